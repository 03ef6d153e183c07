#!/bin/bash
# Diagnostic: build a variant of libwavehip.so whose tetrahedral MFMA kernel records per-wave phase
# timestamps (examples/bin/libwavehip_trace.so); tools/dense_trace.py runs it and prints the timeline.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C="$R/wave_fenics_amd/csrc"
python -c "from wave_fenics_amd import build; build.build()"
mkdir -p "$R/examples/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DWF_DENSE_TRACE -I "$R/include" -I "$C" \
  -c "$C/stiffness_dense.hip" -o "$R/examples/bin/stiffness_dense_trace.o"
OBJS=$(ls "$C"/*.o | grep -v stiffness_dense.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/examples/bin/libwavehip_trace.so" $OBJS "$R/examples/bin/stiffness_dense_trace.o" -ldl
echo "$R/examples/bin/libwavehip_trace.so"

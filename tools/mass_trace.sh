#!/bin/bash
# Diagnostic: variant of libwavehip.so whose dense-mass marching kernel (k_mass_march) records per-wave phase
# timestamps (examples/bin/libwavehip_mstrace.so); tools/mass_trace.py runs it and prints the timeline.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C="$R/wave_fenics_amd/csrc"
python -c "from wave_fenics_amd import build; build.build()"
mkdir -p "$R/examples/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DWF_MASS_TRACE "$@" -I "$R/include" -I "$C" \
  -c "$C/mass_march.hip" -o "$R/examples/bin/mass_march_trace.o"
OBJS=$(ls "$C"/*.o | grep -v mass_march.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/examples/bin/libwavehip_mstrace.so" $OBJS "$R/examples/bin/mass_march_trace.o" -ldl
echo "$R/examples/bin/libwavehip_mstrace.so"

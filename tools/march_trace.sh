#!/bin/bash
# Diagnostic: build a variant of libwavehip.so whose box marching kernel records per-wave phase
# timestamps (examples/bin/libwavehip_mtrace.so); tools/march_trace.py runs it and prints the timeline.
# Extra -D flags (e.g. -DWF_DIAG for the WF_ABLATE masks) can be given as arguments.
# The indexed marching kernel (dense mass, arbitrary-dofmap stiffness) has the same hooks: the script also
# builds examples/bin/libwavehip_itrace.so (-DWF_IDX_TRACE) for tools/idx_trace.py.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C="$R/wave_fenics_amd/csrc"
python -c "from wave_fenics_amd import build; build.build()"
mkdir -p "$R/examples/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DWF_MARCH_TRACE "$@" -I "$R/include" -I "$C" \
  -c "$C/stiffness_march.hip" -o "$R/examples/bin/stiffness_march_trace.o"
OBJS=$(ls "$C"/*.o | grep -v stiffness_march.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/examples/bin/libwavehip_mtrace.so" $OBJS "$R/examples/bin/stiffness_march_trace.o" -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DWF_IDX_TRACE "$@" -I "$R/include" -I "$C" \
  -c "$C/stiffness_march_idx.hip" -o "$R/examples/bin/stiffness_march_idx_trace.o"
OBJS=$(ls "$C"/*.o | grep -v stiffness_march_idx.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/examples/bin/libwavehip_itrace.so" $OBJS "$R/examples/bin/stiffness_march_idx_trace.o" -ldl
echo "$R/examples/bin/libwavehip_mtrace.so"
echo "$R/examples/bin/libwavehip_itrace.so"

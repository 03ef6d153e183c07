#!/bin/bash
# Diagnostic build: libwavehip with the WF_ABLATE run-time flags compiled in (-DWF_DIAG), written
# to examples/bin/libwavehip_diag.so.  Use:  WAVEHIP_LIB=examples/bin/libwavehip_diag.so WF_ABLATE=<mask> python tools/bench_ops.py ...
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C="$R/wave_fenics_amd/csrc"
O="$R/examples/bin/diag_obj"
mkdir -p "$O"
pids=()
for src in tables.cpp mesh_io.cpp generic_plan.cpp function_space.cpp markers.cpp kernels.hip stiffness_march_idx.hip stiffness_march.hip stiffness_march_ks.hip mass_march.hip stiffness_dense.hip tsmm.hip vector_kernels.hip comm.hip cg.hip api.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DWF_DIAG -I "$R/include" -I "$C" -c "$C/$src" -o "$O/${src%.*}.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/examples/bin/libwavehip_diag.so" "$O"/*.o -ldl
echo "$R/examples/bin/libwavehip_diag.so"

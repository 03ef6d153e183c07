#!/bin/bash
# Register / LDS / spill table of every kernel of libwavehip -> profiles/<tag>_resources.txt (runs without a GPU).
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG=${1:-r03}
RAW=$(mktemp)
for f in kernels.hip stiffness_march.hip stiffness_march_idx.hip stiffness_march_ks.hip mass_march.hip stiffness_dense.hip tsmm.hip vector_kernels.hip cg.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage \
    -I "$R/include" -I "$R/wave_fenics_amd/csrc" -c "$R/wave_fenics_amd/csrc/$f" -o /dev/null 2>&1 \
    | grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" \
    | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - - - - - | sed "s/^/$f: /"
done > "$RAW"
python3 "$R/tools/resources_table.py" "$RAW" "$R/profiles/${TAG}_resources.txt"
rm -f "$RAW"

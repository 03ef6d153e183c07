#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   kernel trace + stats, then one PMC pass per counter group (separate passes,
#   /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Output under gpurun_out/prof_<tag>/ ; summarise with tools/summarise_profile.py.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
ARGS=${2:-"--steps 20 --warmup 3 --no-cpu-baseline"}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" $ARGS > "$OUT/kt.log" 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$N" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/pmc_$N.log" 2>&1 || echo "pmc pass $C failed" >> "$OUT/errors.log"
done
find "$OUT" -name "*.csv" | head -50 > "$OUT/files.txt"
echo done

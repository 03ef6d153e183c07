#!/bin/bash
# Collect the measured evidence of a round on the GPU box (run through gpurun):
#   bash tools/evidence.sh r03
# Everything lands under gpurun_out/ev_<tag>/ ; tools/summarise_evidence.py turns it into profiles/<tag>_*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/ev_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
say() { echo "[evidence] $*"; }
prof() {   # prof <name> <program args...>: kernel trace + stats
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$1" -- python3 "${@:2}" > "$OUT/$1.log" 2>&1 || echo "prof $1 failed" >> "$OUT/errors.log"
}
pmc() {    # pmc <name> "<counters>" <program args...>
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$OUT/$1" -- python3 "${@:3}" > "$OUT/$1.log" 2>&1 || echo "pmc $1 failed" >> "$OUT/errors.log"
}
say "bench line (with cpu baseline)"
python3 "$R/bench.py" > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
python3 "$R/bench.py" --generic --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_generic.json"
python3 "$R/bench.py" --no-cpu-baseline --periodic x 2>/dev/null | tail -1 > "$OUT/bench_periodic_x.json"
python3 "$R/bench.py" --no-cpu-baseline --periodic xyz 2>/dev/null | tail -1 > "$OUT/bench_periodic_xyz.json"
python3 "$R/bench.py" --no-cpu-baseline --generic --periodic xyz 2>/dev/null | tail -1 > "$OUT/bench_generic_periodic_xyz.json"
say "kernel trace of bench.py (box and generic)"
prof kt_box "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline
prof kt_generic "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --generic
say "PMC passes (box)"
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  pmc pmc_box_$N "$C" "$R/bench.py" --steps 5 --warmup 2 --settle-steps 0 --no-cpu-baseline
done
say "PMC passes (generic)"
for C in FETCH_SIZE WRITE_SIZE; do
  pmc pmc_generic_$C "$C" "$R/bench.py" --steps 5 --warmup 2 --settle-steps 0 --no-cpu-baseline --generic
done
say "all operators"
export DEGREES=2,3,4,5,6,7
python3 "$R/tools/bench_ops.py" stiffness 2>/dev/null > "$OUT/ops.jsonl"
export DEGREES=2,4,6
python3 "$R/tools/bench_ops.py" mass dense vector rk4 2>/dev/null >> "$OUT/ops.jsonl"
python3 "$R/tools/bench_ops.py" tsmm tet 2>/dev/null >> "$OUT/ops.jsonl"
python3 "$R/tools/bench_shapes.py" 2>/dev/null > "$OUT/shapes.jsonl"
prof kt_ops "$R/tools/bench_ops.py" stiffness mass dense
prof kt_mfma "$R/tools/bench_ops.py" tsmm tet
say "PMC passes for the other kernels quoted in DESIGN.md (P6 stiffness, dense mass, RK4 stage, tetrahedra)"
export DEGREES=6 SETTLE=0    # counters are per launch: no need to hold the load through the power ramp
for C in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  pmc pmc_ops6_$N "$C" "$R/tools/bench_ops.py" stiffness dense
  pmc pmc_rk4_$N "$C" "$R/tools/bench_rk4.py" --steps 3 --warmup 1
  pmc pmc_tet_$N "$C" "$R/tools/bench_ops.py" tet
done
export DEGREES=2,4,6
say "MFMA pipe counters"
pmc pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "$R/tools/bench_ops.py" tsmm tet
unset SETTLE
say "RK4 loop"
python3 "$R/tools/bench_rk4.py" 2>/dev/null | tail -1 > "$OUT/rk4.jsonl"
python3 "$R/tools/bench_rk4.py" --unfused 2>/dev/null | tail -1 >> "$OUT/rk4.jsonl"
python3 "$R/tools/bench_rk4.py" --periodic x 2>/dev/null | tail -1 >> "$OUT/rk4.jsonl"
python3 "$R/tools/bench_rk4.py" --periodic xyz 2>/dev/null | tail -1 >> "$OUT/rk4.jsonl"
prof kt_rk4 "$R/tools/bench_rk4.py" --steps 50
prof kt_rk4_periodic "$R/tools/bench_rk4.py" --steps 50 --periodic xyz
say "launch durations from idle (power ramp)"
for a in "4 400 0" "6 400 0" "6 400 100" "7 400 0"; do python3 "$R/tools/drift.py" $a 2>/dev/null; done > "$OUT/power_ramp.log"
say "generic kernel under hostile orderings"
python3 "$R/tools/bench_generic_orderings.py" 2>/dev/null > "$OUT/generic_orderings.log"
say "f64 MFMA rate, HBM read rate, TSMM and tet timelines (diagnostic binaries built beforehand into examples/bin)"
[ -x "$R/examples/bin/mfma_rate" ] && "$R/examples/bin/mfma_rate" > "$OUT/mfma_f64_rate.log"
[ -x "$R/examples/bin/hbm_read_rate" ] && "$R/examples/bin/hbm_read_rate" > "$OUT/hbm_read_rate.log"
if [ -f "$R/examples/bin/libwavehip_mstrace.so" ]; then
  for p in 2 4 6; do P=$p python3 "$R/tools/mass_trace.py" 2>/dev/null | grep -v amdgpu.ids; done > "$OUT/mass_trace.log"
fi
if [ -x "$R/examples/bin/tsmm_trace" ]; then
  { echo "== 100000 x 125, layout 0"; "$R/examples/bin/tsmm_trace" 100000 125 0; echo "== 100000 x 125, layout 1"; "$R/examples/bin/tsmm_trace" 100000 125 1;
    echo "== 1000000 x 125, layout 0"; "$R/examples/bin/tsmm_trace" 1000000 125 0; } > "$OUT/tsmm_trace.log" 2>&1
fi
[ -f "$R/examples/bin/libwavehip_trace.so" ] && python3 "$R/tools/dense_trace.py" 2>/dev/null > "$OUT/dense_trace.log"
find "$OUT" -name "*.csv" | wc -l
say done

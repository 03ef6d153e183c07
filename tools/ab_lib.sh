#!/bin/bash
# A/B of library variants on the bench line: bash tools/ab_lib.sh "<bench.py args>" lib1.so lib2.so ...  (three alternating rounds)
ARGS=$1; shift
for r in 1 2 3; do
  for L in "$@"; do
    WAVEHIP_LIB=$L python3 bench.py --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$L'.split('/')[-1], 'step %.4f ms  kernel %.4f ms  frac %.3f  from idle %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], (d.get('from_idle') or {}).get('kernel_ms', 0)))"
  done
done

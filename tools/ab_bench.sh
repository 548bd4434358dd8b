#!/bin/bash
# Same-box A/B of library builds on the bench command's timed loop:  bash tools/ab_bench.sh WORKLOAD STEPS lib1.so lib2.so ...  [ROUNDS=2]
WL=$1; STEPS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    BFSM_LIB=$R/$lib timeout -k 10 400 python3 $R/bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline --no-exact --no-extras --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$WL', '$(basename $lib)', 'value', round(d['value'],4), 'ms', round(d['ms_per_step'],4), 'frac', round(d['frac_of_hbm_peak'],4), 'repeats', round(d['repeats']['min'],4), round(d['repeats']['median'],4), round(d['repeats']['max'],4), 'kernels', {k: round(v['ms_per_eval'],3) for k,v in r['per_kernel'].items()}, flush=True)"
  done
done

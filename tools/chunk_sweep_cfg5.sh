#!/bin/bash
# bench.py wall-clock rate of config 5 for several --max-chunk settings (directions resident at once; 48 MiB of scratch each
# with the interleaved layout + P'):  bash tools/chunk_sweep_cfg5.sh [sizes...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for mc in ${@:-576 720 960 1152 1440 1920 2880}; do
  python3 $R/bench.py --workload cfg5 --max-chunk $mc --steps 3 --warmup 1 --no-extras --no-exact --no-cpu-baseline --no-roofline --repeats 2 2>/dev/null |
    python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('cfg5 max_chunk', sys.argv[1], 'value', round(d['value'], 4), 'frac', round(d['frac_of_hbm_peak'], 4), 'repeats', round(d['repeats']['min'], 4), round(d['repeats']['max'], 4), flush=True)" $mc
done

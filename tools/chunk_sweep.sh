#!/bin/bash
# bench.py wall-clock rate (value, median of the repeats) for several --max-chunk settings at cfg3 and cfg4, twice each.
for i in 1 2; do
for mc in 0 384 256 192; do python3 bench.py --workload cfg3 --steps 40 --warmup 5 --no-cpu --no-exact --no-extras --repeats 3 --max-chunk $mc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('cfg3 max_chunk', sys.argv[1], round(d['value'],2), round(d['repeats']['median'],2))" $mc; done
for mc in 1024 640 512 416; do python3 bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu --no-exact --no-extras --repeats 3 --max-chunk $mc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('cfg4 max_chunk', sys.argv[1], round(d['value'],2), round(d['repeats']['median'],2))" $mc; done
done

#!/bin/bash
# Regenerates the judged profile summaries of a round on the GPU box:  bash tools/profile_round.sh TAG [WORKLOAD] [extra bench args]
#   1. rocprofv3 --kernel-trace --stats of the bench command (default workload cfg3 = the default bench command)
#   2./3. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (no tracing domains besides the kernel trace)
#   4. profiles/summarize.py -> profiles/TAG_WL_kernel_stats.csv, TAG_WL_pmc_traffic.json (+ traffic_latest.json for cfg3);
#      tools/launch_gaps.py -> TAG_WL_launch_gaps.txt (per-dispatch durations split by the idle gap in front)
#   5. the bench line itself (reads the fresh traffic_latest.json) -> profiles/TAG_bench_WL.json
# Everything is written under gpurun_out/ (merged back by gpurun); copy gpurun_out/profiles_TAG/* into profiles/.
set -e
TAG=${1:?tag}
WL=${2:-cfg3}
shift; shift || true
EXTRA="$@"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
STEPS=20; [ "$WL" = cfg5 ] && STEPS=5; [ "$WL" = cfg4 ] && STEPS=10
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline --no-exact --no-extras $EXTRA > $O/prof_stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-roofline --no-extras $EXTRA > $O/prof_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-roofline --no-extras $EXTRA > $O/prof_write.log 2>&1
python3 profiles/summarize.py ${TAG}_$WL $O/prof_stats $O/prof_fetch $O/prof_write "tag $TAG: python3 bench.py --workload $WL $EXTRA" $WL
timeout -k 10 600 python3 bench.py --workload $WL $EXTRA > $O/bench_${TAG}_$WL.json 2> $O/bench_${TAG}_$WL.err
mkdir -p $O/profiles_$TAG
python3 tools/launch_gaps.py $O/prof_stats > $O/profiles_$TAG/${TAG}_${WL}_launch_gaps.txt 2>&1 || true
cp profiles/${TAG}_${WL}_kernel_stats.csv profiles/${TAG}_${WL}_pmc_traffic.json $O/profiles_$TAG/
[ "$WL" = cfg3 ] && cp profiles/traffic_latest.json $O/profiles_$TAG/
cp $O/bench_${TAG}_$WL.json $O/profiles_$TAG/${TAG}_bench_$WL.json
head -60 profiles/${TAG}_${WL}_pmc_traffic.json
tail -c 900 $O/bench_${TAG}_$WL.json

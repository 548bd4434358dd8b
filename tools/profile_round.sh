#!/bin/bash
# Regenerates the judged profile summaries of a round on the GPU box:  bash tools/profile_round.sh TAG
#   1. rocprofv3 --kernel-trace --stats of the default bench command (cfg3)
#   2./3. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (no tracing domains besides the kernel trace)
#   4. profiles/summarize.py -> profiles/TAG_cfg3_kernel_stats.csv, TAG_cfg3_pmc_traffic.json, traffic_latest.json
#   5. the bench line itself (reads the fresh traffic_latest.json) -> profiles/TAG_bench_cfg3.json
# Everything is written under gpurun_out/ (merged back by gpurun); copy gpurun_out/profiles_TAG/* into profiles/.
set -e
TAG=${1:?tag}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact > $O/prof_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-roofline > $O/prof_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-roofline > $O/prof_write.log 2>&1
python3 profiles/summarize.py ${TAG}_cfg3 $O/prof_stats $O/prof_fetch $O/prof_write "round-1 build, tag $TAG: default bench command (cfg3, fp64, one chunk of 768 directions)"
timeout -k 10 300 python3 bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
mkdir -p $O/profiles_$TAG
cp profiles/${TAG}_cfg3_kernel_stats.csv profiles/${TAG}_cfg3_pmc_traffic.json profiles/traffic_latest.json $O/profiles_$TAG/
cp $O/bench_$TAG.json $O/profiles_$TAG/${TAG}_bench_cfg3.json
cat profiles/${TAG}_cfg3_pmc_traffic.json | head -40
tail -c 600 $O/bench_$TAG.json

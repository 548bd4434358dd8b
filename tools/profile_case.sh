#!/bin/bash
# Kernel-trace statistics + HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) + SQ counters of one tools/ktimes.py
# case (any workload / mode, e.g. cfg3:h, cfg5, cfg1):  bash tools/profile_case.sh TAG CASE
# Output: gpurun_out/profiles_TAG/TAG_<case>_{kernel_stats.csv,pmc_traffic.json,sq_counters.txt}; copy into profiles/.
set -e
TAG=${1:?tag}; CASE=${2:?case}
NAME=$(echo $CASE | tr ':,' '__')
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; P=$O/profiles_$TAG
cd $R; mkdir -p $P
rm -rf $O/pc_stats $O/pc_fetch $O/pc_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc_stats -- python3 tools/ktimes.py $CASE > $O/pc_stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pc_fetch -- python3 tools/ktimes.py $CASE > $O/pc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pc_write -- python3 tools/ktimes.py $CASE > $O/pc_write.log 2>&1
python3 - $TAG $NAME "$CASE" <<'PY'
import collections, csv, glob, json, os, re, shutil, sys
tag, name, case = sys.argv[1:4]
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"; P = f"{O}/profiles_{tag}"
stats = glob.glob(f"{O}/pc_stats/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"{P}/{tag}_{name}_kernel_stats.csv")
def short(n):
    m = re.search(r"bfsm_(\w*)kernel<\(bfsm::(\w+)\)(\d+)(?:, (\d+))?, (float|double)", n)
    return f"{m.group(1)}{m.group(2)}{m.group(3)}_N{m.group(4)}_{m.group(5)}" if m else None
out = {}
for r in csv.DictReader(open(stats)):
    k = short(r["Name"])
    if k: out.setdefault(k, {}).update(avg_launch_us=float(r["AverageNs"]) / 1e3, calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6)
for cname, d in (("FETCH_SIZE", "pc_fetch"), ("WRITE_SIZE", "pc_write")):
    f = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k: agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items(): out.setdefault(k, {})[cname + "_KiB_per_launch"] = v / n
for k, d in out.items():
    d["hbm_bytes_per_launch"] = (2 * d.get("FETCH_SIZE_KiB_per_launch", 0) + d.get("WRITE_SIZE_KiB_per_launch", 0)) * 1024
    if d.get("avg_launch_us"): d["hbm_GBps"] = d["hbm_bytes_per_launch"] / (d["avg_launch_us"] * 1e-6) / 1e9
out["_note"] = f"tools/ktimes.py {case}; kernel key = <kind enum><index>_N<size>_<type> (K: 5 gain_inv, 6 gain_line, 7 gain_fwd, 8 reduce, 9/10 tail, 11 gain_line_acc, 12 nyq_rows, 13 gain_line_acc_h; SK: 0 small_gain, 1 small_reduce; GK: generic path); HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md"
json.dump(out, open(f"{P}/{tag}_{name}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
for k in sorted(out):
    if isinstance(out[k], dict) and out[k].get("total_ms", 0) > 0.05 * max(v.get("total_ms", 0) for v in out.values() if isinstance(v, dict)):
        print(k, {a: round(b, 1) for a, b in out[k].items()})
PY
bash tools/pmc_sq.sh ${TAG}_$NAME $CASE > $P/${TAG}_${NAME}_sq_counters.txt 2>&1 || true
tail -12 $P/${TAG}_${NAME}_sq_counters.txt

#!/bin/bash
# SQ-side counters of the gain kernels (two passes of 8 SQ counters; own runs, no tracing domains besides kernel-trace)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-exact $BFSM_PMC_ARGS > $O/pmc_sq1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-exact $BFSM_PMC_ARGS > $O/pmc_sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
KIND = {"5": "KA", "6": "KB", "7": "KC"}
for d in ("pmc_sq1", "pmc_sq2"):
    fs = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"\(bfsm::K\)(\d+)", r["Kernel_Name"])
        if not m or m.group(1) not in KIND: continue
        a = agg[KIND[m.group(1)]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    for k in sorted(agg):
        print(d, k, {c: round(v[1] / v[0]) for c, v in sorted(agg[k].items())})
PY

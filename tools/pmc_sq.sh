#!/bin/bash
# SQ-side counters of the gain kernels (two passes of 8 SQ counters; own runs, no tracing domains besides kernel-trace)
#   bash tools/pmc_sq.sh TAG CASE...     (CASE as in tools/ktimes.py; output gpurun_out/pmc_TAG_{sq1,sq2}, summary on stdout
#   and in gpurun_out/pmc_TAG_summary.txt)
export TMPDIR=/tmp
TAG=${1:?tag}; shift
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rm -rf $O/pmc_${TAG}_sq1 $O/pmc_${TAG}_sq2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_${TAG}_sq1 -- python3 tools/ktimes.py "$@" > $O/pmc_${TAG}_sq1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_${TAG}_sq2 -- python3 tools/ktimes.py "$@" > $O/pmc_${TAG}_sq2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/pmc_${TAG}_sq3 -- python3 tools/ktimes.py "$@" > $O/pmc_${TAG}_sq3.log 2>&1
python3 - $TAG <<'PY' | tee $O/pmc_${TAG}_summary.txt
import csv, glob, collections, re, os, sys
tag = sys.argv[1]
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
KIND = {"5": "KA", "6": "KB", "7": "KC", "11": "KBacc", "13": "KBaccH", "14": "KAnyq"}
for d in ("sq1", "sq2", "sq3"):
    fs = glob.glob(f"{O}/pmc_{tag}_{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"\(bfsm::K\)(\d+), (\d+), (float|double)", r["Kernel_Name"])
        if not m or m.group(1) not in KIND: continue
        key = f"{KIND[m.group(1)]}_N{m.group(2)}_{m.group(3)}"
        a = agg[key][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    for k in sorted(agg):
        print(d, k, {c: round(v[1] / v[0]) for c, v in sorted(agg[k].items())})
PY

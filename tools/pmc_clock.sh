#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (MI355X_MICROARCH.md, DVFS give-back)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
rm -rf $O/pmc_clk
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_clk -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-exact > $O/pmc_clk.log 2>&1
python3 - <<'PY'
import csv, glob, collections, re, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
KIND = {"5": "KA", "6": "KB", "7": "KC"}
f = glob.glob(f"{O}/pmc_clk/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(f)):
    m = re.search(r"\(bfsm::K\)(\d+)", r["Kernel_Name"])
    if not m or m.group(1) not in KIND: continue
    a = agg[KIND[m.group(1)]]
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, (n, c, ns) in sorted(agg.items()):
    print(k, "launches", n, "avg_us", round(ns / n / 1e3, 1), "GUI_ACTIVE/8/time = %.2f GHz" % (c / 8 / ns))
PY

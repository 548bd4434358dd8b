# rocprofv3 kernel trace of a tools/*.py script; prints the per-dispatch durations of one kernel kind in launch order
# usage (on the GPU box): bash tools/prof_exact.sh tools/profile_modes.py "64 16 48" 5
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SCRIPT=${1:-tools/profile_modes.py}; ARGS=${2:-"64 16 48"}; KIND=${3:-5}
rm -rf gpurun_out/prof_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 $SCRIPT $ARGS > gpurun_out/prof_trace.log 2>&1
KIND=$KIND python3 - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/prof_trace/**/*kernel_trace.csv", recursive=True)[0]
kind = os.environ["KIND"]
rows = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"])
        for r in csv.DictReader(open(f)) if "(bfsm::K)%s," % kind in r["Kernel_Name"]]
rows.sort()
print("kind", kind, "durations (us), grid:", [(round(d, 1), gx, gy) for _, d, gx, gy in rows])
PY

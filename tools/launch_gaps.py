#!/usr/bin/env python3
"""Per-dispatch durations of the gain kernels from a rocprofv3 kernel trace, split by what preceded the dispatch.

usage: launch_gaps.py <dir with *kernel_trace.csv> [--transitions]
  --transitions : additionally the idle time between consecutive dispatches (end of one -> start of the next), per pair
                  of kernels, for gaps below 100 us (i.e. inside and between queued evaluations)
Answers "why does KA have a slow tail in the --stats summary of a bench session".  Three populations per kernel:
  after idle : the evaluation it belongs to started within 25 ms after the GPU had been idle for >= 10 ms (a new handle
               being created, the CPU baseline running, process start): the first ~6 evaluations after such a pause run
               their first heavy kernel up to 35 % slower, decaying back over ~20 ms;
  after gap  : steady state, but the evaluation started >= 20 us after the previous kernel ended (a host synchronisation
               between evaluations: the blocking-call loop, the boundary of a profiled burst);
  steady     : queued directly behind the previous evaluation (what `value` times).
"""
import csv
import glob
import re
import statistics
import sys

KIND = {"5": "KA gain_inv", "6": "KB gain_line", "7": "KC gain_fwd", "11": "KB' acc", "13": "KB' acc_h", "14": "KA + KN"}
d = sys.argv[1]
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)      # the newest trace in the directory
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
out = {}
prev_end = None
last_idle_end = None     # time at which the last long idle period ended
cls = "after idle"
for s, e, name in rows:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 1e12      # us
    if gap >= 10e3:
        last_idle_end = s
    prev_end = max(prev_end or 0, e)
    m = re.search(r"\(bfsm::K\)(\d+),", name)
    if not m:
        continue
    if m.group(1) == "0":                      # F1a opens an evaluation: classify the whole evaluation here
        if last_idle_end is not None and (s - last_idle_end) / 1e6 <= 25.0:
            cls = "after idle"
        elif gap >= 20.0:
            cls = "after gap"
        else:
            cls = "steady"
    if m.group(1) in KIND:
        out.setdefault(KIND[m.group(1)], {}).setdefault(cls, []).append((e - s) / 1e3)
print(f"kernel trace: {f.split('/')[-1]}")
print("dispatch durations in us")
for k in sorted(out):
    for c in ("steady", "after gap", "after idle"):
        v = out[k].get(c)
        if not v:
            continue
        sd = statistics.pstdev(v) if len(v) > 1 else 0.0
        print(f"{k:14s} {c:10s} n={len(v):4d} mean={statistics.mean(v):9.1f} min={min(v):9.1f} max={max(v):9.1f} sd={sd:7.1f}")

if "--transitions" in sys.argv:
    NAMES = {"0": "F1a", "1": "F1b", "5": "KA", "6": "KB", "7": "KC", "8": "reduce", "9": "tail_inv", "10": "tail_line",
             "11": "KB'", "12": "KN", "13": "KB'H", "14": "KA+KN"}

    def short(name):
        m = re.search(r"\(bfsm::(S?K)\)(\d+),", name)
        if not m:
            return name[:24]
        return ("S" + m.group(2)) if m.group(1) == "SK" else NAMES.get(m.group(2), "K" + m.group(2))
    tr = {}
    pe, pn = None, None
    for s_, e_, name in rows:
        if pe is not None and 0 <= (s_ - pe) < 100e3:
            tr.setdefault((short(pn), short(name)), []).append((s_ - pe) / 1e3)
        pe, pn = e_, name
    print("idle time between consecutive dispatches in us (pairs with >= 20 samples)")
    for k in sorted(tr, key=lambda k: -len(tr[k])):
        v = tr[k]
        if len(v) >= 20:
            print(f"{k[0]:>10s} -> {k[1]:<10s} n={len(v):5d} mean={statistics.mean(v):7.2f} median={statistics.median(v):7.2f} max={max(v):7.1f}")

#!/usr/bin/env python3
"""Single-evaluation rate of the small grids: evaluations queued back to back on one stream, and with a host
synchronisation after each.  usage: latency.py [cfg1 cfg2 ...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd"))
sys.path.insert(0, ROOT)
import torch
import bfsm
from bench import WORKLOADS

c = bfsm.reference_constants()
for name in (sys.argv[1:] or ["cfg1", "cfg2"]):
    w = WORKLOADS[name]
    nv = w["nv"]
    f = torch.from_numpy(bfsm.bkw_solution(nv)[0]).cuda()
    Q = torch.empty_like(f)
    for mode, ex in (("faithful", False), ("exact+hermitian", True)):
        for small in ((True, False) if nv == 16 else (True,)):
            op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(w["n_gl"], 0, c["R"]), bfsm.SphericalDesign(w["n_sph"]),
                                           nv, nv, nv, c["gamma"], c["b_gamma"], c["L"])
            op.setPrecision(w["precision"])
            op.setExactReductions(ex, hermitian=ex)
            op.setSmallPath(small)
            op.initialize()
            s = torch.cuda.current_stream().cuda_stream
            n = 300 if nv == 16 else 100
            for _ in range(200):
                op.computeCollisionAsync(Q, f, s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                op.computeCollisionAsync(Q, f, s)
            ti = (time.perf_counter() - t0) / n      # host time to ISSUE one evaluation (launches + capture query + event record,
            torch.cuda.synchronize()                 # through ctypes); the queue fills after ~1000 launches, so n is kept small enough
            tq = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                op.computeCollision(Q, f)
            tb = (time.perf_counter() - t0) / n
            print(f"{name} {mode:16s} small_path={small!s:5s} queued {1 / tq:9.0f} evals/s ({tq * 1e6:6.1f} us)   blocking {1 / tb:9.0f} evals/s ({tb * 1e6:6.1f} us)   host issue {ti * 1e6:5.1f} us per call", flush=True)
            op.destroy()

#!/usr/bin/env python3
"""Would running the two halves of an evaluation's directions on two streams at once help?  Probe, not product: two
handles, each owning half of the directions (like two ranks on ONE device), driven on two streams concurrently, against
one handle that owns all of them.  usage: overlap_probe.py [cfg2 cfg3 ...]"""
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


def make(w, shard=None):
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(w["n_gl"], 0, c["R"]), bfsm.SphericalDesign(w["n_sph"]),
                                   w["nv"], w["nv"], w["nv"], c["gamma"], c["b_gamma"], c["L"])
    op.setPrecision(w["precision"])
    if shard:
        op.setDirectionShard(*shard)
    op.initialize()
    return op


for name in (sys.argv[1:] or ["cfg2", "cfg3"]):
    w = WORKLOADS[name]
    B = w["n_gl"] * w["n_sph"]
    f = torch.from_numpy(bfsm.bkw_solution(w["nv"])[0]).cuda()
    Q, QA, QB = torch.empty_like(f), torch.empty_like(f), torch.empty_like(f)
    full = make(w)
    n = 200 if w["nv"] <= 32 else 40
    s0 = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        full.computeCollisionAsync(Q, f, s0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        full.computeCollisionAsync(Q, f, s0)
    torch.cuda.synchronize()
    t_full = (time.perf_counter() - t0) / n
    full.destroy()
    for parts in (2, 3, 4):
        ops = [make(w, bfsm.shard_range(B, r, parts)) for r in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        Qs = [torch.empty_like(f) for _ in range(parts)]
        main = torch.cuda.current_stream()
        def once():        # fork from / join into the main stream: evaluation i + 1 starts after ALL of evaluation i
            start = torch.cuda.Event()
            start.record(main)
            for r in range(parts):
                streams[r].wait_event(start)
                ops[r].collidePartial(Qs[r], f, r == 0, streams[r].cuda_stream)
                done = torch.cuda.Event()
                done.record(streams[r])
                main.wait_event(done)
        for _ in range(20):
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            once()
        torch.cuda.synchronize()
        t_par = (time.perf_counter() - t0) / n
        # the same shards one after the other on ONE stream (what the redundant fixed parts cost without any overlap)
        def serial():
            for r in range(parts):
                ops[r].collidePartial(Qs[r], f, r == 0, s0)
        for _ in range(5):
            serial()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            serial()
        torch.cuda.synchronize()
        t_ser = (time.perf_counter() - t0) / n
        print(f"{name}: one handle {t_full * 1e3:.4f} ms | {parts} shards on {parts} streams {t_par * 1e3:.4f} ms ({t_full / t_par:.3f} x) | "
              f"the same shards on one stream {t_ser * 1e3:.4f} ms", flush=True)
        for o in ops:
            o.destroy()

#!/usr/bin/env python3
"""Per-kernel HIP-event times (ms per evaluation, algorithmic TB/s) for a list of cases; BFSM_LIB selects the build.

usage: ktimes.py CASE [CASE ...]     CASE = name[:mode] | nv,n_gl,n_sph,prec[,mode] | NXxNYxNZ,n_gl,n_sph,prec   (mode: f faithful, e exact, h hermitian)
named: cfg1 cfg2 cfg3 cfg4 cfg5 c5s (cfg5 grid, 4 radial nodes) d128 (N=128 fp64, 2 radial nodes x 192) f64 (cfg3 in fp32)
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd"))
import torch
import bfsm

NAMED = {"cfg1": (16, 8, 32, 64), "cfg2": (32, 8, 48, 64), "cfg3": (64, 16, 48, 64), "cfg4": (64, 16, 156, 64),
         "cfg5": (128, 30, 192, 32), "c5s": (128, 4, 192, 32), "d128": (128, 2, 192, 64), "f64": (64, 16, 48, 32)}
c = bfsm.reference_constants()
for case in sys.argv[1:]:
    mode = "f"
    shape = None
    if case.split(":")[0] in NAMED:
        nv, n_gl, n_sph, prec = NAMED[case.split(":")[0]]
        if ":" in case:
            mode = case.split(":")[1]
    else:
        parts = case.split(",")
        shape = tuple(int(a) for a in parts[0].split("x")) if "x" in parts[0] else None      # NXxNYxNZ: a box (size-generic path)
        nv = shape[0] if shape else int(parts[0])
        n_gl, n_sph, prec = (int(a) for a in parts[1:4])
        mode = parts[4] if len(parts) > 4 else "f"
    shape = shape if (case.split(":")[0] not in NAMED and shape) else (nv, nv, nv)
    if shape[0] == shape[1] == shape[2]:
        f = torch.from_numpy(bfsm.bkw_solution(nv)[0]).cuda()
    else:
        import numpy as np
        f = torch.from_numpy(np.random.default_rng(1).random(shape) + 0.1).cuda()
    Q = torch.empty_like(f)
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0, c["R"]), bfsm.SphericalDesign(n_sph), shape[0], shape[1], shape[2],
                                   c["gamma"], c["b_gamma"], c["L"])
    op.setPrecision(prec)
    op.setProfiling(True)
    op.setExactReductions(mode in "eh", hermitian=(mode == "h"))
    if os.environ.get("BFSM_MAX_CHUNK"):          # experiments: directions resident at once
        op.setMaxChunk(int(os.environ["BFSM_MAX_CHUNK"]))
    op.initialize()
    t0 = time.time()
    while time.time() - t0 < 0.5:
        op(Q, f)
    reps, acc, accb = 5, None, None
    for _ in range(reps):
        op(Q, f)
        cn = op.counters()
        cur = [cn.kernel_ms[i] for i in range(len(bfsm.KERNEL_NAMES))]
        acc = cur if acc is None else [a + b for a, b in zip(acc, cur)]
        accb = [cn.kernel_alg_bytes[i] for i in range(len(bfsm.KERNEL_NAMES))]
    tot = sum(acc) / reps
    cb = 16.0 if prec == 64 else 8.0
    alg = (6.0 * n_gl * n_sph + 9) * shape[0] * shape[1] * shape[2] * cb
    s = " ".join(f"{k}={v / reps:.3f}ms/{(b / (v / reps * 1e-3) / 1e12) if v > 0 else 0:.2f}" for k, v, b in zip(bfsm.KERNEL_NAMES, acc, accb))
    print(f"{case:10s} {os.path.basename(os.environ.get('BFSM_LIB', 'default')):28s} sum={tot:.3f}ms alg={alg / tot / 1e9:.2f}TB/s of8={alg / tot / 8e9:.3f} | {s}", flush=True)
    op.destroy()

"""Evaluations per second of bfsm_collide_batch for a BASELINE workload at several batch sizes.
Usage: python tools/batch_rate.py cfg1 1 8 64"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "boltzmann-fourier-spectral-method_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bfsm
from bench import WORKLOADS

name = sys.argv[1]
w = WORKLOADS[name]
nv, n_gl, n_sph, prec = w["nv"], w["n_gl"], w["n_sph"], w["precision"]
c = bfsm.reference_constants()
f1 = torch.from_numpy(bfsm.bkw_solution(nv)[0]).cuda()
B = n_gl * n_sph
for nb in [int(x) for x in sys.argv[2:]]:
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph),
                                   nv, nv, nv, c["gamma"], c["b_gamma"], c["L"])
    op.setPrecision(prec)
    op.setMaxBatch(nb)
    op.initialize()
    f = f1.reshape(1, -1).repeat(nb, 1).contiguous()
    Q = torch.empty_like(f)
    s = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        op.computeCollisionBatch(Q, f, nb, s)
        torch.cuda.synchronize()
    reps = max(3, int(2000 / nb))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        op.computeCollisionBatch(Q, f, nb, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    alg = (6.0 * B + 9.0) * nv ** 3 * (16.0 if prec == 64 else 8.0) * nb
    print(f"{name} batch {nb}: {nb / dt:.0f} evals/s, {alg / dt / 1e12:.2f} TB/s algorithmic", flush=True)
    op.destroy()

#!/usr/bin/env python3
"""Per-kernel HIP-event times of one workload in the three modes (faithful / exact / exact + hermitian)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd"))
import torch
import bfsm

nv, n_gl, n_sph = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 16, 48)))
prec = int(sys.argv[4]) if len(sys.argv) > 4 else 64
c = bfsm.reference_constants()
f = torch.from_numpy(bfsm.bkw_solution(nv)[0]).cuda()
Q = torch.empty_like(f)
for name, ex, he in (("faithful", False, False), ("exact", True, False), ("exact+hermitian", True, True)):
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0, c["R"]), bfsm.SphericalDesign(n_sph), nv, nv, nv,
                                   c["gamma"], c["b_gamma"], c["L"])
    op.setPrecision(prec)
    op.setProfiling(True)
    op.setExactReductions(ex, hermitian=he)
    op.initialize()
    for _ in range(3):
        op(Q, f)
    acc = None
    for _ in range(5):
        op(Q, f)
        cn = op.counters()
        cur = [cn.kernel_ms[i] for i in range(len(bfsm.KERNEL_NAMES))]
        acc = cur if acc is None else [a + b for a, b in zip(acc, cur)]
    print(name, {k: round(v / 5, 3) for k, v in zip(bfsm.KERNEL_NAMES, acc)}, "sum %.3f ms" % (sum(acc) / 5))
    op.destroy()

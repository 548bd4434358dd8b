"""Per-rank cost of a 1/P direction shard on ONE GPU (no collective): what strong scaling can at best look like.

For P in 1,2,4,8 the operator is created on rank 0's shard of P and gain_partial + finish_partial are timed back to
back (the work a rank does between two all-reduces).  speed-up bound = t(P=1) / t(P); the fixed part (F1, reduce,
tail, launch gaps) is what keeps it below P.  Usage: python tools/shard_cost.py [cfg3|cfg4|cfg5]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "boltzmann-fourier-spectral-method_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bfsm
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w = WORKLOADS[name]
nv, n_gl, n_sph, prec = w["nv"], w["n_gl"], w["n_sph"], w["precision"]
B = n_gl * n_sph
c = bfsm.reference_constants()
f = torch.from_numpy(bfsm.bkw_solution(nv)[0]).cuda()
Q = torch.empty_like(f)
s = torch.cuda.current_stream().cuda_stream
base = None
for P in (1, 2, 4, 8):
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph),
                                   nv, nv, nv, c["gamma"], c["b_gamma"], c["L"])
    op.setPrecision(prec)
    op.setDirectionShard(*bfsm.shard_range(B, min(int(os.environ.get('BFSM_SHARD_COST_RANK', '0')), P - 1), P))
    op.initialize()

    fused = os.environ.get("BFSM_SHARD_COST_TWO_CALLS") is None

    def step():
        if fused:
            op.collidePartial(Q, f, True, s)       # what bfsm.sharded_step issues per evaluation
        else:
            op.gainPartial(f, s)
            op.finishPartial(Q, f, True, s)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        step()
        torch.cuda.synchronize()
    steps = 200
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    # the same, synchronising every step (what a caller that waits on the collective each step sees)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        torch.cuda.synchronize()
    ms_sync = 1e3 * (time.perf_counter() - t0) / steps
    base = base or ms
    print(f"{name} P={P}: {ms:.4f} ms/step pipelined ({base / ms:.2f}x of P=1), {ms_sync:.4f} ms/step with a sync per step", flush=True)
    op.destroy()

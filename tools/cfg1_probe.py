"""Config 1 (N=16): the whole-direction gain kernel with and without the loss term riding on workgroup 0, at a few
direction counts around one-workgroup-per-CU (256).  GPU only: gpurun -- python3 tools/cfg1_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd")); sys.path.insert(0, ROOT)
import torch, bfsm
c = bfsm.reference_constants()
f = torch.from_numpy(bfsm.bkw_solution(16)[0]).cuda(); Q = torch.empty_like(f)
for n_gl, n_sph, label in ((8, 32, "cfg1 B=256"), (8, 12, "B=96"), (4, 32, "B=128"), (16, 32, "B=512"), (7, 32, "B=224")):
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0, c["R"]), bfsm.SphericalDesign(n_sph), 16, 16, 16, c["gamma"], c["b_gamma"], c["L"])
    op.setProfiling(True); op.initialize()
    s = torch.cuda.current_stream().cuda_stream
    for wl in (True, False):
        for _ in range(200): op.collidePartial(Q, f, wl, s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000): op.collidePartial(Q, f, wl, s)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 2000
        cn = op.counters()
        print(f"{label:12s} with_loss={wl!s:5s} {t*1e6:6.2f} us/eval  kernels: " + " ".join(f"{k}={v*1e3:.1f}us" for k, v in zip(bfsm.KERNEL_NAMES, cn.kernel_ms) if v > 0), flush=True)
    op.destroy()

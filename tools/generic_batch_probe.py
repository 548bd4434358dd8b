"""Size-generic path: a batch of distributions through one set of launches against the same distributions one call at a
time (evaluations per second, stream-ordered loop).  GPU only: gpurun -- python3 tools/generic_batch_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, bfsm
c = bfsm.reference_constants()
for shape, n_gl, n_sph in (((20, 20, 20), 8, 48), ((32, 64, 16), 8, 48), ((48, 32, 24), 8, 48), ((12, 10, 8), 4, 12)):
    G = shape[0] * shape[1] * shape[2]
    for nb in (1, 8):
        op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0, c["R"]), bfsm.SphericalDesign(n_sph), *shape, c["gamma"], c["b_gamma"], c["L"])
        op.setMaxBatch(nb); op.initialize()
        f = torch.rand(nb, G, dtype=torch.float64, device="cuda") + 0.1
        Q = torch.empty_like(f)
        s = torch.cuda.current_stream().cuda_stream
        reps = 200 if nb == 1 else 40
        for _ in range(5): op.computeCollisionBatch(Q, f, nb, stream=s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): op.computeCollisionBatch(Q, f, nb, stream=s)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
        print(f"{'x'.join(map(str, shape)):10s} B={n_gl * n_sph:4d} batch={nb}: {t * 1e3:7.3f} ms per call = {nb / t:9.1f} evals/s", flush=True)
        op.destroy()

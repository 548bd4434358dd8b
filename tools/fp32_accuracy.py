import sys
sys.path.insert(0, "boltzmann-fourier-spectral-method_amd"); sys.path.insert(0, ".")
import numpy as np, torch, bfsm
from bench import WORKLOADS
w = WORKLOADS["cfg5"]; nv, n_gl, n_sph = w["nv"], w["n_gl"], w["n_sph"]
c = bfsm.reference_constants()
f_h, q_exact, _, dv = bfsm.bkw_solution(nv)
f = torch.from_numpy(f_h).cuda(); out = {}
for prec in (64, 32):
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph), nv, nv, nv, c["gamma"], c["b_gamma"], c["L"])
    op.setPrecision(prec); op.initialize()
    Q = torch.empty_like(f); op(Q, f); out[prec] = Q.cpu().numpy(); op.destroy()
d = np.abs(out[32] - out[64]).max() / np.abs(out[64]).max()
l2 = lambda q: float(np.sqrt(((q - q_exact) ** 2).sum() * dv ** 3))
print("cfg5 full quadrature: max|Q32 - Q64| / max|Q64| = %.3e ; L2 error vs exact BKW: fp64 %.3e, fp32 %.3e" % (d, l2(out[64]), l2(out[32])))

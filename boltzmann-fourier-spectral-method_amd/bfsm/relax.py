"""Space-homogeneous relaxation d f / d t = Q(f, f) -- the natural caller of the collision operator (SURVEY.md 8(f3);
the reference stops at a single evaluation of Q).  f stays resident on the device between evaluations; the stage
combinations are plain torch elementwise ops on the same stream as the operator's kernels."""
import numpy as np

from .bkw import bkw_solution


def ssp_rk3_step(op, f, dt, work, stream=0):
    """One strong-stability-preserving RK3 (Shu-Osher) step, in place on f.  work = (Q, f1, f2) device tensors."""
    Q, f1, f2 = work
    op.computeCollisionAsync(Q, f, stream)
    f1.copy_(f).add_(Q, alpha=dt)                                  # f1 = f + dt Q(f)
    op.computeCollisionAsync(Q, f1, stream)
    f2.copy_(f1).add_(Q, alpha=dt).mul_(0.25).add_(f, alpha=0.75)  # f2 = 3/4 f + 1/4 (f1 + dt Q(f1))
    op.computeCollisionAsync(Q, f2, stream)
    f2.add_(Q, alpha=dt)                                           # f2 + dt Q(f2)
    f.mul_(1.0 / 3.0).add_(f2, alpha=2.0 / 3.0)                    # f = 1/3 f + 2/3 (f2 + dt Q(f2))
    return f


def relax_bkw(op, nv, t0, t1, n_steps, torch, S=5.0):
    """Integrate the BKW initial datum f(t0) to t1 with SSP-RK3 and return a dict of diagnostics against the exact
    BKW solution at t1 (L2 error as the drivers define it, relative mass / energy drift, entropy at both ends)."""
    f0_h, _, L, dv = bkw_solution(nv, S, t0)
    f_exact, _, _, _ = bkw_solution(nv, S, t1)
    f = torch.from_numpy(f0_h.reshape(-1).copy()).cuda()
    work = tuple(torch.empty_like(f) for _ in range(3))
    stream = torch.cuda.current_stream().cuda_stream
    dt = (t1 - t0) / n_steps
    for _ in range(n_steps):
        ssp_rk3_step(op, f, dt, work, stream)
    torch.cuda.synchronize()
    f_h = f.cpu().numpy().reshape(nv, nv, nv)
    v = -L + dv / 2 + np.arange(nv) * dv
    v2 = (v * v)[:, None, None] + (v * v)[None, :, None] + (v * v)[None, None, :]

    def entropy(g):
        gp = np.maximum(g, 1e-300)
        return float((gp * np.log(gp)).sum() * dv ** 3)

    d = np.abs(f_h - f_exact)
    return {
        "l2_error": float(np.sqrt((d * d).sum() * dv ** 3)),
        "linf_error": float(d.max()),
        "mass_drift": float(abs(f_h.sum() - f0_h.sum()) / f0_h.sum()),
        "energy_drift": float(abs((f_h * v2).sum() - (f0_h * v2).sum()) / (f0_h * v2).sum()),
        "entropy_start": entropy(f0_h), "entropy_end": entropy(f_h), "entropy_exact_end": entropy(f_exact),
        "evaluations": 3 * n_steps, "dt": dt,
    }

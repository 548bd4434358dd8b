"""BKW (Bobylev-Krook-Wu) known-answer problem of the reference drivers, in numpy
(maxwell_bkw_cuda.cu:58-107 / maxwell_bkw_fftw.cpp:54-99): Maxwell molecules, gamma = 0, b_gamma = 1/(4 pi),
S = 5, R = 2S, L = (3 + sqrt 2)/2 * S, t = 6.5, grid v_i = -L + dv/2 + i dv."""
import numpy as np


def reference_constants(S=5.0):
    return dict(gamma=0.0, b_gamma=1.0 / (4.0 * np.pi), S=S, R=2.0 * S, L=((3.0 + np.sqrt(2.0)) / 2.0) * S)


def bkw_solution(nv, S=5.0, t=6.5):
    """Returns f_bkw, Q_bkw ([nv,nv,nv] float64), L, dv."""
    L = reference_constants(S)["L"]
    dv = 2 * L / nv
    v = -L + dv / 2 + np.arange(nv) * dv
    K = 1 - np.exp(-t / 6)
    dK = np.exp(-t / 6) / 6
    r_sq = (v * v)[:, None, None] + (v * v)[None, :, None] + (v * v)[None, None, :]
    norm = 1 / (2 * (2 * np.pi * K) ** 1.5)
    f = np.exp(-r_sq / (2 * K)) * ((5 * K - 3) / K + (1 - K) / K ** 2 * r_sq) * norm
    Q = (-3 / (2 * K) + r_sq / (2 * K ** 2)) * f
    Q = Q + norm * np.exp(-r_sq / (2 * K)) * (3 / K ** 2 + (K - 2) / K ** 3 * r_sq)
    Q = Q * dK
    return np.ascontiguousarray(f), np.ascontiguousarray(Q), L, dv


def perturbed_input(f_bkw, seed=0x5EED, amp=0.1):
    """Seeded positive, non-symmetric input f = BKW * (1 + amp*u), u ~ U[0,1) from splitmix64(index + seed)."""
    n = f_bkw.size
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return f_bkw * (1.0 + amp * u.reshape(f_bkw.shape))


def error_norms(Q, Q_exact, dv):
    """L1 = sum|d| dv^3, L2 = sqrt(sum d^2 dv^3), Linf = max|d| (maxwell_bkw_cuda.cu:159-180, with a correct max)."""
    d = np.abs(np.asarray(Q, dtype=np.float64).ravel() - np.asarray(Q_exact, dtype=np.float64).ravel())
    return float(d.sum() * dv ** 3), float(np.sqrt((d * d).sum() * dv ** 3)), float(d.max())

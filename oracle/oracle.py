"""TEST INFRASTRUCTURE ONLY -- Python face of the parity oracle.

Two independent CPU restatements of the reference's FFTW path
(Collisions/FFTWBoltzmannOperator.cpp:147-334):

* ``collide``        -- ctypes call into oracle/libbfsm_oracle.so (plain C + OpenMP, own radix-2 FFT).
* ``collide_numpy``  -- a short numpy restatement on numpy.fft (pocketfft), used to cross-check the C
                        oracle with an FFT that shares no code with it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity pinning: see the header of bfsm_oracle.c (pinned by the reference's Results/*.txt BKW norms;
the reference is unbuildable here, so there is no oracle/_ref).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Desc(ctypes.Structure):
    _fields_ = [
        ("nvx", ctypes.c_int), ("nvy", ctypes.c_int), ("nvz", ctypes.c_int),
        ("n_gl", ctypes.c_int), ("n_sph", ctypes.c_int),
        ("gl_nodes", ctypes.POINTER(ctypes.c_double)), ("gl_wts", ctypes.POINTER(ctypes.c_double)),
        ("sph_wts", ctypes.POINTER(ctypes.c_double)),
        ("sx", ctypes.POINTER(ctypes.c_double)), ("sy", ctypes.POINTER(ctypes.c_double)),
        ("sz", ctypes.POINTER(ctypes.c_double)),
        ("gamma", ctypes.c_double), ("b_gamma", ctypes.c_double), ("L", ctypes.c_double),
    ]


def build(force=False):
    """Compile the C oracle with gcc (a few seconds).  Building the checker is not using it."""
    so = os.path.join(_HERE, "libbfsm_oracle.so")
    src = os.path.join(_HERE, "bfsm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libbfsm_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        L.bfsm_oracle_fft3d.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ctypes.c_int]
        L.bfsm_oracle_fft3d.restype = ctypes.c_int
        L.bfsm_oracle_gauss_legendre.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, dp, dp]
        L.bfsm_oracle_gauss_legendre.restype = ctypes.c_int
        L.bfsm_oracle_collide_ex.argtypes = [ctypes.POINTER(_Desc), dp, dp, dp, ctypes.c_longlong,
                                             ctypes.c_longlong, ctypes.c_int]
        L.bfsm_oracle_collide_ex.restype = ctypes.c_int
        L.bfsm_oracle_threads.restype = ctypes.c_int
        L.bfsm_oracle_bkw.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, dp, dp, dp, dp]
        L.bfsm_oracle_bkw.restype = None
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def gauss_legendre(n, a, b):
    """Ascending GL nodes/weights on [a,b] (GSL glfixed semantics, Quadratures/GaussLegendre.hpp:19-21)."""
    x = np.empty(n)
    w = np.empty(n)
    rc = lib().bfsm_oracle_gauss_legendre(n, a, b, _dp(x), _dp(w))
    if rc:
        raise ValueError(f"gauss_legendre failed rc={rc}")
    return x, w


_DESIGN_DEGREE = {6: 3, 12: 5, 32: 7, 48: 9, 70: 11, 94: 13, 120: 15, 156: 17, 192: 19}


def spherical_design(n, data_dir=None):
    """(x, y, z, w) of the n-point symmetric spherical design; w = 4*pi/n
    (Quadratures/SphericalDesign.cpp:12-24,38-48).  Reads the repo's data tables."""
    if n not in _DESIGN_DEGREE:
        raise ValueError("Invalid value of N")
    if data_dir is None:
        data_dir = os.path.join(os.path.dirname(_HERE), "boltzmann-fourier-spectral-method_amd", "data", "sph_design")
    path = os.path.join(data_dir, f"sym_design_t{_DESIGN_DEGREE[n]:03d}_n{n:03d}.dat")
    rows = [ln.split() for ln in open(path) if ln.strip() and not ln.startswith("#")]
    pts = np.array([[float(v) for v in r] for r in rows[1:]])
    assert pts.shape == (n, 3)
    w = np.full(n, 4 * np.pi / n)
    return pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy(), w


def bkw(nv, S=5.0, t=6.5):
    """BKW known-answer pair of the reference drivers (maxwell_bkw_fftw.cpp:54-99).
    Returns f, Q_exact (nv^3 arrays), L, dv."""
    f = np.empty(nv ** 3)
    q = np.empty(nv ** 3)
    L = ctypes.c_double()
    dv = ctypes.c_double()
    lib().bfsm_oracle_bkw(nv, S, t, _dp(f), _dp(q), ctypes.byref(L), ctypes.byref(dv))
    return f.reshape(nv, nv, nv), q.reshape(nv, nv, nv), L.value, dv.value


def perturbed_input(f_bkw, seed=0x5EED, amp=0.1):
    """Seeded positive, non-symmetric input: f = BKW * (1 + amp*u), u ~ U[0,1) from a fixed 64-bit
    splitmix generator of the linear index -- exercises the Nyquist planes BKW cannot."""
    n = f_bkw.size
    idx = np.arange(n, dtype=np.uint64) + np.uint64(seed)
    with np.errstate(over="ignore"):
        z = (idx + np.uint64(0x9E3779B97F4A7C15)) * np.uint64(1)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return f_bkw * (1.0 + amp * u.reshape(f_bkw.shape))


def error_norms(Q, Q_exact, dv):
    """L1, L2, Linf as the drivers print them (maxwell_bkw_fftw.cpp:145-166), with a correct max."""
    d = np.abs(np.asarray(Q).ravel() - np.asarray(Q_exact).ravel())
    return float(d.sum() * dv ** 3), float(np.sqrt((d ** 2).sum() * dv ** 3)), float(d.max())


def _make_desc(nv, gl, sph, gamma, b_gamma, L):
    nvx, nvy, nvz = (nv, nv, nv) if np.isscalar(nv) else nv
    keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (gl[0], gl[1], sph[3], sph[0], sph[1], sph[2])]
    d = _Desc(nvx, nvy, nvz, len(keep[0]), len(keep[2]), _dp(keep[0]), _dp(keep[1]), _dp(keep[2]),
              _dp(keep[3]), _dp(keep[4]), _dp(keep[5]), gamma, b_gamma, L)
    return d, keep


def collide(f, gl, sph, gamma, b_gamma, L, dir_range=None, threads=0, return_qhat=False):
    """C oracle.  f: real [nvx][nvy][nvz]; gl = (nodes, weights); sph = (x, y, z, w).
    dir_range = (begin, end) over flattened b = r*n_sph + s restricts the gain sum (loss always full)."""
    f = np.ascontiguousarray(f, dtype=np.float64)
    d, keep = _make_desc(f.shape, gl, sph, gamma, b_gamma, L)
    B = d.n_gl * d.n_sph
    b0, b1 = (0, B) if dir_range is None else dir_range
    Q = np.empty_like(f)
    qhat = np.empty(f.shape + (2,)) if return_qhat else None
    rc = lib().bfsm_oracle_collide_ex(ctypes.byref(d), _dp(f), _dp(Q), _dp(qhat) if return_qhat else None,
                                      b0, b1, threads)
    if rc:
        raise RuntimeError(f"bfsm_oracle_collide failed rc={rc}")
    if return_qhat:
        return Q, qhat[..., 0] + 1j * qhat[..., 1]
    return Q


def fft3d(a, sign):
    """C oracle 3-D c2c DFT (unnormalised); sign=-1 forward, +1 backward."""
    a = np.ascontiguousarray(a, dtype=np.complex128).copy()
    v = a.view(np.float64)
    rc = lib().bfsm_oracle_fft3d(a.shape[0], a.shape[1], a.shape[2], _dp(v), sign)
    if rc:
        raise RuntimeError("fft3d failed")
    return a


def collide_numpy(f, gl, sph, gamma, b_gamma, L, dir_range=None, return_qhat=False):
    """numpy restatement of FFTWBoltzmannOperator.cpp:147-334 (independent FFT: numpy.fft)."""
    f = np.asarray(f, dtype=np.float64)
    nx, ny, nz = f.shape
    G = f.size
    rho, wr = gl
    sx, sy, sz, ws = sph
    eps = np.finfo(np.float64).eps

    def sincc(x):  # FFTWBoltzmannOperator.hpp:17-21
        return np.sin(x + eps) / (x + eps)

    def modes(n):  # cpp:50-57
        return np.concatenate([np.arange(0, n // 2), np.arange(-(n // 2), 0)]).astype(np.float64)

    lx, ly, lz = np.meshgrid(modes(nx), modes(ny), modes(nz), indexing="ij")
    norm_l = np.sqrt(lx * lx + ly * ly + lz * lz)
    f_hat = np.fft.fftn(f)                                            # cpp:185-186
    qhat = np.zeros(f.shape, dtype=np.complex128)
    B = len(rho) * len(ws)
    b0, b1 = (0, B) if dir_range is None else dir_range
    for b in range(b0, b1):                                           # cpp:191-196
        r, s = divmod(b, len(ws))
        tmp = -(np.pi / (2 * L)) * rho[r] * (lx * sx[s] + ly * sy[s] + lz * sz[s])   # cpp:205-209
        a = np.cos(tmp) + 1j * np.sin(tmp)
        A1 = np.fft.ifftn(a * f_hat)                                  # cpp:216-230 (ifftn carries the 1/G)
        A2 = np.fft.ifftn(np.conj(a) * f_hat)
        P_hat = np.fft.fftn(A1 * A2)                                  # cpp:233-249
        weight = (1.0 / G) * wr[r] * ws[s] * rho[r] ** (gamma + 2)    # cpp:252
        beta1 = 4 * np.pi * b_gamma * sincc(np.pi * rho[r] * norm_l / (2 * L))   # cpp:261-262
        qhat += weight * beta1 * P_hat                                # cpp:267-270
    beta2 = np.zeros(f.shape)
    for r in range(len(rho)):                                         # cpp:290-293
        beta2 += 16 * np.pi ** 2 * b_gamma * wr[r] * rho[r] ** (gamma + 2) * sincc(np.pi * rho[r] * norm_l / L)
    Q_gain = np.fft.ifftn(qhat) * G                                   # cpp:305 (unnormalised backward)
    Bf = np.fft.ifftn(beta2 * f_hat)                                  # cpp:295-296,309 (1/G folded)
    Q = Q_gain.real - (Bf * f).real                                   # cpp:314-330
    return (Q, qhat) if return_qhat else Q

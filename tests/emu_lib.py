"""ctypes access to tests/emu/libbfsm_emu.so -- the host lock-step emulation of the HIP kernel bodies (test harness)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "emu"), "-s"])
        from bfsm import capi
        # BFSM_EMU_LIB: an alternative build of the same emulator (the AddressSanitizer build of tests/emu/Makefile)
        L = ctypes.CDLL(os.environ.get("BFSM_EMU_LIB") or os.path.join(_HERE, "emu", "libbfsm_emu.so"))
        dp = ctypes.POINTER(ctypes.c_double)
        L.bfsm_emu_collide.argtypes = [ctypes.POINTER(capi.Desc), dp, dp, dp]
        L.bfsm_emu_collide.restype = ctypes.c_int
        L.bfsm_emu_fft3d.argtypes = [ctypes.c_int, ctypes.c_int, dp, ctypes.c_int, ctypes.c_int]
        L.bfsm_emu_fft3d.restype = ctypes.c_int
        ip = ctypes.POINTER(ctypes.c_int)
        L.bfsm_emu_plan.argtypes = [ctypes.POINTER(capi.Desc), ip, ctypes.c_int, ip, ctypes.c_int, ip]
        L.bfsm_emu_plan.restype = ctypes.c_int
        _LIB = L
    return _LIB


def make_desc(nv, gl, sph, gamma, b_gamma, L, precision=64, dir_range=(0, 0), max_chunk=0, flags=0, max_batch=0):
    """gl = (nodes, weights), sph = (x, y, z, w).  Returns (Desc, keepalive)."""
    from bfsm import capi
    dp = ctypes.POINTER(ctypes.c_double)
    keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (gl[0], gl[1], sph[3], sph[0], sph[1], sph[2])]
    nx, ny, nz = (nv, nv, nv) if np.isscalar(nv) else (int(v) for v in nv)     # nv: one extent or (nvx, nvy, nvz)
    d = capi.Desc(nx, ny, nz, len(keep[0]), len(keep[2]), *[a.ctypes.data_as(dp) for a in keep],
                  gamma, b_gamma, L, precision, 0, dir_range[0], dir_range[1], max_chunk, flags, max_batch)
    return d, keep


def collide(f, gl, sph, gamma, b_gamma, L, precision=64, dir_range=(0, 0), max_chunk=0, want_Q=True, flags=0):
    nv = f.shape[0] if f.shape[0] == f.shape[1] == f.shape[2] else f.shape
    d, keep = make_desc(nv, gl, sph, gamma, b_gamma, L, precision, dir_range, max_chunk, flags)
    f = np.ascontiguousarray(f, dtype=np.float64)
    Q = np.empty_like(f)
    qh = np.empty(f.shape + (2,))
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib().bfsm_emu_collide(ctypes.byref(d), f.ctypes.data_as(dp), Q.ctypes.data_as(dp) if want_Q else None,
                                qh.ctypes.data_as(dp))
    if rc:
        raise RuntimeError(f"bfsm_emu_collide rc={rc}")
    qhat_t = qh[..., 0] + 1j * qh[..., 1]            # fused pipeline: [lx][lz][ly]; size-generic path: natural
    if np.isscalar(nv) and nv in (16, 24, 32, 40, 48, 64, 80, 96, 128):
        qhat_t = np.ascontiguousarray(qhat_t.transpose(0, 2, 1))   # -> [lx][ly][lz]
    return (Q if want_Q else None), qhat_t


def fft3d(a, sign, precision=64):
    """a: [batch][N][N][N] complex.  forward: natural in, natural out (un-transposed here); backward likewise."""
    a = np.ascontiguousarray(a, dtype=np.complex128)
    batch, n = a.shape[0], a.shape[1]
    if sign > 0:
        a = np.ascontiguousarray(a.transpose(0, 1, 3, 2))       # natural spectral -> [lx][lz][ly]
    buf = a.copy()
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib().bfsm_emu_fft3d(n, precision, buf.view(np.float64).ctypes.data_as(dp), batch, sign)
    if rc:
        raise RuntimeError(f"bfsm_emu_fft3d rc={rc}")
    if sign < 0:
        buf = np.ascontiguousarray(buf.transpose(0, 1, 3, 2))   # [lx][lz][ly] -> natural
    return buf


def plan(nv, n_gl, n_sph, precision=64, dir_range=(0, 0), max_chunk=0, flags=0, sph=None):
    """Returns (chunks, segments): chunk rows (n_seg, dir0, n, per_group, seg0), segment rows (chunk, d0, n, r)."""
    gl = (np.ones(n_gl), np.ones(n_gl))
    if sph is None:
        sph = (np.ones(n_sph), np.zeros(n_sph), np.zeros(n_sph), np.ones(n_sph))
    d, keep = make_desc(nv, gl, sph, 0.0, 1.0, 1.0, precision, dir_range, max_chunk, flags)
    crow = (ctypes.c_int * (5 * 4096))()
    srow = (ctypes.c_int * (5 * 65536))()
    nseg = ctypes.c_int()
    n = lib().bfsm_emu_plan(ctypes.byref(d), crow, 4096, srow, 65536, ctypes.byref(nseg))
    if n < 0:
        raise ValueError(f"plan rejected rc={-n}")
    return ([tuple(crow[5 * i:5 * i + 5]) for i in range(n)],
            [tuple(srow[5 * i:5 * i + 4]) for i in range(nseg.value)])


class EmuOperator:
    """Host-emulated stand-in for bfsm.HIPBoltzmannOperator with the sharded interface (gainPartial / qhat / finish),
    used by the world-size-2 gloo tests of the multi-GPU logic.  CPU torch tensors instead of device tensors."""

    def __init__(self, nv, gl, sph, gamma, b_gamma, L, dir_range=(0, 0), max_chunk=0):
        import torch
        self.nv, self.gl, self.sph = nv, gl, sph
        self.args = (gamma, b_gamma, L)
        self.dir_range, self.max_chunk = dir_range, max_chunk
        self.qhat = torch.zeros(2 * nv ** 3, dtype=torch.float64)     # the "handle-owned" partial Q_gain_hat

    def gainPartial(self, f, stream=0):
        import torch
        d, keep = make_desc(self.nv, self.gl, self.sph, *self.args, 64, self.dir_range, self.max_chunk)
        fh = np.ascontiguousarray(f.numpy(), dtype=np.float64)
        qh = np.empty(2 * self.nv ** 3)
        dp = ctypes.POINTER(ctypes.c_double)
        rc = lib().bfsm_emu_collide(ctypes.byref(d), fh.ctypes.data_as(dp), None, qh.ctypes.data_as(dp))
        if rc:
            raise RuntimeError(f"bfsm_emu_collide rc={rc}")
        self.qhat.copy_(torch.from_numpy(qh))

    def finishPartial(self, Q, f, with_loss, stream=0):
        self.finish(Q, f, stream, with_loss=with_loss)

    def finish(self, Q, f, stream=0, with_loss=True):
        import torch
        L = lib()
        dp = ctypes.POINTER(ctypes.c_double)
        if not hasattr(L.bfsm_emu_finish, "_typed"):
            from bfsm import capi
            L.bfsm_emu_finish.argtypes = [ctypes.POINTER(capi.Desc), dp, dp, dp, ctypes.c_int]
            L.bfsm_emu_finish.restype = ctypes.c_int
            L.bfsm_emu_finish._typed = True
        d, keep = make_desc(self.nv, self.gl, self.sph, *self.args, 64, self.dir_range, self.max_chunk)
        fh = np.ascontiguousarray(f.numpy(), dtype=np.float64)
        qh = np.ascontiguousarray(self.qhat.numpy())
        out = np.empty(self.nv ** 3)
        rc = L.bfsm_emu_finish(ctypes.byref(d), fh.ctypes.data_as(dp), qh.ctypes.data_as(dp), out.ctypes.data_as(dp),
                               1 if with_loss else 0)
        if rc:
            raise RuntimeError(f"bfsm_emu_finish rc={rc}")
        Q.copy_(torch.from_numpy(out))


def collide_batch(fs, gl, sph, gamma, b_gamma, L, precision=64, max_chunk=0, flags=0):
    """fs: [n_batch][nvx][nvy][nvz]; one emulated bfsm_collide_batch call.  Returns Q with the same shape."""
    fs = np.ascontiguousarray(fs, dtype=np.float64)
    nb = fs.shape[0]
    nv = fs.shape[1] if fs.shape[1] == fs.shape[2] == fs.shape[3] else fs.shape[1:]
    d, keep = make_desc(nv, gl, sph, gamma, b_gamma, L, precision, (0, 0), max_chunk, flags, max_batch=nb)
    L_ = lib()
    dp = ctypes.POINTER(ctypes.c_double)
    if not hasattr(L_.bfsm_emu_collide_batch, "_typed"):
        from bfsm import capi
        L_.bfsm_emu_collide_batch.argtypes = [ctypes.POINTER(capi.Desc), dp, dp, dp, ctypes.c_int]
        L_.bfsm_emu_collide_batch.restype = ctypes.c_int
        L_.bfsm_emu_collide_batch._typed = True
    Q = np.empty_like(fs)
    rc = L_.bfsm_emu_collide_batch(ctypes.byref(d), fs.ctypes.data_as(dp), Q.ctypes.data_as(dp), None, nb)
    if rc:
        raise RuntimeError(f"bfsm_emu_collide_batch rc={rc}")
    return Q


class EmuOperatorFused(EmuOperator):
    """The same with collidePartial (the one-call form bfsm.sharded_step prefers when the operator offers it)."""

    def collidePartial(self, Q, f, with_loss, stream=0):
        import torch
        L = lib()
        dp = ctypes.POINTER(ctypes.c_double)
        if not hasattr(L.bfsm_emu_collide_partial, "_typed"):
            from bfsm import capi
            L.bfsm_emu_collide_partial.argtypes = [ctypes.POINTER(capi.Desc), dp, dp, ctypes.c_int]
            L.bfsm_emu_collide_partial.restype = ctypes.c_int
            L.bfsm_emu_collide_partial._typed = True
        d, keep = make_desc(self.nv, self.gl, self.sph, *self.args, 64, self.dir_range, self.max_chunk)
        fh = np.ascontiguousarray(f.numpy(), dtype=np.float64)
        out = np.empty(self.nv ** 3)
        rc = L.bfsm_emu_collide_partial(ctypes.byref(d), fh.ctypes.data_as(dp), out.ctypes.data_as(dp), 1 if with_loss else 0)
        if rc:
            raise RuntimeError(f"bfsm_emu_collide_partial rc={rc}")
        Q.copy_(torch.from_numpy(out))


def collide_partial(f, gl, sph, gamma, b_gamma, L, precision=64, dir_range=(0, 0), with_loss=True, flags=0, max_chunk=0):
    """Emulated bfsm_collide_partial_async (what bfsm_collide runs): at N = 16 the whole-direction kernels unless
    flags has BFSM_FLAG_NO_SMALL_PATH, otherwise the fused plane-tile sequence."""
    nv = f.shape[0] if f.shape[0] == f.shape[1] == f.shape[2] else f.shape
    d, keep = make_desc(nv, gl, sph, gamma, b_gamma, L, precision, dir_range, max_chunk, flags)
    f = np.ascontiguousarray(f, dtype=np.float64)
    Q = np.empty_like(f)
    dp = ctypes.POINTER(ctypes.c_double)
    L_ = lib()
    if not hasattr(L_.bfsm_emu_collide_partial, "_typed"):
        from bfsm import capi
        L_.bfsm_emu_collide_partial.argtypes = [ctypes.POINTER(capi.Desc), dp, dp, ctypes.c_int]
        L_.bfsm_emu_collide_partial.restype = ctypes.c_int
        L_.bfsm_emu_collide_partial._typed = True
    rc = L_.bfsm_emu_collide_partial(ctypes.byref(d), f.ctypes.data_as(dp), Q.ctypes.data_as(dp), 1 if with_loss else 0)
    if rc:
        raise RuntimeError(f"bfsm_emu_collide_partial rc={rc}")
    return Q

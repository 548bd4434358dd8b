"""HIPBoltzmannOperator -- Python mirror of BoltzmannOperator<CUDA_Backend> (Collisions/CUDABoltzmannOperator.hpp:44-131)
over the C-ABI.  Same life-cycle as the reference object: construct (no work), initialize(), then call it with DEVICE
arrays (torch CUDA tensors of float64) as often as needed.  Errors raise BfsmError (the reference prints and exits)."""
import ctypes

import numpy as np

from . import capi


def shard_range(n_dirs, rank, world):
    """Contiguous, balanced range of flattened quadrature directions b = r*M_sph + s owned by `rank`."""
    base, rem = divmod(n_dirs, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class HIPBoltzmannOperator:
    def __init__(self, gl_quadrature, spherical_quadrature, Nvx, Nvy, Nvz, gamma, b_gamma, L):
        self.gl_quadrature = gl_quadrature
        self.spherical_quadrature = spherical_quadrature
        self.Nvx, self.Nvy, self.Nvz = int(Nvx), int(Nvy), int(Nvz)
        self.gamma, self.b_gamma, self.L = float(gamma), float(b_gamma), float(L)
        self._precision = capi.BFSM_F64
        self._device = 0
        self._dir_range = (0, 0)
        self._max_chunk = 0
        self._flags = 0
        self._max_batch = 0
        self._h = None
        self._lib = None

    # knobs, to be set before initialize() -- like setWisdomFileName in the FFTW backend (FFTWBoltzmannOperator.hpp:39-41)
    def setPrecision(self, bits):
        self._precision = int(bits)

    def setDevice(self, ordinal):
        self._device = int(ordinal)

    def setDirectionShard(self, begin, end):
        self._dir_range = (int(begin), int(end))

    def setMaxChunk(self, n):
        self._max_chunk = int(n)

    def setMaxBatch(self, n):
        """Distributions per computeCollisionBatch call (scratch scales with it)."""
        self._max_batch = int(n)

    def setProfiling(self, on=True):
        self._flags = (self._flags | capi.BFSM_FLAG_PROFILE) if on else (self._flags & ~capi.BFSM_FLAG_PROFILE)

    def setExactReductions(self, on=True, hermitian=False):
        """Opt-in SURVEY 8(f1) reductions (antipodal pairs + one forward FFT per radial node); default off.
        hermitian=True adds BFSM_FLAG_HERMITIAN (only the lx >= 0 planes of A1', A2' are computed and stored)."""
        f = capi.BFSM_FLAG_EXACT_REDUCTIONS | capi.BFSM_FLAG_HERMITIAN
        self._flags &= ~f
        if on:
            self._flags |= capi.BFSM_FLAG_EXACT_REDUCTIONS | (capi.BFSM_FLAG_HERMITIAN if hermitian else 0)

    def setSmallPath(self, on=True):
        """N = 16: use (default) or avoid the whole-direction kernels for single evaluations (BFSM_FLAG_NO_SMALL_PATH)."""
        self._flags = (self._flags & ~capi.BFSM_FLAG_NO_SMALL_PATH) | (0 if on else capi.BFSM_FLAG_NO_SMALL_PATH)

    def getBackendName(self):
        return (self._lib or capi.load_library()).bfsm_backend_name().decode()

    def _check(self, rc):
        if rc != capi.BFSM_OK:
            msg = self._lib.bfsm_last_error(self._h).decode() if self._lib else "?"
            raise capi.BfsmError(rc, msg)

    def initialize(self):
        self._lib = capi.load_library()
        gl, sp = self.gl_quadrature, self.spherical_quadrature
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in
                (gl.getNodes(), gl.getWeights(), sp.getWeights(), sp.getx(), sp.gety(), sp.getz())]
        dp = ctypes.POINTER(ctypes.c_double)
        d = capi.Desc(self.Nvx, self.Nvy, self.Nvz, gl.getNumberOfPoints(), sp.getNumberOfPoints(),
                      *[a.ctypes.data_as(dp) for a in keep],
                      self.gamma, self.b_gamma, self.L, self._precision, self._device,
                      self._dir_range[0], self._dir_range[1], self._max_chunk, self._flags, self._max_batch)
        h = ctypes.c_void_p()
        rc = self._lib.bfsm_create(ctypes.byref(d), ctypes.byref(h))
        if rc != capi.BFSM_OK:
            raise capi.BfsmError(rc, self._lib.bfsm_last_error(None).decode())
        self._h = h

    def _require(self, f, Q=None):
        if self._h is None:
            raise RuntimeError("initialize() has not been called")
        G = self.Nvx * self.Nvy * self.Nvz
        for t in (f, Q):
            if t is None:
                continue
            if not (t.is_cuda and t.dtype.is_floating_point and t.element_size() == 8 and t.is_contiguous() and t.numel() == G):
                raise ValueError("f and Q must be contiguous float64 CUDA tensors with Nvx*Nvy*Nvz elements")

    def computeCollision(self, Q, f_in):
        """Blocking evaluation, device tensors in and out (CUDABoltzmannOperator.cu:119-220)."""
        self._require(f_in, Q)
        self._check(self._lib.bfsm_collide(self._h, _ptr(Q), _ptr(f_in)))

    __call__ = computeCollision

    def computeCollisionAsync(self, Q, f_in, stream=0):
        self._require(f_in, Q)
        self._check(self._lib.bfsm_collide_async(self._h, _ptr(Q), _ptr(f_in), ctypes.c_void_p(stream)))

    def computeCollisionBatch(self, Q, f_in, n_batch, stream=None):
        """n_batch distributions [n_batch][Nvx*Nvy*Nvz] in one set of launches.  Blocking when
        stream is None, otherwise enqueued on that stream."""
        if self._h is None:
            raise RuntimeError("initialize() has not been called")
        G = self.Nvx * self.Nvy * self.Nvz
        for t in (f_in, Q):
            if not (t.is_cuda and t.element_size() == 8 and t.is_contiguous() and t.numel() == n_batch * G):
                raise ValueError("f and Q must be contiguous float64 CUDA tensors with n_batch*Nvx*Nvy*Nvz elements")
        if stream is None:
            self._check(self._lib.bfsm_collide_batch(self._h, _ptr(Q), _ptr(f_in), int(n_batch)))
        else:
            self._check(self._lib.bfsm_collide_batch_async(self._h, _ptr(Q), _ptr(f_in), int(n_batch),
                                                           ctypes.c_void_p(stream)))

    def collideBatchPartial(self, Q, f_in, n_batch, with_loss, stream=0):
        """Batch x direction shard: member i of Q = this shard's partial result for member i [- its loss term]; the
        caller sums Q over the ranks with one collective for the whole batch."""
        G = self.Nvx * self.Nvy * self.Nvz
        for t in (f_in, Q):
            if not (t.is_cuda and t.element_size() == 8 and t.is_contiguous() and t.numel() == n_batch * G):
                raise ValueError("f and Q must be contiguous float64 CUDA tensors with n_batch*Nvx*Nvy*Nvz elements")
        self._check(self._lib.bfsm_collide_batch_partial_async(self._h, _ptr(Q), _ptr(f_in), int(n_batch),
                                                               1 if with_loss else 0, ctypes.c_void_p(stream)))

    def gainPartial(self, f_in, stream=0):
        self._require(f_in)
        self._check(self._lib.bfsm_gain_partial(self._h, _ptr(f_in), ctypes.c_void_p(stream)))

    def finish(self, Q, f_in, stream=0):
        self._require(f_in, Q)
        self._check(self._lib.bfsm_finish(self._h, _ptr(Q), _ptr(f_in), ctypes.c_void_p(stream)))

    def finishPartial(self, Q, f_in, with_loss, stream=0):
        """Q = Re IFFT(this shard's partial Q_gain_hat) [- loss term if with_loss]; the caller sums Q over ranks."""
        self._require(f_in, Q)
        self._check(self._lib.bfsm_finish_partial(self._h, _ptr(Q), _ptr(f_in), 1 if with_loss else 0,
                                                  ctypes.c_void_p(stream)))

    def collidePartial(self, Q, f_in, with_loss, stream=0):
        """gainPartial + finishPartial as one call (slab reduce fused into the tail; qhatBuffer() is not updated)."""
        self._require(f_in, Q)
        self._check(self._lib.bfsm_collide_partial_async(self._h, _ptr(Q), _ptr(f_in), 1 if with_loss else 0,
                                                         ctypes.c_void_p(stream)))

    def qhatBuffer(self):
        """(device pointer, n_elems, precision) of the partial Q_gain_hat owned by the handle."""
        n = ctypes.c_size_t()
        prec = ctypes.c_int()
        p = self._lib.bfsm_qhat_buffer(self._h, ctypes.byref(n), ctypes.byref(prec))
        return p, n.value, prec.value

    def synchronize(self):
        self._check(self._lib.bfsm_synchronize(self._h))

    def fft3d(self, data, batch, sign):
        self._check(self._lib.bfsm_fft3d(self._h, _ptr(data), int(batch), int(sign)))

    def counters(self):
        c = capi.Counters()
        self._check(self._lib.bfsm_get_counters(self._h, ctypes.byref(c)))
        return c

    def destroy(self):
        if self._h is not None and self._lib is not None:
            self._lib.bfsm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

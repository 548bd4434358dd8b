"""ctypes declarations of include/bfsm.h -- one Python name per exported C symbol."""
import ctypes
import os

BFSM_OK = 0
BFSM_F64 = 64
BFSM_F32 = 32
BFSM_FLAG_PROFILE = 1
BFSM_FLAG_EXACT_REDUCTIONS = 2
BFSM_FLAG_HERMITIAN = 4
BFSM_FLAG_NO_SMALL_PATH = 8
KERNEL_NAMES = ("fft_f", "gain_inv", "gain_line", "gain_fwd", "reduce", "tail")
K_COUNT = len(KERNEL_NAMES)

# every symbol include/bfsm.h declares (checked by tests/test_capi_symbols.py)
EXPORTED_SYMBOLS = (
    "bfsm_create", "bfsm_collide", "bfsm_collide_async", "bfsm_collide_batch", "bfsm_collide_batch_async", "bfsm_collide_batch_partial_async", "bfsm_gain_partial", "bfsm_finish", "bfsm_finish_partial", "bfsm_collide_partial_async",
    "bfsm_qhat_buffer",
    "bfsm_synchronize", "bfsm_fft3d", "bfsm_get_counters", "bfsm_destroy", "bfsm_last_error", "bfsm_backend_name",
    "bfsm_version",
)

_dp = ctypes.POINTER(ctypes.c_double)


class Desc(ctypes.Structure):
    """struct bfsm_desc"""
    _fields_ = [
        ("nvx", ctypes.c_int), ("nvy", ctypes.c_int), ("nvz", ctypes.c_int),
        ("n_gl", ctypes.c_int), ("n_sph", ctypes.c_int),
        ("gl_nodes", _dp), ("gl_wts", _dp), ("sph_wts", _dp), ("sx", _dp), ("sy", _dp), ("sz", _dp),
        ("gamma", ctypes.c_double), ("b_gamma", ctypes.c_double), ("L", ctypes.c_double),
        ("precision", ctypes.c_int), ("device", ctypes.c_int),
        ("dir_begin", ctypes.c_longlong), ("dir_end", ctypes.c_longlong),
        ("max_chunk", ctypes.c_int), ("flags", ctypes.c_int), ("max_batch", ctypes.c_int),
    ]


class Counters(ctypes.Structure):
    """struct bfsm_counters"""
    _fields_ = [
        ("alg_bytes_per_eval", ctypes.c_double),
        ("kernel_ms", ctypes.c_double * K_COUNT),
        ("kernel_alg_bytes", ctypes.c_double * K_COUNT),
        ("kernel_launches", ctypes.c_int * K_COUNT),
        ("n_chunks", ctypes.c_int), ("chunk_dirs", ctypes.c_int), ("n_dirs", ctypes.c_longlong),
        ("moved_bytes_per_eval", ctypes.c_double), ("exact_reductions", ctypes.c_int), ("antipodal_merged", ctypes.c_int),
    ]


class BfsmError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"bfsm error {code}: {message}")
        self.code = code


def lib_path():
    if os.environ.get("BFSM_LIB"):            # experiments: an alternative build of the same library
        return os.environ["BFSM_LIB"]
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libbfsm_hip.so")


_LIB = None


def load_library(path=None):
    """dlopen libbfsm_hip.so (built in-tree by `make` / __graft_entry__.build()).  Fails loudly when absent."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or lib_path()
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `make -C boltzmann-fourier-spectral-method_amd` (hipcc, gfx950). "
            "There is no CPU fallback for the collision operator.")
    L = ctypes.CDLL(p)
    vp = ctypes.c_void_p
    L.bfsm_create.argtypes = [ctypes.POINTER(Desc), ctypes.POINTER(vp)]
    L.bfsm_create.restype = ctypes.c_int
    L.bfsm_collide.argtypes = [vp, vp, vp]
    L.bfsm_collide.restype = ctypes.c_int
    L.bfsm_collide_async.argtypes = [vp, vp, vp, vp]
    L.bfsm_collide_async.restype = ctypes.c_int
    L.bfsm_collide_batch.argtypes = [vp, vp, vp, ctypes.c_int]
    L.bfsm_collide_batch.restype = ctypes.c_int
    L.bfsm_collide_batch_async.argtypes = [vp, vp, vp, ctypes.c_int, vp]
    L.bfsm_collide_batch_async.restype = ctypes.c_int
    L.bfsm_collide_batch_partial_async.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]
    L.bfsm_collide_batch_partial_async.restype = ctypes.c_int
    L.bfsm_gain_partial.argtypes = [vp, vp, vp]
    L.bfsm_gain_partial.restype = ctypes.c_int
    L.bfsm_finish.argtypes = [vp, vp, vp, vp]
    L.bfsm_finish.restype = ctypes.c_int
    L.bfsm_finish_partial.argtypes = [vp, vp, vp, ctypes.c_int, vp]
    L.bfsm_finish_partial.restype = ctypes.c_int
    L.bfsm_collide_partial_async.argtypes = [vp, vp, vp, ctypes.c_int, vp]
    L.bfsm_collide_partial_async.restype = ctypes.c_int
    L.bfsm_qhat_buffer.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]
    L.bfsm_qhat_buffer.restype = vp
    L.bfsm_synchronize.argtypes = [vp]
    L.bfsm_synchronize.restype = ctypes.c_int
    L.bfsm_fft3d.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int]
    L.bfsm_fft3d.restype = ctypes.c_int
    L.bfsm_get_counters.argtypes = [vp, ctypes.POINTER(Counters)]
    L.bfsm_get_counters.restype = ctypes.c_int
    L.bfsm_destroy.argtypes = [vp]
    L.bfsm_destroy.restype = ctypes.c_int
    L.bfsm_last_error.argtypes = [vp]
    L.bfsm_last_error.restype = ctypes.c_char_p
    L.bfsm_backend_name.argtypes = []
    L.bfsm_backend_name.restype = ctypes.c_char_p
    L.bfsm_version.argtypes = []
    L.bfsm_version.restype = ctypes.c_int
    if path is None:
        _LIB = L
    return L

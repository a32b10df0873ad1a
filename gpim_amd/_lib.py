"""
ctypes binding of libgpimhip.so (include/gpimhip.h).

The product path has no CPU fallback: if the shared library cannot be loaded, or no
MI355X-class device is visible to torch, every entry point raises RuntimeError.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpimhip.so")

OK, E_BADARG, E_HIP, E_NOT_PD, E_NOMEM = 0, -1, -2, -3, -4
KERNEL_IDS = {"RBF": 0, "Matern52": 1, "RationalQuadratic": 2}
ACQ_IDS = {"cb": 0, "ei": 1, "poi": 2}
MAX_DIM, MAX_PARAMS = 4, 8

c_dp = ctypes.c_void_p      # device pointers travel as plain addresses


class ModelStruct(ctypes.Structure):
    """gpimhip_model_t"""
    _fields_ = [("kernel", ctypes.c_int32), ("dim", ctypes.c_int32), ("n_ls", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("amp_lo", ctypes.c_double), ("amp_hi", ctypes.c_double),
                ("ls_lo", ctypes.c_double * MAX_DIM), ("ls_hi", ctypes.c_double * MAX_DIM),
                ("jitter", ctypes.c_double)]


_lib = None

_PROTOS = {
    "gpimhip_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p]),
    "gpimhip_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "gpimhip_last_error": (ctypes.c_char_p, []),
    "gpimhip_version": (ctypes.c_int, []),
    "gpimhip_shutdown": (ctypes.c_int, []),
    "gpimhip_workspace_bytes": (ctypes.c_int64, [ctypes.c_void_p]),
    "gpimhip_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "gpimhip_fit_completed": (ctypes.c_int, [ctypes.c_void_p]),
    "gpimhip_timing_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "gpimhip_timing_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_int64)]),
    "gpimhip_kmat": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, ctypes.c_int64, c_dp,
                                    ctypes.c_int64, c_dp, ctypes.c_double, c_dp, ctypes.c_int64]),
    "gpimhip_potrf": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int64, c_dp]),
    "gpimhip_nll_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, c_dp, ctypes.c_int64,
                                        c_dp, c_dp, c_dp]),
    "gpimhip_fit_exact": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, c_dp, ctypes.c_int64,
                                         c_dp, ctypes.c_double, ctypes.c_int32, c_dp, c_dp]),
    "gpimhip_predict_exact": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, c_dp,
                                             ctypes.c_int64, c_dp, c_dp, ctypes.c_int64, c_dp, c_dp]),
    "gpimhip_fit_exact_batched": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, ctypes.c_int64,
                                                 c_dp, ctypes.c_int64, ctypes.c_int32, c_dp, ctypes.c_double,
                                                 ctypes.c_int32, c_dp, c_dp]),
    "gpimhip_predict_exact_batched": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp,
                                                     ctypes.c_int64, c_dp, ctypes.c_int64, ctypes.c_int32, c_dp,
                                                     c_dp, ctypes.c_int64, c_dp, c_dp]),
    "gpimhip_vfe_nll_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, c_dp, ctypes.c_int64,
                                            ctypes.c_int64, c_dp, c_dp, c_dp]),
    "gpimhip_fit_vfe": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, c_dp, ctypes.c_int64,
                                       ctypes.c_int64, c_dp, ctypes.c_double, ctypes.c_int32, c_dp, c_dp, c_dp]),
    "gpimhip_predict_vfe": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, c_dp, ctypes.c_int64,
                                           ctypes.c_int64, c_dp, c_dp, ctypes.c_int64, c_dp, c_dp]),
    "gpimhip_fit_vfe_batched": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, ctypes.c_int64, c_dp,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, c_dp, ctypes.c_double,
                                               ctypes.c_int32, c_dp, c_dp, c_dp]),
    "gpimhip_predict_vfe_batched": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, ctypes.c_int64, c_dp,
                                                   ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, c_dp, c_dp,
                                                   ctypes.c_int64, ctypes.c_int64, c_dp, c_dp]),
    "gpimhip_kron_nll_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), ctypes.c_int32,
                                             ctypes.POINTER(ctypes.c_int32), c_dp, c_dp, c_dp, c_dp, c_dp]),
    "gpimhip_fit_kron": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), ctypes.c_int32,
                                        ctypes.POINTER(ctypes.c_int32), c_dp, c_dp, c_dp, ctypes.c_double,
                                        ctypes.c_int32, c_dp, c_dp]),
    "gpimhip_predict_kron": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), ctypes.c_int32,
                                            ctypes.POINTER(ctypes.c_int32), c_dp, c_dp, c_dp,
                                            ctypes.POINTER(ctypes.c_int32), c_dp, c_dp, c_dp]),
    "gpimhip_acq": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, c_dp, c_dp, ctypes.c_int64, ctypes.c_double,
                                   ctypes.c_double, c_dp, c_dp]),
    "gpimhip_nanmax": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, c_dp]),
    "gpimhip_topk": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                    c_dp, c_dp, c_dp]),
    "gpimhip_dist_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "gpimhip_dist_panel_factor": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32,
                                                 ctypes.c_int32, c_dp, c_dp]),
    "gpimhip_dist_setup": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]),
    "gpimhip_dist_panel_pack": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                               c_dp, ctypes.c_int64]),
    "gpimhip_dist_update": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, c_dp, ctypes.c_int64,
                                           ctypes.c_int32, ctypes.c_int32]),
    "gpimhip_dist_solve_update": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, c_dp,
                                                 ctypes.c_int64, ctypes.c_int64, c_dp, ctypes.c_int64, c_dp,
                                                 ctypes.c_int32]),
    "gpimhip_dist_solve_update2": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, c_dp,
                                                  ctypes.c_int64, ctypes.c_int64, c_dp, ctypes.c_int64, c_dp,
                                                  ctypes.c_int32, ctypes.c_int32]),
    "gpimhip_dist_kmat_cols": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, ctypes.c_int64, c_dp,
                                              ctypes.c_int64, ctypes.c_int64, c_dp, ctypes.c_int64]),
    "gpimhip_dist_kinv_update": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, c_dp,
                                                ctypes.c_int64, c_dp, ctypes.c_int64]),
    "gpimhip_dist_kinv_update_n": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_dp,
                                                  ctypes.c_int64, c_dp, ctypes.c_int64]),
    "gpimhip_dist_grad_sums": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, ctypes.c_int64, c_dp,
                                              c_dp, ctypes.c_int64, c_dp, c_dp]),
    "gpimhip_dist_finalize": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), ctypes.c_int64, c_dp, c_dp,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int32, c_dp,
                                             c_dp, c_dp]),
    "gpimhip_dist_finalize_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), ctypes.c_int64, c_dp, c_dp,
                                                 c_dp, ctypes.c_double, ctypes.c_int32, c_dp, c_dp, c_dp]),
    "gpimhip_set_precision": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "gpimhip_set_reflection_shard": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "gpimhip_refl_sums": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, c_dp, ctypes.c_int64, ctypes.c_int32,
                                         c_dp, c_dp]),
    "gpimhip_set_reflection": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_double), c_dp,
                                              ctypes.c_int64, ctypes.c_int64]),
    "gpimhip_acquire_exact": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ModelStruct), c_dp, c_dp, ctypes.c_int64,
                                             c_dp, c_dp, ctypes.c_int64, c_dp, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_double, ctypes.c_double, c_dp, c_dp, c_dp, c_dp]),
    "gpimhip_thin_batch": (ctypes.c_int, [ctypes.c_void_p, c_dp, c_dp, ctypes.c_int32, ctypes.c_int32, c_dp,
                                          ctypes.c_double, ctypes.c_int32, c_dp, c_dp]),
    "gpimhip_dist_vec_forward": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                c_dp, c_dp, c_dp, c_dp]),
    "gpimhip_dist_vec_backward": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                 c_dp, c_dp, c_dp, c_dp]),
    "gpimhip_matvec_t": (ctypes.c_int, [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_dp,
                                        c_dp]),
    "gpimhip_step_plan_host": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                              ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
}
EXPORTS = tuple(_PROTOS)


def load():
    """Loads libgpimhip.so (no GPU needed for the load itself) and types its symbols."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "gpim_amd: %s is missing -- build it with `python -m gpim_amd._build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError("gpim_amd: cannot load %s: %s. There is no CPU fallback." % (LIB_PATH, e))
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    import atexit
    atexit.register(_shutdown)
    return lib


import weakref

_live = weakref.WeakSet()      # every open Handle (created at import: no lazy-initialisation race between threads)


def _shutdown():
    """Interpreter exit: close the handles that are still alive, then release the library's helper streams while the
    HIP runtime is still up (see gpimhip_shutdown)."""
    import gc
    gc.collect()
    for h in list(_live):
        try:
            h.close()
        except Exception:
            pass
    try:
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            _lib.gpimhip_shutdown()
    except Exception:
        pass


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("gpim_amd: no HIP device visible to torch; the MI355X engine has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class NotPositiveDefiniteError(torch.linalg.LinAlgError if hasattr(torch.linalg, "LinAlgError") else RuntimeError):
    """Raised where torch.linalg.cholesky would raise in the reference (gpr.py:192,248)."""


def check(rc):
    if rc == OK:
        return
    msg = load().gpimhip_last_error().decode()
    if rc == E_NOT_PD:
        raise NotPositiveDefiniteError("linalg.cholesky: " + msg)
    if rc == E_BADARG:
        raise ValueError("gpimhip: bad argument. " + msg)
    if rc == E_NOMEM:
        raise MemoryError("gpimhip: " + msg)
    raise RuntimeError("gpimhip: HIP error. " + msg)


class Handle:
    """Owns one gpimhip_handle bound to torch's current stream on the current device."""

    def __init__(self, precision="double", stream=None):
        self.device = require_gpu()
        lib = load()
        h = ctypes.c_void_p()
        stream = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        check(lib.gpimhip_create(ctypes.byref(h), self.device.index, ctypes.c_void_p(stream)))
        self._h = h
        self.lib = lib
        _live.add(self)
        self.precision = precision
        if precision == "single":       # N x N matrices and the O(N^3) products in float (gpimhip_set_precision)
            check(lib.gpimhip_set_precision(h, 32))

    @property
    def h(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.gpimhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Device address of a CUDA tensor (or None): contiguous, or a row-major matrix with padded rows (unit column stride;
    the callee gets the row stride as its leading dimension)."""
    if t is None:
        return None
    assert t.is_cuda and (t.is_contiguous() or (t.dim() == 2 and t.stride(1) == 1))
    return ctypes.c_void_p(t.data_ptr())

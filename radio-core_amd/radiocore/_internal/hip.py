"""ctypes binding of librcfm.so (include/rcfm.h) plus the torch plumbing around it.

This module is the only place that touches the shared library.  There is no
CPU fallback: if the library is missing or no HIP device is present, loading
raises, and every class of the package fails with that error.

PyTorch is used for device memory and streams only (torch.empty on the GPU,
torch.cuda.current_stream); no torch operator computes any part of the path.
"""

import ctypes
import os
import sys
import threading
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCFM_LIB") or os.path.join(os.path.dirname(_HERE), "_lib", "librcfm.so")

RCFM_FM, RCFM_MFM, RCFM_WBFM = 0, 1, 2
RCFM_OPT_LDS_CHAIN, RCFM_OPT_FUSED_TILES, RCFM_OPT_PHASE_LINK, RCFM_OPT_NARROW_TILES, RCFM_OPT_STATE_FENCE = 1, 2, 3, 4, 5   # rcfm_demod_set_option
RCFM_OPT_PILOT_CHAIN, RCFM_OPT_DECIM_TILE, RCFM_OPT_LDS_DEEMPH, RCFM_OPT_PILOT_BLOCKED, RCFM_OPT_GRAPH = 6, 7, 8, 9, 10
RCFM_TUNER_OPT_NARROW_TILES, RCFM_TUNER_OPT_ALIGNED_PLAN = 1, 2                                                                                                # rcfm_tuner_set_option

_ERR_SIZE, _ERR_INDEX, _ERR_RUNTIME, _ERR_ARG, _ERR_STATE = -1, -2, -3, -4, -5

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t
_dbl = ctypes.c_double
_fp = ctypes.POINTER(ctypes.c_float)

# name -> argtypes; every function returns int except rcfm_last_error.
SIGNATURES = {
    "rcfm_version": [],
    "rcfm_device_count": [ctypes.POINTER(_i)],
    "rcfm_malloc": [ctypes.POINTER(_vp), _sz],
    "rcfm_free": [_vp],
    "rcfm_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "rcfm_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "rcfm_memcpy_d2d": [_vp, _vp, _sz, _vp],
    "rcfm_stream_sync": [_vp],
    "rcfm_stream_create": [ctypes.POINTER(_vp)],
    "rcfm_stream_destroy": [_vp],
    "rcfm_arena_create": [_sz, ctypes.POINTER(_vp)],
    "rcfm_arena_adopt": [_vp, _sz, ctypes.POINTER(_vp)],
    "rcfm_arena_bind": [_vp],
    "rcfm_arena_stats": [_vp, ctypes.POINTER(_sz), ctypes.POINTER(_sz), ctypes.POINTER(_sz)],
    "rcfm_arena_destroy": [_vp],
    "rcfm_tuner_create": [_i64, _i, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_vp)],
    "rcfm_tuner_load": [_vp, _vp, _vp],
    "rcfm_tuner_shard": [_vp, _i, _i],
    "rcfm_tuner_run": [_vp, _i, _i, _vp, _vp],
    "rcfm_tuner_spectrum": [_vp, ctypes.POINTER(_vp)],
    "rcfm_tuner_spectrum_layout": [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    "rcfm_tuner_attach_spectrum": [_vp, _vp, _i, _i],
    "rcfm_tuner_window": [_vp, _i, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    "rcfm_tuner_window_layout": [_vp, _i, _i, ctypes.POINTER(_i64), ctypes.POINTER(_i64)],
    "rcfm_tuner_attach_window": [_vp, _vp, _i, _i],
    "rcfm_tuner_adopt": [_vp, _i, _i, _vp],
    "rcfm_tuner_set_option": [_vp, _i, _i],
    "rcfm_tuner_destroy": [_vp],
    "rcfm_demod_create": [_i, _i, _i, _i, _dbl, _i, ctypes.POINTER(_vp)],
    "rcfm_demod_run": [_vp, _i, _i, _vp, _vp, _vp],
    "rcfm_demod_reset_state": [_vp, _vp],
    "rcfm_demod_get_state": [_vp, _fp, _vp],
    "rcfm_demod_set_state": [_vp, _fp, _vp],
    "rcfm_demod_get_taps": [_vp, _fp, _fp],
    "rcfm_demod_bind_state": [_vp, _vp, _i, _i, _vp],
    "rcfm_demod_set_option": [_vp, _i, _i],
    "rcfm_demod_get_option": [_vp, _i, ctypes.POINTER(_i)],
    "rcfm_demod_destroy": [_vp],
    "rcfm_pipeline_run": [_vp, _vp, _i, _i, _vp, _vp],
    "rcfm_host_register": [_vp, _sz],
    "rcfm_host_unregister": [_vp],
    "rcfm_feeder_create": [_sz, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp)],
    "rcfm_feeder_submit": [_vp, _vp],
    "rcfm_feeder_acquire": [_vp, _vp, ctypes.POINTER(_vp)],
    "rcfm_feeder_release": [_vp, _vp],
    "rcfm_feeder_copied": [_vp, ctypes.POINTER(ctypes.c_uint64)],
    "rcfm_feeder_destroy": [_vp],
    "rcfm_comm_unique_id": [_vp],
    "rcfm_comm_init_rank": [_i, _i, _vp, ctypes.POINTER(_vp)],
    "rcfm_gather_audio": [_vp, _i, _vp, _sz, _vp, _vp],
    "rcfm_comm_group_start": [_vp],
    "rcfm_send_bins": [_vp, _i, _vp, _sz, _vp],
    "rcfm_recv_bins": [_vp, _i, _vp, _sz, _vp],
    "rcfm_comm_group_end": [_vp],
    "rcfm_comm_destroy": [_vp],
    "rcfm_resampler_create": [_i, _i, _i, _i, ctypes.POINTER(_vp)],
    "rcfm_resampler_run": [_vp, _vp, _vp, _vp],
    "rcfm_resampler_destroy": [_vp],
    "rcfm_filtfilt": [_i, _i, _fp, _i, _vp, _vp, _vp],
    "rcfm_lfilter_fir": [_i, _i, _fp, _i, _vp, _vp, _vp, _vp],
    "rcfm_hilbert": [_i, _i, _vp, _vp, _vp],
    "rcfm_pll_phase": [_vp, _sz, _dbl, _i, _vp, _vp],
    "rcfm_discriminator": [_i, _i, _vp, _vp, _vp],
    "rcfm_fft_describe": [_i64, _i, _vp],
    "rcfm_fft_describe_plan": [_i64, ctypes.POINTER(_i64), _i, _i, _vp],
    "rcfm_fft_c2c": [_i64, _i, _i, _vp, _vp, _vp],
    "rcfm_fft_c2c_plan": [_i64, ctypes.POINTER(_i64), _i, _i, _i, _i, _vp, _vp, _vp],
    "rcfm_fft_c2c_rocfft": [_i64, _i, _i, _vp, _vp, _vp],
    "rcfm_profile_stage_count": [],
    "rcfm_profile_enable": [ctypes.c_uint64],
    "rcfm_profile_reset": [],
    "rcfm_profile_read": [_i, ctypes.POINTER(_dbl), ctypes.POINTER(_i64)],
}



class FftPass(ctypes.Structure):
    """rcfm_fft_pass (include/rcfm.h)."""
    _fields_ = [("L", ctypes.c_int32), ("nstages", ctypes.c_int32), ("radix", ctypes.c_int32 * 8),
                ("n_o1", _i64), ("n_o2", _i64), ("n_inner", _i64),
                ("in_o1", _i64), ("in_o2", _i64), ("in_i", _i64), ("in_l", _i64),
                ("out_o1", _i64), ("out_o2", _i64), ("out_i", _i64), ("out_k", _i64),
                ("tw_o1", _i64), ("tw_o2", _i64), ("tw_i", _i64),
                ("has_twiddle", ctypes.c_int32), ("load_along_l", ctypes.c_int32), ("in_t", _i64), ("out_t", _i64),
                ("flat_outer", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


class FftPlan(ctypes.Structure):
    """rcfm_fft_plan (include/rcfm.h)."""
    _fields_ = [("n", _i64), ("npass", ctypes.c_int32), ("fine_bits", ctypes.c_int32), ("tmp_stride", _i64),
                ("passes", FftPass * 4)]


_lib = None
_torch = None


def load_library(path=LIB_PATH, strict=True):
    """dlopen librcfm.so and declare every prototype of include/rcfm.h + rcfm_tools.h.  strict=False (tools/ab_libs.py: an
    older build beside the current one) skips entry points that build does not export."""
    if not os.path.exists(path):
        raise ImportError(
            "librcfm.so not found at %s -- build it with `make -C radio-core_amd` "
            "(or __graft_entry__.build()); there is no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        if not strict and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _i
    for name in ("rcfm_last_error", "rcfm_profile_stage_name"):
        getattr(lib, name).restype = ctypes.c_char_p
    lib.rcfm_last_error.argtypes = []
    lib.rcfm_profile_stage_name.argtypes = [_i]
    return lib


def lib():
    """The process-wide librcfm handle.

    PyTorch's ROCm wheel bundles its own libamdhip64; librcfm.so needs the system one.  Two
    HIP runtimes in one process fight over the device (whichever initialises second reports
    "no ROCm-capable device").  Importing torch first makes the dynamic linker resolve
    librcfm's libamdhip64.so.7 dependency to the copy torch already loaded: one runtime.
    """
    global _lib
    if _lib is None:
        try:
            import torch as _t  # noqa: F401  (load order matters, see above)
        except ImportError:
            pass
        _lib = load_library()
    return _lib


def torch():
    """torch with a usable HIP device, or a loud failure."""
    global _torch
    if _torch is None:
        import torch as _t
        if not _t.cuda.is_available():
            raise RuntimeError("radiocore (MI355X build) needs a HIP device; none is visible "
                               "and there is no CPU fallback")
        _torch = _t
    return _torch


def check(status):
    """Map an rcfm_status to the exception the reference raises at that point."""
    if status == 0:
        return
    msg = lib().rcfm_last_error().decode("utf-8", "replace")
    if status == _ERR_SIZE:
        raise ValueError(msg or "input_sig size and input_size mismatch")
    if status == _ERR_INDEX:
        raise IndexError(msg or "list index out of range")
    if status == _ERR_ARG:
        raise ValueError(msg)
    raise RuntimeError("librcfm: %s (status %d)" % (msg, status))


def stream():
    return _vp(torch().cuda.current_stream().cuda_stream)


def ptr(t):
    return _vp(t.data_ptr())


def to_device(x, dtype=None):
    """Host array / sequence / tensor -> contiguous device tensor (H2D if needed)."""
    t = torch()
    if isinstance(x, t.Tensor):
        out = x
    else:
        a = np.asarray(x)
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        elif a.dtype == np.complex128:
            a = a.astype(np.complex64)
        out = t.from_numpy(np.ascontiguousarray(a))
    if dtype is not None and out.dtype != dtype:
        out = out.to(dtype)
    return out.to("cuda", non_blocking=False).contiguous()


def to_host(x):
    return x.cpu().numpy()


def empty(shape, dtype):
    return torch().empty(shape, dtype=dtype, device="cuda")


def float_array(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


class Handle:
    """Owns one opaque librcfm handle and destroys it with the given function."""

    def __init__(self, value, destroy):
        self.value = value
        self._destroy = destroy

    def __del__(self):
        if not self.value:
            return
        try:
            status = self._destroy(self.value)
        except Exception as e:        # interpreter shutdown: the library or ctypes may already be gone
            if not sys.is_finalizing():
                warnings.warn("librcfm handle could not be destroyed: %r" % (e,), ResourceWarning)
            return
        self.value = None
        if status != 0 and not sys.is_finalizing():
            warnings.warn("librcfm destroy returned status %d" % status, ResourceWarning)


# ---- placement (rcfm_arena_*) ----------------------------------------------------------------------------------------

_arena_tls = threading.local()      # like librcfm's own binding (rcfm_arena_bind), the open `with Arena` is per thread


def _arena_stack():
    st = getattr(_arena_tls, "stack", None)
    if st is None:
        st = _arena_tls.stack = []
    return st


def current_arena():
    """The Arena whose `with` block is open on THIS thread (None: hipMalloc per workspace)."""
    st = _arena_stack()
    return st[-1] if st else None


class Arena:
    """ONE placement draw for every large workspace of the handles built inside it (include/rcfm.h, rcfm_arena_*;
    profiles/r04_k_placement.md: where hipMalloc puts a multi-GB workspace moves a cfg4 buffer by 1.5 - 4 %).

        with radiocore.tools.Arena(20 << 30) as arena:      # one 20 GiB device block (0: 1 GiB blocks on demand)
            tuner = Tuner(cuda=True); ... add_channel(..., WBFM(..., cuda=True)) ...
        # the objects remember the arena: their device handles are created lazily (first load / run) inside it

    The arena must outlive those objects (close() refuses while their handles are alive).  No reference counterpart."""

    def __init__(self, block_bytes=0, memory=None):
        """memory: a device tensor the arena lives in instead of blocks of its own (rcfm_arena_adopt; kept alive here)."""
        h = _vp()
        if memory is not None:
            check(lib().rcfm_arena_adopt(ptr(memory), _sz(memory.numel() * memory.element_size()), ctypes.byref(h)))
        else:
            check(lib().rcfm_arena_create(_sz(int(block_bytes)), ctypes.byref(h)))
        self._memory = memory
        self._handle = Handle(h, lib().rcfm_arena_destroy)

    @property
    def value(self):
        return self._handle.value

    def __enter__(self):
        _arena_stack().append(self)
        return self

    def __exit__(self, *exc):
        _arena_stack().pop()
        return False

    def stats(self):
        r, u, n = _sz(), _sz(), _sz()
        check(lib().rcfm_arena_stats(self._handle.value, ctypes.byref(r), ctypes.byref(u), ctypes.byref(n)))
        return {"reserved_bytes": int(r.value), "used_bytes": int(u.value), "live_pieces": int(n.value)}

    def close(self):
        if self._handle.value:
            check(lib().rcfm_arena_destroy(self._handle.value))
            self._handle.value = None


_binding = threading.local()     # what this thread has bound through `bound` (librcfm's binding is per thread too)


class bound:
    """`with bound(arena):` -- tuner / demodulator handles created by librcfm inside the block belong to `arena`
    (None: no-op).  Nests: leaving the block restores the binding the thread had when it entered."""

    def __init__(self, arena):
        self._arena = arena
        self._prev = None

    def __enter__(self):
        if self._arena is not None:
            self._prev = getattr(_binding, "arena", None)
            check(lib().rcfm_arena_bind(self._arena.value))
            _binding.arena = self._arena
        return self

    def __exit__(self, *exc):
        if self._arena is not None:
            prev = self._prev
            if prev is not None and not prev.value:      # closed in the meantime
                prev = None
            check(lib().rcfm_arena_bind(prev.value if prev is not None else None))
            _binding.arena = prev
        return False

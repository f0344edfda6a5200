"""ctypes binding of ``libraft_hip.so`` (C ABI: include/raft_hip.h).

There is no CPU fallback: if the shared library cannot be loaded (and cannot be built
because hipcc is absent) importing the device path raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_LOCK = threading.Lock()
_LIB = None
ABI_VERSION = 219     # include/raft_hip.h RAFT_HIP_VERSION: the ctypes mirrors below describe THIS revision of the structs

c_float_p = C.c_void_p      # raw device pointers travel as void*
c_i64_p = C.POINTER(C.c_int64)
c_int_p = C.POINTER(C.c_int)


class ConvWeights(C.Structure):
    _fields_ = [('wp', C.c_void_p), ('bias', C.c_void_p), ('npad', C.c_int)]


class BasicUpdateWeights(C.Structure):
    _fields_ = [(n, ConvWeights) for n in (
        'convc1', 'convc2', 'convf1', 'convf2', 'conv',
        'gru_zr1', 'gru_q1', 'gru_zr2', 'gru_q2',
        'fh1_mask0', 'fh2', 'mask2', 'gru_ctx1', 'gru_ctx2',
        'convc2_w', 'convf2_w', 'conv_w', 'fh1_mask0_w',
        'gru_zr1_w', 'gru_q1_w', 'gru_zr2_w', 'gru_q2_w', 'fh1_w',
        'gru_zr1_w4', 'gru_q1_w4', 'gru_zr2_w4', 'gru_q2_w4', 'convc1_f', 'gru_ctx1_w4', 'gru_ctx2_w4',
        'convc2_w44', 'conv_w44', 'fh1_mask0_w44', 'fh1_w44', 'convf2_w44')]


class SmallUpdateWeights(C.Structure):
    _fields_ = [(n, ConvWeights) for n in (
        'convc1', 'convf1', 'convf2', 'conv', 'gru_zr', 'gru_q', 'fh1', 'fh2',
        'conv_w', 'gru_zr_w', 'gru_q_w', 'fh1_w')]


class EncoderWeights(C.Structure):
    _fields_ = [('c0', C.c_int), ('c1', C.c_int), ('c2', C.c_int), ('c3', C.c_int), ('cout', C.c_int),
                ('norm', C.c_int), ('conv1', ConvWeights), ('block', (ConvWeights * 3) * 6), ('conv2', ConvWeights),
                ('in_gamma', C.c_void_p * 19), ('in_beta', C.c_void_p * 19), ('block_w', (ConvWeights * 2) * 6), ('block_w44', (ConvWeights * 2) * 6)]


NORM_NONE, NORM_INSTANCE, NORM_FOLDED = 0, 1, 2


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'net', 'x', 'corr', 'coords1', 'flow', 'delta', 'mask', 'ws', 'ctx')]


_P = C.c_void_p
_I = C.c_int
_SIGNATURES = {
    'raft_version': (C.c_int, []),
    'raft_error_string': (C.c_char_p, [_I]),
    'raft_set_option': (_I, [C.c_char_p, C.c_char_p]),
    'raft_get_option': (_I, [C.c_char_p, C.c_char_p, C.c_size_t]),
    'raft_set_thread_concurrency': (_I, [_I]),
    'raft_loop_ctx_create': (_I, [C.POINTER(C.c_void_p)]),
    'raft_loop_ctx_destroy': (_I, [C.c_void_p]),
    'raft_crc32c': (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_size_t]),
    'raft_corr_pyramid_layout': (_I, [_I, _I, _I, _I, c_i64_p, c_int_p, c_int_p]),
    'raft_corr_build_workspace_floats': (C.c_int64, [_I, _I, _I, _I, _I]),
    'raft_corr_build_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, c_i64_p, _P, _P]),
    'raft_corr_lookup_f32': (_I, [_P, c_i64_p, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    'raft_lookup_convc1_f32': (_I, [_P, c_i64_p, _P, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P]),
    'raft_fmap_pyramid_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'raft_corr_lookup_ondemand_f32': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    'raft_bilinear_sampler_f32': (_I, [_P, _P, C.c_int64, _I, _I, _I, _I, _P, _P]),
    'raft_coords_grid_f32': (_I, [_P, _I, _I, _I, _P]),
    'raft_upsample_convex_f32': (_I, [_P, _P, _I, _I, _I, _P, _P]),
    'raft_upflow8_f32': (_I, [_P, _I, _I, _I, _P, _P]),
    'raft_stream_copy_f32': (_I, [_P, _P, C.c_int64, _P]),
    'raft_mfma_probe_f32': (_I, [_P, _I, _I, _P]),
    'raft_metrics_workspace_doubles': (C.c_int64, []),
    'raft_flow_metrics_f32': (_I, [_P, _P, _P, C.c_int64, C.c_float, _P, _P, _P]),
    'raft_sequence_loss_f32': (_I, [_P, _P, _P, C.c_int64, _I, C.c_int64, C.c_double, C.c_float, _P, _P, _P]),
    'raft_sequence_loss_grad_f32': (_I, [_P, _P, _P, C.c_int64, _I, C.c_int64, C.c_double, C.c_float, C.c_float, _P, _P]),
    'raft_corr_lookup_backward_f32': (_I, [_P, c_i64_p, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'raft_relu_backward_f32': (_I, [_P, _P, _P, C.c_int64, _P]),
    'raft_pack_train_conv_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_double), _I, _I, _I, _P, _P, _P]),
    'raft_conv2d_wgrad_workspace_floats': (C.c_int64, [_I, _I, _I, _I, _I, _I, _I]),
    'raft_conv2d_wgrad_f32': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'raft_conv2d_wgrad_multi_f32': (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'raft_conv7x7_c2_f32': (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    'raft_conv7x7_c2_wgrad_workspace_floats': (C.c_int64, [_I]),
    'raft_conv7x7_c2_backward_f32': (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'raft_gru_gate_zr_f32': (_I, [_P, _P, _I, C.c_int64, _P, _P, _P, _P]),
    'raft_gru_gate_q_f32': (_I, [_P, _P, _P, C.c_int64, _P, _P, _P]),
    'raft_gru_gate_q_backward_f32': (_I, [_P, _P, _P, _P, C.c_int64, _P, _P, _P, _P]),
    'raft_gru_gate_r_backward_f32': (_I, [_P, _P, _P, C.c_int64, _P, _P, _P]),
    'raft_axpby_f32': (_I, [C.c_float, _P, C.c_float, _P, _P, C.c_int64, _P]),
    'raft_upsample_convex_backward_workspace_floats': (C.c_int64, [_I, _I, _I]),
    'raft_upsample_convex_backward_f32': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    'raft_sumsq_workspace_doubles': (C.c_int64, []),
    'raft_sumsq_f32': (_I, [_P, C.c_int64, _I, _P, _P, _P]),
    'raft_adamw_step_f32': (_I, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_float, _P]),
    'raft_sumsq_multi_workspace_doubles': (C.c_int64, [_I]),
    'raft_sumsq_multi_f32': (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _I, _P, _P, _P]),
    'raft_adamw_step_multi_f32': (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_int64), _I, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_float, _P]),
    'raft_gemm_f32': (_I, [_P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int64, _I, _I, _I, _I, _I,
                          C.c_float, C.c_float, _P]),
    'raft_corr_build_backward_f32': (_I, [_P, _P, _P, c_i64_p, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'raft_prepare_state_backward_f32': (_I, [_P, _P, _P, _P, _I, _I, C.c_int64, _P, _P]),
    'raft_norm_workspace_doubles': (C.c_int64, [_I, _I]),
    'raft_norm_forward_f32': (_I, [_P, _I, C.c_int64, _I, _P, _P, C.c_float, _I, _P, _P, _P, _P, _P, _P]),
    'raft_norm_backward_f32': (_I, [_P, _P, _P, _P, _P, _I, C.c_int64, _I, _P, _P, _P, _P, _P]),
    'raft_axpby_relu_f32': (_I, [C.c_float, _P, C.c_float, _P, _P, C.c_int64, _P]),
    'raft_f32_to_bf16': (_I, [_P, _P, C.c_int64, _P]),
    'raft_bf16_to_f32': (_I, [_P, _P, C.c_int64, _P]),
    'raft_dropout_f32': (_I, [_P, C.c_int64, C.c_float, C.c_uint64, _P, _P, _P]),
    'raft_dropout_backward_f32': (_I, [_P, _P, C.c_int64, C.c_float, _P, _P]),
    'raft_upflow8_backward_f32': (_I, [_P, _I, _I, _I, _P, _P]),
    'raft_conv2d_f32': (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I,
                             C.c_float, _P, _I, _P]),
    'raft_conv2d_winograd_f32': (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P]),
    'raft_conv2d_winograd4_f32': (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P]),
    'raft_conv1d_winograd_f32': (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P]),
    'raft_conv1d_winograd4_f32': (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, C.c_float, _P, _I, _P]),
    'raft_update_workspace_floats': (C.c_int64, [_I, _I, _I]),
    'raft_prepare_state_f32': (_I, [_P, _I, _I, _I, C.POINTER(State), _P]),
    'raft_gru_context_f32': (_I, [C.POINTER(BasicUpdateWeights), _I, _I, _I, C.POINTER(State), _P]),
    'raft_update_basic_f32': (_I, [C.POINTER(BasicUpdateWeights), _I, _I, _I, C.POINTER(State), _P]),
    'raft_iterate_basic_f32': (_I, [C.POINTER(BasicUpdateWeights), _P, c_i64_p, _I, _I, _I, _I,
                                    C.POINTER(State), _P, _P]),
    'raft_iterate_basic_overlap_f32': (_I, [C.POINTER(BasicUpdateWeights), _P, c_i64_p, _I, _I, _I, _I,
                                            C.POINTER(State), _P, _P, _P, _P, _P]),
    'raft_iterate_basic_ondemand_f32': (_I, [C.POINTER(BasicUpdateWeights), _P, _P, _I, _I, _I, _I, _I,
                                        C.POINTER(State), _P, _P, _P, _P, _P]),
    'raft_iterate_basic_final_f32': (_I, [C.POINTER(BasicUpdateWeights), _P, c_i64_p, _I, _I, _I, _I,
                                          C.POINTER(State), _P, _P, _P, _P, _P]),
    'raft_iterate_basic_timed_f32': (_I, [C.POINTER(BasicUpdateWeights), _P, c_i64_p, _I, _I, _I, _I,
                                          C.POINTER(State), _P, _P, C.POINTER(C.c_float)]),
    'raft_encoder_workspace_floats': (C.c_int64, [C.POINTER(EncoderWeights), _I, _I, _I]),
    'raft_encoder_f32': (_I, [C.POINTER(EncoderWeights), _P, _I, _I, _I, _I, _P, _P, _P]),
    'raft_encoder_pair_f32': (_I, [C.POINTER(EncoderWeights), _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'raft_small_update_workspace_floats': (C.c_int64, [_I, _I, _I]),
    'raft_prepare_state_small_f32': (_I, [_P, _I, _I, _I, C.POINTER(State), _P]),
    'raft_update_small_f32': (_I, [C.POINTER(SmallUpdateWeights), _I, _I, _I, C.POINTER(State), _P]),
    'raft_iterate_small_f32': (_I, [C.POINTER(SmallUpdateWeights), _P, c_i64_p, _I, _I, _I, _I,
                                    C.POINTER(State), _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libraft_hip.so')


def load_library():
    """Load and type the library.  When hipcc is available the build is refreshed first (a digest of sources + flags
    makes that a no-op for an up-to-date .so), so a stale library never meets newer struct mirrors.  When the refresh
    fails (no hipcc, read-only tree, compile error) an existing .so is used only if its build stamp (lib/build.sha256,
    shipped beside it) matches the sources on disk; a .so WITHOUT a stamp cannot be verified and is refused unless
    RAFT_ALLOW_UNVERIFIED_LIB=1; a stamp that names other sources is refused unless RAFT_ALLOW_STALE_LIB=1.
    Either way ``raft_version()`` must equal ``ABI_VERSION``."""
    global _LIB
    if _LIB is not None:                       # fast path: no lock once loaded
        return _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        path = library_path()
        from . import build as _build
        try:
            _build.build_library(verbose=False)
        except Exception as exc:   # noqa: BLE001
            if not os.path.exists(path):
                raise RuntimeError(
                    f'libraft_hip.so is missing at {path} and could not be built ({exc}); '
                    'the RAFT device path has no CPU fallback') from exc
            # A library exists but the refresh failed (compile error, no hipcc, read-only tree).  It may only be used
            # if it was built from exactly these sources; anything else would run old kernels under new tests.
            built = _build.built_digest()
            unknown = built is None                          # prebuilt .so without its stamp: nothing to compare
            stale = (not unknown) and built != _build.source_digest()
            if stale and os.environ.get('RAFT_ALLOW_STALE_LIB') != '1':
                raise RuntimeError(
                    f'{path} was built from different sources than the ones on disk and rebuilding failed: {exc}.  '
                    'Fix the build, or set RAFT_ALLOW_STALE_LIB=1 to load the old library knowingly') from exc
            if unknown and '1' not in (os.environ.get('RAFT_ALLOW_UNVERIFIED_LIB'), os.environ.get('RAFT_ALLOW_STALE_LIB')):
                raise RuntimeError(
                    f'{path} carries no build stamp (lib/build.sha256), so it cannot be checked against the sources on disk, and '
                    f'rebuilding failed: {exc}.  Fix the build, or set RAFT_ALLOW_UNVERIFIED_LIB=1 to load it knowingly') from exc
            import warnings
            warnings.warn(f'libraft_hip.so could not be refreshed ({exc}); loading the existing '
                          f'{"STALE " if stale else ("UNVERIFIED (no build stamp) " if unknown else "up-to-date ")}library at {path}',
                          RuntimeWarning, stacklevel=2)
        try:
            lib = C.CDLL(path)
        except OSError as exc:
            raise RuntimeError(f'cannot load {path}: {exc}; the RAFT device path has no CPU fallback') from exc
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError => symbol missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.raft_version() != ABI_VERSION:
            raise RuntimeError(f'{path} reports ABI {lib.raft_version()}, this binding mirrors ABI {ABI_VERSION}: '
                               'rebuild the library (python -m tf_raft_amd.build --force)')
        _LIB = lib
        return lib


def set_option(name: str, value) -> None:
    """``raft_set_option``: ``value`` None = load-time (environment) state, '' = built-in default."""
    v = None if value is None else str(value).encode()
    check(load_library().raft_set_option(name.encode(), v), f'set_option {name}')


class thread_concurrency:
    """``with thread_concurrency(n):`` -- ``raft_set_thread_concurrency`` for the enclosed launches of the calling thread (n
    independent launch sequences share the device: shapes for CU-time instead of latency), restored on exit."""

    def __init__(self, n: int):
        self.n, self.prev = int(n), 1

    def __enter__(self):
        self.prev = load_library().raft_set_thread_concurrency(self.n)
        return self

    def __exit__(self, *exc):
        load_library().raft_set_thread_concurrency(self.prev)
        return False


def get_option(name: str) -> str:
    buf = C.create_string_buffer(256)
    check(load_library().raft_get_option(name.encode(), buf, 256), f'get_option {name}')
    return buf.value.decode()


def check(rc: int, what: str = '') -> None:
    """Raise on a nonzero return code: ValueError for argument errors, RuntimeError for HIP errors."""
    if rc == 0:
        return
    msg = load_library().raft_error_string(rc).decode()
    text = f'{what}: {msg} (rc={rc})' if what else f'{msg} (rc={rc})'
    if rc < 0:
        raise ValueError(text)
    raise RuntimeError(text)

"""ctypes binding of the C ABI in include/iaf_hip.h (built by iaf_amd/build.py into
iaf_amd/_lib/libiaf_hip.so).  There is NO CPU fallback: if the library is missing or a call
fails, this module raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IAF_HIP_LIB") or os.path.join(_HERE, "_lib", "libiaf_hip.so")   # env override: dev experiments

IAF_OK = 0
IAF_ERR_NULL = -1
IAF_ERR_SHAPE = -2
IAF_ERR_NOT_MULTIPLE = -3
IAF_ERR_NOT_PREPARED = -4
IAF_ERR_WORKSPACE = -5
IAF_ERR_UNSUPPORTED = -6
IAF_ERR_EXCHANGE = -7
IAF_ERR_CAPTURE_SLOTS = -8
IAF_ERR_RANGE = -9
IAF_PRECISION_F32 = 0
IAF_PRECISION_BF16X3 = 1
IAF_PRECISION_F16X2 = 2
IAF_COMM_ID_BYTES = 128
IAF_PACK_F32 = 1
IAF_PACK_BF16X3 = 2
IAF_PACK_F16X2 = 4
IAF_ABI_VERSION = 8                # must equal the library's iaf_abi_version(): a stale libiaf_hip.so is rejected
IAF_VARIANT_TF = 0
IAF_VARIANT_THEANO = 1
IAF_VARIANT_THEANO_FLIPMASK = 2

_c_float_p = ctypes.c_void_p      # device pointers travel as integers (tensor.data_ptr())
_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/iaf_hip.h declares
SIGNATURES = {
    "iaf_abi_version": (ctypes.c_int, []),
    "iaf_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "iaf_device_count": (ctypes.c_int, []),
    "iaf_stack_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_stack_destroy": (ctypes.c_int, [_vp]),
    "iaf_stack_prepare": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "iaf_prep_batch_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int]),
    "iaf_prep_batch_run": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "iaf_prep_batch_destroy": (ctypes.c_int, [_vp]),
    "iaf_stack_workspace_bytes": (ctypes.c_size_t, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_ar_multiconv2d_forward": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "iaf_step_forward": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "iaf_step_inverse": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp,
                                        ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)]),
    "iaf_step_inverse_device": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_float, ctypes.c_int, _vp, _vp]),
    "iaf_stack_set_defer_weightnorm": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_wn_bwd_batch_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int]),
    "iaf_wn_bwd_batch_run": (ctypes.c_int, [_vp] + [ctypes.POINTER(_vp)] * 5 + [_vp]),
    "iaf_wn_bwd_batch_destroy": (ctypes.c_int, [_vp]),
    "iaf_conv3x3_set_defer_weightnorm": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_conv3x3_wn_bwd_batch_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int]),
    "iaf_conv3x3_wn_bwd_batch_run": (ctypes.c_int, [_vp] + [ctypes.POINTER(_vp)] * 5 + [_vp]),
    "iaf_conv3x3_wn_bwd_batch_destroy": (ctypes.c_int, [_vp]),
    "iaf_stack_set_training": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_stack_train_workspace_bytes": (ctypes.c_size_t, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_step_forward_train": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "iaf_step_backward": (ctypes.c_int, [_vp] + [_c_float_p] * 8 + [ctypes.POINTER(_vp)] * 5 + [ctypes.c_int] * 3 +
                          [_vp, ctypes.c_size_t, _vp]),
    "iaf_posterior_block_forward_train": (ctypes.c_int, [_vp] + [_c_float_p] * 9 + [ctypes.c_float] + [_c_float_p] * 3 +
                                          [ctypes.c_int] * 3 + [_vp, ctypes.c_size_t, _vp]),
    "iaf_posterior_block_backward": (ctypes.c_int, [_vp] + [_c_float_p] * 7 + [ctypes.c_float] + [_c_float_p] * 8 +
                                     [ctypes.POINTER(_vp)] * 5 + [ctypes.c_int] * 3 + [_vp, ctypes.c_size_t, _vp]),
    "iaf_adamax_ema_step": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_size_t] + [ctypes.c_float] * 6 + [_vp]),
    "iaf_posterior_block_forward": (ctypes.c_int, [_vp] + [_c_float_p] * 9 + [ctypes.c_float] + [_c_float_p] * 4 +
                                    [ctypes.c_int] * 3 + [_vp, ctypes.c_size_t, _vp]),
    "iaf_gaussian_sample": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_size_t, _vp]),
    "iaf_gaussian_logps": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_size_t, _vp]),
    "iaf_compute_lowerbound": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int, ctypes.c_int, _vp]),
    "iaf_lowerbound_stream_init": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, _vp]),
    "iaf_lowerbound_stream_update": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int, ctypes.c_int, _vp]),
    "iaf_lowerbound_stream_finalize": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int, ctypes.c_int, _vp]),
    "iaf_stack_set_tuning": (ctypes.c_int, [_vp] + [ctypes.c_int] * 5),
    "iaf_stack_set_precision": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_stack_get_precision": (ctypes.c_int, [_vp] + [ctypes.c_int] * 4),
    "iaf_stack_set_tuning_bf3": (ctypes.c_int, [_vp] + [ctypes.c_int] * 6),
    "iaf_stack_set_fuse_first": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_stack_set_fuse_step": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_stack_set_packs": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_gaussian_sample_logsd": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_size_t, _vp]),
    "iaf_gaussian_logps_logsd": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_size_t, _vp]),
    "iaf_kl_free_bits_gate": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float, _c_float_p, _vp]),
    "iaf_up_iaf2_backward_pre": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_float] + [_c_float_p] * 4 + [ctypes.c_int] * 4 + [_vp]),
    "iaf_up_iaf2_backward_post": (ctypes.c_int, [_c_float_p] * 7 + [ctypes.c_int] * 4 + [_vp]),
    "iaf_stack_exchange_errors": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint)]),
    "iaf_stack_range_errors": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint)]),
    "iaf_stack_step_is_f16": (ctypes.c_int, [_vp] + [ctypes.c_int] * 3),
    "iaf_stack_step_exchanges": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_stack_step_pairs": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_stack_set_halo_exchange": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_stack_set_halo_exchange_debug": (ctypes.c_int, [_vp, ctypes.c_uint]),
    "iaf_comm_unique_id": (ctypes.c_int, [_vp]),
    "iaf_comm_create": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_comm_size": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "iaf_allreduce_sum_f32": (ctypes.c_int, [_vp, _c_float_p, ctypes.c_size_t, _vp]),
    "iaf_comm_destroy": (ctypes.c_int, [_vp]),
    "iaf_comm_library": (ctypes.c_char_p, []),
    "iaf_stack_step_is_fused": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_stack_autotune": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_int),
                                          ctypes.POINTER(ctypes.c_float)]),
    "iaf_stack_profile_enable": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "iaf_stack_profile_read": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_int)]),
    "iaf_step_time_layer": (ctypes.c_int, [_vp, ctypes.c_int, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_int,
                                           _vp, ctypes.POINTER(ctypes.c_float)]),
    "iaf_stack_set_debug": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "iaf_layer_work": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int] +
                       [ctypes.POINTER(ctypes.c_double)] * 3),
    "iaf_step_work": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.POINTER(ctypes.c_double)] * 3),
    "iaf_kl_free_bits": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_float, _c_float_p, _vp]),
    "iaf_kl_combine": (ctypes.c_int, [_c_float_p] * 4 + [ctypes.c_size_t, _vp]),
    "iaf_colsum": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, _vp]),
    "iaf_conv3x3_create_masked_theano": (ctypes.c_int, [ctypes.POINTER(_vp)] + [ctypes.c_int] * 4),
    "iaf_resample2": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 5 + [_vp]),
    "iaf_conv3x3_prepare_deconv": (ctypes.c_int, [_vp, _c_float_p, _c_float_p, _c_float_p, _vp]),
    "iaf_noise_from_sample": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_size_t, _vp]),
    "iaf_conv3x3_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int]),
    "iaf_conv3x3_create_masked": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_conv3x3_destroy": (ctypes.c_int, [_vp]),
    "iaf_datainit_normalize": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, _vp]),
    "iaf_discretized_logistic": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_size_t,
                                                ctypes.c_float, _vp]),
    "iaf_conv3x3_prepare": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "iaf_conv3x3_forward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.POINTER(_vp),
                                           ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, _vp]),
    "iaf_conv3x3_set_debug": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "iaf_image_to_float": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, _vp]),
    "iaf_convk_weightnorm": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "iaf_convk_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [ctypes.c_int] * 9 + [_vp]),
    "iaf_deconvk_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp] + [ctypes.c_int] * 9 + [ctypes.c_float, ctypes.c_float, _vp]),
    "iaf_tile_channels": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "iaf_sum_axpy": (ctypes.c_int, [_vp, _vp, ctypes.c_float, _vp, ctypes.c_int, _vp]),
    "iaf_discretized_logistic_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp,
                                                         ctypes.c_int, ctypes.c_size_t, ctypes.c_float, _vp]),
    "iaf_convk_wgrad": (ctypes.c_int, [_vp, _vp, _vp] + [ctypes.c_int] * 10 + [_vp]),
    "iaf_convk_weightnorm_backward": (ctypes.c_int, [_vp] * 6 + [ctypes.c_int] * 5 + [_vp]),
    "iaf_channel_sum": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "iaf_mul_elu_grad": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "iaf_axpby": (ctypes.c_int, [_vp, ctypes.c_float, _vp, ctypes.c_float, _vp, ctypes.c_size_t, _vp]),
    "iaf_affine_transform": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_size_t, _vp]),
    "iaf_clip": (ctypes.c_int, [_vp, ctypes.c_float, ctypes.c_float, _vp, ctypes.c_size_t, _vp]),
    "iaf_conv3x3_forward_stride2": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_int),
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "iaf_conv3x3_forward_deconv": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, _vp]),
    "iaf_conv3x3_prep_batch_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_int]),
    "iaf_conv3x3_prep_batch_run": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "iaf_conv3x3_prep_batch_destroy": (ctypes.c_int, [_vp]),
    "iaf_conv3x3_set_training": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_conv3x3_train_workspace_bytes": (ctypes.c_size_t, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_conv3x3_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp),
                                            ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_float, ctypes.POINTER(_vp),
                                            ctypes.POINTER(ctypes.c_int), ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "iaf_conv3x3_autotune_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp),
                                                     ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_float,
                                                     ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_int), ctypes.c_int, _vp, _vp,
                                                     _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp,
                                                     ctypes.c_size_t, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_int),
                                                     ctypes.POINTER(ctypes.c_float)]),
    "iaf_conv3x3_set_precision": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_conv3x3_range_errors": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint)]),
    "iaf_conv3x3_set_packs": (ctypes.c_int, [_vp, ctypes.c_int]),
    "iaf_conv3x3_runs_bf16x3": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_conv3x3_runs_f16x2": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "iaf_conv3x3_autotune": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.POINTER(_vp),
                                            ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_float)]),
    "iaf_conv3x3_set_tuning": (ctypes.c_int, [_vp] + [ctypes.c_int] * 4),
    "iaf_conv3x3_work": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.POINTER(ctypes.c_double)] * 2),
}


class IafHipError(RuntimeError):
    pass


class UnsupportedError(ValueError):
    """IAF_ERR_UNSUPPORTED: a shape / launch shape the gfx950 kernels do not cover (a ValueError like other bad
    arguments, but its own class so that callers -- and tests -- can tell "not covered" from a genuine argument error)"""


class ExchangeError(IafHipError):
    """IAF_ERR_EXCHANGE: a bounded wait of the halo exchange gave up in an EARLIER launch of the stack -- that launch's outputs
    carry NaN (the caller's NaN check sees it, tf_train.py:283-285); the stack has switched to the kernels that recompute their
    halo rows, so the call that raised this can simply be repeated.  ARStack.set_halo_exchange(True) re-arms the exchange."""


class RangeError(IafHipError):
    """IAF_ERR_RANGE: an operand beyond fp16's largest finite number went into the two-plane fp16 kernels ("f16x2") in an EARLIER launch
    of the stack -- that launch's outputs carry inf / NaN (the caller's NaN check sees it, tf_train.py:283-285); the stack has gone back
    to the bf16x3 kernels, so the call that raised this can be repeated (after another prepare where the stack kept only the fp16 pack).
    ARStack.set_precision("f16x2") re-arms."""


_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IafHipError("HIP engine not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` or `python -m iaf_amd.build`). There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if l.iaf_abi_version() != IAF_ABI_VERSION:
            raise IafHipError("%s has ABI version %d, this package binds version %d: rebuild it (`python -m iaf_amd.build`)"
                              % (LIB_PATH, l.iaf_abi_version(), IAF_ABI_VERSION))
        _lib = l
    return _lib


def error_string(code):
    return lib().iaf_error_string(int(code)).decode()


def check(code):
    """Map a status to the exception kind the reference raises for the same condition
    (SURVEY 8b "Errors"): its channel-divisibility asserts -> AssertionError, other bad
    arguments -> ValueError, device errors -> IafHipError."""
    if code == IAF_OK:
        return
    msg = "iaf_hip: %s (status %d)" % (error_string(code), code)
    if code == IAF_ERR_NOT_MULTIPLE:
        raise AssertionError(msg)          # tf_utils/layers.py:116
    if code == IAF_ERR_UNSUPPORTED:
        raise UnsupportedError(msg)
    if code == IAF_ERR_EXCHANGE:
        raise ExchangeError(msg)
    if code == IAF_ERR_RANGE:
        raise RangeError(msg)
    if code in (IAF_ERR_NULL, IAF_ERR_SHAPE, IAF_ERR_WORKSPACE):
        raise ValueError(msg)
    raise IafHipError(msg)

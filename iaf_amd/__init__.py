"""iaf_amd -- MI355X-native IAF posterior engine (one hot path of openai/iaf):
hand-written gfx950 HIP kernels behind a C ABI (include/iaf_hip.h), driven from Python
wrappers that carry the reference's operator signatures.  PyTorch tensors are storage only."""
from . import _capi  # noqa: F401
from ._capi import ExchangeError, IafHipError, UnsupportedError  # noqa: F401
from .layers import (ARStack, PrepBatch, ConvPrepBatch, WnBwdBatch, WNConv2d, conv2d, ar_conv2d, split, VariableStore, ar_multiconv2d, get_conv_ar_mask, get_linear_ar_mask, multiconv2d,  # noqa: F401
                     variable_scope, default_store, resample2, resize_nearest_neighbor, ar_conv2d_theano)
from .distributions import (DiagonalGaussian, discretized_logistic, compute_lowerbound, gaussian_diag_logps, logsumexp, repeat,  # noqa: F401
                            StreamingLowerBound)
from .iaf_layer import IAFPosterior, IAFLayer  # noqa: F401
from .iw_eval import IWEvaluator  # noqa: F401
from .theano_layer import CVAELayerIAF  # noqa: F401
from .model import CVAE1  # noqa: F401

__all__ = ["ExchangeError", "IafHipError", "UnsupportedError", "ARStack", "PrepBatch", "VariableStore", "ar_multiconv2d", "get_conv_ar_mask", "get_linear_ar_mask", "multiconv2d", "variable_scope",
           "default_store", "DiagonalGaussian", "compute_lowerbound", "gaussian_diag_logps", "logsumexp", "repeat",
           "StreamingLowerBound", "IAFPosterior", "IAFLayer", "IWEvaluator", "CVAELayerIAF", "CVAE1", "WNConv2d", "ConvPrepBatch", "WnBwdBatch", "conv2d", "ar_conv2d", "discretized_logistic", "split", "resample2", "resize_nearest_neighbor", "ar_conv2d_theano"]

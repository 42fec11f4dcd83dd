"""Host-side mirror of tf_utils/distributions.py:5-62 on the HIP engine (same names and
argument meaning).  Random noise is an INPUT here (parity is defined on identical eps); when the
caller passes none we draw it with torch's generator, which is outside the hot path."""
import ctypes
import math

import torch

from . import _capi
from .layers import _ptr, _stream, _check_act


def gaussian_diag_logps(mean, logvar, sample=None, noise=None):
    """distributions.py:5-10."""
    _check_act(mean, "mean")
    _check_act(logvar, "logvar", mean.shape)
    if sample is None:
        sample = DiagonalGaussian(mean, logvar, noise=noise).sample
    _check_act(sample, "sample", mean.shape)
    out = torch.empty_like(mean)
    _capi.check(_capi.lib().iaf_gaussian_logps(_ptr(mean), _ptr(logvar), _ptr(sample), _ptr(out), mean.numel(),
                                               _stream()))
    return out


def gaussian_diag_logps_logsd(mean, logsd, sample):
    """gaussian_diag_logps(mean, 2 * logsd, sample) (tf_train.py:56-57, rand.py:85-86) without materialising 2 * logsd"""
    _check_act(mean, "mean")
    _check_act(logsd, "logsd", mean.shape)
    _check_act(sample, "sample", mean.shape)
    out = torch.empty_like(mean)
    _capi.check(_capi.lib().iaf_gaussian_logps_logsd(_ptr(mean), _ptr(logsd), _ptr(sample), _ptr(out), mean.numel(), _stream()))
    return out


class DiagonalGaussian(object):
    """distributions.py:13-25."""

    def __init__(self, mean, logvar, sample=None, noise=None):
        _check_act(mean, "mean")
        _check_act(logvar, "logvar", mean.shape)
        self.mean, self.logvar = mean, logvar
        if sample is None:
            if noise is None:
                noise = torch.randn_like(mean)
            _check_act(noise, "noise", mean.shape)
            sample = torch.empty_like(mean)
            _capi.check(_capi.lib().iaf_gaussian_sample(_ptr(mean), _ptr(logvar), _ptr(noise), _ptr(sample),
                                                        mean.numel(), _stream()))
        self.sample = sample

    def logps(self, sample):
        return gaussian_diag_logps(self.mean, self.logvar, sample)


def repeat(x, n):
    """distributions.py:40-52 == repeat_interleave on axis 0 (storage op, no arithmetic)."""
    if n == 1:
        return x
    return x.repeat_interleave(n, dim=0)


def compute_lowerbound(log_pxz, sum_kl_costs, k=1):
    """distributions.py:55-62.  Inputs flat [n*k] (image-major, sample-minor) -> [n]."""
    _check_act(log_pxz, "log_pxz")
    _check_act(sum_kl_costs, "sum_kl_costs", log_pxz.shape)
    total = log_pxz.numel()
    assert total % k == 0
    n = total // k
    out = torch.empty(n, dtype=torch.float32, device=log_pxz.device)
    _capi.check(_capi.lib().iaf_compute_lowerbound(_ptr(log_pxz), _ptr(sum_kl_costs), _ptr(out), n, int(k), _stream()))
    return out


def logsumexp(x):
    """distributions.py:35-37: over axis 1 of a [n, k] tensor."""
    _check_act(x, "x")
    n, k = x.shape
    zeros = torch.zeros_like(x)
    # logsumexp(x) = -(compute_lowerbound(x, 0, k)) + log k
    return -compute_lowerbound(x.reshape(-1), zeros.reshape(-1), k) + math.log(float(k))


class StreamingLowerBound(object):
    """k-sample importance-weighted bound without materialising [n, k] (BASELINE config 5,
    k = 10^4): feed (log_pxz, sum_kl) chunks of shape [n, k_chunk]; result() equals
    compute_lowerbound on the concatenation along k."""

    def __init__(self, n, device):
        self.n, self.k = int(n), 0
        self.run_max = torch.empty(n, dtype=torch.float32, device=device)
        self.run_sum = torch.empty(n, dtype=torch.float32, device=device)
        _capi.check(_capi.lib().iaf_lowerbound_stream_init(_ptr(self.run_max), _ptr(self.run_sum), self.n, _stream()))

    def update(self, log_pxz, sum_kl_costs):
        _check_act(log_pxz, "log_pxz")
        _check_act(sum_kl_costs, "sum_kl_costs", log_pxz.shape)
        n, kc = log_pxz.shape
        assert n == self.n
        _capi.check(_capi.lib().iaf_lowerbound_stream_update(_ptr(self.run_max), _ptr(self.run_sum), _ptr(log_pxz),
                                                             _ptr(sum_kl_costs), self.n, int(kc), _stream()))
        self.k += int(kc)

    def result(self):
        out = torch.empty(self.n, dtype=torch.float32, device=self.run_max.device)
        _capi.check(_capi.lib().iaf_lowerbound_stream_finalize(_ptr(self.run_max), _ptr(self.run_sum), _ptr(out),
                                                               self.n, self.k, _stream()))
        return out


def discretized_logistic(mean, logscale, binsize=1 / 256.0, sample=None):
    """distributions.py:28-32 (call site tf_train.py:210: logscale is the scalar variable dec_log_stdv).
    Returns the log-likelihood summed over every axis but the first, [B]."""
    _check_act(mean, "mean")
    _check_act(sample, "sample", mean.shape)
    if not isinstance(logscale, torch.Tensor):
        logscale = torch.full((1,), float(logscale), device=mean.device, dtype=torch.float32)
    scalar = logscale.numel() == 1
    if not scalar:
        _check_act(logscale, "logscale", mean.shape)
    else:
        logscale = logscale.reshape(1).contiguous()
    B = int(mean.shape[0])
    out = torch.empty((B,), device=mean.device, dtype=torch.float32)
    _capi.check(_capi.lib().iaf_discretized_logistic(_ptr(mean), _ptr(logscale), 1 if scalar else 0, _ptr(sample), _ptr(out), B,
                                                     mean.numel() // B, float(binsize), _stream()))
    return out

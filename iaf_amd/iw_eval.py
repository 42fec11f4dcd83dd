"""k-sample importance-weighted ELBO evaluation of the IAF posterior stack (BASELINE configs[4]: 10,000 importance samples
per image, inference only, rows of 256 per pass).

Reference semantics: the evaluation repeats every image k times (`repeat(x, k)`, tf_train.py:168-169; the Theano driver
instead evaluates the whole batch est_marglik times, train.py:194-203), sums the per-layer KL costs over the layer loop
(`kl_cost += cur_cost`, tf_train.py:198-200) and forms, per image,
    -( -log k + logsumexp_k( log p(x|z_k) - sum_kl_k ) )            (compute_lowerbound, distributions.py:55-62).
Here one PASS pushes a batch of n images x 1 importance sample through every layer's fused posterior block (the Theano
driver's order), writes each layer's kl_cost into one row of a [layers, n] matrix, column-sums it, and folds the pass
into a per-image running (max, sum) pair -- the [n, k] weight matrix never exists.  After k passes `result()` equals
compute_lowerbound on the concatenation."""
import ctypes

import torch

from . import _capi
from .distributions import StreamingLowerBound
from .layers import _ptr, _stream


class IWEvaluator(object):
    def __init__(self, stacks, kl_min=0.25):
        """stacks: one prepared ARStack per IAF layer, in any fixed order"""
        self.stacks = list(stacks)
        self.kl_min = float(kl_min)
        self.state = None
        self._kl = self._sum = self._z = self._obj = None

    def reset(self, n, device):
        self.state = StreamingLowerBound(n, device)
        m = len(self.stacks)
        self._kl = torch.empty((m, n), dtype=torch.float32, device=device)
        self._obj = torch.empty((m, n), dtype=torch.float32, device=device)
        self._sum = torch.empty((n, 1), dtype=torch.float32, device=device)
        self._z = [None] * m

    def run_pass(self, layer_inputs, log_pxz):
        """layer_inputs[l] = (qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd, up_context, down_context, eps) of
        layer l for this pass (n rows each); log_pxz [n] = log p(x | z) of the pass (from the decoder, out of scope)."""
        n = int(log_pxz.shape[0])
        if self.state is None or self.state.n != n:
            self.reset(n, log_pxz.device)
        for l, (st, inp) in enumerate(zip(self.stacks, layer_inputs)):
            if self._z[l] is None or self._z[l].shape != inp[0].shape:
                self._z[l] = torch.empty_like(inp[0])
            st.posterior_block(*inp, self.kl_min, out=dict(z=self._z[l], kl_obj=self._obj[l], kl_cost=self._kl[l]))
        _capi.check(_capi.lib().iaf_colsum(_ptr(self._kl), _ptr(self._sum), len(self.stacks), n, _stream()))
        self.state.update(log_pxz.reshape(n, 1), self._sum)

    @property
    def k(self):
        return 0 if self.state is None else self.state.k

    def result(self):
        """[n] per-image bound over the passes so far (== compute_lowerbound(log_pxz, sum_kl, k) on the concatenation)"""
        return self.state.result()

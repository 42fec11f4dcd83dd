"""The IAF part of tf_train.IAFLayer.down (tf_train.py:56-85) as one engine call.

The two convolutions around it (down_conv1 / down_conv2, tf_train.py:53,93) are outside this
path; the boundary is the channel split of down_conv1's output
(tf_train.py:54) and the stored up-pass tensors (tf_train.py:38)."""
from .layers import ARStack


class IAFPosterior(object):
    """Holds the ar_multiconv2d variables of one IAFLayer and evaluates
    posterior sample -> IAF step -> log-det -> KL / free bits."""

    def __init__(self, z_size, h_size, depth_ar=2, kl_min=0.25):
        self.z_size, self.h_size, self.kl_min = z_size, h_size, kl_min
        self.stack = ARStack(z_size, [h_size] * depth_ar)
        # set by up() in the reference (tf_train.py:38)
        self.qz_mean = self.qz_logsd = self.up_context = None

    def load(self, params):
        self.stack.prepare(params)

    def set_up_state(self, qz_mean, qz_logsd, up_context):
        self.qz_mean, self.qz_logsd, self.up_context = qz_mean, qz_logsd, up_context

    def down(self, pz_mean, pz_logsd, rz_mean, rz_logsd, down_context, eps, want_kl_elem=False):
        """Returns dict(z, kl_obj, kl_cost): the values tf_train.py:85-87 hand to the rest of down()."""
        return self.stack.posterior_block(self.qz_mean, self.qz_logsd, rz_mean, rz_logsd, pz_mean, pz_logsd,
                                          self.up_context, down_context, eps, self.kl_min, want_kl_elem)

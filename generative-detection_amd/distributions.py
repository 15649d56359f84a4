"""Diagonal Gaussian posterior with the reference's interface.

Mirrors src/util/distributions.py:5-41 (kl override: self-KL summed over dims [1,2,3], cross-KL with 1e-5 added to
the other variance) on top of the [UPSTREAM] ldm/modules/distributions/distributions.py base (chunk along dim 1,
logvar clamped to [-30, 20], sample = mean + std * randn drawn on the host, mode = mean).

4-d moments on a HIP device (the image posterior, [B, 2*Cz, H, W]) go through the fused HIP kernels; the
small 2-d pose-head posteriors ([B, 16], [8, 2]) use stock torch ops, as SURVEY.md 8(a) row a23 scopes them.
"""
import torch

from . import lib as _lib
from . import ops


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.deterministic = deterministic
        self._hip = parameters.dim() == 4 and parameters.is_cuda
        self._cache = {}

    # mean / logvar / std / var are views or tiny torch ops, built on first use (API compatibility; the hot
    # path uses sample() and kl() below, which read `parameters` directly)
    def _moments(self):
        if "mean" not in self._cache:
            mean, logvar = torch.chunk(self.parameters, 2, dim=1)
            self._cache["mean"] = mean
            self._cache["logvar"] = torch.clamp(logvar, -30.0, 20.0)
        return self._cache["mean"], self._cache["logvar"]

    @property
    def mean(self):
        return self._moments()[0]

    @property
    def logvar(self):
        return self._moments()[1]

    @property
    def std(self):
        if self.deterministic:
            return torch.zeros_like(self.mean)
        return torch.exp(0.5 * self.logvar)

    @property
    def var(self):
        if self.deterministic:
            return torch.zeros_like(self.mean)
        return torch.exp(self.logvar)

    def sample(self, eps=None):
        """mean + std * eps.  eps defaults to the reference's host-side torch.randn draw moved to the device;
        tests inject it (bit-level RNG parity with a CPU draw is otherwise impossible)."""
        shape = (self.parameters.shape[0], self.parameters.shape[1] // 2) + tuple(self.parameters.shape[2:])
        if eps is None:
            eps = torch.randn(shape)
        eps = _lib.upload(eps, self.parameters.device)
        if self.deterministic:
            return self.mean + 0.0 * eps
        if self._hip:
            return ops.gaussian_sample(self.parameters, eps)
        return self.mean + self.std * eps

    def kl(self, other=None):
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            if self._hip:
                return ops.gaussian_kl(self.parameters)
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        o_mean = other.mean.squeeze().unsqueeze(0).to(self.mean)
        o_var = other.var.squeeze().unsqueeze(0).to(self.mean)
        o_logvar = other.logvar.squeeze().unsqueeze(0).to(self.mean)
        dims = list(range(1, o_mean.dim()))
        return 0.5 * torch.sum(torch.pow(self.mean - o_mean, 2) / (o_var + 1e-5)
                               + self.var / (o_var + 1e-5) - 1.0 - self.logvar + o_logvar, dim=dims)

    def nll(self, sample, dims=[1, 2, 3]):
        if self.deterministic:
            return torch.Tensor([0.])
        logtwopi = 1.8378770664093453
        return 0.5 * torch.sum(logtwopi + self.logvar + torch.pow(sample - self.mean, 2) / self.var, dim=dims)

    def mode(self):
        return self.mean

"""ORACLE (test infrastructure): src/util/distributions.py:5-41 on top of the [UPSTREAM]
ldm/modules/distributions/distributions.py DiagonalGaussianDistribution, restated with torch CPU ops.  kl() / kl(other) are held to outputs of the
reference's own file (tests/golden/reference_glue.npz, tests/test_reference_glue.py); the upstream base class members are a restatement."""
import torch


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, eps=None):
        # upstream: self.mean + self.std * torch.randn(self.mean.shape); eps lets tests share the draw with the HIP path
        if eps is None:
            eps = torch.randn(self.mean.shape)
        return self.mean + self.std * eps

    def kl(self, other=None):
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        other_mean = other.mean.squeeze().unsqueeze(0).to(self.mean)
        other_var = other.var.squeeze().unsqueeze(0).to(self.mean)
        other_logvar = other.logvar.squeeze().unsqueeze(0).to(self.mean)
        sum_dim = list(range(1, len(other_mean.size())))
        return 0.5 * torch.sum(torch.pow(self.mean - other_mean, 2) / (other_var + 1e-5)
                               + self.var / (other_var + 1e-5) - 1.0 - self.logvar + other_logvar, dim=sum_dim)

    def mode(self):
        return self.mean

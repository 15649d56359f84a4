from odvae_amd.distributions import DiagonalGaussianDistribution  # noqa: F401

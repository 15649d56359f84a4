"""MI355X-native OD-VAE autoencoder training path (host side).  Import it as `odvae_amd`."""
__version__ = "0.1.0"

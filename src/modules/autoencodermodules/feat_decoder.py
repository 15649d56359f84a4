from odvae_amd.autoencoder import FeatDecoder  # noqa: F401

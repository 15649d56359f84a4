from odvae_amd.autoencoder import FeatEncoder  # noqa: F401

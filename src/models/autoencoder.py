"""`src.models.autoencoder.{PoseAutoencoder, Autoencoder, AutoencoderKL}` (yaml:3) -> generative-detection_amd/autoencoder.py"""
from odvae_amd.autoencoder import Autoencoder, AutoencoderKL, PoseAutoencoder  # noqa: F401

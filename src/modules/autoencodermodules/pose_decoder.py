"""yaml:34 `src.modules.autoencodermodules.pose_decoder.PoseDecoderSpatialVAE`"""
from odvae_amd.pose_modules import PoseDecoder, PoseDecoderSpatialVAE  # noqa: F401

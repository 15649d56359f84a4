"""yaml:46 `src.modules.autoencodermodules.pose_encoder.PoseEncoderSpatialVAE`"""
from odvae_amd.pose_modules import PoseEncoder, PoseEncoderSpatialVAE  # noqa: F401

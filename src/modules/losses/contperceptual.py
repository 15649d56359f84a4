from odvae_amd.losses import LPIPSWithDiscriminator, PoseLoss  # noqa: F401

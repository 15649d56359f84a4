"""`src.modules.losses.PoseLoss` (yaml:16).  The reference ships this package file empty, so the target does not even
resolve there; it is exported here."""
from odvae_amd.losses import LPIPSWithDiscriminator, PoseLoss  # noqa: F401

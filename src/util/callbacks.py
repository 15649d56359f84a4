from odvae_amd.callbacks import Callback, DeviceStatsMonitor, ImageLogger, ModelCheckpoint, TQDMProgressBar, make_grid  # noqa: F401

from odvae_amd.callbacks import Callback, DeviceStatsMonitor, ImageLogger, TQDMProgressBar, make_grid  # noqa: F401

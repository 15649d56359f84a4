"""Training callbacks the reference yaml names under `lightning.callbacks` (configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml:115-131):
`ImageLogger` (src/util/callbacks.py:78-218) -- the caller of `PoseAutoencoder.log_images` (src/models/autoencoder.py:397-432), SURVEY.md
8(f) rank 2 -- plus stand-ins for the two pytorch_lightning callbacks the reference re-exports from the same module
(`TQDMProgressBar`, `DeviceStatsMonitor`), so that every `target:` of the untouched yaml resolves without pytorch_lightning.

Host-side code: grids are assembled with torch on the CPU copy of at most `max_images` images and written with Pillow, as the reference
does (torchvision.utils.make_grid + PIL.Image; torchvision is absent here, `make_grid` restates its published layout rule).
File naming, directory layout, the power-of-two logging schedule and its quirks are the reference's.
"""
import logging
import math
import os

import numpy as np
import torch


class Callback:
    """The hook surface of pytorch_lightning.Callback that trainer.Trainer drives."""

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx):
        pass

    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        pass

    def on_validation_end(self, trainer, pl_module):
        pass


def make_grid(tensor, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid for a [B, C, H, W] batch (normalize=False): a single image is returned as it is; otherwise the
    images are laid out row-major, `nrow` per row, each in a cell of (H + padding) x (W + padding) whose top / left `padding` pixels
    -- and the grid's bottom / right border -- hold `pad_value`.  One-channel images are repeated to three channels."""
    if tensor.dim() == 2:
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 3:
        if tensor.size(0) == 1:
            tensor = torch.cat((tensor, tensor, tensor), 0)
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 4 and tensor.size(1) == 1:
        tensor = torch.cat((tensor, tensor, tensor), 1)
    if tensor.size(0) == 1:
        return tensor.squeeze(0)
    nmaps = tensor.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
    grid = tensor.new_full((tensor.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid[:, y * height + padding:(y + 1) * height, x * width + padding:(x + 1) * width].copy_(tensor[k])
            k += 1
    return grid


class ImageLogger(Callback):
    """src/util/callbacks.py:78-218.  Every time the schedule fires, `pl_module.log_images(batch, split=...)` runs in eval mode under
    no_grad (on the device: the same HIP kernels as the training step's forward), at most `max_images` images per key come to the host,
    are clamped to [-1, 1], and each key is written as `<save_dir>/images/<split>/<key>_gs-<global_step:06>_e-<epoch:06>_b-<batch:06>.png`
    (4 images per row, [-1, 1] -> [0, 255]).  Schedule: steps 1, 2, 4, ..., 2^floor(log2(batch_frequency)) once each, and every
    multiple of `batch_frequency`; never at step 0 unless `log_first_step`.  `logger_log_images` is the reference's hook table keyed by logger
    type (callbacks.py:115-117: `pl.loggers.TensorBoardLogger -> _testtube`): `_testtube` is here too, registered for
    `pytorch_lightning.loggers.TensorBoardLogger` where that package is importable and taken for any logger whose class is NAMED
    TensorBoardLogger and whose `experiment` has `add_image` (torch.utils.tensorboard.SummaryWriter's signature) otherwise."""

    def __init__(self, batch_frequency, max_images, clamp=True, increase_log_steps=True, rescale=True, disabled=False,
                 log_on_batch_idx=False, log_first_step=False, log_images_kwargs=None, disable_local_logging=False):
        super().__init__()
        self.rescale = rescale
        self.batch_freq = batch_frequency
        self.max_images = max_images
        self.logger_log_images = {}
        try:      # the reference's table (callbacks.py:115-117), where Lightning is there to key it
            from pytorch_lightning.loggers import TensorBoardLogger as _TB
            self.logger_log_images[_TB] = self._testtube
        except Exception:  # noqa: BLE001
            pass
        self.log_steps = [2 ** n for n in range(int(np.log2(self.batch_freq)) + 1)]
        if not increase_log_steps:
            self.log_steps = [self.batch_freq]
        self.clamp = clamp
        self.disabled = disabled
        self.log_on_batch_idx = log_on_batch_idx
        self.log_images_kwargs = log_images_kwargs if log_images_kwargs else {}
        self.log_first_step = log_first_step
        self.disable_local_logging = disable_local_logging

    def _testtube(self, pl_module, images, batch_idx, split):
        """callbacks.py:128-139: one grid per key (make_grid's default 8 per row, [-1, 1] -> [0, 1]) into `logger.experiment.add_image`."""
        if not _is_rank_zero():
            return
        for k in images:
            grid = make_grid(images[k])
            grid = (grid + 1.0) / 2.0
            pl_module.logger.experiment.add_image("%s/%s" % (split, k), grid, global_step=pl_module.global_step)

    def log_local(self, save_dir, split, images, global_step, current_epoch, batch_idx):
        from PIL import Image
        root = os.path.join(save_dir, "images", split)
        written = []
        for k in images:
            grid = make_grid(images[k], nrow=4)
            if self.rescale:
                grid = (grid + 1.0) / 2.0  # -1,1 -> 0,1; c,h,w
            grid = grid.transpose(0, 1).transpose(1, 2).squeeze(-1)
            grid = (grid.numpy() * 255).astype(np.uint8)
            path = os.path.join(root, "{}_gs-{:06}_e-{:06}_b-{:06}.png".format(k, global_step, current_epoch, batch_idx))
            os.makedirs(os.path.split(path)[0], exist_ok=True)
            Image.fromarray(grid).save(path)
            written.append(path)
        return written

    def log_img(self, pl_module, batch, batch_idx, split="train"):
        check_idx = batch_idx if self.log_on_batch_idx else pl_module.global_step
        if not (self.check_frequency(check_idx) and callable(getattr(pl_module, "log_images", None)) and self.max_images > 0):
            return []
        is_train = pl_module.training
        if is_train:
            pl_module.eval()
        with torch.no_grad():
            images = pl_module.log_images(batch, split=split, **self.log_images_kwargs)
        for k in images:
            n = min(images[k].shape[0], self.max_images)
            images[k] = images[k][:n]
            if isinstance(images[k], torch.Tensor):
                images[k] = images[k].detach().float().cpu().contiguous()
                if self.clamp:
                    images[k] = torch.clamp(images[k], -1., 1.)
        written = []
        # (`_odvae_logger`: where trainer.Trainer(logger=) parks it when `logger` is the read-only property of a real LightningModule)
        logger = getattr(pl_module, "logger", None) or getattr(pl_module, "_odvae_logger", None)
        if not self.disable_local_logging and _is_rank_zero():
            save_dir = getattr(logger, "save_dir", None) or "."
            written = self.log_local(save_dir, split, images, pl_module.global_step, getattr(pl_module, "current_epoch", 0), batch_idx)
        hook = self.logger_log_images.get(type(logger))
        if hook is None and type(logger).__name__ == "TensorBoardLogger" and hasattr(getattr(logger, "experiment", None), "add_image"):
            hook = self._testtube
        if hook is not None:
            hook(pl_module, images, pl_module.global_step, split)
        if is_train:
            pl_module.train()
        return written

    def check_frequency(self, check_idx):
        if ((check_idx % self.batch_freq) == 0 or (check_idx in self.log_steps)) and (check_idx > 0 or self.log_first_step):
            try:
                self.log_steps.pop(0)
            except IndexError as e:   # the reference logs and carries on once the power-of-two list is used up
                logging.info(e)
            return True
        return False

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx):
        if not self.disabled and (pl_module.global_step > 0 or self.log_first_step):
            self.log_img(pl_module, batch, batch_idx, split="train")

    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        if not self.disabled and pl_module.global_step > 0:
            self.log_img(pl_module, batch, batch_idx, split="val")


class ModelCheckpoint(Callback):
    """The part of pytorch_lightning.callbacks.ModelCheckpoint the reference configures (train.py:228-249): `dirpath`,
    `filename="{epoch:06}"`, `save_last=True`, `save_weights_only=True`, and -- when the model names a `monitor` (yaml:5,
    `val/rec_loss`) -- `monitor` with `save_top_k=3`, mode "min".  Runs at the end of every validation pass
    (`Trainer.validate` / the epoch end of `Trainer.fit`): writes `<dirpath>/<filename>.ckpt` through `trainer.save_checkpoint` when
    the monitored epoch mean is among the k best so far (and removes the file that dropped out), then `last.ckpt`.  [PL-1.9] naming:
    `{epoch:06}` becomes `epoch=000003` (metric names are inserted), a clash gets a `-v1` suffix."""

    def __init__(self, dirpath=None, filename=None, monitor=None, verbose=False, save_last=None, save_top_k=1, save_weights_only=False,
                 mode="min", **_ignored):
        super().__init__()
        if mode not in ("min", "max"):
            raise ValueError("ModelCheckpoint mode %r" % (mode,))
        self.dirpath, self.filename, self.monitor, self.verbose = dirpath, filename, monitor, verbose
        self.save_last, self.save_top_k, self.save_weights_only, self.mode = save_last, save_top_k, save_weights_only, mode
        self.best_k_models, self.kth_best_model_path = {}, ""
        self.best_model_path, self.best_model_score, self.last_model_path = "", None, ""

    @property
    def state_key(self):
        return "ModelCheckpoint{'monitor': %r, 'mode': %r}" % (self.monitor, self.mode)

    def state_dict(self):
        return {"monitor": self.monitor, "best_model_score": self.best_model_score, "best_model_path": self.best_model_path,
                "best_k_models": dict(self.best_k_models), "kth_best_model_path": self.kth_best_model_path,
                "last_model_path": self.last_model_path, "dirpath": self.dirpath}

    def format_checkpoint_name(self, metrics):
        import re
        name = self.filename or "{epoch}-{step}"
        for group in re.findall(r"(\{.*?)[:\}]", name):
            key = group[1:]
            name = name.replace(group, key + "={" + key.replace("/", "_"), 1)       # auto_insert_metric_name
        vals = {k.replace("/", "_"): (v.item() if torch.is_tensor(v) else v) for k, v in metrics.items()}
        return name.format(**vals)

    def _path(self, trainer, metrics):
        """The file name of this save; an existing file of that name that is not one of ours gets a -v<k> suffix ([PL-1.9] versioning).
        Only rank 0 writes, so only rank 0 looks at the disk; under data parallelism it broadcasts its answer, and `best_k_models` (part
        of `state_dict()`) holds the same keys on every rank."""
        base = self.format_checkpoint_name(metrics)
        path, v = os.path.join(self.dirpath or ".", base + ".ckpt"), 0
        if _is_rank_zero():
            while os.path.exists(path) and path not in self.best_k_models:
                v += 1
                path = os.path.join(self.dirpath or ".", "%s-v%d.ckpt" % (base, v))
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            box = [path]
            dist.broadcast_object_list(box, src=0)
            path = box[0]
        return path

    def on_validation_end(self, trainer, pl_module):
        metrics = dict(trainer.callback_metrics)
        metrics.update(epoch=trainer.current_epoch, step=pl_module.global_step)
        sign = 1.0 if self.mode == "min" else -1.0
        if self.monitor is None:
            if self.save_top_k != 0:
                path = self._path(trainer, metrics)
                trainer.save_checkpoint(path, weights_only=self.save_weights_only)
                if self.save_top_k == 1 and self.best_model_path and self.best_model_path != path and _is_rank_zero() and os.path.exists(self.best_model_path):
                    os.remove(self.best_model_path)     # without a monitor PL keeps the latest only (top_k = 1)
                self.best_model_path = path
        elif self.save_top_k != 0:
            if self.monitor not in metrics:
                raise KeyError("ModelCheckpoint(monitor=%r): validation logged %s" % (self.monitor, sorted(trainer.callback_metrics)))
            score = float(metrics[self.monitor])
            full = self.save_top_k > 0 and len(self.best_k_models) >= self.save_top_k
            worst = max(self.best_k_models, key=lambda k: sign * self.best_k_models[k]) if self.best_k_models else None
            if not full or sign * score < sign * self.best_k_models[worst]:
                path = self._path(trainer, metrics)
                trainer.save_checkpoint(path, weights_only=self.save_weights_only)
                if full:
                    del self.best_k_models[worst]
                    if _is_rank_zero() and os.path.exists(worst):
                        os.remove(worst)
                self.best_k_models[path] = score
                self.kth_best_model_path = max(self.best_k_models, key=lambda k: sign * self.best_k_models[k])
                self.best_model_path = min(self.best_k_models, key=lambda k: sign * self.best_k_models[k])
                self.best_model_score = self.best_k_models[self.best_model_path]
                if self.verbose:
                    logging.info("ModelCheckpoint: %s = %.6f -> %s", self.monitor, score, path)
        if self.save_last:
            self.last_model_path = trainer.save_checkpoint(os.path.join(self.dirpath or ".", "last.ckpt"), weights_only=self.save_weights_only)


def default_modelcheckpoint(model, ckptdir):
    """What train.py:228-241 builds when the yaml has no `modelcheckpoint` section."""
    kw = dict(dirpath=ckptdir, filename="{epoch:06}", verbose=True, save_last=True, save_weights_only=True)
    if hasattr(model, "monitor"):
        kw.update(monitor=model.monitor, save_top_k=3)
    return ModelCheckpoint(**kw)


def _is_rank_zero():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class TQDMProgressBar(Callback):
    """Stand-in for pytorch_lightning.callbacks.TQDMProgressBar (yaml:123-127): keeps the constructor, prints nothing."""

    def __init__(self, refresh_rate=1, process_position=0):
        super().__init__()
        self.refresh_rate, self.process_position = refresh_rate, process_position


class DeviceStatsMonitor(Callback):
    """Stand-in for pytorch_lightning.callbacks.DeviceStatsMonitor (yaml:129-130): after each training batch the allocator's
    current / peak bytes go to `pl_module.log` under the names Lightning's accelerator stats use."""

    def __init__(self, cpu_stats=None):
        super().__init__()
        self.cpu_stats = cpu_stats

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx):
        dev = getattr(pl_module, "device", None)
        if dev is not None and dev.type == "cuda":
            pl_module.log("DeviceStatsMonitor.on_train_batch_end/allocated_bytes.all.current", float(torch.cuda.memory_allocated(dev)))
            pl_module.log("DeviceStatsMonitor.on_train_batch_end/allocated_bytes.all.peak", float(torch.cuda.max_memory_allocated(dev)))

"""LightningModule-compatible base used when pytorch_lightning is absent (it is, in this image).

The reference model is a pl.LightningModule (src/models/autoencoder.py:14,66) driven by a PL-1.9 Trainer with
two optimizers under automatic optimisation.  Only the surface the model itself touches is
reproduced: log / log_dict, global_step, device, trainer, learning_rate, toggle/untoggle_optimizer.
If pytorch_lightning is importable the real class is used instead, so the model drops into a real Trainer.
"""
import torch
import torch.nn as nn

try:  # pragma: no cover - not installed here
    import pytorch_lightning as _pl
    LightningModule = _pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self._global_step = 0
            self._logged = {}
            self.trainer = None
            self._toggled = None
            self.logger = None          # anything with a `save_dir` (callbacks.ImageLogger); trainer.Trainer(logger=...) sets it
            self.current_epoch = 0

        # ---- what PL provides and the model reads -----------------------------------------------------
        @property
        def global_step(self):
            """Number of optimizer steps taken so far (PL-1.9: advances once per optimizer.step, i.e. by 2
            per batch with two optimizers)."""
            return self._global_step

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def log(self, name, value, *args, sync_dist=False, **kwargs):
            """sync_dist=True (validation's `val/rec_loss`, src/models/autoencoder.py:359): mean over the ranks."""
            if torch.is_tensor(value):
                value = value.detach()
            if sync_dist:
                from .parallel import all_reduce_mean
                value = all_reduce_mean(value)
            self._logged[name] = value

        def log_dict(self, d, *args, **kwargs):
            for k, v in d.items():
                self.log(k, v)

        @property
        def logged_metrics(self):
            return self._logged

        # ---- PL-1.9 optimizer toggling (automatic optimisation, multiple optimizers) ---------------------
        def toggle_optimizer(self, optimizer, optimizer_idx=None):
            """requires_grad off for every parameter that is not owned by `optimizer`."""
            mine = {id(p) for g in optimizer.param_groups for p in g["params"]}
            saved = {}
            for p in self.parameters():
                saved[p] = p.requires_grad
                if id(p) not in mine:
                    p.requires_grad = False
            self._toggled = saved

        def untoggle_optimizer(self, optimizer_idx=None):
            if self._toggled is not None:
                for p, rg in self._toggled.items():
                    p.requires_grad = rg
                self._toggled = None

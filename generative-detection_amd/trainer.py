"""Minimal training loop with the semantics of the reference's PL-1.9 Trainer for this model.

Per batch, for optimizer_idx in (0, 1): toggle_optimizer -> training_step -> zero_grad -> backward (DDP buckets fire)
-> clip_grad_norm_(gradient_clip_val) -> optimizer.step -> untoggle; global_step advances once per optimizer step,
i.e. by 2 per batch (every `global_step` threshold in the model and the loss depends on that).
Launcher concerns of train.py (loggers, checkpoints callbacks, CLI) are out of scope; `fit` takes any iterable of
batch dicts.
"""
import torch

from .parallel import GradReducer


class _TrainerHandle:
    """What `model.trainer` exposes to the module (the optimizer list drives PL-style toggling)."""

    def __init__(self, optimizers):
        self.optimizers = optimizers


class Trainer:
    def __init__(self, model, gradient_clip_val=None, optimizer_indices=(0, 1), process_group=None, bucket_mb=32.0, precision=None,
                 distributed=None, comm_dtype=None, callbacks=(), logger=None):
        """optimizer_indices: which of the model's optimizers run each batch; (0,) is the "rec+KL only" benchmark
        configuration (discriminator off, optimizer 1 skipped -- SURVEY.md 8(d)).
        comm_dtype: dtype of the gradient buckets on the wire; None = f32, except under precision "bf16" with bucket_mb left at its
        default, where the buckets travel as bf16 in 16 MB pieces (parallel.GradReducer).
        distributed: None = data-parallel exactly when a process group is given or the default group has more than one rank
        (what `strategy: ddp` amounts to, yaml:137); False = never (a single-process reference run inside a rank)."""
        self.model = model
        # callbacks: objects with the pytorch_lightning.Callback hooks used by the reference's yaml (callbacks.ImageLogger, ...);
        # `on_train_batch_end` runs after the last optimizer step of a batch, as under PL's automatic optimisation.  logger: anything
        # with a `save_dir` (ImageLogger writes <save_dir>/images/<split>/...); it becomes `model.logger`.
        self.callbacks = list(callbacks)
        if logger is not None:
            model.logger = logger
        if precision is not None:   # lightning.trainer.precision of the yaml (:139): 32 or "bf16"
            model.set_precision(precision)
        self.clip = gradient_clip_val
        self.optimizer_indices = tuple(optimizer_indices)
        opts, _ = model.configure_optimizers()
        self.optimizers = opts
        for o in opts:
            if hasattr(o, "materialize"):
                o.materialize()
        model.trainer = _TrainerHandle(opts)
        self.reducers = None
        if distributed is None:
            distributed = process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                        and torch.distributed.get_world_size() > 1)
        if distributed:
            # the mean over ranks rides in the loss scale (training_batch), not in a pass over the gradient arena
            if comm_dtype is None and precision is not None and str(precision).lower().startswith("bf16"):
                comm_dtype = torch.bfloat16
                if bucket_mb == 32.0:
                    bucket_mb = 16.0     # f32 megabytes of arena per bucket = 8 MB on the wire: ~18 collectives inside a ~50 ms backward
            self.reducers = [GradReducer(o, process_group=process_group, bucket_mb=bucket_mb, prescaled=True, comm_dtype=comm_dtype)
                             for o in opts]
            self.reducers[0].broadcast_parameters(model)

    # PL-1.9 toggle_optimizer: parameters owned by the *other* optimizers stop requiring grad; parameters in no
    # optimizer (loss.logvar, the frozen LPIPS net) are left alone
    def _toggle(self, idx):
        mine = {id(p) for g in self.optimizers[idx].param_groups for p in g["params"]}
        saved = {}
        for j, opt in enumerate(self.optimizers):
            if j == idx:
                continue
            for g in opt.param_groups:
                for p in g["params"]:
                    if id(p) not in mine and id(p) not in saved:
                        saved[id(p)] = (p, p.requires_grad)
                        p.requires_grad = False
        return saved

    @staticmethod
    def _untoggle(saved):
        for p, rg in saved.values():
            p.requires_grad = rg

    def training_batch(self, batch, batch_idx=0):
        """One batch = one step of each scheduled optimizer.  Returns the list of loss tensors (device, not synced)."""
        model = self.model
        losses = []
        for idx in self.optimizer_indices:
            opt = self.optimizers[idx]
            saved = self._toggle(idx)
            try:
                loss = model.training_step(batch, batch_idx, idx)
                opt.zero_grad(set_to_none=True)   # FusedAdam: gradients are gathered into its arena after the backward (optim.gather_grads)
                red = self.reducers[idx] if self.reducers else None
                if red is not None:
                    red.prepare_for_backward()
                    (loss * red.inv_world).backward()   # sum over ranks of grad(loss / world) = DDP's mean gradient
                    red.finish()
                else:
                    loss.backward()
                if self.clip:
                    if hasattr(opt, "clip_grad_norm_"):
                        opt.clip_grad_norm_(self.clip)
                    else:
                        torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], self.clip)
                opt.step()
            finally:
                self._untoggle(saved)
            model._global_step += 1
            losses.append(loss.detach())
        for cb in self.callbacks:
            cb.on_train_batch_end(self, model, losses, batch, batch_idx)
        return losses

    def fit(self, batches, max_batches=None):
        self.model.train()
        out = []
        for i, batch in enumerate(batches):
            if max_batches is not None and i >= max_batches:
                break
            out.append(self.training_batch(batch, i))
        return out

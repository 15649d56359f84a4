"""Minimal training loop with the semantics of the reference's PL-1.9 Trainer for this model.

Per batch, for optimizer_idx in (0, 1): toggle_optimizer -> training_step -> zero_grad -> backward (DDP buckets fire)
-> clip_grad_norm_(gradient_clip_val) -> optimizer.step -> untoggle; global_step advances once per optimizer step,
i.e. by 2 per batch (every `global_step` threshold in the model and the loss depends on that).
`fit` takes any iterable of batch dicts; `validate` drives `validation_step` (src/models/autoencoder.py:332-363) over a loader with
the epoch mean and the `sync_dist` rank mean of `val/rec_loss`; `save_checkpoint` / `load_checkpoint` write and read the
Lightning-1.9 checkpoint layout the reference's `ModelCheckpoint` produces (train.py:228-249), so a run of this trainer resumes
under the reference and the reverse.  The rest of train.py (CLI, loggers) is out of scope.
"""
import os

import torch

from .parallel import GradReducer


class DeviceHealthError(RuntimeError):
    """A HIP kernel reported, through the library's device-side health counters, that it produced wrong numbers."""


class _TrainerHandle:
    """What `model.trainer` exposes to the module (the optimizer list drives PL-style toggling)."""

    def __init__(self, optimizers):
        self.optimizers = optimizers


class Trainer:
    def __init__(self, model, gradient_clip_val=None, optimizer_indices=(0, 1), process_group=None, bucket_mb=None, precision=None,
                 distributed=None, comm_dtype=None, callbacks=(), logger=None, comm_f32_accumulate=False):
        """optimizer_indices: which of the model's optimizers run each batch; (0,) is the "rec+KL only" benchmark
        configuration (discriminator off, optimizer 1 skipped -- SURVEY.md 8(d)).
        comm_dtype: dtype of the gradient buckets on the wire; None = f32 in every precision -- what the reference's `strategy: ddp`
        all-reduces under `precision: bf16` too (autocast leaves parameters and their gradients f32).  torch.bfloat16 is an opt-in:
        half the bytes per link, ~2^-9 relative rounding on every summed gradient element (deliberate deviations).
        bucket_mb: f32 megabytes of gradient arena per all-reduce; None = 32, or 16 with bf16 buckets (8 MB on the wire: ~18
        collectives inside the bf16 step's ~50 ms backward).
        distributed: None = data-parallel exactly when a process group is given or the default group has more than one rank
        (what `strategy: ddp` amounts to, yaml:137); False = never (a single-process reference run inside a rank)."""
        self.model = model
        # callbacks: objects with the pytorch_lightning.Callback hooks used by the reference's yaml (callbacks.ImageLogger, ...);
        # `on_train_batch_end` runs after the last optimizer step of a batch, as under PL's automatic optimisation.  logger: anything
        # with a `save_dir` (ImageLogger writes <save_dir>/images/<split>/...); it becomes `model.logger`.
        self.callbacks = list(callbacks)
        if logger is not None:
            try:
                model.logger = logger
            except AttributeError:      # a real pytorch_lightning.LightningModule: `logger` is a read-only property of its Trainer
                model._odvae_logger = logger
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
        self.current_epoch = 0            # completed passes over the training loader (PL: trainer.current_epoch)
        # device-side health counters as of the last check_device_health(): GroupNorm team-barrier timeouts (fatal) and exact-softmax
        # fallbacks of the folded attention softmax (correct, slower; reported)
        self.device_health = {"gn_barrier_timeouts": 0, "attn_softmax_fallbacks": 0}
        self.callback_metrics = {}        # name -> 0-d CPU tensor: the last validate()'s epoch means (what ModelCheckpoint monitors)
        self.reducers = None
        if distributed is None:
            distributed = process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                        and torch.distributed.get_world_size() > 1)
        if distributed:
            # the mean over ranks rides in the loss scale (training_batch), not in a pass over the gradient arena
            if bucket_mb is None:
                bucket_mb = 16.0 if comm_dtype == torch.bfloat16 else 32.0
            self.reducers = [GradReducer(o, process_group=process_group, bucket_mb=bucket_mb, prescaled=True, comm_dtype=comm_dtype, f32_accumulate=comm_f32_accumulate)
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

    def check_device_health(self, where=""):
        """Reads the library's two device-side counters (odvae_device_health: one device synchronisation, so it is called only where the
        host waits anyway -- validation, checkpoint save, the end of fit).  A GroupNorm backward whose team barrier gave up has written
        WRONG gradients (csrc/groupnorm.hip: the opt-in team mode, odvae_groupnorm_select_backward(1)): DeviceHealthError, and no
        checkpoint is written from such a state.  Fallbacks of the folded attention softmax are exact; new ones since the last check are
        reported once through `warnings` and to the logger's `log_metrics` when it has one.  Returns the counters."""
        import ctypes
        import warnings
        from . import lib as _lib
        if not torch.cuda.is_available():
            return dict(self.device_health)
        L = _lib.load()
        gn, at = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(L.odvae_device_health(ctypes.byref(gn), ctypes.byref(at), 0, 0), "device_health")
        prev = self.device_health
        self.device_health = {"gn_barrier_timeouts": int(gn.value), "attn_softmax_fallbacks": int(at.value)}
        if gn.value > prev["gn_barrier_timeouts"]:
            raise DeviceHealthError(
                "%d GroupNorm team-barrier wait(s) gave up on the device%s (global_step %d): the gradients of those launches are wrong. "
                "The team mode is opt-in (odvae_groupnorm_select_backward(1)); the default never waits across blocks."
                % (gn.value - prev["gn_barrier_timeouts"], (" before " + where) if where else "", int(self.model.global_step)))
        if at.value > prev["attn_softmax_fallbacks"]:
            new = at.value - prev["attn_softmax_fallbacks"]
            warnings.warn("%d attention block(s) took the exact-softmax fallback since the last check (global_step %d): a row's Cauchy-Schwarz "
                          "bound underflowed in f32; results are exact, each fallback costs three extra passes over its T x T scores"
                          % (new, int(self.model.global_step)), RuntimeWarning, stacklevel=2)
            logger = getattr(self.model, "logger", None) or getattr(self.model, "_odvae_logger", None)
            log = getattr(logger, "log_metrics", None)
            if log is not None:
                log({"device/attn_softmax_fallbacks": float(at.value)}, step=int(self.model.global_step))
        return dict(self.device_health)

    def fit(self, batches, max_batches=None, val_batches=None, max_epochs=1):
        """`max_epochs` passes over `batches` (re-iterated per epoch; at most `max_batches` each).  With `val_batches` every epoch
        ends as PL's does: validation over the whole loader, `on_validation_end` (ModelCheckpoint saves here), then the epoch counter
        advances.  Returns the per-batch loss lists of every epoch, flattened."""
        out = []
        for _ in range(max_epochs):
            self.model.train()
            for i, batch in enumerate(batches):
                if max_batches is not None and i >= max_batches:
                    break
                out.append(self.training_batch(batch, i))
            if val_batches is not None:
                self.validate(val_batches)
            self._set_epoch(self.current_epoch + 1)
        self.check_device_health("the end of fit")
        return out

    def _set_epoch(self, epoch):
        self.current_epoch = int(epoch)
        try:
            self.model.current_epoch = self.current_epoch
        except AttributeError:      # a real LightningModule: read-only, its own Trainer owns the counter
            pass

    # ---- validation (src/models/autoencoder.py:332-363; [PL-1.9] evaluation loop) -----------------------------------------------
    @torch.no_grad()
    def validate(self, batches, max_batches=None):
        """model.eval(), `validation_step` per batch under no_grad, every logged scalar averaged over the epoch the way [PL-1.9] does it
        for `self.log(..., on_epoch=True)` inside validation_step: the mean WEIGHTED BY BATCH SIZE (ResultMetric: value * batch_size
        summed, divided by the cumulated batch size; the batch size is the first dimension of the first tensor in the batch,
        `extract_batch_size`) -- with a ragged last batch this is what the reference's ModelCheckpoint(monitor) sees;
        `val/rec_loss` is logged with `sync_dist=True` (:359), so each batch value already is the mean over the ranks.  The epoch means land in
        `self.callback_metrics` (and are returned); callbacks see `on_validation_batch_end` per batch and `on_validation_end` once.
        The model's training mode is restored."""
        model = self.model
        was_training = model.training
        model.eval()
        sums, counts = {}, {}
        try:
            for i, batch in enumerate(batches):
                if max_batches is not None and i >= max_batches:
                    break
                logged = getattr(model, "_logged", None)
                if logged is not None:
                    logged.clear()
                out = model.validation_step(batch, i)
                bs = _batch_size(batch)
                for k, v in (getattr(model, "logged_metrics", None) or {}).items():
                    if torch.is_tensor(v):
                        if v.numel() != 1:
                            continue
                        v = v.detach().double().reshape(())
                    else:
                        v = torch.tensor(float(v), dtype=torch.float64)
                    v = v * bs
                    sums[k] = v if k not in sums else sums[k] + v.to(sums[k].device)      # stays on the device: no sync per batch
                    counts[k] = counts.get(k, 0) + bs
                for cb in self.callbacks:
                    cb.on_validation_batch_end(self, model, out, batch, i)
        finally:
            model.train(was_training)
        metrics = {k: (sums[k] / counts[k]).float().cpu() for k in sums}      # (the host waits for the device here anyway)
        self.check_device_health("validation ended")
        self.callback_metrics.update(metrics)
        for cb in self.callbacks:
            hook = getattr(cb, "on_validation_end", None)
            if hook is not None:
                hook(self, model)
        return metrics

    # ---- checkpoints: the Lightning-1.9 layout (train.py:228-249 ModelCheckpoint; [PL-1.9] CheckpointConnector.dump_checkpoint) --
    def dump_checkpoint(self, weights_only=False):
        """{"epoch", "global_step", "pytorch-lightning_version", "state_dict"} and, unless `weights_only` (the reference's default,
        train.py:236), {"optimizer_states", "lr_schedulers", "callbacks"}: optimizer_states[i] is optimizer i's `state_dict()` in
        torch.optim.Adam's layout (optim.FusedAdam writes exactly that), lr_schedulers is empty (configure_optimizers returns none,
        autoencoder.py:377).  `state_dict` keys / shapes / dtypes are the reference module tree's (OIHW f32); every
        tensor is copied to the host.  No "loops" entry: PL-1.9 then restores `global_step` and `epoch` from the top-level keys (its
        path for pre-1.6 checkpoints) instead of from a progress-tracker tree this trainer does not keep."""
        model = self.model
        ckpt = {"epoch": self.current_epoch, "global_step": int(model.global_step), "pytorch-lightning_version": "1.9.0",
                "state_dict": {k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items()}}
        if not weights_only:
            def host(o):
                if torch.is_tensor(o):
                    return o.detach().to("cpu", copy=True)
                if isinstance(o, dict):
                    return {k: host(v) for k, v in o.items()}
                if isinstance(o, (list, tuple)):
                    return type(o)(host(v) for v in o)
                return o
            ckpt["optimizer_states"] = [host(o.state_dict()) for o in self.optimizers]
            ckpt["lr_schedulers"] = []
            ckpt["callbacks"] = {}
            for cb in self.callbacks:
                sd = getattr(cb, "state_dict", None)
                if sd is not None:
                    ckpt["callbacks"][getattr(cb, "state_key", type(cb).__qualname__)] = sd()
        return ckpt

    def save_checkpoint(self, path, weights_only=False):
        """Rank 0 writes `path` (atomically: temporary file + rename); every rank returns the path."""
        self.check_device_health("writing %s" % os.path.basename(path))      # never a checkpoint of weights stepped with wrong gradients
        if _rank() == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            tmp = "%s.part" % path
            torch.save(self.dump_checkpoint(weights_only=weights_only), tmp)
            os.replace(tmp, path)
        return path

    def load_checkpoint(self, path, strict=True, trusted=False):
        """Resume from a checkpoint in that layout -- one of this trainer's or one Lightning wrote for the reference model: weights
        (strict by default), `global_step`, `epoch`, and the optimizer states when present (a weights-only file leaves the
        optimizers fresh, as Lightning does).  The file is read with torch's restricted unpickler (`weights_only=True`: tensors, numbers,
        strings, containers).  A real Lightning checkpoint may also carry arbitrary Python objects (`hyper_parameters` as an OmegaConf
        DictConfig, callback state with timedelta / Path): unpickling those executes code, so it happens only with `trusted=True`."""
        import pickle
        try:
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            if not trusted:
                raise pickle.UnpicklingError(
                    "%s holds objects torch's restricted unpickler refuses (%s). If the file comes from a source you trust -- a Lightning "
                    "checkpoint with hyper_parameters / callback state -- call load_checkpoint(path, trusted=True)." % (path, str(e).splitlines()[0])) from e
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
        res = self.model.load_state_dict(ckpt["state_dict"], strict=strict)
        from . import ops
        ops.PACK_CACHE.bump()           # load_state_dict copies into .data of the arena views: no version bump the pack cache could see
        self.model._global_step = int(ckpt.get("global_step", 0))
        self._set_epoch(ckpt.get("epoch", 0))
        states = ckpt.get("optimizer_states")
        if states is not None:
            if len(states) != len(self.optimizers):
                raise ValueError("checkpoint holds %d optimizer states, the model configures %d optimizers" % (len(states), len(self.optimizers)))
            for o, sd in zip(self.optimizers, states):
                o.load_state_dict(sd)
        return res


def _batch_size(batch):
    """[PL-1.9] extract_batch_size: the first dimension of the first tensor found in the batch (dicts / sequences walked in order); 1 if none."""
    if torch.is_tensor(batch):
        return int(batch.shape[0]) if batch.dim() > 0 else 1
    if isinstance(batch, dict):
        batch = list(batch.values())
    if isinstance(batch, (list, tuple)):
        for v in batch:
            if torch.is_tensor(v):
                return int(v.shape[0]) if v.dim() > 0 else 1
            if isinstance(v, (dict, list, tuple)) and any(torch.is_tensor(x) for x in (v.values() if isinstance(v, dict) else v)):
                return _batch_size(v)
    return 1


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0

"""ImageLogger (src/util/callbacks.py:78-218) on the host: the power-of-two logging schedule, file naming and
pixel values of the grids, torchvision's make_grid layout rule, and that every `target:` under the untouched yaml's
`lightning.callbacks` (yaml:115-131) resolves through the src.* shims.  A fake module stands in for the model (CPU tensors)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, "tests", "golden", "autoencoder_kl_16x16x16.yaml")


class FakeModule(nn.Module):
    def __init__(self, save_dir):
        super().__init__()
        self.global_step, self.current_epoch = 0, 3
        self.calls = []

        class L:
            pass
        self.logger = L()
        self.logger.save_dir = save_dir

    def log_images(self, batch, split="train", **kw):
        self.calls.append((self.global_step, split, self.training))
        g = torch.Generator().manual_seed(self.global_step)
        return {"inputs_rgb": torch.rand(5, 3, 6, 4, generator=g) * 3 - 1.5, "reconstructions_rgb": torch.rand(5, 3, 6, 4, generator=g) * 2 - 1}


def test_make_grid_layout():
    from odvae_amd.callbacks import make_grid
    x = torch.arange(5 * 3 * 2 * 3, dtype=torch.float32).reshape(5, 3, 2, 3)
    g = make_grid(x, nrow=4)                       # 2 rows of cells (4 + 1 images), cell = (2 + 2) x (3 + 2), outer border 2
    assert tuple(g.shape) == (3, 2 * 4 + 2, 4 * 5 + 2)
    assert torch.equal(g[:, 2:4, 2:5], x[0]) and torch.equal(g[:, 2:4, 7:10], x[1]) and torch.equal(g[:, 6:8, 2:5], x[4])
    assert g[:, :2].abs().sum() == 0 and g[:, :, :2].abs().sum() == 0 and g[:, 6:8, 7:].abs().sum() == 0   # padding and empty cells
    assert torch.equal(make_grid(x[:1]), x[0])    # a single image comes back as it is
    assert tuple(make_grid(torch.ones(2, 1, 4, 4)).shape) == (3, 8, 14)   # one channel -> three


def test_image_logger_schedule_files_and_pixels(tmp_path):
    from PIL import Image
    from odvae_amd.callbacks import ImageLogger, make_grid
    cb = ImageLogger(batch_frequency=10, max_images=3, increase_log_steps=True)
    assert cb.log_steps == [1, 2, 4, 8]
    m = FakeModule(str(tmp_path)).train()
    fired = []
    for step in range(0, 25):
        m.global_step = step
        before = len(m.calls)
        cb.on_train_batch_end(None, m, None, {}, batch_idx=step // 2)
        if len(m.calls) > before:
            fired.append(step)
    assert fired == [1, 2, 4, 8, 10, 20]           # powers of two once each, then every multiple of batch_frequency; never step 0
    assert all(not training for _, _, training in m.calls) and m.training   # log_images ran in eval mode, the mode was restored
    files = sorted(os.listdir(os.path.join(tmp_path, "images", "train")))
    assert len(files) == 12 and "inputs_rgb_gs-000004_e-000003_b-000002.png" in files
    # pixel values: first max_images images, clamped to [-1, 1], 4 per row, (x + 1) / 2 * 255 truncated to uint8
    m.global_step = 4
    want = m.log_images({})["inputs_rgb"][:3].clamp(-1, 1)
    grid = ((make_grid(want, nrow=4) + 1) / 2).permute(1, 2, 0).numpy()
    got = np.asarray(Image.open(os.path.join(tmp_path, "images", "train", "inputs_rgb_gs-000004_e-000003_b-000002.png")))
    assert got.shape == grid.shape and np.array_equal(got, (grid * 255).astype(np.uint8))
    # validation hook writes under images/val and never at global_step 0
    m.global_step = 0
    cb.on_validation_batch_end(None, m, None, {}, 0, 0)
    assert not os.path.exists(os.path.join(tmp_path, "images", "val"))
    m.global_step = 30
    cb.on_validation_batch_end(None, m, None, {}, 7, 0)
    assert sorted(os.listdir(os.path.join(tmp_path, "images", "val")))[0] == "inputs_rgb_gs-000030_e-000003_b-000007.png"


def test_image_logger_tensorboard_hook(tmp_path):
    """The reference's `logger_log_images` table (src/util/callbacks.py:115-117, 128-139): with a TensorBoard logger every logged key also
    goes to `logger.experiment.add_image(f"{split}/{key}", grid in [0, 1], global_step=...)`, one grid (make_grid, 8 per row) per key."""
    from odvae_amd.callbacks import ImageLogger, make_grid

    class Experiment:
        def __init__(self):
            self.images = []

        def add_image(self, tag, img, global_step=None):
            self.images.append((tag, img.clone(), global_step))

    class TensorBoardLogger:       # (what pytorch_lightning.loggers.TensorBoardLogger looks like to the callback)
        def __init__(self, save_dir):
            self.save_dir, self.experiment = save_dir, Experiment()

    m = FakeModule(str(tmp_path))
    m.logger = TensorBoardLogger(str(tmp_path))
    cb = ImageLogger(batch_frequency=4, max_images=5, clamp=True)
    m.global_step = 2
    cb.on_train_batch_end(None, m, None, {}, 0)
    got = m.logger.experiment.images
    assert [t for t, _, _ in got] == ["train/inputs_rgb", "train/reconstructions_rgb"] and all(gs == 2 for _, _, gs in got)
    g = torch.Generator().manual_seed(2)
    want = (make_grid(torch.clamp(torch.rand(5, 3, 6, 4, generator=g) * 3 - 1.5, -1.0, 1.0)) + 1.0) / 2.0
    assert torch.allclose(got[0][1], want) and got[0][1].min() >= 0.0 and got[0][1].max() <= 1.0
    assert any(f.endswith(".png") for _, _, fs in os.walk(str(tmp_path)) for f in fs)      # the local files are written as well
    # a logger that is not a TensorBoard logger gets nothing
    m2 = FakeModule(str(tmp_path / "other"))
    m2.logger.experiment = Experiment()
    m2.global_step = 2
    ImageLogger(batch_frequency=4, max_images=5).on_train_batch_end(None, m2, None, {}, 0)
    assert m2.logger.experiment.images == []


def test_yaml_callbacks_resolve():
    from odvae_amd.config import Config, instantiate_from_config
    cfg = Config.load(YAML)
    cbs = {k: instantiate_from_config(v) for k, v in cfg.lightning.callbacks.items()}
    assert set(cbs) == {"image_logger", "progress_bar", "device_stats_monitor"}
    il = cbs["image_logger"]
    assert type(il).__name__ == "ImageLogger" and il.batch_freq == 1000 and il.max_images == 1
    assert il.log_steps == [2 ** n for n in range(10)]


class _TinyVal(torch.nn.Module):
    """A LightningModule-shaped toy: one optimizer, a validation_step that logs a scripted `val/rec_loss`."""

    def __new__(cls, *a, **k):
        from odvae_amd.lightning import LightningModule

        class Impl(LightningModule):
            def __init__(self, scores):
                super().__init__()
                self.lin = torch.nn.Linear(4, 4)
                self.learning_rate = 1e-2
                self.monitor = "val/rec_loss"
                self.scores = list(scores)

            def training_step(self, batch, batch_idx, optimizer_idx):
                return self.lin(batch).pow(2).mean()

            def validation_step(self, batch, batch_idx):
                s = self.scores[0]
                self.log("val/rec_loss", torch.tensor(s + 0.1 * batch_idx), sync_dist=True)
                self.log("val/other", 1.0)
                return None

            def configure_optimizers(self):
                return [torch.optim.Adam(self.lin.parameters(), lr=self.learning_rate, betas=(0.5, 0.9))], []
        return Impl(*a, **k)


def test_modelcheckpoint_keeps_the_three_best_and_last(tmp_path):
    """train.py:228-249: filename {epoch:06}, save_last, save_weights_only, monitor = model.monitor, save_top_k 3 (mode min).  Five
    epochs with scripted validation means 0.55, 0.35, 0.75, 0.25, 0.65 (two batches each: s, s + 0.1): the files on disk are always the
    three best epochs + last.ckpt, a worse epoch writes only last.ckpt, `epoch` / `global_step` in the files follow the loop."""
    from odvae_amd.callbacks import default_modelcheckpoint
    from odvae_amd.config import Config, instantiate_from_config
    from odvae_amd.trainer import Trainer
    scores = [0.5, 0.3, 0.7, 0.2, 0.6]
    model = _TinyVal(scores)
    ckdir = os.path.join(tmp_path, "checkpoints")
    cb = default_modelcheckpoint(model, ckdir)
    assert cb.monitor == "val/rec_loss" and cb.save_top_k == 3 and cb.save_last and cb.save_weights_only
    trainer = Trainer(model, optimizer_indices=(0,), callbacks=[cb])
    train = [torch.randn(2, 4) for _ in range(3)]
    val = [torch.randn(2, 4) for _ in range(2)]
    seen = []
    for epoch in range(5):
        trainer.fit(train, val_batches=val, max_epochs=1)
        model.scores.pop(0)
        seen.append(sorted(os.listdir(ckdir)))
        assert abs(float(trainer.callback_metrics["val/rec_loss"]) - (scores[epoch] + 0.05)) < 1e-6     # the epoch MEAN over two batches
        last = torch.load(os.path.join(ckdir, "last.ckpt"))
        assert last["epoch"] == epoch and last["global_step"] == 3 * (epoch + 1) and "optimizer_states" not in last
    assert seen[0] == ["epoch=000000.ckpt", "last.ckpt"]
    assert seen[2] == ["epoch=000000.ckpt", "epoch=000001.ckpt", "epoch=000002.ckpt", "last.ckpt"]
    assert seen[3] == ["epoch=000000.ckpt", "epoch=000001.ckpt", "epoch=000003.ckpt", "last.ckpt"]      # 0.75 dropped out
    assert seen[4] == seen[3]                                                                           # 0.65 is not among the best three
    assert cb.best_model_path.endswith("epoch=000003.ckpt") and abs(cb.best_model_score - 0.25) < 1e-6
    assert cb.kth_best_model_path.endswith("epoch=000000.ckpt")
    # the same callback through the reference's own config route (target: pytorch_lightning.callbacks.ModelCheckpoint, train.py:229)
    cfg = Config({"target": "pytorch_lightning.callbacks.ModelCheckpoint",
                  "params": {"dirpath": ckdir, "filename": "{epoch:06}", "verbose": True, "save_last": True, "save_weights_only": True}})
    assert type(instantiate_from_config(cfg)).__name__ == "ModelCheckpoint"
    # full checkpoint + resume: weights, optimizer moments, counters
    path = trainer.save_checkpoint(os.path.join(tmp_path, "full.ckpt"))
    model2 = _TinyVal(scores)
    trainer2 = Trainer(model2, optimizer_indices=(0,))
    trainer2.load_checkpoint(path)
    assert model2.global_step == 15 and trainer2.current_epoch == 5
    assert torch.equal(model2.lin.weight, model.lin.weight)
    s1, s2 = trainer.optimizers[0].state_dict()["state"], trainer2.optimizers[0].state_dict()["state"]
    assert all(torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) and float(s1[k]["step"]) == float(s2[k]["step"]) == 15.0 for k in s1)


def test_validation_means_are_weighted_by_batch_size():
    """[PL-1.9] reduces an on_epoch `self.log` inside validation_step as sum(value * batch_size) / sum(batch_size) (ResultMetric), not as the
    plain mean over batches: with a ragged last batch the two differ, and the monitored `val/rec_loss` decides which checkpoints survive
    (train.py:238-241).  Three batches of 4, 4 and 1 samples with per-batch values 1.0, 2.0, 10.0: PL's mean is 22 / 9, the plain one 13 / 3."""
    from odvae_amd.trainer import Trainer, _batch_size
    model = _TinyVal([1.0])
    vals = iter([1.0, 2.0, 10.0])

    def validation_step(batch, batch_idx, _m=model):
        v = torch.tensor(next(vals))
        _m.log("val/rec_loss", v, sync_dist=True)
        _m.log_dict({"val/other": 2 * v})
    model.validation_step = validation_step
    trainer = Trainer(model, optimizer_indices=(0,))
    got = trainer.validate([torch.randn(4, 4), torch.randn(4, 4), torch.randn(1, 4)])
    assert abs(float(got["val/rec_loss"]) - 22.0 / 9.0) < 1e-6 and abs(float(got["val/other"]) - 44.0 / 9.0) < 1e-6
    assert _batch_size({"class_name": ["car"] * 3, "patch": torch.zeros(5, 3, 8, 8), "yaw": torch.zeros(7)}) == 5
    assert _batch_size([{"a": torch.zeros(6, 2)}, torch.zeros(2)]) == 6 and _batch_size({"names": ["x"]}) == 1


def test_load_checkpoint_refuses_pickled_objects_unless_trusted(tmp_path):
    """torch >= 2.6 loads with weights_only=True by default; a Lightning checkpoint may carry arbitrary objects (hyper_parameters as an OmegaConf
    DictConfig, callback state).  load_checkpoint reads tensor-only files as is and unpickles objects only when the caller says `trusted`."""
    import pathlib
    import pickle
    from odvae_amd.trainer import Trainer
    model = _TinyVal([0.5])
    trainer = Trainer(model, optimizer_indices=(0,))
    ckpt = trainer.dump_checkpoint()
    # the full PL-1.9 key set around it: loops, callbacks keyed by PL's ModelCheckpoint state_key, hyper_parameters
    ckpt["loops"] = {"fit_loop": {"state_dict": {}, "epoch_progress": {"current": {"completed": 3}}}}
    ckpt["callbacks"] = {"ModelCheckpoint{'monitor': 'val/rec_loss', 'mode': 'min', 'every_n_train_steps': 0, 'every_n_epochs': 1, "
                         "'train_time_interval': None}": {"best_model_score": torch.tensor(0.25), "best_model_path": "/logs/x/epoch=000003.ckpt"}}
    ckpt["hyper_parameters"] = {"embed_dim": 16, "monitor": "val/rec_loss"}
    plain = os.path.join(tmp_path, "plain.ckpt")
    torch.save(ckpt, plain)
    trainer.load_checkpoint(plain)                                   # tensors, numbers, strings, containers: the restricted unpickler takes it
    ckpt["callbacks"]["Timer"] = {"where": pathlib.PurePosixPath("/logs/x")}    # an object of a class outside torch's allow-list
    objs = os.path.join(tmp_path, "objects.ckpt")
    torch.save(ckpt, objs)
    with pytest.raises(pickle.UnpicklingError, match="trusted=True"):
        trainer.load_checkpoint(objs)
    trainer.load_checkpoint(objs, trusted=True)
    assert model.global_step == ckpt["global_step"]

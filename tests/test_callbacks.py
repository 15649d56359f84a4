"""ImageLogger (src/util/callbacks.py:78-218) on the host: the power-of-two logging schedule, file naming and
pixel values of the grids, torchvision's make_grid layout rule, and that every `target:` under the untouched yaml's
`lightning.callbacks` (yaml:115-131) resolves through the src.* shims.  A fake module stands in for the model (CPU tensors)."""
import os

import numpy as np
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


def test_yaml_callbacks_resolve():
    from odvae_amd.config import Config, instantiate_from_config
    cfg = Config.load(YAML)
    cbs = {k: instantiate_from_config(v) for k, v in cfg.lightning.callbacks.items()}
    assert set(cbs) == {"image_logger", "progress_bar", "device_stats_monitor"}
    il = cbs["image_logger"]
    assert type(il).__name__ == "ImageLogger" and il.batch_freq == 1000 and il.max_images == 1
    assert il.log_steps == [2 ** n for n in range(10)]

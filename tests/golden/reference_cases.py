"""Case definitions shared by tests/golden/make_reference_goldens.py (which runs the REFERENCE's Python on them in the build container)
and the tests that replay them on the oracle and on the HIP path: seeded inputs are re-made here instead of being stored, so the committed
fixture (reference_glue.npz) holds expected outputs only."""
import math

import numpy as np
import torch

LOSS_KW = dict(encoder_pretrain_steps=10, pose_conditioned_generation_steps=20, disc_start=15, kl_weight_obj=1.0, kl_weight_bbox=1e-6,
               disc_weight=0.5, pose_weight=80000, fill_factor_weight=500000, class_weight=1000000, bbox_weight=200000, pose_loss_fn="l1",
               mask_weight=0, mask_loss_fn="l2", disc_in_channels=3, num_classes=11, perceptual_weight=1.0, disc_factor=1.0, train_on_yaw=True)
# (name, global_step, class ids): the three regimes of contperceptual.py:222-224 / :307-321 on either side of disc_start, a batch with
# the masked class id 1 in it, a batch of nothing but id 1 (every masked mean takes its `else 0` branch)
LOSS_CASES = [("pose_only_step5", 5, [0, 1, 3, 1]), ("rec_no_pixel_step12", 12, [0, 1, 3, 1]), ("rec_no_pixel_disc_on_step20", 20, [0, 2, 3, 1]),
              ("vae_step40", 40, [0, 1, 3, 9]), ("vae_step40_all_masked", 40, [1, 1, 1, 1]), ("pose_only_step5_all_masked", 5, [1, 1, 1, 1]),
              ("at_threshold_step10", 10, [0, 0, 5, 1]), ("at_threshold_step30", 30, [4, 1, 0, 7])]
LABELS = ['car', 'truck', 'trailer', 'bus', 'construction_vehicle', 'bicycle', 'motorcycle', 'pedestrian', 'traffic_cone', 'barrier',
          'background']


def stats_table():
    """Stand-in for dataset_stats/combined/all.pkl (not shipped): {label: {"t3","l","h","w": tensor([mean, logvar])}} (contperceptual.py:84-104)."""
    g = torch.Generator().manual_seed(77)
    return {lab: {k: torch.randn(2, generator=g) * 0.3 for k in ("t3", "l", "h", "w")} for lab in LABELS}


def loss_inputs(seed, class_ids):
    g = torch.Generator().manual_seed(seed)
    B, H = 4, 64
    d = {"rgb_gt": torch.rand(B, 3, H, H, generator=g) * 2 - 1, "feat": torch.randn(B, 8, H, H, generator=g),
         "last_w": torch.randn(3, 8, 3, 3, generator=g) * 0.1, "last_b": torch.randn(3, generator=g) * 0.1,
         "dec_pose": torch.randn(B, 19, generator=g), "pose_gt": torch.randn(B, 4, generator=g), "bbox_gt": torch.randn(B, 3, generator=g),
         "fill_factor_gt": torch.rand(B, generator=g), "moments": torch.randn(B, 32, 4, 4, generator=g) * 0.5,
         "bbox_moments": torch.randn(B, 16, generator=g) * 0.5, "class_id": torch.tensor(class_ids, dtype=torch.int64)}
    mask = torch.zeros(B, 1, H, H)
    for i in range(B):   # a different box per sample, one of them the whole image
        x0, y0 = 4 * i, 6 * i
        mask[i, :, y0:H - 3 * i, x0:H - 5 * i] = 1.0
    d["mask_2d_bbox"] = mask
    return d




def step_batch(B=4, height=64):
    """The batch of the training_step / validation_step cases: SURVEY.md 8(b) schema, one sample of the masked class id 1, two partial boxes."""
    g = torch.Generator().manual_seed(5)
    ids = [0, 1, 3, 0][:B]
    b = {"patch": torch.rand(B, 3, height, height, generator=g), "pose_6d": torch.randn(B, 4, generator=g),
         "yaw": (torch.rand(B, generator=g) * 2 - 1) * math.pi, "class_id": torch.tensor(ids, dtype=torch.int64),
         "class_name": [LABELS[c] for c in ids], "bbox_sizes": torch.randn(B, 3, generator=g), "fill_factor": torch.rand(B, generator=g),
         "mask_2d_bbox": torch.ones(B, 1, height, height), "pose_6d_perturbed": torch.randn(B, 1, 4, generator=g),
         "yaw_perturbed": (torch.rand(B, generator=g) * 2 - 1) * math.pi}
    b["mask_2d_bbox"][1, :, :20, :] = 0.0
    b["mask_2d_bbox"][2, :, :, 40:] = 0.0
    return b


def digest_of(t, limit=4096, keep=2048):
    """(norm, samples): the L2 norm in f64 and -- for a tensor above `limit` elements -- every stride-th element, else all of it."""
    flat = t.detach().cpu().reshape(-1)
    stride = 1 if flat.numel() <= limit else int(math.ceil(flat.numel() / keep))
    return np.float64(flat.double().norm().item()), flat[::stride].numpy().copy()


def digest(arrays, key, t):
    arrays[key + ".norm"], arrays[key + ".samples"] = digest_of(t)


# ---- patch extraction (src/data/datasets/nuscenes.py:90-194 `_generate_patch`) --------------------------------------------------------
PATCH_S = 96            # network resolution of the fixture (256 in the yaml; the arithmetic is the same, the fixture 7x smaller)
PATCH_IMAGE_HW = (300, 420)


def patch_image():
    """The synthetic camera image of the patch fixtures: half smooth ramps, half noise, seeded."""
    h, w = PATCH_IMAGE_HW
    rng = np.random.default_rng(2024)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    return np.where(rng.random((h, w, 1)) < 0.5, smooth, noise).astype(np.uint8)


def patch_instances():
    """(bbox [x1, y1, x2, y2], center_2d) of the fixtures: ordinary boxes, boxes over every image border, a projected centre outside the image
    (dropped), a box wholly outside with its centre inside (the reference's corner case, :117-136), wide and tall boxes (padding pixels), an
    extent of every PATCH_SIZES snap class, fractional coordinates, an empty box."""
    h, w = PATCH_IMAGE_HW
    rng = np.random.default_rng(77)
    out = [([100.2, 80.7, 180.9, 150.1], [140.5, 115.3]), ([-20.5, -10.0, 60.0, 45.5], [19.7, 17.2]), ([380.0, 250.0, 470.0, 330.0], [419.0, 289.6]),
           ([10.0, 200.0, 130.0, 260.0], [70.0, 230.0]), ([200.0, 20.0, 240.0, 170.0], [220.4, 95.0]), ([150.0, 100.0, 250.0, 200.0], [-3.0, 150.0]),
           ([150.0, 100.0, 250.0, 200.0], [200.0, 300.0]), ([430.0, 100.0, 500.0, 160.0], [415.0, 130.0]), ([-90.0, 40.0, -10.0, 100.0], [2.0, 70.0]),
           ([50.0, 50.0, 50.0, 50.0], [50.0, 50.0]), ([5.0, 5.0, 185.0, 150.0], [95.0, 77.5]), ([300.0, 150.0, 420.0, 299.0], [360.0, 224.5]),
           ([120.9, 130.9, 170.1, 181.1], [145.5, 156.0]), ([0.0, 0.0, 420.0, 300.0], [210.0, 150.0])]
    for _ in range(26):
        cx, cy = rng.uniform(-10, w + 10), rng.uniform(-10, h + 10)
        ex, ey = rng.uniform(4, 170), rng.uniform(4, 170)
        out.append(([float(cx - ex / 2), float(cy - ey / 2), float(cx + ex / 2), float(cy + ey / 2)],
                    [float(cx + rng.uniform(-6, 6)), float(cy + rng.uniform(-6, 6))]))
    return out

"""Object-patch extraction on the GPU: the step right before the hot path.

The reference cuts one square patch per annotated object out of a decoded camera image with PIL, resizes it to the
network resolution and rasterises the object's 2-d box into a mask (src/data/datasets/nuscenes.py:90-194, one Python call
per instance inside DataLoader workers).  Here the decoded u8 camera images live in HBM and one kernel launch produces the
whole batch: `patch` [B,3,S,S] f32 in [0,1] (channels_last) and `mask_2d_bbox` [B,1,S,S], bit-identical to the PIL path
(csrc/patch_u8.hip).  JPEG decoding, annotation parsing and the pose targets stay where they are (out of scope).

Host side, per instance (plain integer arithmetic, no device sync):
  * `plan_patch`   -- which square to cut: nuscenes.py:97-158 (centre rejection, square box around the floored projected
                      centre, snapping to PATCH_SIZES under `perturb_scale`, the four border clamps, padding pixels)
  * `mask_slice`   -- numpy's slice rule for `mask_bool[y1:y2, x1:x2] = True` (:178-187; negative starts wrap, as there)
  * `resample_table` -- Pillow's coefficient windows for (crop size -> S), cached per crop size
"""
import math

import numpy as np
import torch

from . import lib as _lib

PATCH_SIZES = (50, 100, 200, 400)   # nuscenes.py:55
PRECISION_BITS = 32 - 8 - 2         # Pillow Resample.c (8 bits per channel)


def snap_to_patch_size(extent):
    """Closest entry of PATCH_SIZES, the first one on ties (`list.index(min(...))`, nuscenes.py:129-130,140-141)."""
    best = PATCH_SIZES[0]
    for s in PATCH_SIZES[1:]:
        if abs(extent - s) < abs(extent - best):
            best = s
    return best


class PatchPlan:
    """Crop square [x1, x1+size) x [y1, y1+size) in camera-image pixels + what the caller derives from it."""
    __slots__ = ("x1", "y1", "size", "padding_pixels", "mask_x", "mask_y")

    def __init__(self, x1, y1, size, padding_pixels, mask_x, mask_y):
        self.x1, self.y1, self.size, self.padding_pixels, self.mask_x, self.mask_y = x1, y1, size, padding_pixels, mask_x, mask_y


def mask_slice(lo, hi, n):
    """(start, stop) that `array[lo:hi]` addresses on an axis of length n (Python slice semantics, step 1)."""
    start, stop, _ = slice(lo, hi).indices(n)
    return start, max(stop, start)


def plan_patch(bbox, center_2d, img_w, img_h, perturb_scale):
    """The crop the reference takes for one object, or None where it drops the instance (nuscenes.py:97-166).
    bbox = [x1, y1, x2, y2] floats (exterior rectangle of the projected 3-d box), center_2d = projected centre."""
    cx, cy = float(center_2d[0]), float(center_2d[1])
    if cx < 0 or cy < 0 or cx >= img_w or cy >= img_h:          # :102-103  less than half of the object visible
        return None
    bx1, by1, bx2, by2 = (int(v) for v in bbox)                   # :109  truncation toward zero
    width, height = bx2 - bx1, by2 - by1
    size = max(width, height)
    ccx, ccy = int(math.floor(cx)), int(math.floor(cy))
    outside = bx1 >= img_w or by1 >= img_h or bx2 <= 0 or by2 <= 0
    if outside:                                                   # :117-136  only width/height survive this branch
        width = min(img_w, bx2) - max(0, bx1)
        height = min(img_h, by2) - max(0, by1)
    elif perturb_scale:                                           # :138-150
        size = snap_to_patch_size(size)
        half = size // 2
        if ccx - half < 0:
            ccx = half
        if ccy - half < 0:
            ccy = half
        if ccx + half > img_w:
            ccx = img_w - half
        if ccy + half > img_h:
            ccy = img_h - half
    half = size // 2
    side = 2 * half                                               # :157-160  (c - half, c + half)
    if side <= 0:                                                 # PIL raises / divides by zero -> instance dropped (:164-174)
        return None
    x1, y1 = ccx - half, ccy - half
    pad = width - height if width > height else 0                 # :152-155
    # :181-187  bbox corners relative to the crop, truncated, then numpy slice assignment on a (side, side) array
    mx = mask_slice(int(float(bbox[0]) - x1), int(float(bbox[2]) - x1), side)
    my = mask_slice(int(float(bbox[1]) - y1), int(float(bbox[3]) - y1), side)
    return PatchPlan(x1, y1, side, pad, mx, my)


def resample_table(in_size, out_size):
    """int32 [out_size][8] = {k0..k4, first source index, taps, nearest source index}: Pillow's BILINEAR windows for
    in_size -> out_size (Resample.c precompute_coeffs + normalize_coeffs_8bpc, evaluated in f64 in the same operation
    order) and the NEAREST source index (Geometry.c ImagingScaleAffine: xo = scale/2, then xo += scale per step)."""
    if in_size >= 2 * out_size:
        raise ValueError("crop %d -> %d: Image.resize(reducing_gap=1.0) would box-reduce first; not supported" % (in_size, out_size))
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                                    # bilinear support 1.0
    inv = 1.0 / filterscale
    idx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (idx + 0.5) * scale
    first = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    last = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size)
    taps = last - first
    assert taps.max() <= 5
    w = np.zeros((out_size, 5), np.float64)
    total = np.zeros(out_size, np.float64)
    for t in range(5):
        a = np.abs(((t + first).astype(np.float64) - center + 0.5) * inv)
        wt = np.where((a < 1.0) & (t < taps), 1.0 - a, 0.0)
        w[:, t] = wt
        total = total + wt                                        # left-to-right accumulation, as the C loop
    w = np.where(total[:, None] != 0.0, w / np.where(total == 0.0, 1.0, total)[:, None], w)
    k = np.trunc(0.5 + w * float(1 << PRECISION_BITS)).astype(np.int64)
    tab = np.zeros((out_size, 8), np.int32)
    tab[:, :5] = k
    tab[:, 5] = first
    tab[:, 6] = taps
    xo = 0.0 + scale * 0.5
    for i in range(out_size):
        tab[i, 7] = int(xo)
        xo += scale
    return tab


class PatchBatch:
    """What `_generate_patch` returns, batched: patch [n,3,S,S], mask [n,1,S,S] (device), patch_size [n,2] f32 (w, h) of
    the crop, resampling_factor [(fx, fy)], padding_pixels_resampled [n] and `kept` = indices of the instances that
    were not dropped."""
    __slots__ = ("patch", "mask", "patch_size", "resampling_factor", "padding_pixels_resampled", "kept", "plans")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


class GpuPatcher:
    def __init__(self, patch_height=256, patch_aspect_ratio=1.0, perturb_scale=False, device="cuda:0"):
        self.size = (patch_height, int(patch_height * patch_aspect_ratio))   # nuscenes.py:67
        if self.size[0] != self.size[1]:
            raise ValueError("the reference asserts equal resampling factors (nuscenes.py:172): square patches only")
        self.S = int(patch_height)
        self.perturb_scale = bool(perturb_scale)
        self.device = torch.device(device)
        self._slot = {}
        self._host_tables = []
        self._tables = None

    def _table_slot(self, crop):
        slot = self._slot.get(crop)
        if slot is None:
            slot = self._slot[crop] = len(self._host_tables)
            self._host_tables.append(resample_table(crop, self.S))
            self._tables = None
        return slot

    def _device_tables(self):
        if self._tables is None:
            self._tables = torch.from_numpy(np.stack(self._host_tables)).to(self.device)
        return self._tables

    def __call__(self, images, instances):
        """images: list of u8 [H,W,3] tensors on the device (decoded camera images); instances: iterable of
        (image_index, bbox[4], center_2d[2]).  One launch for all kept instances."""
        staged = self.stage(images, instances)
        return self.launch(staged) if staged["n"] else PatchBatch(
            patch=None, mask=None, patch_size=None, resampling_factor=[], padding_pixels_resampled=[], kept=[], plans=[])

    def stage(self, images, instances):
        """Host half: plan every instance and ship pointers, geometry and mask rectangles in one H2D copy."""
        plans, kept = [], []
        for i, (img_idx, bbox, center) in enumerate(instances):
            img = images[img_idx]
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous() or not 6 <= img.numel() < 2 ** 31:
                raise ValueError("camera images must be contiguous u8 [H,W,3] tensors")
            if not img.is_cuda:
                raise _lib.HipLibraryError("camera images must live on the HIP device (no CPU fallback), got %s" % img.device)
            plan = plan_patch(bbox, center, img.shape[1], img.shape[0], self.perturb_scale)
            if plan is not None:
                plans.append((img_idx, plan))
                kept.append(i)
        n = len(plans)
        if n == 0:
            return {"n": 0}
        # staging buffer: [n] pointers (8 B) | [n][8] geometry | [n][4] mask rectangle
        ptr_bytes = (8 * n + 15) // 16 * 16
        host = np.zeros(ptr_bytes + 32 * n + 16 * n, np.uint8)
        ptrs = host[:8 * n].view(np.int64)
        geom = host[ptr_bytes:ptr_bytes + 32 * n].view(np.int32).reshape(n, 8)
        rect = host[ptr_bytes + 32 * n:].view(np.int32).reshape(n, 4)
        for j, (img_idx, plan) in enumerate(plans):
            img = images[img_idx]
            ptrs[j] = img.data_ptr()
            geom[j, :6] = (img.shape[0], img.shape[1], plan.x1, plan.y1, plan.size, self._table_slot(plan.size))
            rect[j] = (plan.mask_x[0], plan.mask_x[1], plan.mask_y[0], plan.mask_y[1])
        return {"n": n, "ptr_bytes": ptr_bytes, "dev": torch.from_numpy(host).to(self.device, non_blocking=True),
                "tables": self._device_tables(), "plans": [p for _, p in plans], "kept": kept,
                "images": [images[k] for k, _ in plans]}   # keeps the camera images alive until the launch is issued

    def launch(self, staged):
        """Device half: one kernel launch on the current stream."""
        L = _lib.load()
        n, S, plans = staged["n"], self.S, staged["plans"]
        patch = torch.empty((n, S, S, 3), dtype=torch.float32, device=self.device)
        mask = torch.empty((n, 1, S, S), dtype=torch.float32, device=self.device)
        base, tables = staged["dev"].data_ptr(), staged["tables"]
        _lib.check(L.odvae_patch_crop_resize_u8(base, base + staged["ptr_bytes"], base + staged["ptr_bytes"] + 32 * n,
                                                tables.data_ptr(), tables.shape[0], n, S, patch.data_ptr(), mask.data_ptr(),
                                                _lib.stream_ptr()), "patch_crop_resize")
        factor = [(S / p.size, S / p.size) for p in plans]
        return PatchBatch(patch=patch.permute(0, 3, 1, 2), mask=mask,
                          patch_size=torch.tensor([[p.size, p.size] for p in plans], dtype=torch.float32),
                          resampling_factor=factor,
                          padding_pixels_resampled=[p.padding_pixels * f[0] for p, f in zip(plans, factor)],
                          kept=staged["kept"], plans=plans)

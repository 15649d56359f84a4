"""ORACLE (test infrastructure; only tests/ may import this): the reference's per-instance patch extraction,
src/data/datasets/nuscenes.py:90-194 (`_generate_patch`), restated on CPU.

Pixel arithmetic: PINNED.  The reference does it with Pillow (`Image.crop`, `Image.resize(BILINEAR, reducing_gap=1.0)`,
`Image.resize(NEAREST)`) and torchvision's ToTensor; Pillow 12.2.0 is importable in this image, so `generate_patch_pil`
calls the very same library entry points, and `pillow_bilinear_u8` / `pillow_nearest_index` restate Pillow's 8-bit
fixed-point resampler (Resample.c) and nearest mapping (Geometry.c ImagingScaleAffine) in numpy; tests/test_patches.py
checks the restatement against Pillow itself.  torchvision is absent: ToTensor is restated as u8 -> f32 -> / 255 (its
documented behaviour; bool masks become 0/1).
Box arithmetic (which square to cut, which instances to drop, padding pixels, the mask slice): PINNED since round 5 -- tests/golden/
make_reference_patch_goldens.py runs the reference's own `NuScenesBase._generate_patch` (nuscenes.py imported unmodified, with stand-ins for the
absent mmdet3d / pytorch3d / torchvision names; Pillow is the real library) on 40 instances incl. every border and the corner case, and
tests/test_patches.py holds this file to its outputs bit for bit (tests/test_patches_gpu.py the device path).
"""
import math

import numpy as np

PATCH_SIZES = [50, 100, 200, 400]   # nuscenes.py:55
PRECISION_BITS = 32 - 8 - 2


def to_tensor(arr):
    """torchvision.transforms.ToTensor on a PIL image given as numpy: HWC u8 -> CHW f32 / 255; bool (mode "1") -> 0/1."""
    a = np.asarray(arr)
    if a.dtype == bool:
        a = a.astype(np.uint8) * 255
    if a.ndim == 2:
        a = a[:, :, None]
    return (a.astype(np.float32) / np.float32(255)).transpose(2, 0, 1)


def generate_patch_pil(img_u8, bbox, center_2d, patch_size=(256, 256), perturb_scale=False):
    """nuscenes.py:90-194 with PIL doing the pixel work.  img_u8: HWC u8 array (the decoded camera image).
    Returns (patch CHW f32, patch_size_sq [2] f32, resampling_factor, padding_pixels_resampled, mask 1HW f32) or five
    Nones where the reference drops the instance."""
    from PIL import Image
    img_pil = Image.fromarray(img_u8)
    none5 = (None, None, None, None, None)
    if center_2d[0] < 0 or center_2d[1] < 0 or center_2d[0] >= img_pil.size[0] or center_2d[1] >= img_pil.size[1]:
        return none5
    x1, y1, x2, y2 = bbox
    corner_case = False
    try:
        x1, y1, x2, y2 = int(x1), int(y1), int(x2), int(y2)
        width, height = x2 - x1, y2 - y1
        floored_center = np.floor(center_2d).astype(np.int32)
        box_size = max(int(width), int(height))
        if x1 >= img_pil.size[0] or y1 >= img_pil.size[1] or x2 <= 0 or y2 <= 0:
            corner_case = True
            x1, y1 = max(0, x1), max(0, y1)
            x2, y2 = min(img_pil.size[0], x2), min(img_pil.size[1], y2)
            width, height = x2 - x1, y2 - y1
            # (:125-136 recompute x1..y2 and center_2d from a snapped size; every one of them is overwritten or unused below)
        if perturb_scale and not corner_case:
            diffs = [abs(box_size - s) for s in PATCH_SIZES]
            box_size = PATCH_SIZES[diffs.index(min(diffs))]
            if floored_center[0] - box_size // 2 < 0:
                floored_center[0] = box_size // 2
            if floored_center[1] - box_size // 2 < 0:
                floored_center[1] = box_size // 2
            if floored_center[0] + box_size // 2 > img_pil.size[0]:
                floored_center[0] = img_pil.size[0] - box_size // 2
            if floored_center[1] + box_size // 2 > img_pil.size[1]:
                floored_center[1] = img_pil.size[1] - box_size // 2
        padding_pixels = int(width) - int(height) if int(width) > int(height) else 0
        y1 = floored_center[1] - box_size // 2
        y2 = floored_center[1] + box_size // 2
        x1 = floored_center[0] - box_size // 2
        x2 = floored_center[0] + box_size // 2
        patch = img_pil.crop((x1, y1, x2, y2))
        patch_size_sq = np.asarray(patch.size, np.float32)
    except Exception:
        return none5
    rw, rh = patch_size
    try:
        resampling_factor = (rw / patch.size[0], rh / patch.size[1])
        assert resampling_factor[0] == resampling_factor[1]
    except ZeroDivisionError:
        return none5
    patch_resized = patch.resize((rw, rh), resample=Image.Resampling.BILINEAR, reducing_gap=1.0)
    mask_bool = np.zeros((patch.size[1], patch.size[0]), dtype=bool)
    x1_full, y1_full, x2_full, y2_full = bbox
    mask_bool[int(y1_full - y1):int(y2_full - y1), int(x1_full - x1):int(x2_full - x1)] = True
    mask_resized = Image.fromarray(mask_bool).resize((rw, rh), resample=Image.Resampling.NEAREST, reducing_gap=1.0)
    return (to_tensor(np.asarray(patch_resized)), patch_size_sq, resampling_factor, padding_pixels * resampling_factor[0],
            to_tensor(np.asarray(mask_resized)))


# ---- Pillow's resampler restated (scalar loops; used on small cases) ------------------------------------------------
def pillow_bilinear_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs (triangle filter, support 1) + normalize_coeffs_8bpc."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    kk = np.zeros((out_size, ksize), np.int64)
    bounds = np.zeros((out_size, 2), np.int64)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w, ww = [], 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w.append(1.0 - a if a < 1.0 else 0.0)
            ww += w[-1]
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def pillow_bilinear_u8(img, out_size):
    """ImagingResample for 8-bit images: horizontal pass (rounded to u8), then vertical pass."""
    h, w, c = img.shape
    if (h, w) == (out_size, out_size):
        return img.copy()
    kx, bx = pillow_bilinear_coeffs(w, out_size)
    ky, by = pillow_bilinear_coeffs(h, out_size)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.zeros((h, out_size, c), np.uint8)
    for xx in range(out_size):
        x0, n = bx[xx]
        acc = np.full((h, c), half, np.int64)
        for x in range(n):
            acc += img[:, x0 + x, :].astype(np.int64) * kx[xx, x]
        tmp[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((out_size, out_size, c), np.uint8)
    for yy in range(out_size):
        y0, n = by[yy]
        acc = np.full((out_size, c), half, np.int64)
        for y in range(n):
            acc += tmp[y0 + y].astype(np.int64) * ky[yy, y]
        out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def pillow_nearest_index(in_size, out_size):
    """Geometry.c ImagingScaleAffine: source index per output index for Image.resize(NEAREST)."""
    step = float(in_size) / out_size
    xo = 0.0 + step * 0.5
    idx = []
    for _ in range(out_size):
        idx.append(-1 if xo < 0.0 else int(xo))
        xo += step
    return np.asarray(idx)

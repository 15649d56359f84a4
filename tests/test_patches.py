"""CPU checks for the patch-extraction row: the oracle's restatement of Pillow's resampler against
Pillow itself (the reference's own dependency, nuscenes.py:159-192), and the product's host logic (crop planning, mask
slice rule, coefficient tables) against the oracle.  No GPU: the kernel's arithmetic is emulated here from the product's
tables only to validate those tables."""
import numpy as np
import pytest
import torch

from oracle import patches as oracle

PIL_Image = pytest.importorskip("PIL.Image")


def _pil_bilinear(img, S):
    return np.asarray(PIL_Image.fromarray(img).resize((S, S), resample=PIL_Image.Resampling.BILINEAR, reducing_gap=1.0))


@pytest.mark.parametrize("S", [256, 96])
def test_oracle_bilinear_restatement_matches_pillow(S):
    rng = np.random.default_rng(0)
    for size in [50, 100, 200, 400, 37, 255, 257, 300, 2 * S - 1, 3, S, 129]:
        if size >= 2 * S:
            continue
        img = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
        assert np.array_equal(oracle.pillow_bilinear_u8(img, S), _pil_bilinear(img, S)), size


@pytest.mark.parametrize("S", [256, 200, 512, 96])
def test_oracle_nearest_index_matches_pillow(S):
    for size in list(range(1, 64)) + [100, 199, 200, 201, 255, 256, 257, 400, 511, 513, 777]:
        x = np.arange(size)
        lo = np.tile((x & 255).astype(np.uint8), (2, 1))
        hi = np.tile((x >> 8).astype(np.uint8), (2, 1))
        ref = [np.asarray(PIL_Image.fromarray(a).resize((S, 2), resample=PIL_Image.Resampling.NEAREST, reducing_gap=1.0))[0]
               for a in (lo, hi)]
        assert np.array_equal(oracle.pillow_nearest_index(size, S), ref[0].astype(int) + 256 * ref[1].astype(int)), size


def test_to_tensor_division_matches_torch_cpu():
    v = np.arange(256, dtype=np.uint8)
    ref = torch.from_numpy(v.copy()).to(torch.float32).div(255).numpy()
    assert np.array_equal(oracle.to_tensor(v.reshape(16, 16, 1))[0].reshape(-1), ref)


@pytest.mark.parametrize("S", [256, 512, 96])
def test_product_tables_match_oracle(S):
    from odvae_amd.patches import resample_table
    for size in [1, 2, 3, 37, 50, 100, 200, 255, 256, 257, 400, 2 * S - 1]:
        if size >= 2 * S:
            continue
        tab = resample_table(size, S)
        kk, bounds = oracle.pillow_bilinear_coeffs(size, S)
        assert kk.shape[1] <= 5
        assert np.array_equal(tab[:, :kk.shape[1]], kk) and not tab[:, kk.shape[1]:5].any(), size
        assert np.array_equal(tab[:, 5:7], bounds), size
        assert np.array_equal(tab[:, 7], oracle.pillow_nearest_index(size, S)), size
    with pytest.raises(ValueError):
        resample_table(2 * S, S)


def _emulate(img, plan, tab, S):
    """The kernel's integer arithmetic (csrc/patch_u8.hip) from the product's table, in numpy: validates the tables/plan."""
    h, w, _ = img.shape
    crop = np.zeros((plan.size, plan.size, 3), np.int64)
    ys, xs = np.arange(plan.size) + plan.y1, np.arange(plan.size) + plan.x1
    vy, vx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
    crop[np.ix_(vy, vx)] = img[np.ix_(ys[vy], xs[vx])]
    half = 1 << 21
    hor = np.full((plan.size, S, 3), half, np.int64)
    for t in range(5):
        src = np.minimum(tab[:, 5] + t, plan.size - 1)
        hor += crop[:, src, :] * tab[None, :, t, None]
    hor = np.clip(hor >> 22, 0, 255)
    out = np.full((S, S, 3), half, np.int64)
    for t in range(5):
        src = np.minimum(tab[:, 5] + t, plan.size - 1)
        out += hor[src] * tab[:, t, None, None]
    out = np.clip(out >> 22, 0, 255).astype(np.uint8)
    near = tab[:, 7]
    my = (near >= plan.mask_y[0]) & (near < plan.mask_y[1])
    mx = (near >= plan.mask_x[0]) & (near < plan.mask_x[1])
    return out, (my[:, None] & mx[None, :])


def random_instances(rng, n, img_w, img_h, max_extent=480):
    out = []
    for _ in range(n):
        w, h = rng.uniform(4, max_extent), rng.uniform(4, max_extent)
        cx, cy = rng.uniform(-30, img_w + 30), rng.uniform(-30, img_h + 30)
        jitter = rng.uniform(-0.2, 0.2, 2) * (w, h)
        bbox = [cx - w / 2 + jitter[0], cy - h / 2 + jitter[1], cx + w / 2 + jitter[0], cy + h / 2 + jitter[1]]
        out.append((bbox, [cx, cy]))
    # hand-made edge cases: box hanging over each border, box wholly outside with the centre inside, tiny box
    out += [([-40.5, 10.2, 80.7, 90.1], [20.0, 50.0]), ([img_w - 60.3, img_h - 70.8, img_w + 55.0, img_h + 20.0], [img_w - 3.0, img_h - 2.5]),
            ([img_w + 5.0, 10.0, img_w + 90.0, 70.0], [img_w - 1.0, 40.0]), ([100.2, 100.7, 101.9, 101.2], [100.5, 100.9]),
            ([300.0, 200.0, 420.0, 260.0], [360.0, 230.0]), ([10.9, 20.9, 410.9, 220.9], [210.9, 120.9])]
    return out


@pytest.mark.parametrize("perturb_scale", [False, True])
def test_plan_and_tables_reproduce_the_pil_path(perturb_scale):
    from odvae_amd.patches import plan_patch, resample_table
    rng = np.random.default_rng(3)
    img_h, img_w, S = 450, 800, 256
    img = rng.integers(0, 256, (img_h, img_w, 3), dtype=np.uint8)
    dropped = kept = 0
    for bbox, center in random_instances(rng, 60, img_w, img_h):
        ref = oracle.generate_patch_pil(img, bbox, center, (S, S), perturb_scale)
        plan = plan_patch(bbox, center, img_w, img_h, perturb_scale)
        if ref[0] is None:
            assert plan is None, (bbox, center)
            dropped += 1
            continue
        assert plan is not None, (bbox, center)
        kept += 1
        assert tuple(ref[1]) == (plan.size, plan.size)
        assert ref[3] == plan.padding_pixels * (S / plan.size)
        got, mask = _emulate(img, plan, resample_table(plan.size, S), S)
        assert np.array_equal(oracle.to_tensor(got), ref[0]), (bbox, center)
        assert np.array_equal(mask.astype(np.float32)[None], ref[4]), (bbox, center)
    assert kept > 20 and dropped > 0


# ---- held to the REFERENCE's own code (tests/golden/make_reference_patch_goldens.py ran NuScenesBase._generate_patch unmodified) ----------
def _ref_gold():
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if here not in sys.path:
        sys.path.insert(0, here)
    import reference_cases as rc
    return np.load(os.path.join(here, "reference_patches.npz")), rc


def _crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@pytest.mark.parametrize("perturb", [False, True])
def test_oracle_patches_match_the_reference_run(perturb):
    """Which instances are dropped, the crop (through the resized pixels), the box mask, the crop size, the resampling factor and the padding
    pixels, for 40 instances incl. every border, the box-outside-image corner case, an empty box and fractional coordinates: BIT-exact
    (CRC-32 + sum of the u8 patch / mask bytes; four instances carry the bytes themselves)."""
    g, rc = _ref_gold()
    img = rc.patch_image()
    assert int(img.astype(np.int64).sum()) == int(g["image_sum"])
    S, p = int(g["S"]), int(perturb)
    kept = []
    for i, (bbox, center) in enumerate(rc.patch_instances()):
        out = oracle.generate_patch_pil(img, bbox, center, (S, S), perturb)
        pre = "p%d.%d" % (p, i)
        if out[0] is None:
            assert pre + ".dropped" in g.files, (i, bbox, center)
            continue
        kept.append(i)
        patch, size_sq, factor, pad, mask = out
        p8, m8 = np.round(patch * 255.0).astype(np.uint8), np.round(mask).astype(np.uint8)
        assert _crc(p8) == int(g[pre + ".patch_crc"]) and int(p8.astype(np.int64).sum()) == int(g[pre + ".patch_sum"]), i
        assert _crc(m8) == int(g[pre + ".mask_crc"]) and int(m8.astype(np.int64).sum()) == int(g[pre + ".mask_sum"]), i
        if pre + ".patch_u8" in g.files:
            assert np.array_equal(p8, g[pre + ".patch_u8"]) and np.array_equal(m8, g[pre + ".mask_u8"])
        assert np.array_equal(np.asarray(size_sq, np.float32), g[pre + ".size_sq"])
        assert tuple(factor) == tuple(g[pre + ".factor"]) and float(pad) == float(g[pre + ".padding_resampled"])
    assert kept == list(g["p%d.kept" % p]) and len(kept) >= 30


@pytest.mark.parametrize("perturb", [False, True])
def test_product_crop_plans_match_the_reference_run(perturb):
    """The product's host logic (patches.plan_patch: the integer arithmetic the GPU kernel is driven by) against the same fixture: dropped set,
    crop size (= the reference's patch.size), padding pixels."""
    from odvae_amd.patches import plan_patch
    g, rc = _ref_gold()
    h, w = rc.PATCH_IMAGE_HW
    S, p = int(g["S"]), int(perturb)
    kept = []
    for i, (bbox, center) in enumerate(rc.patch_instances()):
        plan = plan_patch(bbox, center, w, h, perturb)
        pre = "p%d.%d" % (p, i)
        if plan is None:
            assert pre + ".dropped" in g.files, i
            continue
        kept.append(i)
        assert np.array_equal(np.asarray([plan.size, plan.size], np.float32), g[pre + ".size_sq"]), i
        assert float(plan.padding_pixels) * (S / plan.size) == float(g[pre + ".padding_resampled"]), i
    assert kept == list(g["p%d.kept" % p])

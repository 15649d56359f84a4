"""Patch extraction on the device (csrc/patch_u8.hip through the C ABI) vs the PIL path of the reference's dataset
(oracle/patches.py: Image.crop -> Image.resize(BILINEAR, reducing_gap=1.0) -> ToTensor, and the NEAREST-resized box mask;
src/data/datasets/nuscenes.py:90-194).  Byte/integer work: the bar is bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
pytest.importorskip("PIL.Image")

from test_patches import random_instances  # noqa: E402


def _images(rng, shapes):
    imgs = []
    for h, w in shapes:
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
        noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        imgs.append(np.where(rng.random((h, w, 1)) < 0.5, smooth, noise).astype(np.uint8))
    return imgs


@pytest.mark.parametrize("S,perturb_scale", [(256, False), (256, True), (512, True), (224, False)])
def test_patch_batch_is_bit_identical_to_pil(hip_lib, S, perturb_scale):
    from odvae_amd.patches import GpuPatcher
    from oracle import patches as oracle
    rng = np.random.default_rng(11 + S)
    host_imgs = _images(rng, [(900, 1600), (450, 800), (300, 333)])   # nuScenes cameras are 1600x900
    dev_imgs = [torch.from_numpy(a).to("cuda:0") for a in host_imgs]
    inst = []
    for k, a in enumerate(host_imgs):
        for bbox, center in random_instances(rng, 24, a.shape[1], a.shape[0], max_extent=min(480, 2 * S - 8)):
            inst.append((k, bbox, center))
    patcher = GpuPatcher(patch_height=S, perturb_scale=perturb_scale)
    out = patcher(dev_imgs, inst)
    torch.cuda.synchronize()
    refs = [oracle.generate_patch_pil(host_imgs[k], bbox, center, (S, S), perturb_scale) for k, bbox, center in inst]
    kept = [i for i, r in enumerate(refs) if r[0] is not None]
    assert out.kept == kept and len(kept) > 30 and len(kept) < len(inst)
    assert out.patch.shape == (len(kept), 3, S, S) and out.mask.shape == (len(kept), 1, S, S)
    assert out.patch.is_contiguous(memory_format=torch.channels_last)
    patch, mask = out.patch.cpu().numpy(), out.mask.cpu().numpy()
    for j, i in enumerate(kept):
        ref = refs[i]
        assert np.array_equal(patch[j], ref[0]), ("patch", i, inst[i])
        assert np.array_equal(mask[j], ref[4]), ("mask", i, inst[i])
        assert np.array_equal(out.patch_size[j].numpy(), ref[1])
        assert out.resampling_factor[j] == ref[2]
        assert out.padding_pixels_resampled[j] == ref[3]


def test_every_byte_value_and_identity_size(hip_lib):
    """crop size == S is a plain copy in Pillow; all 256 byte values go through the u8 -> f32 / 255 conversion."""
    from odvae_amd.patches import GpuPatcher
    S = 64
    img = (np.arange(S * S * 3) % 256).astype(np.uint8).reshape(S, S, 3)
    out = GpuPatcher(patch_height=S)([torch.from_numpy(img).to("cuda:0")], [(0, [0.0, 0.0, float(S), float(S)], [S / 2, S / 2])])
    ref = torch.from_numpy(img.copy()).permute(2, 0, 1).to(torch.float32).div(255)
    assert torch.equal(out.patch[0].cpu(), ref)
    assert out.mask.min().item() == 1.0


def test_oversize_crop_and_bad_images_fail_loudly(hip_lib):
    from odvae_amd.patches import GpuPatcher
    img = torch.zeros((900, 1600, 3), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(ValueError):
        GpuPatcher(patch_height=128)([img], [(0, [100.0, 100.0, 500.0, 400.0], [300.0, 250.0])])   # 400 -> 128 needs reduce()
    with pytest.raises(ValueError):
        GpuPatcher(patch_height=128)([img.float()], [(0, [100.0, 100.0, 150.0, 140.0], [125.0, 120.0])])
    with pytest.raises(Exception):
        GpuPatcher(patch_height=128)([img.cpu()], [(0, [100.0, 100.0, 150.0, 140.0], [125.0, 120.0])])


@pytest.mark.parametrize("perturb", [False, True])
def test_device_patches_match_the_reference_run(hip_lib, perturb):
    """csrc/patch_u8.hip through the C ABI against outputs of the REFERENCE's own `NuScenesBase._generate_patch` (tests/golden/
    reference_patches.npz, written by make_reference_patch_goldens.py): dropped set, crop size, padding pixels, and the u8 patch / mask bytes
    by CRC-32 + sum -- bit-exact.  Instances whose crop is >= 2 S go through Pillow's box pre-reduction (`reducing_gap=1.0`), which the device
    path refuses loudly (the 256 / 512 network resolutions never meet it for PATCH_SIZES <= 400); they are checked for exactly that."""
    import zlib
    from odvae_amd.patches import GpuPatcher, plan_patch
    from test_patches import _ref_gold
    g, rc = _ref_gold()
    img = rc.patch_image()
    h, w = rc.PATCH_IMAGE_HW
    S, p = int(g["S"]), int(perturb)
    dev_img = torch.from_numpy(img).to("cuda:0")
    inst, big = [], []
    for i, (bbox, center) in enumerate(rc.patch_instances()):
        plan = plan_patch(bbox, center, w, h, perturb)
        if plan is not None and plan.size >= 2 * S:
            big.append(i)
            with pytest.raises(ValueError):
                GpuPatcher(patch_height=S, perturb_scale=perturb)([dev_img], [(0, bbox, center)])
            continue
        inst.append((i, bbox, center))
    out = GpuPatcher(patch_height=S, perturb_scale=perturb)([dev_img], [(0, b, c) for _, b, c in inst])
    torch.cuda.synchronize()
    kept_ref = [i for i in g["p%d.kept" % p] if i not in big]
    assert [inst[j][0] for j in out.kept] == kept_ref and len(kept_ref) >= 20
    patch = (out.patch.cpu().numpy() * 255.0).round().astype(np.uint8)
    mask = out.mask.cpu().numpy().round().astype(np.uint8)
    for j, i in enumerate(kept_ref):
        pre = "p%d.%d" % (p, i)
        assert np.array_equal(patch[j].astype(np.float32) / np.float32(255), out.patch[j].cpu().numpy())      # the f32 values ARE u8 / 255
        assert zlib.crc32(np.ascontiguousarray(patch[j]).tobytes()) == int(g[pre + ".patch_crc"]), i
        assert zlib.crc32(np.ascontiguousarray(mask[j]).tobytes()) == int(g[pre + ".mask_crc"]), i
        assert np.array_equal(out.patch_size[j].numpy(), g[pre + ".size_sq"])
        assert tuple(out.resampling_factor[j]) == tuple(g[pre + ".factor"])
        assert float(out.padding_pixels_resampled[j]) == float(g[pre + ".padding_resampled"])

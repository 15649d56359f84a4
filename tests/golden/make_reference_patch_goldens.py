"""Generates tests/golden/reference_patches.npz by RUNNING the reference's own `NuScenesBase._generate_patch`
(src/data/datasets/nuscenes.py:90-194, imported unmodified from /root/reference) on a synthetic camera image: the step right in front of the
hot path, whose box arithmetic -- which square to cut, which instances to drop, the padding pixels, the mask slice --
oracle/patches.py had only restated ("PARITY UNPINNED").  The module imports mmdet3d, pytorch3d and torchvision, none of which exist in this
image; they get stand-ins for exactly the names imported: empty base classes / registries for mmdet3d and pytorch3d (nothing of them runs
inside `_generate_patch`), and torchvision's `Compose` / `ToTensor` restated as their documented behaviour (u8 HWC -> f32 CHW / 255, mode "1"
-> 0 / 1).  Pillow -- what does the pixel work -- is the real library.  Build container only; fixtures = inputs' seeds + expected outputs.

    python tests/golden/make_reference_patch_goldens.py
"""
import os
import sys
import tempfile
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
OUT = os.path.join(HERE, "reference_patches.npz")
sys.path[:0] = [HERE, ROOT]
from reference_cases import PATCH_S, patch_image, patch_instances  # noqa: E402


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            _module(parent)
        setattr(sys.modules[parent], child, m)
    return m


def install_standins():
    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    class _Base(object):
        def __init__(self, *a, **k):
            pass

    _module("mmdet3d.registry", DATASETS=_Registry())
    _module("mmdet3d.datasets.nuscenes_dataset", NuScenesDataset=_Base)
    _module("pytorch3d.renderer.cameras", _R=None, _T=None, PerspectiveCameras=_Base, _FocalLengthType=object)
    _module("pytorch3d.common.datatypes", Device=object)
    nothing = lambda *a, **k: None
    _module("pytorch3d.transforms", Transform3d=_Base, euler_angles_to_matrix=nothing, matrix_to_euler_angles=nothing, se3_log_map=nothing,
            se3_exp_map=nothing)

    class ToTensor:      # torchvision.transforms.ToTensor on a PIL image: HWC u8 -> CHW f32 / 255; mode "1" -> 0 / 1
        def __call__(self, pic):
            a = np.asarray(pic)
            if a.dtype == bool:
                a = a.astype(np.uint8) * 255
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    _module("torchvision.transforms", Compose=Compose, ToTensor=ToTensor)
    _module("torchvision.transforms.functional", InterpolationMode=object)
    _module("torchvision.ops")


def main():
    from PIL import Image
    install_standins()
    for name in [n for n in sys.modules if n == "src" or n.startswith("src.")]:
        del sys.modules[name]
    sys.path.insert(0, REFERENCE)
    import src.data.datasets.nuscenes as ref
    assert ref.__file__.startswith(REFERENCE + "/")
    img = patch_image()
    f = tempfile.NamedTemporaryFile(suffix=".png", delete=False)
    f.close()
    Image.fromarray(img).save(f.name)
    arrays = {"image_sum": np.int64(img.astype(np.int64).sum()), "S": np.int64(PATCH_S)}
    inst = patch_instances()
    for perturb in (False, True):
        this = types.SimpleNamespace(perturb_scale=perturb, patch_size=(PATCH_S, PATCH_S))
        kept = []
        for i, (bbox, center) in enumerate(inst):
            cam = types.SimpleNamespace(bbox=list(bbox), center_2d=list(center))
            try:
                patch, size_sq, factor, pad, mask = ref.NuScenesBase._generate_patch(this, f.name, cam)
            except ValueError as e:      # (Image.resize of a crop >= 2 S with reducing_gap would box-reduce first: not in this fixture)
                raise
            pre = "p%d.%d" % (int(perturb), i)
            if patch is None:
                arrays[pre + ".dropped"] = np.int64(1)
                continue
            kept.append(i)
            p8 = np.round(patch.numpy() * 255.0).astype(np.uint8)
            assert np.array_equal(p8.astype(np.float32) / np.float32(255), patch.numpy())
            m8 = np.round(mask.numpy()).astype(np.uint8)
            # bit-exactness is checked through CRC-32 + sum of the bytes; the first four kept instances also carry the bytes themselves
            arrays[pre + ".patch_crc"] = np.int64(zlib.crc32(np.ascontiguousarray(p8).tobytes()))
            arrays[pre + ".patch_sum"] = np.int64(p8.astype(np.int64).sum())
            arrays[pre + ".mask_crc"] = np.int64(zlib.crc32(np.ascontiguousarray(m8).tobytes()))
            arrays[pre + ".mask_sum"] = np.int64(m8.astype(np.int64).sum())
            if len(kept) <= 4:
                arrays[pre + ".patch_u8"] = p8
                arrays[pre + ".mask_u8"] = m8
            arrays[pre + ".size_sq"] = size_sq.numpy()
            arrays[pre + ".factor"] = np.asarray(factor, np.float64)
            arrays[pre + ".padding_resampled"] = np.float64(pad)
        arrays["p%d.kept" % int(perturb)] = np.asarray(kept, np.int64)
    os.unlink(f.name)
    np.savez_compressed(OUT, **arrays)
    print("wrote %s: %d arrays, %d bytes; kept %s / %s of %d" % (OUT, len(arrays), os.path.getsize(OUT), len(arrays["p0.kept"]), len(arrays["p1.kept"]), len(inst)))


if __name__ == "__main__":
    main()

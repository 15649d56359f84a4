"""Generates tests/golden/pose_encoder_ref.npz by importing the REFERENCE's own
src/modules/autoencodermodules/pose_encoder.py (the one hot-path-adjacent reference module that imports without the
un-vendored ldm/taming submodules).  Run in the build container only: /root/reference does not travel.

    python tests/golden/make_pose_encoder_golden.py
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/src/modules/autoencodermodules/pose_encoder.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_encoder_ref.npz")


def main():
    spec = importlib.util.spec_from_file_location("ref_pose_encoder", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(23)
    # small instance (n = m = 4) keeps the fixture at a few hundred KB; same code path as n = m = 16
    net = mod.PoseEncoderSpatialVAE(num_classes=11, num_channels=16, n=4, m=4, activation="swish", hidden_dim=64, num_layers=2)
    z = torch.randn(3, 19, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        y = net(z)
    arrays = {"z": z.numpy(), "y": y.numpy(), "grid": net.x.numpy()}
    for k, v in net.state_dict().items():
        arrays["sd." + k] = v.numpy()
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()

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
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "pose_encoder_ref.npz")
OUT_FULL = os.path.join(HERE, "pose_encoder_ref_full.npz")


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
    full(mod)


def full(mod):
    """The yaml's OWN instance (yaml:46-54: 11 classes, 16 channels, n = m = 16, hidden 500, 2 layers, swish; 3 089 984
    parameters) run by the reference's module, forward and -- through torch autograd on the reference's forward --
    backward.  The weights are NOT stored (12 MB): they follow from the state_dict keys by
    odvae_amd.synthetic.fill_state_procedural, which the tests re-run on their own module (same keys, same shapes).
    Stored: z [4,19], y [4,4096], the upstream gradient gy, dL/dz, and per parameter the gradient's L2 norm plus its
    first 64 entries."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from odvae_amd.synthetic import fill_state_procedural
    net = mod.PoseEncoderSpatialVAE(num_classes=11, num_channels=16, n=16, m=16, activation="swish", hidden_dim=500, num_layers=2)
    fill_state_procedural(net, seed=23)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(4, 19, generator=g).requires_grad_(True)
    gy = torch.randn(4, 16 * 16 * 16, generator=g) / 64.0
    y = net(z)
    y.backward(gy)
    arrays = {"z": z.detach().numpy(), "y": y.detach().numpy(), "gy": gy.numpy(), "dz": z.grad.numpy()}
    for k, p in net.named_parameters():
        arrays["gnorm." + k] = np.float64(p.grad.double().norm().item())
        arrays["ghead." + k] = p.grad.reshape(-1)[:64].numpy().copy()
    np.savez_compressed(OUT_FULL, **arrays)
    print("wrote", OUT_FULL, os.path.getsize(OUT_FULL), "bytes")


if __name__ == "__main__":
    main()

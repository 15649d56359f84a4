"""The ONE fixture set produced by reference code -- tests/golden/pose_encoder_ref.npz and pose_encoder_ref_full.npz, both
written by importing /root/reference/src/modules/autoencodermodules/pose_encoder.py:59-131 in the build container
(tests/golden/make_pose_encoder_golden.py) -- against the path that runs on the training step: PoseEncoderSpatialVAE on
cuda:0, i.e. csrc/linear_f32.hip through the C ABI (odvae_linear_f32), forward and backward.  The full file is the yaml's own
instance (n = m = 16, hidden 500, 3 089 984 parameters), the shapes `_encode_pose` (src/models/autoencoder.py:162-174) runs
at 256 x 256.  Tolerance 1e-5 of max|ref| (fp32; the kernel sums K in a different order than the reference's CPU GEMM)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def close(a, b, what, tol=1e-5):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max()
    assert err <= tol * max(np.abs(b).max(), 1e-30), "%s: max err %.3e vs max|ref| %.3e" % (what, err, np.abs(b).max())


def test_small_instance_on_device_matches_reference_output(hip_lib):
    from odvae_amd.pose_modules import PoseEncoderSpatialVAE
    g = np.load(os.path.join(GOLD, "pose_encoder_ref.npz"))
    net = PoseEncoderSpatialVAE(num_classes=11, num_channels=16, n=4, m=4, activation="swish", hidden_dim=64, num_layers=2)
    res = net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    net = net.to(DEV)
    with torch.no_grad():
        y = net(torch.from_numpy(g["z"]).to(DEV))
    assert y.is_cuda
    close(y.cpu().numpy(), g["y"], "y (n = m = 4)")


def test_yaml_instance_on_device_matches_reference_forward_and_backward(hip_lib):
    from odvae_amd import ops
    from odvae_amd.pose_modules import PoseEncoderSpatialVAE
    from odvae_amd.synthetic import fill_state_procedural
    g = np.load(os.path.join(GOLD, "pose_encoder_ref_full.npz"))
    net = PoseEncoderSpatialVAE(num_classes=11, num_channels=16, n=16, m=16, activation="swish", hidden_dim=500, num_layers=2)
    fill_state_procedural(net, seed=23)
    net = net.to(DEV)
    calls = []
    real = ops.linear_act
    try:
        ops.linear_act = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        import odvae_amd.pose_modules as pm
        assert pm.ops is ops
        z = torch.from_numpy(g["z"]).to(DEV).requires_grad_(True)
        y = net(z)
        y.backward(torch.from_numpy(g["gy"]).to(DEV))
    finally:
        ops.linear_act = real
    assert len(calls) == 4            # coord_linear, latent_linear, layers.1 (+swish), layers.3: all on linear_f32.hip
    close(y.detach().cpu().numpy(), g["y"], "y")
    close(z.grad.cpu().numpy(), g["dz"], "dL/dz")
    for k, p in net.named_parameters():
        want = float(g["gnorm." + k])
        assert abs(p.grad.double().norm().item() - want) <= 1e-5 * want, k
        close(p.grad.reshape(-1)[:64].cpu().numpy(), g["ghead." + k], "grad head " + k, tol=2e-5)

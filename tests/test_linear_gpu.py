"""Pose-head dense layers on linear_f32.hip against torch on the host: single layers at every shape the two
MLPs use (4096 -> 500 -> 500 -> 27 tanh; 512 -> 1024, 19 -> 4 without bias, 1024 -> 500 -> 4096 swish) plus ragged shapes that leave
partial tiles in every dimension, and the two modules end to end against the oracle's PoseEncoderSpatialVAE / PoseDecoderSpatialVAE
(src/modules/autoencodermodules/pose_encoder.py:59-131, pose_decoder.py:60-97) with the same state_dict -- outputs, input gradients
and every parameter gradient.  f32 with a different summation order: 1e-4 of the tensor's scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ACTS = {None: lambda t: t, "tanh": torch.tanh, "swish": torch.nn.functional.silu, "relu": torch.relu}


def close(a, b, what, tol=1e-4):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs().max().item()
    assert err <= tol * max(1e-6, b.abs().max().item()), "%s: %.3e vs ref scale %.3e" % (what, err, b.abs().max().item())


@pytest.mark.parametrize("m,n,k,act,bias", [(32, 500, 4096, "tanh", True), (32, 500, 500, "tanh", True), (32, 27, 500, None, True),
                                            (1, 1024, 512, None, True), (32, 4, 19, None, False), (32, 500, 1024, "swish", True),
                                            (32, 4096, 500, None, True), (5, 33, 70, "relu", True), (67, 31, 9, "swish", False),
                                            (2, 500, 4096, "tanh", True)])
def test_linear_act_matches_torch(hip_lib, m, n, k, act, bias):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(m * 1000 + n + k)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g) * 0.1 if bias else None
    dy = torch.randn(m, n, generator=g)

    ref_in = [t.double().requires_grad_() for t in (x, w)] + ([b.double().requires_grad_()] if bias else [])
    ref = ACTS[act](torch.nn.functional.linear(*ref_in))
    ref.backward(dy.double())

    dev_in = [t.to(DEV).requires_grad_() for t in (x, w)] + ([b.to(DEV).requires_grad_()] if bias else [None])
    y = ops.linear_act(dev_in[0], dev_in[1], dev_in[2], act)
    y.backward(dy.to(DEV))
    close(y, ref, "y")
    for name, d, r in zip(("dx", "dw", "db"), dev_in, ref_in):
        close(d.grad, r.grad, name)
    # deterministic: fixed split-K order, no atomics
    y2 = ops.linear_act(dev_in[0], dev_in[1], dev_in[2], act)
    assert torch.equal(y, y2)


def test_linear_act_rejects_bad_arguments(hip_lib):
    from odvae_amd import ops
    from odvae_amd.lib import HipLibraryError
    with pytest.raises(ValueError):
        ops.linear_act(torch.zeros(2, 8, device=DEV), torch.zeros(4, 9, device=DEV))
    with pytest.raises((HipLibraryError, ValueError, RuntimeError)):
        ops.linear_act(torch.zeros(2, 8), torch.zeros(4, 8))           # host tensors never reach the kernel


@pytest.mark.parametrize("batch", [32, 3])
def test_pose_mlps_match_oracle_modules(hip_lib, batch):
    from odvae_amd import pose_modules
    from oracle import autoencoder as O
    torch.manual_seed(7)
    for ours_cls, ref_cls, d_in, kw in ((pose_modules.PoseDecoderSpatialVAE, O.PoseDecoderSpatialVAE, 4096, dict(activation="tanh", hidden_dim=500, num_layers=2)),
                                        (pose_modules.PoseEncoderSpatialVAE, O.PoseEncoderSpatialVAE, 19, dict(activation="swish", hidden_dim=500, num_layers=2)),
                                        (pose_modules.PoseEncoderSpatialVAE, O.PoseEncoderSpatialVAE, 19, dict(activation="swish", hidden_dim=96, num_layers=3))):
        ref = ref_cls(num_classes=11, num_channels=16, n=16, m=16, **kw).double()
        ours = ours_cls(num_classes=11, num_channels=16, n=16, m=16, **kw)
        ours.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        ours.to(DEV)
        d = d_in if d_in != 19 else 8 + 11
        x = torch.randn(batch, d)
        xr = x.double().requires_grad_()
        xo = x.to(DEV).requires_grad_()
        yr = ref(xr)
        yo = ours(xo)
        close(yo, yr, ours_cls.__name__ + " out")
        dy = torch.randn(yr.shape)
        yr.backward(dy.double())
        yo.backward(dy.to(DEV))
        close(xo.grad, xr.grad, ours_cls.__name__ + " dx")
        pr = dict(ref.named_parameters())
        for name, p in ours.named_parameters():
            close(p.grad, pr[name].grad, ours_cls.__name__ + " " + name)

"""PatchGAN / LPIPS-style kernels and modules vs torch CPU fp32 (oracle/losses.py modules with the same state_dict).
Tolerances: per-op forward 2e-4, backward 5e-4 (relative to max|ref|); whole networks 1e-3 / 5e-3."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, tol, what=""):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs().max().item()
    ref = max(1e-6, b.abs().max().item())
    assert err <= tol * ref, "%s: max err %.3e > %.1e * %.3e" % (what, err, tol, ref)


@pytest.mark.parametrize("n,cin,cout,h,w,stride,bias", [(2, 3, 64, 32, 32, 2, True), (2, 64, 128, 16, 16, 2, False),
                                                        (1, 32, 64, 9, 7, 1, False), (2, 64, 1, 8, 8, 1, True)])
def test_conv4x4(hip_lib, n, cin, cout, h, w, stride, bias):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(cin * cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 4, 4, generator=g) / math.sqrt(16 * cin)
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    y_ref = F.conv2d(xr, wr, br, stride=stride, padding=1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd, wd = x.to(DEV).requires_grad_(True), wt.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if bias else None
    y = ops.conv4x4(xd, wd, bd, stride)
    close(y, y_ref, 2e-4, "conv4x4 fwd")
    y.backward(gy.to(DEV))
    close(xd.grad, xr.grad, 5e-4, "conv4x4 dx")
    close(wd.grad, wr.grad, 5e-4, "conv4x4 dw")
    if bias:
        close(bd.grad, br.grad, 5e-4, "conv4x4 db")


@pytest.mark.parametrize("train", [True, False])
def test_batchnorm_lrelu(hip_lib, train):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 128, 6, 5, generator=g) * 2 + 1
    bn_ref = torch.nn.BatchNorm2d(128)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.randn(128, generator=g)); bn_ref.bias.copy_(torch.randn(128, generator=g))
        bn_ref.running_mean.copy_(torch.randn(128, generator=g) * 0.1); bn_ref.running_var.copy_(torch.rand(128, generator=g) + 0.5)
    import copy
    bn = copy.deepcopy(bn_ref).to(DEV)
    bn_ref.train(train); bn.train(train)
    xr = x.clone().requires_grad_(True)
    y_ref = F.leaky_relu(bn_ref(xr), 0.2)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    y = ops.batchnorm_lrelu(xd, bn, 0.2)
    close(y, y_ref, 2e-4, "bn fwd")
    y.backward(gy.to(DEV))
    close(xd.grad, xr.grad, 5e-4, "bn dx")
    close(bn.weight.grad, bn_ref.weight.grad, 5e-4, "bn dgamma")
    close(bn.bias.grad, bn_ref.bias.grad, 5e-4, "bn dbeta")
    close(bn.running_mean, bn_ref.running_mean, 1e-5, "running_mean")
    close(bn.running_var, bn_ref.running_var, 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked)


def test_small_ops(hip_lib):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 64, 8, 6, generator=g)
    xr = x.clone().requires_grad_(True)
    gy = torch.randn(2, 64, 8, 6, generator=g)
    F.leaky_relu(xr, 0.2).backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    y = ops.leaky_relu(xd, 0.2)
    close(y, F.leaky_relu(x, 0.2), 1e-6, "lrelu")
    y.backward(gy.to(DEV))
    close(xd.grad, xr.grad, 1e-6, "lrelu bwd")
    # max pool with ties (post-ReLU zeros): gradient goes to the first maximum, like torch
    xp = torch.relu(torch.randn(2, 64, 8, 8, generator=g))
    xpr = xp.clone().requires_grad_(True)
    yp_ref = F.max_pool2d(xpr, 2, 2)
    gp = torch.randn(yp_ref.shape, generator=g)
    yp_ref.backward(gp)
    xpd = xp.to(DEV).requires_grad_(True)
    yp = ops.maxpool2x2(xpd)
    close(yp, yp_ref, 1e-6, "maxpool")
    yp.backward(gp.to(DEV))
    close(xpd.grad, xpr.grad, 1e-6, "maxpool bwd")
    # scaling layer
    shift = torch.tensor([-.030, -.088, -.188])[None, :, None, None]; scale = torch.tensor([.458, .448, .450])[None, :, None, None]
    xs = torch.randn(2, 3, 8, 8, generator=g)
    xsr = xs.clone().requires_grad_(True)
    ((xsr - shift) / scale).backward(torch.ones(2, 3, 8, 8))
    xsd = xs.to(DEV).requires_grad_(True)
    ys = ops.scale_shift(xsd, shift.to(DEV), scale.to(DEV))
    close(ys, (xs - shift) / scale, 1e-6, "scaling")
    ys.backward(torch.ones(2, 3, 8, 8, device=DEV))
    close(xsd.grad, xsr.grad, 1e-6, "scaling bwd")
    # conv3x3 + fused ReLU
    wt = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g)
    xc = torch.randn(2, 64, 8, 8, generator=g)
    xcr = xc.clone().requires_grad_(True)
    yc_ref = F.relu(F.conv2d(xcr, wt, b, padding=1))
    gc = torch.randn(yc_ref.shape, generator=g)
    yc_ref.backward(gc)
    xcd = xc.to(DEV).requires_grad_(True)
    yc = ops.conv3x3(xcd, wt.to(DEV), b.to(DEV), None, 0, relu=True)
    close(yc, yc_ref, 2e-4, "conv+relu")
    yc.backward(gc.to(DEV))
    close(xcd.grad, xcr.grad, 5e-4, "conv+relu dx")


@pytest.mark.parametrize("c", [64, 128, 512])
def test_lpips_layer_distance(hip_lib, c):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(c)
    f0 = torch.relu(torch.randn(2, c, 6, 5, generator=g))
    f1 = torch.relu(torch.randn(2, c, 6, 5, generator=g))
    w = torch.rand(1, c, 1, 1, generator=g) / c
    f1r = f1.clone().requires_grad_(True)
    n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True)) + 1e-10)
    n1 = f1r / (torch.sqrt(torch.sum(f1r ** 2, dim=1, keepdim=True)) + 1e-10)
    ref = F.conv2d((n0 - n1) ** 2, w).mean([1, 2, 3])
    gw = torch.randn(2, generator=g)
    (ref * gw).sum().backward()
    f1d = f1.to(DEV).requires_grad_(True)
    out = ops.lpips_layer_distance(f0.to(DEV), f1d, w.to(DEV))
    close(out, ref, 2e-5, "lpips dist")
    (out * gw.to(DEV)).sum().backward()
    close(f1d.grad, f1r.grad, 2e-4, "lpips dist bwd")


def test_discriminator_matches_oracle(hip_lib):
    from odvae_amd.gan import NLayerDiscriminator, weights_init
    from oracle.losses import NLayerDiscriminator as RefD
    torch.manual_seed(5)
    ref = RefD().apply(weights_init)
    net = NLayerDiscriminator()
    res = net.load_state_dict(ref.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    net = net.to(DEV)
    ref.train(); net.train()
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    xr = x.clone().requires_grad_(True)
    y_ref = ref(xr)
    assert tuple(y_ref.shape) == (2, 1, 6, 6)
    gy = torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(7))
    y_ref.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd)
    close(y, y_ref, 1e-3, "D fwd")
    y.backward(gy.to(DEV))
    close(xd.grad, xr.grad, 5e-3, "D dx")
    refp = dict(ref.named_parameters())
    for name, p in net.named_parameters():
        close(p.grad, refp[name].grad, 5e-3, "D grad " + name)
    refb = dict(ref.named_buffers())
    for name, b in net.named_buffers():
        close(b.float(), refb[name].float(), 1e-4, "D buffer " + name)


def test_lpips_style_matches_oracle(hip_lib):
    from odvae_amd.gan import LPIPSStyle
    from oracle.losses import LPIPSStyle as RefL
    net = LPIPSStyle()
    ref = RefL()
    res = ref.load_state_dict(net.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    net = net.to(DEV).eval(); ref.eval()
    g = torch.Generator().manual_seed(8)
    x0 = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    x1 = (x0 + 0.3 * torch.randn(2, 3, 32, 32, generator=g)).clamp(-1, 1)
    x1r = x1.clone().requires_grad_(True)
    d_ref = ref(x0, x1r)
    assert tuple(d_ref.shape) == (2, 1, 1, 1)
    d_ref.sum().backward()
    x1d = x1.to(DEV).requires_grad_(True)
    d = net(x0.to(DEV), x1d)
    close(d, d_ref, 1e-3, "lpips fwd")
    d.sum().backward()
    close(x1d.grad, x1r.grad, 5e-3, "lpips dx")


def test_lpips_from_upstream_keyed_checkpoint(hip_lib, tmp_path):
    """A checkpoint carrying `loss.perceptual_loss.*` in the UPSTREAM key scheme goes through `ckpt_path` into the HIP model and
    straight into the oracle; both then give the same perceptual distance and input gradient (reference:
    src/modules/losses/contperceptual.py:3,5; src/models/autoencoder.py:97-98)."""
    import os
    import warnings
    from test_checkpoint import YAML, _oracle, synthetic_upstream_lpips_state
    from odvae_amd import synthetic
    from odvae_amd.config import instantiate_from_config
    sd = synthetic_upstream_lpips_state(seed=21)
    torch.manual_seed(4)
    ref = _oracle()
    full = ref.state_dict()
    for k, v in sd.items():
        full["loss.perceptual_loss." + k] = v
    path = os.path.join(tmp_path, "last.ckpt")
    torch.save({"state_dict": full}, path)
    ref.load_state_dict(torch.load(path)["state_dict"], strict=True)
    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32, perceptual_weight=1.0)
    mcfg.params["ckpt_path"] = path
    model = instantiate_from_config(mcfg).to(DEV)
    lp, lp_ref = model.loss.perceptual_loss.eval(), ref.loss.perceptual_loss.eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error")            # real (loaded) weights: the synthetic-weights warning must stay quiet
        model.loss._check_perceptual_weights()
    g = torch.Generator().manual_seed(9)
    x0 = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    x1 = (x0 + 0.25 * torch.randn(2, 3, 64, 64, generator=g)).clamp(-1, 1)
    x1r = x1.clone().requires_grad_(True)
    d_ref = lp_ref(x0, x1r)
    d_ref.sum().backward()
    x1d = x1.to(DEV).requires_grad_(True)
    d = lp(x0.to(DEV), x1d)
    close(d, d_ref, 1e-3, "lpips(ckpt) fwd")
    d.sum().backward()
    close(x1d.grad, x1r.grad, 5e-3, "lpips(ckpt) dx")
    # and a fresh loss with perceptual_weight > 0 on the stand-in weights does warn
    mcfg2, _ = synthetic.model_config(YAML, latent_hw=4, ch=32, perceptual_weight=1.0)
    model2 = instantiate_from_config(mcfg2).to(DEV)
    with pytest.warns(RuntimeWarning, match="SYNTHETIC"):
        model2.loss._check_perceptual_weights()

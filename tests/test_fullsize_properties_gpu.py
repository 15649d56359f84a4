"""BASELINE.json full sizes (B=32, 256x256, 128 channels; T=4096 tokens), where the CPU oracle cannot run in test time:
parity through size-independent properties and through exact f64 spot checks of sampled outputs.

  conv3x3 forward / data gradient / weight gradient: sampled output elements recomputed in f64 from the gathered operands
      (each is one dot product of the reference formula, F.conv2d's definition), plus linearity in the input;
  GroupNorm: per-(sample, group) mean 0 / variance 1 of the normalised tensor, and invariance to a per-group affine
      change of the input;
  attention: softmax rows sum to 1, so a value tensor that is constant over tokens must come back unchanged, and the
      gradient w.r.t. q, k of that output is 0;
  whole step at full width (ch=128, 256x256): bit-identical repeat, and the directional derivative of the loss along a
      random direction matches <grad, d> (central differences).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
B, C, H = 32, 128, 256


def _sample(rng, *dims, k=24):
    return [tuple(int(rng.integers(0, d)) for d in dims) for _ in range(k)]


def test_conv3x3_full_size_spot_checks_and_linearity(hip_lib):
    from odvae_amd import ops
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, H, H, C, device=DEV, generator=g).permute(0, 3, 1, 2)          # channels_last memory
    w = (torch.randn(C, C, 3, 3, device=DEV, generator=g) / math.sqrt(9 * C)).requires_grad_(True)
    bias = torch.randn(C, device=DEV, generator=g).requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    y = ops.conv3x3(xg, w, bias)
    dy = torch.randn(B, H, H, C, device=DEV, generator=g).permute(0, 3, 1, 2)
    y.backward(dy)
    xc, wc, bc = x.permute(0, 2, 3, 1).cpu().double().numpy(), w.detach().cpu().double().numpy(), bias.detach().cpu().double().numpy()
    dyc = dy.permute(0, 2, 3, 1).cpu().double().numpy()
    yc, dxc, dwc = y.detach().permute(0, 2, 3, 1).cpu().numpy(), xg.grad.permute(0, 2, 3, 1).cpu().numpy(), w.grad.cpu().numpy()
    xp = np.pad(xc, ((0, 0), (1, 1), (1, 1), (0, 0)))
    dyp = np.pad(dyc, ((0, 0), (1, 1), (1, 1), (0, 0)))
    rng = np.random.default_rng(0)
    corners = [(0, 0, 0, 0), (B - 1, H - 1, H - 1, C - 1), (3, 0, H - 1, 5), (7, H - 1, 0, 64)]
    for n, oy, ox, co in _sample(rng, B, H, H, C) + corners:
        want = bc[co] + np.einsum("hwc,chw->", xp[n, oy:oy + 3, ox:ox + 3, :], wc[co])
        assert abs(yc[n, oy, ox, co] - want) <= 2e-5 * max(1.0, abs(want)), ("fwd", n, oy, ox, co)
    for n, iy, ix, ci in _sample(rng, B, H, H, C) + corners:
        # dx[n,iy,ix,ci] = sum_{kh,kw,co} w[co,ci,kh,kw] * dy[n, iy-kh+1, ix-kw+1, co]
        patch = dyp[n, iy:iy + 3, ix:ix + 3, :][::-1, ::-1, :]
        want = np.einsum("hwc,chw->", patch, wc[:, ci])
        assert abs(dxc[n, iy, ix, ci] - want) <= 2e-5 * max(1.0, abs(want)), ("dgrad", n, iy, ix, ci)
    for co, ci, kh, kw in _sample(rng, C, C, 3, 3, k=6):
        want = np.einsum("nhw,nhw->", xp[:, kh:kh + H, kw:kw + H, ci], dyc[..., co])    # 2M-term dot product
        assert abs(dwc[co, ci, kh, kw] - want) <= 2e-4 * math.sqrt(B * H * H), ("wgrad", co, ci, kh, kw)
    assert np.allclose(bias.grad.cpu().numpy(), dyc.sum((0, 1, 2)), rtol=0, atol=2e-4 * math.sqrt(B * H * H))
    # linearity in the input (bias enters once)
    x2 = torch.randn(B, H, H, C, device=DEV, generator=g).permute(0, 3, 1, 2)
    with torch.no_grad():
        lhs = ops.conv3x3(x + x2, w, bias)
        rhs = y.detach() + ops.conv3x3(x2, w, None)
    assert (lhs - rhs).abs().max().item() <= 2e-5 * rhs.abs().max().item()


def test_groupnorm_full_size_statistics_and_invariance(hip_lib):
    from odvae_amd import ops
    g = torch.Generator(device=DEV).manual_seed(2)
    x = (torch.randn(B, H, H, C, device=DEV, generator=g) * 3.0 + 1.5).permute(0, 3, 1, 2)
    ones, zeros = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    y = ops.group_norm(x, ones, zeros, 32, 1e-6, False)
    yg = y.permute(0, 2, 3, 1).reshape(B, H * H, 32, C // 32).double()
    mean, var = yg.mean((1, 3)), yg.var((1, 3), unbiased=False)
    assert mean.abs().max().item() < 1e-5 and (var - 1).abs().max().item() < 1e-4
    # GroupNorm output does not change when a group is scaled and shifted as a whole
    scale = torch.rand(32, device=DEV, generator=g) + 0.5
    shift = torch.randn(32, device=DEV, generator=g)
    x2 = (x.permute(0, 2, 3, 1).reshape(B, H, H, 32, C // 32) * scale[:, None] + shift[:, None]).reshape(B, H, H, C).permute(0, 3, 1, 2)
    y2 = ops.group_norm(x2, ones, zeros, 32, 1e-6, False)
    assert (y2 - y).abs().max().item() < 2e-4


def test_attention_full_token_count_row_sum_property(hip_lib):
    from odvae_amd import ops
    t, c, nb = 4096, 256, 2
    g = torch.Generator(device=DEV).manual_seed(3)
    qk = torch.randn(nb, t, 2 * c, device=DEV, generator=g)
    vconst = torch.randn(nb, 1, c, device=DEV, generator=g).expand(nb, t, c)
    qkv = torch.cat([qk, vconst], dim=2).reshape(nb, 64, 64, 3 * c).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    qkv.requires_grad_(True)
    out = ops.attention_qkv(qkv)                                  # [nb, c, 64, 64]
    got = out.permute(0, 2, 3, 1).reshape(nb, t, c)
    assert (got - vconst).abs().max().item() < 1e-5 * vconst.abs().max().item() + 1e-5
    out.backward(torch.randn_like(out))
    gq = qkv.grad.permute(0, 2, 3, 1).reshape(nb, t, 3 * c)
    # with v constant over tokens the output does not depend on the scores: d/dq = d/dk = 0
    assert gq[..., :2 * c].abs().max().item() < 1e-5 * gq[..., 2 * c:].abs().max().item()


def test_full_width_step_is_repeatable_and_gradient_matches_directional_derivative(hip_lib):
    import os
    from odvae_amd import synthetic
    yaml = os.path.join(os.path.dirname(__file__), "golden", "autoencoder_kl_16x16x16.yaml")
    torch.manual_seed(23)
    model = synthetic.build_model(yaml, batch_size_for_lr=4).to(DEV).train()       # ch = 128, the yaml's width
    model._global_step = 1   # > encoder_pretrain_steps: the reconstruction and KL terms are part of the total
    batch = synthetic.make_batch(4, 256, seed=31)
    noise = synthetic.make_noise(4, 16, dropout_p=0.7, seed=32)

    def loss_of():
        model.injected_noise = noise
        return model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, 0)

    model.zero_grad(set_to_none=True)
    l0 = loss_of(); l0.backward()
    g0 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    l1 = loss_of(); l1.backward()
    assert torch.equal(l0.detach(), l1.detach())
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, g0[n]), n
    # directional derivative along the (normalised) gradient direction restricted to the decoder's conv weights
    names = [n for n in g0 if n.startswith("decoder.") and n.endswith("conv1.weight")]
    params = dict(model.named_parameters())
    gnorm = math.sqrt(sum(g0[n].double().pow(2).sum().item() for n in names))
    wnorm = math.sqrt(sum(params[n].detach().double().pow(2).sum().item() for n in names))
    eps = 5e-4 * abs(l0.item()) / gnorm ** 2          # predicted loss change 1e-3 * |L|: far above fp32 resolution
    assert eps * gnorm < 0.05 * wnorm, "step too large for a derivative check"
    with torch.no_grad():
        for n in names:
            params[n].add_(g0[n], alpha=eps)
        lp = loss_of().item()
        for n in names:
            params[n].add_(g0[n], alpha=-2 * eps)
        lm = loss_of().item()
        for n in names:
            params[n].add_(g0[n], alpha=eps)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - gnorm ** 2) <= 0.05 * gnorm ** 2, (fd, gnorm ** 2, eps)

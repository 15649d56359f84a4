"""bf16 kernels at BASELINE.json's full sizes, where the CPU oracle cannot run in seconds: checks that do not need it.

* conv_bf16 / conv_wgrad_bf16 at B=32, 128 channels, 256x256 (the largest layers of configs[1]) -- f64 recomputation, from the same
  bf16 inputs, of sampled output pixels, data-gradient pixels and weight-gradient entries.
* fused attention at T = 16 384 tokens, C = 256 (configs[4]'s attention level) -- forward and backward against the materialised-
  scores formula evaluated with torch f32 ops ON THE DEVICE (a checker: 1 GiB of scores for one image), plus bit-reproducibility.
Tolerances as in test_bf16_gpu.py: bf16 outputs 1e-2 of max|ref|, f32 outputs 1e-3; attention gradients 2e-2."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def test_conv_bf16_fullsize_f64_spot_checks(hip_lib):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(123)
    n, c, h = 32, 128, 256
    x = torch.randn(n, h, h, c, generator=g).to(BF).to(DEV).permute(0, 3, 1, 2).requires_grad_(True)
    w = (torch.randn(c, c, 3, 3, generator=g) / math.sqrt(9 * c)).to(BF).float().to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(c, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(n, h, h, c, generator=g).to(BF).to(DEV).permute(0, 3, 1, 2)
    y = ops.conv3x3(x, w, b, None, 0)
    y.backward(dy)
    xd, wd, dyd = x.detach().double(), w.detach().double(), dy.double()
    idx = torch.Generator().manual_seed(5)
    ymax, dxmax = y.detach().abs().max().item(), x.grad.abs().max().item()
    for _ in range(24):
        i = int(torch.randint(0, n, (1,), generator=idx)); oy = int(torch.randint(0, h, (1,), generator=idx))
        ox = int(torch.randint(0, h, (1,), generator=idx))
        if _ < 4:   # corners and borders
            oy, ox = [(0, 0), (h - 1, h - 1), (0, h - 1), (h - 1, 0)][_]
        y0, y1, x0, x1 = max(oy - 1, 0), min(oy + 2, h), max(ox - 1, 0), min(ox + 2, h)
        patch = torch.zeros(c, 3, 3, dtype=torch.float64, device=DEV)
        patch[:, y0 - oy + 1:y1 - oy + 1, x0 - ox + 1:x1 - ox + 1] = xd[i, :, y0:y1, x0:x1]
        want = (wd * patch[None]).sum((1, 2, 3)) + b.detach().double()
        assert (y[i, :, oy, ox].double() - want).abs().max().item() <= 1e-2 * ymax
        # data gradient: dx[ci] = sum_{co, kh, kw} dy[co, oy - kh + 1, ox - kw + 1] w[co, ci, kh, kw]
        gpatch = torch.zeros(c, 3, 3, dtype=torch.float64, device=DEV)
        for kh in range(3):
            for kw in range(3):
                yy, xx = oy - kh + 1, ox - kw + 1
                if 0 <= yy < h and 0 <= xx < h:
                    gpatch[:, kh, kw] = dyd[i, :, yy, xx]
        want_dx = (wd * gpatch[:, None]).sum((0, 2, 3))
        assert (x.grad[i, :, oy, ox].double() - want_dx).abs().max().item() <= 1e-2 * dxmax
    # weight gradient entries: dw[co, ci, kh, kw] = sum_{n, y, x} dy[n, co, y, x] x[n, ci, y + kh - 1, x + kw - 1]
    xp = F.pad(xd, (1, 1, 1, 1))
    wmax = w.grad.abs().max().item()
    for _ in range(6):
        co = int(torch.randint(0, c, (1,), generator=idx)); ci = int(torch.randint(0, c, (1,), generator=idx))
        kh, kw = int(torch.randint(0, 3, (1,), generator=idx)), int(torch.randint(0, 3, (1,), generator=idx))
        want = (dyd[:, co] * xp[:, ci, kh:kh + h, kw:kw + h]).sum().item()
        assert abs(w.grad[co, ci, kh, kw].item() - want) <= 1e-3 * wmax + 1e-3 * abs(want)
    want_db = dyd.sum((0, 2, 3))
    assert (b.grad.double() - want_db).abs().max().item() <= 1e-3 * want_db.abs().max().item()


def test_flash_attention_16384_tokens(hip_lib):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(77)
    n, c, h = 1, 256, 128
    t = h * h
    qkv = torch.randn(n, h, h, 3 * c, generator=g).to(BF).to(DEV).permute(0, 3, 1, 2).requires_grad_(True)
    do = torch.randn(n, h, h, c, generator=g).to(BF).to(DEV).permute(0, 3, 1, 2)
    o = ops.attention_qkv(qkv)
    o.backward(do)
    got_dqkv = qkv.grad.detach().clone()
    qkv.grad = None
    o2 = ops.attention_qkv(qkv)
    o2.backward(do)
    assert torch.equal(o2, o) and torch.equal(qkv.grad, got_dqkv), "not bit-reproducible at T = 16384"
    # checker: materialised scores in f32 on the device (1 GiB), autograd for the gradients
    ref_in = qkv.detach().float().reshape(n, 3, c, t).requires_grad_(True)
    q, k, v = ref_in.unbind(1)
    s = torch.bmm(q.transpose(1, 2), k) * (float(c) ** -0.5)
    p = torch.softmax(s, dim=2)
    o_ref = torch.bmm(v, p.transpose(1, 2)).reshape(n, c, h, h)
    o_ref.backward(do.float())
    err = (o.float() - o_ref).abs().max().item()
    assert err <= 1e-2 * o_ref.abs().max().item(), err
    ref_g = ref_in.grad.reshape(n, 3 * c, h, h)
    for name, sl in (("dq", slice(0, c)), ("dk", slice(c, 2 * c)), ("dv", slice(2 * c, 3 * c))):
        e = (got_dqkv[:, sl].float() - ref_g[:, sl]).abs().max().item()
        assert e <= 2e-2 * ref_g[:, sl].abs().max().item(), (name, e)

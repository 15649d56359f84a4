"""bf16 mixed-precision kernels (BASELINE.json configs[4]) vs torch CPU.

Per kernel the reference is torch CPU **f32 arithmetic on the same bf16-rounded inputs**: products of bf16 numbers are exact in
f32 and both sides accumulate in f32, so what remains is summation order plus ONE rounding of a bf16 output.  Tolerances
(relative to max|ref|): f32 outputs (weight / bias gradients, statistics) 1e-3; bf16 outputs 1e-2 (bf16 has 8 significant bits:
half an ulp is 2^-9 = 2e-3 of the value itself, plus the rounding of bf16 intermediates inside multi-kernel ops).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
F32_TOL, BF_TOL = 1e-3, 1e-2


def close(a, b, tol, what=""):
    a = a.detach().float().cpu().double()
    b = b.detach().float().cpu().double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    assert torch.isfinite(a).all(), what + ": non-finite values"
    err = (a - b).abs().max().item()
    ref = max(1e-6, b.abs().max().item())
    assert err <= tol * ref, "%s: max err %.3e > %.1e * %.3e" % (what, err, tol, ref)


def rb(t):
    """round to bf16 and back: the values both sides see"""
    return t.to(BF).float()


def cl_bf16(t):
    """device bf16 tensor in NHWC memory"""
    return t.to(DEV).to(BF).contiguous(memory_format=torch.channels_last)


def ref_conv(mode, x, w, b):
    if mode == 0:
        return F.conv2d(x, w, b, padding=1)
    if mode == 1:
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    if mode == 2:
        return F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    return F.conv2d(x, w, b)


@pytest.mark.parametrize("mode,n,cin,cout,h,w", [
    (0, 2, 64, 128, 16, 16), (0, 1, 128, 256, 12, 20), (0, 2, 32, 64, 8, 8), (0, 1, 256, 32, 8, 16), (0, 2, 96, 128, 8, 16),
    (1, 2, 64, 64, 16, 16), (1, 1, 128, 128, 24, 8), (2, 2, 64, 64, 8, 8), (2, 1, 128, 128, 6, 10), (4, 2, 64, 192, 8, 8),
    (4, 1, 256, 256, 16, 16), (4, 3, 32, 64, 4, 4),
    # larger maps: whole tiles, ragged tiles in both directions, one / two / four 32-channel chunks, two output-channel blocks
    (0, 2, 128, 128, 32, 32), (0, 1, 256, 256, 20, 36), (0, 1, 128, 256, 17, 16), (0, 3, 32, 128, 16, 48), (0, 1, 96, 192, 33, 18)])
def test_conv_bf16_fwd_bwd(hip_lib, mode, n, cin, cout, h, w):
    from odvae_amd import ops
    _conv_bf16_fwd_bwd(ops, mode, n, cin, cout, h, w)


def _conv_bf16_fwd_bwd(ops, mode, n, cin, cout, h, w):
    g = torch.Generator().manual_seed(mode * 1000 + cin + cout + h)
    k = 1 if mode == 4 else 3
    x = rb(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = rb(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    b = (0.1 * torch.randn(cout, generator=g)).requires_grad_(True)
    y_ref = ref_conv(mode, x, wt, b)
    res = rb(torch.randn(y_ref.shape, generator=g)).requires_grad_(True)
    y_ref = y_ref + res
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)

    xd = cl_bf16(x.detach()).requires_grad_(True)
    wd, bd = wt.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    rd = cl_bf16(res.detach()).requires_grad_(True)
    y = ops.conv1x1(xd, wd, bd, rd) if mode == 4 else ops.conv3x3(xd, wd, bd, rd, mode)
    assert y.dtype == BF
    close(y, y_ref, BF_TOL, "y")
    y.backward(cl_bf16(dy))
    close(xd.grad, x.grad, BF_TOL, "dx")
    close(rd.grad, res.grad, 1e-6, "dresidual")
    close(wd.grad, wt.grad, F32_TOL, "dw")
    close(bd.grad, b.grad, F32_TOL, "db")


def test_conv1x1_bf16_weight_gradient_over_image_groups(hip_lib):
    """Past 2 GiB of activations the 1x1 weight gradient is reduced group of images by group of images (as forward and data
    gradient are); forced here at a small size: the grouped result equals the one-launch result up to f32 summation order."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(77)
    x = cl_bf16(torch.randn(5, 64, 8, 8, generator=g))
    wt = torch.randn(128, 64, 1, 1, generator=g).div(8).to(DEV)
    b = (0.1 * torch.randn(128, generator=g)).to(DEV)
    dy = cl_bf16(torch.randn(5, 128, 8, 8, generator=g))
    grads = {}
    for grp in (0, 2):
        ops.WGRAD_1X1_GROUP = grp
        try:
            xd, wd, bd = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
            ops.conv1x1(xd, wd, bd, None).backward(dy)
            grads[grp] = (wd.grad.clone(), bd.grad.clone())
        finally:
            ops.WGRAD_1X1_GROUP = 0
    close(grads[2][0], grads[0][0], 1e-5, "dw grouped vs one launch")
    close(grads[2][1], grads[0][1], 1e-5, "db grouped vs one launch")
    wref = (dy.float().permute(1, 0, 2, 3).reshape(128, -1) @ x.float().permute(1, 0, 2, 3).reshape(64, -1).t()).reshape(128, 64, 1, 1)
    close(grads[2][0], wref, F32_TOL, "dw grouped vs f32 product")


def test_conv_bf16_f32_ends(hip_lib):
    """The f32 ends of the network: 3-channel image padded to 8 zero channels into conv_in; conv_out writing / differentiating a
    3-channel f32 reconstruction; encoder.conv_out writing f32 moments."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(77)
    # conv_in: f32 image -> bf16 features
    img = torch.randn(2, 3, 16, 16, generator=g)
    w_in = rb(torch.randn(64, 3, 3, 3, generator=g) / 5).requires_grad_(True)
    b_in = (0.1 * torch.randn(64, generator=g)).requires_grad_(True)
    y_ref = F.conv2d(rb(img), w_in, b_in, padding=1)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    wd, bd = w_in.detach().to(DEV).requires_grad_(True), b_in.detach().to(DEV).requires_grad_(True)
    xb = ops.to_bf16(img.to(DEV).contiguous(memory_format=torch.channels_last), pad_channels_to=8)
    assert tuple(xb.shape) == (2, 8, 16, 16) and xb.dtype == BF
    y = ops.conv3x3(xb, wd, bd, None, 0)
    close(y, y_ref, BF_TOL, "conv_in y")
    y.backward(cl_bf16(dy))
    close(wd.grad, w_in.grad, F32_TOL, "conv_in dw")
    close(bd.grad, b_in.grad, F32_TOL, "conv_in db")
    # conv_out: bf16 features -> f32 3-channel image, f32 gradient coming back
    x = rb(torch.randn(2, 64, 16, 16, generator=g)).requires_grad_(True)
    w_out = rb(torch.randn(3, 64, 3, 3, generator=g) / 24).requires_grad_(True)
    b_out = (0.1 * torch.randn(3, generator=g)).requires_grad_(True)
    r_ref = F.conv2d(x, w_out, b_out, padding=1)
    dr = torch.randn(r_ref.shape, generator=g)
    r_ref.backward(rb(dr))           # the kernel casts the f32 gradient to bf16 before the products
    xd = cl_bf16(x.detach()).requires_grad_(True)
    wo, bo = w_out.detach().to(DEV).requires_grad_(True), b_out.detach().to(DEV).requires_grad_(True)
    r = ops.conv3x3(xd, wo, bo, None, 0, out_f32=True)
    assert r.dtype == torch.float32 and tuple(r.shape) == (2, 3, 16, 16)
    close(r, r_ref, F32_TOL, "conv_out y")
    r.backward(dr.to(DEV).contiguous(memory_format=torch.channels_last))
    close(xd.grad, x.grad, BF_TOL, "conv_out dx")
    close(wo.grad, w_out.grad, F32_TOL, "conv_out dw")
    close(bo.grad, b_out.grad * 0 + dr.sum((0, 2, 3)), F32_TOL, "conv_out db")   # bias gradient sums the f32 gradient itself
    # latent hand-off: f32 z -> bf16, gradient back in f32
    z = torch.randn(2, 16, 4, 4, generator=g)
    zd = z.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    zb = ops.to_bf16(zd)
    close(zb, rb(z), 1e-6, "to_bf16")
    zb.backward(cl_bf16(torch.ones(2, 16, 4, 4)))
    assert zd.grad.dtype == torch.float32 and torch.equal(zd.grad.cpu(), torch.ones(2, 16, 4, 4))


@pytest.mark.parametrize("n,c,h,w,swish", [(2, 64, 8, 8, True), (1, 128, 12, 20, False), (2, 32, 16, 16, True), (1, 512, 4, 4, True), (2, 256, 16, 8, True)])
def test_groupnorm_bf16(hip_lib, n, c, h, w, swish):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(c + h)
    x = rb(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(c, generator=g)).requires_grad_(True)
    u = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    y_ref = u * torch.sigmoid(u) if swish else u
    dy = rb(torch.randn(y_ref.shape, generator=g))
    dskip = rb(torch.randn(y_ref.shape, generator=g))
    (y_ref * dy).sum().backward(retain_graph=True)
    dx_ref = x.grad.clone() + dskip
    xd = cl_bf16(x.detach()).requires_grad_(True)
    gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
    y, skip = ops.group_norm_skip(xd, gd, bd, 32, 1e-6, swish)
    assert y.dtype == BF
    close(y, y_ref, BF_TOL, "gn y")
    torch.autograd.backward([y, skip], [cl_bf16(dy), cl_bf16(dskip)])
    close(xd.grad, dx_ref, BF_TOL, "gn dx (+ skip)")
    close(gd.grad, gamma.grad, 2e-3, "gn dgamma")
    close(bd.grad, beta.grad, 2e-3, "gn dbeta")


def ref_attention(qkv):
    n, c3, h, w = qkv.shape
    c = c3 // 3
    q, k, v = qkv.reshape(n, 3, c, h * w).unbind(1)            # [n, c, t]
    s = torch.bmm(q.transpose(1, 2), k) * (float(c) ** -0.5)     # [n, tq, tk]
    p = torch.softmax(s, dim=2)
    return torch.bmm(v, p.transpose(1, 2)).reshape(n, c, h, w)  # o[c, tq] = sum_k v[c, k] p[tq, k]


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 16, 16), (2, 128, 4, 4), (1, 256, 32, 32), (2, 512, 8, 8), (1, 64, 6, 6), (3, 128, 5, 9),
                                     (1, 256, 64, 64),
                                     # wave-pair kernels (C = 128 / 256) on token counts that are multiples of neither the 32-row tile
                                     # nor the 128-row block, fewer tiles than ring stages, and a batch that is a multiple of 8 (the
                                     # XCD-grouped block order)
                                     (1, 256, 5, 9), (2, 256, 12, 11), (1, 256, 50, 82), (8, 128, 6, 7), (16, 256, 3, 3)])
def test_flash_attention_bf16(hip_lib, n, c, h, w):
    """Fused attention forward + backward vs the materialised-scores formula in f32 (T = 16 ... 4096, head dim 64 ... 512, token
    counts that are not multiples of the 32-key / 128-query tiles)."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(c + h * w)
    qkv = rb(torch.randn(n, 3 * c, h, w, generator=g)).requires_grad_(True)
    o_ref = ref_attention(qkv)
    do = rb(torch.randn(o_ref.shape, generator=g))
    o_ref.backward(do)
    qd = cl_bf16(qkv.detach()).requires_grad_(True)
    o = ops.attention_qkv(qd)
    assert o.dtype == BF
    close(o, o_ref, BF_TOL, "attention o")
    o.backward(cl_bf16(do))
    # the kernels are deterministic (no atomics; dQ has its own kernel): a second run is bit-identical -- also a race screen for
    # the double-buffered LDS pipeline
    q2 = cl_bf16(qkv.detach()).requires_grad_(True)
    o2 = ops.attention_qkv(q2)
    o2.backward(cl_bf16(do))
    assert torch.equal(o2, o) and torch.equal(q2.grad, qd.grad), "attention is not bit-reproducible"
    dref = qkv.grad
    # dq / dk / dv have very different magnitudes: compare each against its own scale
    for name, sl in (("dq", slice(0, c)), ("dk", slice(c, 2 * c)), ("dv", slice(2 * c, 3 * c))):
        close(qd.grad[:, sl], dref[:, sl], 2e-2, "attention " + name)


def test_flash_attention_peaked_rows(hip_lib):
    """Large score ranges (online-softmax rescaling, exp2 of very negative numbers) and rows dominated by one key."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(5)
    n, c, h, w = 1, 64, 16, 16
    qkv = torch.randn(n, 3 * c, h, w, generator=g)
    qkv[:, :c] *= 6.0
    qkv[:, c:2 * c] *= 6.0
    qkv = rb(qkv)
    o_ref = ref_attention(qkv)
    o = ops.attention_qkv(cl_bf16(qkv))
    close(o, o_ref, 2e-2, "peaked attention")


def test_attn_and_resnet_blocks_bf16(hip_lib):
    """AttnBlock / ResnetBlock modules end to end in bf16 against the oracle modules in f32 on the same bf16-rounded input."""
    from odvae_amd import synthetic
    from odvae_amd.modules import AttnBlock, ResnetBlock
    from oracle.ldm_model import AttnBlock as RefAttn, ResnetBlock as RefRes
    g = torch.Generator().manual_seed(3)
    for name, mod, ref in (("attn", AttnBlock(64), RefAttn(64)),
                           ("res", ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, temb_channels=0),
                            RefRes(in_channels=64, out_channels=128, dropout=0.0, temb_channels=0))):
        synthetic.fill_state_procedural(mod, seed=9)
        ref.load_state_dict(mod.state_dict())
        mod = mod.to(DEV)
        x = rb(torch.randn(2, 64, 16, 16, generator=g)).requires_grad_(True)
        y_ref = ref(x) if name == "attn" else ref(x, None)
        dy = rb(torch.randn(y_ref.shape, generator=g))
        y_ref.backward(dy)
        xd = cl_bf16(x.detach()).requires_grad_(True)
        y = mod(xd)
        assert y.dtype == BF
        close(y, y_ref, 2e-2, name + " y")
        y.backward(cl_bf16(dy))
        close(xd.grad, x.grad, 3e-2, name + " dx")
        refp = dict(ref.named_parameters())
        for k, p in mod.named_parameters():
            assert p.grad.dtype == torch.float32
            if k == "k.bias":
                # softmax is invariant to a per-row shift of the scores, so d k.bias is exactly 0 in exact arithmetic; with bf16
                # dK rows the column sum leaves rounding noise: bound it by the scale of the sibling bias gradients
                assert p.grad.abs().max().item() <= 3e-2 * refp["v.bias"].grad.abs().max().item(), "attn dk.bias noise"
                continue
            close(p.grad, refp[k].grad, 3e-2, "%s d%s" % (name, k))


@pytest.mark.parametrize("n,cin,cout,h,w,res", [(2, 64, 128, 16, 16, False), (3, 128, 256, 20, 36, True), (1, 256, 512, 8, 16, True),
                                                (2, 128, 128, 33, 18, False), (4, 128, 128, 64, 64, True)])
def test_groupnorm_statistics_from_the_bf16_conv_epilogue(hip_lib, monkeypatch, n, cin, cout, h, w, res):
    """SURVEY.md 2.1, GroupNorm row, mixed-precision path: the stride-1 3x3 kernel leaves (sum, sum of squares) of its bf16-ROUNDED output
    per output tile and channel group (4 / 8 / 16 channels per group, ragged tiles, two and four 128-channel blocks), and the GroupNorm
    behind it runs finalize + apply only.  Checked: the partials against f64 sums of the tensor the conv wrote, the normalised output
    and every gradient against the path with the statistics pass (same kernels otherwise), and against torch on the host."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(n + cin + cout + h)
    x = rb(torch.randn(n, cin, h, w, generator=g))
    wt = rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = 0.1 * torch.randn(cout, generator=g)
    r = rb(torch.randn(n, cout, h, w, generator=g)) if res else None
    gamma, beta = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    gy = cl_bf16(torch.randn(n, cout, h, w, generator=g))
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "GN_FUSED_STATS", fused)
        xd = cl_bf16(x).requires_grad_(True)
        wd = wt.to(DEV).requires_grad_(True)
        gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
        y = ops.conv3x3(xd, wd, b.to(DEV), cl_bf16(r) if res else None, gn_stats=True)
        part = ops._gn_partials_of(y, 32)
        assert (part is not None) == fused
        if fused:
            assert part.shape == (n, hip_lib.odvae_conv_bf16_stats_chunks(h, w), 32, 2)
            yc = y.detach().float().double().cpu().reshape(n, 32, cout // 32, h * w)
            want = torch.stack([yc.sum(dim=(2, 3)), (yc * yc).sum(dim=(2, 3))], dim=-1)
            got = part.double().cpu().sum(dim=1)
            assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        z = ops.group_norm(y, gd, bd, 32, 1e-6, swish=True)
        z.backward(gy)
        outs.append((y.detach(), z.detach(), xd.grad, wd.grad, gd.grad, bd.grad))
    assert torch.equal(outs[0][0], outs[1][0])                 # the conv output does not depend on the switch
    for a, c, what, tol in zip(outs[0][1:], outs[1][1:], ("z", "dx", "dw", "dgamma", "dbeta"), (8e-3, 8e-3, 1e-3, 1e-3, 1e-3)):
        # mean / rstd differ by f32 summation order only; z and dx are bf16 tensors: a last-place flip is 2^-8 of the value
        assert (a.float() - c.float()).abs().max().item() <= tol * c.float().abs().max().item(), what
    ref_y = rb(F.conv2d(x, wt, b, padding=1) + (r if res else 0.0))
    ref_z = F.silu(F.group_norm(ref_y, 32, gamma, beta, eps=1e-6))
    close(outs[0][1], ref_z, 2e-2, "conv -> GroupNorm(statistics from the epilogue) + swish vs torch")

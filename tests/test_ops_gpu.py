"""Per-kernel parity: HIP path (through the C ABI) vs torch CPU fp32 functional ops on the same seeded inputs.
Tolerance: max|hip - ref| <= TOL * max(1, max|ref|), TOL = 2e-4 forward / 5e-4 backward unless noted
(fp32 with different summation order; K up to a few thousand)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-4
BWD_TOL = 5e-4


def close(a, b, tol, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert err <= tol * ref, "%s: max err %.3e > %.1e * %.3e" % (what, err, tol, ref)


def dev():
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------------
@pytest.fixture(params=[-1, 0, 1, 2], ids=["staging-per-shape", "staging-registers", "staging-lds-dma-32", "staging-lds-dma-16"])
def gemm_staging(request, hip_lib):
    """Both operand-staging forms of gemm_f32.hip (and the per-shape default) under the GEMM-backed tests; they compute the same
    products in the same order."""
    prev = hip_lib.odvae_gemm_select_staging(request.param)
    yield request.param
    hip_lib.odvae_gemm_select_staging(prev)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k,batch", [(128, 128, 64, 1), (200, 72, 52, 3), (16, 16, 512, 2), (4, 260, 36, 1), (260, 132, 100, 2)])
def test_gemm(hip_lib, gemm_staging, ta, tb, m, n, k, batch):
    from odvae_amd import ops
    if ta and m % 4:
        pytest.skip("transA needs M % 4 == 0")
    if not tb and n % 4:
        pytest.skip("transB=0 needs N % 4 == 0")
    g = torch.Generator().manual_seed(m * 131 + n * 17 + k + batch + ta * 2 + tb)
    a = torch.randn(batch, *((k, m) if ta else (m, k)), generator=g)
    b = torch.randn(batch, *((n, k) if tb else (k, n)), generator=g)
    bias = torch.randn(n, generator=g)
    res = torch.randn(batch, m, n, generator=g)
    prod = torch.bmm(a.transpose(1, 2) if ta else a, b.transpose(1, 2) if tb else b)
    ad, bd, biasd, resd = a.to(dev()), b.to(dev()), bias.to(dev()), res.to(dev())
    lda, sa, ldb, sb = a.shape[2], a.shape[1] * a.shape[2], b.shape[2], b.shape[1] * b.shape[2]
    # bias + residual (the residual seeds the accumulators, which requires alpha == 1)
    c = torch.full((batch, m, n), float("nan"), device=dev())
    ops.gemm(ta, tb, m, n, k, 1.0, ad, lda, sa, bd, ldb, sb, c, n, m * n, biasd, resd, batch)
    close(c, prod + bias + res, FWD_TOL * math.sqrt(k / 32), "gemm bias+residual")
    # alpha + bias, no residual
    c2 = torch.full((batch, m, n), float("nan"), device=dev())
    ops.gemm(ta, tb, m, n, k, 0.5, ad, lda, sa, bd, ldb, sb, c2, n, m * n, biasd, None, batch)
    close(c2, 0.5 * prod + bias, FWD_TOL * math.sqrt(k / 32), "gemm alpha+bias")
    from odvae_amd import lib
    with pytest.raises(lib.HipLibraryError):
        ops.gemm(ta, tb, m, n, k, 0.5, ad, lda, sa, bd, ldb, sb, c2, n, m * n, biasd, resd, batch)


def test_gemm_staging_forms_are_bit_identical(hip_lib):
    """Register-staged and LDS-DMA forms (32- and 16-wide steps): same fragments, same MFMA order -- the results are equal bit for bit (all four layouts, a K
    that is no multiple of the 32-wide step, ragged M / N, batch stride)."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(77)
    m, n, k, batch = 260, 132, 100, 3
    prev = hip_lib.odvae_gemm_select_staging(-1)
    try:
        for ta in (0, 1):
            for tb in (0, 1):
                a = torch.randn(batch, *((k, m) if ta else (m, k)), generator=g).to(dev())
                b = torch.randn(batch, *((n, k) if tb else (k, n)), generator=g).to(dev())
                outs = []
                for mode in (0, 1, 2):
                    hip_lib.odvae_gemm_select_staging(mode)
                    c = torch.empty(batch, m, n, device=dev())
                    ops.gemm(ta, tb, m, n, k, 1.0, a, a.shape[2], a.shape[1] * a.shape[2], b, b.shape[2], b.shape[1] * b.shape[2], c, n, m * n, None, None, batch)
                    outs.append(c)
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (ta, tb)
    finally:
        hip_lib.odvae_gemm_select_staging(prev)


def test_gemm_splitk(hip_lib, gemm_staging):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(7)
    m, n, k = 64, 96, 16384
    a = torch.randn(k, m, generator=g)  # stored [K][M] (transA), the wgrad shape
    b = torch.randn(k, n, generator=g)
    ref = (a.double().t() @ b.double()).float()
    assert hip_lib.odvae_gemm_f32_workspace_bytes(m, n, k, 1) > 0
    c = torch.empty(m, n, device=dev())
    ops.gemm(1, 0, m, n, k, 1.0, a.to(dev()), m, 0, b.to(dev()), n, 0, c, n, 0)
    close(c, ref, 1e-5 * math.sqrt(k), "gemm split-K")


# ------------------------------------------------------------------------------------------------------
def ref_conv(mode, x, w, b):
    if mode == 0:
        return F.conv2d(x, w, b, stride=1, padding=1)
    if mode == 1:
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2, padding=0)
    return F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, stride=1, padding=1)


CONV_CASES = [
    # mode, N, Cin, Cout, H, W
    (0, 2, 32, 64, 24, 16),
    (0, 1, 3, 128, 16, 32),
    (0, 2, 128, 3, 16, 16),
    (0, 2, 16, 32, 4, 4),
    (0, 1, 64, 160, 20, 12),
    (0, 1, 256, 256, 8, 8),
    (0, 2, 3, 160, 8, 48),      # thin-side weight gradient, channel tiles past Cb
    (0, 1, 160, 2, 12, 16),     # thin output side with 2 channels
    (0, 3, 1, 64, 5, 16),       # thin input side with 1 channel
    (0, 1, 64, 128, 20, 12),    # 8-wave Winograd kernel, ragged 8x16 pixel tiles on both axes
    (0, 2, 36, 256, 6, 10),     # 8-wave Winograd kernel, Cin padded to the 16-channel chunk, two co blocks
    (1, 2, 32, 32, 16, 32),
    (1, 1, 128, 128, 8, 8),
    (2, 2, 32, 32, 8, 16),
    (2, 1, 128, 128, 4, 4),
    (2, 2, 64, 160, 11, 21),    # ragged tiles, Cout not a multiple of 128
    (2, 1, 40, 24, 9, 5),       # narrow Cout (32-wide tile), Cin padded to 64
    (2, 1, 256, 256, 16, 16),
]


@pytest.mark.parametrize("mode,n,cin,cout,h,w", CONV_CASES)
def test_conv3x3_fwd_bwd(hip_lib, mode, n, cin, cout, h, w):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(1000 * mode + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = ref_conv(mode, xr, wr, br)
    res = torch.randn(y_ref.shape, generator=g)
    resr = res.clone().requires_grad_(True)
    y_ref = y_ref + resr
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)

    xd = x.to(dev()).requires_grad_(True)
    wd = wt.to(dev()).requires_grad_(True)
    bd = b.to(dev()).requires_grad_(True)
    resd = res.to(dev()).requires_grad_(True)
    y = ops.conv3x3(xd, wd, bd, resd, mode)
    close(y, y_ref, FWD_TOL, "conv fwd")
    y.backward(gy.to(dev()))
    close(xd.grad, xr.grad, BWD_TOL, "conv dx")
    close(wd.grad, wr.grad, BWD_TOL * math.sqrt(n * h * w / 64), "conv dw")
    close(bd.grad, br.grad, BWD_TOL * math.sqrt(n * h * w / 64), "conv db")
    close(resd.grad, resr.grad, 1e-6, "conv dres")


@pytest.mark.parametrize("case", [(0, 2, 32, 64, 24, 16), (0, 1, 64, 160, 20, 12), (0, 1, 256, 256, 8, 8)])
def test_conv3x3_direct_kernel_still_matches(hip_lib, monkeypatch, case):
    """ODVAE_CONV_WINOGRAD=0: the direct implicit-GEMM kernel (the in-tree reference of the Winograd path) on the shapes
    that Winograd serves by default."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "WINOGRAD", False)
    monkeypatch.setattr(ops, "WGRAD_WINOGRAD", False)   # and the direct weight-gradient kernel on the 256 -> 256 case
    test_conv3x3_fwd_bwd(hip_lib, *case)


# Winograd against the direct kernel, in max|difference| / max|y|.  F(2x2, 3x3) differs by summation order only; F(4x4, 3x3)
# (interpolation points 0, +-1, +-2: transform coefficients up to 8 and 1/24) is an order of magnitude less exact in f32:
# measured 0.5e-5 .. 2.4e-5 against an f64 convolution over Cin = 64 .. 512 (tools/wino4_time.py), bound here at 5e-5.
WINO_TOL = {False: 1e-5, True: 5e-5}


@pytest.mark.parametrize("f4", [False, True])
def test_winograd_and_direct_kernels_agree(hip_lib, monkeypatch, f4):
    """Same f32 inputs through both kernels: forward and data gradient agree to a few 1e-6 (F(2x2)) / 1e-5 (F(4x4)) of max|y|."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "WINOGRAD4", f4)
    assert bool(ops._wino4_ok(32, 48, 128, 128)) == f4
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 128, 32, 48, generator=g).to(dev())
    w = (torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(9 * 128)).to(dev())
    gy = torch.randn(2, 128, 32, 48, generator=g).to(dev())
    outs = []
    for wino in (True, False):
        monkeypatch.setattr(ops, "WINOGRAD", wino)
        xd = x.clone().requires_grad_(True)
        y = ops.conv3x3(xd, w.clone().requires_grad_(True))
        y.backward(gy)
        outs.append((y.detach(), xd.grad))
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= WINO_TOL[f4] * b.abs().max().item()


@pytest.mark.parametrize("f4", [False, True])
@pytest.mark.parametrize("n,cin,cout,h,w", [(4, 128, 128, 128, 128), (5, 64, 128, 112, 160), (3, 128, 256, 64, 96), (2, 64, 512, 64, 128),
                                            (8, 128, 128, 128, 128), (3, 64, 128, 160, 320)])
def test_winograd_persistent_form(hip_lib, monkeypatch, n, cin, cout, h, w, f4):
    """Layers with at least two tiles per block run the persistent form of the Winograd kernel (the chunk pipeline carries on
    across the tile boundary; conv3x3_wino_f32.hip): 512 / 700 / 288 / 256 tiles over 256 / 256 / 128 / 64 blocks per output-channel
    block, i.e. even and uneven tile counts per block and one, two and four co blocks.  Forward (bias + residual) against torch on
    the host, forward and data gradient against the direct kernel, and a bit-identical repeat."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "WINOGRAD4", f4)     # F(4x4): 16 x 32-pixel tiles, 64-channel blocks -- the last two shapes give it 256 / 300 tiles over 128 blocks
    g = torch.Generator().manual_seed(n * 7 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    gy = torch.randn(n, cout, h, w, generator=g).to(dev())
    ref = torch.nn.functional.conv2d(x, wt, b, padding=1) + res
    outs = []
    for wino in (True, False):
        monkeypatch.setattr(ops, "WINOGRAD", wino)
        xd = x.to(dev()).requires_grad_(True)
        y = ops.conv3x3(xd, wt.to(dev()).requires_grad_(True), b.to(dev()), res.to(dev()))
        y.backward(gy)
        outs.append((y.detach(), xd.grad))
    close(outs[0][0], ref, FWD_TOL, "persistent winograd fwd vs torch")
    for a, d in zip(*outs):
        assert (a - d).abs().max().item() <= WINO_TOL[f4] * d.abs().max().item()
    monkeypatch.setattr(ops, "WINOGRAD", True)
    with torch.no_grad():
        again = ops.conv3x3(x.to(dev()), wt.to(dev()), b.to(dev()), res.to(dev()))
    assert torch.equal(again, outs[0][0])


@pytest.mark.parametrize("n,cin,cout,h,w,res,bias", [(1, 64, 64, 16, 32, True, True), (2, 72, 88, 20, 36, False, True), (1, 512, 512, 32, 32, True, False),
                                                     (2, 64, 192, 36, 68, False, False)])
def test_winograd4_through_the_c_abi(hip_lib, n, cin, cout, h, w, res, bias):
    """odvae_conv3x3_pack_wino4_f32 + odvae_conv3x3_wino4_f32 called directly (minimum tile, partial blocks in both directions, channel counts
    that are no multiple of the 64-channel block, 64 chunks; with / without bias and residual) against an f64 convolution on the host, forward
    pack and data-gradient pack; then the contract: what `supported` names, and the error returns for shapes the kernel cannot take."""
    from odvae_amd import lib as _lib, ops
    L = hip_lib
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, h, w, cin, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g) if bias else None
    r = torch.randn(n, h, w, cout, generator=g) if res else None
    gy = torch.randn(n, h, w, cout, generator=g)
    xd, wd, gyd = x.to(dev()), wt.to(dev()), gy.to(dev())
    fwd = torch.empty(L.odvae_conv3x3_wino4_pack_floats(cin, cout), device=dev())
    dgr = torch.empty(L.odvae_conv3x3_wino4_pack_floats(cout, cin), device=dev())
    _lib.check(L.odvae_conv3x3_pack_wino4_f32(wd.data_ptr(), cout, cin, fwd.data_ptr(), dgr.data_ptr(), _lib.stream_ptr()), "pack_wino4")
    y = torch.empty(n, h, w, cout, device=dev())
    bd, rd = (b.to(dev()) if bias else None), (r.to(dev()) if res else None)
    _lib.check(L.odvae_conv3x3_wino4_f32(xd.data_ptr(), n, h, w, cin, fwd.data_ptr(), cout, _lib.ptr(bd), _lib.ptr(rd), y.data_ptr(), 0,
                                         _lib.stream_ptr()), "wino4 fwd")
    dx = torch.empty(n, h, w, cin, device=dev())
    _lib.check(L.odvae_conv3x3_wino4_f32(gyd.data_ptr(), n, h, w, cout, dgr.data_ptr(), cin, None, None, dx.data_ptr(), 0, _lib.stream_ptr()), "wino4 dgrad")
    x64 = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    ref = F.conv2d(x64, wt.double(), b.double() if bias else None, padding=1)
    dref, = torch.autograd.grad(ref, x64, gy.permute(0, 3, 1, 2).double())
    if res:
        ref = ref + r.permute(0, 3, 1, 2).double()
    for got, want, what in ((y, ref, "forward"), (dx, dref, "data gradient")):
        err = (got.cpu().permute(0, 3, 1, 2).double() - want.detach()).abs().max().item()
        assert err <= WINO_TOL[True] * want.abs().max().item(), "%s: %.3e of max|y|" % (what, err / want.abs().max().item())
    assert L.odvae_conv3x3_wino4_supported(h, w, cin, cout) == 1
    assert L.odvae_conv3x3_wino4_supported(h + 2, w, cin, cout) == 0          # H, W in multiples of 4
    assert L.odvae_conv3x3_wino4_supported(h, 16, cin, cout) == 0             # at least one 16 x 32 block
    assert L.odvae_conv3x3_wino4_supported(h, w, cin + 4, cout) == 0          # channels in chunks of 8
    assert L.odvae_conv3x3_wino4_supported(h, w, 32, cout) == 0               # thin layers stay on F(2x2)
    assert L.odvae_conv3x3_wino4_f32(xd.data_ptr(), n, h + 2, w, cin, fwd.data_ptr(), cout, None, None, y.data_ptr(), 0, _lib.stream_ptr()) != 0
    assert b"multiples of 4" in L.odvae_last_error()
    assert L.odvae_conv3x3_wino4_f32(xd.data_ptr(), n, h, w, cin + 4, fwd.data_ptr(), cout, None, None, y.data_ptr(), 0, _lib.stream_ptr()) != 0
    assert L.odvae_conv3x3_wino4_f32(None, n, h, w, cin, fwd.data_ptr(), cout, None, None, y.data_ptr(), 0, _lib.stream_ptr()) != 0
    assert L.odvae_conv3x3_wino4_f32(xd.data_ptr(), n, h, w, cin, fwd.data_ptr(), cout, None, None, y.data_ptr(), 1, _lib.stream_ptr()) != 0   # no fused ReLU


@pytest.mark.parametrize("n,cin,cout,h,w,res", [(2, 64, 128, 16, 32, False), (3, 128, 256, 36, 68, True), (1, 256, 512, 32, 32, True),
                                                (2, 64, 64, 20, 36, False), (8, 128, 128, 128, 128, True)])
def test_groupnorm_statistics_from_the_conv_epilogue(hip_lib, monkeypatch, n, cin, cout, h, w, res):
    """SURVEY.md 2.1, GroupNorm row: the F(4x4) conv's output transform leaves (sum, sum of squares) of y per output tile and channel group
    (4 / 8 / 16 / 2 channels per group; partial blocks; the persistent form; with / without a residual), and the GroupNorm that reads y runs
    finalize + apply only.  Checked: the partials against sums of the tensor the conv wrote (f64 on the host), GroupNorm + swish through
    the fused path against torch on the host and against the unfused HIP path, and the backward pass through both layers."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "WINOGRAD4", True)
    g = torch.Generator().manual_seed(n + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g)
    r = torch.randn(n, cout, h, w, generator=g) if res else None
    gamma, beta = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    gy = torch.randn(n, cout, h, w, generator=g).to(dev())
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "GN_FUSED_STATS", fused)
        xd = x.to(dev()).requires_grad_(True)
        wd = wt.to(dev()).requires_grad_(True)
        gd, bd = gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
        y = ops.conv3x3(xd, wd, b.to(dev()), r.to(dev()) if res else None, gn_stats=True)
        part = ops._gn_partials_of(y, 32)
        assert (part is not None) == fused
        if fused:
            assert part.shape == (n, hip_lib.odvae_conv3x3_wino4_stats_chunks(h, w), 32, 2)
            yc = y.detach().double().cpu().reshape(n, 32, cout // 32, h * w)
            want = torch.stack([yc.sum(dim=(2, 3)), (yc * yc).sum(dim=(2, 3))], dim=-1)          # [n][32][2]
            got = part.double().cpu().sum(dim=1)
            assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        z = ops.group_norm(y, gd, bd, 32, 1e-6, swish=True)
        z.backward(gy)
        outs.append((y.detach(), z.detach(), xd.grad, wd.grad, gd.grad, bd.grad))
    ref_y = F.conv2d(x, wt, b, padding=1) + (r if res else 0.0)
    ref_z = F.silu(F.group_norm(ref_y, 32, gamma, beta, eps=1e-6))
    close(outs[0][1], ref_z, 5e-4, "conv -> GroupNorm(stats from the epilogue) + swish vs torch")
    assert torch.equal(outs[0][0], outs[1][0])                      # the conv output itself does not depend on the switch
    for a, c, what in zip(outs[0][1:], outs[1][1:], ("z", "dx", "dw", "dgamma", "dbeta")):
        assert (a - c).abs().max().item() <= 2e-5 * c.abs().max().item(), what


@pytest.mark.parametrize("n,c,cout,h,w,skip", [(2, 64, 64, 16, 32, False), (3, 128, 64, 48, 64, True), (1, 256, 128, 20, 36, True),
                                               (2, 64, 128, 32, 32, False)])
def test_groupnorm_backward_sums_from_the_conv_data_gradient(hip_lib, monkeypatch, n, c, cout, h, w, skip):
    """conv3x3(swish(GroupNorm(x))): the conv's data-gradient launch (F(4x4)) also leaves the per-channel sums of the GroupNorm's
    backward (odvae_conv3x3_wino4_gnbwd_f32 -> odvae_groupnorm_bwd_partials_f32), so the GroupNorm's reduce pass does not run.  Against
    torch CPU and against the same graph with the fusion off; ragged tiles, three images, Cin != Cout, a skip gradient folded in."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "WINOGRAD4", True)
    g = torch.Generator().manual_seed(c + h + n)
    x = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3
    wt = torch.randn(cout, c, 3, 3, generator=g) / (3.0 * c ** 0.5)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    gy = torch.randn(n, cout, h, w, generator=g)
    gs = torch.randn(n, c, h, w, generator=g)
    xr, wr, gr, br = (t.clone().requires_grad_(True) for t in (x, wt, gamma, beta))
    ref = F.conv2d(F.silu(F.group_norm(xr, 32, gr, br, eps=1e-6)), wr, None, padding=1)
    (ref * gy).sum().backward() if not skip else ((ref * gy).sum() + (xr * gs).sum()).backward()
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "GN_FUSED_BWD", fused)
        hits = ops.GN_FUSED_BWD_HITS
        xd, wd, gd, bd = (t.to(dev()).requires_grad_(True) for t in (x, wt, gamma, beta))
        xin = xd.contiguous(memory_format=torch.channels_last)
        if skip:
            a, xs = ops.group_norm_skip(xin, gd, bd, 32, 1e-6, swish=True)
        else:
            a, xs = ops.group_norm(xin, gd, bd, 32, 1e-6, swish=True), None
        y = ops.conv3x3(a, wd, None, None)
        loss = (y * gy.to(dev())).sum() + ((xs * gs.to(dev())).sum() if skip else 0.0)
        loss.backward()
        assert (ops.GN_FUSED_BWD_HITS - hits) == (1 if fused else 0)
        outs[fused] = (y.detach(), xd.grad, wd.grad, gd.grad, bd.grad)
    for got, other, want, what in zip(outs[True], outs[False], (ref, xr.grad, wr.grad, gr.grad, br.grad), ("y", "dx", "dw", "dgamma", "dbeta")):
        close(got, want, 5e-4 if what in ("y", "dx") else BWD_TOL * 4, "GroupNorm backward sums from the data gradient: " + what)
        assert (got - other).abs().max().item() <= 3e-5 * max(1.0, other.abs().max().item()), what + " vs the two-kernel form"


def test_groupnorm_backward_sums_are_not_used_for_another_gradient(hip_lib, monkeypatch):
    """Two consumers of swish(GroupNorm(x)): autograd sums their gradients into a new tensor, which is not the one the first conv's data
    gradient made its sums from -- the GroupNorm must run its own reduce pass."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "WINOGRAD4", True)
    monkeypatch.setattr(ops, "GN_FUSED_BWD", True)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 16, 32, generator=g)
    wt = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    gamma, beta = torch.randn(64, generator=g), torch.randn(64, generator=g)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    ar = F.silu(F.group_norm(xr, 32, gr, br, eps=1e-6))
    (F.conv2d(ar, wt, None, padding=1).sum() + (ar * ar).sum()).backward()
    xd, gd, bd = (t.to(dev()).requires_grad_(True) for t in (x, gamma, beta))
    hits = ops.GN_FUSED_BWD_HITS
    a = ops.group_norm(xd.contiguous(memory_format=torch.channels_last), gd, bd, 32, 1e-6, swish=True)
    (ops.conv3x3(a, wt.to(dev()), None, None).sum() + (a * a).sum()).backward()
    assert ops.GN_FUSED_BWD_HITS == hits
    close(xd.grad, xr.grad, 5e-4, "dx with two consumers")
    close(gd.grad, gr.grad, BWD_TOL * 4, "dgamma with two consumers")


def test_epilogue_statistics_are_dropped_when_the_tensor_was_written_to(hip_lib, monkeypatch):
    """The statistics ride on the conv's output tensor OBJECT, tagged with its storage pointer, version counter and shape.  An in-place
    write between the conv and the GroupNorm (a hook, a future fusion) changes the version: the GroupNorm must then take its own
    statistics pass and still normalise the values the tensor holds NOW."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "WINOGRAD4", True)
    monkeypatch.setattr(ops, "GN_FUSED_STATS", True)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 16, 32, generator=g).to(dev())
    wt = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(dev())
    gamma, beta = torch.randn(64, generator=g).to(dev()), torch.randn(64, generator=g).to(dev())
    y = ops.conv3x3(x, wt, None, None, gn_stats=True)
    assert ops._gn_partials_of(y, 32) is not None
    assert ops._gn_partials_of(y, 16) is None                     # another group count: not these statistics
    assert ops._gn_partials_of(y.clone(), 32) is None             # a copy is another object
    y.mul_(3.0).add_(1.0)                                         # in-place: same object, same storage, new version
    assert ops._gn_partials_of(y, 32) is None
    z = ops.group_norm(y, gamma, beta, 32, 1e-6, swish=True)
    ref = F.silu(F.group_norm(y.detach().cpu().contiguous(), 32, gamma.cpu(), beta.cpu(), eps=1e-6))
    close(z, ref, 5e-4, "GroupNorm after an in-place write (stale epilogue statistics must not be used)")


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 128, 128, 16, 16), (1, 128, 256, 8, 12), (3, 256, 128, 6, 10),
                                            (2, 128, 128, 2, 2), (1, 384, 128, 4, 34)])
def test_winograd_domain_weight_gradient(hip_lib, n, cin, cout, h, w):
    """odvae_conv3x3_wgrad_wino_f32 through the C ABI against torch CPU (weight and bias gradient) and against the direct
    weight-gradient kernel: ragged last chunk (tiles not a multiple of 16), tiles of one chunk in different images, a
    single tile row, three channel blocks."""
    from odvae_amd import lib as _lib, ops
    L = hip_lib
    g = torch.Generator().manual_seed(n * 1000 + cin + h)
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    b = torch.zeros(cout, requires_grad=True)
    F.conv2d(x, wt, b, padding=1).backward(dy)
    xd = x.to(dev()).permute(0, 2, 3, 1).contiguous()
    dyd = dy.to(dev()).permute(0, 2, 3, 1).contiguous()
    assert L.odvae_conv3x3_wgrad_wino_supported(n, h, w, cin, cout) == 1
    dw = torch.empty(cout, cin, 3, 3, device=dev()); db = torch.empty(cout, device=dev())
    wp, wn = ops._ws(L.odvae_conv3x3_wgrad_wino_workspace_bytes(n, h, w, cin, cout), xd)
    _lib.check(L.odvae_conv3x3_wgrad_wino_f32(xd.data_ptr(), dyd.data_ptr(), n, h, w, cin, cout, dw.data_ptr(), db.data_ptr(),
                                              wp, wn, _lib.stream_ptr()), "wgrad_wino")
    dw2 = torch.empty_like(dw); db2 = torch.empty_like(db)
    wp, wn = ops._ws(L.odvae_conv3x3_wgrad_workspace_bytes(0, n, h, w, cin, cout), xd)
    _lib.check(L.odvae_conv3x3_wgrad_f32(0, xd.data_ptr(), dyd.data_ptr(), n, h, w, cin, h, w, cout, dw2.data_ptr(), db2.data_ptr(),
                                         wp, wn, _lib.stream_ptr()), "wgrad")
    close(dw, wt.grad, BWD_TOL, "winograd-domain dw vs torch")
    close(db, b.grad, BWD_TOL, "winograd-domain dbias vs torch")
    close(dw, dw2.cpu(), 1e-5, "winograd-domain dw vs direct kernel")
    # unsupported shapes are refused, not mis-served
    assert L.odvae_conv3x3_wgrad_wino_supported(n, h + 1, w, cin, cout) == 0
    assert L.odvae_conv3x3_wgrad_wino_supported(n, h, w, cin + 4, cout) == 0
    rc = L.odvae_conv3x3_wgrad_wino_f32(xd.data_ptr(), dyd.data_ptr(), n, h, w, cin + 4, cout, dw.data_ptr(), db.data_ptr(),
                                        wp, wn, _lib.stream_ptr())
    assert rc != 0


def test_upsample_conv_dense_form_still_matches(hip_lib, monkeypatch):
    """ODVAE_UPCONV_DENSE=1 keeps mode 2 (dense 3x3 at 2x resolution; data gradient = mode 0 + 2x2 sum-pool) for A/B."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "UPCONV_BY_PARITY", False)
    test_conv3x3_fwd_bwd(hip_lib, 2, 2, 64, 160, 11, 21)


def test_conv3x3_no_bias_no_res(hip_lib):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 32, 16, 16, generator=g)
    wt = torch.randn(32, 32, 3, 3, generator=g) / 17
    y = ops.conv3x3(x.to(dev()), wt.to(dev()))
    close(y, F.conv2d(x, wt, None, 1, 1), FWD_TOL, "conv plain")


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,h,w,swish", [(2, 32, 8, 8, True), (2, 64, 16, 8, False), (3, 128, 12, 20, True),
                                             (1, 512, 16, 16, True), (2, 256, 4, 4, False)])
def test_groupnorm(hip_lib, n, c, h, w, swish):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    gamma = torch.randn(c, generator=g)
    beta = torch.randn(c, generator=g)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    y_ref = F.group_norm(xr, 32, gr, br, eps=1e-6)
    if swish:
        y_ref = y_ref * torch.sigmoid(y_ref)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd, gd, bd = (t.to(dev()).requires_grad_(True) for t in (x, gamma, beta))
    y = ops.group_norm(xd, gd, bd, 32, 1e-6, swish)
    close(y, y_ref, FWD_TOL, "gn fwd")
    y.backward(gy.to(dev()))
    close(xd.grad, xr.grad, BWD_TOL, "gn dx")
    close(gd.grad, gr.grad, BWD_TOL * 4, "gn dgamma")
    close(bd.grad, br.grad, BWD_TOL * 4, "gn dbeta")


def test_group_norm_skip_sums_both_gradients(hip_lib):
    """group_norm_skip hands x out a second time for the block's skip connection and folds that branch's gradient into
    the GroupNorm backward pass: dx = dGN(dy) + dskip, exactly what autograd's separate add would produce."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 64, 8, 8, generator=g)
    gamma, beta = torch.randn(64, generator=g), torch.randn(64, generator=g)
    gy, gs = torch.randn(2, 64, 8, 8, generator=g), torch.randn(2, 64, 8, 8, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.group_norm(xr, 32, gamma, beta, eps=1e-6)
    (y_ref * torch.sigmoid(y_ref) * gy).sum().backward()
    ref = xr.grad + gs
    xd = x.to(dev()).requires_grad_(True)
    y, xs = ops.group_norm_skip(xd, gamma.to(dev()), beta.to(dev()), 32, 1e-6, True)
    assert xs.data_ptr() == xd.data_ptr() or torch.equal(xs, xd)
    ((y * gy.to(dev())).sum() + (xs * gs.to(dev())).sum()).backward()
    close(xd.grad, ref, BWD_TOL, "gn skip dx")
    # the skip output alone (GroupNorm branch unused) still passes its gradient through
    xd2 = x.to(dev()).requires_grad_(True)
    _, xs2 = ops.group_norm_skip(xd2, gamma.to(dev()), beta.to(dev()), 32, 1e-6, True)
    (xs2 * gs.to(dev())).sum().backward()
    close(xd2.grad, gs, 1e-7, "gn skip only")


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 32, 64, 8, 8), (1, 128, 128, 16, 16), (3, 32, 16, 4, 4), (2, 16, 16, 4, 4)])
def test_conv1x1(hip_lib, gemm_staging, n, cin, cout, h, w):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(cin * cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    xr, wr, br, rr = (t.clone().requires_grad_(True) for t in (x, wt, b, res))
    y_ref = F.conv2d(xr, wr, br) + rr
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd, wd, bd, rd = (t.to(dev()).requires_grad_(True) for t in (x, wt, b, res))
    y = ops.conv1x1(xd, wd, bd, rd)
    close(y, y_ref, FWD_TOL, "1x1 fwd")
    y.backward(gy.to(dev()))
    close(xd.grad, xr.grad, BWD_TOL, "1x1 dx")
    close(wd.grad, wr.grad, BWD_TOL * 2, "1x1 dw")
    close(bd.grad, br.grad, BWD_TOL * 2, "1x1 db")
    close(rd.grad, rr.grad, 1e-6, "1x1 dres")


@pytest.mark.parametrize("fused", [True, False], ids=["softmax-bwd-in-gemm", "softmax-bwd-separate"])
@pytest.mark.parametrize("n,c,h,w", [(2, 32, 4, 4), (2, 64, 16, 16), (1, 256, 8, 16), (2, 64, 10, 18)])
def test_attention(hip_lib, gemm_staging, monkeypatch, n, c, h, w, fused):
    """fused: the softmax backward rides in the epilogue of the dP product (odvae_gemm_softmax_bwd_f32 + odvae_rowdot_f32);
    separate: bmm, then odvae_softmax_rows_bwd_f32.  (10 x 18 = 180 tokens: ragged 128-wide tiles on both axes.)"""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "FUSED_SOFTMAX_BWD", fused)
    g = torch.Generator().manual_seed(c + h * w)
    qkv = torch.randn(n, 3 * c, h, w, generator=g)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr[:, :c], qr[:, c:2 * c], qr[:, 2 * c:]
    t = h * w
    w_ = torch.bmm(q.reshape(n, c, t).permute(0, 2, 1), k.reshape(n, c, t)) * (c ** -0.5)
    w_ = F.softmax(w_, dim=2)
    o_ref = torch.bmm(v.reshape(n, c, t), w_.permute(0, 2, 1)).reshape(n, c, h, w)
    go = torch.randn(o_ref.shape, generator=g)
    o_ref.backward(go)
    outs = {}
    for folded in (True, False):      # folded: softmax inside the two forward products (exp against a row bound, row sums in the PV product)
        monkeypatch.setattr(ops, "ATTN_FOLDED_SOFTMAX", folded)
        qd = qkv.to(dev()).requires_grad_(True)
        o = ops.attention_qkv(qd)
        close(o, o_ref, FWD_TOL, "attn fwd (folded softmax: %s)" % folded)
        o.backward(go.to(dev()))
        close(qd.grad, qr.grad, BWD_TOL, "attn dqkv (folded softmax: %s)" % folded)
        if folded:
            assert int(ops._ATTN_LAST_FLAG.item()) == 0      # the bound held: no fallback ran
        outs[folded] = (o.detach(), qd.grad)
    for a, b, what in zip(outs[True], outs[False], ("o", "dqkv")):
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item()), what + ": folded vs separate softmax"   # (exp arguments are ~|bound| instead of ~|max| below zero: their f32 rounding scales with it)


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 16, 16), (1, 128, 10, 18)])
def test_attention_folded_softmax_falls_back_when_the_row_bound_underflows(hip_lib, monkeypatch, n, c, h, w):
    """|q_i| max_j |k_j| C^-1/2 far above the largest score of the row (long q and k at right angles): every exponential relative to
    the bound is 0 in f32, the PV product sees a zero row sum and raises the device flag, and the predicated fallback launches redo the
    block with the row maximum -- same result as the separate softmax pass, forward and backward, without a host round trip."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(c + h)
    qkv = torch.randn(n, 3 * c, h, w, generator=g)
    qkv[:, 0] += 400.0            # q: a large component along channel 0
    qkv[:, c + 1] += 400.0        # k: a large component along channel 1  ->  bound ~ 1.6e5 / sqrt(c), scores O(100)
    t = h * w
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr[:, :c], qr[:, c:2 * c], qr[:, 2 * c:]
    w_ = F.softmax(torch.bmm(q.reshape(n, c, t).permute(0, 2, 1), k.reshape(n, c, t)) * (c ** -0.5), dim=2)
    o_ref = torch.bmm(v.reshape(n, c, t), w_.permute(0, 2, 1)).reshape(n, c, h, w)
    go = torch.randn(o_ref.shape, generator=g)
    o_ref.backward(go)
    outs = {}
    for folded in (True, False):
        monkeypatch.setattr(ops, "ATTN_FOLDED_SOFTMAX", folded)
        qd = qkv.to(dev()).requires_grad_(True)
        o = ops.attention_qkv(qd)
        if folded:
            assert int(ops._ATTN_LAST_FLAG.item()) == 1      # the fallback did run
        o.backward(go.to(dev()))
        outs[folded] = (o.detach(), qd.grad)
    assert torch.equal(outs[True][0], outs[False][0])         # the fallback IS the separate-pass computation
    close(outs[True][0], o_ref, FWD_TOL, "attn fwd after the fallback")
    close(outs[True][1], outs[False][1], 1e-6, "attn dqkv after the fallback vs the separate pass")
    close(outs[True][1], qr.grad, BWD_TOL * 4, "attn dqkv after the fallback")


@pytest.mark.parametrize("n,c,h,w,budget_images", [(5, 64, 16, 16, 2), (3, 32, 10, 18, 1), (6, 256, 8, 16, 4)])
def test_attention_with_score_budget(hip_lib, monkeypatch, n, c, h, w, budget_images):
    """Past ops.ATTN_SCORE_BUDGET bytes of T x T scores the f32 AttnBlock runs group by group through one score buffer and recomputes
    P in the backward (what lets the f32 path run 512 x 512 at B = 32: 16 384 tokens = 1 GiB of scores per image and block).
    Forced here at small sizes, including a last group that is not full; same reference and tolerances as test_attention."""
    from odvae_amd import ops
    t = h * w
    monkeypatch.setattr(ops, "ATTN_SCORE_BUDGET", budget_images * t * t * 4)
    g = torch.Generator().manual_seed(c + h * w + n)
    qkv = torch.randn(n, 3 * c, h, w, generator=g)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr[:, :c], qr[:, c:2 * c], qr[:, 2 * c:]
    w_ = F.softmax(torch.bmm(q.reshape(n, c, t).permute(0, 2, 1), k.reshape(n, c, t)) * (c ** -0.5), dim=2)
    o_ref = torch.bmm(v.reshape(n, c, t), w_.permute(0, 2, 1)).reshape(n, c, h, w)
    go = torch.randn(o_ref.shape, generator=g)
    o_ref.backward(go)
    qd = qkv.to(dev()).requires_grad_(True)
    o = ops.attention_qkv(qd)
    assert o.grad_fn is not None and "Recompute" in type(o.grad_fn).__name__     # the budgeted path really ran
    close(o, o_ref, FWD_TOL, "attn fwd (score budget)")
    o.backward(go.to(dev()))
    close(qd.grad, qr.grad, BWD_TOL, "attn dqkv (score budget)")
    # same numbers as the all-resident path with its softmax as a separate pass (same kernels, same order inside a group)
    monkeypatch.setattr(ops, "ATTN_FOLDED_SOFTMAX", False)
    monkeypatch.setattr(ops, "ATTN_SCORE_BUDGET", 1 << 40)
    q2 = qkv.to(dev()).requires_grad_(True)
    o2 = ops.attention_qkv(q2)
    o2.backward(go.to(dev()))
    assert torch.equal(o2, o) and torch.equal(q2.grad, qd.grad)


# ------------------------------------------------------------------------------------------------------
def test_rescale_minmax(hip_lib):
    from odvae_amd import ops
    x = torch.rand(3, 3, 20, 12, generator=torch.Generator().manual_seed(3))
    ref = 2.0 * (x - x.min()) / (x.max() - x.min()) - 1.0
    y = ops.rescale_minmax(x.to(dev()))
    assert y.shape == x.shape and y.stride() == (20 * 12 * 3, 1, 12 * 3, 3)
    close(y, ref, 1e-6, "rescale")


def test_gaussian(hip_lib):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(11)
    mom = torch.randn(3, 8, 4, 4, generator=g) * 3
    mom[0, 5, 0, 0] = 25.0   # beyond the clamp
    mom[1, 6, 1, 1] = -40.0
    eps = torch.randn(3, 4, 4, 4, generator=g)
    mr = mom.clone().requires_grad_(True)
    mean, logvar = torch.chunk(mr, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    z_ref = mean + torch.exp(0.5 * logvar) * eps
    kl_ref = 0.5 * torch.sum(mean ** 2 + torch.exp(logvar) - 1.0 - logvar, dim=[1, 2, 3])
    gz = torch.randn(z_ref.shape, generator=g)
    gk = torch.randn(3, generator=g)
    (z_ref * gz).sum().backward(retain_graph=True)
    g1 = mr.grad.clone(); mr.grad = None
    (kl_ref * gk).sum().backward()
    g2 = mr.grad.clone()
    md = mom.to(dev()).requires_grad_(True)
    z = ops.gaussian_sample(md, eps.to(dev()))
    close(z, z_ref, 1e-5, "sample")
    z.backward(gz.to(dev()))
    close(md.grad, g1, 1e-5, "sample bwd")
    md.grad = None
    kl = ops.gaussian_kl(md)
    close(kl, kl_ref, 1e-5, "kl")
    kl.backward(gk.to(dev()))
    close(md.grad, g2, 1e-5, "kl bwd")


def test_l1_masked(hip_lib):
    from odvae_amd import ops
    g = torch.Generator().manual_seed(13)
    x = torch.randn(3, 3, 16, 8, generator=g)
    xr = torch.randn(3, 3, 16, 8, generator=g).requires_grad_(True)
    m = (torch.rand(3, 1, 16, 8, generator=g) > 0.3).float()
    ref = torch.abs(x * m - xr * m).sum(dim=[1, 2, 3])
    gw = torch.randn(3, generator=g)
    (ref * gw).sum().backward()
    xd = x.to(dev())
    xrd = xr.detach().to(dev()).requires_grad_(True)
    out = ops.l1_masked_sum(xd, xrd, m.to(dev()))
    close(out, ref, 1e-5, "l1")
    (out * gw.to(dev())).sum().backward()
    close(xrd.grad, xr.grad, 1e-6, "l1 bwd")
    out2 = ops.l1_masked_sum(xd, xrd, None)
    close(out2, torch.abs(x - xr).sum(dim=[1, 2, 3]), 1e-5, "l1 nomask")


def test_adam_and_norm(hip_lib):
    from odvae_amd import lib
    L = hip_lib
    g = torch.Generator().manual_seed(17)
    n = 100003
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3, betas=(0.5, 0.9))
    p = p0.to(dev()); m = torch.zeros_like(p); v = torch.zeros_like(p)
    out = torch.empty(2, device=dev())
    ws = torch.empty(4096, device=dev())
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * 0.01 * step
        ref_p.grad = grad.clone()
        total = torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        gd = grad.to(dev())
        lib.check(L.odvae_grad_norm_f32(gd.data_ptr(), n, 1.0, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, lib.stream_ptr()), "norm")
        assert abs(out[0].item() - total.item()) <= 1e-5 * total.item()
        lib.check(L.odvae_adam_step_f32(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.5, 0.9, 1e-8,
                                        step, out.data_ptr(), lib.stream_ptr()), "adam")
        close(p, ref_p, 1e-6, "adam step %d" % step)


def test_fails_loudly_on_cpu_tensor(hip_lib):
    from odvae_amd import ops, lib
    with pytest.raises(lib.HipLibraryError):
        ops.group_norm(torch.randn(1, 32, 4, 4), torch.ones(32), torch.zeros(32), 32, 1e-6, True)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 8, 16), (1, 128, 128, 18, 34), (2, 256, 128, 16, 16), (8, 128, 128, 64, 64)])
def test_upsample_conv_on_the_f4x4_kernel(hip_lib, monkeypatch, n, cin, cout, h, w):
    """[UPSTREAM] Upsample.forward (nearest 2x, then conv3x3) with the forward and the data gradient on the Winograd F(4x4,3x3) kernel
    (`odvae_conv3x3_wino4_up_f32`: the halo of the never-formed upsampled image is read from x[iy >> 1][ix >> 1]) against torch on the
    host and against the parity-class kernels (modes 5 / 6); the output's GroupNorm statistics come from the same epilogue.
    Shapes: one tile, ragged tiles in both directions, Cin != Cout, the persistent form."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(n + cin + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    b = torch.randn(cout, generator=g)
    gy = torch.randn(n, cout, 2 * h, 2 * w, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, br, padding=1)
    y_ref.backward(gy)
    outs = {}
    for f4 in (True, "two-step", False):     # "two-step": full-resolution data gradient + odvae_upsample2x_bwd_f32 instead of the pooled epilogue
        monkeypatch.setattr(ops, "UPCONV_WINOGRAD4", bool(f4))
        monkeypatch.setattr(ops, "UPCONV_POOLED_DGRAD", f4 is True)
        xd, wd, bd = x.to(dev()).requires_grad_(True), wt.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
        y = ops.conv3x3(xd, wd, bd, None, mode=2, gn_stats=True)
        part = ops._gn_partials_of(y, 32)
        assert (part is not None) == (bool(f4) and cout % 32 == 0)
        if part is not None:
            yc = y.detach().double().cpu().reshape(n, 32, cout // 32, 4 * h * w)
            want = torch.stack([yc.sum(dim=(2, 3)), (yc * yc).sum(dim=(2, 3))], dim=-1)
            assert (part.double().cpu().sum(dim=1) - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        y.backward(gy.to(dev()))
        outs[f4] = (y.detach(), xd.grad, wd.grad, bd.grad)
    for (a, c, ref, what) in zip(outs[True], outs[False], (y_ref, xr.grad, wr.grad, br.grad), ("y", "dx", "dw", "db")):
        close(a, ref, 5e-4 if what in ("y", "dx") else BWD_TOL * 4, "upsample conv on F(4x4): " + what)
        assert (a - c).abs().max().item() <= 6e-5 * max(1.0, c.abs().max().item()), what + " vs the parity-class kernels"
    assert (outs[True][1] - outs["two-step"][1]).abs().max().item() <= 2e-6 * max(1.0, outs["two-step"][1].abs().max().item()), "pooled vs two-step dx"


@pytest.mark.parametrize("n,cout,h,w", [(2, 3, 16, 32), (1, 3, 20, 36), (3, 2, 8, 8), (2, 3, 64, 96), (1, 1, 9, 33)])
def test_thin_output_conv(hip_lib, n, cout, h, w):
    """decoder.conv_out (128 -> 3, [UPSTREAM] Decoder.forward) on `conv3x3_thin_out_kernel`: the nine taps of the output channels as one
    32-wide MFMA operand over the tile and its halo ring, then a nine-term gather -- forward against torch on the host (whole and
    ragged tiles, image borders inside a tile, 1-3 output channels); the backward of this layer runs the thin-side kernels tested above."""
    from odvae_amd import ops
    g = torch.Generator().manual_seed(n + cout + h)
    x = torch.randn(n, 128, h, w, generator=g)
    wt = torch.randn(cout, 128, 3, 3, generator=g) / math.sqrt(9 * 128)
    b = torch.randn(cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, padding=1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd, wd, bd = x.to(dev()).requires_grad_(True), wt.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    y = ops.conv3x3(xd, wd, bd)
    close(y, y_ref, FWD_TOL, "thin-output conv forward")
    y.backward(gy.to(dev()))
    close(xd.grad, xr.grad, BWD_TOL, "dx")
    close(wd.grad, wr.grad, BWD_TOL * 4, "dw")
    close(bd.grad, br.grad, BWD_TOL * 4, "db")
    # without a bias
    close(ops.conv3x3(xd.detach(), wd.detach(), None), F.conv2d(x, wt, None, padding=1), FWD_TOL, "thin-output conv, no bias")


@pytest.mark.parametrize("kind,shapes", [("wino4", [(64, 64), (128, 64), (64, 128), (256, 256)]), ("wino", [(64, 64), (128, 128), (128, 64)])])
def test_batched_weight_packs_equal_the_single_launches(hip_lib, monkeypatch, kind, shapes):
    """After a weight update the pack cache refills every stale pack of a kind with ONE launch (`odvae_*_pack_*_batch`, a device table of
    weight / pack pointers): the packs it leaves are bit-identical to what the per-weight launches write, forward and data-gradient
    side, in place (same buffers as before the update), including weights whose data-gradient pack was never asked for."""
    from odvae_amd import ops
    monkeypatch.setattr(ops, "PACK_CACHE", ops._PackCache())
    g = torch.Generator().manual_seed(len(shapes))
    ws = [torch.randn(s[0], s[1], *( (s[2], s[2]) if len(s) > 2 else (3, 3)), generator=g).to(dev()) for s in shapes]
    first = [ops.pack_conv3x3(w, True, i != 1, kind) for i, w in enumerate(ws)]      # weight 1: forward pack only
    ptrs = [(f.data_ptr(), None if d is None else d.data_ptr()) for f, d in first]
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5).add_(0.25)
    ops.PACK_CACHE.bump()
    got = [ops.pack_conv3x3(w, True, i != 1, kind) for i, w in enumerate(ws)]
    assert len(ops.PACK_CACHE._tables) == 1                      # the batched path ran
    for i, w in enumerate(ws):
        f, d = ops._pack_conv3x3_now(w, True, i != 1, kind)
        assert (got[i][0].data_ptr(), None if got[i][1] is None else got[i][1].data_ptr()) == ptrs[i]
        assert torch.equal(got[i][0], f)
        assert (d is None and got[i][1] is None) or torch.equal(got[i][1], d)

"""GroupNorm backward, read-once form (csrc/groupnorm.hip `gn_bwd_fused_kernel`, csrc/bf16_ops.hip `gnb_bwd_fused_kernel`): the blocks of a
sample keep their share of x / dy in registers between the sums and the apply phase and meet at a per-(sample, channel slab) barrier in
L2.  Checked through the C ABI against torch on the host, against the two-kernel form (`odvae_groupnorm_select_backward(0)`), for
bit-identical repeats (the team sums run in a fixed order), and -- the property that matters most for a kernel with an in-launch
barrier -- that no wait ever gave up (`odvae_groupnorm_fused_timeouts() == 0`).  [UPSTREAM] Normalize / nonlinearity via
src/modules/autoencodermodules/feat_encoder.py:2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"

# (n, c, h, w): one block per item (T = 1); five members with a ragged last one; 16 channels per group; 2 channels per group;
# 32 members; one channel per group; several items per team (the refill path); 128 members = the 256 x 256 level's geometry
SHAPES = [(2, 128, 16, 16), (3, 256, 36, 68), (1, 512, 32, 32), (2, 64, 20, 36), (4, 128, 128, 128), (2, 32, 8, 8), (32, 128, 64, 64),
          (2, 128, 256, 256)]


def _raw_bwd(L, x, dy, gamma, beta, mean, rstd, swish, skip, groups=32):
    """x, dy, skip: [N][H][W][C] in memory (the C ABI's NHWC)."""
    from odvae_amd import lib as _lib, ops
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    wp, wn = ops._ws(L.odvae_groupnorm_workspace_bytes(n, h * w, c, groups), x)
    _lib.check(L.odvae_groupnorm_bwd_f32(x.data_ptr(), dy.data_ptr(), n, h * w, c, groups, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), int(swish), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), _lib.ptr(skip), wp, wn,
                                         _lib.stream_ptr()), "groupnorm_bwd")
    return dx, dg, db


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("swish,with_skip", [(True, True), (True, False), (False, True)])
def test_read_once_backward_matches_torch_and_the_two_kernel_form(hip_lib, shape, swish, with_skip):
    L = hip_lib
    n, c, h, w = shape
    g = torch.Generator().manual_seed(n * 7 + c + h)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)       # raw C-ABI layout [N][HW][C]
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    dy = torch.randn(n, c, h, w, generator=g)
    sk = torch.randn(n, c, h, w, generator=g) if with_skip else None
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    xg = x.reshape(n, 32, -1).double()
    mean = xg.mean(dim=2).float()
    rstd = (1.0 / torch.sqrt(xg.var(dim=2, unbiased=False) + 1e-6)).float()
    xd, dyd, skd = nhwc(x), nhwc(dy), (nhwc(sk) if with_skip else None)
    args = (xd, dyd, gamma.to(DEV), beta.to(DEV), mean.to(DEV), rstd.to(DEV), swish, skd)
    prev = L.odvae_groupnorm_select_backward(1)          # the fused form or an error: the test must not pass on the fallback
    try:
        fused = _raw_bwd(L, *args)
        again = _raw_bwd(L, *args)
        L.odvae_groupnorm_select_backward(0)
        two = _raw_bwd(L, *args)
    finally:
        L.odvae_groupnorm_select_backward(prev)
    assert L.odvae_groupnorm_fused_timeouts() == 0
    for a, b in zip(fused, again):
        assert torch.equal(a, b)                         # fixed-order team sums: bit-identical repeats
    # torch on the host (f64 would be better still, f32 autograd is what the other GroupNorm tests use)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.group_norm(xr, 32, gr, br, eps=1e-6)
    if swish:
        y = F.silu(y)
    y.backward(dy)
    want_dx = xr.grad + (sk if with_skip else 0.0)
    for name, got, other, want, tol in (("dx", fused[0].permute(0, 3, 1, 2).cpu(), two[0].permute(0, 3, 1, 2).cpu(), want_dx, 5e-4),
                                        ("dgamma", fused[1].cpu(), two[1].cpu(), gr.grad, 2e-3), ("dbeta", fused[2].cpu(), two[2].cpu(), br.grad, 2e-3)):
        scale = max(1.0, want.abs().max().item())
        assert (got.double() - want.double()).abs().max().item() <= tol * scale, name
        assert (got.double() - other.double()).abs().max().item() <= 2e-5 * scale, name + " vs the two-kernel form"


def test_read_once_backward_refuses_shapes_it_cannot_hold(hip_lib):
    """C not a multiple of the 32-channel slab: mode 1 reports it, the default mode silently takes the two-kernel form."""
    from odvae_amd import ops
    L = hip_lib
    n, c, h, w = 2, 48, 8, 8
    x = torch.randn(n, h, w, c, device=DEV)
    dy = torch.randn(n, h, w, c, device=DEV)
    gamma, beta = torch.randn(c, device=DEV), torch.randn(c, device=DEV)
    mean, rstd = torch.zeros(n, 16, device=DEV), torch.ones(n, 16, device=DEV)
    dx, dg, db = torch.empty_like(x), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    wp, wn = ops._ws(L.odvae_groupnorm_workspace_bytes(n, h * w, c, 16), x)
    call = lambda: L.odvae_groupnorm_bwd_f32(x.data_ptr(), dy.data_ptr(), n, h * w, c, 16, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1,
                                             dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, wp, wn, torch.cuda.current_stream().cuda_stream)
    prev = L.odvae_groupnorm_select_backward(1)
    try:
        assert call() != 0 and b"fused" in L.odvae_last_error()
        L.odvae_groupnorm_select_backward(-1)
        assert call() == 0
    finally:
        L.odvae_groupnorm_select_backward(prev)
    torch.cuda.synchronize()
    assert torch.isfinite(dx).all()

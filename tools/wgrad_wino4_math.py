"""The arithmetic of DESIGN.md §7's next item -- the stride-1 3x3 weight gradient in the F(4x4,3x3) domain -- checked on the CPU before a
kernel exists (no GPU, no library: torch only; `python tools/wgrad_wino4_math.py`).

The weight gradient of one 4x4 output tile is itself a small correlation, dW[k] = sum_i dy[i] x[i + k] (k = 0..2, i = 0..3) per axis: three
outputs of a four-tap filter over six inputs, i.e. F(3, 4) on the SAME six interpolation points (0, +-1, +-2, inf) the forward's F(4, 3) uses.
Hence
    dW[co][ci] = Aw^T [ sum over tiles and images of (Gw dy_tile Gw^T) .* (B^T x_tile B) ] Aw
with B^T the forward kernel's own input transform (the 6x6 V tiles are the forward's, bit for bit), Gw (6x4) the Vandermonde of the points
with the forward G's row factors, and Aw^T (3x6) the first three rows of the forward's A^T with the point at infinity moved to the last kept
power.  36 products per (tile, co, ci) instead of 9 x 16 = 144: the 4x the sizing in DESIGN.md uses.

Prints (i) the exactness of the identity in f64, (ii) the f32 deviation of the transform-domain sum against the direct f32 sum on the step's
operand statistics (unit-variance x, dy with the 1e-4 scale of a mean-reduced loss), reduction lengths of the 256^2 and 64^2 levels."""
import torch
import torch.nn.functional as F

BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                   [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
POINTS = [0.0, 1.0, -1.0, 2.0, -2.0]
ROWF = [1 / 4, -1 / 6, -1 / 6, 1 / 24, 1 / 24]      # the forward G's row factors, 1 / prod_{j != i} (p_i - p_j)
GW = torch.zeros(6, 4, dtype=torch.float64)
for r, (p, f) in enumerate(zip(POINTS, ROWF)):
    GW[r] = torch.tensor([f * p ** k for k in range(4)], dtype=torch.float64)
GW[5, 3] = 1.0
AWT = torch.zeros(3, 6, dtype=torch.float64)
for c, p in enumerate(POINTS):
    AWT[:, c] = torch.tensor([p ** k for k in range(3)], dtype=torch.float64)
AWT[2, 5] = 1.0


def tiles_of(x, dy):
    """x [N, Ci, H, W], dy [N, Co, H, W] (H, W multiples of 4) -> x tiles [N, Ci, T, 6, 6] (1-pixel zero halo), dy tiles [N, Co, T, 4, 4]"""
    n, ci, h, w = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    xt = xp.unfold(2, 6, 4).unfold(3, 6, 4).reshape(n, ci, -1, 6, 6)
    dt = dy.unfold(2, 4, 4).unfold(3, 4, 4).reshape(n, dy.shape[1], -1, 4, 4)
    return xt, dt


def wgrad_wino4(x, dy, dtype):
    xt, dt = tiles_of(x.to(dtype), dy.to(dtype))
    bt, gw, awt = BT.to(dtype), GW.to(dtype), AWT.to(dtype)
    v = bt @ xt @ bt.T                                   # [N, Ci, T, 6, 6]
    u = gw @ dt @ gw.T                                   # [N, Co, T, 6, 6]
    m = torch.einsum("notab,nitab->oiab", u, v)          # the 36 [Co x Ci] products, reduced over tiles and images
    return awt @ m @ awt.T                               # [Co, Ci, 3, 3]


def wgrad_direct(x, dy, dtype):
    w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, dtype=dtype, requires_grad=True)
    F.conv2d(x.to(dtype), w, padding=1).backward(dy.to(dtype))
    return w.grad


def main():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 16, 24, generator=g, dtype=torch.float64)
    dy = torch.randn(2, 7, 16, 24, generator=g, dtype=torch.float64)
    ref = wgrad_direct(x, dy, torch.float64)
    e = (wgrad_wino4(x, dy, torch.float64) - ref).abs().max().item() / ref.abs().max().item()
    print("identity in f64: rel dev %.2e" % e)
    assert e < 1e-12
    for (n, h) in ((2, 256), (2, 64), (32, 64)):
        x = torch.randn(n, 8, h, h, generator=g, dtype=torch.float64)
        dy = torch.randn(n, 8, h, h, generator=g, dtype=torch.float64) * 1e-4
        ref = wgrad_direct(x, dy, torch.float64)
        s = ref.abs().max().item()
        ew = (wgrad_wino4(x, dy, torch.float32).double() - ref).abs().max().item() / s
        ed = (wgrad_direct(x, dy, torch.float32).double() - ref).abs().max().item() / s
        print("f32, N=%d %dx%d (reduction over %d tiles): transform domain %.2e, direct %.2e" % (n, h, h, n * (h // 4) ** 2, ew, ed))


if __name__ == "__main__":
    main()

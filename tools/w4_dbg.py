import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
b, cin, cout, h, wd = 8, 128, 128, 128, 128
x = torch.randn(b, h, wd, cin, device=dev).permute(0, 3, 1, 2)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
with torch.no_grad():
    ops.WINOGRAD4 = True
    y = ops.conv3x3(x, w, None)
    ops.WINOGRAD4 = False
    r = ops.conv3x3(x, w, None)
d = (y - r).abs()
bad = ~torch.isfinite(y) | (d > 1e-3 * r.abs().max())
print("bad fraction", bad.float().mean().item(), "nan", (~torch.isfinite(y)).float().mean().item())
# per 16x32 tile and 64-channel block
bt = bad.view(b, 2, 64, 8, 16, 4, 32).float().mean(dim=(2, 4, 6))    # [n][coblk][ty][tx]
for n in range(b):
    for cb in range(2):
        print("n%d cb%d" % (n, cb), ["".join("X" if v > 0.5 else ("x" if v > 0 else ".") for v in row) for row in bt[n, cb].tolist()])
# which channels / pixels inside a bad tile
idx = bad.nonzero()
if len(idx):
    print("first bad", idx[0].tolist(), "y", y[tuple(idx[0].tolist())].item(), "ref", r[tuple(idx[0].tolist())].item())
    n0, c0, y0, x0 = idx[0].tolist()
    ty, tx = y0 // 16, x0 // 32
    tile = bad[n0, :, ty * 16:(ty + 1) * 16, tx * 32:(tx + 1) * 32]
    print("bad per channel (first 64):", tile[:64].float().mean(dim=(1, 2)).tolist()[:16])
    print("bad per row:", tile.float().mean(dim=(0, 2)).tolist())
    print("bad per col:", [round(v, 2) for v in tile.float().mean(dim=(0, 1)).tolist()])

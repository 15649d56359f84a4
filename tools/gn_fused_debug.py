import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from odvae_amd import lib as _lib
from test_groupnorm_fused_gpu import _raw_bwd
L = _lib.load()
DEV = "cuda:0"
for (n, c, h, w) in [(2, 128, 16, 16), (3, 256, 36, 68), (2, 32, 8, 8), (32, 128, 64, 64)]:
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(n, h, w, c, generator=g) * 2 + 0.5).to(DEV)
    dy = torch.randn(n, h, w, c, generator=g).to(DEV)
    sk = torch.randn(n, h, w, c, generator=g).to(DEV)
    gamma, beta = torch.randn(c, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
    xg = x.float().reshape(n, h * w, 32, c // 32)
    mean = xg.mean(dim=(1, 3)).contiguous(); rstd = (1.0 / torch.sqrt(xg.var(dim=(1, 3), unbiased=False) + 1e-6)).contiguous()
    L.odvae_groupnorm_select_backward(1)
    outs = [_raw_bwd(L, x, dy, gamma, beta, mean, rstd, True, sk) for _ in range(4)]
    L.odvae_groupnorm_select_backward(0)
    two = _raw_bwd(L, x, dy, gamma, beta, mean, rstd, True, sk)
    L.odvae_groupnorm_select_backward(-1)
    torch.cuda.synchronize()
    for k, name in enumerate(("dx", "dgamma", "dbeta")):
        d = [(outs[i][k] - outs[0][k]).abs().max().item() for i in range(1, 4)]
        nd = [(outs[i][k] != outs[0][k]).sum().item() for i in range(1, 4)]
        print((n, c, h, w), name, "repeat max diff", d, "differing", nd, "| vs two-kernel", (outs[0][k] - two[k]).abs().max().item(), "scale", two[k].abs().max().item())
print("timeouts", L.odvae_groupnorm_fused_timeouts())

# where are the wrong elements?
n, c, h, w = 2, 128, 16, 16
g = torch.Generator().manual_seed(1)
x = (torch.randn(n, h, w, c, generator=g) * 2 + 0.5).to(DEV)
dy = torch.randn(n, h, w, c, generator=g).to(DEV)
sk = torch.randn(n, h, w, c, generator=g).to(DEV)
gamma, beta = torch.randn(c, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
xg = x.float().reshape(n, h * w, 32, c // 32)
mean = xg.mean(dim=(1, 3)).contiguous(); rstd = (1.0 / torch.sqrt(xg.var(dim=(1, 3), unbiased=False) + 1e-6)).contiguous()
L.odvae_groupnorm_select_backward(0)
two = _raw_bwd(L, x, dy, gamma, beta, mean, rstd, True, sk)[0].reshape(n, h * w, c)
two_ns = _raw_bwd(L, x, dy, gamma, beta, mean, rstd, True, None)[0].reshape(n, h * w, c)
for rep in range(3):
    L.odvae_groupnorm_select_backward(1)
    f = _raw_bwd(L, x, dy, gamma, beta, mean, rstd, True, sk)[0].reshape(n, h * w, c)
    bad = ((f - two).abs() > 1e-3).nonzero()
    print("rep", rep, "bad", bad.shape[0], "samples", sorted(set(bad[:, 0].tolist())), "pixels", sorted(set(bad[:, 1].tolist()))[:40], "channels", sorted(set(bad[:, 2].tolist())))
    if bad.shape[0]:
        i = bad[0]
        print("   first bad", i.tolist(), "got", f[i[0], i[1], i[2]].item(), "want", two[i[0], i[1], i[2]].item(), "without skip", two_ns[i[0], i[1], i[2]].item(), "skip", sk.reshape(n, h * w, c)[i[0], i[1], i[2]].item())
L.odvae_groupnorm_select_backward(-1)
skf = sk.reshape(n, h * w, c)
d = (f - two_ns)          # what the kernel added as "skip"
print("added at [0,33,17]:", d[0, 33, 17].item(), "| skip[0, 33, 16..19]", skf[0, 33, 16:20].tolist(), "| skip[0, 65, 16..19]", skf[0, 65, 16:20].tolist(), "| skip[0,1,16..19]", skf[0, 1, 16:20].tolist())
# search the skip tensor for the value the kernel added
hit = ((skf - d[0, 33, 17]).abs() < 1e-6).nonzero()
print("that value sits at", hit.tolist()[:8])
hit = ((skf - d[0, 35, 21]).abs() < 1e-6).nonzero()
print("[0,35,21] added", d[0, 35, 21].item(), "sits at", hit.tolist()[:8])
# no-skip run: is the GroupNorm term itself wrong?
L.odvae_groupnorm_select_backward(1)
f_ns = _raw_bwd(L, x, dy, gamma, beta, mean, rstd, True, None)[0].reshape(n, h * w, c)
L.odvae_groupnorm_select_backward(-1)
bad_ns = ((f_ns - two_ns).abs() > 1e-3).nonzero()
print("without a skip tensor: bad", bad_ns.shape[0])
# host recomputation of the term for every (pixel, channel) with the constants of channel 17, sample 0
xs = x.reshape(n, h * w, c)[0].double(); dys = dy.reshape(n, h * w, c)[0].double()
G = 32; cpg = c // G
mu = mean[0].double().repeat_interleave(cpg); rs = rstd[0].double().repeat_interleave(cpg)
xh = (xs - mu) * rs
u = xh * gamma.double() + beta.double()
sg = torch.sigmoid(u)
du = dys * sg * (1 + u * (1 - sg))
A = (du * xh).sum(0); B = du.sum(0)
ds1 = (A * gamma.double()).reshape(G, cpg).sum(1).repeat_interleave(cpg) / (h * w * cpg)
ds2 = (B * gamma.double()).reshape(G, cpg).sum(1).repeat_interleave(cpg) / (h * w * cpg)
term = rs * (du * gamma.double() - (ds2 + xh * ds1))
print("host term at [33,17]", term[33, 17].item(), "two-kernel", two_ns[0, 33, 17].item(), "fused(no skip)", f_ns[0, 33, 17].item(), "fused(skip) - skip", (f[0, 33, 17] - skf[0, 33, 17]).item())
cc = 17
cand = rs[cc] * (du * gamma.double()[cc] - (ds2[cc] + xh * ds1[cc]))      # other (pixel, channel)'s xhat / du with channel 17's constants
target = (f[0, 33, 17] - skf[0, 33, 17]).double()
hit = ((cand - target).abs() < 1e-5).nonzero()
print("fused(skip) - skip matches xhat/du of", hit.tolist()[:6])
target2 = f[0, 33, 17].double()
for name, t in (("skip of same pixel other comps", skf[0, 33].double()), ("skip px 65", skf[0, 65].double()), ("skip px 1", skf[0, 1].double()), ("skip px 97", skf[0, 97].double())):
    h2 = ((term[33, 17] + t - target2).abs() < 1e-5).nonzero()
    print("  true term +", name, "->", h2.tolist()[:6])

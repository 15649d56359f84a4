"""Does an HBM-bound GroupNorm backward hide under an MFMA-bound weight-gradient kernel when the two run on different streams?
128 ch @256x256, B=32 (1 GiB tensors): serial vs two-stream wall time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops

dev = "cuda:0"
B, C, H = 32, 128, 256
x = torch.randn(B, H, H, C, device=dev).permute(0, 3, 1, 2)
dy = torch.randn(B, H, H, C, device=dev).permute(0, 3, 1, 2)
w = (torch.randn(C, C, 3, 3, device=dev) * 0.03).requires_grad_(True)
gx = torch.randn(B, H, H, C, device=dev).permute(0, 3, 1, 2).requires_grad_(True)
gamma = torch.ones(C, device=dev, requires_grad=True)
beta = torch.zeros(C, device=dev, requires_grad=True)
gdy = torch.randn(B, H, H, C, device=dev).permute(0, 3, 1, 2)

y = ops.conv3x3(x, w)              # graph for the weight gradient only (x needs no gradient)
gy = ops.group_norm(gx, gamma, beta, 32, 1e-6, True)


def wgrad():
    torch.autograd.grad(y, w, dy, retain_graph=True)


def gn_bwd():
    torch.autograd.grad(gy, (gx, gamma, beta), gdy, retain_graph=True)


def dgrad():
    with torch.no_grad():
        ops.conv3x3(dy, w)


def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_streams(a, b):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): a()
        with torch.cuda.stream(s2): b()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run


print("wgrad alone          %.3f ms" % timeit(wgrad))
print("gn backward alone    %.3f ms" % timeit(gn_bwd))
print("dgrad (conv) alone   %.3f ms" % timeit(dgrad))
print("wgrad ; gn_bwd       %.3f ms (serial)" % timeit(lambda: (wgrad(), gn_bwd())))
print("wgrad || gn_bwd      %.3f ms (two streams)" % timeit(two_streams(wgrad, gn_bwd)))
print("dgrad ; gn_bwd       %.3f ms (serial)" % timeit(lambda: (dgrad(), gn_bwd())))
print("dgrad || gn_bwd      %.3f ms (two streams)" % timeit(two_streams(dgrad, gn_bwd)))
print("wgrad ; dgrad        %.3f ms (serial)" % timeit(lambda: (wgrad(), dgrad())))
print("wgrad || dgrad       %.3f ms (two streams)" % timeit(two_streams(wgrad, dgrad)))

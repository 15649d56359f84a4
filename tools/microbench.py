"""GPU micro-benchmarks of the dominant kernels at the config-2 layer shapes (B per-GPU images, 256x256).
Prints achieved TFLOP/s (MFMA-bound kernels) or GB/s (HBM-bound) per case.  Not part of the test-suite."""
import sys
import os
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    L = lib.load()
    print("B =", B)
    for (mode, cin, cout, h) in [(0, 128, 128, 256), (0, 128, 128, 128), (0, 256, 256, 64), (0, 512, 512, 16), (0, 256, 128, 128),
                                 (1, 128, 128, 256), (2, 128, 128, 128), (0, 128, 3, 256), (0, 3, 128, 256)]:
        x = torch.randn(B, h, h, cin, device=dev).permute(0, 3, 1, 2)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        pack, dpack = ops.pack_conv3x3(w, True, True)
        ho = h if mode == 0 else (h // 2 if mode == 1 else 2 * h)
        flops = 2.0 * 9 * cin * cout * B * ho * ho
        t = timeit(lambda: ops._conv3x3_raw(mode, x, pack, cin, cout, b, None))
        print("conv3x3 fwd  mode%d %4d->%4d @%3d: %8.3f ms  %7.1f TFLOP/s" % (mode, cin, cout, h, t, flops / t / 1e9))
        dy = torch.randn(B, ho, ho, cout, device=dev).permute(0, 3, 1, 2)
        dw = torch.empty_like(w); db = torch.empty_like(b)
        need = L.odvae_conv3x3_wgrad_workspace_bytes(mode, B, ho, ho, cin, cout)
        wp, wn = lib.workspace.get(need, dev)
        def wg():
            lib.check(L.odvae_conv3x3_wgrad_f32(mode, x.data_ptr(), dy.data_ptr(), B, h, h, cin, ho, ho, cout, dw.data_ptr(), db.data_ptr(), wp, wn, lib.stream_ptr()), "wgrad")
        t = timeit(wg)
        print("conv3x3 wgrad mode%d %4d->%4d @%3d: %8.3f ms  %7.1f TFLOP/s" % (mode, cin, cout, h, t, flops / t / 1e9))
    for (m, n, k, batch, ta, tb) in [(B * 65536 // 4, 256, 128, 1, 0, 1), (4096, 4096, 256, B, 0, 1), (4096, 256, 4096, B, 0, 0), (4096, 256, 4096, B, 1, 0), (256, 256, B * 4096, 1, 1, 0)]:
        a = torch.randn(batch, (k if ta else m), (m if ta else k), device=dev)
        bm = torch.randn(batch, (n if tb else k), (k if tb else n), device=dev)
        c = torch.empty(batch, m, n, device=dev)
        t = timeit(lambda: ops.gemm(ta, tb, m, n, k, 1.0, a, a.shape[2], a.shape[1] * a.shape[2], bm, bm.shape[2], bm.shape[1] * bm.shape[2], c, n, m * n, batch=batch))
        print("gemm ta%d tb%d M%6d N%5d K%6d b%2d: %8.3f ms  %7.1f TFLOP/s" % (ta, tb, m, n, k, batch, t, 2.0 * m * n * k * batch / t / 1e9))
    for (c, h) in [(128, 256), (256, 64), (512, 16)]:
        x = torch.randn(B, h, h, c, device=dev).permute(0, 3, 1, 2)
        g = torch.randn(c, device=dev); bt = torch.randn(c, device=dev)
        t = timeit(lambda: ops.group_norm(x, g, bt, 32, 1e-6, True))
        byts = x.numel() * 4 * 3
        print("groupnorm+swish fwd C%4d @%3d: %8.3f ms  %7.1f GB/s (3 passes)" % (c, h, t, byts / t / 1e6))
        xg = x.detach().requires_grad_(True)
        y = ops.group_norm(xg, g, bt, 32, 1e-6, True)
        dy = torch.randn_like(y)
        t = timeit(lambda: torch.autograd.grad(y, xg, dy, retain_graph=True))
        print("groupnorm+swish bwd C%4d @%3d: %8.3f ms  %7.1f GB/s (5 passes)" % (c, h, t, x.numel() * 4 * 5 / t / 1e6))


if __name__ == "__main__":
    main()

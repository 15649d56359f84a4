"""GEMM timings at the attention / 1x1 shapes of the step (B = 32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib as _lib
if os.environ.get("ODVAE_PROBE_LIB"):      # A/B builds of the library (tools/bin/, not shipped)
    _lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]

def timeit(fn, iters=30):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dev = torch.device("cuda:0"); B = 32
cases = [("S=QK^T   NT", 4096, 4096, 256, B, 0, 1), ("O=PV     NN", 4096, 256, 4096, B, 0, 0), ("dV=P^TdO TN", 4096, 256, 4096, B, 1, 0),
         ("dP=dOV^T NT", 4096, 4096, 256, B, 0, 1), ("1x1 fwd qkv NT", B * 4096, 768, 256, 1, 0, 1), ("1x1 dgrad NN", B * 4096, 256, 768, 1, 0, 0),
         ("1x1 wgrad TN", 768, 256, B * 4096, 1, 1, 0), ("nin 128->256 @128", B * 16384, 256, 128, 1, 0, 1)]
for name, m, n, k, batch, ta, tb in cases:
    a = torch.randn(batch, (k if ta else m), (m if ta else k), device=dev)
    b = torch.randn(batch, (n if tb else k), (k if tb else n), device=dev)
    c = torch.empty(batch, m, n, device=dev)
    t = timeit(lambda: ops.gemm(ta, tb, m, n, k, 1.0, a, a.shape[2], a.shape[1] * a.shape[2], b, b.shape[2], b.shape[1] * b.shape[2], c, n, m * n, batch=batch))
    print("%-18s M%7d N%5d K%7d b%2d: %8.3f ms  %6.1f TFLOP/s" % (name, m, n, k, batch, t, 2.0 * m * n * k * batch / t / 1e9))

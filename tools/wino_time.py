"""Time the default 3x3 conv forward on two layer shapes (used for rocprofv3 --pmc passes and ablation builds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib as _lib
if os.environ.get("ODVAE_PROBE_LIB"):      # A/B builds of the library (tools/bin/, not shipped)
    _lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]
dev = "cuda:0"
# correctness first (A/B builds are not covered by the test-suite): odd tile counts, a residual, several chunk counts
for (b, cin, cout, h, wd) in [] if os.environ.get("WINO_NOCHECK") else [(1, 16, 128, 8, 16), (2, 32, 128, 10, 18), (1, 48, 128, 16, 16), (2, 128, 128, 64, 48), (1, 64, 256, 24, 40), (3, 256, 128, 16, 16), (1, 512, 512, 32, 32),
                               (4, 128, 128, 128, 128), (2, 64, 256, 128, 128), (5, 128, 128, 112, 160), (3, 256, 512, 64, 96)]:   # the last four: several tiles per block (persistent form)
    x = torch.randn(b, h, wd, cin, device=dev).permute(0, 3, 1, 2)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    res = torch.randn(b, h, wd, cout, device=dev).permute(0, 3, 1, 2)
    with torch.no_grad():
        y = ops.conv3x3(x, w, bias, residual=res)
        ref = (torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), bias.double().cpu(), padding=1) + res.double().cpu())
    err = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print("check %dx%dx%dx%d->%d: max err / max|y| = %.2e %s" % (b, cin, h, wd, cout, err, "ok" if err < 2e-5 else "WRONG"), flush=True)
    assert err < 2e-5
for (b, cin, cout, h) in ([(32,128,128,256)] if os.environ.get("WINO_ONLY_BIG") else [(32,128,128,256),(32,256,256,64)]):
    x = torch.randn(b, h, h, cin, device=dev).permute(0,3,1,2)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    with torch.no_grad():
        for _ in range(60): y = ops.conv3x3(x, w, bias)     # clock ramp: the first launches of a process run 5-15 % slower
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): y = ops.conv3x3(x, w, bias)
        e1.record(); torch.cuda.synchronize()
    print("B%d %d->%d @%d: %.3f ms" % (b, cin, cout, h, e0.elapsed_time(e1) / 100), flush=True)


"""Time the default 3x3 conv forward on two layer shapes (used for rocprofv3 --pmc passes and ablation builds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib as _lib
if os.environ.get("ODVAE_PROBE_LIB"):      # A/B builds of the library (tools/bin/, not shipped)
    _lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]
dev = "cuda:0"
for (b, cin, cout, h) in [(32,128,128,256),(32,256,256,64)]:
    x = torch.randn(b, h, h, cin, device=dev).permute(0,3,1,2)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    with torch.no_grad():
        y = ops.conv3x3(x, w, bias); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = ops.conv3x3(x, w, bias)
        e1.record(); torch.cuda.synchronize()
    print("B%d %d->%d @%d: %.3f ms" % (b, cin, cout, h, e0.elapsed_time(e1) / 10), flush=True)

"""Winograd vs direct 3x3 conv forward at the step's layer shapes: time and max relative deviation (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops
dev = "cuda:0"
for (b, cin, cout, h) in [(32,128,128,256),(32,128,128,128),(32,256,256,64),(32,256,256,32),(32,512,512,16)]:
    x = torch.randn(b, h, h, cin, device=dev).permute(0,3,1,2)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    for mode in ("wino", "dense"):
        ops.WINOGRAD = mode == "wino"
        with torch.no_grad():
            y = ops.conv3x3(x, w, bias); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): y = ops.conv3x3(x, w, bias)
            e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        if mode == "wino": yw = y
        print("%s B%d %d->%d @%d: %.3f ms  %.1f TFLOP/s (direct-equivalent)" % (mode, b, cin, cout, h, t, 2.0*9*cin*cout*b*h*h/t/1e9), flush=True)
    print("   max |wino - dense| / max|dense| = %.2e" % ((yw - y).abs().max().item() / y.abs().max().item()))

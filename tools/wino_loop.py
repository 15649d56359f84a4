"""Run the default 3x3 conv forward (128->128 @256^2, B=32) back to back for a few seconds (tools/clock_watch.sh samples the clocks meanwhile)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib as _lib
if os.environ.get("ODVAE_PROBE_LIB"):
    _lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]
dev = "cuda:0"
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
x = torch.randn(32, 256, 256, 128, device=dev).permute(0, 3, 1, 2)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.05
bias = torch.randn(128, device=dev)
with torch.no_grad():
    y = ops.conv3x3(x, w, bias); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): y = ops.conv3x3(x, w, bias)
        e1.record(); torch.cuda.synchronize(); n += 1
        print("%.3f ms" % (e0.elapsed_time(e1) / 100), flush=True)

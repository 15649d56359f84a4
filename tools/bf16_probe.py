"""Launch one bf16 kernel family a few times (for rocprofv3 --pmc / --kernel-trace runs and quick timings).
usage: python tools/bf16_probe.py flash_fwd|flash_bwd|conv|wgrad [N] [C] [H] [iters]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib as _lib  # noqa: E402
if os.environ.get("ODVAE_PROBE_LIB"):      # A/B builds of the library (tools/bin/, not shipped)
    _lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]

which = sys.argv[1] if len(sys.argv) > 1 else "flash_fwd"
N, C, H = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (8, 256, 64)))
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dev = torch.device("cuda:0")
BF = torch.bfloat16
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def nhwc(n, c, h, w):
    return torch.randn(n, h, w, c, device=dev).to(BF).permute(0, 3, 1, 2)


if which.startswith("flash"):
    qkv = nhwc(N, 3 * C, H, H).requires_grad_(which == "flash_bwd")
    do = nhwc(N, C, H, H)
    flops = 4.0 * (H * H) ** 2 * C * N * (1 if which == "flash_fwd" else 2.5)

    def run():
        o = ops.attention_qkv(qkv)
        if which == "flash_bwd":
            o.backward(do)
            qkv.grad = None
else:
    x = nhwc(N, C, H, H).requires_grad_(True)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.03).requires_grad_(which == "wgrad")
    b = torch.zeros(C, device=dev)
    dy = nhwc(N, C, H, H)
    flops = 2.0 * 9 * C * C * N * H * H

    def run():
        if which == "conv":
            with torch.no_grad():
                ops.conv3x3(x, w, b, None, 0)
        else:
            x.requires_grad_(False)
            y = ops.conv3x3(x, w, b, None, 0)
            y.backward(dy)
            w.grad = None

for i in range(iters + 1):
    if i == 1:
        ev0.record()
    run()
ev1.record()
torch.cuda.synchronize()
t = ev0.elapsed_time(ev1) / iters
print("%s N%d C%d %dx%d: %.3f ms  %.1f TFLOP/s (algorithmic)" % (which, N, C, H, H, t, flops / t / 1e9))

"""GroupNorm + swish forward / backward timing on one tensor shape (f32 or bf16); also served the Infinity-Cache sub-batching
experiment recorded in DESIGN.md 7 (rejected).  NOTE: every iteration ends in a device synchronisation, so the BACKWARD figure brackets the host's launch
gaps of the autograd backward (four launches issued into an empty queue) as well as its kernels -- read per-kernel rates from a kernel trace or the PMC cycle
counts (profiles/r05_bf16_conv_pmc.txt), not from here.  usage: python tools/gn_probe.py [f32|bf16] [N C H iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib as _lib  # noqa: E402
if os.environ.get("ODVAE_PROBE_LIB"):      # A/B builds of the library (tools/bin/, not shipped)
    _lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
N, C, H = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (32, 128, 256)))
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 40
dev = torch.device("cuda:0")
x = torch.randn(N, H, H, C, device=dev).to(dt).permute(0, 3, 1, 2).requires_grad_(True)
g, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
dy = torch.randn(N, H, H, C, device=dev).to(dt).permute(0, 3, 1, 2)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for i in range(iters + 1):
    e[0].record()
    y, skip = ops.group_norm_skip(x, g, b, 32, 1e-6, True)
    e[1].record()
    torch.autograd.backward([y, skip], [dy, dy])
    e[2].record()
    torch.cuda.synchronize()
    x.grad = None
    if i:
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
nb = x.numel() * x.element_size()
print("GN+swish %s N%d C%d %dx%d (%.0f MB/tensor): fwd %.3f ms = %.2f TB/s of 2 algorithmic passes | bwd (incl. folded skip gradient) %.3f ms = %.2f TB/s of 4 passes"
      % (sys.argv[1] if len(sys.argv) > 1 else "f32", N, C, H, H, nb / 1e6, tf / iters, 2 * nb / (tf / iters) / 1e9, tb / iters, 4 * nb / (tb / iters) / 1e9))

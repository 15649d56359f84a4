"""Launch one conv configuration a few times (for rocprofv3 --pmc / --kernel-trace runs)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, cin, cout, h = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (32, 128, 128, 256)))
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
L = lib.load()
x = torch.randn(B, h, h, cin, device=dev).permute(0, 3, 1, 2)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
pack, _ = ops.pack_conv3x3(w, True, False)
dy = torch.randn(B, h, h, cout, device=dev).permute(0, 3, 1, 2)
dw = torch.empty_like(w); db = torch.empty_like(b)
need = L.odvae_conv3x3_wgrad_workspace_bytes(0, B, h, h, cin, cout)
wp, wn = lib.workspace.get(need, dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(iters + 1):
    if i == 1:
        ev0.record()
    if which == "fwd":
        ops._conv3x3_raw(0, x, pack, cin, cout, b, None)
    else:
        lib.check(L.odvae_conv3x3_wgrad_f32(0, x.data_ptr(), dy.data_ptr(), B, h, h, cin, h, h, cout, dw.data_ptr(), db.data_ptr(), wp, wn, lib.stream_ptr()), "wgrad")
ev1.record()
torch.cuda.synchronize()
t = ev0.elapsed_time(ev1) / iters
print("%s B%d %d->%d @%d: %.3f ms  %.1f TFLOP/s" % (which, B, cin, cout, h, t, 2.0 * 9 * cin * cout * B * h * h / t / 1e9))

"""Winograd-domain weight gradient vs the direct weight-gradient kernel (both through the C ABI) and vs torch CPU on small
shapes; timing on the step's layer shapes.  python tools/wgrad_wino_probe.py [check|time]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import lib as _lib, ops

dev = "cuda:0"
if os.environ.get("ODVAE_PROBE_LIB"):      # A/B builds of the library (tools/bin/, not shipped)
    _lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]
L = _lib.load()


def run(kind, x, dy, cin, cout):
    n, h, w = x.shape[0], x.shape[1], x.shape[2]
    dw = torch.empty(cout, cin, 3, 3, device=dev)
    db = torch.empty(cout, device=dev)
    if kind == "wino":
        need = L.odvae_conv3x3_wgrad_wino_workspace_bytes(n, h, w, cin, cout)
        wp, wn = ops._ws(need, x)
        _lib.check(L.odvae_conv3x3_wgrad_wino_f32(x.data_ptr(), dy.data_ptr(), n, h, w, cin, cout, dw.data_ptr(), db.data_ptr(),
                                                  wp, wn, _lib.stream_ptr()), "wgrad_wino")
    else:
        need = L.odvae_conv3x3_wgrad_workspace_bytes(0, n, h, w, cin, cout)
        wp, wn = ops._ws(need, x)
        _lib.check(L.odvae_conv3x3_wgrad_f32(0, x.data_ptr(), dy.data_ptr(), n, h, w, cin, h, w, cout, dw.data_ptr(), db.data_ptr(),
                                             wp, wn, _lib.stream_ptr()), "wgrad")
    return dw, db


def check():
    for (n, cin, cout, h, w) in [(2, 128, 128, 16, 16), (1, 128, 256, 8, 12), (3, 256, 128, 6, 10), (2, 128, 128, 2, 2), (1, 384, 128, 4, 34)]:
        g = torch.Generator().manual_seed(n * 1000 + cin + h)
        x = torch.randn(n, cin, h, w, generator=g)
        dy = torch.randn(n, cout, h, w, generator=g)
        wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
        b = torch.zeros(cout, requires_grad=True)
        torch.nn.functional.conv2d(x, wt, b, padding=1).backward(dy)
        xd = x.to(dev).permute(0, 2, 3, 1).contiguous()
        dyd = dy.to(dev).permute(0, 2, 3, 1).contiguous()
        assert L.odvae_conv3x3_wgrad_wino_supported(n, h, w, cin, cout) == 1
        dw, db = run("wino", xd, dyd, cin, cout)
        dw2, db2 = run("direct", xd, dyd, cin, cout)
        torch.cuda.synchronize()
        s = wt.grad.abs().max().item()
        e1 = (dw.cpu() - wt.grad).abs().max().item() / s
        e2 = (dw2.cpu() - wt.grad).abs().max().item() / s
        eb = (db.cpu() - b.grad).abs().max().item() / b.grad.abs().max().item()
        print("N%d %d->%d %dx%d: wino rel err %.2e (direct %.2e), bias %.2e" % (n, cin, cout, h, w, e1, e2, eb), flush=True)
        assert e1 < 2e-4 and eb < 2e-4


def time_():
    for (b, cin, cout, h) in [(32, 128, 128, 256), (32, 128, 128, 128), (32, 256, 256, 64), (32, 512, 512, 16), (32, 256, 512, 32)]:
        x = torch.randn(b, h, h, cin, device=dev)
        dy = torch.randn(b, h, h, cout, device=dev)
        for kind in ("direct", "wino"):
            for _ in range(30): run(kind, x, dy, cin, cout)      # clock ramp
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): run(kind, x, dy, cin, cout)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 40
            gf = 2.0 * 9 * cin * cout * b * h * h / 1e9
            print("B%d %d->%d @%d %-6s %.3f ms  %.1f TFLOP/s (direct-form FLOPs)" % (b, cin, cout, h, kind, ms, gf / ms), flush=True)
        a, _ = run("direct", x, dy, cin, cout); c, _ = run("wino", x, dy, cin, cout); torch.cuda.synchronize()
        print("   max |wino - direct| / max|direct| = %.2e" % ((a - c).abs().max().item() / a.abs().max().item()), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    check() if what == "check" else time_()

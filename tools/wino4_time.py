"""F(4x4,3x3) Winograd conv (conv3x3_wino4_f32.hip) next to the F(2x2) kernel: correctness against an f64 CPU convolution on
odd shapes (partial blocks, several chunk counts, residual, ReLU, data gradient), then steady-state time on the step's layers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import ops, lib as _lib
if os.environ.get("ODVAE_PROBE_LIB"):
    _lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]
dev = "cuda:0"
F = torch.nn.functional
ops.WINOGRAD4 = True
shapes = [(1, 64, 64, 16, 32), (2, 64, 64, 20, 36), (1, 72, 96, 16, 32), (2, 128, 128, 64, 48), (1, 64, 256, 24, 40), (3, 256, 128, 16, 32),
          (1, 512, 512, 32, 32), (2, 128, 128, 128, 128), (2, 8 * 9, 8 * 11, 36, 68),
          (8, 128, 128, 128, 128), (2, 128, 128, 256, 512), (5, 64, 256, 112, 160), (3, 256, 512, 64, 96)]   # several tiles per block (persistent form)
for (b, cin, cout, h, wd) in [] if os.environ.get("WINO_NOCHECK") else shapes:
    assert ops._wino4_ok(h, wd, cin, cout), (h, wd, cin, cout)
    x = torch.randn(b, h, wd, cin, device=dev).permute(0, 3, 1, 2).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).requires_grad_(True)
    bias = torch.randn(cout, device=dev)
    res = torch.randn(b, h, wd, cout, device=dev).permute(0, 3, 1, 2)
    y = ops.conv3x3(x, w, bias, residual=res)
    dy = torch.randn_like(y)
    dx, = torch.autograd.grad(y, x, dy)
    xc, wc = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu()
    ref = F.conv2d(xc, wc, bias.double().cpu(), padding=1) + res.double().cpu()
    dref, = torch.autograd.grad(ref, xc, dy.double().cpu())
    err = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    derr = (dx.double().cpu() - dref).abs().max().item() / dref.abs().max().item()
    with torch.no_grad():
        yr = ops.conv3x3(x, w, None, relu=True)      # (fused ReLU: the F(2x2) kernel, see ops._Conv3x3)
        rr = F.relu(F.conv2d(xc, wc, None, padding=1))
    rerr = (yr.double().cpu() - rr).abs().max().item() / rr.abs().max().item()
    ok = max(err, derr, rerr) < 3e-5
    print("check %dx%dx%dx%d->%d: fwd %.2e dgrad %.2e relu %.2e %s" % (b, cin, h, wd, cout, err, derr, rerr, "ok" if ok else "WRONG"), flush=True)
    assert ok
for (b, cin, cout, h) in ([(32, 128, 128, 256)] if os.environ.get("WINO_ONLY_BIG") else [(32, 128, 128, 256), (32, 256, 256, 128), (32, 512, 512, 64), (32, 256, 256, 64)]):
    x = torch.randn(b, h, h, cin, device=dev).permute(0, 3, 1, 2)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    out = []
    for f4 in (False, True):
        ops.WINOGRAD4 = f4
        with torch.no_grad():
            for _ in range(4 if os.environ.get("WINO_ONLY_BIG") else 40): y = ops.conv3x3(x, w, bias)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2 if os.environ.get("WINO_ONLY_BIG") else 60): y = ops.conv3x3(x, w, bias)
            e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 60)
    ops.WINOGRAD4 = True
    with torch.no_grad():       # the same with the GroupNorm statistics of y written by the output transform
        for _ in range(4 if os.environ.get("WINO_ONLY_BIG") else 40): y = ops.conv3x3(x, w, bias, gn_stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2 if os.environ.get("WINO_ONLY_BIG") else 60): y = ops.conv3x3(x, w, bias, gn_stats=True)
        e1.record(); torch.cuda.synchronize()
    t_stats = e0.elapsed_time(e1) / (2 if os.environ.get("WINO_ONLY_BIG") else 60)
    fl = 2.0 * 9 * cin * cout * b * h * h
    print("B%d %d->%d @%d: F(2x2) %.3f ms  F(4x4) %.3f ms  (x%.2f; %.0f / %.0f direct-form TFLOP/s; F(4x4) executes %.3f of the f32 MFMA peak)"
          % (b, cin, cout, h, out[0], out[1], out[0] / out[1], fl / out[0] / 1e9, fl / out[1] / 1e9, fl / 4 / out[1] / 1e9 / 157.3), flush=True)
    print("      with GroupNorm statistics in the output transform: %.3f ms (+%.1f %%)" % (t_stats, 100 * (t_stats / out[1] - 1)), flush=True)

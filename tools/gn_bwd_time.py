"""GroupNorm backward: read-once (fused) form against the two-kernel form on the training step's shapes (B = 32), in one process.
Prints per shape: ms per call of each form (HIP events around 20 calls after 3 warm-up calls, alternating forms), the algorithmic bytes
(x, dy, skip read + dx written) and what that is in TB/s and as a fraction of the 8 TB/s HBM peak.  usage: python tools/gn_bwd_time.py [--bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from odvae_amd import lib as _lib, ops
    L = _lib.load()
    bf16 = "--bf16" in sys.argv
    dt = torch.bfloat16 if bf16 else torch.float32
    esz = 2 if bf16 else 4
    dev = torch.device("cuda:0")
    shapes = [(32, 128, 256, 256), (32, 128, 128, 128), (32, 256, 128, 128), (32, 256, 64, 64), (32, 256, 32, 32), (32, 512, 32, 32), (32, 512, 16, 16)]
    if "--big" in sys.argv:
        shapes = [(32, 128, 512, 512), (32, 256, 256, 256)] + shapes
    fn = L.odvae_groupnorm_bwd_bf16 if bf16 else L.odvae_groupnorm_bwd_f32
    wsf = L.odvae_groupnorm_bf16_workspace_bytes if bf16 else L.odvae_groupnorm_workspace_bytes
    for n, c, h, w in shapes:
        x = (torch.randn(n, h, w, c, device=dev) * 2 + 0.5).to(dt)
        dy = torch.randn(n, h, w, c, device=dev).to(dt)
        sk = torch.randn(n, h, w, c, device=dev).to(dt)
        dx = torch.empty_like(x)
        gamma, beta = torch.randn(c, device=dev), torch.randn(c, device=dev)
        xg = x.float().reshape(n, h * w, 32, c // 32)
        mean = xg.mean(dim=(1, 3)).contiguous()
        rstd = (1.0 / torch.sqrt(xg.var(dim=(1, 3), unbiased=False) + 1e-6)).contiguous()
        dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
        wp, wn = ops._ws(wsf(n, h * w, c, 32), x)

        def call(skip):
            _lib.check(fn(x.data_ptr(), dy.data_ptr(), n, h * w, c, 32, gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1,
                          dx.data_ptr(), dg.data_ptr(), db.data_ptr(), sk.data_ptr() if skip else None, wp, wn, _lib.stream_ptr()), "gn bwd")
        res = {}
        for skip in (False, True):
            for mode in (0, 1):
                L.odvae_groupnorm_select_backward(mode)
                for _ in range(3):
                    call(skip)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    call(skip)
                e1.record()
                torch.cuda.synchronize()
                res[(skip, mode)] = e0.elapsed_time(e1) / 20
        L.odvae_groupnorm_select_backward(-1)
        for skip in (False, True):
            nbytes = (3 + skip) * n * h * w * c * esz
            t0, t1 = res[(skip, 0)], res[(skip, 1)]
            print("%s N=%d C=%d %dx%d skip=%d: two-kernel %.3f ms (%.2f TB/s, %.2f of peak) | read-once %.3f ms (%.2f TB/s, %.2f of peak) | x%.2f"
                  % ("bf16" if bf16 else "f32", n, c, h, w, skip, t0, nbytes / t0 / 1e9, nbytes / t0 / 1e9 / 8, t1, nbytes / t1 / 1e9, nbytes / t1 / 1e9 / 8, t0 / t1), flush=True)
    print("fused timeouts:", L.odvae_groupnorm_fused_timeouts())


if __name__ == "__main__":
    main()

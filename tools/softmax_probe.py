import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import lib
L = lib.load()
dev = torch.device("cuda:0")
rows, cols = 32 * 4096, 4096
x = torch.randn(rows, cols, device=dev); y = torch.empty_like(x); dp = torch.randn(rows, cols, device=dev)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
f = t(lambda: lib.check(L.odvae_softmax_rows_f32(x.data_ptr(), y.data_ptr(), rows, cols, 0.0625, lib.stream_ptr()), "sm"))
b = t(lambda: lib.check(L.odvae_softmax_rows_bwd_f32(y.data_ptr(), dp.data_ptr(), dp.data_ptr(), rows, cols, 0.0625, lib.stream_ptr()), "smb"))
gb = rows * cols * 4 / 1e9
print("softmax fwd %.3f ms (%.0f GB/s for 2 passes)  bwd %.3f ms (%.0f GB/s for 3 passes)" % (f, 2 * gb / f * 1e3, b, 3 * gb / b * 1e3))

"""Does a re-read that fits the 256 MiB Infinity Cache stream faster than a read from HBM?  torch.sum (a bandwidth-bound read) over ONE buffer
again and again against the same bytes spread over many buffers (1.5 GiB cycle, never resident).  usage: python tools/mall_probe.py"""
import torch
dev = torch.device("cuda:0")
for mb in (32, 64, 128, 192, 256, 384):
    n = mb * 1024 * 1024 // 4
    k = max(2, int(1536 / mb))
    bufs = [torch.randn(n, device=dev) for _ in range(k)]
    def run(cycle, reps=40):
        for i in range(5): bufs[i % cycle].sum()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps): bufs[i % cycle].sum()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    same, spread = run(1), run(k)
    # write-then-read: a kernel writes the buffer, the next reads it (the GroupNorm backward's reduce -> apply pattern is read -> read, the conv -> GroupNorm one write -> read)
    def wr(cycle, reps=40):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            b = bufs[i % cycle]; b.mul_(1.0001); b.sum()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    print("%4d MB: read same buffer %.3f ms (%.2f TB/s) | spread over %d buffers %.3f ms (%.2f TB/s) | x%.2f || mul_ + sum: same %.3f ms, spread %.3f ms"
          % (mb, same, mb * 1.048576e6 / same / 1e9, k, spread, mb * 1.048576e6 / spread / 1e9, spread / same, wr(1), wr(k)), flush=True)
    del bufs
    torch.cuda.empty_cache()

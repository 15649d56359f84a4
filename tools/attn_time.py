"""f32 attention block (ops.attention_qkv) at the benchmark's shape (N = 32, C = 256, 64 x 64 tokens): forward and backward time with the
row softmax folded into the products (ODVAE_ATTN_FOLDED_SOFTMAX=1, default) and as a separate pass (0), alternating, HIP events.
usage: python tools/attn_time.py [N] [C] [H]      (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from odvae_amd import ops
    n, c, h = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 256, 64)))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = (torch.randn(n, 3 * c, h, h, device=dev, generator=g) * 0.6).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(n, c, h, h, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    for rnd in range(2):
        for folded in (False, True):
            ops.ATTN_FOLDED_SOFTMAX = folded
            tf = tb = 0.0
            for it in range(5):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                o = ops.attention_qkv(qkv)
                e[1].record()
                o.backward(go)
                e[2].record()
                torch.cuda.synchronize()
                qkv.grad = None
                if it >= 2:
                    tf += e[0].elapsed_time(e[1]) / 3
                    tb += e[1].elapsed_time(e[2]) / 3
            print("folded=%d  forward %.3f ms  backward %.3f ms  (flag %s)" % (folded, tf, tb, int(ops._ATTN_LAST_FLAG.item()) if folded else "-"), flush=True)


if __name__ == "__main__":
    main()

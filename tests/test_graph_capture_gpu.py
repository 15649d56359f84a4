"""INTEGRATION.md says of the C ABI: "every call enqueues on the hipStream_t it is given; no device sync, no host allocation, graph-capture
safe".  This test holds it to that: a chain of the hot path's entry points -- F(4x4) Winograd conv with the GroupNorm statistics epilogue,
GroupNorm + swish, the fused q/k/v 1x1 conv, single-head attention with the folded softmax (incl. its predicated fallback launches), the
projection 1x1 conv with the residual folded in -- is captured into a hipGraph on a side stream after one eager warm-up (weight packs and
the workspace exist by then) and replayed on NEW input contents; the replay must equal the eager result bit for bit.  Forward only: the one attempt to
capture forward + torch-autograd backward of the same chain (round 5) ended in a segmentation fault inside `capture_end` of this ROCm 7.2 /
torch 2.10 build -- a host-side crash of the runtime, not a kernel fault -- and was not pursued (DESIGN.md 7 lists it with the other things
that stand between the training loop and a hipGraph, and why removing them has not been worth it)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _chain(x, p):
    from odvae_amd import ops
    h = ops.conv3x3(x, p["w1"], p["b1"], None, 0, gn_stats=True)
    h = ops.group_norm(h, p["g"], p["be"], 32, 1e-6, True)
    qkv = ops.conv1x1(h, p["wq"], p["bq"], None)
    a = ops.attention_qkv(qkv)
    return ops.conv1x1(a, p["wo"], p["bo"], h)


def _params(c, requires_grad):
    g = torch.Generator().manual_seed(3)
    mk = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV).requires_grad_(requires_grad)
    return {"w1": mk(c, c, 3, 3, scale=(9 * c) ** -0.5), "b1": mk(c, scale=0.1), "g": (1 + 0.1 * torch.randn(c, generator=g)).to(DEV).requires_grad_(requires_grad),
            "be": mk(c, scale=0.1), "wq": mk(3 * c, c, 1, 1, scale=c ** -0.5), "bq": mk(3 * c, scale=0.1),
            "wo": mk(c, c, 1, 1, scale=c ** -0.5), "bo": mk(c, scale=0.1)}


def test_forward_chain_replays_from_a_graph(hip_lib):
    c, n, hw = 64, 2, 32
    p = _params(c, False)
    gen = torch.Generator().manual_seed(5)
    xs = [torch.randn(n, hw, hw, c, generator=gen).permute(0, 3, 1, 2) for _ in range(3)]
    static_x = xs[0].to(DEV).clone()
    with torch.no_grad():
        want = []
        for x in xs:                                   # eager results (the first call also makes the packs and sizes the workspace)
            want.append(_chain(x.to(DEV), p).clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            _chain(static_x, p)                        # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            static_y = _chain(static_x, p)
        for x, w in zip(xs, want):
            static_x.copy_(x.to(DEV))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(static_y, w)            # same kernels, same order, same data: bit-identical

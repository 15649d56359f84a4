"""Encoder / Decoder parity: HIP modules vs the CPU oracle (oracle/ldm_model.py) with the same state_dict.
Width-reduced model (ch=32, same ch_mult / attention placement as autoencoder_kl_16x16x16.yaml) at 64x64, B=2.
Tolerance (fp32, ~60 layers deep): outputs 1e-3, parameter gradients 3e-3, relative to max|ref|."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DD = dict(double_z=True, z_channels=16, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 1, 2, 2, 4],
          num_res_blocks=2, attn_resolutions=[16], dropout=0.0)


def rel_err(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())


@pytest.mark.parametrize("which", ["encoder", "decoder"])
def test_encoder_decoder_match_oracle(hip_lib, which):
    from odvae_amd import modules
    from oracle import ldm_model
    torch.manual_seed(23)
    ref = getattr(ldm_model, which.capitalize())(**DD)
    net = getattr(modules, which.capitalize())(**DD)
    missing = net.load_state_dict(ref.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net = net.to("cuda:0")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 64, 64, generator=g) if which == "encoder" else torch.randn(2, 16, 4, 4, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = ref(xr)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = x.to("cuda:0").requires_grad_(True)
    y = net(xd)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert rel_err(y, y_ref) < 1e-3, "forward rel err %.3e" % rel_err(y, y_ref)
    y.backward(gy.to("cuda:0"))
    assert rel_err(xd.grad, xr.grad) < 3e-3, "input grad rel err %.3e" % rel_err(xd.grad, xr.grad)
    worst = ("", 0.0)
    ref_params = dict(ref.named_parameters())
    # some gradients are analytically zero (attention k.bias: softmax is shift-invariant), so errors are
    # measured against max(|ref grad|, 1e-3 * largest gradient in the net)
    scale = max(p.grad.abs().max().item() for p in ref_params.values())
    for name, p in net.named_parameters():
        r = ref_params[name].grad.double()
        e = (p.grad.detach().cpu().double() - r).abs().max().item() / max(r.abs().max().item(), 1e-3 * scale)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] < 3e-3, "param grad %s rel err %.3e" % worst


def test_attn_block_16384_tokens_with_score_budget_matches_oracle(hip_lib, monkeypatch):
    """AttnBlock at BASELINE.json configs[4]'s attention resolution in fp32: 128 x 128 = 16 384 tokens, 1 GiB of scores per image.
    With the score budget at one image the block runs image by image through one score buffer and recomputes P in the backward
    (ops._AttentionRecompute) -- the form that lets the f32 path take 512 x 512 at B = 32.  Reference: oracle/ldm_model.AttnBlock
    (materialised scores, torch CPU), B = 2, C = 64.  Tolerances as for the other module tests."""
    from odvae_amd import modules, ops
    from oracle import ldm_model
    torch.manual_seed(5)
    c, hw = 64, 128
    monkeypatch.setattr(ops, "ATTN_SCORE_BUDGET", (hw * hw) ** 2 * 4)
    ref = ldm_model.AttnBlock(c)
    net = modules.AttnBlock(c)
    assert not net.load_state_dict(ref.state_dict(), strict=True).missing_keys
    net = net.to("cuda:0")
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, c, hw, hw, generator=g)
    gy = torch.randn(2, c, hw, hw, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = ref(xr)
    y_ref.backward(gy)
    xd = x.to("cuda:0").requires_grad_(True)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    y = net(xd)
    y.backward(gy.to("cuda:0"))
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < 3.5 * 2 ** 30, "score buffers outlived their group: %.2f GiB" % (peak / 2 ** 30)   # P + dS of ONE image + small tensors
    assert rel_err(y, y_ref) < 1e-3
    assert rel_err(xd.grad, xr.grad) < 3e-3
    scale = max(pr.grad.abs().max().item() for pr in ref.parameters())
    for (name, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
        # k.bias has a gradient of exactly zero in exact arithmetic (softmax is invariant to a per-query shift of the scores): both
        # sides hold rounding noise there, so every parameter is measured against at least 1e-3 of the largest gradient
        e = (p.grad.detach().cpu().double() - pr.grad.double()).abs().max().item() / max(pr.grad.abs().max().item(), 1e-3 * scale)
        assert e < 3e-3, "%s: %.3e" % (name, e)

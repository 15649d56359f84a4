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

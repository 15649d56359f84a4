"""HIP path vs the COMMITTED fixtures of SURVEY.md 8(c) (tests/golden/ops_tiny.npz, model_ch32.npz; generator
tests/golden/make_op_fixtures.py): no live oracle on the GPU box, only numbers in the repo.  Tolerances as in
test_ops_gpu.py / test_model_gpu.py: per-op forward 2e-4, backward 5e-4 of max|ref|; model-level 1e-3 / 5e-3."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
YAML = os.path.join(GOLD, "autoencoder_kl_16x16x16.yaml")
DEV = "cuda:0"


def close(a, b, tol, what):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    err = np.abs(a - b).max()
    ref = max(1.0, np.abs(b).max())
    assert err <= tol * ref, "%s: max err %.3e > %.1e * %.3e" % (what, err, tol, ref)


@pytest.fixture(scope="module")
def ops_fix():
    return np.load(os.path.join(GOLD, "ops_tiny.npz"))


def _t(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).requires_grad_(grad)


def test_groupnorm_swish_fixture(hip_lib, ops_fix):
    from odvae_amd import ops
    g = ops_fix
    x, gamma, beta = _t(g["gn_swish.x"], True), _t(g["gn_swish.gamma"], True), _t(g["gn_swish.beta"], True)
    y = ops.group_norm(x, gamma, beta, 32, 1e-6, True)
    close(y, g["gn_swish.y"], 2e-4, "gn y")
    y.backward(_t(g["gn_swish.dy"]))
    close(x.grad, g["gn_swish.dx"], 5e-4, "gn dx")
    close(gamma.grad, g["gn_swish.dgamma"], 5e-4, "gn dgamma")
    close(beta.grad, g["gn_swish.dbeta"], 5e-4, "gn dbeta")


@pytest.mark.parametrize("name,mode", [("conv3x3", 0), ("downsample", 1), ("upsample", 2)])
def test_conv3x3_fixtures(hip_lib, ops_fix, name, mode):
    from odvae_amd import ops
    g = ops_fix
    x, w, b = _t(g[name + ".x"], True), _t(g[name + ".w"], True), _t(g[name + ".b"], True)
    y = ops.conv3x3(x, w, b, None, mode)
    close(y, g[name + ".y"], 2e-4, name + " y")
    y.backward(_t(g[name + ".dy"]))
    close(x.grad, g[name + ".dx"], 5e-4, name + " dx")
    close(w.grad, g[name + ".dw"], 5e-4, name + " dw")
    close(b.grad, g[name + ".db"], 5e-4, name + " db")


def test_conv1x1_fixture(hip_lib, ops_fix):
    from odvae_amd import ops
    g = ops_fix
    x, w, b = _t(g["conv1x1.x"], True), _t(g["conv1x1.w"], True), _t(g["conv1x1.b"], True)
    y = ops.conv1x1(x, w, b)
    close(y, g["conv1x1.y"], 2e-4, "1x1 y")
    y.backward(_t(g["conv1x1.dy"]))
    close(x.grad, g["conv1x1.dx"], 5e-4, "1x1 dx")
    close(w.grad, g["conv1x1.dw"], 5e-4, "1x1 dw")
    close(b.grad, g["conv1x1.db"], 5e-4, "1x1 db")


@pytest.mark.parametrize("name", ["attn", "resblock"])
def test_block_fixtures(hip_lib, ops_fix, name):
    """AttnBlock(32) over 64 tokens; ResnetBlock 32 -> 64 with the 1x1 shortcut (residual in the conv epilogue)."""
    from odvae_amd import synthetic
    from odvae_amd.modules import AttnBlock, ResnetBlock
    g = ops_fix
    mod = AttnBlock(32) if name == "attn" else ResnetBlock(in_channels=32, out_channels=64, dropout=0.0, temb_channels=0)
    synthetic.fill_state_procedural(mod, seed=5)
    mod = mod.to(DEV)
    x = _t(g[name + ".x"], True)
    y = mod(x)
    close(y, g[name + ".y"], 5e-4, name + " y")
    y.backward(_t(g[name + ".dy"]))
    close(x.grad, g[name + ".dx"], 2e-3, name + " dx")
    for k, p in mod.named_parameters():
        close(p.grad, g["%s.grad.%s" % (name, k)], 2e-3, "%s d%s" % (name, k))


def test_model_ch32_fixture(hip_lib):
    """One training step of the width-reduced model against the committed oracle outputs: latent, reconstruction, every
    logged scalar, the total and the norm of every parameter gradient."""
    from odvae_amd import synthetic
    g = np.load(os.path.join(GOLD, "model_ch32.npz"))
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32)
    synthetic.fill_state_procedural(model, seed=23)
    model = model.to(DEV).train()
    model._global_step = 1
    model.loss.log_exact_g_loss = True     # the fixture holds the reference's g_loss = -mean D(x_rec) with the discriminator off
    batch = synthetic.make_batch(2, 64, seed=23)
    noise = synthetic.make_noise(2, 4, seed=24)
    model.injected_noise = noise
    loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, 0)
    loss.backward()
    close(loss, g["loss"], 1e-3, "total loss")
    logged = model.logged_metrics
    for k in g.files:
        if k.startswith("log."):
            v = logged[k[4:]]
            close(torch.as_tensor(float(v)), g[k], 2e-3, k)
    rgb = model._rgb_input(batch)
    with torch.no_grad():
        model._global_step = 1
        dec_obj, dec_pose, post, _ = model.forward(rgb)
    close(dec_obj, g["dec_obj"], 2e-3, "reconstruction")
    close(dec_pose, g["dec_pose"], 2e-3, "dec_pose")
    close(post.parameters, g["moments"], 1e-3, "moments")
    params = dict(model.named_parameters())
    for name, want in zip(g["grad_names"], g["grad_norms"]):
        p = params[str(name)]
        got = 0.0 if p.grad is None else p.grad.double().norm().item()
        assert abs(got - want) <= 5e-3 * max(want, 1e-3 * float(g["grad_norms"].max())), (str(name), got, want)
    close(params["decoder.conv_out.weight"].grad, g["grad.decoder.conv_out.weight"], 5e-3, "d conv_out.weight")
    close(params["encoder.conv_in.weight"].grad, g["grad.encoder.conv_in.weight"], 5e-3, "d conv_in.weight")

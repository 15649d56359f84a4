"""Generates the SURVEY.md 8(c) fixtures 1-2 with the build's CPU oracle (torch.nn.functional ops + oracle/):

  tests/golden/ops_tiny.npz     per-op cases at tiny shapes: inputs, weights, outputs, upstream gradient, input and weight
                                gradients of GroupNorm+swish, conv3x3 s1, Downsample conv (pad (0,1,0,1), s2), Upsample conv
                                (nearest 2x), conv1x1, AttnBlock, ResnetBlock with a 1x1 shortcut.
  tests/golden/model_ch32.npz   width-reduced model (ch=32, same ch_mult / attention / resolution), 64x64, B=2, seed 23,
                                global_step 1: latent z, reconstruction, every scalar log term, total loss, the norm of every
                                parameter gradient.  Weights are NOT stored: synthetic.fill_state_procedural(seed 23) derives
                                them from the state_dict keys; batch and noise come from synthetic.make_batch / make_noise.

These pin the ORACLE (parity unpinned: the reference has no fixtures and cannot be imported) and let the GPU parity tests run
against committed numbers instead of a live oracle.      python tests/golden/make_op_fixtures.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
YAML = os.path.join(GOLD, "autoencoder_kl_16x16x16.yaml")


def _rand(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


def op_cases():
    """name -> dict of numpy arrays.  Every case: forward on x (and parameters), backward from a fixed upstream gradient dy."""
    from oracle.ldm_model import AttnBlock, ResnetBlock
    g = torch.Generator().manual_seed(2310)
    out = {}

    def record(name, y, dy, leaves):
        y.backward(dy)
        out[name + ".y"] = y.detach().numpy()
        out[name + ".dy"] = dy.numpy()
        for k, t in leaves.items():
            out["%s.%s" % (name, k)] = t.detach().numpy()
            out["%s.d%s" % (name, k)] = t.grad.numpy()

    # GroupNorm(32, eps 1e-6) + swish
    x = _rand(g, 2, 64, 8, 8).requires_grad_(True)
    gamma = (1 + 0.1 * _rand(g, 64)).requires_grad_(True)
    beta = (0.1 * _rand(g, 64)).requires_grad_(True)
    h = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    record("gn_swish", h * torch.sigmoid(h), _rand(g, 2, 64, 8, 8), dict(x=x, gamma=gamma, beta=beta))
    # conv3x3 stride 1 pad 1
    x = _rand(g, 2, 32, 8, 8).requires_grad_(True)
    w = _rand(g, 64, 32, 3, 3, scale=0.06).requires_grad_(True)
    b = _rand(g, 64, scale=0.1).requires_grad_(True)
    record("conv3x3", F.conv2d(x, w, b, padding=1), _rand(g, 2, 64, 8, 8), dict(x=x, w=w, b=b))
    # Downsample: pad (0,1,0,1) + conv3x3 stride 2
    x = _rand(g, 2, 32, 8, 8).requires_grad_(True)
    w = _rand(g, 32, 32, 3, 3, scale=0.06).requires_grad_(True)
    b = _rand(g, 32, scale=0.1).requires_grad_(True)
    record("downsample", F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2), _rand(g, 2, 32, 4, 4), dict(x=x, w=w, b=b))
    # Upsample: nearest 2x + conv3x3
    x = _rand(g, 2, 32, 8, 8).requires_grad_(True)
    w = _rand(g, 32, 32, 3, 3, scale=0.06).requires_grad_(True)
    b = _rand(g, 32, scale=0.1).requires_grad_(True)
    record("upsample", F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1),
           _rand(g, 2, 32, 16, 16), dict(x=x, w=w, b=b))
    # conv1x1
    x = _rand(g, 2, 32, 8, 8).requires_grad_(True)
    w = _rand(g, 64, 32, 1, 1, scale=0.18).requires_grad_(True)
    b = _rand(g, 64, scale=0.1).requires_grad_(True)
    record("conv1x1", F.conv2d(x, w, b), _rand(g, 2, 64, 8, 8), dict(x=x, w=w, b=b))
    # AttnBlock (C = 32, T = 64) and ResnetBlock 32 -> 64 with nin_shortcut: module parameters by state_dict key
    from odvae_amd import synthetic
    for name, mod, cin in (("attn", AttnBlock(32), 32), ("resblock", ResnetBlock(in_channels=32, out_channels=64, dropout=0.0, temb_channels=0), 32)):
        synthetic.fill_state_procedural(mod, seed=5)
        x = _rand(g, 2, cin, 8, 8).requires_grad_(True)
        y = mod(x) if name == "attn" else mod(x, None)
        dy = _rand(g, *y.shape)
        y.backward(dy)
        out[name + ".x"], out[name + ".dx"] = x.detach().numpy(), x.grad.numpy()
        out[name + ".y"], out[name + ".dy"] = y.detach().numpy(), dy.numpy()
        for k, p in mod.named_parameters():
            out["%s.grad.%s" % (name, k)] = p.grad.numpy()
    return out


def build_oracle_ch32():
    from odvae_amd import synthetic
    from oracle.autoencoder import PoseAutoencoder
    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32)
    p = mcfg.params.to_container()
    ref = PoseAutoencoder(p["ddconfig"], dict(p["lossconfig"]["params"]), p["embed_dim"], p["pose_decoder_config"]["params"],
                          p["pose_encoder_config"]["params"], feat_dims=p["feat_dims"], dropout_prob_init=p["dropout_prob_init"],
                          dropout_prob_final=p["dropout_prob_final"], dropout_warmup_steps=p["dropout_warmup_steps"],
                          pose_conditioned_generation_steps=p["pose_conditioned_generation_steps"])
    synthetic.fill_state_procedural(ref, seed=23)
    ref.global_step = 1
    return ref


def model_case():
    from odvae_amd import synthetic
    ref = build_oracle_ch32().train()
    batch = synthetic.make_batch(2, 64, seed=23)
    noise = synthetic.make_noise(2, 4, seed=24)
    loss, log, aux = ref.training_step(batch, 0, noise)
    loss.backward()
    out = {"loss": np.array(loss.item()), "dec_obj": aux["dec_obj"].detach().numpy(), "dec_pose": aux["dec_pose"].detach().numpy(),
           "moments": aux["posterior"].parameters.detach().numpy(), "z": aux["posterior"].sample(noise["posterior_eps"]).detach().numpy()}
    for k, v in log.items():
        if not torch.is_tensor(v) or v.numel() == 1:
            out["log." + k] = np.array(float(v))
    names, norms = [], []
    for k, p in ref.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(p.grad.double().norm().item())
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms)
    out["grad.decoder.conv_out.weight"] = ref.decoder.conv_out.weight.grad.numpy()
    out["grad.encoder.conv_in.weight"] = ref.encoder.conv_in.weight.grad.numpy()
    return out


if __name__ == "__main__":
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    np.savez_compressed(os.path.join(GOLD, "ops_tiny.npz"), **op_cases())
    np.savez_compressed(os.path.join(GOLD, "model_ch32.npz"), **model_case())
    for f in ("ops_tiny.npz", "model_ch32.npz"):
        print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, "KiB")

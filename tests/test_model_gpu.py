"""Whole-path parity: PoseAutoencoder training step on the HIP kernels vs the CPU oracle (oracle/autoencoder.py),
same weights, same batch, same injected noise.  Width-reduced network (ch=32) at 64x64 and the benchmark's own network (ch=128) at
256x256, B=2, rec+KL only.
Stated fp32 tolerances = at most 4x what the path was MEASURED to use (tools/parity_margins.py -> profiles/r04_parity_margins.json;
deviations relative to max|ref|, gradients to max(|ref grad|, 1e-3 * largest gradient)):
  ch=32 @64x64 (no layer wide enough for F(4x4)):  outputs / loss terms 2e-5 (measured 6.7e-6), gradients 4.4e-4 (1.2e-4: a conv bias),
  ch=128 @256x256, Winograd F(4x4,3x3) (default):  outputs / loss terms 8e-5 (1.7e-5), gradients 4.6e-3 (2.8e-5 ... 2.3e-3 between builds
      that differ ONLY in summation order: the reconstruction term is an L1, its gradient sign(x_rec - x) flips at a pixel whose
      |x_rec - x| is below the forward's own deviation (1.7e-5 of the value range), and ONE flipped value of the 393 216 moves
      every gradient behind it by 2 / numel of that pixel's Jacobian -- measured with two builds that differ in the split-K reduce's
      order only (profiles/r04_sign_flip_ab.txt): 1 flipped value -> decoder 9.3e-4, post_quant_conv.weight 2.3e-3; 0 flipped -> 6.7e-6,
      1.8e-5.  tools/parity_margins.py reports the count (l1_sign_flips).  The tolerance covers two or three flips),
  the same on F(2x2,3x3) (ODVAE_CONV_WINOGRAD4=0):  outputs 2.7e-5 (5.5e-6), gradients 4e-5 (0.9e-5 ... 1.1e-5),
  3-step loss curves 4e-6 (1.3e-6 ... 2.1e-6); weights after 3 Adam steps use 0.08-0.16 of their bound (2.2 lr per step + 5e-3 max|w|)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

YAML = os.path.join(os.path.dirname(__file__), "golden", "autoencoder_kl_16x16x16.yaml")


def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())


def build_pair(perceptual_weight=0.0, disc_factor=0.0, latent_hw=4, activation_checkpoint=False, phase="vae", ch=32):
    from odvae_amd import synthetic
    from odvae_amd.config import instantiate_from_config
    from oracle.autoencoder import PoseAutoencoder as OraclePA
    torch.manual_seed(23)
    mcfg, cfg = synthetic.model_config(YAML, latent_hw=latent_hw, ch=ch, perceptual_weight=perceptual_weight,
                                       disc_factor=disc_factor, phase=phase)
    if activation_checkpoint:
        mcfg.params.ddconfig["activation_checkpoint"] = activation_checkpoint
    model = instantiate_from_config(mcfg)
    model.learning_rate = 12 * cfg.model.base_learning_rate
    p = mcfg.params.to_container()
    p["ddconfig"].pop("activation_checkpoint", None)
    lk = dict(p["lossconfig"]["params"])
    ref = OraclePA(p["ddconfig"], lk, p["embed_dim"], p["pose_decoder_config"]["params"], p["pose_encoder_config"]["params"],
                   feat_dims=p.get("feat_dims", [16, 16, 16]), dropout_prob_init=p["dropout_prob_init"], dropout_prob_final=p["dropout_prob_final"],
                   dropout_warmup_steps=p["dropout_warmup_steps"],
                   pose_conditioned_generation_steps=p["pose_conditioned_generation_steps"],
                   add_noise_to_z_obj=p["add_noise_to_z_obj"], train_on_yaw=p["train_on_yaw"])
    res = ref.load_state_dict(model.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ref.learning_rate = model.learning_rate
    return model.to("cuda:0"), ref


@pytest.mark.parametrize("global_step", [1, 0])
def test_training_step_matches_oracle(hip_lib, global_step):
    """global_step 1: every loss term (the rec / KL terms join when step > encoder_pretrain_steps, contperceptual.py:307);
    global_step 0: the pose-only total of the very first step."""
    model, ref = build_pair()
    check_step(model, ref, global_step, height=64, latent_hw=4, tol_out=2e-5, tol_grad=4.4e-4)


def test_training_step_at_headline_shapes_matches_oracle(hip_lib):
    """The benchmark's own network and resolution (ch=128, 256x256, z = 16x16x16; B=2 so that the oracle finishes in
    seconds): every kernel runs at the channel counts, tile counts and 4 096 attention tokens of BASELINE.json configs[1],
    through the Winograd, parity-class upsample and thin-side paths the width-reduced tests only touch in part."""
    from odvae_amd import ops
    model, ref = build_pair(latent_hw=16, ch=None)
    f4 = ops.WINOGRAD and ops.WINOGRAD4
    check_step(model, ref, 1, height=256, latent_hw=16, tol_out=8e-5 if f4 else 2.7e-5, tol_grad=4.6e-3 if f4 else 4e-5)


def test_headline_gradients_with_the_l1_sign_taken_from_the_oracle(hip_lib):
    """The flip-insensitive twin of the test above.  The 4.6e-3 gradient bound there exists because the pixel term
    is an L1: sign(x_rec - x) may differ from the oracle's at a pixel whose |x_rec - x| is below the forward's own 2e-5 deviation, and one
    flipped value moves every gradient behind it.  Here the reconstruction term is made LINEAR with the sign tensor fixed by the oracle:
        total = [pose / class / box / fill-factor / KL terms, evaluated by each side's own loss with the pixel term off]
                + sum(x_rec * G),   G = mask * sign(x_rec_oracle - x) / ((exp(logvar) + 1e-8) * #unmasked samples)
    (G is exactly the gradient the oracle's L1 NLL sends into the reconstruction), on the benchmark's own network and resolution
    (ch = 128, 256 x 256, 4 096 attention tokens, Winograd F(4x4) convs).  Every kernel of the decoder's and encoder's backward runs as
    in the step test, from the same upstream gradient on both sides, so the bound is set by the kernels alone: 1.2e-4
    (measured: profiles/r05_parity_margins.txt)."""
    from odvae_amd import synthetic
    model, ref = build_pair(latent_hw=16, ch=None)
    model.train(); ref.train()
    model._global_step = ref.global_step = 1
    for m in (model, ref):      # pixel term off inside the loss (contperceptual.py:222-224); everything else as in the headline step
        m.loss.pose_conditioned_generation_steps = 10 ** 9
    batch = synthetic.make_batch(2, 256, seed=5)
    noise = synthetic.make_noise(2, 16, dropout_p=0.7, seed=6)
    # oracle
    rgb_ref = ref._rescale(batch["patch"].float())
    pose_ref = batch["pose_6d"].clone().float()
    pose_ref[:, 3] = batch["yaw"]
    dec_ref, dpose_ref, post_ref, bpost_ref = ref.forward(rgb_ref, noise)
    nbg = float((batch["class_id"] != 1).sum())
    with torch.no_grad():
        G = batch["mask_2d_bbox"] * torch.sign(dec_ref * batch["mask_2d_bbox"] - rgb_ref * batch["mask_2d_bbox"]) \
            / ((torch.exp(ref.loss.logvar) + 1e-8) * nbg)
    rest_ref, _ = ref.loss(rgb_ref, None, pose_ref, dec_ref, dpose_ref, batch["class_id"], batch["class_name"], batch["bbox_sizes"],
                           batch["fill_factor"].float(), post_ref, bpost_ref, 0, 1, batch["mask_2d_bbox"], last_layer=ref.decoder.conv_out.weight)
    (rest_ref + (dec_ref * G).sum()).backward()
    # HIP path
    model.injected_noise = noise
    b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    rgb, mask_gt, pose_gt, class_gt, labels, bbox_gt, fill_gt, mask2d = model._unpack(b)
    dec, dpose, post, bpost = model.forward(rgb)
    rest, _ = model.loss(rgb, mask_gt, pose_gt, dec, dpose, class_gt, labels, bbox_gt, fill_gt, post, bpost, 0, 1, mask2d,
                         last_layer=model.get_last_layer())
    assert rel(dec, dec_ref) < 8e-5 and rel(rest, rest_ref) < 8e-5
    (rest + (dec * G.to(dec.device)).sum()).backward()
    ref_params = dict(ref.named_parameters())
    scale = max(p.grad.abs().max().item() for p in ref_params.values() if p.grad is not None)
    worst, compared = ("", 0.0), set()
    for name, p in model.named_parameters():
        rg = ref_params[name].grad
        if rg is None or p.grad is None:
            assert (rg is None or rg.abs().max().item() == 0.0) and (p.grad is None or p.grad.abs().max().item() == 0.0), name
            continue
        compared.add(name.split(".")[0])
        e = (p.grad.detach().cpu().double() - rg.double()).abs().max().item() / max(rg.abs().max().item(), 1e-3 * scale)
        if e > worst[1]:
            worst = (name, e)
    print("headline gradients, L1 sign from the oracle: worst %s %.3e" % worst)
    assert {"encoder", "decoder", "quant_conv_obj", "post_quant_conv", "pose_encoder", "pose_decoder"} <= compared, compared
    assert ref_params["decoder.conv_in.weight"].grad.abs().max().item() > 0 and ref_params["encoder.conv_in.weight"].grad.abs().max().item() > 0
    assert worst[1] < 1.2e-4, "param grad %s rel err %.3e" % worst      # <= 4.3x the measured 2.8e-5


def check_step(model, ref, global_step, height, latent_hw, tol_out=1e-3, tol_grad=5e-3):
    from odvae_amd import synthetic
    model.train(); ref.train()
    model.loss.log_exact_g_loss = True
    model._global_step = ref.global_step = global_step
    batch = synthetic.make_batch(2, height, seed=5)
    noise = synthetic.make_noise(2, latent_hw, dropout_p=0.7, seed=6)
    model.injected_noise = noise
    loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, 0)
    loss_ref, log_ref, aux = ref.training_step(batch, 0, noise)
    logs = model.logged_metrics
    assert rel(loss, loss_ref) < tol_out, (loss.item(), loss_ref.item())
    for key in ("kl_loss_obj", "nll_loss", "rec_loss", "pose_loss", "class_loss", "bbox_loss", "kl_loss_bbox", "fill_factor_loss",
                "g_loss"):   # g_loss with the discriminator off: -mean D(x_rec), evaluated without a graph (contperceptual.py:285-292)
        assert rel(logs["train/" + key], log_ref["train/" + key]) < tol_out, (key, rel(logs["train/" + key], log_ref["train/" + key]))
    # latent / reconstruction
    dec_obj, dec_pose, post, _ = model.forward(model._rescale(batch["patch"].to("cuda:0")))
    assert rel(post.parameters, aux["posterior"].parameters) < tol_out
    assert rel(dec_obj, aux["dec_obj"]) < tol_out
    assert rel(dec_pose, aux["dec_pose"]) < tol_out
    loss.backward()
    loss_ref.backward()
    ref_params = dict(ref.named_parameters())
    scale = max(p.grad.abs().max().item() for p in ref_params.values() if p.grad is not None)
    worst = ("", 0.0)
    compared = set()
    for name, p in model.named_parameters():
        rg = ref_params[name].grad
        if rg is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, name
            continue
        if p.grad is None:   # disc_factor = 0: the reference multiplies D's output by an exact 0, here D is not evaluated
            assert name.startswith("loss.discriminator") and rg.abs().max().item() == 0.0, name
            continue
        compared.add(name.split(".")[0])
        e = (p.grad.detach().cpu().double() - rg.double()).abs().max().item() / max(rg.abs().max().item(), 1e-3 * scale)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] < tol_grad, "param grad %s rel err %.3e" % worst
    if global_step > 0:   # the reconstruction terms reach the decoder and, through z, the encoder
        assert {"encoder", "decoder", "quant_conv_obj", "post_quant_conv"} <= compared, compared
        assert ref_params["decoder.conv_in.weight"].grad.abs().max().item() > 0


@pytest.mark.parametrize("ch,height,latent_hw", [(32, 64, 4), (None, 256, 16)], ids=["narrow-64", "headline-256"])
def test_gan_lpips_training_batch_matches_oracle(hip_lib, ch, height, latent_hw):
    """BASELINE.json configs[3]: PatchGAN + LPIPS-style loss, both optimizers (generator step with the adaptive weight
    from two partial backward passes, then the discriminator step), two batches; in miniature and ("headline-256") on
    the benchmark's own network and resolution."""
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    from oracle.autoencoder import train_batch
    model, ref = build_pair(perceptual_weight=1.0, disc_factor=1.0, ch=ch, latent_hw=latent_hw)
    model.train(); ref.train()
    # the product's perceptual net pins itself to eval mode (gan.LPIPSStyle.train, DESIGN.md 7 "deliberate deviations"); the oracle,
    # like [UPSTREAM] taming LPIPS, would run its Dropout(0.5) lin layers stochastically after .train() -- parity is stated against
    # its eval-mode (expected) value, so the oracle's metric is switched back by hand
    assert not model.loss.perceptual_loss.training and ref.loss.perceptual_loss.training
    ref.loss.perceptual_loss.eval()
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0, 1))
    ref_opts = ref.configure_optimizers()
    for step in range(2):
        batch = synthetic.make_batch(2, height, seed=300 + step)
        noises = {0: synthetic.make_noise(2, latent_hw, dropout_p=0.7, seed=400 + 2 * step),
                  1: synthetic.make_noise(2, latent_hw, dropout_p=0.7, seed=401 + 2 * step)}
        got = []
        for idx in (0, 1):   # one optimizer at a time so each forward sees its own injected noise
            model.injected_noise = noises[idx]
            trainer.optimizer_indices = (idx,)
            got.append(trainer.training_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, step)[0])
            if idx == 0:
                logs = dict(model.logged_metrics)
        out = train_batch(ref, ref_opts, batch, noises, optimizer_indices=(0, 1), clip=1.0)
        for idx in (0, 1):
            a, b = got[idx].item(), out[idx][0].item()
            assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (step, idx, a, b)
        for key in ("g_loss", "d_weight", "nll_loss", "rec_loss"):
            assert rel(logs["train/" + key], out[0][1]["train/" + key]) < 5e-3, (step, key, logs["train/" + key], out[0][1]["train/" + key])
    assert model.global_step == 4 and ref.global_step == 4
    ref_sd = ref.state_dict()
    # Adam moves a weight by ~lr*sign(g) per step, so a gradient that is zero up to rounding (BatchNorm biases start
    # at 0, attention k.bias) can differ by up to 2*lr per step between two correct implementations
    lr, steps = model.learning_rate, 2
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32 and k.startswith(("decoder", "loss.discriminator")):
            diff = (v.detach().cpu().double() - ref_sd[k].double()).abs().max().item()
            assert diff <= 2.2 * lr * steps + 5e-3 * ref_sd[k].abs().max().item(), (k, diff)


@pytest.mark.parametrize("ch,height,latent_hw", [(32, 64, 4), (None, 256, 16)], ids=["narrow-64", "headline-256"])
def test_three_step_loss_curve_matches_oracle(hip_lib, ch, height, latent_hw):
    """Three optimizer steps (clip + Adam) on both sides; "headline-256" is the benchmark's own network and resolution."""
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    from oracle.autoencoder import train_batch
    model, ref = build_pair(ch=ch, latent_hw=latent_hw)
    model.train(); ref.train()
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,))
    ref_opts = ref.configure_optimizers()
    curve, curve_ref = [], []
    for step in range(3):
        batch = synthetic.make_batch(2, height, seed=100 + step)
        noise = synthetic.make_noise(2, latent_hw, dropout_p=0.7, seed=200 + step)
        model.injected_noise = noise
        losses = trainer.training_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, step)
        out = train_batch(ref, ref_opts, batch, {0: noise}, optimizer_indices=(0,), clip=1.0)
        curve.append(losses[0].item()); curve_ref.append(out[0][0].item())
    for a, b in zip(curve, curve_ref):
        assert abs(a - b) <= 4e-6 * max(1.0, abs(b)), (curve, curve_ref)      # measured 1.5e-7 / 9e-7 (profiles/r04_parity_margins.json)
    assert model.global_step == 3 and ref.global_step == 3
    # weights after three Adam steps
    ref_sd = ref.state_dict()
    # Adam's update is ~lr*sign(g): a gradient that is zero up to rounding (attention k.bias) may move a weight by
    # up to 2*lr per step differently in two correct implementations
    lr, steps = model.learning_rate, 3
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32 and k.startswith(("encoder", "decoder", "quant", "post_quant", "pose_")):
            diff = (v.detach().cpu().double() - ref_sd[k].double()).abs().max().item()
            assert diff <= 2.2 * lr * steps + 5e-3 * ref_sd[k].abs().max().item(), (k, diff)


def test_eight_step_curve_f4x4_vs_f2x2_winograd(hip_lib, monkeypatch):
    """The benchmark's own network (ch = 128) at 128 x 128, B = 4, eight optimizer steps, once with the stride-1 convs on the
    Winograd F(4x4,3x3) kernel (default) and once on F(2x2,3x3) (ODVAE_CONV_WINOGRAD4=0; ten times closer to the direct form per
    layer): same weights, batches and noise.  The loss curves agree to 2e-5 per step (measured 7e-7), the logged reconstruction / KL terms of
    the last step to 1e-3, the weights after eight Adam steps to 2.2 lr per step (a gradient that is zero up to rounding may
    move a weight either way) + 2e-3 of the tensor's largest entry."""
    from odvae_amd import ops, synthetic
    from odvae_amd.config import instantiate_from_config
    from odvae_amd.trainer import Trainer
    runs = {}
    for f4 in (True, False):
        monkeypatch.setattr(ops, "WINOGRAD4", f4)
        torch.manual_seed(23)
        mcfg, cfg = synthetic.model_config(YAML, latent_hw=8)
        model = instantiate_from_config(mcfg)
        model.learning_rate = 12 * cfg.model.base_learning_rate
        model = model.to("cuda:0").train()
        trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,))
        curve = []
        for step in range(8):
            batch = synthetic.make_batch(4, 128, seed=300 + step)
            model.injected_noise = synthetic.make_noise(4, 8, dropout_p=0.7, seed=400 + step)
            curve.append(trainer.training_batch(batch, step)[0].item())
        logs = {k: float(model.logged_metrics["train/" + k]) for k in ("rec_loss", "nll_loss", "kl_loss_obj")}
        runs[f4] = (curve, logs, {k: v.detach().clone() for k, v in model.state_dict().items() if v.dtype == torch.float32}, model.learning_rate)
        assert ops._wino4_ok(128, 128, 128, 128) == f4
    (c4, l4, w4, lr), (c2, l2, w2, _) = runs[True], runs[False]
    print("F(4x4) vs F(2x2) curve: max relative difference %.2e over %s" % (max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(c4, c2)), c2))
    for a, b in zip(c4, c2):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (c4, c2)      # measured 7e-7
    for k in l4:
        assert abs(l4[k] - l2[k]) <= 1e-3 * max(1.0, abs(l2[k])), (k, l4[k], l2[k])
    for k in w4:
        if k.startswith(("encoder", "decoder", "quant", "post_quant")):
            diff = (w4[k] - w2[k]).abs().max().item()
            assert diff <= 2.2 * lr * 8 + 2e-3 * w2[k].abs().max().item(), (k, diff)


def test_config1_golden_curve_at_full_width(hip_lib):
    """BASELINE.json configs[0] (yaml at full width, 64x64, B=2, 10 steps) on the HIP path against the committed fixture
    tests/golden/oracle_config1.npz: same initial weights (the generator's seeded oracle), same batches and noise."""
    import importlib.util
    import numpy as np
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_goldens", os.path.join(gold_dir, "make_oracle_goldens.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gold = np.load(os.path.join(gold_dir, "oracle_config1.npz"))
    ref = gen.build_oracle()                                     # seeded initial weights only; it is not trained here
    model = synthetic.build_model(YAML, batch_size_for_lr=2, latent_hw=4)
    res = model.load_state_dict(ref.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to("cuda:0").train()
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,))
    curve = []
    for step in range(10):
        batch = synthetic.make_batch(2, 64, seed=1000 + step)
        model.injected_noise = synthetic.make_noise(2, 4, dropout_p=0.7, seed=2000 + step)
        curve.append(trainer.training_batch(batch, step)[0].item())
        if step == 0:
            for k in ("kl_loss_obj", "nll_loss", "rec_loss", "pose_loss", "class_loss", "bbox_loss", "kl_loss_bbox"):
                assert abs(float(model.logged_metrics["train/" + k]) - float(gold["log." + k])) <= 1e-3 * max(1.0, abs(float(gold["log." + k]))), k
    assert np.allclose(np.array(curve), gold["curve"], rtol=3e-3), (curve, gold["curve"])


def _step_with_grads(model, batch, noise):
    model.zero_grad(set_to_none=True)
    model._global_step = 1   # > encoder_pretrain_steps (0): the reconstruction / KL terms are in the total, the decoder gets gradients
    model.injected_noise = noise
    loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, 0)
    loss.backward()
    return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def test_decoder_recompute_is_bit_identical(hip_lib):
    """ddconfig.activation_checkpoint (BASELINE.json configs[4]: "activation-checkpointed Decoder") recomputes each
    decoder unit in backward; the kernels are deterministic, so loss and every gradient must be bit-identical."""
    from odvae_amd import synthetic
    plain, _ = build_pair()
    ckpt, _ = build_pair(activation_checkpoint=True)
    ckpt.load_state_dict(plain.state_dict())
    assert ckpt.decoder.activation_checkpoint and not plain.decoder.activation_checkpoint
    plain.train(); ckpt.train()
    batch = synthetic.make_batch(2, 64, seed=5)
    noise = synthetic.make_noise(2, 4, dropout_p=0.7, seed=6)
    l0, g0 = _step_with_grads(plain, batch, noise)
    l1, g1 = _step_with_grads(ckpt, batch, noise)
    assert torch.equal(l0, l1)
    assert g0.keys() == g1.keys() and g0["decoder.conv_in.weight"].abs().max().item() > 0
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("precision", [None, "bf16"], ids=["f32", "bf16"])
def test_decoder_norm_checkpoint_policy_is_bit_identical_and_saves_memory(hip_lib, precision):
    """ddconfig.activation_checkpoint = "norm": the convs of the decoder keep (x, mean, rstd) of the Normalize in front of them instead of its
    normalised + activated output and re-make that tensor with one GroupNorm apply pass in their backward (ops.remake_from_norm).  Same kernels
    on the same values: loss and every gradient bit-identical to the plain model; the peak memory of forward + backward is lower."""
    from odvae_amd import synthetic
    plain, _ = build_pair(ch=64, latent_hw=8)
    norm, _ = build_pair(ch=64, latent_hw=8, activation_checkpoint="norm")
    norm.load_state_dict(plain.state_dict())
    assert norm.decoder.activation_checkpoint == "norm"
    if precision:
        plain.set_precision(precision); norm.set_precision(precision)
    plain.train(); norm.train()
    batch = synthetic.make_batch(4, 128, seed=5)
    noise = synthetic.make_noise(4, 8, dropout_p=0.7, seed=6)
    peaks = []
    outs = []
    for m in (plain, norm):
        torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        outs.append(_step_with_grads(m, batch, noise))
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated() - base)
    (l0, g0), (l1, g1) = outs
    assert torch.equal(l0, l1)
    assert g0.keys() == g1.keys() and g0["decoder.conv_in.weight"].abs().max().item() > 0
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    assert peaks[1] < 0.93 * peaks[0], peaks        # (the encoder's activations and the losses are untouched: the step's peak falls by the decoder's share)


def test_config5_geometry_512_matches_oracle(hip_lib):
    """BASELINE.json configs[4] geometry in fp32: 512x512 crop, z = 32x32x16, checkpointed decoder, B=1, width-reduced
    (ch=32).  The attention blocks see 128x128 = 16384 tokens (a 1 GiB score matrix per block)."""
    from odvae_amd import synthetic
    model, ref = build_pair(latent_hw=32, activation_checkpoint=True)
    model.train(); ref.train()
    batch = synthetic.make_batch(1, 512, seed=7)
    noise = synthetic.make_noise(1, 32, dropout_p=0.7, seed=8)
    loss, grads = _step_with_grads(model, batch, noise)
    assert grads["decoder.conv_in.weight"].abs().max().item() > 0
    ref.global_step = 1
    loss_ref, log_ref, aux = ref.training_step(batch, 0, noise)
    assert rel(loss, loss_ref) < 1e-3, (loss.item(), loss_ref.item())
    for key in ("kl_loss_obj", "nll_loss", "rec_loss"):
        assert rel(model.logged_metrics["train/" + key], log_ref["train/" + key]) < 1e-3, key
    loss_ref.backward()
    ref_params = dict(ref.named_parameters())
    scale = max(p.grad.abs().max().item() for p in ref_params.values() if p.grad is not None)
    for name, g in grads.items():
        rg = ref_params[name].grad
        if rg is None:
            assert g.abs().max().item() == 0.0, name
            continue
        e = (g.cpu().double() - rg.double()).abs().max().item() / max(rg.abs().max().item(), 1e-3 * scale)
        assert e < 5e-3, "param grad %s rel err %.3e" % (name, e)


@pytest.mark.parametrize("gan,precision", [(False, None), (True, None), (False, "bf16")], ids=["rec+KL", "gan+lpips", "rec+KL-bf16"])
def test_training_batch_never_synchronises_the_host(hip_lib, gan, precision):
    """A training batch (both optimizers with the GAN on) issues into the stream without a single host synchronisation
    (torch.cuda.set_sync_debug_mode("error") raises at the first one): the host may run a whole step ahead of the device.  Round 3 found
    one with it -- the pose encoder's coordinate grid, a pageable host tensor, was copied to the device in every forward."""
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    kw = dict(perceptual_weight=1.0, disc_factor=1.0, disc_start=0) if gan else {}
    torch.manual_seed(23)
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32, **kw).to("cuda:0").train()
    model._global_step = 1
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0, 1) if gan else (0,), precision=precision)
    batch = synthetic.make_batch(2, 64, seed=23)
    batch = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in batch.items()}

    def step(i):
        b = dict(batch)
        b["pose_6d"] = batch["pose_6d"].clone()
        return trainer.training_batch(b, i)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(2):          # first steps build packs, workspaces and the staging ring
            step(i)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            losses = step(2)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    assert all(torch.isfinite(l).all() for l in losses)


@pytest.mark.parametrize("gan,precision,ckpt", [(False, None, False), (True, None, False), (False, "bf16", False), (False, "bf16", True)],
                         ids=["rec+KL", "gan+lpips", "rec+KL-bf16", "rec+KL-bf16-ckpt"])
def test_steady_state_steps_hold_no_more_device_memory(hip_lib, gan, precision, ckpt):
    """Nothing a training step allocates outlives it: after three warm-up steps the bytes the caching allocator holds live
    (`allocated_bytes.all.current` at the step boundary), its live-block count and the weight-pack cache's entry count stay put over six more steps.
    Round 4 found the bf16 path leaking 14 weight packs per step (AttnBlock's torch.cat'ed q/k/v weight is a new tensor every
    forward; the pack cache kept each one's packs until 4 096 entries had piled up): 6-10 MB and one to three hipMalloc calls -- device
    synchronisations -- per step, i.e. the 79 ms steps with 96-230 ms outliers of the bf16 side run."""
    from odvae_amd import ops, synthetic
    from odvae_amd.trainer import Trainer
    kw = dict(perceptual_weight=1.0, disc_factor=1.0, disc_start=0) if gan else {}
    torch.manual_seed(23)
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32, **kw).to("cuda:0").train()
    model.decoder.activation_checkpoint = ckpt
    model._global_step = 1
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0, 1) if gan else (0,), precision=precision)
    batch = synthetic.make_batch(2, 64, seed=23)
    batch = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in batch.items()}

    def step(i):
        b = dict(batch)
        b["pose_6d"] = batch["pose_6d"].clone()
        trainer.training_batch(b, i)

    def held():
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats()
        return st["allocated_bytes.all.current"], st["allocation.all.current"], len(ops.PACK_CACHE.store)
    import gc
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(3):
            step(i)
        gc.collect()
        base = held()
        seen = []
        for i in range(3, 9):
            step(i)
            gc.collect()
            seen.append(held())
    assert all(s == base for s in seen), (base, seen)

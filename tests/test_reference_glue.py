"""The ORACLE against outputs of the REFERENCE's own Python (CPU only).

tests/golden/reference_glue.npz was written by tests/golden/make_reference_goldens.py, which imports the reference's
src/util/distributions.py, src/modules/losses/contperceptual.py and src/models/autoencoder.py UNMODIFIED from /root/reference (with
stand-ins for the absent third-party names that point at the oracle's restatement of the upstream layers) and runs
DiagonalGaussianDistribution.kl, PoseAutoencoder._get_dropout_prob, PoseLoss.forward (both optimizer indices, the three global_step
regimes, masked class id 1, an all-masked batch) and PoseAutoencoder.training_step / validation_step.  Here the oracle's own
restatement of those 860 lines (oracle/distributions.py, oracle/losses.py, oracle/autoencoder.py) must give the same numbers: both sides
run the same torch CPU ops underneath, so the tolerance is summation-order noise (1e-5 relative), not a modelling tolerance.
tests/test_reference_glue_gpu.py replays the same cases on the HIP path.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)
from reference_cases import LABELS, LOSS_CASES, LOSS_KW, digest_of, loss_inputs, stats_table, step_batch  # noqa: E402

YAML = os.path.join(GOLD, "autoencoder_kl_16x16x16.yaml")
RTOL = 1e-5


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "reference_glue.npz"))


def close(a, b, rtol=RTOL, atol=0.0):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(float(np.abs(b).max()) if b.size else 0.0, 1e-30)
    return a.shape == b.shape and float(np.abs(a - b).max() if b.size else 0.0) <= rtol * scale + atol


def check_digest(gold, key, t, rtol=RTOL, atol=0.0):
    norm, samples = digest_of(t)
    gn = float(gold[key + ".norm"])
    assert abs(norm - gn) <= rtol * max(gn, 1e-30) + atol, (key, norm, gn)
    # elementwise: relative to the largest entry of the whole tensor (the norm bounds it), not of the sampled subset
    assert samples.shape == gold[key + ".samples"].shape, key
    scale = max(float(np.abs(gold[key + ".samples"]).max()), 1e-30)
    assert float(np.abs(samples.astype(np.float64) - gold[key + ".samples"]).max()) <= rtol * scale + atol, key


def check_param_grads(gold, pre, named_params, rtol=RTOL, floor=0.0):
    """Gradient digests: L2 norm and first 64 entries per parameter.  `floor` (a fraction of the LARGEST gradient norm of the case) is the
    scale below which a gradient is compared absolutely: a conv bias in front of a GroupNorm has a mathematically (near-)zero gradient, and
    what is left of it is cancellation noise of either side's f32 sums (the same rule as tests/test_model_gpu.py)."""
    named_params = list(named_params)
    top = max([float(gold["%s.gnorm.%s" % (pre, k)]) for k, _ in named_params if "%s.gnorm.%s" % (pre, k) in gold.files] or [0.0])
    seen = 0
    for k, p in named_params:
        nk = "%s.gnorm.%s" % (pre, k)
        if nk not in gold.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        gn = float(gold[nk])
        scale = max(gn, floor * top, 1e-12)
        assert p.grad is not None, k
        assert abs(float(p.grad.double().norm()) - gn) <= rtol * scale, (k, float(p.grad.double().norm()), gn)
        head = gold["%s.ghead.%s" % (pre, k)]
        got = p.grad.detach().cpu().reshape(-1)[:64].numpy()
        assert float(np.abs(got.astype(np.float64) - head).max()) <= rtol * scale, k
        seen += 1
    return seen


def test_kl_matches_the_reference(gold):
    from oracle.distributions import DiagonalGaussianDistribution as D
    assert close(D(torch.from_numpy(gold["kl.self.params"])).kl(), gold["kl.self.out"])
    a, b = torch.from_numpy(gold["kl.other.params_self"]), torch.from_numpy(gold["kl.other.params_other"])
    out = D(a).kl(D(b))
    assert out.shape == (8,)   # QUIRK kept: [8,1] against [1,8] broadcasts to [8,8]; entry i sums the cross terms over all prior dims
    assert close(out, gold["kl.other.out"])
    assert close(D(a, deterministic=True).kl(), gold["kl.deterministic.out"])


def _oracle_model(phase):
    from odvae_amd import synthetic
    from oracle.autoencoder import PoseAutoencoder
    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32, phase=phase, perceptual_weight=1.0, disc_factor=1.0, disc_start=0)
    p = mcfg.params.to_container()
    return PoseAutoencoder(p["ddconfig"], dict(p["lossconfig"]["params"]), p["embed_dim"], p["pose_decoder_config"]["params"],
                           p["pose_encoder_config"]["params"], feat_dims=p["feat_dims"], dropout_prob_init=p["dropout_prob_init"],
                           dropout_prob_final=p["dropout_prob_final"], dropout_warmup_steps=p["dropout_warmup_steps"],
                           pose_conditioned_generation_steps=p["pose_conditioned_generation_steps"],
                           add_noise_to_z_obj=p["add_noise_to_z_obj"], train_on_yaw=p["train_on_yaw"])


def test_dropout_schedule_matches_the_reference(gold):
    """autoencoder.py:184-206 with the yaml's own thresholds (30 000 / 45 000 / 45 000, 1.0 -> 0.7), at every boundary +- 1."""
    model = _oracle_model("asis")
    E, G, W, p0, p1 = gold["dropout.params"]
    assert (model.encoder_pretrain_steps, model.pose_conditioned_generation_steps, model.dropout_warmup_steps) == (E, G, W)
    for s, want in zip(gold["dropout.steps"], gold["dropout.probs"]):
        model.global_step = int(s)
        assert model._get_dropout_prob() == pytest.approx(float(want), rel=0, abs=1e-15), int(s)
    # QUIRK kept: the ramp subtracts encoder_pretrain_steps only (:200), so it STARTS below dropout_prob_init at step E + G
    k = list(gold["dropout.steps"]).index(int(E + G))
    assert gold["dropout.probs"][k] == pytest.approx(p0 - (p0 - p1) * G / W)


def _oracle_loss():
    from odvae_amd.synthetic import fill_state_procedural
    from oracle.losses import PoseLoss
    torch.manual_seed(5)
    loss = PoseLoss(dataset_stats=stats_table(), **LOSS_KW)
    fill_state_procedural(loss, seed=31)
    with torch.no_grad():
        loss.logvar.fill_(0.3)
    loss.train()
    loss.perceptual_loss.eval()
    return loss


@pytest.mark.parametrize("case", range(len(LOSS_CASES)), ids=[c[0] for c in LOSS_CASES])
def test_pose_loss_forward_matches_the_reference(gold, case):
    """contperceptual.py:214-375 run by the reference, replayed on oracle/losses.py: total, every logged term both sides hold, the
    gradients at every input of the loss and at the discriminator's parameters."""
    from odvae_amd.synthetic import fill_state_procedural
    from oracle.distributions import DiagonalGaussianDistribution as D
    name, gs, cls = LOSS_CASES[case]
    assert int(gold["loss.%s.global_step" % name]) == gs
    loss = _oracle_loss()
    assert sorted(loss.state_dict().keys()) == list(gold["loss.state_keys"])
    d = loss_inputs(200 + case, cls)
    for opt in (0, 1):
        pre = "loss.%s.opt%d" % (name, opt)
        if pre + ".raises" in gold.files:
            # the reference cannot TRAIN on a batch of nothing but the masked class id once the adaptive weight is on (:294-299: autograd.grad of
            # a graph-less 0, then `assert not self.training`); neither can the oracle, which restates exactly that
            assert opt == 0 and set(cls) == {1}
            leaves = {k: d[k].clone().requires_grad_(True) for k in ("feat", "last_w", "last_b", "dec_pose", "moments", "bbox_moments")}
            dec_obj = torch.nn.functional.conv2d(leaves["feat"], leaves["last_w"], leaves["last_b"], padding=1)
            with pytest.raises(AssertionError):
                loss(d["rgb_gt"], None, d["pose_gt"], dec_obj, leaves["dec_pose"], d["class_id"], [LABELS[c] for c in cls], d["bbox_gt"],
                     d["fill_factor_gt"], D(leaves["moments"]), D(leaves["bbox_moments"]), opt, gs, d["mask_2d_bbox"], last_layer=leaves["last_w"])
            continue
        leaves = {k: d[k].clone().requires_grad_(True) for k in ("feat", "last_w", "last_b", "dec_pose", "moments", "bbox_moments")}
        dec_obj = torch.nn.functional.conv2d(leaves["feat"], leaves["last_w"], leaves["last_b"], padding=1)
        fill_state_procedural(loss.discriminator, seed=31)
        for p in loss.parameters():
            p.grad = None
        out, log = loss(d["rgb_gt"], None, d["pose_gt"], dec_obj, leaves["dec_pose"], d["class_id"], [LABELS[c] for c in cls], d["bbox_gt"],
                        d["fill_factor_gt"], D(leaves["moments"]), D(leaves["bbox_moments"]), opt, gs, d["mask_2d_bbox"],
                        last_layer=leaves["last_w"], split="train")
        assert close(out, gold[pre + ".loss"]), (pre, float(out), float(gold[pre + ".loss"]))
        compared = 0
        for k, v in log.items():
            assert pre + ".log." + k in gold.files, k       # the oracle logs nothing the reference does not
            assert close(torch.as_tensor(v).float(), gold[pre + ".log." + k], atol=1e-12), (pre, k, float(v), float(gold[pre + ".log." + k]))
            compared += 1
        assert compared >= (11 if opt == 0 else 3)
        if pre + ".gnorm.discriminator.main.0.weight" in gold.files or any(f.startswith(pre + ".grad.") for f in gold.files):
            out.backward()
            for k, t in leaves.items():
                if pre + ".grad." + k + ".norm" in gold.files:
                    check_digest(gold, pre + ".grad." + k, t.grad, atol=1e-12)
                else:
                    assert t.grad is None or float(t.grad.abs().max()) == 0.0, (pre, k)
            check_param_grads(gold, pre, loss.named_parameters())


def test_training_and_validation_step_match_the_reference(gold):
    """autoencoder.py:295-330 (training_step, optimizer 0 and 1) and :332-363 (validation_step) run by the reference's PoseAutoencoder on
    the oracle's Encoder / Decoder, replayed on oracle/autoencoder.py with the recorded host draws: loss, logged terms, every parameter's
    gradient."""
    from odvae_amd.synthetic import fill_state_procedural
    model = _oracle_model("vae")
    assert sorted(model.state_dict().keys()) == list(gold["step.state_keys"])
    model.learning_rate = 12 * 4.5e-6
    opts = model.configure_optimizers()
    assert sum(q.numel() for g in opts[0].param_groups for q in g["params"]) == int(gold["step.opt0.nparams"])
    assert sum(q.numel() for g in opts[1].param_groups for q in g["params"]) == int(gold["step.opt1.nparams"])
    batch = step_batch()
    model.train()
    model.loss.perceptual_loss.eval()
    for opt_idx in (0, 1):
        pre = "step.train.opt%d" % opt_idx
        fill_state_procedural(model, seed=23)
        model.global_step = int(gold[pre + ".global_step"])
        noise = {k: torch.from_numpy(gold[pre + ".noise." + k]) for k in ("posterior_eps", "dropout_mask", "z_noise", "bbox_eps")}
        for q in model.parameters():
            q.grad = None
        out, log, _ = model.training_step(batch, opt_idx, noise)
        assert model.dropout_prob == pytest.approx(float(gold[pre + ".dropout_prob"]))
        assert close(out, gold[pre + ".loss"]), (float(out), float(gold[pre + ".loss"]))
        assert close(out, gold[pre + ".log." + ("aeloss" if opt_idx == 0 else "discloss")])
        for k, v in log.items():
            assert close(torch.as_tensor(v).float(), gold[pre + ".log." + k], atol=1e-12), (k, float(v), float(gold[pre + ".log." + k]))
        out.backward()
        # whole-network backward in f32 on either side (thread count changes the conv / GEMM summation order): 1e-4 of max(own norm, 1e-3 of the largest)
        n = check_param_grads(gold, pre, model.named_parameters(), rtol=1e-4, floor=1e-3)
        assert n > (200 if opt_idx == 0 else 10), n
    fill_state_procedural(model, seed=23)
    model.eval()
    model.global_step = int(gold["step.val.global_step"])
    noise = {k: torch.from_numpy(gold["step.val.noise." + k]) for k in ("posterior_eps", "dropout_mask", "z_noise", "bbox_eps")}
    with torch.no_grad():
        logs = model.validation_step(batch, noise)
    assert "val/rec_loss" in logs and "val/disc_loss" in logs
    for k, v in logs.items():
        assert close(torch.as_tensor(v).float(), gold["step.val.log." + k], atol=1e-12), (k, float(v), float(gold["step.val.log." + k]))


def test_headline_network_step_matches_the_reference(gold):
    """BASELINE.json configs[1]'s own network (ch = 128, 256 x 256, 4 096 attention tokens, rec+KL only) at B = 2: the reference's training_step run on
    the oracle's Encoder / Decoder against the oracle's own training_step -- loss, logged terms, every parameter gradient (1e-4 of max(own norm,
    1e-3 of the largest): f32 sums in a different thread count)."""
    from odvae_amd import synthetic
    from odvae_amd.synthetic import fill_state_procedural
    from oracle.autoencoder import PoseAutoencoder
    mcfg, _ = synthetic.model_config(YAML, latent_hw=16, ch=None, phase="vae", perceptual_weight=0.0, disc_factor=0.0, disc_start=0)
    p = mcfg.params.to_container()
    model = PoseAutoencoder(p["ddconfig"], dict(p["lossconfig"]["params"]), p["embed_dim"], p["pose_decoder_config"]["params"],
                            p["pose_encoder_config"]["params"], feat_dims=p.get("feat_dims", [16, 16, 16]), dropout_prob_init=p["dropout_prob_init"],
                            dropout_prob_final=p["dropout_prob_final"], dropout_warmup_steps=p["dropout_warmup_steps"],
                            pose_conditioned_generation_steps=p["pose_conditioned_generation_steps"],
                            add_noise_to_z_obj=p["add_noise_to_z_obj"], train_on_yaw=p["train_on_yaw"])
    assert sum(q.numel() for q in model.parameters()) == int(gold["headline.nparams"])
    fill_state_procedural(model, seed=23)
    model.train()
    model.global_step = 1
    pre = "headline.train.opt0"
    noise = {k: torch.from_numpy(gold[pre + ".noise." + k]) for k in ("posterior_eps", "dropout_mask", "z_noise", "bbox_eps")}
    out, log, _ = model.training_step(synthetic.make_batch(2, 256, seed=5), 0, noise)
    assert close(out, gold[pre + ".loss"]), (float(out), float(gold[pre + ".loss"]))
    for k, v in log.items():
        assert close(torch.as_tensor(v).float(), gold[pre + ".log." + k], atol=1e-12), (k, float(v), float(gold[pre + ".log." + k]))
    out.backward()
    n = check_param_grads(gold, pre, model.named_parameters(), rtol=1e-4, floor=1e-3)
    assert n > 250, n

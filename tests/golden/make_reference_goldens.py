"""Generates tests/golden/reference_glue.npz by RUNNING the reference's own Python for the part of the hot path that lives in
/root/reference:

    src/util/distributions.py              DiagonalGaussianDistribution.kl (:10-41)
    src/modules/losses/contperceptual.py   PoseLoss.__init__ / forward and every helper it calls (:26-375)
    src/models/autoencoder.py              PoseAutoencoder.__init__, _get_dropout_prob (:184-206), forward (:208-257),
                                           training_step (:295-330), validation_step (:332-363), configure_optimizers (:365-377)

Those three files import third-party packages that are absent from this image and un-vendored in the reference (`ldm`, `taming`,
`mmdet`, `pytorch_lightning`: SURVEY.md 8(c)).  This script registers minimal `sys.modules` entries for exactly the names the three
files import and points them at the ORACLE's restatement of the upstream pieces (oracle/ldm_model.py Encoder / Decoder,
oracle/losses.py NLayerDiscriminator / LPIPSStyle / hinge_d_loss / adopt_weight / focal loss, oracle/distributions.py base class
members) plus three upstream method bodies restated below ([UPSTREAM] AutoencoderKL.get_input / decode / get_last_layer and the
LPIPSWithDiscriminator constructor) and a 15-line LightningModule stand-in.  The reference files themselves are imported UNMODIFIED
from /root/reference.

What that pins and what it does not: the ~860 lines of glue that ARE in /root/reference (mask / phase / global_step logic, the order
and weights of the loss terms, the log dict, the kl(other) broadcast, the dropout schedule, the training / validation step plumbing) are
executed, not transcribed, so a transcription slip in oracle/losses.py or oracle/autoencoder.py shows as a mismatch with these
fixtures.  The upstream arithmetic underneath (conv stack, attention, PatchGAN, LPIPS structure) is still the oracle's restatement:
"parity unpinned" stays for that layer.

Build container only: /root/reference does not travel; the fixtures (inputs + expected outputs, no source text) do.

    python tests/golden/make_reference_goldens.py
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, ROOT]
REFERENCE = "/root/reference"
OUT = os.path.join(HERE, "reference_glue.npz")
YAML = os.path.join(HERE, "autoencoder_kl_16x16x16.yaml")


# ---------------------------------------------------------------------------------------------------------------------------------
# stand-ins for the absent third-party names (only what the three reference files import)
# ---------------------------------------------------------------------------------------------------------------------------------
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            _module(parent)
        setattr(sys.modules[parent], child, m)
    if not hasattr(m, "__path__"):
        m.__path__ = []   # behaves as a package for `from a.b.c import d`
    return m


DRAWS = []   # every tensor the base distribution's sample() draws, in order (the reference samples with torch.randn on the host)


def install_standins():
    from oracle import distributions as odist
    from oracle import ldm_model as omodel
    from oracle import losses as olosses

    class LightningModule(nn.Module):
        """What the reference uses of pytorch_lightning.LightningModule [PL-1.9]: global_step, device, log, log_dict."""

        def __init__(self):
            super().__init__()
            self._gs, self.logged = 0, {}

        global_step = property(lambda self: self._gs, lambda self, v: setattr(self, "_gs", v))
        device = property(lambda self: torch.device("cpu"))

        def log(self, name, value, **kw):
            self.logged[name] = value

        def log_dict(self, d, **kw):
            self.logged.update(d)

    _module("pytorch_lightning", LightningModule=LightningModule)

    class LDMDiagonalGaussianDistribution(object):
        """[UPSTREAM] ldm/modules/distributions/distributions.py base class, members as in oracle/distributions.py; sample() draws with
        torch.randn on the host like upstream and records the draw."""

        def __init__(self, parameters, deterministic=False):
            odist.DiagonalGaussianDistribution.__init__(self, parameters, deterministic)

        def sample(self):
            eps = torch.randn(self.mean.shape)
            DRAWS.append(eps.clone())
            return self.mean + self.std * eps

        def mode(self):
            return self.mean

    _module("ldm.modules.distributions.distributions", DiagonalGaussianDistribution=LDMDiagonalGaussianDistribution)
    _module("ldm.modules.diffusionmodules.model", Encoder=omodel.Encoder, Decoder=omodel.Decoder)

    def get_obj_from_str(string):
        import importlib
        module, cls = string.rsplit(".", 1)
        return getattr(importlib.import_module(module), cls)

    def instantiate_from_config(config):   # [UPSTREAM] ldm/util.py
        if "target" not in config:
            raise KeyError("Expected key `target` to instantiate.")
        return get_obj_from_str(config["target"])(**config.get("params", dict()))

    _module("ldm.util", instantiate_from_config=instantiate_from_config, get_obj_from_str=get_obj_from_str)

    class AutoencoderKL(LightningModule):
        """[UPSTREAM] ldm/models/autoencoder.py AutoencoderKL: the three methods PoseAutoencoder inherits and calls."""

        def get_input(self, batch, k):
            x = batch[k]
            if len(x.shape) == 3:
                x = x[..., None]
            return x.permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format).float()

        def decode(self, z):
            return self.decoder(self.post_quant_conv(z))

        def get_last_layer(self):
            return self.decoder.conv_out.weight

    _module("ldm.models.autoencoder", AutoencoderKL=AutoencoderKL)

    class LPIPSWithDiscriminator(nn.Module):
        """[UPSTREAM] ldm/modules/losses/contperceptual.py constructor + calculate_adaptive_weight, on the oracle's pieces."""

        def __init__(self, disc_start, logvar_init=0.0, kl_weight=1.0, pixelloss_weight=1.0, disc_num_layers=3, disc_in_channels=3,
                     disc_factor=1.0, disc_weight=1.0, perceptual_weight=1.0, use_actnorm=False, disc_conditional=False, disc_loss="hinge"):
            super().__init__()
            assert disc_loss in ["hinge", "vanilla"]
            self.kl_weight, self.pixel_weight = kl_weight, pixelloss_weight
            self.perceptual_loss = olosses.LPIPSStyle().eval()
            self.perceptual_weight = perceptual_weight
            self.logvar = nn.Parameter(torch.ones(size=()) * logvar_init)
            self.discriminator = olosses.NLayerDiscriminator(input_nc=disc_in_channels, n_layers=disc_num_layers,
                                                             use_actnorm=use_actnorm).apply(olosses.weights_init)
            self.discriminator_iter_start = disc_start
            self.disc_loss = olosses.hinge_d_loss
            self.disc_factor, self.discriminator_weight, self.disc_conditional = disc_factor, disc_weight, disc_conditional

        def calculate_adaptive_weight(self, nll_loss, g_loss, last_layer=None):
            return olosses.PoseLoss.calculate_adaptive_weight(self, nll_loss, g_loss, last_layer)

    _module("ldm.modules.losses.contperceptual", LPIPSWithDiscriminator=LPIPSWithDiscriminator)
    _module("taming.modules.losses.vqperceptual", adopt_weight=olosses.adopt_weight)

    class FocalLoss(nn.Module):
        """mmdet 3.3.0 FocalLoss() defaults on class-index targets (use_sigmoid, gamma 2, alpha 0.25, reduction mean)."""

        def forward(self, pred, target):
            return olosses.sigmoid_focal_loss_mean(pred, target)

    _module("mmdet.models.losses.focal_loss", FocalLoss=FocalLoss)


def import_reference():
    """The reference's `src` package, unmodified, ahead of this repository's own `src` shims on sys.path."""
    for name in [n for n in sys.modules if n == "src" or n.startswith("src.")]:
        del sys.modules[name]
    sys.path.insert(0, REFERENCE)
    import src.models.autoencoder as ref_ae
    import src.modules.losses.contperceptual as ref_loss
    import src.util.distributions as ref_dist
    for m in (ref_ae, ref_loss, ref_dist):
        assert m.__file__.startswith(REFERENCE + "/"), m.__file__
    return ref_ae, ref_loss, ref_dist


# ---------------------------------------------------------------------------------------------------------------------------------
def np_(t):
    return t.detach().cpu().numpy().copy() if torch.is_tensor(t) else np.asarray(t)


def grad_digest(arrays, prefix, named_params):
    """Per parameter: the gradient's L2 norm (f64) and its first 64 entries -- enough to catch a wrong gradient, small enough to commit."""
    for k, p in named_params:
        if p.grad is None:
            continue
        arrays["%s.gnorm.%s" % (prefix, k)] = np.float64(p.grad.double().norm().item())
        arrays["%s.ghead.%s" % (prefix, k)] = np_(p.grad.reshape(-1)[:64])


def case_kl(arrays, ref_dist):
    g = torch.Generator().manual_seed(101)
    p4 = torch.randn(3, 8, 4, 4, generator=g)
    arrays["kl.self.params"] = np_(p4)
    arrays["kl.self.out"] = np_(ref_dist.DiagonalGaussianDistribution(p4).kl())
    # kl(other) as compute_pose_kl_loss calls it (contperceptual.py:200-203): [8,2] moments against a prior built from [8,2] moments
    a, b = torch.randn(8, 2, generator=g), torch.randn(8, 2, generator=g)
    arrays["kl.other.params_self"], arrays["kl.other.params_other"] = np_(a), np_(b)
    arrays["kl.other.out"] = np_(ref_dist.DiagonalGaussianDistribution(a).kl(ref_dist.DiagonalGaussianDistribution(b)))
    arrays["kl.deterministic.out"] = np_(ref_dist.DiagonalGaussianDistribution(a, deterministic=True).kl())


from reference_cases import LABELS, LOSS_CASES, LOSS_KW, digest, loss_inputs, step_batch, stats_table  # noqa: E402


def stats_pickle():
    stats = stats_table()
    f = tempfile.NamedTemporaryFile(suffix=".pkl", delete=False)
    pickle.dump(stats, f)
    f.close()
    return f.name, stats


def case_loss(arrays, ref_loss, ref_dist):
    from odvae_amd.synthetic import fill_state_procedural
    path, stats = stats_pickle()
    torch.manual_seed(5)
    loss = ref_loss.PoseLoss(dataset_stats_path=path, **LOSS_KW)
    os.unlink(path)
    fill_state_procedural(loss, seed=31)       # weights follow from (seed, key, shape): the tests re-make them on their own module
    with torch.no_grad():
        loss.logvar.fill_(0.3)
    loss.train()
    loss.perceptual_loss.eval()                # the frozen metric in eval mode, as every consumer of these fixtures runs it
    for ci, (name, gs, cls) in enumerate(LOSS_CASES):
        d = loss_inputs(200 + ci, cls)
        arrays["loss.%s.global_step" % name] = np.int64(gs)     # the inputs follow from reference_cases.loss_inputs(200 + case index, ids)
        for opt in (0, 1):
            leaves = {k: d[k].clone().requires_grad_(True) for k in ("feat", "last_w", "last_b", "dec_pose", "moments", "bbox_moments")}
            dec_obj = torch.nn.functional.conv2d(leaves["feat"], leaves["last_w"], leaves["last_b"], padding=1)
            post = ref_dist.DiagonalGaussianDistribution(leaves["moments"])
            bpost = ref_dist.DiagonalGaussianDistribution(leaves["bbox_moments"])
            loss.zero_grad()
            # BatchNorm running statistics move with every discriminator forward in train mode: same starting buffers for every call
            fill_state_procedural(loss.discriminator, seed=31)
            for k, p in loss.named_parameters():
                p.grad = None
            pre = "loss.%s.opt%d" % (name, opt)
            try:
                out, log = loss(d["rgb_gt"], None, d["pose_gt"], dec_obj, leaves["dec_pose"], d["class_id"], [LABELS[c] for c in cls],
                                d["bbox_gt"], d["fill_factor_gt"], post, bpost, opt, gs, d["mask_2d_bbox"], last_layer=leaves["last_w"],
                                split="train")
            except AssertionError:
                # a TRAINING batch made of the masked class only, past encoder_pretrain_steps with the discriminator on: nll_loss is the
                # graph-less torch.tensor(0.0) of :155, autograd.grad raises, and the handler's `assert not self.training` (:298) fires --
                # the reference cannot train on such a batch.  Recorded as that; the product returns d_weight = 0 instead.
                arrays[pre + ".raises"] = np.array("AssertionError")
                continue
            arrays[pre + ".loss"] = np_(out)
            for k, v in log.items():
                arrays[pre + ".log." + k] = np_(torch.as_tensor(v).float())
            if out.requires_grad:
                out.backward()
                for k, t in leaves.items():
                    if t.grad is not None:
                        digest(arrays, pre + ".grad." + k, t.grad)
                grad_digest(arrays, pre, loss.named_parameters())
    # the discriminator's weights as the procedural fill leaves them are not stored; its keys are (tests assert the same key set)
    arrays["loss.state_keys"] = np.array(sorted(loss.state_dict().keys()))


def model_kwargs(phase, latent_hw=4, ch=32, perceptual_weight=1.0, disc_factor=1.0):
    from odvae_amd import synthetic
    mcfg, _ = synthetic.model_config(YAML, latent_hw=latent_hw, ch=ch, phase=phase, perceptual_weight=perceptual_weight, disc_factor=disc_factor,
                                     disc_start=0)
    p = mcfg.params.to_container()
    stats = p["lossconfig"]["params"].pop("dataset_stats")
    f = tempfile.NamedTemporaryFile(suffix=".pkl", delete=False)
    pickle.dump(stats, f)
    f.close()
    p["lossconfig"]["params"]["dataset_stats_path"] = f.name
    p["lossconfig"]["target"] = "src.modules.losses.contperceptual.PoseLoss"   # the reference's package __init__ is empty
    return p, f.name


def replay_draws(seed, B, latent, p_drop, add_noise=True, perturbed=False):
    """The host RNG stream of ONE reference forward (autoencoder.py:227-244), replayed call by call after the same manual_seed:
    posterior randn, nn.Dropout's mask, Normal(0,1).sample, bbox-posterior randn."""
    torch.manual_seed(seed)
    shape = (B, 16, latent, latent)
    eps = torch.randn(shape)
    mask = torch.nn.functional.dropout(torch.ones(shape), p_drop, True) if p_drop > 0 else torch.ones(shape)
    zn = torch.distributions.normal.Normal(0, 1).sample(shape) if add_noise else torch.zeros(shape)
    beps = torch.randn(B, 8)
    out = {"posterior_eps": eps, "dropout_mask": mask, "z_noise": zn, "bbox_eps": beps}
    if perturbed:      # log_images: _perturbed_pose_forward samples the posterior once more (autoencoder.py:387-390)
        out["posterior_eps_perturbed"] = torch.randn(shape)
    return out


def case_dropout_schedule(arrays, ref_ae):
    p, path = model_kwargs("asis")
    torch.manual_seed(3)
    model = ref_ae.PoseAutoencoder(**p)
    os.unlink(path)
    E, G, W = model.encoder_pretrain_steps, model.pose_conditioned_generation_steps, model.dropout_warmup_steps
    steps = sorted({0, 1, E - 1, E, E + 1, E + G - 1, E + G, E + G + 1, E + G + W // 3, E + G + W - 1, E + G + W, E + G + W + 1, 10 ** 6})
    probs = []
    for s in steps:
        model.global_step = s
        probs.append(model._get_dropout_prob())
    arrays["dropout.steps"], arrays["dropout.probs"] = np.array(steps, dtype=np.int64), np.array(probs, dtype=np.float64)
    arrays["dropout.params"] = np.array([E, G, W, model.dropout_prob_init, model.dropout_prob_final], dtype=np.float64)


def case_steps(arrays, ref_ae):
    """training_step (both optimizer indices) and validation_step of the reference's PoseAutoencoder, width-reduced (ch = 32, 64x64, latent
    4x4x16, pose MLPs n = m = 4), VAE phase, LPIPS-style + PatchGAN terms on; gradients by torch autograd on the reference's own graph."""
    from odvae_amd import synthetic
    from odvae_amd.synthetic import fill_state_procedural
    p, path = model_kwargs("vae")
    torch.manual_seed(3)
    model = ref_ae.PoseAutoencoder(**p)
    os.unlink(path)
    fill_state_procedural(model, seed=23)
    model.learning_rate = 12 * 4.5e-6
    arrays["step.state_keys"] = np.array(sorted(model.state_dict().keys()))
    B = 4
    batch = step_batch()          # reference_cases.step_batch: seeded, one sample of the masked class id, two partial boxes

    def ref_batch():
        b = {k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in batch.items()}
        b["patch"] = b["patch"].permute(0, 2, 3, 1)      # the reference's get_input + .permute(0,2,3,1) net to identity on NCHW ...
        b["patch"] = b["patch"].permute(0, 3, 1, 2)      # ... only for NCHW storage: hand it NCHW, as its dataset does (nuscenes.py:190-191)
        return b

    opts, _ = model.configure_optimizers()
    arrays["step.opt0.nparams"] = np.int64(sum(q.numel() for g in opts[0].param_groups for q in g["params"]))
    arrays["step.opt1.nparams"] = np.int64(sum(q.numel() for g in opts[1].param_groups for q in g["params"]))
    model.train()
    model.loss.perceptual_loss.eval()
    for opt_idx, gs, seed in ((0, 3, 900), (1, 4, 901)):
        fill_state_procedural(model, seed=23)             # BatchNorm buffers back to the same start
        model.global_step = gs
        del DRAWS[:]
        torch.manual_seed(seed)
        for q in model.parameters():
            q.grad = None
        out = model.training_step(ref_batch(), 0, opt_idx)
        noise = replay_draws(seed, B, 4, model.dropout_prob)
        assert len(DRAWS) == 2 and torch.equal(DRAWS[0], noise["posterior_eps"]) and torch.equal(DRAWS[1], noise["bbox_eps"]), \
            "the replayed RNG stream is not the one the reference's forward consumed"
        pre = "step.train.opt%d" % opt_idx
        arrays[pre + ".global_step"], arrays[pre + ".dropout_prob"] = np.int64(gs), np.float64(model.dropout_prob)
        for k, v in noise.items():
            arrays[pre + ".noise." + k] = np_(v)
        arrays[pre + ".loss"] = np_(out)
        for k, v in model.logged.items():
            arrays[pre + ".log." + k] = np_(torch.as_tensor(v).float())
        model.logged = {}
        out.backward()
        grad_digest(arrays, pre, model.named_parameters())
    # validation_step: eval mode, no graph; the loss is evaluated for optimizer 0 and 1 with split="val" (:347-357)
    fill_state_procedural(model, seed=23)
    model.eval()
    model.global_step = 7
    del DRAWS[:]
    torch.manual_seed(902)
    with torch.no_grad():
        model.validation_step(ref_batch(), 0)
    noise = replay_draws(902, B, 4, model.dropout_prob)
    assert torch.equal(DRAWS[0], noise["posterior_eps"]) and torch.equal(DRAWS[1], noise["bbox_eps"])
    arrays["step.val.global_step"] = np.int64(7)
    for k, v in noise.items():
        arrays["step.val.noise." + k] = np_(v)
    for k, v in model.logged.items():
        arrays["step.val.log." + k] = np_(torch.as_tensor(v).float())
    # log_images (:397-432): no graph, forward + the perturbed-pose decode; the three image sets as digests
    fill_state_procedural(model, seed=23)
    model.global_step = 9
    del DRAWS[:]
    torch.manual_seed(903)
    imgs = model.log_images(ref_batch())
    noise = replay_draws(903, B, 4, model.dropout_prob, perturbed=True)
    assert len(DRAWS) == 3 and torch.equal(DRAWS[0], noise["posterior_eps"]) and torch.equal(DRAWS[1], noise["bbox_eps"]) \
        and torch.equal(DRAWS[2], noise["posterior_eps_perturbed"])
    arrays["step.images.global_step"] = np.int64(9)
    for k, v in noise.items():
        arrays["step.images.noise." + k] = np_(v)
    assert sorted(imgs) == ["inputs_rgb", "perturbed_pose_reconstruction_rgb", "reconstructions_rgb"]
    for k, v in imgs.items():
        digest(arrays, "step.images." + k, v)


def case_headline(arrays, ref_ae):
    """BASELINE.json configs[1] itself, as far as the CPU allows: the yaml's OWN network (ch = 128, 71 M parameters in optimizer 0, 4 096 attention
    tokens) at 256 x 256, rec+KL only (perceptual_weight = 0, disc_factor = 0), B = 2, one training_step of optimizer 0 at global_step 1 run by the
    reference's PoseAutoencoder.training_step / PoseLoss.forward, backward by autograd: loss, logged terms, per-parameter gradient digests.  Weights
    follow from the state_dict keys (fill_state_procedural), the batch from synthetic.make_batch(2, 256, seed=5), the noise is stored."""
    from odvae_amd import synthetic
    from odvae_amd.synthetic import fill_state_procedural
    p, path = model_kwargs("vae", latent_hw=16, ch=None, perceptual_weight=0.0, disc_factor=0.0)
    torch.manual_seed(3)
    model = ref_ae.PoseAutoencoder(**p)
    os.unlink(path)
    fill_state_procedural(model, seed=23)
    model.train()
    model.global_step = 1
    batch = synthetic.make_batch(2, 256, seed=5)
    del DRAWS[:]
    torch.manual_seed(910)
    out = model.training_step({k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in batch.items()}, 0, 0)
    noise = replay_draws(910, 2, 16, model.dropout_prob)
    assert len(DRAWS) == 2 and torch.equal(DRAWS[0], noise["posterior_eps"]) and torch.equal(DRAWS[1], noise["bbox_eps"])
    pre = "headline.train.opt0"
    arrays[pre + ".dropout_prob"] = np.float64(model.dropout_prob)
    for k, v in noise.items():
        arrays[pre + ".noise." + k] = np_(v)
    arrays[pre + ".loss"] = np_(out)
    for k, v in model.logged.items():
        arrays[pre + ".log." + k] = np_(torch.as_tensor(v).float())
    out.backward()
    grad_digest(arrays, pre, model.named_parameters())
    arrays["headline.nparams"] = np.int64(sum(q.numel() for q in model.parameters()))


def main():
    torch.set_num_threads(4)
    install_standins()
    ref_ae, ref_loss, ref_dist = import_reference()
    arrays = {}
    case_kl(arrays, ref_dist)
    case_dropout_schedule(arrays, ref_ae)
    case_loss(arrays, ref_loss, ref_dist)
    case_steps(arrays, ref_ae)
    case_headline(arrays, ref_ae)
    np.savez_compressed(OUT, **arrays)
    print("wrote %s: %d arrays, %d bytes" % (OUT, len(arrays), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()

"""Checkpoint interchange (src/models/autoencoder.py:44-45,97-98 + [UPSTREAM] init_from_ckpt):
a Lightning-style checkpoint {"state_dict": ...} written from the ORACLE's module tree (the reference's key scheme) loads
into the HIP-backed model through ckpt_path / ignore_keys, and the model's own state_dict loads back into the oracle.
CPU only: no kernel runs, only the parameter trees meet."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, "tests", "golden", "autoencoder_kl_16x16x16.yaml")


def _oracle(ch=32, latent_hw=4):
    from odvae_amd import synthetic
    from oracle.autoencoder import PoseAutoencoder
    mcfg, _ = synthetic.model_config(YAML, latent_hw=latent_hw, ch=ch)
    p = mcfg.params.to_container()
    return PoseAutoencoder(p["ddconfig"], dict(p["lossconfig"]["params"]), p["embed_dim"], p["pose_decoder_config"]["params"],
                           p["pose_encoder_config"]["params"], feat_dims=p["feat_dims"])


def test_lightning_checkpoint_round_trip(tmp_path):
    from odvae_amd import synthetic
    from odvae_amd.config import instantiate_from_config
    torch.manual_seed(7)
    ref = _oracle()
    path = os.path.join(tmp_path, "last.ckpt")
    torch.save({"state_dict": ref.state_dict(), "global_step": 123, "epoch": 4}, path)

    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32)
    mcfg.params["ckpt_path"] = path
    model = instantiate_from_config(mcfg)          # PoseAutoencoder.__init__ -> init_from_ckpt
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref.state_dict().keys())
    # QUIRK kept from the reference (autoencoder.py:97-104): init_from_ckpt runs BEFORE the pose MLPs are constructed, so
    # ckpt_path never restores pose_encoder.* / pose_decoder.* (strict=False hides it); everything else is restored.
    for k, v in ref.state_dict().items():
        if k.startswith(("pose_encoder", "pose_decoder")):
            assert not torch.equal(sd[k], v) or v.numel() == 0, k
        else:
            assert torch.equal(sd[k], v), k
    model.load_state_dict(ref.state_dict(), strict=True)      # an explicit load restores them too
    assert torch.equal(model.state_dict()["pose_decoder.layers.0.weight"], ref.state_dict()["pose_decoder.layers.0.weight"])

    # ignore_keys drops whole prefixes (how the authors re-initialise the loss / discriminator)
    torch.manual_seed(8)
    mcfg2, _ = synthetic.model_config(YAML, latent_hw=4, ch=32)
    mcfg2.params["ckpt_path"] = path
    mcfg2.params["ignore_keys"] = ["loss.discriminator", "pose_decoder"]
    model2 = instantiate_from_config(mcfg2)
    sd2 = model2.state_dict()
    assert torch.equal(sd2["encoder.conv_in.weight"], ref.state_dict()["encoder.conv_in.weight"])
    assert not torch.equal(sd2["loss.discriminator.main.0.weight"], ref.state_dict()["loss.discriminator.main.0.weight"])
    assert not torch.equal(sd2["pose_decoder.layers.0.weight"], ref.state_dict()["pose_decoder.layers.0.weight"])

    # and back: the HIP model's state_dict is a valid reference checkpoint
    back = _oracle()
    res = back.load_state_dict(model2.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys


# [UPSTREAM] key scheme of taming's LPIPS as it sits in a reference checkpoint under `loss.perceptual_loss.` -- written out
# by hand (torchvision vgg16 feature indices of the 13 convs, the five lin heads, the two ScalingLayer buffers), NOT derived
# from either implementation under test
UPSTREAM_LPIPS_SHAPES = {"scaling_layer.shift": (1, 3, 1, 1), "scaling_layer.scale": (1, 3, 1, 1)}
for _sl, _convs in {"slice1": [(0, 3, 64), (2, 64, 64)], "slice2": [(5, 64, 128), (7, 128, 128)],
                    "slice3": [(10, 128, 256), (12, 256, 256), (14, 256, 256)],
                    "slice4": [(17, 256, 512), (19, 512, 512), (21, 512, 512)],
                    "slice5": [(24, 512, 512), (26, 512, 512), (28, 512, 512)]}.items():
    for _i, _ci, _co in _convs:
        UPSTREAM_LPIPS_SHAPES["net.%s.%d.weight" % (_sl, _i)] = (_co, _ci, 3, 3)
        UPSTREAM_LPIPS_SHAPES["net.%s.%d.bias" % (_sl, _i)] = (_co,)
for _k, _c in enumerate([64, 128, 256, 512, 512]):
    UPSTREAM_LPIPS_SHAPES["lin%d.model.1.weight" % _k] = (1, _c, 1, 1)


def synthetic_upstream_lpips_state(seed=11):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in UPSTREAM_LPIPS_SHAPES.items():
        if k.endswith("shift"):
            sd[k] = torch.tensor([-.030, -.088, -.188]).view(shape)
        elif k.endswith("scale"):
            sd[k] = torch.tensor([.458, .448, .450]).view(shape)
        elif k.startswith("lin"):
            sd[k] = torch.rand(shape, generator=g) / shape[1]
        elif k.endswith("bias"):
            sd[k] = torch.randn(shape, generator=g) * 0.05
        else:
            sd[k] = torch.randn(shape, generator=g) * (2.0 / (9 * shape[1])) ** 0.5
    return sd


def test_lpips_subtree_uses_the_upstream_key_scheme(tmp_path):
    """A checkpoint written with the upstream keys loads strict=True with nothing missing or unexpected, into the product
    tree and into the oracle's; through ckpt_path the perceptual weights really arrive (they used to be dropped silently)."""
    from odvae_amd import synthetic
    from odvae_amd.config import instantiate_from_config
    from odvae_amd.gan import LPIPSStyle
    from oracle.losses import LPIPSStyle as OracleLPIPS
    sd = synthetic_upstream_lpips_state()
    for net in (LPIPSStyle(), OracleLPIPS()):
        assert set(net.state_dict().keys()) == set(UPSTREAM_LPIPS_SHAPES)
        res = net.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    prod = LPIPSStyle()
    assert prod.has_synthetic_weights()
    prod.load_state_dict(sd, strict=True)
    assert not prod.has_synthetic_weights()

    # whole-model checkpoint in the reference's layout: oracle tree + the upstream LPIPS tensors under loss.perceptual_loss.
    torch.manual_seed(9)
    ref = _oracle()
    full = ref.state_dict()
    for k, v in sd.items():
        assert "loss.perceptual_loss." + k in full
        full["loss.perceptual_loss." + k] = v
    path = os.path.join(tmp_path, "epoch=000001.ckpt")
    torch.save({"state_dict": full}, path)
    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32, perceptual_weight=1.0)
    mcfg.params["ckpt_path"] = path
    model = instantiate_from_config(mcfg)
    got = model.state_dict()
    for k, v in sd.items():
        assert torch.equal(got["loss.perceptual_loss." + k], v), k
    assert not model.loss.perceptual_loss.has_synthetic_weights()

    # the two upstream download files (torchvision vgg16 state_dict, taming vgg.pth) through load_weights()
    vgg_file = {("features." + k.split(".", 2)[2]): v for k, v in sd.items() if k.startswith("net.")}
    vgg_file["classifier.0.weight"] = torch.zeros(2, 2)      # present in the real file, ignored
    lin_file = {k: v for k, v in sd.items() if k.startswith("lin")}
    torch.save(vgg_file, os.path.join(tmp_path, "vgg16.pth"))
    torch.save(lin_file, os.path.join(tmp_path, "vgg.pth"))
    fresh = LPIPSStyle().load_weights(vgg16=os.path.join(tmp_path, "vgg16.pth"), lins=os.path.join(tmp_path, "vgg.pth"))
    for k, v in sd.items():
        assert torch.equal(fresh.state_dict()[k], v), k

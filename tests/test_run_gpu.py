"""The untouched reference yaml end to end on the GPU: thin runner (config merge + dotlist + lr rule + PL loop) in the
encoder-pre-training phase the yaml starts in, and log_images against the oracle's decode."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, "tests", "golden", "autoencoder_kl_16x16x16.yaml")


def test_runner_with_untouched_yaml_phase_one(hip_lib):
    from odvae_amd import run
    model = run.main(["-b", YAML, "--steps", "2", "--height", "64", "model.params.ddconfig.ch=32", "data.params.batch_size=2"])
    assert model.global_step == 4                       # two optimizers per batch
    assert abs(model.learning_rate - 1 * 1 * 2 * 4.5e-6) < 1e-12
    logs = model.logged_metrics
    assert float(logs["dropout_prob"]) == 1.0           # phase 1: dropout p = 1, decoder skipped, pose losses only
    assert float(logs["train/d_weight"]) == 0.0
    assert torch.isfinite(logs["train/total_loss"]) and torch.isfinite(logs["train/disc_loss"])
    # phase 1 trains only what feeds the pose head: decoder weights untouched by Adam
    assert model.loss.encoder_pretrain_steps == 30000 and model.encoder_pretrain_steps == 30000


def test_log_images_matches_oracle_decode(hip_lib):
    from odvae_amd import synthetic
    from oracle.autoencoder import PoseAutoencoder as OraclePA
    torch.manual_seed(3)
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32)
    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32)
    p = mcfg.params.to_container()
    ref = OraclePA(p["ddconfig"], dict(p["lossconfig"]["params"]), p["embed_dim"], p["pose_decoder_config"]["params"],
                   p["pose_encoder_config"]["params"], feat_dims=p["feat_dims"], dropout_prob_final=p["dropout_prob_final"],
                   dropout_warmup_steps=p["dropout_warmup_steps"],
                   pose_conditioned_generation_steps=p["pose_conditioned_generation_steps"])
    ref.load_state_dict(model.state_dict())
    model = model.to("cuda:0").eval(); ref.eval()
    batch = synthetic.make_batch(2, 64, seed=9)
    noise = synthetic.make_noise(2, 4, seed=10)
    model.injected_noise = noise
    out = model.log_images({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    assert set(out) == {"reconstructions_rgb", "perturbed_pose_reconstruction_rgb", "inputs_rgb"}
    x = ref._rescale(batch["patch"].float())
    with torch.no_grad():
        dec_obj, dec_pose, post, _ = ref.forward(x, noise, training=False)
    err = (out["inputs_rgb"].cpu() - x).abs().max().item()
    assert err < 1e-5, err
    rec = out["reconstructions_rgb"].cpu()
    assert (rec - dec_obj).abs().max().item() <= 1e-3 * max(1.0, dec_obj.abs().max().item())
    assert tuple(out["perturbed_pose_reconstruction_rgb"].shape) == (2, 3, 64, 64)


def test_perturbed_pose_reconstruction_matches_oracle(hip_lib):
    """log_images' second image (src/models/autoencoder.py:379-432): a fresh posterior sample, the decoded pose with its yaw
    replaced by `yaw_perturbed`, pose-encoded and added to z, decoded -- against the same arithmetic on the oracle's modules."""
    from odvae_amd import synthetic
    from oracle.autoencoder import PoseAutoencoder as OraclePA
    torch.manual_seed(4)
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32)
    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32)
    p = mcfg.params.to_container()
    ref = OraclePA(p["ddconfig"], dict(p["lossconfig"]["params"]), p["embed_dim"], p["pose_decoder_config"]["params"],
                   p["pose_encoder_config"]["params"], feat_dims=p["feat_dims"], dropout_prob_final=p["dropout_prob_final"],
                   dropout_warmup_steps=p["dropout_warmup_steps"],
                   pose_conditioned_generation_steps=p["pose_conditioned_generation_steps"])
    ref.load_state_dict(model.state_dict())
    model = model.to("cuda:0").eval(); ref.eval()
    batch = synthetic.make_batch(2, 64, seed=11)
    noise = synthetic.make_noise(2, 4, seed=12)
    noise["posterior_eps_perturbed"] = torch.randn(2, 16, 4, 4, generator=torch.Generator().manual_seed(13))
    model.injected_noise = noise
    out = model.log_images({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    x = ref._rescale(batch["patch"].float())
    with torch.no_grad():
        _, dec_pose, post, _ = ref.forward(x, noise, training=False)
        z = post.sample(noise["posterior_eps_perturbed"])
        pose = dec_pose.clone()
        pose[:, 3] = batch["yaw_perturbed"]          # train_on_yaw: only the yaw slot is perturbed (:284-293,379-386)
        want = ref.decode(z + ref.pose_encoder(pose).view(-1, *ref.feature_dims))
    got = out["perturbed_pose_reconstruction_rgb"].cpu()
    assert (got - want).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())
    assert (got - out["reconstructions_rgb"].cpu()).abs().max().item() > 1e-3      # it really is a different image


def test_runner_with_yaml_callbacks_writes_image_grids(hip_lib, tmp_path):
    """`-l <logdir>`: the callbacks of the untouched yaml (yaml:115-131) run behind the training loop; ImageLogger calls log_images on
    the device at global steps 2, 4 (its power-of-two schedule; PL counts two optimizer steps per batch) and writes the three PNGs of
    src/util/callbacks.py:141-160 per firing, max_images = 1 (yaml:119) -> a bare 64 x 64 image, values in the open range."""
    import numpy as np
    from PIL import Image
    from odvae_amd import run
    model = run.main(["-b", YAML, "--steps", "2", "--height", "64", "-l", str(tmp_path), "model.params.ddconfig.ch=32", "data.params.batch_size=2"])
    assert model.training and model.global_step == 4
    files = sorted(os.listdir(os.path.join(tmp_path, "images", "train")))
    assert files == sorted("%s_gs-%06d_e-000000_b-%06d.png" % (k, gs, b) for gs, b in ((2, 0), (4, 1))
                           for k in ("inputs_rgb", "reconstructions_rgb", "perturbed_pose_reconstruction_rgb")), files
    img = np.asarray(Image.open(os.path.join(tmp_path, "images", "train", "inputs_rgb_gs-000004_e-000000_b-000001.png")))
    assert img.shape == (64, 64, 3) and img.min() == 0 and img.max() >= 254      # _rescale maps the batch to [-1, 1]
    assert "DeviceStatsMonitor.on_train_batch_end/allocated_bytes.all.current" in model.logged_metrics

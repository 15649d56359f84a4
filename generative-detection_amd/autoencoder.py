"""AutoencoderKL / Autoencoder / PoseAutoencoder on the HIP kernels, with the reference's LightningModule surface.

Mirrors src/models/autoencoder.py (PoseAutoencoder :34-443, Autoencoder :29-32) and the [UPSTREAM]
ldm/models/autoencoder.py AutoencoderKL it derives from (encode / decode / get_input / get_last_layer /
init_from_ckpt / configure_optimizers).  Constructor signature, attribute names, state_dict keys, batch-dict keys,
global_step thresholds and logged names are the reference's; the arithmetic is in generative-detection_amd/ops.py.
"""
import logging

import torch
import torch.nn as nn

from . import lib as _lib
from . import ops
from .config import instantiate_from_config
from .distributions import DiagonalGaussianDistribution
from .lightning import LightningModule
from .modules import Conv1x1, Decoder, Encoder
from .optim import make_adam

POSE_6D_DIM = 4
FILL_FACTOR_DIM = 1
LHW_DIM = 3
BBOX_DIM = POSE_6D_DIM + LHW_DIM + FILL_FACTOR_DIM


class FeatEncoder(Encoder):
    """src/modules/autoencodermodules/feat_encoder.py:4-6"""


class FeatDecoder(Decoder):
    """src/modules/autoencodermodules/feat_decoder.py:4-6"""


class AutoencoderKL(LightningModule):
    """[UPSTREAM] ldm.models.autoencoder.AutoencoderKL (the plain KL autoencoder BASELINE.json names)."""

    def __init__(self, ddconfig, lossconfig, embed_dim, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.loss = instantiate_from_config(lossconfig)
        assert ddconfig["double_z"]
        self.quant_conv = Conv1x1(2 * ddconfig["z_channels"], 2 * embed_dim)
        self.post_quant_conv = Conv1x1(embed_dim, ddconfig["z_channels"])
        self.embed_dim = embed_dim
        if colorize_nlabels is not None:
            assert type(colorize_nlabels) == int
            self.register_buffer("colorize", torch.randn(3, colorize_nlabels, 1, 1))
        if monitor is not None:
            self.monitor = monitor
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def set_precision(self, precision):
        """The trainer's `precision` (configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml:139; PL-1.9 accepts 32, "32", 16, "bf16"):
        32 = f32 everywhere; "bf16" = mixed precision -- bf16 activations inside Encoder / Decoder on the bf16 MFMA kernels, f32
        master weights, gradients, statistics, latent, reconstruction and losses.  fp16 is not offered (no loss scaler here)."""
        p = str(precision).lower()
        if p in ("32", "32-true", "fp32"):
            dt = torch.float32
        elif p in ("bf16", "bf16-mixed"):
            dt = torch.bfloat16
        else:
            raise ValueError("precision %r: this build computes in 32 (f32) or bf16 (mixed precision)" % (precision,))
        self.encoder.compute_dtype = dt
        self.decoder.compute_dtype = dt
        return self

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                print("Deleting key {} from state_dict.".format(k))
                del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def encode(self, x):
        moments = self.quant_conv(self.encoder(x))
        return DiagonalGaussianDistribution(moments)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior

    def get_input(self, batch, k):
        x = batch[k]
        if len(x.shape) == 3:
            x = x[..., None]
        return x.permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format).float()

    def training_step(self, batch, batch_idx, optimizer_idx):
        inputs = self.get_input(batch, self.image_key).to(self.device)
        reconstructions, posterior = self(inputs)
        if optimizer_idx == 0:
            aeloss, log_dict_ae = self.loss(inputs, reconstructions, posterior, optimizer_idx, self.global_step,
                                            last_layer=self.get_last_layer(), split="train")
            self.log("aeloss", aeloss, prog_bar=True, logger=True, on_step=True, on_epoch=True)
            self.log_dict(log_dict_ae, prog_bar=False, logger=True, on_step=True, on_epoch=False)
            return aeloss
        discloss, log_dict_disc = self.loss(inputs, reconstructions, posterior, optimizer_idx, self.global_step,
                                            last_layer=self.get_last_layer(), split="train")
        self.log("discloss", discloss, prog_bar=True, logger=True, on_step=True, on_epoch=True)
        self.log_dict(log_dict_disc, prog_bar=False, logger=True, on_step=True, on_epoch=False)
        return discloss

    def configure_optimizers(self):
        lr = self.learning_rate
        opt_ae = make_adam(list(self.encoder.parameters()) + list(self.decoder.parameters())
                           + list(self.quant_conv.parameters()) + list(self.post_quant_conv.parameters()),
                           lr=lr, betas=(0.5, 0.9))
        opt_disc = make_adam(self.loss.discriminator.parameters(), lr=lr, betas=(0.5, 0.9))
        return [opt_ae, opt_disc], []

    def get_last_layer(self):
        return self.decoder.conv_out.weight

    def to_rgb(self, x):
        """Random fixed 1x1 projection of a C-channel map to three channels, rescaled to [-1, 1] over the batch
        (src/models/autoencoder.py:438-443; [UPSTREAM] AutoencoderKL.to_rgb, used when `colorize_nlabels` is given).
        `colorize` is a buffer drawn once, as in the reference."""
        if not hasattr(self, "colorize"):
            self.register_buffer("colorize", torch.randn(3, x.shape[1], 1, 1).to(x))
        return ops.rescale_minmax(ops.conv1x1(x, self.colorize))


class Autoencoder(AutoencoderKL):
    """src/models/autoencoder.py:29-32"""


class PoseAutoencoder(AutoencoderKL):
    def __init__(self, ddconfig, lossconfig, embed_dim, euler_convention, ckpt_path=None, ignore_keys=[],
                 image_mask_key=None, image_rgb_key="patch", pose_key="pose_6d", fill_factor_key="fill_factor",
                 pose_perturbed_key="pose_6d_perturbed", class_key="class_id", bbox_key="bbox_sizes",
                 colorize_nlabels=None, monitor=None, activation="relu", feat_dims=[16, 16, 16],
                 pose_decoder_config=None, pose_encoder_config=None, dropout_prob_init=1.0, dropout_prob_final=0.7,
                 dropout_warmup_steps=5000, pose_conditioned_generation_steps=10000, add_noise_to_z_obj=True,
                 train_on_yaw=True):
        LightningModule.__init__(self)  # the reference skips AutoencoderKL.__init__ as well (:66)
        self.encoder_pretrain_steps = lossconfig["params"]["encoder_pretrain_steps"]
        self.train_on_yaw = train_on_yaw
        self.dropout_prob_final, self.dropout_prob_init = dropout_prob_final, dropout_prob_init
        self.dropout_prob = dropout_prob_init
        self.dropout_warmup_steps = dropout_warmup_steps
        self.pose_conditioned_generation_steps = pose_conditioned_generation_steps
        self.add_noise_to_z_obj = add_noise_to_z_obj
        self.feature_dims = feat_dims
        self.image_rgb_key, self.image_mask_key = image_rgb_key, image_mask_key
        self.pose_key, self.pose_perturbed_key = pose_key, pose_perturbed_key
        self.class_key, self.bbox_key, self.fill_factor_key = class_key, bbox_key, fill_factor_key
        self.encoder = FeatEncoder(**ddconfig)
        self.decoder = FeatDecoder(**ddconfig)
        lossconfig["params"]["train_on_yaw"] = self.train_on_yaw
        self.loss = instantiate_from_config(lossconfig)
        assert ddconfig["double_z"]
        zc = ddconfig["z_channels"]
        self.quant_conv_obj = Conv1x1(2 * zc, 2 * embed_dim)
        self.quant_conv_pose = Conv1x1(2 * zc, embed_dim)
        self.post_quant_conv = Conv1x1(embed_dim, zc)
        self.embed_dim = embed_dim
        if colorize_nlabels is not None:
            assert type(colorize_nlabels) == int
            self.register_buffer("colorize", torch.randn(3, colorize_nlabels, 1, 1))
        if monitor is not None:
            self.monitor = monitor
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)
        self.num_classes = lossconfig["params"]["num_classes"]
        self.pose_decoder = instantiate_from_config(pose_decoder_config)
        self.pose_encoder = instantiate_from_config(pose_encoder_config)
        self.z_channels = zc
        self.euler_convention = euler_convention
        self.feat_dims = feat_dims
        # test hook: {"posterior_eps", "dropout_mask", "z_noise", "bbox_eps"} tensors replace the host RNG draws
        self.injected_noise = None

    # ---- pose head (:126-174) ---------------------------------------------------------------------------------
    def _decode_pose_to_distribution(self, z, sample_posterior=True):
        c_pred = z[..., -self.num_classes:]
        bbox_moments = z[..., :2 * BBOX_DIM].to(self.device)
        return DiagonalGaussianDistribution(bbox_moments), c_pred

    def _decode_pose(self, x, sample_posterior=True):
        flat = x.reshape(x.size(0), -1)  # logical NCHW order, as x.view(B,-1) in the reference
        z = self.pose_decoder(flat)
        bbox_posterior, c_pred = self._decode_pose_to_distribution(z)
        if sample_posterior:
            bbox_pred = bbox_posterior.sample(self._noise("bbox_eps"))
        else:
            bbox_pred = bbox_posterior.mode()
        return torch.cat([bbox_pred, c_pred], dim=-1), bbox_posterior

    def _encode_pose(self, x):
        flat = self.pose_encoder(x)
        return flat.view(flat.size(0), self.feature_dims[0], self.feature_dims[1], self.feature_dims[2])

    def _noise(self, key):
        return None if self.injected_noise is None else self.injected_noise.get(key)

    # ---- encode / forward (:176-257) -------------------------------------------------------------------------
    def encode(self, x):
        x = x.to(self.device)
        h = self.encoder(x)
        moments_obj = self.quant_conv_obj(h)
        pose_feat = self.quant_conv_pose(h)
        return DiagonalGaussianDistribution(moments_obj), pose_feat

    def _get_dropout_prob(self):
        step, pre, gen = self.global_step, self.encoder_pretrain_steps, self.pose_conditioned_generation_steps
        if step < pre + gen:
            return self.dropout_prob_init
        if step < self.dropout_warmup_steps + pre + gen:
            # QUIRK (:200): the ramp is measured from encoder_pretrain_steps, not from pre+gen
            return self.dropout_prob_init - (self.dropout_prob_init - self.dropout_prob_final) * (step - pre) / self.dropout_warmup_steps
        return self.dropout_prob_final

    def _dropout(self, z, p):
        """nn.Dropout(p) keeps with prob 1-p and scales by 1/(1-p); p = 1 zeroes z.  QUIRK kept (autoencoder.py:233-235):
        the reference builds a fresh nn.Dropout inside forward, and a fresh module is in training mode, so the dropout
        is active in validation and log_images too."""
        mask = self._noise("dropout_mask")
        if mask is None:
            if p >= 1.0:
                mask = torch.zeros(z.shape)
            else:
                mask = (torch.rand(z.shape) >= p).float() / (1.0 - p)
        return ops.latent_combine(z, _lib.upload(mask, z.device), None)

    def forward(self, input_im, sample_posterior=True):
        posterior_obj, pose_feat = self.encode(input_im)
        z_obj = posterior_obj.sample(self._noise("posterior_eps")) if sample_posterior else posterior_obj.mode()
        self.dropout_prob = self._get_dropout_prob()
        if self.dropout_prob > 0:
            z_obj = self._dropout(z_obj, self.dropout_prob)
        if self.add_noise_to_z_obj:
            z_noise = self._noise("z_noise")
            if z_noise is None:
                z_noise = torch.randn(tuple(z_obj.shape))  # Normal(0,1).sample(...) on the host, as the reference (:239-240)
            z_obj = ops.latent_combine(z_obj, None, _lib.upload(z_noise, self.device))
        dec_pose, bbox_posterior = self._decode_pose(pose_feat, sample_posterior)
        if self.global_step < self.encoder_pretrain_steps:
            n, _, h, w = input_im.shape
            dec_obj = torch.zeros((n, h, w, input_im.shape[1]), device=self.device).permute(0, 3, 1, 2)
        else:
            enc_pose = self._encode_pose(dec_pose)
            assert z_obj.shape == enc_pose.shape, f"z_obj shape: {z_obj.shape}, enc_pose shape: {enc_pose.shape}"
            dec_obj = self.decode(ops.latent_combine(z_obj, None, enc_pose))
        return dec_obj, dec_pose, posterior_obj, bbox_posterior

    # ---- batch access (:259-293) ---------------------------------------------------------------------------------
    def get_pose_input(self, batch, k):
        x = batch[k]
        if self.train_on_yaw:
            x[:, 3] = batch["yaw"]
        return x.to(memory_format=torch.contiguous_format).float()

    def get_mask_input(self, batch, k):
        return None if k is None else batch[k]

    def get_class_input(self, batch, k):
        return batch[k]

    def get_bbox_input(self, batch, k):
        return batch[k]

    def get_fill_factor_input(self, batch, k):
        return batch[k]

    def _get_perturbed_pose(self, batch, k):
        x = batch[k].squeeze(1)
        if self.train_on_yaw:
            x = torch.zeros_like(x)
            x[:, -1] = batch["yaw_perturbed"]
        return x

    def _rgb_input(self, batch):
        """get_input(...).permute(0,2,3,1) is the identity on an NCHW `patch` (:296); _rescale on the device."""
        x = batch[self.image_rgb_key]
        if x.dim() == 3:
            x = x[..., None]
        return self._rescale(x.float().to(self.device))

    def _unpack(self, batch):
        rgb_gt = self._rgb_input(batch)
        pose_gt = self.get_pose_input(batch, self.pose_key).to(self.device)
        mask_gt = self.get_mask_input(batch, self.image_mask_key)
        mask_gt = mask_gt.to(self.device) if mask_gt is not None else None
        class_gt = self.get_class_input(batch, self.class_key).to(self.device)
        bbox_gt = self.get_bbox_input(batch, self.bbox_key).to(self.device)
        fill_factor_gt = self.get_fill_factor_input(batch, self.fill_factor_key).to(self.device).float()
        return rgb_gt, mask_gt, pose_gt, class_gt, batch["class_name"], bbox_gt, fill_factor_gt, batch["mask_2d_bbox"]

    # ---- steps (:295-363) ---------------------------------------------------------------------------------------------
    def training_step(self, batch, batch_idx, optimizer_idx):
        rgb_gt, mask_gt, pose_gt, class_gt, class_gt_label, bbox_gt, fill_factor_gt, mask_2d_bbox = self._unpack(batch)
        dec_obj, dec_pose, posterior_obj, bbox_posterior = self.forward(rgb_gt)
        self.log("dropout_prob", self.dropout_prob, prog_bar=True, logger=True, on_step=True, on_epoch=True)
        loss, log_dict = self.loss(rgb_gt, mask_gt, pose_gt, dec_obj, dec_pose, class_gt, class_gt_label, bbox_gt,
                                   fill_factor_gt, posterior_obj, bbox_posterior, optimizer_idx, self.global_step,
                                   mask_2d_bbox, last_layer=self.get_last_layer(), split="train")
        name = "aeloss" if optimizer_idx == 0 else "discloss"
        self.log(name, loss, prog_bar=True, logger=True, on_step=True, on_epoch=True)
        self.log_dict(log_dict, prog_bar=False, logger=True, on_step=True, on_epoch=False)
        return loss

    def validation_step(self, batch, batch_idx):
        rgb_gt, mask_gt, pose_gt, class_gt, class_gt_label, bbox_gt, fill_factor_gt, mask_2d_bbox = self._unpack(batch)
        dec_obj, dec_pose, posterior_obj, bbox_posterior = self.forward(rgb_gt)
        logs = []
        for optimizer_idx in (0, 1):
            _, log = self.loss(rgb_gt, mask_gt, pose_gt, dec_obj, dec_pose, class_gt, class_gt_label, bbox_gt,
                               fill_factor_gt, posterior_obj, bbox_posterior, optimizer_idx, self.global_step,
                               mask_2d_bbox, last_layer=self.get_last_layer(), split="val")
            logs.append(log)
        log_dict_ae, log_dict_disc = logs
        self.log("val/rec_loss", log_dict_ae["val/rec_loss"], sync_dist=True)
        del log_dict_ae["val/rec_loss"]
        self.log_dict(log_dict_ae)
        self.log_dict(log_dict_disc)
        return self.log_dict

    def configure_optimizers(self):
        lr = self.learning_rate
        ae_params = (list(self.encoder.parameters()) + list(self.decoder.parameters())
                     + list(self.quant_conv_obj.parameters()) + list(self.quant_conv_pose.parameters())
                     + list(self.post_quant_conv.parameters()) + list(self.pose_encoder.parameters())
                     + list(self.pose_decoder.parameters()))  # loss.logvar is in no optimizer, as in the reference
        opt_ae = make_adam(ae_params, lr=lr, betas=(0.5, 0.9))
        opt_disc = make_adam(self.loss.discriminator.parameters(), lr=lr, betas=(0.5, 0.9))
        return [opt_ae, opt_disc], []

    # ---- image logging (:379-432) ---------------------------------------------------------------------------------------
    def _perturb_poses(self, batch, dec_pose):
        yaw = self._get_perturbed_pose(batch, self.pose_perturbed_key).squeeze()[:, -1]
        out = dec_pose.clone()
        out[:, 3] = yaw
        assert out.shape == dec_pose.shape
        return out.to(self.device)

    def _perturbed_pose_forward(self, posterior_obj, dec_pose, batch, sample_posterior=True):
        z_obj = posterior_obj.sample(self._noise("posterior_eps_perturbed")) if sample_posterior else posterior_obj.mode()
        enc_pose = self._encode_pose(self._perturb_poses(batch, dec_pose))
        return self.decode(ops.latent_combine(z_obj, None, enc_pose))

    @torch.no_grad()
    def log_images(self, batch, only_inputs=False, **kwargs):
        log = dict()
        x_rgb = self._rgb_input(batch)
        if not only_inputs:
            xrec, poserec, posterior_obj, _ = self.forward(x_rgb)
            xrec_perturbed = self._perturbed_pose_forward(posterior_obj, poserec, batch)
            log["reconstructions_rgb"] = xrec[:, :3, :, :].clone().detach()
            log["perturbed_pose_reconstruction_rgb"] = xrec_perturbed[:, :3, :, :].clone().detach()
        log["inputs_rgb"] = x_rgb.clone().detach()
        return log

    def _rescale(self, x):
        return ops.rescale_minmax(x)

"""ORACLE (test infrastructure): the reference's PoseAutoencoder training path restated in plain PyTorch on the CPU.

Follows src/models/autoencoder.py (forward :208-257, encode :176-182, pose head :126-174, dropout schedule :184-206,
training_step :295-330, _rescale :434-436, configure_optimizers :365-377) with the [UPSTREAM] ldm AutoencoderKL.decode
(post_quant_conv then decoder).  RNG draws are parameters so the HIP path can be fed the same noise.
PINNED since round 5 against outputs of the reference's OWN src/models/autoencoder.py, imported unmodified in the build container with stand-ins for the
absent third-party names (tests/golden/make_reference_goldens.py -> reference_glue.npz; tests/test_reference_glue.py: dropout schedule, training_step for both
optimizer indices, validation_step, and the headline network's step, loss / logs / every parameter gradient).  The Encoder / Decoder underneath (ldm_model.py)
remain a restatement of absent upstream code: parity unpinned for that layer.
"""
import numpy as np
import torch
import torch.nn as nn

from .distributions import DiagonalGaussianDistribution
from .ldm_model import Decoder, Encoder
from .losses import PoseLoss

POSE_6D_DIM, LHW_DIM, FILL_FACTOR_DIM = 4, 3, 1
BBOX = POSE_6D_DIM + LHW_DIM + FILL_FACTOR_DIM


class PoseEncoderSpatialVAE(nn.Module):
    """src/modules/autoencodermodules/pose_encoder.py:59-131 (verified against the reference's own importable module:
    tests/golden/pose_encoder_ref.npz)."""

    def __init__(self, num_classes=2, num_channels=16, n=16, m=16, activation="swish", hidden_dim=500, num_layers=2):
        super().__init__()
        act = {"swish": nn.SiLU, "tanh": nn.Tanh}.get(activation, nn.ReLU)
        self.num_coords, self.feat_size, self.in_dim = n * m, 4, 2
        self.h_dim, self.x_dim = n * m * 4, 2 * n * m
        self.coord_linear = nn.Linear(self.x_dim, self.h_dim)
        self.latent_linear = nn.Linear(POSE_6D_DIM + LHW_DIM + FILL_FACTOR_DIM + num_classes, self.feat_size, bias=False)
        layers = [act()]
        for layer_id in range(1, num_layers):
            layers += [nn.Linear(self.h_dim if layer_id == 1 else hidden_dim, hidden_dim), act()]
        layers.append(nn.Linear(hidden_dim, num_channels * n * m))
        self.layers = nn.Sequential(*layers)
        x0, x1 = np.meshgrid(np.linspace(-1, 1, m), np.linspace(1, -1, n))
        self.x = torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1)).float()

    def forward(self, z):
        b = z.size(0)
        x = self.x.expand(b, self.num_coords, self.in_dim).to(z).contiguous().view(b, self.num_coords * self.in_dim)
        h_x = self.coord_linear(x)
        h_z = self.latent_linear(z).unsqueeze(1).expand(b, self.num_coords, self.feat_size).reshape(b, self.h_dim)
        return self.layers(h_x + h_z)


class PoseDecoderSpatialVAE(nn.Module):
    """src/modules/autoencodermodules/pose_decoder.py:60-97"""

    def __init__(self, num_classes=2, num_channels=16, n=16, m=16, activation="tanh", hidden_dim=500, num_layers=2, **kw):
        super().__init__()
        act = nn.Tanh if activation == "tanh" else nn.ReLU
        layers = [nn.Linear(num_channels * n * m, hidden_dim), act()]
        for _ in range(1, num_layers):
            layers += [nn.Linear(hidden_dim, hidden_dim), act()]
        layers.append(nn.Linear(hidden_dim, 2 * BBOX + num_classes))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x)


class PoseAutoencoder(nn.Module):
    def __init__(self, ddconfig, loss_kwargs, embed_dim, pose_decoder_kwargs, pose_encoder_kwargs, feat_dims=(16, 16, 16),
                 dropout_prob_init=1.0, dropout_prob_final=0.7, dropout_warmup_steps=5000,
                 pose_conditioned_generation_steps=10000, add_noise_to_z_obj=True, train_on_yaw=True):
        super().__init__()
        self.encoder_pretrain_steps = loss_kwargs["encoder_pretrain_steps"]
        self.dropout_prob_init, self.dropout_prob_final = dropout_prob_init, dropout_prob_final
        self.dropout_warmup_steps = dropout_warmup_steps
        self.pose_conditioned_generation_steps = pose_conditioned_generation_steps
        self.add_noise_to_z_obj, self.train_on_yaw = add_noise_to_z_obj, train_on_yaw
        self.feature_dims = list(feat_dims)
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.loss = PoseLoss(train_on_yaw=train_on_yaw, **loss_kwargs)
        zc = ddconfig["z_channels"]
        self.quant_conv_obj = nn.Conv2d(2 * zc, 2 * embed_dim, 1)
        self.quant_conv_pose = nn.Conv2d(2 * zc, embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, zc, 1)
        self.num_classes = loss_kwargs["num_classes"]
        self.pose_decoder = PoseDecoderSpatialVAE(**pose_decoder_kwargs)
        self.pose_encoder = PoseEncoderSpatialVAE(**pose_encoder_kwargs)
        self.global_step = 0
        self.learning_rate = None

    @staticmethod
    def _rescale(x):
        return 2. * (x - x.min()) / (x.max() - x.min()) - 1.

    def _get_dropout_prob(self):
        gs, pre, gen = self.global_step, self.encoder_pretrain_steps, self.pose_conditioned_generation_steps
        if gs < pre:
            return self.dropout_prob_init
        elif gs < pre + gen:
            return self.dropout_prob_init
        elif gs < self.dropout_warmup_steps + pre + gen:
            return self.dropout_prob_init - (self.dropout_prob_init - self.dropout_prob_final) * (gs - pre) / self.dropout_warmup_steps
        return self.dropout_prob_final

    def encode(self, x):
        h = self.encoder(x)
        return DiagonalGaussianDistribution(self.quant_conv_obj(h)), self.quant_conv_pose(h)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def forward(self, input_im, noise, training=True):
        """noise: dict with posterior_eps [B,Cz,h,w], dropout_mask (already scaled keep mask, same shape), z_noise, bbox_eps [B,8]"""
        posterior_obj, pose_feat = self.encode(input_im)
        z_obj = posterior_obj.sample(noise["posterior_eps"])
        self.dropout_prob = self._get_dropout_prob()
        if self.dropout_prob > 0:
            # nn.Dropout(p)(z) with the drawn mask made explicit.  The reference constructs the Dropout module inside
            # forward (:233-235); a fresh module is in training mode, so this applies in eval / log_images as well.
            z_obj = z_obj * noise["dropout_mask"]
        if self.add_noise_to_z_obj:
            z_obj = z_obj + noise["z_noise"]
        z = self.pose_decoder(pose_feat.view(pose_feat.size(0), -1))
        c_pred = z[..., -self.num_classes:]
        bbox_posterior = DiagonalGaussianDistribution(torch.cat([z[..., :BBOX], z[..., BBOX:2 * BBOX]], dim=-1))
        dec_pose = torch.cat([bbox_posterior.sample(noise["bbox_eps"]), c_pred], dim=-1)
        if self.global_step < self.encoder_pretrain_steps:
            dec_obj = torch.zeros_like(input_im)
        else:
            enc_pose = self.pose_encoder(dec_pose).view(-1, *self.feature_dims)
            dec_obj = self.decode(z_obj + enc_pose)
        return dec_obj, dec_pose, posterior_obj, bbox_posterior

    def training_step(self, batch, optimizer_idx, noise):
        rgb_gt = self._rescale(batch["patch"].float())
        pose_gt = batch["pose_6d"].clone().float()
        if self.train_on_yaw:
            pose_gt[:, 3] = batch["yaw"]
        dec_obj, dec_pose, posterior_obj, bbox_posterior = self.forward(rgb_gt, noise)
        loss, log = self.loss(rgb_gt, None, pose_gt, dec_obj, dec_pose, batch["class_id"], batch["class_name"],
                              batch["bbox_sizes"], batch["fill_factor"].float(), posterior_obj, bbox_posterior, optimizer_idx,
                              self.global_step, batch["mask_2d_bbox"], last_layer=self.decoder.conv_out.weight, split="train")
        return loss, log, dict(rgb_gt=rgb_gt, dec_obj=dec_obj, dec_pose=dec_pose, posterior=posterior_obj)

    def validation_step(self, batch, noise):
        """src/models/autoencoder.py:332-363: one forward, the loss evaluated for optimizer 0 and 1 with split="val";
        returns the merged log dict (val/rec_loss included)."""
        rgb_gt = self._rescale(batch["patch"].float())
        pose_gt = batch["pose_6d"].clone().float()
        if self.train_on_yaw:
            pose_gt[:, 3] = batch["yaw"]
        dec_obj, dec_pose, posterior_obj, bbox_posterior = self.forward(rgb_gt, noise)
        logs = {}
        for optimizer_idx in (0, 1):
            _, log = self.loss(rgb_gt, None, pose_gt, dec_obj, dec_pose, batch["class_id"], batch["class_name"],
                               batch["bbox_sizes"], batch["fill_factor"].float(), posterior_obj, bbox_posterior, optimizer_idx,
                               self.global_step, batch["mask_2d_bbox"], last_layer=self.decoder.conv_out.weight, split="val")
            logs.update(log)
        return logs

    def configure_optimizers(self):
        lr = self.learning_rate
        ae = (list(self.encoder.parameters()) + list(self.decoder.parameters()) + list(self.quant_conv_obj.parameters())
              + list(self.quant_conv_pose.parameters()) + list(self.post_quant_conv.parameters())
              + list(self.pose_encoder.parameters()) + list(self.pose_decoder.parameters()))
        return [torch.optim.Adam(ae, lr=lr, betas=(0.5, 0.9)),
                torch.optim.Adam(self.loss.discriminator.parameters(), lr=lr, betas=(0.5, 0.9))]


def train_batch(model, optimizers, batch, noise_per_opt, optimizer_indices=(0, 1), clip=1.0):
    """One PL-1.9 style batch on the oracle: per optimizer forward, zero_grad, backward, clip, step; global_step += 1 each."""
    out = []
    for idx in optimizer_indices:
        opt = optimizers[idx]
        others = [p for j, o in enumerate(optimizers) if j != idx for g in o.param_groups for p in g["params"]]
        for p in others:
            p.requires_grad = False
        loss, log, aux = model.training_step(batch, idx, noise_per_opt[idx])
        opt.zero_grad()
        loss.backward()
        if clip:
            torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], clip)
        opt.step()
        for p in others:
            p.requires_grad = True
        model.global_step += 1
        out.append((loss.detach(), log, aux))
    return out

"""ORACLE (test infrastructure): plain-PyTorch CPU restatement of the loss side of the hot path.

Follows src/modules/losses/contperceptual.py of the reference line by line in meaning (PoseLoss :26-375) plus the
un-vendored pieces it imports: [UPSTREAM] ldm/modules/losses/contperceptual.py (LPIPSWithDiscriminator ctor,
calculate_adaptive_weight), taming/modules/losses/vqperceptual.py (adopt_weight, hinge_d_loss),
taming/modules/discriminator/model.py (NLayerDiscriminator, weights_init), taming/modules/losses/lpips.py (LPIPS
structure) and mmdet FocalLoss defaults.  PoseLoss itself (contperceptual.py, which IS in the reference) is PINNED since round 5: the reference's file is
imported unmodified and run (tests/golden/make_reference_goldens.py) and tests/test_reference_glue.py holds this restatement to its outputs on eight cases x both
optimizer indices (loss, every logged term, gradients).  The upstream pieces (PatchGAN, LPIPS structure, hinge loss, adaptive weight, focal loss) stay
restatements of code that is absent here: parity unpinned for those.
Full-size tensors are materialised exactly as the reference does (rec_loss [B,3,H,W], broadcasts, host branches).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .distributions import DiagonalGaussianDistribution

POSE_6D_DIM, LHW_DIM, FILL_FACTOR_DIM = 4, 3, 1
BACKGROUND_CLASS_IDX = 1
BBOX_DIM = POSE_6D_DIM + LHW_DIM + FILL_FACTOR_DIM


def adopt_weight(weight, global_step, threshold=0, value=0.):
    if global_step < threshold:
        weight = value
    return weight


def hinge_d_loss(logits_real, logits_fake):
    loss_real = torch.mean(F.relu(1. - logits_real))
    loss_fake = torch.mean(F.relu(1. + logits_fake))
    return 0.5 * (loss_real + loss_fake)


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find('BatchNorm') != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc=3, ndf=64, n_layers=3, use_actnorm=False):
        super().__init__()
        norm_layer = nn.BatchNorm2d
        use_bias = False  # norm_layer is BatchNorm2d
        kw, padw = 4, 1
        sequence = [nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev, nf_mult = nf_mult, min(2 ** n, 8)
            sequence += [nn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias),
                         norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        nf_mult_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        sequence += [nn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                     norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        sequence += [nn.Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.main = nn.Sequential(*sequence)

    def forward(self, input):
        return self.main(input)


# [UPSTREAM] taming/modules/losses/lpips.py: torchvision vgg16().features cut into five nn.Sequential slices whose
# members keep their feature index as name; entries: int c = Conv2d(->c, 3x3, pad 1), "R" = ReLU, "M" = MaxPool2d(2, 2)
VGG16_FEATURES = [64, "R", 64, "R", "M", 128, "R", 128, "R", "M", 256, "R", 256, "R", 256, "R", "M",
                  512, "R", 512, "R", 512, "R", "M", 512, "R", 512, "R", 512, "R"]
VGG16_SLICE_BOUNDS = [(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)]


class _ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, inp):
        return (inp - self.shift) / self.scale


class _NetLinLayer(nn.Module):
    def __init__(self, chn_in, chn_out=1):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout(), nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False))


class _Vgg16(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in VGG16_FEATURES:
            if v == "R":
                layers.append(nn.ReLU(inplace=False))
            elif v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers.append(nn.Conv2d(cin, v, kernel_size=3, padding=1))
                cin = v
        for k, (lo, hi) in enumerate(VGG16_SLICE_BOUNDS):
            sl = nn.Sequential()
            for x in range(lo, hi):
                sl.add_module(str(x), layers[x])
            setattr(self, "slice%d" % (k + 1), sl)

    def forward(self, x):
        outs = []
        for k in range(5):
            x = getattr(self, "slice%d" % (k + 1))(x)
            outs.append(x)
        return outs


class LPIPSStyle(nn.Module):
    """[UPSTREAM] taming LPIPS structure and state_dict keys (scaling_layer.*, net.slice{k}.{idx}.*, lin{k}.model.1.weight)
    with whatever weights are loaded into it."""

    def __init__(self):
        super().__init__()
        self.scaling_layer = _ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.net = _Vgg16()
        for k, c in enumerate(self.chns):
            setattr(self, "lin%d" % k, _NetLinLayer(c))
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, input, target):
        f0, f1 = self.net(self.scaling_layer(input)), self.net(self.scaling_layer(target))
        val = 0
        for k in range(len(self.chns)):
            n0 = f0[k] / (torch.sqrt(torch.sum(f0[k] ** 2, dim=1, keepdim=True)) + 1e-10)
            n1 = f1[k] / (torch.sqrt(torch.sum(f1[k] ** 2, dim=1, keepdim=True)) + 1e-10)
            val = val + getattr(self, "lin%d" % k).model((n0 - n1) ** 2).mean([2, 3], keepdim=True)
        return val


def sigmoid_focal_loss_mean(pred, target, gamma=2.0, alpha=0.25):
    num_classes = pred.size(1)
    target = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
    pred_sigmoid = pred.sigmoid()
    pt = (1 - pred_sigmoid) * target + pred_sigmoid * (1 - target)
    focal_weight = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
    return (F.binary_cross_entropy_with_logits(pred, target, reduction='none') * focal_weight).mean()


class PoseLoss(nn.Module):
    def __init__(self, disc_start, dataset_stats, logvar_init=0.0, disc_num_layers=3, disc_in_channels=3, disc_factor=1.0,
                 disc_weight=1.0, perceptual_weight=1.0, train_on_yaw=True, kl_weight_obj=1.0, kl_weight_bbox=1e-6,
                 pose_weight=1.0, mask_weight=0.0, class_weight=1.0, bbox_weight=1.0, fill_factor_weight=1.0,
                 pose_loss_fn="l1", mask_loss_fn="l2", encoder_pretrain_steps=0, pose_conditioned_generation_steps=7000,
                 num_classes=1, **unused):
        super().__init__()
        self.perceptual_loss = LPIPSStyle().eval()
        self.perceptual_weight = perceptual_weight
        self.logvar = nn.Parameter(torch.ones(size=()) * logvar_init)
        self.discriminator = NLayerDiscriminator(input_nc=disc_in_channels, n_layers=disc_num_layers).apply(weights_init)
        self.discriminator_iter_start = disc_start
        self.disc_factor, self.discriminator_weight = disc_factor, disc_weight
        self.pose_conditioned_generation_steps = pose_conditioned_generation_steps
        self.encoder_pretrain_steps = encoder_pretrain_steps
        self.pose_weight, self.mask_weight, self.fill_factor_weight = pose_weight, mask_weight, fill_factor_weight
        self.class_weight, self.bbox_weight = class_weight, bbox_weight
        self.kl_weight_obj, self.kl_weight_bbox = kl_weight_obj, kl_weight_bbox
        self.train_on_yaw, self.num_classes = train_on_yaw, num_classes
        self.pose_loss = nn.L1Loss(reduction="none") if pose_loss_fn == "l1" else nn.MSELoss(reduction="none")
        self.rot_loss_fn = nn.SmoothL1Loss(reduction="none")
        self.bbox_loss_fn = nn.MSELoss(reduction="none")
        self.fill_factor_loss_fn = nn.MSELoss(reduction="none")
        self.bbox_distribution_dict = self._create_distribution_from_dataset_stats(dataset_stats)

    def _create_distribution_from_dataset_stats(self, dataset_stats):  # contperceptual.py:82-109
        dist_dict = {}
        for label, stats in dataset_stats.items():
            bbox_means, bbox_logvars = torch.zeros(BBOX_DIM), torch.zeros(BBOX_DIM)
            rot_param = "yaw" if self.train_on_yaw else "v3"
            for idx, key in enumerate(["t1", "t2", "t3", rot_param, "l", "h", "w", "fill_factor"]):
                if key == "yaw":
                    mean, logvar = 0.0, 2 * torch.log(torch.tensor(math.pi))
                elif key == "t1" or key == "t2":
                    mean, logvar = 0.0, 2 * torch.log(torch.tensor(1.0))
                elif key == "fill_factor":
                    mean, logvar = 0.5, 2 * torch.log(torch.tensor(math.sqrt(2)))
                else:
                    mean, logvar = stats[key]
                bbox_means[idx], bbox_logvars[idx] = mean, logvar
            parameters = torch.cat((bbox_means.unsqueeze(1), bbox_logvars.unsqueeze(1)), dim=1)
            dist_dict[label] = DiagonalGaussianDistribution(parameters)
        return dist_dict

    def calculate_adaptive_weight(self, nll_loss, g_loss, last_layer):
        nll_grads = torch.autograd.grad(nll_loss, last_layer, retain_graph=True)[0]
        g_grads = torch.autograd.grad(g_loss, last_layer, retain_graph=True)[0]
        d_weight = torch.norm(nll_grads) / (torch.norm(g_grads) + 1e-4)
        d_weight = torch.clamp(d_weight, 0.0, 1e4).detach()
        return d_weight * self.discriminator_weight

    def pose_terms(self, dec_pose, pose_gt, bbox_gt, fill_factor_gt, class_gt, class_gt_label, bbox_posterior, mask_bg=None):
        """The pose-head terms of contperceptual.py:111-132 (compute_pose_loss), :176-181 (compute_class_loss), :183-189
        (compute_bbox_loss), :191-205 (compute_pose_kl_loss), :207-212 (compute_fill_factor_loss), evaluated the way
        forward() evaluates them (:259-271,278): masked sums over the batch divided by the number of samples whose class id is
        not BACKGROUND_CLASS_IDX, 0 when there is none.  Returns the five unweighted terms and the four per-sample parts of
        the pose term (t1 / t2 / t3 / v3, [B] each)."""
        zero = torch.tensor(0.0)
        if mask_bg is None:
            mask_bg = torch.zeros_like(class_gt)
            mask_bg[class_gt != BACKGROUND_CLASS_IDX] = 1
        nbg = torch.sum(mask_bg)
        pose_rec = dec_pose[:, :POSE_6D_DIM]
        lhw_rec = dec_pose[:, POSE_6D_DIM:POSE_6D_DIM + LHW_DIM]
        fill_factor_rec = dec_pose[:, POSE_6D_DIM + LHW_DIM:BBOX_DIM]
        class_probs = dec_pose[:, BBOX_DIM:]
        class_loss = sigmoid_focal_loss_mean(class_probs, class_gt)
        bbox_loss = self.bbox_loss_fn(bbox_gt, lhw_rec) * mask_bg.unsqueeze(1)
        bbox_loss = torch.sum(bbox_loss) / nbg if nbg > 0 else zero
        # compute_pose_loss(pose_gt, pose_rec, mask_bg): (gt, pred) in the (pred, gt) slots (:269)
        pred, gt = pose_gt, pose_rec
        t1_loss, t2_loss, t3_loss = (self.pose_loss(pred[:, i], gt[:, i]) for i in range(3))
        if self.train_on_yaw:
            v3_loss = self.rot_loss_fn(torch.sin(pred[:, 3]), torch.sin(gt[:, 3]))
        else:
            v3_loss = self.pose_loss(pred[:, 3], gt[:, 3])
        pose_loss = (t1_loss + t2_loss + t3_loss + v3_loss) * mask_bg
        pose_loss = torch.sum(pose_loss) / nbg if nbg > 0 else zero
        fill_factor_loss = self.fill_factor_loss_fn(fill_factor_gt, fill_factor_rec.squeeze()) * mask_bg
        fill_factor_loss = torch.sum(fill_factor_loss) / nbg if nbg > 0 else zero
        pose_kl = torch.zeros(len(class_gt_label), bbox_posterior.mean.size(1))
        for idx, label in enumerate(class_gt_label):
            if label == "background":
                continue
            cur = DiagonalGaussianDistribution(torch.cat((bbox_posterior.mean[idx].unsqueeze(1),
                                                          bbox_posterior.logvar[idx].unsqueeze(1)), dim=1))
            pose_kl[idx] = cur.kl(self.bbox_distribution_dict[label])
        kl_loss_obj_bbox = torch.sum(pose_kl) / nbg if nbg > 0 else zero
        return {"pose_loss": pose_loss, "class_loss": class_loss, "bbox_loss": bbox_loss, "fill_factor_loss": fill_factor_loss,
                "kl_loss_bbox": kl_loss_obj_bbox, "t1": t1_loss, "t2": t2_loss, "t3": t3_loss, "v3": v3_loss}

    def forward(self, rgb_gt, mask_gt, pose_gt, dec_obj, dec_pose, class_gt, class_gt_label, bbox_gt, fill_factor_gt,
                posterior_obj, bbox_posterior, optimizer_idx, global_step, mask_2d_bbox, last_layer=None, split="train"):
        assert mask_gt is None
        zero = torch.tensor(0.0)
        use_pixel_loss = not (global_step < (self.encoder_pretrain_steps + self.pose_conditioned_generation_steps))
        mask_bg = torch.zeros_like(class_gt)
        mask_bg[class_gt != BACKGROUND_CLASS_IDX] = 1
        nbg = torch.sum(mask_bg)
        inputs, reconstructions = rgb_gt, dec_obj
        inputs_rgb, reconstructions_rgb = rgb_gt, reconstructions[:, :3, :, :]
        if mask_2d_bbox is not None:  # :251-257
            inputs = inputs * mask_2d_bbox
            reconstructions = reconstructions * mask_2d_bbox
            inputs_rgb = inputs_rgb * mask_2d_bbox
            reconstructions_rgb = reconstructions_rgb * mask_2d_bbox
        t = self.pose_terms(dec_pose, pose_gt, bbox_gt, fill_factor_gt, class_gt, class_gt_label, bbox_posterior, mask_bg)
        class_loss, bbox_loss, pose_loss, fill_factor_loss = t["class_loss"], t["bbox_loss"], t["pose_loss"], t["fill_factor_loss"]
        weighted_class_loss = self.class_weight * class_loss
        weighted_bbox_loss = self.bbox_weight * bbox_loss
        weighted_pose_loss = self.pose_weight * pose_loss
        mask_loss, weighted_mask_loss = zero, zero  # use_mask_loss is forced off when no mask channel exists (:232,248)
        weighted_fill_factor_loss = self.fill_factor_weight * fill_factor_loss

        # _get_rec_loss :134-145
        if use_pixel_loss:
            rec_loss = torch.abs(inputs_rgb.contiguous() - reconstructions_rgb.contiguous())
        else:
            rec_loss = torch.zeros_like(inputs_rgb.contiguous())
        if self.perceptual_weight > 0:
            p_loss = self.perceptual_loss(inputs_rgb.contiguous(), reconstructions_rgb.contiguous())
            rec_loss = rec_loss + self.perceptual_weight * p_loss
        # _get_nll_loss :147-158
        nll_full = rec_loss / (torch.exp(self.logvar) + 1e-8) + self.logvar
        masked = nll_full * mask_bg.unsqueeze(1).unsqueeze(1).unsqueeze(1)
        nll_loss = torch.sum(masked) / nbg if nbg > 0 else zero
        weighted_nll_loss = nll_loss
        # _get_kl_loss :160-164
        kl_loss_obj = posterior_obj.kl() * mask_bg
        kl_loss_obj = torch.sum(kl_loss_obj) / nbg if nbg > 0 else zero
        kl_loss_obj_bbox = t["kl_loss_bbox"]

        bg4 = mask_bg.unsqueeze(1).unsqueeze(1).unsqueeze(1)
        if optimizer_idx == 0:
            logits_fake = self.discriminator(reconstructions.contiguous()) * bg4
            g_loss = -torch.mean(logits_fake)
            if self.disc_factor > 0.0 and global_step > self.encoder_pretrain_steps:
                try:
                    d_weight = self.calculate_adaptive_weight(nll_loss, g_loss, last_layer=last_layer)
                except RuntimeError:   # no graph under torch.no_grad(): validation (contperceptual.py:295-299)
                    assert not self.training
                    d_weight = torch.tensor(0.0)
            else:
                d_weight = torch.tensor(0.0)
            disc_factor = adopt_weight(self.disc_factor, global_step, threshold=self.discriminator_iter_start)
            pose_only = (weighted_pose_loss + weighted_class_loss + weighted_bbox_loss + weighted_fill_factor_loss
                         + (self.kl_weight_bbox * kl_loss_obj_bbox))
            if self.encoder_pretrain_steps == -1:
                loss = pose_only
            elif global_step > self.encoder_pretrain_steps:
                loss = (weighted_pose_loss + weighted_mask_loss + weighted_nll_loss + weighted_class_loss + weighted_bbox_loss
                        + weighted_fill_factor_loss + (self.kl_weight_obj * kl_loss_obj)
                        + (self.kl_weight_bbox * kl_loss_obj_bbox) + d_weight * disc_factor * g_loss)
            else:
                loss = pose_only
            log = {"total_loss": loss.detach(), "kl_loss_obj": kl_loss_obj.detach(), "nll_loss": nll_loss.detach(),
                   "rec_loss": rec_loss.detach().mean(), "d_weight": d_weight.detach(), "g_loss": g_loss.detach(),
                   "pose_loss": pose_loss.detach(), "class_loss": class_loss.detach(), "bbox_loss": bbox_loss.detach(),
                   "kl_loss_bbox": kl_loss_obj_bbox.detach(), "fill_factor_loss": fill_factor_loss.detach()}
            return loss, {"%s/%s" % (split, k): v for k, v in log.items()}
        logits_real = self.discriminator(inputs.contiguous().detach()) * bg4
        logits_fake = self.discriminator(reconstructions.contiguous().detach()) * bg4
        disc_factor = adopt_weight(self.disc_factor, global_step, threshold=self.discriminator_iter_start)
        d_loss = disc_factor * hinge_d_loss(logits_real, logits_fake)
        return d_loss, {"%s/disc_loss" % split: d_loss.detach(), "%s/logits_real" % split: logits_real.detach().mean(),
                        "%s/logits_fake" % split: logits_fake.detach().mean()}

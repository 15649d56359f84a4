"""PatchGAN discriminator and the LPIPS-style perceptual network.

* NLayerDiscriminator / weights_init: [UPSTREAM] taming/modules/discriminator/model.py (pix2pix PatchGAN):
  Conv(4x4,s2,p1)+LeakyReLU(0.2), two Conv(4x4,s2,p1,no bias)+BatchNorm2d+LeakyReLU, one Conv(4x4,s1,p1,no bias)+BN+
  LeakyReLU, Conv(4x4,s1,p1) -> logits [B,1,30,30] at 256x256 (src/modules/losses/contperceptual.py:285).
  state_dict keys main.{0,2,3,5,6,8,9,11} as in the reference checkpoint layout.
* LPIPSStyle: [UPSTREAM] taming/modules/losses/lpips.py structure (ScalingLayer, VGG16 feature slices relu1_2 ... relu5_3,
  channel-unit-normalise, squared difference, 1x1 "lin" heads, spatial mean, sum over the five taps).  The real LPIPS
  weights are downloads (torchvision VGG16 + vgg.pth) that do not exist offline, so the weights here are seeded
  synthetic and frozen unless `LPIPSStyle.load_weights()` is given the two files or a checkpoint carrying
  `loss.perceptual_loss.*` is loaded over them: "LPIPS-style", as BASELINE.json words it.
"""
import torch
import torch.nn as nn

from . import ops


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm") != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class Conv4x4(nn.Conv2d):
    """4x4 convolution parameter holder, forward through the HIP k x k kernel."""

    def __init__(self, cin, cout, stride, bias=True):
        super().__init__(cin, cout, kernel_size=4, stride=stride, padding=1, bias=bias)

    def forward(self, x):
        return ops.conv4x4(x, self.weight, self.bias, self.stride[0])


class BatchNormLReLU(nn.BatchNorm2d):
    """BatchNorm2d (batch statistics in training, running statistics in eval; not synchronised across ranks,
    as in the reference) fused with the LeakyReLU(0.2) that follows it in the PatchGAN."""

    def forward(self, x):
        return ops.batchnorm_lrelu(x, self, 0.2)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, x):
        return ops.leaky_relu(x, self.negative_slope)


class _Fused(nn.Identity):
    """Placeholder keeping nn.Sequential indices equal to upstream where an activation was fused away."""


class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc=3, ndf=64, n_layers=3, use_actnorm=False):
        super().__init__()
        if use_actnorm:
            raise NotImplementedError("ActNorm discriminator is not used by the OD-VAE configs")
        seq = [Conv4x4(input_nc, ndf, 2, bias=True), LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [Conv4x4(ndf * prev, ndf * mult, 2, bias=False), BatchNormLReLU(ndf * mult), _Fused()]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [Conv4x4(ndf * prev, ndf * mult, 1, bias=False), BatchNormLReLU(ndf * mult), _Fused()]
        seq += [Conv4x4(ndf * mult, 1, 1, bias=True)]
        self.main = nn.Sequential(*seq)

    def forward(self, input):
        return self.main(input)


class _VggConv(nn.Conv2d):
    def __init__(self, cin, cout):
        super().__init__(cin, cout, kernel_size=3, padding=1)

    def forward(self, x):
        return ops.conv3x3(x, self.weight, self.bias, None, 0, relu=True)


# [UPSTREAM] taming/modules/losses/lpips.py `vgg16`: torchvision's vgg16().features split at the five taps
# relu1_2 / relu2_2 / relu3_3 / relu4_3 / relu5_3; each conv keeps its torchvision feature index as its module name
# (slice1 = features[0:4], slice2 = [4:9], slice3 = [9:16], slice4 = [16:23], slice5 = [23:30]; ReLU and MaxPool carry no
# parameters), so the keys read net.slice{k}.{idx}.{weight,bias}.
VGG16_SLICES = [
    ("slice1", [(0, 3, 64), (2, 64, 64)]),
    ("slice2", [(5, 64, 128), (7, 128, 128)]),
    ("slice3", [(10, 128, 256), (12, 256, 256), (14, 256, 256)]),
    ("slice4", [(17, 256, 512), (19, 512, 512), (21, 512, 512)]),
    ("slice5", [(24, 512, 512), (26, 512, 512), (28, 512, 512)]),
]
LPIPS_CHNS = [64, 128, 256, 512, 512]


class ScalingLayer(nn.Module):
    """[UPSTREAM] lpips.ScalingLayer: (x - shift) / scale with the two buffers in the state_dict."""

    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, x):
        return ops.scale_shift(x, self.shift, self.scale)


class _Vgg16Features(nn.Module):
    def __init__(self):
        super().__init__()
        for name, convs in VGG16_SLICES:
            sl = nn.Module()
            for idx, cin, cout in convs:
                sl.add_module(str(idx), _VggConv(cin, cout))
            self.add_module(name, sl)

    def forward(self, h):
        outs = []
        for k, (name, convs) in enumerate(VGG16_SLICES):
            if k > 0:
                h = ops.maxpool2x2(h)     # features[4], [9], [16], [23] open slices 2..5
            sl = getattr(self, name)
            for idx, _, _ in convs:
                h = getattr(sl, str(idx))(h)   # conv + the ReLU that follows it (fused epilogue)
            outs.append(h)
        return outs


class NetLinLayer(nn.Module):
    """[UPSTREAM] lpips.NetLinLayer: Sequential(Dropout, Conv2d(chn_in, 1, 1, bias=False)) -> key lin{k}.model.1.weight.
    The Dropout holds the key index; it never drops anything here: LPIPSStyle pins itself to eval mode (see its `train`), and
    the 1x1 conv is folded into ops.lpips_layer_distance."""

    def __init__(self, chn_in, chn_out=1, use_dropout=True):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)

    @property
    def weight(self):
        return self.model[-1].weight


class LPIPSStyle(nn.Module):
    """d(x, y) = sum_k mean_hw lin_k( (normalize(f_k(x)) - normalize(f_k(y)))^2 ), shape [B,1,1,1].

    Module tree = [UPSTREAM] taming LPIPS (`scaling_layer`, `net.slice{1..5}.{idx}`, `lin{0..4}.model.1`), so the
    `loss.perceptual_loss.*` entries of a reference checkpoint load with strict=True.  Construction fills seeded synthetic
    weights (the real ones are two downloads that do not exist offline); `load_weights` takes the two upstream files."""

    def __init__(self, seed=1234, use_dropout=True):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.chns = list(LPIPS_CHNS)
        self.net = _Vgg16Features()
        gen = torch.Generator().manual_seed(seed)
        with torch.no_grad():   # He-normal stand-in for the pretrained VGG16 weights (no network here)
            for name, convs in VGG16_SLICES:
                for idx, cin, cout in convs:
                    conv = getattr(getattr(self.net, name), str(idx))
                    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (9 * cin)) ** 0.5)
                    conv.bias.zero_()
        for k, chn in enumerate(self.chns):
            lin = NetLinLayer(chn, use_dropout=use_dropout)
            with torch.no_grad():
                lin.weight.copy_(torch.rand(lin.weight.shape, generator=gen) / chn)  # non-negative, like LPIPS lins
            self.add_module("lin%d" % k, lin)
        for p in self.parameters():
            p.requires_grad = False
        self._synthetic_mark = self._fingerprint()

    def train(self, mode=True):
        """DELIBERATE DEVIATION: the perceptual net stays in eval mode whatever the parent module is switched to.
        [UPSTREAM] builds it as `LPIPS().eval()` but defines no `train` override, so a Lightning fit's recursive `model.train()`
        re-arms the `Dropout(0.5)` in front of every lin layer and the reference's training-time LPIPS value is that of a random
        half of the feature channels, doubled -- an unbiased, noisy estimate of the eval-mode distance computed here (and by
        every LPIPS evaluation outside training).  The metric is frozen (`requires_grad = False`), so nothing else depends on
        the mode."""
        return super().train(False)

    # ---- weights ---------------------------------------------------------------------------------------------------------------
    def _fingerprint(self):
        with torch.no_grad():
            return (float(self.lin0.weight.double().sum()), float(self.net.slice1._modules["0"].weight.double().sum()))

    def has_synthetic_weights(self):
        """True while the construction-time stand-in weights are still in place (nothing was loaded over them)."""
        now = self._fingerprint()   # summation order differs between host and device: compare with a tolerance
        return all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(now, self._synthetic_mark))

    def load_weights(self, vgg16=None, lins=None):
        """The two files [UPSTREAM] LPIPS.__init__ fetches: `vgg16` = torchvision vgg16 state_dict (keys features.{idx}.*;
        classifier.* ignored) and `lins` = taming's vgg.pth (keys lin{k}.model.1.weight).  Each may be a path or a dict.
        Raises KeyError / RuntimeError when a tensor is missing or has the wrong shape."""
        def _sd(x):
            return torch.load(x, map_location="cpu") if isinstance(x, (str, bytes)) or hasattr(x, "__fspath__") else x
        with torch.no_grad():
            if vgg16 is not None:
                sd = _sd(vgg16)
                for name, convs in VGG16_SLICES:
                    for idx, _, _ in convs:
                        conv = getattr(getattr(self.net, name), str(idx))
                        conv.weight.copy_(sd["features.%d.weight" % idx])
                        conv.bias.copy_(sd["features.%d.bias" % idx])
            if lins is not None:
                sd = _sd(lins)
                for k in range(len(self.chns)):
                    getattr(self, "lin%d" % k).weight.copy_(sd["lin%d.model.1.weight" % k])
        ops.PACK_CACHE.bump()
        return self

    def features(self, x):
        return self.net(self.scaling_layer(x))

    def forward(self, input, target):
        f0, f1 = self.features(input), self.features(target)
        total = None
        for k in range(len(self.chns)):
            d = ops.lpips_layer_distance(f0[k], f1[k], getattr(self, "lin%d" % k).weight)  # [B]
            total = d if total is None else total + d
        return total.reshape(-1, 1, 1, 1)

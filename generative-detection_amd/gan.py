"""PatchGAN discriminator and the LPIPS-style perceptual network.

* NLayerDiscriminator / weights_init: [UPSTREAM] taming/modules/discriminator/model.py (pix2pix PatchGAN):
  Conv(4x4,s2,p1)+LeakyReLU(0.2), two Conv(4x4,s2,p1,no bias)+BatchNorm2d+LeakyReLU, one Conv(4x4,s1,p1,no bias)+BN+
  LeakyReLU, Conv(4x4,s1,p1) -> logits [B,1,30,30] at 256x256 (src/modules/losses/contperceptual.py:285).
  state_dict keys main.{0,2,3,5,6,8,9,11} as in the reference checkpoint layout.
* LPIPSStyle: [UPSTREAM] taming/modules/losses/lpips.py structure (ScalingLayer, VGG16 feature slices relu1_2 ... relu5_3,
  channel-unit-normalise, squared difference, 1x1 "lin" heads, spatial mean, sum over the five taps).  The real LPIPS
  weights are downloads (torchvision VGG16 + vgg.pth) that do not exist offline, so the weights here are seeded
  synthetic and frozen unless `load_weights()` is given files: "LPIPS-style", as BASELINE.json words it.
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


VGG16_CFG = [(64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512)]


class LPIPSStyle(nn.Module):
    """d(x, y) = sum_k mean_hw lin_k( (normalize(f_k(x)) - normalize(f_k(y)))^2 ), shape [B,1,1,1]."""

    def __init__(self, seed=1234):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])
        gen = torch.Generator().manual_seed(seed)
        self.slices = nn.ModuleList()
        cin = 3
        for widths in VGG16_CFG:
            convs = nn.ModuleList()
            for cout in widths:
                conv = _VggConv(cin, cout)
                with torch.no_grad():  # He-normal stand-in for the pretrained VGG16 weights (no network here)
                    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (9 * cin)) ** 0.5)
                    conv.bias.zero_()
                convs.append(conv)
                cin = cout
            self.slices.append(convs)
        self.lins = nn.ModuleList()
        for widths in VGG16_CFG:
            lin = nn.Conv2d(widths[-1], 1, 1, bias=False)
            with torch.no_grad():
                lin.weight.copy_(torch.rand(lin.weight.shape, generator=gen) / widths[-1])  # non-negative, like LPIPS lins
            self.lins.append(lin)
        for p in self.parameters():
            p.requires_grad = False

    def features(self, x):
        h = ops.scale_shift(x, self.shift, self.scale)
        outs = []
        for k, convs in enumerate(self.slices):
            if k > 0:
                h = ops.maxpool2x2(h)
            for conv in convs:
                h = conv(h)
            outs.append(h)
        return outs

    def forward(self, input, target):
        f0, f1 = self.features(input), self.features(target)
        total = None
        for k in range(len(VGG16_CFG)):
            d = ops.lpips_layer_distance(f0[k], f1[k], self.lins[k].weight)  # [B]
            total = d if total is None else total + d
        return total.reshape(-1, 1, 1, 1)

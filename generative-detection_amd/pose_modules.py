"""Pose head MLPs (spatial-VAE style), interface- and state_dict-compatible with
src/modules/autoencodermodules/pose_encoder.py:59-131 (PoseEncoderSpatialVAE; legacy PoseEncoder :14-57) and
src/modules/autoencodermodules/pose_decoder.py:60-97 (PoseDecoderSpatialVAE; legacy PoseDecoder :12-57).
The nn.Linear modules only hold the parameters (state_dict keys as upstream); on the device every dense layer runs on
ops.linear_act (csrc/linear_f32.hip: split-K 32 x 32 MFMA tiles, bias + activation fused in the reduce), SURVEY.md 8(a) row a23.
Host tensors (the CPU-side interface tests) take torch's own path.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

_FUSABLE = {nn.Tanh: "tanh", nn.SiLU: "swish", nn.ReLU: "relu"}


def _run_layers(seq, x):
    """nn.Sequential of Linear / activation modules: on the device each Linear (+ the activation behind it) is one
    ops.linear_act call; anything else (Softplus of the legacy heads) runs as the module it is."""
    if not x.is_cuda:
        return seq(x)
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            act = _FUSABLE.get(type(mods[i + 1])) if i + 1 < len(mods) else None
            x = ops.linear_act(x, m.weight, m.bias, act)
            i += 2 if act else 1
        else:
            x = m(x)
            i += 1
    return x

POSE_DIM = 4
LHW_DIM = 3
FILL_FACTOR_DIM = 1


def _act_class(name, table, default):
    return table.get(name, default)


def _mlp3(d_in, d_h1, d_h2, d_out, activation):
    if activation == "relu":
        act = nn.ReLU()
    elif activation == "softplus":
        act = nn.Softplus()
    else:
        raise ValueError("Invalid activation function. Please provide a valid activation function in ['relu', 'softplus'].")
    return nn.Sequential(nn.Linear(d_in, d_h1), act, nn.Linear(d_h1, d_h2), act, nn.Linear(d_h2, d_out))


class PoseEncoder(nn.Module):
    """Legacy pose -> image-feature MLP (pose_encoder.py:14-57): hidden widths enc/8 and enc/4."""

    def __init__(self, enc_feat_dims, pose_feat_dims, activation="relu"):
        super().__init__()
        self.fc = _mlp3(pose_feat_dims, enc_feat_dims // 8, enc_feat_dims // 4, enc_feat_dims, activation)

    def forward(self, x):
        return _run_layers(self.fc, x)


class PoseDecoder(nn.Module):
    """Legacy image-feature -> pose MLP (pose_decoder.py:12-57): hidden widths enc/4 and enc/8."""

    def __init__(self, enc_feat_dims, pose_feat_dims, activation="relu"):
        super().__init__()
        self.fc = _mlp3(enc_feat_dims, enc_feat_dims // 4, enc_feat_dims // 8, pose_feat_dims, activation)

    def forward(self, x):
        return _run_layers(self.fc, x)


class PoseEncoderSpatialVAE(nn.Module):
    """19-d (pose, lhw, fill, class logits) -> num_channels*n*m feature map, conditioned on a fixed n x m
    coordinate grid: h = coord_linear(grid) + tile(latent_linear(z)); y = layers(h)."""

    def __init__(self, num_classes=2, num_channels=16, n=16, m=16, activation="swish", hidden_dim=500, num_layers=2):
        super().__init__()
        act = _act_class(activation, {"swish": nn.SiLU, "tanh": nn.Tanh}, nn.ReLU)
        self.num_channels, self.n, self.m = num_channels, n, m
        self.in_dim = 2
        self.num_coords = n * m
        self.feat_size = 4
        self.h_dim = self.num_coords * self.feat_size
        self.x_dim = self.in_dim * self.num_coords
        self.z_dim = POSE_DIM + LHW_DIM + FILL_FACTOR_DIM + num_classes
        self.coord_linear = nn.Linear(self.x_dim, self.h_dim)
        if self.z_dim > 0:
            self.latent_linear = nn.Linear(self.z_dim, self.feat_size, bias=False)
        seq = [act()]
        width_in = self.h_dim
        for _ in range(1, num_layers):
            seq += [nn.Linear(width_in, hidden_dim), act()]
            width_in = hidden_dim
        seq.append(nn.Linear(hidden_dim, num_channels * n * m))
        self.layers = nn.Sequential(*seq)
        # grid: x runs -1..1 left to right, y runs 1..-1 top to bottom; a plain attribute (not a buffer), as in the reference
        gx, gy = np.meshgrid(np.linspace(-1, 1, m), np.linspace(1, -1, n))
        self.x = torch.from_numpy(np.stack([gx.ravel(), gy.ravel()], 1)).float()

    def forward(self, z):
        if z.dim() < 2:
            z = z.unsqueeze(0)
        b = z.size(0)
        if z.is_cuda:
            # the grid is the same for every row: one row of coord_linear, broadcast over the batch.  Its device copy is made once:
            # `self.x` is a host tensor (a plain attribute, as in the reference), and `.to(device)` of pageable host memory blocks the
            # host until the stream has drained -- one full pipeline stall per forward (found with torch.cuda.set_sync_debug_mode).
            key = (z.device, z.dtype)
            if getattr(self, "_x_dev_key", None) != key:
                self._x_dev, self._x_dev_key = self.x.to(z).reshape(1, self.x_dim), key
            h = ops.linear_act(self._x_dev, self.coord_linear.weight, self.coord_linear.bias)
            h_z = ops.linear_act(z, self.latent_linear.weight)
        else:
            grid = self.x.to(z).expand(b, self.num_coords, self.in_dim).reshape(b, self.x_dim)
            h = self.coord_linear(grid)
            h_z = self.latent_linear(z)                                 # b x feat_size
        h = h + h_z.unsqueeze(1).expand(b, self.num_coords, self.feat_size).reshape(b, self.h_dim)
        return _run_layers(self.layers, h)


class PoseDecoderSpatialVAE(nn.Module):
    """flattened feature map -> 2*(pose+lhw+fill) moments + class logits."""

    def __init__(self, num_classes=2, num_channels=16, n=16, m=16, activation="tanh", **kwargs):
        super().__init__()
        act = nn.Tanh if activation == "tanh" else nn.ReLU
        self.n = num_channels * n * m
        self.latent_dim = 2 * (POSE_DIM + LHW_DIM + FILL_FACTOR_DIM) + num_classes
        hidden_dim = kwargs.get("hidden_dim", 500)
        num_layers = kwargs.get("num_layers", 2)
        if kwargs.get("resid", False):
            raise NotImplementedError("resid=True needs ResidLinear, which the reference never defines (pose_decoder.py:87)")
        seq = [nn.Linear(self.n, hidden_dim), act()]
        for _ in range(1, num_layers):
            seq += [nn.Linear(hidden_dim, hidden_dim), act()]
        seq.append(nn.Linear(hidden_dim, self.latent_dim))
        self.layers = nn.Sequential(*seq)

    def forward(self, x):
        return _run_layers(self.layers, x)

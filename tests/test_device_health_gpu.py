"""Device-side health counters reach the training loop.

Two kernels can report trouble only through a word in device memory: the opt-in team mode of the read-once GroupNorm backward
(csrc/groupnorm.hip: a team barrier that gives up after 0.2 s lets the block run on with partial sums -- wrong gradients) and the folded
attention softmax (ops._Attention: a row whose Cauchy-Schwarz bound underflows takes an exact fallback -- correct, slower).  The Trainer
reads both through odvae_device_health wherever the host already waits: after validation, before a checkpoint is written, at the end of
fit.  A barrier timeout raises DeviceHealthError and no checkpoint is written; softmax fallbacks are reported once per new count."""
import ctypes
import os
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
YAML = os.path.join(os.path.dirname(__file__), "golden", "autoencoder_kl_16x16x16.yaml")


def _trainer():
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    torch.manual_seed(0)
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32).to("cuda:0").train()
    model._global_step = 1
    return Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,)), synthetic.make_batch(2, 64, seed=3)


def _counters(L):
    gn, at = ctypes.c_int(-1), ctypes.c_int(-1)
    assert L.odvae_device_health(ctypes.byref(gn), ctypes.byref(at), 0, 0) == 0
    return gn.value, at.value


def test_a_real_attention_fallback_is_counted_on_the_device(hip_lib):
    """Scores far beyond what a Cauchy-Schwarz bound can cover in f32 (|q||k| ~ 1e4 while q.k ~ 0 on most rows): every exponential of those
    rows underflows against the bound, the flag goes up, the predicated launches redo the block exactly -- and count themselves."""
    from odvae_amd import ops
    if not ops.ATTN_FOLDED_SOFTMAX:
        pytest.skip("folded softmax switched off")
    _, before = _counters(hip_lib)
    n, c, h, w = 1, 64, 16, 16
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(n, 3 * c, h, w, generator=g)
    qkv[:, :2 * c] *= 60.0                                  # |q_i| |k_j| C^-1/2 ~ 60 * 60 * 8 = 3e4 >> 88 (f32 exp range), q_i . k_j ~ N(0, 3e4^2 / 64)
    y = ops.attention_qkv(qkv.to("cuda:0"))
    s = torch.einsum("ncq,nck->nqk", qkv[:, :c].reshape(n, c, -1).double(), qkv[:, c:2 * c].reshape(n, c, -1).double()) * c ** -0.5
    want = torch.einsum("nqk,nck->ncq", torch.softmax(s, dim=-1), qkv[:, 2 * c:].reshape(n, c, -1).double()).reshape(n, c, h, w)
    assert (y.cpu().double() - want).abs().max().item() < 1e-3 * want.abs().max().item()      # the fallback's result is the exact softmax
    assert int(ops._ATTN_LAST_FLAG.item()) == 1
    _, after = _counters(hip_lib)
    assert after == before + 1


def test_trainer_reports_softmax_fallbacks_once_and_raises_on_a_barrier_timeout(hip_lib, tmp_path):
    from odvae_amd.trainer import DeviceHealthError
    trainer, batch = _trainer()
    dev = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in batch.items()}
    trainer.check_device_health()                                             # baseline: whatever earlier tests left in the counters
    base = dict(trainer.device_health)
    trainer.fit([dev], max_batches=1)                                         # a clean step: nothing new, no warning, no error
    assert trainer.device_health == base
    # test hook: two fallbacks "happen" on the device
    assert hip_lib.odvae_device_health(None, None, 0, 2) == 0
    with pytest.warns(RuntimeWarning, match="2 attention block"):
        trainer.validate([dev])
    assert trainer.device_health["attn_softmax_fallbacks"] == base["attn_softmax_fallbacks"] + 2
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                        # ... and are not reported a second time
        trainer.validate([dev])
    # test hook: one GroupNorm team barrier "gives up"
    assert hip_lib.odvae_device_health(None, None, 1, 0) == 0
    path = str(tmp_path / "last.ckpt")
    with pytest.raises(DeviceHealthError, match="GroupNorm team-barrier"):
        trainer.save_checkpoint(path)
    assert not os.path.exists(path) and not os.path.exists(path + ".part")    # nothing written from a poisoned state
    # the error is reported once per new timeout: the counter is sticky on the device, the trainer remembers what it has seen
    trainer.save_checkpoint(path)
    assert os.path.exists(path)
    assert trainer.device_health["gn_barrier_timeouts"] == base["gn_barrier_timeouts"] + 1
    assert hip_lib.odvae_groupnorm_fused_timeouts() == trainer.device_health["gn_barrier_timeouts"]
    assert hip_lib.odvae_device_health(None, None, -1, 0) == 0               # leave the device counter as other tests expect it: 0
    assert hip_lib.odvae_groupnorm_fused_timeouts() == 0

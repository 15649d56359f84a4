"""SURVEY.md 8(f) rank 1, the writer side (train.py:228-249 ModelCheckpoint; src/models/autoencoder.py:332-363 validation_step):
what `Trainer.save_checkpoint` writes, the reference's model + torch.optim.Adam can resume from, and the reverse -- checked by taking
the SAME third optimizer step on both sides of the hand-over; `Trainer.validate` against the oracle's validation_step means.
Tolerances are those of tests/test_model_gpu.py (loss 2e-3, weights 2.2 lr per step + 5e-3 of the tensor's largest entry)."""
import os

import pytest
import torch

from test_model_gpu import build_pair

pytestmark = pytest.mark.gpu


def _batch(step, height=64):
    from odvae_amd import synthetic
    return synthetic.make_batch(2, height, seed=500 + step), synthetic.make_noise(2, height // 16, dropout_p=0.7, seed=600 + step)


def _hip_step(trainer, model, step):
    batch, noise = _batch(step)
    model.injected_noise = noise
    return trainer.training_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, step)[0].item()


def _ref_step(ref, opts, step):
    from oracle.autoencoder import train_batch
    batch, noise = _batch(step)
    return train_batch(ref, opts, batch, {0: noise}, optimizer_indices=(0,), clip=1.0)[0][0].item()


def _weights_close(model, ref, steps):
    lr, ref_sd = model.learning_rate, ref.state_dict()
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32 and k.startswith(("encoder", "decoder", "quant", "post_quant", "pose_")):
            diff = (v.detach().cpu().double() - ref_sd[k].double()).abs().max().item()
            assert diff <= 2.2 * lr * steps + 5e-3 * ref_sd[k].abs().max().item(), (k, diff)


def test_hip_checkpoint_resumes_under_the_reference_optimizer(hip_lib, tmp_path):
    """HIP model: two optimizer steps -> last.ckpt (full Lightning layout) -> a FRESH oracle model + torch.optim.Adam load it (strict)
    -> both sides take step three.  The oracle that resumed from the file must land where the HIP trainer lands, and where an oracle
    that ran all three steps itself lands."""
    from odvae_amd.trainer import Trainer
    model, ref_all = build_pair()
    model.train(); ref_all.train()
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,))
    opts_all = ref_all.configure_optimizers()
    for step in range(2):
        a, b = _hip_step(trainer, model, step), _ref_step(ref_all, opts_all, step)
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (step, a, b)
    path = trainer.save_checkpoint(os.path.join(tmp_path, "ckpt", "last.ckpt"))
    ckpt = torch.load(path, map_location="cpu")
    assert set(ckpt) >= {"epoch", "global_step", "pytorch-lightning_version", "state_dict", "optimizer_states", "lr_schedulers", "callbacks"}
    assert ckpt["global_step"] == 2 and ckpt["epoch"] == 0 and ckpt["lr_schedulers"] == [] and len(ckpt["optimizer_states"]) == 2
    assert all(not t.is_cuda for t in ckpt["state_dict"].values())
    # a fresh reference-side model resumes from the file alone
    _, ref_new = build_pair()
    ref_new.train()
    torch.manual_seed(99)
    for p in ref_new.parameters():          # make sure nothing survives from the shared seed
        p.data.normal_()
    res = ref_new.load_state_dict(ckpt["state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ref_new.global_step = ckpt["global_step"]
    opts_new = ref_new.configure_optimizers()
    for o, sd in zip(opts_new, ckpt["optimizer_states"]):
        o.load_state_dict(sd)               # torch.optim.Adam accepts FusedAdam's state layout as its own
    # the file carries torch.optim.Adam's own bookkeeping: the same parameters have state, with the same per-parameter step counts, as in
    # the oracle optimizer that ran the two steps itself (the decoder joined at the second step: the very first total is pose-only)
    st, st_all = opts_new[0].state_dict()["state"], opts_all[0].state_dict()["state"]
    assert len(st) > 100 and set(st) == set(st_all)
    assert all(float(st[k]["step"]) == float(st_all[k]["step"]) for k in st)
    assert {float(v["step"]) for v in st.values()} == {1.0, 2.0}
    top = max(v["exp_avg"].abs().max().item() for v in st_all.values())
    for k in st:        # first moments: 5e-3 of the tensor's largest entry (a gradient that is zero up to rounding is compared on the global scale)
        m = st_all[k]["exp_avg"]
        assert (st[k]["exp_avg"] - m).abs().max().item() <= 5e-3 * max(m.abs().max().item(), 1e-3 * top), k
    l_hip, l_new, l_all = _hip_step(trainer, model, 2), _ref_step(ref_new, opts_new, 2), _ref_step(ref_all, opts_all, 2)
    assert abs(l_new - l_all) <= 2e-3 * max(1.0, abs(l_all)), (l_new, l_all)
    assert abs(l_hip - l_new) <= 2e-3 * max(1.0, abs(l_new)), (l_hip, l_new)
    assert model.global_step == ref_new.global_step == ref_all.global_step == 3
    _weights_close(model, ref_new, 3)
    # the resumed oracle and the uninterrupted one differ only by what the HIP steps differ from the oracle's (the file carried everything)
    for (k, a), (_, b) in zip(ref_new.state_dict().items(), ref_all.state_dict().items()):
        if a.dtype == torch.float32 and k.startswith(("encoder", "decoder", "quant", "post_quant", "pose_")):
            assert (a - b).abs().max().item() <= 2.2 * model.learning_rate * 3 + 5e-3 * b.abs().max().item(), k


def test_reference_checkpoint_resumes_on_the_hip_trainer(hip_lib, tmp_path):
    """The reverse: the oracle + torch.optim.Adam run two steps and write a Lightning-layout checkpoint; a fresh HIP model + Trainer
    `load_checkpoint` it (weights strict, optimizer moments and step counts into the FusedAdam arenas, global_step) and take step three."""
    from odvae_amd.trainer import Trainer
    _, ref = build_pair()
    ref.train()
    opts = ref.configure_optimizers()
    for step in range(2):
        _ref_step(ref, opts, step)
    path = os.path.join(tmp_path, "epoch=000000.ckpt")
    torch.save({"epoch": 0, "global_step": ref.global_step, "pytorch-lightning_version": "1.9.0", "state_dict": ref.state_dict(),
                "optimizer_states": [o.state_dict() for o in opts], "lr_schedulers": [], "callbacks": {}}, path)
    model, _ = build_pair()
    with torch.no_grad():
        for p in model.parameters():
            p.normal_()
    model.train()
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,))
    res = trainer.load_checkpoint(path)
    assert not res.missing_keys and not res.unexpected_keys
    assert model.global_step == 2 and trainer.current_epoch == 0
    l_hip, l_ref = _hip_step(trainer, model, 2), _ref_step(ref, opts, 2)
    assert abs(l_hip - l_ref) <= 2e-3 * max(1.0, abs(l_ref)), (l_hip, l_ref)
    _weights_close(model, ref, 1)           # one HIP step on top of the oracle's two
    # a weights-only file (the reference's default, train.py:236) leaves the optimizers fresh
    wpath = trainer.save_checkpoint(os.path.join(tmp_path, "w.ckpt"), weights_only=True)
    assert "optimizer_states" not in torch.load(wpath, map_location="cpu")


def test_validate_matches_the_oracle_epoch_means_and_feeds_modelcheckpoint(hip_lib, tmp_path):
    """`Trainer.validate` over three batches: every logged `val/*` scalar is the mean over the batches of what the oracle's
    validation_step logs (eval mode, no gradients, both loss branches, autoencoder.py:332-363), the model returns to training mode,
    and the yaml-shaped ModelCheckpoint (monitor val/rec_loss, top 3, save_last, weights only) writes epoch=... + last.ckpt."""
    from odvae_amd.callbacks import default_modelcheckpoint
    from odvae_amd.trainer import Trainer
    model, ref = build_pair()
    model.monitor = "val/rec_loss"
    model.train(); ref.eval()
    model._global_step = ref.global_step = 5
    model.loss.log_exact_g_loss = True
    cb = default_modelcheckpoint(model, os.path.join(tmp_path, "checkpoints"))
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,), callbacks=[cb])

    def loader():
        for i in range(3):
            batch, noise = _batch(10 + i)
            model.injected_noise = noise
            yield batch
    got = trainer.validate(loader())
    assert model.training
    want = {}
    with torch.no_grad():
        for i in range(3):
            batch, noise = _batch(10 + i)
            for k, v in ref.validation_step(batch, noise).items():
                want[k] = want.get(k, 0.0) + float(v) / 3.0
    assert "val/rec_loss" in got and set(want) <= set(got), sorted(set(want) - set(got))
    for k, v in want.items():
        assert abs(float(got[k]) - v) <= 2e-3 * max(1.0, abs(v)), (k, float(got[k]), v)
    files = sorted(os.listdir(os.path.join(tmp_path, "checkpoints")))
    assert files == ["epoch=000000.ckpt", "last.ckpt"], files
    ck = torch.load(os.path.join(tmp_path, "checkpoints", "last.ckpt"), map_location="cpu")
    assert "optimizer_states" not in ck and ck["global_step"] == 5          # save_weights_only, as train.py:236 configures
    assert abs(cb.best_model_score - float(got["val/rec_loss"])) < 1e-6

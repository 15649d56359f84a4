"""Edge cases of the training step the reference's own code paths single out, HIP path vs oracle:
samples of class id 1 are masked out of every loss term (BACKGROUND_CLASS_IDX = 1, contperceptual.py:17,228), a batch
with no unmasked sample takes the `else 0` branches (:129,155-163,187,204,210), the 2-d box mask multiplies input and
reconstruction (autoencoder.py:252-257), the untouched yaml starts in the encoder-pretraining phase (decoder skipped,
dropout p = 1; autoencoder.py:184-206,246-247), and a batch of one.  Tolerances as in test_model_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_model_gpu import build_pair, rel  # noqa: E402


def _batch(classes, size=64, seed=11, holes=True):
    from odvae_amd import synthetic
    batch = synthetic.make_batch(len(classes), size, seed=seed)
    batch["class_id"] = torch.tensor(classes, dtype=torch.int64)
    batch["class_name"] = [synthetic.LABELS[c] for c in classes]
    if holes:   # a real 2-d box mask: zero outside a rectangle that differs per sample
        m = torch.zeros(len(classes), 1, size, size)
        for i in range(len(classes)):
            m[i, :, 4 + 3 * i:size - 6, 2 * i:size - 9 - i] = 1.0
        batch["mask_2d_bbox"] = m
    return batch


def _compare(model, ref, batch, noise, grad_tol=5e-3, global_step=1):
    model.train(); ref.train()
    model.loss.log_exact_g_loss = True                   # log g_loss = -mean D(x_rec) as the reference does with the discriminator off
    model._global_step = ref.global_step = global_step   # > encoder_pretrain_steps (0): the rec / KL terms are in the total
    model.zero_grad(set_to_none=True)
    model.injected_noise = noise
    loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, 0)
    loss_ref, log_ref, _ = ref.training_step(batch, 0, noise)
    assert torch.isfinite(loss).all()
    assert abs(loss.item() - loss_ref.item()) <= 1e-3 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    for key, want in log_ref.items():
        got = model.logged_metrics.get(key)
        if got is None or not torch.is_tensor(want) or want.numel() != 1:
            continue
        a, b = float(got), float(want)
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (key, a, b)
    if loss.requires_grad:
        loss.backward()
    if loss_ref.requires_grad:
        loss_ref.backward()
    ref_params = dict(ref.named_parameters())
    grads = [p.grad.abs().max().item() for p in ref_params.values() if p.grad is not None]
    scale = max(grads) if grads else 0.0
    for name, p in model.named_parameters():
        rg = ref_params[name].grad
        g = p.grad
        if rg is None or scale == 0.0:
            assert g is None or g.abs().max().item() <= 1e-6 * max(scale, 1.0), name
            continue
        if g is None:   # disc_factor = 0: the reference multiplies D's output by an exact 0, here D is not evaluated
            assert name.startswith("loss.discriminator") and rg.abs().max().item() == 0.0, name
            continue
        e = (g.detach().cpu().double() - rg.double()).abs().max().item() / max(rg.abs().max().item(), 1e-3 * scale)
        assert e < grad_tol, "param grad %s rel err %.3e" % (name, e)


def test_masked_class_and_box_mask(hip_lib):
    from odvae_amd import synthetic
    model, ref = build_pair()
    _compare(model, ref, _batch([0, 1, 3, 1]), synthetic.make_noise(4, 4, dropout_p=0.7, seed=21))


def test_batch_without_any_unmasked_sample(hip_lib):
    from odvae_amd import synthetic
    model, ref = build_pair()
    _compare(model, ref, _batch([1, 1]), synthetic.make_noise(2, 4, dropout_p=0.7, seed=22))


def test_untouched_thresholds_start_in_encoder_pretraining(hip_lib):
    from odvae_amd import synthetic
    model, ref = build_pair(phase="asis")
    assert model._get_dropout_prob() == 1.0
    _compare(model, ref, _batch([0, 2, 1], holes=False), synthetic.make_noise(3, 4, dropout_p=1.0, seed=23), global_step=0)


def test_validation_step_logs_match_oracle(hip_lib):
    """validation_step (autoencoder.py:332-363): eval mode, no graph, GAN + LPIPS-style terms on -- the adaptive weight
    cannot be differentiated there and falls back to 0 (contperceptual.py:295-299); z-dropout stays active because the
    reference builds a fresh nn.Dropout inside forward."""
    from odvae_amd import synthetic
    model, ref = build_pair(perceptual_weight=1.0, disc_factor=1.0)
    model.eval(); ref.eval()
    model._global_step = ref.global_step = 3
    batch = _batch([0, 1, 4], seed=41)
    noise = synthetic.make_noise(3, 4, dropout_p=0.7, seed=42)
    model.injected_noise = noise
    with torch.no_grad():
        model.validation_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0)
        want = ref.validation_step(batch, noise)
    got = model.logged_metrics
    checked = 0
    for key, w in want.items():
        if not torch.is_tensor(w) or w.numel() != 1:
            continue
        assert key in got, key
        a, b = float(got[key]), float(w)
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (key, a, b)
        checked += 1
    assert checked >= 10 and float(got["val/d_weight"]) == 0.0 and "val/disc_loss" in got


def test_batch_of_one(hip_lib):
    from odvae_amd import synthetic
    model, ref = build_pair()
    _compare(model, ref, _batch([5]), synthetic.make_noise(1, 4, dropout_p=0.7, seed=24))


def test_to_rgb_matches_reference_formula(hip_lib):
    """to_rgb (src/models/autoencoder.py:438-443): a fixed random 1x1 projection to three channels, min-max rescaled over the batch."""
    import torch.nn.functional as F
    model, _ = build_pair()
    x = torch.randn(2, 16, 12, 20, generator=torch.Generator().manual_seed(3)).to("cuda:0")
    y = model.to_rgb(x)
    w = model.colorize.detach().cpu()
    assert tuple(w.shape) == (3, 16, 1, 1) and "colorize" in dict(model.named_buffers())
    ref = F.conv2d(x.cpu(), w)
    ref = 2.0 * (ref - ref.min()) / (ref.max() - ref.min()) - 1.0
    assert tuple(y.shape) == (2, 3, 12, 20) and (y.cpu() - ref).abs().max().item() < 1e-5
    assert torch.equal(model.to_rgb(x), y)           # the projection is drawn once

"""Pose-head loss terms from ONE HIP kernel (pose_f32.hip) against the ORACLE's restatement of
src/modules/losses/contperceptual.py:111-132,176-212 (oracle/losses.PoseLoss.pose_terms, the same code its forward() runs) -- values
of all nine outputs and the gradients w.r.t. dec_pose and the box-posterior moments, on batches that exercise the masks (class id 1 =
BACKGROUND_CLASS_IDX, label "background", a batch with nothing unmasked, a clamped log-variance, B = 1 and B = 300, l2,
train_on_yaw = False).  A second test keeps the product's own per-term torch-op methods honest against the same kernel.
Tolerance 2e-5 relative (f32, different summation order)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
YAML = os.path.join(os.path.dirname(__file__), "golden", "autoencoder_kl_16x16x16.yaml")
DEV = "cuda:0"


def close(a, b, what, tol=2e-5, floor=1.0):
    """Elementwise: |a - b| <= tol * max(floor, |b|) -- the box KL reaches 1e8 beside terms of order 1, so one scale for the whole
    vector would check nothing but the largest entry."""
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    excess = ((a - b).abs() - tol * b.abs().clamp_min(floor)).max().item()
    assert excess <= 0, "%s: worst excess %.3e (max |ref| %.3e)" % (what, excess, b.abs().max().item())


def make_loss(pose_loss_fn="l1", train_on_yaw=True):
    from odvae_amd import synthetic
    from odvae_amd.config import instantiate_from_config
    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32)
    lc = mcfg.params.lossconfig
    lc.params["pose_loss_fn"] = pose_loss_fn
    lc.params["train_on_yaw"] = train_on_yaw
    if not train_on_yaw:   # the prior then reads the "v3" statistics (contperceptual.py:84-104); the stand-in only carries t3 / l / h / w
        for stats in lc.params["dataset_stats"].values():
            stats["v3"] = torch.tensor([0.1, -0.2])
    return instantiate_from_config(lc).to(DEV)


CASES = [("l1", True, "mixed"), ("l2", True, "mixed"), ("l1", False, "mixed"), ("l1", True, "all_masked"), ("l1", True, "one"),
         ("l1", True, "big")]


def make_oracle_loss(pose_loss_fn, train_on_yaw):
    from odvae_amd import synthetic
    from oracle.losses import PoseLoss as OraclePoseLoss
    mcfg, _ = synthetic.model_config(YAML, latent_hw=4, ch=32)
    lk = dict(mcfg.params.lossconfig.params.to_container())
    lk["pose_loss_fn"] = pose_loss_fn
    lk["train_on_yaw"] = train_on_yaw
    if not train_on_yaw:
        for stats in lk["dataset_stats"].values():
            stats["v3"] = torch.tensor([0.1, -0.2])
    return OraclePoseLoss(**lk)


def edge_batch(pose_loss_fn, yaw, case, nc):
    from odvae_amd import synthetic
    g = torch.Generator().manual_seed(CASES.index((pose_loss_fn, yaw, case)) + 11)
    B = {"one": 1, "big": 300}.get(case, 7)
    dec_pose = torch.randn(B, 8 + nc, generator=g) * 1.5
    moments = torch.randn(B, 16, generator=g)
    moments[0, 8] = 25.0       # log-variance above the clamp: value clamped, no gradient
    moments[-1, 9] = -40.0
    pose_gt = torch.randn(B, 4, generator=g) * 2
    bbox_gt, fill_gt = torch.randn(B, 3, generator=g), torch.rand(B, generator=g)
    ids = torch.randint(0, nc + 1, (B,), generator=g)      # nc itself = "no positive column" in the focal one-hot
    if case == "all_masked":
        ids[:] = 1
    elif B > 2:
        ids[1] = 1
    labels = [synthetic.LABELS[min(int(i), len(synthetic.LABELS) - 1)] for i in ids]
    if B > 3:
        labels[2] = "background"
    return dec_pose, moments, pose_gt, bbox_gt, fill_gt, ids, labels


W9 = [1.0, 0.7, 1.3, 0.4, 2e-3, 0, 0, 0, 0]


def run_kernel(loss, dec_pose, moments, pose_gt, bbox_gt, fill_gt, ids, labels):
    from odvae_amd.distributions import DiagonalGaussianDistribution
    dp = dec_pose.to(DEV).requires_grad_(True)
    mo = moments.to(DEV).requires_grad_(True)
    out = loss._fused_pose_terms(dp, pose_gt.to(DEV), bbox_gt.to(DEV), fill_gt.to(DEV), ids.to(DEV), labels,
                                 DiagonalGaussianDistribution(mo))
    (out * torch.tensor(W9, device=DEV)).sum().backward()
    return out.detach().cpu(), dp.grad.cpu(), mo.grad.cpu()


@pytest.mark.parametrize("pose_loss_fn,yaw,case", CASES)
def test_fused_pose_terms_match_oracle(hip_lib, pose_loss_fn, yaw, case):
    """pose_f32.hip vs oracle/losses.PoseLoss.pose_terms (CPU, the code path oracle forward() runs)."""
    from oracle.distributions import DiagonalGaussianDistribution as OracleDG
    loss = make_loss(pose_loss_fn, yaw)
    ref = make_oracle_loss(pose_loss_fn, yaw)
    dec_pose, moments, pose_gt, bbox_gt, fill_gt, ids, labels = edge_batch(pose_loss_fn, yaw, case, loss.num_classes)
    out, g_pose, g_mom = run_kernel(loss, dec_pose, moments, pose_gt, bbox_gt, fill_gt, ids, labels)

    dp = dec_pose.clone().requires_grad_(True)
    mo = moments.clone().requires_grad_(True)
    t = ref.pose_terms(dp, pose_gt, bbox_gt, fill_gt, ids, labels, OracleDG(mo))
    want = torch.stack([t["pose_loss"], t["class_loss"], t["bbox_loss"], t["fill_factor_loss"], t["kl_loss_bbox"],
                        t["t1"].mean(), t["t2"].mean(), t["t3"].mean(), t["v3"].mean()])
    close(out, want, "pose terms vs oracle")
    total = (want * torch.tensor(W9)).sum()
    if total.requires_grad:
        total.backward()
    close(g_pose, dp.grad if dp.grad is not None else torch.zeros_like(dp), "d dec_pose vs oracle", floor=1e-2)
    close(g_mom, mo.grad if mo.grad is not None else torch.zeros_like(mo), "d moments vs oracle", floor=1e-2)
    assert g_mom[0, 8].item() == 0.0 and g_mom[-1, 9].item() == 0.0     # clamped log-variances carry no gradient


@pytest.mark.parametrize("pose_loss_fn,yaw,case", CASES[:4])
def test_per_term_methods_match_fused_kernel(hip_lib, pose_loss_fn, yaw, case):
    """The product's own per-term methods (the `fused_pose_terms = False` path) against the same kernel."""
    from odvae_amd.distributions import DiagonalGaussianDistribution
    loss = make_loss(pose_loss_fn, yaw)
    dec_pose, moments, pose_gt, bbox_gt, fill_gt, ids, labels = edge_batch(pose_loss_fn, yaw, case, loss.num_classes)
    out, g_pose, g_mom = run_kernel(loss, dec_pose, moments, pose_gt, bbox_gt, fill_gt, ids, labels)
    dp = dec_pose.to(DEV).requires_grad_(True)
    mo = moments.to(DEV).requires_grad_(True)
    pose_gt, bbox_gt, fill_gt, class_gt = pose_gt.to(DEV), bbox_gt.to(DEV), fill_gt.to(DEV), ids.to(DEV)
    mask_bg = (class_gt != 1).long()
    post = DiagonalGaussianDistribution(mo)
    class_loss, _ = loss.compute_class_loss(class_gt, dp[:, 8:])
    bbox_loss, _ = loss.compute_bbox_loss(bbox_gt, dp[:, 4:7], mask_bg)
    pose_loss, _, t1, t2, t3, v3 = loss.compute_pose_loss(pose_gt, dp[:, :4], mask_bg)
    fill_loss, _ = loss.compute_fill_factor_loss(fill_gt, dp[:, 7:8].squeeze(), mask_bg)
    kl = loss.compute_pose_kl_loss(post, mask_bg, labels)
    ref = torch.stack([pose_loss, class_loss, bbox_loss, fill_loss, kl, t1.mean(), t2.mean(), t3.mean(), v3.mean()])
    close(out, ref, "pose terms")
    (ref * torch.tensor(W9, device=DEV)).sum().backward()
    close(g_pose, dp.grad, "d dec_pose", floor=1e-2)
    close(g_mom, mo.grad, "d moments", floor=1e-2)


def test_training_step_same_with_and_without_fused_pose_terms(hip_lib):
    """Whole step: every logged scalar and the pose-MLP gradients agree between the fused kernel and the per-term torch ops."""
    from test_model_gpu import build_pair
    from odvae_amd import synthetic
    res = {}
    for fused in (True, False):
        model, _ = build_pair()
        model.train()
        model._global_step = 1
        model.loss.fused_pose_terms = fused
        batch = synthetic.make_batch(4, 64, seed=5)
        batch["class_id"] = torch.tensor([0, 1, 3, 10])
        batch["class_name"] = ["car", "truck", "bus", "background"]
        model.injected_noise = synthetic.make_noise(4, 4, dropout_p=0.7, seed=6)
        loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, 0)
        loss.backward()
        res[fused] = (loss.detach(), dict(model.logged_metrics), {k: p.grad.detach().clone() for k, p in model.named_parameters()
                                                                  if p.grad is not None and k.startswith(("pose_", "quant_conv_pose"))})
    close(res[True][0], res[False][0], "total loss")
    for k, v in res[False][1].items():
        if torch.is_tensor(v) and v.numel() == 1:
            close(torch.as_tensor(float(res[True][1][k])), torch.as_tensor(float(v)), k)
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) > 4
    for k, v in res[False][2].items():
        close(res[True][2][k], v, "grad " + k, tol=1e-4, floor=max(1.0, v.abs().max().item()))

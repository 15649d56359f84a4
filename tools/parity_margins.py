"""How much of each stated tolerance does the HIP path use?  Runs the whole-step parity comparisons of tests/test_model_gpu.py (HIP training
step vs the CPU oracle: same weights, batch and injected noise) with the stride-1 3x3 convs on Winograd F(4x4,3x3) (default) and on
F(2x2,3x3), at the width-reduced network (ch = 32, 64 x 64) and at the benchmark's own network and resolution (ch = 128, 256 x 256, B = 2), and
writes the worst relative deviation per quantity -- total loss, every logged term, latent moments, reconstruction, pose output, every
parameter gradient (worst tensor named), 3-step loss curve -- to gpurun_out/parity_margins.json (copied to profiles/r04_parity_margins.json).
Relative deviations use the tests' own denominators: max|ref| for tensors, max(|ref grad|, 1e-3 * largest gradient) per parameter.
usage: python tools/parity_margins.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


if os.environ.get("ODVAE_PROBE_LIB"):      # A/B builds of the library (tools/ab_build.py)
    from odvae_amd import lib as _ab_lib
    _ab_lib.LIB_PATH = os.environ["ODVAE_PROBE_LIB"]


def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())


def one_step(ch, height, latent_hw, global_step=1, gan=False):
    from odvae_amd import synthetic
    from test_model_gpu import build_pair
    kw = dict(perceptual_weight=1.0, disc_factor=1.0) if gan else {}
    model, ref = build_pair(ch=ch, latent_hw=latent_hw, **kw)
    model.train(); ref.train()
    if gan:
        ref.loss.perceptual_loss.eval()
    model.loss.log_exact_g_loss = True
    model._global_step = ref.global_step = global_step
    batch = synthetic.make_batch(2, height, seed=5)
    noise = synthetic.make_noise(2, latent_hw, dropout_p=0.7, seed=6)
    model.injected_noise = noise
    loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, 0)
    loss_ref, log_ref, aux = ref.training_step(batch, 0, noise)
    logs = model.logged_metrics
    out = {"total_loss": rel(loss, loss_ref), "terms": {}}
    for key, v in log_ref.items():
        if key in logs and torch.is_tensor(v) and v.numel() == 1:
            out["terms"][key] = rel(torch.as_tensor(logs[key]).float(), v.float())
    dec_obj, dec_pose, post, _ = model.forward(model._rescale(batch["patch"].to("cuda:0")))
    out["latent_moments"] = rel(post.parameters, aux["posterior"].parameters)
    out["reconstruction"] = rel(dec_obj, aux["dec_obj"])
    out["dec_pose"] = rel(dec_pose, aux["dec_pose"])
    # pixels where the L1 term's gradient sign(x_hat - x) differs between the two paths (|x_hat - x| below the forward deviation): each
    # one moves the gradients by a fixed amount, 2 / numel of the reconstruction gradient at that pixel
    target = model._rescale(batch["patch"].to("cuda:0")).detach().cpu().double()
    sa, sb = torch.sign(dec_obj.detach().cpu().double() - target), torch.sign(aux["dec_obj"].detach().cpu().double() - target)
    out["l1_sign_flips"] = {"pixels": int((sa != sb).sum().item()), "of": sa.numel()}
    loss.backward(); loss_ref.backward()
    ref_params = dict(ref.named_parameters())
    scale = max(p.grad.abs().max().item() for p in ref_params.values() if p.grad is not None)
    worst, per_prefix = ("", 0.0), {}
    for name, p in model.named_parameters():
        rg = ref_params[name].grad
        if rg is None or p.grad is None:
            continue
        e = (p.grad.detach().cpu().double() - rg.double()).abs().max().item() / max(rg.abs().max().item(), 1e-3 * scale)
        pre = name.split(".")[0]
        per_prefix[pre] = max(per_prefix.get(pre, 0.0), e)
        if e > worst[1]:
            worst = (name, e)
    out["gradient_worst"] = {"tensor": worst[0], "rel": worst[1]}
    out["gradient_worst_by_module"] = per_prefix
    del model, ref
    torch.cuda.empty_cache()
    return out


def curve(ch, height, latent_hw, steps=3):
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    from oracle.autoencoder import train_batch
    from test_model_gpu import build_pair
    model, ref = build_pair(ch=ch, latent_hw=latent_hw)
    model.train(); ref.train()
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,))
    ref_opts = ref.configure_optimizers()
    worst = 0.0
    for step in range(steps):
        batch = synthetic.make_batch(2, height, seed=100 + step)
        noise = synthetic.make_noise(2, latent_hw, dropout_p=0.7, seed=200 + step)
        model.injected_noise = noise
        a = trainer.training_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, step)[0].item()
        b = train_batch(ref, ref_opts, batch, {0: noise}, optimizer_indices=(0,), clip=1.0)[0][0].item()
        worst = max(worst, abs(a - b) / max(1.0, abs(b)))
    lr, ref_sd, wworst = model.learning_rate, ref.state_dict(), ("", 0.0)
    for k, v in model.state_dict().items():
        if v.dtype == torch.float32 and k.startswith(("encoder", "decoder", "quant", "post_quant", "pose_")):
            d = (v.detach().cpu().double() - ref_sd[k].double()).abs().max().item()
            used = d / (2.2 * lr * steps + 5e-3 * ref_sd[k].abs().max().item())
            if used > wworst[1]:
                wworst = (k, used)
    del model, ref
    torch.cuda.empty_cache()
    return {"loss_curve_rel": worst, "weights_worst_fraction_of_bound": {"tensor": wworst[0], "fraction": wworst[1]}}


def main():
    from odvae_amd import ops
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_margins.json")
    res = {"tolerances_in_the_tests": {"outputs_and_losses": 1e-3, "gradients": 5e-3, "loss_curve": 2e-3,
                                       "weights": "2.2 lr per step + 5e-3 max|w|"},
           "note": "relative deviations HIP vs CPU oracle (parity unpinned: the oracle is this build's restatement); f32 path; B = 2"}
    if os.environ.get("ODVAE_MARGINS_ONLY") == "ch128":      # one comparison only (for A/B builds): the benchmark's network, F(4x4), step 1
        ops.WINOGRAD4 = True
        print(json.dumps(one_step(None, 256, 16, 1), indent=1), flush=True)
        return
    for f4 in (True, False):
        ops.WINOGRAD4 = f4
        tag = "F(4x4,3x3)" if f4 else "F(2x2,3x3)"
        res[tag] = {"ch32_64x64_step1": one_step(32, 64, 4, 1), "ch32_64x64_step0": one_step(32, 64, 4, 0),
                    "ch128_256x256_step1": one_step(None, 256, 16, 1),
                    "ch32_64x64_3step_curve": curve(32, 64, 4), "ch128_256x256_3step_curve": curve(None, 256, 16)}
        print(tag, json.dumps({k: (v.get("gradient_worst") or v) for k, v in res[tag].items()}), flush=True)
    ops.WINOGRAD4 = True
    res["F(4x4,3x3)"]["gan_lpips_ch32_64x64_step1"] = one_step(32, 64, 4, 1, gan=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print("written", out_path)


if __name__ == "__main__":
    main()

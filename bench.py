#!/usr/bin/env python
"""Headline benchmark: VAE training images/s (fwd + bwd + optimizer) at 256x256, z = 16x16x16 (BASELINE.json).

Workload = BASELINE.json configs[1]: configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml (fixture copy under
tests/golden/), 256x256 synthetic crops, B = 32 per GPU, fp32, rec+KL only (perceptual_weight = 0, disc_factor = 0,
optimizer 1 skipped), VAE phase.  One process per GPU; N > 1 is launched by torch.distributed.run
and shards the minibatch (weak scaling) with the bucketed RCCL all-reduce of generative-detection_amd/parallel.py.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around every launch of the dominant
kernel (the stride-1 3x3 convolution, forward + data-gradient: fused Winograd F(2x2,3x3), or the direct implicit-GEMM kernel
under ODVAE_CONV_WINOGRAD=0) during the timed steps;
`cpu_baseline` times the CPU oracle (a port: the reference itself cannot be imported) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
YAML = os.path.join(ROOT, "tests", "golden", "autoencoder_kl_16x16x16.yaml")
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--gan", action="store_true",
                    help="BASELINE.json configs[3]: PatchGAN discriminator + LPIPS-style loss, both optimizers per batch")
    ap.add_argument("--ckpt-decoder", action="store_true",
                    help="BASELINE.json configs[4]: activation-checkpointed Decoder (each up level / mid block is recomputed in backward)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: initialise RCCL with world size 1 and run the bucketed reducer anyway")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: cgroup quota, else affinity mask, capped at the one-GPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(res, batch=1, steps=1):
    """Oracle (plain torch on the host cores) on a bounded sample of the same workload: same yaml, same step
    definition (fwd + bwd + clip + Adam), rec+KL only, at `res` x `res`, batch `batch`."""
    from odvae_amd import synthetic
    from oracle.autoencoder import PoseAutoencoder as OraclePA, train_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    lat = res // 16
    mcfg, _ = synthetic.model_config(YAML, latent_hw=lat)
    p = mcfg.params.to_container()
    torch.manual_seed(23)
    ref = OraclePA(p["ddconfig"], dict(p["lossconfig"]["params"]), p["embed_dim"], p["pose_decoder_config"]["params"],
                   p["pose_encoder_config"]["params"], feat_dims=p.get("feat_dims", [16, 16, 16]),
                   dropout_prob_init=p["dropout_prob_init"],
                   dropout_prob_final=p["dropout_prob_final"], dropout_warmup_steps=p["dropout_warmup_steps"],
                   pose_conditioned_generation_steps=p["pose_conditioned_generation_steps"])
    ref.learning_rate = 12 * 4.5e-6
    ref.global_step = 1   # as on the GPU: past the very first step, whose total holds the pose terms only (contperceptual.py:307)
    opts = ref.configure_optimizers()
    batch_d = synthetic.make_batch(batch, res, seed=23)
    noise = synthetic.make_noise(batch, lat, seed=24)
    t0 = time.time()
    for _ in range(steps):
        train_batch(ref, opts, batch_d, {0: noise}, optimizer_indices=(0,), clip=1.0)
    dt = time.time() - t0
    return {"value": batch * steps / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d step(s), B=%d, %dx%d, fp32, rec+KL only, torch %s CPU oracle" % (steps, batch, res, res, torch.__version__)}


def main():
    args = parse()
    # RCCL prints banner lines ("Hostname", "Librccl path") on fd 1; the contract is ONE JSON line on stdout, so
    # everything else that lands on fd 1 is routed to stderr and the result is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
    from odvae_amd import ops, synthetic
    from odvae_amd.trainer import Trainer

    torch.manual_seed(23)
    lat = args.res // 16
    gan = dict(perceptual_weight=1.0, disc_factor=1.0, disc_start=0) if args.gan else {}
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=lat, **gan).to(dev)
    model.train()
    if args.ckpt_decoder:
        model.decoder.activation_checkpoint = True
    # steady state: past the very first optimizer step, whose total holds the pose terms only (`global_step >
    # encoder_pretrain_steps`, contperceptual.py:307) and would skip the decoder's backward pass -- every timed step does the
    # full forward + backward + optimizer work even with --warmup 0
    model._global_step = 1
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0, 1) if args.gan else (0,),
                      process_group=dist.group.WORLD if use_dist else None)
    batch = synthetic.make_batch(args.batch, args.res, seed=23 + rank)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}  # inputs resident in HBM

    def step(i):
        b = dict(batch)
        b["pose_6d"] = batch["pose_6d"].clone()  # training_step writes yaw into it
        return trainer.training_batch(b, i)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_kernel_events:
        ops.KERNEL_EVENTS.enable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    roof = ops.KERNEL_EVENTS.summary("conv3x3_128x128") if not args.no_kernel_events else None
    others, extra_ms = {}, None
    if roof is not None:   # ONE more step, outside the timed region, with the secondary kernel families bracketed as well
        ops.KERNEL_EVENTS.extra = True
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(args.warmup + args.steps)
        torch.cuda.synchronize()
        extra_ms = (time.perf_counter() - t1) * 1e3
        others = {k: ops.KERNEL_EVENTS.summary(k) for k in ("conv3x3_wgrad_wino", "gemm_f32")}
    ops.KERNEL_EVENTS.disable()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "VAE train images/s (fwd+bwd+opt) at %dx%d z=%dx%dx16" % (args.res, args.res, lat, lat),
            "value": args.batch * world * args.steps / elapsed,
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "autoencoder_kl_16x16x16.yaml, %dx%d synthetic crops, B=%d/GPU, %s, VAE phase"
                       % (args.res, args.res, args.batch,
                          ("PatchGAN + LPIPS-style loss, optimizers 0 and 1 per batch" if args.gan
                           else "rec+KL only (discriminator off, optimizer 0)")
                          + (", activation-checkpointed Decoder" if args.ckpt_decoder else "")),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world},
        }
        if roof is not None:
            traffic = None   # HBM bytes per launch of the same kernel, from committed rocprofv3 --pmc passes of this command
            tpath = os.path.join(ROOT, "profiles", "r01_conv3x3_traffic.json")
            if os.path.exists(tpath) and not args.gan and args.batch == 32 and args.res == 256 and not args.ckpt_decoder:
                traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
            wino = ops.WINOGRAD
            out["roofline"] = {"bound": "mfma", "achieved": roof["tflops"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": roof["tflops"] / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_conv3x3_traffic.json)",
                               "kernel": ("conv3x3_wino8_kernel (stride-1 3x3 conv fwd + dgrad, Winograd F(2x2,3x3), 32 tiles x 128 co per block)"
                                          if wino else "conv3x3_kernel_v2<MODE 0,KC 32,2,2,2,2> (3x3 conv fwd + dgrad, 128px x 128co tile)"),
                               "launches": roof["launches"], "avg_launch_ms": roof["avg_ms"],
                               "algorithmic_gflop_per_launch": roof["gflop_per_launch"],
                               "algorithmic_bytes_per_launch": roof["bytes_per_launch"],
                               "share_of_step_time": roof["total_ms"] / (ms * args.steps)}
            if wino:   # `achieved` counts the direct form's 2*9*Cin*Cout FLOP per pixel; the kernel executes 16/36 of them
                out["roofline"]["executed_tflops"] = roof["tflops"] * 16.0 / 36.0
                out["roofline"]["frac_executed"] = roof["tflops"] * 16.0 / 36.0 / PEAK_F32_MFMA_TFLOPS
                out["roofline"]["note"] = ("algorithmic (direct-convolution) FLOP/s can exceed the f32 MFMA peak: Winograd F(2x2,3x3) "
                                           "needs 16 instead of 36 multiply-adds per 2x2 output tile; frac_executed prices the "
                                           "multiply-adds actually issued against the same peak")
            # the next two MFMA-bound kernel families of the step, measured the same way (HIP events around every launch)
            names = {"conv3x3_wgrad_wino": "conv3x3_wgrad_wino_kernel (weight gradient of the stride-1 3x3 convs in the Winograd domain)",
                     "gemm_f32": "gemm_f32_kernel (attention products incl. the fused softmax backward, 1x1 convs and their gradients)"}
            out["roofline_others"] = []
            for key, r in others.items():
                if r is None:
                    continue
                e = {"kernel": names[key], "bound": "mfma", "achieved": r["tflops"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": r["tflops"] / PEAK_F32_MFMA_TFLOPS, "launches": r["launches"], "avg_launch_ms": r["avg_ms"],
                     "share_of_step_time": r["total_ms"] / extra_ms,
                     "measured_on": "one extra step after the timed region (HIP events around every launch of this family)"}
                if key == "conv3x3_wgrad_wino":
                    e["executed_tflops"] = r["tflops"] * 16.0 / 36.0
                    e["frac_executed"] = e["executed_tflops"] / PEAK_F32_MFMA_TFLOPS
                out["roofline_others"].append(e)
        print("[bench] GPU leg done: %.2f images/s, %.1f ms/step" % (out["value"], ms), file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.res, batch=2, steps=2)   # about 10 s of host work on 16 cores
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

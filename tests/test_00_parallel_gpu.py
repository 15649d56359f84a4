"""Data-parallel path on device tensors (reference: `strategy: ddp`,
configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml:137, train.py:162), real PoseAutoencoder + FusedAdam + GradReducer + Trainer.

1. two ranks share the one GPU of the test box over **gloo** (it moves CUDA tensors through the host; RCCL refuses two ranks on one
   device):  (a) the bucketed all-reduce over the flat gradient arena keeps both ranks in lock-step and every bucket is reduced;
   (b) the averaged gradient arena equals the gradient of ONE process on the concatenated batch (what DDP promises), in the VAE phase
   and in the encoder-pretraining phase of the untouched yaml, where the decoder gets no gradient and buckets are only partly touched;
2. one rank over **nccl** (= RCCL; a one-GPU box can host exactly one RCCL rank): the same Trainer + GradReducer step through real RCCL
   collectives is bit-identical to the non-distributed step.
The RCCL launch with N > 1 (`bench.py --gpus N`) can only be exercised on a multi-GPU node.

Every case spawns its ranks BEFORE this pytest process touches the GPU (fork + exec from a GPU-initialised process is forbidden on this
pool), which is why the file name sorts first; device_count() does not initialise the GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, "tests", "golden", "autoencoder_kl_16x16x16.yaml")
OWNED = ("encoder", "decoder", "quant", "post_quant", "pose_")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _shard(rank, step=0):
    """One rank's minibatch.  Every sample carries a pixel at 0 and one at 1, so `_rescale`'s batch min / max
    (src/models/autoencoder.py:434-436; per rank, not synchronised) is the same on a shard and on the concatenated batch."""
    from odvae_amd import synthetic
    batch = synthetic.make_batch(2, 64, seed=50 + 10 * step + rank)
    batch["patch"][:, :, 0, 0] = 0.0
    batch["patch"][:, :, 0, 1] = 1.0
    return batch, synthetic.make_noise(2, 4, seed=70 + 10 * step + rank)


def _concat(parts):
    out = {}
    for k in parts[0]:
        v = [p[k] for p in parts]
        out[k] = torch.cat(v, 0) if torch.is_tensor(v[0]) else sum(v, [])
    return out


def _grad_arena(model, opt, red, batch, noise):
    """training_step -> backward (-> bucketed all-reduce) exactly as Trainer.training_batch does, stopping before clip / step."""
    model.injected_noise = noise
    loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, 0)
    opt.zero_grad(set_to_none=True)
    partial = []
    if red is not None:
        red.prepare_for_backward()
        (loss * red.inv_world).backward()
        partial = sorted(red._touched_buckets - red._launched)     # buckets only some of whose parameters got a gradient
        red.finish()
    else:
        loss.backward()
    opt.gather_grads()
    torch.cuda.synchronize()
    return opt.flat_grad.detach().cpu().clone(), partial


def _gloo_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    out = {}
    # ---- (b) averaged gradients = full-batch gradients, both phases ------------------------------------------------------------
    for phase in ("vae", "asis"):
        torch.manual_seed(1000 + rank)   # different initial weights per rank: the broadcast must align them
        model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32, phase=phase).to("cuda:0").train()
        trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,), bucket_mb=1.0)
        model._global_step = 1 if phase == "vae" else 0      # "asis": 0 < encoder_pretrain_steps = 30000 -> decoder skipped
        batch, noise = _shard(rank)
        arena, partial = _grad_arena(model, trainer.optimizers[0], trainer.reducers[0], batch, noise)
        res = {"arena": arena, "partial": partial, "nbuckets": len(trainer.reducers[0].buckets),
               "launched": sorted(trainer.reducers[0].launch_order)}
        if rank == 0:
            single = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32, phase=phase)
            single.load_state_dict(model.state_dict())
            single = single.to("cuda:0").train()
            t1 = Trainer(single, gradient_clip_val=1.0, optimizer_indices=(0,), distributed=False)
            assert t1.reducers is None
            single._global_step = model._global_step
        shards = [_shard(r) for r in range(world)]
        if rank == 0:
            full_noise = {k: torch.cat([s[1][k] for s in shards], 0) for k in shards[0][1]}
            res["arena_full"], _ = _grad_arena(single, t1.optimizers[0], None, _concat([s[0] for s in shards]), full_noise)
            name_of = {id(p): n for n, p in single.named_parameters()}
            res["names"] = [(name_of[id(p)], off, cnt) for p, off, cnt in t1.optimizers[0].param_slices()]
        out[phase] = res
        del model, trainer
    # ---- (a) lock-step over two optimizer steps --------------------------------------------------------------------------------
    torch.manual_seed(1000 + rank)
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32).to("cuda:0").train()
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,), bucket_mb=4.0)
    red = trainer.reducers[0]
    losses = []
    for step in range(2):
        batch, noise = _shard(rank, step)
        model.injected_noise = noise
        losses.append(trainer.training_batch(batch, step)[0].item())
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if k.startswith(OWNED)}
    out.update({"sd": sd, "losses": losses, "nbuckets": len(red.buckets), "order": list(red.launch_order)})
    torch.save(out, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def _nccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    runs = {}
    for mode in ("rccl", "single", "rccl-bf16-f32acc"):
        torch.manual_seed(23)
        model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32).to("cuda:0").train()
        if mode == "rccl":
            trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,), process_group=dist.group.WORLD, bucket_mb=1.0)
            assert trainer.reducers is not None and dist.get_backend() == "nccl"
        elif mode == "rccl-bf16-f32acc":
            # bf16 buckets with f32 accumulation: all_to_all_single + all_gather_into_tensor of RCCL itself (world size 1: the sum of one
            # shard is that shard, so every gradient is exactly its own bf16 rounding)
            trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,), process_group=dist.group.WORLD, bucket_mb=1.0,
                              comm_dtype=torch.bfloat16, comm_f32_accumulate=True)
            assert trainer.reducers[0].f32_accumulate
        else:
            trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,), distributed=False)
            assert trainer.reducers is None
        model._global_step = 1
        losses = []
        for step in range(2):
            batch, noise = _shard(0, step)
            model.injected_noise = noise
            losses.append(trainer.training_batch(batch, step)[0].item())
        torch.cuda.synchronize()
        runs[mode] = {"sd": {k: v.detach().cpu() for k, v in model.state_dict().items() if k.startswith(OWNED)}, "losses": losses}
        if mode.startswith("rccl"):
            red = trainer.reducers[0]
            runs[mode].update(nbuckets=len(red.buckets), order=list(red.launch_order))
        del model, trainer
    torch.save(runs, os.path.join(out_dir, "nccl.pt"))
    dist.destroy_process_group()


def _spawnable():
    if torch.cuda.device_count() < 1:
        pytest.skip("no HIP device")
    if torch.cuda.is_initialized():
        pytest.skip("GPU already initialised in this process; run this file in its own pytest invocation")


def test_two_ranks_one_gpu_lockstep_and_full_batch_gradient(tmp_path):
    _spawnable()
    world, port = 2, _free_port()
    mp.spawn(_gloo_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    # (a) lock-step
    assert r0["nbuckets"] >= 2 and sorted(r0["order"]) == list(range(r0["nbuckets"])) and r0["order"] == r1["order"]
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k   # identical averaged gradients -> bit-identical weights
    assert all(torch.isfinite(torch.tensor(r["losses"])).all() for r in (r0, r1))
    assert r0["losses"] != r1["losses"]                  # the ranks really saw different shards
    # (b) mean over ranks of the shard gradients = gradient of the concatenated batch (fp32: different summation split only)
    for phase in ("vae", "asis"):
        a0, a1, full = r0[phase]["arena"].double(), r1[phase]["arena"].double(), r0[phase]["arena_full"].double()
        assert torch.equal(a0, a1), phase                # both ranks hold the same reduced arena
        gmax = full.abs().max().item()
        assert gmax > 0
        worst = 0.0
        for name, off, n in r0[phase]["names"]:
            ref = full[off:off + n]
            err = (a0[off:off + n] - ref).abs().max().item()
            scale = max(ref.abs().max().item(), 1e-3 * gmax)
            worst = max(worst, err / scale)
            assert err <= 2e-3 * scale, "%s %s: %.3e vs scale %.3e" % (phase, name, err, scale)
        print("phase %s: worst per-parameter gradient deviation %.2e of its scale" % (phase, worst))
    # the pretraining phase leaves the decoder without gradients: some buckets are never launched by the count-down and
    # at least one is flushed by finish() with only part of its parameters touched
    asis = r0["asis"]
    dec = [(o, n) for name, o, n in asis["names"] if name.startswith("decoder.")]
    assert dec and all(float(asis["arena_full"][o:o + n].abs().max()) == 0.0 for o, n in dec)
    assert len(asis["partial"]) >= 1 or len(asis["launched"]) < asis["nbuckets"]
    assert len(r0["vae"]["launched"]) == r0["vae"]["nbuckets"]


def test_rccl_world1_step_is_bit_identical_to_single_process(tmp_path):
    _spawnable()
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    runs = torch.load(os.path.join(tmp_path, "nccl.pt"))
    a, b = runs["rccl"], runs["single"]
    assert a["nbuckets"] >= 2 and sorted(a["order"]) == list(range(a["nbuckets"]))
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    for k in b["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k
    # bf16 buckets, f32-accumulated, through RCCL's all-to-all / all-gather: every bucket travelled; the first step's loss is the same
    # (the exchange comes after it), the second differs by what two Adam steps on bf16-rounded gradients can move the weights
    c = runs["rccl-bf16-f32acc"]
    assert c["nbuckets"] >= 2 and sorted(c["order"]) == list(range(c["nbuckets"]))
    assert c["losses"][0] == b["losses"][0] and abs(c["losses"][1] - b["losses"][1]) <= 2e-3 * abs(b["losses"][1])
    lr = 4.5e-6 * 12
    for k in b["sd"]:
        assert (c["sd"][k] - b["sd"][k]).abs().max().item() <= 2 * 2.2 * lr + 1e-6, k

"""N > 1 path on CPU: two gloo ranks, the bucketed GradReducer + Trainer loop of the product, driven with a small torch
model (the HIP ops need a GPU; the reducer and the trainer are device-agnostic).  Checks: averaged gradients equal the
single-process gradients of the concatenated batch, buckets are launched while backward is still running (reverse
order), parameters are broadcast from rank 0, and two ranks stay bit-identical over several optimizer steps."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class TinyNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 64)
        self.b = nn.Linear(64, 64)
        self.c = nn.Linear(64, 64)
        self.d = nn.Linear(64, 4)

    def forward(self, x):
        return self.d(torch.tanh(self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))))


class TinyLightning(nn.Module):
    """Two-optimizer module with the surface Trainer uses (training_step / configure_optimizers / _global_step)."""

    def __init__(self):
        super().__init__()
        self.gen, self.disc = TinyNet(), TinyNet()
        self._global_step = 0
        self.learning_rate = 1e-2

    def training_step(self, batch, batch_idx, optimizer_idx):
        x, y = batch
        if optimizer_idx == 0:
            return ((self.gen(x) - y) ** 2).mean() + 0.1 * self.disc(self.gen(x).detach().repeat(1, 2)).mean() * 0
        return (self.disc(x) ** 2).mean()

    def configure_optimizers(self):
        return [torch.optim.Adam(self.gen.parameters(), lr=self.learning_rate, betas=(0.5, 0.9)),
                torch.optim.Adam(self.disc.parameters(), lr=self.learning_rate, betas=(0.5, 0.9))], []


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from odvae_amd.parallel import GradReducer
    from odvae_amd.trainer import Trainer
    torch.manual_seed(100 + rank)            # different init per rank: broadcast must fix it
    model = TinyLightning()
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0, 1), bucket_mb=0.02)
    assert trainer.reducers is not None and len(trainer.reducers[0].buckets) >= 2
    g = torch.Generator().manual_seed(5)
    full_x, full_y = torch.randn(8, 8, generator=g), torch.randn(8, 4, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)
    # --- gradient equivalence on the first backward (before any optimizer step) ---
    red = trainer.reducers[0]
    loss = model.training_step((full_x[shard], full_y[shard]), 0, 0)
    red.prepare_for_backward()
    assert red.prescaled                     # the Trainer's reducers take the mean through the loss scale
    (loss * red.inv_world).backward()
    order = list(red.launch_order)
    red.finish()
    grads = {n: p.grad.clone() for n, p in model.gen.named_parameters()}
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    # --- several batches through the trainer ---
    for i in range(3):
        trainer.training_batch((full_x[shard] + i, full_y[shard]), i)
    # sync_dist=True of self.log (validation's val/rec_loss, autoencoder.py:359): mean over ranks
    from odvae_amd.lightning import LightningModule
    lm = LightningModule()
    lm.log("val/rec_loss", torch.tensor(float(rank + 1)), sync_dist=True)
    lm.log("local", torch.tensor(float(rank + 1)))
    synced, local = float(lm.logged_metrics["val/rec_loss"]), float(lm.logged_metrics["local"])
    # a generic (non-prescaled) reducer still averages in finish()
    net2 = TinyNet()
    opt2 = torch.optim.SGD(net2.parameters(), lr=0.1)
    red2 = GradReducer(opt2, bucket_mb=0.02)
    red2.broadcast_parameters(net2)
    red2.prepare_for_backward()
    net2(full_x[shard]).pow(2).mean().backward()
    red2.finish()
    g2 = torch.cat([p.grad.reshape(-1) for p in net2.parameters()]).clone()
    # bf16 gradient buckets (the exchange of the bf16 step): cast -> all-reduce -> back into the f32 gradients
    net3 = TinyNet()
    opt3 = torch.optim.SGD(net3.parameters(), lr=0.1)
    red3 = GradReducer(opt3, bucket_mb=0.02, comm_dtype=torch.bfloat16)
    red3.broadcast_parameters(net3)
    red3.prepare_for_backward()
    net3(full_x[shard]).pow(2).mean().backward()
    red3.finish()
    g3 = torch.cat([p.grad.reshape(-1) for p in net3.parameters()]).clone()
    # bf16 on the wire with f32 accumulation: all-to-all of shards -> f32 sum on the owning rank -> one rounding -> all-gather
    net5 = TinyNet()
    opt5 = torch.optim.SGD(net5.parameters(), lr=0.1)
    red5 = GradReducer(opt5, bucket_mb=0.02, comm_dtype=torch.bfloat16, f32_accumulate=True)
    red5.broadcast_parameters(net5)
    red5.prepare_for_backward()
    net5(full_x[shard]).pow(2).mean().backward()
    local5 = torch.cat([p.grad.reshape(-1) for p in net5.parameters()]).clone()      # this rank's own gradient (the arena, before the exchange)
    red5.finish()
    g5 = torch.cat([p.grad.reshape(-1) for p in net5.parameters()]).clone()
    # the same through the Trainer: comm_dtype=torch.bfloat16 is an opt-in (the default keeps f32 buckets, whatever the precision)
    assert all(r.comm_dtype is None for r in trainer.reducers)
    torch.manual_seed(300 + rank)
    model4 = TinyLightning()
    trainer4 = Trainer(model4, gradient_clip_val=1.0, optimizer_indices=(0, 1), bucket_mb=0.02, comm_dtype=torch.bfloat16)
    red4 = trainer4.reducers[0]
    assert red4.comm_dtype == torch.bfloat16
    loss4 = model4.training_step((full_x[shard], full_y[shard]), 0, 0)
    red4.prepare_for_backward()
    (loss4 * red4.inv_world).backward()
    red4.finish()
    grads4 = {n: p.grad.clone() for n, p in model4.gen.named_parameters()}
    sd4 = {k: v.clone() for k, v in model4.state_dict().items()}
    torch.save({"g5": g5, "local5": local5, "nb5": len(red5.buckets),
                "grads4": grads4, "sd4": sd4, "g3": g3, "sd3": net3.state_dict(), "nb3": len(red3.buckets), "order3": list(red3.launch_order), "synced": synced, "local": local, "g2": g2, "sd2": net2.state_dict(), "grads": grads, "sd0": sd0, "sd": model.state_dict(), "order": order, "nbuckets": len(red.buckets),
                "global_step": model._global_step}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    # broadcast: both ranks started from rank 0's weights
    for k in r0["sd0"]:
        assert torch.equal(r0["sd0"][k], r1["sd0"][k]), k
    # single-process reference: same weights, full batch -> mean loss over 8 = average of the two shard gradients
    ref = TinyLightning()
    ref.load_state_dict(r0["sd0"])
    g = torch.Generator().manual_seed(5)
    full_x, full_y = torch.randn(8, 8, generator=g), torch.randn(8, 4, generator=g)
    ref.training_step((full_x, full_y), 0, 0).backward()
    for n, p in ref.gen.named_parameters():
        assert torch.allclose(r0["grads"][n], p.grad, atol=1e-6), n
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n
    # Trainer with bf16 buckets (opt-in): averaged gradient = full-batch gradient up to the bf16 rounding of each rank's pre-scaled
    # share (2^-9 relative per addend; stated bound 2^-7 of each tensor's largest entry), identical on both ranks
    ref4 = TinyLightning()
    ref4.load_state_dict(r0["sd4"])
    ref4.training_step((full_x, full_y), 0, 0).backward()
    for n, p in ref4.gen.named_parameters():
        assert torch.equal(r0["grads4"][n], r1["grads4"][n]), n
        assert (r0["grads4"][n] - p.grad).abs().max() <= 2.0 ** -7 * p.grad.abs().max() + 1e-12, n
        assert not torch.equal(r0["grads4"][n], p.grad) or p.grad.abs().max() == 0, n     # (it did travel as bf16)
    # buckets fire during backward, last layers first (reverse arena order), and every bucket fired
    assert r0["order"] == r1["order"] and len(r0["order"]) == r0["nbuckets"]
    assert r0["order"][0] == r0["nbuckets"] - 1 and r0["order"][-1] == 0
    # ranks stay in lock-step; global_step advanced by 2 per batch (one per optimizer step, PL-1.9)
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    assert r0["global_step"] == 6 and r1["global_step"] == 6
    assert r0["synced"] == r1["synced"] == 1.5 and (r0["local"], r1["local"]) == (1.0, 2.0)
    ref2 = TinyNet()
    ref2.load_state_dict(r0["sd2"])
    ref2(full_x).pow(2).mean().backward()
    want = torch.cat([p.grad.reshape(-1) for p in ref2.parameters()])
    assert torch.allclose(r0["g2"], want, atol=1e-6) and torch.equal(r0["g2"], r1["g2"])
    # bf16 buckets: every bucket travelled, both ranks hold the same gradients, and they are the full-batch gradients up to the
    # bf16 rounding of the two shard gradients (2^-8 each) and of their sum
    ref3 = TinyNet()
    ref3.load_state_dict(r0["sd3"])
    ref3(full_x).pow(2).mean().backward()
    want3 = torch.cat([p.grad.reshape(-1) for p in ref3.parameters()])
    assert r0["nb3"] >= 2 and sorted(r0["order3"]) == list(range(r0["nb3"]))
    assert torch.equal(r0["g3"], r1["g3"])
    assert (r0["g3"] - want3).abs().max().item() <= 2.0 ** -6 * want3.abs().max().item()
    assert not torch.equal(r0["g3"], want3)          # it really went through bf16
    # bf16 on the wire, f32 sums: EXACTLY bf16(mean-less sum in f32 of the two bf16-rounded shard gradients) averaged by the generic
    # reducer's finish() -- one rounding of the sum, not one per hop
    assert r0["nb5"] >= 2 and torch.equal(r0["g5"], r1["g5"])
    exact5 = (r0["local5"].to(torch.bfloat16).float() + r1["local5"].to(torch.bfloat16).float()).to(torch.bfloat16).float() * 0.5
    assert torch.equal(r0["g5"], exact5)


def test_front_bucket_is_small():
    """The bucket at the front of the arena holds the model's first parameters -- the gradients a backward pass produces LAST -- so its
    collective is the one nothing can hide; like torch DDP's 1 MB first bucket it is cut small (GradReducer(first_bucket_mb=...)).  Layout only:
    single process, gloo group of one."""
    import torch.distributed as dist
    from odvae_amd.parallel import GradReducer
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        dist.init_process_group("gloo", rank=0, world_size=1)
        own = True
    else:
        own = False
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(*[torch.nn.Linear(256, 256) for _ in range(8)])      # 8 x 65 792 parameters = 0.25 MB each
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        red = GradReducer(opt, bucket_mb=1.0, first_bucket_mb=0.25)
        sizes = [(e - s) * 4 / 2 ** 20 for s, e in red.buckets]
        assert sizes[0] <= 0.27 and all(x >= 1.0 for x in sizes[1:-1]), sizes
        assert sum(e - s for s, e in red.buckets) == red.arena.numel() and red.buckets[0][0] == 0
        # the front bucket is the LAST one whose collective is issued
        red.prepare_for_backward()
        net(torch.randn(4, 256)).sum().backward()
        red.finish()
        assert red.launch_order[-1] == 0 and sorted(red.launch_order) == list(range(len(red.buckets))), red.launch_order
        same = GradReducer(torch.optim.SGD(net.parameters(), lr=0.1), bucket_mb=1.0, first_bucket_mb=1.0)
        assert (same.buckets[0][1] - same.buckets[0][0]) * 4 / 2 ** 20 >= 1.0
    finally:
        if own:
            dist.destroy_process_group()

"""Soak run: N training steps of the benchmark configuration (B = 32, 256 x 256, rec+KL only; --bf16 for mixed precision) on fresh synthetic
batches, checking what a short parity test cannot: the loss stays finite and falls, no attention block ever needed the folded softmax's
fallback (device flags), no GroupNorm team barrier timed out, steady-state steps allocate nothing.
usage: python tools/soak.py [steps] [--bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
YAML = os.path.join(ROOT, "tests", "golden", "autoencoder_kl_16x16x16.yaml")


def main():
    from odvae_amd import lib as _lib, ops, synthetic
    from odvae_amd.trainer import Trainer
    steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
    bf16 = "--bf16" in sys.argv
    dev = torch.device("cuda:0")
    torch.manual_seed(23)
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=16).to(dev).train()
    model._global_step = 1
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,), precision="bf16" if bf16 else None)
    losses = []
    fallbacks0 = _lib.load().odvae_attn_softmax_fallbacks(0)
    alloc0 = None
    for i in range(steps):
        batch = synthetic.make_batch(32, 256, seed=1000 + i)
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
        loss = trainer.training_batch(batch, i)[0]
        losses.append(loss.detach())
        if i == 5:
            alloc0 = torch.cuda.memory_stats(dev)["num_device_alloc"]
    torch.cuda.synchronize()
    ls = [float(x) for x in losses]
    assert all(l == l and abs(l) < 1e30 for l in ls), "non-finite loss"
    n_flag = _lib.load().odvae_attn_softmax_fallbacks(0) - fallbacks0     # the device-side counter: every attention block of every step
    timeouts = _lib.load().odvae_groupnorm_fused_timeouts()
    allocs = torch.cuda.memory_stats(dev)["num_device_alloc"] - alloc0
    k = max(1, steps // 6)
    print("%s: %d steps; loss first %d mean %.1f -> last %d mean %.1f; folded-softmax fallbacks %d; GroupNorm barrier timeouts %d; device allocations after step 5: %d"
          % ("bf16" if bf16 else "f32", steps, k, sum(ls[:k]) / k, k, sum(ls[-k:]) / k, n_flag, timeouts, allocs))
    assert sum(ls[-k:]) < sum(ls[:k]) and n_flag == 0 and timeouts == 0


if __name__ == "__main__":
    main()

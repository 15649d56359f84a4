"""Which allocations reach hipMalloc in steady state?  Runs the bf16 (or f32) training step, then records the caching allocator's
history over a few more steps and prints every `segment_alloc` (= a device allocation: a synchronising call) with the python frames of
the request that caused it, plus per-step device-allocation counts.  usage: python tools/alloc_probe.py [--f32] [--gc-off] [--steps N]"""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
YAML = os.path.join(ROOT, "tests", "golden", "autoencoder_kl_16x16x16.yaml")


def main():
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    f32 = "--f32" in sys.argv
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 12
    warm = int(sys.argv[sys.argv.index("--warmup") + 1]) if "--warmup" in sys.argv else 6
    dev = torch.device("cuda:0")
    torch.manual_seed(23)
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=16).to(dev).train()
    model._global_step = 1
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,), precision=None if f32 else "bf16")
    data = synthetic.make_batch(32, 256, seed=23)
    data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}

    def step(i):
        b = dict(data)
        b["pose_6d"] = data["pose_6d"].clone()
        return trainer.training_batch(b, i)

    def n_alloc():
        return torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    for i in range(warm):
        a0 = n_alloc()
        step(i)
        torch.cuda.synchronize()
        print("warm-up step %d: %d device allocations, reserved %.2f GB, gc counts %s" % (i, n_alloc() - a0, torch.cuda.memory_reserved(dev) / 1e9, gc.get_count()), flush=True)
    if "--gc-off" in sys.argv:
        gc.collect()
        gc.freeze()
        gc.disable()
    if "--history" in sys.argv:
        torch.cuda.memory._record_memory_history(max_entries=200000)
    gc_log, gc_t0 = [], [0.0]

    def on_gc(phase, info):
        if phase == "start":
            gc_t0[0] = time.perf_counter()
        else:
            gc_log.append((info["generation"], (time.perf_counter() - gc_t0[0]) * 1e3))
    gc.callbacks.append(on_gc)
    for i in range(steps):
        del gc_log[:]
        a0 = n_alloc()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        step(warm + i)
        ev1.record()
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats(dev)
        print("    gc in this step: %d collections, %.1f ms in all, by generation %s" % (len(gc_log), sum(t for _, t in gc_log),
              {g: "%d x, %.1f ms" % (sum(1 for gg, _ in gc_log if gg == g), sum(t for gg, t in gc_log if gg == g)) for g in (0, 1, 2)}))
        print("step %d: %d device allocations, %.2f ms, reserved %.3f GB | small pool: allocated %.2f MB, reserved %.2f MB, inactive split %.2f MB, live blocks %d | large pool: allocated %.2f MB, reserved %.2f MB"
              % (i, n_alloc() - a0, ev0.elapsed_time(ev1), torch.cuda.memory_reserved(dev) / 1e9, st["allocated_bytes.small_pool.current"] / 1e6,
                 st["reserved_bytes.small_pool.current"] / 1e6, st["inactive_split_bytes.small_pool.current"] / 1e6, st["allocation.small_pool.current"],
                 st["allocated_bytes.large_pool.current"] / 1e6, st["reserved_bytes.large_pool.current"] / 1e6), flush=True)
    if "--history" not in sys.argv:
        return
    snap = torch.cuda.memory._snapshot()
    torch.cuda.memory._record_memory_history(enabled=None)
    traces = snap.get("device_traces", [[]])[0]
    pending = None
    shown = 0
    for ev in traces:
        if ev["action"] == "segment_alloc":
            shown += 1
            frames = [f for f in ev.get("frames", []) if "/repo/" in f.get("filename", "")][:6]
            print("segment_alloc %.1f MB" % (ev["size"] / 1e6), " <- ".join("%s:%d %s" % (os.path.basename(f["filename"]), f["line"], f["name"]) for f in frames))
    print("segment allocations recorded:", shown)


if __name__ == "__main__":
    main()

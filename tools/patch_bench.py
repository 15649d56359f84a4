"""Time csrc/patch_u8.hip on one batch of nuScenes-shaped work (32 instances cut from six 1600x900 camera images) and the
PIL path of the reference on the same instances.  Algorithmic HBM bytes = crop bytes read once + patch and mask written
once.  Usage (GPU box): python tools/patch_bench.py [S]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odvae_amd.patches import GpuPatcher  # noqa: E402
from oracle import patches as oracle      # noqa: E402  (timed as the CPU baseline only)


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(0)
    host = [rng.integers(0, 256, (900, 1600, 3), dtype=np.uint8) for _ in range(6)]
    dev = [torch.from_numpy(a).to("cuda:0") for a in host]
    inst = []
    while len(inst) < 32:
        k = len(inst) % 6
        w, h = rng.uniform(30, 450), rng.uniform(30, 450)
        cx, cy = rng.uniform(0, 1600), rng.uniform(0, 900)
        inst.append((k, [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], [cx, cy]))
    patcher = GpuPatcher(patch_height=S, perturb_scale=True)
    out = patcher(dev, inst)
    torch.cuda.synchronize()
    n = len(out.kept)
    crop_bytes = sum(p.size * p.size * 3 for p in out.plans)
    algo = crop_bytes + n * S * S * 4 * 4
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        out = patcher(dev, inst)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    # kernel only: replay the launch with HIP events around it
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    staged = patcher.stage(dev, inst)
    for _ in range(reps):
        e0.record()
        patcher.launch(staged)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    t0 = time.perf_counter()
    for k, bbox, c in inst:
        oracle.generate_patch_pil(host[k], bbox, c, (S, S), True)
    cpu = time.perf_counter() - t0
    print("S=%d  instances=%d  algorithmic bytes=%.1f MB" % (S, n, algo / 1e6))
    print("GPU: %.1f us per batch wall (host plan + H2D + launch; %.0f patches/s), kernel %.1f us (min of HIP-event brackets) -> %.0f GB/s algorithmic"
          % (wall * 1e6, n / wall, min(times) * 1e3, algo / (min(times) * 1e-3) / 1e9))
    print("CPU (PIL path, 1 core): %.1f ms per batch -> %.0f patches/s" % (cpu * 1e3, n / cpu))


if __name__ == "__main__":
    main()

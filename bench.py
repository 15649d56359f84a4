#!/usr/bin/env python
"""Headline benchmark: VAE training images/s (fwd + bwd + optimizer) at 256x256, z = 16x16x16 (BASELINE.json).

Workload = BASELINE.json configs[1]: configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml (fixture copy under
tests/golden/), 256x256 synthetic crops, B = 32 per GPU, fp32, rec+KL only (perceptual_weight = 0, disc_factor = 0,
optimizer 1 skipped), VAE phase.  One process per GPU; N > 1 is launched by torch.distributed.run
and shards the minibatch (weak scaling) with the bucketed RCCL all-reduce of generative-detection_amd/parallel.py.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around every launch of the dominant
kernel (the stride-1 3x3 convolution, forward + data-gradient: fused Winograd F(4x4,3x3); F(2x2,3x3) under ODVAE_CONV_WINOGRAD4=0,
the direct implicit-GEMM kernel under ODVAE_CONV_WINOGRAD=0) during the timed steps;
`cpu_baseline` times the CPU oracle (a port: the reference itself cannot be imported) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# Host threads: the step's host side is one Python thread issuing launches plus a few tiny CPU tensor ops (the reference draws its noise on the
# CPU).  Left alone, torch sizes its intra-op pool by the machine (hundreds of hardware threads on a GPU host) and OpenMP workers SPIN after
# each parallel region; inside a container with a CPU quota that burns the quota and the kernel throttles the whole cgroup until the next
# 100 ms period -- seen as single steps with 100-200 ms of host enqueue time (`host_enqueue_ms_by_step`, `cgroup_cpu`) in otherwise flat
# runs.  Passive waiting must be set before the OpenMP runtime loads; the pool size is set in main().
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
YAML = os.path.join(ROOT, "tests", "golden", "autoencoder_kl_16x16x16.yaml")
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_bf16, dense
PEAK_HBM_GBS = 8000.0          # HBM3E
ALGORITHMIC_GFLOP_PER_IMAGE = {256: 1449.9, 512: 8901.6, 64: 75.5}   # fwd + bwd, VAE only


LINE_LIMIT = 4000   # bytes of the ONE stdout line: the driver parses it from a bounded tail (round 4's 22.6 KB line was not parsed)


def _r(v, sig=6):
    """Strict-JSON number: 6 significant digits, no NaN / Infinity (those become null)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, v))
    return v


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d}


def compact_record(full, detail_path=None):
    """The ONE line for stdout, from the full record: the contract's keys, `roofline` (dominant kernel), `roofline_step`,
    `cpu_baseline`, and per side run only {value, ms_per_step, steps, dtype}.  Everything else (per-step arrays, sysfs states, GC,
    cgroup, allocator, kernel variants, the other kernel families) stays in the full record, written to `detail_path`."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "ranks_joined", "backend", "steps", "warmup", "ms_per_step", "higher_is_better",
                       "scaling", "vs_baseline", "dtype", "data", "selftest", "ranks_in_lockstep"))
    if "config" in full:
        out["config"] = full["config"]
    for k in ("host_enqueue_ms_per_step", "gpu_elapsed_ms_per_step", "peak_device_memory_gb"):
        if full.get(k) is not None:
            out[k] = _r(full[k], 5)
    if isinstance(full.get("gpu_step_ms"), dict):
        out["gpu_step_ms_median"] = _r(full["gpu_step_ms"].get("median"), 5)
    if isinstance(full.get("roofline"), dict):
        out["roofline"] = _pick(full["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                                   "algorithmic_gflop_per_launch", "issued_gflop_per_launch", "algorithmic_tflops", "kernel",
                                                   "launches", "sampled_1_in", "avg_launch_ms", "share_of_step_time"))
    if isinstance(full.get("roofline_step"), dict):
        out["roofline_step"] = _pick(full["roofline_step"], ("bound", "achieved", "peak", "unit", "frac", "issued_tflop_per_step",
                                                             "algorithmic_tflop_per_step"))
    if isinstance(full.get("roofline_others"), list):    # one short row per further kernel family: [name, bound, frac, share of the step]
        out["roofline_others"] = [[str(e.get("kernel", "")).split(" ")[0], e.get("bound"), _r(e.get("frac"), 3), _r(e.get("share_of_step_time"), 3)]
                                  for e in full["roofline_others"]]
    if isinstance(full.get("cpu_baseline"), dict):
        out["cpu_baseline"] = _pick(full["cpu_baseline"], ("value", "unit", "cores", "kind", "sample"))
        if full["cpu_baseline"].get("cores_busy") is not None:
            out["cpu_baseline"]["cores_busy"] = _r(full["cpu_baseline"]["cores_busy"], 3)
        if isinstance(full["cpu_baseline"].get("config1"), dict):
            out["cpu_baseline"]["config1"] = _pick(full["cpu_baseline"]["config1"], ("value", "cores", "sample"))
    if isinstance(full.get("other_configs"), dict):
        oc = {}
        for k, v in full["other_configs"].items():
            key = v.get("short", k) if isinstance(v, dict) else k
            oc[key] = ({"error": str(v.get("error"))[:120]} if isinstance(v, dict) and "error" in v
                       else _pick(v, ("value", "ms_per_step", "steps", "dtype")))
        out["other_configs"] = oc
    if detail_path:
        out["detail"] = detail_path
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    for drop in ("roofline_others", "other_configs", "gpu_step_ms_median", "peak_device_memory_gb", "host_enqueue_ms_per_step"):
        if len(line) <= LINE_LIMIT:
            break
        out.pop(drop, None)     # never the contract keys, `roofline` or `cpu_baseline`
        line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    return line


def _json_safe(o):
    """The full record as strict JSON too (NaN / Infinity -> null)."""
    if isinstance(o, dict):
        return {str(k): _json_safe(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_json_safe(v) for v in o]
    if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
        return None
    return o


def emit(full, json_fd, name="bench_detail.json"):
    """Full record -> bench_detail.json beside this script (and a copy under gpurun_out/ when that directory exists, so a gpurun call
    brings it home) and stderr stays free of it; the compact line -> the saved stdout descriptor."""
    rel = None
    full = _json_safe(full)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(full, f, indent=1, allow_nan=False)
                rel = rel or os.path.relpath(os.path.join(d, name), ROOT)
            except OSError as e:
                sys.stderr.write("[bench] could not write the detail record in %s: %s\n" % (d, e))
    line = compact_record(full, rel)
    os.write(json_fd, (line + "\n").encode())
    return line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)     # >= 50 timed steps after >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--gan", action="store_true",
                    help="BASELINE.json configs[3]: PatchGAN discriminator + LPIPS-style loss, both optimizers per batch")
    ap.add_argument("--bf16", action="store_true",
                    help="BASELINE.json configs[4]: bf16 mixed precision (bf16 activations in HBM, fp32 master weights and accumulation)")
    ap.add_argument("--ckpt-decoder", action="store_true",
                    help="BASELINE.json configs[4]: activation-checkpointed Decoder (each up level / mid block is recomputed in backward)")
    ap.add_argument("--ckpt-policy", choices=("unit", "norm"), default="unit",
                    help="with --ckpt-decoder: unit = torch.utils.checkpoint per ResnetBlock(+AttnBlock) (BASELINE.json configs[4]); norm = keep the conv "
                         "outputs, re-make only the GroupNorm+swish tensors in the backward (modules.Decoder)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: initialise RCCL with world size 1 and run the bucketed reducer anyway")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU rehearsal of the N-rank launch path: ranks rendezvous over gloo, count themselves and run the "
                         "bucketed reducer on host tensors; no HIP kernel runs and the printed line is marked as such")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short side measurements (configs[4] in bf16, the headline shape in bf16) the default run appends")
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


def cpu_baseline(res, batch=1, steps=1, warmup=1):
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
    for _ in range(warmup):
        train_batch(ref, opts, batch_d, {0: noise}, optimizer_indices=(0,), clip=1.0)
    cg0, c0 = cgroup_cpu(), os.times()
    t0 = time.time()
    for _ in range(steps):
        train_batch(ref, opts, batch_d, {0: noise}, optimizer_indices=(0,), clip=1.0)
    dt = time.time() - t0
    cg1, c1 = cgroup_cpu(), os.times()
    cpu_s = (c1.user - c0.user) + (c1.system - c0.system)
    # cores_busy: CPU seconds this process burnt per wall second of the timed leg -- what "cores" really delivered (a quota-throttled or
    # oversubscribed host shows here, and in the cgroup's throttle counters, not in the thread count)
    return {"value": batch * steps / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d warm-up + %d timed step(s), B=%d, %dx%d, fp32, rec+KL only, torch %s CPU oracle"
                      % (warmup, steps, batch, res, res, torch.__version__),
            "wall_s": dt, "cores_busy": cpu_s / dt if dt > 0 else None,
            "cgroup_cpu_delta": {k: cg1[k] - cg0[k] for k in cg1 if k in cg0}}


def host_enqueue_ms(step, first_index, reps=3):
    """Host time of ONE step issued into an empty queue (python + autograd + launch calls, nothing to wait for): the minimum over
    `reps` single steps, each behind a device synchronisation.  The loop figure `host_loop_ms_per_step` cannot tell this from
    back-pressure -- once the host is a few steps ahead the runtime blocks it, and it then reads the GPU's own time per step."""
    best = None
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(first_index + r)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    torch.cuda.synchronize()
    return best


def _amdgpu_sysfs(dev):
    """sysfs directory of the amdgpu card behind torch device `dev` (matched by PCI address), or None."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(dev)
        want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:  # noqa: BLE001
        want = None
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    for c in cards:
        try:
            if want and os.path.basename(os.path.realpath(c)).startswith(want):
                return c
        except OSError:
            pass
    return cards[0] if len(cards) == 1 else None


def gpu_state(dev):
    """Shader / memory clock, power draw and cap, temperature as the kernel driver reports them right now (sysfs; no GPU call, so
    it is safe between timed regions).  Missing files are skipped: the record says what the box let an ordinary user read."""
    import glob
    out = {}
    base = _amdgpu_sysfs(dev)
    if base is None:
        return {"error": "no amdgpu sysfs entry matched the device"}

    def rd(path):
        try:
            return open(path).read().strip()
        except OSError:
            return None
    for key, fn in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk")):
        txt = rd(os.path.join(base, fn))
        if txt:
            cur = [l for l in txt.splitlines() if l.rstrip().endswith("*")]
            out[key + "_mhz_current"] = cur[0].split(":")[1].replace("*", "").strip() if cur else None
            out[key + "_levels"] = [l.split(":")[1].replace("*", "").strip() for l in txt.splitlines() if ":" in l]
    for hw in glob.glob(os.path.join(base, "hwmon", "hwmon*")):
        for key, fn, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("power_cap_w", "power1_cap", 1e-6),
                               ("sclk_hwmon_mhz", "freq1_input", 1e-6), ("mclk_hwmon_mhz", "freq2_input", 1e-6),
                               ("temp_edge_c", "temp1_input", 1e-3), ("temp_junction_c", "temp2_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3)):
            v = rd(os.path.join(hw, fn))
            if v is not None and key not in out:
                try:
                    out[key] = float(v) * scale
                except ValueError:
                    pass
    v = rd(os.path.join(base, "gpu_busy_percent"))
    if v is not None:
        out["gpu_busy_percent"] = v
    return out


def cgroup_cpu():
    """CPU-bandwidth throttling of this container so far (cgroup v2 cpu.stat): periods, throttled periods, throttled microseconds."""
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            if k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec"):
                out[k] = int(v)
    except (OSError, ValueError):
        pass
    return out


_ALLOC_KEYS = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms", "num_sync_all_streams")


def alloc_counters(dev):
    st = torch.cuda.memory_stats(dev)
    d = {k: int(st.get(k, 0)) for k in _ALLOC_KEYS}
    d["reserved_gb"] = st.get("reserved_bytes.all.current", 0) / 1e9
    d["allocated_gb"] = st.get("allocated_bytes.all.current", 0) / 1e9
    return d


class GcWatch:
    """Python garbage collections inside a timed region: count and host milliseconds per generation (a full, generation-2 collection of
    this process's ~10^5 objects stalls the host for tens of milliseconds; whether one coincides with a slow step is in the record)."""

    def __init__(self):
        self.log, self._t0 = [], 0.0

    def _cb(self, phase, info):
        if phase == "start":
            self._t0 = time.perf_counter()
        else:
            self.log.append((info["generation"], (time.perf_counter() - self._t0) * 1e3))

    def __enter__(self):
        import gc
        gc.callbacks.append(self._cb)
        return self

    def __exit__(self, *exc):
        import gc
        gc.callbacks.remove(self._cb)

    def summary(self):
        return {"gen%d" % g: {"collections": sum(1 for gg, _ in self.log if gg == g), "host_ms": round(sum(t for gg, t in self.log if gg == g), 2),
                              "longest_ms": round(max([t for gg, t in self.log if gg == g] or [0.0]), 2)} for g in (0, 1, 2)}


def timed_loop(step, first_index, steps, dev):
    """K steps with one HIP event in front of each and one behind the last, on the stream the kernels are launched on: wall time
    (host clock, device-synchronised at both ends), the GPU-side elapsed time of the same loop, per-step GPU times, the host's enqueue
    time, the caching allocator's device-call counters over the loop (a hipMalloc / hipFree inside it is a device synchronisation)
    and the clocks / power the driver reports before and after."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    state0, a0, cg0 = gpu_state(dev), alloc_counters(dev), cgroup_cpu()
    torch.cuda.synchronize()
    host_t = [0.0] * (steps + 1)      # host clock at each step boundary: a step whose ENQUEUE took long names the host as the stalled side
    with GcWatch() as gcw:
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i].record()
            host_t[i] = time.perf_counter()
            step(first_index + i)
        ev[steps].record()
        host_t[steps] = time.perf_counter()
        host = host_t[steps] - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    state1, a1, cg1 = gpu_state(dev), alloc_counters(dev), cgroup_cpu()
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    diag = {"gpu_elapsed_ms_per_step": ev[0].elapsed_time(ev[steps]) / steps,
            "gpu_step_ms": {"min": per[0], "median": per[len(per) // 2], "max": per[-1],
                            "all": [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(steps)] if steps <= 64 else None},
            "host_enqueue_ms_by_step": [round((host_t[i + 1] - host_t[i]) * 1e3, 2) for i in range(steps)] if steps <= 64 else None,
            "allocator_delta": {k: a1[k] - a0[k] for k in _ALLOC_KEYS},
            "python_gc_in_timed_loop": gcw.summary(),
            "cgroup_cpu_delta": {k: cg1[k] - cg0[k] for k in cg1 if k in cg0}, "torch_num_threads": torch.get_num_threads(),
            "allocator_after": {"reserved_gb": a1["reserved_gb"], "allocated_gb": a1["allocated_gb"]},
            "gpu_state_before": state0, "gpu_state_after": state1}
    return wall, host, diag


def side_run(dev, res, batch, steps, warmup, ckpt, precision, gan=False, force_dist=False):
    """A second, smaller measurement in the same process (one GPU): images/s of another BASELINE.json configuration, no kernel
    events.  The caching allocator is emptied first, so a run never starts inside what the previous one left reserved.  Returns the
    dict that goes under `other_configs` of the one JSON line, or the error text -- it never fails the headline.
    force_dist: the data-parallel path on one GPU -- an RCCL group of world size 1, the bucketed GradReducer and the pre-scaled loss."""
    import gc
    from odvae_amd import synthetic
    from odvae_amd.trainer import Trainer
    model = trainer = None
    try:
        gc.collect()
        torch.cuda.empty_cache()
        torch.manual_seed(23)
        lat = res // 16
        gkw = dict(perceptual_weight=1.0, disc_factor=1.0, disc_start=0) if gan else {}
        model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=lat, **gkw).to(dev).train()
        model.decoder.activation_checkpoint = ckpt if ckpt in ("unit", "norm") else bool(ckpt)
        model._global_step = 1
        group = None
        if force_dist:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29541")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            group = dist.group.WORLD
        trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0, 1) if gan else (0,), precision=precision, process_group=group)
        data = synthetic.make_batch(batch, res, seed=23)
        data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}

        def step(i):
            b = dict(data)
            b["pose_6d"] = data["pose_6d"].clone()
            return trainer.training_batch(b, i)
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        dt, host, diag = timed_loop(step, warmup, steps, dev)
        if force_dist and trainer.reducers:      # one more step, outside the timed loop, with the bucket schedule recorded (events on the compute stream)
            trainer.reducers[0].record_timeline = True
            step(warmup + steps)
            diag["dp_bucket_timeline"] = trainer.reducers[0].timeline()
            trainer.reducers[0].record_timeline = False
        host1 = host_enqueue_ms(step, warmup + steps)
        out = {"value": batch * steps / dt, "unit": "images/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
               "host_enqueue_ms_per_step": host1, "host_over_gpu": host1 / (dt / steps * 1e3), "host_loop_ms_per_step": host / steps * 1e3,
               "dtype": "bf16" if str(precision) == "bf16" else "f32", "peak_device_memory_gb": torch.cuda.max_memory_allocated(dev) / 1e9,
               # the same count of images over the MEDIAN step time: what the run gives when no single step is stalled by the host (`value` is the mean,
               # stalls included; `gpu_step_ms` / `host_enqueue_ms_by_step` / `cgroup_cpu_delta` / `python_gc_in_timed_loop` say which steps and why)
               "images_per_s_at_median_step": batch / (diag["gpu_step_ms"]["median"] * 1e-3),
               "backend": "rccl (world size 1, bucketed GradReducer)" if force_dist else "none",
               "config": {"workload": "autoencoder_kl_16x16x16.yaml, %dx%d synthetic crops, z=%dx%dx16, B=%d/GPU, %s%s, %s"
                          % (res, res, lat, lat, batch,
                             "PatchGAN + LPIPS-style loss, optimizers 0 and 1 per batch" if gan else "rec+KL only",
                             (", Decoder checkpointed by the 'norm' policy" if ckpt == "norm" else ", activation-checkpointed Decoder") if ckpt else "",
                             "bf16 mixed precision" if str(precision) == "bf16" else "fp32")}}
        out.update(diag)
        return out
    except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        del model, trainer
        gc.collect()
        torch.cuda.empty_cache()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rank environment: this process touches no GPU (no HIP call, no
    torch.cuda.*: a process that has initialised the device must not spawn-by-exec on this pool) and starts N fresh ranks
    with torch.distributed.run, one per GPU, over RCCL (the reference: `strategy: ddp`, yaml:137, train.py:162).  Rank 0's
    one JSON line is forwarded on stdout; the exit code is non-zero unless exactly N ranks joined."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer mappings need it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [l for l in proc.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write("[bench] %d-rank launch failed (rc %d, %d JSON lines)\n" % (args.gpus, proc.returncode, len(lines)))
        return proc.returncode or 1
    out = json.loads(lines[-1])
    if out.get("n_gpus") != args.gpus or out.get("ranks_joined") != args.gpus:
        sys.stderr.write("[bench] asked for %d ranks, result reports n_gpus=%r ranks_joined=%r\n"
                         % (args.gpus, out.get("n_gpus"), out.get("ranks_joined")))
        return 3
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    return 0


def launcher_selftest(args, json_fd):
    """The N-rank plumbing without a GPU (tests/test_bench_launcher.py): env rendezvous, rank count by all-reduce, the
    bucketed GradReducer on host tensors, max-over-ranks timing, one JSON line from rank 0."""
    import torch.distributed as dist
    from odvae_amd.parallel import GradReducer
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    joined = torch.ones(1)
    dist.all_reduce(joined)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 16))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    red = GradReducer(opt, bucket_mb=0.002, prescaled=True)
    red.broadcast_parameters(net)
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        x = torch.randn(4, 16, generator=torch.Generator().manual_seed(100 * rank + i))
        opt.zero_grad()
        red.prepare_for_backward()
        (net(x).pow(2).mean() * red.inv_world).backward()
        red.finish()
        opt.step()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    digest = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).double().sum()
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        out = {"metric": "launcher self-test (gloo, host tensors; NOT a benchmark)", "value": 4 * world * args.steps / t.item(),
               "unit": "samples/s", "n_gpus": world, "ranks_joined": int(joined.item()), "backend": "gloo",
               "steps": args.steps, "warmup": 0, "ranks_in_lockstep": bool(lo.item() == hi.item()), "selftest": True}
        emit(out, json_fd)
    dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))   # before anything touches the GPU
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus and not args.force_dist:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))
    # RCCL prints banner lines ("Hostname", "Librccl path") on fd 1; the contract is ONE JSON line on stdout, so
    # everything else that lands on fd 1 is routed to stderr and the result is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.launcher_selftest:
        return launcher_selftest(args, json_fd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    if os.environ.get("ODVAE_BENCH_HOST_THREADS", "") != "0":      # (0: leave torch's default pool, for A/B)
        torch.set_num_threads(int(os.environ.get("ODVAE_BENCH_HOST_THREADS", "0")) or max(1, min(4, host_cores() // max(1, world))))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
        joined = torch.ones(1, device=dev)
        dist.all_reduce(joined)                     # the ranks count themselves over RCCL
        ranks_joined = int(joined.item())
    else:
        ranks_joined = 1
    from odvae_amd import ops, synthetic
    from odvae_amd.trainer import Trainer

    torch.manual_seed(23)
    lat = args.res // 16
    gan = dict(perceptual_weight=1.0, disc_factor=1.0, disc_start=0) if args.gan else {}
    model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=lat, **gan).to(dev)
    model.train()
    if args.ckpt_decoder:
        model.decoder.activation_checkpoint = args.ckpt_policy
    # --bf16: the trainer's `precision: bf16` (yaml:139) -- bf16 activations on the bf16 MFMA kernels + fused attention, f32 master
    # weights; with N > 1 the gradient buckets then travel as bf16 too (parallel.GradReducer)
    # steady state: past the very first optimizer step, whose total holds the pose terms only (`global_step >
    # encoder_pretrain_steps`, contperceptual.py:307) and would skip the decoder's backward pass -- every timed step does the
    # full forward + backward + optimizer work even with --warmup 0
    model._global_step = 1
    trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0, 1) if args.gan else (0,),
                      process_group=dist.group.WORLD if use_dist else None, precision="bf16" if args.bf16 else None)
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
        # A HIP timing event is a barrier packet with a release: the next kernel starts with nothing of its predecessor's tail
        # overlapped.  Around every 0.8 ms f32 conv that costs 0.3 % of the step; around every 0.29 ms bf16 conv it cost 25 %
        # (200 vs 251 images/s).  In bf16 mode one launch in eight of the dominant family is bracketed (17 per step).
        ops.KERNEL_EVENTS.sample = 8 if args.bf16 else 1
    sample = 8 if (args.bf16 and not args.no_kernel_events) else 1   # launches / share below are scaled back by it (estimates when > 1)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one per step boundary, on the launch stream
    state0, alloc0, cg0 = gpu_state(dev), alloc_counters(dev), cgroup_cpu()
    gcw = GcWatch().__enter__()
    t0 = time.perf_counter()
    host_t = [0.0] * (args.steps + 1)
    for i in range(args.steps):
        step_ev[i].record()
        host_t[i] = time.perf_counter()
        step(args.warmup + i)
    step_ev[args.steps].record()
    host_t[args.steps] = time.perf_counter()
    host_s = host_t[args.steps] - t0      # the host has enqueued every launch of the timed steps; the GPU is still working
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gcw.__exit__()
    state1, alloc1, cg1 = gpu_state(dev), alloc_counters(dev), cgroup_cpu()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    dom_key = "conv_bf16" if args.bf16 else "conv3x3_128x128"
    mfma_peak = PEAK_BF16_MFMA_TFLOPS if args.bf16 else PEAK_F32_MFMA_TFLOPS
    roof = ops.KERNEL_EVENTS.summary(dom_key) if not args.no_kernel_events else None
    wino4 = False
    if roof is not None and not args.bf16:
        r4 = ops.KERNEL_EVENTS.summary("conv3x3_wino4")      # the F(4x4,3x3) kernel takes the >= 64-channel stride-1 convs: the dominant kernel
        if r4 is not None and r4["total_ms"] > roof["total_ms"]:
            roof, wino4 = r4, True
    ops.KERNEL_EVENTS.issued_timed = ops.KERNEL_EVENTS.issued
    others, extra_ms = {}, None
    if roof is not None:   # ONE more step, outside the timed region, with the secondary kernel families bracketed as well
        ops.KERNEL_EVENTS.extra = True
        ops.KERNEL_EVENTS.sample = 1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(args.warmup + args.steps)
        torch.cuda.synchronize()
        extra_ms = (time.perf_counter() - t1) * 1e3
        others = {k: ops.KERNEL_EVENTS.summary(k) for k in ("conv3x3_wgrad_wino", "gemm_f32", "conv_wgrad_bf16", "conv1x1_bf16",
                                                            "flash_attn", "groupnorm")}
    ops.KERNEL_EVENTS.disable()
    host1_ms = host_enqueue_ms(step, args.warmup + args.steps + 1) if not use_dist else None   # no events, outside the timed region

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "VAE train images/s (fwd+bwd+opt) at %dx%d z=%dx%dx16" % (args.res, args.res, lat, lat),
            "value": args.batch * world * args.steps / elapsed,
            "unit": "images/s",
            "n_gpus": world, "ranks_joined": ranks_joined, "backend": "rccl" if use_dist else "none", "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.bf16 else "f32", "data": "synthetic",
            # python + autograd + launch calls of one step issued into an EMPTY queue (minimum of three single steps behind a device
            # synchronisation, after the timed region): the step is host-bound when this approaches ms_per_step.  host_loop_ms_per_step
            # is the enqueue time inside the timed loop, back-pressure of the launch queue included (it reads ~ms_per_step by itself).
            "host_enqueue_ms_per_step": host1_ms, "host_over_gpu": (host1_ms / (elapsed / args.steps * 1e3)) if host1_ms else None,
            "host_loop_ms_per_step": host_s / args.steps * 1e3,
            # GPU-side view of the same timed loop (rank 0): one HIP event per step boundary on the launch stream
            "gpu_elapsed_ms_per_step": step_ev[0].elapsed_time(step_ev[args.steps]) / args.steps,
            "gpu_step_ms": (lambda per: {"min": min(per), "median": sorted(per)[len(per) // 2], "max": max(per),
                                         "all": [round(v, 3) for v in per] if len(per) <= 64 else None})(
                [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]),
            "host_enqueue_ms_by_step": [round((host_t[i + 1] - host_t[i]) * 1e3, 2) for i in range(args.steps)] if args.steps <= 64 else None,
            "allocator_delta": {k: alloc1[k] - alloc0[k] for k in _ALLOC_KEYS},
            "python_gc_in_timed_loop": gcw.summary(),
            "cgroup_cpu_delta": {k: cg1[k] - cg0[k] for k in cg1 if k in cg0}, "torch_num_threads": torch.get_num_threads(),
            "gpu_state_before": state0, "gpu_state_after": state1,
            "config": {"workload": "autoencoder_kl_16x16x16.yaml, %dx%d synthetic crops, B=%d/GPU, %s, VAE phase"
                       % (args.res, args.res, args.batch,
                          ("PatchGAN + LPIPS-style loss, optimizers 0 and 1 per batch" if args.gan
                           else "rec+KL only (discriminator off, optimizer 0)")
                          + ((", activation-checkpointed Decoder" + (" (norm policy)" if args.ckpt_policy == "norm" else "")) if args.ckpt_decoder else "")
                          + (", bf16 mixed precision (bf16 activations, f32 master weights / accumulation)" if args.bf16 else ", fp32")),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world},
        }
        if roof is not None:
            traffic = None   # HBM bytes per launch of the same kernel, from committed rocprofv3 --pmc passes of this command
            tfile = None
            for cand in (("r05_bf16_conv_traffic.json",) if args.bf16 else ("r05_conv3x3_traffic.json",)):   # this round's PMC passes only
                if os.path.exists(os.path.join(ROOT, "profiles", cand)):
                    tfile = cand
                    break
            if tfile and not args.gan and args.batch == 32 and args.res == 256 and not args.ckpt_decoder:
                tj = json.load(open(os.path.join(ROOT, "profiles", tfile)))
                if args.bf16 or ("wino4" in tj["kernel"]) == wino4:      # a traffic file of another kernel is not this kernel's traffic
                    traffic = tj["hbm_bytes_per_launch"]
            wino = ops.WINOGRAD and not args.bf16
            # multiply-adds ISSUED, counted per launch on the host: Winograd F(2x2,3x3) issues 16 of the direct form's 36 per 2x2 tile,
            # F(4x4,3x3) 36 of 144 per 4x4 tile
            ratio = roof["issued_tflops"] / roof["tflops"]
            issued = roof["issued_tflops"]
            out["roofline"] = {"bound": "mfma", "achieved": issued, "peak": mfma_peak, "unit": "TFLOP/s",
                               "frac": issued / mfma_peak, "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/%s)" % tfile,
                               "kernel": ("conv_bf16_kernel (3x3 conv fwd + dgrad, all modes, bf16 implicit GEMM, 128 px x 128 co per block)" if args.bf16
                                          else "conv3x3_wino4_kernel (stride-1 3x3 conv fwd + dgrad, Winograd F(4x4,3x3), 32 tiles of 4x4 x 64 co per block)"
                                          if wino4 else "conv3x3_wino8_kernel (stride-1 3x3 conv fwd + dgrad, Winograd F(2x2,3x3), 32 tiles x 128 co per block)"
                                          if wino else "conv3x3_kernel_v2<MODE 0,KC 32,2,2,2,2> (3x3 conv fwd + dgrad, 128px x 128co tile)"),
                               "launches": roof["launches"] * sample, "timed_launches": roof["launches"], "sampled_1_in": sample,
                               "avg_launch_ms": roof["avg_ms"],
                               "issued_gflop_per_launch": roof["gflop_per_launch"] * ratio,
                               "algorithmic_gflop_per_launch": roof["gflop_per_launch"],
                               "algorithmic_tflops": roof["tflops"],
                               "algorithmic_bytes_per_launch": roof["bytes_per_launch"],
                               "share_of_step_time": roof["total_ms"] * sample / (ms * args.steps),
                               "note": ("achieved/frac price the multiply-adds actually ISSUED to the f32 MFMA pipe; "
                                        "algorithmic_tflops is the direct-convolution work (2*9*Cin*Cout per pixel, SURVEY.md 8(d)) "
                                        "the same launches deliver" + (", 4x the issued work under Winograd F(4x4,3x3)" if wino4 else
                                                                       ", 36/16 of the issued work under Winograd" if wino else ""))}
            if wino4:
                # the family is one kernel template with different epilogues (rocprof lists them as conv3x3_wino4_kernel<EPI, UP>): the
                # aggregate above prices ALL of them on their matrix work, including the launches whose output transform also carries the
                # GroupNorm statistics / the first pass of a GroupNorm backward (x read once more, ~14 vector operations per element)
                out["roofline"]["variants"] = []
                for v in ops.KERNEL_EVENTS.variants("conv3x3_wino4"):
                    rv = ops.KERNEL_EVENTS.summary("conv3x3_wino4|" + v)
                    out["roofline"]["variants"].append({"epilogue": v, "launches": rv["launches"], "avg_launch_ms": rv["avg_ms"],
                                                        "achieved": rv["issued_tflops"], "frac": rv["issued_tflops"] / mfma_peak,
                                                        "total_ms_per_step": rv["total_ms"] / args.steps})
            # whole step: every multiply-add issued by the MFMA kernels of the timed steps (host-side count per launch)
            step_issued = ops.KERNEL_EVENTS.issued_timed / args.steps
            out["roofline_step"] = {"bound": "mfma", "issued_tflop_per_step": step_issued / 1e12,
                                    "achieved": step_issued / (ms * 1e-3) / 1e12, "peak": mfma_peak, "unit": "TFLOP/s",
                                    "frac": step_issued / (ms * 1e-3) / 1e12 / mfma_peak,
                                    "algorithmic_tflop_per_step": ALGORITHMIC_GFLOP_PER_IMAGE.get(args.res, 0.0) * args.batch / 1e3
                                    if not args.gan else None}
            # the next kernel families of the step, measured the same way (HIP events around every launch) on one extra step
            names = {"conv3x3_wgrad_wino": "conv3x3_wgrad_wino_kernel (weight gradient of the stride-1 3x3 convs in the Winograd domain)",
                     "gemm_f32": "gemm_f32_kernel (attention products incl. the fused softmax backward, 1x1 convs and their gradients)",
                     "flash_attn": "flash_fwd / flash_dq / flash_dkv kernels (fused attention: scores never leave the CU)",
                     "conv_wgrad_bf16": "conv_wgrad_bf16_kernel (weight gradient of the 3x3 / 1x1 convs, transposed LDS reads)",
                     "conv1x1_bf16": "conv_bf16_kernel<MODE 4> (1x1 convs: q/k/v/proj_out, nin_shortcut, and their data gradients)",
                     "groupnorm": "gn_* (GroupNorm(32, eps 1e-6) + swish, forward and backward incl. the folded skip gradient)"}
            out["roofline_others"] = []
            for key, r in others.items():
                if r is None:
                    continue
                e = {"kernel": names[key], "launches": r["launches"], "avg_launch_ms": r["avg_ms"], "total_ms": r["total_ms"],
                     "share_of_step_time": r["total_ms"] / extra_ms,
                     "measured_on": "one extra step after the timed region (HIP events around every launch of this family)"}
                if key == "groupnorm":
                    e.update({"bound": "hbm", "achieved": r["tbytes_per_s"] * 1e3, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": r["tbytes_per_s"] * 1e3 / PEAK_HBM_GBS,
                              "algorithmic_bytes_per_step": r["bytes_per_launch"] * r["launches"],
                              "note": "algorithmic bytes = x read + y written (forward), x, dy (, skip gradient) read + dx written (backward)"})
                    # this round's PMC passes only: a file from before a kernel change is not evidence for these kernels
                    fam = os.path.join(ROOT, "profiles", "r04_family_traffic.json")      # (the GroupNorm kernels are unchanged since these passes)
                    if not (args.gan or args.bf16 or args.ckpt_decoder) and args.res == 256 and args.batch == 32 and os.path.exists(fam):
                        # HBM bytes the family actually moves (statistics / reduce are passes of their own), from the committed PMC
                        # passes of this command (FETCH_SIZE x2 + WRITE_SIZE over `steps_counted` steps), over the time measured live here
                        fj = json.load(open(fam))
                        ks, nsteps = fj["kernels"], float(fj["steps_counted"])
                        moved = sum((v["fetch_bytes_total_corrected"] + v["write_bytes_total"]) / nsteps for k, v in ks.items() if k.startswith("gn_"))
                        e.update({"traffic": moved, "traffic_unit": "HBM bytes per step (PMC, profiles/r04_family_traffic.json)",
                                  "achieved_on_traffic": moved / (e["total_ms"] * 1e-3) / 1e9,
                                  "frac_on_traffic": moved / (e["total_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS})
                else:
                    k = 16.0 / 36.0 if key == "conv3x3_wgrad_wino" else 1.0
                    e.update({"bound": "mfma", "achieved": r["tflops"] * k, "peak": mfma_peak, "unit": "TFLOP/s",
                              "frac": r["tflops"] * k / mfma_peak, "algorithmic_tflops": r["tflops"]})
                out["roofline_others"].append(e)
        out["peak_device_memory_gb"] = torch.cuda.max_memory_allocated(dev) / 1e9
        print("[bench] GPU leg done: %.2f images/s, %.1f ms/step" % (out["value"], ms), file=sys.stderr, flush=True)
        default_cfg = not (args.gan or args.bf16 or args.ckpt_decoder) and args.res == 256 and args.batch == 32
        if world == 1 and default_cfg and not args.no_other_configs:
            # the other measured configurations of BASELINE.json, appended for the record (the headline above is untouched:
            # its model is released first, the side runs carry no events)
            del trainer, model
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            # order: the bf16 run at the headline shape goes FIRST (fresh allocator, nothing bf16 has run yet) and is repeated LAST,
            # after the 512x512 run and the GAN run, so the record itself shows whether its rate depends on what ran before it
            oc = {}
            # short keys (they travel in the one stdout line); each entry's config.workload in bench_detail.json says what it ran
            oc["bf16_256_B32"] = side_run(dev, 256, 32, 30, 5, False, "bf16")                       # configs[1] shape in bf16 mixed precision
            oc["cfg4_bf16_512_ckpt_unit"] = side_run(dev, 512, 32, 4, 2, True, "bf16")               # configs[4]: 512x512, z=32x32x16, checkpointed Decoder
            # configs[4] geometry with the 'norm' checkpoint policy (conv outputs kept, only GroupNorm+swish re-made in the backward: more
            # memory, less recompute; NOT the unit-checkpointed configuration of the line above)
            oc["cfg4_bf16_512_ckpt_norm"] = side_run(dev, 512, 32, 4, 2, "norm", "bf16")
            oc["cfg3_f32_256_gan_lpips"] = side_run(dev, 256, 32, 4, 2, False, 32, gan=True)          # configs[3]: PatchGAN + LPIPS-style, both optimizers
            oc["bf16_256_B32_again"] = side_run(dev, 256, 32, 30, 5, False, "bf16")                   # repeated after the 512x512 and GAN runs
            # the data-parallel path's overhead is the difference between THESE two adjacent runs (the headline ran first, on a cooler chip
            # and an empty allocator: a side run of the same step after the others has measured up to 1.7 % slower than it)
            oc["f32_256_B32_single"] = side_run(dev, 256, 32, 6, 2, False, 32)
            oc["f32_256_B32_rccl_world1"] = side_run(dev, 256, 32, 6, 2, False, 32, force_dist=True)  # RCCL world size 1, bucketed reducer
            out["other_configs"] = oc
            try:
                import torch.distributed as dist
                if dist.is_initialized() and not use_dist:
                    dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            print("[bench] side runs done", file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            # one warm-up step, then >= 3 timed steps (B=2 at 256x256: ~4.5 s each on 16 cores)
            out["cpu_baseline"] = cpu_baseline(args.res, batch=2 if args.res <= 256 else 1, steps=3 if args.res <= 256 else 1, warmup=1)
            # BASELINE.json configs[0] beside it: the reference's own CPU-runnable case (64x64, B=2, 10 steps)
            out["cpu_baseline"]["config1"] = cpu_baseline(64, batch=2, steps=10, warmup=1)
        sys.stdout.flush()
        # the default invocation owns bench_detail.json; any other configuration writes a file of its own name (a profiling run of --gan must
        # not overwrite the headline's record)
        tag = "".join(t for t, on in (("_gan", args.gan), ("_bf16", args.bf16), ("_res%d" % args.res, args.res != 256), ("_b%d" % args.batch, args.batch != 32),
                                      ("_ckpt", args.ckpt_decoder), ("_n%d" % world, world > 1), ("_dist1", args.force_dist and world == 1)) if on)
        emit(out, json_fd, "bench_detail%s.json" % tag)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

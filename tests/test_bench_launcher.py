"""`python bench.py --gpus N` must start N ranks itself (it used to run one rank and print n_gpus: 1).
CPU rehearsal with gloo: the parent spawns torch.distributed.run, the ranks rendezvous on 127.0.0.1, count themselves
by all-reduce, run the product's bucketed GradReducer on host tensors and rank 0's ONE JSON line comes back through
the parent.  The RCCL leg of the same code path needs a multi-GPU node (the driver's SCALE run)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout)


def test_gpus2_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # the contract: ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_joined"] == 2 and out["backend"] == "gloo"
    assert out["ranks_in_lockstep"] is True and out["selftest"] is True


def test_gpus8_launch_path_on_gloo():
    """The N = 8 case of the driver's SCALE run, rehearsed on host tensors (8 gloo ranks on this container's cores): the parent starts
    eight ranks, all eight join, the bucketed reducer keeps them in lockstep, and ONE compact strict-JSON line comes back."""
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "0", "--launcher-selftest"], timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, lines
    out = json.loads(lines[0], parse_constant=_no_constants)
    assert out["n_gpus"] == 8 and out["ranks_joined"] == 8 and out["ranks_in_lockstep"] is True


def test_world_size_mismatch_is_an_error():
    # a launcher that started a different number of ranks than --gpus names must not pass silently
    r = _run(["--gpus", "4", "--launcher-selftest"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"--gpus 4" in r.stderr


def test_parent_does_not_touch_the_gpu():
    """The launching parent must not initialise HIP (a GPU-initialised process may not exec children on the GPU pool):
    the launch branch sits before any torch.cuda call in main()."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args)") < main.index("torch.cuda")
    launch = src[src.index("def launch_ranks"):src.index("def launcher_selftest")]
    assert "torch.cuda" not in launch.replace("torch.cuda.*", "")


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("odvae_bench", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _no_constants(name):
    raise AssertionError("non-strict JSON constant %s in the bench line" % name)


def test_stdout_line_is_compact_strict_json(tmp_path):
    """Round 4's line was 22.6 KB and the driver parsed nothing from it.  The ONE stdout line is built by compact_record() from the
    full record; fed round 4's real full record (plus a NaN and an Infinity) it must stay under 4 KB, be strict JSON, and still
    carry the contract's keys, `roofline`, `roofline_step` and `cpu_baseline`; per side run only value / ms_per_step / steps / dtype."""
    b = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_record_final.json")))
    full["host_over_gpu"] = float("nan")
    full["roofline"]["traffic"] = float("inf")
    full["other_configs"]["a failed side run"] = {"error": "RuntimeError: " + "x" * 5000}
    line = b.compact_record(full, "bench_detail.json")
    assert len(line) < 4096 and "\n" not in line
    out = json.loads(line, parse_constant=_no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "ranks_joined", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_step", "cpu_baseline", "detail"):
        assert k in out, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel", "launches", "avg_launch_ms",
              "share_of_step_time"):
        assert k in out["roofline"], k
    assert out["roofline"]["traffic"] is None            # Infinity -> null
    assert set(out["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert out["config"]["workload"] and "model" not in out["config"]
    for k, v in out["other_configs"].items():
        assert set(v) <= {"value", "ms_per_step", "steps", "dtype", "error"}, (k, v)
    # and the full record goes to a file as strict JSON, the line to the descriptor
    r, w = os.pipe()
    old_root = b.ROOT
    b.ROOT = str(tmp_path)
    try:
        b.emit(full, w)
    finally:
        b.ROOT = old_root
    os.close(w)
    got = os.read(r, 1 << 16).decode()
    os.close(r)
    assert got.endswith("\n") and got.count("\n") == 1 and len(got) < 4096
    json.loads(got, parse_constant=_no_constants)
    detail = json.load(open(os.path.join(str(tmp_path), "bench_detail.json")), parse_constant=_no_constants)
    assert detail["gpu_step_ms"]["all"] and "other_configs" in detail


def test_compact_line_sheds_optional_blocks_before_the_limit():
    b = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_record_final.json")))
    full["other_configs"] = {"side run %03d %s" % (i, "y" * 60): {"value": 1.0, "ms_per_step": 2.0, "steps": 3, "dtype": "f32"} for i in range(60)}
    line = b.compact_record(full, None)
    assert len(line) <= b.LINE_LIMIT
    out = json.loads(line, parse_constant=_no_constants)
    assert "roofline" in out and "cpu_baseline" in out and "other_configs" not in out

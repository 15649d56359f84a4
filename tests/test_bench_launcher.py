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

"""CPU: `python bench.py --gpus N` must run N ranks or fail -- never print an N-GPU line from one rank.

The launcher (bench.launch_ranks) starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
process tree; `--launch-check` runs only the process-group side of the bench (no kernel), so the whole path -- parent,
torch.distributed.run, N ranks, rendezvous on 127.0.0.1, an all-reduce, rank 0's JSON line relayed by the parent -- runs on a
GPU-less box over gloo.  On a multi-GPU node the same command without CNR_DIST_BACKEND checks it over RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(CNR_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_launches_two_ranks():
    p = _run(["--gpus", "2", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["backend"] == "gloo" and out["launch_check"] is True


def test_world_size_must_match_gpus():
    """Started under a launcher whose WORLD_SIZE disagrees with --gpus: refuse (in round 2 this printed n_gpus: 1)."""
    p = _run(["--gpus", "8", "--launch-check"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode != 0 and "must agree" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_failed_rank_fails_the_parent():
    p = _run(["--gpus", "2", "--launch-check"], env_extra={"CNR_DIST_BACKEND": "no_such_backend"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_nccl_launch_refused_without_enough_gpus():
    """The real backend on a box with fewer GPUs than ranks: a message, not a hang and not a smaller run."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = _run(["--gpus", "2", "--launch-check"], env_extra={"CNR_DIST_BACKEND": "nccl"})
    assert p.returncode != 0 and "GPU(s)" in p.stderr


def test_launch_timeout_kills_the_child_tree():
    """a rank bring-up or collective that never returns must end the bench with a message and a non-zero exit.
    One second is less than two ranks need to import torch: the deadline fires, the child process group is killed."""
    import time
    t0 = time.time()
    p = _run(["--gpus", "2", "--launch-check", "--launch-timeout", "1"])
    assert p.returncode != 0 and "did not finish within" in p.stderr, p.stderr[-1000:]
    assert time.time() - t0 < 60
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]

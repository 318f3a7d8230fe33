"""CPU: `python bench.py --gpus N` starts N ranks itself.  The launcher, the rendezvous, the
barrier / MAX-over-ranks timing and the rank-0-prints contract run over gloo with a stub step
(`--selftest-backend gloo`: no hot-path work is done or claimed, the line says so)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=240)


@pytest.mark.timeout(300)
def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-backend", "gloo"])
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert out["steps"] == 3 and out["warmup"] == 1 and out["data"] == "stub"
    assert out["ms_per_step"] > 0 and out["scaling"] == "weak"


@pytest.mark.timeout(300)
def test_gpus_n_without_n_gpus_fails_loudly():
    """The real (RCCL) path on a box with fewer GPUs than asked for: non-zero exit, no JSON line."""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env_extra={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert p.returncode != 0
    assert "GPU(s) are visible" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(300)
def test_world_size_mismatch_is_an_error():
    """Under a launcher (WORLD_SIZE set) --gpus must equal the world size: never print n_gpus != --gpus."""
    p = _run(["--gpus", "4", "--selftest-backend", "gloo"],
             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"},
             drop=())
    assert p.returncode == 2
    assert "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


@pytest.mark.timeout(300)
def test_a_dying_rank_stops_the_job():
    """--steps 0 makes rank 0 raise (division by zero when it formats the line) while rank 1 waits in the final
    barrier: the parent must notice, stop rank 1 and report a non-zero exit instead of hanging."""
    p = _run(["--gpus", "2", "--steps", "0", "--selftest-backend", "gloo"])
    assert p.returncode != 0


@pytest.mark.timeout(300)
def test_train_line_launcher_path_over_gloo():
    """`bench.py --gpus 2 --train nrms`: the same spawn / rendezvous / rank count as the forward line, then the training
    line's own plumbing -- weights broadcast from rank 0, uniform shard layout, ONE embedding all-gather and ONE gradient
    bucket all-reduce per step, MAX-over-ranks timing -- with a CPU stand-in model over gloo; rank 0 prints one line."""
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--train", "nrms", "--selftest-backend", "gloo"])
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["data"] == "stub"
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak" and out["loss_finite"] is True
    assert out["unit"] == "impressions/s" and out["value"] > 0 and out["ms_per_step"] > 0
    assert "two history encodes" in out["config"]["workload"] and "all-reduce" in out["config"]["parallelism"]
    # ... and with one asynchronous bucket per tower
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--train", "nrms", "--overlap-buckets", "--selftest-backend", "gloo"])
    assert p.returncode == 0, p.stderr
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["loss_finite"] is True and "per tower" in out["config"]["parallelism"]

"""bench.py's N-rank launch path, rehearsed on CPU: `python bench.py --gpus 2` must start two ranks itself (no
WORLD_SIZE in the environment), gather every row exactly once and report n_gpus == 2 (the flag used to be
ignored).  Uses gloo; the GPU form differs only in the backend ("nccl" = RCCL) and the step function."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_flag_spawns_ranks_gloo():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["gloo_ranks"] == 2 and res["gather_complete"] is True
    assert res["steps"] == 3 and res["warmup"] == 1


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "1", "--launcher-selftest"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_single_rank_selftest():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--batch", "2", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1

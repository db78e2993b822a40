"""`python bench.py --gpus N` launches its own ranks: the parent never touches the GPU, starts N
children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's line and fails if a child fails.  Driven
here with --dry-run (gloo on the CPU, no HIP library) at N = 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("gather", ["all", "rank0"])
def test_bench_launches_its_own_ranks_dry_run(gather):
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run", "--gather", gather])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["self_launched"] is True
    assert out["images_gathered"] == 6 and out["image_ids"] == [0, 1, 2, 3, 4, 5]      # both ranks reported their block


def test_bench_single_rank_dry_run_needs_no_launcher():
    r = _run(["--gpus", "1", "--steps", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_ranks_seen"] == 1 and out["self_launched"] is False


def test_bench_launcher_fails_when_a_rank_fails():
    # an outer launcher's WORLD_SIZE that disagrees with --gpus is refused by every rank; the self-launcher reports failure
    r = _run(["--gpus", "2", "--dry-run", "--steps", "1", "--no-such-flag"])
    assert r.returncode != 0


def test_bench_under_an_outer_launcher_is_one_rank():
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "3", "RANK": "0"}, timeout=60)
    assert r.returncode != 0

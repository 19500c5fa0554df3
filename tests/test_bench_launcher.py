"""bench.py --gpus N without a launcher must start N ranks itself (VERDICT r1 / ADVICE r1: it used to run ONE rank and
print a normal-looking n_gpus=1 line).  CPU-only: --launcher-selftest makes the ranks rendezvous on gloo and run the
bench's own gather plumbing, nothing else.  The frame loops this shards: gs_trainer.py:463,551,616."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_launcher_spawns_two_ranks_and_relays_rank0():
    r = run(["--gpus", "2", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["rank_ids"] == [0, 1] and out["max_over_ranks_ok"]
    assert "[launcher] started 2 ranks" in r.stderr


def test_world_size_mismatch_is_an_error_not_a_single_rank_number():
    r = run(["--gpus", "4", "--launcher-selftest"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_rccl_mode_refuses_more_ranks_than_gpus():
    """Here there is no GPU at all: one rank per GPU over RCCL cannot be formed, and the launcher says so instead of
    quietly measuring fewer ranks."""
    r = run(["--gpus", "2"])
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_a_rank_that_dies_takes_the_others_down_promptly():
    """the launcher used to wait on rank 0 only -- with rank 1 dead before the rendezvous, rank 0 sat in
    init_process_group until the process-group timeout.  Now every rank is watched, as torchrun does."""
    import time
    t0 = time.monotonic()
    r = run(["--gpus", "2", "--launcher-selftest"], {"HGS_SELFTEST_FAIL_RANK": "1", "HGS_BENCH_INIT_TIMEOUT_S": "600"}, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank 1 exited with code 7" in r.stderr and "stopping the other ranks" in r.stderr
    assert time.monotonic() - t0 < 60.0


def test_launcher_deadline():
    r = run(["--gpus", "2", "--launcher-selftest"], {"HGS_SELFTEST_HANG_RANK": "1", "HGS_BENCH_DEADLINE_S": "8"}, timeout=120)
    assert r.returncode != 0 and "deadline" in r.stderr and r.stdout.strip() == ""

"""bench.py must never label a 1-GPU run as N GPUs (round-1 advisor finding): without WORLD_SIZE it starts the N ranks
itself, and when the ranks cannot get a GPU each the whole command fails instead of printing a line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, timeout=timeout)


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4", "--ne", "2", "--qsize", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"WORLD_SIZE=2" in r.stdout and b'"n_gpus"' not in r.stdout


def test_gpus_2_without_launcher_starts_two_ranks_or_fails():
    """here (no GPU) the two self-launched ranks must refuse to run; the parent relays the failure"""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible: the self-launched run would succeed")
    r = _run(["--gpus", "2", "--ne", "2", "--qsize", "1", "--steps", "3", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0, r.stdout[-2000:]
    assert b'"n_gpus"' not in r.stdout
    assert b"2 ranks but" in r.stdout      # message of the rank processes: they were really started as 2 ranks

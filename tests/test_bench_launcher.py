"""CPU tests of the harness around the engine: `bench.py --gpus N` starts N ranks by itself (before anything touches a GPU), refuses a
mislabelled job, and the Python mirror picks the residency flags the C ABI expects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_launches_that_many_ranks():
    """MSM_BENCH_LAUNCH_ONLY=1: every child reports its rank environment and leaves before importing torch or the engine; rank 0's
    line is the one forwarded."""
    p = _run(["--gpus", "3"], {"MSM_BENCH_LAUNCH_ONLY": "1"})
    assert p.returncode == 0, p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line == {"rank": 0, "local_rank": 0, "n_gpus": 3, "master": "127.0.0.1", "argv_gpus": 3}


def test_gpus_flag_must_match_the_launcher():
    p = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "MSM_BENCH_LAUNCH_ONLY": "1"})
    assert p.returncode != 0 and "mislabelled" in p.stderr


def test_more_gpus_than_the_node_has_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("a node with 64 GPUs")
    p = _run(["--gpus", "64"], {})
    assert p.returncode != 0 and "GPU" in p.stderr
    assert p.stdout.strip() == ""        # never a single-GPU line in disguise


def test_residency_flags_of_the_python_mirror():
    import msm_amd
    f = msm_amd.Curve._residency_flags
    assert f(False, False) == 0
    assert f(True, True) == msm_amd.FLAG_DEVICE
    assert f(True, False) == msm_amd.FLAG_DEVICE_BASES          # resident SRS, scalars from the host each call
    with pytest.raises(msm_amd.MsmAmdError):
        f(False, True)

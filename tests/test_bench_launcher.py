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
    """The pre-check of the launcher itself (the branch the driver's `--gpus N` takes on a real node): GPUs counted from sysfs -- here
    a stated count, this container has no KFD topology -- and NEVER through torch or HIP: launch_ranks asserts that neither torch
    nor the engine has been imported by the parent, on the refusing and on the launching path."""
    p = _run(["--gpus", "64"], {"MSM_BENCH_FAKE_GPU_COUNT": "8"})
    assert p.returncode == 2 and "shows 8 GPU" in p.stderr and "AssertionError" not in p.stderr
    assert p.stdout.strip() == ""        # never a single-GPU line in disguise


def test_the_launcher_counts_gpus_without_a_gpu_runtime():
    """count_gpus_sysfs reads /sys/class/kfd only; a node without that topology gives None (the ranks then find out themselves), and
    importing bench.py pulls in neither torch nor the engine."""
    code = ("import sys; sys.argv = ['bench.py']; import bench; r = bench.count_gpus_sysfs(); "
            "assert r is None or isinstance(r, int); assert 'torch' not in sys.modules and 'msm_amd' not in sys.modules; print('ok', r)")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and p.stdout.startswith("ok"), p.stderr


def test_a_rank_without_a_gpu_fails_the_job():
    """Where sysfs says nothing the launcher starts the ranks and the ranks find out: with no GPU at all (this container) every rank
    exits non-zero, the launcher ends the job with that code and prints no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a node with GPUs")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"MSM_BENCH_LAUNCH_TIMEOUT": "100"}, timeout=200)
    assert p.returncode != 0
    assert not any(line.startswith("{") for line in p.stdout.splitlines())


def test_residency_flags_of_the_python_mirror():
    import msm_amd
    f = msm_amd.Curve._residency_flags
    assert f(False, False) == 0
    assert f(True, True) == msm_amd.FLAG_DEVICE
    assert f(True, False) == msm_amd.FLAG_DEVICE_BASES          # resident SRS, scalars from the host each call
    with pytest.raises(msm_amd.MsmAmdError):
        f(False, True)

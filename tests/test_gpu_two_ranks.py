"""The multi-rank code of csrc/dist.hip with MORE THAN ONE rank, on a box with one GPU (round 4).

RCCL refuses two ranks on one device, and the driver's one-GPU boxes are all this suite ever sees -- so until round 4 dist.hip's
sequencing, failure marks and split planning had only ever run with a world of one.  Here two freshly started processes both open GPU 0
and form a group through the C ABI; the collective library dist.hip loads is named by MSM_AMD_RCCL_LIB and is tests/host/fake_rccl.c:
the same entry points (ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommAbort / ncclCommCount / ncclCommDestroy), stream-
ordered and asynchronous like the real ones, carried over POSIX shared memory.  Everything else is the product's own code: the split
(msm_dist_split), the slices, the shard calls on the GPU, tickets, exchanges in submission order, header checks, the bounded wait and
the abort path, the host combine.  The two real-RCCL tests of tests/test_gpu_configs.py stay as they are (they need two GPUs).

Reference seams the split follows: wasmcurves/src/build_multiexp.js:96-104, :469 (per-window _chunk), :251-371 (one point slice)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


def build_fake_rccl():
    src, out = os.path.join(HOST, "fake_rccl.c"), os.path.join(HOST, "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", out,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"])
    return out


def run_ranks(scenario, world=2, timeout=400):
    lib = build_fake_rccl()
    with tempfile.TemporaryDirectory(prefix="msm_two_ranks_") as scratch:
        env = dict(os.environ, MSM_AMD_RCCL_LIB=lib, FAKE_RCCL_TIMEOUT_MS="60000")
        logdir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(logdir):
            env["MSM_TWO_RANK_LOG"] = logdir      # progress of every rank (a hung run shows where it stopped)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (the ranks' stderr goes to files when the run keeps logs: a hung run is killed from outside, and what it had printed is the evidence)
        errs = [open(os.path.join(logdir, f"two_ranks_{scenario}_rank{r}.stderr"), "w+") if "MSM_TWO_RANK_LOG" in env else tempfile.TemporaryFile("w+") for r in range(world)]
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py"), scenario, str(r), str(world), scratch],
                                  env=env, stdout=subprocess.PIPE, stderr=errs[r], text=True) for r in range(world)]
        outs = []
        try:
            for r, p in enumerate(procs):
                so, _ = p.communicate(timeout=timeout)
                errs[r].seek(0)
                outs.append((so, errs[r].read()))
        finally:
            for p in procs:      # (these are this test's own children, by handle)
                if p.poll() is None:
                    p.kill()
            for f in errs:
                f.close()
    res = []
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exited {p.returncode}\nstdout:\n{so}\nstderr:\n{se[-4000:]}"
        res.append(json.loads(so.strip().splitlines()[-1]))
    return res


def test_two_ranks_one_gpu_through_the_c_entry_points():
    """msm_amd_dist_init on two processes; msm_amd_multiexp_dist[_async]; the planned and both forced splits with slice-local inputs
    (msm_amd_dist_plan / msm_amd_multiexp_dist_local), ragged and empty slices; G1 and G2 tickets in flight together and waited for in
    OPPOSITE orders on the two ranks from three threads each; a call that fails on one rank only (that rank: its own error, the other:
    MSM_AMD_EPEER, the group goes on); prepared tables over unequal slices sharing the sequence with ordinary calls; ranks that
    disagree on the split or on the window layout (MSM_AMD_EINVAL on both instead of a wrong point); msm_amd_dist_destroy with
    tickets nobody waited for.  Rank 0 checks every result against the oracle."""
    res = run_ranks("exchange")
    for r in res:
        assert r["ok"], r
    assert len({r["digest"] for r in res}) == 1 and res[0]["digest"]


def test_a_rank_that_never_enters_the_exchange():
    """One rank submits a call the other never does: after dist_timeout_ms the waiting rank aborts its communicator (ncclCommAbort),
    returns MSM_AMD_EPEER, fails the ticket queued behind it and every later call of the group AT ONCE, keeps serving plain calls;
    msm_amd_dist_destroy + msm_amd_dist_init on both ranks bring the group back."""
    res = run_ranks("timeout")
    for r in res:
        assert r["ok"], r
    assert len({r["digest"] for r in res}) == 1 and res[0]["digest"]


def test_bench_gpus_2_takes_the_c_abi_exchange_on_one_gpu():
    """`python bench.py --gpus 2` end to end through csrc/dist.hip with two ranks: bench.py starts its ranks itself (the parent never
    touches torch or HIP), the ranks share GPU 0 (MSM_BENCH_BACKEND=gloo carries the id, the barrier and the MAX over ranks), the engine's
    own communicator is the shared-memory stand-in (MSM_AMD_RCCL_LIB) -- the driver's N > 1 flow, minus xGMI.  The line must say that
    the C-ABI exchange ran with two ranks, and the result must be the one-GPU run's."""
    lib = build_fake_rccl()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MSM_BENCH_BACKEND="gloo", MSM_AMD_RCCL_LIB=lib)
    args = ["--steps", "6", "--warmup", "2", "--logn", "16", "--no-cpu-baseline"]
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env, capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0, p2.stderr[-3000:]
    two = json.loads(p2.stdout.strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and two["config"]["exchange"].startswith("C ABI") and two["config"]["rccl_ranks"] == 2, two["config"]
    assert two["config"]["split"]["gp"] * two["config"]["split"]["gw"] == 2
    env1 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env1, capture_output=True, text=True, timeout=600)
    assert p1.returncode == 0, p1.stderr[-3000:]
    one = json.loads(p1.stdout.strip().splitlines()[-1])
    assert one["result_x_prefix"] == two["result_x_prefix"]

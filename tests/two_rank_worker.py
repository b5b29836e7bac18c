"""One rank of tests/test_gpu_two_ranks.py: a freshly started process that opens GPU 0 (every rank shares the one GPU of the box),
joins a group through the C ABI -- csrc/dist.hip, with the collective library named by MSM_AMD_RCCL_LIB: tests/host/fake_rccl.c over
shared memory -- and runs the scenario named on the command line.  The ranks talk to each other through files in a scratch directory
(the 128-byte id, barriers): no torch.distributed, nothing but the C ABI in the data path.

usage: two_rank_worker.py <scenario> <rank> <world> <scratch dir>;  prints one JSON line {"rank", "ok", "digest", "notes"}"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zprize-wasm-msm_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

import golden_util as gu  # noqa: E402
import msm_amd  # noqa: E402
from oracle import oracle as orcm  # noqa: E402  (the checker: rank 0 compares what the group computed with it)

scenario, rank, world, scratch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
notes = []
_T0 = time.time()
_LOG = os.environ.get("MSM_TWO_RANK_LOG")      # a directory: progress lines of every rank (what a hung run was doing)


def say(what):
    if _LOG:
        with open(os.path.join(_LOG, f"two_ranks_{scenario}_rank{rank}.log"), "a") as f:
            f.write(f"[{time.time() - _T0:7.2f}] {what}\n")


def put(name, data=b"x"):
    tmp = os.path.join(scratch, f".{name}.{rank}.tmp")
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, os.path.join(scratch, name))


def get(name, timeout=150.0):
    path = os.path.join(scratch, name)
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: waited {timeout} s for {name}")
        time.sleep(0.002)
    with open(path, "rb") as f:
        return f.read()


def barrier(tag):
    put(f"bar.{tag}.{rank}")
    for r in range(world):
        get(f"bar.{tag}.{r}")


def join(eng, tag):
    """rank 0 makes the id, everybody joins (msm_amd_dist_unique_id / msm_amd_dist_init)"""
    if rank == 0:
        put(f"id.{tag}", msm_amd.dist_unique_id())
    eng.dist_init(rank, world, get(f"id.{tag}"))


def check(cond, what):
    say(("ok: " if cond else "FAILED: ") + what)
    if not cond:
        notes.append("FAILED: " + what)
    return bool(cond)


def slices(C, group, bases, scalars, plan):
    asz = C.affine_bytes(group)
    return (bases[plan["first"] * asz:(plan["first"] + plan["count"]) * asz], scalars[plan["first"] * 32:(plan["first"] + plan["count"]) * 32])


def main():
    ok = True
    say("imports done")
    eng = msm_amd.Engine(0)
    say("engine created")
    digest = ""
    try:
        join(eng, "a")
        say("joined the group")
        eng.set_option("slots", 4)
        check(eng.dist_info() == (rank, world), "dist_info")
        cnt, dead = eng.dist_comm_count()
        ok &= check(cnt == world and not dead, f"the communicator counts {cnt} ranks")
        curve, group, n = "bn128", 1, 6000
        C = eng.curve(curve)
        bases, scalars = gu.random_case(curve, group, "gpu:two-ranks", n)
        exp = None
        if rank == 0:
            orcm.build()
            O = orcm.Oracle(curve, group)
            exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))

        if scenario == "exchange":
            # (1) every rank holds everything (msm_amd_multiexp_dist): resident inputs, one synchronous call, two in flight
            d_b = eng.alloc(len(bases)).upload(bases)
            d_s = eng.alloc(len(scalars)).upload(scalars)
            r1 = C.multiexp_dist(group, d_b, d_s, 32, n)
            t = [C.multiexp_dist_async(group, d_b, d_s, 32, n) for _ in range(2)]
            r2, r3 = t[0].wait(), t[1].wait()
            ok &= check(r1 == r2 == r3, "the same bytes from three calls")
            if rank == 0:
                ok &= check(r1 == exp, "multiexp_dist against the oracle")
            digest = r1.hex()[:24]
            # (2) every rank holds only ITS slice (msm_amd_dist_plan + msm_amd_multiexp_dist_local): the planned split and both forced ones
            for gw in (0, 1, 2):
                eng.set_option("dist_gw", gw)
                plan = eng.dist_plan(curve, group, 32, n)
                if gw:
                    ok &= check(plan["gw"] == gw and plan["gp"] == world // gw, f"forced split gw = {gw}")
                lb, ls = slices(C, group, bases, scalars, plan)
                ok &= check(C.multiexp_dist_local(group, lb, ls, 32, plan["count"], n) == r1, f"slice-local inputs, dist_gw = {gw}: {plan}")
            eng.set_option("dist_gw", 0)
            # ragged slices: a point count the slices do not divide, and one so small that the last slice is EMPTY on the last rank
            for nn in (5999, 1025, 1):
                eng.set_option("dist_gw", 1)
                plan = eng.dist_plan(curve, group, 32, nn)
                lb, ls = slices(C, group, bases[:nn * C.affine_bytes(group)], scalars[:nn * 32], plan)
                got = C.multiexp_dist_local(group, lb, ls, 32, plan["count"], nn)
                if rank == 0:
                    ok &= check(got == O.normalize(O.multiexp_affine(bases, scalars, 32, nn)), f"ragged slices at n = {nn}: {plan}")
            eng.set_option("dist_gw", 0)
            # (3) a G1 and a G2 call in flight together; the two ranks wait for the tickets in OPPOSITE orders, each from three threads:
            #     the exchanges still pair up call by call (submission order), 144-byte and 288-byte partials never meet
            C2 = eng.curve("bls12381")
            b1, s1 = gu.random_case("bls12381", 1, "gpu:two-ranks:g1", 3000)
            b2, s2 = gu.random_case("bls12381", 2, "gpu:two-ranks:g2", 500)
            tk = [C2.multiexp_dist_async(1, b1, s1, 32, 3000), C2.multiexp_dist_async(2, b2, s2, 32, 500), C2.multiexp_dist_async(1, b1, s1, 32, 3000)]
            got = [None] * 3

            def waiter(k):
                try:
                    got[k] = tk[k].wait()
                except Exception as e:   # noqa: BLE001
                    notes.append(f"waiter {k}: {e}")

            order = (2, 1, 0) if rank == 0 else (0, 2, 1)
            th = []
            for k in order:
                x = threading.Thread(target=waiter, args=(k,))
                x.start()
                th.append(x)
                time.sleep(0.05)      # the later tickets are well inside msm_amd_dist_wait before the first one is asked for
            for x in th:
                x.join(timeout=300)
            ok &= check(got[0] is not None and got[0] == got[2] and got[1] is not None, "mixed G1 / G2 tickets waited for in opposite orders")
            if rank == 0:
                O1, O2 = orcm.Oracle("bls12381", 1), orcm.Oracle("bls12381", 2)
                ok &= check(got[0] == O1.normalize(O1.multiexp_affine(b1, s1, 32, 3000)), "the G1 call against the oracle")
                ok &= check(got[1] == O2.normalize(O2.multiexp_affine(b2, s2, 32, 500)), "the G2 call against the oracle")
            # (4) a rank whose part fails: rank 1 passes a slice length that contradicts the plan -- it gets its own error, rank 0
            #     MSM_AMD_EPEER, nobody hangs, and the next call is fine
            plan = eng.dist_plan(curve, group, 32, n)
            lb, ls = slices(C, group, bases, scalars, plan)
            try:
                C.multiexp_dist_local(group, lb, ls, 32, plan["count"] - (1 if rank == 1 else 0), n)
                ok &= check(False, "a contradicting slice length went through")
            except msm_amd.MsmAmdError as e:
                ok &= check(e.code == (msm_amd.EINVAL if rank == 1 else msm_amd.EPEER), f"failure on one rank: rank {rank} got {e.code}")
            ok &= check(C.multiexp_dist_local(group, lb, ls, 32, plan["count"], n) == r1, "the call after a failed one")
            # (5) prepared tables in the group, UNEQUAL slices: rank 0 holds 4500 points, rank 1 the other 1500
            cut = 4500
            asz = C.affine_bytes(group)
            mine_b = bases[:cut * asz] if rank == 0 else bases[cut * asz:]
            mine_s = scalars[:cut * 32] if rank == 0 else scalars[cut * 32:]
            pb = C.prepare_bases(group, mine_b, cut if rank == 0 else n - cut)
            try:
                ok &= check(pb.multiexp_dist(mine_s) == r1, "prepared tables over unequal slices")
                ta, tb = pb.multiexp_dist_async(mine_s), C.multiexp_dist_async(group, d_b, d_s, 32, n)    # prepared and ordinary calls share the sequence
                ok &= check(tb.wait() == r1 and ta.wait() == r1, "a prepared and an ordinary call in flight, waited for out of order")
            finally:
                pb.release()
            # (6) ranks that disagree about the split (an option set on one rank only): MSM_AMD_EINVAL on BOTH, not a wrong point
            if rank == 1:
                eng.set_option("dist_gw", 2)
            else:
                eng.set_option("dist_gw", 1)
            try:
                C.multiexp_dist(group, d_b, d_s, 32, n)
                ok &= check(False, "ranks with different splits got a result")
            except msm_amd.MsmAmdError as e:
                ok &= check(e.code == msm_amd.EINVAL and "different splits" in str(e), f"different splits: {e}")
            eng.set_option("dist_gw", 0)
            # ... and about the window layout inside one point slice (window groups of one slice must use the same windows)
            eng.set_option("dist_gw", 2)
            if rank == 1:
                eng.set_window_bits(12)
            try:
                C.multiexp_dist(group, d_b, d_s, 32, n)
                ok &= check(False, "ranks with different window layouts got a result")
            except msm_amd.MsmAmdError as e:
                ok &= check(e.code == msm_amd.EINVAL and "window layouts" in str(e), f"different layouts: {e}")
            eng.set_window_bits(0)
            eng.set_option("dist_gw", 0)
            ok &= check(C.multiexp_dist(group, d_b, d_s, 32, n) == r1, "the group goes on after both disagreements")
            # (7) leaving the group with a call submitted and never waited for: its slot is collected, the context is an ordinary one again
            for _ in range(4):
                C.multiexp_dist_async(group, d_b, d_s, 32, n)      # four slots, all taken, nobody waits
            barrier("orphans")       # (both ranks have queued theirs: nobody destroys while the other still submits)
            eng.dist_destroy()
            got = [C.multiexp_async(group, d_b, d_s, 32, n) for _ in range(4)]     # all four slots are free again
            ok &= check(all(t.wait() == r1 for t in got), "four plain calls after a destroy with four orphaned tickets")

        elif scenario == "timeout":
            # one rank never enters an exchange: the other gives up after dist_timeout_ms, aborts the communicator, and every later call of
            # the (dead) group returns MSM_AMD_EPEER at once; destroy + a new init on both ranks bring the group back
            eng.set_option("dist_timeout_ms", 1500)
            r1 = C.multiexp_dist(group, bases, scalars, 32, n)
            if rank == 0:
                ok &= check(r1 == exp, "multiexp_dist against the oracle")
                t0 = time.time()
                t_a = C.multiexp_dist_async(group, bases, scalars, 32, n)
                t_b = C.multiexp_dist_async(group, bases, scalars, 32, n)     # a second one behind it: must not block either
                try:
                    t_a.wait()
                    ok &= check(False, "an exchange nobody else entered completed")
                except msm_amd.MsmAmdError as e:
                    dt = time.time() - t0
                    ok &= check(e.code == msm_amd.EPEER and "did not complete" in str(e) and 1.0 < dt < 30.0, f"timeout after {dt:.2f} s: {e}")
                t0 = time.time()
                try:
                    t_b.wait()
                    ok &= check(False, "a call of a dead group completed")
                except msm_amd.MsmAmdError as e:
                    ok &= check(e.code == msm_amd.EPEER and "dead" in str(e) and time.time() - t0 < 1.0, f"the ticket behind it: {e}")
                try:
                    C.multiexp_dist(group, bases, scalars, 32, n)
                    ok &= check(False, "a new call of a dead group was accepted")
                except msm_amd.MsmAmdError as e:
                    ok &= check(e.code == msm_amd.EPEER and "dead" in str(e), f"a new call: {e}")
                ok &= check(eng.dist_comm_count()[1], "dist_comm_count reports the group dead")
                ok &= check(C.g1m_multiexpAffine(bases, scalars, 32, n) == exp, "plain calls still work on the context of a dead group")
            barrier("after-timeout")
            eng.dist_destroy()
            join(eng, "b")
            r2 = C.multiexp_dist(group, bases, scalars, 32, n)
            ok &= check(r2 == r1, "the group after destroy + init")
            digest = r2.hex()[:24]
        else:
            raise SystemExit(f"unknown scenario {scenario}")
    finally:
        eng.close()
    print(json.dumps({"rank": rank, "ok": bool(ok) and not any(x.startswith("FAILED") for x in notes), "digest": digest, "notes": notes}), flush=True)


if __name__ == "__main__":
    main()

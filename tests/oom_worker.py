"""Child process of tests/test_gpu_round4.py::test_a_nearly_full_device_gives_enomem_or_the_right_bytes: the device filled to a
few hundred, then a few dozen megabytes free under a live context, and the calls that used to END THE PROCESS there (DESIGN.md section 7:
the HSA runtime aborts when it cannot allocate a queue's scratch at a dispatch).  Every call must return the right bytes or
MSM_AMD_ENOMEM; the process must reach its last line.  Prints one JSON line."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zprize-wasm-msm_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

import golden_util as gu  # noqa: E402
import msm_amd  # noqa: E402
from oracle import oracle as orcm  # noqa: E402


def free_mb(hip):
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value >> 20


def fill(eng, hip, leave_mb):
    """allocate until about leave_mb are free; returns the buffers (kept alive by the caller)"""
    held = []
    for chunk_mb in (16384, 1024, 64, 8):
        while free_mb(hip) > leave_mb + chunk_mb:
            try:
                held.append(eng.alloc(chunk_mb << 20))
            except msm_amd.MsmAmdError:
                break
    return held


def main():
    notes, ok = [], True
    orcm.build()
    eng = msm_amd.Engine(0)            # (reserves the MSM path's scratch on every stream: msm_amd_create)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]

    def attempt(what, fn, exp):
        nonlocal ok
        try:
            r = fn()
            good = r == exp
            notes.append(f"{what}: {'right bytes' if good else 'WRONG BYTES'} ({free_mb(hip)} MB free)")
            ok &= good
        except msm_amd.MsmAmdError as e:
            good = e.code == msm_amd.ENOMEM
            notes.append(f"{what}: error {e.code}{'' if good else ' (not ENOMEM): ' + str(e)} ({free_mb(hip)} MB free)")
            ok &= good

    try:
        cases = {}
        for curve, group, n in (("bls12381", 2, 3000), ("bls12381", 1, 20000), ("bn128", 2, 2000)):
            bases, uniform = gu.random_case(curve, group, f"gpu:oom:{curve}:{group}", n)
            # skewed: a third of the scalars equal -> long buckets, queued segments, the tail-combine kernels do real work
            sc = bytearray(uniform)
            for i in range(0, n, 3):
                sc[32 * i:32 * i + 32] = uniform[:32]
            sc = bytes(sc)
            O = orcm.Oracle(curve, group)
            cases[(curve, group)] = (bases, sc, n, O.normalize(O.multiexp_affine(bases, sc, 32, n)))
        held = fill(eng, hip, 220)
        notes.append(f"device filled: {free_mb(hip)} MB free")
        for (curve, group), (bases, sc, n, exp) in cases.items():
            C = eng.curve(curve)
            f = C.g1m_multiexpAffine if group == 1 else C.g2m_multiexpAffine
            attempt(f"first {curve} G{group} call with queued segments", lambda: f(bases, sc, 32, n), exp)
            attempt(f"second {curve} G{group} call", lambda: f(bases, sc, 32, n), exp)
        # entry points whose kernels need far more scratch than the context reserved: refused cleanly, never dispatched
        C = eng.curve("bls12381")
        b2, s2, n2, e2 = cases[("bls12381", 2)]
        try:
            flags = C.batch_in_group_affine(2, b2[:192 * 64]) if hasattr(C, "batch_in_group_affine") else None
            notes.append(f"batch_in_group_affine: ran ({free_mb(hip)} MB free)" if flags is not None else "batch_in_group_affine: no binding")
        except msm_amd.MsmAmdError as e:
            notes.append(f"batch_in_group_affine: error {e.code}")
            ok &= e.code == msm_amd.ENOMEM
        try:
            pb = C.prepare_bases(2, b2, n2)
            attempt("prepared table built on the full device", lambda: pb.multiexp(s2), e2)
            pb.release()
        except msm_amd.MsmAmdError as e:
            notes.append(f"prepare_bases: error {e.code} ({free_mb(hip)} MB free)")
            ok &= e.code == msm_amd.ENOMEM
        held += fill(eng, hip, 24)
        notes.append(f"device filled further: {free_mb(hip)} MB free")
        for (curve, group), (bases, sc, n, exp) in cases.items():
            C = eng.curve(curve)
            f = C.g1m_multiexpAffine if group == 1 else C.g2m_multiexpAffine
            attempt(f"{curve} G{group} call on the full device", lambda: f(bases, sc, 32, n), exp)
        # a second context cannot reserve its scratch any more: the create fails cleanly
        try:
            e2nd = msm_amd.Engine(0)
            e2nd.close()
            notes.append("a second context was created (enough memory for its scratch after all)")
        except msm_amd.MsmAmdError as e:
            notes.append(f"second context refused: {e}")
        for h in held:
            h.free()
        b, sc, n, exp = cases[("bls12381", 1)]
        attempt("after the memory came back", lambda: eng.curve("bls12381").g1m_multiexpAffine(b, sc, 32, n), exp)
    finally:
        eng.close()
    print(json.dumps({"ok": bool(ok), "notes": notes}), flush=True)


if __name__ == "__main__":
    main()

import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "zprize-wasm-msm_amd/python"); sys.path.insert(0, ".")
import golden_util as gu, msm_amd
eng = msm_amd.Engine(0)
curve, group = sys.argv[1], int(sys.argv[2])
C = eng.curve(curve)
from oracle import oracle as orc
O = orc.Oracle(curve, group)
zero = O.zero()
bad = 0
for rec in gu.golden(curve)["groups"][f"g{group}"]["msm"]:
    bases, scalars = gu.msm_inputs(curve, group, rec)
    affine = rec["fn"] == "multiexpAffine"
    fn = getattr(C, f"g{group}m_multiexp" + ("Affine" if affine else ""))
    t = time.time()
    out = fn(bases, scalars, rec["scalar_size"], rec["n"])
    ok = out == gu.canon(curve, group, bytes.fromhex(rec["out_norm"]), zero)
    bad += not ok
    print(rec["id"], rec["n"], "ok" if ok else "MISMATCH", "%.3fs" % (time.time() - t), flush=True)
print("ALL DONE mismatches:", bad, flush=True)

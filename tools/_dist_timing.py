import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import numpy as np
import msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
n = 1 << 16
pts = C.gen_points(1, 5, n)
sc = eng.alloc(n * 32).upload(np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8).tobytes())
eng.dist_init(0, 1, msm_amd.dist_unique_id())
for name, f in (("shard", lambda: C.multiexp_shard(1, pts, sc, 32, n, 0, 1)), ("dist", lambda: C.multiexp_dist(1, pts, sc, 32, n))):
    for _ in range(3): f()
    t = time.perf_counter()
    for _ in range(30): f()
    print(name, "%.3f ms per call" % ((time.perf_counter() - t) / 30 * 1e3), flush=True)
eng.close()

#!/usr/bin/env python3
"""kernel statistics of a call whose scalars are 90 % ones (one hot bucket of 0.9 n entries): run under rocprofv3 --kernel-trace --stats"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
n = 1 << 20
pts = C.gen_points(1, 0x5151, n)
g = torch.Generator(device="cuda"); g.manual_seed(11)
s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); s[:, 31] &= 0x3F
r = torch.rand(n, device="cuda", generator=g)
one = r < float(os.environ.get("P_ONE", "0.9"))
s[one] = 0; s[one, 0] = 1
s = s.contiguous()
torch.cuda.synchronize()
for _ in range(12):
    C._multiexp(1, 0, pts, s, 32, n)
print(eng.timings(), eng.geometry())

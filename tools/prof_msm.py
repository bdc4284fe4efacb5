#!/usr/bin/env python3
"""one MSM of 2^lg terms, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_msm.py LG`"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpc_bulletproof_amd as mb   # noqa: E402
import oracle_lib as o            # noqa: E402
lg = int(sys.argv[1])
n = 1 << lg
base = o.gens("G", 4096)
pts = (base * ((n + 4095) // 4096))[:64 * n]
sc = o.random_scalars(lg, n)
gpu = mb.BpGpu(0)
for _ in range(3):
    gpu.msm(sc, pts)

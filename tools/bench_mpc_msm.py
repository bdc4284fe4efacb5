#!/usr/bin/env python3
"""BASELINE.json configs[4] (two-party r1cs_mpc prover, per-party MSMs on one GPU each): latency of the local work
of StarkPoint::msm_authenticated_iter -- three MSMs (shares, MACs, public modifiers) over one point vector -- at the
sizes the reference's integration tests produce (SimpleCircuit: N <= 3; k = 8 shuffle: N <= 29) and
a few larger ones, against the CPU oracle doing the same three MSMs on one thread.  A latency test, not throughput."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpc_bulletproof_amd as mb   # noqa: E402
import oracle_lib as o            # noqa: E402

gpu = mb.BpGpu(0)
base = o.gens("G", 2048) + o.gens("H", 2048)
for n in (3, 29, 255, 1025, 4096):
    pts = base[:64 * n]
    sc = o.random_scalars(500 + n, 3 * n)
    r = gpu.msm_shared(3, n, sc, pts)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        r = gpu.msm_shared(3, n, sc, pts)
    tg = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    rc = o.msm_batch(sc, pts * 3, 3, n)
    tc = time.perf_counter() - t0
    assert rc == r
    print(f"N={n:5d}: gpu {tg * 1e3:7.2f} ms per authenticated MSM (3 sets, host buffers)   cpu-oracle 1T {tc * 1e3:8.2f} ms   x{tc / tg:6.1f}")

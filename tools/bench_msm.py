#!/usr/bin/env python3
"""MSM micro-benchmark (SURVEY 8d 'Micro'): GPU bpgpu_msm vs the CPU oracle's Pippenger, host buffers in/out."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpc_bulletproof_amd as mb   # noqa: E402
import oracle_lib as o            # noqa: E402

gpu = mb.BpGpu(0)
base = o.gens("G", 4096)
for lg in [int(x) for x in os.environ.get("BENCH_MSM_LOG2", "7,10,12,14,17,20").split(",")]:
    n = 1 << lg
    pts = (base * ((n + 4095) // 4096))[:64 * n]
    sc = o.random_scalars(lg, n)
    gpu.msm(sc, pts)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        r = gpu.msm(sc, pts)
    tg = (time.perf_counter() - t0) / reps
    tc = None
    if lg <= 14:
        t0 = time.perf_counter()
        rc = o.msm(sc, pts, 2)
        tc = time.perf_counter() - t0
        assert rc == r
    # operands resident in HBM (what a caller that keeps its points on the device sees)
    d_sc, d_pts, d_out = gpu.to_device(sc), gpu.to_device(pts), gpu.malloc(64)
    gpu.msm_batch_dev(1, n, d_sc, d_pts, d_out)
    gpu.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        gpu.msm_batch_dev(1, n, d_sc, d_pts, d_out)
    gpu.sync()
    td = (time.perf_counter() - t0) / reps
    assert gpu.download(d_out, 64) == r and gpu.input_flag() == 0
    for d in (d_sc, d_pts, d_out):
        gpu.free(d)
    print(f"n=2^{lg:<2d} resident {td * 1e3:8.2f} ms ({n / td / 1e6:8.3f} Mterm/s, {96 * n / td / 1e9:6.2f} GB/s algorithmic) | "
          f"host buffers {tg * 1e3:8.2f} ms ({n / tg / 1e6:7.3f} Mterm/s incl. copies + validation)"
          + (f"   cpu-oracle 1T {tc * 1e3:9.1f} ms  x{tc / tg:6.1f}" if tc else ""))

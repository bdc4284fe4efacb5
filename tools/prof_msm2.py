#!/usr/bin/env python3
"""Resident MSMs of 2^LG terms for the profiler (tools/prof_msm_all.sh): points k_i * G made by the product's own generator
multiplication, seeded scalars; REPS calls on device-resident operands after one warm-up.  Usage: prof_msm2.py LG [REPS]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_bulletproof_amd as mb   # noqa: E402

N = 0x0800000000000010ffffffffffffffffb781126dcae7b2321e66a241adc64d2f
lg = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << lg
rnd = random.Random(1000 + lg)
gpu = mb.BpGpu(0)
base_k = b"".join(rnd.randrange(1, N).to_bytes(32, "little") for _ in range(4096))
base = gpu.generator_mul(base_k)
pts = (base * ((n + 4095) // 4096))[:64 * n]
sc = b"".join(rnd.randrange(N).to_bytes(32, "little") for _ in range(n))
d_sc, d_pts, d_out = gpu.to_device(sc), gpu.to_device(pts), gpu.malloc(64)
gpu.msm_batch_dev(1, n, d_sc, d_pts, d_out)
gpu.sync()
time.sleep(0.3)            # the trace is cut at this gap
t0 = time.perf_counter()
for _ in range(reps):
    gpu.msm_batch_dev(1, n, d_sc, d_pts, d_out)
gpu.sync()
dt = (time.perf_counter() - t0) / reps
assert gpu.input_flag() == 0
print(f"n=2^{lg}: {dt * 1e3:.3f} ms per resident MSM = {n / dt / 1e6:.1f} Mterm/s = {96 * n / dt / 1e9:.2f} GB/s algorithmic")

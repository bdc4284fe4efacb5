#!/usr/bin/env python3
"""Does libbpgpu.so's load-time default of GPU_MAX_HW_QUEUES reach the HIP runtime?  One process: import torch, then the library,
then ONE bpgpu_r1cs_verify_stream_dev call over 64 x 1024 proofs (bench.py's workload cache), five times.  Run it plain and with
GPU_MAX_HW_QUEUES=4 exported: the rates differ by ~3x when the default is effective.  Usage: queue_probe.py <workload-cache>.1024"""
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
before = os.environ.get("GPU_MAX_HW_QUEUES")
import torch  # noqa: E402,F401
import mpc_bulletproof_amd as mb  # noqa: E402

wl = pickle.load(open(sys.argv[1], "rb"))
n1, n2, k, m = wl["dims"]
gpu = mb.BpGpu(0)
circ = gpu.circuit_create(*wl["csr"], n1 + n2, m)
gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], int(os.environ.get("WINDOW_BITS", "16")))
reps, nb = 64, 1024
dp, ds, dc = gpu.to_device(wl["points"] * reps), gpu.to_device(wl["scalars"] * reps), gpu.to_device(wl["challenges"] * reps)
dok = gpu.malloc(4 * nb * reps)
ts = []
for _ in range(6):
    gpu.sync()
    t0 = time.perf_counter()
    gpu.r1cs_verify_stream_dev(gens, circ, nb * reps, n1, k, dp, ds, dc, dok)
    gpu.sync()
    ts.append(time.perf_counter() - t0)
assert gpu.download(dok, 4 * nb * reps) == (1).to_bytes(4, "little") * (nb * reps)
ts = sorted(ts[1:])
print(f"GPU_MAX_HW_QUEUES before the imports: {before!r}, after: {os.environ.get('GPU_MAX_HW_QUEUES')!r}; "
      f"one call over {nb * reps} proofs: {ts[len(ts) // 2] * 1e3:.2f} ms = {nb * reps / ts[len(ts) // 2] / 1e6:.2f} M verifications/s")

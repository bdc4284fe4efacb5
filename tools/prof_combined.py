#!/usr/bin/env python3
"""Drive the combined batch check (bpgpu_r1cs_verify_combined_dev) of the bench workload: solo steps (one context, sync after
each) then pipelined bursts.  Usage: prof_combined.py WORKLOAD.pkl [inflight] -- run it under rocprofv3 --kernel-trace for the
per-kernel picture (tools/kernel_stats.py / tools/timeline.py read the database)."""
import os
import pickle
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ.setdefault("BPGPU_SINGLE_STREAM", "1")
import torch                       # noqa: E402
import mpc_bulletproof_amd as mb   # noqa: E402

wl = pickle.load(open(sys.argv[1], "rb"))
nb = len(wl["scalars"]) // 160
n1, n2, k, m = wl["dims"]
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctxs = [mb.BpGpu(0) for _ in range(inflight)]
gpu = ctxs[0]
circ = gpu.circuit_create(*wl["csr"], n1 + n2, m)
gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], int(os.environ.get("BPGPU_WINDOW_BITS", "20")))
d_pts, d_sc, d_ch = gpu.to_device(wl["points"]), gpu.to_device(wl["scalars"]), gpu.to_device(wl["challenges"])
rnd = random.Random(0xC0B1)
d_rho = gpu.to_device(b"".join(rnd.getrandbits(250).to_bytes(32, "little") for _ in range(nb)))
d_parts = [gpu.malloc(64) for _ in ctxs]
cnt = [0]


def step():
    i = cnt[0] % len(ctxs)
    cnt[0] += 1
    ctxs[i].r1cs_verify_combined_dev(gens, circ, nb, n1, k, d_pts, d_sc, d_ch, d_rho, d_parts[i])


for _ in range(3 * inflight):
    step()
torch.cuda.synchronize()
assert all(c.download(d, 64) == bytes(64) for c, d in zip(ctxs, d_parts))
for K in [int(x) for x in os.environ.get("BURST_KS", "1,1,1,20,20,64,64,512,512").split(",")]:
    cnt[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K:5d}: submit {1e3 * (t1 - t0):7.2f} ms, total {1e3 * (t2 - t0):7.2f} ms = {1e3 * (t2 - t0) / K:6.3f} ms/step = {nb * K / (t2 - t0) / 1e6:5.2f} M/s")

#!/usr/bin/env python3
"""Occupancy picture of ONE burst of K verification steps on an idle GPU, from the library's own HIP-event pairs around every launch
(bpgpu_profile_intervals; rocprofv3's kernel trace serialises the queues and shows 3-4 kernels at a time where 20 run).  Per time bin:
how many launches of each kernel of the chain are on the chip, the waves they bring per SIMD and the issue demand of those waves
(waves x the kernel's solo issuing share, profiles/pmc_constants.json) -- a bin whose demand is below 1 per SIMD cannot be at the peak.
Usage: burst_gantt_ev.py workload.pkl [inflight] [K] [bin_us]"""
import os
import pickle
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
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
binus = float(sys.argv[4]) if len(sys.argv) > 4 else 100.0
ctxs = [mb.BpGpu(0) for _ in range(inflight)]
gpu = ctxs[0]
circ = gpu.circuit_create(*wl["csr"], n1 + n2, m)
gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], 20)
d_pts, d_sc, d_ch = gpu.to_device(wl["points"]), gpu.to_device(wl["scalars"]), gpu.to_device(wl["challenges"])
d_oks = [gpu.malloc(4 * nb) for _ in ctxs]
cnt = [0]
if os.environ.get("BURST_LATENCY_MODE"):
    for c in ctxs:
        c.set_latency_mode(True)


def step():
    i = cnt[0] % len(ctxs)
    cnt[0] += 1
    ctxs[i].r1cs_verify_batch_dev(gens, circ, nb, n1, k, d_pts, d_sc, d_ch, d_oks[i])


for _ in range(2000):
    step()
torch.cuda.synchronize()
# the same burst, untimed by events, for reference
for _ in range(3):
    cnt[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    print(f"plain burst K={K}: {1e3 * (time.perf_counter() - t0):.3f} ms")
for c in ctxs:
    c.profile_enable(True)
WAVES = {"verify_front": 112, "verify_scalars": 1024, "verify_windows": 1024, "verify_groups": 128, "verify_back": 320, "verify_verdict": 16}
ISSUE = {"verify_front": 0.72, "verify_scalars": 0.43, "verify_windows": 0.74, "verify_groups": 0.83, "verify_back": 0.71, "verify_verdict": 0.41}
for rep in range(3):
    cnt[0] = 0
    torch.cuda.synchronize()
    ep = gpu.profile_epoch()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0)
    iv = []
    for c in ctxs:
        iv += c.profile_intervals(ep)
print(f"event-timed burst K={K}: {wall:.3f} ms, {len(iv)} launches")
names = list(WAVES)
t_end = max(b for _, a, b in iv)
print("   t(us)  " + " ".join(f"{n[7:]:>8s}" for n in names) + "   waves/SIMD  issue demand/SIMD")
nbin = int(t_end * 1e3 / binus) + 1
for i in range(nbin):
    lo, hi = i * binus / 1e3, (i + 1) * binus / 1e3
    c = {n: 0.0 for n in names}
    for n, a, b in iv:
        if n in c:
            ov = min(b, hi) - max(a, lo)
            if ov > 0:
                c[n] += ov / (hi - lo)
    w = sum(c[n] * WAVES[n] for n in names) / 1024
    d = sum(c[n] * WAVES[n] * ISSUE[n] for n in names) / 1024
    print(f"{i * binus:8.0f}  " + " ".join(f"{c[n]:8.1f}" for n in names) + f"   {w:8.2f}  {d:8.2f}")
for n in names:
    d = [(b - a) * 1e3 for x, a, b in iv if x == n]
    if d:
        st = [a * 1e3 for x, a, b in iv if x == n]
        en = [b * 1e3 for x, a, b in iv if x == n]
        print(f"{n[7:]:8s} x{len(d):3d}: duration min {min(d):7.1f} avg {sum(d) / len(d):7.1f} max {max(d):7.1f} us; "
              f"first start {min(st):7.1f} last start {max(st):7.1f} first end {min(en):7.1f} last end {max(en):7.1f}")

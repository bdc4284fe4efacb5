#!/usr/bin/env python3
"""Phases of the verification workload with unix-time stamps for tools/power_phases.sh (rocm-smi sampled beside it): idle | 5 s of continuous
steps | 5 s of 20-step bursts | idle.  Usage: power_phases.py workload.pkl"""
import os, pickle, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ.setdefault("BPGPU_SINGLE_STREAM", "1")
import torch
import mpc_bulletproof_amd as mb
wl = pickle.load(open(sys.argv[1], "rb"))
nb = len(wl["scalars"]) // 160
n1, n2, k, m = wl["dims"]
ctxs = [mb.BpGpu(0) for _ in range(20)]
gpu = ctxs[0]
circ = gpu.circuit_create(*wl["csr"], n1 + n2, m)
gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], 20)
d_pts, d_sc, d_ch = gpu.to_device(wl["points"]), gpu.to_device(wl["scalars"]), gpu.to_device(wl["challenges"])
d_oks = [gpu.malloc(4 * nb) for _ in ctxs]
cnt = [0]
def step():
    i = cnt[0] % 20; cnt[0] += 1
    ctxs[i].r1cs_verify_batch_dev(gens, circ, nb, n1, k, d_pts, d_sc, d_ch, d_oks[i])
def stamp(w): print(f"{time.time():.2f} {w}", flush=True)
for _ in range(200): step()
torch.cuda.synchronize()
stamp("idle"); time.sleep(3)
stamp("continuous"); t = time.time(); n = 0
while time.time() - t < 5:
    for _ in range(40): step()
    n += 40
    if n % 400 == 0: torch.cuda.synchronize()
torch.cuda.synchronize(); stamp(f"continuous-end {n * nb / (time.time() - t) / 1e6:.2f}")
stamp("bursts"); t = time.time(); r = []
while time.time() - t < 5:
    cnt[0] = 0; torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); r.append(nb * 20 / (time.perf_counter() - t0) / 1e6)
r.sort(); stamp(f"bursts-end {r[len(r)//2]:.2f}")
stamp("idle"); time.sleep(3); stamp("end")

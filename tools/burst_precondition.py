#!/usr/bin/env python3
"""What state does a 20-step burst want the GPU in?  After a long continuous run: a burst at once, after a pause of X ms, after n bursts."""
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
def burst(K=20):
    cnt[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize()
    return nb * K / (time.perf_counter() - t0) / 1e6
def hot(sec):
    t = time.perf_counter()
    while time.perf_counter() - t < sec:
        for _ in range(40): step()
    torch.cuda.synchronize()
for _ in range(2000): step()
torch.cuda.synchronize()
for pause in (0, 2, 10, 50, 200, 1000):
    r = []
    for rep in range(4):
        hot(0.3)
        time.sleep(pause / 1e3)
        r.append(burst())
    print(f"0.3 s of continuous steps, pause {pause:5d} ms, one burst: " + " ".join(f"{x:.2f}" for x in r), flush=True)
for nbursts in (1, 2, 4, 8):
    r = []
    for rep in range(4):
        hot(0.3)
        for _ in range(nbursts): burst()
        r.append(burst())
    print(f"0.3 s of continuous steps, {nbursts} untimed bursts, one burst: " + " ".join(f"{x:.2f}" for x in r), flush=True)
for sec in (0.0, 0.02, 0.1):
    r = []
    for rep in range(4):
        time.sleep(1.0)
        if sec: hot(sec)
        r.append(burst())
    print(f"1 s idle, {sec} s of continuous steps, one burst: " + " ".join(f"{x:.2f}" for x in r), flush=True)

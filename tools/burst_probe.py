#!/usr/bin/env python3
"""How long do bursts of K verification steps take after a device synchronisation?  (explains why short bench runs
under-report the steady state: see DESIGN.md 6)  Uses bench.py's workload cache: run bench.py --workload-cache X first."""
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
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 16
prof_on = len(sys.argv) > 3 and sys.argv[3] == "prof"
ctxs = [mb.BpGpu(0) for _ in range(inflight)]
gpu = ctxs[0]
circ = gpu.circuit_create(*wl["csr"], n1 + n2, m)
gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], 20)
d_pts, d_sc, d_ch = gpu.to_device(wl["points"]), gpu.to_device(wl["scalars"]), gpu.to_device(wl["challenges"])
d_oks = [gpu.malloc(4 * nb) for _ in ctxs]
cnt = [0]


def step():
    i = cnt[0] % len(ctxs)
    cnt[0] += 1
    ctxs[i].r1cs_verify_batch_dev(gens, circ, nb, n1, k, d_pts, d_sc, d_ch, d_oks[i])


for c in ctxs:
    c.profile_enable(prof_on)
    if os.environ.get("BURST_LATENCY_MODE"):   # the un-pipelined caller's setting (bpgpu_set_latency_mode)
        c.set_latency_mode(True)
for _ in range(3000):
    step()
torch.cuda.synchronize()
NT = int(os.environ.get("BURST_THREADS", "0"))    # experiment: the K steps submitted from NT host threads (step j on context j % inflight)
if NT:
    import threading
    go = [threading.Barrier(NT + 1), threading.Barrier(NT + 1)]
    todo = [0]

    def worker(t):
        while True:
            go[0].wait()
            if todo[0] < 0:
                return
            for j in range(t, todo[0], NT):
                ctxs[j % len(ctxs)].r1cs_verify_batch_dev(gens, circ, nb, n1, k, d_pts, d_sc, d_ch, d_oks[j % len(ctxs)])
            go[1].wait()

    ths = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(NT)]
    for t in ths:
        t.start()
for K in [int(x) for x in os.environ.get("BURST_KS", "1,2,4,8,16,32,64,64,256,256,1024,1024,4096").split(",")]:
    cnt[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if NT:
        todo[0] = K
        go[0].wait()
        go[1].wait()
    else:
        for _ in range(K):
            step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K:5d}: submit {1e3 * (t1 - t0):7.2f} ms, total {1e3 * (t2 - t0):7.2f} ms = {1e3 * (t2 - t0) / K:6.3f} ms/step = {nb * K / (t2 - t0) / 1e6:5.2f} M/s")
    if prof_on:
        for c in ctxs:
            c.profile_read()

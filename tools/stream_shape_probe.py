#!/usr/bin/env python3
"""One bpgpu_r1cs_verify_stream_dev call over N proofs (a burst on an idle GPU) under different (batch, lanes) shapes of the lane ring.
Usage: stream_shape_probe.py <workload-cache>.1024 [N=20480]"""
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                       # noqa: E402,F401
import mpc_bulletproof_amd as mb   # noqa: E402

cache = sys.argv[1]
wl = pickle.load(open(cache, "rb"))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20480
nb = len(wl["scalars"]) // 160
n1, n2, k, m = wl["dims"]
reps = (N + nb - 1) // nb


def many(ext, one):
    with open(cache + ext, "rb") as f:
        buf = f.read(reps * len(one))
    return (buf * (reps * len(one) // len(buf) + 1))[:reps * len(one)]


gpu = mb.BpGpu(0)
circ = gpu.circuit_create(*wl["csr"], n1 + n2, m)
gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], 20)
d_p, d_s, d_c = gpu.to_device(many(".pts", wl["points"])), gpu.to_device(many(".sc", wl["scalars"])), gpu.to_device(many(".ch", wl["challenges"]))
d_o = gpu.malloc(4 * nb * reps)
shapes = [(1024, 20), (2048, 10), (2048, 20), (4096, 5), (4096, 8), (5120, 4), (10240, 2), (512, 20), (512, 40), (1024, 12), (1024, 16)]
for batch, lanes in shapes:
    gpu.set_option("stream_batch", batch)
    gpu.set_option("stream_lanes", lanes)
    ts = []
    for r in range(8):
        gpu.sync()
        t0 = time.perf_counter()
        gpu.r1cs_verify_stream_dev(gens, circ, nb * reps, n1, k, d_p, d_s, d_c, d_o)
        gpu.sync()
        ts.append(time.perf_counter() - t0)
    assert gpu.download(d_o, 4 * nb * reps) == (1).to_bytes(4, "little") * (nb * reps)
    ts = sorted(ts[2:])
    print(f"N={nb * reps} batch={batch:6d} lanes={lanes:3d}: median {1e3 * ts[len(ts) // 2]:7.3f} ms = {nb * reps / ts[len(ts) // 2] / 1e6:5.2f} M/s   best {1e3 * ts[0]:7.3f} ms", flush=True)

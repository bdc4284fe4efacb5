#!/usr/bin/env python3
"""Where does an H2D-inclusive step lose time?  (a) one pinned 2 MB upload + sync; (b) bench.py's h2d_inclusive leg on 20 contexts;
(c) the same after another 20 streams (the lanes of a stream call) exist in the process.  Usage: h2d_probe.py <workload-cache>.1024"""
import ctypes as C
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BPGPU_SINGLE_STREAM", "1")
import torch  # noqa: E402
import mpc_bulletproof_amd as mb  # noqa: E402

wl = pickle.load(open(sys.argv[1], "rb"))
n1, n2, k, m = wl["dims"]
pts, sc, ch = wl["points"], wl["scalars"], wl["challenges"]
nb = 1024
ctxs = [mb.BpGpu(0) for _ in range(20)]
gpu = ctxs[0]
circ = gpu.circuit_create(*wl["csr"], n1 + n2, m)
gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], 16)
packed = pts + sc + ch
h_in = mb.lib.host_alloc(len(packed), packed)
d_in = [c.malloc(len(packed)) for c in ctxs]
d_ok = [c.malloc(4 * nb) for c in ctxs]
for _ in range(3):
    gpu.sync()
    t0 = time.perf_counter()
    gpu.upload_async(d_in[0], h_in, len(packed))
    gpu.sync()
    print(f"(a) pinned upload of {len(packed)} B + sync: {(time.perf_counter() - t0) * 1e6:.0f} us")


def hstep(i):
    j = i % len(ctxs)
    c, dp = ctxs[j], d_in[j]
    c.upload_async(dp, h_in, len(packed))
    c.r1cs_verify_batch_dev(gens, circ, nb, n1, k, dp, C.c_void_p(dp.value + len(pts)), C.c_void_p(dp.value + len(pts) + len(sc)), d_ok[j])


def leg(tag, steps=200):
    for i in range(40):
        hstep(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        hstep(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{tag}: {dt / steps * 1e3:.3f} ms/step = {nb * steps / dt / 1e6:.2f} M/s")


leg("(b) 20 contexts")
d3 = [gpu.to_device(x * 20) for x in (pts, sc, ch)]
dok = gpu.malloc(4 * nb * 20)
gpu.r1cs_verify_stream_dev(gens, circ, nb * 20, n1, k, d3[0], d3[1], d3[2], dok)
gpu.sync()
leg("(c) 20 contexts + 20 lanes alive")

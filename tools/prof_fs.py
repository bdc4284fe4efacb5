#!/usr/bin/env python3
"""The device-transcript verification leg alone (bench.py's `with_device_transcript`), for
`rocprofv3 --kernel-trace -- python3 tools/prof_fs.py WORKLOAD.pkl [steps] [inflight]` (workload: bench.py --workload-cache)."""
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("BPGPU_SINGLE_STREAM", "1")
import torch                       # noqa: E402
import mpc_bulletproof_amd as mb   # noqa: E402

wl = pickle.load(open(sys.argv[1], "rb"))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 512
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 16
nb = len(wl["scalars"]) // 160
n1, n2, k, m = wl["dims"]
ctxs = [mb.BpGpu(0) for _ in range(inflight)]
gpu = ctxs[0]
circ = gpu.circuit_create(*wl["csr"], n1 + n2, m)
gens = gpu.gens_create(wl["G"], wl["H"], wl["B"], wl["B"], 20)
d_pts, d_sc = gpu.to_device(wl["points"]), gpu.to_device(wl["scalars"])
d_init = gpu.to_device(wl["init_state"] * nb)
d_oks = [gpu.malloc(4 * nb) for _ in ctxs]


def fstep(i):
    j = i % len(ctxs)
    ctxs[j].r1cs_verify_batch_fs_dev(gens, circ, nb, n1, k, d_init, d_pts, d_sc, d_oks[j])


for i in range(4 * len(ctxs)):
    fstep(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    fstep(i)
t_submit = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for c, d in zip(ctxs, d_oks):
    assert c.download(d, 4 * nb) == (1).to_bytes(4, "little") * nb
print(f"{steps} steps, {inflight} in flight: {dt / steps * 1e3:.4f} ms/step = {nb * steps / dt / 1e6:.3f} M verifications/s (host submission {t_submit / steps * 1e3:.4f} ms/step)")

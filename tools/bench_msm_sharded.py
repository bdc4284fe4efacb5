#!/usr/bin/env python3
"""One large MSM split by term range across the ranks of a node (BASELINE configs[3]'s 98 347-term
mega_check, or any size): each rank runs bpgpu_msm over its contiguous slice on its own GPU, the <= 8 partial points
are all-gathered (RCCL) and added.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_msm_sharded.py [LOG2N]
BPGPU_BENCH_REHEARSAL=1: all ranks share device 0 and use gloo (control-flow rehearsal on a one-GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                              # noqa: E402
import torch.distributed as dist          # noqa: E402
import mpc_bulletproof_amd as mb          # noqa: E402
from mpc_bulletproof_amd import sharding  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n = 98347 if lg == 0 else 1 << lg
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
rehearsal = bool(os.environ.get("BPGPU_BENCH_REHEARSAL"))
dev = 0 if rehearsal else local
torch.cuda.set_device(dev)
if world > 1:
    dist.init_process_group("gloo" if rehearsal else "nccl", **({} if rehearsal else {"device_id": torch.device("cuda", dev)}))
gpu = mb.BpGpu(dev)
# synthetic operands, identical on every rank: points k_i * G from the generator table kernel, scalars from a LCG
import random
rnd = random.Random(1234)
ks = b"".join(rnd.getrandbits(250).to_bytes(32, "little") for _ in range(min(n, 4096)))
base = gpu.generator_mul(ks)
pts = (base * ((n + 4095) // 4096))[:64 * n]
sc = b"".join(rnd.getrandbits(250).to_bytes(32, "little") for _ in range(n))
d_sc, d_pts, d_out = gpu.to_device(sc), gpu.to_device(pts), gpu.malloc(64)   # resident operands (every rank holds all terms)
full = gpu.msm(sc, pts) if rank == 0 else None
for rep in range(3):
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:
        r = sharding.sharded_msm_dev(gpu, d_sc, d_pts, n, d_out)
    else:
        gpu.msm_batch_dev(1, n, d_sc, d_pts, d_out)
        r = gpu.download(d_out, 64)
    dt = time.perf_counter() - t0
    if world > 1:
        dt = sharding.max_over_ranks(dt)
if rank == 0:
    assert r == full, "sharded result differs from the single-GPU MSM"
    print(f"n = {n} terms over {world} rank(s): {dt * 1e3:.2f} ms per MSM (operands resident in HBM, slices addressed in place, partial-point all-gather + one point-sum launch), "
          f"{n / dt / 1e6:.1f} M terms/s; result equals the single-GPU MSM")
if world > 1:
    dist.destroy_process_group()
gpu.close()

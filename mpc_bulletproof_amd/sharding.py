"""Multi-GPU plumbing for the one place the path shards: independent proofs.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  Proofs are independent units, so ranks own contiguous slices of the batch and the
data path needs NO collective; the only exchanges are (a) gathering per-rank accept bits / results for
the caller and (b) the latency-bound all-gather of <= 8 partial points (64 B each) when a single
combined batch check is wanted.  No torch types cross the C ABI: tensors are just byte carriers.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of `total` units owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def allgather_bytes(local: bytes, group=None):
    """Every rank contributes a byte string (lengths may differ); returns the list of all of them."""
    world = dist.get_world_size(group)
    dev = _device()
    n = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    mx = max(int(s.item()) for s in sizes)
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
    if local:
        buf[:len(local)] = torch.frombuffer(bytearray(local), dtype=torch.uint8).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return [bytes(o[:int(s.item())].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]


def max_over_ranks(x: float, group=None) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_accept_bits(local_ok, total, group=None):
    """local_ok: list of 0/1 for this rank's slice (shard_bounds order) -> full list on every rank."""
    parts = allgather_bytes(bytes(local_ok), group)
    out = []
    for p in parts:
        out.extend(p)
    assert len(out) == total
    return out


def combine_partial_points(local_point: bytes, points_sum, group=None) -> bytes:
    """All-gather one partial MSM result (64-byte affine point, zeros = identity) per rank and add them locally with
    `points_sum(points_bytes) -> 64 bytes` (BpGpu.points_sum: one launch; RCCL has no elliptic-curve reduction op)."""
    return points_sum(b"".join(allgather_bytes(local_point, group)))


def sharded_msm(scalars: bytes, points: bytes, msm_fn, points_sum, group=None) -> bytes:
    """One LARGE multi-scalar multiplication split by term range (the 98 347-term mega_check of
    the 2^14-shuffle): rank r computes the sum over its contiguous slice with `msm_fn(scalars, points) -> 64 B`
    (BpGpu.msm on the GPU box), the <= 8 partial points are all-gathered and added locally.  Every rank
    returns the same full result.  Operands arrive as host byte strings: see sharded_msm_dev for resident ones."""
    n = len(scalars) // 32
    assert len(points) == 64 * n
    lo, hi = shard_bounds(n, dist.get_rank(group), dist.get_world_size(group))
    part = msm_fn(scalars[32 * lo:32 * hi], points[64 * lo:64 * hi]) if hi > lo else bytes(64)
    return combine_partial_points(part, points_sum, group)


def sharded_msm_dev(gpu, d_scalars, d_points, n, d_out, group=None) -> bytes:
    """The same with the operands RESIDENT in this rank's HBM (every rank holds the full scalar / point vectors, e.g. the
    generators and the scalars its own scalar-assembly kernels wrote): the rank's slice is addressed in place
    (bpgpu_msm_batch_dev on a pointer offset), only the 64-byte partial crosses PCIe, and the partials are summed with one
    point-sum launch.  gpu: BpGpu; d_*: device pointers (c_void_p); d_out: 64 B of device memory.  Returns the full sum."""
    import ctypes as C
    lo, hi = shard_bounds(n, dist.get_rank(group), dist.get_world_size(group))
    if hi > lo:
        gpu.msm_batch_dev(1, hi - lo, C.c_void_p(d_scalars.value + 32 * lo), C.c_void_p(d_points.value + 64 * lo), d_out)
        part = gpu.download(d_out, 64)
    else:
        part = bytes(64)
    return combine_partial_points(part, gpu.points_sum, group)

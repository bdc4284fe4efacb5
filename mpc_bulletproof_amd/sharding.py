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


# ---- the all-gather a native RankGroup needs (host mirror: mpc_bulletproof::RankGroup; harness: bph_allgather_fn) -----------------
import ctypes as _C

ALLGATHER_FN = _C.CFUNCTYPE(None, _C.POINTER(_C.c_uint8), _C.c_size_t, _C.POINTER(_C.c_uint8), _C.c_void_p)


def allgather_callback(group=None):
    """A C callback `void f(const uint8_t *mine, size_t bytes, uint8_t *out, void *user)` over torch.distributed: every rank
    contributes `bytes` bytes, `out` receives world x bytes in rank order (backend "nccl" = RCCL over xGMI on a GPU node: the byte
    carrier is a device tensor; "gloo": a CPU tensor).  Keep the returned object alive while native code may call it."""
    def cb(mine, nbytes, out, user):
        world = dist.get_world_size(group)
        t = torch.frombuffer(bytearray(_C.string_at(mine, nbytes)), dtype=torch.uint8).to(_device())
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        data = b"".join(o.cpu().numpy().tobytes() for o in outs)
        _C.memmove(out, data, len(data))
    return ALLGATHER_FN(cb)


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


def sharded_ipp_create(gpu, transcript, n, w, B, Gf, Hf, G, H, a, b, window_bits=8, group=None):
    """InnerProductProof::create (inner_product_proof.rs:49-193) for ONE large proof with a, b, G, H dealt CYCLICALLY over the
    ranks (BASELINE configs[3]): rank r owns the indices i = r (mod ranks).  A fold pairs i with i + h, and
    ranks | h while h >= ranks, so every fold is local and each rank runs an ordinary resident-generator IPP session of length
    n / ranks on its own sub-vectors (its own fixed-base tables: 1 / ranks of the memory).  Per round the only exchange is the
    all-gather of the ranks' partial L and R (2 x 64 B each; the c_L Q terms add up with them) and one point-sum launch; the
    Fiat-Shamir transcript runs redundantly on every rank.  When a rank's session is down to one element, (a_r, b_r) and
    its folded generators (G'_r, H'_r) are all-gathered and the last log2(ranks) rounds run on every rank as the literal
    schedule (bpgpu_ipp_begin) over those `ranks` elements.  Same L, R, a, b bytes as the single-GPU proof.

    gpu: BpGpu.  transcript: object with append_message(label, bytes) and challenge_scalar(label) -> 32 LE bytes, already past
    innerproduct_domain_sep(n); every rank feeds its own copy the same bytes.  w: 32 B (Q = w * B); B: 64 B; Gf, Hf, a, b: n x 32 B;
    G, H: n x 64 B (full vectors; only this rank's residue class is uploaded).  Returns (L_list, R_list, a, b)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert n & (n - 1) == 0 and world & (world - 1) == 0 and world <= n
    sub = n // world

    def strided(buf, sz):
        return b"".join(buf[sz * i:sz * i + sz] for i in range(rank, n, world))

    Ls, Rs = [], []

    def challenge(L, R):
        transcript.append_message(b"L", L)
        transcript.append_message(b"R", R)
        u = transcript.challenge_scalar(b"u")
        Ls.append(L)
        Rs.append(R)
        return u

    gens = gpu.gens_create(strided(G, 64), strided(H, 64), B, B, window_bits)
    s = gpu.ipp_begin_gens(gens, 1, sub, w, strided(Gf, 32), strided(Hf, 32), strided(a, 32), strided(b, 32))
    try:
        while gpu.ipp_len(s) > 1:
            Lr, Rr = gpu.ipp_round(s, 1)
            parts = allgather_bytes(Lr + Rr, group)
            L = gpu.points_sum(b"".join(p[:64] for p in parts))
            R = gpu.points_sum(b"".join(p[64:] for p in parts))
            u = challenge(L, R)
            gpu.ipp_fold(s, u, gpu.batch_inverse(u))
        ar, br = gpu.ipp_finish(s, 1)
        Gr, Hr = gpu.ipp_folded_gens(s, 1)
    finally:
        gpu.ipp_destroy(s)
        gpu.gens_destroy(gens)
    if world == 1:
        return Ls, Rs, ar, br
    parts = allgather_bytes(ar + br + Gr + Hr, group)          # rank order = residue order = element order of the tail
    a_t, b_t = b"".join(p[:32] for p in parts), b"".join(p[32:64] for p in parts)
    G_t, H_t = b"".join(p[64:128] for p in parts), b"".join(p[128:192] for p in parts)
    one = (1).to_bytes(32, "little") * world
    Q = gpu.msm(w, B)
    t = gpu.ipp_begin(1, world, Q, one, one, G_t, H_t, True, a_t, b_t)
    try:
        while gpu.ipp_len(t) > 1:
            L, R = gpu.ipp_round(t, 1)
            u = challenge(L, R)
            gpu.ipp_fold(t, u, gpu.batch_inverse(u))
        a_f, b_f = gpu.ipp_finish(t, 1)
    finally:
        gpu.ipp_destroy(t)
    return Ls, Rs, a_f, b_f

"""world_size-2 and world_size-8 CPU (gloo) tests of the multi-GPU plumbing: proof-level sharding, result gathering and
the partial-point combine.  The per-rank compute is done by the CPU oracle here (no GPU in this
container); on the GPU box the same functions run over RCCL with the HIP path."""
import os
import socket
import sys

import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _points_sum(points):
    import oracle_lib as o
    acc = bytes(64)
    for i in range(0, len(points), 64):
        acc = o.point_add(acc, points[i:i + 64])
    return acc


def _worker(rank, world, port, n_bits, nb, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bp_helpers as bh
    import oracle_lib as o
    from mpc_bulletproof_amd import sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        recs, cap = bh.make_range_batch(n_bits, nb, tamper={1, nb - 2})   # same deterministic batch on every rank
        lo, hi = sh.shard_bounds(nb, rank, world)
        local_ok = [1 if o.r1cs_verify(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) == 0 else 0
                    for proof, com in recs[lo:hi]]
        full = sh.gather_accept_bits(local_ok, nb)
        # combined check: each rank reduces its shard to one point; partials are all-gathered and added
        local_pt = bytes(64)
        for proof, com in recs[lo:hi]:
            s = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap)
            local_pt = o.point_add(local_pt, s.mega_check())
            s.close()
        total_pt = sh.combine_partial_points(local_pt, _points_sum)
        tmax = sh.max_over_ranks(float(rank + 1))
        # one large MSM split by term range; 37 terms -> uneven slices
        sc, pts = o.random_scalars(91, 37), (o.gens("G", 32) + o.gens("H", 32))[:64 * 37]
        big = sh.sharded_msm(sc, pts, o.msm, _points_sum)
        # the all-gather callback a native RankGroup calls (host mirror: Prover::prove / Verifier::verify with a rank group)
        import ctypes as C
        cb = sh.allgather_callback()
        mine = (C.c_uint8 * 96)(*([rank + 1] * 96))
        out = (C.c_uint8 * (96 * world))()
        cb(mine, 96, out, None)
        q.put((rank, lo, hi, full, total_pt, tmax, big, bytes(out)))
    finally:
        dist.destroy_process_group()


import pytest   # noqa: E402


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_verification_gloo(world):
    """world 2, and world 8 -- the node the driver scales to; with 7 proofs one of the eight ranks owns an EMPTY shard (no accept
    bits to contribute, the identity as its partial point)."""
    sys.path.insert(0, HERE)
    import bp_helpers as bh
    import oracle_lib as o
    from mpc_bulletproof_amd import sharding as sh
    n_bits, nb = 4, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_bits, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards partition the batch
    assert [r[1:3] for r in res] == [sh.shard_bounds(nb, r, world) for r in range(world)]
    assert res[0][1] == 0 and res[-1][2] == nb and all(res[i][2] == res[i + 1][1] for i in range(world - 1))
    if world > nb:
        assert any(r[1] == r[2] for r in res)      # a rank with nothing to verify
    # every rank sees the same, correct, accept bits
    recs, cap = bh.make_range_batch(n_bits, nb, tamper={1, nb - 2})
    want = [1 if o.r1cs_verify(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) == 0 else 0 for proof, com in recs]
    assert want == [1, 0, 1, 1, 1, 0, 1]
    assert all(r[3] == want for r in res)
    # the combined point equals the single-process sum of all mega_check points
    acc = bytes(64)
    for proof, com in recs:
        s = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap)
        acc = o.point_add(acc, s.mega_check())
        s.close()
    assert all(r[4] == acc for r in res) and acc != bytes(64)
    assert all(r[5] == float(world) for r in res)
    sc, pts = o.random_scalars(91, 37), (o.gens("G", 32) + o.gens("H", 32))[:64 * 37]
    assert all(r[6] == o.msm(sc, pts) for r in res)
    assert all(r[7] == b"".join(bytes([k + 1] * 96) for k in range(world)) for r in res)


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from mpc_bulletproof_amd import sharding as sh
    for total in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            b = [sh.shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1

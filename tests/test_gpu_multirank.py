"""The multi-rank paths of mpc_bulletproof_amd.sharding on the HIP path: two fresh rank processes on ONE GPU (gloo for the
rendezvous and the all-gathers -- RCCL refuses two ranks per device; the 8-GPU RCCL run is the driver's), every sum
computed by libbpgpu.so, checked against the single-process GPU result and the CPU oracle."""
import json
import os
import socket
import subprocess
import sys

import pytest

import bp_helpers as bh
import oracle_lib as o

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(tmp_path, world, args, port, timeout=600):
    """start `world` rank_worker.py children (output to files: a full pipe must never stall a rank that the other one is waiting
    for in a collective), wait for all of them, and kill whatever is left on a timeout or a failure -- no orphan keeps the GPU"""
    import time
    procs, logs = [], []
    try:
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            logs.append(open(tmp_path / f"rank{r}.log", "w"))
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "rank_worker.py")] + args, env=env,
                                          stdout=logs[-1], stderr=subprocess.STDOUT))
        deadline = time.time() + timeout
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
                break                      # one rank failed (its peers would wait for it forever) or time is up
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        for f in logs:
            f.close()
    outs = [(tmp_path / f"rank{r}.log").read_text() for r in range(world)]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


def test_sharded_msm_and_combined_check_two_ranks_one_gpu(tmp_path):
    world, n_terms, n_bits, nb = 2, 2999, 8, 9
    port = _free_port()
    prefix = str(tmp_path / "rank")
    outs = _run_ranks(tmp_path, world, [prefix, str(n_terms), str(n_bits), str(nb)], port)
    res = [json.load(open(f"{prefix}.{r}")) for r in range(world)]
    # term-range-sharded MSM: host-buffer and resident-operand forms, both ranks, == the oracle's single MSM
    sc = o.random_scalars(4100, n_terms)
    pts = ((o.gens("G", 512) + o.gens("H", 512)) * 3)[:64 * n_terms]
    want = o.msm(sc, pts).hex()
    assert all(r["big_host"] == want and r["big_dev"] == want for r in res)
    # proof-level sharding: the gathered accept bits and the sum of the per-rank combined-check partials
    recs, cap = bh.make_range_batch(n_bits, nb, tamper={1})
    sessions = [o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) for proof, com in recs]
    assert all(r["ok"] == [1 if s.rc == 0 else 0 for s in sessions] == [0 if i == 1 else 1 for i in range(nb)] for r in res)
    rho = o.random_scalars(777, nb)
    comb = o.msm(rho, b"".join(s.mega_check() for s in sessions), 1).hex()
    assert all(r["comb"] == comb for r in res) and comb != bytes(64).hex()
    assert [(r["lo"], r["hi"]) for r in res] == [(0, 5), (5, 9)] and all(r["tmax"] == 2.0 for r in res)
    # the vector-sharded inner-product proof (sharding.sharded_ipp_create): byte-identical to the oracle's InnerProductProof::create
    for n_ipp in (2, 32):
        Gp, Hp, B = o.gens("G", n_ipp), o.gens("H", n_ipp), o.generator()
        av, bv = o.random_scalars(51, n_ipp), o.random_scalars(52, n_ipp)
        Gf, Hf, w = o.scalars([1] * (n_ipp // 2) + [7] * (n_ipp - n_ipp // 2)), o.random_scalars(54, n_ipp), o.random_scalars(55, 1)
        L, R, ao, bo, _ = o.ipp_create(b"innerproducttest", n_ipp, o.point_mul(w, B), Gf, Hf, Gp, Hp, av, bv)
        for r in res:
            got = r["ipp"][str(n_ipp)]
            assert (got["L"], got["R"], got["a"], got["b"]) == (L.hex(), R.hex(), ao.hex(), bo.hex()), n_ipp


def test_one_proof_sharded_over_two_ranks_full_size(tmp_path):
    """BASELINE configs[3] in its sharded form, at full size, through the host mirror: the k-shuffle proof (k = 2^4 and 2^14:
    65 533 constraints, 2^15 generators per side, a 98 347-term mega_check) proved and verified by TWO rank processes on one GPU
    (gloo all-gathers of the partial points; RCCL needs a GPU per rank), every multi-scalar multiplication split by generator /
    point range.  Both ranks return the single-process proof byte for byte and accept it; a non-permutation is rejected on both.
    The 98 347-term MSM itself, split by term range with the operands resident (sharded_msm_dev), equals the oracle-free identity."""
    import ctypes as C
    import random
    world = 2
    prefix = str(tmp_path / "rank")
    _run_ranks(tmp_path, world, [prefix, "98347", "8", "9", "4,14"], _free_port(), timeout=900)
    res = [json.load(open(f"{prefix}.{r}")) for r in range(world)]
    Gp, Gd = o.gens("G", 512, dlogs=True)
    Hp, Hd = o.gens("H", 512, dlogs=True)
    sc = o.random_scalars(4100, 98347)
    dl = ((Gd + Hd) * 97)[:32 * 98347]
    want = o.point_mul(o.inner_product(sc, dl), o.generator()).hex()         # MSM(s_i, k_i G) = (sum s_i k_i) G
    assert all(r["big_host"] == want and r["big_dev"] == want for r in res)
    host = C.CDLL(os.path.join(HERE, "host", "libbph_capi.so"))
    for lg in (4, 14):
        ks = 1 << lg
        rnd = random.Random(1000 + lg)
        xs = [rnd.getrandbits(64) for _ in range(ks)]
        ys = list(xs)
        rnd.shuffle(ys)
        cap = max(2, 1 << (2 * (ks - 1) - 1).bit_length())
        arr = (C.c_uint64 * (2 * ks))(*(xs + ys))
        proof, plen, com, ms = (C.c_uint8 * 8192)(), C.c_size_t(0), (C.c_uint8 * (2 * ks * 64))(), (C.c_double * 6)()
        assert host.bph_shuffle_prove_verify(C.c_size_t(ks), arr, C.c_uint64(4242 + lg), C.c_size_t(cap), proof, C.byref(plen), com, ms) == 0
        single = bytes(proof)[:plen.value].hex()
        for r in res:
            s = r["shuffle"][str(lg)]
            assert s["rc"] == 0 and s["rc_bad"] != 0 and s["proof"] == single, (lg, s["rc"], s["rc_bad"])
            # ... and with every rank's prover / verifier bound to the gadget's ParametricCircuit: the same bytes and verdicts
            assert s["rc_param"] == 0 and s["rc_param_bad"] == s["rc_bad"] and s["proof_param"] == single, (lg, s["rc_param"], s["rc_param_bad"])


def test_points_sum(gpu_ctx):
    G = o.generator()
    pts = [o.point_mul(o.s2b(k), G) for k in (3, 5, 11)] + [bytes(64), o.point_mul(o.s2b(o.N - 3), G)]
    assert gpu_ctx.points_sum(b"".join(pts)) == o.point_mul(o.s2b(16), G)
    assert gpu_ctx.points_sum(b"") == bytes(64)
    assert gpu_ctx.points_sum(pts[0] + pts[4]) == bytes(64)                       # P + (-P)
    assert gpu_ctx.points_sum(pts[1] * 200) == o.point_mul(o.s2b(1000), G)        # more points than lanes' first pass
    import mpc_bulletproof_amd as m
    bad = bytearray(pts[0])
    bad[1] ^= 1
    with pytest.raises(m.BpGpuError):
        gpu_ctx.points_sum(bytes(bad) + pts[1])


@pytest.fixture(scope="module")
def gpu_ctx():
    import mpc_bulletproof_amd as m
    g = m.BpGpu(0)
    yield g
    g.close()


# ------------------------------------------------------------------ arkworks in-memory forms (zero-copy hand-over)
def _ark_scalar(x):
    return (x * (1 << 256) % o.N).to_bytes(32, "little")


def _ark_point(xy, z):
    """affine boundary bytes -> ark Projective bytes (Jacobian, Montgomery R = 2^256) with the given Z"""
    P = o.P
    R = 1 << 256
    if xy == bytes(64):
        return (R % P).to_bytes(32, "little") * 2 + bytes(32)          # Projective::zero() = (1 : 1 : 0)
    x, y = int.from_bytes(xy[:32], "little"), int.from_bytes(xy[32:], "little")
    X, Y = x * z * z % P, y * z * z * z % P
    return b"".join((c * R % P).to_bytes(32, "little") for c in (X, Y, z))


def _ark_point_affine(b):
    P = o.P
    Ri = pow(1 << 256, -1, P)
    X, Y, Z = (int.from_bytes(b[32 * i:32 * i + 32], "little") * Ri % P for i in range(3))
    if Z == 0:
        return bytes(64)
    zi = pow(Z, -1, P)
    return (X * zi * zi % P).to_bytes(32, "little") + (Y * zi * zi * zi % P).to_bytes(32, "little")


@pytest.mark.parametrize("n", [0, 1, 40, 600, 3000])
def test_msm_and_conversions_in_arkworks_memory_form(gpu_ctx, n):
    """bpgpu_msm_ark takes Scalars as ark-ff Montgomery limbs (x 2^256 mod n) and StarkPoints as ark-ec Jacobian coordinates in
    Montgomery form (random Z per point, one identity) and returns the sum in the same form: equal, as a point, to the oracle's MSM of the
    canonical encodings.  The vector conversions round-trip with the boundary encodings."""
    import random
    rnd = random.Random(60 + n)
    sc = o.random_scalars(900 + n, n)
    pts = ((o.gens("G", 512) + o.gens("H", 512)) * 3)[:64 * n]
    if n >= 40:
        pts = pts[:64 * 7] + bytes(64) + pts[64 * 8:]                 # an identity operand
    xs = [int.from_bytes(sc[32 * i:32 * i + 32], "little") for i in range(n)]
    ark_sc = b"".join(_ark_scalar(x) for x in xs)
    ark_pts = b"".join(_ark_point(pts[64 * i:64 * i + 64], rnd.randrange(1, o.P) if i % 3 else 1) for i in range(n))
    got = gpu_ctx.msm_ark(ark_sc, ark_pts)
    assert _ark_point_affine(got) == (o.msm(sc, pts) if n else bytes(64))
    if n:
        assert gpu_ctx.scalars_from_ark(ark_sc) == sc and gpu_ctx.scalars_to_ark(sc) == ark_sc
        assert gpu_ctx.points_from_ark(ark_pts) == pts
        back = gpu_ctx.points_to_ark(pts)
        assert all(_ark_point_affine(back[96 * i:96 * i + 96]) == pts[64 * i:64 * i + 64] for i in range(n))
        assert gpu_ctx.points_to_ark(pts[:64]) == _ark_point(pts[:64], 1)         # Z = 1 out, byte for byte
    import mpc_bulletproof_amd as m
    if n == 40:
        bad = bytearray(ark_pts)
        bad[96 * 5 + 3] ^= 1                                                  # off the curve
        with pytest.raises(m.BpGpuError):
            gpu_ctx.msm_ark(ark_sc, bytes(bad))
        with pytest.raises(m.BpGpuError):
            gpu_ctx.scalars_from_ark(o.N.to_bytes(32, "little"))              # limbs >= modulus

"""One rank of the two-process HIP-path test (tests/test_gpu_multirank.py): torch.distributed over gloo, both ranks on
cuda:0, every compute call through libbpgpu.so.  Started as a fresh interpreter (never forked from a GPU-initialised
process).  Writes its results as JSON to argv[1].<rank>."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def main():
    out_prefix, n_terms, n_bits, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    import bp_helpers as bh
    import oracle_lib as o          # inputs only (seeded scalars, generators, oracle-made proofs): the sums are the GPU's
    import mpc_bulletproof_amd as m
    from mpc_bulletproof_amd import sharding as sh
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gpu = m.BpGpu(0)
    try:
        sc = o.random_scalars(4100, n_terms)
        pts = ((o.gens("G", 512) + o.gens("H", 512)) * ((n_terms + 1023) // 1024))[:64 * n_terms]
        big_host = sh.sharded_msm(sc, pts, gpu.msm, gpu.points_sum)
        d_sc, d_pts, d_out = gpu.to_device(sc), gpu.to_device(pts), gpu.malloc(64)
        big_dev = sh.sharded_msm_dev(gpu, d_sc, d_pts, n_terms, d_out)
        # proofs sharded by index: accept bits gathered, one combined-check partial per rank summed
        recs, cap = bh.make_range_batch(n_bits, nb, tamper={1})
        lo, hi = sh.shard_bounds(nb, rank, world)
        s0 = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], recs[0][1], recs[0][0], cap)
        rp, kind, idx, coeff = s0.csr()
        circ = gpu.circuit_create(rp, kind, idx, coeff, s0.n1 + s0.n2, s0.m)
        gens = gpu.gens_create(o.gens("G", cap), o.gens("H", cap), o.generator(), o.generator(), 8)
        p_b = q_b = c_b = b""
        for proof, com in recs[lo:hi]:
            s = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap)
            k, p, q = bh.verify_inputs(proof, com)
            p_b, q_b, c_b = p_b + p, q_b + q, c_b + s.challenges()
            s.close()
        ok, _, _ = gpu.r1cs_verify_batch(gens, circ, hi - lo, s0.n1, s0.k, s0.m, p_b, q_b, c_b)
        full_ok = sh.gather_accept_bits(ok, nb)
        rho = o.random_scalars(777, nb)
        part = gpu.r1cs_verify_combined(gens, circ, hi - lo, s0.n1, s0.k, s0.m, p_b, q_b, c_b, rho[32 * lo:32 * hi])
        comb = sh.combine_partial_points(part, gpu.points_sum)
        tmax = sh.max_over_ranks(float(rank + 1))
        # one inner-product proof with a, b, G, H dealt cyclically over the ranks: the host transcript is the
        # oracle's Python model of it, every arithmetic step is the GPU's
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pymodel as pm

        class Tr:
            def __init__(self, n):
                self.t = pm.Transcript(b"innerproducttest")
                self.t.innerproduct_domain_sep(n)

            def append_message(self, label, data):
                self.t.append_message(label, data)

            def challenge_scalar(self, label):
                return pm.s2b(self.t.challenge_scalar(label))

        ipp = {}
        for n_ipp in (2, 32):
            Gp, Hp, B = o.gens("G", n_ipp), o.gens("H", n_ipp), o.generator()
            av, bv = o.random_scalars(51, n_ipp), o.random_scalars(52, n_ipp)
            Gf, Hf, w = o.scalars([1] * (n_ipp // 2) + [7] * (n_ipp - n_ipp // 2)), o.random_scalars(54, n_ipp), o.random_scalars(55, 1)
            Ls, Rs, aa, bb = sh.sharded_ipp_create(gpu, Tr(n_ipp), n_ipp, w, B, Gf, Hf, Gp, Hp, av, bv)
            ipp[str(n_ipp)] = {"L": b"".join(Ls).hex(), "R": b"".join(Rs).hex(), "a": aa.hex(), "b": bb.hex()}
        # ONE proof split over the ranks through the host mirror (Prover::prove / Verifier::verify with a RankGroup): the k-shuffle
        # gadget; at k = 2^14 this is BASELINE configs[3] at full size (n+ = 2^15 generators per side, a 98 347-term mega_check)
        shuffle = {}
        if len(sys.argv) > 5:
            import ctypes as C
            import random
            host = C.CDLL(os.path.join(HERE, "host", "libbph_capi.so"))
            cb = sh.allgather_callback()
            for lg in [int(x) for x in sys.argv[5].split(",")]:
                ks = 1 << lg
                rnd = random.Random(1000 + lg)
                xs = [rnd.getrandbits(64) for _ in range(ks)]
                ys = list(xs)
                rnd.shuffle(ys)
                cap = max(2, 1 << (2 * (ks - 1) - 1).bit_length()) if ks > 1 else 2
                arr = (C.c_uint64 * (2 * ks))(*(xs + ys))
                proof, plen, com, ms = (C.c_uint8 * 8192)(), C.c_size_t(0), (C.c_uint8 * (2 * ks * 64))(), (C.c_double * 6)()
                rc = host.bph_shuffle_prove_verify_sharded(C.c_size_t(ks), arr, C.c_uint64(4242 + lg), C.c_size_t(cap), C.c_size_t(rank), C.c_size_t(world),
                                                           cb, None, proof, C.byref(plen), com, ms)
                bad = list(ys)
                bad[3] ^= 1                                     # not a permutation any more: the sharded verifier must reject on every rank
                arr_bad = (C.c_uint64 * (2 * ks))(*(xs + bad))
                p2, l2, c2, m2 = (C.c_uint8 * 8192)(), C.c_size_t(0), (C.c_uint8 * (2 * ks * 64))(), (C.c_double * 6)()
                rc_bad = host.bph_shuffle_prove_verify_sharded(C.c_size_t(ks), arr_bad, C.c_uint64(4242 + lg), C.c_size_t(cap), C.c_size_t(rank),
                                                               C.c_size_t(world), cb, None, p2, C.byref(l2), c2, m2)
                # the same with prover and verifier bound to the shuffle's ParametricCircuit: same proof bytes, same verdicts
                p3, l3, c3, m3 = (C.c_uint8 * 8192)(), C.c_size_t(0), (C.c_uint8 * (2 * ks * 64))(), (C.c_double * 6)()
                rc_p = host.bph_shuffle_prove_verify_sharded_param(C.c_size_t(ks), arr, C.c_uint64(4242 + lg), C.c_size_t(cap), C.c_size_t(rank),
                                                                   C.c_size_t(world), cb, None, p3, C.byref(l3), c3, m3) if ks > 1 else 0
                rc_p_bad = host.bph_shuffle_prove_verify_sharded_param(C.c_size_t(ks), arr_bad, C.c_uint64(4242 + lg), C.c_size_t(cap), C.c_size_t(rank),
                                                                       C.c_size_t(world), cb, None, p2, C.byref(l2), c2, m2) if ks > 1 else rc_bad
                shuffle[str(lg)] = {"rc": rc, "rc_bad": rc_bad, "proof": bytes(proof)[:plen.value].hex(), "prove_ms": ms[3], "verify_ms": ms[5],
                                    "rc_param": rc_p, "rc_param_bad": rc_p_bad, "proof_param": (bytes(p3)[:l3.value].hex() if ks > 1 else bytes(proof)[:plen.value].hex())}
        with open(f"{out_prefix}.{rank}", "w") as f:
            json.dump({"big_host": big_host.hex(), "big_dev": big_dev.hex(), "ok": full_ok, "comb": comb.hex(), "tmax": tmax,
                       "lo": lo, "hi": hi, "ipp": ipp, "shuffle": shuffle}, f)
    finally:
        gpu.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

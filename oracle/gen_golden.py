#!/usr/bin/env python3
"""Generate tests/golden/*.json from the pure-Python model (oracle/pymodel.py).

Run in the build container only:  python oracle/gen_golden.py
The fixtures are data (inputs + expected outputs, hex); they pin the C oracle
(tests/test_oracle.py) and the HIP path (tests/test_gpu_*.py).  No reference file is
read: the reference (Rust) cannot be executed here, and its tests hold no serialized
vectors -- its KATs are restated literally in tests/test_oracle.py.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pymodel as pm  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def hs(x):
    return pm.s2b(x).hex()


def hp(pt):
    return pm.p2b(pt).hex()


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=0, sort_keys=True)
    print(name, os.path.getsize(os.path.join(OUT, name)), "bytes")


def flat_proof(p):
    k = len(p["L_vec"])
    return (k.to_bytes(4, "little") + bytes(4)
            + b"".join(pm.p2b(p[x]) for x in ("A_I1", "A_O1", "S1", "A_I2", "A_O2", "S2",
                                              "T_1", "T_3", "T_4", "T_5", "T_6"))
            + b"".join(pm.s2b(p[x]) for x in ("t_x", "t_x_blinding", "e_blinding"))
            + b"".join(pm.p2b(x) for x in p["L_vec"]) + b"".join(pm.p2b(x) for x in p["R_vec"])
            + pm.s2b(p["a"]) + pm.s2b(p["b"])).hex()


# ----------------------------------------------------------------------------- primitives
rng = pm.SplitMix64(0xB0117E7)
prim = {"keccak256": [], "hash_to_scalar": [], "scalar_ops": [], "batch_inverse": {}, "points": {}}
for msg in (b"", b"abc", b"\x00" * 135, b"\x01" * 136, b"\x02" * 137, bytes(range(256)) * 2):
    prim["keccak256"].append({"in": msg.hex(), "out": pm.keccak256(msg).hex()})
for i in range(4):
    low = pm.keccak256(bytes([i]))
    prim["hash_to_scalar"].append({"low": low.hex(), "out": hs(pm.hash_to_scalar(low))})
edge = [0, 1, 2, pm.N - 1, pm.N - 2, (pm.N + 1) // 2, 2**251, 2**128 - 1]
vals = edge + [rng.scalar() for _ in range(8)]
for i, a in enumerate(vals):
    b = vals[(i * 7 + 3) % len(vals)]
    prim["scalar_ops"].append({"a": hs(a), "b": hs(b), "add": hs(a + b), "sub": hs(a - b), "mul": hs(a * b),
                               "inv_a": hs(pm.inv(a)) if a % pm.N else None})
nz = [v for v in vals if v % pm.N]
prim["batch_inverse"] = {"in": [hs(v) for v in nz], "out": [hs(v) for v in pm.batch_inverse(nz)]}
prim["splitmix_scalars"] = {"seed": 12345, "out": [hs(x) for x in (lambda r: [r.scalar() for _ in range(4)])(pm.SplitMix64(12345))]}
G = pm.G
P2 = pm.pt_add(G, G)
k1, k2 = rng.scalar(), rng.scalar()
A, B = pm.pt_mul(k1, G), pm.pt_mul(k2, G)
prim["points"] = {
    "G": hp(G), "2G": hp(P2), "3G": hp(pm.pt_add(P2, G)), "nm1G": hp(pm.pt_mul(pm.N - 1, G)),
    "k1": hs(k1), "k2": hs(k2), "A": hp(A), "B": hp(B), "A+B": hp(pm.pt_add(A, B)),
    "A+A": hp(pm.pt_add(A, A)), "A-A": hp(pm.pt_add(A, pm.pt_neg(A))), "A+inf": hp(pm.pt_add(A, pm.INF)),
    "negA": hp(pm.pt_neg(A)),
}
gens = pm.BulletproofGens(16, 2)
prim["generators"] = {
    "G0": [hp(p) for p in gens.G_vec[0][:16]], "H0": [hp(p) for p in gens.H_vec[0][:16]],
    "G1": [hp(p) for p in gens.G_vec[1][:2]], "H1": [hp(p) for p in gens.H_vec[1][:2]],
    "G0_dlog": [hs(k) for k in gens.G_dlog[0][:16]], "H0_dlog": [hs(k) for k in gens.H_dlog[0][:16]],
}
dump("primitives.json", prim)

# ----------------------------------------------------------------------------- MSM
msm_cases = []
pts16 = gens.G_vec[0][:16]


def msm_case(name, scal, pts):
    msm_cases.append({"name": name, "scalars": [hs(s) for s in scal], "points": [hp(p) for p in pts],
                      "out": hp(pm.msm(scal, pts))})


msm_case("empty", [], [])
msm_case("one", [rng.scalar()], [G])
msm_case("zero_scalar", [0, rng.scalar()], [pts16[0], pts16[1]])
msm_case("identity_point", [rng.scalar(), rng.scalar(), rng.scalar()], [pts16[0], pm.INF, pts16[2]])
msm_case("duplicate_points", [rng.scalar(), rng.scalar(), rng.scalar()], [G, G, pts16[3]])  # B == B_blinding
s = rng.scalar()
msm_case("cancels_to_identity", [s, pm.N - s], [pts16[4], pts16[4]])
msm_case("p_plus_minus_p", [1, 1], [pts16[5], pm.pt_neg(pts16[5])])
msm_case("small_scalars", [1, 2, 3, 4, 5, 6, 7, 8], pts16[:8])
msm_case("max_scalars", [pm.N - 1] * 4, pts16[:4])
msm_case("random16", [rng.scalar() for _ in range(16)], pts16)
msm_case("random33", [rng.scalar() for _ in range(33)], pts16 + gens.H_vec[0][:16] + [G])
dump("msm.json", msm_cases)

# ----------------------------------------------------------------------------- IPP
ipp = {"fold": [], "verification_scalars": [], "create": []}
for n in (1, 2, 4):
    u = rng.scalar()
    ui = pm.inv(u)
    a = [rng.scalar() for _ in range(2 * n)]
    b = [rng.scalar() for _ in range(2 * n)]
    Gv, Hv = gens.G_vec[0][:2 * n], gens.H_vec[0][:2 * n]
    ao, bo, Go, Ho = pm.fold_witness(u, ui, a[:n], a[n:], b[:n], b[n:], Gv[:n], Gv[n:], Hv[:n], Hv[n:])
    ipp["fold"].append({"n": n, "u": hs(u), "u_inv": hs(ui), "a": [hs(x) for x in a], "b": [hs(x) for x in b],
                        "G": [hp(x) for x in Gv], "H": [hp(x) for x in Hv], "a_out": [hs(x) for x in ao],
                        "b_out": [hs(x) for x in bo], "G_out": [hp(x) for x in Go], "H_out": [hp(x) for x in Ho]})
for k in (0, 1, 2, 3, 5):
    ch = [rng.scalar() for _ in range(k)]
    us, uis, sv = pm.verification_scalars_from_challenges(ch, 1 << k)
    ipp["verification_scalars"].append({"challenges": [hs(x) for x in ch], "u_sq": [hs(x) for x in us],
                                        "u_inv_sq": [hs(x) for x in uis], "s": [hs(x) for x in sv]})
for n in (1, 2, 4, 8):
    a = [rng.scalar() for _ in range(n)]
    b = [rng.scalar() for _ in range(n)]
    Q = pm.pt_mul(rng.scalar(), G)
    y_inv = rng.scalar()
    Hf = pm.exp_iter(y_inv, n)
    Gf = [1] * n if n != 4 else [rng.scalar() for _ in range(n)]  # benches/inner_product.rs uses random factors
    Gv, Hv = gens.G_vec[0][:n], gens.H_vec[0][:n]
    Pp = pm.msm([a[i] * Gf[i] % pm.N for i in range(n)] + [b[i] * Hf[i] % pm.N for i in range(n)]
                + [pm.inner_product(a, b)], Gv + Hv + [Q])
    trace = []
    L, R, aa, bb = pm.ipp_create(pm.Transcript(b"innerproducttest"), Q, Gf, Hf, Gv, Hv, a, b, trace)
    assert pm.ipp_verify(L, R, aa, bb, n, pm.Transcript(b"innerproducttest"), Gf, Hf, Pp, Q, Gv, Hv)
    ipp["create"].append({"n": n, "label": b"innerproducttest".hex(), "Q": hp(Q), "G_factors": [hs(x) for x in Gf],
                          "H_factors": [hs(x) for x in Hf], "a": [hs(x) for x in a], "b": [hs(x) for x in b],
                          "P": hp(Pp), "L": [hp(x) for x in L], "R": [hp(x) for x in R], "a_out": hs(aa),
                          "b_out": hs(bb), "challenges": [hs(u) for u, _ in trace]})
dump("ipp.json", ipp)

# ----------------------------------------------------------------------------- R1CS
pc = pm.PedersenGens()
bp = pm.BulletproofGens(16, 1)
r1cs = {"range": [], "shuffle": [], "example": []}


def verify_record(ve, proof, extra):
    trace = {}
    ok = ve.verify(proof, bp, trace)
    rec = dict(extra)
    rec["ok"] = ok
    if "y" in trace:
        rec.update({"challenges": [hs(trace[c]) for c in ("y", "z", "u", "x", "w", "r")] + [hs(c) for c in trace["ipp_u"]],
                    "n1": trace["n1"], "n2": trace["n2"], "padded_n": trace["padded_n"],
                    "wL": [hs(x) for x in trace["wL"]], "wR": [hs(x) for x in trace["wR"]],
                    "wO": [hs(x) for x in trace["wO"]], "wV": [hs(x) for x in trace["wV"]], "wc": hs(trace["wc"]),
                    "mega_check": hp(trace.get("mega_check"))})
    return rec


for (v, nb, seed) in ((2, 2, 1), (3, 2, 2), (4, 2, 3), (11, 4, 4), (173, 8, 5), (255, 8, 6), (256, 8, 7)):
    tr = pm.Transcript(b"RangeProofTest")
    pr = pm.Prover(pc, tr)
    r = pm.SplitMix64(seed)
    com, var = pr.commit(v, r.scalar())
    pm.range_proof_gadget(pr, pm.lc_var(var), v, nb)
    proof = pr.prove(bp, r)
    tr = pm.Transcript(b"RangeProofTest")
    ve = pm.Verifier(pc, tr)
    var = ve.commit(com)
    pm.range_proof_gadget(ve, pm.lc_var(var), None, nb)
    # capture the MSM terms too for the smallest cases
    rec = verify_record(ve, proof, {"v": v, "n_bits": nb, "seed": seed, "label": b"RangeProofTest".hex(),
                                    "commitments": [hp(com)], "proof": flat_proof(proof)})
    assert rec["ok"] == (v < (1 << nb)), (v, nb)
    r1cs["range"].append(rec)

for (k, seed, bad) in ((1, 11, False), (2, 12, False), (3, 13, False), (4, 14, False), (5, 15, False), (3, 16, True)):
    r = pm.SplitMix64(seed)           # blinding factors (commit order, then the prover's)
    rv = pm.SplitMix64(seed + 1000)   # witness values
    inp = [rv.next_u64() for _ in range(k)]
    out = inp[1:] + inp[:1]
    if bad:
        out[0] ^= 1

    def mk_tr():
        t = pm.Transcript(b"ShuffleProofTest")
        t.append_message(b"dom-sep", b"ShuffleProof")
        t.append_u64(b"k", k)
        return t

    pr = pm.Prover(pc, mk_tr())
    cv = [pr.commit(v, r.scalar()) for v in inp + out]
    pm.shuffle_gadget(pr, [c[1] for c in cv[:k]], [c[1] for c in cv[k:]])
    proof = pr.prove(bp, r)
    ve = pm.Verifier(pc, mk_tr())
    vv = [ve.commit(c[0]) for c in cv]
    pm.shuffle_gadget(ve, vv[:k], vv[k:])
    rec = verify_record(ve, proof, {"k": k, "seed": seed, "values": inp + out, "label": b"ShuffleProofTest".hex(),
                                    "commitments": [hp(c[0]) for c in cv], "proof": flat_proof(proof)})
    assert rec["ok"] == (not bad)
    r1cs["shuffle"].append(rec)

for c2 in (9, 10):
    r = pm.SplitMix64(100 + c2)
    pr = pm.Prover(pc, pm.Transcript(b"R1CSExampleGadget"))
    cv = [pr.commit(x, r.scalar()) for x in (3, 4, 6, 1, 40)]
    pm.example_gadget(pr, *[pm.lc_var(c[1]) for c in cv], pm.lc_const(c2))
    proof = pr.prove(bp, r)
    ve = pm.Verifier(pc, pm.Transcript(b"R1CSExampleGadget"))
    vv = [pm.lc_var(ve.commit(c[0])) for c in cv]
    pm.example_gadget(ve, *vv, pm.lc_const(c2))
    rec = verify_record(ve, proof, {"c2": c2, "seed": 100 + c2, "values": [3, 4, 6, 1, 40, c2],
                                    "label": b"R1CSExampleGadget".hex(), "commitments": [hp(c[0]) for c in cv],
                                    "proof": flat_proof(proof)})
    assert rec["ok"] == (c2 == 9)
    r1cs["example"].append(rec)
dump("r1cs.json", r1cs)

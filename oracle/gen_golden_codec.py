#!/usr/bin/env python3
"""Golden vectors for the wire codec from the Python model: compressed points, rejected
encodings and the wire form of the committed golden proofs (tests/golden/r1cs.json) -> tests/golden/codec.json.
Test infrastructure; run from the repo root:  python oracle/gen_golden_codec.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pymodel as pm   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
H = bytes.fromhex


def parse_flat(b):
    k = int.from_bytes(b[:4], "little")
    off = 8
    p = {}
    for name in ("A_I1", "A_O1", "S1", "A_I2", "A_O2", "S2", "T_1", "T_3", "T_4", "T_5", "T_6"):
        p[name] = pm.b2p(b[off:off + 64])
        off += 64
    for name in ("t_x", "t_x_blinding", "e_blinding"):
        p[name] = pm.b2s(b[off:off + 32])
        off += 32
    p["L_vec"] = [pm.b2p(b[off + 64 * i:off + 64 * i + 64]) for i in range(k)]
    off += 64 * k
    p["R_vec"] = [pm.b2p(b[off + 64 * i:off + 64 * i + 64]) for i in range(k)]
    off += 64 * k
    p["a"], p["b"] = pm.b2s(b[off:off + 32]), pm.b2s(b[off + 32:off + 64])
    return p


rng = pm.SplitMix64(0xC0DEC)
out = {"points": [], "invalid": [], "proofs": []}
ks = [1, 2, 3, 5, pm.N - 1, pm.N - 2, 2**128, 2**251] + [rng.scalar() for _ in range(16)]
for k in ks:
    for pt in (pm.pt_mul(k, pm.G),):
        out["points"].append({"xy": pm.p2b(pt).hex(), "compressed": pm.point_compress(pt).hex()})
out["points"].append({"xy": pm.p2b(pm.INF).hex(), "compressed": pm.point_compress(pm.INF).hex()})
# rejected encodings: x off the curve (both signs), x >= p, both flags, infinity flag is accepted whatever x says
x = 5
while pm.fp_sqrt((x ** 3 + x + pm.CURVE_B) % pm.P) is not None:
    x += 1
bad = x.to_bytes(32, "little")
out["invalid"] += [bad.hex(), (bad[:31] + bytes([bad[31] | 0x80])).hex(), pm.P.to_bytes(32, "little").hex(),
                   (pm.P + 5).to_bytes(32, "little").hex(), (bytes(31) + b"\xc0").hex()]
for b in out["invalid"]:
    try:
        pm.point_decompress(H(b))
        raise SystemExit("model accepts " + b)
    except pm.FormatError:
        pass
r1cs = json.load(open(os.path.join(OUT, "r1cs.json")))
for kind in ("range", "shuffle", "example"):
    for rec in r1cs[kind]:
        p = parse_flat(H(rec["proof"]))
        wire = pm.r1cs_proof_to_bytes(p)
        q = pm.r1cs_proof_from_bytes(wire)
        assert q == p, "round trip"
        out["proofs"].append({"kind": kind, "flat": rec["proof"], "wire": wire.hex()})
with open(os.path.join(OUT, "codec.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("codec.json", os.path.getsize(os.path.join(OUT, "codec.json")), "bytes;", len(out["points"]), "points,", len(out["proofs"]), "proofs")

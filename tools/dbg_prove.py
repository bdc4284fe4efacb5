import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o
host = C.CDLL(os.path.join(ROOT, "mpc_bulletproof_amd", "libbphost.so"))
label = b"RangeProofTest"
for (nb, nvals, n_bits) in ((1, 4, 64), (1, 8, 64), (2, 16, 64), (1, 16, 64)):
    n = nvals * n_bits
    vals = [((0x9E3779B97F4A7C15 * (i + 1 + 31 * p)) & ((1 << 64) - 1)) for p in range(nb) for i in range(nvals)]
    arr = (C.c_uint64 * len(vals))(*vals)
    plen = C.c_size_t(0); proofs = (C.c_uint8 * (nb * 4096))(); com = (C.c_uint8 * (nb * nvals * 64))()
    rc = host.bph_range_prove_batch(C.c_size_t(nb), C.c_size_t(nvals), C.c_size_t(n_bits), o._buf(label), C.c_size_t(len(label)), arr, C.c_uint64(900), C.c_size_t(n), proofs, C.byref(plen), com)
    L = plen.value
    rc2, proof_o, com_o = o.r1cs_prove(o.K_RANGE_MULTI, n_bits | (nvals << 16), label, vals[:nvals], 900, n)
    pg = bytes(proofs)[:L]
    print(nb, nvals, n, "rc", rc, rc2, "com eq", bytes(com)[:nvals*64] == com_o, "proof eq", pg == proof_o)
    if pg != proof_o:
        names = ["hdr"] + ["A_I1","A_O1","S1","A_I2","A_O2","S2","T_1","T_3","T_4","T_5","T_6"]
        off = 8
        for nm in names[1:]:
            print("  ", nm, pg[off:off+64] == proof_o[off:off+64]); off += 64
        for nm in ("t_x","t_xb","e_b"):
            print("  ", nm, pg[off:off+32] == proof_o[off:off+32]); off += 32
        k = (L - off - 64) // 128
        for j in range(k):
            print("   L", j, pg[off+64*j:off+64*j+64] == proof_o[off+64*j:off+64*j+64], "R", j, pg[off+64*k+64*j:off+64*k+64*j+64] == proof_o[off+64*k+64*j:off+64*k+64*j+64])
        break

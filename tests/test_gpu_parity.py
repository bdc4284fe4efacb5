"""GPU parity tests: every exported function of libbpgpu.so (called through the C ABI) against the
CPU oracle and the committed golden vectors, bit for bit.  Run with `-m gpu` on an MI355X."""
import pytest

import bp_helpers as bh
import oracle_lib as o

pytestmark = pytest.mark.gpu
H = bytes.fromhex
N = o.N


@pytest.fixture(scope="module")
def gpu():
    import mpc_bulletproof_amd as m
    g = m.BpGpu(0)
    yield g
    g.close()


@pytest.fixture
def opts(gpu):
    """set launch-route options of the shared context for ONE test (bpgpu_set_option); the previous values come back afterwards"""
    old = {}

    def set_(**kw):
        for k, v in kw.items():
            old.setdefault(k, gpu.get_option(k))
            gpu.set_option(k, v)
    yield set_
    for k, v in old.items():
        gpu.set_option(k, v)


def J(c, k):
    return b"".join(map(H, c[k]))


# ------------------------------------------------------------------ scalar field
def test_batch_inverse(gpu, golden_primitives):
    bi = golden_primitives["batch_inverse"]
    assert gpu.batch_inverse(J(bi, "in")) == J(bi, "out")
    for n in (1, 2, 5, 64, 1000):
        s = o.random_scalars(n, n)
        assert gpu.batch_inverse(s) == o.batch_inverse(s)
    assert gpu.batch_inverse(b"") == b""


def test_batch_inverse_rejects(gpu):
    import mpc_bulletproof_amd as m
    with pytest.raises(m.BpGpuError) as e:
        gpu.batch_inverse(o.s2b(5) + bytes(32))          # zero element
    assert e.value.code == m.lib.E_ARG
    with pytest.raises(m.BpGpuError):
        gpu.batch_inverse(N.to_bytes(32, "little"))      # non-canonical


def test_inner_product(gpu):
    assert gpu.inner_product(o.scalars([1, 2, 3, 4]), o.scalars([2, 3, 4, 5])) == o.s2b(40)   # inner_product_proof.rs:620-635
    assert gpu.inner_product(b"", b"") == bytes(32)
    for n in (1, 63, 64, 65, 1000, 70000):
        a, b = o.random_scalars(3 * n, n), o.random_scalars(3 * n + 1, n)
        assert gpu.inner_product(a, b) == o.inner_product(a, b)
    import mpc_bulletproof_amd as m
    with pytest.raises(m.BpGpuError):
        gpu.inner_product(o.scalars([1, 2]), o.scalars([1]))


# ------------------------------------------------------------------ MSM
def test_msm_golden(gpu, golden_msm):
    for c in golden_msm:
        assert gpu.msm(J(c, "scalars"), J(c, "points")) == H(c["out"]), c["name"]


def test_msm_random_and_dlog_identity(gpu):
    for n in (7, 64, 154, 1000):
        Gp, Gd = o.gens("G", n, dlogs=True)
        sc = o.random_scalars(100 + n, n)
        got = gpu.msm(sc, Gp)
        assert got == o.msm(sc, Gp)
        assert got == o.point_mul(o.inner_product(sc, Gd), o.generator())   # needs no oracle MSM


def test_msm_batch(gpu):
    nb, n = 5, 33
    Gp = o.gens("G", n)
    sc = o.random_scalars(9, nb * n)
    pts = Gp * nb
    assert gpu.msm_batch(nb, n, sc, pts) == o.msm_batch(sc, pts, nb, n)


@pytest.mark.parametrize("nb,n", [(1, 1), (1, 2), (3, 16), (4, 32), (3, 37), (2, 600), (1, 32767), (1, 32768), (1, 32769)])
def test_msm_window_parallel_group_shapes(gpu, nb, n):
    """The window-parallel MSM launches (msm_wp_batch: groups of 16 points, <= 32 for a small instance, identity-padded tails,
    one final sum per instance) at the shapes where the grouping changes, and on either side of the 2^15-term limit above which
    k_pip.hip takes over.  Identity points, zero scalars and a P / -P pair are mixed in.  Checked against the oracle for the small
    shapes and against MSM(s_i, k_i G) = (sum s_i k_i) G for the large ones."""
    sys_path_oracle()
    import pymodel as pm
    Gp, Gd = o.gens("G", n, dlogs=True)
    pts, sc = bytearray(Gp * nb), bytearray(o.random_scalars(400 + n, nb * n))
    dl = list(o.unscalars(Gd)) * nb
    if n >= 16:
        for inst in range(nb):
            b = inst * n
            pts[64 * (b + 3):64 * (b + 4)] = bytes(64); dl[b + 3] = 0                           # identity point
            sc[32 * (b + 5):32 * (b + 6)] = bytes(32)                                           # zero scalar
            pts[64 * (b + 7):64 * (b + 8)] = pm.p2b(pm.pt_neg(pm.b2p(bytes(pts[64 * (b + 6):64 * (b + 7)]))))   # -P next to P
            dl[b + 7] = (N - dl[b + 6]) % N
            sc[32 * (b + 7):32 * (b + 8)] = sc[32 * (b + 6):32 * (b + 7)]
    got = gpu.msm_batch(nb, n, bytes(sc), bytes(pts)) if nb > 1 else gpu.msm(bytes(sc), bytes(pts))
    svals = o.unscalars(bytes(sc))
    for inst in range(nb):
        tot = sum(svals[inst * n + i] * dl[inst * n + i] for i in range(n)) % N
        assert got[64 * inst:64 * inst + 64] == o.point_mul(o.s2b(tot), o.generator()), (nb, n, inst)
    if nb * n <= 2000:
        assert got == o.msm_batch(bytes(sc), bytes(pts), nb, n)


@pytest.mark.parametrize("n", [3, 29, 700])
def test_msm_shared_points(gpu, n):
    """The three local MSMs of msm_authenticated_iter (shares, MACs, public modifiers) over one point vector:
    SimpleCircuit-sized (n <= 3), k=8 shuffle-sized (29) and a bucket-method size."""
    nsets = 3
    G = o.generator()
    sc = o.random_scalars(600 + n, nsets * n)
    k = o.random_scalars(700 + n, n)
    pts = b"".join(o.point_mul(k[32 * i:32 * i + 32], G) for i in range(n)) if n < 100 else \
        (o.gens("G", 512) + o.gens("H", 512))[:64 * n]
    want = o.msm_batch(sc, pts * nsets, nsets, n)
    assert gpu.msm_shared(nsets, n, sc, pts) == want
    import mpc_bulletproof_amd as m
    bad = bytearray(pts)
    bad[5] ^= 1
    with pytest.raises(m.BpGpuError) as e:
        gpu.msm_shared(nsets, n, sc, bytes(bad))
    assert e.value.code == m.lib.E_ARG


def test_msm_batch_device_resident(gpu):
    """bpgpu_msm_batch_dev: operands and result stay in HBM; same bytes as the host-buffer call; a malformed
    operand raises the input flag."""
    nb, n = 3, 40
    sc = o.random_scalars(77, nb * n)
    pts = (o.gens("G", 64) + o.gens("H", 64))[:64 * nb * n]
    want = o.msm_batch(sc, pts, nb, n)
    d_sc, d_pts, d_out = gpu.to_device(sc), gpu.to_device(pts), gpu.malloc(64 * nb)
    gpu.msm_batch_dev(nb, n, d_sc, d_pts, d_out)
    assert gpu.download(d_out, 64 * nb) == want and gpu.input_flag() == 0
    bad = bytearray(pts)
    bad[64 * 17 + 1] ^= 2
    d_bad = gpu.to_device(bytes(bad))
    gpu.msm_batch_dev(nb, n, d_sc, d_bad, d_out)
    gpu.sync()
    assert gpu.input_flag() == 1
    for d in (d_sc, d_pts, d_out, d_bad):
        gpu.free(d)


def test_msm_from_page_locked_staging(gpu):
    """bpgpu_host_alloc / bpgpu_host_free: operands handed over from page-locked staging memory give the same result
    as from ordinary buffers (the fast hand-over path of INTEGRATION.md)."""
    import ctypes as C
    import mpc_bulletproof_amd as m
    lib = C.CDLL(m.lib.SO_PATH)
    lib.bpgpu_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.bpgpu_host_free.argtypes = [C.c_void_p]
    n = 3000
    sc, pts = o.random_scalars(4711, n), (o.gens("G", 1024) * 3)[:64 * n]
    want = gpu.msm(sc, pts)
    assert want == o.msm(sc, pts)
    ps, pp = C.c_void_p(), C.c_void_p()
    assert lib.bpgpu_host_alloc(C.c_size_t(32 * n), C.byref(ps)) == 0 and ps.value
    assert lib.bpgpu_host_alloc(C.c_size_t(64 * n), C.byref(pp)) == 0 and pp.value
    try:
        C.memmove(ps, sc, 32 * n)
        C.memmove(pp, pts, 64 * n)
        out = (C.c_uint8 * 64)()
        lib.bpgpu_msm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        assert lib.bpgpu_msm(gpu.ctx, ps, pp, C.c_size_t(n), out) == 0
        assert bytes(out) == want
    finally:
        lib.bpgpu_host_free(ps)
        lib.bpgpu_host_free(pp)
    lib.bpgpu_host_free(None)     # freeing nothing is allowed


def test_msm_rejects_bad_input(gpu):
    import mpc_bulletproof_amd as m
    G = o.generator()
    bad = bytearray(G)
    bad[0] ^= 1
    with pytest.raises(m.BpGpuError) as e:
        gpu.msm(o.s2b(3), bytes(bad))
    assert e.value.code == m.lib.E_ARG
    with pytest.raises(m.BpGpuError):
        gpu.msm(N.to_bytes(32, "little"), G)


@pytest.mark.parametrize("c", [4, 8, 20])
def test_msm_gens(gpu, c):
    cap = 16     # 20-bit windows: 13 x 2^19 table rows per generator (436 MB each, 15 GB for the 34 generators)
    Gp, Hp, B = o.gens("G", cap), o.gens("H", cap), o.generator()
    g = gpu.gens_create(Gp, Hp, B, B, c)
    try:
        for n in (16, 8, 1, 0):
            nb = 3
            sc = o.random_scalars(50 + n, nb * (2 + 2 * n))
            pts = B + B + Gp[:64 * n] + Hp[:64 * n]
            want = o.msm_batch(sc, pts * nb, nb, 2 + 2 * n)
            assert gpu.msm_gens(g, nb, n, sc) == want
            # ark-ff Montgomery limbs in (x * 2^256 mod n), same sums out; a limb vector >= n is rejected
            sc_ark = b"".join((int.from_bytes(sc[i:i + 32], "little") * (1 << 256) % N).to_bytes(32, "little") for i in range(0, len(sc), 32))
            assert gpu.msm_gens(g, nb, n, sc_ark, ark=True) == want
        import mpc_bulletproof_amd as m
        with pytest.raises(m.BpGpuError):
            gpu.msm_gens(g, 1, 0, N.to_bytes(32, "little") + bytes(32), ark=True)
        # edge scalars: 0, 1, n-1 and equal scalars on B == B_blinding (duplicate points)
        sc = o.scalars([5, 5] + [0, 1, N - 1, 2] * 8)
        assert gpu.msm_gens(g, 1, 16, sc) == o.msm(sc, B + B + Gp + Hp)
        sc = o.scalars([7, N - 7] + [0] * 32)
        assert gpu.msm_gens(g, 1, 16, sc) == bytes(64)
    finally:
        gpu.gens_destroy(g)


def test_msm_gens_with_identity_generators(gpu):
    """The ABI accepts the identity (64 zero bytes) as a generator: its table rows are the identity and must
    contribute nothing, whatever the scalar (the fixed-base lanes drop such rows before the mixed addition that
    assumes a non-identity operand)."""
    cap = 8
    Gp, Hp, B = bytearray(o.gens("G", cap)), bytearray(o.gens("H", cap)), o.generator()
    Gp[64 * 2:64 * 3] = bytes(64)            # G_2 = identity
    Hp[0:64] = bytes(64)                     # H_0 = identity
    Gp, Hp = bytes(Gp), bytes(Hp)
    for c in (8, 16):
        g = gpu.gens_create(Gp, Hp, B, bytes(64), c)      # B_blinding = identity as well
        try:
            for nb in (1, 5):
                sc = o.random_scalars(700 + nb + c, nb * (2 + 2 * cap))
                want = o.msm_batch(sc, (B + bytes(64) + Gp + Hp) * nb, nb, 2 + 2 * cap)
                assert gpu.msm_gens(g, nb, cap, sc) == want
        finally:
            gpu.gens_destroy(g)


# ------------------------------------------------------------------ IPP
def test_fold_witness(gpu, golden_ipp):
    for c in golden_ipp["fold"]:
        got = gpu.fold_witness(c["n"], H(c["u"]), H(c["u_inv"]), J(c, "a"), J(c, "b"), J(c, "G"), J(c, "H"))
        assert got == (J(c, "a_out"), J(c, "b_out"), J(c, "G_out"), J(c, "H_out"))
    n = 40
    u = o.random_scalars(1, 1)
    ui = o.sc_inv(u)
    a, b = o.random_scalars(2, 2 * n), o.random_scalars(3, 2 * n)
    G, Hh = o.gens("G", 2 * n), o.gens("H", 2 * n)
    assert gpu.fold_witness(n, u, ui, a, b, G, Hh) == o.fold_witness(n, u, ui, a, b, G, Hh)


def test_verification_scalars(gpu, golden_ipp):
    for c in golden_ipp["verification_scalars"]:
        k = len(c["challenges"])
        assert gpu.verification_scalars(J(c, "challenges"), 1 << k) == (J(c, "u_sq"), J(c, "u_inv_sq"), J(c, "s"))
    ch = o.random_scalars(4, 10)
    assert gpu.verification_scalars(ch, 1024) == o.verification_scalars(ch, 1024)
    import mpc_bulletproof_amd as m
    with pytest.raises(m.BpGpuError) as e:
        gpu.verification_scalars(ch, 512)     # n != 2^k -> VerificationError (inner_product_proof.rs:265-267)
    assert e.value.code == m.lib.E_LEN


# ------------------------------------------------------------------ R1CS
def _session(kind, param, rec, values_verify, cap=16):
    label = H(rec["label"])
    com = b"".join(map(H, rec["commitments"]))
    return o.VerifySession(kind, param, label, values_verify, com, H(rec["proof"]), cap), com


def _gens(gpu, cap, c=8):
    return gpu.gens_create(o.gens("G", cap), o.gens("H", cap), o.generator(), o.generator(), c)


def _check_record(gpu, g, kind, param, rec, values_verify):
    s, com = _session(kind, param, rec, values_verify)
    rp, kind_, idx, coeff = s.csr()
    circ = gpu.circuit_create(rp, kind_, idx, coeff, s.n1 + s.n2, s.m)
    try:
        z = H(rec["challenges"][1])
        assert gpu.flatten_constraints(circ, s.n1 + s.n2, s.m, z) == \
            (J(rec, "wL"), J(rec, "wR"), J(rec, "wO"), J(rec, "wV"), H(rec["wc"]))
        k, points, scalars = bh.verify_inputs(H(rec["proof"]), com)
        ok, mega, full = gpu.r1cs_verify_batch(g, circ, 1, s.n1, k, s.m, points, scalars, s.challenges(),
                                               want_mega=True, want_scalars=True)
        assert mega == H(rec["mega_check"])
        assert ok == [1 if rec["ok"] else 0]
        assert full == s.msm_terms()[0]
    finally:
        gpu.circuit_destroy(circ)


def test_r1cs_golden_records(gpu, golden_r1cs):
    g = _gens(gpu, 16)
    try:
        for rec in golden_r1cs["range"]:
            _check_record(gpu, g, o.K_RANGE, rec["n_bits"], rec, [])
        for rec in golden_r1cs["shuffle"]:
            _check_record(gpu, g, o.K_SHUFFLE, rec["k"], rec, [])
        for rec in golden_r1cs["example"]:
            _check_record(gpu, g, o.K_EXAMPLE, 0, rec, [rec["values"][5]])
    finally:
        gpu.gens_destroy(g)


@pytest.mark.parametrize("n_bits,nb,c", [(8, 12, 8), (64, 6, 8), (32, 4, 4), (8, 70, 20)])
def test_range_verify_batch(gpu, n_bits, nb, c):
    """Batched verification of the n-bit range gadget (config 2 shape at small batch): accept bits,
    mega_check points (also for tampered proofs) and all MSM scalars equal the oracle's."""
    tamper = {1, nb - 1}
    recs, cap = bh.make_range_batch(n_bits, nb, tamper=tamper)
    sessions = [o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) for proof, com in recs]
    s0 = sessions[0]
    rp, kind, idx, coeff = s0.csr()
    circ = gpu.circuit_create(rp, kind, idx, coeff, s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, c)
    try:
        pts = sc = ch = b""
        for (proof, com), s in zip(recs, sessions):
            k, p, q = bh.verify_inputs(proof, com)
            pts += p
            sc += q
            ch += s.challenges()
        ok, mega, full = gpu.r1cs_verify_batch(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch, True, True)
        for i, s in enumerate(sessions):
            assert ok[i] == (1 if s.rc == 0 else 0) == (0 if i in tamper else 1)
            assert mega[64 * i:64 * i + 64] == s.mega_check()
            assert full[32 * s.nterms * i:32 * s.nterms * (i + 1)] == s.msm_terms()[0]
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)


@pytest.mark.parametrize("vkeys", [False, True])
def test_prover_session_commit_and_polys(gpu, vkeys):
    """Resident-witness prover sessions (bpgpu_r1cs_prover_commit twice: 5 phase-1 + 3 phase-2 multipliers, then
    bpgpu_r1cs_prover_session_polys): the six commitments per prover equal the oracle's MSMs of prover.rs:465-494 / :532-565 over
    [B_blinding, G_lo.., H_lo..]; with vector_keys the blinding vectors are the oracle's BlindVec v1 expansion of the keys; the
    polynomial outputs (t_1..t_6, wV, l(x), r(x)) equal those of bpgpu_r1cs_prover_polys on the same operands."""
    import random
    rnd = random.Random(4242 + vkeys)
    N_ = o.N
    n, nb, cap, split = 8, 3, 16, 5
    rc, proof, com = o.r1cs_prove(o.K_RANGE, n, b"RangeProofTest", [200], 5, cap)
    s = o.VerifySession(o.K_RANGE, n, b"RangeProofTest", [], com, proof, cap)
    rp, kd, idx, coeff = s.csr()
    circ = gpu.circuit_create(rp, kd, idx, coeff, n, s.m)
    g = _gens(gpu, cap, 8)
    Gp, Hp, B = o.gens("G", cap), o.gens("H", cap), o.generator()
    ark = lambda vals: b"".join((v * (1 << 256) % N_).to_bytes(32, "little") for v in vals)       # noqa: E731
    pk = lambda vals: b"".join(o.s2b(v) for v in vals)                                               # noqa: E731
    flat = lambda rows, lo, hi: [v for r in rows for v in r[lo:hi]]                                   # noqa: E731
    wit = {k_: [[rnd.randrange(N_) if rnd.random() < 0.5 else rnd.randrange(2) for _ in range(n)] for _ in range(nb)] for k_ in ("aL", "aR", "aO")}
    sL = [[0] * n for _ in range(nb)]
    sR = [[0] * n for _ in range(nb)]
    ses = None
    try:
        for lo, hi in ((0, split), (split, n)):
            cnt = hi - lo
            blinds = [[rnd.randrange(N_) for _ in range(3)] for _ in range(nb)]
            if vkeys:
                keys = [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(nb)]
                for p in range(nb):
                    sL[p][lo:hi] = o.unscalars(o.blind_vector(keys[p], 0, cnt))
                    sR[p][lo:hi] = o.unscalars(o.blind_vector(keys[p], 1, cnt))
                ses, got = gpu.r1cs_prover_commit(g, ses, nb, cnt, ark(flat(wit["aL"], lo, hi)), ark(flat(wit["aR"], lo, hi)),
                                                  ark(flat(wit["aO"], lo, hi)), ark(flat(blinds, 0, 3)), vector_keys=b"".join(keys))
            else:
                for p in range(nb):
                    sL[p][lo:hi] = [rnd.randrange(N_) for _ in range(cnt)]
                    sR[p][lo:hi] = [rnd.randrange(N_) for _ in range(cnt)]
                ses, got = gpu.r1cs_prover_commit(g, ses, nb, cnt, ark(flat(wit["aL"], lo, hi)), ark(flat(wit["aR"], lo, hi)),
                                                  ark(flat(wit["aO"], lo, hi)), ark(flat(blinds, 0, 3)),
                                                  s_L=ark(flat(sL, lo, hi)), s_R=ark(flat(sR, lo, hi)))
            gp, hp = Gp[64 * lo:64 * hi], Hp[64 * lo:64 * hi]
            for p in range(nb):
                want = [o.msm(pk([blinds[p][0]] + wit["aL"][p][lo:hi] + wit["aR"][p][lo:hi]), B + gp + hp),
                        o.msm(pk([blinds[p][1]] + wit["aO"][p][lo:hi]), B + gp),
                        o.msm(pk([blinds[p][2]] + sL[p][lo:hi] + sR[p][lo:hi]), B + gp + hp)]
                assert [got[64 * (3 * p + w):64 * (3 * p + w + 1)] for w in range(3)] == want, (lo, p)
        ys, zs, xs = ([rnd.randrange(1, N_) for _ in range(nb)] for _ in range(3))
        t, wv = gpu.r1cs_prover_session_polys(ses, circ, nb, s.m, pk(ys), pk(zs))
        lv, rv = gpu.r1cs_prover_eval(ses, nb, n, pk(xs))
        t2, wv2, h2 = gpu.r1cs_prover_polys(circ, nb, n, s.m, pk(ys), pk([pow(y, -1, N_) for y in ys]), pk(zs), pk(flat(wit["aL"], 0, n)),
                                            pk(flat(wit["aR"], 0, n)), pk(flat(wit["aO"], 0, n)), pk(flat(sL, 0, n)), pk(flat(sR, 0, n)))
        lv2, rv2 = gpu.r1cs_prover_eval(h2, nb, n, pk(xs))
        gpu.prover_destroy(h2)
        assert (t, wv, lv, rv) == (t2, wv2, lv2, rv2)
        import mpc_bulletproof_amd as m
        with pytest.raises(m.BpGpuError):      # a second polynomial build on the same session
            gpu.r1cs_prover_session_polys(ses, circ, nb, s.m, pk(ys), pk(zs))
        with pytest.raises(m.BpGpuError):      # both sources of blinding vectors / neither
            gpu.r1cs_prover_commit(g, None, nb, 2, ark([1] * 2 * nb), ark([1] * 2 * nb), ark([1] * 2 * nb), ark([1] * 3 * nb))
    finally:
        if ses is not None:
            gpu.prover_destroy(ses)
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        s.close()


@pytest.mark.parametrize("tnp", [1, 2, 4, 8])
@pytest.mark.parametrize("nb", [1, 2, 3, 5])
def test_verify_handful_of_proofs_every_table_lane_shape(gpu, opts, nb, tnp):
    """Regression for the round-2 device fault: a handful of proofs through the window-parallel launches at every
    points-per-table-lane setting (the staging is per BLOCK of 64 lanes, so for few proofs the block rounding decides the
    scratch layout: verify_wp_scratch_bytes takes the maximum over the lane shapes, and the entry point refuses a layout that
    does not fit instead of launching).  Accept bits, mega_check points (tampered proofs included) and all MSM scalars as the
    oracle's."""
    opts(table_np=tnp)
    test_range_verify_batch(gpu, 8, nb, 8)


@pytest.mark.parametrize("lg", [1, 2, 3, 5, 7, 10])
@pytest.mark.parametrize("tnp", [0, 8])
def test_verify_single_dummy_circuit_proof(gpu, opts, lg, tnp):
    """ONE proof of the reference's bench circuit (benches/r1cs.rs:24-33: n = 2^lg multipliers in a chain) -- the shape of the
    round-2 fault (nb = 1, window-parallel launches with the generator half as its own chunked launch for the larger sizes):
    verdict, mega_check point and every MSM scalar as the oracle's, for the proof and for a tampered copy."""
    n = 1 << lg
    opts(table_np=tnp)
    rc, proof, com = o.r1cs_prove(o.K_DUMMY, n, b"test", [], 40 + lg, n)
    assert rc == 0
    bad = bytearray(proof)
    bad[8 + 11 * 64 + 3] ^= 4                                  # t_x
    g = _gens(gpu, n, 8)
    try:
        for pr in (proof, bytes(bad)):
            s = o.VerifySession(o.K_DUMMY, n, b"test", [], com, pr, n)
            rp, kind, idx, coeff = s.csr()
            circ = gpu.circuit_create(rp, kind, idx, coeff, s.n1 + s.n2, s.m)
            try:
                k, pts, sc = bh.verify_inputs(pr, com)
                ok, mega, full = gpu.r1cs_verify_batch(g, circ, 1, s.n1, s.k, s.m, pts, sc, s.challenges(), True, True)
                assert ok == [1 if pr is proof else 0] and (s.rc == 0) == (pr is proof)
                assert mega == s.mega_check() and full == s.msm_terms()[0]
            finally:
                gpu.circuit_destroy(circ)
                s.close()
    finally:
        gpu.gens_destroy(g)


@pytest.mark.parametrize("on_host", [False, True, "page_locked"])
def test_verify_stream_over_the_lane_ring(gpu, opts, on_host):
    """bpgpu_r1cs_verify_stream(_dev): 150 proofs (three tampered) of the 8-bit range gadget in ONE call, cut into batches of 16
    that take turns on three lanes: the verdicts equal the oracle's proof by proof; a second call on the same context reuses the
    lanes; a following single-batch call on the parent context sees the same results (the lanes were joined).  on_host: operands in
    pageable host memory (staged copies on every lane's stream) or page-locked (read in place by the kernels: no copy commands)."""
    import mpc_bulletproof_amd as m
    nb, tamper = 150, {2, 77, 149}
    opts(stream_batch=16, stream_lanes=3)
    recs, cap = bh.make_range_batch(8, nb, tamper=tamper)
    s0 = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], recs[0][1], recs[0][0], cap)
    circ = gpu.circuit_create(*s0.csr(), s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    try:
        pts = sc = ch = b""
        for proof, com in recs:
            s = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], com, proof, cap)
            k, p, q = bh.verify_inputs(proof, com)
            pts, sc, ch = pts + p, sc + q, ch + s.challenges()
            s.close()
        want = [0 if i in tamper else 1 for i in range(nb)]
        for rep in range(2):
            if on_host == "page_locked":
                hp, hs, hc = m.lib.host_alloc(len(pts) + 64, bytes(64) + pts), m.lib.host_alloc(len(sc), sc), m.lib.host_alloc(len(ch), ch)
                import ctypes as C_
                ok = gpu.r1cs_verify_stream(g, circ, nb, s0.n1, s0.k, s0.m, C_.c_void_p(hp.value + 64), hs, hc)     # (an interior address too)
                raw = gpu.r1cs_verify_stream(g, circ, nb, s0.n1, s0.k, s0.m, C_.c_void_p(hp.value + 64), hs, hc, raw=True)
                assert raw == b"".join(v.to_bytes(4, "little") for v in ok)
                for h in (hp, hs, hc):
                    m.lib.host_free(h)
            elif on_host:
                ok = gpu.r1cs_verify_stream(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch)
            else:
                dp, ds, dc, dok = gpu.to_device(pts), gpu.to_device(sc), gpu.to_device(ch), gpu.malloc(4 * nb)
                gpu.r1cs_verify_stream_dev(g, circ, nb, s0.n1, s0.k, dp, ds, dc, dok)
                ok = [int.from_bytes(gpu.download(dok, 4 * nb)[4 * i:4 * i + 4], "little") for i in range(nb)]   # (download syncs the parent stream)
                for d in (dp, ds, dc, dok):
                    gpu.free(d)
            assert ok == want, rep
        ok1, _, _ = gpu.r1cs_verify_batch(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch, False, False)
        assert list(ok1) == want
        assert gpu.r1cs_verify_stream(g, circ, 0, s0.n1, s0.k, s0.m, b"", b"", b"") == []
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        s0.close()


@pytest.mark.parametrize("on_host", [False, True])
def test_verify_screened_combined_check_first_per_proof_on_failure(gpu, opts, on_host):
    """bpgpu_r1cs_verify_screened(_dev): 150 proofs of the 8-bit range gadget in batches of 16 over three lanes.  All valid: every
    batch passes its combined check, no batch takes the per-proof path, every verdict is 1.  With three tampered proofs (in two
    batches) and one proof whose point bytes are off the curve (a third batch): exactly those three batches fall back to the
    per-proof path and the verdicts equal the oracle's proof by proof -- what bpgpu_r1cs_verify_stream returns."""
    import random
    nb, tamper = 150, {2, 77, 78}
    opts(stream_batch=16, screen_batch=16, stream_lanes=3)
    good, cap = bh.make_range_batch(8, nb)
    bad, _ = bh.make_range_batch(8, nb, tamper=tamper)
    s0 = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], good[0][1], good[0][0], cap)
    circ = gpu.circuit_create(*s0.csr(), s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    rnd = random.Random(4242)
    rho = b"".join(o.s2b(rnd.randrange(1, o.N)) for _ in range(nb))
    try:
        def pack(recs):
            pts = sc = ch = b""
            for proof, com in recs:
                s = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], com, proof, cap)
                k, p, q = bh.verify_inputs(proof, com)
                pts, sc, ch = pts + p, sc + q, ch + s.challenges()
                s.close()
            return pts, sc, ch

        def run(pts, sc, ch):
            if on_host:
                return gpu.r1cs_verify_screened(g, circ, nb, s0.n1, s0.k, pts, sc, ch, rho)
            dp, ds, dc, dr, dok = gpu.to_device(pts), gpu.to_device(sc), gpu.to_device(ch), gpu.to_device(rho), gpu.malloc(4 * nb)
            nf = gpu.r1cs_verify_screened_dev(g, circ, nb, s0.n1, s0.k, dp, ds, dc, dr, dok)
            ok = [int.from_bytes(gpu.download(dok, 4 * nb)[4 * i:4 * i + 4], "little") for i in range(nb)]
            for d in (dp, ds, dc, dr, dok):
                gpu.free(d)
            return ok, nf
        pts, sc, ch = pack(good)
        for rep in range(2):
            ok, nf = run(pts, sc, ch)
            assert ok == [1] * nb and nf == 0, rep
        pts, sc, ch = pack(bad)
        nvar = 11 + s0.m + 2 * s0.k
        off = (120 * nvar + 1) * 64                   # proof 120 (batch 7), its second point: y := y + 1 leaves the curve
        y = (int.from_bytes(pts[off + 32:off + 64], "little") + 1) % (1 << 251)
        pts = pts[:off + 32] + y.to_bytes(32, "little") + pts[off + 64:]
        want = [0 if (i in tamper or i == 120) else 1 for i in range(nb)]
        ok, nf = run(pts, sc, ch)
        assert ok == want and nf == 3                  # batches 0 (proof 2), 4 (proofs 77, 78), 7 (proof 120)
        ok_stream = gpu.r1cs_verify_stream(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch)
        assert ok_stream == want
        # a zeroed weight buffer (a caller's bug) must not accept anything: every batch holds a zero weight and is verified per proof
        rho_ok, rho = rho, bytes(32 * nb)
        ok, nf = run(pts, sc, ch)
        assert ok == want and nf == (nb + 15) // 16
        rho = rho_ok[:32 * 77] + bytes(32) + rho_ok[32 * 78:]     # ONE zero weight, on a tampered proof: its batch falls back
        ok, nf = run(pts, sc, ch)
        assert ok == want and nf == 3
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        s0.close()


def test_verify_screened_random_shapes(gpu, opts):
    """The screened call on ragged shapes: random proof counts, screening batch sizes that do not divide them, one to five lanes,
    random sets of tampered proofs (incl. none and all) -- the verdicts always equal the per-proof call's, and the number of
    fallback batches is the number of batches that hold a tampered proof."""
    import random
    rnd = random.Random(31337)
    pool_n = 120
    good, cap = bh.make_range_batch(8, pool_n, seed0=7000)
    bad, _ = bh.make_range_batch(8, pool_n, seed0=7000, tamper=set(range(pool_n)))
    s0 = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], good[0][1], good[0][0], cap)
    circ = gpu.circuit_create(*s0.csr(), s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    try:
        def inputs(rec):
            proof, com = rec
            s = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], com, proof, cap)
            k, p, q = bh.verify_inputs(proof, com)
            c = s.challenges()
            s.close()
            return p, q, c
        gi, bi = [inputs(r) for r in good], [inputs(r) for r in bad]
        for trial in range(10):
            nb = rnd.choice((1, 2, 17, 50, 97, 120))
            batch, lanes = rnd.choice((1, 7, 16, 33, 64, 200)), rnd.randrange(1, 6)
            frac = rnd.choice((0.0, 0.02, 0.1, 0.5, 1.0))
            tam = {i for i in range(nb) if rnd.random() < frac}
            opts(stream_batch=rnd.choice((8, 16, 64)), screen_batch=batch, stream_lanes=lanes)
            pick = [bi[i] if i in tam else gi[i] for i in range(nb)]
            pts, sc, ch = (b"".join(x[j] for x in pick) for j in range(3))
            rho = b"".join(o.s2b(rnd.randrange(1, o.N)) for _ in range(nb))
            want = [0 if i in tam else 1 for i in range(nb)]
            ok, nf = gpu.r1cs_verify_screened(g, circ, nb, s0.n1, s0.k, pts, sc, ch, rho)
            assert ok == want, (trial, nb, batch, lanes, sorted(tam))
            assert nf == len({i // batch for i in tam}), (trial, nb, batch, lanes, sorted(tam))
            ok1, _, _ = gpu.r1cs_verify_batch(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch, False, False)
            assert list(ok1) == want
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        s0.close()


def test_verify_screened_with_device_transcript(gpu, opts):
    """bpgpu_r1cs_verify_screened_fs_dev: the transcript replayed on the device per batch, then the batch's combined check.  150 valid
    proofs: no fallback.  With two tampered proofs, one off-curve point and one proof whose A_I1 is the identity (the transcript's
    validate_and_append_point rejects it): exactly those four batches fall back, and the verdicts are bpgpu_r1cs_verify_batch_fs's
    (= the oracle's whole Verifier::verify)."""
    import random
    sys_path_oracle()
    import pymodel as pm
    nb, tamper = 150, {5, 99}
    opts(stream_batch=16, screen_batch=16, stream_lanes=3)
    good, cap = bh.make_range_batch(8, nb)
    bad, _ = bh.make_range_batch(8, nb, tamper=tamper)
    s0 = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], good[0][1], good[0][0], cap)
    circ = gpu.circuit_create(*s0.csr(), s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    rnd = random.Random(777)
    rho = b"".join(o.s2b(rnd.randrange(1, o.N)) for _ in range(nb))
    init = pm.Transcript(b"RangeProofTest").state * nb
    nvar = 11 + s0.m + 2 * s0.k
    try:
        def pack(recs):
            pts = sc = b""
            for proof, com in recs:
                k, p, q = bh.verify_inputs(proof, com)
                pts, sc = pts + p, sc + q
            return pts, sc

        def run(pts, sc):
            di, dp, ds, dr, dok = gpu.to_device(init), gpu.to_device(pts), gpu.to_device(sc), gpu.to_device(rho), gpu.malloc(4 * nb)
            nf = gpu.r1cs_verify_screened_fs_dev(g, circ, nb, s0.n1, s0.k, di, dp, ds, dr, dok)
            ok = [int.from_bytes(gpu.download(dok, 4 * nb)[4 * i:4 * i + 4], "little") for i in range(nb)]
            for d in (di, dp, ds, dr, dok):
                gpu.free(d)
            return ok, nf
        pts, sc = pack(good)
        assert run(pts, sc) == ([1] * nb, 0)
        pts, sc = pack(bad)
        off = (120 * nvar + 1) * 64                   # proof 120: A_O1 off the curve
        y = (int.from_bytes(pts[off + 32:off + 64], "little") + 1) % (1 << 251)
        pts = pts[:off + 32] + y.to_bytes(32, "little") + pts[off + 64:]
        off = 40 * nvar * 64                          # proof 40: A_I1 = identity
        pts = pts[:off] + bytes(64) + pts[off + 64:]
        want, _, _ = gpu.r1cs_verify_batch_fs(g, circ, nb, s0.n1, s0.k, s0.m, init, pts, sc, want_mega=False)
        assert want == [0 if i in (5, 99, 120, 40) else 1 for i in range(nb)]
        ok, nf = run(pts, sc)
        assert ok == want and nf == 4                  # batches 0 (proof 5), 2 (40), 6 (99), 7 (120)
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        s0.close()


@pytest.mark.parametrize("world", [2, 3, 8])     # 8: more ranks than some of the vectors have entries (empty shares)
def test_shard_partial_sums_add_up(gpu, world):
    """bpgpu_set_shard / bpgpu_r1cs_verify_shard (ONE large proof over the GPUs of a node) on one context, rank after
    rank: the partial commitments of a prover session, the partial L, R of every IPP round and the partial mega_check points of a
    verification each add up (bpgpu_points_sum) to what the unsharded call returns -- incl. a tampered proof's non-identity point."""
    import random
    rnd = random.Random(77 + world)
    N_ = o.N
    n, nb, cap = 8, 2, 8
    ark = lambda vals: b"".join((v * (1 << 256) % N_).to_bytes(32, "little") for v in vals)       # noqa: E731
    pk = lambda vals: b"".join(o.s2b(v) for v in vals)                                               # noqa: E731
    g = _gens(gpu, cap, 8)
    try:
        # --- prover commitments: explicit blinding vectors so that every rank's session holds the same operands
        wit = [[rnd.randrange(N_) for _ in range(nb * n)] for _ in range(5)]
        blinds = [rnd.randrange(N_) for _ in range(3 * nb)]
        ses, full = gpu.r1cs_prover_commit(g, None, nb, n, ark(wit[0]), ark(wit[1]), ark(wit[2]), ark(blinds), s_L=ark(wit[3]), s_R=ark(wit[4]))
        gpu.prover_destroy(ses)
        parts = []
        for r in range(world):
            gpu.set_shard(r, world)
            ses, part = gpu.r1cs_prover_commit(g, None, nb, n, ark(wit[0]), ark(wit[1]), ark(wit[2]), ark(blinds), s_L=ark(wit[3]), s_R=ark(wit[4]))
            gpu.prover_destroy(ses)
            parts.append(part)
        gpu.set_shard(0, 1)
        for j in range(3 * nb):
            assert gpu.points_sum(b"".join(p_[64 * j:64 * j + 64] for p_ in parts)) == full[64 * j:64 * j + 64], j
        # --- IPP rounds over resident generators: partial L, R per round; the scalar state is folded on every rank alike
        a, b = [rnd.randrange(N_) for _ in range(n)], [rnd.randrange(N_) for _ in range(n)]
        Gf, Hf, w = [1] * n, [rnd.randrange(1, N_) for _ in range(n)], [rnd.randrange(1, N_)]

        def rounds(rank, world_):
            gpu.set_shard(rank, world_)
            s_ = gpu.ipp_begin_gens(g, 1, n, pk(w), pk(Gf), pk(Hf), pk(a), pk(b))
            gpu.set_shard(0, 1)
            out, rr = [], random.Random(5)
            while gpu.ipp_len(s_) > 1:
                out.append(gpu.ipp_round(s_, 1))
                u = rr.randrange(1, N_)
                gpu.ipp_fold(s_, pk([u]), pk([pow(u, -1, N_)]))
            fin = gpu.ipp_finish(s_, 1)
            gpu.ipp_destroy(s_)
            return out, fin
        ref_rounds, ref_fin = rounds(0, 1)
        per_rank = [rounds(r, world) for r in range(world)]
        assert all(fin == ref_fin for _, fin in per_rank)
        for j, (L, R) in enumerate(ref_rounds):
            assert gpu.points_sum(b"".join(pr[0][j][0] for pr in per_rank)) == L and gpu.points_sum(b"".join(pr[0][j][1] for pr in per_rank)) == R, j
        # --- verification: partial mega_check points (valid proof -> identity; tampered -> the oracle's point)
        recs, cap2 = bh.make_range_batch(8, 2, tamper={1})
        for proof, com in recs:
            s = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], com, proof, cap2)
            circ = gpu.circuit_create(*s.csr(), s.n1 + s.n2, s.m)
            k, pts, sc = bh.verify_inputs(proof, com)
            parts = [gpu.r1cs_verify_shard(g, circ, s.n1, s.k, pts, sc, s.challenges(), r, world) for r in range(world)]
            assert gpu.points_sum(b"".join(parts)) == s.mega_check()
            assert (s.mega_check() == bytes(64)) == (s.rc == 0)
            gpu.circuit_destroy(circ)
            s.close()
        import mpc_bulletproof_amd as m
        with pytest.raises(m.BpGpuError):
            gpu.set_shard(3, 3)
    finally:
        gpu.set_shard(0, 1)
        gpu.gens_destroy(g)


def test_one_context_shared_by_concurrent_threads(gpu):
    """"thread-safe and re-entrant" -- the reference calls this arithmetic from rayon workers (inner_product_proof.rs:233-247)
    and fabric executor threads (transcript.rs:155-161, r1cs_mpc/mpc_prover.rs:621-657).  Eight threads drive ONE bpgpu_ctx at once
    with a mix of bpgpu_msm (window-parallel and bucket sizes), bpgpu_msm_batch, bpgpu_batch_inverse, bpgpu_msm_shared,
    bpgpu_inner_product, bpgpu_r1cs_verify_batch (with a tampered proof) and an erroring call (a malformed point); every result of
    every repetition must equal the oracle's, whatever the interleaving.  (ctypes releases the GIL inside the calls.)"""
    import threading
    import mpc_bulletproof_amd as m
    Gp = o.gens("G", 600)
    jobs = []          # (name, callable, expected)
    for n in (3, 41, 600):
        sc = o.random_scalars(5000 + n, n)
        pts = Gp[:64 * n]
        jobs.append((f"msm{n}", (lambda sc=sc, pts=pts: gpu.msm(sc, pts)), o.msm(sc, pts)))
    sc = o.random_scalars(5100, 4 * 20)
    jobs.append(("msm_batch", (lambda sc=sc: gpu.msm_batch(4, 20, sc, Gp[:64 * 20] * 4)), o.msm_batch(sc, Gp[:64 * 20] * 4, 4, 20)))
    inv_in = o.random_scalars(5200, 300)
    jobs.append(("batch_inverse", (lambda: gpu.batch_inverse(inv_in)), o.batch_inverse(inv_in)))
    ssc = o.random_scalars(5300, 3 * 29)
    jobs.append(("msm_shared", (lambda: gpu.msm_shared(3, 29, ssc, Gp[:64 * 29])), o.msm_batch(ssc, Gp[:64 * 29] * 3, 3, 29)))
    ia, ib = o.random_scalars(5400, 500), o.random_scalars(5401, 500)
    jobs.append(("inner_product", (lambda: gpu.inner_product(ia, ib)), o.inner_product(ia, ib)))
    nbv = 9
    recs, cap = bh.make_range_batch(8, nbv, seed0=5500, tamper={4})
    sess = [o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], com, proof, cap) for proof, com in recs]
    s0 = sess[0]
    circ = gpu.circuit_create(*s0.csr(), s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    vp = vs = vc = b""
    for (proof, com), s_ in zip(recs, sess):
        k, p_, q_ = bh.verify_inputs(proof, com)
        vp, vs, vc = vp + p_, vs + q_, vc + s_.challenges()
    want_v = ([1 if s_.rc == 0 else 0 for s_ in sess], b"".join(s_.mega_check() for s_ in sess))
    jobs.append(("verify", (lambda: gpu.r1cs_verify_batch(g, circ, nbv, s0.n1, s0.k, s0.m, vp, vs, vc, True, False)[:2]), want_v))
    badpts = bytearray(Gp[:64 * 5])
    badpts[7] ^= 1

    def erroring():
        try:
            gpu.msm(o.random_scalars(1, 5), bytes(badpts))
        except m.BpGpuError as e:
            return e.code
        return 0
    jobs.append(("malformed", erroring, m.lib.E_ARG))
    errors = []
    start = threading.Barrier(8)

    def worker(t):
        try:
            start.wait()
            for rep in range(6):
                for j in range(len(jobs)):
                    name, fn, want = jobs[(j + t * 3 + rep) % len(jobs)]
                    got = fn()
                    if got != want:
                        errors.append((t, rep, name))
        except Exception as e:      # noqa: BLE001
            errors.append((t, "exception", repr(e)))
    try:
        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors[:5]
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        for s_ in sess:
            s_.close()


def test_shard_malformed_point_is_a_collective_verdict(gpu):
    """on the large-proof route (nvar > 256: one proof's points validated slice by slice) an off-curve point is seen only
    by the rank whose share holds it.  That rank must not fail alone (the others would wait in the all-gather) and must not return a
    sum that silently omits the term: it returns the poison encoding (64 x 0xFF), every other rank its ordinary partial, and
    bpgpu_points_sum over the gathered partials fails with BPGPU_E_ARG -- the same on every rank.  The small-proof route likewise."""
    import mpc_bulletproof_amd as m
    for kind, param, vals, cap, where in ((o.K_RANGE_MULTI, 1 | (256 << 16), [i & 1 for i in range(256)], 256, 6 + 200),
                                          (o.K_RANGE, 8, [77], 8, 1)):
        rc, proof, com = o.r1cs_prove(kind, param, b"RangeProofTest", vals, 4242, cap)
        assert rc == 0
        s = o.VerifySession(kind, param, b"RangeProofTest", [], com, proof, cap)
        assert s.rc == 0
        nvar = 11 + s.m + 2 * s.k
        assert (nvar > 256) == (kind == o.K_RANGE_MULTI)
        g = _gens(gpu, cap, 8)
        circ = gpu.circuit_create(*s.csr(), s.n1 + s.n2, s.m)
        try:
            k, pts, sc = bh.verify_inputs(proof, com)
            world = 2
            good = [gpu.r1cs_verify_shard(g, circ, s.n1, s.k, pts, sc, s.challenges(), r, world) for r in range(world)]
            assert gpu.points_sum(b"".join(good)) == bytes(64) and bytes([0xFF]) * 64 not in good
            bad = bytearray(pts)
            bad[64 * where + 32] ^= 1                       # y ^ 1: not on the curve
            parts = [gpu.r1cs_verify_shard(g, circ, s.n1, s.k, bytes(bad), sc, s.challenges(), r, world) for r in range(world)]
            owner = 0 if where < (nvar + 1) // 2 else 1     # contiguous shares, the first one larger by at most one
            assert parts[owner] == bytes([0xFF]) * 64
            if nvar > 256:       # large-proof route: a rank validates its own slice of the points only
                assert parts[1 - owner] == good[1 - owner]
            else:                # small-proof route: every rank's table launch reads all points (the other ranks' SCALARS are zeroed)
                assert parts[1 - owner] == bytes([0xFF]) * 64
            with pytest.raises(m.BpGpuError) as e:
                gpu.points_sum(b"".join(parts))
            assert e.value.code == m.lib.E_ARG
            ok, _, _ = gpu.r1cs_verify_batch(g, circ, 1, s.n1, s.k, s.m, bytes(bad), sc, s.challenges(), want_mega=False)
            assert ok == [0]                                 # (the unsharded call rejects the same proof)
            # a non-canonical proof scalar: every rank assembles all scalars, so every rank poisons its partial
            bsc = bytes([0xFF]) * 32 + sc[32:]
            parts = [gpu.r1cs_verify_shard(g, circ, s.n1, s.k, pts, bsc, s.challenges(), r, world) for r in range(world)]
            assert parts == [bytes([0xFF]) * 64] * world
        finally:
            gpu.circuit_destroy(circ)
            gpu.gens_destroy(g)
            s.close()


def test_options_setter_rejects_bad_values(gpu):
    import mpc_bulletproof_amd as m
    for name, bad in (("verify_straus_np", 5), ("table_np", 3), ("vs_large_min", 0), ("ipp_literal", 2), ("msm_wp_max", -1)):
        with pytest.raises(m.BpGpuError):
            gpu.set_option(name, bad)
    assert gpu.get_option("msm_wp_max") == 1 << 15 and gpu.get_option("verify_window_parallel") == 1
    with gpu.options(msm_wp_max=7):
        assert gpu.get_option("msm_wp_max") == 7
    assert gpu.get_option("msm_wp_max") == 1 << 15
    for name, bad in (("horner_form", 4), ("groups_form", 4), ("fixed_lpm", 48), ("stream_batch", 0), ("stream_lanes", 65), ("pippenger_min", 1),
                      ("fixed_chunk_gens", -2), ("fixed_chunk_gens", 65)):
        with pytest.raises(m.BpGpuError):
            gpu.set_option(name, bad)


def test_options_seeded_from_the_environment_are_validated(monkeypatch):
    """a new context's options may be seeded from the environment, through the SAME validator as bpgpu_set_option --
    BPGPU_STREAM_BATCH=0 (a division by zero before), a negative lane count, a table_np that is not a lane shape and garbage leave
    the defaults in force; valid values are taken."""
    import mpc_bulletproof_amd as m
    for k, v in (("BPGPU_STREAM_BATCH", "0"), ("BPGPU_SCREEN_BATCH", "-5"), ("BPGPU_STREAM_LANES", "0"), ("BPGPU_TABLE_NP", "3"),
                 ("BPGPU_FIXED_LPM", "17"), ("BPGPU_HORNER_FORM", "x1"), ("BPGPU_MSM_WP_MAX", "4096"), ("BPGPU_GROUPS_FORM", "2")):
        monkeypatch.setenv(k, v)
    g2 = m.BpGpu(0)
    try:
        assert g2.get_option("stream_batch") == 1024 and g2.get_option("screen_batch") == 2560 and g2.get_option("stream_lanes") == 20
        assert g2.get_option("table_np") == 0 and g2.get_option("fixed_lpm") == 0 and g2.get_option("horner_form") == 0
        assert g2.get_option("msm_wp_max") == 4096 and g2.get_option("groups_form") == 2
    finally:
        g2.close()


def test_circuit_from_arkworks_coefficients(gpu):
    """bpgpu_circuit_create_ark: constraint coefficients as ark-ff Montgomery limbs (x * 2^256 mod n) give the same flattened
    weights as the canonical-bytes circuit (range gadget and 2-phase shuffle rows); a limb vector >= n is rejected."""
    import mpc_bulletproof_amd as m
    for okind, param, label, vals in ((o.K_RANGE, 8, b"RangeProofTest", [77]), (o.K_SHUFFLE, 4, b"ShuffleProofTest", [5, 9, 2, 7, 2, 5, 7, 9])):
        rc, proof, com = o.r1cs_prove(okind, param, label, vals, 3, 16)
        assert rc == 0
        s = o.VerifySession(okind, param, label, [], com, proof, 16)
        rp, kind, idx, coeff = s.csr()
        ark = b"".join((int.from_bytes(coeff[i:i + 32], "little") * (1 << 256) % N).to_bytes(32, "little") for i in range(0, len(coeff), 32))
        n_mul, mm = s.n1 + s.n2, s.m
        c1 = gpu.circuit_create(rp, kind, idx, coeff, n_mul, mm)
        c2 = gpu.circuit_create(rp, kind, idx, ark, n_mul, mm, ark=True)
        try:
            z = o.s2b(0x1234567 + param)
            assert gpu.flatten_constraints(c1, n_mul, mm, z) == gpu.flatten_constraints(c2, n_mul, mm, z)
        finally:
            gpu.circuit_destroy(c1)
            gpu.circuit_destroy(c2)
            s.close()
        with pytest.raises(m.BpGpuError):
            gpu.circuit_create(rp, kind, idx, N.to_bytes(32, "little") + ark[32:], n_mul, mm, ark=True)


def test_verify_scalars_large_proof_path(gpu, golden_r1cs, opts):
    """The grid-split scalar assembly used for large proofs (padded_n or m >= 4096: the 2^14-shuffle of BASELINE
    configs[3]) forced onto small circuits: every MSM scalar, mega_check point and accept bit as the oracle's."""
    opts(vs_large_min=1)
    g = _gens(gpu, 16)
    try:
        for rec in golden_r1cs["range"]:
            _check_record(gpu, g, o.K_RANGE, rec["n_bits"], rec, [])
        for rec in golden_r1cs["shuffle"]:
            _check_record(gpu, g, o.K_SHUFFLE, rec["k"], rec, [])
        for rec in golden_r1cs["example"]:
            _check_record(gpu, g, o.K_EXAMPLE, 0, rec, [rec["values"][5]])
    finally:
        gpu.gens_destroy(g)
    test_range_verify_batch(gpu, 64, 5, 8)
    # a shuffle big enough for several blocks per kernel: k = 300 -> n = 598, padded_n = 1024, m = 600, q = 1197
    k = 300
    import random
    rnd = random.Random(5)
    x = [rnd.getrandbits(64) for _ in range(k)]
    y = list(x)
    rnd.shuffle(y)
    rc, proof, com = o.r1cs_prove(o.K_SHUFFLE, k, b"ShuffleProofTest", x + y, 31, 1024)
    assert rc == 0
    s = o.VerifySession(o.K_SHUFFLE, k, b"ShuffleProofTest", [], com, proof, 1024)
    rp, kind, idx, coeff = s.csr()
    circ = gpu.circuit_create(rp, kind, idx, coeff, s.n1 + s.n2, s.m)
    g = _gens(gpu, 1024, 4)
    try:
        kk, points, scalars = bh.verify_inputs(proof, com)
        for tam in (False, True):
            sess = s
            sc = scalars
            if tam:
                bad = bytearray(proof)
                bad[-1] ^= 1       # ipp b
                sess = o.VerifySession(o.K_SHUFFLE, k, b"ShuffleProofTest", [], com, bytes(bad), 1024)
                _, points, sc = bh.verify_inputs(bytes(bad), com)
            ok, mega, full = gpu.r1cs_verify_batch(g, circ, 1, s.n1, kk, s.m, points, sc, sess.challenges(), True, True)
            assert ok == [0 if tam else 1]
            assert mega == sess.mega_check()
            assert full == sess.msm_terms()[0]
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)


@pytest.mark.parametrize("np_,fuse,wp,c", [(4, True, 1, 8), (4, True, 1, 16), (4, True, 0, 8), (3, True, 0, 8), (2, True, 0, 8),
                                           (4, False, 0, 8), (1, False, 0, 8)])
def test_verify_batch_launch_variants(gpu, opts, np_, fuse, wp, c):
    """Batches of >= 64 proofs take the window-parallel path (tables | windows | Horner + verdict, 18 proof points =
    4.5 table lanes per proof); option verify_window_parallel = 0 selects the fused Straus launch (role-major lanes + small
    fixed-base MSMs in one kernel, a remainder launch for 18 mod 4 points), verify_no_fuse = 1 the separate launches.
    Accept bits, mega_check points (tampered proofs included) and all MSM scalars must equal the oracle's in every
    variant."""
    opts(verify_straus_np=np_, verify_window_parallel=wp, verify_no_fuse=0 if fuse else 1)
    test_range_verify_batch(gpu, 8, 70, c)


@pytest.mark.parametrize("horner,groups,lpm", [(1, 1, 16), (2, 2, 32), (3, 3, 64), (3, 1, 0), (2, 3, 0)])
def test_verify_batch_horner_forms(gpu, opts, horner, groups, lpm):
    """The Horner pass and its first stage in each of their forms -- a lane, a DPP quad (ec29_quad.cuh), a whole wave with
    row-distributed field arithmetic (ec29_row.cuh) per proof / per group of 8 windows -- and the lanes per fixed-base MSM of the
    back launch, selected through the per-context options in the DEFAULT mode (70 proofs, tampered ones among them): accept bits,
    mega_check points and all MSM scalars equal the oracle's in every combination."""
    opts(horner_form=horner, groups_form=groups, fixed_lpm=lpm)
    test_range_verify_batch(gpu, 8, 70, 8)


@pytest.mark.parametrize("gens,nb,c", [(-1, 70, 8), (1, 70, 8), (3, 70, 8), (5, 130, 8), (18, 65, 8), (64, 7, 8), (5, 70, 16), (0, 300, 8)])
def test_verify_batch_generator_half_a_proof_per_lane(gpu, opts, gens, nb, c):
    """The generator half of the back launch walked a PROOF per lane, a run of `gens` generators per wave (fixed_chunk_body), the
    partial sums of a proof added in the verdict launch -- for runs of 1 generator (18 partials of the 8-bit gadget's 18 generators)
    up to one run of all of them, batches that leave dead lanes in their last wave (7, 65, 70, 130 proofs), tampered proofs among them:
    accept bits, mega_check points and all MSM scalars equal the oracle's; -1 forces the lanes-per-proof form, 0 is the default
    rule (the proof-per-lane walk from 256 proofs on)."""
    opts(fixed_chunk_gens=gens)
    test_range_verify_batch(gpu, 8, nb, c)


@pytest.mark.parametrize("n_bits,nb,c", [(8, 71, 8), (8, 12, 8), (64, 65, 20)])
def test_verify_batch_latency_mode(gpu, n_bits, nb, c):
    """bpgpu_set_latency_mode (the un-pipelined caller's setting: 1 point per table lane, the generator half on a side stream with
    64 lanes per MSM, a whole wave per proof -- row-distributed arithmetic -- in both Horner stages): same accept bits, mega_check
    points and MSM scalars as the oracle."""
    gpu.set_latency_mode(True)
    try:
        test_range_verify_batch(gpu, n_bits, nb, c)
    finally:
        gpu.set_latency_mode(False)


@pytest.mark.parametrize("wp", [1, 0])
def test_verify_batch_rejects_malformed_proofs_one_by_one(gpu, opts, wp):
    """An off-curve proof point, a non-canonical proof scalar or a non-canonical challenge makes THAT proof's accept bit
    0 and leaves the other verdicts of a 70-proof batch alone (the reference rejects a malformed proof with
    FormatError / VerificationError on its own: r1cs/proof.rs:128-207, verifier.rs:401-444); the call itself
    succeeds.  Both the window-parallel launches (validation inside the table lanes / the scalar assembly) and the
    fused Straus launch.  The context-wide input flag is raised as a diagnostic and cleared by reading it."""
    opts(verify_window_parallel=wp)
    recs, cap = bh.make_range_batch(8, 70)
    s0 = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], recs[0][1], recs[0][0], cap)
    rp, kind, idx, coeff = s0.csr()
    circ = gpu.circuit_create(rp, kind, idx, coeff, s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    try:
        pts = sc = ch = b""
        for proof, com in recs:
            s = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], com, proof, cap)
            k, p, q = bh.verify_inputs(proof, com)
            pts += p
            sc += q
            ch += s.challenges()
            s.close()
        ok, _, _ = gpu.r1cs_verify_batch(g, circ, 70, s0.n1, s0.k, s0.m, pts, sc, ch)
        assert ok == [1] * 70 and gpu.input_flag() == 0
        nvar = 11 + s0.m + 2 * s0.k
        bad_pts = bytearray(pts)
        bad_pts[64 * (nvar * 37 + 9) + 3] ^= 4          # proof 37, point 9: off the curve
        bad_pts[64 * (nvar * 2 + nvar - 1):64 * (nvar * 2 + nvar)] = o.P.to_bytes(32, "little") + bytes(32)   # proof 2, last point: x = p
        ok, _, _ = gpu.r1cs_verify_batch(g, circ, 70, s0.n1, s0.k, s0.m, bytes(bad_pts), sc, ch)
        assert ok == [0 if i in (2, 37) else 1 for i in range(70)]
        assert gpu.input_flag() == 1 and gpu.input_flag() == 0
        bad_sc = bytearray(sc)
        bad_sc[32 * (5 * 51 + 2):32 * (5 * 51 + 3)] = N.to_bytes(32, "little")      # proof 51: e_blinding = n
        ok, _, _ = gpu.r1cs_verify_batch(g, circ, 70, s0.n1, s0.k, s0.m, pts, bytes(bad_sc), ch)
        assert ok == [0 if i == 51 else 1 for i in range(70)]
        bad_ch = bytearray(ch)
        bad_ch[32 * ((6 + s0.k) * 3 + 7):32 * ((6 + s0.k) * 3 + 8)] = b"\xff" * 32  # proof 3: u_2 >= n
        ok, _, _ = gpu.r1cs_verify_batch(g, circ, 70, s0.n1, s0.k, s0.m, pts, sc, bytes(bad_ch))
        assert ok == [0 if i == 3 else 1 for i in range(70)]
        # all three at once, through the device-resident entry point with 3 calls in flight before the first read
        d_ok = [gpu.malloc(4 * 70) for _ in range(3)]
        d_in = [(gpu.to_device(a), gpu.to_device(b), gpu.to_device(c)) for a, b, c in
                ((bytes(bad_pts), sc, ch), (pts, bytes(bad_sc), ch), (pts, sc, ch))]
        for (dp, ds, dc), dk in zip(d_in, d_ok):
            gpu.r1cs_verify_batch_dev(g, circ, 70, s0.n1, s0.k, dp, ds, dc, dk)
        gpu.sync()
        got = [list(int.from_bytes(gpu.download(dk, 4 * 70)[4 * i:4 * i + 4], "little") for i in range(70)) for dk in d_ok]
        assert got[0] == [0 if i in (2, 37) else 1 for i in range(70)]
        assert got[1] == [0 if i == 51 else 1 for i in range(70)] and got[2] == [1] * 70
        for t in d_in:
            for d in t:
                gpu.free(d)
        for d in d_ok:
            gpu.free(d)
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)


def test_verify_batch_generators_too_short(gpu):
    import mpc_bulletproof_amd as m
    recs, cap = bh.make_range_batch(8, 1)
    s = o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], recs[0][1], recs[0][0], cap)
    rp, kind, idx, coeff = s.csr()
    circ = gpu.circuit_create(rp, kind, idx, coeff, 8, 1)
    g = _gens(gpu, 4)
    try:
        k, p, q = bh.verify_inputs(*recs[0])
        with pytest.raises(m.BpGpuError) as e:
            gpu.r1cs_verify_batch(g, circ, 1, 8, k, 1, p, q, s.challenges())
        assert e.value.code == m.lib.E_GENS      # R1CSError::InvalidGeneratorsLength (verifier.rs:421-423)
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)


# ------------------------------------------------------------------ prover polynomials
@pytest.mark.parametrize("kind,param,nb", [(0, 8, 3), (0, 64, 2), (1, 5, 2)])
def test_prover_polys_and_eval_unit_parity(gpu, kind, param, nb):
    """bpgpu_r1cs_prover_polys / _eval against r1cs/prover.rs:587-619, 659-672 and util.rs:152-181 restated literally on
    Python integers: t_1..t_6 (special_inner_product), wV, and l_vec / r_vec with the zero / -y^i padding -- directly,
    not through whole-proof bytes.  Circuits: the 8- and 64-bit range gadget (n = 8, 64: padded = n) and a 5-shuffle
    (n = 8 multipliers of a 2-phase circuit).  The weights come from the oracle's flattened_constraints."""
    okind = [o.K_RANGE, o.K_SHUFFLE][kind]
    label = [b"RangeProofTest", b"ShuffleProofTest"][kind]
    import random
    rnd = random.Random(77 + param)
    vals = [rnd.getrandbits(param)] if kind == 0 else (lambda x: x + sorted(x))([rnd.getrandbits(32) for _ in range(param)])
    cap = 64
    rc, proof, com = o.r1cs_prove(okind, param, label, vals, 5, cap)
    assert rc == 0
    s = o.VerifySession(okind, param, label, [], com, proof, cap)
    n, m = s.n1 + s.n2, s.m
    padded = 1 << max(0, (n - 1).bit_length())
    rp, kd, idx, coeff = s.csr()
    circ = gpu.circuit_create(rp, kd, idx, coeff, n, m)
    N_ = o.N
    I = lambda b, i: int.from_bytes(b[32 * i:32 * i + 32], "little")      # noqa: E731
    try:
        ys = [rnd.randrange(1, N_) for _ in range(nb)]
        zs = [rnd.randrange(1, N_) for _ in range(nb)]
        xs = [rnd.randrange(1, N_) for _ in range(nb)]
        wit = {k_: [[rnd.randrange(N_) if k_ in ("sL", "sR") or rnd.random() < 0.5 else rnd.randrange(2) for _ in range(n)]
                    for _ in range(nb)] for k_ in ("aL", "aR", "aO", "sL", "sR")}
        pk = lambda rows: b"".join(o.s2b(v) for r in rows for v in r)       # noqa: E731
        t, wv, h = gpu.r1cs_prover_polys(circ, nb, n, m, pk([ys]), pk([[pow(y, -1, N_) for y in ys]]), pk([zs]),
                                         pk(wit["aL"]), pk(wit["aR"]), pk(wit["aO"]), pk(wit["sL"]), pk(wit["sR"]))
        lv, rv = gpu.r1cs_prover_eval(h, nb, padded, pk([xs]))
        gpu.prover_destroy(h)
        # the same call with every input in ark-ff Montgomery form (x * 2^256 mod n as 32 little-endian bytes)
        ark = lambda rows: b"".join((v * (1 << 256) % N_).to_bytes(32, "little") for r in rows for v in r)       # noqa: E731
        t2, wv2, h2 = gpu.r1cs_prover_polys(circ, nb, n, m, ark([ys]), ark([[pow(y, -1, N_) for y in ys]]), ark([zs]),
                                            ark(wit["aL"]), ark(wit["aR"]), ark(wit["aO"]), ark(wit["sL"]), ark(wit["sR"]), ark=True)
        lv2, rv2 = gpu.r1cs_prover_eval(h2, nb, padded, pk([xs]))
        gpu.prover_destroy(h2)
        assert (t2, wv2, lv2, rv2) == (t, wv, lv, rv)
        for p in range(nb):
            y, z, x = ys[p], zs[p], xs[p]
            wL, wR, wO, wV, _ = s.flatten(o.s2b(z))
            yi = pow(y, -1, N_)
            l1 = [(wit["aL"][p][i] + pow(yi, i, N_) * I(wR, i)) % N_ for i in range(n)]
            l2, l3 = wit["aO"][p], wit["sL"][p]
            r0 = [(I(wO, i) - pow(y, i, N_)) % N_ for i in range(n)]
            r1 = [(pow(y, i, N_) * wit["aR"][p][i] + I(wL, i)) % N_ for i in range(n)]
            r3 = [pow(y, i, N_) * wit["sR"][p][i] % N_ for i in range(n)]
            ip = lambda a, b: sum(u * v for u, v in zip(a, b)) % N_       # noqa: E731
            want_t = [ip(l1, r0), (ip(l1, r1) + ip(l2, r0)) % N_, (ip(l2, r1) + ip(l3, r0)) % N_, (ip(l1, r3) + ip(l3, r1)) % N_,
                      ip(l2, r3), ip(l3, r3)]
            assert [I(t, 6 * p + j) for j in range(6)] == want_t, p
            assert wv[32 * m * p:32 * m * (p + 1)] == wV, p
            want_l = [(x * (l1[i] + x * (l2[i] + x * l3[i]))) % N_ for i in range(n)] + [0] * (padded - n)
            want_r = [(r0[i] + x * (r1[i] + x * (x * r3[i]))) % N_ for i in range(n)] + [(-pow(y, i, N_)) % N_ for i in range(n, padded)]
            assert [I(lv, padded * p + i) for i in range(padded)] == want_l, p
            assert [I(rv, padded * p + i) for i in range(padded)] == want_r, p
    finally:
        gpu.circuit_destroy(circ)
        s.close()


# ------------------------------------------------------------------ IPP prover (lock-step session)
def _ipp_create_gpu(gpu, label, nb, n, Q, Gf, Hf, G, H, shared, a, b, gens=None, w=None):
    """Drive InnerProductProof::create (inner_product_proof.rs:49-193) with the transcript on the host
    (the oracle's Python model of it) and all arithmetic on the GPU."""
    import pymodel as pm
    trs = [pm.Transcript(label) for _ in range(nb)]
    for t in trs:
        t.innerproduct_domain_sep(n)
    s = gpu.ipp_begin_gens(gens, nb, n, w, Gf, Hf, a, b) if gens is not None else gpu.ipp_begin(nb, n, Q, Gf, Hf, G, H, shared, a, b)
    Ls, Rs, chs = [b""] * nb, [b""] * nb, [b""] * nb
    try:
        while gpu.ipp_len(s) > 1:
            L, R = gpu.ipp_round(s, nb)
            u = b""
            for p, t in enumerate(trs):
                t.append_message(b"L", L[64 * p:64 * p + 64])
                t.append_message(b"R", R[64 * p:64 * p + 64])
                up = pm.s2b(t.challenge_scalar(b"u"))
                u += up
                Ls[p] += L[64 * p:64 * p + 64]
                Rs[p] += R[64 * p:64 * p + 64]
                chs[p] += up
            gpu.ipp_fold(s, u, gpu.batch_inverse(u))
        aa, bb = gpu.ipp_finish(s, nb)
    finally:
        gpu.ipp_destroy(s)
    return Ls, Rs, aa, bb, chs


@pytest.mark.parametrize("literal", [0, 1, 2])
def test_ipp_create_golden(gpu, golden_ipp, opts, literal):
    """InnerProductProof::create of ONE proof over arbitrary generators: through tables built for the session (default: rounds are
    table lookups) and through the literal generator-folding schedule (option ipp_literal = 1) -- the same bytes."""
    sys_path_oracle()
    if literal == 2:            # above the bound of the table route (ipp_table_max_n): the O(n)-memory literal schedule takes over
        opts(ipp_table_max_n=1)
    else:
        opts(ipp_literal=literal)
    for c in golden_ipp["create"]:
        n = c["n"]
        Gp, Hp = o.gens("G", n), o.gens("H", n)
        Ls, Rs, aa, bb, chs = _ipp_create_gpu(gpu, H(c["label"]), 1, n, H(c["Q"]), J(c, "G_factors"), J(c, "H_factors"),
                                              Gp, Hp, True, J(c, "a"), J(c, "b"))
        assert (Ls[0], Rs[0], aa, bb, chs[0]) == (J(c, "L"), J(c, "R"), H(c["a_out"]), H(c["b_out"]), J(c, "challenges"))


@pytest.mark.parametrize("shared", [True, False])
def test_ipp_create_batched(gpu, shared):
    sys_path_oracle()
    nb, n = 3, 16
    Gp, Hp = o.gens("G", n), o.gens("H", n)
    a, b = o.random_scalars(21, nb * n), o.random_scalars(22, nb * n)
    Gf, Hf = o.random_scalars(23, nb * n), o.random_scalars(24, nb * n)
    Q = b"".join(o.point_mul(o.random_scalars(30 + p, 1), o.generator()) for p in range(nb))
    if shared:
        G, Hh = Gp, Hp
    else:   # per-proof generator sets (rotations of the chain)
        G = b"".join(Gp[64 * p:] + Gp[:64 * p] for p in range(nb))
        Hh = b"".join(Hp[64 * p:] + Hp[:64 * p] for p in range(nb))
    Ls, Rs, aa, bb, _ = _ipp_create_gpu(gpu, b"innerproducttest", nb, n, Q, Gf, Hf, G, Hh, shared, a, b)
    for p in range(nb):
        sl = slice(32 * n * p, 32 * n * (p + 1))
        Gq = Gp if shared else G[64 * n * p:64 * n * (p + 1)]
        Hq = Hp if shared else Hh[64 * n * p:64 * n * (p + 1)]
        L, R, ao, bo, _ = o.ipp_create(b"innerproducttest", n, Q[64 * p:64 * p + 64], Gf[sl], Hf[sl], Gq, Hq, a[sl], b[sl])
        assert (Ls[p], Rs[p], aa[32 * p:32 * p + 32], bb[32 * p:32 * p + 32]) == (L, R, ao, bo)


@pytest.mark.parametrize("n,cap,c", [(16, 16, 8), (8, 16, 4), (1024, 1024, 4), (1, 4, 8)])
def test_ipp_create_resident_generators(gpu, n, cap, c):
    """bpgpu_ipp_begin_gens: the IPP over resident generator tables (L, R as fixed-base MSMs over the ORIGINAL
    generators, generator folding replaced by coefficient updates) yields the oracle's proof bytes."""
    sys_path_oracle()
    nb = 3 if n <= 16 else 2
    Gp, Hp, B = o.gens("G", cap), o.gens("H", cap), o.generator()
    g = gpu.gens_create(Gp, Hp, B, B, c)
    try:
        a, b = o.random_scalars(41, nb * n), o.random_scalars(42, nb * n)
        Gf, Hf = o.random_scalars(43, nb * n), o.random_scalars(44, nb * n)
        w = o.random_scalars(45, nb)
        Ls, Rs, aa, bb, _ = _ipp_create_gpu(gpu, b"innerproducttest", nb, n, None, Gf, Hf, None, None, True, a, b, gens=g, w=w)
        for p in range(nb):
            sl = slice(32 * n * p, 32 * n * (p + 1))
            Q = o.point_mul(w[32 * p:32 * p + 32], B)
            L, R, ao, bo, _ = o.ipp_create(b"innerproducttest", n, Q, Gf[sl], Hf[sl], Gp[:64 * n], Hp[:64 * n], a[sl], b[sl])
            assert (Ls[p], Rs[p], aa[32 * p:32 * p + 32], bb[32 * p:32 * p + 32]) == (L, R, ao, bo)
    finally:
        gpu.gens_destroy(g)


@pytest.mark.parametrize("nb,n", [(11, 32), (70, 16)])
def test_ipp_rounds_of_many_provers_grouped_table_walk(gpu, nb, n):
    """From 8 provers on the round MSMs of a resident-generator IPP take k_fixed_msm_ipp_g (a wave = 8 MSMs x 8 pair-lanes).
    11 provers (a unit with clamped lanes), n = 32, and 70 provers (eight full units and one with six live lanes), n = 16: L, R of
    every round and the final a, b equal the oracle's."""
    sys_path_oracle()
    cap, c = 32, 8
    Gp, Hp, B = o.gens("G", cap), o.gens("H", cap), o.generator()
    g = gpu.gens_create(Gp, Hp, B, B, c)
    try:
        a, b = o.random_scalars(141, nb * n), o.random_scalars(142, nb * n)
        Gf, Hf = o.random_scalars(143, nb * n), o.random_scalars(144, nb * n)
        w = o.random_scalars(145, nb)
        Ls, Rs, aa, bb, _ = _ipp_create_gpu(gpu, b"innerproducttest", nb, n, None, Gf, Hf, None, None, True, a, b, gens=g, w=w)
        for p in range(nb):
            sl = slice(32 * n * p, 32 * n * (p + 1))
            Q = o.point_mul(w[32 * p:32 * p + 32], B)
            L, R, ao, bo, _ = o.ipp_create(b"innerproducttest", n, Q, Gf[sl], Hf[sl], Gp[:64 * n], Hp[:64 * n], a[sl], b[sl])
            assert (Ls[p], Rs[p], aa[32 * p:32 * p + 32], bb[32 * p:32 * p + 32]) == (L, R, ao, bo), p
    finally:
        gpu.gens_destroy(g)


def sys_path_oracle():
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    if d not in sys.path:
        sys.path.insert(0, d)


# ------------------------------------------------------------------ large MSM (bucket method)
@pytest.mark.parametrize("route", ["default", "pip2", "bucket"])
def test_msm_pippenger_edge_cases(gpu, opts, route):
    sys_path_oracle()
    """zero / one / n-1 scalars, identity points, duplicates, P and -P pairs must come out exactly as the oracle's, through
    each of the three MSM routes (see test_msm_pippenger_sizes)."""
    import pymodel as pm
    if route != "default":
        opts(msm_wp_max=0)
    if route == "pip2":
        opts(msm_pip2_single=1)
    n = 640
    Gp = o.gens("G", n)
    pts = bytearray(Gp)
    sc = bytearray(o.random_scalars(5, n))
    for i in range(0, n, 7):
        pts[64 * i:64 * i + 64] = bytes(64)                       # identity points
    for i in range(1, n, 11):
        sc[32 * i:32 * i + 32] = o.s2b([0, 1, N - 1, 2, N - 2][i % 5])
    for i in range(3, n - 1, 50):                                  # duplicates and opposite pairs
        pts[64 * (i + 1):64 * (i + 2)] = pts[64 * i:64 * i + 64]
    G = o.generator()
    negG = pm.p2b(pm.pt_neg(pm.G))
    pts[64 * 2:64 * 3], pts[64 * 4:64 * 5] = G, negG
    sc[32 * 2:32 * 3] = sc[32 * 4:32 * 5]
    assert gpu.msm(bytes(sc), bytes(pts)) == o.msm(bytes(sc), bytes(pts))
    # everything cancels
    half = o.random_scalars(6, 300)
    neg = b"".join(o.s2b(N - v) for v in o.unscalars(half))
    assert gpu.msm(half + neg, Gp[:64 * 300] * 2) == bytes(64)


@pytest.mark.parametrize("route", ["default", "pip2", "bucket"])
@pytest.mark.parametrize("n", [512, 5000, 98347])
def test_msm_pippenger_sizes(gpu, opts, n, route):
    """98 347 = the C4 verification MSM size.  Checked against the oracle-free identity
    MSM(s_i, k_i G) = (sum s_i k_i) G and, for the smaller sizes, the oracle's own Pippenger.  Routes: default = the
    window-parallel launches up to 2^15 terms, k_pip.hip above; pip2 = k_pip2.hip's one-instance pipeline (2^8..2^16 terms);
    bucket = k_pip.hip / the Straus lanes at every size."""
    if route != "default":
        opts(msm_wp_max=0)
    if route == "pip2":
        opts(msm_pip2_single=1)
    Gp, Gd = o.gens("G", n, dlogs=True)
    sc = o.random_scalars(1234 + n, n)
    got = gpu.msm(sc, Gp)
    assert got == o.point_mul(o.inner_product(sc, Gd), o.generator())
    if n <= 5000:
        assert got == o.msm(sc, Gp, 2)


def test_msm_pippenger_two_level_sort_edge_cases(gpu):
    """n >= 2^15 takes the LDS-staged two-level counting sort (coarse bins per tile, fine sort per bin): skewed and
    degenerate key distributions -- all scalars equal (one bucket per window holds everything), 0 / 1 / n-1 scalars,
    bit-valued scalars, identity points, duplicate and opposite points, a ragged last tile -- against the oracle-free
    identity MSM(s_i, k_i G) = (sum s_i k_i) G."""
    n = 40000 + 37                                                  # not a multiple of the 2048-key tile
    Gp, Gd = o.gens("G", n, dlogs=True)
    G = o.generator()
    ks = o.unscalars(Gd)

    def check(sc_list, pts=Gp, dl=ks):
        sc = o.scalars(sc_list)
        want = o.point_mul(o.s2b(sum(a * b for a, b in zip(sc_list, dl)) % N), G)
        assert gpu.msm(sc, pts) == want

    rnd = o.unscalars(o.random_scalars(31337, n))
    check([rnd[0]] * n)                                             # every key of a window in ONE bucket
    check([i & 1 for i in range(n)])                                # bits: digit 1 in window 0 only
    check([[0, 1, N - 1, 2, N - 2, rnd[i]][i % 6] for i in range(n)])
    check([rnd[i] >> 200 for i in range(n)])                        # small scalars: the upper windows are empty
    # identity points, duplicates, opposite pairs
    pts, dl = bytearray(Gp), list(ks)
    for i in range(0, n, 13):
        pts[64 * i:64 * i + 64] = bytes(64)
        dl[i] = 0
    for i in range(5, n - 1, 101):
        pts[64 * (i + 1):64 * (i + 2)] = pts[64 * i:64 * i + 64]
        dl[i + 1] = dl[i]
    sc = list(rnd)
    for i in range(7, n - 1, 211):                                  # s P + (n - s) P = 0 on a duplicated point
        pts[64 * (i + 1):64 * (i + 2)] = pts[64 * i:64 * i + 64]
        dl[i + 1] = dl[i]
        sc[i + 1] = N - sc[i]
    check(sc, bytes(pts), dl)


def test_msm_batch_pippenger_two_level(gpu):
    """two instances of 33 000 terms each through the batched entry point (segments = instance x window)"""
    nb, n = 2, 33000
    Gp, Gd = o.gens("H", n, dlogs=True)
    sc = o.random_scalars(78, nb * n)
    got = gpu.msm_batch(nb, n, sc, Gp * nb)
    for b in range(nb):
        assert got[64 * b:64 * b + 64] == o.point_mul(o.inner_product(sc[32 * n * b:32 * n * (b + 1)], Gd), o.generator())


def test_msm_batch_pippenger(gpu):
    nb, n = 2, 700
    Gp = o.gens("H", n)
    sc = o.random_scalars(77, nb * n)
    assert gpu.msm_batch(nb, n, sc, Gp * nb) == o.msm_batch(sc, Gp * nb, nb, n)


# ------------------------------------------------------------------ combined batch check
@pytest.mark.parametrize("tamper", [set(), {3}])
def test_combined_batch_check(gpu, tamper):
    """sum_p rho_p * mega_check_p as one point: equals the oracle's weighted sum of the per-proof
    mega_check points (identity iff every proof is valid)."""
    n_bits, nb = 8, 9
    recs, cap = bh.make_range_batch(n_bits, nb, tamper=tamper)
    sessions = [o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) for proof, com in recs]
    s0 = sessions[0]
    rp, kind, idx, coeff = s0.csr()
    circ = gpu.circuit_create(rp, kind, idx, coeff, s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    try:
        pts = sc = ch = b""
        for (proof, com), s in zip(recs, sessions):
            k, p, q = bh.verify_inputs(proof, com)
            pts, sc, ch = pts + p, sc + q, ch + s.challenges()
        rho = o.random_scalars(4242, nb)
        got = gpu.r1cs_verify_combined(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch, rho)
        want = o.msm(rho, b"".join(s.mega_check() for s in sessions), 1)
        assert got == want
        assert (got == bytes(64)) == (not tamper)
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)


def test_ipp_create_bucket_rounds(gpu):
    """n = 512: the first rounds (2h + 1 >= 257 terms) use the batched bucket-method MSM."""
    sys_path_oracle()
    nb, n = 2, 512
    Gp, Hp = o.gens("G", n), o.gens("H", n)
    a, b = o.random_scalars(41, nb * n), o.random_scalars(42, nb * n)
    Gf, Hf = o.scalars([1] * (nb * n)), o.random_scalars(44, nb * n)
    Q = b"".join(o.point_mul(o.random_scalars(50 + p, 1), o.generator()) for p in range(nb))
    Ls, Rs, aa, bb, _ = _ipp_create_gpu(gpu, b"innerproducttest", nb, n, Q, Gf, Hf, Gp, Hp, True, a, b)
    for p in range(nb):
        sl = slice(32 * n * p, 32 * n * (p + 1))
        L, R, ao, bo, _ = o.ipp_create(b"innerproducttest", n, Q[64 * p:64 * p + 64], Gf[sl], Hf[sl], Gp, Hp, a[sl], b[sl])
        assert (Ls[p], Rs[p], aa[32 * p:32 * p + 32], bb[32 * p:32 * p + 32]) == (L, R, ao, bo)


def test_msm_gens_capacity_1024(gpu):
    """capacity 1024 (BASELINE config 3: n = 1024 multipliers): 2050 resident generators."""
    cap = 1024
    Gp, Hp, B = o.gens("G", cap), o.gens("H", cap), o.generator()
    g = gpu.gens_create(Gp, Hp, B, B, 8)
    try:
        sc = o.random_scalars(99, 2 + 2 * cap)
        assert gpu.msm_gens(g, 1, cap, sc) == o.msm(sc, B + B + Gp + Hp)
    finally:
        gpu.gens_destroy(g)


# ------------------------------------------------------------------ BASELINE full size (configs[1])
def test_verify_batch_full_size_1024x64bit(gpu):
    """1024 x 64-bit range-gadget verifications in one call (BASELINE.json configs[1]).  48 distinct oracle
    proofs are tiled to 1024 slots and a known pattern of slots is tampered: the accept bits must follow the
    pattern exactly, every mega_check must equal its source proof's, and the combined batch check must be
    the identity iff no slot is tampered."""
    n_bits, distinct, nb = 64, 48, 1024
    recs, cap = bh.make_range_batch(n_bits, distinct, seed0=7000)
    sessions = [o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) for proof, com in recs]
    s0 = sessions[0]
    assert (s0.nterms, s0.k, s0.q) == (154, 6, 129)          # 154-term MSM, 129 constraints
    rp, kind, idx, coeff = s0.csr()
    circ = gpu.circuit_create(rp, kind, idx, coeff, 64, 1)
    g = _gens(gpu, cap, 8)
    try:
        base = []
        for (proof, com), s in zip(recs, sessions):
            k, p, q = bh.verify_inputs(proof, com)
            base.append((p, q, s.challenges(), s.mega_check()))
        bad = {i for i in range(nb) if i % 97 == 5}
        pts = sc = ch = b""
        for i in range(nb):
            p, q, c, _ = base[i % distinct]
            if i in bad:
                q = bytes([q[0] ^ 1]) + q[1:]               # flip a bit of t_x
            pts, sc, ch = pts + p, sc + q, ch + c
        ok, mega, _ = gpu.r1cs_verify_batch(g, circ, nb, 64, 6, 1, pts, sc, ch, want_mega=True)
        assert ok == [0 if i in bad else 1 for i in range(nb)]
        for i in range(nb):
            if i not in bad:
                assert mega[64 * i:64 * i + 64] == bytes(64)
        # one tampered slot, checked against the oracle's mega_check for the same tampered proof
        i = min(bad)
        proof, com = recs[i % distinct]
        tp = bytearray(proof)
        tp[8 + 11 * 64] ^= 1
        st = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, bytes(tp), cap)
        # (the tampered t_x changes the transcript; the GPU was given the ORIGINAL challenges, so compare through the MSM)
        sc_t, pts_t = sessions[i % distinct].msm_terms()
        assert st.rc == -1 and mega[64 * i:64 * i + 64] != bytes(64)
        # combined check: identity for the clean batch, not for the tampered one
        rho = o.random_scalars(31337, nb)
        clean_sc = b"".join(base[j % distinct][1] for j in range(nb))
        assert gpu.r1cs_verify_combined(g, circ, nb, 64, 6, 1, pts, clean_sc, ch, rho) == bytes(64)
        assert gpu.r1cs_verify_combined(g, circ, nb, 64, 6, 1, pts, sc, ch, rho) != bytes(64)
        # a batch size that fills neither the last 4-proof block of the fixed-base lanes nor the last table wave,
        # over 16-bit-window tables (the bench uses 20-bit ones: same kernels, other template instance)
        nb2 = 1027
        g16 = _gens(gpu, cap, 16)
        try:
            p2 = pts + pts[:3 * len(base[0][0])]
            s2 = sc + sc[:3 * 160]
            c2 = ch + ch[:3 * len(base[0][2])]
            ok2, mega2, _ = gpu.r1cs_verify_batch(g16, circ, nb2, 64, 6, 1, p2, s2, c2, want_mega=True)
            assert ok2 == ok + ok[:3] and mega2 == mega + mega[:3 * 64]
        finally:
            gpu.gens_destroy(g16)
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)


@pytest.mark.parametrize("chunk_gens", [0, -1])
def test_verify_batch_c20_benchmarked_instance(gpu, opts, chunk_gens):
    """The template instances bench.py times -- 20-bit-window generator tables, nb >= 1024: the generator half a proof per lane
    (k_verify_back_q<20, 1> + k_verify_verdict_q, the default from 256 proofs on) and, forced by the option, 16 lanes per
    fixed-base MSM (k_verify_back<20, 16> + k_verify_verdict); k_verify_front, k_verify_windows -- on a batch with tampered
    proofs: accept bits, the non-identity mega_check points and all MSM scalars of every slot equal the oracle's.  Capacity 8
    keeps the table at 18 generators x 13 windows x 2^19 rows x 64 B = 7.9 GB.  Also nb = 70 (32 lanes per MSM) in
    test_range_verify_batch[8-70-20]."""
    opts(fixed_chunk_gens=chunk_gens)
    n_bits, distinct, nb = 8, 24, 1030
    recs, cap = bh.make_range_batch(n_bits, distinct, seed0=9100)
    variants = []          # (points, scalars, challenges, ok, mega, full) for clean and tampered versions of each proof
    for t in range(2):
        for i, (proof, com) in enumerate(recs):
            if t:
                bad = bytearray(proof)
                bad[8 + 11 * 64 + (i % 3) * 32] ^= 1 + (i % 7)       # t_x / t_x_blinding / e_blinding
                proof = bytes(bad)
            s = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap)
            k, p, q = bh.verify_inputs(proof, com)
            assert (s.rc == 0) == (t == 0)
            variants.append((p, q, s.challenges(), 1 if s.rc == 0 else 0, s.mega_check(), s.msm_terms()[0]))
            if t == 0 and i == 0:
                s0 = s
                rp, kind, idx, coeff = s.csr()
            else:
                s.close()
    circ = gpu.circuit_create(rp, kind, idx, coeff, s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 20)
    try:
        pick = [(i % distinct) + (distinct if i % 41 == 7 or i == nb - 1 else 0) for i in range(nb)]
        pts = b"".join(variants[j][0] for j in pick)
        sc = b"".join(variants[j][1] for j in pick)
        ch = b"".join(variants[j][2] for j in pick)
        ok, mega, full = gpu.r1cs_verify_batch(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch, True, True)
        nt = s0.nterms
        for i, j in enumerate(pick):
            assert ok[i] == variants[j][3], i
            assert mega[64 * i:64 * i + 64] == variants[j][4], i
            assert full[32 * nt * i:32 * nt * (i + 1)] == variants[j][5], i
        assert sum(ok) == nb - len([i for i in range(nb) if i % 41 == 7 or i == nb - 1])
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        s0.close()


@pytest.mark.parametrize("nb", [1024, 1027])
def test_verify_batch_bench_instance_64bit_c20_default_mode(gpu, nb):
    """EXACTLY the composite bench.py times: the 64-bit gadget (padded n = 64, k = 6 -> the vs_prep fast path +
    k_verify_scalars_fast), 20-bit-window tables of the 130 generators (57 GB), nb >= 1024 in the default mode
    (k_verify_front<4> -> k_verify_scalars_fast -> k_verify_windows -> k_verify_horner_groups -> k_verify_back_q<20,1> ->
    k_verify_verdict_q), on a batch that holds tampered proofs: accept bits, every mega_check point (identity for the valid
    proofs, the oracle's non-identity point for the tampered ones) and all 154 MSM scalars of every slot in
    verifier.rs:517-532 order equal the oracle's.  nb = 1027 leaves the last 4-proof block / table wave partly empty."""
    n_bits, distinct = 64, 20
    recs, cap = bh.make_range_batch(n_bits, distinct, seed0=9400)
    variants = []
    for t in range(2):
        for i, (proof, com) in enumerate(recs):
            if t:
                bad = bytearray(proof)
                bad[8 + 11 * 64 + (i % 3) * 32] ^= 1 + (i % 7)       # t_x / t_x_blinding / e_blinding
                proof = bytes(bad)
            s = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap)
            k, p, q = bh.verify_inputs(proof, com)
            assert (s.rc == 0) == (t == 0)
            variants.append((p, q, s.challenges(), 1 if s.rc == 0 else 0, s.mega_check(), s.msm_terms()[0]))
            if t == 0 and i == 0:
                s0 = s
                rp, kind, idx, coeff = s.csr()
            else:
                s.close()
    assert (s0.nterms, s0.k, s0.n1) == (154, 6, 64)
    circ = gpu.circuit_create(rp, kind, idx, coeff, s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 20)
    try:
        tam = {i for i in range(nb) if i % 53 == 11 or i in (0, 1023, nb - 1)}
        pick = [(i % distinct) + (distinct if i in tam else 0) for i in range(nb)]
        pts = b"".join(variants[j][0] for j in pick)
        sc = b"".join(variants[j][1] for j in pick)
        ch = b"".join(variants[j][2] for j in pick)
        ok, mega, full = gpu.r1cs_verify_batch(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch, True, True)
        nt = s0.nterms
        for i, j in enumerate(pick):
            assert ok[i] == variants[j][3], i
            assert mega[64 * i:64 * i + 64] == variants[j][4], i
            assert (mega[64 * i:64 * i + 64] == bytes(64)) == (i not in tam), i
            assert full[32 * nt * i:32 * nt * (i + 1)] == variants[j][5], i
        assert ok == [0 if i in tam else 1 for i in range(nb)]
        # ... and the call bench.py makes (operands resident, accept bits only: no mega_check / scalar outputs requested)
        d_p, d_s, d_c, d_o = gpu.to_device(pts), gpu.to_device(sc), gpu.to_device(ch), gpu.malloc(4 * nb)
        try:
            gpu.r1cs_verify_batch_dev(g, circ, nb, s0.n1, s0.k, d_p, d_s, d_c, d_o)
            gpu.sync()
            assert gpu.download(d_o, 4 * nb) == b"".join(v.to_bytes(4, "little") for v in ok)
        finally:
            for d in (d_p, d_s, d_c, d_o):
                gpu.free(d)
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        s0.close()


# ------------------------------------------------------------------ device-side transcript
def test_verify_with_device_transcript(gpu):
    """The whole of Verifier::verify for a 1-phase circuit on the device: the challenges the GPU derives
    (keccak256 hash chain, hash_to_scalar) equal the oracle's transcript replay; accept bits and mega_check
    points equal the host-transcript path; identity points are rejected where the reference validates."""
    sys_path_oracle()
    import pymodel as pm
    for n_bits, nb in ((8, 6), (64, 4)):
        tamper = {1}
        recs, cap = bh.make_range_batch(n_bits, nb, seed0=4000, tamper=tamper)
        sessions = [o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) for proof, com in recs]
        s0 = sessions[0]
        rp, kind, idx, coeff = s0.csr()
        circ = gpu.circuit_create(rp, kind, idx, coeff, s0.n1 + s0.n2, s0.m)
        g = _gens(gpu, cap, 8)
        try:
            init = pm.Transcript(b"RangeProofTest").state * nb
            pts = sc = b""
            for proof, com in recs:
                k, p, q = bh.verify_inputs(proof, com)
                pts, sc = pts + p, sc + q
            ok, mega, ch = gpu.r1cs_verify_batch_fs(g, circ, nb, s0.n1, s0.k, s0.m, init, pts, sc)
            for i, s in enumerate(sessions):
                assert ch[32 * (6 + s.k) * i:32 * (6 + s.k) * (i + 1)] == s.challenges(), i
                assert ok[i] == (1 if s.rc == 0 else 0) == (0 if i in tamper else 1)
                assert mega[64 * i:64 * i + 64] == s.mega_check()
            # identity A_I1 / T_1 / L_0: VerificationError from validate_and_append_point
            nvar = 11 + s0.m + 2 * s0.k
            for slot in (0, 6 + s0.m, 11 + s0.m):
                bad = bytearray(pts)
                bad[64 * slot:64 * slot + 64] = bytes(64)          # proof 0
                ok2, _, _ = gpu.r1cs_verify_batch_fs(g, circ, nb, s0.n1, s0.k, s0.m, init, bytes(bad), sc)
                assert ok2[0] == 0 and ok2[2:] == ok[2:]
                kk, p0, q0 = bh.verify_inputs(*recs[0])
                flat = bytearray(recs[0][0])
                off = {0: 8, 6 + s0.m: 8 + 6 * 64, 11 + s0.m: 8 + 11 * 64 + 96}[slot]
                flat[off:off + 64] = bytes(64)
                assert o.r1cs_verify(o.K_RANGE, n_bits, b"RangeProofTest", [], recs[0][1], bytes(flat), cap) == -1
            assert nvar * 64 * nb == len(pts)
        finally:
            gpu.gens_destroy(g)
            gpu.circuit_destroy(circ)


def test_whole_verify_differential_fuzz(gpu):
    """Random byte flips ANYWHERE in the proof or the commitment (points knocked off the curve, non-canonical coordinates and
    scalars, identity points, flipped transcript material): the GPU's whole Verifier::verify (device transcript + verification,
    one batch) returns the oracle's verdict for every proof.  tools/fuzz_verify.py is the long form of this test."""
    import random
    sys_path_oracle()
    import pymodel as pm
    rnd = random.Random(9001)
    for n_bits, nb in ((8, 96), (64, 40)):
        cap = 1 << max(0, (n_bits - 1).bit_length())
        recs = []
        for i in range(nb):
            rc, proof, com = o.r1cs_prove(o.K_RANGE, n_bits, b"RangeProofTest", [rnd.getrandbits(n_bits)], rnd.getrandbits(40), cap)
            assert rc == 0
            proof, com = bytearray(proof), bytearray(com)
            mode = rnd.randrange(8)
            if mode >= 2:
                for _ in range(rnd.choice((1, 1, 2, 3))):
                    if mode == 7:
                        com[rnd.randrange(len(com))] ^= 1 << rnd.randrange(8)
                    elif mode == 6:
                        slot = rnd.randrange(11)
                        proof[8 + 64 * slot:8 + 64 * slot + 64] = bytes(64)
                    elif mode == 5:
                        proof[8 + 32 * rnd.randrange((len(proof) - 8) // 32) + 31] |= 0xF0
                    else:
                        proof[8 + rnd.randrange(len(proof) - 8)] ^= 1 << rnd.randrange(8)
            recs.append((bytes(proof), bytes(com)))
        good = o.r1cs_prove(o.K_RANGE, n_bits, b"RangeProofTest", [1], 7, cap)
        s0 = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], good[2], good[1], cap)
        circ = gpu.circuit_create(*s0.csr(), s0.n1 + s0.n2, s0.m)
        g = _gens(gpu, cap, 8)
        try:
            pts = sc = b""
            for proof, com in recs:
                k, p, q = bh.verify_inputs(proof, com)
                pts, sc = pts + p, sc + q
            ok, _, _ = gpu.r1cs_verify_batch_fs(g, circ, nb, s0.n1, s0.k, s0.m, pm.Transcript(b"RangeProofTest").state * nb, pts, sc, want_mega=False)
            want = [1 if o.r1cs_verify(o.K_RANGE, n_bits, b"RangeProofTest", [], com, proof, cap) == 0 else 0 for proof, com in recs]
            assert ok == want
            assert 0 < sum(want) < nb          # the sample holds accepted and rejected proofs
        finally:
            gpu.gens_destroy(g)
            gpu.circuit_destroy(circ)
            s0.close()


def test_two_phase_circuits_batched_with_device_transcript(gpu):
    """SURVEY 8f N1 for circuits with RANDOMIZED constraints (verifier.rs:366-385): four k-shuffle proofs with different
    inputs -- hence four different gadget challenges z -- verified in ONE batch against ONE parametric circuit (coefficient
    = c0 + chi * c1: the shuffle gadget's (x_i - z) terms, tests/r1cs.rs:23-62).  With the transcript on the device
    (bpgpu_r1cs_verify_batch_fs2: phase separator + challenge_scalar("shuffle challenge") inside the schedule) every
    challenge equals the oracle's replay, -chi equals the oracle's numeric `One` coefficients, and accept bits, mega_check
    points (one proof tampered) and all MSM scalars equal the oracle's; the same through bpgpu_r1cs_verify_batch_param with
    host-supplied challenges."""
    sys_path_oracle()
    import pymodel as pm
    import random
    k_sh, nb, cap = 6, 4, 16
    rnd = random.Random(2025)
    recs, sessions = [], []
    for p in range(nb):
        x = [rnd.getrandbits(40) for _ in range(k_sh)]
        y = list(x)
        rnd.shuffle(y)
        rc, proof, com = o.r1cs_prove(o.K_SHUFFLE, k_sh, b"ShuffleProofTest", x + y, 300 + p, cap)
        assert rc == 0
        if p == 2:
            bad = bytearray(proof)
            bad[8 + 11 * 64 + 64] ^= 1                      # e_blinding
            proof = bytes(bad)
        recs.append((proof, com))
        sessions.append(o.VerifySession(o.K_SHUFFLE, k_sh, b"ShuffleProofTest", [], com, proof, cap))
    s0 = sessions[0]
    n, m, q = s0.n1 + s0.n2, s0.m, s0.q
    assert s0.n1 == 0 and n == 2 * (k_sh - 1) and q == 2 * n + 1 and m == 2 * k_sh
    # parametric CSR from the numeric one: in the shuffle circuit every `One` term is a (-z) (kind 4: linear_combination.rs:15-28)
    rp, kd, ix, cf = s0.csr()
    rows0, rows1 = [[] for _ in range(q)], [[] for _ in range(q)]
    for r in range(q):
        for t in range(rp[r], rp[r + 1]):
            if kd[t] == 4:
                rows1[r].append((4, 0, (o.N - 1).to_bytes(32, "little")))     # chi * (-1)
            else:
                rows0[r].append((kd[t], ix[t], cf[32 * t:32 * t + 32]))
    prp, pkd, pix, pcf = [0], [], [], b""
    for row in rows0 + rows1:
        for a, b, c in row:
            pkd.append(a)
            pix.append(b)
            pcf += c
        prp.append(len(pkd))
    circ = gpu.circuit_create_param(q, 1, prp, pkd, pix, pcf, n, m)
    g = _gens(gpu, cap, 8)
    try:
        t = pm.Transcript(b"ShuffleProofTest")
        t.append_message(b"dom-sep", b"ShuffleProof")           # tests/r1cs.rs:80-81
        t.append_u64(b"k", k_sh)
        init = t.state * nb
        pts = sc = b""
        for proof, com in recs:
            kk, p_, q_ = bh.verify_inputs(proof, com)
            pts, sc = pts + p_, sc + q_
        ok, mega, ch, chi = gpu.r1cs_verify_batch_fs2(g, circ, nb, 0, s0.k, m, 1, init, b"shuffle challenge", pts, sc)
        chis = set()
        for i, s in enumerate(sessions):
            assert ch[32 * (6 + s.k) * i:32 * (6 + s.k) * (i + 1)] == s.challenges(), i
            z = int.from_bytes(chi[32 * i:32 * i + 32], "little")
            rp_i, kd_i, ix_i, cf_i = s.csr()
            ones = {cf_i[32 * t_:32 * t_ + 32] for t_ in range(len(kd_i)) if kd_i[t_] == 4}
            assert ones == {((o.N - z) % o.N).to_bytes(32, "little")}, i                      # the oracle's rows carry -z
            chis.add(z)
            assert ok[i] == (1 if s.rc == 0 else 0) == (0 if i == 2 else 1), i
            assert mega[64 * i:64 * i + 64] == s.mega_check(), i
        assert len(chis) == nb
        allch = b"".join(s.challenges() for s in sessions)
        ok2, mega2, full2 = gpu.r1cs_verify_batch_param(g, circ, nb, 0, s0.k, m, pts, sc, allch, chi, True, True)
        assert ok2 == ok and mega2 == mega
        for i, s in enumerate(sessions):
            assert full2[32 * s.nterms * i:32 * s.nterms * (i + 1)] == s.msm_terms()[0], i
        # a parametric circuit without its gadget challenges is refused
        import mpc_bulletproof_amd as m_
        with pytest.raises(m_.BpGpuError):
            gpu.r1cs_verify_batch(g, circ, nb, 0, s0.k, m, pts, sc, allch)
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)
        for s in sessions:
            s.close()


# ------------------------------------------------------------------ wire codec
def test_points_codec_golden(gpu, golden_codec):
    """32-byte compressed points <-> affine: the model's vectors (signs, identity) both ways, and the rejected
    encodings (x off the curve with either sign flag, x >= p, both flags)."""
    xy = b"".join(H(r["xy"]) for r in golden_codec["points"])
    comp = b"".join(H(r["compressed"]) for r in golden_codec["points"])
    assert gpu.points_compress(xy) == comp
    got, ok = gpu.points_decompress(comp)
    assert got == xy and ok == [1] * len(golden_codec["points"])
    bad = b"".join(H(b) for b in golden_codec["invalid"])
    got, ok = gpu.points_decompress(bad + comp[:32])
    assert ok == [0] * len(golden_codec["invalid"]) + [1]
    assert got[:64 * len(golden_codec["invalid"])] == bytes(64 * len(golden_codec["invalid"]))
    import mpc_bulletproof_amd as m
    off = bytearray(xy[:64])
    off[0] ^= 1
    with pytest.raises(m.BpGpuError) as e:
        gpu.points_compress(bytes(off))
    assert e.value.code == m.lib.E_ARG


def test_points_codec_roundtrip_many(gpu):
    """decompress(compress(P)) == P for 4096 generator-chain points and their negatives (size-independent property;
    every decompression is a square root in F_p through the 24-digit discrete logarithm)."""
    sys_path_oracle()
    import pymodel as pm
    n = 2048
    pts = o.gens("G", n // 2) + o.gens("H", n // 2)
    neg = b"".join(pts[64 * i:64 * i + 32] + ((pm.P - int.from_bytes(pts[64 * i + 32:64 * i + 64], "little")) % pm.P).to_bytes(32, "little")
                   for i in range(n))
    allp = pts + neg
    comp = gpu.points_compress(allp)
    # a point and its negative share x and differ in the sign flag only
    for i in range(0, n, 97):
        a, b = comp[32 * i:32 * i + 32], comp[32 * (n + i):32 * (n + i) + 32]
        assert a[:31] == b[:31] and (a[31] ^ b[31]) == 0x80
    got, ok = gpu.points_decompress(comp)
    assert ok == [1] * (2 * n) and got == allp


def test_proof_wire_format_golden(golden_codec):
    """R1CSProof::to_bytes / from_bytes through the host mirror (device point codec underneath) against the model's
    wire bytes of the committed golden proofs (1-phase and 2-phase), and the reference's length / version errors."""
    import ctypes as C
    import os
    host = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libbph_capi.so"))
    out, ln = (C.c_uint8 * 8192)(), C.c_size_t(0)
    for rec in golden_codec["proofs"]:
        flat, wire = H(rec["flat"]), H(rec["wire"])
        assert host.bph_proof_flat_to_wire(o._buf(flat), C.c_size_t(len(flat)), out, C.byref(ln)) == 0
        assert bytes(out)[:ln.value] == wire
        assert host.bph_proof_wire_to_flat(o._buf(wire), C.c_size_t(len(wire)), out, C.byref(ln)) == 0
        assert bytes(out)[:ln.value] == flat
    wire = H(golden_codec["proofs"][0]["wire"])
    bad_cases = [b"", bytes([2]) + wire[1:], wire[:-1], wire[:1 + 10 * 32], wire + bytes(32),
                 wire[:1] + H(golden_codec["invalid"][0]) + wire[33:]]
    for bad in bad_cases:
        assert host.bph_proof_wire_to_flat(o._buf(bad if bad else b"\0"), C.c_size_t(len(bad)), out, C.byref(ln)) == -3, bad[:4]


def test_verify_batch_from_wire_format(gpu, golden_codec):
    """bpgpu_r1cs_verify_batch_wire: reference wire-format proofs + compressed commitments -> accept bits entirely on
    the device (unpack, point decompression, transcript, verification).  70 proofs of the 8-bit gadget: valid ones
    accepted, a tampered scalar rejected by the verification, an undecodable point / a wrong version byte rejected
    as the reference's FormatError would."""
    sys_path_oracle()
    import pymodel as pm
    import mpc_bulletproof_amd as m

    def flat_to_dict(b):
        k, pts11, sc3, L, R, ab = bh.parse_flat_proof(b)
        names = ("A_I1", "A_O1", "S1", "A_I2", "A_O2", "S2", "T_1", "T_3", "T_4", "T_5", "T_6")
        p = {nm: pm.b2p(pts11[64 * i:64 * i + 64]) for i, nm in enumerate(names)}
        for i, nm in enumerate(("t_x", "t_x_blinding", "e_blinding")):
            p[nm] = pm.b2s(sc3[32 * i:32 * i + 32])
        p["L_vec"] = [pm.b2p(L[64 * i:64 * i + 64]) for i in range(k)]
        p["R_vec"] = [pm.b2p(R[64 * i:64 * i + 64]) for i in range(k)]
        p["a"], p["b"] = pm.b2s(ab[:32]), pm.b2s(ab[32:])
        return p

    nb, n_bits = 70, 8
    recs, cap = bh.make_range_batch(n_bits, nb, tamper={5})
    s0 = o.VerifySession(o.K_RANGE, n_bits, b"RangeProofTest", [], recs[0][1], recs[0][0], cap)
    rp, kind, idx, coeff = s0.csr()
    circ = gpu.circuit_create(rp, kind, idx, coeff, s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    try:
        wires = [pm.r1cs_proof_to_bytes(flat_to_dict(proof)) for proof, _ in recs]
        coms = [pm.point_compress(pm.b2p(com)) for _, com in recs]
        plen = len(wires[0])
        assert plen == 1 + 11 * 32 + (2 * s0.k + 2) * 32 and all(len(w) == plen for w in wires)
        bad_point = bytearray(wires[9])
        bad_point[1 + 32 * 4:1 + 32 * 5] = H(golden_codec["invalid"][0])       # T_3 := an x off the curve
        wires[9] = bytes(bad_point)
        wires[11] = bytes([1]) + wires[11][1:]                                 # version byte contradicts the length
        big = bytearray(wires[20])                                             # t_x + n encodes the same scalar
        off = 1 + 8 * 32
        big[off:off + 32] = (int.from_bytes(big[off:off + 32], "big") + pm.N).to_bytes(32, "big")
        wires[20] = bytes(big)
        init = pm.Transcript(b"RangeProofTest").state * nb
        ok = gpu.r1cs_verify_batch_wire(g, circ, nb, s0.n1, plen, b"".join(wires), b"".join(coms), init)
        want = [1] * nb
        want[5] = want[9] = want[11] = 0
        assert ok == want
        with pytest.raises(m.BpGpuError) as e:
            gpu.r1cs_verify_batch_wire(g, circ, nb, s0.n1, plen - 32, b"".join(w[:-32] for w in wires), b"".join(coms), init)
        assert e.value.code == m.lib.E_LEN
    finally:
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)


@pytest.mark.parametrize("resident", [True, False])
def test_ipp_rounds_with_device_transcript(gpu, resident):
    """bpgpu_ipp_run_fs: all rounds of InnerProductProof::create with the hash chain on the device give the oracle's
    L, R, a, b (and leave the transcripts where the host's would be), over resident generator tables and in the
    generator-folding mode."""
    sys_path_oracle()
    import pymodel as pm
    nb, n, k = 3, 32, 5
    Gp, Hp, B = o.gens("G", n), o.gens("H", n), o.generator()
    a, b = o.random_scalars(61, nb * n), o.random_scalars(62, nb * n)
    Gf, Hf = o.random_scalars(63, nb * n), o.random_scalars(64, nb * n)
    w = o.random_scalars(65, nb)
    Q = b"".join(o.point_mul(w[32 * p:32 * p + 32], B) for p in range(nb))
    trs = [pm.Transcript(b"innerproducttest") for _ in range(nb)]
    for t in trs:
        t.innerproduct_domain_sep(n)
    states = b"".join(t.state for t in trs)
    g = gpu.gens_create(Gp, Hp, B, B, 8) if resident else None
    s = gpu.ipp_begin_gens(g, nb, n, w, Gf, Hf, a, b) if resident else gpu.ipp_begin(nb, n, Q, Gf, Hf, Gp, Hp, True, a, b)
    try:
        L, R, aa, bb, st_out = gpu.ipp_run_fs(s, nb, k, states)
        for p in range(nb):
            sl = slice(32 * n * p, 32 * n * (p + 1))
            Lo, Ro, ao, bo, ch = o.ipp_create(b"innerproducttest", n, Q[64 * p:64 * p + 64], Gf[sl], Hf[sl], Gp, Hp, a[sl], b[sl])
            assert (L[64 * k * p:64 * k * (p + 1)], R[64 * k * p:64 * k * (p + 1)]) == (Lo, Ro)
            assert (aa[32 * p:32 * p + 32], bb[32 * p:32 * p + 32]) == (ao, bo)
            t = trs[p]
            for r in range(k):
                t.append_message(b"L", Lo[64 * r:64 * r + 64])
                t.append_message(b"R", Ro[64 * r:64 * r + 64])
                assert pm.s2b(t.challenge_scalar(b"u")) == ch[32 * r:32 * r + 32]
            assert st_out[32 * p:32 * p + 32] == t.state
    finally:
        gpu.ipp_destroy(s)
        if g is not None:
            gpu.gens_destroy(g)


@pytest.mark.parametrize("kind,param,values_fn,nverify", [
    ("multi", 8 | (2 << 16), lambda i: [(37 * i + 5) % 256, (91 * i + 200) % 256], 0),      # m = 2, n = 16, k = 4 -> 21 points
    ("multi", 4 | (3 << 16), lambda i: [i % 16, (5 * i + 3) % 16, (11 * i + 7) % 16], 0),   # m = 3, n = 12 -> padded 16, 22 points
    ("example", 0, lambda i: [3 + i, 4, 6, 1, 0, 9], 1),        # m = 5, n = 1, k = 0 -> 16 points (c2 = 9 is public: one circuit)
])
def test_verify_batch_other_circuits_all_launch_variants(gpu, opts, kind, param, values_fn, nverify):
    """Batches of 70 proofs of circuits with several commitments / padding / k = 0 through the window-parallel, fused
    and separate launch paths: accept bits, mega_check points and MSM scalars as the oracle's (tampered included)."""
    nb = 70
    okind = o.K_RANGE_MULTI if kind == "multi" else o.K_EXAMPLE
    label = b"RangeProofTest" if kind == "multi" else b"ExampleGadget"
    cap = 16
    recs = []
    for i in range(nb):
        vals = values_fn(i)
        if kind == "example":     # (a1 + a2)(b1 + b2) = c1 + c2 with c2 public
            a1, a2, b1, b2 = vals[0], vals[1], vals[2], vals[3]
            c2 = vals[5]
            vals = [a1, a2, b1, b2, (a1 + a2) * (b1 + b2) - c2, c2]
        rc, proof, com = o.r1cs_prove(okind, param, label, vals, 4000 + i, cap)
        assert rc == 0
        if i in (2, 41):
            bad = bytearray(proof)
            bad[8 + 64 * 11 + 5] ^= 1      # t_x
            proof = bytes(bad)
        recs.append((proof, com, vals[-nverify:] if nverify else []))
    sessions = [o.VerifySession(okind, param, label, vv, com, proof, cap) for proof, com, vv in recs]
    s0 = sessions[0]
    rp, kd, ix, coeff = s0.csr()
    pts = sc = ch = b""
    for (proof, com, _), s in zip(recs, sessions):
        k, p, q = bh.verify_inputs(proof, com)
        pts += p
        sc += q
        ch += s.challenges()
    want_ok = [1 if s.rc == 0 else 0 for s in sessions]
    assert want_ok.count(0) == 2
    for route in ({}, {"verify_window_parallel": 0}, {"verify_no_fuse": 1}):
        opts(verify_window_parallel=1, verify_no_fuse=0)
        opts(**route)
        circ = gpu.circuit_create(rp, kd, ix, coeff, s0.n1 + s0.n2, s0.m)
        g = _gens(gpu, cap, 8)
        try:
            ok, mega, full = gpu.r1cs_verify_batch(g, circ, nb, s0.n1, s0.k, s0.m, pts, sc, ch, True, True)
            assert ok == want_ok, route
            for i, s in enumerate(sessions):
                assert mega[64 * i:64 * i + 64] == s.mega_check(), (route, i)
                assert full[32 * s.nterms * i:32 * s.nterms * (i + 1)] == s.msm_terms()[0], (route, i)
        finally:
            gpu.gens_destroy(g)
            gpu.circuit_destroy(circ)
    for s in sessions:
        s.close()


def test_profile_events_select_and_cap(gpu):
    """bpgpu_profile_enable / _select / _read (the HIP-event timing bench.py's roofline uses): with a kind mask only the selected
    launches are timed; at most 256 launches per kind are kept between two reads; verdicts are unaffected.  bpgpu_profile_intervals
    returns the same launches as (kind, start, end) relative to an epoch: ordered, non-negative, and inside the wall-clock window."""
    recs, cap = bh.make_range_batch(8, 70, tamper={3})
    sessions = [o.VerifySession(o.K_RANGE, 8, b"RangeProofTest", [], com, proof, cap) for proof, com in recs]
    s0 = sessions[0]
    circ = gpu.circuit_create(*s0.csr(), s0.n1 + s0.n2, s0.m)
    g = _gens(gpu, cap, 8)
    try:
        pts = sc = ch = b""
        for (proof, com), s in zip(recs, sessions):
            k, p, q = bh.verify_inputs(proof, com)
            pts, sc, ch = pts + p, sc + q, ch + s.challenges()
        want = [1 if s.rc == 0 else 0 for s in sessions]
        gpu.profile_enable(True)
        ok, _, _ = gpu.r1cs_verify_batch(g, circ, 70, s0.n1, s0.k, s0.m, pts, sc, ch, False, False)
        assert list(ok) == want
        every = {n: c for n, (ms, c) in gpu.profile_read().items() if c}
        assert {"verify_front", "verify_scalars", "verify_windows", "verify_groups", "verify_back", "verify_verdict"} <= set(every)
        gpu.profile_select(["verify_back"])
        for _ in range(260):
            ok, _, _ = gpu.r1cs_verify_batch(g, circ, 70, s0.n1, s0.k, s0.m, pts, sc, ch, False, False)
        assert list(ok) == want
        only = {n: (ms, c) for n, (ms, c) in gpu.profile_read().items() if c}
        assert set(only) == {"verify_back"} and only["verify_back"][1] == 256 and only["verify_back"][0] > 0
        gpu.profile_select(None)
        import time
        epoch = gpu.profile_epoch()
        t0 = time.perf_counter()
        for _ in range(3):
            gpu.r1cs_verify_batch(g, circ, 70, s0.n1, s0.k, s0.m, pts, sc, ch, False, False)
        wall_ms = (time.perf_counter() - t0) * 1e3
        iv = gpu.profile_intervals(epoch)
        six = ["verify_front", "verify_scalars", "verify_windows", "verify_groups", "verify_back", "verify_verdict"]
        iv = [x for x in iv if x[0] != "fixed_msm"]                      # (the generator half as its own launch, when it does not ride in `back`)
        assert len(iv) == 18 and {n for n, _, _ in iv} == set(six)
        assert all(0 <= a <= b <= wall_ms + 1 for _, a, b in iv)
        chain = sorted((a, b, n) for n, a, b in iv)[:6]                  # the first batch's six launches, in stream order
        assert [n for _, _, n in chain] == six
        assert all(chain[i][1] <= chain[i + 1][0] + 1e-3 for i in range(5))
        assert gpu.profile_intervals(epoch) == []                       # read-and-clear
    finally:
        gpu.profile_enable(False)
        gpu.gens_destroy(g)
        gpu.circuit_destroy(circ)

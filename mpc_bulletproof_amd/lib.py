"""ctypes binding of libbpgpu.so (include/bpgpu.h).  Byte-level, no torch types."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libbpgpu.so")

SYMBOLS = [
    "bpgpu_device_count", "bpgpu_create", "bpgpu_destroy", "bpgpu_strerror", "bpgpu_last_error", "bpgpu_sync",
    "bpgpu_stream", "bpgpu_set_latency_mode", "bpgpu_set_shard", "bpgpu_r1cs_verify_shard", "bpgpu_set_option", "bpgpu_get_option", "bpgpu_input_flag", "bpgpu_profile_enable", "bpgpu_profile_select", "bpgpu_profile_read", "bpgpu_profile_epoch", "bpgpu_profile_intervals", "bpgpu_malloc", "bpgpu_free", "bpgpu_upload", "bpgpu_download", "bpgpu_upload_async", "bpgpu_download_async", "bpgpu_host_alloc", "bpgpu_host_free",
    "bpgpu_batch_inverse", "bpgpu_inner_product", "bpgpu_msm", "bpgpu_msm_batch", "bpgpu_msm_batch_dev", "bpgpu_points_sum", "bpgpu_msm_ark", "bpgpu_scalars_from_ark", "bpgpu_scalars_to_ark", "bpgpu_points_from_ark", "bpgpu_points_to_ark", "bpgpu_msm_shared", "bpgpu_points_decompress", "bpgpu_points_compress", "bpgpu_gens_create",
    "bpgpu_gens_destroy", "bpgpu_gens_capacity", "bpgpu_msm_gens", "bpgpu_msm_gens_ark", "bpgpu_fold_witness",
    "bpgpu_verification_scalars", "bpgpu_ipp_begin", "bpgpu_ipp_begin_gens", "bpgpu_ipp_destroy", "bpgpu_ipp_len", "bpgpu_ipp_round",
    "bpgpu_ipp_fold", "bpgpu_ipp_finish", "bpgpu_ipp_folded_gens", "bpgpu_ipp_run_fs", "bpgpu_r1cs_prover_polys", "bpgpu_r1cs_prover_polys_ark", "bpgpu_r1cs_prover_eval", "bpgpu_r1cs_prover_ipp_begin", "bpgpu_prover_destroy", "bpgpu_r1cs_prover_commit", "bpgpu_r1cs_prover_session_polys", "bpgpu_r1cs_prover_session_polys_param",
    "bpgpu_generator_mul", "bpgpu_circuit_create", "bpgpu_circuit_create_ark", "bpgpu_circuit_create_param", "bpgpu_circuit_destroy", "bpgpu_flatten_constraints",
    "bpgpu_r1cs_verify_batch", "bpgpu_r1cs_verify_batch_dev", "bpgpu_r1cs_verify_stream", "bpgpu_r1cs_verify_stream_dev", "bpgpu_r1cs_verify_screened", "bpgpu_r1cs_verify_screened_dev", "bpgpu_r1cs_verify_screened_fs_dev", "bpgpu_r1cs_verify_combined",
    "bpgpu_r1cs_verify_combined_dev", "bpgpu_r1cs_verify_batch_fs", "bpgpu_r1cs_verify_batch_fs_dev",
    "bpgpu_r1cs_verify_batch_wire", "bpgpu_r1cs_verify_batch_wire_dev", "bpgpu_r1cs_verify_batch_param", "bpgpu_r1cs_verify_batch_fs2",
    "bpgpu_r1cs_verify_batch_fs2_dev",
]


class BpGpuError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__(f"bpgpu error {code}: {what}")


def load():
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: the HIP extension has not been built "
            "(run __graft_entry__.build()).  There is no CPU fallback.")
    lib = C.CDLL(SO_PATH)
    lib.bpgpu_strerror.restype = C.c_char_p
    lib.bpgpu_last_error.restype = C.c_char_p
    lib.bpgpu_stream.restype = C.c_void_p
    lib.bpgpu_gens_capacity.restype = C.c_size_t
    lib.bpgpu_ipp_len.restype = C.c_size_t
    return lib


_lib = load()
_lib.bpgpu_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
_lib.bpgpu_host_free.argtypes = [C.c_void_p]


def host_alloc(nbytes, data=None):
    """page-locked staging memory (bpgpu_host_alloc), optionally filled with `data`; returns a c_void_p"""
    p = C.c_void_p()
    rc = _lib.bpgpu_host_alloc(C.c_size_t(nbytes), C.byref(p))
    if rc or not p.value:
        raise BpGpuError(rc or E_OOM, "bpgpu_host_alloc")
    if data is not None:
        C.memmove(p, data if isinstance(data, bytes) else bytes(data), len(data))
    return p


def host_free(p):
    _lib.bpgpu_host_free(p)
E_ARG, E_LEN, E_DEVICE, E_OOM, E_GENS = -1, -2, -3, -4, -5
# bpgpu_set_option (include/bpgpu.h BPGPU_OPT_*)
OPT = {"msm_wp_max": 1, "msm_pip2_single": 2, "verify_no_fuse": 3, "verify_window_parallel": 4, "verify_straus_np": 5,
       "ipp_literal": 6, "vs_large_min": 7, "table_np": 8, "ipp_table_max_n": 9, "stream_lanes": 10, "stream_batch": 11, "screen_batch": 12,
       "horner_form": 13, "horner_row_max": 14, "pippenger_min": 15, "ipp_pippenger_min": 16, "fixed_lpm": 17, "groups_form": 18, "fixed_chunk_gens": 19}
# bpgpu_profile_read kinds (include/bpgpu.h BPGPU_PROF_KINDS)
PROF_NAMES = ["verify_scalars", "fixed_msm", "points_from_boundary", "straus", "verify_finalize", "transcript", "verify_msm",
              "verify_windows", "verify_front", "verify_groups", "verify_back", "verify_verdict", "combined_front_scalars_digits",
              "combined_sort_accum_reduce", "combined_unused", "combined_final", "prover_commit", "prover_polys", "msm_gens", "ipp_begin",
              "ipp_rounds", "ipp_round_msm", "reserved22", "reserved23"]
PROF_KINDS = len(PROF_NAMES)


def _buf(b):
    """a read-only operand for the C ABI WITHOUT copying it: the address of the bytes object's own storage (kept alive by the
    caller's reference for the duration of the call).  Copying 100 MB of operands twice on the Python side was what made a 2^20-term
    bpgpu_msm from "host buffers" look 15x slower than the resident call."""
    if not isinstance(b, bytes):
        b = bytes(b)
    return C.c_char_p(b if len(b) else b"\0")


def _inout(b):
    """an IN/OUT operand (bpgpu_batch_inverse inverts in place): a private, writable copy"""
    b = bytes(b)
    return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if len(b) else b"\0")


def _out(n):
    return (C.c_uint8 * max(n, 1))()


def device_count():
    return _lib.bpgpu_device_count()


class BpGpu:
    """One context on one device.  All byte encodings as in include/bpgpu.h."""

    def __init__(self, device=0):
        self.ctx = C.c_void_p()
        rc = _lib.bpgpu_create(device, C.byref(self.ctx))
        if rc:
            raise BpGpuError(rc, _lib.bpgpu_strerror(rc).decode())

    def close(self):
        if self.ctx:
            _lib.bpgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise BpGpuError(rc, _lib.bpgpu_strerror(rc).decode() + " | " + _lib.bpgpu_last_error(self.ctx).decode())

    # ---- plumbing
    def set_option(self, name, value):
        """launch-route option of this context (OPT keys; include/bpgpu.h BPGPU_OPT_*)"""
        self._ck(_lib.bpgpu_set_option(self.ctx, C.c_int(OPT[name]), C.c_int64(int(value))))

    def get_option(self, name):
        v = C.c_int64()
        self._ck(_lib.bpgpu_get_option(self.ctx, C.c_int(OPT[name]), C.byref(v)))
        return v.value

    def options(self, **kw):
        """context manager: set options, restore the previous values on exit"""
        gpu = self

        class _Scope:
            def __enter__(s):
                s.old = {k: gpu.get_option(k) for k in kw}
                for k, v in kw.items():
                    gpu.set_option(k, v)
                return gpu

            def __exit__(s, *exc):
                for k, v in s.old.items():
                    gpu.set_option(k, v)
                return False
        return _Scope()

    def sync(self):
        self._ck(_lib.bpgpu_sync(self.ctx))

    def stream(self):
        return _lib.bpgpu_stream(self.ctx)

    def malloc(self, nbytes):
        p = C.c_void_p()
        self._ck(_lib.bpgpu_malloc(self.ctx, C.c_size_t(nbytes), C.byref(p)))
        return p

    def free(self, p):
        self._ck(_lib.bpgpu_free(self.ctx, p))

    def upload(self, dptr, data):
        self._ck(_lib.bpgpu_upload(self.ctx, dptr, _buf(data), C.c_size_t(len(data))))

    def upload_async(self, dptr, host_ptr, nbytes):
        """enqueue a copy from (page-locked) host memory on the context's stream; host_ptr: integer address / c_void_p"""
        self._ck(_lib.bpgpu_upload_async(self.ctx, dptr, C.c_void_p(host_ptr if isinstance(host_ptr, int) else host_ptr.value), C.c_size_t(nbytes)))

    def to_device(self, data):
        p = self.malloc(len(data))
        self.upload(p, data)
        return p

    def download(self, dptr, nbytes):
        o = _out(nbytes)
        self._ck(_lib.bpgpu_download(self.ctx, o, dptr, C.c_size_t(nbytes)))
        return bytes(o)[:nbytes]

    def input_flag(self):
        v = C.c_int(0)
        self._ck(_lib.bpgpu_input_flag(self.ctx, C.byref(v)))
        return v.value

    def set_latency_mode(self, on=True):
        self._ck(_lib.bpgpu_set_latency_mode(self.ctx, 1 if on else 0))

    def profile_enable(self, on=True):
        self._ck(_lib.bpgpu_profile_enable(self.ctx, 1 if on else 0))

    def profile_select(self, names=None):
        """restrict the event timing to the named kinds (PROF_NAMES); None = all"""
        mask = 0xffffffff if names is None else sum(1 << PROF_NAMES.index(n) for n in names)
        self._ck(_lib.bpgpu_profile_select(self.ctx, C.c_uint32(mask)))

    def profile_read(self):
        ms = (C.c_double * PROF_KINDS)()
        cnt = (C.c_uint64 * PROF_KINDS)()
        self._ck(_lib.bpgpu_profile_read(self.ctx, ms, cnt))
        return {n: (ms[i], int(cnt[i])) for i, n in enumerate(PROF_NAMES)}

    def profile_epoch(self):
        """reference event for profile_intervals (of any context of this device); owned by the context"""
        _lib.bpgpu_profile_epoch.restype = C.c_void_p
        e = _lib.bpgpu_profile_epoch(self.ctx)
        if not e:
            raise BpGpuError(E_DEVICE, "bpgpu_profile_epoch")
        return C.c_void_p(e)

    def profile_intervals(self, epoch, cap=8192):
        """-> [(kind name, start ms, end ms)] of every timed launch since the last read, relative to `epoch`"""
        kind, a, b, n = (C.c_int32 * cap)(), (C.c_double * cap)(), (C.c_double * cap)(), C.c_size_t(0)
        self._ck(_lib.bpgpu_profile_intervals(self.ctx, epoch, C.c_size_t(cap), kind, a, b, C.byref(n)))
        return [(PROF_NAMES[kind[i]], a[i], b[i]) for i in range(n.value)]

    # ---- scalar field
    def batch_inverse(self, scalars):
        n = len(scalars) // 32
        b = _inout(scalars)
        self._ck(_lib.bpgpu_batch_inverse(self.ctx, b, C.c_size_t(n)))
        return bytes(b)[:32 * n]

    def msm_batch_dev(self, nb, n, d_scalars, d_points, d_out):
        self._ck(_lib.bpgpu_msm_batch_dev(self.ctx, C.c_size_t(nb), C.c_size_t(n), d_scalars, d_points, d_out))

    # ---- arkworks in-memory forms (include/bpgpu.h): scalars 32 B = x 2^256 mod n, points 96 B = Jacobian coords c 2^256 mod p
    def msm_ark(self, scalars_mont, points_jac_mont):
        n = len(scalars_mont) // 32
        if len(points_jac_mont) != 96 * n:
            raise BpGpuError(E_LEN, "msm_ark: length mismatch")
        o = _out(96)
        self._ck(_lib.bpgpu_msm_ark(self.ctx, _buf(scalars_mont), _buf(points_jac_mont), C.c_size_t(n), o))
        return bytes(o)

    def _ark(self, fn, data, isz, osz):
        n = len(data) // isz
        o = _out(osz * n)
        self._ck(fn(self.ctx, _buf(data), C.c_size_t(n), o))
        return bytes(o)[:osz * n]

    def scalars_from_ark(self, b):
        return self._ark(_lib.bpgpu_scalars_from_ark, b, 32, 32)

    def scalars_to_ark(self, b):
        return self._ark(_lib.bpgpu_scalars_to_ark, b, 32, 32)

    def points_from_ark(self, b):
        return self._ark(_lib.bpgpu_points_from_ark, b, 96, 64)

    def points_to_ark(self, b):
        return self._ark(_lib.bpgpu_points_to_ark, b, 64, 96)

    def points_sum(self, points):
        """sum of the 64-byte points in `points` (no scalars) -> 64 bytes"""
        n = len(points) // 64
        out = _out(64)
        self._ck(_lib.bpgpu_points_sum(self.ctx, _buf(points), C.c_size_t(n), out))
        return bytes(out)

    def msm_shared(self, nsets, n, scalars, points):
        """nsets MSMs over one point vector (msm_authenticated_iter's share / MAC / modifier MSMs)."""
        out = _out(64 * nsets)
        self._ck(_lib.bpgpu_msm_shared(self.ctx, C.c_size_t(nsets), C.c_size_t(n), _buf(scalars), _buf(points), out))
        return bytes(out)[:64 * nsets]

    def points_decompress(self, compressed):
        """-> (xy bytes n x 64, ok list) for n x 32-byte compressed points"""
        n = len(compressed) // 32
        xy, ok = _out(64 * n), (C.c_int32 * max(n, 1))()
        self._ck(_lib.bpgpu_points_decompress(self.ctx, _buf(compressed), C.c_size_t(n), xy, ok))
        return bytes(xy)[:64 * n], list(ok)[:n]

    def points_compress(self, xy):
        n = len(xy) // 64
        out = _out(32 * n)
        self._ck(_lib.bpgpu_points_compress(self.ctx, _buf(xy), C.c_size_t(n), out))
        return bytes(out)[:32 * n]

    def inner_product(self, a, b):
        if len(a) != len(b):
            raise BpGpuError(E_LEN, "inner_product(a,b): lengths of vectors do not match")
        o = _out(32)
        self._ck(_lib.bpgpu_inner_product(self.ctx, _buf(a), _buf(b), C.c_size_t(len(a) // 32), o))
        return bytes(o)

    # ---- MSM
    def msm(self, scalars, points):
        n = len(scalars) // 32
        if len(points) != 64 * n:
            raise BpGpuError(E_LEN, "msm: length mismatch")
        o = _out(64)
        self._ck(_lib.bpgpu_msm(self.ctx, _buf(scalars), _buf(points), C.c_size_t(n), o))
        return bytes(o)

    def msm_batch(self, nb, n, scalars, points):
        if len(scalars) != 32 * nb * n or len(points) != 64 * nb * n:
            raise BpGpuError(E_LEN, "msm_batch: length mismatch")
        o = _out(64 * nb)
        self._ck(_lib.bpgpu_msm_batch(self.ctx, C.c_size_t(nb), C.c_size_t(n), _buf(scalars), _buf(points), o))
        return bytes(o)[:64 * nb]

    # ---- generators
    def gens_create(self, G, H, B, B_blinding, window_bits=8):
        cap = len(G) // 64
        h = C.c_void_p()
        self._ck(_lib.bpgpu_gens_create(self.ctx, _buf(G), _buf(H), C.c_size_t(cap), _buf(B), _buf(B_blinding),
                                        window_bits, C.byref(h)))
        return h

    def gens_destroy(self, h):
        _lib.bpgpu_gens_destroy(self.ctx, h)

    def msm_gens(self, gens, nb, n, scalars, ark=False):
        """ark=True: the scalars are ark-ff Montgomery limbs (x * 2^256 mod n), converted on the device"""
        o = _out(64 * nb)
        fn = _lib.bpgpu_msm_gens_ark if ark else _lib.bpgpu_msm_gens
        self._ck(fn(self.ctx, gens, C.c_size_t(nb), C.c_size_t(n), _buf(scalars), o))
        return bytes(o)[:64 * nb]

    # ---- IPP
    def fold_witness(self, n, u, u_inv, a, b, G, H):
        if not (len(a) == len(b) == 64 * n and len(G) == len(H) == 128 * n):
            raise BpGpuError(E_LEN, "fold_witness: length mismatch")
        ao, bo, Go, Ho = _out(32 * n), _out(32 * n), _out(64 * n), _out(64 * n)
        self._ck(_lib.bpgpu_fold_witness(self.ctx, C.c_size_t(n), _buf(u), _buf(u_inv), _buf(a), _buf(b), _buf(G),
                                         _buf(H), ao, bo, Go, Ho))
        return bytes(ao)[:32 * n], bytes(bo)[:32 * n], bytes(Go)[:64 * n], bytes(Ho)[:64 * n]

    def verification_scalars(self, challenges, n):
        k = len(challenges) // 32
        a, b, s = _out(32 * k), _out(32 * k), _out(32 * n)
        self._ck(_lib.bpgpu_verification_scalars(self.ctx, _buf(challenges), C.c_size_t(k), C.c_size_t(n), a, b, s))
        return bytes(a)[:32 * k], bytes(b)[:32 * k], bytes(s)[:32 * n]

    # lock-step InnerProductProof::create (transcript on the host)
    def ipp_begin(self, nb, n, Q, Gf, Hf, G, H, shared_gens, a, b):
        h = C.c_void_p()
        self._ck(_lib.bpgpu_ipp_begin(self.ctx, C.c_size_t(nb), C.c_size_t(n), _buf(Q), _buf(Gf), _buf(Hf), _buf(G),
                                      _buf(H), 1 if shared_gens else 0, _buf(a), _buf(b), C.byref(h)))
        return h

    def ipp_begin_gens(self, gens, nb, n, w, Gf, Hf, a, b):
        """Resident-generator session: G, H = gens[:n], Q = w * B (no generator folding on the device)."""
        h = C.c_void_p()
        self._ck(_lib.bpgpu_ipp_begin_gens(self.ctx, gens, C.c_size_t(nb), C.c_size_t(n), _buf(w), _buf(Gf), _buf(Hf),
                                           _buf(a), _buf(b), C.byref(h)))
        return h

    def ipp_len(self, s):
        return _lib.bpgpu_ipp_len(s)

    def ipp_round(self, s, nb):
        L, R = _out(64 * nb), _out(64 * nb)
        self._ck(_lib.bpgpu_ipp_round(self.ctx, s, L, R))
        return bytes(L)[:64 * nb], bytes(R)[:64 * nb]

    def ipp_fold(self, s, u, u_inv):
        self._ck(_lib.bpgpu_ipp_fold(self.ctx, s, _buf(u), _buf(u_inv)))

    def ipp_finish(self, s, nb):
        a, b = _out(32 * nb), _out(32 * nb)
        self._ck(_lib.bpgpu_ipp_finish(self.ctx, s, a, b))
        return bytes(a)[:32 * nb], bytes(b)[:32 * nb]

    def ipp_folded_gens(self, s, nb):
        G, H = _out(64 * nb), _out(64 * nb)
        self._ck(_lib.bpgpu_ipp_folded_gens(self.ctx, s, G, H))
        return bytes(G)[:64 * nb], bytes(H)[:64 * nb]

    def ipp_run_fs(self, s, nb, k, states):
        """all rounds with the transcript on the device -> (L bytes nb*k*64, R, a, b, states_out)"""
        L, R = _out(64 * nb * max(k, 1)), _out(64 * nb * max(k, 1))
        a, b, so = _out(32 * nb), _out(32 * nb), _out(32 * nb)
        self._ck(_lib.bpgpu_ipp_run_fs(self.ctx, s, _buf(states), L, R, a, b, so))
        return bytes(L)[:64 * nb * k], bytes(R)[:64 * nb * k], bytes(a)[:32 * nb], bytes(b)[:32 * nb], bytes(so)[:32 * nb]

    def ipp_destroy(self, s):
        _lib.bpgpu_ipp_destroy(self.ctx, s)

    def generator_mul(self, scalars):
        n = len(scalars) // 32
        o = _out(64 * n)
        self._ck(_lib.bpgpu_generator_mul(self.ctx, _buf(scalars), C.c_size_t(n), o))
        return bytes(o)[:64 * n]

    def prover_polys(self, circuit, nb, n, m, y, y_inv, z, a_L, a_R, a_O, s_L, s_R):
        t, wV, h = _out(32 * 6 * nb), _out(32 * nb * m), C.c_void_p()
        self._ck(_lib.bpgpu_r1cs_prover_polys(self.ctx, circuit, C.c_size_t(nb), _buf(y), _buf(y_inv), _buf(z), _buf(a_L),
                                              _buf(a_R), _buf(a_O), _buf(s_L), _buf(s_R), t, wV, C.byref(h)))
        return bytes(t)[:32 * 6 * nb], bytes(wV)[:32 * nb * m], h

    def prover_eval(self, sess, nb, padded_n, x):
        lv, rv = _out(32 * nb * padded_n), _out(32 * nb * padded_n)
        self._ck(_lib.bpgpu_r1cs_prover_eval(self.ctx, sess, C.c_size_t(padded_n), _buf(x), lv, rv))
        return bytes(lv)[:32 * nb * padded_n], bytes(rv)[:32 * nb * padded_n]

    def prover_destroy(self, sess):
        _lib.bpgpu_prover_destroy(self.ctx, sess)

    # ---- R1CS
    def circuit_create(self, row_ptr, kind, idx, coeff, n_mul, m, ark=False):
        """ark=True: coefficients as ark-ff Montgomery limbs (x * 2^256 mod n)"""
        q = len(row_ptr) - 1
        nnz = len(kind)
        h = C.c_void_p()
        rp = (C.c_uint32 * (q + 1))(*row_ptr)
        kd = (C.c_uint32 * max(nnz, 1))(*kind)
        ix = (C.c_uint32 * max(nnz, 1))(*idx)
        fn = _lib.bpgpu_circuit_create_ark if ark else _lib.bpgpu_circuit_create
        self._ck(fn(self.ctx, C.c_size_t(q), rp, kd, ix, _buf(coeff), C.c_size_t(n_mul), C.c_size_t(m), C.byref(h)))
        return h

    def circuit_destroy(self, h):
        _lib.bpgpu_circuit_destroy(self.ctx, h)

    def flatten_constraints(self, circuit, n_mul, m, z, want_wc=True):
        nb = len(z) // 32
        wL, wR, wO = _out(32 * nb * n_mul), _out(32 * nb * n_mul), _out(32 * nb * n_mul)
        wV, wc = _out(32 * nb * m), _out(32 * nb)
        self._ck(_lib.bpgpu_flatten_constraints(self.ctx, circuit, C.c_size_t(nb), _buf(z), wL, wR, wO, wV,
                                                wc if want_wc else None))

        def cut(b, k):
            return bytes(b)[:32 * nb * k]

        return cut(wL, n_mul), cut(wR, n_mul), cut(wO, n_mul), cut(wV, m), bytes(wc)[:32 * nb]

    def r1cs_prover_polys(self, circuit, nb, n, m, y, y_inv, z, a_L, a_R, a_O, s_L, s_R, ark=False):
        """prover.rs:587-619 for nb provers of one circuit -> (t_coeffs nb x 6 x 32 B, wV nb x m x 32 B, prover handle);
        ark=True: every input scalar in ark-ff Montgomery form"""
        for v in (a_L, a_R, a_O, s_L, s_R):
            if len(v) != 32 * nb * n:
                raise BpGpuError(E_LEN, "r1cs_prover_polys: length mismatch")
        t, wv, h = _out(32 * 6 * nb), _out(32 * nb * max(m, 1)), C.c_void_p()
        fn = _lib.bpgpu_r1cs_prover_polys_ark if ark else _lib.bpgpu_r1cs_prover_polys
        self._ck(fn(self.ctx, circuit, C.c_size_t(nb), _buf(y), _buf(y_inv), _buf(z), _buf(a_L), _buf(a_R),
                    _buf(a_O), _buf(s_L), _buf(s_R), t, wv, C.byref(h)))
        return bytes(t)[:32 * 6 * nb], bytes(wv)[:32 * nb * m], h

    def r1cs_prover_eval(self, prover, nb, padded_n, x):
        """prover.rs:659-672: l_vec, r_vec (nb x padded_n) = l(x), r(x) with the zero / -y^i padding"""
        lv, rv = _out(32 * nb * padded_n), _out(32 * nb * padded_n)
        self._ck(_lib.bpgpu_r1cs_prover_eval(self.ctx, prover, C.c_size_t(padded_n), _buf(x), lv, rv))
        return bytes(lv)[:32 * nb * padded_n], bytes(rv)[:32 * nb * padded_n]

    def r1cs_prover_commit(self, gens, session, nb, n_new, a_L, a_R, a_O, blindings, s_L=None, s_R=None, vector_keys=None):
        """prover.rs:457-494 / :519-565 on a resident-witness session (None: open one) -> (session, A_I A_O S nb x 3 x 64 B).
        All scalars in ark-ff Montgomery form; the blinding vectors explicit (s_L, s_R) or from vector_keys (nb x 32 B)."""
        h = session if session is not None else C.c_void_p()
        out = _out(64 * 3 * nb)
        opt = lambda b: _buf(b) if b is not None else None     # noqa: E731
        self._ck(_lib.bpgpu_r1cs_prover_commit(self.ctx, gens, C.byref(h), C.c_size_t(nb), C.c_size_t(n_new), opt(a_L), opt(a_R),
                                               opt(a_O), opt(s_L), opt(s_R), opt(vector_keys), _buf(blindings), out))
        return h, bytes(out)[:64 * 3 * nb]

    def r1cs_prover_session_polys(self, session, circuit, nb, m, y, z):
        """prover.rs:587-619 on the session's planes -> (t_coeffs nb x 6 x 32 B, wV nb x m x 32 B); y, z canonical LE"""
        t, wv = _out(32 * 6 * nb), _out(32 * nb * max(m, 1))
        self._ck(_lib.bpgpu_r1cs_prover_session_polys(self.ctx, session, circuit, _buf(y), _buf(z), t, wv))
        return bytes(t)[:32 * 6 * nb], bytes(wv)[:32 * nb * m]

    def prover_destroy(self, prover):
        _lib.bpgpu_prover_destroy(self.ctx, prover)

    def r1cs_verify_batch(self, gens, circuit, nb, n1, k, m, points, scalars, challenges, want_mega=True,
                          want_scalars=False):
        np_ = 1 << k
        nvar, nterms = 11 + m + 2 * k, 13 + m + 2 * np_ + 2 * k
        if len(points) != 64 * nb * nvar or len(scalars) != 160 * nb or len(challenges) != 32 * nb * (6 + k):
            raise BpGpuError(E_LEN, "r1cs_verify_batch: length mismatch")
        ok = (C.c_int32 * max(nb, 1))()
        mega = _out(64 * nb) if want_mega else None
        full = _out(32 * nb * nterms) if want_scalars else None
        self._ck(_lib.bpgpu_r1cs_verify_batch(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                              _buf(points), _buf(scalars), _buf(challenges), ok, mega, full))
        return (list(ok)[:nb], bytes(mega)[:64 * nb] if want_mega else None,
                bytes(full)[:32 * nb * nterms] if want_scalars else None)

    def r1cs_verify_batch_fs(self, gens, circuit, nb, n1, k, m, init_states, points, scalars, want_mega=True):
        nvar = 11 + m + 2 * k
        if len(points) != 64 * nb * nvar or len(scalars) != 160 * nb or len(init_states) != 32 * nb:
            raise BpGpuError(E_LEN, "r1cs_verify_batch_fs: length mismatch")
        ok = (C.c_int32 * max(nb, 1))()
        mega = _out(64 * nb) if want_mega else None
        ch = _out(32 * nb * (6 + k))
        self._ck(_lib.bpgpu_r1cs_verify_batch_fs(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                                 _buf(init_states), _buf(points), _buf(scalars), ok, mega, ch))
        return list(ok)[:nb], (bytes(mega)[:64 * nb] if want_mega else None), bytes(ch)[:32 * nb * (6 + k)]

    def circuit_create_param(self, q, nchi, row_ptr, kind, idx, coeff, n_mul, m):
        """two-phase circuit with coefficients affine in nchi gadget challenges: a CSR of (1 + nchi) * q rows (include/bpgpu.h)"""
        nnz = len(kind)
        rp = (C.c_uint32 * len(row_ptr))(*row_ptr)
        kd = (C.c_uint32 * max(nnz, 1))(*kind)
        ix = (C.c_uint32 * max(nnz, 1))(*idx)
        h = C.c_void_p()
        self._ck(_lib.bpgpu_circuit_create_param(self.ctx, C.c_size_t(q), C.c_size_t(nchi), rp, kd, ix, _buf(coeff), C.c_size_t(n_mul),
                                                 C.c_size_t(m), C.byref(h)))
        return h

    def r1cs_verify_batch_param(self, gens, circuit, nb, n1, k, m, points, scalars, challenges, gadget_challenges, want_mega=True,
                                want_scalars=False):
        np_ = 1 << k
        nterms = 13 + m + 2 * np_ + 2 * k
        ok = (C.c_int32 * max(nb, 1))()
        mega = _out(64 * nb) if want_mega else None
        full = _out(32 * nb * nterms) if want_scalars else None
        self._ck(_lib.bpgpu_r1cs_verify_batch_param(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k), _buf(points),
                                                    _buf(scalars), _buf(challenges), _buf(gadget_challenges), ok, mega, full))
        return (list(ok)[:nb], bytes(mega)[:64 * nb] if want_mega else None, bytes(full)[:32 * nb * nterms] if want_scalars else None)

    def r1cs_verify_batch_fs2(self, gens, circuit, nb, n1, k, m, nchi, init_states, gadget_label, points, scalars, want_mega=True):
        """-> (ok, mega, challenges nb x (6 + k) x 32, gadget challenges nb x nchi x 32)"""
        ok = (C.c_int32 * max(nb, 1))()
        mega = _out(64 * nb) if want_mega else None
        ch, chi = _out(32 * nb * (6 + k)), _out(32 * nb * max(nchi, 1))
        lab = (gadget_label + bytes(32))[:32]
        self._ck(_lib.bpgpu_r1cs_verify_batch_fs2(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k), _buf(init_states),
                                                  _buf(lab), _buf(points), _buf(scalars), ok, mega, ch, chi))
        return list(ok)[:nb], (bytes(mega)[:64 * nb] if want_mega else None), bytes(ch)[:32 * nb * (6 + k)], bytes(chi)[:32 * nb * nchi]

    def r1cs_verify_batch_fs_dev(self, gens, circuit, nb, n1, k, d_init, d_points, d_scalars, d_ok, d_mega=None, d_ch=None):
        self._ck(_lib.bpgpu_r1cs_verify_batch_fs_dev(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                                     d_init, d_points, d_scalars, d_ok, d_mega, d_ch))

    def r1cs_verify_batch_wire(self, gens, circuit, nb, n1, proof_len, proofs, commitments, init_states):
        """wire-format proofs (R1CSProof::to_bytes) + compressed commitments -> accept bits, all on the device"""
        if len(proofs) != nb * proof_len or len(init_states) != 32 * nb:
            raise BpGpuError(E_LEN, "r1cs_verify_batch_wire: length mismatch")
        ok = (C.c_int32 * max(nb, 1))()
        self._ck(_lib.bpgpu_r1cs_verify_batch_wire(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(proof_len),
                                                   _buf(proofs), _buf(commitments), _buf(init_states), ok))
        return list(ok)[:nb]

    def r1cs_verify_batch_wire_dev(self, gens, circuit, nb, n1, proof_len, d_proofs, d_commitments, d_init, d_ok):
        self._ck(_lib.bpgpu_r1cs_verify_batch_wire_dev(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1),
                                                       C.c_size_t(proof_len), d_proofs, d_commitments, d_init, d_ok))

    def r1cs_verify_combined(self, gens, circuit, nb, n1, k, m, points, scalars, challenges, rho):
        nvar = 11 + m + 2 * k
        if len(points) != 64 * nb * nvar or len(scalars) != 160 * nb or len(challenges) != 32 * nb * (6 + k) or len(rho) != 32 * nb:
            raise BpGpuError(E_LEN, "r1cs_verify_combined: length mismatch")
        o = _out(64)
        self._ck(_lib.bpgpu_r1cs_verify_combined(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                                 _buf(points), _buf(scalars), _buf(challenges), _buf(rho), o))
        return bytes(o)

    def r1cs_verify_combined_dev(self, gens, circuit, nb, n1, k, d_points, d_scalars, d_challenges, d_rho, d_out):
        self._ck(_lib.bpgpu_r1cs_verify_combined_dev(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                                     d_points, d_scalars, d_challenges, d_rho, d_out))

    def set_shard(self, rank, world):
        """this context's share of ONE large proof split over the GPUs of a node (include/bpgpu.h bpgpu_set_shard)"""
        self._ck(_lib.bpgpu_set_shard(self.ctx, C.c_size_t(rank), C.c_size_t(world)))

    def r1cs_verify_shard(self, gens, circuit, n1, k, points, scalars, challenges, rank, world, gadget_challenges=None):
        """rank's partial mega_check point of one proof (64 bytes)"""
        out = _out(64)
        self._ck(_lib.bpgpu_r1cs_verify_shard(self.ctx, gens, circuit, C.c_size_t(n1), C.c_size_t(k), _buf(points), _buf(scalars),
                                              _buf(challenges), _buf(gadget_challenges) if gadget_challenges is not None else None,
                                              C.c_size_t(rank), C.c_size_t(world), out))
        return bytes(out)[:64]

    def r1cs_verify_stream_dev(self, gens, circuit, nb, n1, k, d_points, d_scalars, d_challenges, d_ok):
        """any number of proofs in one call: batches over the context's ring of lanes (asynchronous; sync() waits for all)"""
        self._ck(_lib.bpgpu_r1cs_verify_stream_dev(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                                   d_points, d_scalars, d_challenges, d_ok))

    def r1cs_verify_stream(self, gens, circuit, nb, n1, k, m, points, scalars, challenges, raw=False):
        """the same from host memory (bytes, or c_void_p of page-locked memory for all three operands) -> [ok]
        (raw = True: the nb x int32 verdicts as bytes -- turning 65 536 verdicts into a Python list costs milliseconds)"""
        ok = (C.c_int32 * max(nb, 1))()
        wrap = lambda b: b if isinstance(b, C.c_void_p) else _buf(b)     # noqa: E731
        self._ck(_lib.bpgpu_r1cs_verify_stream(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                               wrap(points), wrap(scalars), wrap(challenges), ok))
        return bytes(ok)[:4 * nb] if raw else ok[:nb]

    def r1cs_verify_screened_dev(self, gens, circuit, nb, n1, k, d_points, d_scalars, d_challenges, d_rho, d_ok):
        """combined check per batch first, per-proof path only for the batches that fail it (device-resident operands and verdicts;
        sync() before reading d_ok) -> number of batches that took the per-proof path"""
        nf = C.c_size_t(0)
        self._ck(_lib.bpgpu_r1cs_verify_screened_dev(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                                     d_points, d_scalars, d_challenges, d_rho, d_ok, C.byref(nf)))
        return nf.value

    def r1cs_verify_screened_fs_dev(self, gens, circuit, nb, n1, k, d_init_states, d_points, d_scalars, d_rho, d_ok):
        """the same with the transcript replayed on the device (whole Verifier::verify; 1-phase circuits) -> fallback batches"""
        nf = C.c_size_t(0)
        self._ck(_lib.bpgpu_r1cs_verify_screened_fs_dev(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                                        d_init_states, d_points, d_scalars, d_rho, d_ok, C.byref(nf)))
        return nf.value

    def r1cs_verify_screened(self, gens, circuit, nb, n1, k, points, scalars, challenges, rho):
        """the same from host memory (bytes or c_void_p of page-locked memory) -> ([ok], fallback batches)"""
        ok = (C.c_int32 * max(nb, 1))()
        nf = C.c_size_t(0)
        wrap = lambda b: b if isinstance(b, C.c_void_p) else _buf(b)     # noqa: E731
        self._ck(_lib.bpgpu_r1cs_verify_screened(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1), C.c_size_t(k),
                                                 wrap(points), wrap(scalars), wrap(challenges), wrap(rho), ok, C.byref(nf)))
        return list(ok)[:nb], nf.value

    def r1cs_verify_batch_dev(self, gens, circuit, nb, n1, k, d_points, d_scalars, d_challenges, d_ok, d_mega=None,
                              d_full=None):
        self._ck(_lib.bpgpu_r1cs_verify_batch_dev(self.ctx, gens, circuit, C.c_size_t(nb), C.c_size_t(n1),
                                                  C.c_size_t(k), d_points, d_scalars, d_challenges, d_ok, d_mega,
                                                  d_full))

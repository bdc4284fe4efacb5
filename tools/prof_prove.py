import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o
host = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
nb, nvals, n_bits = int(sys.argv[1]), 16, 64
n = nvals * n_bits
label = b"RangeProofTest"
vals = [((0x9E3779B97F4A7C15 * (i + 1 + 31 * p)) & ((1 << 64) - 1)) for p in range(nb) for i in range(nvals)]
arr = (C.c_uint64 * len(vals))(*vals)
plen = C.c_size_t(0); proofs = (C.c_uint8 * (nb * 4096))(); com = (C.c_uint8 * (nb * nvals * 64))()
for it in range(2):
    time.sleep(0.1)      # an idle gap between the calls: tools/prof_prove.sh cuts the kernel trace there
    t0 = time.perf_counter()
    rc = host.bph_range_prove_batch(C.c_size_t(nb), C.c_size_t(nvals), C.c_size_t(n_bits), o._buf(label), C.c_size_t(len(label)), arr, C.c_uint64(900), C.c_size_t(n), proofs, C.byref(plen), com)
    print(rc, time.perf_counter() - t0)

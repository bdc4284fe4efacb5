// mpc_bulletproof.cpp -- host-side orchestration of the reference's API over the bpgpu C ABI.
// Citations are to renegade-fi/mpc-bulletproof (paths relative to its root).
#include "mpc_bulletproof.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <sys/random.h>
#include <unistd.h>
#include <cerrno>
#include <chrono>
#include <exception>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <map>
#include <cstdio>
#include <cstdlib>

#include <cstring>

namespace mpc_bulletproof {

typedef unsigned __int128 u128;

// ================================================================ Scalar (host glue, F_n) =========
namespace {
const uint64_t NMOD[4] = {0x1e66a241adc64d2fULL, 0xb781126dcae7b232ULL, 0xffffffffffffffffULL, 0x0800000000000010ULL};
const uint64_t N0INV = 0xbb6b3c4ce8bde631ULL;   // -n^-1 mod 2^64
const uint64_t R2[4] = {0x6021b3f1ea1c688dULL, 0x509cf64d14ce60b9ULL, 0xbaf0ab4cf78bbabbULL, 0x07d9e57c2333766eULL};
const uint64_t RONE[4] = {0x51925a0bf4fca74fULL, 0xc75ec4b46df16beeULL, 0x0000000000000008ULL, 0x07fffffffffffdf1ULL};

bool geq_n(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] != NMOD[i]) return a[i] > NMOD[i];
  }
  return true;
}
void sub_n(uint64_t a[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - NMOD[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
void mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * N0INV;
    c = ((u128)m * NMOD[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * NMOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || geq_n(t)) sub_n(t);
  memcpy(r, t, 32);
}
}  // namespace

Scalar Scalar::one() { Scalar s; memcpy(s.v_, RONE, 32); return s; }
Scalar Scalar::from(uint64_t x) { Scalar s; uint64_t t[4] = {x, 0, 0, 0}; mont_mul(s.v_, t, R2); return s; }
Scalar Scalar::from_bytes_le(const uint8_t b[32]) {
  uint64_t t[4];
  for (int i = 0; i < 4; i++) { t[i] = 0; for (int j = 7; j >= 0; j--) t[i] = (t[i] << 8) | b[8 * i + j]; }
  if (geq_n(t)) throw ProofException(ProofError::FormatError);
  Scalar s;
  mont_mul(s.v_, t, R2);
  return s;
}
Scalar Scalar::from_le_bytes_mod_order_wide(const uint8_t b[64]) {
  uint64_t lo[4], hi[4];
  for (int h = 0; h < 2; h++)
    for (int i = 0; i < 4; i++) {
      uint64_t w = 0;
      for (int j = 7; j >= 0; j--) w = (w << 8) | b[32 * h + 8 * i + j];
      (h ? hi : lo)[i] = w;
    }
  while (geq_n(lo)) sub_n(lo);
  while (geq_n(hi)) sub_n(hi);
  Scalar l, h;
  mont_mul(l.v_, lo, R2);
  mont_mul(h.v_, hi, R2);
  mont_mul(h.v_, h.v_, R2);   // hi * 2^256
  return l + h;
}
void Scalar::to_bytes_le(uint8_t out[32]) const {
  // witness vectors of R1CS gadgets are mostly 0 and 1 (bits, a_O of a range proof): no reduction for those
  if ((v_[0] | v_[1] | v_[2] | v_[3]) == 0) { memset(out, 0, 32); return; }
  if (memcmp(v_, RONE, 32) == 0) { memset(out, 0, 32); out[0] = 1; return; }
  uint64_t one[4] = {1, 0, 0, 0}, t[4];
  mont_mul(t, v_, one);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(t[i] >> (8 * j));
}
void Scalar::to_bytes_be(uint8_t out[32]) const {
  uint8_t le[32];
  to_bytes_le(le);
  for (int i = 0; i < 32; i++) out[i] = le[31 - i];
}
Scalar Scalar::from_be_bytes_mod_order(const uint8_t b[32]) {
  uint8_t wide[64] = {0};
  for (int i = 0; i < 32; i++) wide[i] = b[31 - i];
  return from_le_bytes_mod_order_wide(wide);
}
Scalar Scalar::operator+(const Scalar &o) const {
  Scalar r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)v_[i] + o.v_[i]; r.v_[i] = (uint64_t)c; c >>= 64; }
  if (geq_n(r.v_)) sub_n(r.v_);
  return r;
}
Scalar Scalar::operator-() const {
  Scalar r;
  if (*this == Scalar()) return r;
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)NMOD[i] - v_[i] - (uint64_t)br; r.v_[i] = (uint64_t)d; br = (d >> 64) & 1; }
  return r;
}
Scalar Scalar::operator-(const Scalar &o) const { return *this + (-o); }
Scalar Scalar::operator*(const Scalar &o) const { Scalar r; mont_mul(r.v_, v_, o.v_); return r; }
Scalar Scalar::inverse() const {
  uint64_t e[4] = {NMOD[0] - 2, NMOD[1], NMOD[2], NMOD[3]};
  Scalar acc = Scalar::one(), base = *this;
  for (int i = 0; i < 256; i++) {
    if ((e[i / 64] >> (i % 64)) & 1) acc = acc * base;
    base = base * base;
  }
  return acc;
}

uint64_t SeededRng::next_u64() {
  s_ += 0x9E3779B97F4A7C15ULL;
  uint64_t z = s_;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// ten 64-bit words of OS entropy, drawn once per process: the key of the constraint rows' running hash (CsCore::hash_term)
static void os_entropy(uint8_t *out, size_t len) {
  size_t got = 0;
  while (got < len) {
    ssize_t r = getrandom(out + got, len - got, 0);
    if (r < 0) { if (errno == EINTR) continue; break; }
    got += (size_t)r;
  }
  if (got < len) {   // kernels without getrandom(2)
    FILE *f = std::fopen("/dev/urandom", "rb");
    if (f) { got += std::fread(out + got, 1, len - got, f); std::fclose(f); }
  }
  if (got < len) throw std::runtime_error("no entropy from getrandom(2) or /dev/urandom");
}
static const uint64_t *row_hash_key() {
  static const struct Key { uint64_t k[10]; Key() { os_entropy((uint8_t *)k, sizeof k); } } key;
  return key.k;
}
// Keccak sponge DRBG: the 1600-bit state is seeded with 32 bytes of getrandom(2); output is squeezed 136 bytes per permutation
// (17 words; the first version hashed twice per 32 bytes, and 256 provers drawing 525 000 blinding scalars spent 150 ms in it);
// rekey() absorbs its material into the rate and permutes.  Forward secrecy: the rate is zeroed before EVERY permutation (the
// sponge "forget" step: the next state depends on the 512-bit capacity only), so a captured state reveals nothing of the
// blocks already handed out -- Keccak-f is invertible, but the inverse needs the rate words that were overwritten.
namespace { void permute_words(uint64_t s[25]); }
OsRng::OsRng(bool vector_keys) : vk_(vector_keys) {
  uint8_t key[32];
  os_entropy(key, sizeof key);
  memset(st_, 0, sizeof st_);
  memcpy(st_, key, 32);
  st_[4] ^= 0x01;                        // domain byte after the key, pad bit at the end of the rate
  st_[16] ^= 0x8000000000000000ULL;
  permute_words(st_);
  used_ = 0;
}
OsRng::OsRng(const uint8_t key[32], bool vector_keys) : vk_(vector_keys) {
  memset(st_, 0, sizeof st_);
  memcpy(st_, key, 32);
  st_[4] ^= 0x01;
  st_[16] ^= 0x8000000000000000ULL;
  permute_words(st_);
  used_ = 0;
}
void OsRng::refill() {
  ++blocks_;
  for (int i = 0; i < 17; i++) st_[i] = 0;   // forget: every block (17 stores against a 24-round permutation)
  permute_words(st_);
  used_ = 0;
}
uint64_t OsRng::next_u64() {
  if (used_ >= 17) refill();
  return st_[used_++];
}
void OsRng::rekey(const uint8_t *material, size_t len) {
  // absorb: 0x02 || material, padded, 136 bytes at a time
  size_t off = 0;
  uint8_t blk[136];
  bool first = true;
  do {
    memset(blk, 0, sizeof blk);
    size_t cap = first ? 135 : 136, take = len - off < cap ? len - off : cap;
    if (first) blk[0] = 0x02;
    if (take) memcpy(blk + (first ? 1 : 0), material + off, take);
    off += take;
    const bool last = off == len && take < cap;
    if (last) { blk[(first ? 1 : 0) + take] ^= 0x01; blk[135] ^= 0x80; }
    for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); st_[i] ^= w; }
    permute_words(st_);
    first = false;
    if (last) break;
  } while (true);
  used_ = 0;   // what was left of the old block is dropped: the state just changed under it anyway
}
Scalar Rng::scalar() {
  uint8_t b[64] = {0};
  for (int i = 0; i < 4; i++) { uint64_t w = next_u64(); for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(w >> (8 * j)); }
  return Scalar::from_le_bytes_mod_order_wide(b);
}

void Rng::scalars(Scalar *out, size_t count) {
  if (count < 4096) { for (size_t i = 0; i < count; i++) out[i] = scalar(); return; }
  std::vector<uint64_t> w(4 * count);
  for (auto &x : w) x = next_u64();
  const size_t chunks = 64;
  parallel_for(chunks, [&](size_t c) {
    for (size_t i = count * c / chunks; i < count * (c + 1) / chunks; i++) {
      uint8_t b[64] = {0};
      memcpy(b, &w[4 * i], 32);           // little-endian host
      out[i] = Scalar::from_le_bytes_mod_order_wide(b);
    }
  });
}

StarkPoint StarkPoint::generator() {
  static const uint8_t GEN[64] = {0xca,0xcf,0x43,0xc9,0x8b,0x3d,0x72,0x3d,0xe0,0x19,0x18,0x0d,0x9b,0xfd,0xac,0xde,0xc7,0xf0,0x40,0x5a,0x41,0xed,0xec,0x7b,0x1b,0x97,0x99,0x85,0xc1,0x15,0xef,0x01,
                                  0x1f,0xdc,0xe8,0x36,0x0c,0x00,0x73,0x28,0xa3,0x43,0xbe,0x1a,0xd1,0xec,0x53,0xde,0x62,0xec,0x46,0xdf,0x01,0x48,0xbe,0xb7,0x30,0x97,0xa4,0x0a,0x06,0x68,0x56,0x00};
  StarkPoint p;
  memcpy(p.xy.data(), GEN, 64);
  return p;
}

// ================================================================ host parallelism ================
// A persistent pool: creating and joining 16 threads costs ~0.4 ms, and one prove_batch call runs a dozen parallel loops.
// Worker t always takes the t-th contiguous slice of a loop, so the thread that built a prover's constraint system is the
// one that later packs and frees it (its allocations stay in that thread's malloc arena).  A loop started from inside a
// running loop -- on a worker OR on the calling thread while it runs its own slice -- runs serially (the caller holds the
// non-recursive run_mu_ during its slice: a nested run() from there would lock it again).
namespace {
class Pool {
 public:
  explicit Pool(unsigned n) : nworkers_(n > 1 ? n - 1 : 0) {
    for (unsigned t = 0; t < nworkers_; t++) th_.emplace_back([this, t] { worker(t + 1); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto &x : th_) x.join();
  }
  unsigned width() const { return nworkers_ + 1; }
  static bool in_worker() { return tl_worker_; }
  // slices [n t / nt, n (t + 1) / nt) for t < nt = min(n, width); the caller takes slice 0
  void run(size_t n, const std::function<void(size_t)> &f) {
    std::lock_guard<std::mutex> one(run_mu_);             // one loop at a time (callers on other threads queue up)
    const size_t nt = n < width() ? n : width();
    std::vector<std::exception_ptr> err(nt);
    {
      std::lock_guard<std::mutex> lk(mu_);
      n_ = n; nt_ = nt; f_ = &f; err_ = err.data(); pending_ = nt - 1; gen_++;
    }
    cv_.notify_all();
    {
      struct InLoop { bool was; InLoop() : was(tl_worker_) { tl_worker_ = true; } ~InLoop() { tl_worker_ = was; } } in_loop;
      slice(0);
    }
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return pending_ == 0; });
    f_ = nullptr;
    lk.unlock();
    for (auto &e : err) if (e) std::rethrow_exception(e);
  }
 private:
  void slice(size_t t) {
    try { for (size_t i = n_ * t / nt_; i < n_ * (t + 1) / nt_; i++) (*f_)(i); } catch (...) { err_[t] = std::current_exception(); }
  }
  void worker(unsigned t) {
    tl_worker_ = true;
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return gen_ != seen; });
      seen = gen_;
      if (stop_) return;
      if (t >= nt_) continue;                              // this loop is narrower than the pool
      lk.unlock();
      slice(t);
      lk.lock();
      if (--pending_ == 0) done_.notify_one();
    }
  }
  unsigned nworkers_;
  std::vector<std::thread> th_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_;
  uint64_t gen_ = 0;
  bool stop_ = false;
  size_t n_ = 0, nt_ = 0, pending_ = 0;
  const std::function<void(size_t)> *f_ = nullptr;
  std::exception_ptr *err_ = nullptr;
  static thread_local bool tl_worker_;
};
thread_local bool Pool::tl_worker_ = false;
}  // namespace
void parallel_for(size_t n, const std::function<void(size_t)> &f, size_t min_n) {
  static const unsigned nthreads = [] {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned t = hw ? (hw < 16 ? hw : 16) : 4;
    if (const char *e = getenv("BPH_THREADS")) { int v = atoi(e); if (v > 0) t = (unsigned)v; }
    return t;
  }();
  if (n < min_n || nthreads <= 1 || Pool::in_worker()) { for (size_t i = 0; i < n; i++) f(i); return; }
  // lives until process exit (no static-destruction-order races with worker threads); a forked child starts its own
  static std::mutex pool_mu;
  static Pool *pool = nullptr;
  static pid_t pool_pid = 0;
  {
    std::lock_guard<std::mutex> lk(pool_mu);
    if (!pool || pool_pid != getpid()) { pool = new Pool(nthreads); pool_pid = getpid(); }
  }
  pool->run(n, f);
}

// ================================================================ byte packing helpers ============
namespace {
// grow-only uninitialised byte buffers for the big scalar uploads (a zero-initialised std::vector costs a
// single-threaded memset + page faults of tens of MB per call)
// grow-only staging buffer for the large packed operands (50 MB of commitment scalars for 256 provers): page-locked
// through the C ABI when it can be, so that the copy to the device is one DMA at PCIe speed
struct RawBuf {
  uint8_t *p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  void release() { if (pinned) bpgpu_host_free(p); else free(p); p = nullptr; cap = 0; }
  ~RawBuf() { if (!pinned) free(p); }   // a pinned buffer of a thread that outlives the HIP runtime is left to the process exit
  uint8_t *ensure(size_t n) {
    if (n > cap) {
      release();
      size_t want = n + n / 4 + 1;                       // grow-only with head room: pinning is not cheap
      void *q = nullptr;
      if (!getenv("BPH_PAGEABLE_STAGING") && bpgpu_host_alloc(want, &q) == BPGPU_OK && q) { p = (uint8_t *)q; pinned = true; }
      else { p = (uint8_t *)malloc(want); pinned = false; }
      if (!p) throw std::bad_alloc();
      cap = want;
    }
    return p;
  }
};
inline void pack_range(uint8_t *dst, const Scalar *src, size_t n) { for (size_t i = 0; i < n; i++) src[i].to_bytes_le(dst + 32 * i); }
// the bulk operands of the prover (witness and blinding vectors) go to the device in their in-memory Montgomery form
inline void pack_range_ark(uint8_t *dst, const Scalar *src, size_t n) { for (size_t i = 0; i < n; i++) src[i].to_ark_le(dst + 32 * i); }
constexpr size_t PACK_CHUNK = 4096;   // scalars per parallel work item
std::vector<uint8_t> pack_scalars(const std::vector<Scalar> &v) {
  std::vector<uint8_t> o(v.size() * 32);
  const size_t n = v.size(), chunks = (n + PACK_CHUNK - 1) / PACK_CHUNK;
  parallel_for(chunks, [&](size_t c) {
    for (size_t i = c * PACK_CHUNK; i < n && i < (c + 1) * PACK_CHUNK; i++) v[i].to_bytes_le(&o[32 * i]);
  }, 4);
  return o;
}
std::vector<Scalar> unpack_scalars(const uint8_t *b, size_t n) {
  std::vector<Scalar> o(n);
  const size_t chunks = (n + PACK_CHUNK - 1) / PACK_CHUNK;
  parallel_for(chunks, [&](size_t c) {
    for (size_t i = c * PACK_CHUNK; i < n && i < (c + 1) * PACK_CHUNK; i++) o[i] = Scalar::from_bytes_le(b + 32 * i);
  }, 4);
  return o;
}
std::vector<uint8_t> pack_points(const std::vector<StarkPoint> &v) {
  std::vector<uint8_t> o(v.size() * 64);
  for (size_t i = 0; i < v.size(); i++) memcpy(&o[64 * i], v[i].xy.data(), 64);
  return o;
}
std::vector<StarkPoint> unpack_points(const uint8_t *b, size_t n) {
  std::vector<StarkPoint> o(n);
  for (size_t i = 0; i < n; i++) memcpy(o[i].xy.data(), b + 64 * i, 64);
  return o;
}
}  // namespace

// ================================================================ Device ==========================
Device::Device(int index) : index_(index) {
  int rc = bpgpu_create(index, &ctx_);
  if (rc) throw DeviceException(rc, std::string("bpgpu_create: ") + bpgpu_strerror(rc));
}
Device::~Device() { bpgpu_destroy(ctx_); }
static int g_default_device_index = 0;
void Device::set_default_index(int index) { g_default_device_index = index; }
Device &Device::default_device() { static Device d(g_default_device_index); return d; }
// Circuits, generator tables and the circuit cache live on the DEFAULT device's GPU: a Device handed to prove_batch / the sharded
// verify must be a further context of that same GPU (a worker thread's own stream and pool).  Kernels of another GPU's context
// would dereference this GPU's memory.
static Device &same_gpu(Device *device) {
  Device &def = Device::default_device();
  if (!device) return def;
  if (device->index() != def.index())
    throw std::invalid_argument("Device of another GPU index than the default device's: select the GPU with Device::set_default_index "
                                "before the first use (one process per GPU)");
  return *device;
}
void Device::check(int rc, const char *what) const {
  if (rc) throw DeviceException(rc, std::string(what) + ": " + bpgpu_strerror(rc) + " | " + bpgpu_last_error(ctx_));
}
StarkPoint Device::msm(const std::vector<Scalar> &scalars, const std::vector<StarkPoint> &points) const {
  if (scalars.size() != points.size()) throw std::invalid_argument("msm: length mismatch");
  StarkPoint out;
  auto s = pack_scalars(scalars);
  auto p = pack_points(points);
  check(bpgpu_msm(ctx_, s.data(), p.data(), scalars.size(), out.xy.data()), "bpgpu_msm");
  return out;
}

// ================================================================ keccak / transcript =============
namespace {
const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
inline uint64_t rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
// Keccak-f[1600] on the flat state s[x + 5 y], rounds unrolled over the lanes (the generic double loop with modular indices ran
// at ~1.5 us per permutation: the 2^14-shuffle's transcript, the generator chains and the OS-keyed blinding stream are all
// tens of thousands of permutations)
void permute_scalar(uint64_t s[25]) {
  for (int r = 0; r < 24; r++) {
    const uint64_t c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20], c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21];
    const uint64_t c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22], c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23];
    const uint64_t c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];
    const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1);
    for (int y = 0; y < 25; y += 5) { s[y] ^= d0; s[y + 1] ^= d1; s[y + 2] ^= d2; s[y + 3] ^= d3; s[y + 4] ^= d4; }
    uint64_t b[25];
    b[0] = s[0];              b[10] = rotl(s[1], 1);   b[20] = rotl(s[2], 62);  b[5] = rotl(s[3], 28);   b[15] = rotl(s[4], 27);
    b[16] = rotl(s[5], 36);   b[1] = rotl(s[6], 44);   b[11] = rotl(s[7], 6);   b[21] = rotl(s[8], 55);  b[6] = rotl(s[9], 20);
    b[7] = rotl(s[10], 3);    b[17] = rotl(s[11], 10); b[2] = rotl(s[12], 43);  b[12] = rotl(s[13], 25); b[22] = rotl(s[14], 39);
    b[23] = rotl(s[15], 41);  b[8] = rotl(s[16], 45);  b[18] = rotl(s[17], 15); b[3] = rotl(s[18], 21);  b[13] = rotl(s[19], 8);
    b[14] = rotl(s[20], 18);  b[24] = rotl(s[21], 2);  b[9] = rotl(s[22], 61);  b[19] = rotl(s[23], 56); b[4] = rotl(s[24], 14);
    for (int y = 0; y < 25; y += 5) {
      s[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
      s[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
      s[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
      s[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
      s[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
    }
    s[0] ^= KRC[r];
  }
}
#if defined(__x86_64__)
// The same permutation on AVX-512 (the GPU boxes' EPYC 9575F and most server parts have it; chosen at run time): the state as five
// PLANE registers (register y, element x = lane A[x + 5y]), ~40 instructions a round instead of ~250 scalar ones with the state
// spilled to memory -- the transcript is a dependent chain of these (32 768 commitments of a 2^14-shuffle, 65 536 generator-chain
// steps), so the permutation's latency is the host's.  theta and rho act inside / across the plane registers; pi is a per-register
// permutation that leaves the NEW state column-major (register X, element Y), where chi is one three-input logic op across
// registers; a 5 x 5 transpose (unpack + two-source permutes) restores the plane form.
__attribute__((target("avx512f,avx512vl"))) void permute_avx512(uint64_t s[25]) {
  const __mmask8 M5 = 0x1F;
  __m512i P0 = _mm512_maskz_loadu_epi64(M5, s), P1 = _mm512_maskz_loadu_epi64(M5, s + 5), P2 = _mm512_maskz_loadu_epi64(M5, s + 10),
          P3 = _mm512_maskz_loadu_epi64(M5, s + 15), P4 = _mm512_maskz_loadu_epi64(M5, s + 20);
  const __m512i IDX_M = _mm512_setr_epi64(4, 0, 1, 2, 3, 5, 6, 7), IDX_P = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7);
  // rho: rotation of lane (x, y), register y
  const __m512i R0 = _mm512_setr_epi64(0, 1, 62, 28, 27, 0, 0, 0), R1 = _mm512_setr_epi64(36, 44, 6, 55, 20, 0, 0, 0),
                R2 = _mm512_setr_epi64(3, 10, 43, 25, 39, 0, 0, 0), R3 = _mm512_setr_epi64(41, 45, 15, 21, 8, 0, 0, 0),
                R4 = _mm512_setr_epi64(18, 2, 61, 56, 14, 0, 0, 0);
  // pi: B[X, Y] = A[x, y = X] with x = (3 Y + X) mod 5 -> T_X[Y] = P_X[(3 Y + X) mod 5]
  const __m512i T0 = _mm512_setr_epi64(0, 3, 1, 4, 2, 5, 6, 7), T1 = _mm512_setr_epi64(1, 4, 2, 0, 3, 5, 6, 7),
                T2 = _mm512_setr_epi64(2, 0, 3, 1, 4, 5, 6, 7), T3 = _mm512_setr_epi64(3, 1, 4, 2, 0, 5, 6, 7),
                T4 = _mm512_setr_epi64(4, 2, 0, 3, 1, 5, 6, 7);
  // transpose back: P_y = {E0[y], E1[y], E2[y], E3[y], E4[y]} from the unpacked pairs (indices 0..7 first source, 8..15 second)
  const __m512i Q01 = _mm512_setr_epi64(0, 1, 8, 9, 0, 0, 0, 0), Q23 = _mm512_setr_epi64(2, 3, 10, 11, 0, 0, 0, 0),
                Q45 = _mm512_setr_epi64(4, 5, 12, 13, 0, 0, 0, 0);
  const __m512i S0 = _mm512_set1_epi64(0), S1 = _mm512_set1_epi64(1), S2 = _mm512_set1_epi64(2), S3 = _mm512_set1_epi64(3),
                S4 = _mm512_set1_epi64(4);
  const __mmask8 M4 = 0x10;      // element 4
  for (int r = 0; r < 24; r++) {
    // theta
    __m512i C = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(P0, P1, P2, 0x96), P3, P4, 0x96);
    const __m512i Cm = _mm512_permutexvar_epi64(IDX_M, C), Cp = _mm512_rol_epi64(_mm512_permutexvar_epi64(IDX_P, C), 1);
    P0 = _mm512_ternarylogic_epi64(P0, Cm, Cp, 0x96); P1 = _mm512_ternarylogic_epi64(P1, Cm, Cp, 0x96);
    P2 = _mm512_ternarylogic_epi64(P2, Cm, Cp, 0x96); P3 = _mm512_ternarylogic_epi64(P3, Cm, Cp, 0x96);
    P4 = _mm512_ternarylogic_epi64(P4, Cm, Cp, 0x96);
    // rho, pi
    const __m512i B0 = _mm512_permutexvar_epi64(T0, _mm512_rolv_epi64(P0, R0)), B1 = _mm512_permutexvar_epi64(T1, _mm512_rolv_epi64(P1, R1)),
                  B2 = _mm512_permutexvar_epi64(T2, _mm512_rolv_epi64(P2, R2)), B3 = _mm512_permutexvar_epi64(T3, _mm512_rolv_epi64(P3, R3)),
                  B4 = _mm512_permutexvar_epi64(T4, _mm512_rolv_epi64(P4, R4));
    // chi (a ^ (~b & c) = 0xD2), iota on lane (0, 0)
    __m512i E0 = _mm512_ternarylogic_epi64(B0, B1, B2, 0xD2);
    const __m512i E1 = _mm512_ternarylogic_epi64(B1, B2, B3, 0xD2), E2 = _mm512_ternarylogic_epi64(B2, B3, B4, 0xD2),
                  E3 = _mm512_ternarylogic_epi64(B3, B4, B0, 0xD2), E4 = _mm512_ternarylogic_epi64(B4, B0, B1, 0xD2);
    E0 = _mm512_xor_si512(E0, _mm512_maskz_set1_epi64(1, (long long)KRC[r]));
    // column-major (register X, element Y) -> planes
    const __m512i u01l = _mm512_unpacklo_epi64(E0, E1), u01h = _mm512_unpackhi_epi64(E0, E1), u23l = _mm512_unpacklo_epi64(E2, E3),
                  u23h = _mm512_unpackhi_epi64(E2, E3);
    P0 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u01l, Q01, u23l), M4, S0, E4);
    P1 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u01h, Q01, u23h), M4, S1, E4);
    P2 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u01l, Q23, u23l), M4, S2, E4);
    P3 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u01h, Q23, u23h), M4, S3, E4);
    P4 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u01l, Q45, u23l), M4, S4, E4);
  }
  _mm512_mask_storeu_epi64(s, M5, P0); _mm512_mask_storeu_epi64(s + 5, M5, P1); _mm512_mask_storeu_epi64(s + 10, M5, P2);
  _mm512_mask_storeu_epi64(s + 15, M5, P3); _mm512_mask_storeu_epi64(s + 20, M5, P4);
}
const bool HAVE_AVX512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && !getenv("BPH_KECCAK_SCALAR");
inline void permute(uint64_t s[25]) { if (HAVE_AVX512) permute_avx512(s); else permute_scalar(s); }
#else
inline void permute(uint64_t s[25]) { permute_scalar(s); }
#endif
void permute_words(uint64_t s[25]) { permute(s); }
std::vector<uint8_t> pad_label(const std::string &l) {   // merlin fork pad_label: source absent; see DESIGN.md
  size_t k = (l.size() + 31) / 32 * 32;
  if (k < 32) k = 32;
  std::vector<uint8_t> o(k, 0);
  memcpy(o.data(), l.data(), l.size());
  return o;
}
}  // namespace
namespace {
// keccak256 as a stream: bytes are XORed into the state where they fall (little-endian host: byte i of the rate is byte i of the
// state words), so a message assembled from pieces needs no buffer
struct Keccak256 {
  static constexpr size_t RATE = 136;
  uint64_t A[25];
  size_t pos = 0;
  Keccak256() { memset(A, 0, sizeof A); }
  void absorb(const uint8_t *in, size_t len) {
    uint8_t *st = (uint8_t *)A;
    while (len) {
      size_t take = std::min(len, RATE - pos);
      if (take == RATE) {                    // whole block: word by word
        for (size_t i = 0; i < RATE / 8; i++) { uint64_t w; memcpy(&w, in + 8 * i, 8); A[i] ^= w; }
      } else {                               // a piece of a block: eight bytes at a time (unaligned accesses are fine on x86-64)
        size_t i = 0;
        for (; i + 8 <= take; i += 8) { uint64_t w, t; memcpy(&w, in + i, 8); memcpy(&t, st + pos + i, 8); t ^= w; memcpy(st + pos + i, &t, 8); }
        for (; i < take; i++) st[pos + i] ^= in[i];
      }
      pos += take; in += take; len -= take;
      if (pos == RATE) { permute(A); pos = 0; }
    }
  }
  void zeros(size_t len) {                  // absorbing zero bytes only moves the position
    while (len) {
      size_t take = std::min(len, RATE - pos);
      pos += take; len -= take;
      if (pos == RATE) { permute(A); pos = 0; }
    }
  }
  void finish(uint8_t out[32]) {            // original Keccak padding 0x01 .. 0x80
    uint8_t *st = (uint8_t *)A;
    st[pos] ^= 0x01;
    st[RATE - 1] ^= 0x80;
    permute(A);
    memcpy(out, A, 32);
  }
  void label(const std::string &l) {        // pad_label(l) without materialising it
    size_t k = (l.size() + 31) / 32 * 32;
    if (k < 32) k = 32;
    absorb((const uint8_t *)l.data(), l.size());
    zeros(k - l.size());
  }
};
}  // namespace
void keccak256(const uint8_t *in, size_t len, uint8_t out[32]) {   // original Keccak padding 0x01
  Keccak256 k;
  k.absorb(in, len);
  k.finish(out);
}
Scalar hash_to_scalar(const uint8_t low[32]) {
  uint8_t buf[64];
  memcpy(buf, low, 32);
  keccak256(low, 32, buf + 32);
  return Scalar::from_le_bytes_mod_order_wide(buf);
}
Transcript::Transcript(const std::string &label) {
  auto a = pad_label("bp-hashchain-v1"), b = pad_label(label);
  a.insert(a.end(), b.begin(), b.end());
  keccak256(a.data(), a.size(), state_);
}
void Transcript::append_message(const std::string &label, const uint8_t *msg, size_t len) {
  Keccak256 k;                              // state || 0x00 || pad_label(label) || u32le(len) || msg
  const uint8_t tag = 0x00;
  uint8_t lb[4];
  for (int j = 0; j < 4; j++) lb[j] = (uint8_t)((uint64_t)len >> (8 * j));
  k.absorb(state_, 32);
  k.absorb(&tag, 1);
  k.label(label);
  k.absorb(lb, 4);
  k.absorb(msg, len);
  k.finish(state_);
}
void Transcript::append_u64(const std::string &label, uint64_t x) {
  uint8_t b[8];
  for (int j = 0; j < 8; j++) b[j] = (uint8_t)(x >> (8 * j));
  append_message(label, b, 8);
}
void Transcript::challenge_bytes(const std::string &label, uint8_t out[32]) {
  Keccak256 k;                              // state || 0x01 || pad_label(label)
  const uint8_t tag = 0x01;
  k.absorb(state_, 32);
  k.absorb(&tag, 1);
  k.label(label);
  k.finish(state_);
  memcpy(out, state_, 32);
}
static void dom_sep(Transcript &t, const std::string &s) { auto p = pad_label(s); t.append_message("dom-sep", p.data(), p.size()); }
void Transcript::innerproduct_domain_sep(uint64_t n) { dom_sep(*this, "ipp v1"); append_u64("n", n); }   // transcript.rs:70-73
void Transcript::r1cs_domain_sep() { dom_sep(*this, "r1cs v1"); }
void Transcript::r1cs_1phase_domain_sep() { dom_sep(*this, "r1cs-1phase"); }
void Transcript::r1cs_2phase_domain_sep() { dom_sep(*this, "r1cs-2phase"); }
void Transcript::append_scalar(const std::string &label, const Scalar &s) { auto b = s.to_bytes(); append_message(label, b.data(), 32); }
void Transcript::append_point(const std::string &label, const StarkPoint &p) { append_message(label, p.xy.data(), 64); }
void Transcript::validate_and_append_point(const std::string &label, const StarkPoint &p) {
  if (p.is_identity()) throw ProofException(ProofError::VerificationError);   // transcript.rs:101-113
  append_point(label, p);
}
Scalar Transcript::challenge_scalar(const std::string &label) {
  uint8_t b[32];
  challenge_bytes(label, b);
  return hash_to_scalar(b);
}

// ================================================================ generators ======================
PedersenGens::PedersenGens() : B(StarkPoint::generator()), B_blinding(StarkPoint::generator()) {}
StarkPoint PedersenGens::commit(const Scalar &value, const Scalar &blinding) const {
  return Device::default_device().msm({value, blinding}, {B, B_blinding});
}
std::vector<StarkPoint> PedersenGens::commit_batch(const BulletproofGens &bp_gens, const std::vector<Scalar> &values,
                                                   const std::vector<Scalar> &blindings) const {
  if (values.size() != blindings.size()) throw std::invalid_argument("commit_batch: length mismatch");
  size_t nb = values.size();
  std::vector<Scalar> vec(2 * nb);
  for (size_t i = 0; i < nb; i++) { vec[2 * i] = values[i]; vec[2 * i + 1] = blindings[i]; }
  auto bytes = pack_scalars(vec);
  std::vector<uint8_t> o(nb * 64 + 1);
  Device &d = Device::default_device();
  d.check(bpgpu_msm_gens(d.ctx(), bp_gens.device_tables(*this), nb, 0, bytes.data(), o.data()), "bpgpu_msm_gens");
  return unpack_points(o.data(), nb);
}
static std::vector<StarkPoint> generators_chain(char which, uint32_t party, size_t skip, size_t count) {
  // GeneratorsChain::new + fast_forward + next, generators.rs:82-124
  std::string lab = "GeneratorsChain";
  lab.push_back(which);
  for (int j = 0; j < 4; j++) lab.push_back((char)(party >> (8 * j)));
  auto padded = pad_label(lab);
  uint8_t state[32], nx[32];
  keccak256(padded.data(), padded.size(), state);
  // On-disk cache, opt-in: BPH_GENS_CACHE_DIR=<dir> keeps every chain's points in
  // <dir>/gens_<G|H><party>.bin = records of [32-byte chain state BEFORE the element | 64-byte point]; a stretch is served
  // from the file when the record at `skip` carries the very state the hash chain has reached there (so a cache written for
  // another label, transcript definition or curve can never be mistaken for this one), and extended otherwise.
  // 65 536 generators: 90 ms of host hashing + 8 MB of file instead of 2 x 65 536 scalar multiplications.
  std::vector<uint8_t> sc(count * 32), out(count * 64), states(count * 32);
  const char *dir = getenv("BPH_GENS_CACHE_DIR");
  std::string path;
  std::vector<uint8_t> cached;
  if (dir && *dir) {
    path = std::string(dir) + "/gens_" + which + std::to_string(party) + ".bin";
    if (FILE *f = std::fopen(path.c_str(), "rb")) {
      std::fseek(f, 0, SEEK_END);
      long sz = std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      if (sz > 0 && sz % 96 == 0) { cached.resize((size_t)sz); if (std::fread(cached.data(), 1, cached.size(), f) != cached.size()) cached.clear(); }
      std::fclose(f);
    }
  }
  for (size_t i = 0; i < skip; i++) { keccak256(state, 32, nx); memcpy(state, nx, 32); }
  const size_t have = cached.size() / 96;
  size_t served = 0;      // elements [skip, skip + served) come from the file
  for (size_t i = 0; i < count; i++) {
    memcpy(&states[32 * i], state, 32);
    if (served == i && skip + i < have && memcmp(&cached[96 * (skip + i)], state, 32) == 0) {
      memcpy(&out[64 * i], &cached[96 * (skip + i) + 32], 64);
      served = i + 1;
    }
    keccak256(state, 32, nx);
    memcpy(state, nx, 32);
    hash_to_scalar(state).to_bytes_le(&sc[32 * i]);
  }
  if (served < count) {
    Device &d = Device::default_device();
    d.check(bpgpu_generator_mul(d.ctx(), sc.data() + 32 * served, count - served, out.data() + 64 * served), "bpgpu_generator_mul");
    if (!path.empty() && skip + served <= have) {       // extend (or repair from the first mismatch on) atomically
      std::vector<uint8_t> file(cached.begin(), cached.begin() + 96 * (skip + served));
      for (size_t i = served; i < count; i++) {
        file.insert(file.end(), &states[32 * i], &states[32 * i] + 32);
        file.insert(file.end(), &out[64 * i], &out[64 * i] + 64);
      }
      const std::string tmp = path + ".tmp" + std::to_string((unsigned long)getpid());
      if (FILE *f = std::fopen(tmp.c_str(), "wb")) {
        const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
        std::fclose(f);
        if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
      }
    }
  }
  return unpack_points(out.data(), count);
}
BulletproofGens::BulletproofGens(size_t cap, size_t parties) : party_capacity(parties), G_vec_(parties), H_vec_(parties) {
  increase_capacity(cap);
}
BulletproofGens::~BulletproofGens() {
  if (tables_) bpgpu_gens_destroy(Device::default_device().ctx(), tables_);
}
void BulletproofGens::increase_capacity(size_t new_capacity) {
  if (gens_capacity >= new_capacity) return;
  // the 2 * party_capacity hash chains are independent (generators.rs:216-233): one host thread each; the
  // scalar -> point step of every chain is one bpgpu_generator_mul call (thread-safe)
  parallel_for(2 * party_capacity, [&](size_t t) {
    const size_t i = t / 2;
    auto v = generators_chain(t % 2 ? 'H' : 'G', (uint32_t)i, gens_capacity, new_capacity - gens_capacity);
    auto &dst = t % 2 ? H_vec_[i] : G_vec_[i];
    dst.insert(dst.end(), v.begin(), v.end());
  });
  gens_capacity = new_capacity;
}
std::vector<StarkPoint> BulletproofGens::Share::G(size_t n) const {
  const auto &v = gens->G_vec_[share];
  return std::vector<StarkPoint>(v.begin(), v.begin() + std::min(n, v.size()));
}
std::vector<StarkPoint> BulletproofGens::Share::H(size_t n) const {
  const auto &v = gens->H_vec_[share];
  return std::vector<StarkPoint>(v.begin(), v.begin() + std::min(n, v.size()));
}
bpgpu_gens *BulletproofGens::device_tables(const PedersenGens &pc, int window_bits) const {
  if (window_bits == 0) {
    if (const char *e = getenv("BPH_WINDOW_BITS")) window_bits = atoi(e);
    if (window_bits == 0) window_bits = gens_capacity <= 1024 ? 16 : (gens_capacity <= 4096 ? 12 : 8);
  }
  std::array<uint8_t, 128> key;
  memcpy(key.data(), pc.B.xy.data(), 64);
  memcpy(key.data() + 64, pc.B_blinding.xy.data(), 64);
  Device &d = Device::default_device();
  if (tables_ && tables_cap_ == gens_capacity && tables_pc_ == key) return tables_;
  if (tables_) { bpgpu_gens_destroy(d.ctx(), tables_); tables_ = nullptr; }
  auto g = pack_points(G_vec_[0]), h = pack_points(H_vec_[0]);
  d.check(bpgpu_gens_create(d.ctx(), g.data(), h.data(), gens_capacity, pc.B.xy.data(), pc.B_blinding.xy.data(), window_bits, &tables_),
          "bpgpu_gens_create");
  tables_cap_ = gens_capacity;
  tables_pc_ = key;
  return tables_;
}

// ================================================================ wire codec =========
std::vector<uint8_t> compress_points(const std::vector<StarkPoint> &pts) {
  std::vector<uint8_t> out(pts.size() * 32);
  if (pts.empty()) return out;
  Device &d = Device::default_device();
  auto xy = pack_points(pts);
  d.check(bpgpu_points_compress(d.ctx(), xy.data(), pts.size(), out.data()), "bpgpu_points_compress");
  return out;
}
std::vector<StarkPoint> decompress_points(const uint8_t *b, size_t n) {
  std::vector<StarkPoint> pts(n);
  if (!n) return pts;
  Device &d = Device::default_device();
  std::vector<uint8_t> xy(n * 64);
  std::vector<int32_t> ok(n);
  d.check(bpgpu_points_decompress(d.ctx(), b, n, xy.data(), ok.data()), "bpgpu_points_decompress");
  for (size_t i = 0; i < n; i++) {
    if (!ok[i]) throw ProofException(ProofError::FormatError);
    memcpy(pts[i].xy.data(), &xy[64 * i], 64);
  }
  return pts;
}
std::vector<uint8_t> InnerProductProof::to_bytes() const {                                // :387-398
  std::vector<StarkPoint> pts;
  for (size_t i = 0; i < L_vec.size(); i++) { pts.push_back(L_vec[i]); pts.push_back(R_vec[i]); }
  std::vector<uint8_t> out = compress_points(pts);
  out.resize(out.size() + 64);
  a.to_bytes_be(&out[out.size() - 64]);
  b.to_bytes_be(&out[out.size() - 32]);
  return out;
}
InnerProductProof InnerProductProof::from_bytes(const uint8_t *s, size_t len) {           // :419-455
  if (len < 64 || len % 32) throw ProofException(ProofError::FormatError);
  size_t num_points = (len - 64) / 32;
  if (num_points % 2) throw ProofException(ProofError::FormatError);
  size_t lg_n = num_points / 2;
  if (lg_n >= 32) throw ProofException(ProofError::FormatError);
  auto pts = decompress_points(s, num_points);
  InnerProductProof p;
  for (size_t i = 0; i < lg_n; i++) { p.L_vec.push_back(pts[2 * i]); p.R_vec.push_back(pts[2 * i + 1]); }
  p.a = Scalar::from_be_bytes_mod_order(s + 64 * lg_n);
  p.b = Scalar::from_be_bytes_mod_order(s + 64 * lg_n + 32);
  return p;
}
namespace r1cs {
static bool missing_phase2(const R1CSProof &p) { return p.A_I2.is_identity() && p.A_O2.is_identity() && p.S2.is_identity(); }   // proof.rs:121-123
size_t R1CSProof::serialized_size() const { return 1 + (missing_phase2(*this) ? 11 : 14) * 32 + ipp_proof.serialized_size(); }
std::vector<uint8_t> R1CSProof::to_bytes() const {
  const bool one = missing_phase2(*this);
  std::vector<StarkPoint> pts{A_I1, A_O1, S1};
  if (!one) { pts.push_back(A_I2); pts.push_back(A_O2); pts.push_back(S2); }
  for (auto *q : {&T_1, &T_3, &T_4, &T_5, &T_6}) pts.push_back(*q);
  const size_t head = pts.size();
  for (size_t i = 0; i < ipp_proof.L_vec.size(); i++) { pts.push_back(ipp_proof.L_vec[i]); pts.push_back(ipp_proof.R_vec[i]); }
  auto cp = compress_points(pts);     // all points of the proof in one device call
  std::vector<uint8_t> out;
  out.reserve(serialized_size());
  out.push_back(one ? 0 : 1);         // ONE_PHASE_COMMITMENTS / TWO_PHASE_COMMITMENTS
  out.insert(out.end(), cp.begin(), cp.begin() + head * 32);
  uint8_t sb[32];
  for (const Scalar *x : {&t_x, &t_x_blinding, &e_blinding}) { x->to_bytes_be(sb); out.insert(out.end(), sb, sb + 32); }
  out.insert(out.end(), cp.begin() + head * 32, cp.end());
  ipp_proof.a.to_bytes_be(sb); out.insert(out.end(), sb, sb + 32);
  ipp_proof.b.to_bytes_be(sb); out.insert(out.end(), sb, sb + 32);
  return out;
}
R1CSProof R1CSProof::from_bytes(const uint8_t *s, size_t len) {
  if (!len) throw R1CSException(R1CSError::FormatError);
  const uint8_t version = s[0];
  s++; len--;
  if (len % 32) throw R1CSException(R1CSError::FormatError);
  if (version > 1) throw R1CSException(R1CSError::FormatError);
  const size_t head = version == 0 ? 8 : 11;
  if (len < (head + 3) * 32) throw R1CSException(R1CSError::FormatError);
  const size_t ipp_len = len - (head + 3) * 32;
  // InnerProductProof::from_bytes' own length checks (inner_product_proof.rs:419-436)
  if (ipp_len < 64 || ((ipp_len - 64) / 32) % 2 || (ipp_len - 64) / 64 >= 32) throw R1CSException(R1CSError::FormatError);
  const size_t npts_ipp = (ipp_len - 64) / 32;
  // gather every compressed point of the proof for one device decompression
  std::vector<uint8_t> cp(s, s + head * 32);
  const uint8_t *ipp = s + (head + 3) * 32;
  cp.insert(cp.end(), ipp, ipp + npts_ipp * 32);
  std::vector<StarkPoint> pts;
  try { pts = decompress_points(cp.data(), head + npts_ipp); } catch (const ProofException &) { throw R1CSException(R1CSError::FormatError); }
  R1CSProof p;
  size_t k = 0;
  p.A_I1 = pts[k++]; p.A_O1 = pts[k++]; p.S1 = pts[k++];
  if (version == 1) { p.A_I2 = pts[k++]; p.A_O2 = pts[k++]; p.S2 = pts[k++]; }
  p.T_1 = pts[k++]; p.T_3 = pts[k++]; p.T_4 = pts[k++]; p.T_5 = pts[k++]; p.T_6 = pts[k++];
  p.t_x = Scalar::from_be_bytes_mod_order(s + head * 32);
  p.t_x_blinding = Scalar::from_be_bytes_mod_order(s + head * 32 + 32);
  p.e_blinding = Scalar::from_be_bytes_mod_order(s + head * 32 + 64);
  for (size_t i = 0; i < npts_ipp / 2; i++) { p.ipp_proof.L_vec.push_back(pts[k++]); p.ipp_proof.R_vec.push_back(pts[k++]); }
  p.ipp_proof.a = Scalar::from_be_bytes_mod_order(ipp + npts_ipp * 32);
  p.ipp_proof.b = Scalar::from_be_bytes_mod_order(ipp + npts_ipp * 32 + 32);
  return p;
}
}  // namespace r1cs

// ================================================================ util / inner product =============
namespace util {
std::vector<Scalar> exp_iter(const Scalar &x, size_t n) {
  std::vector<Scalar> o(n);
  Scalar cur = Scalar::one();
  for (size_t i = 0; i < n; i++) { o[i] = cur; cur *= x; }
  return o;
}
Scalar sum_of_powers_slow(const Scalar &x, size_t n) {
  Scalar acc, cur = Scalar::one();
  for (size_t i = 0; i < n; i++) { acc += cur; cur *= x; }
  return acc;
}
Scalar sum_of_powers(const Scalar &x, size_t n) {
  if (n & (n - 1)) return sum_of_powers_slow(x, n);
  if (n == 0 || n == 1) return Scalar::from(n);
  size_t m = n;
  Scalar result = Scalar::one() + x, factor = x;
  while (m > 2) { factor = factor * factor; result = result + factor * result; m /= 2; }
  return result;
}
}  // namespace util

Scalar inner_product(const std::vector<Scalar> &a, const std::vector<Scalar> &b) {
  if (a.size() != b.size()) throw std::invalid_argument("inner_product(a,b): lengths of vectors do not match");
  Device &d = Device::default_device();
  auto pa = pack_scalars(a), pb = pack_scalars(b);
  uint8_t out[32];
  d.check(bpgpu_inner_product(d.ctx(), pa.data(), pb.data(), a.size(), out), "bpgpu_inner_product");
  return Scalar::from_bytes_le(out);
}

InnerProductProof InnerProductProof::create(Transcript &transcript, const StarkPoint &Q, const std::vector<Scalar> &G_factors,
                                            const std::vector<Scalar> &H_factors, std::vector<StarkPoint> G_vec,
                                            std::vector<StarkPoint> H_vec, std::vector<Scalar> a_vec, std::vector<Scalar> b_vec) {
  size_t n = G_vec.size();
  if (H_vec.size() != n || a_vec.size() != n || b_vec.size() != n || G_factors.size() != n || H_factors.size() != n)
    throw std::invalid_argument("InnerProductProof::create: length mismatch");          // asserts :62-67
  if (!n || (n & (n - 1))) throw std::invalid_argument("InnerProductProof::create: n must be a power of two");   // :70
  transcript.innerproduct_domain_sep(n);                                                  // :72
  Device &d = Device::default_device();
  InnerProductProof proof;
  bpgpu_ipp *s = nullptr;
  auto pG = pack_points(G_vec), pH = pack_points(H_vec);
  auto pa = pack_scalars(a_vec), pb = pack_scalars(b_vec), pgf = pack_scalars(G_factors), phf = pack_scalars(H_factors);
  d.check(bpgpu_ipp_begin(d.ctx(), 1, n, Q.xy.data(), pgf.data(), phf.data(), pG.data(), pH.data(), 1, pa.data(), pb.data(), &s),
          "bpgpu_ipp_begin");
  try {
    if (!getenv("BPH_HOST_IPP_TRANSCRIPT")) {   // all rounds on the device, hash chain included (bpgpu_ipp_run_fs)
      size_t k = 0;
      for (size_t t = n; t > 1; t >>= 1) k++;
      std::vector<uint8_t> L(k * 64 + 1), R(k * 64 + 1);
      uint8_t st_out[32], a[32], b[32];
      d.check(bpgpu_ipp_run_fs(d.ctx(), s, transcript.state(), L.data(), R.data(), a, b, st_out), "bpgpu_ipp_run_fs");
      transcript.set_state(st_out);
      for (size_t r = 0; r < k; r++) {
        StarkPoint Lp, Rp;
        memcpy(Lp.xy.data(), &L[64 * r], 64);
        memcpy(Rp.xy.data(), &R[64 * r], 64);
        proof.L_vec.push_back(Lp);
        proof.R_vec.push_back(Rp);
      }
      proof.a = Scalar::from_bytes_le(a);
      proof.b = Scalar::from_bytes_le(b);
      bpgpu_ipp_destroy(d.ctx(), s);
      return proof;
    }
    while (bpgpu_ipp_len(s) > 1) {
      StarkPoint L, R;
      d.check(bpgpu_ipp_round(d.ctx(), s, L.xy.data(), R.xy.data()), "bpgpu_ipp_round");
      proof.L_vec.push_back(L);
      proof.R_vec.push_back(R);
      transcript.append_point("L", L);                                                    // :119-123 / :177-181
      transcript.append_point("R", R);
      Scalar u = transcript.challenge_scalar("u"), u_inv = u.inverse();
      auto bu = u.to_bytes(), bi = u_inv.to_bytes();
      d.check(bpgpu_ipp_fold(d.ctx(), s, bu.data(), bi.data()), "bpgpu_ipp_fold");
    }
    uint8_t a[32], b[32];
    d.check(bpgpu_ipp_finish(d.ctx(), s, a, b), "bpgpu_ipp_finish");
    proof.a = Scalar::from_bytes_le(a);
    proof.b = Scalar::from_bytes_le(b);
  } catch (...) {
    bpgpu_ipp_destroy(d.ctx(), s);
    throw;
  }
  bpgpu_ipp_destroy(d.ctx(), s);
  return proof;
}

InnerProductProof::VerificationScalars InnerProductProof::verification_scalars(size_t n, Transcript &transcript,
                                                                               std::vector<Scalar> *challenges_out) const {
  size_t lg_n = L_vec.size();
  if (lg_n >= 32) throw ProofException(ProofError::VerificationError);                    // :259-264
  if (n != ((size_t)1 << lg_n)) throw ProofException(ProofError::VerificationError);      // :265-267
  transcript.innerproduct_domain_sep(n);
  std::vector<Scalar> ch;
  for (size_t i = 0; i < lg_n; i++) {                                                     // :273-278
    transcript.validate_and_append_point("L", L_vec[i]);
    transcript.validate_and_append_point("R", R_vec.at(i));
    ch.push_back(transcript.challenge_scalar("u"));
  }
  if (challenges_out) *challenges_out = ch;
  Device &d = Device::default_device();
  auto pc = pack_scalars(ch);
  std::vector<uint8_t> us(lg_n * 32 + 1), uis(lg_n * 32 + 1), s(n * 32);
  d.check(bpgpu_verification_scalars(d.ctx(), pc.data(), lg_n, n, us.data(), uis.data(), s.data()), "bpgpu_verification_scalars");
  VerificationScalars v;
  v.u_sq = unpack_scalars(us.data(), lg_n);
  v.u_inv_sq = unpack_scalars(uis.data(), lg_n);
  v.s = unpack_scalars(s.data(), n);
  return v;
}

void InnerProductProof::verify(size_t n, Transcript &transcript, const std::vector<Scalar> &G_factors,
                               const std::vector<Scalar> &H_factors, const StarkPoint &P, const StarkPoint &Q,
                               const std::vector<StarkPoint> &G, const std::vector<StarkPoint> &H) const {
  auto v = verification_scalars(n, transcript);
  std::vector<Scalar> sc;
  std::vector<StarkPoint> pts;
  sc.push_back(a * b);
  pts.push_back(Q);
  for (size_t i = 0; i < n && i < G.size(); i++) { sc.push_back((a * v.s[i]) * G_factors.at(i)); pts.push_back(G[i]); }   // :336-340
  for (size_t i = 0; i < n && i < H.size(); i++) { sc.push_back((b * v.s[n - 1 - i]) * H_factors.at(i)); pts.push_back(H[i]); }   // :343-348
  for (size_t i = 0; i < L_vec.size(); i++) { sc.push_back(-v.u_sq[i]); pts.push_back(L_vec[i]); }
  for (size_t i = 0; i < R_vec.size(); i++) { sc.push_back(-v.u_inv_sq[i]); pts.push_back(R_vec[i]); }
  StarkPoint expect_P = Device::default_device().msm(sc, pts);
  if (expect_P != P) throw ProofException(ProofError::VerificationError);
}

// ================================================================ r1cs ============================
namespace r1cs {

void LinearCombination::add_term(const Variable &v, const Scalar &c) {
  auto it = terms.find(v);
  if (it == terms.end()) terms[v] = c; else it->second = it->second + c;
}
LinearCombination LinearCombination::operator+(const LinearCombination &o) const {
  LinearCombination r = *this;
  for (auto &kv : o.terms) r.add_term(kv.first, kv.second);
  return r;
}
LinearCombination LinearCombination::operator-() const {
  LinearCombination r;
  for (auto &kv : terms) r.terms[kv.first] = -kv.second;
  return r;
}
LinearCombination LinearCombination::operator-(const LinearCombination &o) const {   // (no negated temporary: gadgets subtract a lot)
  LinearCombination r = *this;
  for (auto &kv : o.terms) {
    auto it = r.terms.find(kv.first);
    if (it == r.terms.end()) r.terms[kv.first] = -kv.second; else it->second = it->second - kv.second;
  }
  return r;
}
LinearCombination LinearCombination::operator*(const Scalar &s) const {
  LinearCombination r;
  for (auto &kv : terms) r.terms[kv.first] = kv.second * s;
  return r;
}

std::vector<uint8_t> R1CSProof::to_flat_bytes() const {
  size_t k = ipp_proof.L_vec.size();
  std::vector<uint8_t> o(8 + 11 * 64 + 96 + 128 * k + 64, 0);
  for (int j = 0; j < 4; j++) o[j] = (uint8_t)(k >> (8 * j));
  uint8_t *p = o.data() + 8;
  const StarkPoint *pts[11] = {&A_I1, &A_O1, &S1, &A_I2, &A_O2, &S2, &T_1, &T_3, &T_4, &T_5, &T_6};
  for (auto *q : pts) { memcpy(p, q->xy.data(), 64); p += 64; }
  t_x.to_bytes_le(p); t_x_blinding.to_bytes_le(p + 32); e_blinding.to_bytes_le(p + 64); p += 96;
  for (auto &q : ipp_proof.L_vec) { memcpy(p, q.xy.data(), 64); p += 64; }
  for (auto &q : ipp_proof.R_vec) { memcpy(p, q.xy.data(), 64); p += 64; }
  ipp_proof.a.to_bytes_le(p); ipp_proof.b.to_bytes_le(p + 32);
  return o;
}
R1CSProof R1CSProof::from_flat_bytes(const std::vector<uint8_t> &b) {
  if (b.size() < 8) throw R1CSException(R1CSError::FormatError);
  size_t k = b[0] | (b[1] << 8) | (b[2] << 16) | ((size_t)b[3] << 24);
  if (k >= 32 || b.size() != 8 + 11 * 64 + 96 + 128 * k + 64) throw R1CSException(R1CSError::FormatError);
  R1CSProof pr;
  const uint8_t *p = b.data() + 8;
  StarkPoint *pts[11] = {&pr.A_I1, &pr.A_O1, &pr.S1, &pr.A_I2, &pr.A_O2, &pr.S2, &pr.T_1, &pr.T_3, &pr.T_4, &pr.T_5, &pr.T_6};
  for (auto *q : pts) { memcpy(q->xy.data(), p, 64); p += 64; }
  try {
    pr.t_x = Scalar::from_bytes_le(p); pr.t_x_blinding = Scalar::from_bytes_le(p + 32); pr.e_blinding = Scalar::from_bytes_le(p + 64);
    p += 96;
    pr.ipp_proof.L_vec = unpack_points(p, k); p += 64 * k;
    pr.ipp_proof.R_vec = unpack_points(p, k); p += 64 * k;
    pr.ipp_proof.a = Scalar::from_bytes_le(p); pr.ipp_proof.b = Scalar::from_bytes_le(p + 32);
  } catch (const ProofException &) { throw R1CSException(R1CSError::FormatError); }
  return pr;
}

// ---- shared constraint-system core ---------------------------------------------------------------
// The constraint rows, append-only, in blocks that never move: a std::vector of 160-byte rows re-allocates (and moves every row
// it holds) a dozen times while the 2^14-shuffle pushes its 65 533 rows -- a fifth of the gadget's time.
class RowStore {
 public:
  static constexpr size_t BLOCK = 256;     // 40 KB: below malloc's mmap threshold, so the blocks of the next proof come from the freed ones (no page faults)
  void push_back(LinearCombination &&lc) {
    if (n_ == blocks_.size() * BLOCK) { std::unique_ptr<LinearCombination[]> b(new LinearCombination[BLOCK]); blocks_.push_back(std::move(b)); }
    blocks_[n_ / BLOCK][n_ % BLOCK] = std::move(lc);
    n_++;
  }
  size_t size() const { return n_; }
  const LinearCombination &operator[](size_t i) const { return blocks_[i / BLOCK][i % BLOCK]; }
  struct const_iterator {
    const RowStore *s; size_t i;
    const LinearCombination &operator*() const { return (*s)[i]; }
    const_iterator &operator++() { ++i; return *this; }
    bool operator!=(const const_iterator &o) const { return i != o.i; }
  };
  const_iterator begin() const { return {this, 0}; }
  const_iterator end() const { return {this, n_}; }
 private:
  std::vector<std::unique_ptr<LinearCombination[]>> blocks_;
  size_t n_ = 0;
};
static const Scalar kOne = Scalar::one(), kMinusOne = -Scalar::one();
class CsCore {
 public:
  CsCore(bool prover, const PedersenGens &pc, Transcript &t, RandomizedConstraintSystem *self)
      : is_prover(prover), pc_gens(pc), tr(t), self_(self) { tr.r1cs_domain_sep(); }
  bool is_prover;
  PedersenGens pc_gens;
  Transcript &tr;
  RandomizedConstraintSystem *self_;
  // ParametricCircuit's probe runs: the gadget challenges come from here instead of the transcript, their labels are recorded
  const std::vector<Scalar> *probe_chi = nullptr;
  std::vector<std::string> *probe_labels = nullptr;
  size_t probe_next = 0;
  // a PROVER bound to a ParametricCircuit (Prover::use_circuit): the gadgets run for their witness only -- constraint rows are not
  // stored, hashed or uploaded (the circuit holds them) -- and the gadget challenges drawn from the transcript are kept for the device
  const ParametricCircuit *param = nullptr;
  std::vector<Scalar> chi_drawn;
  size_t rows_dropped = 0;
  RowStore constraints;
  // 128-bit running hash of the rows (variables + coefficients, in row order), updated as they are pushed: lock-step provers
  // must share their constraint rows, and comparing 255 x 2064 rows with the first prover's cost every batch as much as
  // building them (BPH_CHECK_ROWS=1 still does it); the hash also keys the cache of uploaded circuits
  // The hash is KEYED: ten words drawn once per process from the OS (row_hash_key()) replace the additive / xor constants of the
  // term mixing and the initial state, so that coefficients derived from untrusted input cannot be chosen offline to collide
  // (an unkeyed multiply-rotate hash with public constants invites that; a collision would prove against another circuit's rows).
  uint64_t rows_hash[2] = {row_hash_key()[8], row_hash_key()[9]};
  size_t rows_nnz = 0;
  // (one dependent multiply-rotate step per TERM; the five words of a term are mixed by independent multiplications -- hashing
  // word by word was a chain of 26 dependent multiplications per row, a third of the time the 2^14-shuffle's gadget takes)
  static uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
  const uint64_t *hkey_ = row_hash_key();
  void hash_term(uint64_t v, const uint64_t w[4]) {
    const uint64_t *k = hkey_;
    const uint64_t a = (v + k[0]) * 0xD6E8FEB86659FD93ULL ^ rotl((w[0] ^ k[1]) * 0xC2B2AE3D27D4EB4FULL, 13) ^
                       rotl((w[1] + k[2]) * 0x9FB21C651E98DF25ULL, 26) ^ rotl((w[2] ^ k[3]) * 0xFF51AFD7ED558CCDULL, 39) ^
                       rotl((w[3] + k[4]) * 0xC4CEB9FE1A85EC53ULL, 52);
    const uint64_t b = (v ^ k[5]) * 0x94D049BB133111EBULL + rotl((w[0] + k[6]) * 0xBF58476D1CE4E5B9ULL, 17) +
                       rotl((w[1] ^ k[7]) * 0xE7037ED1A0B428DBULL, 31) + rotl((w[2] + k[3]) * 0x8EBC6AF09C88C6E3ULL, 43) +
                       rotl((w[3] ^ k[1]) * 0x589965CC75374CC3ULL, 55);
    rows_hash[0] = rotl(rows_hash[0] ^ a, 29) * 0x9E3779B97F4A7C15ULL;
    rows_hash[1] = rotl(rows_hash[1] + b, 31) * 0xC2B2AE3D27D4EB4FULL;
  }
  void push_row(LinearCombination &&lc) {
    if (param) { rows_dropped++; return; }
    const uint64_t head[4] = {lc.terms.size(), 0, 0, 0};
    hash_term(0xA5A5A5A500000000ULL, head);
    for (auto &kv : lc.terms) {
      if (kv.first.kind == Variable::Zero) continue;
      uint64_t w[4];
      kv.second.to_ark_le((uint8_t *)w);
      hash_term(((uint64_t)kv.first.kind << 56) ^ (uint64_t)kv.first.index, w);
      rows_nnz++;
    }
    constraints.push_back(std::move(lc));
  }
  std::vector<Scalar> a_L, a_R, a_O, v, v_blinding;    // prover
  std::vector<StarkPoint> V;                            // both (prover keeps them for convenience)
  size_t num_vars = 0;                                  // verifier
  std::vector<ConstraintSystem::Callback> deferred;
  long pending_multiplier = -1;
  StarkPoint mega;

  size_t multipliers() const { return is_prover ? a_L.size() : num_vars; }
  Scalar eval(const LinearCombination &lc) const {      // prover.rs:179-194 / verifier.rs:168-174
    Scalar acc;
    if (!is_prover) return acc;
    for (auto &kv : lc.terms) {
      const Scalar *val = nullptr;
      switch (kv.first.kind) {
        case Variable::MultiplierLeft: val = &a_L.at(kv.first.index); break;
        case Variable::MultiplierRight: val = &a_R.at(kv.first.index); break;
        case Variable::MultiplierOutput: val = &a_O.at(kv.first.index); break;
        case Variable::Committed: val = &v.at(kv.first.index); break;
        case Variable::One: val = &kOne; break;
        default: continue;
      }
      // gadget rows are mostly +-1 * variable and constant terms: those cost an addition, not a multiplication
      if (val == &kOne) acc += kv.second;
      else if (kv.second == kOne) acc += *val;
      else if (kv.second == kMinusOne) acc -= *val;
      else acc += kv.second * *val;
    }
    return acc;
  }
  std::array<Variable, 3> new_multiplier(const Scalar &l, const Scalar &r) {
    size_t i;
    if (is_prover) { i = a_L.size(); a_L.push_back(l); a_R.push_back(r); a_O.push_back(l * r); }
    else i = num_vars++;
    return {Variable{Variable::MultiplierLeft, i}, Variable{Variable::MultiplierRight, i}, Variable{Variable::MultiplierOutput, i}};
  }
  std::array<Variable, 3> multiply(LinearCombination &&left, LinearCombination &&right) {   // prover.rs:99-125 / verifier.rs:99-120
    auto vars = new_multiplier(eval(left), eval(right));
    if (param) { rows_dropped += 2; return vars; }      // (the two rows live in the parametric circuit)
    left.add_term(vars[0], kMinusOne);
    right.add_term(vars[1], kMinusOne);
    push_row(std::move(left));
    push_row(std::move(right));
    return vars;
  }
  Variable allocate(const Scalar *assignment) {                                         // prover.rs:127-146 / verifier.rs:122-135
    if (is_prover && !assignment) throw R1CSException(R1CSError::MissingAssignment);
    if (pending_multiplier < 0) {
      size_t i = multipliers();
      pending_multiplier = (long)i;
      if (is_prover) { a_L.push_back(*assignment); a_R.push_back(Scalar()); a_O.push_back(Scalar()); }
      else num_vars++;
      return Variable{Variable::MultiplierLeft, i};
    }
    size_t i = (size_t)pending_multiplier;
    pending_multiplier = -1;
    if (is_prover) { a_R[i] = *assignment; a_O[i] = a_L[i] * a_R[i]; }
    return Variable{Variable::MultiplierRight, i};
  }
  void create_randomized_constraints() {                                                // prover.rs:383-402 / verifier.rs:366-385
    pending_multiplier = -1;
    if (deferred.empty()) { tr.r1cs_1phase_domain_sep(); return; }
    tr.r1cs_2phase_domain_sep();
    auto cbs = std::move(deferred);
    deferred.clear();
    for (auto &cb : cbs) cb(*self_);
  }
  bool same_rows(const CsCore &o) const {
    if (constraints.size() != o.constraints.size()) return false;
    for (size_t r = 0; r < constraints.size(); r++) {
      const auto &a = constraints[r].terms, &b = o.constraints[r].terms;
      if (a.size() != b.size()) return false;
      auto ia = a.begin();
      auto ib = b.begin();
      for (; ia != a.end(); ++ia, ++ib)
        if (ia->first < ib->first || ib->first < ia->first || ia->second != ib->second) return false;
    }
    return true;
  }
  // constraint rows -> CSR arrays of the C ABI (the reference's Vec<LinearCombination>)
  // ark = true: coefficients in their in-memory Montgomery form (bpgpu_circuit_create_ark converts them on the device)
  void csr(std::vector<uint32_t> &rp, std::vector<uint32_t> &kind, std::vector<uint32_t> &idx, std::vector<uint8_t> &coeff, bool ark = false) const {
    size_t nnz = 0;
    for (auto &lc : constraints) nnz += lc.terms.size();
    rp.clear(); kind.clear(); idx.clear(); coeff.clear();
    rp.reserve(constraints.size() + 1); kind.reserve(nnz); idx.reserve(nnz); coeff.resize(nnz * 32);
    rp.push_back(0);
    size_t t = 0;
    for (auto &lc : constraints) {
      for (auto &kv : lc.terms) {
        if (kv.first.kind == Variable::Zero) continue;
        kind.push_back(kv.first.kind);
        idx.push_back((uint32_t)kv.first.index);
        if (ark) kv.second.to_ark_le(&coeff[32 * t]); else kv.second.to_bytes_le(&coeff[32 * t]);
        t++;
      }
      rp.push_back((uint32_t)kind.size());
    }
    coeff.resize(t * 32);
  }
  // the same arrays, written straight into page-locked staging memory by the thread pool (the 2^14-shuffle builds and uploads
  // 65 533 rows / 196 600 terms for every proof: its rows carry the gadget's challenge)
  bpgpu_circuit *upload_circuit(size_t n_mul, size_t m) const {
    const size_t q = constraints.size();
    static thread_local RawBuf buf;
    static thread_local std::vector<uint32_t> rp;
    rp.resize(q + 1);
    size_t nnz = 0;
    for (size_t r = 0; r < q; r++) {
      rp[r] = (uint32_t)nnz;
      for (auto &kv : constraints[r].terms) nnz += kv.first.kind != Variable::Zero;
    }
    rp[q] = (uint32_t)nnz;
    const size_t cap = nnz ? nnz : 1;
    uint8_t *base = buf.ensure(cap * 40);
    uint8_t *coeff = base;
    uint32_t *kind = (uint32_t *)(base + cap * 32), *idx = kind + cap;
    kind[0] = idx[0] = 0;
    const size_t CH = 2048, chunks = (q + CH - 1) / CH;
    const uint32_t *row_ptr = rp.data();      // (a lambda does not capture a thread_local: a pool thread would see its own, empty `rp`)
    parallel_for(chunks, [&](size_t c) {
      for (size_t r = c * CH; r < q && r < (c + 1) * CH; r++) {
        size_t t = row_ptr[r];
        for (auto &kv : constraints[r].terms) {
          if (kv.first.kind == Variable::Zero) continue;
          kind[t] = kv.first.kind;
          idx[t] = (uint32_t)kv.first.index;
          kv.second.to_ark_le(coeff + 32 * t);
          t++;
        }
      }
    }, 4);
    Device &d = Device::default_device();
    bpgpu_circuit *c = nullptr;
    d.check(bpgpu_circuit_create_ark(d.ctx(), q, row_ptr, kind, idx, coeff, n_mul, m, &c), "bpgpu_circuit_create_ark");
    return c;
  }
};

static size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

// Uploaded circuits by row hash: a batch of provers -- and the next batch, and the one a second worker thread is preparing --
// share ONE device copy of their constraint rows (building the CSR arrays, transposing and uploading them, and the hipFree of
// the old copy, which waits for the whole device to idle, cost ~1.5 ms per batch of 256 x 2064 rows).  A circuit is plain
// device memory: usable from every context of the device.  Never more than 8 idle entries.
namespace {
struct CircuitCache {
  struct Entry { uint64_t h[2]; size_t n, m, q, nnz; bpgpu_circuit *c; int users; uint64_t stamp; };
  std::mutex mu;
  std::vector<Entry> e;
  uint64_t clock = 0;
  bpgpu_circuit *acquire(const CsCore &cs, size_t n, size_t m) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &x : e)
      if (x.h[0] == cs.rows_hash[0] && x.h[1] == cs.rows_hash[1] && x.n == n && x.m == m && x.q == cs.constraints.size() && x.nnz == cs.rows_nnz) {
        x.users++; x.stamp = ++clock;
        return x.c;
      }
    bpgpu_circuit *c = cs.upload_circuit(n, m);      // (under the lock: a second thread asking for the same circuit waits for it)
    size_t idle = 0, oldest = (size_t)-1;
    for (size_t i = 0; i < e.size(); i++)
      if (e[i].users == 0) { idle++; if (oldest == (size_t)-1 || e[i].stamp < e[oldest].stamp) oldest = i; }
    if (idle >= 8) { bpgpu_circuit_destroy(Device::default_device().ctx(), e[oldest].c); e.erase(e.begin() + (long)oldest); }
    e.push_back(Entry{{cs.rows_hash[0], cs.rows_hash[1]}, n, m, cs.constraints.size(), cs.rows_nnz, c, 1, ++clock});
    return c;
  }
  void release(bpgpu_circuit *c) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &x : e) if (x.c == c) { x.users--; return; }
  }
};
CircuitCache &circuit_cache() { static CircuitCache *c = new CircuitCache(); return *c; }   // (lives until exit, like the pool)
}  // namespace

// BPH_TIMING=1: phase timings of prove_batch / verify on stderr
struct Lap {
  bool on = getenv("BPH_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void operator()(const char *what) {
    if (!on) return;
    auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[bph]   %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count());
    t0 = t;
  }
};

// ---- Prover ----------------------------------------------------------------------------------------
Prover::Prover(const PedersenGens &pc, Transcript &t) : c_(new CsCore(true, pc, t, this)) {}
Prover::~Prover() {}
Transcript &Prover::transcript() { return c_->tr; }
size_t Prover::num_constraints() const { return c_->constraints.size(); }
size_t Prover::num_multipliers() const { return c_->a_O.size(); }
std::array<Variable, 3> Prover::multiply(LinearCombination l, LinearCombination r) { return c_->multiply(std::move(l), std::move(r)); }
Variable Prover::allocate(const Scalar *a) { return c_->allocate(a); }
std::array<Variable, 3> Prover::allocate_multiplier(const std::pair<Scalar, Scalar> *in) {
  if (!in) throw R1CSException(R1CSError::MissingAssignment);                           // prover.rs:152
  return c_->new_multiplier(in->first, in->second);
}
Variable Prover::commit_public(const Scalar &v) { return commit(v, Scalar::one()).second; }   // prover.rs:171-173
void Prover::constrain(LinearCombination lc) { c_->push_row(std::move(lc)); }
Scalar Prover::eval(const LinearCombination &lc) const { return c_->eval(lc); }
void Prover::specify_randomized_constraints(Callback cb) { c_->deferred.push_back(std::move(cb)); }
Scalar Prover::challenge_scalar(const std::string &label) {
  Scalar x = c_->tr.challenge_scalar(label);
  if (c_->param) c_->chi_drawn.push_back(x);
  return x;
}
void Prover::use_circuit(const ParametricCircuit &circuit) {
  if (c_->constraints.size() || c_->a_L.size() || !c_->deferred.empty()) throw std::invalid_argument("Prover::use_circuit: before the gadgets are added");
  c_->param = &circuit;
}
bool Prover::constraints_satisfied() const {
  for (auto &lc : c_->constraints) if (c_->eval(lc) != Scalar::zero()) return false;
  return true;
}
std::pair<StarkPoint, Variable> Prover::commit(const Scalar &v, const Scalar &v_blinding) {
  size_t i = c_->v.size();
  c_->v.push_back(v);
  c_->v_blinding.push_back(v_blinding);
  StarkPoint V = c_->pc_gens.commit(v, v_blinding);
  c_->V.push_back(V);
  c_->tr.append_point("V", V);
  return {V, Variable{Variable::Committed, i}};
}

Variable Prover::commit_precomputed(const Scalar &v, const Scalar &v_blinding, const StarkPoint &V) {
  size_t i = c_->v.size();
  c_->v.push_back(v);
  c_->v_blinding.push_back(v_blinding);
  c_->V.push_back(V);
  c_->tr.append_point("V", V);
  return Variable{Variable::Committed, i};
}

R1CSProof Prover::prove(const BulletproofGens &bp_gens) {
  OsRng rng;
  return prove(bp_gens, rng);
}
R1CSProof Prover::prove(const BulletproofGens &bp_gens, Rng &rng) {
  std::vector<Prover *> ps{this};
  std::vector<Rng *> rs{&rng};
  return prove_batch(ps, bp_gens, rs)[0];
}

// Lock-step Prover::prove (prover.rs:412-727) for nb provers of circuits with identical constraint rows
// (1-phase gadgets, or 2-phase ones whose randomized rows happen to coincide): every device call is
// batched over the provers; only the transcripts run per prover on the host.
R1CSProof Prover::prove(const BulletproofGens &bp_gens, RankGroup &group, Rng *rng, Device *device) {
  std::vector<Prover *> ps{this};
  std::unique_ptr<Rng> own;
  if (!rng) {   // rank 0's entropy for everyone: the ranks must draw the same blinding factors
    OsRng seed_src;
    uint8_t mine[32];
    for (int i = 0; i < 4; i++) { uint64_t w = seed_src.next_u64(); memcpy(mine + 8 * i, &w, 8); }
    std::vector<uint8_t> all(32 * group.size());
    group.all_gather(mine, 32, all.data());
    own.reset(new OsRng(all.data(), true));
    rng = own.get();
  }
  std::vector<Rng *> rs{rng};
  return prove_batch(ps, bp_gens, rs, device, &group)[0];
}
// partial points of the ranks -> their sums (count points per rank, in place in `mine`)
// Returns the first non-zero bpgpu_points_sum code AFTER every column has been tried: every rank sees the same gathered bytes, so
// every rank gets the same code -- a partial that is not a point (the poison encoding a rank's bpgpu_r1cs_verify_shard returns
// for a malformed operand in its share) fails the whole group together, with no rank left waiting in a collective.
static int combine_partials_rc(Device &d, RankGroup &g, uint8_t *mine, size_t count) {
  const size_t w = g.size();
  std::vector<uint8_t> all(w * count * 64), col(w * 64);
  g.all_gather(mine, count * 64, all.data());
  int first = 0;
  for (size_t j = 0; j < count; j++) {
    for (size_t r = 0; r < w; r++) memcpy(&col[64 * r], &all[(r * count + j) * 64], 64);
    const int rc = bpgpu_points_sum(d.ctx(), col.data(), w, mine + 64 * j);
    if (rc && !first) first = rc;
  }
  return first;
}
static void combine_partials(Device &d, RankGroup &g, uint8_t *mine, size_t count) {
  d.check(combine_partials_rc(d, g, mine, count), "bpgpu_points_sum");
}
std::vector<R1CSProof> Prover::prove_batch(std::vector<Prover *> &provers, const BulletproofGens &bp_gens,
                                           std::vector<Rng *> &rngs, Device *device, RankGroup *group) {
  const size_t nb = provers.size();
  if (!nb || rngs.size() != nb) throw std::invalid_argument("prove_batch: one Rng per prover");
  Device &d = same_gpu(device);
  if (group && group->size() <= 1) group = nullptr;
  if (group && nb != 1) throw std::invalid_argument("prove_batch: a rank group shards ONE proof");
  struct ShardGuard {      // the context computes this rank's partial sums for the duration of the call
    Device &d; bool on;
    ~ShardGuard() { if (on) bpgpu_set_shard(d.ctx(), 0, 1); }
  } shard_guard{d, group != nullptr};
  if (group) d.check(bpgpu_set_shard(d.ctx(), group->rank(), group->size()), "bpgpu_set_shard");
  std::vector<CsCore *> cs(nb);
  for (size_t p = 0; p < nb; p++) cs[p] = provers[p]->c_.get();
  const PedersenGens &pc = cs[0]->pc_gens;
  bpgpu_gens *gens = bp_gens.device_tables(pc);
  const size_t n1 = cs[0]->a_L.size(), m = cs[0]->v.size();
  const bool vkeys = rngs[0]->vector_keys();
  for (size_t p = 0; p < nb; p++) {
    CsCore *c = cs[p];
    if (c->a_L.size() != n1 || c->v.size() != m) throw std::invalid_argument("prove_batch: circuits differ in shape");
    if (rngs[p]->vector_keys() != vkeys) throw std::invalid_argument("prove_batch: the provers' Rngs differ in vector_keys()");
    c->tr.append_u64("m", c->v.size());                                                  // prover.rs:420
  }
  if (bp_gens.gens_capacity < n1) throw R1CSException(R1CSError::InvalidGeneratorsLength);   // :450-452
  std::vector<R1CSProof> proofs(nb);
  Lap lap;
  std::vector<Scalar> i_b1(nb), o_b1(nb), s_b1(nb), i_b2(nb), o_b2(nb), s_b2(nb);
  // prover.rs:435-445: the blinding RNG is bound to the transcript state and to the witness blindings
  // (build_rng().rekey_with_witness_bytes("v_blinding", ..)); the OS entropy is already in an OsRng's key
  // (many commitments -- the 2^14-shuffle has 32 768 -- are absorbed as the keccak256 digests of runs of 128 blindings, hashed
  // on the thread pool: one rekey per blinding was 32 768 dependent permutations, 13 ms)
  parallel_for(nb, [&](size_t p) {
    rngs[p]->rekey(cs[p]->tr.state(), 32);
    const auto &vb = cs[p]->v_blinding;
    const size_t RUN = 128, runs = (vb.size() + RUN - 1) / RUN;
    if (runs <= 1) {
      std::vector<uint8_t> all(vb.size() * 32);
      for (size_t i = 0; i < vb.size(); i++) vb[i].to_bytes_le(&all[32 * i]);
      if (!all.empty()) rngs[p]->rekey(all.data(), all.size());
    } else {
      std::vector<uint8_t> dig(runs * 32);
      parallel_for(runs, [&](size_t r) {
        const size_t lo = r * RUN, hi = std::min(vb.size(), lo + RUN);
        std::vector<uint8_t> buf((hi - lo) * 32);
        for (size_t i = lo; i < hi; i++) vb[i].to_bytes_le(&buf[32 * (i - lo)]);
        keccak256(buf.data(), buf.size(), &dig[32 * r]);
      });
      rngs[p]->rekey(dig.data(), dig.size());
    }
  }, 16);
  // One phase of commitments, prover.rs:457-494 (lo = 0) / :519-565 (lo = n1): blinding factors, then A_I, A_O, S over
  // [B_blinding, G_lo.., H_lo..] -- the witness planes go to the device once (bpgpu_r1cs_prover_commit keeps them in the
  // session for the polynomial build), the blinding vectors s_L, s_R either with them or, Rng::vector_keys(), as one 32-byte
  // key per prover that the device expands
  bpgpu_prover *ps = nullptr;
  struct SessionGuard { Device &d; bpgpu_prover *&ps; ~SessionGuard() { if (ps) bpgpu_prover_destroy(d.ctx(), ps); } } session_guard{d, ps};
  auto commit_phase = [&](size_t lo, size_t hi, std::vector<Scalar> &ib, std::vector<Scalar> &ob, std::vector<Scalar> &sb, int which_phase) {
    const size_t cnt = hi - lo, plane = nb * cnt * 32;
    static thread_local RawBuf buf;
    uint8_t *base = buf.ensure((vkeys ? 3 : 5) * plane + nb * 3 * 32 + nb * 32 + 1);
    const uint8_t *paL = base, *paR = base + plane, *paO = base + 2 * plane;
    uint8_t *psL = base + 3 * plane, *psR = base + 4 * plane;
    uint8_t *pbl = base + (vkeys ? 3 : 5) * plane, *pkey = pbl + nb * 3 * 32;
    // one prover: its witness vectors already lie in memory as the planes the call takes (a Scalar IS its four Montgomery words)
    static_assert(sizeof(Scalar) == 32, "Scalar must be its 32 in-memory bytes");
    const bool direct = nb == 1 && cnt > 0;
    if (direct) {
      paL = (const uint8_t *)(cs[0]->a_L.data() + lo); paR = (const uint8_t *)(cs[0]->a_R.data() + lo); paO = (const uint8_t *)(cs[0]->a_O.data() + lo);
    }
    parallel_for(nb, [&](size_t p) {
      Rng &r = *rngs[p];
      ib[p] = r.scalar(); ob[p] = r.scalar(); sb[p] = r.scalar();                         // :457-459 / :519-521
      ib[p].to_ark_le(pbl + (p * 3) * 32); ob[p].to_ark_le(pbl + (p * 3 + 1) * 32); sb[p].to_ark_le(pbl + (p * 3 + 2) * 32);
      if (!direct) {
        pack_range_ark(base + p * cnt * 32, cs[p]->a_L.data() + lo, cnt);
        pack_range_ark(base + plane + p * cnt * 32, cs[p]->a_R.data() + lo, cnt);
        pack_range_ark(base + 2 * plane + p * cnt * 32, cs[p]->a_O.data() + lo, cnt);
      }
      if (!cnt) return;
      if (vkeys) {                                                                        // :461-462 / :526-527, as a key
        for (int i = 0; i < 4; i++) { uint64_t w = r.next_u64(); memcpy(pkey + p * 32 + 8 * i, &w, 8); }   // little-endian host
      } else {                                                 // (inside a running loop, i.e. for nb >= 2, the inner loop runs serially)
        std::vector<Scalar> v(cnt);
        r.scalars(v.data(), cnt);
        pack_range_ark(psL + p * cnt * 32, v.data(), cnt);
        r.scalars(v.data(), cnt);
        pack_range_ark(psR + p * cnt * 32, v.data(), cnt);
      }
    });
    lap("prove:   pack planes");
    std::vector<uint8_t> o(nb * 3 * 64);
    int rc = bpgpu_r1cs_prover_commit(d.ctx(), gens, &ps, nb, cnt, paL, paR, paO, vkeys ? nullptr : psL, vkeys ? nullptr : psR,
                                      vkeys && cnt ? pkey : nullptr, pbl, o.data());
    if (rc == BPGPU_E_GENS) throw R1CSException(R1CSError::InvalidGeneratorsLength);
    d.check(rc, "bpgpu_r1cs_prover_commit");
    if (group) combine_partials(d, *group, o.data(), 3);
    for (size_t p = 0; p < nb; p++) {
      StarkPoint *dst[3] = {which_phase == 1 ? &proofs[p].A_I1 : &proofs[p].A_I2, which_phase == 1 ? &proofs[p].A_O1 : &proofs[p].A_O2,
                            which_phase == 1 ? &proofs[p].S1 : &proofs[p].S2};
      for (int k = 0; k < 3; k++) memcpy(dst[k]->xy.data(), &o[(p * 3 + k) * 64], 64);
    }
  };
  commit_phase(0, n1, i_b1, o_b1, s_b1, 1);
  lap("prove: phase-1 commit");
  parallel_for(nb, [&](size_t p) {
    cs[p]->tr.append_point("A_I1", proofs[p].A_I1);
    cs[p]->tr.append_point("A_O1", proofs[p].A_O1);
    cs[p]->tr.append_point("S1", proofs[p].S1);
    cs[p]->create_randomized_constraints();                                              // :501
  });
  lap("prove: randomized constraints");
  const size_t n = cs[0]->a_L.size(), n2 = n - n1, padded_n = next_pow2(n);
  for (auto *c : cs) if (c->a_L.size() != n) throw std::invalid_argument("prove_batch: circuits differ after randomization");
  if (bp_gens.gens_capacity < padded_n) throw R1CSException(R1CSError::InvalidGeneratorsLength);   // :511-513
  if (n2 > 0) commit_phase(n1, n, i_b2, o_b2, s_b2, 2);                                    // else identity, :566-576
  lap("prove: phase-2 commit");
  std::vector<Scalar> y(nb), z(nb);
  parallel_for(nb, [&](size_t p) {
    cs[p]->tr.append_point("A_I2", proofs[p].A_I2);
    cs[p]->tr.append_point("A_O2", proofs[p].A_O2);
    cs[p]->tr.append_point("S2", proofs[p].S2);
    y[p] = cs[p]->tr.challenge_scalar("y");                                               // :584-585
    z[p] = cs[p]->tr.challenge_scalar("z");
  });
  // device: flattened constraints, l/r coefficient vectors, t_1..t_6 -- :587-619 (y^-1, :593, on the device too).  One
  // circuit for all provers: the constraint rows must coincide (their running hashes; BPH_CHECK_ROWS=1: row by row).
  const ParametricCircuit *param = cs[0]->param;
  for (size_t p = 1; p < nb; p++) if (cs[p]->param != param) throw std::invalid_argument("prove_batch: provers of different circuits");
  std::vector<uint8_t> chi_bytes;
  if (param) {
    // the provers were bound to a ParametricCircuit: its device circuit serves all of them; what must agree is the shape the
    // gadgets produced (multipliers per phase, commitments, rows, challenges drawn) and the circuit's
    const size_t nchi = param->challenge_labels().size();
    for (size_t p = 0; p < nb; p++)
      if (n != param->n() || n1 != param->n1() || m != param->m() || cs[p]->chi_drawn.size() != nchi || cs[p]->rows_dropped != param->num_constraints())
        throw std::invalid_argument("prove_batch: the gadgets built on a prover do not match its ParametricCircuit");
    chi_bytes.resize(nb * nchi * 32);
    for (size_t p = 0; p < nb; p++) for (size_t j = 0; j < nchi; j++) cs[p]->chi_drawn[j].to_bytes_le(&chi_bytes[(p * nchi + j) * 32]);
  } else {
    for (size_t p = 1; p < nb; p++)
      if (cs[p]->rows_hash[0] != cs[0]->rows_hash[0] || cs[p]->rows_hash[1] != cs[0]->rows_hash[1] || cs[p]->rows_nnz != cs[0]->rows_nnz ||
          cs[p]->constraints.size() != cs[0]->constraints.size())
        throw std::invalid_argument("prove_batch: constraint rows differ between provers");
    if (getenv("BPH_CHECK_ROWS"))
      parallel_for(nb, [&](size_t p) {
        if (p && !cs[p]->same_rows(*cs[0])) throw std::invalid_argument("prove_batch: constraint rows differ between provers");
      });
  }
  lap("prove: transcript y z, rows");
  bpgpu_circuit *circ = param ? param->device_circuit() : circuit_cache().acquire(*cs[0], n, m);
  struct CircuitGuard { bpgpu_circuit *c; ~CircuitGuard() { if (c) circuit_cache().release(c); } } circuit_guard{param ? nullptr : circ};
  lap("prove: circuit");
  if (!ps) {   // a circuit without multipliers in either phase cannot happen after the phase-1 call; kept for clarity
    throw std::logic_error("prove_batch: no prover session");
  }
  std::vector<uint8_t> tco(nb * 6 * 32), wVb(nb * m * 32 + 1);
  {
    auto by = pack_scalars(y), bz = pack_scalars(z);
    if (param) d.check(bpgpu_r1cs_prover_session_polys_param(d.ctx(), ps, circ, by.data(), bz.data(), chi_bytes.data(), tco.data(), wVb.data()),
                       "bpgpu_r1cs_prover_session_polys_param");
    else d.check(bpgpu_r1cs_prover_session_polys(d.ctx(), ps, circ, by.data(), bz.data(), tco.data(), wVb.data()),
                 "bpgpu_r1cs_prover_session_polys");
  }
  lap("prove: prover_polys");
  auto t = unpack_scalars(tco.data(), nb * 6);   // per prover: t1 t2 t3 t4 t5 t6
  auto wV = unpack_scalars(wVb.data(), nb * m);
  std::vector<Scalar> tb(nb * 6);                                                         // :621-625 (tb2 filled below)
  {   // T_1, T_3, T_4, T_5, T_6 = commit(t_i, tb_i): 5 nb two-term MSMs over (B, B_blinding) -- :627-631
    std::vector<Scalar> vec(nb * 5 * 2);
    const int idx[5] = {0, 2, 3, 4, 5};
    parallel_for(nb, [&](size_t p) {
      for (int k = 0; k < 5; k++) {
        tb[p * 6 + idx[k]] = rngs[p]->scalar();
        vec[(p * 5 + k) * 2] = t[p * 6 + idx[k]];
        vec[(p * 5 + k) * 2 + 1] = tb[p * 6 + idx[k]];
      }
    }, 32);
    auto bytes = pack_scalars(vec);
    std::vector<uint8_t> o(nb * 5 * 64);
    d.check(bpgpu_msm_gens(d.ctx(), gens, nb * 5, 0, bytes.data(), o.data()), "bpgpu_msm_gens");
    for (size_t p = 0; p < nb; p++) {
      StarkPoint *dst[5] = {&proofs[p].T_1, &proofs[p].T_3, &proofs[p].T_4, &proofs[p].T_5, &proofs[p].T_6};
      for (int k = 0; k < 5; k++) memcpy(dst[k]->xy.data(), &o[(p * 5 + k) * 64], 64);
    }
  }
  std::vector<Scalar> u(nb), x(nb), w(nb);
  parallel_for(nb, [&](size_t p) {
    Transcript &tr = cs[p]->tr;
    tr.append_point("T_1", proofs[p].T_1);
    tr.append_point("T_3", proofs[p].T_3);
    tr.append_point("T_4", proofs[p].T_4);
    tr.append_point("T_5", proofs[p].T_5);
    tr.append_point("T_6", proofs[p].T_6);
    u[p] = tr.challenge_scalar("u");                                                      // :639-640
    x[p] = tr.challenge_scalar("x");
    Scalar tb2;
    if (m >= 4096) {                                                                      // :644-648 (32 768 terms for the 2^14-shuffle)
      std::vector<Scalar> part(64);
      parallel_for(64, [&](size_t c) {
        Scalar acc;
        for (size_t i = m * c / 64; i < m * (c + 1) / 64; i++) acc += wV[p * m + i] * cs[p]->v_blinding[i];
        part[c] = acc;
      });
      for (auto &x : part) tb2 += x;
    } else {
      for (size_t i = 0; i < m; i++) tb2 += wV[p * m + i] * cs[p]->v_blinding[i];
    }
    tb[p * 6 + 1] = tb2;
    auto poly6 = [&](const Scalar *c6) {                                                  // util.rs:192-194
      return x[p] * (c6[0] + x[p] * (c6[1] + x[p] * (c6[2] + x[p] * (c6[3] + x[p] * (c6[4] + x[p] * c6[5])))));
    };
    proofs[p].t_x = poly6(&t[p * 6]);                                                     // :659-660
    proofs[p].t_x_blinding = poly6(&tb[p * 6]);
    Scalar i_b = i_b1[p] + u[p] * i_b2[p], o_b = o_b1[p] + u[p] * o_b2[p], s_b = s_b1[p] + u[p] * s_b2[p];   // :674-676
    proofs[p].e_blinding = x[p] * (i_b + x[p] * (o_b + x[p] * s_b));                      // :678
    tr.append_scalar("t_x", proofs[p].t_x);
    tr.append_scalar("t_x_blinding", proofs[p].t_x_blinding);
    tr.append_scalar("e_blinding", proofs[p].e_blinding);
    w[p] = tr.challenge_scalar("w");                                                      // :686
  }, 8);
  lap("prove: T commits, x, blindings");
  bpgpu_ipp *ipp = nullptr;
  for (size_t p = 0; p < nb; p++) cs[p]->tr.innerproduct_domain_sep(padded_n);              // inner_product_proof.rs:72
  if (!getenv("BPH_IPP_FOLD_GENERATORS")) {
    // l(x), r(x), the G/H factors (:661-672, 689-697) and the IPP operands never leave the device; Q_p = w_p * B
    // (:687) and G, H = bp_gens are resident: the session runs over the generator tables
    auto bx = pack_scalars(x), bu = pack_scalars(u), bw = pack_scalars(w);
    int rc = bpgpu_r1cs_prover_ipp_begin(d.ctx(), ps, gens, padded_n, n1, bx.data(), bu.data(), nullptr /* the session's y^-1 */, bw.data(), &ipp);
    d.check(rc, "bpgpu_r1cs_prover_ipp_begin");
  } else {   // the reference's literal schedule (operands through the host, generators folded every round), for A/B runs
    std::vector<uint8_t> lv(nb * padded_n * 32), rv(nb * padded_n * 32);
    {
      auto bx = pack_scalars(x);
      int rc = bpgpu_r1cs_prover_eval(d.ctx(), ps, padded_n, bx.data(), lv.data(), rv.data());   // :661-672
      d.check(rc, "bpgpu_r1cs_prover_eval");
    }
    std::vector<Scalar> Gf(nb * padded_n), Hf(nb * padded_n);
    parallel_for(nb, [&](size_t p) {
      auto exp_y_inv = util::exp_iter(y[p].inverse(), padded_n);
      for (size_t i = 0; i < padded_n; i++) {
        Gf[p * padded_n + i] = i < n1 ? Scalar::one() : u[p];
        Hf[p * padded_n + i] = exp_y_inv[i] * Gf[p * padded_n + i];
      }
    });
    auto pgf = pack_scalars(Gf), phf = pack_scalars(Hf);
    std::vector<uint8_t> Qb(nb * 64);
    std::vector<Scalar> vec(nb * 2);
    for (size_t p = 0; p < nb; p++) vec[2 * p] = w[p];
    auto bytes = pack_scalars(vec);
    d.check(bpgpu_msm_gens(d.ctx(), gens, nb, 0, bytes.data(), Qb.data()), "bpgpu_msm_gens");
    auto pG = pack_points(bp_gens.share(0).G(padded_n)), pH = pack_points(bp_gens.share(0).H(padded_n));
    d.check(bpgpu_ipp_begin(d.ctx(), nb, padded_n, Qb.data(), pgf.data(), phf.data(), pG.data(), pH.data(), 1, lv.data(), rv.data(), &ipp),
            "bpgpu_ipp_begin");
  }
  lap("prove: ipp_begin");
  try {
    if (!getenv("BPH_HOST_IPP_TRANSCRIPT") && !group) {
      // the k rounds back to back on the device, hash chain included (bpgpu_ipp_run_fs); the host transcripts are
      // advanced to the same state afterwards
      size_t k = 0;
      for (size_t t = padded_n; t > 1; t >>= 1) k++;
      std::vector<uint8_t> st_in(nb * 32), st_out(nb * 32), L(nb * k * 64 + 1), R(nb * k * 64 + 1), a(nb * 32), b(nb * 32);
      for (size_t p = 0; p < nb; p++) memcpy(&st_in[32 * p], cs[p]->tr.state(), 32);
      d.check(bpgpu_ipp_run_fs(d.ctx(), ipp, st_in.data(), L.data(), R.data(), a.data(), b.data(), st_out.data()), "bpgpu_ipp_run_fs");
      for (size_t p = 0; p < nb; p++) {
        cs[p]->tr.set_state(&st_out[32 * p]);
        for (size_t r = 0; r < k; r++) {
          StarkPoint Lp, Rp;
          memcpy(Lp.xy.data(), &L[(p * k + r) * 64], 64);
          memcpy(Rp.xy.data(), &R[(p * k + r) * 64], 64);
          proofs[p].ipp_proof.L_vec.push_back(Lp);
          proofs[p].ipp_proof.R_vec.push_back(Rp);
        }
        proofs[p].ipp_proof.a = Scalar::from_bytes_le(&a[32 * p]);
        proofs[p].ipp_proof.b = Scalar::from_bytes_le(&b[32 * p]);
      }
    } else {
    std::vector<uint8_t> L(nb * 64), R(nb * 64), ub(nb * 32), uib(nb * 32);
    while (bpgpu_ipp_len(ipp) > 1) {
      d.check(bpgpu_ipp_round(d.ctx(), ipp, L.data(), R.data()), "bpgpu_ipp_round");
      if (group) {       // this rank's partial L, R -> the sums over the ranks (nb == 1)
        uint8_t lr[128];
        memcpy(lr, L.data(), 64); memcpy(lr + 64, R.data(), 64);
        combine_partials(d, *group, lr, 2);
        memcpy(L.data(), lr, 64); memcpy(R.data(), lr + 64, 64);
      }
      parallel_for(nb, [&](size_t p) {
        StarkPoint Lp, Rp;
        memcpy(Lp.xy.data(), &L[64 * p], 64);
        memcpy(Rp.xy.data(), &R[64 * p], 64);
        proofs[p].ipp_proof.L_vec.push_back(Lp);
        proofs[p].ipp_proof.R_vec.push_back(Rp);
        cs[p]->tr.append_point("L", Lp);                                                  // :119-123 / :177-181
        cs[p]->tr.append_point("R", Rp);
        cs[p]->tr.challenge_scalar("u").to_bytes_le(&ub[32 * p]);
      }, 64);
      uib = ub;
      d.check(bpgpu_batch_inverse(d.ctx(), uib.data(), nb), "bpgpu_batch_inverse");
      d.check(bpgpu_ipp_fold(d.ctx(), ipp, ub.data(), uib.data()), "bpgpu_ipp_fold");
    }
    std::vector<uint8_t> a(nb * 32), b(nb * 32);
    d.check(bpgpu_ipp_finish(d.ctx(), ipp, a.data(), b.data()), "bpgpu_ipp_finish");
    for (size_t p = 0; p < nb; p++) {
      proofs[p].ipp_proof.a = Scalar::from_bytes_le(&a[32 * p]);
      proofs[p].ipp_proof.b = Scalar::from_bytes_le(&b[32 * p]);
    }
    }
  } catch (...) {
    bpgpu_ipp_destroy(d.ctx(), ipp);
    throw;
  }
  bpgpu_ipp_destroy(d.ctx(), ipp);
  lap("prove: ipp rounds");
  return proofs;
}

// ---- Verifier --------------------------------------------------------------------------------------
Verifier::Verifier(const PedersenGens &pc, Transcript &t) : c_(new CsCore(false, pc, t, this)) {}
Verifier::~Verifier() {}
Transcript &Verifier::transcript() { return c_->tr; }
size_t Verifier::num_constraints() const { return c_->constraints.size(); }
size_t Verifier::num_multipliers() const { return c_->num_vars; }
std::array<Variable, 3> Verifier::multiply(LinearCombination l, LinearCombination r) { return c_->multiply(std::move(l), std::move(r)); }
Variable Verifier::allocate(const Scalar *a) { return c_->allocate(a); }
std::array<Variable, 3> Verifier::allocate_multiplier(const std::pair<Scalar, Scalar> *) { return c_->new_multiplier(Scalar(), Scalar()); }
Variable Verifier::commit_public(const Scalar &v) { return commit(c_->pc_gens.commit(v, Scalar::one())); }   // verifier.rs:153-160
void Verifier::constrain(LinearCombination lc) { c_->push_row(std::move(lc)); }
Scalar Verifier::eval(const LinearCombination &) const { return Scalar::zero(); }
void Verifier::specify_randomized_constraints(Callback cb) { c_->deferred.push_back(std::move(cb)); }
Scalar Verifier::challenge_scalar(const std::string &label) {
  if (c_->probe_labels) {           // a ParametricCircuit probe: substituted value, label recorded
    c_->probe_labels->push_back(label);
    const size_t i = c_->probe_next++;
    return c_->probe_chi && i < c_->probe_chi->size() ? (*c_->probe_chi)[i] : Scalar::zero();
  }
  return c_->tr.challenge_scalar(label);
}
StarkPoint Verifier::last_mega_check() const { return c_->mega; }
Variable Verifier::commit(const StarkPoint &V) {
  size_t i = c_->V.size();
  c_->V.push_back(V);
  c_->tr.append_point("V", V);
  return Variable{Variable::Committed, i};
}

void Verifier::circuit_csr(std::vector<uint32_t> &row_ptr, std::vector<uint32_t> &kind, std::vector<uint32_t> &idx,
                           std::vector<uint8_t> &coeff) const {
  c_->csr(row_ptr, kind, idx, coeff);
}

Verifier::BatchInputs Verifier::transcript_replay(const R1CSProof &proof, const BulletproofGens &bp_gens) {
  return replay(proof, bp_gens, nullptr, nullptr);
}
Verifier::BatchInputs Verifier::transcript_replay(const R1CSProof &proof, const BulletproofGens &bp_gens, const ParametricCircuit &circuit,
                                                  std::vector<uint8_t> &gadget_challenges) {
  return replay(proof, bp_gens, &circuit, &gadget_challenges);
}
Verifier::BatchInputs Verifier::replay(const R1CSProof &proof, const BulletproofGens &bp_gens, const ParametricCircuit *pc,
                                       std::vector<uint8_t> *chi) {
  CsCore &c = *c_;
  Transcript &tr = c.tr;
  BatchInputs in;
  if (pc && (c.V.size() != pc->m() || c.num_vars != 0 || c.constraints.size() != 0 || !c.deferred.empty()))
    throw std::invalid_argument("verify with a ParametricCircuit: make the circuit's commitments, and only them (the gadgets live in the circuit)");
  try {
    tr.append_u64("m", c.V.size());                                                     // verifier.rs:398
    in.n1 = pc ? pc->n1() : c.num_vars;
    tr.validate_and_append_point("A_I1", proof.A_I1);                                   // :401-406
    tr.validate_and_append_point("A_O1", proof.A_O1);
    tr.validate_and_append_point("S1", proof.S1);
    if (pc) {                                                                           // :366-385 with the gadget closures' only transcript
      tr.r1cs_2phase_domain_sep();                                                      //  effect: their challenge_scalar calls, in order
      chi->resize(32 * pc->challenge_labels().size());
      for (size_t j = 0; j < pc->challenge_labels().size(); j++) tr.challenge_scalar(pc->challenge_labels()[j]).to_bytes_le(chi->data() + 32 * j);
    } else {
      c.create_randomized_constraints();                                                // :409
    }
    in.n = pc ? pc->n() : c.num_vars; in.padded_n = next_pow2(in.n); in.m = c.V.size();
    if (bp_gens.gens_capacity < in.padded_n) throw R1CSException(R1CSError::InvalidGeneratorsLength);   // :421-423
    tr.append_point("A_I2", proof.A_I2);                                                // :428-430
    tr.append_point("A_O2", proof.A_O2);
    tr.append_point("S2", proof.S2);
    Scalar y = tr.challenge_scalar("y"), z = tr.challenge_scalar("z");
    tr.validate_and_append_point("T_1", proof.T_1);                                     // :435-444
    tr.validate_and_append_point("T_3", proof.T_3);
    tr.validate_and_append_point("T_4", proof.T_4);
    tr.validate_and_append_point("T_5", proof.T_5);
    tr.validate_and_append_point("T_6", proof.T_6);
    Scalar u = tr.challenge_scalar("u"), x = tr.challenge_scalar("x");
    tr.append_scalar("t_x", proof.t_x);                                                 // :449-453
    tr.append_scalar("t_x_blinding", proof.t_x_blinding);
    tr.append_scalar("e_blinding", proof.e_blinding);
    Scalar w = tr.challenge_scalar("w");
    // transcript half of verification_scalars (inner_product_proof.rs:259-278)
    size_t k = proof.ipp_proof.L_vec.size();
    if (k >= 32 || in.padded_n != ((size_t)1 << k) || proof.ipp_proof.R_vec.size() != k) throw ProofException(ProofError::VerificationError);
    in.k = k;
    tr.innerproduct_domain_sep(in.padded_n);
    std::vector<Scalar> ch{y, z, u, x, w, Scalar()};
    for (size_t i = 0; i < k; i++) {
      tr.validate_and_append_point("L", proof.ipp_proof.L_vec[i]);
      tr.validate_and_append_point("R", proof.ipp_proof.R_vec[i]);
      ch.push_back(tr.challenge_scalar("u"));
    }
    ch[5] = tr.challenge_scalar("r");                                                   // :506
    // the points in bpgpu_r1cs_verify_batch's order, written once (a StarkPoint IS its 64 boundary bytes: the 32 768 commitments
    // of a 2^14-shuffle are one 2 MB copy, not two vectors of points and a packing pass)
    static_assert(sizeof(StarkPoint) == 64, "StarkPoint must be its 64 boundary bytes");
    in.points.resize((11 + c.V.size() + 2 * k) * 64);
    uint8_t *dst = in.points.data();
    auto put = [&](const StarkPoint *q, size_t count) { if (count) memcpy(dst, q, 64 * count); dst += 64 * count; };
    for (auto *q : {&proof.A_I1, &proof.A_O1, &proof.S1, &proof.A_I2, &proof.A_O2, &proof.S2}) put(q, 1);
    put(c.V.data(), c.V.size());
    for (auto *q : {&proof.T_1, &proof.T_3, &proof.T_4, &proof.T_5, &proof.T_6}) put(q, 1);
    put(proof.ipp_proof.L_vec.data(), k);
    put(proof.ipp_proof.R_vec.data(), k);
    in.scalars = pack_scalars({proof.t_x, proof.t_x_blinding, proof.e_blinding, proof.ipp_proof.a, proof.ipp_proof.b});
    in.challenges = pack_scalars(ch);
  } catch (const ProofException &) {
    throw R1CSException(R1CSError::VerificationError);                                  // From<ProofError>, errors.rs:179-189
  }
  return in;
}

void Verifier::verify(const R1CSProof &proof, const BulletproofGens &bp_gens, RankGroup &group, Device *device) {
  if (group.size() <= 1) { verify(proof, bp_gens); return; }
  CsCore &c = *c_;
  BatchInputs in = transcript_replay(proof, bp_gens);          // (every rank replays the transcript: sequential hashing, no shares)
  Device &d = same_gpu(device);
  bpgpu_gens *gens = bp_gens.device_tables(c.pc_gens);
  bpgpu_circuit *circ = c.upload_circuit(in.n, in.m);
  uint8_t part[64];
  int rc = bpgpu_r1cs_verify_shard(d.ctx(), gens, circ, in.n1, in.k, in.points.data(), in.scalars.data(), in.challenges.data(), nullptr,
                                   group.rank(), group.size(), part);
  bpgpu_circuit_destroy(d.ctx(), circ);
  // E_GENS / E_LEN depend on the shapes only, which every rank holds alike: all ranks throw here together.  A malformed point or
  // scalar is seen by the rank whose share holds it: that rank's partial is the poison encoding, and the sum fails on ALL ranks.
  if (rc == BPGPU_E_GENS) throw R1CSException(R1CSError::InvalidGeneratorsLength);
  d.check(rc, "bpgpu_r1cs_verify_shard");
  const int rc2 = combine_partials_rc(d, group, part, 1);
  if (rc2 == BPGPU_E_ARG) throw R1CSException(R1CSError::FormatError);
  d.check(rc2, "bpgpu_points_sum");
  memcpy(c.mega.xy.data(), part, 64);
  if (!c.mega.is_identity()) throw R1CSException(R1CSError::VerificationError);      // :549-551
}
void Verifier::verify(const R1CSProof &proof, const BulletproofGens &bp_gens) {
  CsCore &c = *c_;
  Lap lap;
  BatchInputs in = transcript_replay(proof, bp_gens);
  lap("verify: transcript replay");
  // device: flatten, inversions, scalar assembly, mega_check MSM, identity test -- :457-553
  Device &d = Device::default_device();
  bpgpu_gens *gens = bp_gens.device_tables(c.pc_gens);
  bpgpu_circuit *circ = c.upload_circuit(in.n, in.m);
  lap("verify: upload_circuit");
  int32_t ok = 0;
  int rc = bpgpu_r1cs_verify_batch(d.ctx(), gens, circ, 1, in.n1, in.k, in.points.data(), in.scalars.data(),
                                   in.challenges.data(), &ok, c.mega.xy.data(), nullptr);
  lap("verify: bpgpu_r1cs_verify_batch");
  bpgpu_circuit_destroy(d.ctx(), circ);
  if (rc == BPGPU_E_GENS) throw R1CSException(R1CSError::InvalidGeneratorsLength);
  if (rc == BPGPU_E_ARG) throw R1CSException(R1CSError::FormatError);
  d.check(rc, "bpgpu_r1cs_verify_batch");
  if (!ok) throw R1CSException(R1CSError::VerificationError);                           // :549-551
}

// ---- ParametricCircuit --------------------------------------------------------------------------------------------------
namespace {
struct ProbeRows { size_t n1 = 0, n = 0; std::vector<std::vector<std::pair<Variable, Scalar>>> rows; std::vector<std::string> labels; };
}
ParametricCircuit::ParametricCircuit(size_t m, const std::function<void(Verifier &, const std::vector<Variable> &)> &gadgets) : m_(m) {
  auto run = [&](const std::vector<Scalar> &chi, ProbeRows &out) {
    Transcript tr("parametric circuit probe");
    Verifier v(PedersenGens(), tr);
    std::vector<Variable> vars;
    const StarkPoint dummy = StarkPoint::generator();
    for (size_t i = 0; i < m; i++) vars.push_back(v.commit(dummy));
    gadgets(v, vars);
    CsCore &c = *v.c_;
    out.n1 = c.num_vars;
    c.probe_chi = &chi; c.probe_labels = &out.labels; c.probe_next = 0;
    c.create_randomized_constraints();
    c.probe_chi = nullptr; c.probe_labels = nullptr;
    out.n = c.num_vars;
    out.rows.resize(c.constraints.size());
    for (size_t r = 0; r < c.constraints.size(); r++)
      for (auto &kv : c.constraints[r].terms)
        if (kv.first.kind != Variable::Zero) out.rows[r].push_back(kv);
  };
  ProbeRows r0;
  run({}, r0);
  const size_t nchi = r0.labels.size();
  if (nchi == 0) throw std::invalid_argument("ParametricCircuit: the gadgets draw no challenge (use an ordinary circuit)");
  if (nchi > 8) throw std::invalid_argument("ParametricCircuit: more than 8 gadget challenges");
  n1_ = r0.n1; n_ = r0.n; q_ = r0.rows.size(); labels_ = r0.labels;
  // c_j = rows(e_j) - rows(0), term by term (a term may be absent from one of the two: coefficient 0)
  typedef std::map<Variable, Scalar> Row;
  auto as_map = [](const std::vector<std::pair<Variable, Scalar>> &v) { Row mrow; for (auto &kv : v) mrow[kv.first] = kv.second; return mrow; };
  std::vector<Row> c0(q_);
  for (size_t r = 0; r < q_; r++) c0[r] = as_map(r0.rows[r]);
  std::vector<std::vector<Row>> cj(nchi, std::vector<Row>(q_));
  for (size_t j = 0; j < nchi; j++) {
    std::vector<Scalar> e(nchi);
    e[j] = Scalar::one();
    ProbeRows rj;
    run(e, rj);
    if (rj.rows.size() != q_ || rj.n != n_ || rj.n1 != n1_ || rj.labels != labels_)
      throw std::invalid_argument("ParametricCircuit: the circuit's shape depends on its challenges");
    for (size_t r = 0; r < q_; r++) {
      Row mj = as_map(rj.rows[r]);
      for (auto &kv : mj) { auto it = c0[r].find(kv.first); const Scalar d = it == c0[r].end() ? kv.second : kv.second - it->second; if (d != Scalar::zero()) cj[j][r][kv.first] = d; }
      for (auto &kv : c0[r]) if (!mj.count(kv.first) && kv.second != Scalar::zero()) cj[j][r][kv.first] = -kv.second;
    }
  }
  {   // the affine model, checked on a random point: every term of every row
    OsRng rng(false);
    std::vector<Scalar> rho(nchi);
    for (auto &x : rho) x = rng.scalar();
    ProbeRows rr;
    run(rho, rr);
    if (rr.rows.size() != q_ || rr.n != n_ || rr.n1 != n1_) throw std::invalid_argument("ParametricCircuit: the circuit's shape depends on its challenges");
    for (size_t r = 0; r < q_; r++) {
      Row want = c0[r];
      for (size_t j = 0; j < nchi; j++) for (auto &kv : cj[j][r]) want[kv.first] = want[kv.first] + rho[j] * kv.second;
      Row got = as_map(rr.rows[r]);
      for (auto &kv : want) { auto it = got.find(kv.first); if ((it == got.end() ? Scalar::zero() : it->second) != kv.second) throw std::invalid_argument("ParametricCircuit: a constraint coefficient is not affine in the gadget challenges"); }
      for (auto &kv : got) if (!want.count(kv.first) && kv.second != Scalar::zero()) throw std::invalid_argument("ParametricCircuit: a constraint coefficient is not affine in the gadget challenges");
    }
  }
  // CSR of bpgpu_circuit_create_param: (1 + nchi) q rows, block 0 = c0, block j = c_j
  std::vector<uint32_t> rp{0}, kind, idx;
  std::vector<uint8_t> coeff;
  auto emit = [&](const Row &row) {
    for (auto &kv : row) {
      if (kv.second == Scalar::zero()) continue;
      kind.push_back((uint32_t)kv.first.kind); idx.push_back((uint32_t)kv.first.index);
      coeff.resize(coeff.size() + 32);
      kv.second.to_bytes_le(coeff.data() + coeff.size() - 32);
    }
    rp.push_back((uint32_t)kind.size());
  };
  for (size_t r = 0; r < q_; r++) emit(c0[r]);
  for (size_t j = 0; j < nchi; j++) for (size_t r = 0; r < q_; r++) emit(cj[j][r]);
  Device &d = Device::default_device();
  if (kind.empty()) { kind.push_back(0); idx.push_back(0); coeff.resize(32); }   // (never dereferenced: row_ptr ends at 0)
  d.check(bpgpu_circuit_create_param(d.ctx(), q_, nchi, rp.data(), kind.data(), idx.data(), coeff.data(), n_, m_, &circ_), "bpgpu_circuit_create_param");
}
ParametricCircuit::~ParametricCircuit() { if (circ_) bpgpu_circuit_destroy(Device::default_device().ctx(), circ_); }

void Verifier::verify(const R1CSProof &proof, const BulletproofGens &bp_gens, const ParametricCircuit &circuit, RankGroup &group, Device *device) {
  if (group.size() <= 1) { verify(proof, bp_gens, circuit); return; }
  CsCore &c = *c_;
  std::vector<uint8_t> chi;
  BatchInputs in = replay(proof, bp_gens, &circuit, &chi);
  Device &d = same_gpu(device);
  bpgpu_gens *gens = bp_gens.device_tables(c.pc_gens);
  uint8_t part[64];
  int rc = bpgpu_r1cs_verify_shard(d.ctx(), gens, circuit.device_circuit(), in.n1, in.k, in.points.data(), in.scalars.data(), in.challenges.data(),
                                   chi.data(), group.rank(), group.size(), part);
  if (rc == BPGPU_E_GENS) throw R1CSException(R1CSError::InvalidGeneratorsLength);
  d.check(rc, "bpgpu_r1cs_verify_shard");
  const int rc2 = combine_partials_rc(d, group, part, 1);      // (a malformed operand: the poison partial fails every rank together)
  if (rc2 == BPGPU_E_ARG) throw R1CSException(R1CSError::FormatError);
  d.check(rc2, "bpgpu_points_sum");
  memcpy(c.mega.xy.data(), part, 64);
  if (!c.mega.is_identity()) throw R1CSException(R1CSError::VerificationError);
}
void Verifier::verify(const R1CSProof &proof, const BulletproofGens &bp_gens, const ParametricCircuit &circuit) {
  CsCore &c = *c_;
  Lap lap;
  std::vector<uint8_t> chi;
  BatchInputs in = replay(proof, bp_gens, &circuit, &chi);
  lap("verify (parametric): transcript replay");
  Device &d = Device::default_device();
  bpgpu_gens *gens = bp_gens.device_tables(c.pc_gens);
  int32_t ok = 0;
  int rc = bpgpu_r1cs_verify_batch_param(d.ctx(), gens, circuit.device_circuit(), 1, in.n1, in.k, in.points.data(), in.scalars.data(),
                                         in.challenges.data(), chi.data(), &ok, c.mega.xy.data(), nullptr);
  lap("verify (parametric): bpgpu_r1cs_verify_batch_param");
  if (rc == BPGPU_E_GENS) throw R1CSException(R1CSError::InvalidGeneratorsLength);
  if (rc == BPGPU_E_ARG) throw R1CSException(R1CSError::FormatError);
  d.check(rc, "bpgpu_r1cs_verify_batch_param");
  if (!ok) throw R1CSException(R1CSError::VerificationError);                           // :549-551
}

}  // namespace r1cs
}  // namespace mpc_bulletproof

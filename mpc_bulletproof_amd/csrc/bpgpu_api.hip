// bpgpu_api.hip -- the C ABI of include/bpgpu.h on top of the HIP kernels.
// No CPU fallback: every entry point needs a live HIP device and fails with BPGPU_E_DEVICE otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/bpgpu.h"
#include "kernels.h"
#include "fe29.cuh"

using namespace bpk;

namespace {

struct Slot { void *p = nullptr; size_t cap = 0; };

}  // namespace

struct bpgpu_ctx {
  int device = 0;
  hipStream_t st = nullptr, st2 = nullptr;
  hipEvent_t ev1 = nullptr, ev2 = nullptr;
  std::mutex mu;
  std::string err;
  int *d_flag = nullptr;          // device int: bad-input flag
  void *sqrt_tab = nullptr;       // F_p square-root tables of the point codec (built on first use)
  struct bpgpu_gens *gen_tab = nullptr;   // 16-bit-window table of the curve generator (bpgpu_generator_mul)
  Slot ws[30];                    // grow-only workspace slots
  // optional per-kernel HIP-event timing (bench.py roofline): kind -> list of (start, stop)
  bool prof = false;
  bool latency_mode = false;      // bpgpu_set_latency_mode
  size_t shard_rank = 0, shard_world = 1;   // bpgpu_set_shard: this context's share of ONE large proof split over the GPUs of a node
  // bpgpu_set_option: launch-route selectors of THIS context (tests walk every route through them; a multi-tenant host gives
  // each tenant its own context).  The BPGPU_* environment variables of the same names only seed the defaults, once, in bpgpu_create.
  int64_t opt[BPGPU_OPT_COUNT] = {};
  std::vector<hipEvent_t> prof_ev[BPGPU_PROF_KINDS];
  std::vector<hipEvent_t> prof_pool;   // recycled events
  std::vector<hipEvent_t> prof_epochs; // reference events handed out by bpgpu_profile_epoch (alive as long as the context)
  bool prof_skip[BPGPU_PROF_KINDS] = {};
  uint32_t prof_mask = 0xffffffffu;    // bpgpu_profile_select
  // Device buffers of the prover / IPP sessions, recycled between sessions: hipFree waits for the WHOLE device to idle, which
  // serialises two contexts that pipeline batches (one host thread each), and a session is a dozen allocations.
  struct PoolBlk { void *p; size_t cap; bool used; };
  std::vector<PoolBlk> pool;
  // bpgpu_r1cs_verify_stream: the ring of lanes (child contexts: one stream + workspaces each) this context spreads the batches
  // of one call over; created on first use, owned by the parent
  std::vector<bpgpu_ctx *> lanes;
  hipEvent_t lane_ev = nullptr;   // fork / join marker of a stream call
  void *pinned = nullptr;         // page-locked staging for the verdicts of a host-memory stream call (grow-only)
  size_t pinned_cap = 0;
  // device-transcript schedule cache (m, k, padded_n) -> steps already resident in ws slot 15
  size_t sched_key[3] = {(size_t)-1, (size_t)-1, (size_t)-1};
  int sched_len = 0;
};
struct ProfScope {   // records start/stop events on `st` around a launch when profiling is on (events come from a per-context pool)
  bpgpu_ctx *c; int kind; hipStream_t st;
  // start and stop marks of a kind alternate (scopes of one kind do not nest).  At most PROF_CAP timed launches per kind are
  // kept between two reads: a long run samples its first launches instead of creating thousands of events inside the timed
  // region (2 048 steps: 1 200 hipEventCreate calls on the sampled context cost the run 5 % of its throughput).
  static constexpr size_t PROF_CAP = 256;
  static void mark(bpgpu_ctx *c, int kind, hipStream_t st) {
    auto &v = c->prof_ev[kind];
    if (!((c->prof_mask >> kind) & 1u)) return;
    if (c->prof_skip[kind]) { c->prof_skip[kind] = false; return; }                       // the stop of a skipped start
    if ((v.size() & 1) == 0 && v.size() >= 2 * PROF_CAP) { c->prof_skip[kind] = true; return; }
    hipEvent_t e = nullptr;
    if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    c->prof_ev[kind].push_back(e);
  }
  ProfScope(bpgpu_ctx *c_, int kind_, hipStream_t st_) : c(c_), kind(kind_), st(st_) { if (c->prof) mark(c, kind, st); }
  ~ProfScope() { if (c->prof) mark(c, kind, st); }
};
// the same with an explicit end: entry points that wait for their results close the span BEFORE the host-side wait (error paths
// close it on the way out)
struct ProfSpan {
  bpgpu_ctx *c; int kind; hipStream_t st; bool open;
  ProfSpan(bpgpu_ctx *c_, int kind_, hipStream_t st_) : c(c_), kind(kind_), st(st_), open(c_->prof) { if (open) ProfScope::mark(c, kind, st); }
  void close() { if (open) { ProfScope::mark(c, kind, st); open = false; } }
  ~ProfSpan() { close(); }
};
static void prof_mark_cb(void *c, int kind, hipStream_t st) { ProfScope::mark((bpgpu_ctx *)c, kind, st); }
struct bpgpu_gens {
  size_t cap = 0;
  int c = 0;
  AffDev *points = nullptr;       // [B, B_blinding, G_0..G_{cap-1}, H_0..H_{cap-1}]
  AffDev *table = nullptr;        // (2 + 2 cap) * W * 2^(c-1)
};
// lock-step InnerProductProof::create state for nb proofs (device resident between rounds)
struct bpgpu_ipp {
  size_t nb = 0, n0 = 0, n = 0;
  bool first = true, shared_gens = false;
  Words8 *a[2] = {nullptr, nullptr}, *b[2] = {nullptr, nullptr};   // ping-pong, nb x n
  AffDev *G[2] = {nullptr, nullptr}, *H[2] = {nullptr, nullptr};   // [0]: input (n0 or nb x n0), [1]/[0] folded
  AffDev *Q = nullptr;
  Words8 *Gf = nullptr, *Hf = nullptr;                             // nb x n0 (first round only)
  Words8 *t1 = nullptr, *t2 = nullptr, *t3 = nullptr, *t4 = nullptr;   // nb x n0/2 temporaries
  Words8 *cLR = nullptr, *uu = nullptr;                            // nb x 2 each (uu: u | u_inv as 2 arrays of nb)
  JacRaw *res = nullptr, *sums = nullptr;                          // nb x 2 x (n0 + 1), nb x 2
  AffDev *mpts = nullptr;                                          // nb x 2 x (n0 + 1): contiguous MSM operands
  Words8 *msc = nullptr;                                           //   (bucket-method rounds)
  Words8 *out_xy = nullptr;                                        // nb x 2 points
  const JacRaw *tail_partials = nullptr;                           // the last round MSM's chunk partials, when the fused round tail sums them
  size_t tail_chunks = 0;
  int cur = 0;                                                     // index of the live a/b/G/H buffers
  bpgpu_gens *own_gens = nullptr;                                  // tables built for this session only (bpgpu_ipp_begin, one proof)
  const bpgpu_gens *gens = nullptr;                                // resident-generator mode: no G/H buffers,
  Words8 *cG = nullptr, *cH = nullptr, *w = nullptr;               //   coefficient vectors nb x n0 and Q = w * B
  size_t slo = 0, shi = (size_t)-1;                                // bpgpu_set_shard at session start: the generators whose terms this
  bool with_q = true;                                              //   rank's L, R carry (the c Q term belongs to rank 0)
};
struct bpgpu_circuit {
  size_t q = 0, n = 0, m = 0, nnz = 0, nchi = 0;   // nchi: gadget challenges the coefficients are affine in (kernels.h CircuitDev)
  uint32_t *col_ptr = nullptr, *row = nullptr;
  Words8 *coeff = nullptr;
};

#define HIPCK(ctx, call)                                                                    \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) {                                                                \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
      return e__ == hipErrorOutOfMemory ? BPGPU_E_OOM : BPGPU_E_DEVICE;                     \
    }                                                                                       \
  } while (0)
#define CK(x)                \
  do {                       \
    int rc__ = (x);          \
    if (rc__ != BPGPU_OK) return rc__; \
  } while (0)

// session buffers (ctx->mu held): best fit among the free blocks of at most twice the size, else a new allocation; released blocks
// stay with the context (at most 48 free ones: beyond that the largest goes back to the device)
static bool pool_alloc(bpgpu_ctx *ctx, void **out, size_t bytes) {
  *out = nullptr;
  bytes = (bytes + 255) & ~(size_t)255;
  if (!bytes) bytes = 256;
  int best = -1;
  for (size_t i = 0; i < ctx->pool.size(); i++) {
    auto &b = ctx->pool[i];
    if (!b.used && b.cap >= bytes && b.cap <= 2 * bytes + 4096 && (best < 0 || b.cap < ctx->pool[best].cap)) best = (int)i;
  }
  if (best >= 0) { ctx->pool[best].used = true; *out = ctx->pool[best].p; return true; }
  void *p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {   // give the cached blocks back and try once more
    for (auto &b : ctx->pool) if (!b.used && b.p) { (void)hipFree(b.p); b.p = nullptr; }
    std::vector<bpgpu_ctx::PoolBlk> keep;
    for (auto &b : ctx->pool) if (b.p) keep.push_back(b);
    ctx->pool.swap(keep);
    if (hipMalloc(&p, bytes) != hipSuccess) return false;
  }
  try { ctx->pool.push_back({p, bytes, true}); } catch (const std::bad_alloc &) { (void)hipFree(p); return false; }
  *out = p;
  return true;
}
static void pool_release(bpgpu_ctx *ctx, void *p) {
  if (!p) return;
  size_t nfree = 0, largest = (size_t)-1;
  for (size_t i = 0; i < ctx->pool.size(); i++) {
    auto &b = ctx->pool[i];
    if (b.p == p) b.used = false;
    if (!b.used) { nfree++; if (largest == (size_t)-1 || b.cap > ctx->pool[largest].cap) largest = i; }
  }
  if (nfree > 48 && largest != (size_t)-1) {
    (void)hipFree(ctx->pool[largest].p);
    ctx->pool.erase(ctx->pool.begin() + (long)largest);
  }
}
static int ws_get(bpgpu_ctx *ctx, int slot, size_t bytes, void **out) {
  Slot &s = ctx->ws[slot];
  if (bytes < 256) bytes = 256;
  if (s.cap < bytes) {
    if (s.p) { HIPCK(ctx, hipStreamSynchronize(ctx->st)); HIPCK(ctx, hipStreamSynchronize(ctx->st2)); HIPCK(ctx, hipFree(s.p)); s.p = nullptr; s.cap = 0; }
    size_t want = bytes + bytes / 4;
    HIPCK(ctx, hipMalloc(&s.p, want));
    s.cap = want;
  }
  *out = s.p;
  return BPGPU_OK;
}
// scratch for the per-lane Straus tables; launches on ctx->st are in order, so they share it
static int straus_ws(bpgpu_ctx *ctx, int np, size_t n, void **out) { return ws_get(ctx, 13, straus_scratch_bytes(np, n), out); }
static int flag_reset(bpgpu_ctx *ctx) { HIPCK(ctx, hipMemsetAsync(ctx->d_flag, 0, sizeof(int), ctx->st)); return BPGPU_OK; }
static int flag_read(bpgpu_ctx *ctx, int *v) {
  HIPCK(ctx, hipMemcpyAsync(v, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->st));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
static int h2d(bpgpu_ctx *ctx, void *d, const void *h, size_t n) {
  if (n) HIPCK(ctx, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->st));
  return BPGPU_OK;
}
static int d2h(bpgpu_ctx *ctx, void *h, const void *d, size_t n) {
  if (n) HIPCK(ctx, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, ctx->st));
  return BPGPU_OK;
}
static int launch_ok(bpgpu_ctx *ctx) { HIPCK(ctx, hipGetLastError()); return BPGPU_OK; }
// One launch that brings a batch's three operand arrays from page-locked host memory into the lane's buffers: wide, coalesced reads
// over the bus by 256 waves (no DMA command, no cross-engine dependency), so that the chain's latency-bound kernels -- the table
// lanes, the scalar assembly -- read HBM, not the bus.  Sizes in 16-byte words.
__global__ void __launch_bounds__(256) k_fetch3(const uint4 *s0, uint4 *d0, size_t n0, const uint4 *s1, uint4 *d1, size_t n1,
                                                const uint4 *s2, uint4 *d2, size_t n2) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1 + n2; i += stride) {
    if (i < n0) d0[i] = s0[i];
    else if (i < n0 + n1) d1[i - n0] = s1[i - n0];
    else d2[i - n0 - n1] = s2[i - n0 - n1];
  }
}
// the device-side address of a page-locked (device-mapped) host allocation, or nullptr for pageable / unknown memory
static const void *host_device_alias(const void *h) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, h) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
  return a.devicePointer;
}

// ---- small device helpers that live with the API ------------------------------------------------
__global__ void k_coeff_to_mont(Words8 *io, size_t n, int *bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  for (int j = 0; j < 8; j++) w[j] = io[i].w[j];
  if (!bp::words_lt_mod<bp::FN>(w)) { atomicOr(bad, 1); return; }
  bp::Fn x = bp::canon(bp::to_mont(bp::unpack<bp::FN>(w)));
  bp::pack(w, x);
  for (int j = 0; j < 8; j++) io[i].w[j] = w[j];
}
// the same from ark-ff Montgomery limbs (x 2^256 mod n): one multiplication by 2^266 instead of the host's de-Montgomery
__global__ void k_coeff_ark_to_mont(Words8 *io, size_t n, int *bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  for (int j = 0; j < 8; j++) w[j] = io[i].w[j];
  if (!bp::words_lt_mod<bp::FN>(w)) { atomicOr(bad, 1); return; }
  constexpr int32_t C[bp::NL] = FN_ARK_MONT;
  bp::Fn k;
  for (int j = 0; j < bp::NL; j++) k.v[j] = C[j];
  bp::Fn x = bp::canon(bp::mul(bp::unpack<bp::FN>(w), k));
  bp::pack(w, x);
  for (int j = 0; j < 8; j++) io[i].w[j] = w[j];
}
static int msm_gens_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, size_t nb, size_t n, const uint32_t *dsc, JacRaw *dres,
                        hipStream_t st, int part_slot = 12, int lpm = 0, int kinds = 0) {
  size_t chunks = fixed_msm_chunks(g->c, n, nb, kinds);
  void *dpart = nullptr;
  if (chunks > 1) CK(ws_get(ctx, part_slot, nb * chunks * sizeof(JacRaw), &dpart));
  fixed_msm(st, g->c, g->table, n, g->cap, dsc, (2 + 2 * n) * 8, dres, nb, (JacRaw *)dpart, lpm, kinds);
  return BPGPU_OK;
}

extern "C" {
#pragma GCC visibility push(default)

int bpgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char *bpgpu_strerror(int code) {
  switch (code) {
    case BPGPU_OK: return "ok";
    case BPGPU_E_ARG: return "malformed argument (non-canonical scalar, off-curve point, null pointer)";
    case BPGPU_E_LEN: return "length mismatch";
    case BPGPU_E_DEVICE: return "HIP device error (no CPU fallback exists)";
    case BPGPU_E_OOM: return "out of device memory";
    case BPGPU_E_GENS: return "generator capacity too small";
    default: return "unknown";
  }
}
// The pipelined entry points keep ~20 kernels' worth of independent batches in flight, one stream each; the HIP runtime maps
// streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default: everything beyond serialises).  The runtime reads the variable
// when it initialises, so it is set here, when the library is loaded, unless the host process has chosen a value itself.
__attribute__((constructor)) static void bpgpu_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "24", 0); }

static int ctx_create(int device, bool single_stream, bpgpu_ctx **out);
// the one validator of option values: bpgpu_set_option and the environment seeding of a new context both go through it
static bool opt_valid(int option, int64_t value) {
  if (option <= 0 || option >= BPGPU_OPT_COUNT) return false;
  switch (option) {
    case BPGPU_OPT_MSM_WP_MAX: return value >= 0;                                          // 0 = never
    case BPGPU_OPT_VERIFY_STRAUS_NP: return value >= 1 && value <= 4;
    case BPGPU_OPT_VS_LARGE_MIN: return value >= 1;
    case BPGPU_OPT_TABLE_NP: return value == 0 || value == 1 || value == 2 || value == 4 || value == 8;
    case BPGPU_OPT_IPP_TABLE_MAX_N: return value >= 0;
    case BPGPU_OPT_STREAM_LANES: return value >= 1 && value <= 64;
    case BPGPU_OPT_STREAM_BATCH: return value >= 1 && value <= ((int64_t)1 << 20);
    case BPGPU_OPT_SCREEN_BATCH: return value >= 1 && value <= ((int64_t)1 << 20);
    case BPGPU_OPT_HORNER_FORM: return value >= 0 && value <= 3;
    case BPGPU_OPT_HORNER_ROW_MAX: return value >= 0 && value <= ((int64_t)1 << 20);
    case BPGPU_OPT_PIPPENGER_MIN: return value >= 2 && value <= ((int64_t)1 << 30);
    case BPGPU_OPT_IPP_PIPPENGER_MIN: return value >= 2 && value <= ((int64_t)1 << 30);
    case BPGPU_OPT_FIXED_LPM: return value == 0 || value == 16 || value == 32 || value == 64;
    case BPGPU_OPT_GROUPS_FORM: return value >= 0 && value <= 3;
    case BPGPU_OPT_FIXED_CHUNK_GENS: return value >= -1 && value <= 64;
    default: return value == 0 || value == 1;
  }
}
int bpgpu_create(int device, bpgpu_ctx **out) {
  // BPGPU_SINGLE_STREAM=1: one stream per context (deeply pipelined callers overlap ACROSS contexts and
  // hardware queues are a limited resource: GPU_MAX_HW_QUEUES)
  const bool single = getenv("BPGPU_SINGLE_STREAM") && atoi(getenv("BPGPU_SINGLE_STREAM")) != 0;
  return ctx_create(device, single, out);
}
static int ctx_create(int device, bool single, bpgpu_ctx **out) {
  if (!out) return BPGPU_E_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return BPGPU_E_DEVICE;
  bpgpu_ctx *ctx = new (std::nothrow) bpgpu_ctx();
  if (!ctx) return BPGPU_E_OOM;
  ctx->device = device;
  {
    static const struct { int opt; const char *env; int64_t dflt; } seed[] = {
        {BPGPU_OPT_MSM_WP_MAX, "BPGPU_MSM_WP_MAX", (int64_t)1 << 15}, {BPGPU_OPT_MSM_PIP2_SINGLE, "BPGPU_PIP2_SINGLE", 0},
        {BPGPU_OPT_VERIFY_NO_FUSE, "BPGPU_NO_FUSE", 0},               {BPGPU_OPT_VERIFY_WINDOW_PARALLEL, "BPGPU_WINDOW_PARALLEL", 1},
        {BPGPU_OPT_VERIFY_STRAUS_NP, "BPGPU_STRAUS_NP", 4},           {BPGPU_OPT_IPP_LITERAL, "BPGPU_IPP_LITERAL", 0},
        {BPGPU_OPT_VS_LARGE_MIN, "BPGPU_VS_LARGE_MIN", 4096},         {BPGPU_OPT_TABLE_NP, "BPGPU_TABLE_NP", 0},
        {BPGPU_OPT_IPP_TABLE_MAX_N, "BPGPU_IPP_TABLE_MAX_N", (int64_t)1 << 16},
        {BPGPU_OPT_STREAM_LANES, "BPGPU_STREAM_LANES", 20},           {BPGPU_OPT_STREAM_BATCH, "BPGPU_STREAM_BATCH", 1024},
        {BPGPU_OPT_SCREEN_BATCH, "BPGPU_SCREEN_BATCH", 2560},         {BPGPU_OPT_HORNER_FORM, "BPGPU_HORNER_FORM", 0},
        {BPGPU_OPT_HORNER_ROW_MAX, "BPGPU_HORNER_ROW_MAX", 1536},     {BPGPU_OPT_PIPPENGER_MIN, "BPGPU_PIPPENGER_MIN", 512},
        {BPGPU_OPT_IPP_PIPPENGER_MIN, "BPGPU_IPP_PIPPENGER_MIN", 257}, {BPGPU_OPT_FIXED_LPM, "BPGPU_FIXED_LPM", 0},
        {BPGPU_OPT_GROUPS_FORM, "BPGPU_GROUPS_FORM", 0}, {BPGPU_OPT_FIXED_CHUNK_GENS, "BPGPU_FIXED_CHUNK_GENS", 0}};
    // the environment only SEEDS a new context's options, through the same validation as bpgpu_set_option: a value that the
    // setter would refuse (a batch size of 0, a negative lane count, ...) leaves the default in force
    for (auto &s : seed) {
      ctx->opt[s.opt] = s.dflt;
      const char *e = getenv(s.env);
      if (e && *e) {
        char *end = nullptr;
        const long long v = strtoll(e, &end, 10);
        if (end && *end == 0 && opt_valid(s.opt, (int64_t)v)) ctx->opt[s.opt] = (int64_t)v;
      }
    }
  }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->st, hipStreamNonBlocking) != hipSuccess ||
      (single ? ((ctx->st2 = ctx->st), hipSuccess) : hipStreamCreateWithFlags(&ctx->st2, hipStreamNonBlocking)) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev1, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev2, hipEventDisableTiming) != hipSuccess ||
      hipMalloc((void **)&ctx->d_flag, sizeof(int)) != hipSuccess) {
    delete ctx;
    return BPGPU_E_DEVICE;
  }
  *out = ctx;
  return BPGPU_OK;
}
void bpgpu_destroy(bpgpu_ctx *ctx) {
  if (!ctx) return;
  for (auto *l : ctx->lanes) bpgpu_destroy(l);
  ctx->lanes.clear();
  if (ctx->lane_ev) hipEventDestroy(ctx->lane_ev);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->st);
  hipStreamSynchronize(ctx->st2);
  for (auto &s : ctx->ws) if (s.p) hipFree(s.p);
  for (auto &b : ctx->pool) if (b.p) hipFree(b.p);
  hipFree(ctx->d_flag);
  hipFree(ctx->sqrt_tab);
  if (ctx->gen_tab) { hipFree(ctx->gen_tab->points); hipFree(ctx->gen_tab->table); delete ctx->gen_tab; }
  hipEventDestroy(ctx->ev1);
  hipEventDestroy(ctx->ev2);
  for (auto &v : ctx->prof_ev) for (auto e : v) hipEventDestroy(e);
  for (auto e : ctx->prof_pool) hipEventDestroy(e);
  for (auto e : ctx->prof_epochs) hipEventDestroy(e);
  if (ctx->st2 != ctx->st) hipStreamDestroy(ctx->st2);
  hipStreamDestroy(ctx->st);
  delete ctx;
}
const char *bpgpu_last_error(bpgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }
int bpgpu_sync(bpgpu_ctx *ctx) {
  if (!ctx) return BPGPU_E_ARG;
  HIPCK(ctx, hipStreamSynchronize(ctx->st));      // (a stream call joins its lanes into ctx->st before it returns)
  HIPCK(ctx, hipStreamSynchronize(ctx->st2));
  return BPGPU_OK;
}
void *bpgpu_stream(bpgpu_ctx *ctx) { return ctx ? (void *)ctx->st : nullptr; }
int bpgpu_set_latency_mode(bpgpu_ctx *ctx, int on) {
  if (!ctx) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (on && ctx->st2 == ctx->st) {       // a single-stream context (BPGPU_SINGLE_STREAM): the un-pipelined caller gets its side stream now
    HIPCK(ctx, hipSetDevice(ctx->device));
    hipStream_t s2;
    if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess) ctx->st2 = s2; else (void)hipGetLastError();
  }
  ctx->latency_mode = on != 0;
  return BPGPU_OK;
}
static void shard_bounds(size_t total, size_t rank, size_t world, size_t *lo, size_t *hi) {   // contiguous, sizes differ by at most one
  const size_t base = total / world, rem = total % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}
int bpgpu_set_shard(bpgpu_ctx *ctx, size_t rank, size_t world) {
  if (!ctx || !world || rank >= world) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->shard_rank = rank; ctx->shard_world = world;
  return BPGPU_OK;
}
int bpgpu_set_option(bpgpu_ctx *ctx, int option, int64_t value) {
  if (!ctx || !opt_valid(option, value)) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->opt[option] = value;
  return BPGPU_OK;
}
int bpgpu_get_option(bpgpu_ctx *ctx, int option, int64_t *value) {
  if (!ctx || !value || option <= 0 || option >= BPGPU_OPT_COUNT) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  *value = ctx->opt[option];
  return BPGPU_OK;
}
int bpgpu_profile_enable(bpgpu_ctx *ctx, int on) {
  if (!ctx) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->prof = on != 0;
  return BPGPU_OK;
}
int bpgpu_profile_select(bpgpu_ctx *ctx, uint32_t kind_mask) {
  if (!ctx) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (int k = 0; k < BPGPU_PROF_KINDS; k++)      // a pair may not straddle a change of the mask
    if ((ctx->prof_ev[k].size() & 1) || ctx->prof_skip[k]) return BPGPU_E_ARG;
  ctx->prof_mask = kind_mask;
  return BPGPU_OK;
}
int bpgpu_profile_read(bpgpu_ctx *ctx, double ms_sum[BPGPU_PROF_KINDS], uint64_t launches[BPGPU_PROF_KINDS]) {
  if (!ctx || !ms_sum || !launches) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  HIPCK(ctx, hipStreamSynchronize(ctx->st2));
  for (int k = 0; k < BPGPU_PROF_KINDS; k++) { ms_sum[k] = 0; launches[k] = 0; }
  std::vector<bpgpu_ctx *> all{ctx};
  all.insert(all.end(), ctx->lanes.begin(), ctx->lanes.end());     // the lanes of a stream call report through their parent
  for (bpgpu_ctx *c : all) {
    if (c != ctx) HIPCK(ctx, hipStreamSynchronize(c->st));
    for (int k = 0; k < BPGPU_PROF_KINDS; k++) {
      auto &v = c->prof_ev[k];
      for (size_t i = 0; i + 1 < v.size(); i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, v[i], v[i + 1]) == hipSuccess) { ms_sum[k] += ms; launches[k]++; }
      }
      for (auto e : v) c->prof_pool.push_back(e);
      v.clear();
    }
  }
  return BPGPU_OK;
}
// Timed intervals instead of sums: every (start, stop) pair recorded since the last read, in milliseconds relative to `epoch`
// (bpgpu_profile_epoch of ANY context of this device), so that a caller can take the union over kernels, kinds and contexts --
// the time the GPU was busy -- which sums of overlapping launches cannot give.
void *bpgpu_profile_epoch(bpgpu_ctx *ctx) {
  if (!ctx) return nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  hipEvent_t e = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess || hipEventCreate(&e) != hipSuccess) return nullptr;
  if (hipEventRecord(e, ctx->st) != hipSuccess || hipEventSynchronize(e) != hipSuccess) { hipEventDestroy(e); return nullptr; }
  try { ctx->prof_epochs.push_back(e); } catch (const std::bad_alloc &) { hipEventDestroy(e); return nullptr; }
  return (void *)e;                  // owned by the context: valid as a reference until bpgpu_destroy
}
int bpgpu_profile_intervals(bpgpu_ctx *ctx, void *epoch, size_t cap, int32_t *kind, double *start_ms, double *end_ms, size_t *count) {
  if (!ctx || !epoch || !count || (cap && (!kind || !start_ms || !end_ms))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  HIPCK(ctx, hipStreamSynchronize(ctx->st2));
  size_t n = 0;
  std::vector<bpgpu_ctx *> all{ctx};
  all.insert(all.end(), ctx->lanes.begin(), ctx->lanes.end());
  for (bpgpu_ctx *c : all) {
    if (c != ctx) HIPCK(ctx, hipStreamSynchronize(c->st));
    for (int k = 0; k < BPGPU_PROF_KINDS; k++) {
      auto &v = c->prof_ev[k];
      for (size_t i = 0; i + 1 < v.size(); i += 2) {
        float a = 0, b = 0;
        if (n < cap && hipEventElapsedTime(&a, (hipEvent_t)epoch, v[i]) == hipSuccess &&
            hipEventElapsedTime(&b, (hipEvent_t)epoch, v[i + 1]) == hipSuccess) {
          kind[n] = k; start_ms[n] = a; end_ms[n] = b; n++;
        }
      }
      for (auto e : v) c->prof_pool.push_back(e);
      v.clear();
    }
  }
  *count = n;
  return BPGPU_OK;
}
int bpgpu_input_flag(bpgpu_ctx *ctx, int *bad) {
  if (!ctx || !bad) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipStreamSynchronize(ctx->st2));
  CK(flag_read(ctx, bad));
  return flag_reset(ctx);   // read-and-clear: the verification entry points never reset it themselves
}
int bpgpu_host_alloc(size_t bytes, void **out) {
  if (!out) return BPGPU_E_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return BPGPU_E_DEVICE;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 4, hipHostMallocDefault);
  if (e != hipSuccess) { *out = nullptr; return e == hipErrorOutOfMemory ? BPGPU_E_OOM : BPGPU_E_DEVICE; }
  return BPGPU_OK;
}
void bpgpu_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}
int bpgpu_malloc(bpgpu_ctx *ctx, size_t bytes, void **dptr) {
  if (!ctx || !dptr) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  HIPCK(ctx, hipMalloc(dptr, bytes ? bytes : 4));
  return BPGPU_OK;
}
int bpgpu_free(bpgpu_ctx *ctx, void *dptr) {
  if (!ctx) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  HIPCK(ctx, hipStreamSynchronize(ctx->st2));
  if (dptr) HIPCK(ctx, hipFree(dptr));
  return BPGPU_OK;
}
int bpgpu_upload(bpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  CK(h2d(ctx, dst, src, bytes));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
// asynchronous on the context's stream: the caller keeps `src` / `dst` alive (and page-locked: bpgpu_host_alloc, for the
// copy to overlap anything) until bpgpu_sync
int bpgpu_upload_async(bpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return h2d(ctx, dst, src, bytes);
}
int bpgpu_download_async(bpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return d2h(ctx, dst, src, bytes);
}
int bpgpu_download(bpgpu_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipStreamSynchronize(ctx->st2));
  CK(d2h(ctx, dst, src, bytes));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- scalar field */
int bpgpu_batch_inverse(bpgpu_ctx *ctx, uint8_t *scalars, size_t n) {
  if (!ctx || (n && !scalars)) return BPGPU_E_ARG;
  if (!n) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *d;
  CK(ws_get(ctx, 0, n * 32, &d));
  CK(flag_reset(ctx));
  CK(h2d(ctx, d, scalars, n * 32));
  scalars_check(ctx->st, (Words8 *)d, n, ctx->d_flag);
  batch_inverse(ctx->st, (Words8 *)d, n, ctx->d_flag);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, scalars, d, n * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
int bpgpu_inner_product(bpgpu_ctx *ctx, const uint8_t *a, const uint8_t *b, size_t n, uint8_t out[32]) {
  if (!ctx || !out || (n && (!a || !b))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *da, *db, *dout, *scr;
  CK(ws_get(ctx, 0, n * 32, &da));
  CK(ws_get(ctx, 1, n * 32, &db));
  CK(ws_get(ctx, 2, 32, &dout));
  CK(ws_get(ctx, 3, inner_product_scratch_bytes(n), &scr));
  CK(flag_reset(ctx));
  CK(h2d(ctx, da, a, n * 32));
  CK(h2d(ctx, db, b, n * 32));
  scalars_check(ctx->st, (Words8 *)da, n, ctx->d_flag);
  scalars_check(ctx->st, (Words8 *)db, n, ctx->d_flag);
  inner_product(ctx->st, (Words8 *)da, (Words8 *)db, n, (Words8 *)dout, scr);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, out, dout, 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

// MSMs of up to 2^15 terms as a batch of independent <= 32-point sums through the window-parallel launches of the verifier
// (k_ec.hip: point tables -> 64 window sums per group, lane = window -> Horner on quads) and one final sum per instance: no
// sort, no buckets, and the only long dependency chain is the 252 quad doublings every MSM ends in.  64 windows x 16 points is
// 2.4x the additions of the bucket method at these sizes, but they are a few 10^7 wave-instructions on an otherwise idle
// chip: one call is 0.6-0.7 ms from 2 to 2^14 terms and 0.77 ms at 2^15 (the bucket launches of k_pip.hip: 0.86 ms there, and
// ahead from 2^16 terms on: 0.96 against 1.05 ms, 1.07 against 1.59 ms at 2^17; a Straus lane per term + a sum: 1.05 ms).
// points: ABI bytes (validated in the table launch; *d_flag on a malformed one) or, converted = true, AffDev rows.
// *done: handled here (n <= 2^15, BPGPU_OPT_MSM_WP_MAX overrides, and at most 2^16 groups in all).
static void wp_options(const bpgpu_ctx *ctx, VerifyWp &v) {   // the per-context launch-route options of the window-parallel chain
  v.horner_form = (int)ctx->opt[BPGPU_OPT_HORNER_FORM];
  v.row_max = (size_t)ctx->opt[BPGPU_OPT_HORNER_ROW_MAX];
  v.fixed_lpm = (int)ctx->opt[BPGPU_OPT_FIXED_LPM];
  v.groups_form = (int)ctx->opt[BPGPU_OPT_GROUPS_FORM];
  v.fixed_chunk_gens = (int)ctx->opt[BPGPU_OPT_FIXED_CHUNK_GENS];
}
static int msm_wp_batch(bpgpu_ctx *ctx, size_t nb, size_t n, const void *dsc, const void *points, bool converted, JacRaw *dsum, bool *done,
                        int *bad = nullptr, size_t max_n = 0) {
  const size_t wp_max = max_n ? max_n : (size_t)ctx->opt[BPGPU_OPT_MSM_WP_MAX];
  *done = false;
  if (!nb || !n || n > wp_max) return BPGPU_OK;
  size_t G, per;
  if (n <= 32) { G = n; per = 1; }
  else { G = 16; per = (n + 15) / 16; }
  const size_t ng = nb * per, np = per * G;
  if (ng > ((size_t)1 << 16)) return BPGPU_OK;
  const void *sc = dsc, *pts = points;
  if (np != n) {   // ragged tails: identity points with zero scalars -- 64 zero bytes are the identity in either form
    void *dp, *ds;
    CK(ws_get(ctx, 24, nb * np * 64, &dp));     // (slots of their own: the verifier calls this with its slot-7 / slot-9 buffers as operands)
    CK(ws_get(ctx, 25, nb * np * 32, &ds));
    HIPCK(ctx, hipMemsetAsync(dp, 0, nb * np * 64, ctx->st));
    HIPCK(ctx, hipMemsetAsync(ds, 0, nb * np * 32, ctx->st));
    HIPCK(ctx, hipMemcpy2DAsync(dp, np * 64, points, n * 64, n * 64, nb, hipMemcpyDeviceToDevice, ctx->st));
    HIPCK(ctx, hipMemcpy2DAsync(ds, np * 32, dsc, n * 32, n * 32, nb, hipMemcpyDeviceToDevice, ctx->st));
    pts = dp; sc = ds;
  }
  void *dwp;
  CK(ws_get(ctx, 12, verify_wp_scratch_bytes(ng, G), &dwp));
  VerifyWp v{(const AffDev *)pts, ng, G, dwp, bad ? bad : ctx->d_flag, nullptr, true, converted};
  wp_options(ctx, v);
  if (!verify_wp_layout_fits(v)) { ctx->err = "internal: window-parallel scratch layout exceeds its buffer (msm)"; return BPGPU_E_DEVICE; }
  VerifyDims d{};
  verify_wp_front_launch(ctx->st, v, d, nullptr, nullptr, 0, false);
  verify_wp_windows(ctx->st, v, (const uint32_t *)sc);
  if (per == 1) {
    verify_wp_groups(ctx->st, v);
    verify_wp_back(ctx->st, v, 8, nullptr, 0, 0, nullptr, 0, nullptr);
    HIPCK(ctx, hipMemcpyAsync(dsum, verify_wp_varsum(v), nb * sizeof(JacRaw), hipMemcpyDeviceToDevice, ctx->st));
  } else {
    // the instances' window sums are added up first: ONE pair of Horner stages per MSM (not per instance, with a sum of `per` points behind it)
    void *dwp2;
    CK(ws_get(ctx, 28, verify_wp_scratch_bytes(nb, G), &dwp2));
    VerifyWp v2 = v;
    v2.nb = nb;
    v2.scratch = dwp2;
    if (!verify_wp_layout_fits(v2)) { ctx->err = "internal: window-parallel scratch layout exceeds its buffer (msm, reduced)"; return BPGPU_E_DEVICE; }
    verify_wp_reduce_instances(ctx->st, v, v2, per);
    verify_wp_groups(ctx->st, v2);
    verify_wp_back(ctx->st, v2, 8, nullptr, 0, 0, nullptr, 0, nullptr);
    HIPCK(ctx, hipMemcpyAsync(dsum, verify_wp_varsum(v2), nb * sizeof(JacRaw), hipMemcpyDeviceToDevice, ctx->st));
  }
  *done = true;
  return BPGPU_OK;
}
/* ---------------------------------------------------------------- MSM (general points) */
// device-resident core: dsc / dxy hold the boundary encodings in HBM, dout receives nb x 64 B; asynchronous on ctx->st
static int msm_batch_dev_locked(bpgpu_ctx *ctx, size_t nb, size_t n, const void *dsc, const void *dxy, void *dout) {
  size_t tot = nb * n;
  void *dpts, *dres, *dsum;
  CK(ws_get(ctx, 2, tot * sizeof(AffDev), &dpts));
  CK(ws_get(ctx, 3, tot * sizeof(JacRaw), &dres));
  CK(ws_get(ctx, 4, nb * sizeof(JacRaw), &dsum));
  const size_t pip_min = (size_t)ctx->opt[BPGPU_OPT_PIPPENGER_MIN];
  if (n >= pip_min) {
    // 32-bit bucket ids, sorted entries (term index | sign bit) and offsets: reject what they cannot address
    const size_t cW = 252 / (size_t)pippenger_window(n) + 1, chalf = (size_t)1 << (pippenger_window(n) - 1);
    if (n >= ((size_t)1 << 31) / nb || nb * cW * chalf >= ((size_t)1 << 31) || tot * cW >= ((size_t)1 << 32)) return BPGPU_E_LEN;
  }
  {
    bool done = false;
    scalars_check(ctx->st, (const Words8 *)dsc, tot, ctx->d_flag);
    CK(msm_wp_batch(ctx, nb, n, dsc, dxy, false, (JacRaw *)dsum, &done));
    if (done) {
      jac_to_boundary(ctx->st, (JacRaw *)dsum, (Words8 *)dout, nb);
      return launch_ok(ctx);
    }
  }
  // k_pip2.hip's one-instance pipeline carries the combined batch check; for a lone MSM the window-parallel launches (up to 2^15
  // terms) and k_pip.hip (from 2^16) are both faster now.  BPGPU_OPT_MSM_PIP2_SINGLE routes 2^8..2^16 terms through it (tests).
  const bool pip2_single = ctx->opt[BPGPU_OPT_MSM_PIP2_SINGLE] != 0;
  if (pip2_single && nb == 1 && n >= pip_min && pippenger2_supported(n)) {   // one mid-size instance: seven launches (k_pip2.hip)
    const int c2 = pippenger2_window(n);
    void *dpip;
    CK(ws_get(ctx, 14, pippenger2_scratch_bytes(n, c2), &dpip));
    pippenger2_boundary(ctx->st, (const Words8 *)dxy, (const Words8 *)dsc, n, c2, (Words8 *)dout, (AffDev *)dpts, dpip, ctx->d_flag);
    return launch_ok(ctx);
  }
  scalars_check(ctx->st, (const Words8 *)dsc, tot, ctx->d_flag);
  points_from_boundary(ctx->st, (const Words8 *)dxy, (AffDev *)dpts, tot, ctx->d_flag);
  if (n >= pip_min) {   // bucket method
    int c = pippenger_window(n);
    void *dpip;
    if (nb == 1) {
      CK(ws_get(ctx, 14, pippenger_scratch_bytes(n, c), &dpip));
      pippenger(ctx->st, (AffDev *)dpts, (const uint32_t *)dsc, n, c, (JacRaw *)dsum, dpip);
    } else {
      CK(ws_get(ctx, 14, pippenger_scratch_bytes_batch(nb, n, c), &dpip));
      pippenger_batch(ctx->st, (AffDev *)dpts, (const uint32_t *)dsc, nb, n, c, (JacRaw *)dsum, 1, dpip);
    }
  } else {
    StrausArgs a{};
    a.pts[0] = (AffDev *)dpts; a.pt_stride[0] = 1;
    a.sc[0] = (const uint32_t *)dsc; a.sc_stride[0] = 8;
    void *dstr;
    CK(straus_ws(ctx, 1, tot, &dstr));
    straus(ctx->st, 1, a, (JacRaw *)dres, tot, dstr);
    segmented_sum(ctx->st, (JacRaw *)dres, (JacRaw *)dsum, nb, n);
  }
  jac_to_boundary(ctx->st, (JacRaw *)dsum, (Words8 *)dout, nb);
  return launch_ok(ctx);
}
static int msm_batch_locked(bpgpu_ctx *ctx, size_t nb, size_t n, const uint8_t *scalars, const uint8_t *points,
                            uint8_t *out) {
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t tot = nb * n;
  if (!nb) return BPGPU_OK;
  if (!n) { memset(out, 0, nb * 64); return BPGPU_OK; }
  void *dsc, *dxy, *dout;
  CK(ws_get(ctx, 0, tot * 32, &dsc));
  CK(ws_get(ctx, 1, tot * 64, &dxy));
  CK(ws_get(ctx, 5, nb * 64, &dout));
  CK(flag_reset(ctx));
  CK(h2d(ctx, dsc, scalars, tot * 32));
  CK(h2d(ctx, dxy, points, tot * 64));
  CK(msm_batch_dev_locked(ctx, nb, n, dsc, dxy, dout));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, out, dout, nb * 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
int bpgpu_msm_batch_dev(bpgpu_ctx *ctx, size_t nb, size_t n, const void *scalars_dev, const void *points_dev, void *out_dev) {
  if (!ctx || (nb && !out_dev) || (nb && n && (!scalars_dev || !points_dev))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  if (!nb) return BPGPU_OK;
  if (!n) { HIPCK(ctx, hipMemsetAsync(out_dev, 0, nb * 64, ctx->st)); return BPGPU_OK; }
  CK(flag_reset(ctx));   // the flag reports on the most recent *_dev call
  return msm_batch_dev_locked(ctx, nb, n, scalars_dev, points_dev, out_dev);
}
int bpgpu_msm(bpgpu_ctx *ctx, const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && (!scalars || !points))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return msm_batch_locked(ctx, 1, n, scalars, points, out);
}
int bpgpu_msm_batch(bpgpu_ctx *ctx, size_t nb, size_t n, const uint8_t *scalars, const uint8_t *points,
                    uint8_t *out) {
  if (!ctx || (nb && !out) || (nb && n && (!scalars || !points))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return msm_batch_locked(ctx, nb, n, scalars, points, out);
}

/* nsets MSMs over ONE point vector (the share / MAC / public-modifier MSMs of msm_authenticated_iter) */
/* ---------------------------------------------------------------- arkworks in-memory forms (k_ark.hip) */
// sum_i scalars[i] * pts[i] for validated device operands (plain canonical scalars, Montgomery affine points) -> one JacRaw
static int msm_core_locked(bpgpu_ctx *ctx, size_t n, const uint32_t *dsc, const AffDev *dpts, JacRaw *dsum) {
  const size_t pip_min = (size_t)ctx->opt[BPGPU_OPT_PIPPENGER_MIN];
  bool done = false;
  CK(msm_wp_batch(ctx, 1, n, dsc, dpts, true, dsum, &done));
  if (done) return BPGPU_OK;
  void *dpip;
  const bool pip2_single = ctx->opt[BPGPU_OPT_MSM_PIP2_SINGLE] != 0;
  if (pip2_single && n >= pip_min && pippenger2_supported(n)) {
    const int c2 = pippenger2_window(n);
    CK(ws_get(ctx, 14, pippenger2_scratch_bytes(n, c2), &dpip));
    pippenger2(ctx->st, dpts, dsc, n, c2, dsum, dpip, ctx->d_flag);
  } else if (n >= pip_min) {
    const int c = pippenger_window(n);
    const size_t cW = 252 / (size_t)c + 1;
    if (n >= ((size_t)1 << 31) || cW * ((size_t)1 << (c - 1)) >= ((size_t)1 << 31) || n * cW >= ((size_t)1 << 32)) return BPGPU_E_LEN;
    CK(ws_get(ctx, 14, pippenger_scratch_bytes(n, c), &dpip));
    pippenger(ctx->st, dpts, dsc, n, c, dsum, dpip);
  } else {
    void *dres, *dstr;
    CK(ws_get(ctx, 13, straus_scratch_bytes(1, n), &dstr));
    CK(ws_get(ctx, 12, n * sizeof(JacRaw), &dres));
    StrausArgs sa{};
    sa.pts[0] = dpts; sa.pt_stride[0] = 1;
    sa.sc[0] = dsc; sa.sc_stride[0] = 8;
    straus(ctx->st, 1, sa, (JacRaw *)dres, n, dstr);
    segmented_sum(ctx->st, (JacRaw *)dres, dsum, 1, n);
  }
  return BPGPU_OK;
}
int bpgpu_msm_ark(bpgpu_ctx *ctx, const uint8_t *scalars_mont, const uint8_t *points_jac_mont, size_t n, uint8_t out_jac_mont[96]) {
  if (!ctx || !out_jac_mont || (n && (!scalars_mont || !points_jac_mont))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *dsc, *dpj, *daff, *djac, *dsum, *dout;
  CK(ws_get(ctx, 0, (n ? n : 1) * 32, &dsc));
  CK(ws_get(ctx, 1, (n ? n : 1) * 96, &dpj));
  CK(ws_get(ctx, 2, (n ? n : 1) * sizeof(AffDev), &daff));
  CK(ws_get(ctx, 3, (n ? n : 1) * sizeof(JacRaw), &djac));
  CK(ws_get(ctx, 4, sizeof(JacRaw), &dsum));
  CK(ws_get(ctx, 5, 96, &dout));
  CK(flag_reset(ctx));
  if (n) {
    CK(h2d(ctx, dsc, scalars_mont, n * 32));
    CK(h2d(ctx, dpj, points_jac_mont, n * 96));
    scalars_from_ark(ctx->st, (const Words8 *)dsc, (Words8 *)dsc, n, ctx->d_flag);                 // in place
    points_from_ark(ctx->st, (const Words8 *)dpj, (JacRaw *)djac, n, ctx->d_flag);
    batch_normalize(ctx->st, (const JacRaw *)djac, (AffDev *)daff, n, 8);
    CK(msm_core_locked(ctx, n, (const uint32_t *)dsc, (const AffDev *)daff, (JacRaw *)dsum));
  } else {
    HIPCK(ctx, hipMemsetAsync(dsum, 0, sizeof(JacRaw), ctx->st));                                  // Z = 0: the identity
  }
  points_to_ark(ctx->st, (const JacRaw *)dsum, (Words8 *)dout, 1);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, out_jac_mont, dout, 96));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
static int ark_convert_locked(bpgpu_ctx *ctx, int what, const uint8_t *in, size_t n, uint8_t *out) {
  static const size_t in_sz[4] = {32, 32, 96, 64}, out_sz[4] = {32, 32, 64, 96};
  if (!n) return BPGPU_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *din, *dout, *djac, *daff;
  CK(ws_get(ctx, 0, n * in_sz[what], &din));
  CK(ws_get(ctx, 1, n * out_sz[what], &dout));
  CK(ws_get(ctx, 3, n * sizeof(JacRaw), &djac));
  CK(ws_get(ctx, 2, n * sizeof(AffDev), &daff));
  CK(flag_reset(ctx));
  CK(h2d(ctx, din, in, n * in_sz[what]));
  switch (what) {
    case 0: scalars_from_ark(ctx->st, (const Words8 *)din, (Words8 *)dout, n, ctx->d_flag); break;
    case 1: scalars_to_ark(ctx->st, (const Words8 *)din, (Words8 *)dout, n, ctx->d_flag); break;
    case 2:
      points_from_ark(ctx->st, (const Words8 *)din, (JacRaw *)djac, n, ctx->d_flag);
      jac_to_boundary(ctx->st, (const JacRaw *)djac, (Words8 *)dout, n);
      break;
    default:
      points_from_boundary(ctx->st, (const Words8 *)din, (AffDev *)daff, n, ctx->d_flag);
      aff_to_jacraw(ctx->st, (const AffDev *)daff, (JacRaw *)djac, n);
      points_to_ark(ctx->st, (const JacRaw *)djac, (Words8 *)dout, n);
      break;
  }
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, out, dout, n * out_sz[what]));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
int bpgpu_scalars_from_ark(bpgpu_ctx *ctx, const uint8_t *scalars_mont, size_t n, uint8_t *scalars_le) {
  if (!ctx || (n && (!scalars_mont || !scalars_le))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return ark_convert_locked(ctx, 0, scalars_mont, n, scalars_le);
}
int bpgpu_scalars_to_ark(bpgpu_ctx *ctx, const uint8_t *scalars_le, size_t n, uint8_t *scalars_mont) {
  if (!ctx || (n && (!scalars_mont || !scalars_le))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return ark_convert_locked(ctx, 1, scalars_le, n, scalars_mont);
}
int bpgpu_points_from_ark(bpgpu_ctx *ctx, const uint8_t *points_jac_mont, size_t n, uint8_t *points_xy) {
  if (!ctx || (n && (!points_jac_mont || !points_xy))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return ark_convert_locked(ctx, 2, points_jac_mont, n, points_xy);
}
int bpgpu_points_to_ark(bpgpu_ctx *ctx, const uint8_t *points_xy, size_t n, uint8_t *points_jac_mont) {
  if (!ctx || (n && (!points_jac_mont || !points_xy))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return ark_convert_locked(ctx, 3, points_xy, n, points_jac_mont);
}
int bpgpu_points_sum(bpgpu_ctx *ctx, const uint8_t *points, size_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && !points)) return BPGPU_E_ARG;
  if (!n) { memset(out, 0, 64); return BPGPU_OK; }
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *dxy, *dout;
  CK(ws_get(ctx, 0, n * 64, &dxy));
  CK(ws_get(ctx, 4, 64, &dout));
  CK(flag_reset(ctx));
  CK(h2d(ctx, dxy, points, n * 64));
  points_sum(ctx->st, (const Words8 *)dxy, n, (Words8 *)dout, ctx->d_flag);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, out, dout, 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
int bpgpu_msm_shared(bpgpu_ctx *ctx, size_t nsets, size_t n, const uint8_t *scalars, const uint8_t *points,
                     uint8_t *out) {
  if (!ctx || (nsets && !out) || (nsets && n && (!scalars || !points))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  if (!nsets) return BPGPU_OK;
  if (!n) { memset(out, 0, nsets * 64); return BPGPU_OK; }
  const size_t tot = nsets * n;
  void *dsc, *dxy, *dpts, *dres, *dsum, *dout;
  CK(ws_get(ctx, 0, tot * 32, &dsc));
  CK(ws_get(ctx, 1, n * 64, &dxy));
  CK(ws_get(ctx, 3, tot * sizeof(JacRaw), &dres));
  CK(ws_get(ctx, 4, nsets * sizeof(JacRaw), &dsum));
  CK(ws_get(ctx, 5, nsets * 64, &dout));
  const size_t pip_min = (size_t)ctx->opt[BPGPU_OPT_PIPPENGER_MIN];
  const bool bucket = n >= pip_min;
  CK(ws_get(ctx, 2, (bucket ? tot + n : n) * sizeof(AffDev), &dpts));
  CK(flag_reset(ctx));
  CK(h2d(ctx, dsc, scalars, tot * 32));
  CK(h2d(ctx, dxy, points, n * 64));
  scalars_check(ctx->st, (Words8 *)dsc, tot, ctx->d_flag);
  points_from_boundary(ctx->st, (Words8 *)dxy, (AffDev *)dpts, n, ctx->d_flag);   // validated and converted once
  bool wp_done = false;
  if (!bucket || n <= ((size_t)1 << 15)) {   // window-parallel launches over replicas of the converted points (msm_wp_batch)
    void *drep;
    CK(ws_get(ctx, 8, tot * sizeof(AffDev), &drep));
    gather_points(ctx->st, (AffDev *)dpts, 0, n, nsets, (AffDev *)drep, n);
    CK(msm_wp_batch(ctx, nsets, n, dsc, drep, true, (JacRaw *)dsum, &wp_done));
  }
  if (wp_done) {
  } else if (bucket) {   // one batched bucket-method launch; the instances read replicas of the converted points
    AffDev *rep = (AffDev *)dpts + n;
    gather_points(ctx->st, (AffDev *)dpts, 0, n, nsets, rep, n);
    int c = pippenger_window(n);
    void *dpip;
    CK(ws_get(ctx, 14, pippenger_scratch_bytes_batch(nsets, n, c), &dpip));
    pippenger_batch(ctx->st, rep, (const uint32_t *)dsc, nsets, n, c, (JacRaw *)dsum, 1, dpip);
  } else {        // one Straus lane per (set, term), the sets share the point array
    StrausArgs a{};
    a.pts[0] = (AffDev *)dpts; a.pt_stride[0] = 1; a.pt_outer[0] = 0;
    a.sc[0] = (uint32_t *)dsc; a.sc_stride[0] = 8; a.sc_outer[0] = n * 8;
    a.inner = n;
    void *dstr;
    CK(straus_ws(ctx, 1, tot, &dstr));
    straus(ctx->st, 1, a, (JacRaw *)dres, tot, dstr);
    segmented_sum(ctx->st, (JacRaw *)dres, (JacRaw *)dsum, nsets, n);
  }
  jac_to_boundary(ctx->st, (JacRaw *)dsum, (Words8 *)dout, nsets);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, out, dout, nsets * 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- point wire codec */
int bpgpu_points_decompress(bpgpu_ctx *ctx, const uint8_t *compressed, size_t n, uint8_t *xy, int32_t *ok) {
  if (!ctx || (n && (!compressed || !xy || !ok))) return BPGPU_E_ARG;
  if (!n) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  if (!ctx->sqrt_tab) {
    void *t = nullptr;
    if (hipMalloc(&t, sqrt_table_bytes()) != hipSuccess) return BPGPU_E_OOM;
    sqrt_tables_build(ctx->st, t);
    ctx->sqrt_tab = t;
  }
  void *din, *dxy, *dok;
  CK(ws_get(ctx, 0, n * 32, &din));
  CK(ws_get(ctx, 1, n * 64, &dxy));
  CK(ws_get(ctx, 5, n * 4, &dok));
  CK(h2d(ctx, din, compressed, n * 32));
  points_decompress(ctx->st, (const Words8 *)din, (Words8 *)dxy, (int32_t *)dok, n, ctx->sqrt_tab);
  CK(launch_ok(ctx));
  CK(d2h(ctx, xy, dxy, n * 64));
  CK(d2h(ctx, ok, dok, n * 4));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
int bpgpu_points_compress(bpgpu_ctx *ctx, const uint8_t *xy, size_t n, uint8_t *compressed) {
  if (!ctx || (n && (!xy || !compressed))) return BPGPU_E_ARG;
  if (!n) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *dxy, *dpts, *dout;
  CK(ws_get(ctx, 1, n * 64, &dxy));
  CK(ws_get(ctx, 2, n * sizeof(AffDev), &dpts));
  CK(ws_get(ctx, 0, n * 32, &dout));
  CK(flag_reset(ctx));
  CK(h2d(ctx, dxy, xy, n * 64));
  points_from_boundary(ctx->st, (const Words8 *)dxy, (AffDev *)dpts, n, ctx->d_flag);   // canonical + on-curve checks
  points_compress(ctx->st, (const Words8 *)dxy, (Words8 *)dout, n);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, compressed, dout, n * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- resident generators */
int bpgpu_gens_create(bpgpu_ctx *ctx, const uint8_t *G, const uint8_t *H, size_t cap, const uint8_t B[64],
                      const uint8_t Bb[64], int c, bpgpu_gens **out) try {
  if (!ctx || !out || !B || !Bb || (cap && (!G || !H))) return BPGPU_E_ARG;
  if (!(c == 4 || c == 8 || c == 10 || c == 12 || c == 14 || c == 16 || c == 20)) return BPGPU_E_ARG;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t ng = 2 + 2 * cap;
  std::vector<uint8_t> host(ng * 64);   // (before `g`: a bad_alloc here must not leak it)
  bpgpu_gens *g = new (std::nothrow) bpgpu_gens();
  if (!g) return BPGPU_E_OOM;
  g->cap = cap;
  g->c = c;
  memcpy(&host[0], B, 64);
  memcpy(&host[64], Bb, 64);
  if (cap) { memcpy(&host[128], G, cap * 64); memcpy(&host[128 + cap * 64], H, cap * 64); }
  size_t entries = fixed_table_entries(c, ng);
  size_t W = 252 / c + 1;
  void *dxy = nullptr, *scratch = nullptr;
  auto fail = [&](int rc) {
    if (dxy) hipFree(dxy);
    if (scratch) hipFree(scratch);
    if (g->points) hipFree(g->points);
    if (g->table) hipFree(g->table);
    delete g;
    return rc;
  };
#define GCK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e__); return fail(e__ == hipErrorOutOfMemory ? BPGPU_E_OOM : BPGPU_E_DEVICE); } } while (0)
  GCK(hipMalloc(&dxy, ng * 64));
  GCK(hipMalloc((void **)&g->points, ng * sizeof(AffDev)));
  GCK(hipMalloc((void **)&g->table, entries * sizeof(AffDev)));
  // the Jacobian staging of the table is built for groups of generators so that it stays below ~8 GB
  // (a 20-bit-window table of the 64-bit gadget's 130 generators is 57 GB; staged whole it would need 95 GB more)
  const size_t per_gen = W * ((size_t)1 << (c - 1));                 // table rows per generator
  size_t group = ((size_t)8 << 30) / ((per_gen + W) * sizeof(JacRaw));
  if (group < 1) group = 1;
  if (group > ng) group = ng;
  GCK(hipMalloc(&scratch, group * (W + per_gen) * sizeof(JacRaw)));
  GCK(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), ctx->st));
  GCK(hipMemcpyAsync(dxy, host.data(), ng * 64, hipMemcpyHostToDevice, ctx->st));
  points_from_boundary(ctx->st, (Words8 *)dxy, g->points, ng, ctx->d_flag);
  for (size_t g0 = 0; g0 < ng; g0 += group) {
    size_t cnt = ng - g0 < group ? ng - g0 : group;
    fixed_table_build(ctx->st, c, g->points + g0, cnt, g->table + g0 * per_gen, (JacRaw *)scratch);
  }
  GCK(hipGetLastError());
  int bad = 0;
  GCK(hipMemcpyAsync(&bad, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->st));
  GCK(hipStreamSynchronize(ctx->st));
#undef GCK
  hipFree(dxy); dxy = nullptr;
  hipFree(scratch); scratch = nullptr;
  if (bad) return fail(BPGPU_E_ARG);
  *out = g;
  return BPGPU_OK;
} catch (const std::bad_alloc &) {   // host-side staging (std::vector): no exception crosses the C ABI
  return BPGPU_E_OOM;
}
void bpgpu_gens_destroy(bpgpu_ctx *ctx, bpgpu_gens *g) {
  if (!g) return;
  if (ctx) { std::lock_guard<std::mutex> lk(ctx->mu); hipStreamSynchronize(ctx->st); hipStreamSynchronize(ctx->st2); }
  hipFree(g->points);
  hipFree(g->table);
  delete g;
}
size_t bpgpu_gens_capacity(const bpgpu_gens *g) { return g ? g->cap : 0; }

// ark = the scalars are ark-ff Montgomery limbs (x 2^256 mod n, what the reference's Scalar holds in memory): converted in
// place on the device (one multiplication per scalar) instead of one de-Montgomery per scalar on the host
static int msm_gens_impl(bpgpu_ctx *ctx, const bpgpu_gens *g, size_t nb, size_t n, const uint8_t *scalars, uint8_t *out, bool ark) {
  if (!ctx || !g || (nb && (!scalars || !out))) return BPGPU_E_ARG;
  if (n > g->cap) return BPGPU_E_GENS;
  if (!nb) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t per = 2 + 2 * n, tot = nb * per;
  void *dsc, *dres, *dout;
  CK(ws_get(ctx, 0, tot * 32, &dsc));
  CK(ws_get(ctx, 4, nb * sizeof(JacRaw), &dres));
  CK(ws_get(ctx, 5, nb * 64, &dout));
  CK(flag_reset(ctx));
  CK(h2d(ctx, dsc, scalars, tot * 32));
  ProfSpan span(ctx, 18, ctx->st);
  if (ark) scalars_from_ark(ctx->st, (const Words8 *)dsc, (Words8 *)dsc, tot, ctx->d_flag);
  else scalars_check(ctx->st, (Words8 *)dsc, tot, ctx->d_flag);
  CK(msm_gens_dev(ctx, g, nb, n, (uint32_t *)dsc, (JacRaw *)dres, ctx->st));
  jac_to_boundary(ctx->st, (JacRaw *)dres, (Words8 *)dout, nb);
  span.close();
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, out, dout, nb * 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
int bpgpu_msm_gens(bpgpu_ctx *ctx, const bpgpu_gens *g, size_t nb, size_t n, const uint8_t *scalars, uint8_t *out) {
  return msm_gens_impl(ctx, g, nb, n, scalars, out, false);
}
int bpgpu_msm_gens_ark(bpgpu_ctx *ctx, const bpgpu_gens *g, size_t nb, size_t n, const uint8_t *scalars_ark, uint8_t *out) {
  return msm_gens_impl(ctx, g, nb, n, scalars_ark, out, true);
}

/* ---------------------------------------------------------------- IPP */
int bpgpu_fold_witness(bpgpu_ctx *ctx, size_t n, const uint8_t u[32], const uint8_t u_inv[32], const uint8_t *a,
                       const uint8_t *b, const uint8_t *G, const uint8_t *H, uint8_t *a_out, uint8_t *b_out,
                       uint8_t *G_out, uint8_t *H_out) {
  if (!ctx || !u || !u_inv || (n && (!a || !b || !G || !H || !a_out || !b_out || !G_out || !H_out))) return BPGPU_E_ARG;
  if (!n) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  // slot 0: scalars u, u_inv, a[2n], b[2n], a_out[n], b_out[n]
  void *dsc, *dxy, *dpts, *dres, *dout;
  size_t nsc = 2 + 4 * n + 2 * n;
  CK(ws_get(ctx, 0, nsc * 32, &dsc));
  CK(ws_get(ctx, 1, 4 * n * 64, &dxy));
  CK(ws_get(ctx, 2, 4 * n * sizeof(AffDev), &dpts));
  CK(ws_get(ctx, 3, 2 * n * sizeof(JacRaw), &dres));
  CK(ws_get(ctx, 5, 2 * n * 64, &dout));
  Words8 *w = (Words8 *)dsc;
  Words8 *du = w, *dui = w + 1, *da = w + 2, *db = w + 2 + 2 * n, *dao = w + 2 + 4 * n, *dbo = w + 2 + 5 * n;
  CK(flag_reset(ctx));
  CK(h2d(ctx, du, u, 32));
  CK(h2d(ctx, dui, u_inv, 32));
  CK(h2d(ctx, da, a, 2 * n * 32));
  CK(h2d(ctx, db, b, 2 * n * 32));
  CK(h2d(ctx, dxy, G, 2 * n * 64));
  CK(h2d(ctx, (uint8_t *)dxy + 2 * n * 64, H, 2 * n * 64));
  scalars_check(ctx->st, w, 2 + 4 * n, ctx->d_flag);
  points_from_boundary(ctx->st, (Words8 *)dxy, (AffDev *)dpts, 4 * n, ctx->d_flag);
  fold_scalars(ctx->st, n, du, dui, da, db, dao, dbo);
  AffDev *dG = (AffDev *)dpts, *dH = dG + 2 * n;
  StrausArgs sg{};   // G' = u^-1 G_L + u G_R
  sg.pts[0] = dG; sg.pts[1] = dG + n; sg.pt_stride[0] = sg.pt_stride[1] = 1;
  sg.sc[0] = (uint32_t *)dui; sg.sc[1] = (uint32_t *)du; sg.sc_stride[0] = sg.sc_stride[1] = 0;
  void *dstr;
  CK(straus_ws(ctx, 2, n, &dstr));
  straus(ctx->st, 2, sg, (JacRaw *)dres, n, dstr);
  StrausArgs sh{};   // H' = u H_L + u^-1 H_R
  sh.pts[0] = dH; sh.pts[1] = dH + n; sh.pt_stride[0] = sh.pt_stride[1] = 1;
  sh.sc[0] = (uint32_t *)du; sh.sc[1] = (uint32_t *)dui; sh.sc_stride[0] = sh.sc_stride[1] = 0;
  straus(ctx->st, 2, sh, (JacRaw *)dres + n, n, dstr);
  jac_to_boundary(ctx->st, (JacRaw *)dres, (Words8 *)dout, 2 * n);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, a_out, dao, n * 32));
  CK(d2h(ctx, b_out, dbo, n * 32));
  CK(d2h(ctx, G_out, dout, n * 64));
  CK(d2h(ctx, H_out, (uint8_t *)dout + n * 64, n * 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
int bpgpu_verification_scalars(bpgpu_ctx *ctx, const uint8_t *challenges, size_t k, size_t n, uint8_t *u_sq,
                               uint8_t *u_inv_sq, uint8_t *s) {
  if (!ctx || !s || (k && (!challenges || !u_sq || !u_inv_sq))) return BPGPU_E_ARG;
  if (k >= 32 || n != ((size_t)1 << k)) return BPGPU_E_LEN;   // inner_product_proof.rs:259-267
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *d;
  CK(ws_get(ctx, 0, (3 * k + n + 1) * 32, &d));
  Words8 *w = (Words8 *)d;
  CK(flag_reset(ctx));
  CK(h2d(ctx, w, challenges, k * 32));
  scalars_check(ctx->st, w, k, ctx->d_flag);
  verification_scalars(ctx->st, w, k, n, w + k, w + 2 * k, w + 3 * k);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, u_sq, w + k, k * 32));
  CK(d2h(ctx, u_inv_sq, w + 2 * k, k * 32));
  CK(d2h(ctx, s, w + 3 * k, n * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- R1CS */
static int circuit_create_impl(bpgpu_ctx *ctx, size_t q_real, size_t nchi, const uint32_t *row_ptr, const uint32_t *kind,
                               const uint32_t *idx, const uint8_t *coeff, size_t n_mul, size_t m, bpgpu_circuit **out, bool ark = false) try {
  if (!ctx || !out || !row_ptr) return BPGPU_E_ARG;
  *out = nullptr;
  const size_t q = q_real * (1 + nchi);     // CSR rows: block j (rows j q_real ..) holds the chi_j parts of the coefficients
  size_t nnz = row_ptr[q];
  if (nnz && (!kind || !idx || !coeff)) return BPGPU_E_ARG;
  // CSR (row-major, as the reference holds constraints) -> column-major by output variable, on the device (k_scalar.hip
  // circuit_transpose): the host only checks that the row pointers are monotone.  (The transposition used to run here, single-
  // threaded: 2 ms for the 2^14-shuffle's 196 600 terms, paid by every proof of a circuit with randomized constraints.)
  size_t nout = 3 * n_mul + m + 1;
  for (size_t r = 0; r < q; r++) if (row_ptr[r + 1] < row_ptr[r]) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  bpgpu_circuit *c = new (std::nothrow) bpgpu_circuit();
  if (!c) return BPGPU_E_OOM;
  c->q = q_real; c->n = n_mul; c->m = m; c->nnz = nnz; c->nchi = nchi;
  // (plain allocations, not the context's pool: a circuit is shared by contexts and may outlive the one that made it)
  // ONE allocation [coeff | col_ptr | row] (a proof of a circuit with randomized constraints pays for this every time: three
  // hipMalloc + three hipFree were ~1 ms of the 2^14-shuffle's prove and verify)
  auto fail = [&](int rc) { hipFree(c->coeff); delete c; return rc; };
  const size_t b_coeff = (nnz ? nnz : 1) * 32, b_col = ((nout + 1) * 4 + 255) / 256 * 256, b_row = (nnz ? nnz : 1) * 4;
  if (hipMalloc((void **)&c->coeff, b_coeff + b_col + b_row) != hipSuccess) { c->coeff = nullptr; return fail(BPGPU_E_OOM); }
  c->col_ptr = (uint32_t *)((uint8_t *)c->coeff + b_coeff);
  c->row = (uint32_t *)((uint8_t *)c->coeff + b_coeff + b_col);
  void *drp, *dkd, *dix, *dcf, *dfill;
  int rc;
  if ((rc = ws_get(ctx, 0, (q + 1) * 4, &drp)) || (rc = ws_get(ctx, 1, (nnz ? nnz : 1) * 4, &dkd)) || (rc = ws_get(ctx, 2, (nnz ? nnz : 1) * 4, &dix)) ||
      (rc = ws_get(ctx, 3, (nnz ? nnz : 1) * 32, &dcf)) || (rc = ws_get(ctx, 4, nout * 4, &dfill)))
    return fail(rc);
  if (hipMemsetAsync(ctx->d_flag, 0, sizeof(int), ctx->st) != hipSuccess ||
      hipMemcpyAsync(drp, row_ptr, (q + 1) * 4, hipMemcpyHostToDevice, ctx->st) != hipSuccess ||
      (nnz && (hipMemcpyAsync(dkd, kind, nnz * 4, hipMemcpyHostToDevice, ctx->st) != hipSuccess ||
               hipMemcpyAsync(dix, idx, nnz * 4, hipMemcpyHostToDevice, ctx->st) != hipSuccess ||
               hipMemcpyAsync(dcf, coeff, nnz * 32, hipMemcpyHostToDevice, ctx->st) != hipSuccess)))
    return fail(BPGPU_E_DEVICE);
  circuit_transpose(ctx->st, q, (const uint32_t *)drp, (const uint32_t *)dkd, (const uint32_t *)dix, (const Words8 *)dcf, n_mul, m, ark,
                    c->col_ptr, (uint32_t *)dfill, c->row, c->coeff, ctx->d_flag);
  int bad = 0;
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&bad, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->st) != hipSuccess ||
      hipStreamSynchronize(ctx->st) != hipSuccess)
    return fail(BPGPU_E_DEVICE);
  if (bad) return fail(BPGPU_E_ARG);
  *out = c;
  return BPGPU_OK;
} catch (const std::bad_alloc &) {   // host-side staging (std::vector): no exception crosses the C ABI
  return BPGPU_E_OOM;
}
int bpgpu_circuit_create(bpgpu_ctx *ctx, size_t q, const uint32_t *row_ptr, const uint32_t *kind, const uint32_t *idx,
                         const uint8_t *coeff, size_t n_mul, size_t m, bpgpu_circuit **out) {
  return circuit_create_impl(ctx, q, 0, row_ptr, kind, idx, coeff, n_mul, m, out);
}
int bpgpu_circuit_create_ark(bpgpu_ctx *ctx, size_t q, const uint32_t *row_ptr, const uint32_t *kind, const uint32_t *idx,
                             const uint8_t *coeff_ark, size_t n_mul, size_t m, bpgpu_circuit **out) {
  return circuit_create_impl(ctx, q, 0, row_ptr, kind, idx, coeff_ark, n_mul, m, out, true);
}
int bpgpu_circuit_create_param(bpgpu_ctx *ctx, size_t q, size_t nchi, const uint32_t *row_ptr, const uint32_t *kind,
                               const uint32_t *idx, const uint8_t *coeff, size_t n_mul, size_t m, bpgpu_circuit **out) {
  if (nchi > 8) return BPGPU_E_LEN;
  return circuit_create_impl(ctx, q, nchi, row_ptr, kind, idx, coeff, n_mul, m, out);
}
void bpgpu_circuit_destroy(bpgpu_ctx *ctx, bpgpu_circuit *c) {
  if (!c) return;
  if (ctx) { std::lock_guard<std::mutex> lk(ctx->mu); hipStreamSynchronize(ctx->st); hipStreamSynchronize(ctx->st2); }
  hipFree(c->coeff);     // [coeff | col_ptr | row] is one allocation
  delete c;
}
static CircuitDev circuit_dev(const bpgpu_circuit *c) {
  CircuitDev d;
  d.col_ptr = c->col_ptr; d.row = c->row; d.coeff = c->coeff;
  d.q = c->q; d.n = c->n; d.m = c->m; d.nnz = c->nnz;
  d.nchi = c->nchi; d.qz = c->q * (1 + c->nchi);
  return d;
}
int bpgpu_flatten_constraints(bpgpu_ctx *ctx, const bpgpu_circuit *c, size_t nb, const uint8_t *z, uint8_t *wL,
                              uint8_t *wR, uint8_t *wO, uint8_t *wV, uint8_t *wc) {
  if (!ctx || !c || (nb && (!z || (c->n && (!wL || !wR || !wO)) || (c->m && !wV)))) return BPGPU_E_ARG;
  if (c->nchi) return BPGPU_E_ARG;     // parametric circuits are flattened inside the verification entry points (they need chi)
  if (!nb) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t n = c->n, m = c->m;
  void *dz, *dw, *dzp;
  CK(ws_get(ctx, 0, nb * 32, &dz));
  CK(ws_get(ctx, 1, nb * (3 * n + m + 1) * 32, &dw));
  CK(ws_get(ctx, 6, nb * (c->q ? c->q : 1) * 9 * 4, &dzp));
  Words8 *w = (Words8 *)dw;
  Words8 *dL = w, *dR = w + nb * n, *dO = w + 2 * nb * n, *dV = w + 3 * nb * n, *dC = w + 3 * nb * n + nb * m;
  CK(flag_reset(ctx));
  CK(h2d(ctx, dz, z, nb * 32));
  scalars_check(ctx->st, (Words8 *)dz, nb, ctx->d_flag);
  flatten(ctx->st, circuit_dev(c), nb, (Words8 *)dz, 8, dL, dR, dO, dV, dC, (int32_t *)dzp);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, wL, dL, nb * n * 32));
  CK(d2h(ctx, wR, dR, nb * n * 32));
  CK(d2h(ctx, wO, dO, nb * n * 32));
  CK(d2h(ctx, wV, dV, nb * m * 32));
  if (wc) CK(d2h(ctx, wc, dC, nb * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

static int verify_batch_dev_locked(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                   size_t k, const void *points, const void *scalars, const void *challenges,
                                   void *ok, void *mega, void *full_sc, const void *chi = nullptr, const size_t *shard = nullptr) {
  // shard = {rank, world} (nb == 1): this rank's share of ONE proof's mega_check -- the generators and the proof points of its
  // slice; `mega` receives the partial sum (the ranks all-gather and add them: SURVEY 8e.2), `ok` says whether the PARTIAL is the identity
  if (k >= 32) return BPGPU_E_LEN;
  if (shard && (nb != 1 || !shard[1] || shard[0] >= shard[1])) return BPGPU_E_ARG;
  if ((c->nchi != 0) != (chi != nullptr)) return BPGPU_E_ARG;   // a parametric circuit needs its gadget challenges, and only it
  size_t np = (size_t)1 << k, n = c->n, m = c->m;
  if (n > np || n1 > n || (np > 1 && n <= np / 2 && n != 0)) return BPGPU_E_LEN;   // padded_n = next_pow2(n)
  if (np > g->cap) return BPGPU_E_GENS;
  if (!nb) return BPGPU_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t nvar = 11 + m + 2 * k, nfix = 2 + 2 * np;
  void *dpts, *dfix, *dvar, *dzp, *dvres, *dfres;
  CK(ws_get(ctx, 7, nb * nvar * sizeof(AffDev), &dpts));
  CK(ws_get(ctx, 8, nb * nfix * 32, &dfix));
  CK(ws_get(ctx, 9, nb * nvar * 32, &dvar));
  VerifyDims d{nb, n1, n, np, k, m, (const Words8 *)chi, (size_t)ctx->opt[BPGPU_OPT_VS_LARGE_MIN]};
  CK(ws_get(ctx, 6, verify_scalars_scratch_ints(circuit_dev(c), d) * 4, &dzp));
  CK(ws_get(ctx, 10, nb * nvar * sizeof(JacRaw), &dvres));
  // (nb x parts partial sums when the generator half is walked a proof per lane: BPGPU_OPT_FIXED_CHUNK_GENS)
  size_t fparts = 1;
  if (fixed_msm_chunks(g->c, np, nb) == 1 && verify_wp_supported(nb, nvar, g->c, np)) {
    VerifyWp vt{};
    vt.nb = nb;
    vt.latency_mode = ctx->latency_mode;
    wp_options(ctx, vt);
    fparts = verify_wp_fixed_parts(vt, np);
  }
  CK(ws_get(ctx, 11, nb * fparts * sizeof(JacRaw), &dfres));
  void *dstr;
  CK(straus_ws(ctx, 4, nb * nvar, &dstr));
  // per-proof canonicity bits of the scalar assembly (every entry is written by the kernel: no reset)
  void *dbadsc;
  CK(ws_get(ctx, 20, 2 * nb * sizeof(int32_t), &dbadsc));
  int32_t *dbadpt = (int32_t *)dbadsc + nb;   // per-proof malformed-point bits of the Straus / separate-launch paths
  // default: window-parallel variable-base part -- front [tables | inversion pass], scalars, windows, groups,
  // back [Horner | fixed-base MSMs], verdict (k_ec.hip).  Every launch is on ctx->st.
  const bool no_fuse = ctx->opt[BPGPU_OPT_VERIFY_NO_FUSE] != 0;
  const bool no_wp = ctx->opt[BPGPU_OPT_VERIFY_WINDOW_PARALLEL] == 0;
  // The generator half rides in the back launch when it is small (<= 16 384 (generator, window) pairs per proof, one chunk);
  // otherwise -- few proofs of a mid-size circuit, or a table window the fused kernel is not built for -- it is its own
  // chunked launch ahead of the Horner pass.  The window kernel walks a proof's points serially in every window lane: up to
  // 256 proof points (the 2^14-shuffle's 32 809 take the Straus launches below).
  const bool fused_fixed = fixed_msm_chunks(g->c, np, nb) == 1 && verify_wp_supported(nb, nvar, g->c, np);
  if (!no_fuse && !no_wp && nvar && (fused_fixed || nvar <= 256)) {
    void *dwp;
    CK(ws_get(ctx, 12, verify_wp_scratch_bytes(nb, nvar), &dwp));
    VerifyWp v{(const AffDev *)points, nb, nvar, dwp, ctx->d_flag, (const int32_t *)dbadsc, ctx->latency_mode, false, (int)ctx->opt[BPGPU_OPT_TABLE_NP]};
    wp_options(ctx, v);
    if (!verify_wp_layout_fits(v)) { ctx->err = "internal: window-parallel scratch layout exceeds its buffer (verify)"; return BPGPU_E_DEVICE; }
    int32_t *aux = nullptr;
    size_t aux_stride = 0;
    const bool fuse_prep = verify_scalars_aux(circuit_dev(c), d, (int32_t *)dzp, &aux, &aux_stride);
    const bool fast = fuse_prep && verify_scalars_fast_shape(circuit_dev(c), d);   // wave-sized proofs: the serial part of the assembly in the front launch's lanes
    { ProfScope ps(ctx, 8, ctx->st);
      verify_wp_front_launch(ctx->st, v, d, (const Words8 *)challenges, aux, aux_stride, fuse_prep, fast ? (const Words8 *)scalars : nullptr,
                             fast ? (Words8 *)dfix : nullptr, fast ? (Words8 *)dvar : nullptr, fast ? (Words8 *)full_sc : nullptr); }
    { ProfScope ps(ctx, 0, ctx->st);
      verify_scalars(ctx->st, circuit_dev(c), d, (const Words8 *)challenges, (const Words8 *)scalars, (Words8 *)dfix,
                     (Words8 *)dvar, (Words8 *)full_sc, (int32_t *)dzp, ctx->d_flag, (int32_t *)dbadsc, fuse_prep, fast); }
    if (shard) {   // a small proof: the other ranks' terms are simply zeroed
      size_t slo, shi, vlo, vhi;
      shard_bounds(np, shard[0], shard[1], &slo, &shi);
      shard_bounds(nvar, shard[0], shard[1], &vlo, &vhi);
      shard_mask(ctx->st, (Words8 *)dfix, np, slo, shi, shard[0] == 0, (Words8 *)dvar, nvar, vlo, vhi);
    }
    // Latency mode with a second stream: the generator half needs nothing but the scalars, so it runs BESIDE the window sums,
    // the first Horner stage and the Horner pass instead of sharing the back launch with the wave-per-proof Horner rows (which
    // then have the SIMDs to themselves): a lone batch's chain loses the ~0.15 ms the two halves spent taking turns.
    const bool side = fused_fixed && ctx->latency_mode && ctx->st2 != ctx->st && !shard;
    if (side) {
      HIPCK(ctx, hipEventRecord(ctx->ev1, ctx->st));
      HIPCK(ctx, hipStreamWaitEvent(ctx->st2, ctx->ev1, 0));
      { ProfScope ps(ctx, 1, ctx->st2);
        CK(msm_gens_dev(ctx, g, nb, np, (const uint32_t *)dfix, (JacRaw *)dfres, ctx->st2, 23, 64)); }
      HIPCK(ctx, hipEventRecord(ctx->ev2, ctx->st2));
    }
    { ProfScope ps(ctx, 7, ctx->st);
      verify_wp_windows(ctx->st, v, (const uint32_t *)dvar); }
    { ProfScope ps(ctx, 9, ctx->st);
      verify_wp_groups(ctx->st, v); }
    if (!fused_fixed) { ProfScope ps(ctx, 1, ctx->st);
      CK(msm_gens_dev(ctx, g, nb, np, (const uint32_t *)dfix, (JacRaw *)dfres, ctx->st, 23)); }
    { ProfScope ps(ctx, 10, ctx->st);
      verify_wp_back(ctx->st, v, g->c, fused_fixed && !side ? g->table : nullptr, np, g->cap, (const uint32_t *)dfix, (2 + 2 * np) * 8, (JacRaw *)dfres); }
    if (side) HIPCK(ctx, hipStreamWaitEvent(ctx->st, ctx->ev2, 0));
    { ProfScope ps(ctx, 11, ctx->st);
      verify_wp_verdict(ctx->st, v, (const JacRaw *)dfres, (int32_t *)ok, (Words8 *)mega, fused_fixed && !side ? verify_wp_fixed_parts(v, np) : 1); }
    return launch_ok(ctx);
  }
  // scalar assembly, then fixed-base part on st2 while st runs the variable-base part
  // (canonicity of the scalars and challenges is checked inside verify_scalars)
  HIPCK(ctx, hipMemsetAsync(dbadpt, 0, nb * sizeof(int32_t), ctx->st));
  {
    ProfScope ps(ctx, 0, ctx->st);
    verify_scalars(ctx->st, circuit_dev(c), d, (const Words8 *)challenges, (const Words8 *)scalars, (Words8 *)dfix,
                   (Words8 *)dvar, (Words8 *)full_sc, (int32_t *)dzp, ctx->d_flag, (int32_t *)dbadsc);
  }
  size_t vlo = 0, vhi = nvar;
  if (shard) {     // generator half: the other ranks' scalars are zeroed (a zero digit costs the table-lookup lanes nothing but the
    size_t slo, shi;   // walk); proof-point half: only this rank's slice of the points is touched at all (below)
    shard_bounds(np, shard[0], shard[1], &slo, &shi);
    shard_bounds(nvar, shard[0], shard[1], &vlo, &vhi);
    shard_mask(ctx->st, (Words8 *)dfix, np, slo, shi, shard[0] == 0, (Words8 *)dvar, nvar, vlo, vhi);
  }
  // proof points: `vnp` points per lane share one doubling chain (lanes per proof = ceil(nvar / vnp)).  Lanes
  // are ROLE-major (lane = role * nb + proof): the 64 lanes of a wave hold the same proof element of 64 proofs,
  // so the identity points of 1-phase proofs (A_I2, A_O2, S2) are skipped wave-uniformly inside k_straus.
  const int vnp_opt = (int)ctx->opt[BPGPU_OPT_VERIFY_STRAUS_NP];
  const int vnp = vnp_opt < 1 ? 1 : (vnp_opt > 4 ? 4 : vnp_opt);
  const size_t lanes = nvar / vnp, rem = nvar - lanes * vnp;   // `rem` leftover points run one per lane
  const size_t nres = lanes + rem;
  StrausArgs am{}, ar{};
  for (int j = 0; j < vnp; j++) {
    am.pts[j] = (AffDev *)dpts + j * lanes; am.pt_stride[j] = nvar; am.pt_outer[j] = 1;
    am.sc[j] = (uint32_t *)dvar + j * lanes * 8; am.sc_stride[j] = nvar * 8; am.sc_outer[j] = 8;
  }
  am.inner = nb; am.out_outer = 1; am.out_stride = nres;
  ar.pts[0] = (AffDev *)dpts + vnp * lanes; ar.pt_stride[0] = nvar; ar.pt_outer[0] = 1;
  ar.sc[0] = (uint32_t *)dvar + vnp * lanes * 8; ar.sc_stride[0] = nvar * 8; ar.sc_outer[0] = 8;
  ar.inner = nb; ar.out_outer = 1; ar.out_stride = nres;
  bool fused = false;
  if (nb == 1 && !no_wp && !no_fuse && nvar <= ((size_t)1 << 16)) {
    // ONE large proof (the 2^14-shuffle: 32 809 proof points): its variable-base half as 16-point groups through the window-
    // parallel launches (msm_wp_batch) instead of a Straus lane per 4 points, a one-point remainder launch and a 256-deep
    // serial sum (1.9 + 0.9 + 1.15 ms -> 0.8 ms); the generator half runs on the second stream as below.
    HIPCK(ctx, hipEventRecord(ctx->ev1, ctx->st));
    HIPCK(ctx, hipStreamWaitEvent(ctx->st2, ctx->ev1, 0));
    { ProfScope ps(ctx, 1, ctx->st2);
      CK(msm_gens_dev(ctx, g, nb, np, (const uint32_t *)dfix, (JacRaw *)dfres, ctx->st2, 23)); }
    HIPCK(ctx, hipEventRecord(ctx->ev2, ctx->st2));
    bool done = false;
    { ProfScope ps(ctx, 3, ctx->st);
      if (vhi > vlo) CK(msm_wp_batch(ctx, 1, vhi - vlo, (const uint8_t *)dvar + vlo * 32, (const uint8_t *)points + vlo * 64, false, (JacRaw *)dvres, &done,
                                     (int *)dbadpt, (size_t)1 << 16));
      else { HIPCK(ctx, hipMemsetAsync(dvres, 0, sizeof(JacRaw), ctx->st)); done = true; }    // (zero limbs = the identity)
    }
    HIPCK(ctx, hipStreamWaitEvent(ctx->st, ctx->ev2, 0));
    if (done) {
      ProfScope ps(ctx, 4, ctx->st);
      verify_finalize(ctx->st, (JacRaw *)dvres, 1, (JacRaw *)dfres, nb, (int32_t *)ok, (Words8 *)mega, (const int32_t *)dbadsc, dbadpt);
      return launch_ok(ctx);
    }
  }
  if (!no_fuse && lanes && fixed_msm_chunks(g->c, np, nb) == 1) {
    // one launch for both halves of the MSM, reading the proof points straight from the ABI bytes
    StrausArgs af = am;
    for (int j = 0; j < vnp; j++) af.pts[j] = (const AffDev *)points + j * lanes;
    af.from_boundary = 1; af.bad = ctx->d_flag; af.bad_inner = dbadpt;
    af.prio = 0;
    ProfScope ps(ctx, 6, ctx->st);
    fused = verify_msm_fused(ctx->st, vnp, af, (JacRaw *)dvres, nb * lanes, dstr, g->c, g->table, np, g->cap,
                             (const uint32_t *)dfix, (2 + 2 * np) * 8, (JacRaw *)dfres, nb);
    if (fused && rem) {
      StrausArgs bf = ar;
      bf.pts[0] = (const AffDev *)points + vnp * lanes;
      bf.from_boundary = 1; bf.bad = ctx->d_flag; bf.bad_inner = dbadpt;
      void *dstr2;
      CK(ws_get(ctx, 12, straus_scratch_bytes(1, nb * rem), &dstr2));
      straus(ctx->st, 1, bf, (JacRaw *)dvres + lanes, nb * rem, dstr2);
    }
  }
  if (!fused) {
    HIPCK(ctx, hipEventRecord(ctx->ev1, ctx->st));
    HIPCK(ctx, hipStreamWaitEvent(ctx->st2, ctx->ev1, 0));
    {
      ProfScope ps(ctx, 1, ctx->st2);
      CK(msm_gens_dev(ctx, g, nb, np, (const uint32_t *)dfix, (JacRaw *)dfres, ctx->st2));
    }
    HIPCK(ctx, hipEventRecord(ctx->ev2, ctx->st2));
    {
      ProfScope ps(ctx, 2, ctx->st);
      points_from_boundary(ctx->st, (const Words8 *)points, (AffDev *)dpts, nb * nvar, ctx->d_flag, dbadpt, nvar);
    }
    {
      ProfScope ps(ctx, 3, ctx->st);
      if (lanes) straus(ctx->st, vnp, am, (JacRaw *)dvres, nb * lanes, dstr);
      if (rem) straus(ctx->st, 1, ar, (JacRaw *)dvres + lanes, nb * rem, dstr);
    }
    HIPCK(ctx, hipStreamWaitEvent(ctx->st, ctx->ev2, 0));
  }
  {
    ProfScope ps(ctx, 4, ctx->st);
    verify_finalize(ctx->st, (JacRaw *)dvres, nres, (JacRaw *)dfres, nb, (int32_t *)ok, (Words8 *)mega, (const int32_t *)dbadsc, dbadpt);
  }
  return launch_ok(ctx);
}
int bpgpu_r1cs_verify_batch_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                size_t k, const void *points, const void *scalars, const void *challenges,
                                void *ok, void *mega, void *full_sc) {
  if (!ctx || !g || !c || (nb && (!points || !scalars || !challenges || !ok))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_batch_dev_locked(ctx, g, c, nb, n1, k, points, scalars, challenges, ok, mega, full_sc);
}
/* ONE proof's mega_check split over the GPUs of a node by term range (BASELINE configs[3]: 98 347 terms): this
 * rank's partial point.  Every rank runs the (cheap, O(n)) scalar assembly in full and the MSM over its share only. */
int bpgpu_r1cs_verify_shard(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t n1, size_t k, const uint8_t *points,
                            const uint8_t *scalars, const uint8_t *challenges, const uint8_t *gadget_challenges, size_t rank, size_t world,
                            uint8_t partial_xy[64]) {
  if (!ctx || !g || !c || !points || !scalars || !challenges || !partial_xy || !world || rank >= world) return BPGPU_E_ARG;
  if ((c->nchi != 0) != (gadget_challenges != nullptr)) return BPGPU_E_ARG;
  if (k >= 32) return BPGPU_E_LEN;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t m = c->m, nvar = 11 + m + 2 * k;
  void *dP, *dS, *dC, *dok, *dmega, *dchi = nullptr;
  CK(ws_get(ctx, 0, nvar * 64, &dP));
  CK(ws_get(ctx, 1, 5 * 32, &dS));
  CK(ws_get(ctx, 2, (6 + k) * 32, &dC));
  CK(ws_get(ctx, 3, 4, &dok));
  CK(ws_get(ctx, 4, 64, &dmega));
  if (c->nchi) { CK(ws_get(ctx, 21, c->nchi * 32, &dchi)); CK(h2d(ctx, dchi, gadget_challenges, c->nchi * 32)); }
  CK(flag_reset(ctx));
  CK(h2d(ctx, dP, points, nvar * 64));
  CK(h2d(ctx, dS, scalars, 5 * 32));
  CK(h2d(ctx, dC, challenges, (6 + k) * 32));
  if (dchi) scalars_check(ctx->st, (const Words8 *)dchi, c->nchi, ctx->d_flag);
  const size_t shard[2] = {rank, world};
  void *dbits;                       // the proof's malformed-scalar | malformed-point bits (slot 20 of verify_batch_dev_locked, nb = 1)
  CK(ws_get(ctx, 20, 2 * sizeof(int32_t), &dbits));
  HIPCK(ctx, hipMemsetAsync(dbits, 0, 2 * sizeof(int32_t), ctx->st));
  CK(verify_batch_dev_locked(ctx, g, c, 1, n1, k, dP, dS, dC, dok, dmega, nullptr, dchi, shard));
  // A malformed operand (off-curve / non-canonical point, non-canonical scalar or challenge) is seen only by the rank whose share
  // holds it: the context flag, or -- on the large-proof route, which validates this rank's slice of the points into the proof's
  // own bits -- the per-proof bits of workspace slot 20 (scalars | points).  The verdict must be COLLECTIVE: this rank returns
  // BPGPU_OK with the poison encoding (64 bytes 0xFF: not a point), and bpgpu_points_sum over the gathered partials fails with
  // BPGPU_E_ARG on every rank alike.  (An error code on one rank only would leave the others waiting in their all-gather.)
  int bad = 0;
  CK(flag_read(ctx, &bad));
  int32_t bits[2] = {0, 0};
  CK(d2h(ctx, bits, dbits, sizeof bits));
  CK(d2h(ctx, partial_xy, dmega, 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  if (bad || bits[0] || bits[1]) memset(partial_xy, 0xFF, 64);
  return BPGPU_OK;
}
int bpgpu_r1cs_verify_batch(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                            size_t k, const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges,
                            int32_t *ok, uint8_t *mega, uint8_t *msm_scalars) {
  if (!ctx || !g || !c || (nb && (!points || !scalars || !challenges || !ok))) return BPGPU_E_ARG;
  if (k >= 32) return BPGPU_E_LEN;
  if (!nb) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t np = (size_t)1 << k, m = c->m, nvar = 11 + m + 2 * k, nterms = 13 + m + 2 * np + 2 * k;
  void *dP, *dS, *dC, *dok, *dmega, *dfull = nullptr;
  CK(ws_get(ctx, 0, nb * nvar * 64, &dP));
  CK(ws_get(ctx, 1, nb * 5 * 32, &dS));
  CK(ws_get(ctx, 2, nb * (6 + k) * 32, &dC));
  CK(ws_get(ctx, 3, nb * 4, &dok));
  CK(ws_get(ctx, 4, nb * 64, &dmega));
  if (msm_scalars) CK(ws_get(ctx, 5, nb * nterms * 32, &dfull));
  CK(h2d(ctx, dP, points, nb * nvar * 64));
  CK(h2d(ctx, dS, scalars, nb * 5 * 32));
  CK(h2d(ctx, dC, challenges, nb * (6 + k) * 32));
  // a malformed proof (off-curve / non-canonical point, non-canonical scalar or challenge) is rejected on its own:
  // ok[p] = 0, the other verdicts stand (the reference's per-proof FormatError / VerificationError)
  CK(verify_batch_dev_locked(ctx, g, c, nb, n1, k, dP, dS, dC, dok, mega ? dmega : nullptr, dfull));
  CK(d2h(ctx, ok, dok, nb * 4));
  if (mega) CK(d2h(ctx, mega, dmega, nb * 64));
  if (msm_scalars) CK(d2h(ctx, msm_scalars, dfull, nb * nterms * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* A STREAM of verification batches in one call: the proofs are cut into batches of `batch` proofs (default 1024) that take turns
 * on the context's ring of lanes (child contexts: a stream and a set of workspaces each), so that the six-launch chains of
 * ~20 batches overlap on the GPU -- what a caller otherwise builds out of twenty contexts.  The lanes fork from ctx->st (inputs
 * uploaded asynchronously on it are visible) and are joined back into it. */
static int stream_lanes(bpgpu_ctx *ctx, size_t want) {
  if (!ctx->lane_ev) HIPCK(ctx, hipEventCreateWithFlags(&ctx->lane_ev, hipEventDisableTiming));
  while (ctx->lanes.size() < want) {
    bpgpu_ctx *l = nullptr;
    int rc = ctx_create(ctx->device, true, &l);
    if (rc) return rc;
    try { ctx->lanes.push_back(l); } catch (const std::bad_alloc &) { bpgpu_destroy(l); return BPGPU_E_OOM; }
  }
  for (bpgpu_ctx *l : ctx->lanes) {   // the lanes follow their parent's settings
    for (int i = 0; i < BPGPU_OPT_COUNT; i++) l->opt[i] = ctx->opt[i];
    l->latency_mode = false;          // (a stream call is the pipelined case by definition)
    l->prof = ctx->prof; l->prof_mask = ctx->prof_mask;
  }
  return BPGPU_OK;
}
static int verify_stream_locked(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges, uint8_t *ok, int on_host) {
  // on_host: 0 = operands and verdicts resident; 1 = pageable host memory (staged copies on the lane's stream); 2 = device-mapped
  // page-locked host memory (one fetch launch per batch, verdicts written in place)
  if (k >= 32) return BPGPU_E_LEN;
  if (!nb) return BPGPU_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t batch = (size_t)ctx->opt[BPGPU_OPT_STREAM_BATCH], nlanes_opt = (size_t)ctx->opt[BPGPU_OPT_STREAM_LANES];
  const size_t nchunks = (nb + batch - 1) / batch, nl = nchunks < nlanes_opt ? nchunks : nlanes_opt;
  CK(stream_lanes(ctx, nl));
  const size_t m = c->m, nvar = 11 + m + 2 * k, nch = 6 + k;
  HIPCK(ctx, hipEventRecord(ctx->lane_ev, ctx->st));
  for (size_t l = 0; l < nl; l++) HIPCK(ctx, hipStreamWaitEvent(ctx->lanes[l]->st, ctx->lane_ev, 0));
  int rc = BPGPU_OK;
  for (size_t ci = 0; ci < nchunks && rc == BPGPU_OK; ci++) {
    bpgpu_ctx *ln = ctx->lanes[ci % nl];
    const size_t lo = ci * batch, cnt = nb - lo < batch ? nb - lo : batch;
    const uint8_t *P = points + lo * nvar * 64, *S = scalars + lo * 5 * 32, *Cc = challenges + lo * nch * 32;
    uint8_t *O = ok + lo * 4;
    if (on_host == 2) {
      void *dP, *dS, *dC;
      if (!(rc = ws_get(ln, 0, batch * nvar * 64, &dP)) && !(rc = ws_get(ln, 1, batch * 5 * 32, &dS)) && !(rc = ws_get(ln, 2, batch * nch * 32, &dC))) {
        hipLaunchKernelGGL(k_fetch3, dim3(64), dim3(256), 0, ln->st, (const uint4 *)P, (uint4 *)dP, cnt * nvar * 4, (const uint4 *)S, (uint4 *)dS, cnt * 10,
                           (const uint4 *)Cc, (uint4 *)dC, cnt * nch * 2);
        rc = verify_batch_dev_locked(ln, g, c, cnt, n1, k, dP, dS, dC, O, nullptr, nullptr);
      }
    } else if (on_host) {       // operands and verdicts in pageable host memory: staged through the lane's own buffers
      void *dP, *dS, *dC, *dok;
      (void)((rc = ws_get(ln, 0, batch * nvar * 64, &dP)) || (rc = ws_get(ln, 1, batch * 5 * 32, &dS)) ||
             (rc = ws_get(ln, 2, batch * nch * 32, &dC)) || (rc = ws_get(ln, 3, batch * 4, &dok)) ||
             (rc = h2d(ln, dP, P, cnt * nvar * 64)) || (rc = h2d(ln, dS, S, cnt * 5 * 32)) || (rc = h2d(ln, dC, Cc, cnt * nch * 32)) ||
             (rc = verify_batch_dev_locked(ln, g, c, cnt, n1, k, dP, dS, dC, dok, nullptr, nullptr)) ||
             (rc = d2h(ln, O, dok, cnt * 4)));
    } else rc = verify_batch_dev_locked(ln, g, c, cnt, n1, k, P, S, Cc, O, nullptr, nullptr);
    if (rc) ctx->err = ln->err;
  }
  // join (also on the way out of an error: the batches already submitted keep running, and nothing the caller does next on this
  // context may overtake them): ctx->st -- and with it bpgpu_sync / the caller's next call -- waits for every lane
  for (size_t l = 0; l < nl; l++) {
    if (hipEventRecord(ctx->lanes[l]->ev1, ctx->lanes[l]->st) != hipSuccess || hipStreamWaitEvent(ctx->st, ctx->lanes[l]->ev1, 0) != hipSuccess) {
      if (rc == BPGPU_OK) { ctx->err = "bpgpu_r1cs_verify_stream: joining the lanes failed"; rc = BPGPU_E_DEVICE; }
    }
  }
  return rc;
}
int bpgpu_r1cs_verify_stream_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                 const void *points_dev, const void *scalars_dev, const void *challenges_dev, void *ok_dev) {
  if (!ctx || !g || !c || (nb && (!points_dev || !scalars_dev || !challenges_dev || !ok_dev))) return BPGPU_E_ARG;
  if (c->nchi) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_stream_locked(ctx, g, c, nb, n1, k, (const uint8_t *)points_dev, (const uint8_t *)scalars_dev,
                              (const uint8_t *)challenges_dev, (uint8_t *)ok_dev, 0);
}
int bpgpu_r1cs_verify_stream(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                             const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges, int32_t *ok) {
  if (!ctx || !g || !c || (nb && (!points || !scalars || !challenges || !ok))) return BPGPU_E_ARG;
  if (c->nchi) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!nb) return BPGPU_OK;
  // the verdicts come back through page-locked staging: an asynchronous copy into the caller's (pageable) array would make every
  // batch's download a host-side wait for its lane -- and serialise the lanes
  HIPCK(ctx, hipSetDevice(ctx->device));
  if (ctx->pinned_cap < nb * 4) {
    if (ctx->pinned) { HIPCK(ctx, hipStreamSynchronize(ctx->st)); (void)hipHostFree(ctx->pinned); ctx->pinned = nullptr; ctx->pinned_cap = 0; }
    HIPCK(ctx, hipHostMalloc(&ctx->pinned, nb * 4 + nb, hipHostMallocDefault));
    ctx->pinned_cap = nb * 4 + nb;
  }
  // Operands in PAGE-LOCKED host memory (bpgpu_host_alloc, hipHostMalloc, hipHostRegister) are fetched by ONE kernel launch per batch
  // on its lane (k_fetch3: wide coalesced reads over the bus -- 2 080 bytes per proof, 8 GB/s at 4 M proofs/s) and the verdicts are
  // written straight into the page-locked staging.  No copy command is enqueued at all: the per-batch H2D / D2H copies on 20 lanes
  // (four DMA commands and as many cross-engine dependencies per 1024 proofs) held the host-memory stream at 2-3 M/s against 4.1
  // resident; reading the operands in place from the chain's own kernels put bus latency on its latency-bound links (3.4 M/s).
  // Pageable operands take the staged copies on each lane's stream.
  const void *dpts = host_device_alias(points), *dsc = host_device_alias(scalars), *dch = host_device_alias(challenges);
  const void *dok = host_device_alias(ctx->pinned);
  if (dpts && dsc && dch && dok)
    CK(verify_stream_locked(ctx, g, c, nb, n1, k, (const uint8_t *)dpts, (const uint8_t *)dsc, (const uint8_t *)dch, (uint8_t *)dok,
                            (((uintptr_t)dpts | (uintptr_t)dsc | (uintptr_t)dch) & 15) ? 0 : 2));   // (16-byte words; unaligned buffers are read in place)
  else
    CK(verify_stream_locked(ctx, g, c, nb, n1, k, points, scalars, challenges, (uint8_t *)ctx->pinned, 1));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  memcpy(ok, ctx->pinned, nb * 4);
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- Verifier::verify with the transcript on the device */
// the transcript half: schedule (cached per context) + k_verify_transcript.  *chp = the challenges (challenges_out or a workspace),
// *dbad = per-proof "a validated point is the identity" flags, *dchi = the gadget challenges of a parametric circuit (or null)
static int fs_transcript_locked(bpgpu_ctx *ctx, const bpgpu_circuit *c, size_t nb, size_t k, const void *init_states, const void *points,
                                const void *scalars, void *challenges_out, const uint8_t *gadget_label, void *chi_out, Words8 **chp,
                                void **dbad_out, void **dchi_out) try {
  if (k >= 32) return BPGPU_E_LEN;
  if (c->nchi > 1 || (c->nchi == 1 && !gadget_label)) return BPGPU_E_ARG;   // one gadget challenge label per schedule
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t m = c->m, np = (size_t)1 << k, nchi = c->nchi;
  void *dsteps, *dch, *dbad, *dchi = nullptr;
  if (nchi) { if (chi_out) dchi = chi_out; else CK(ws_get(ctx, 21, nb * nchi * 32, &dchi)); }
  CK(ws_get(ctx, 19, transcript_schedule_max(m, k) * sizeof(TrStep) + 64, &dsteps));
  CK(ws_get(ctx, 17, nb * (6 + k) * 32, &dch));
  CK(ws_get(ctx, 18, nb * 4, &dbad));
  if (ctx->sched_key[0] != m || ctx->sched_key[1] != k || ctx->sched_key[2] != np + (nchi << 40)) {
    std::vector<TrStep> steps(transcript_schedule_max(m, k));
    ctx->sched_len = transcript_schedule(steps.data(), m, k, np, nchi);
    HIPCK(ctx, hipMemcpyAsync(dsteps, steps.data(), ctx->sched_len * sizeof(TrStep), hipMemcpyHostToDevice, ctx->st));
    HIPCK(ctx, hipStreamSynchronize(ctx->st));   // `steps` is a local
    ctx->sched_key[0] = m; ctx->sched_key[1] = k; ctx->sched_key[2] = np + (nchi << 40);
  }
  *chp = challenges_out ? (Words8 *)challenges_out : (Words8 *)dch;
  *dbad_out = dbad;
  *dchi_out = dchi;
  ProfScope ps(ctx, 5, ctx->st);
  verify_transcript(ctx->st, nb, m, k, (const TrStep *)dsteps, ctx->sched_len, (const Words8 *)init_states,
                    (const Words8 *)points, (const Words8 *)scalars, *chp, (int32_t *)dbad, gadget_label, (Words8 *)dchi, nchi);
  return BPGPU_OK;
} catch (const std::bad_alloc &) {   // host-side staging (std::vector): no exception crosses the C ABI
  return BPGPU_E_OOM;
}
static int verify_fs_locked(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                            const void *init_states, const void *points, const void *scalars, void *ok, void *mega,
                            void *challenges_out, const uint8_t *gadget_label = nullptr, void *chi_out = nullptr) {
  if (k >= 32) return BPGPU_E_LEN;
  if (!nb) return BPGPU_OK;
  Words8 *chp;
  void *dbad, *dchi;
  CK(fs_transcript_locked(ctx, c, nb, k, init_states, points, scalars, challenges_out, gadget_label, chi_out, &chp, &dbad, &dchi));
  CK(verify_batch_dev_locked(ctx, g, c, nb, n1, k, points, scalars, chp, ok, mega, nullptr, dchi));
  and_not(ctx->st, (int32_t *)ok, (const int32_t *)dbad, nb);
  return launch_ok(ctx);
}
int bpgpu_r1cs_verify_batch_fs_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                   const void *init_states, const void *points, const void *scalars, void *ok, void *mega,
                                   void *challenges_out) {
  if (!ctx || !g || !c || (nb && (!init_states || !points || !scalars || !ok))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_fs_locked(ctx, g, c, nb, n1, k, init_states, points, scalars, ok, mega, challenges_out);
}
int bpgpu_r1cs_verify_batch_fs(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                               const uint8_t *init_states, const uint8_t *points, const uint8_t *scalars, int32_t *ok,
                               uint8_t *mega, uint8_t *challenges_out) {
  if (!ctx || !g || !c || (nb && (!init_states || !points || !scalars || !ok))) return BPGPU_E_ARG;
  if (k >= 32) return BPGPU_E_LEN;
  if (!nb) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t m = c->m, nvar = 11 + m + 2 * k;
  void *dP, *dS, *dI, *dok, *dmega, *dcho;
  CK(ws_get(ctx, 0, nb * nvar * 64, &dP));
  CK(ws_get(ctx, 1, nb * 5 * 32, &dS));
  CK(ws_get(ctx, 2, nb * 32, &dI));
  CK(ws_get(ctx, 3, nb * 4 + nb * 64, &dok));
  dmega = (uint8_t *)dok + ((nb * 4 + 63) / 64) * 64;
  CK(ws_get(ctx, 16, nb * (6 + k) * 32, &dcho));
  CK(h2d(ctx, dP, points, nb * nvar * 64));
  CK(h2d(ctx, dS, scalars, nb * 5 * 32));
  CK(h2d(ctx, dI, init_states, nb * 32));
  CK(verify_fs_locked(ctx, g, c, nb, n1, k, dI, dP, dS, dok, mega ? dmega : nullptr, dcho));
  CK(d2h(ctx, ok, dok, nb * 4));
  if (mega) CK(d2h(ctx, mega, dmega, nb * 64));
  if (challenges_out) CK(d2h(ctx, challenges_out, dcho, nb * (6 + k) * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* two-phase circuits (parametric constraint weights): host challenges + gadget challenges, or the transcript on the device */
int bpgpu_r1cs_verify_batch_param(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                  size_t k, const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges,
                                  const uint8_t *gadget_challenges, int32_t *ok, uint8_t *mega, uint8_t *msm_scalars) {
  if (!ctx || !g || !c || (nb && (!points || !scalars || !challenges || !ok)) || (c && c->nchi && !gadget_challenges)) return BPGPU_E_ARG;
  if (k >= 32) return BPGPU_E_LEN;
  if (!nb) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t np = (size_t)1 << k, m = c->m, nvar = 11 + m + 2 * k, nterms = 13 + m + 2 * np + 2 * k;
  void *dP, *dS, *dC, *dok, *dmega, *dfull = nullptr, *dchi = nullptr;
  CK(ws_get(ctx, 0, nb * nvar * 64, &dP));
  CK(ws_get(ctx, 1, nb * 5 * 32, &dS));
  CK(ws_get(ctx, 2, nb * (6 + k) * 32, &dC));
  CK(ws_get(ctx, 3, nb * 4, &dok));
  CK(ws_get(ctx, 4, nb * 64, &dmega));
  if (msm_scalars) CK(ws_get(ctx, 5, nb * nterms * 32, &dfull));
  if (c->nchi) { CK(ws_get(ctx, 21, nb * c->nchi * 32, &dchi)); CK(h2d(ctx, dchi, gadget_challenges, nb * c->nchi * 32)); }
  CK(h2d(ctx, dP, points, nb * nvar * 64));
  CK(h2d(ctx, dS, scalars, nb * 5 * 32));
  CK(h2d(ctx, dC, challenges, nb * (6 + k) * 32));
  // a non-canonical gadget challenge rejects ITS proof (ok[p] = 0), like any other non-canonical scalar or challenge
  void *dchibad = nullptr;
  if (dchi) {
    CK(ws_get(ctx, 18, nb * 4, &dchibad));
    HIPCK(ctx, hipMemsetAsync(dchibad, 0, nb * 4, ctx->st));
    scalars_check_proof(ctx->st, (const Words8 *)dchi, nb * c->nchi, c->nchi, ctx->d_flag, (int32_t *)dchibad);
  }
  CK(verify_batch_dev_locked(ctx, g, c, nb, n1, k, dP, dS, dC, dok, mega ? dmega : nullptr, dfull, dchi));
  if (dchibad) and_not(ctx->st, (int32_t *)dok, (const int32_t *)dchibad, nb);
  CK(d2h(ctx, ok, dok, nb * 4));
  if (mega) CK(d2h(ctx, mega, dmega, nb * 64));
  if (msm_scalars) CK(d2h(ctx, msm_scalars, dfull, nb * nterms * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
int bpgpu_r1cs_verify_batch_fs2_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                    const void *init_states, const uint8_t gadget_label[32], const void *points, const void *scalars,
                                    void *ok, void *mega, void *challenges_out, void *gadget_challenges_out) {
  if (!ctx || !g || !c || (nb && (!init_states || !points || !scalars || !ok))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_fs_locked(ctx, g, c, nb, n1, k, init_states, points, scalars, ok, mega, challenges_out, gadget_label, gadget_challenges_out);
}
int bpgpu_r1cs_verify_batch_fs2(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                const uint8_t *init_states, const uint8_t gadget_label[32], const uint8_t *points,
                                const uint8_t *scalars, int32_t *ok, uint8_t *mega, uint8_t *challenges_out,
                                uint8_t *gadget_challenges_out) {
  if (!ctx || !g || !c || (nb && (!init_states || !points || !scalars || !ok))) return BPGPU_E_ARG;
  if (k >= 32) return BPGPU_E_LEN;
  if (!nb) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t m = c->m, nvar = 11 + m + 2 * k;
  void *dP, *dS, *dI, *dok, *dmega, *dcho, *dchi;
  CK(ws_get(ctx, 0, nb * nvar * 64, &dP));
  CK(ws_get(ctx, 1, nb * 5 * 32, &dS));
  CK(ws_get(ctx, 2, nb * 32, &dI));
  CK(ws_get(ctx, 3, nb * 4 + nb * 64, &dok));
  dmega = (uint8_t *)dok + ((nb * 4 + 63) / 64) * 64;
  CK(ws_get(ctx, 16, nb * (6 + k) * 32, &dcho));
  CK(ws_get(ctx, 22, nb * (c->nchi ? c->nchi : 1) * 32, &dchi));
  CK(h2d(ctx, dP, points, nb * nvar * 64));
  CK(h2d(ctx, dS, scalars, nb * 5 * 32));
  CK(h2d(ctx, dI, init_states, nb * 32));
  CK(verify_fs_locked(ctx, g, c, nb, n1, k, dI, dP, dS, dok, mega ? dmega : nullptr, dcho, gadget_label, c->nchi ? dchi : nullptr));
  CK(d2h(ctx, ok, dok, nb * 4));
  if (mega) CK(d2h(ctx, mega, dmega, nb * 64));
  if (challenges_out) CK(d2h(ctx, challenges_out, dcho, nb * (6 + k) * 32));
  if (gadget_challenges_out && c->nchi) CK(d2h(ctx, gadget_challenges_out, dchi, nb * c->nchi * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- Verifier::verify from wire-format proofs */
// proof_len -> (two_phase, k): 1 + (11 | 14) * 32 + (2k + 2) * 32; the two families never share a length
static bool wire_dims(size_t proof_len, int *two_phase, size_t *k) {
  if (proof_len < 1 + 13 * 32 || (proof_len - 1) % 32) return false;
  size_t el = (proof_len - 1) / 32;
  if (el >= 13 && (el - 13) % 2 == 0) { *two_phase = 0; *k = (el - 13) / 2; return *k < 32; }
  if (el >= 16 && (el - 16) % 2 == 0) { *two_phase = 1; *k = (el - 16) / 2; return *k < 32; }
  return false;
}
static int verify_wire_locked(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                              size_t proof_len, const void *proofs, const void *commitments, const void *init_states,
                              void *ok) {
  int two_phase = 0;
  size_t k = 0;
  if (!wire_dims(proof_len, &two_phase, &k)) return BPGPU_E_LEN;
  if (!nb) return BPGPU_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  if (!ctx->sqrt_tab) {
    void *t = nullptr;
    if (hipMalloc(&t, sqrt_table_bytes()) != hipSuccess) return BPGPU_E_OOM;
    sqrt_tables_build(ctx->st, t);
    ctx->sqrt_tab = t;
  }
  const size_t m = c->m, nvar = 11 + m + 2 * k;
  void *dcomp, *dxy, *dsc, *dfmt;
  CK(ws_get(ctx, 0, nb * nvar * 32, &dcomp));
  CK(ws_get(ctx, 1, nb * nvar * 64, &dxy));
  CK(ws_get(ctx, 2, nb * 5 * 32, &dsc));
  CK(ws_get(ctx, 15, nb * 4 + nb * nvar * 4, &dfmt));
  int32_t *fmt_ok = (int32_t *)dfmt, *dec_ok = fmt_ok + nb;
  wire_unpack(ctx->st, (const uint8_t *)proofs, proof_len, (const uint8_t *)commitments, nb, m, k, two_phase,
              (Words8 *)dcomp, (Words8 *)dsc, fmt_ok);
  points_decompress(ctx->st, (const Words8 *)dcomp, (Words8 *)dxy, dec_ok, nb * nvar, ctx->sqrt_tab);
  // undecodable points come out as the identity: the transcript / MSM run on them, the verdict is forced to 0 below
  CK(verify_fs_locked(ctx, g, c, nb, n1, k, init_states, dxy, dsc, ok, nullptr, nullptr));
  wire_and_ok(ctx->st, (int32_t *)ok, fmt_ok, dec_ok, nb, nvar);
  return launch_ok(ctx);
}
int bpgpu_r1cs_verify_batch_wire_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                     size_t proof_len, const void *proofs_dev, const void *commitments_dev,
                                     const void *init_states_dev, void *ok_dev) {
  if (!ctx || !g || !c || (nb && (!proofs_dev || !init_states_dev || !ok_dev || (c->m && !commitments_dev)))) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_wire_locked(ctx, g, c, nb, n1, proof_len, proofs_dev, commitments_dev, init_states_dev, ok_dev);
}
int bpgpu_r1cs_verify_batch_wire(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                 size_t proof_len, const uint8_t *proofs, const uint8_t *commitments,
                                 const uint8_t *init_states, int32_t *ok) {
  if (!ctx || !g || !c || (nb && (!proofs || !init_states || !ok || (c->m && !commitments)))) return BPGPU_E_ARG;
  if (!nb) return BPGPU_OK;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t m = c->m;
  void *dP, *dC, *dI, *dok;
  CK(ws_get(ctx, 3, nb * proof_len + 64, &dP));
  CK(ws_get(ctx, 4, nb * (m ? m : 1) * 32, &dC));
  CK(ws_get(ctx, 5, nb * 32, &dI));
  CK(ws_get(ctx, 16, nb * 4, &dok));
  CK(h2d(ctx, dP, proofs, nb * proof_len));
  if (m) CK(h2d(ctx, dC, commitments, nb * m * 32));
  CK(h2d(ctx, dI, init_states, nb * 32));
  CK(verify_wire_locked(ctx, g, c, nb, n1, proof_len, dP, dC, dI, dok));
  CK(d2h(ctx, ok, dok, nb * 4));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- combined batch check
 * sum_p rho_p * mega_check_p as ONE point: the generator terms collapse to a single fixed-base MSM with
 * scalars sum_p rho_p * s_{p,g}; the proof-specific points go through one bucket-method MSM of
 * nb * (11 + m + 2k) terms.  All proofs are valid iff the (all-GPU) sum of the partial points is the
 * identity, up to the 2^-252 soundness of the random weights.  Not a reference API: offered
 * beside the per-proof mode for verifier services; the per-proof mode stays the parity path. */
static int verify_combined_locked(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                  size_t k, const void *points, const void *scalars, const void *challenges,
                                  const void *rho, void *partial_xy) {
  if (k >= 32) return BPGPU_E_LEN;
  size_t np = (size_t)1 << k, n = c->n, m = c->m;
  if (n > np || n1 > n || (np > 1 && n <= np / 2 && n != 0)) return BPGPU_E_LEN;
  if (np > g->cap) return BPGPU_E_GENS;
  if (!nb) return BPGPU_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t nvar = 11 + m + 2 * k, nfix = 2 + 2 * np, tot = nb * nvar;
  void *dpts, *dfix, *dvar, *dzp, *dfsum, *dtwo, *dsum, *dpip;
  int cw = pippenger_window(tot);
  CK(ws_get(ctx, 7, tot * sizeof(AffDev), &dpts));
  CK(ws_get(ctx, 8, nb * nfix * 32, &dfix));
  CK(ws_get(ctx, 9, tot * 32, &dvar));
  if (c->nchi) return BPGPU_E_ARG;
  VerifyDims d{nb, n1, n, np, k, m, nullptr, (size_t)ctx->opt[BPGPU_OPT_VS_LARGE_MIN]};
  CK(ws_get(ctx, 6, verify_scalars_scratch_ints(circuit_dev(c), d) * 4, &dzp));
  CK(ws_get(ctx, 10, nfix * 32, &dfsum));
  CK(ws_get(ctx, 11, 2 * sizeof(JacRaw), &dtwo));
  CK(ws_get(ctx, 15, sizeof(JacRaw), &dsum));
  CK(ws_get(ctx, 14, pippenger_scratch_bytes(tot, cw), &dpip));
  CK(flag_reset(ctx));
  if (verify_combined2_supported(nb, nvar, g->c, np)) {   // eight launches on one stream (k_pip2.hip)
    void *dc2;
    CK(ws_get(ctx, 14, verify_combined2_scratch_bytes(nb, nvar, nfix), &dc2));
    CombinedArgs ca{circuit_dev(c), d, nvar, (const Words8 *)points, (const Words8 *)scalars, (const Words8 *)challenges,
                    (const Words8 *)rho, (Words8 *)dfix, (Words8 *)dvar, (int32_t *)dzp, g->table, g->cap, g->c, dc2, ctx->d_flag,
                    (Words8 *)partial_xy, ctx->prof ? &prof_mark_cb : nullptr, ctx};
    verify_combined2(ctx->st, ca);
    return launch_ok(ctx);
  }
  // (canonicity of the scalars and challenges is checked inside verify_scalars)
  scalars_check(ctx->st, (const Words8 *)rho, nb, ctx->d_flag);
  verify_scalars(ctx->st, circuit_dev(c), d, (const Words8 *)challenges, (const Words8 *)scalars, (Words8 *)dfix,
                 (Words8 *)dvar, nullptr, (int32_t *)dzp, ctx->d_flag);
  // generator part on stream 2: weighted column sums, then one fixed-base MSM
  HIPCK(ctx, hipEventRecord(ctx->ev1, ctx->st));
  HIPCK(ctx, hipStreamWaitEvent(ctx->st2, ctx->ev1, 0));
  sc_weighted_colsum(ctx->st2, nb, nfix, (const Words8 *)dfix, (const Words8 *)rho, (Words8 *)dfsum);
  CK(msm_gens_dev(ctx, g, 1, np, (const uint32_t *)dfsum, (JacRaw *)dtwo, ctx->st2));
  HIPCK(ctx, hipEventRecord(ctx->ev2, ctx->st2));
  // proof-specific points on stream 1: scale by rho, bucket-method MSM
  points_from_boundary(ctx->st, (const Words8 *)points, (AffDev *)dpts, tot, ctx->d_flag);
  sc_scale_rows(ctx->st, nb, nvar, (Words8 *)dvar, (const Words8 *)rho);
  pippenger(ctx->st, (const AffDev *)dpts, (const uint32_t *)dvar, tot, cw, (JacRaw *)dtwo + 1, dpip);
  HIPCK(ctx, hipStreamWaitEvent(ctx->st, ctx->ev2, 0));
  segmented_sum(ctx->st, (const JacRaw *)dtwo, (JacRaw *)dsum, 1, 2);
  jac_to_boundary(ctx->st, (const JacRaw *)dsum, (Words8 *)partial_xy, 1);
  return launch_ok(ctx);
}
int bpgpu_r1cs_verify_combined_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                   size_t k, const void *points, const void *scalars, const void *challenges,
                                   const void *rho, void *partial_xy_dev) {
  if (!ctx || !g || !c || (nb && (!points || !scalars || !challenges || !rho)) || !partial_xy_dev) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_combined_locked(ctx, g, c, nb, n1, k, points, scalars, challenges, rho, partial_xy_dev);
}
int bpgpu_r1cs_verify_combined(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                               size_t k, const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges,
                               const uint8_t *rho, uint8_t partial_xy[64]) {
  if (!ctx || !g || !c || (nb && (!points || !scalars || !challenges || !rho)) || !partial_xy) return BPGPU_E_ARG;
  if (k >= 32) return BPGPU_E_LEN;
  if (!nb) { memset(partial_xy, 0, 64); return BPGPU_OK; }
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  size_t m = c->m, nvar = 11 + m + 2 * k;
  void *dP, *dS, *dC, *dR, *dout;
  CK(ws_get(ctx, 0, nb * nvar * 64, &dP));
  CK(ws_get(ctx, 1, nb * 5 * 32, &dS));
  CK(ws_get(ctx, 2, nb * (6 + k) * 32, &dC));
  CK(ws_get(ctx, 3, nb * 32, &dR));
  CK(ws_get(ctx, 4, 64, &dout));
  CK(h2d(ctx, dP, points, nb * nvar * 64));
  CK(h2d(ctx, dS, scalars, nb * 5 * 32));
  CK(h2d(ctx, dC, challenges, nb * (6 + k) * 32));
  CK(h2d(ctx, dR, rho, nb * 32));
  CK(verify_combined_locked(ctx, g, c, nb, n1, k, dP, dS, dC, dR, dout));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, partial_xy, dout, 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* Screened stream: the combined check of every batch first (one point per batch), the per-proof path only for the batches whose
 * point is not the identity (or that hold a malformed input).  Phase 1 and phase 2 each fork over the lanes and join; between them
 * the host reads 68 bytes per batch. */
// init_states != null: the transcript is replayed on the device (challenges unused): per batch k_verify_transcript first, and a proof
// whose transcript replay fails (a validated point is the identity) sends its batch to the per-proof path like any other failure
static int verify_screened_locked(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                  const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges, const uint8_t *rho,
                                  uint8_t *ok, bool on_host, size_t *fallback_batches, const uint8_t *init_states = nullptr) try {
  if (k >= 32) return BPGPU_E_LEN;
  if (fallback_batches) *fallback_batches = 0;
  if (!nb) return BPGPU_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  // proofs per combined check: the check is a chain of nine mostly latency-bound launches with little work per proof, so its
  // batches are larger than the per-proof path's (2560 x 24 points still fit the one-instance bucket pipeline of k_pip2.hip:
  // 11.2 M/s against 8.1 M/s at 1024); a batch that fails is re-verified proof by proof as a whole
  const size_t m = c->m, nvar = 11 + m + 2 * k, nch = 6 + k;
  size_t batch = (size_t)ctx->opt[BPGPU_OPT_SCREEN_BATCH];
  const size_t fit = ((size_t)1 << 16) / nvar / 64 * 64;          // proof points of one check <= 2^16
  if (fit >= 256 && batch > fit) batch = fit;
  const size_t nlanes_opt = (size_t)ctx->opt[BPGPU_OPT_STREAM_LANES];
  const size_t nchunks = (nb + batch - 1) / batch, nl = nchunks < nlanes_opt ? nchunks : nlanes_opt;
  CK(stream_lanes(ctx, nl));
  void *dpart, *dflag;
  CK(ws_get(ctx, 26, nchunks * 64, &dpart));
  CK(ws_get(ctx, 27, nchunks * 4, &dflag));
  std::vector<uint8_t> hpart(nchunks * 64);
  std::vector<int> hflag(nchunks);
  auto fork = [&]() -> int {
    HIPCK(ctx, hipEventRecord(ctx->lane_ev, ctx->st));
    for (size_t l = 0; l < nl; l++) HIPCK(ctx, hipStreamWaitEvent(ctx->lanes[l]->st, ctx->lane_ev, 0));
    return BPGPU_OK;
  };
  auto join = [&](int rc) -> int {
    for (size_t l = 0; l < nl; l++)
      if (hipEventRecord(ctx->lanes[l]->ev1, ctx->lanes[l]->st) != hipSuccess || hipStreamWaitEvent(ctx->st, ctx->lanes[l]->ev1, 0) != hipSuccess)
        if (rc == BPGPU_OK) { ctx->err = "bpgpu_r1cs_verify_screened: joining the lanes failed"; rc = BPGPU_E_DEVICE; }
    return rc;
  };
  // stage a batch's operands on its lane (host variant); returns device pointers either way
  // (*Cc = the challenges, or with a device transcript the batch's 32-byte initial states)
  const uint8_t *third = init_states ? init_states : challenges;
  const size_t third_bytes = init_states ? 32 : nch * 32;
  auto operands = [&](bpgpu_ctx *ln, size_t lo, size_t cnt, const void **P, const void **S, const void **Cc, const void **R, void **dok) -> int {
    if (!on_host) {
      *P = points + lo * nvar * 64; *S = scalars + lo * 5 * 32; *Cc = third + lo * third_bytes; *R = rho + lo * 32; *dok = ok + lo * 4;
      return BPGPU_OK;
    }
    void *dP, *dS, *dC, *dR;
    int rc;
    if ((rc = ws_get(ln, 0, batch * nvar * 64, &dP)) || (rc = ws_get(ln, 1, batch * 5 * 32, &dS)) || (rc = ws_get(ln, 2, batch * third_bytes, &dC)) ||
        (rc = ws_get(ln, 3, batch * 32, &dR)) || (rc = ws_get(ln, 4, batch * 4, dok)) ||
        (rc = h2d(ln, dP, points + lo * nvar * 64, cnt * nvar * 64)) || (rc = h2d(ln, dS, scalars + lo * 5 * 32, cnt * 5 * 32)) ||
        (rc = h2d(ln, dC, third + lo * third_bytes, cnt * third_bytes)) || (rc = h2d(ln, dR, rho + lo * 32, cnt * 32)))
      return rc;
    *P = dP; *S = dS; *Cc = dC; *R = dR;
    return BPGPU_OK;
  };
  // ---- phase 1: one combined check per batch
  CK(fork());
  int rc = BPGPU_OK;
  for (size_t ci = 0; ci < nchunks && rc == BPGPU_OK; ci++) {
    bpgpu_ctx *ln = ctx->lanes[ci % nl];
    const size_t lo = ci * batch, cnt = nb - lo < batch ? nb - lo : batch;
    const void *P, *S, *Cc, *R;
    void *dok;
    rc = operands(ln, lo, cnt, &P, &S, &Cc, &R, &dok);
    void *dbad = nullptr, *dchi = nullptr;
    if (rc == BPGPU_OK && init_states) {       // the batch's challenges from the device transcript
      Words8 *chp = nullptr;
      rc = fs_transcript_locked(ln, c, cnt, k, Cc, P, S, nullptr, nullptr, nullptr, &chp, &dbad, &dchi);
      Cc = chp;
    }
    if (rc == BPGPU_OK) rc = verify_combined_locked(ln, g, c, cnt, n1, k, P, S, Cc, R, (uint8_t *)dpart + 64 * ci);
    if (rc == BPGPU_OK && dbad) or_flag(ln->st, (const int32_t *)dbad, cnt, ln->d_flag);      // (after the combined check: it resets the flag)
    if (rc == BPGPU_OK) zero_flag(ln->st, (const Words8 *)R, cnt, ln->d_flag);                // a zero weight voids the check for its batch
    if (rc == BPGPU_OK && hipMemcpyAsync((int *)dflag + ci, ln->d_flag, 4, hipMemcpyDeviceToDevice, ln->st) != hipSuccess) rc = BPGPU_E_DEVICE;
    if (rc) ctx->err = ln->err;
  }
  rc = join(rc);
  if (rc) return rc;
  CK(d2h(ctx, hpart.data(), dpart, nchunks * 64));
  CK(d2h(ctx, hflag.data(), dflag, nchunks * 4));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  // ---- phase 2: all-accept for the batches that passed, the per-proof verification for the others
  size_t nfall = 0;
  CK(fork());
  for (size_t ci = 0; ci < nchunks && rc == BPGPU_OK; ci++) {
    const size_t lo = ci * batch, cnt = nb - lo < batch ? nb - lo : batch;
    bool pass = hflag[ci] == 0;
    for (size_t i = 0; i < 64 && pass; i++) pass = hpart[64 * ci + i] == 0;
    if (pass) {
      if (on_host) for (size_t i = 0; i < cnt; i++) ((int32_t *)ok)[lo + i] = 1;
      else if (hipMemsetD32Async((hipDeviceptr_t)(ok + lo * 4), 1, cnt, ctx->st) != hipSuccess) rc = BPGPU_E_DEVICE;
      continue;
    }
    bpgpu_ctx *ln = ctx->lanes[nfall++ % nl];
    const void *P, *S, *Cc, *R;
    void *dok;
    (void)((rc = operands(ln, lo, cnt, &P, &S, &Cc, &R, &dok)) ||
           (rc = init_states ? verify_fs_locked(ln, g, c, cnt, n1, k, Cc, P, S, dok, nullptr, nullptr)
                             : verify_batch_dev_locked(ln, g, c, cnt, n1, k, P, S, Cc, dok, nullptr, nullptr)) ||
           (on_host && (rc = d2h(ln, ok + lo * 4, dok, cnt * 4))));
    if (rc) ctx->err = ln->err;
  }
  rc = join(rc);
  if (fallback_batches) *fallback_batches = nfall;
  return rc;
} catch (const std::bad_alloc &) {
  return BPGPU_E_OOM;
}
int bpgpu_r1cs_verify_screened_fs_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                      const void *init_states_dev, const void *points_dev, const void *scalars_dev, const void *rho_dev,
                                      void *ok_dev, size_t *fallback_batches) {
  if (!ctx || !g || !c || (nb && (!init_states_dev || !points_dev || !scalars_dev || !rho_dev || !ok_dev))) return BPGPU_E_ARG;
  if (c->nchi) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_screened_locked(ctx, g, c, nb, n1, k, (const uint8_t *)points_dev, (const uint8_t *)scalars_dev, nullptr,
                                (const uint8_t *)rho_dev, (uint8_t *)ok_dev, false, fallback_batches, (const uint8_t *)init_states_dev);
}
int bpgpu_r1cs_verify_screened_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                   const void *points_dev, const void *scalars_dev, const void *challenges_dev, const void *rho_dev,
                                   void *ok_dev, size_t *fallback_batches) {
  if (!ctx || !g || !c || (nb && (!points_dev || !scalars_dev || !challenges_dev || !rho_dev || !ok_dev))) return BPGPU_E_ARG;
  if (c->nchi) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return verify_screened_locked(ctx, g, c, nb, n1, k, (const uint8_t *)points_dev, (const uint8_t *)scalars_dev,
                                (const uint8_t *)challenges_dev, (const uint8_t *)rho_dev, (uint8_t *)ok_dev, false, fallback_batches);
}
int bpgpu_r1cs_verify_screened(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                               const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges, const uint8_t *rho, int32_t *ok,
                               size_t *fallback_batches) {
  if (!ctx || !g || !c || (nb && (!points || !scalars || !challenges || !rho || !ok))) return BPGPU_E_ARG;
  if (c->nchi) return BPGPU_E_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (fallback_batches) *fallback_batches = 0;
  if (!nb) return BPGPU_OK;
  HIPCK(ctx, hipSetDevice(ctx->device));
  if (ctx->pinned_cap < nb * 4) {       // verdicts of the fallback batches come back through page-locked staging (bpgpu_r1cs_verify_stream)
    if (ctx->pinned) { HIPCK(ctx, hipStreamSynchronize(ctx->st)); (void)hipHostFree(ctx->pinned); ctx->pinned = nullptr; ctx->pinned_cap = 0; }
    HIPCK(ctx, hipHostMalloc(&ctx->pinned, nb * 4 + nb, hipHostMallocDefault));
    ctx->pinned_cap = nb * 4 + nb;
  }
  CK(verify_screened_locked(ctx, g, c, nb, n1, k, points, scalars, challenges, rho, (uint8_t *)ctx->pinned, true, fallback_batches));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  memcpy(ok, ctx->pinned, nb * 4);
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- IPP prover session */
// (ctx->mu held) the session's buffers go back to the context's pool
static void ipp_free_all(bpgpu_ctx *ctx, bpgpu_ipp *s) {
  void *all[] = {s->a[0], s->a[1], s->b[0], s->b[1], s->G[0], s->G[1], s->H[0], s->H[1], s->Q, s->Gf, s->Hf, s->t1, s->t2, s->t3, s->t4,
                 s->cLR, s->uu, s->res, s->sums, s->out_xy, s->mpts, s->msc, s->cG, s->cH, s->w};
  for (void *p : all) pool_release(ctx, p);
  delete s;
}
int bpgpu_ipp_begin(bpgpu_ctx *ctx, size_t nb, size_t n, const uint8_t *Q, const uint8_t *G_factors,
                    const uint8_t *H_factors, const uint8_t *G, const uint8_t *H, int shared_gens, const uint8_t *a,
                    const uint8_t *b, bpgpu_ipp **out) {
  if (!ctx || !out || !nb || !Q || !G_factors || !H_factors || !G || !H || !a || !b) return BPGPU_E_ARG;
  if (!n || (n & (n - 1))) return BPGPU_E_LEN;   // assert!(n.is_power_of_two()), inner_product_proof.rs:70
  *out = nullptr;
  // ONE proof over arbitrary generators (the reference's `ipp-prover` criterion bench, benches/inner_product.rs:34-64; the tail of a
  // vector-sharded proof): the literal schedule folds G and H every round -- two dependent 252-doubling chains, ~2.8 ms per round
  // however small n is.  Instead build fixed-base tables for THESE generators once (B = B_blinding = Q, w = 1) and run the
  // resident-generator session: a round becomes table lookups.  Same group elements, same bytes.  BPGPU_OPT_IPP_LITERAL keeps the
  // literal schedule (nb > 1 with per-proof Q always takes it).  The tables are per session (256 KB per generator at c = 8 up to
  // n = 1024, 32 KB at c = 4 above: 2 GB at n = 2^15) and their build grows with n, so the route is bounded: n <=
  // BPGPU_OPT_IPP_TABLE_MAX_N (default 2^16), tables + staging within half of the free device memory, and a table build that
  // runs out of memory falls back to the literal schedule, whose footprint is O(n).
  int64_t literal = 0, table_max_n = 0;
  (void)bpgpu_get_option(ctx, BPGPU_OPT_IPP_LITERAL, &literal);
  (void)bpgpu_get_option(ctx, BPGPU_OPT_IPP_TABLE_MAX_N, &table_max_n);
  if (nb == 1 && n >= 2 && !literal && n <= (size_t)table_max_n) {
    const int c = n <= 1024 ? 8 : 4;
    const size_t per_gen = (252 / c + 1) * ((size_t)1 << (c - 1)), ng = 2 + 2 * n;
    size_t stage_gens = ((size_t)8 << 30) / ((per_gen + 252 / c + 1) * sizeof(JacRaw));
    if (stage_gens > ng) stage_gens = ng;
    const size_t need = ng * per_gen * sizeof(AffDev) + stage_gens * (per_gen + 252 / c + 1) * sizeof(JacRaw) + ng * 192;
    size_t free_b = 0, total_b = 0;
    bool fits = false;
    { std::lock_guard<std::mutex> lk(ctx->mu);
      fits = hipSetDevice(ctx->device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 2; }
    if (fits) {
      bpgpu_gens *g = nullptr;
      int rc = bpgpu_gens_create(ctx, G, H, n, Q, Q, c, &g);
      if (rc == BPGPU_OK) {
        uint8_t one[32] = {1};
        rc = bpgpu_ipp_begin_gens(ctx, g, 1, n, one, G_factors, H_factors, a, b, out);
        if (rc == BPGPU_OK) { (*out)->own_gens = g; return BPGPU_OK; }
        bpgpu_gens_destroy(ctx, g);
      }
      if (rc != BPGPU_E_OOM) return rc;     // out of memory: the literal schedule below
    }
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  bpgpu_ipp *s = new (std::nothrow) bpgpu_ipp();
  if (!s) return BPGPU_E_OOM;
  s->nb = nb; s->n0 = s->n = n; s->shared_gens = shared_gens != 0;
  size_t tot = nb * n, gtot = shared_gens ? n : tot, half = nb * (n > 1 ? n / 2 : 1);
  void *stage = nullptr;
  bool okk = true;
  auto M = [&](void **p, size_t bytes) { if (okk && !pool_alloc(ctx, p, bytes)) okk = false; };
  M((void **)&s->a[0], tot * 32); M((void **)&s->b[0], tot * 32); M((void **)&s->a[1], half * 32); M((void **)&s->b[1], half * 32);
  M((void **)&s->G[0], (gtot > half ? gtot : half) * sizeof(AffDev)); M((void **)&s->H[0], (gtot > half ? gtot : half) * sizeof(AffDev));
  M((void **)&s->G[1], half * sizeof(AffDev)); M((void **)&s->H[1], half * sizeof(AffDev));
  M((void **)&s->Q, nb * sizeof(AffDev)); M((void **)&s->Gf, tot * 32); M((void **)&s->Hf, tot * 32);
  M((void **)&s->t1, half * 32); M((void **)&s->t2, half * 32); M((void **)&s->t3, half * 32); M((void **)&s->t4, half * 32);
  M((void **)&s->cLR, nb * 2 * 32); M((void **)&s->uu, nb * 2 * 32);
  M((void **)&s->res, nb * 2 * (n + 1) * sizeof(JacRaw)); M((void **)&s->sums, nb * 2 * sizeof(JacRaw));
  M((void **)&s->out_xy, nb * 2 * 64);
  M((void **)&s->mpts, nb * 2 * (n + 1) * sizeof(AffDev)); M((void **)&s->msc, nb * 2 * (n + 1) * 32);
  M(&stage, (2 * gtot + nb) * 64);
  if (!okk) { pool_release(ctx, stage); ipp_free_all(ctx, s); return BPGPU_E_OOM; }
  int rc = BPGPU_OK;
  do {
    if ((rc = flag_reset(ctx))) break;
    if ((rc = h2d(ctx, s->a[0], a, tot * 32)) || (rc = h2d(ctx, s->b[0], b, tot * 32)) ||
        (rc = h2d(ctx, s->Gf, G_factors, tot * 32)) || (rc = h2d(ctx, s->Hf, H_factors, tot * 32))) break;
    uint8_t *st8 = (uint8_t *)stage;
    if ((rc = h2d(ctx, st8, G, gtot * 64)) || (rc = h2d(ctx, st8 + gtot * 64, H, gtot * 64)) ||
        (rc = h2d(ctx, st8 + 2 * gtot * 64, Q, nb * 64))) break;
    scalars_check(ctx->st, s->a[0], tot, ctx->d_flag);
    scalars_check(ctx->st, s->b[0], tot, ctx->d_flag);
    scalars_check(ctx->st, s->Gf, tot, ctx->d_flag);
    scalars_check(ctx->st, s->Hf, tot, ctx->d_flag);
    points_from_boundary(ctx->st, (Words8 *)st8, s->G[0], gtot, ctx->d_flag);
    points_from_boundary(ctx->st, (Words8 *)(st8 + gtot * 64), s->H[0], gtot, ctx->d_flag);
    points_from_boundary(ctx->st, (Words8 *)(st8 + 2 * gtot * 64), s->Q, nb, ctx->d_flag);
    if ((rc = launch_ok(ctx))) break;
    int bad = 0;
    if ((rc = flag_read(ctx, &bad))) break;
    if (bad) rc = BPGPU_E_ARG;
  } while (0);
  (void)hipStreamSynchronize(ctx->st);   // the staging buffer goes back to the pool: its conversion launches have completed
  pool_release(ctx, stage);
  if (rc) { ipp_free_all(ctx, s); return rc; }
  *out = s;
  return BPGPU_OK;
}
int bpgpu_ipp_begin_gens(bpgpu_ctx *ctx, const bpgpu_gens *g, size_t nb, size_t n, const uint8_t *w,
                         const uint8_t *G_factors, const uint8_t *H_factors, const uint8_t *a, const uint8_t *b,
                         bpgpu_ipp **out) {
  if (!ctx || !g || !out || !nb || !w || !G_factors || !H_factors || !a || !b) return BPGPU_E_ARG;
  if (!n || (n & (n - 1))) return BPGPU_E_LEN;   // assert!(n.is_power_of_two()), inner_product_proof.rs:70
  if (n > g->cap) return BPGPU_E_GENS;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  bpgpu_ipp *s = new (std::nothrow) bpgpu_ipp();
  if (!s) return BPGPU_E_OOM;
  s->nb = nb; s->n0 = s->n = n; s->gens = g;
  if (ctx->shard_world > 1) { shard_bounds(n, ctx->shard_rank, ctx->shard_world, &s->slo, &s->shi); s->with_q = ctx->shard_rank == 0; }
  size_t tot = nb * n, half = nb * (n > 1 ? n / 2 : 1);
  bool okk = true;
  auto M = [&](void **p, size_t bytes) { if (okk && !pool_alloc(ctx, p, bytes)) okk = false; };
  M((void **)&s->a[0], tot * 32); M((void **)&s->b[0], tot * 32); M((void **)&s->a[1], half * 32); M((void **)&s->b[1], half * 32);
  M((void **)&s->cG, tot * 32); M((void **)&s->cH, tot * 32); M((void **)&s->w, nb * 32);
  M((void **)&s->cLR, nb * 2 * 32); M((void **)&s->uu, nb * 2 * 32);
  M((void **)&s->sums, nb * 2 * sizeof(JacRaw)); M((void **)&s->out_xy, nb * 2 * 64);
  M((void **)&s->msc, nb * 2 * (2 + 2 * n) * 32);
  if (!okk) { ipp_free_all(ctx, s); return BPGPU_E_OOM; }
  int rc = BPGPU_OK;
  do {
    if ((rc = flag_reset(ctx))) break;
    if ((rc = h2d(ctx, s->a[0], a, tot * 32)) || (rc = h2d(ctx, s->b[0], b, tot * 32)) ||
        (rc = h2d(ctx, s->cG, G_factors, tot * 32)) || (rc = h2d(ctx, s->cH, H_factors, tot * 32)) ||
        (rc = h2d(ctx, s->w, w, nb * 32))) break;
    scalars_check(ctx->st, s->a[0], tot, ctx->d_flag);
    scalars_check(ctx->st, s->b[0], tot, ctx->d_flag);
    scalars_check(ctx->st, s->cG, tot, ctx->d_flag);
    scalars_check(ctx->st, s->cH, tot, ctx->d_flag);
    scalars_check(ctx->st, s->w, nb, ctx->d_flag);
    if ((rc = launch_ok(ctx))) break;
    int bad = 0;
    if ((rc = flag_read(ctx, &bad))) break;
    if (bad) rc = BPGPU_E_ARG;
  } while (0);
  if (rc) { ipp_free_all(ctx, s); return rc; }
  *out = s;
  return BPGPU_OK;
}
void bpgpu_ipp_destroy(bpgpu_ctx *ctx, bpgpu_ipp *s) {
  if (!s || !ctx) return;        // (a session belongs to the context it was opened on)
  bpgpu_gens *own = s->own_gens;
  { std::lock_guard<std::mutex> lk(ctx->mu); hipStreamSynchronize(ctx->st); hipStreamSynchronize(ctx->st2); ipp_free_all(ctx, s); }
  if (own) bpgpu_gens_destroy(ctx, own);
}
size_t bpgpu_ipp_len(const bpgpu_ipp *s) { return s ? s->n : 0; }

/* c_L, c_R and the two MSMs of one round -- inner_product_proof.rs:87-114 (first) / :156-172.
 * Device part: out_xy[2p], out_xy[2p + 1] = L_p, R_p in boundary form (2 Words8 per point). */
static int ipp_round_dev(bpgpu_ctx *ctx, bpgpu_ipp *s, Words8 *out_xy) {
  hipStream_t st = ctx->st;
  const size_t nb = s->nb, n = s->n, h = n / 2, seg = 2 * h + 1;
  Words8 *a = s->a[s->cur], *b = s->b[s->cur];
  if (s->gens) {   // resident generators: two table-lookup MSMs over the original generators per proof
    sc_dot_batched(st, nb, h, a, n, b + h, n, s->cLR, 2);        // c_L = <a_L, b_R>
    sc_dot_batched(st, nb, h, a + h, n, b, n, s->cLR + 1, 2);    // c_R = <a_R, b_L>
    ipp_gens_scalars(st, nb, s->n0, n, a, b, s->cG, s->cH, s->cLR, s->w, s->msc, s->slo, s->shi, s->with_q);
    {
      size_t chunks = fixed_msm_ipp_chunks(s->gens->c, s->n0, nb * 2);
      void *dpart = nullptr;
      if (chunks > 1) CK(ws_get(ctx, 12, nb * 2 * chunks * sizeof(JacRaw), &dpart));
      ProfScope ps(ctx, 21, st);
      // (out_xy == nullptr: the caller's fused round tail sums the chunk partials and converts the points itself)
      const bool tail_sums = !out_xy && chunks > 1 && chunks <= 256;
      fixed_msm_ipp(st, s->gens->c, s->gens->table, s->n0, s->gens->cap, n, (const uint32_t *)s->msc, s->sums, nb * 2, (JacRaw *)dpart, !tail_sums);
      s->tail_partials = tail_sums ? (const JacRaw *)dpart : nullptr;
      s->tail_chunks = tail_sums ? chunks : 0;
    }
    if (out_xy) jac_to_boundary(st, s->sums, out_xy, nb * 2);
    return launch_ok(ctx);
  }
  const AffDev *G = s->G[s->cur], *H = s->H[s->cur];
  const bool shared = s->first && s->shared_gens;
  const size_t gouter = shared ? 0 : n;
  void *dstr;
  CK(straus_ws(ctx, 1, nb * h, &dstr));
  sc_dot_batched(st, nb, h, a, n, b + h, n, s->cLR, 2);        // c_L = <a_L, b_R>
  sc_dot_batched(st, nb, h, a + h, n, b, n, s->cLR + 1, 2);    // c_R = <a_R, b_L>
  const Words8 *sLa = a, *sLb = b + h, *sRa = a + h, *sRb = b;
  size_t so = n;   // outer stride (Words8) of the scalar sources
  if (s->first) {  // fold the factors into the scalars, :90-114
    sc_mul_strided(st, nb, h, a, n, 1, s->Gf + h, s->n0, 1, s->t1);       // a_L * G_factors[n..2n]
    sc_mul_strided(st, nb, h, b + h, n, 1, s->Hf, s->n0, 1, s->t2);       // b_R * H_factors[0..n]
    sc_mul_strided(st, nb, h, a + h, n, 1, s->Gf, s->n0, 1, s->t3);       // a_R * G_factors[0..n]
    sc_mul_strided(st, nb, h, b, n, 1, s->Hf + h, s->n0, 1, s->t4);       // b_L * H_factors[n..2n]
    sLa = s->t1; sLb = s->t2; sRa = s->t3; sRb = s->t4; so = h;
  }
  const size_t pip_min = (size_t)ctx->opt[BPGPU_OPT_IPP_PIPPENGER_MIN];
  {
    // make the 2 nb MSM instances contiguous: L = [a_L | b_R | c_L] x [G_R | H_L | Q], then R = [a_R | b_L | c_R] x [G_L | H_R | Q]
    const size_t io = 2 * seg;
    gather_points(st, G + h, gouter, h, nb, s->mpts, io);            gather_scalars(st, sLa, so, h, nb, s->msc, io);
    gather_points(st, H, gouter, h, nb, s->mpts + h, io);            gather_scalars(st, sLb, so, h, nb, s->msc + h, io);
    gather_points(st, s->Q, 1, 1, nb, s->mpts + 2 * h, io);          gather_scalars(st, s->cLR, 2, 1, nb, s->msc + 2 * h, io);
    gather_points(st, G, gouter, h, nb, s->mpts + seg, io);          gather_scalars(st, sRa, so, h, nb, s->msc + seg, io);
    gather_points(st, H + h, gouter, h, nb, s->mpts + seg + h, io);  gather_scalars(st, sRb, so, h, nb, s->msc + seg + h, io);
    gather_points(st, s->Q, 1, 1, nb, s->mpts + seg + 2 * h, io);    gather_scalars(st, s->cLR + 1, 2, 1, nb, s->msc + seg + 2 * h, io);
  }
  if (seg >= pip_min) {   // bucket method, one batched launch for all instances
    int cw = pippenger_window(seg);
    void *dpip;
    CK(ws_get(ctx, 14, pippenger_scratch_bytes_batch(nb * 2, seg, cw), &dpip));
    pippenger_batch(st, s->mpts, (const uint32_t *)s->msc, nb * 2, seg, cw, s->sums, 1, dpip);
  } else {                // short rounds: one Straus lane per term (a single launch), then a tree sum per instance
    StrausArgs x{};
    x.pts[0] = s->mpts; x.pt_stride[0] = 1;
    x.sc[0] = (const uint32_t *)s->msc; x.sc_stride[0] = 8;
    CK(straus_ws(ctx, 1, nb * 2 * seg, &dstr));
    straus(st, 1, x, s->res, nb * 2 * seg, dstr);
    segmented_sum(st, s->res, s->sums, nb * 2, seg);
  }
  jac_to_boundary(st, s->sums, out_xy, nb * 2);
  return launch_ok(ctx);
}
int bpgpu_ipp_round(bpgpu_ctx *ctx, bpgpu_ipp *s, uint8_t *L, uint8_t *R) try {
  if (!ctx || !s || !L || !R) return BPGPU_E_ARG;
  if (s->n < 2) return BPGPU_E_LEN;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t nb = s->nb;
  CK(ipp_round_dev(ctx, s, s->out_xy));
  std::vector<uint8_t> tmp(nb * 128);
  CK(d2h(ctx, tmp.data(), s->out_xy, nb * 128));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  for (size_t p = 0; p < nb; p++) { memcpy(L + 64 * p, &tmp[128 * p], 64); memcpy(R + 64 * p, &tmp[128 * p + 64], 64); }
  return BPGPU_OK;
} catch (const std::bad_alloc &) {   // host-side staging (std::vector): no exception crosses the C ABI
  return BPGPU_E_OOM;
}
/* fold_witness with the round's challenges -- inner_product_proof.rs:125-146 (first) / :183-184 */
// device part of the fold: du / dui = the round's challenges and their inverses (nb each) already in HBM
static int ipp_fold_dev(bpgpu_ctx *ctx, bpgpu_ipp *s, Words8 *du, Words8 *dui) {
  hipStream_t st = ctx->st;
  const size_t nb = s->nb, n = s->n, h = n / 2;
  const int cur = s->cur, nxt = cur ^ 1;
  if (s->gens) {
    ipp_gens_fold(st, nb, s->n0, n, du, dui, s->cG, s->cH);
    fold_scalars_batched(st, nb, h, du, dui, s->a[cur], s->b[cur], s->a[nxt], s->b[nxt]);
    s->cur = nxt; s->n = h; s->first = false;
    return launch_ok(ctx);
  }
  const AffDev *G = s->G[cur], *H = s->H[cur];
  const bool shared = s->first && s->shared_gens;
  const size_t gouter = shared ? 0 : n;
  JacRaw *fres = s->res;   // reuse: nb x h (G) then nb x h (H)
  StrausArgs g{}, hh{};
  g.pts[0] = G; g.pts[1] = G + h; hh.pts[0] = H; hh.pts[1] = H + h;
  for (int j = 0; j < 2; j++) { g.pt_stride[j] = hh.pt_stride[j] = 1; g.pt_outer[j] = hh.pt_outer[j] = gouter; }
  g.inner = hh.inner = h;
  if (s->first) {   // G_i <- G_factors_i * G_i folded into the fold scalars, :125-134
    sc_mul_strided(st, nb, h, s->Gf, s->n0, 1, dui, 1, 0, s->t1);       // u^-1 * gf_i
    sc_mul_strided(st, nb, h, s->Gf + h, s->n0, 1, du, 1, 0, s->t2);    // u    * gf_{h+i}
    sc_mul_strided(st, nb, h, s->Hf, s->n0, 1, du, 1, 0, s->t3);        // u    * hf_i
    sc_mul_strided(st, nb, h, s->Hf + h, s->n0, 1, dui, 1, 0, s->t4);   // u^-1 * hf_{h+i}
    g.sc[0] = (uint32_t *)s->t1; g.sc[1] = (uint32_t *)s->t2; hh.sc[0] = (uint32_t *)s->t3; hh.sc[1] = (uint32_t *)s->t4;
    for (int j = 0; j < 2; j++) { g.sc_stride[j] = hh.sc_stride[j] = 8; g.sc_outer[j] = hh.sc_outer[j] = h * 8; }
  } else {
    g.sc[0] = (uint32_t *)dui; g.sc[1] = (uint32_t *)du; hh.sc[0] = (uint32_t *)du; hh.sc[1] = (uint32_t *)dui;
    for (int j = 0; j < 2; j++) { g.sc_stride[j] = hh.sc_stride[j] = 0; g.sc_outer[j] = hh.sc_outer[j] = 8; }
  }
  // the G and the H fold are independent 252-doubling chains: run them side by side on the context's two streams
  void *dstr, *dstr2;
  CK(straus_ws(ctx, 2, nb * h, &dstr));
  CK(ws_get(ctx, 12, straus_scratch_bytes(2, nb * h), &dstr2));
  hipStream_t st2 = ctx->st2;
  HIPCK(ctx, hipEventRecord(ctx->ev1, st));
  HIPCK(ctx, hipStreamWaitEvent(st2, ctx->ev1, 0));
  straus(st, 2, g, fres, nb * h, dstr);
  straus(st2, 2, hh, fres + nb * h, nb * h, dstr2);
  // when the input generators are shared the folded ones become per-proof: write them to buffer nxt
  batch_normalize(st, fres, s->G[nxt], nb * h, 8);
  batch_normalize(st2, fres + nb * h, s->H[nxt], nb * h, 8);
  HIPCK(ctx, hipEventRecord(ctx->ev2, st2));
  fold_scalars_batched(st, nb, h, du, dui, s->a[cur], s->b[cur], s->a[nxt], s->b[nxt]);
  HIPCK(ctx, hipStreamWaitEvent(st, ctx->ev2, 0));
  s->cur = nxt; s->n = h; s->first = false;
  return launch_ok(ctx);
}
int bpgpu_ipp_fold(bpgpu_ctx *ctx, bpgpu_ipp *s, const uint8_t *u, const uint8_t *u_inv) {
  if (!ctx || !s || !u || !u_inv) return BPGPU_E_ARG;
  if (s->n < 2) return BPGPU_E_LEN;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t nb = s->nb;
  Words8 *du = s->uu, *dui = s->uu + nb;
  CK(flag_reset(ctx));
  CK(h2d(ctx, du, u, nb * 32));
  CK(h2d(ctx, dui, u_inv, nb * 32));
  scalars_check(ctx->st, s->uu, 2 * nb, ctx->d_flag);
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;        // checked before the session state advances
  return ipp_fold_dev(ctx, s, du, dui);
}
/* InnerProductProof::create's whole round loop on the device (SURVEY 8f N1 applied to the prover): per round the
 * L, R MSMs, transcript.append_point("L"), ("R"), challenge_scalar("u") (inner_product_proof.rs:119-123,177-181)
 * with the keccak hash chain in a kernel, u^-1 and the fold -- no host round trip between rounds. */
int bpgpu_ipp_run_fs(bpgpu_ctx *ctx, bpgpu_ipp *s, const uint8_t *states_in, uint8_t *L_out, uint8_t *R_out,
                     uint8_t *a_out, uint8_t *b_out, uint8_t *states_out) try {
  if (!ctx || !s || !states_in || !a_out || !b_out) return BPGPU_E_ARG;
  if (s->shi != (size_t)-1) return BPGPU_E_ARG;    // a sharded session's L, R are partial sums: its rounds need the ranks' exchange
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t nb = s->nb;
  size_t k = 0;
  for (size_t t = s->n; t > 1; t >>= 1) k++;
  if (k && (!L_out || !R_out)) return BPGPU_E_ARG;
  void *dstates, *dlr, *dzero;
  CK(ws_get(ctx, 16, nb * 32, &dstates));
  CK(ws_get(ctx, 17, (k ? k : 1) * nb * 128, &dlr));
  CK(ws_get(ctx, 18, 4, &dzero));
  CK(h2d(ctx, dstates, states_in, nb * 32));
  Words8 *du = s->uu, *dui = s->uu + nb;
  ProfSpan span(ctx, 20, ctx->st);
  for (size_t r = 0; r < k; r++) {
    Words8 *lr = (Words8 *)dlr + r * nb * 4;             // 2 points x 2 Words8 per proof
    if (s->gens) {   // resident generators: point conversion, the three transcript steps and u^-1 in ONE launch (k_ipp_round_tail)
      CK(ipp_round_dev(ctx, s, nullptr));
      ipp_round_tail(ctx->st, nb, s->sums, (uint64_t *)dstates, lr, du, dui, s->tail_partials, s->tail_chunks);
    } else {
      CK(ipp_round_dev(ctx, s, lr));
      ipp_round_challenge(ctx->st, nb, (uint64_t *)dstates, lr, du);
      HIPCK(ctx, hipMemcpyAsync(dui, du, nb * 32, hipMemcpyDeviceToDevice, ctx->st));
      batch_inverse(ctx->st, dui, nb, (int *)dzero);       // challenges are non-zero up to 2^-252
    }
    CK(ipp_fold_dev(ctx, s, du, dui));
  }
  span.close();
  std::vector<uint8_t> tmp(k * nb * 128);
  if (k) CK(d2h(ctx, tmp.data(), dlr, k * nb * 128));
  CK(d2h(ctx, a_out, s->a[s->cur], nb * 32));
  CK(d2h(ctx, b_out, s->b[s->cur], nb * 32));
  if (states_out) CK(d2h(ctx, states_out, dstates, nb * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  for (size_t p = 0; p < nb; p++)       // proof-major outputs: L_out[p][r], R_out[p][r]
    for (size_t r = 0; r < k; r++) {
      memcpy(L_out + (p * k + r) * 64, &tmp[(r * nb + p) * 128], 64);
      memcpy(R_out + (p * k + r) * 64, &tmp[(r * nb + p) * 128 + 64], 64);
    }
  return BPGPU_OK;
} catch (const std::bad_alloc &) {   // host-side staging (std::vector): no exception crosses the C ABI
  return BPGPU_E_OOM;
}
/* final a, b -- inner_product_proof.rs:187-192 */
// The generators a resident-generator session has folded SO FAR, as points: G'_t = sum_{i = t mod n} cG[i] G_i (and H'), n = the
// session's current length.  For n == 1 this is the pair (G', H') the remaining state (a, b) refers to -- what a rank of a
// vector-sharded IPP (sharding.sharded_ipp_create: SURVEY 8e.2) hands to the final log2(ranks) rounds.  Only n == 1 is exposed.
int bpgpu_ipp_folded_gens(bpgpu_ctx *ctx, bpgpu_ipp *s, uint8_t *G_out, uint8_t *H_out) try {
  if (!ctx || !s || !G_out || !H_out || !s->gens) return BPGPU_E_ARG;
  if (s->n != 1) return BPGPU_E_LEN;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t nb = s->nb, n0 = s->n0, per = 2 + 2 * n0;
  // MSM 2p over [B, Bb, G.., H..]: scalars cG[p] on the G block; MSM 2p + 1: cH[p] on the H block
  HIPCK(ctx, hipMemsetAsync(s->msc, 0, nb * 2 * per * 32, ctx->st));
  HIPCK(ctx, hipMemcpy2DAsync((uint8_t *)s->msc + 2 * 32, 2 * per * 32, s->cG, n0 * 32, n0 * 32, nb, hipMemcpyDeviceToDevice, ctx->st));
  HIPCK(ctx, hipMemcpy2DAsync((uint8_t *)s->msc + (per + 2 + n0) * 32, 2 * per * 32, s->cH, n0 * 32, n0 * 32, nb, hipMemcpyDeviceToDevice, ctx->st));
  CK(msm_gens_dev(ctx, s->gens, 2 * nb, n0, (const uint32_t *)s->msc, s->sums, ctx->st));
  jac_to_boundary(ctx->st, s->sums, s->out_xy, 2 * nb);
  CK(launch_ok(ctx));
  std::vector<uint8_t> tmp(nb * 2 * 64);
  CK(d2h(ctx, tmp.data(), s->out_xy, nb * 2 * 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  for (size_t p = 0; p < nb; p++) { memcpy(G_out + 64 * p, &tmp[128 * p], 64); memcpy(H_out + 64 * p, &tmp[128 * p + 64], 64); }
  return BPGPU_OK;
} catch (const std::bad_alloc &) {   // host-side staging (std::vector): no exception crosses the C ABI
  return BPGPU_E_OOM;
}
int bpgpu_ipp_finish(bpgpu_ctx *ctx, bpgpu_ipp *s, uint8_t *a_out, uint8_t *b_out) {
  if (!ctx || !s || !a_out || !b_out) return BPGPU_E_ARG;
  if (s->n != 1) return BPGPU_E_LEN;
  std::lock_guard<std::mutex> lk(ctx->mu);
  CK(d2h(ctx, a_out, s->a[s->cur], s->nb * 32));
  CK(d2h(ctx, b_out, s->b[s->cur], s->nb * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

/* ---------------------------------------------------------------- R1CS prover polynomials */
/* One call = prover.rs:587-619: flattened_constraints(z), exp_y / exp_y_inv, the l/r coefficient
 * vectors and t_1..t_6 (util.rs:152-170) for nb provers of one circuit; then prover.rs:659-672
 * (l(x), r(x), padding) once the host transcript has produced x. */
struct bpgpu_prover {
  size_t nb = 0, n = 0, m = 0;
  int32_t *polys = nullptr;   // [6][nb][n][9]
  Words8 *y = nullptr;        // nb
  // resident-witness sessions (bpgpu_r1cs_prover_commit): the witness and blinding planes, nb x wn plain canonical words, stay in
  // HBM from the phase commitments to the polynomial build
  size_t wn = 0;
  Words8 *aL = nullptr, *aR = nullptr, *aO = nullptr, *sL = nullptr, *sR = nullptr;
  Words8 *yinv = nullptr;     // nb (set by bpgpu_r1cs_prover_session_polys)
};
static void prover_free_all(bpgpu_ctx *ctx, bpgpu_prover *s) {   // ctx->mu held
  void *all[] = {s->polys, s->y, s->aL, s->aR, s->aO, s->sL, s->sR, s->yinv};
  for (void *p : all) pool_release(ctx, p);
  delete s;
}
static int prover_polys_impl(bpgpu_ctx *ctx, const bpgpu_circuit *c, size_t nb, const uint8_t *y, const uint8_t *y_inv,
                             const uint8_t *z, const uint8_t *a_L, const uint8_t *a_R, const uint8_t *a_O,
                             const uint8_t *s_L, const uint8_t *s_R, uint8_t *t_coeffs, uint8_t *wV,
                             bpgpu_prover **out, bool ark) {
  if (!ctx || !c || !out || !nb || !y || !y_inv || !z || !t_coeffs || (c->m && !wV)) return BPGPU_E_ARG;
  if (c->nchi) return BPGPU_E_ARG;      // the prover knows its gadget challenges when it builds the rows: numeric circuits only
  size_t n = c->n, m = c->m;
  if (n && (!a_L || !a_R || !a_O || !s_L || !s_R)) return BPGPU_E_ARG;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  bpgpu_prover *s = new (std::nothrow) bpgpu_prover();
  if (!s) return BPGPU_E_OOM;
  s->nb = nb; s->n = n; s->m = m;
  auto fail = [&](int rc) { prover_free_all(ctx, s); return rc; };
  if (!pool_alloc(ctx, (void **)&s->polys, (6 * nb * (n ? n : 1) * 9) * 4) || !pool_alloc(ctx, (void **)&s->y, nb * 32))
    return fail(BPGPU_E_OOM);
  void *din, *dzp, *dout;
  size_t tot = nb * n;
  int rc;
  if ((rc = ws_get(ctx, 0, (3 * nb + 5 * tot) * 32, &din)) || (rc = ws_get(ctx, 6, nb * (c->q ? c->q : 1) * 9 * 4, &dzp)) ||
      (rc = ws_get(ctx, 1, (nb * 6 + nb * m) * 32, &dout)))
    return fail(rc);
  Words8 *w = (Words8 *)din;
  Words8 *dy = w, *dyi = w + nb, *dz = w + 2 * nb, *dL = w + 3 * nb, *dR = dL + tot, *dO = dR + tot, *dsL = dO + tot, *dsR = dsL + tot;
  Words8 *dt = (Words8 *)dout, *dwV = dt + nb * 6;
  if ((rc = flag_reset(ctx)) || (rc = h2d(ctx, dy, y, nb * 32)) || (rc = h2d(ctx, dyi, y_inv, nb * 32)) ||
      (rc = h2d(ctx, dz, z, nb * 32)) || (rc = h2d(ctx, dL, a_L, tot * 32)) || (rc = h2d(ctx, dR, a_R, tot * 32)) ||
      (rc = h2d(ctx, dO, a_O, tot * 32)) || (rc = h2d(ctx, dsL, s_L, tot * 32)) || (rc = h2d(ctx, dsR, s_R, tot * 32)))
    return fail(rc);
  if (ark) scalars_from_ark(ctx->st, w, w, 3 * nb + 5 * tot, ctx->d_flag);
  else scalars_check(ctx->st, w, 3 * nb + 5 * tot, ctx->d_flag);
  if (hipMemcpyAsync(s->y, dy, nb * 32, hipMemcpyDeviceToDevice, ctx->st) != hipSuccess) return fail(BPGPU_E_DEVICE);
  CircuitDev cd = circuit_dev(c);
  zpow_table(ctx->st, nb, c->q, dz, 8, (int32_t *)dzp);   // (parametric circuits are rejected above: the prover builds numeric rows)
  prover_polys(ctx->st, cd, nb, dy, dyi, dL, dR, dO, dsL, dsR, (const int32_t *)dzp, s->polys, dwV);
  prover_tcoeffs(ctx->st, nb, n, s->polys, dt);
  if ((rc = launch_ok(ctx))) return fail(rc);
  int bad = 0;
  if ((rc = flag_read(ctx, &bad))) return fail(rc);
  if (bad) return fail(BPGPU_E_ARG);
  if ((rc = d2h(ctx, t_coeffs, dt, nb * 6 * 32)) || (m && (rc = d2h(ctx, wV, dwV, nb * m * 32)))) return fail(rc);
  if (hipStreamSynchronize(ctx->st) != hipSuccess) return fail(BPGPU_E_DEVICE);
  *out = s;
  return BPGPU_OK;
}
int bpgpu_r1cs_prover_polys(bpgpu_ctx *ctx, const bpgpu_circuit *c, size_t nb, const uint8_t *y, const uint8_t *y_inv,
                            const uint8_t *z, const uint8_t *a_L, const uint8_t *a_R, const uint8_t *a_O,
                            const uint8_t *s_L, const uint8_t *s_R, uint8_t *t_coeffs, uint8_t *wV,
                            bpgpu_prover **out) {
  return prover_polys_impl(ctx, c, nb, y, y_inv, z, a_L, a_R, a_O, s_L, s_R, t_coeffs, wV, out, false);
}
int bpgpu_r1cs_prover_polys_ark(bpgpu_ctx *ctx, const bpgpu_circuit *c, size_t nb, const uint8_t *y, const uint8_t *y_inv,
                                const uint8_t *z, const uint8_t *a_L, const uint8_t *a_R, const uint8_t *a_O,
                                const uint8_t *s_L, const uint8_t *s_R, uint8_t *t_coeffs, uint8_t *wV,
                                bpgpu_prover **out) {
  return prover_polys_impl(ctx, c, nb, y, y_inv, z, a_L, a_R, a_O, s_L, s_R, t_coeffs, wV, out, true);
}
int bpgpu_r1cs_prover_eval(bpgpu_ctx *ctx, bpgpu_prover *s, size_t padded_n, const uint8_t *x, uint8_t *l_vec,
                           uint8_t *r_vec) {
  if (!ctx || !s || !x || !l_vec || !r_vec) return BPGPU_E_ARG;
  if (padded_n < s->n || (padded_n & (padded_n - 1))) return BPGPU_E_LEN;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *dx, *dout;
  size_t nb = s->nb;
  CK(ws_get(ctx, 0, nb * 32, &dx));
  CK(ws_get(ctx, 1, 2 * nb * padded_n * 32, &dout));
  Words8 *dl = (Words8 *)dout, *dr = dl + nb * padded_n;
  CK(flag_reset(ctx));
  CK(h2d(ctx, dx, x, nb * 32));
  scalars_check(ctx->st, (Words8 *)dx, nb, ctx->d_flag);
  prover_eval(ctx->st, nb, s->n, padded_n, (Words8 *)dx, s->y, s->polys, dl, dr);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, l_vec, dl, nb * padded_n * 32));
  CK(d2h(ctx, r_vec, dr, nb * padded_n * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
/* prover.rs:659-708 without leaving the device: l(x), r(x) with their padding, the G/H factors and the
 * resident-generator IPP session that consumes them */
int bpgpu_r1cs_prover_ipp_begin(bpgpu_ctx *ctx, bpgpu_prover *ps, const bpgpu_gens *g, size_t padded_n, size_t n1,
                                const uint8_t *x, const uint8_t *u, const uint8_t *y_inv, const uint8_t *w,
                                bpgpu_ipp **out) {
  if (!ctx || !ps || !g || !x || !u || (!y_inv && !ps->yinv) || !w || !out || !ps->polys) return BPGPU_E_ARG;
  if (!padded_n || padded_n < ps->n || (padded_n & (padded_n - 1)) || n1 > ps->n) return BPGPU_E_LEN;
  if (padded_n > g->cap) return BPGPU_E_GENS;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t nb = ps->nb, n = padded_n;
  bpgpu_ipp *s = new (std::nothrow) bpgpu_ipp();
  if (!s) return BPGPU_E_OOM;
  s->nb = nb; s->n0 = s->n = n; s->gens = g;
  if (ctx->shard_world > 1) { shard_bounds(n, ctx->shard_rank, ctx->shard_world, &s->slo, &s->shi); s->with_q = ctx->shard_rank == 0; }
  size_t tot = nb * n, half = nb * (n > 1 ? n / 2 : 1);
  bool okk = true;
  auto M = [&](void **p, size_t bytes) { if (okk && !pool_alloc(ctx, p, bytes)) okk = false; };
  M((void **)&s->a[0], tot * 32); M((void **)&s->b[0], tot * 32); M((void **)&s->a[1], half * 32); M((void **)&s->b[1], half * 32);
  M((void **)&s->cG, tot * 32); M((void **)&s->cH, tot * 32); M((void **)&s->w, nb * 32);
  M((void **)&s->cLR, nb * 2 * 32); M((void **)&s->uu, nb * 2 * 32);
  M((void **)&s->sums, nb * 2 * sizeof(JacRaw)); M((void **)&s->out_xy, nb * 2 * 64);
  M((void **)&s->msc, nb * 2 * (2 + 2 * n) * 32);
  if (!okk) { ipp_free_all(ctx, s); return BPGPU_E_OOM; }
  int rc = BPGPU_OK;
  do {
    void *din;
    if ((rc = ws_get(ctx, 0, 3 * nb * 32, &din))) break;
    Words8 *dx = (Words8 *)din, *du = dx + nb, *dyi = du + nb;
    if ((rc = flag_reset(ctx))) break;
    if ((rc = h2d(ctx, dx, x, nb * 32)) || (rc = h2d(ctx, du, u, nb * 32)) || (rc = h2d(ctx, s->w, w, nb * 32))) break;
    if (y_inv) { if ((rc = h2d(ctx, dyi, y_inv, nb * 32))) break; }
    else if (hipMemcpyAsync(dyi, ps->yinv, nb * 32, hipMemcpyDeviceToDevice, ctx->st) != hipSuccess) { rc = BPGPU_E_DEVICE; break; }
    ProfSpan span(ctx, 19, ctx->st);
    scalars_check(ctx->st, dx, 3 * nb, ctx->d_flag);
    scalars_check(ctx->st, s->w, nb, ctx->d_flag);
    prover_eval(ctx->st, nb, ps->n, n, dx, ps->y, ps->polys, s->a[0], s->b[0]);
    ipp_r1cs_factors(ctx->st, nb, n, n1, du, dyi, s->cG, s->cH);
    span.close();
    if ((rc = launch_ok(ctx))) break;
    int bad = 0;
    if ((rc = flag_read(ctx, &bad))) break;
    if (bad) rc = BPGPU_E_ARG;
  } while (0);
  if (rc) { ipp_free_all(ctx, s); return rc; }
  *out = s;
  return BPGPU_OK;
}
void bpgpu_prover_destroy(bpgpu_ctx *ctx, bpgpu_prover *s) {
  if (!s || !ctx) return;        // (a session belongs to the context it was opened on)
  std::lock_guard<std::mutex> lk(ctx->mu);
  hipStreamSynchronize(ctx->st);
  prover_free_all(ctx, s);
}

/* ---- resident-witness prover sessions: prover.rs:457-494 / :519-565 (phase commitments) and :587-619 (polynomials) with the
 * witness uploaded ONCE and the blinding vectors optionally drawn on the device ------------------------------------------ */
int bpgpu_r1cs_prover_commit(bpgpu_ctx *ctx, const bpgpu_gens *g, bpgpu_prover **session, size_t nb, size_t n_new,
                             const uint8_t *a_L, const uint8_t *a_R, const uint8_t *a_O, const uint8_t *s_L, const uint8_t *s_R,
                             const uint8_t *vector_keys, const uint8_t *blindings, uint8_t *commitments) {
  if (!ctx || !g || !session || !nb || !blindings || !commitments) return BPGPU_E_ARG;
  if (n_new && (!a_L || !a_R || !a_O)) return BPGPU_E_ARG;
  const bool explicit_vec = s_L && s_R;
  if (n_new && (explicit_vec == (vector_keys != nullptr) || (!s_L) != (!s_R))) return BPGPU_E_ARG;   // exactly one source of s_L, s_R
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  bpgpu_prover *s = *session;
  const bool fresh = s == nullptr;
  if (!fresh && s->nb != nb) return BPGPU_E_LEN;
  const size_t wn0 = fresh ? 0 : s->wn, wn = wn0 + n_new;
  if (wn > g->cap) return BPGPU_E_GENS;
  if (fresh) {
    s = new (std::nothrow) bpgpu_prover();
    if (!s) return BPGPU_E_OOM;
    s->nb = nb;
  }
  auto fail = [&](int rc) { if (fresh) prover_free_all(ctx, s); return rc; };
  // new planes of nb x wn; the multipliers of the earlier phase are carried over
  Words8 *pl[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  Words8 **old[5] = {&s->aL, &s->aR, &s->aO, &s->sL, &s->sR};
  const size_t tot_new = nb * n_new;
  void *din = nullptr, *drows = nullptr, *dres = nullptr, *dout = nullptr;
  int rc;
  if ((rc = ws_get(ctx, 0, (5 * tot_new + 4 * nb) * 32, &din)) || (rc = ws_get(ctx, 1, nb * 3 * (2 + 2 * wn) * 32, &drows)) ||
      (rc = ws_get(ctx, 4, nb * 3 * sizeof(JacRaw), &dres)) || (rc = ws_get(ctx, 5, nb * 3 * 64, &dout)))
    return fail(rc);
  if (n_new) {
    bool okk = true;
    for (int i = 0; i < 5; i++) okk = okk && pool_alloc(ctx, (void **)&pl[i], nb * wn * 32);
    if (!okk) { for (auto p : pl) pool_release(ctx, p); return fail(BPGPU_E_OOM); }
    if (wn0) for (int i = 0; i < 5; i++)
      if (hipMemcpy2DAsync(pl[i], wn * 32, *old[i], wn0 * 32, wn0 * 32, nb, hipMemcpyDeviceToDevice, ctx->st) != hipSuccess) {
        for (auto p : pl) pool_release(ctx, p);
        return fail(BPGPU_E_DEVICE);
      }
    for (int i = 0; i < 5; i++) { pool_release(ctx, *old[i]); *old[i] = pl[i]; }   // (stream-ordered: any reuse runs after the copies)
    s->wn = wn;
  }
  // staging: [a_L | a_R | a_O | (s_L | s_R)] nb x n_new each, then blindings nb x 3 (all ark), then the keys nb x 32 bytes (raw)
  Words8 *w = (Words8 *)din;
  const size_t nvec = explicit_vec ? 5 : 3;
  Words8 *dbl = w + nvec * tot_new, *dkeys = dbl + 3 * nb;
  const uint8_t *src[5] = {a_L, a_R, a_O, s_L, s_R};
  if ((rc = flag_reset(ctx))) return fail(rc);
  for (size_t i = 0; i < nvec && n_new; i++) if ((rc = h2d(ctx, w + i * tot_new, src[i], tot_new * 32))) return fail(rc);
  if ((rc = h2d(ctx, dbl, blindings, nb * 3 * 32))) return fail(rc);
  if (n_new && !explicit_vec && (rc = h2d(ctx, dkeys, vector_keys, nb * 32))) return fail(rc);
  ProfSpan span(ctx, 16, ctx->st);
  if (n_new) {
    if (wn0 == 0) {          // contiguous planes: convert straight into them
      for (size_t i = 0; i < nvec; i++) scalars_from_ark(ctx->st, w + i * tot_new, *old[i], tot_new, ctx->d_flag);
    } else {
      scalars_from_ark(ctx->st, w, w, nvec * tot_new, ctx->d_flag);
      for (size_t i = 0; i < nvec; i++)
        if (hipMemcpy2DAsync(*old[i] + wn0, wn * 32, w + i * tot_new, n_new * 32, n_new * 32, nb, hipMemcpyDeviceToDevice, ctx->st) != hipSuccess)
          return fail(BPGPU_E_DEVICE);
    }
    if (!explicit_vec) blind_vectors(ctx->st, dkeys, nb, n_new, s->sL, s->sR, wn, wn0);
  }
  scalars_from_ark(ctx->st, dbl, dbl, 3 * nb, ctx->d_flag);
  size_t slo = 0, shi = wn;
  if (ctx->shard_world > 1) shard_bounds(wn, ctx->shard_rank, ctx->shard_world, &slo, &shi);   // this rank's generators: partial commitments
  commit_rows(ctx->st, nb, wn, wn0, wn, s->aL, s->aR, s->aO, s->sL, s->sR, dbl, (Words8 *)drows, slo, shi, ctx->shard_rank == 0);
  // (three classes of rows per prover -- A_I, A_O: bit vectors; S: dense -- so that a wave of the MSM-per-lane walk holds one class)
  if ((rc = msm_gens_dev(ctx, g, nb * 3, wn, (const uint32_t *)drows, (JacRaw *)dres, ctx->st, 12, 0, 3))) return fail(rc);
  jac_to_boundary(ctx->st, (JacRaw *)dres, (Words8 *)dout, nb * 3);
  span.close();
  if ((rc = launch_ok(ctx))) return fail(rc);
  int bad = 0;
  if ((rc = flag_read(ctx, &bad))) return fail(rc);
  if (bad) return fail(BPGPU_E_ARG);
  if ((rc = d2h(ctx, commitments, dout, nb * 3 * 64))) return fail(rc);
  if (hipStreamSynchronize(ctx->st) != hipSuccess) return fail(BPGPU_E_DEVICE);
  *session = s;
  return BPGPU_OK;
}
static int prover_session_polys_locked(bpgpu_ctx *ctx, bpgpu_prover *s, const bpgpu_circuit *c, const uint8_t *y, const uint8_t *z,
                                       const uint8_t *chi, uint8_t *t_coeffs, uint8_t *wV);
int bpgpu_r1cs_prover_session_polys(bpgpu_ctx *ctx, bpgpu_prover *s, const bpgpu_circuit *c, const uint8_t *y, const uint8_t *z,
                                    uint8_t *t_coeffs, uint8_t *wV) {
  if (!ctx || !s || !c || !y || !z || !t_coeffs || (c->m && !wV)) return BPGPU_E_ARG;
  if (c->nchi) return BPGPU_E_ARG;
  return prover_session_polys_locked(ctx, s, c, y, z, nullptr, t_coeffs, wV);
}
int bpgpu_r1cs_prover_session_polys_param(bpgpu_ctx *ctx, bpgpu_prover *s, const bpgpu_circuit *c, const uint8_t *y, const uint8_t *z,
                                          const uint8_t *gadget_challenges, uint8_t *t_coeffs, uint8_t *wV) {
  if (!ctx || !s || !c || !y || !z || !t_coeffs || (c->m && !wV)) return BPGPU_E_ARG;
  if (!c->nchi || !gadget_challenges) return BPGPU_E_ARG;
  return prover_session_polys_locked(ctx, s, c, y, z, gadget_challenges, t_coeffs, wV);
}
static int prover_session_polys_locked(bpgpu_ctx *ctx, bpgpu_prover *s, const bpgpu_circuit *c, const uint8_t *y, const uint8_t *z,
                                       const uint8_t *chi, uint8_t *t_coeffs, uint8_t *wV) {
  if (c->n != s->wn || s->polys) return BPGPU_E_LEN;     // the circuit's multipliers are the session's; one polynomial build per session
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  const size_t nb = s->nb, n = c->n, m = c->m;
  if (!pool_alloc(ctx, (void **)&s->polys, (6 * nb * (n ? n : 1) * 9) * 4) || !pool_alloc(ctx, (void **)&s->y, nb * 32) ||
      !pool_alloc(ctx, (void **)&s->yinv, nb * 32)) {
    pool_release(ctx, s->polys); pool_release(ctx, s->y); pool_release(ctx, s->yinv);
    s->polys = nullptr; s->y = s->yinv = nullptr;
    return BPGPU_E_OOM;
  }
  s->n = n; s->m = m;
  void *din, *dzp, *dout, *dchi = nullptr;
  const size_t qz = (1 + c->nchi) * (c->q ? c->q : 1);    // a parametric circuit's z-power table carries one block per gadget challenge
  CK(ws_get(ctx, 0, 2 * nb * 32, &din));
  CK(ws_get(ctx, 6, nb * qz * 9 * 4, &dzp));
  CK(ws_get(ctx, 1, (nb * 6 + nb * m) * 32, &dout));
  if (c->nchi) CK(ws_get(ctx, 21, nb * c->nchi * 32, &dchi));
  Words8 *dz = (Words8 *)din, *dt = (Words8 *)dout, *dwV = dt + nb * 6;
  CK(flag_reset(ctx));
  CK(h2d(ctx, s->y, y, nb * 32));
  CK(h2d(ctx, dz, z, nb * 32));
  if (c->nchi) { CK(h2d(ctx, dchi, chi, nb * c->nchi * 32)); scalars_check(ctx->st, (const Words8 *)dchi, nb * c->nchi, ctx->d_flag); }
  scalars_check(ctx->st, s->y, nb, ctx->d_flag);
  scalars_check(ctx->st, dz, nb, ctx->d_flag);
  ProfSpan span(ctx, 17, ctx->st);
  HIPCK(ctx, hipMemcpyAsync(s->yinv, s->y, nb * 32, hipMemcpyDeviceToDevice, ctx->st));
  batch_inverse(ctx->st, s->yinv, nb, ctx->d_flag);       // y^-1, prover.rs:593 (a zero challenge raises the flag: E_ARG)
  zpow_table(ctx->st, nb, c->q, dz, 8, (int32_t *)dzp, c->nchi, (const Words8 *)dchi);
  prover_polys(ctx->st, circuit_dev(c), nb, s->y, s->yinv, s->aL, s->aR, s->aO, s->sL, s->sR, (const int32_t *)dzp, s->polys, dwV);
  prover_tcoeffs(ctx->st, nb, n, s->polys, dt);
  span.close();
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, t_coeffs, dt, nb * 6 * 32));
  if (m) CK(d2h(ctx, wV, dwV, nb * m * 32));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}
/* scalars[i] * (curve generator): GeneratorsChain::next (generators.rs:112-124), Q = w * B (prover.rs:687) */
int bpgpu_generator_mul(bpgpu_ctx *ctx, const uint8_t *scalars, size_t n, uint8_t *out) {
  if (!ctx || (n && (!scalars || !out))) return BPGPU_E_ARG;
  if (!n) return BPGPU_OK;
  static const uint8_t GEN[64] = {0xca,0xcf,0x43,0xc9,0x8b,0x3d,0x72,0x3d,0xe0,0x19,0x18,0x0d,0x9b,0xfd,0xac,0xde,0xc7,0xf0,0x40,0x5a,0x41,0xed,0xec,0x7b,0x1b,0x97,0x99,0x85,0xc1,0x15,0xef,0x01,
                                  0x1f,0xdc,0xe8,0x36,0x0c,0x00,0x73,0x28,0xa3,0x43,0xbe,0x1a,0xd1,0xec,0x53,0xde,0x62,0xec,0x46,0xdf,0x01,0x48,0xbe,0xb7,0x30,0x97,0xa4,0x0a,0x06,0x68,0x56,0x00};
  bool have;
  { std::lock_guard<std::mutex> lk(ctx->mu); have = ctx->gen_tab != nullptr; }
  if (!have) {   // fixed-base table of the generator: 16 windows x 2^15 multiples (34 MB), built once per context
    bpgpu_gens *t = nullptr;
    int rc = bpgpu_gens_create(ctx, nullptr, nullptr, 0, GEN, GEN, 16, &t);   // takes the context lock itself
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->gen_tab) ctx->gen_tab = t;
    else { hipFree(t->points); hipFree(t->table); delete t; }   // another thread won the race
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCK(ctx, hipSetDevice(ctx->device));
  void *dsc, *dres, *dout;
  CK(ws_get(ctx, 0, n * 32, &dsc));
  CK(ws_get(ctx, 3, n * sizeof(JacRaw), &dres));
  CK(ws_get(ctx, 5, n * 64, &dout));
  CK(flag_reset(ctx));
  CK(h2d(ctx, dsc, scalars, n * 32));
  scalars_check(ctx->st, (Words8 *)dsc, n, ctx->d_flag);
  fixed_single16(ctx->st, ctx->gen_tab->table, (const uint32_t *)dsc, (JacRaw *)dres, n);   // 16 table additions per scalar
  jac_to_boundary(ctx->st, (JacRaw *)dres, (Words8 *)dout, n);
  CK(launch_ok(ctx));
  int bad = 0;
  CK(flag_read(ctx, &bad));
  if (bad) return BPGPU_E_ARG;
  CK(d2h(ctx, out, dout, n * 64));
  HIPCK(ctx, hipStreamSynchronize(ctx->st));
  return BPGPU_OK;
}

#pragma GCC visibility pop
}  // extern "C"


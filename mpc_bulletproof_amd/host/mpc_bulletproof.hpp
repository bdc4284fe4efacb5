// mpc_bulletproof.hpp -- host-side mirror of the reference crate's API for the accelerated path
// (renegade-fi/mpc-bulletproof src/lib.rs:30-49): Transcript, PedersenGens, BulletproofGens,
// InnerProductProof, inner_product, r1cs::{Prover, Verifier, R1CSProof, Variable, LinearCombination}.
// Same names, argument meaning and error behaviour; every group / vector operation goes through the
// C ABI of include/bpgpu.h to the MI355X (there is no CPU arithmetic path for points here).
// The Rust toolchain is absent from the build image, so this mirror is C++ (the reference is
// compiled code); INTEGRATION.md shows the Rust-side binding of the same C ABI.
#pragma once
#include <algorithm>
#include <array>
#include <cstring>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/bpgpu.h"

#pragma GCC visibility push(default)
namespace mpc_bulletproof {

// ---- errors (src/errors.rs:13-55,150-177) -------------------------------------------------------
enum class ProofError { VerificationError, FormatError, WrongNumBlindingFactors, InvalidBitsize, InvalidAggregation,
                        InvalidGeneratorsLength };
enum class R1CSError { InvalidGeneratorsLength, FormatError, VerificationError, MissingAssignment, GadgetError };
struct ProofException : std::runtime_error { ProofError e; explicit ProofException(ProofError e_) : std::runtime_error("ProofError"), e(e_) {} };
struct R1CSException : std::runtime_error { R1CSError e; explicit R1CSException(R1CSError e_) : std::runtime_error("R1CSError"), e(e_) {} };
struct DeviceException : std::runtime_error { int code; DeviceException(int c, const std::string &w) : std::runtime_error(w), code(c) {} };

// ---- Scalar: host glue arithmetic in F_n (mpc_stark::algebra::scalar::Scalar) -------------------
// Only what the host-side orchestration needs (transcript reduction, witness synthesis in
// ConstraintSystem::multiply, Poly6::eval, blinding combinations).  Vector work is on the device.
class Scalar {
 public:
  Scalar() : v_{0, 0, 0, 0} {}
  static Scalar zero() { return Scalar(); }
  static Scalar one();
  static Scalar from(uint64_t x);
  static Scalar from_bytes_le(const uint8_t b[32]);        // canonical; throws ProofException(FormatError)
  static Scalar from_le_bytes_mod_order_wide(const uint8_t b[64]);
  void to_bytes_le(uint8_t out[32]) const;
  // the limbs as they lie in memory (= ark-ff's Fp256<MontBackend<_, 4>>: x * 2^256 mod n, little endian): what the
  // *_ark entry points of the C ABI take -- no de-Montgomery on the host
  void to_ark_le(uint8_t out[32]) const { memcpy(out, v_, 32); }
  void to_bytes_be(uint8_t out[32]) const;                 // Scalar::to_bytes_be (proof wire format)
  static Scalar from_be_bytes_mod_order(const uint8_t b[32]);
  std::array<uint8_t, 32> to_bytes() const { std::array<uint8_t, 32> o; to_bytes_le(o.data()); return o; }
  Scalar operator+(const Scalar &o) const;
  Scalar operator-(const Scalar &o) const;
  Scalar operator*(const Scalar &o) const;
  Scalar operator-() const;
  Scalar &operator+=(const Scalar &o) { return *this = *this + o; }
  Scalar &operator-=(const Scalar &o) { return *this = *this - o; }
  Scalar &operator*=(const Scalar &o) { return *this = *this * o; }
  bool operator==(const Scalar &o) const { return v_[0] == o.v_[0] && v_[1] == o.v_[1] && v_[2] == o.v_[2] && v_[3] == o.v_[3]; }
  bool operator!=(const Scalar &o) const { return !(*this == o); }
  Scalar inverse() const;     // Fermat on the host (single values; batches go to bpgpu_batch_inverse)
 private:
  uint64_t v_[4];             // Montgomery form, R = 2^256
};

// Randomness of the prover (blinding factors).  The reference builds it from the transcript, re-keyed with the
// witness blindings and finalized with thread_rng() (src/r1cs/prover.rs:435-445): Prover::prove re-keys whatever Rng
// it is given the same way (rekey() with the transcript state and every v_blinding) before the first draw.
//   OsRng      -- the default and the only one for production use: a Keccak-f[1600] sponge keyed with 32 bytes from
//                 getrandom(2) (/dev/urandom as the fallback), 136 output bytes per permutation; forward-secure: the rate
//                 is zeroed before every permutation, so a captured state does not reveal earlier output blocks;
//                 rekey() absorbs caller material into the state.
//   SeededRng  -- TEST / BENCH ONLY: the splitmix64 stream of a 64-bit seed, so that proofs can be replayed and
//                 compared byte for byte with the CPU oracle; rekey() is a no-op.  Proofs made with it are NOT
//                 zero-knowledge (64 bits of non-cryptographic state).
class Rng {
 public:
  virtual ~Rng() = default;
  virtual uint64_t next_u64() = 0;
  virtual void rekey(const uint8_t *material, size_t len) = 0;
  // true: Prover::prove does not draw the blinding VECTORS s_L, s_R (prover.rs:461-462, 526-527) scalar by scalar from this
  // stream; it draws one 32-byte key per phase in their place and the device expands it ("BlindVec v1", include/bpgpu.h
  // bpgpu_r1cs_prover_commit): no host-side draws, nothing to upload.  OsRng's default.
  virtual bool vector_keys() const { return false; }
  Scalar scalar();            // 4 x u64 little-endian limbs (+ 4 zero limbs) reduced mod n
  // count scalars, the same stream as count calls of scalar(): the words are drawn in order, the reductions mod n run on the
  // thread pool (one prover drawing 2 x 32 766 blinding factors spent 6 ms in them)
  void scalars(Scalar *out, size_t count);
};
class OsRng final : public Rng {
 public:
  explicit OsRng(bool vector_keys = true);   // throws std::runtime_error if the OS gives no entropy
  // the same generator from 32 caller-supplied bytes: the ranks of a sharded proof must draw IDENTICAL blinding factors, so
  // rank 0's OS entropy is handed to all of them (Prover::prove with a RankGroup does this)
  OsRng(const uint8_t key[32], bool vector_keys);
  uint64_t next_u64() override;
  void rekey(const uint8_t *material, size_t len) override;
  bool vector_keys() const override { return vk_; }
 private:
  bool vk_;
  void refill();
  uint64_t st_[25];           // Keccak-f[1600] sponge state; words 0..16 are the rate
  uint64_t blocks_ = 0;
  int used_ = 17;
};
class SeededRng final : public Rng {
 public:
  explicit SeededRng(uint64_t seed, bool vector_keys = false) : s_(seed), vk_(vector_keys) {}
  uint64_t next_u64() override;
  void rekey(const uint8_t *, size_t) override {}
  bool vector_keys() const override { return vk_; }   // (the oracle's vector-key mode replays such proofs)
 private:
  uint64_t s_;
  bool vk_;
};

// Host-side data parallelism (the reference uses rayon): f(i) for i in [0, n) on up to BPH_THREADS threads
// (default min(16, hardware threads)); exceptions are rethrown on the caller's thread.  Serial when n < min_n.
void parallel_for(size_t n, const std::function<void(size_t)> &f, size_t min_n = 2);

// ---- StarkPoint: canonical affine bytes; arithmetic happens on the device ------------------------
struct StarkPoint {
  std::array<uint8_t, 64> xy{};   // x || y little-endian; zeros = identity (src/util.rs:274-289)
  static StarkPoint identity() { return StarkPoint(); }
  static StarkPoint generator();
  bool is_identity() const { for (auto c : xy) if (c) return false; return true; }
  bool operator==(const StarkPoint &o) const { return xy == o.xy; }
  bool operator!=(const StarkPoint &o) const { return !(*this == o); }
};

// ---- RankGroup: the ranks (one process per GPU) that split ONE large proof -- SURVEY 8e.2 -----------
// The path's only exchange is an all-gather of a few partial points per call (64 bytes each): RCCL over xGMI on a GPU node
// (torch.distributed "nccl" behind mpc_bulletproof_amd/sharding.py's callback), gloo in the CPU-rendezvous tests.
class RankGroup {
 public:
  virtual ~RankGroup() {}
  virtual size_t rank() const = 0;
  virtual size_t size() const = 0;
  // every rank contributes `bytes` bytes; `out` receives size() x bytes in rank order, on every rank
  virtual void all_gather(const uint8_t *mine, size_t bytes, uint8_t *out) = 0;
};

// ---- Device: owns the bpgpu context --------------------------------------------------------------
class Device {
 public:
  explicit Device(int index = 0);
  ~Device();
  Device(const Device &) = delete;
  Device &operator=(const Device &) = delete;
  bpgpu_ctx *ctx() const { return ctx_; }
  int index() const { return index_; }   // the GPU this context lives on
  static Device &default_device();   // lazily created; device 0 unless set_default_index was called first
  // one process per GPU: a rank selects its GPU (its local rank) before the first use of the default device
  static void set_default_index(int index);
  void check(int rc, const char *what) const;
  // StarkPoint::msm_iter / msm
  StarkPoint msm(const std::vector<Scalar> &scalars, const std::vector<StarkPoint> &points) const;
 private:
  bpgpu_ctx *ctx_ = nullptr;
  int index_ = 0;
};

// ---- Transcript: stand-in for merlin::HashChainTranscript + TranscriptProtocol (src/transcript.rs) -
void keccak256(const uint8_t *in, size_t len, uint8_t out[32]);
Scalar hash_to_scalar(const uint8_t low[32]);                  // src/util.rs:252-267
class Transcript {
 public:
  explicit Transcript(const std::string &label);
  void append_message(const std::string &label, const uint8_t *msg, size_t len);
  void append_u64(const std::string &label, uint64_t x);
  void challenge_bytes(const std::string &label, uint8_t out[32]);
  // TranscriptProtocol (src/transcript.rs:25-57)
  void innerproduct_domain_sep(uint64_t n);
  void r1cs_domain_sep();
  void r1cs_1phase_domain_sep();
  void r1cs_2phase_domain_sep();
  void append_scalar(const std::string &label, const Scalar &s);
  void append_point(const std::string &label, const StarkPoint &p);
  void validate_and_append_point(const std::string &label, const StarkPoint &p);   // throws ProofException(VerificationError)
  Scalar challenge_scalar(const std::string &label);
  const uint8_t *state() const { return state_; }   // the 32-byte hash-chain state (input of the device transcript)
  void set_state(const uint8_t s[32]) { memcpy(state_, s, 32); }   // resume after a device-side stretch of the chain
 private:
  uint8_t state_[32];
};

// ---- generators (src/generators.rs) ----------------------------------------------------------------
struct PedersenGens {
  StarkPoint B, B_blinding;
  PedersenGens();                                              // default(): both = curve generator (:61-70)
  StarkPoint commit(const Scalar &value, const Scalar &blinding) const;   // :41-43
  // many commitments in one device call (fixed-base tables of (B, B_blinding) from `bp_gens`)
  std::vector<StarkPoint> commit_batch(const class BulletproofGens &bp_gens, const std::vector<Scalar> &values,
                                       const std::vector<Scalar> &blindings) const;
};
class BulletproofGens {
 public:
  BulletproofGens(size_t gens_capacity, size_t party_capacity);   // :182-191
  ~BulletproofGens();
  void increase_capacity(size_t new_capacity);                    // :210-235
  size_t gens_capacity = 0, party_capacity = 0;
  struct Share {                                                   // BulletproofGensShare :302-320
    const BulletproofGens *gens; size_t share;
    std::vector<StarkPoint> G(size_t n) const;
    std::vector<StarkPoint> H(size_t n) const;
  };
  Share share(size_t j) const { return Share{this, j}; }
  // resident fixed-base tables of share 0 for (B, B_blinding) -- built on first use
  // window_bits 0 = by capacity (BPH_WINDOW_BITS overrides): 16-bit windows up to 1024 generators per side (4.5 GB at 64,
  // 69 GB at 1024: 16 table additions per term instead of 19 with 14-bit windows -- the prover's IPP rounds are 2 n-term
  // table-lookup MSMs each), 12-bit up to 4096 (24 GB), 8-bit beyond (17 GB at 32768)
  bpgpu_gens *device_tables(const PedersenGens &pc, int window_bits = 0) const;
 private:
  std::vector<std::vector<StarkPoint>> G_vec_, H_vec_;
  mutable bpgpu_gens *tables_ = nullptr;
  mutable size_t tables_cap_ = 0;
  mutable std::array<uint8_t, 128> tables_pc_{};
};

// ---- inner product proof (src/inner_product_proof.rs) ----------------------------------------------
Scalar inner_product(const std::vector<Scalar> &a, const std::vector<Scalar> &b);   // :463-472 (device)
struct InnerProductProof {
  std::vector<StarkPoint> L_vec, R_vec;
  Scalar a, b;
  // :49-193 -- panics (std::invalid_argument) on length mismatch / non power of two like the reference asserts
  static InnerProductProof create(Transcript &transcript, const StarkPoint &Q, const std::vector<Scalar> &G_factors,
                                  const std::vector<Scalar> &H_factors, std::vector<StarkPoint> G_vec,
                                  std::vector<StarkPoint> H_vec, std::vector<Scalar> a_vec, std::vector<Scalar> b_vec);
  struct VerificationScalars { std::vector<Scalar> u_sq, u_inv_sq, s; };
  // :254-310 -- throws ProofException(VerificationError)
  VerificationScalars verification_scalars(size_t n, Transcript &transcript, std::vector<Scalar> *challenges = nullptr) const;
  // :317-372 -- throws ProofException(VerificationError)
  void verify(size_t n, Transcript &transcript, const std::vector<Scalar> &G_factors, const std::vector<Scalar> &H_factors,
              const StarkPoint &P, const StarkPoint &Q, const std::vector<StarkPoint> &G, const std::vector<StarkPoint> &H) const;
  bool operator==(const InnerProductProof &o) const { return L_vec == o.L_vec && R_vec == o.R_vec && a == o.a && b == o.b; }
  // wire format (:379-455): L_0 R_0 .. L_{k-1} R_{k-1} as 32-byte compressed points, then a, b big-endian
  size_t serialized_size() const { return L_vec.size() * 2 * 32 + 64; }
  std::vector<uint8_t> to_bytes() const;
  static InnerProductProof from_bytes(const uint8_t *b, size_t len);   // throws ProofException(FormatError)
};
// StarkPoint::to_bytes / from_bytes for many points in one device call (bpgpu_points_compress / _decompress)
std::vector<uint8_t> compress_points(const std::vector<StarkPoint> &pts);
std::vector<StarkPoint> decompress_points(const uint8_t *b, size_t n);   // throws ProofException(FormatError)

namespace util {
std::vector<Scalar> exp_iter(const Scalar &x, size_t n);          // src/util.rs:73-76
Scalar sum_of_powers(const Scalar &x, size_t n);                  // src/util.rs:218-234
Scalar sum_of_powers_slow(const Scalar &x, size_t n);             // src/util.rs:237-239
}  // namespace util

// ---- r1cs (src/r1cs/*) ------------------------------------------------------------------------------
namespace r1cs {

struct Variable {                                                  // linear_combination.rs:15-28
  enum Kind : uint32_t { MultiplierLeft = 0, MultiplierRight = 1, MultiplierOutput = 2, Committed = 3, One = 4, Zero = 5 };
  Kind kind; size_t index;
  bool operator<(const Variable &o) const { return kind != o.kind ? kind < o.kind : index < o.index; }
  static Variable one() { return Variable{One, 0}; }
};
// terms of a linear combination, ordered by variable (the reference's HashMap).  Up to three terms live INSIDE the object --
// almost every row of a gadget is `var`, `a + b - 1` or `(x - z) - mul_left`, and one heap allocation per row is 2 x 10^6
// allocations (and as many frees) when 256 provers build 2064 constraints each --; longer rows (the range gadget's final
// sum) spill into a sorted vector.
class TermMap {
 public:
  typedef std::pair<Variable, Scalar> value_type;
  typedef value_type *iterator;
  typedef const value_type *const_iterator;
  // (the inline slots are raw storage: a fresh TermMap -- gadgets make and move about a dozen per multiplier -- does not zero 144
  // bytes it is about to overwrite; value_type is trivially copyable and trivially destructible)
  TermMap() {}
  TermMap(const TermMap &o) : n_(o.n_) { if (o.heap_) heap_.reset(new std::vector<value_type>(*o.heap_)); else copy_inl(o); }
  TermMap(TermMap &&o) noexcept : n_(o.n_), heap_(std::move(o.heap_)) { if (!heap_) copy_inl(o); o.n_ = 0; }
  TermMap &operator=(const TermMap &o) { if (this != &o) { TermMap t(o); *this = std::move(t); } return *this; }
  TermMap &operator=(TermMap &&o) noexcept {
    n_ = o.n_; heap_ = std::move(o.heap_);
    if (!heap_) copy_inl(o);
    o.n_ = 0;
    return *this;
  }
  iterator begin() { return heap_ ? heap_->data() : inl(); }
  iterator end() { return begin() + n_; }
  const_iterator begin() const { return heap_ ? heap_->data() : inl(); }
  const_iterator end() const { return begin() + n_; }
  size_t size() const { return n_; }
  iterator find(const Variable &k) { auto it = lower(k); return it != end() && !(k < it->first) ? it : end(); }
  Scalar &operator[](const Variable &k) {
    iterator it = lower(k);
    if (it != end() && !(k < it->first)) return it->second;
    const size_t pos = (size_t)(it - begin());
    if (!heap_ && n_ < INLINE) {
      value_type *a = inl();
      if (n_ > pos) memmove((void *)(a + pos + 1), (const void *)(a + pos), (n_ - pos) * sizeof(value_type));
      new (a + pos) value_type(k, Scalar());
      n_++;
      return a[pos].second;
    }
    if (!heap_) { heap_.reset(new std::vector<value_type>(inl(), inl() + n_)); heap_->reserve(2 * INLINE + 2); }
    heap_->insert(heap_->begin() + (long)pos, value_type(k, Scalar()));
    n_++;
    return (*heap_)[pos].second;
  }
 private:
  static constexpr size_t INLINE = 3;
  iterator lower(const Variable &k) {
    if (n_ > 16) return std::lower_bound(begin(), end(), k, [](const value_type &a, const Variable &b) { return a.first < b; });
    iterator it = begin(), e = end();
    while (it != e && it->first < k) ++it;      // a handful of terms per row: linear beats binary
    return it;
  }
  static_assert(std::is_trivially_copyable<Variable>::value && std::is_trivially_copyable<Scalar>::value &&
                std::is_trivially_destructible<value_type>::value, "inline terms are moved as bytes");
  value_type *inl() { return reinterpret_cast<value_type *>(raw_); }
  const value_type *inl() const { return reinterpret_cast<const value_type *>(raw_); }
  void copy_inl(const TermMap &o) { if (o.n_) memcpy((void *)raw_, (const void *)o.raw_, o.n_ * sizeof(value_type)); }
  alignas(value_type) unsigned char raw_[INLINE * sizeof(value_type)];
  size_t n_ = 0;
  std::unique_ptr<std::vector<value_type>> heap_;
};
class LinearCombination {                                          // linear_combination.rs:118-121
 public:
  LinearCombination() {}
  LinearCombination(const Variable &v) { terms[v] = Scalar::one(); }
  LinearCombination(const Scalar &s) { terms[Variable::one()] = s; }
  void add_term(const Variable &v, const Scalar &c);               // :129-135
  LinearCombination operator+(const LinearCombination &o) const;
  LinearCombination operator-(const LinearCombination &o) const;
  LinearCombination operator-() const;
  LinearCombination operator*(const Scalar &s) const;
  TermMap terms;
};
inline LinearCombination operator*(const Variable &v, const Scalar &s) { LinearCombination l; l.terms[v] = s; return l; }

struct R1CSProof {                                                 // proof.rs:35-67
  StarkPoint A_I1, A_O1, S1, A_I2, A_O2, S2, T_1, T_3, T_4, T_5, T_6;
  Scalar t_x, t_x_blinding, e_blinding;
  InnerProductProof ipp_proof;
  // "flat v0" byte layout shared with the oracle (NOT the reference wire format, whose 32-byte point
  // compression lives in the absent mpc-stark crate -- proof.rs:82-109)
  std::vector<uint8_t> to_flat_bytes() const;
  static R1CSProof from_flat_bytes(const std::vector<uint8_t> &b);   // throws R1CSException(FormatError)
  // the reference wire format (proof.rs:82-207): version byte, 8 or 11 compressed points, 3 scalars, the IPP
  size_t serialized_size() const;                                     // proof.rs:111-119
  std::vector<uint8_t> to_bytes() const;                              // proof.rs:82-109
  static R1CSProof from_bytes(const uint8_t *b, size_t len);          // proof.rs:128-207; throws R1CSException(FormatError)
};

class RandomizedConstraintSystem;
// constraint_system.rs:55-208
class ConstraintSystem {
 public:
  virtual ~ConstraintSystem() {}
  virtual Transcript &transcript() = 0;
  virtual size_t num_constraints() const = 0;
  virtual size_t num_multipliers() const = 0;
  virtual std::array<Variable, 3> multiply(LinearCombination left, LinearCombination right) = 0;
  virtual Variable allocate(const Scalar *assignment) = 0;        // throws R1CSException(MissingAssignment)
  virtual std::array<Variable, 3> allocate_multiplier(const std::pair<Scalar, Scalar> *input_assignments) = 0;
  virtual Variable commit_public(const Scalar &value) = 0;
  virtual void constrain(LinearCombination lc) = 0;
  virtual Scalar eval(const LinearCombination &lc) const = 0;
  using Callback = std::function<void(RandomizedConstraintSystem &)>;
  virtual void specify_randomized_constraints(Callback cb) = 0;
};
class RandomizedConstraintSystem : public ConstraintSystem {
 public:
  virtual Scalar challenge_scalar(const std::string &label) = 0;
};

class CsCore;   // shared implementation
class Verifier;
// ONE device circuit for EVERY proof of a gadget whose randomized (second-phase) constraints are AFFINE in the challenge scalars
// it draws -- coefficient of a term = c0 + sum_j chi_j c_j, e.g. the shuffle gadget's (x_i - z) (tests/r1cs.rs:23-62).  Without
// it the verifier re-executes the gadget for every proof (verifier.rs:366-385: the 2^14-shuffle's 65 533 constraint rows, half of
// a verification's wall clock) and uploads the rows again; with it the rows cross the ABI ONCE per circuit shape
// (bpgpu_circuit_create_param) and a verification is the transcript replay plus one bpgpu_r1cs_verify_batch_param call.
class ParametricCircuit {
 public:
  // `gadgets(verifier, vars)` adds the circuit's gadgets to a scratch verifier that holds `m` committed variables (vars, in
  // commitment order).  It is run nchi + 2 times with the gadget challenges replaced by probe values (all zero; one unit vector
  // per challenge; a random point, on which the affine model is CHECKED term by term: a gadget that is not affine in its
  // challenges -- or whose shape depends on them -- throws std::invalid_argument).  At most 8 challenges (bpgpu.h).
  ParametricCircuit(size_t m, const std::function<void(Verifier &, const std::vector<Variable> &)> &gadgets);
  ~ParametricCircuit();
  ParametricCircuit(const ParametricCircuit &) = delete;
  ParametricCircuit &operator=(const ParametricCircuit &) = delete;
  size_t n1() const { return n1_; }                       // first-phase multipliers
  size_t n() const { return n_; }                         // all multipliers
  size_t m() const { return m_; }
  size_t num_constraints() const { return q_; }
  const std::vector<std::string> &challenge_labels() const { return labels_; }
  bpgpu_circuit *device_circuit() const { return circ_; }
 private:
  size_t n1_ = 0, n_ = 0, m_ = 0, q_ = 0;
  std::vector<std::string> labels_;
  bpgpu_circuit *circ_ = nullptr;
};
class Prover : public RandomizedConstraintSystem {
 public:
  Prover(const PedersenGens &pc_gens, Transcript &transcript);     // prover.rs:285-300
  ~Prover();
  std::pair<StarkPoint, Variable> commit(const Scalar &v, const Scalar &v_blinding);   // :319-329
  // same, with the commitment V = commit(v, v_blinding) already computed (PedersenGens::commit_batch)
  Variable commit_precomputed(const Scalar &v, const Scalar &v_blinding, const StarkPoint &V);
  // Bind this prover to a ParametricCircuit (before any gadget is added): the gadgets then run for their WITNESS only -- constraint
  // rows are neither stored, hashed nor uploaded, the circuit's device copy serves the proof (bpgpu_r1cs_prover_session_polys_param
  // with the gadget challenges this prover's transcript produced).  prove() checks that the gadgets produced the circuit's shape.
  // The proof is byte for byte the one the unbound prover makes.
  void use_circuit(const ParametricCircuit &circuit);
  R1CSProof prove(const BulletproofGens &bp_gens);                 // :412-727, blinding factors from a fresh OsRng
  R1CSProof prove(const BulletproofGens &bp_gens, Rng &rng);       // the same with the randomness injected (tests: SeededRng)
  // the same for nb provers of one circuit in lock-step (every device call batched over the provers).  `device`: the context
  // the batch runs on (default: the process-wide one).  A caller that streams batches gives each of its worker threads a
  // Device of its own: while one thread waits for its batch's kernels the other builds, packs and hashes the next batch, and
  // the kernels of the two overlap on the GPU (tests/host/capi.cpp bph_range_prove_stream; bench.py r1cs_prove).
  static std::vector<R1CSProof> prove_batch(std::vector<Prover *> &provers, const BulletproofGens &bp_gens,
                                            std::vector<Rng *> &rngs, Device *device = nullptr, RankGroup *group = nullptr);
  // ONE large proof split over the ranks of `group` (BASELINE configs[3]): every rank holds the same prover (same
  // witness, same transcript) and calls this; each computes the multi-scalar multiplications over its share of the generators,
  // the partial points are all-gathered and added.  Every rank returns the same proof -- byte for byte the single-GPU one.
  // Without an Rng the blinding factors come from rank 0's OS entropy, handed to all ranks.
  R1CSProof prove(const BulletproofGens &bp_gens, RankGroup &group, Rng *rng = nullptr, Device *device = nullptr);
  bool constraints_satisfied() const;                              // :405-409
  Transcript &transcript() override;
  size_t num_constraints() const override;
  size_t num_multipliers() const override;
  std::array<Variable, 3> multiply(LinearCombination left, LinearCombination right) override;
  Variable allocate(const Scalar *assignment) override;
  std::array<Variable, 3> allocate_multiplier(const std::pair<Scalar, Scalar> *input_assignments) override;
  Variable commit_public(const Scalar &value) override;
  void constrain(LinearCombination lc) override;
  Scalar eval(const LinearCombination &lc) const override;
  void specify_randomized_constraints(Callback cb) override;
  Scalar challenge_scalar(const std::string &label) override;
 private:
  std::unique_ptr<CsCore> c_;
};
class Verifier : public RandomizedConstraintSystem {
 public:
  Verifier(const PedersenGens &pc_gens, Transcript &transcript);   // verifier.rs:270-282
  ~Verifier();
  Variable commit(const StarkPoint &commitment);                   // :298-306
  void verify(const R1CSProof &proof, const BulletproofGens &bp_gens);   // :393-554; throws R1CSException
  // the same with the mega_check's 13 + m + 2n + 2 lg n terms split over the ranks of `group` (every rank gets the verdict)
  void verify(const R1CSProof &proof, const BulletproofGens &bp_gens, RankGroup &group, Device *device = nullptr);
  // Verifier::verify against a ParametricCircuit: the caller makes the commitments (commit(), in the circuit's order) and does NOT
  // add the gadgets -- the replay draws the gadget challenges under the circuit's labels, right after the phase separator, as the
  // gadgets' own closures would (verifier.rs:366-385), and the device takes the weights c0 + sum_j chi_j c_j.  Same verdicts.
  void verify(const R1CSProof &proof, const BulletproofGens &bp_gens, const ParametricCircuit &circuit);
  void verify(const R1CSProof &proof, const BulletproofGens &bp_gens, const ParametricCircuit &circuit, RankGroup &group, Device *device = nullptr);
  // The host half of verify(): the transcript replay (verifier.rs:398-455,506; inner_product_proof.rs:259-278) and
  // the operand layout of bpgpu_r1cs_verify_batch -- what a service that batches many proofs of one circuit
  // collects per proof.  Throws like verify() on identity points / bad lengths.
  struct BatchInputs {
    size_t n1 = 0, n = 0, padded_n = 0, m = 0, k = 0;
    std::vector<uint8_t> points;       // (11 + m + 2k) x 64
    std::vector<uint8_t> scalars;      // 5 x 32: t_x t_x_blinding e_blinding a b
    std::vector<uint8_t> challenges;   // (6 + k) x 32: y z u x w r u_1..u_k
  };
  BatchInputs transcript_replay(const R1CSProof &proof, const BulletproofGens &bp_gens);
  // the same for a parametric circuit; `gadget_challenges` receives the nchi x 32 bytes of bpgpu_r1cs_verify_batch_param
  BatchInputs transcript_replay(const R1CSProof &proof, const BulletproofGens &bp_gens, const ParametricCircuit &circuit,
                                std::vector<uint8_t> &gadget_challenges);
  // the constraint rows as the CSR arrays of bpgpu_circuit_create (valid after transcript_replay / verify)
  void circuit_csr(std::vector<uint32_t> &row_ptr, std::vector<uint32_t> &kind, std::vector<uint32_t> &idx,
                   std::vector<uint8_t> &coeff) const;
  // the mega_check point of the last verify() (identity <=> accepted); parity hook
  StarkPoint last_mega_check() const;
  Transcript &transcript() override;
  size_t num_constraints() const override;
  size_t num_multipliers() const override;
  std::array<Variable, 3> multiply(LinearCombination left, LinearCombination right) override;
  Variable allocate(const Scalar *assignment) override;
  std::array<Variable, 3> allocate_multiplier(const std::pair<Scalar, Scalar> *input_assignments) override;
  Variable commit_public(const Scalar &value) override;
  void constrain(LinearCombination lc) override;
  Scalar eval(const LinearCombination &lc) const override;
  void specify_randomized_constraints(Callback cb) override;
  Scalar challenge_scalar(const std::string &label) override;
 private:
  friend class ParametricCircuit;
  BatchInputs replay(const R1CSProof &proof, const BulletproofGens &bp_gens, const ParametricCircuit *pc, std::vector<uint8_t> *chi);
  std::unique_ptr<CsCore> c_;
};

}  // namespace r1cs
}  // namespace mpc_bulletproof
#pragma GCC visibility pop

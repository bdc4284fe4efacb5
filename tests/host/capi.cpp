// capi.cpp -- flat entry points of the host mirror (libbphost.so) with the same shape as the CPU
// oracle's flat API, so tests compare proof bytes / accept results one to one.
// Return codes: 0 Ok, -1 VerificationError, -2 InvalidGeneratorsLength, -3 malformed/FormatError,
// -4 MissingAssignment, -10 device failure (no CPU fallback).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "gadgets.hpp"

using namespace mpc_bulletproof;
using namespace mpc_bulletproof::r1cs;

static std::vector<StarkPoint> unpack_points_pub(const uint8_t *b, size_t n) {
  std::vector<StarkPoint> o(n);
  for (size_t i = 0; i < n; i++) memcpy(o[i].xy.data(), b + 64 * i, 64);
  return o;
}

enum { K_RANGE = 0, K_SHUFFLE = 1, K_EXAMPLE = 2, K_DUMMY = 3, K_RANGE_MULTI = 4 };

// `seed` of the proving entry points: BPH_SEED_OS_ENTROPY (all ones) = blinding factors from OsRng (getrandom(2)-keyed DRBG,
// what a deployment must use); any other value = the replayable SeededRng stream -- TEST / BENCH ONLY, such proofs are
// not zero-knowledge (the parity tests and bench.py need proofs they can replay against the CPU oracle).
// The blinding VECTORS (s_L, s_R): OsRng hands Prover::prove one key per phase and the device expands it (Rng::vector_keys;
// BPH_HOST_VECTORS=1 keeps the scalar-by-scalar host draws, for A/B runs); the SeededRng does the same after
// bph_set_seeded_vector_keys(1), which the oracle's vector-key mode replays.
static const uint64_t BPH_SEED_OS_ENTROPY = ~(uint64_t)0;
static bool g_seeded_vector_keys = false;
static std::unique_ptr<Rng> make_rng(uint64_t seed) {
  if (seed == BPH_SEED_OS_ENTROPY) return std::unique_ptr<Rng>(new OsRng(getenv("BPH_HOST_VECTORS") == nullptr));
  return std::unique_ptr<Rng>(new SeededRng(seed, g_seeded_vector_keys));
}

// BulletproofGens are created once and reused by a real caller (generators.rs:182); the flat batch entry points
// keep one instance per capacity so that repeated calls do not rebuild generators and device tables
static const BulletproofGens &cached_gens(size_t capacity) {
  static std::mutex mu;
  // never destroyed: at process exit the HIP runtime may already be gone when static destructors run
  static auto *cache = new std::map<size_t, BulletproofGens *>();
  std::lock_guard<std::mutex> lk(mu);
  BulletproofGens *&slot = (*cache)[capacity];
  if (!slot) slot = new BulletproofGens(capacity, 1);
  return *slot;
}

static int map_error(const R1CSException &e) {
  switch (e.e) {
    case R1CSError::VerificationError: return -1;
    case R1CSError::InvalidGeneratorsLength: return -2;
    case R1CSError::FormatError: return -3;
    case R1CSError::MissingAssignment: return -4;
    default: return -5;
  }
}
static Transcript start_transcript(int kind, size_t param, const uint8_t *label, size_t len) {
  Transcript t(std::string((const char *)label, len));
  if (kind == K_SHUFFLE) {   // tests/r1cs.rs:80-81
    t.append_message("dom-sep", (const uint8_t *)"ShuffleProof", 12);
    t.append_u64("k", param);
  }
  return t;
}
#define GUARD(...)                                    \
  try { __VA_ARGS__ } catch (const R1CSException &e) { return map_error(e); } \
  catch (const ProofException &) { return -1; }       \
  catch (const DeviceException &) { return -10; }     \
  catch (const std::exception &) { return -3; }

extern "C" {
#pragma GCC visibility push(default)

void bph_set_seeded_vector_keys(int on) { g_seeded_vector_keys = on != 0; }
// one process per GPU: the rank's device index, before anything else touches the default device
void bph_set_device(int index) { Device::set_default_index(index); }

int bph_r1cs_prove(int kind, size_t param, const uint8_t *label, size_t label_len, const uint64_t *values,
                   size_t nvalues, uint64_t seed, size_t gens_capacity, uint8_t *proof_out, size_t *proof_len,
                   uint8_t *commitments_out, size_t *m_out) {
  GUARD({
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);   // kept alive as a caller would (generators + tables)
    Transcript transcript = start_transcript(kind, param, label, label_len);
    Prover prover(pc_gens, transcript);
    std::unique_ptr<Rng> rng_owner = make_rng(seed);
    Rng &rng = *rng_owner;
    std::vector<StarkPoint> commitments;
    std::vector<Variable> vars;
    auto commit = [&](uint64_t v) {
      auto cv = prover.commit(Scalar::from(v), rng.scalar());
      commitments.push_back(cv.first);
      vars.push_back(cv.second);
    };
    if (kind == K_RANGE) {
      if (nvalues != 1) return -3;
      commit(values[0]);
      gadgets::range_proof(prover, LinearCombination(vars[0]), &values[0], param);
    } else if (kind == K_RANGE_MULTI) {   // several values range-proved in ONE constraint system: param = n_bits | nvals << 16
      const size_t nbits = param & 0xffff, nv = param >> 16;
      if (nvalues != nv) return -3;
      for (size_t i = 0; i < nv; i++) {
        commit(values[i]);
        gadgets::range_proof(prover, LinearCombination(vars[i]), &values[i], nbits);
      }
    } else if (kind == K_SHUFFLE) {
      if (nvalues != 2 * param) return -3;
      for (size_t i = 0; i < 2 * param; i++) commit(values[i]);
      gadgets::shuffle_gadget(prover, std::vector<Variable>(vars.begin(), vars.begin() + param),
                              std::vector<Variable>(vars.begin() + param, vars.end()));
    } else if (kind == K_EXAMPLE) {
      if (nvalues != 6) return -3;
      for (int i = 0; i < 5; i++) commit(values[i]);
      gadgets::example_gadget(prover, vars[0], vars[1], vars[2], vars[3], vars[4], LinearCombination(Scalar::from(values[5])));
    } else if (kind == K_DUMMY) {
      Scalar val = rng.scalar();
      auto cv = prover.commit(val, Scalar::one());   // commit_public, prover.rs:171-173
      commitments.push_back(cv.first);
      gadgets::dummy_circuit(prover, cv.second, param);
    } else return -3;
    R1CSProof proof = prover.prove(bp_gens, rng);
    auto bytes = proof.to_flat_bytes();
    memcpy(proof_out, bytes.data(), bytes.size());
    *proof_len = bytes.size();
    for (size_t i = 0; i < commitments.size(); i++) memcpy(commitments_out + 64 * i, commitments[i].xy.data(), 64);
    *m_out = commitments.size();
    return 0;
  })
}

int bph_r1cs_verify(int kind, size_t param, const uint8_t *label, size_t label_len, const uint64_t *values,
                    size_t nvalues, const uint8_t *commitments, size_t m, const uint8_t *proof, size_t proof_len,
                    size_t gens_capacity, uint8_t mega_out[64]) {
  GUARD({
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);   // kept alive as a caller would (generators + tables)
    Transcript transcript = start_transcript(kind, param, label, label_len);
    Verifier verifier(pc_gens, transcript);
    std::vector<Variable> vars;
    for (size_t i = 0; i < m; i++) {
      StarkPoint V;
      memcpy(V.xy.data(), commitments + 64 * i, 64);
      vars.push_back(verifier.commit(V));
    }
    if (kind == K_RANGE && m == 1) gadgets::range_proof(verifier, LinearCombination(vars[0]), nullptr, param);
    else if (kind == K_RANGE_MULTI && m == (param >> 16)) { for (size_t i = 0; i < m; i++) gadgets::range_proof(verifier, LinearCombination(vars[i]), nullptr, param & 0xffff); }
    else if (kind == K_SHUFFLE && m == 2 * param)
      gadgets::shuffle_gadget(verifier, std::vector<Variable>(vars.begin(), vars.begin() + param),
                              std::vector<Variable>(vars.begin() + param, vars.end()));
    else if (kind == K_EXAMPLE && m == 5 && nvalues == 1)
      gadgets::example_gadget(verifier, vars[0], vars[1], vars[2], vars[3], vars[4], LinearCombination(Scalar::from(values[0])));
    else if (kind == K_DUMMY && m == 1) gadgets::dummy_circuit(verifier, vars[0], param);
    else return -3;
    R1CSProof pr = R1CSProof::from_flat_bytes(std::vector<uint8_t>(proof, proof + proof_len));
    if (mega_out) memset(mega_out, 0xff, 64);
    try {
      verifier.verify(pr, bp_gens);
    } catch (const R1CSException &e) {
      if (mega_out) memcpy(mega_out, verifier.last_mega_check().xy.data(), 64);
      return map_error(e);
    }
    if (mega_out) memcpy(mega_out, verifier.last_mega_check().xy.data(), 64);
    return 0;
  })
}

// nb provers, each proving `nvals` values of `n_bits` bits in ONE constraint system (BASELINE config 3 shape:
// 16 x 64-bit -> n = 1024 multipliers, q = 2064, m = 16), all in lock-step.  values: nb x nvals.
// proofs_out: nb x proof_len (returned), commitments_out: nb x nvals x 64.
int bph_range_prove_batch(size_t nb, size_t nvals, size_t n_bits, const uint8_t *label, size_t label_len,
                          const uint64_t *values, uint64_t seed0, size_t gens_capacity, uint8_t *proofs_out,
                          size_t *proof_len, uint8_t *commitments_out) {
  GUARD({
    const bool timing = getenv("BPH_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      auto t = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[bph] %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
      T0 = t;
    };
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);
    lap("generators");
    std::vector<std::unique_ptr<Transcript>> trs;
    std::vector<std::unique_ptr<Prover>> provers;
    std::vector<std::unique_ptr<Rng>> rngs;
    std::vector<Prover *> pp;
    std::vector<Rng *> rr;
    // all nb * nvals Pedersen commitments in one device call (the blinding factors are drawn first, in
    // the order tests/r1cs.rs:684 draws them)
    std::vector<Scalar> vs, bls;
    vs.resize(nb * nvals); bls.resize(nb * nvals);
    for (size_t p = 0; p < nb; p++) rngs.emplace_back(seed0 == BPH_SEED_OS_ENTROPY ? make_rng(seed0) : make_rng(seed0 + p));
    parallel_for(nb, [&](size_t p) {
      for (size_t j = 0; j < nvals; j++) { vs[p * nvals + j] = Scalar::from(values[p * nvals + j]); bls[p * nvals + j] = rngs[p]->scalar(); }
    });
    auto Vs = pc_gens.commit_batch(bp_gens, vs, bls);
    lap("tables + commitments");
    for (size_t p = 0; p < nb; p++) {
      trs.emplace_back(new Transcript(std::string((const char *)label, label_len)));
      provers.emplace_back(new Prover(pc_gens, *trs.back()));
      pp.push_back(provers.back().get());
      rr.push_back(rngs[p].get());
    }
    parallel_for(nb, [&](size_t p) {   // the provers build their constraint systems independently
      for (size_t j = 0; j < nvals; j++) {
        uint64_t v = values[p * nvals + j];
        size_t ix = p * nvals + j;
        memcpy(commitments_out + ix * 64, Vs[ix].xy.data(), 64);
        Variable var = provers[p]->commit_precomputed(vs[ix], bls[ix], Vs[ix]);
        gadgets::range_proof(*provers[p], LinearCombination(var), &v, n_bits);
      }
    });
    lap("circuit building");
    auto proofs = Prover::prove_batch(pp, bp_gens, rr);
    lap("prove_batch");
    for (size_t p = 0; p < nb; p++) {
      auto bytes = proofs[p].to_flat_bytes();
      *proof_len = bytes.size();
      memcpy(proofs_out + p * bytes.size(), bytes.data(), bytes.size());
    }
    lap("serialise");
    parallel_for(nb, [&](size_t p) { provers[p].reset(); trs[p].reset(); });   // constraint systems: many small allocations
    lap("teardown");
    return 0;
  })
}

// A STREAM of prover batches (bench.py r1cs_prove; BASELINE configs[2] shape): nbatch batches of nb provers, each prover
// range-proving nvals values of n_bits bits in one constraint system, proved by `threads` worker threads that each own a Device
// (context + stream): while one thread waits for its batch's kernels, another builds / packs / hashes the next batch, and the
// kernels of batches in flight overlap on the GPU.  values: nb x nvals (every batch proves the same values under fresh
// blinding factors).  prebuild = 1: the constraint systems and commitments of ALL batches are built first, untimed -- the
// reference's own bench times Prover::prove only (benches/r1cs.rs:36-55, 95-108: "only time proof generation") --, and the
// timed region is the stream of prove_batch calls incl. dropping the provers (Prover::prove consumes self); prebuild = 0: circuit
// building and commitments are inside the timed region too.  proofs_out: nbatch x nb x proof_len; commitments_out: nbatch x nb x
// nvals x 64.  ms_out[0] = wall of the timed region, [1] = sum over batches of building, [2] = of prove_batch, [3] = of teardown.
// profile = 1: HIP-event timing of the device phases on every worker's context (bpgpu_profile_*): ms_out[4] = GPU-busy time = the
// union over all contexts of the phase intervals, [5] / [6] = summed duration / launches of the round MSM (the dominant kernel),
// [7..11] = summed durations of the round loops, phase commitments, polynomial builds, T commitments, IPP set-ups.
int bph_range_prove_stream(size_t nbatch, size_t threads, int prebuild, int profile, size_t nb, size_t nvals, size_t n_bits, const uint8_t *label,
                           size_t label_len, const uint64_t *values, uint64_t seed0, size_t gens_capacity, uint8_t *proofs_out,
                           size_t *proof_len, uint8_t *commitments_out, double ms_out[12]) {
  GUARD({
    if (!nbatch || !threads || !nb) return -3;
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);
    bp_gens.device_tables(pc_gens);                    // built before the clock starts (a caller keeps its generators)
    struct Batch {
      std::vector<std::unique_ptr<Transcript>> trs;
      std::vector<std::unique_ptr<Prover>> provers;
      std::vector<std::unique_ptr<Rng>> rngs;
      double ms_build = 0, ms_prove = 0, ms_drop = 0;
    };
    std::vector<Batch> batches(nbatch);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    auto build = [&](size_t bi, Device &dev) {
      (void)dev;
      auto t0 = now();
      Batch &B = batches[bi];
      std::vector<Scalar> vs(nb * nvals), bls(nb * nvals);
      for (size_t p = 0; p < nb; p++) B.rngs.emplace_back(seed0 == BPH_SEED_OS_ENTROPY ? make_rng(seed0) : make_rng(seed0 + bi * nb + p));
      parallel_for(nb, [&](size_t p) {
        for (size_t j = 0; j < nvals; j++) { vs[p * nvals + j] = Scalar::from(values[p * nvals + j]); bls[p * nvals + j] = B.rngs[p]->scalar(); }
      });
      auto Vs = pc_gens.commit_batch(bp_gens, vs, bls);
      for (size_t p = 0; p < nb; p++) {
        B.trs.emplace_back(new Transcript(std::string((const char *)label, label_len)));
        B.provers.emplace_back(new Prover(pc_gens, *B.trs.back()));
      }
      uint8_t *com = commitments_out + bi * nb * nvals * 64;
      parallel_for(nb, [&](size_t p) {
        for (size_t j = 0; j < nvals; j++) {
          uint64_t v = values[p * nvals + j];
          size_t ix = p * nvals + j;
          memcpy(com + ix * 64, Vs[ix].xy.data(), 64);
          Variable var = B.provers[p]->commit_precomputed(vs[ix], bls[ix], Vs[ix]);
          gadgets::range_proof(*B.provers[p], LinearCombination(var), &v, n_bits);
        }
      });
      B.ms_build = ms(t0, now());
    };
    std::mutex out_mu;
    auto prove = [&](size_t bi, Device &dev) {
      Batch &B = batches[bi];
      auto t0 = now();
      std::vector<Prover *> pp;
      std::vector<Rng *> rr;
      for (size_t p = 0; p < nb; p++) { pp.push_back(B.provers[p].get()); rr.push_back(B.rngs[p].get()); }
      auto proofs = Prover::prove_batch(pp, bp_gens, rr, &dev);
      auto t1 = now();
      size_t plen = 0;
      for (size_t p = 0; p < nb; p++) {
        auto bytes = proofs[p].to_flat_bytes();
        plen = bytes.size();
        memcpy(proofs_out + (bi * nb + p) * plen, bytes.data(), plen);
      }
      { std::lock_guard<std::mutex> lk(out_mu); *proof_len = plen; }
      parallel_for(nb, [&](size_t p) { B.provers[p].reset(); B.trs[p].reset(); });   // Prover::prove consumes the prover
      B.ms_prove = ms(t0, t1);
      B.ms_drop = ms(t1, now());
    };
    std::vector<std::unique_ptr<Device>> devs;
    for (size_t t = 0; t < threads; t++) devs.emplace_back(new Device(0));
    if (prebuild) for (size_t bi = 0; bi < nbatch; bi++) build(bi, *devs[0]);
    void *epoch = nullptr;
    if (profile) {
      epoch = bpgpu_profile_epoch(devs[0]->ctx());
      if (!epoch) return -10;
      for (auto &dv : devs) bpgpu_profile_enable(dv->ctx(), 1);
    }
    std::atomic<size_t> next{0};
    std::vector<std::exception_ptr> errs(threads);
    auto T0 = now();
    std::vector<std::thread> th;
    for (size_t t = 0; t < threads; t++)
      th.emplace_back([&, t] {
        try {
          for (;;) {
            size_t bi = next.fetch_add(1);
            if (bi >= nbatch) break;
            if (!prebuild) build(bi, *devs[t]);
            prove(bi, *devs[t]);
          }
        } catch (...) { errs[t] = std::current_exception(); }
      });
    for (auto &x : th) x.join();
    ms_out[0] = ms(T0, now());
    for (auto &e : errs) if (e) std::rethrow_exception(e);
    ms_out[1] = ms_out[2] = ms_out[3] = 0;
    for (auto &B : batches) { ms_out[1] += B.ms_build; ms_out[2] += B.ms_prove; ms_out[3] += B.ms_drop; }
    for (int i = 4; i < 12; i++) ms_out[i] = 0;
    if (profile) {
      std::vector<std::pair<double, double>> iv;
      const size_t cap = 1 << 16;
      std::vector<int32_t> kind(cap);
      std::vector<double> a(cap), b(cap);
      for (auto &dv : devs) {
        size_t cnt = 0;
        bpgpu_profile_enable(dv->ctx(), 0);
        if (bpgpu_profile_intervals(dv->ctx(), epoch, cap, kind.data(), a.data(), b.data(), &cnt)) return -10;
        for (size_t i = 0; i < cnt; i++) {
          const double dt = b[i] - a[i];
          switch (kind[i]) {
            case 21: ms_out[5] += dt; ms_out[6] += 1; break;          // nested inside kind 20: not part of the union
            case 20: ms_out[7] += dt; iv.emplace_back(a[i], b[i]); break;
            case 16: ms_out[8] += dt; iv.emplace_back(a[i], b[i]); break;
            case 17: ms_out[9] += dt; iv.emplace_back(a[i], b[i]); break;
            case 18: ms_out[10] += dt; iv.emplace_back(a[i], b[i]); break;
            case 19: ms_out[11] += dt; iv.emplace_back(a[i], b[i]); break;
            default: iv.emplace_back(a[i], b[i]); break;
          }
        }
      }
      std::sort(iv.begin(), iv.end());
      double busy = 0, cs_ = 0, ce = -1;
      for (auto &x : iv) {
        if (ce < 0 || x.first > ce) { if (ce >= 0) busy += ce - cs_; cs_ = x.first; ce = x.second; }
        else if (x.second > ce) ce = x.second;
      }
      if (ce >= 0) busy += ce - cs_;
      ms_out[4] = busy;
    }
    return 0;
  })
}

// BASELINE config 4: one k-shuffle proof (tests/r1cs.rs:64-164, benches/shuffle.rs), prove then verify, with the
// phases timed.  values = x_0..x_{k-1} | y_0..y_{k-1}.  ms[0..6] = generators, fixed-base tables + 2k commitments,
// prover circuit + phase-1 commit, prove, verifier circuit, verify.  Returns the verify result code.
int bph_shuffle_prove_verify(size_t k, const uint64_t *values, uint64_t seed, size_t gens_capacity, uint8_t *proof_out,
                             size_t *proof_len, uint8_t *commitments_out, double ms[6]) {
  GUARD({
    auto T0 = std::chrono::steady_clock::now();
    int li = 0;
    auto lap = [&]() {
      auto t = std::chrono::steady_clock::now();
      ms[li++] = std::chrono::duration<double, std::milli>(t - T0).count();
      T0 = t;
    };
    const char *label = "ShuffleProofTest";
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);   // kept alive as a caller would (generators + tables)
    lap();
    std::unique_ptr<Rng> rng_owner = make_rng(seed);
    Rng &rng = *rng_owner;
    std::vector<Scalar> vs, bls;
    for (size_t i = 0; i < 2 * k; i++) { vs.push_back(Scalar::from(values[i])); bls.push_back(rng.scalar()); }
    auto Vs = pc_gens.commit_batch(bp_gens, vs, bls);
    lap();
    R1CSProof proof;
    {
      Transcript transcript = start_transcript(K_SHUFFLE, k, (const uint8_t *)label, strlen(label));
      Prover prover(pc_gens, transcript);
      std::vector<Variable> vars;
      for (size_t i = 0; i < 2 * k; i++) vars.push_back(prover.commit_precomputed(vs[i], bls[i], Vs[i]));
      gadgets::shuffle_gadget(prover, std::vector<Variable>(vars.begin(), vars.begin() + k),
                              std::vector<Variable>(vars.begin() + k, vars.end()));
      lap();
      proof = prover.prove(bp_gens, rng);
      lap();
    }
    auto bytes = proof.to_flat_bytes();
    memcpy(proof_out, bytes.data(), bytes.size());
    *proof_len = bytes.size();
    for (size_t i = 0; i < 2 * k; i++) memcpy(commitments_out + 64 * i, Vs[i].xy.data(), 64);
    Transcript transcript = start_transcript(K_SHUFFLE, k, (const uint8_t *)label, strlen(label));
    Verifier verifier(pc_gens, transcript);
    std::vector<Variable> vars;
    for (size_t i = 0; i < 2 * k; i++) vars.push_back(verifier.commit(Vs[i]));
    gadgets::shuffle_gadget(verifier, std::vector<Variable>(vars.begin(), vars.begin() + k),
                            std::vector<Variable>(vars.begin() + k, vars.end()));
    lap();
    verifier.verify(proof, bp_gens);
    lap();
    return 0;
  })
}

static ParametricCircuit *shuffle_param_circuit(size_t k, double *build_ms) {
  static std::mutex mu;
  static std::map<size_t, std::unique_ptr<ParametricCircuit>> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto &slot = cache[k];
  if (build_ms) *build_ms = 0;
  if (!slot) {
    auto t0 = std::chrono::steady_clock::now();
    slot.reset(new ParametricCircuit(2 * k, [k](Verifier &v, const std::vector<Variable> &vars) {
      gadgets::shuffle_gadget(v, std::vector<Variable>(vars.begin(), vars.begin() + k), std::vector<Variable>(vars.begin() + k, vars.end()));
    }));
    if (build_ms) *build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return slot.get();
}
// Prover::prove of the k-shuffle with the prover BOUND to the ParametricCircuit (Prover::use_circuit: the gadget runs for its witness
// only, no constraint rows are built or uploaded).  Same inputs and outputs as bph_shuffle_prove_verify's prover half; ms_out[0] =
// the commitments (one device call), [1] = gadget / circuit building (commit calls + witness), [2] = Prover::prove.
int bph_shuffle_prove_param(size_t k, const uint64_t *values, uint64_t seed, size_t gens_capacity, uint8_t *proof_out, size_t *proof_len,
                            uint8_t *commitments_out, double ms_out[3]) {
  GUARD({
    const char *label = "ShuffleProofTest";
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);
    ParametricCircuit *pc = shuffle_param_circuit(k, nullptr);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::unique_ptr<Rng> rng_owner = make_rng(seed);
    Rng &rng = *rng_owner;
    auto t0 = now();
    std::vector<Scalar> vs, bls;
    for (size_t i = 0; i < 2 * k; i++) { vs.push_back(Scalar::from(values[i])); bls.push_back(rng.scalar()); }
    auto Vs = pc_gens.commit_batch(bp_gens, vs, bls);
    auto t1 = now();
    Transcript transcript = start_transcript(K_SHUFFLE, k, (const uint8_t *)label, strlen(label));
    Prover prover(pc_gens, transcript);
    prover.use_circuit(*pc);
    std::vector<Variable> vars;
    for (size_t i = 0; i < 2 * k; i++) vars.push_back(prover.commit_precomputed(vs[i], bls[i], Vs[i]));
    gadgets::shuffle_gadget(prover, std::vector<Variable>(vars.begin(), vars.begin() + k), std::vector<Variable>(vars.begin() + k, vars.end()));
    auto t2 = now();
    R1CSProof proof = prover.prove(bp_gens, rng);
    auto t3 = now();
    ms_out[0] = ms(t0, t1); ms_out[1] = ms(t1, t2); ms_out[2] = ms(t2, t3);
    auto bytes = proof.to_flat_bytes();
    memcpy(proof_out, bytes.data(), bytes.size());
    *proof_len = bytes.size();
    for (size_t i = 0; i < 2 * k; i++) memcpy(commitments_out + 64 * i, Vs[i].xy.data(), 64);
    return 0;
  })
}

// Verifier::verify of a k-shuffle proof against a ParametricCircuit (host mirror: one device circuit per shape, cached here per k):
// reps verifications of the same proof; ms_out[0] = building the circuit (0 when it was cached), [1] = the commit calls of one
// verification (the transcript's dependent hash chain), [2] = median of Verifier::verify(proof, gens, circuit).
// Returns 0 when every repetition accepted, else the mapped R1CSError of the first rejection.
int bph_shuffle_verify_param(size_t k, const uint8_t *commitments, const uint8_t *proof_bytes, size_t proof_len, size_t gens_capacity,
                             size_t reps, double ms_out[3]) {
  GUARD({
    const char *label = "ShuffleProofTest";
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    ms_out[0] = ms_out[1] = ms_out[2] = 0;
    ParametricCircuit *pc = shuffle_param_circuit(k, &ms_out[0]);
    R1CSProof proof = R1CSProof::from_flat_bytes(std::vector<uint8_t>(proof_bytes, proof_bytes + proof_len));
    std::vector<StarkPoint> Vs(2 * k);
    for (size_t i = 0; i < 2 * k; i++) memcpy(Vs[i].xy.data(), commitments + 64 * i, 64);
    std::vector<double> tv;
    for (size_t rep = 0; rep < (reps ? reps : 1); rep++) {
      Transcript transcript = start_transcript(K_SHUFFLE, k, (const uint8_t *)label, strlen(label));
      Verifier verifier(pc_gens, transcript);
      auto t0 = now();
      for (size_t i = 0; i < 2 * k; i++) verifier.commit(Vs[i]);
      auto t1 = now();
      verifier.verify(proof, bp_gens, *pc);
      auto t2 = now();
      ms_out[1] = ms(t0, t1);
      tv.push_back(ms(t1, t2));
    }
    std::sort(tv.begin(), tv.end());
    ms_out[2] = tv[tv.size() / 2];
    return 0;
  })
}

// The same proof split over the ranks of a group: every rank (process) calls this with the same inputs and its own
// rank; `allgather(mine, bytes, out, user)` must gather `bytes` bytes from every rank into out[world x bytes] in rank order
// (torch.distributed behind a ctypes callback in the tests and in bench.py).  Every rank returns the same proof and verdict.
typedef void (*bph_allgather_fn)(const uint8_t *mine, size_t bytes, uint8_t *out, void *user);
namespace {
struct CallbackGroup final : RankGroup {
  size_t r, w; bph_allgather_fn fn; void *user;
  CallbackGroup(size_t r_, size_t w_, bph_allgather_fn f, void *u) : r(r_), w(w_), fn(f), user(u) {}
  size_t rank() const override { return r; }
  size_t size() const override { return w; }
  void all_gather(const uint8_t *mine, size_t bytes, uint8_t *out) override { fn(mine, bytes, out, user); }
};
}  // namespace
static int shuffle_sharded(size_t k, const uint64_t *values, uint64_t seed, size_t gens_capacity, size_t rank, size_t world,
                           bph_allgather_fn allgather, void *user, uint8_t *proof_out, size_t *proof_len,
                           uint8_t *commitments_out, double ms[6], bool param);
int bph_shuffle_prove_verify_sharded(size_t k, const uint64_t *values, uint64_t seed, size_t gens_capacity, size_t rank, size_t world,
                                     bph_allgather_fn allgather, void *user, uint8_t *proof_out, size_t *proof_len,
                                     uint8_t *commitments_out, double ms[6]) {
  return shuffle_sharded(k, values, seed, gens_capacity, rank, world, allgather, user, proof_out, proof_len, commitments_out, ms, false);
}
// the same with prover and verifier bound to the shuffle's ParametricCircuit (every rank holds its own device copy)
int bph_shuffle_prove_verify_sharded_param(size_t k, const uint64_t *values, uint64_t seed, size_t gens_capacity, size_t rank, size_t world,
                                           bph_allgather_fn allgather, void *user, uint8_t *proof_out, size_t *proof_len,
                                           uint8_t *commitments_out, double ms[6]) {
  return shuffle_sharded(k, values, seed, gens_capacity, rank, world, allgather, user, proof_out, proof_len, commitments_out, ms, true);
}
static int shuffle_sharded(size_t k, const uint64_t *values, uint64_t seed, size_t gens_capacity, size_t rank, size_t world,
                           bph_allgather_fn allgather, void *user, uint8_t *proof_out, size_t *proof_len,
                           uint8_t *commitments_out, double ms[6], bool param) {
  GUARD({
    ParametricCircuit *pc = param ? shuffle_param_circuit(k, nullptr) : nullptr;
    if (!world || rank >= world || !allgather) return -3;
    auto T0 = std::chrono::steady_clock::now();
    int li = 0;
    auto lap = [&]() {
      auto t = std::chrono::steady_clock::now();
      ms[li++] = std::chrono::duration<double, std::milli>(t - T0).count();
      T0 = t;
    };
    CallbackGroup group(rank, world, allgather, user);
    const char *label = "ShuffleProofTest";
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);
    lap();
    // the Pedersen blindings of the 2k values and the prover's blinding factors: the same stream on every rank (a replayable
    // seed, or -- BPH_SEED_OS_ENTROPY -- rank 0's OS entropy handed to all of them)
    std::unique_ptr<Rng> rng_owner;
    if (seed == BPH_SEED_OS_ENTROPY) {
      OsRng src;
      uint8_t mine[32];
      for (int i = 0; i < 4; i++) { uint64_t w64 = src.next_u64(); memcpy(mine + 8 * i, &w64, 8); }
      std::vector<uint8_t> all(32 * world);
      group.all_gather(mine, 32, all.data());
      rng_owner.reset(new OsRng(all.data(), getenv("BPH_HOST_VECTORS") == nullptr));
    } else rng_owner = make_rng(seed);
    Rng &rng = *rng_owner;
    std::vector<Scalar> vs, bls;
    for (size_t i = 0; i < 2 * k; i++) { vs.push_back(Scalar::from(values[i])); bls.push_back(rng.scalar()); }
    auto Vs = pc_gens.commit_batch(bp_gens, vs, bls);
    lap();
    R1CSProof proof;
    {
      Transcript transcript = start_transcript(K_SHUFFLE, k, (const uint8_t *)label, strlen(label));
      Prover prover(pc_gens, transcript);
      if (pc) prover.use_circuit(*pc);
      std::vector<Variable> vars;
      for (size_t i = 0; i < 2 * k; i++) vars.push_back(prover.commit_precomputed(vs[i], bls[i], Vs[i]));
      gadgets::shuffle_gadget(prover, std::vector<Variable>(vars.begin(), vars.begin() + k),
                              std::vector<Variable>(vars.begin() + k, vars.end()));
      lap();
      proof = prover.prove(bp_gens, group, &rng);
      lap();
    }
    auto bytes = proof.to_flat_bytes();
    memcpy(proof_out, bytes.data(), bytes.size());
    *proof_len = bytes.size();
    for (size_t i = 0; i < 2 * k; i++) memcpy(commitments_out + 64 * i, Vs[i].xy.data(), 64);
    Transcript transcript = start_transcript(K_SHUFFLE, k, (const uint8_t *)label, strlen(label));
    Verifier verifier(pc_gens, transcript);
    std::vector<Variable> vars;
    for (size_t i = 0; i < 2 * k; i++) vars.push_back(verifier.commit(Vs[i]));
    if (!pc) gadgets::shuffle_gadget(verifier, std::vector<Variable>(vars.begin(), vars.begin() + k),
                                     std::vector<Variable>(vars.begin() + k, vars.end()));
    lap();
    if (pc) verifier.verify(proof, bp_gens, *pc, group); else verifier.verify(proof, bp_gens, group);
    lap();
    return 0;
  })
}

// Operands of bpgpu_r1cs_verify_batch for nb proofs of the n_bits range gadget (m = 1), produced by the host mirror:
// transcript replay per proof (parallel over proofs), the circuit's CSR rows (buffers sized by the caller:
// row_ptr 2 n_bits + 2, kind/idx 8 n_bits + 8 entries, coeff 32 x that), dims = n1, n, k, m, q, nnz, and the
// 32-byte transcript state after Transcript::new(label) (input of bpgpu_r1cs_verify_batch_fs).
int bph_range_verify_inputs(size_t nb, size_t n_bits, const uint8_t *label, size_t label_len, const uint8_t *commitments,
                            const uint8_t *proofs, size_t proof_len, size_t gens_capacity, uint8_t *points_out,
                            uint8_t *scalars_out, uint8_t *challenges_out, uint8_t init_state_out[32], size_t dims_out[6],
                            uint32_t *row_ptr, uint32_t *kind, uint32_t *idx, uint8_t *coeff) {
  GUARD({
    PedersenGens pc_gens;
    const BulletproofGens &bp_gens = cached_gens(gens_capacity);
    std::vector<Verifier::BatchInputs> ins(nb);
    std::vector<uint32_t> rp, kd, ix;
    std::vector<uint8_t> co;
    parallel_for(nb, [&](size_t p) {
      Transcript transcript(std::string((const char *)label, label_len));
      Verifier verifier(pc_gens, transcript);
      StarkPoint V;
      memcpy(V.xy.data(), commitments + 64 * p, 64);
      Variable var = verifier.commit(V);
      gadgets::range_proof(verifier, LinearCombination(var), nullptr, n_bits);
      R1CSProof pr = R1CSProof::from_flat_bytes(std::vector<uint8_t>(proofs + p * proof_len, proofs + (p + 1) * proof_len));
      ins[p] = verifier.transcript_replay(pr, bp_gens);
      if (p == 0) verifier.circuit_csr(rp, kd, ix, co);
    });
    const auto &a = ins[0];
    for (size_t p = 0; p < nb; p++) {
      if (ins[p].k != a.k || ins[p].n != a.n || ins[p].m != a.m) return -3;
      memcpy(points_out + p * a.points.size(), ins[p].points.data(), a.points.size());
      memcpy(scalars_out + p * a.scalars.size(), ins[p].scalars.data(), a.scalars.size());
      memcpy(challenges_out + p * a.challenges.size(), ins[p].challenges.data(), a.challenges.size());
    }
    Transcript t0(std::string((const char *)label, label_len));
    memcpy(init_state_out, t0.state(), 32);
    dims_out[0] = a.n1; dims_out[1] = a.n; dims_out[2] = a.k; dims_out[3] = a.m; dims_out[4] = rp.size() - 1; dims_out[5] = kd.size();
    memcpy(row_ptr, rp.data(), rp.size() * 4);
    memcpy(kind, kd.data(), kd.size() * 4);
    memcpy(idx, ix.data(), ix.size() * 4);
    memcpy(coeff, co.data(), co.size());
    return 0;
  })
}

// wire codec (proof.rs:82-207): "flat v0" <-> reference wire bytes; out buffers sized by the caller (<= flat size)
int bph_proof_flat_to_wire(const uint8_t *flat, size_t flat_len, uint8_t *wire_out, size_t *wire_len) {
  GUARD({
    R1CSProof p = R1CSProof::from_flat_bytes(std::vector<uint8_t>(flat, flat + flat_len));
    auto w = p.to_bytes();
    if (w.size() != p.serialized_size()) return -5;
    memcpy(wire_out, w.data(), w.size());
    *wire_len = w.size();
    return 0;
  })
}
int bph_proof_wire_to_flat(const uint8_t *wire, size_t wire_len, uint8_t *flat_out, size_t *flat_len) {
  GUARD({
    R1CSProof p = R1CSProof::from_bytes(wire, wire_len);
    auto f = p.to_flat_bytes();
    memcpy(flat_out, f.data(), f.size());
    *flat_len = f.size();
    return 0;
  })
}

int bph_compress_points(const uint8_t *xy, size_t n, uint8_t *out) {
  GUARD({
    auto c = compress_points(unpack_points_pub(xy, n));
    memcpy(out, c.data(), c.size());
    return 0;
  })
}

int bph_generator(uint8_t out[64]) {
  StarkPoint g = StarkPoint::generator();
  memcpy(out, g.xy.data(), 64);
  return 0;
}

int bph_gens(int which, uint32_t party, size_t n, uint8_t *out) {
  GUARD({
    BulletproofGens g(n, party + 1);
    auto pts = which == 'G' ? g.share(party).G(n) : g.share(party).H(n);
    for (size_t i = 0; i < n; i++) memcpy(out + 64 * i, pts[i].xy.data(), 64);
    return 0;
  })
}

int bph_ipp_create(const uint8_t *label, size_t label_len, size_t n, const uint8_t Q[64], const uint8_t *Gf,
                   const uint8_t *Hf, const uint8_t *G, const uint8_t *H, const uint8_t *a, const uint8_t *b,
                   uint8_t *L_out, uint8_t *R_out, uint8_t a_out[32], uint8_t b_out[32]) {
  GUARD({
    auto sv = [&](const uint8_t *p) { std::vector<Scalar> v(n); for (size_t i = 0; i < n; i++) v[i] = Scalar::from_bytes_le(p + 32 * i); return v; };
    auto pv = [&](const uint8_t *p) { std::vector<StarkPoint> v(n); for (size_t i = 0; i < n; i++) memcpy(v[i].xy.data(), p + 64 * i, 64); return v; };
    Transcript t(std::string((const char *)label, label_len));
    StarkPoint q;
    memcpy(q.xy.data(), Q, 64);
    InnerProductProof p = InnerProductProof::create(t, q, sv(Gf), sv(Hf), pv(G), pv(H), sv(a), sv(b));
    for (size_t i = 0; i < p.L_vec.size(); i++) { memcpy(L_out + 64 * i, p.L_vec[i].xy.data(), 64); memcpy(R_out + 64 * i, p.R_vec[i].xy.data(), 64); }
    p.a.to_bytes_le(a_out);
    p.b.to_bytes_le(b_out);
    return 0;
  })
}

int bph_ipp_verify(const uint8_t *label, size_t label_len, size_t n, const uint8_t *Gf, const uint8_t *Hf,
                   const uint8_t P[64], const uint8_t Q[64], const uint8_t *G, const uint8_t *H, const uint8_t *L,
                   const uint8_t *R, size_t k, const uint8_t a[32], const uint8_t b[32]) {
  GUARD({
    auto sv = [&](const uint8_t *p, size_t c) { std::vector<Scalar> v(c); for (size_t i = 0; i < c; i++) v[i] = Scalar::from_bytes_le(p + 32 * i); return v; };
    auto pv = [&](const uint8_t *p, size_t c) { std::vector<StarkPoint> v(c); for (size_t i = 0; i < c; i++) memcpy(v[i].xy.data(), p + 64 * i, 64); return v; };
    InnerProductProof pr;
    pr.L_vec = pv(L, k);
    pr.R_vec = pv(R, k);
    pr.a = Scalar::from_bytes_le(a);
    pr.b = Scalar::from_bytes_le(b);
    Transcript t(std::string((const char *)label, label_len));
    StarkPoint pp, qq;
    memcpy(pp.xy.data(), P, 64);
    memcpy(qq.xy.data(), Q, 64);
    pr.verify(n, t, sv(Gf, n), sv(Hf, n), pp, qq, pv(G, n), pv(H, n));
    return 0;
  })
}

#pragma GCC visibility pop
}

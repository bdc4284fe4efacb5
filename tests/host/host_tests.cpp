// host_tests.cpp -- the reference's own tests restated against the C++ mirror (GPU required).
// tests/r1cs.rs, src/inner_product_proof.rs:474-636, src/util.rs:291-346, src/generators.rs:348-415.
#include <cstdio>
#include <cstdlib>

#include "gadgets.hpp"

using namespace mpc_bulletproof;
using namespace mpc_bulletproof::r1cs;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

// ---- tests/r1cs.rs:69-134 ShuffleProof::prove / verify
static bool kshuffle_helper(size_t k, bool corrupt = false) {
  PedersenGens pc_gens;
  BulletproofGens bp_gens(std::max<size_t>(1, 2 * k) == 1 ? 1 : (size_t)1 << (64 - __builtin_clzll(2 * k - 1)), 1);
  SeededRng rng(1000 + k);
  std::vector<Scalar> input, output;
  for (size_t i = 0; i < k; i++) input.push_back(Scalar::from(rng.next_u64()));
  output.assign(input.rbegin(), input.rend());
  if (corrupt) output[0] = output[0] + Scalar::one();
  std::vector<StarkPoint> ic, oc;
  R1CSProof proof;
  {
    Transcript t("ShuffleProofTest");
    t.append_message("dom-sep", (const uint8_t *)"ShuffleProof", 12);
    t.append_u64("k", k);
    Prover prover(pc_gens, t);
    std::vector<Variable> iv, ov;
    for (auto &v : input) { auto c = prover.commit(v, rng.scalar()); ic.push_back(c.first); iv.push_back(c.second); }
    for (auto &v : output) { auto c = prover.commit(v, rng.scalar()); oc.push_back(c.first); ov.push_back(c.second); }
    gadgets::shuffle_gadget(prover, iv, ov);
    proof = prover.prove(bp_gens, rng);
  }
  Transcript t("ShuffleProofTest");
  t.append_message("dom-sep", (const uint8_t *)"ShuffleProof", 12);
  t.append_u64("k", k);
  Verifier verifier(pc_gens, t);
  std::vector<Variable> iv, ov;
  for (auto &c : ic) iv.push_back(verifier.commit(c));
  for (auto &c : oc) ov.push_back(verifier.commit(c));
  gadgets::shuffle_gadget(verifier, iv, ov);
  try { verifier.verify(proof, bp_gens); } catch (const R1CSException &) { return false; }
  return true;
}

// ---- tests/r1cs.rs:232-340 example gadget round trip (with the serialization variant)
static bool example_gadget_roundtrip(uint64_t a1, uint64_t a2, uint64_t b1, uint64_t b2, uint64_t c1, uint64_t c2, bool serialize) {
  PedersenGens pc_gens;
  BulletproofGens bp_gens(128, 1);
  SeededRng rng(7);
  std::vector<StarkPoint> commitments;
  R1CSProof proof;
  {
    Transcript t("R1CSExampleGadget");
    Prover prover(pc_gens, t);
    std::vector<Variable> vars;
    for (uint64_t x : {a1, a2, b1, b2, c1}) { auto c = prover.commit(Scalar::from(x), rng.scalar()); commitments.push_back(c.first); vars.push_back(c.second); }
    gadgets::example_gadget(prover, vars[0], vars[1], vars[2], vars[3], vars[4], LinearCombination(Scalar::from(c2)));
    proof = prover.prove(bp_gens, rng);
  }
  if (serialize) proof = R1CSProof::from_flat_bytes(proof.to_flat_bytes());
  Transcript t("R1CSExampleGadget");
  Verifier verifier(pc_gens, t);
  std::vector<Variable> vars;
  for (auto &c : commitments) vars.push_back(verifier.commit(c));
  gadgets::example_gadget(verifier, vars[0], vars[1], vars[2], vars[3], vars[4], LinearCombination(Scalar::from(c2)));
  try { verifier.verify(proof, bp_gens); } catch (const R1CSException &) { return false; }
  return true;
}

// ---- tests/r1cs.rs:672-703 range_proof_helper
static bool range_proof_helper(uint64_t v_val, size_t n) {
  PedersenGens pc_gens;
  BulletproofGens bp_gens(128, 1);
  SeededRng rng(99 + n);
  R1CSProof proof;
  StarkPoint commitment;
  {
    Transcript t("RangeProofTest");
    Prover prover(pc_gens, t);
    auto cv = prover.commit(Scalar::from(v_val), rng.scalar());
    commitment = cv.first;
    gadgets::range_proof(prover, LinearCombination(cv.second), &v_val, n);
    proof = prover.prove(bp_gens, rng);
  }
  Transcript t("RangeProofTest");
  Verifier verifier(pc_gens, t);
  Variable var = verifier.commit(commitment);
  gadgets::range_proof(verifier, LinearCombination(var), nullptr, n);
  try { verifier.verify(proof, bp_gens); } catch (const R1CSException &) { return false; }
  return true;
}

// ---- src/inner_product_proof.rs:507-583 test_helper_create
static void ipp_test_helper_create(size_t n) {
  SeededRng rng(5 + n);
  BulletproofGens bp_gens(n, 1);
  auto G = bp_gens.share(0).G(n), H = bp_gens.share(0).H(n);
  StarkPoint Q = Device::default_device().msm({rng.scalar()}, {StarkPoint::generator()});
  std::vector<Scalar> a(n), b(n);
  for (auto &x : a) x = rng.scalar();
  for (auto &x : b) x = rng.scalar();
  Scalar c = inner_product(a, b);
  std::vector<Scalar> G_factors(n, Scalar::one());
  Scalar y_inv = rng.scalar();
  auto H_factors = util::exp_iter(y_inv, n);
  std::vector<Scalar> sc = a;
  for (size_t i = 0; i < n; i++) sc.push_back(b[i] * H_factors[i]);
  sc.push_back(c);
  std::vector<StarkPoint> pts = G;
  pts.insert(pts.end(), H.begin(), H.end());
  pts.push_back(Q);
  StarkPoint P = Device::default_device().msm(sc, pts);
  Transcript tp("innerproducttest");
  auto proof = InnerProductProof::create(tp, Q, G_factors, H_factors, G, H, a, b);
  Transcript tv("innerproducttest");
  bool ok = true;
  try { proof.verify(n, tv, G_factors, H_factors, P, Q, G, H); } catch (const ProofException &) { ok = false; }
  CHECK(ok);
  if (n > 1) {   // a wrong P must be rejected
    Transcript tv2("innerproducttest");
    bool rejected = false;
    try { proof.verify(n, tv2, G_factors, H_factors, Q, Q, G, H); } catch (const ProofException &) { rejected = true; }
    CHECK(rejected);
  }
}

int main() {
  // src/util.rs:295-345
  {
    auto e = util::exp_iter(Scalar::from(2), 4);
    CHECK(e[0] == Scalar::from(1) && e[1] == Scalar::from(2) && e[2] == Scalar::from(4) && e[3] == Scalar::from(8));
    Scalar x = Scalar::from(10);
    uint64_t want[7] = {0, 1, 11, 111, 1111, 11111, 111111};
    for (size_t n = 0; n < 7; n++) CHECK(util::sum_of_powers_slow(x, n) == Scalar::from(want[n]));
    for (size_t n : {0, 1, 2, 4, 8, 16, 32, 64}) CHECK(util::sum_of_powers(x, n) == util::sum_of_powers_slow(x, n));
  }
  // src/inner_product_proof.rs:620-635
  CHECK(inner_product({Scalar::from(1), Scalar::from(2), Scalar::from(3), Scalar::from(4)},
                      {Scalar::from(2), Scalar::from(3), Scalar::from(4), Scalar::from(5)}) == Scalar::from(40));
  // src/generators.rs:381-413 resizing == fresh
  {
    BulletproofGens g(64, 1), g2(32, 1);
    g2.increase_capacity(64);
    CHECK(g.share(0).G(64) == g2.share(0).G(64) && g.share(0).H(64) == g2.share(0).H(64));
  }
  // src/inner_product_proof.rs:595-618
  for (size_t n : {1, 2, 4, 32, 64}) ipp_test_helper_create(n);
  // tests/r1cs.rs:171-214
  for (size_t k : {1, 2, 3, 4, 5, 6, 7, 24, 42}) CHECK(kshuffle_helper(k));
  CHECK(!kshuffle_helper(5, true));
  // tests/r1cs.rs:541-587
  CHECK(example_gadget_roundtrip(3, 4, 6, 1, 40, 9, false));
  CHECK(!example_gadget_roundtrip(3, 4, 6, 1, 40, 10, false));
  CHECK(example_gadget_roundtrip(3, 4, 6, 1, 40, 9, true));
  CHECK(!example_gadget_roundtrip(3, 4, 6, 1, 40, 10, true));
  // tests/r1cs.rs:654-670
  for (size_t n : {2, 10, 32, 63}) {
    uint64_t max = (uint64_t)(((unsigned __int128)1 << n) - 1);
    for (uint64_t v : {(uint64_t)0, max / 3, max}) CHECK(range_proof_helper(v, n));
    CHECK(!range_proof_helper(max + 1, n));
  }
  std::printf(failures ? "host_tests: %d FAILURES\n" : "host_tests: all passed\n", failures);
  return failures ? 1 : 0;
}

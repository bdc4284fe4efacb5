// gadgets.hpp -- the reference's test / bench circuits written against the mirrored API, line for
// line with tests/r1cs.rs and benches/r1cs.rs of renegade-fi/mpc-bulletproof.
#pragma once
#include "mpc_bulletproof.hpp"

namespace mpc_bulletproof {
namespace gadgets {
using namespace r1cs;

// tests/r1cs.rs:23-62  ShuffleProof::gadget
inline void shuffle_gadget(RandomizedConstraintSystem &cs0, std::vector<Variable> x, std::vector<Variable> y) {
  if (x.size() != y.size()) throw std::invalid_argument("shuffle: length mismatch");
  size_t k = x.size();
  if (k == 1) {
    cs0.constrain(LinearCombination(y[0]) - LinearCombination(x[0]));
    return;
  }
  cs0.specify_randomized_constraints([x, y, k](RandomizedConstraintSystem &cs) {
    Scalar z = cs.challenge_scalar("shuffle challenge");
    // Make last x multiplier for i = k-1 and k-2
    auto last_mulx = cs.multiply(LinearCombination(x[k - 1]) - LinearCombination(z), LinearCombination(x[k - 2]) - LinearCombination(z));
    Variable out = last_mulx[2];
    // Make multipliers for x from i == [0, k-3]
    for (size_t i = k - 2; i-- > 0;) out = cs.multiply(LinearCombination(out), LinearCombination(x[i]) - LinearCombination(z))[2];
    Variable first_mulx_out = out;
    auto last_muly = cs.multiply(LinearCombination(y[k - 1]) - LinearCombination(z), LinearCombination(y[k - 2]) - LinearCombination(z));
    out = last_muly[2];
    for (size_t i = k - 2; i-- > 0;) out = cs.multiply(LinearCombination(out), LinearCombination(y[i]) - LinearCombination(z))[2];
    // Constrain last x mul output and last y mul output to be equal
    cs.constrain(LinearCombination(first_mulx_out) - LinearCombination(out));
  });
}

// tests/r1cs.rs:217-228  example_gadget: (a1 + a2) * (b1 + b2) = (c1 + c2)
inline void example_gadget(ConstraintSystem &cs, LinearCombination a1, LinearCombination a2, LinearCombination b1,
                           LinearCombination b2, LinearCombination c1, LinearCombination c2) {
  auto v = cs.multiply(a1 + a2, b1 + b2);
  cs.constrain(c1 + c2 - LinearCombination(v[2]));
}

// tests/r1cs.rs:620-652  range_proof: v in [0, 2^n)
inline void range_proof(ConstraintSystem &cs, LinearCombination v, const uint64_t *v_assignment, size_t n) {
  Scalar exp_2 = Scalar::one();
  for (size_t i = 0; i < n; i++) {
    // Create low-level variables and add them to constraints
    std::pair<Scalar, Scalar> asg;
    if (v_assignment) {
      uint64_t bit = (*v_assignment >> i) & 1;
      asg = {Scalar::from(1 - bit), Scalar::from(bit)};
    }
    auto abo = cs.allocate_multiplier(v_assignment ? &asg : nullptr);
    // Enforce a * b = 0, so one of (a,b) is zero
    cs.constrain(LinearCombination(abo[2]));
    // Enforce that a = 1 - b, so they both are 1 or 0.
    cs.constrain(LinearCombination(abo[0]) + (LinearCombination(abo[1]) - LinearCombination(Scalar::one())));
    // Add `-b_i*2^i` to the linear combination
    v.add_term(abo[1], -exp_2);   // v -= b * exp_2 (SubAssign in place, as the reference)
    exp_2 = exp_2 + exp_2;
  }
  // Enforce that v = Sum(b_i * 2^i, i = 0..n-1)
  cs.constrain(v);
}

// benches/r1cs.rs:24-33  DummyCircuit::apply_constraints with a given public value
inline void dummy_circuit(ConstraintSystem &cs, Variable var, size_t n_constraints) {
  for (size_t i = 0; i < n_constraints; i++) var = cs.multiply(LinearCombination(var), LinearCombination(var))[2];
}

}  // namespace gadgets
}  // namespace mpc_bulletproof

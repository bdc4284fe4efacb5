// pool_test.cpp -- CPU-only unit test of the host mirror's thread pool (no device call): nested parallel_for from the
// calling thread's own slice and from workers, Rng::scalars (a parallel_for above 4096 draws) from inside a loop -- the
// shape of Prover::prove_batch with nb >= 2 provers of >= 4096 multipliers -- and exception propagation.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include "mpc_bulletproof.hpp"

using namespace mpc_bulletproof;

int main() {
  int failures = 0;
  {   // nested loops: every (p, c) pair exactly once
    std::vector<std::atomic<int>> hits(8 * 64);
    for (auto &h : hits) h = 0;
    parallel_for(8, [&](size_t p) { parallel_for(64, [&](size_t c) { hits[p * 64 + c]++; }); });
    for (auto &h : hits) if (h != 1) failures++;
  }
  {   // the prover's shape: nb provers each drawing >= 4096 blinding scalars; the stream must equal the serial one
    const size_t nb = 3, cnt = 5000;
    std::vector<std::vector<Scalar>> got(nb, std::vector<Scalar>(cnt));
    std::vector<SeededRng> rngs;
    for (size_t p = 0; p < nb; p++) rngs.emplace_back(100 + p);
    parallel_for(nb, [&](size_t p) { rngs[p].scalars(got[p].data(), cnt); });
    for (size_t p = 0; p < nb; p++) {
      SeededRng ref(100 + p);
      for (size_t i = 0; i < cnt; i++) if (got[p][i] != ref.scalar()) { failures++; break; }
    }
  }
  {   // an exception thrown in a nested loop reaches the outer caller; the pool stays usable
    bool caught = false;
    try {
      parallel_for(4, [&](size_t p) { parallel_for(16, [&](size_t c) { if (p == 0 && c == 7) throw std::runtime_error("x"); }); });
    } catch (const std::runtime_error &) { caught = true; }
    if (!caught) failures++;
    std::atomic<int> n{0};
    parallel_for(100, [&](size_t) { n++; });
    if (n != 100) failures++;
  }
  std::printf(failures ? "pool_test: %d FAILURES\n" : "pool_test: all passed\n", failures);
  return failures ? 1 : 0;
}

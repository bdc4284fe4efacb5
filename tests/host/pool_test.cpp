// pool_test.cpp -- CPU-only unit test of the host mirror's thread pool (no device call): nested parallel_for from the
// calling thread's own slice and from workers, Rng::scalars (a parallel_for above 4096 draws) from inside a loop -- the
// shape of Prover::prove_batch with nb >= 2 provers of >= 4096 multipliers -- and exception propagation.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <string>
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
  {   // keccak256 known answers (empty, "abc", 1280 bytes, and the three lengths around the 136-byte rate): the test runs once with the
      // AVX-512 permutation (where the CPU has it) and once with BPH_KECCAK_SCALAR=1 -- both must reproduce them
    struct KAT { size_t len; int fill; const char *hex; };
    const KAT kats[] = {{0, 0, "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"},
                        {3, -1, "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"},
                        {1280, -2, "9d1ff092a1e205a727a9d8ff51db9d07399185e733560cf9107ae14b77d3a080"},
                        {135, 0x5a, "03c527855334eb2e62b3b9b4d02ab76721707d3dde5fb218369640ee2edc7f3a"},
                        {136, 0x5a, "ddc757d2caa82320e140f35833c18e8cc3b230b2b9a48def3d98461ffae81716"},
                        {137, 0x5a, "37a14cb79c82d4b7d837a3f8ea134a324b138a5e4e3bb0814b75e52b17975b9c"}};
    for (const KAT &k : kats) {
      std::vector<uint8_t> m(k.len);
      for (size_t i = 0; i < k.len; i++) m[i] = k.fill == -1 ? (uint8_t)"abc"[i] : (k.fill == -2 ? (uint8_t)(i & 255) : (uint8_t)k.fill);
      uint8_t d[32];
      keccak256(m.data(), m.size(), d);
      char hex[65];
      for (int i = 0; i < 32; i++) std::snprintf(hex + 2 * i, 3, "%02x", d[i]);
      if (std::string(hex) != k.hex) { std::printf("keccak256(%zu bytes) = %s\n", k.len, hex); failures++; }
    }
  }
  std::printf(failures ? "pool_test: %d FAILURES\n" : "pool_test: all passed\n", failures);
  return failures ? 1 : 0;
}

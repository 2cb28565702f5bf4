// Host-only harness for the product's tree builders (nbody-simulation_amd/csrc/tree_build.hpp), built by
// tests/test_native_sanitizers.py with -fsanitize=address,undefined and with -fsanitize=thread (GPU sanitizers are
// not available on the pool; the builders are plain C++ and run their subtrees on std::threads).
// Checks: no sanitizer report, and the trees built with 1 thread and with many threads are identical.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../nbody-simulation_amd/csrc/tree_build.hpp"

using namespace nbody;

template <class T> static bool same(const TreeHost<T>& a, const TreeHost<T>& b) {
  if (a.size() != b.size() || a.order != b.order || a.mass_u32 != b.mass_u32 || a.max_depth != b.max_depth) return false;
  return std::memcmp(a.geom0.data(), b.geom0.data(), a.size() * sizeof(typename TreeHost<T>::G4)) == 0 &&
         std::memcmp(a.geom1.data(), b.geom1.data(), a.size() * sizeof(typename TreeHost<T>::G4)) == 0 &&
         std::memcmp(a.link.data(), b.link.data(), a.size() * sizeof(typename TreeHost<T>::L4)) == 0;
}

template <class T> static int run(int n, unsigned seed) {
  std::mt19937 rng(seed);
  std::normal_distribution<double> g(50000.0, 9000.0);
  std::vector<T> pos(2 * (size_t)n);
  std::vector<uint32_t> w((size_t)n);
  for (int i = 0; i < n; ++i) {
    pos[2 * i] = (T)std::min(99999.0, std::max(1.0, g(rng)));
    pos[2 * i + 1] = (T)std::min(99999.0, std::max(1.0, g(rng)));
    w[i] = 1 + (uint32_t)(i % 7);
  }
  TreeHost<T> a, b, qa, qb;
  setenv("NBODY_BUILD_THREADS", "1", 1);
  build_bvh<T>(pos.data(), w.data(), n, 64, a);
  build_quad<T>(pos.data(), w.data(), n, (T)0, (T)0, (T)100000, qa);
  setenv("NBODY_BUILD_THREADS", "8", 1);
  build_bvh<T>(pos.data(), w.data(), n, 64, b);
  build_quad<T>(pos.data(), w.data(), n, (T)0, (T)0, (T)100000, qb);
  if (!same(a, b) || !same(qa, qb)) { std::printf("MISMATCH n=%d\n", n); return 1; }
  if (a.overflow || qa.overflow) { std::printf("unexpected overflow n=%d\n", n); return 1; }
  std::printf("n=%d bvh nodes %zu depth %d | quad nodes %zu depth %d : 1 thread == 8 threads\n", n, a.size(), a.max_depth,
              qa.size(), qa.max_depth);
  return 0;
}

int main() {
  int rc = 0;
  for (int n : {0, 1, 9, 65, 1000, 70000, 300000}) {
    rc |= run<float>(n, 1u + (unsigned)n);
    rc |= run<double>(n, 7u + (unsigned)n);
  }
  // degenerate input must report overflow, not crash or hang
  std::vector<float> same_pt(2 * 100, 5.0f);
  TreeHost<float> t;
  build_bvh<float>(same_pt.data(), nullptr, 100, 64, t);
  if (!t.overflow) { std::printf("bvh: degenerate input not flagged\n"); rc = 1; }
  build_quad<float>(same_pt.data(), nullptr, 100, 0.f, 0.f, 100000.f, t);
  if (!t.overflow) { std::printf("quad: degenerate input not flagged\n"); rc = 1; }
  std::printf(rc ? "FAILED\n" : "OK\n");
  return rc;
}

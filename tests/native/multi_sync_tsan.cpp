// The multi-device context's barrier and worker pool (csrc/multi_sync.hpp) under ThreadSanitizer / ASan, no GPU:
//   1. plain barriers: every rank gets the AND of the votes, generation after generation;
//   2. a rank that fails BEFORE a barrier breaks it: the others are released with `false`, nobody hangs, the next job works again;
//   3. ADVICE r03: a rank that fails AFTER the final barrier (break_all while a slower rank is still waking from that barrier):
//      the slow rank must still see the completed barrier's `true` — it guards a collective the fast rank has already enqueued.
#include <atomic>
#include <chrono>
#include <cstdio>

#include "../../nbody-simulation_amd/csrc/multi_sync.hpp"

using nbody::Barrier;
using nbody::Pool;

static int fails = 0;
#define CHECK(c)                                                  \
  do {                                                            \
    if (!(c)) {                                                   \
      std::printf("MISMATCH %s:%d %s\n", __FILE__, __LINE__, #c); \
      ++fails;                                                    \
    }                                                             \
  } while (0)

int main() {
  const int G = 4;
  Barrier bar;
  bar.n = G;
  Pool pool;
  pool.barrier = &bar;
  pool.start(G);

  // 1. votes
  for (int round = 0; round < 200; ++round) {
    std::atomic<int> yes{0};
    const int dissent = round % (G + 1);  // rank `dissent` votes no (G: nobody)
    int rc = pool.run([&](int d) -> int {
      for (int k = 0; k < 3; ++k) {
        const bool r = bar.arrive(!(k == 1 && d == dissent));
        if (r) ++yes;
      }
      return 0;
    });
    CHECK(rc == 0);
    CHECK(yes.load() == (dissent < G ? 2 * G : 3 * G));
  }

  // 2. a rank that fails before the barrier
  for (int round = 0; round < 200; ++round) {
    std::atomic<int> got_true{0};
    int who = -1;
    int rc = pool.run([&](int d) -> int {
      if (d == round % G) return NBODY_ERR_NOMEM;  // never arrives
      if (bar.arrive(true)) ++got_true;
      return 0;
    }, &who);
    CHECK(rc == NBODY_ERR_NOMEM && who == round % G);
    CHECK(got_true.load() == 0);
    rc = pool.run([&](int) { return bar.arrive(true) ? 0 : NBODY_ERR_INVALID; });  // mended
    CHECK(rc == 0);
  }

  // 3. a rank that fails after the final barrier, while the others are still waking up from it
  int late_false = 0;
  for (int round = 0; round < 3000; ++round) {
    std::atomic<int> got_true{0};
    int rc = pool.run([&](int d) -> int {
      if (d != 0) {  // the slow ranks: asleep inside the barrier when rank 0, arriving last, completes it
        if (bar.arrive(true)) ++got_true;
        return 0;
      }
      while (true) {  // arrive last
        std::unique_lock<std::mutex> lk(bar.m);
        if (bar.waiting == G - 1) break;
        lk.unlock();
        std::this_thread::yield();
      }
      const bool r = bar.arrive(true);
      if (r) ++got_true;
      return NBODY_ERR_HIP;  // ... and fail right away: Pool::loop breaks the barrier while the others still sleep
    });
    CHECK(rc == NBODY_ERR_HIP);
    if (got_true.load() != G) ++late_false;
  }
  CHECK(late_false == 0);
  if (late_false) std::printf("a completed barrier read false on %d of 3000 rounds\n", late_false);

  pool.shutdown();
  std::printf(fails ? "FAILED\n" : "OK\n");
  return fails ? 1 : 0;
}

// Host-side synchronisation of the multi-device context (multi.hip): the barrier the workers meet at before every collective
// and the pool that runs one job per device.  Plain C++ (no HIP), so that the CPU sanitizer tests can drive it
// (tests/native/multi_sync_tsan.cpp).  Internal to the library.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/nbody_hip.h"

namespace nbody {

// All workers meet here before a collective: either every rank joins it or none does (a rank that failed to enqueue
// its kernels must not leave the others waiting inside RCCL).
// A worker that leaves its job with an error (or an exception) BREAKS the barrier: whoever waits, or arrives later in
// the same job, gets `false` at once instead of waiting for a rank that will never come; Pool::run mends it before
// the next job.
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int n = 1, waiting = 0;
  uint64_t gen = 0;
  bool acc = true, result = true, broken = false;
  bool arrive(bool ok) {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    acc = acc && ok;
    if (++waiting == n) {
      result = acc;
      acc = true;
      waiting = 0;
      ++gen;
      cv.notify_all();
      return result;
    }
    const uint64_t g = gen;
    cv.wait(lk, [&] { return gen != g || broken; });
    // The generation decides, not the flag: a barrier that COMPLETED keeps its result for every rank that took part, even when a
    // faster rank has failed and broken the barrier since (it may already have enqueued the collective this barrier guards: a slow
    // rank that now read `false` would skip it, and the collective would lack a member — ADVICE r03).  `result` is still this
    // generation's: the next one cannot complete before this waiter has returned.  Only a waiter whose generation never completed
    // is released with `false`.
    return gen != g ? result : false;
  }
  void break_all() {
    std::unique_lock<std::mutex> lk(m);
    broken = true;
    cv.notify_all();
  }
  void mend() {
    std::unique_lock<std::mutex> lk(m);
    broken = false;
    acc = true;
    waiting = 0;
  }
};

struct Pool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::function<int(int)> job;
  uint64_t gen = 0;
  int pending = 0;
  std::vector<int> rc;
  bool stop = false;
  Barrier* barrier = nullptr;  // broken by a worker that fails, mended before every job
  void start(int n) {
    rc.assign((size_t)n, 0);
    for (int d = 0; d < n; ++d) th.emplace_back([this, d] { loop(d); });
  }
  void loop(int d) {
    uint64_t seen = 0;
    for (;;) {
      std::function<int(int)> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen;
        f = job;
      }
      int r;
      try {
        r = f(d);
      } catch (...) {  // nothing unwinds out of a worker (std::bad_alloc of a host-side builder, say)
        r = NBODY_ERR_NOMEM;
      }
      if (r != 0 && barrier) barrier->break_all();  // the others must not wait for this rank at a later barrier
      std::unique_lock<std::mutex> lk(m);
      rc[(size_t)d] = r;
      if (--pending == 0) cv_done.notify_all();
    }
  }
  // Runs f(d) on every worker; returns the first non-zero result (its device in *who).
  int run(const std::function<int(int)>& f, int* who = nullptr) {
    if (barrier) barrier->mend();  // (no worker is inside a job here)
    std::unique_lock<std::mutex> lk(m);
    job = f;
    pending = (int)th.size();
    ++gen;
    cv_job.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
    for (size_t d = 0; d < rc.size(); ++d)
      if (rc[d]) {
        if (who) *who = (int)d;
        return rc[d];
      }
    return 0;
  }
  void shutdown() {
    {
      std::unique_lock<std::mutex> lk(m);
      stop = true;
      cv_job.notify_all();
    }
    for (auto& t : th) t.join();
    th.clear();
  }
};

}  // namespace nbody

// Shared host-side declarations of libnbody_hip (context, error plumbing, timers).  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/nbody_hip.h"

struct nbody_timer {
  struct Pair {
    hipEvent_t a, b;
  };
  std::vector<Pair> pool;     // recorded, not yet read
  std::vector<Pair> free_;    // reusable
  double total_ms = 0.0;
  int64_t launches = 0;
  hipError_t begin(hipStream_t s, Pair* out);
  hipError_t end(hipStream_t s, const Pair& p);
  hipError_t drain();
  ~nbody_timer();
};

namespace nbody {

// Scoped bracket: records an event pair around a kernel launch when a timer is attached.
struct TimerScope {
  nbody_timer* t;
  hipStream_t s;
  nbody_timer::Pair p{};
  bool live = false;
  TimerScope(nbody_timer* timer, hipStream_t stream) : t(timer), s(stream) {
    if (t && t->begin(s, &p) == hipSuccess) live = true;
  }
  ~TimerScope() {
    if (live) (void)t->end(s, p);
  }
};

template <class T> struct Vec2T;
template <> struct Vec2T<float> { using type = float2; };
template <> struct Vec2T<double> { using type = double2; };

}  // namespace nbody

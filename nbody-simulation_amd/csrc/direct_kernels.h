// Launch interface of the direct O(N^2) kernels (direct_kernels.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbody {

struct DirectArgs {
  const float2* pos_all;  // [n_src] all source positions
  const float* mass_all;  // [n_src] `weight as f32`
  int n_src;
  int tgt_begin;  // targets are sources [tgt_begin, tgt_begin + n_tgt)
  int n_tgt;
  float2* vel;      // [n_tgt] in/out, or null (acceleration only)
  float2* pos_out;  // [n_tgt] out, or null
  float2* acc_out;  // [n_tgt] out, or null
  float2* partial;  // [gsplit][n_tgt] scratch when the grid splits the sources
  float delta;
  float clamp;
  float uniform_mass;  // > 0: every mass equals this value (mass_all is not read by the FAST kernel)
  const int* gate;  // optional device flag: the kernel runs only when (*gate != 0) == (run_if != 0)
  int run_if;
};

struct DirectConfig {
  int tpt = 1;        // targets per thread: 1, 2, 4
  int wsplit = 1;     // waves of a block sharing one target group and splitting the sources: 1 or 4
  int gsplit = 1;     // source split over blockIdx.y (partials + direct_finish)
  bool use_lds = false;
  bool use_asm = true;  // hand-ordered 8-pair block (LDS flavour, tpt 1)
};

hipError_t launch_direct_fast(hipStream_t s, const DirectArgs& a, const DirectConfig& c);
hipError_t launch_direct_exact(hipStream_t s, const DirectArgs& a);
hipError_t launch_hazard_scan(hipStream_t s, const float* xy, long n_floats, int* flag);
hipError_t launch_weights_to_mass(hipStream_t s, const uint32_t* w, float* m, long n);

}  // namespace nbody

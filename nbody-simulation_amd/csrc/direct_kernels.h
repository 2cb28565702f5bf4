// Launch interface of the direct O(N^2) kernels (direct_kernels.hip, nearfar.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nbody {

// Decision words at the start of the workspace (int each), written on the stream before the step's kernels.
enum { kFlagHazard = 0, kFlagFallback = 1, kFlagNearCount = 2, kFlagState = 3 };
// kFlagState: 0 = near/far split (main pass without the clamp), 1 = single clamped FAST pass, 2 = EXACT kernel.

struct DirectArgs {
  const float2* pos_all;  // [n_src] all positions: targets, integration base, EXACT sources, near sources
  const float2* src_pos;  // [n_src] sources of the FAST main pass (pos_all, or the far copy of nearfar.hip)
  const float* mass_all;  // [n_src] `weight as f32`
  int n_src;
  int tgt_begin;  // targets are bodies [tgt_begin, tgt_begin + n_tgt)
  int n_tgt;
  float2* vel;      // [n_tgt] in/out, or null (acceleration only)
  float2* pos_out;  // [n_tgt] out, or null
  float2* acc_out;  // [n_tgt] out, or null
  float2* partial;  // [gsplit][n_tgt] scratch
  int to_partial;   // main pass writes partial sums (direct_finish completes the step)
  float delta;
  float clamp;
  float uniform_mass;  // > 0: every mass equals this value (mass_all is not read by the FAST main pass)
  const int* flags;    // decision words
  int run_state;       // the kernel runs only when flags[kFlagState] == run_state; < 0: always
  const uint32_t* near_list;  // near sources (ascending body index), count in flags[kFlagNearCount]
  int src_couples;            // src_pos is in couples {xA, xB, yA, yB} (nearfar.hip, far_store<true>), n_src a multiple of kFarPad
  const float* src_minv;      // free per-body masses, streamed main pass: 1 / mass in the far copy's slot order (nearfar.hip); null otherwise
  const float* tile_mass;     // mass classes (capi.hip): src_pos is ordered by mass class, every 1024-source tile holds ONE class
                              // (classes padded with far-away points) and tile_mass[tile] is its mass; null otherwise
};

struct DirectConfig {
  int tpt = 1;     // targets per thread: 1 or 2
  int gsplit = 1;  // source split over blockIdx.y
  int use_asm = 3;      // tpt 1: 0 the compiler's schedule, 1 the hand-ordered 8-pair block, 2 packed couples (two pairs per packed op) through
                        // LDS, 3 packed couples with the far sources streamed through SGPRs (equal masses, no clamp; otherwise as 2)
  bool nearfar = true;  // per-step near/far split of the sources
};

struct NearFarLayout {
  size_t table_keys, table_bytes, is_near, scan, near_list, pos_far, minv_far, cub_temp, cub_temp_bytes, total;
};
NearFarLayout nearfar_layout(int64_t n_src);
// heavy_base > 0: bodies whose mass differs from it join the near list (sparse-heavy scenes; `mass` is then read)
// rank (optional, mass classes): body i's slot in the far copy (class order, classes padded to whole tiles); the n_pad_slots
// slots of pad_slots hold no body and receive the far-away point.  The far copy then has n + n_pad_slots entries.
constexpr int64_t kMaxMassClasses = 32;
constexpr int64_t kDirectTile = 1024;
constexpr float kFastBig = 1152921504606846976.0f;   // 2^60: FAST's domain (direct_kernels.hip, direct_hazard_scan)
constexpr float kFastTiny = 2.384185791015625e-07f;  // 2^-22
constexpr int kFarPad = 16;  // a far copy in couples is padded with far-away points to a multiple of this many sources
__host__ __device__ inline int64_t far_padded(int64_t n_slots) { return (n_slots + kFarPad - 1) / kFarPad * kFarPad; }
hipError_t launch_nearfar(hipStream_t s, const float2* pos, const float* mass, float heavy_base, int n, float clamp, int use_hazard,
                          int* flags, char* scratch, const NearFarLayout& L, const float2** pos_far, const uint32_t** near_list,
                          const uint32_t* rank = nullptr, const uint32_t* pad_slots = nullptr, int n_pad_slots = 0, bool couples = false,
                          const float** minv_far = nullptr);
// minv_far (optional, couples without classes): receives the far copy's second array, 1 / mass in slot order — the streamed
// per-body-mass main pass reads it (direct_stream_m)
hipError_t launch_decide_simple(hipStream_t s, int use_hazard, int* flags);

hipError_t launch_direct_fast(hipStream_t s, const DirectArgs& a, const DirectConfig& c, bool noclamp);
hipError_t launch_direct_finish(hipStream_t s, const DirectArgs& a, int n_gsplit, bool add_near);
hipError_t launch_direct_exact(hipStream_t s, const DirectArgs& a);
hipError_t launch_hazard_scan(hipStream_t s, const float* xy, long n_floats, int* flags);
hipError_t launch_weights_to_mass(hipStream_t s, const uint32_t* w, float* m, long n);

}  // namespace nbody

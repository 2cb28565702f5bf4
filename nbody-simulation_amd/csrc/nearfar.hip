// Near/far split of the sources of a direct step (gfx950).
//
// The clamp of the reference's force law (`if distance < 0.001 { distance = 0.001 }`, /root/reference
// src/main.rs:247-249) only ever changes a pair whose bodies are closer than sqrt(clamp).  v_max_f32 is a
// half-rate instruction on gfx950 and costs 10 % of the direct kernel (profiles/r01_noclamp_ab.txt), so the
// sources are split per step:
//   far  = bodies with no other body in their own or the 8 surrounding cells of a grid of pitch h >= sqrt(clamp).
//          For such a source every pair has max(|dx|,|dy|) > h, hence d2 > clamp: the clamp is the identity and
//          the main kernel runs without it.
//   near = everything else (a few dozen bodies out of a million in the benchmark's Plummer sphere).  They are
//          moved out of the main pass (their slot in the source array holds a point at (1e30,1e30): d2 overflows
//          to +inf, 1/inf = 0, the pair contributes exactly 0) and added by `direct_finish` with the clamp.
// Only the order of summation changes (near sources last), which FAST arithmetic does not promise anyway.
//
// Steps, all on the stream, no host round trip: cell keys -> radix sort (hipCUB) -> neighbour test (own cell: the
// adjacent sorted entries; (cx, cy+-1): adjacent in key order; (cx+-1, cy-1..cy+1): one binary search each) ->
// exclusive scan -> compacted ascending index list + far copy of the positions -> decision word.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "direct_kernels.h"

namespace nbody {

namespace {

constexpr int kCellBias = 1 << 30;  // cell indices are stored biased, valid while |x / h| < 2^30

__global__ __launch_bounds__(256) void nf_cell_keys(const float2* __restrict__ pos, int n, double inv_h,
                                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                    int* __restrict__ flags) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float2 p = pos[i];
  double fx = floor((double)p.x * inv_h), fy = floor((double)p.y * inv_h);
  bool bad = !(fx > -(double)kCellBias + 2 && fx < (double)kCellBias - 2 && fy > -(double)kCellBias + 2 &&
               fy < (double)kCellBias - 2);  // also catches NaN
  if (bad) {
    atomicOr(&flags[kFlagFallback], 1);
    fx = fy = 0;
  }
  uint32_t cx = (uint32_t)((long)fx + kCellBias), cy = (uint32_t)((long)fy + kCellBias);
  keys[i] = ((uint64_t)cx << 32) | cy;
  idx[i] = (uint32_t)i;
}

__device__ __forceinline__ int lower_bound_u64(const uint64_t* __restrict__ a, int n, uint64_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// is_near[i] = 1 when another body shares body i's cell or one of the 8 cells around it.
// `heavy_base` > 0: a body whose mass differs from it is listed as near too — the main pass then runs with the one mass
// hoisted out of the sum, and direct_finish adds the few odd bodies with their own masses (the reference's scene: two
// heavy bodies among 151 000 of weight 1, main.rs:282-291).
__global__ __launch_bounds__(256) void nf_mark(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx,
                                               int n, const float* __restrict__ mass, float heavy_base,
                                               uint32_t* __restrict__ is_near) {
  int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint64_t k = keys[r];
  bool near = false;
  // same cell, or (cx, cy-1) / (cx, cy+1): those are the entries next to r in key order
  if (r > 0) near |= (k - keys[r - 1]) <= 1;
  if (r + 1 < n) near |= (keys[r + 1] - k) <= 1;
  if (!near) {
    // columns cx-1 and cx+1, rows cy-1 .. cy+1: three consecutive keys each
    const uint64_t col = (uint64_t)1 << 32;
    uint64_t lo = k - col - 1;
    int p = lower_bound_u64(keys, n, lo);
    near |= (p < n && keys[p] <= lo + 2);
    if (!near) {
      lo = k + col - 1;
      p = lower_bound_u64(keys, n, lo);
      near |= (p < n && keys[p] <= lo + 2);
    }
  }
  const uint32_t body = idx[r];
  if (heavy_base > 0.f) near |= mass[body] != heavy_base;
  is_near[body] = near ? 1u : 0u;
}

__global__ __launch_bounds__(256) void nf_compact(const float2* __restrict__ pos, const uint32_t* __restrict__ is_near,
                                                  const uint32_t* __restrict__ scan, int n, float2* __restrict__ pos_far,
                                                  uint32_t* __restrict__ near_list) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float2 p = pos[i];
  if (is_near[i]) {
    near_list[scan[i]] = (uint32_t)i;  // ascending body index: the near sum has a fixed order
    p = make_float2(1e30f, 1e30f);
  }
  pos_far[i] = p;
}

// state: 0 = near/far split (no clamp in the main pass), 1 = single clamped pass, 2 = EXACT kernel.
__global__ void nf_decide(const uint32_t* __restrict__ is_near, const uint32_t* __restrict__ scan, int n, int max_near,
                          int use_hazard, int* __restrict__ flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int m = n > 0 ? (int)(scan[n - 1] + is_near[n - 1]) : 0;
  flags[kFlagNearCount] = m;
  int state = (flags[kFlagFallback] != 0 || m > max_near) ? 1 : 0;
  if (use_hazard && flags[kFlagHazard] != 0) state = 2;
  flags[kFlagState] = state;
}

__global__ void nf_decide_simple(int use_hazard, int* __restrict__ flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  flags[kFlagNearCount] = 0;
  flags[kFlagState] = (use_hazard && flags[kFlagHazard] != 0) ? 2 : 1;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// Layout of the near/far scratch inside the caller's workspace (all 256-B aligned).
NearFarLayout nearfar_layout(int64_t n_src) {
  NearFarLayout L{};
  size_t n = (size_t)(n_src > 0 ? n_src : 1);
  size_t off = 0;
  L.keys0 = off; off += align_up(n * 8);
  L.keys1 = off; off += align_up(n * 8);
  L.idx0 = off; off += align_up(n * 4);
  L.idx1 = off; off += align_up(n * 4);
  L.is_near = off; off += align_up(n * 4);
  L.scan = off; off += align_up(n * 4);
  L.near_list = off; off += align_up(n * 4);
  L.pos_far = off; off += align_up(n * 8);
  L.cub_temp = off;
  L.cub_temp_bytes = align_up((size_t)8 << 20);  // radix sort with double buffers + scan need far less
  off += L.cub_temp_bytes;
  L.total = off;
  return L;
}

// Enqueues the split.  On return flags[kFlagState] is (will be) valid on the stream.
hipError_t launch_nearfar(hipStream_t s, const float2* pos, const float* mass, float heavy_base, int n, float clamp, int use_hazard,
                          int* flags, char* scratch, const NearFarLayout& L, const float2** pos_far, const uint32_t** near_list) {
  uint64_t* k0 = (uint64_t*)(scratch + L.keys0);
  uint64_t* k1 = (uint64_t*)(scratch + L.keys1);
  uint32_t* i0 = (uint32_t*)(scratch + L.idx0);
  uint32_t* i1 = (uint32_t*)(scratch + L.idx1);
  uint32_t* is_near = (uint32_t*)(scratch + L.is_near);
  uint32_t* scan = (uint32_t*)(scratch + L.scan);
  uint32_t* list = (uint32_t*)(scratch + L.near_list);
  float2* far = (float2*)(scratch + L.pos_far);
  *pos_far = far;
  *near_list = list;
  const double h = sqrt((double)clamp) * 1.001;  // pitch strictly above sqrt(clamp), margin >> f32 rounding of d2
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(nf_cell_keys, dim3(blocks), dim3(256), 0, s, pos, n, 1.0 / h, k0, i0, flags);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipcub::DoubleBuffer<uint64_t> dk(k0, k1);
  hipcub::DoubleBuffer<uint32_t> dv(i0, i1);
  size_t need = 0;
  e = hipcub::DeviceRadixSort::SortPairs(nullptr, need, dk, dv, n, 0, 64, s);
  if (e != hipSuccess) return e;
  size_t need_scan = 0;
  e = hipcub::DeviceScan::ExclusiveSum(nullptr, need_scan, is_near, scan, n, s);
  if (e != hipSuccess) return e;
  if (need > L.cub_temp_bytes || need_scan > L.cub_temp_bytes) {
    // never expected; stay correct by taking the single clamped pass
    hipLaunchKernelGGL(nf_decide_simple, dim3(1), dim3(1), 0, s, use_hazard, flags);
    return hipGetLastError();
  }
  size_t tb = L.cub_temp_bytes;
  e = hipcub::DeviceRadixSort::SortPairs(scratch + L.cub_temp, tb, dk, dv, n, 0, 64, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nf_mark, dim3(blocks), dim3(256), 0, s, dk.Current(), dv.Current(), n, mass, heavy_base, is_near);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  tb = L.cub_temp_bytes;
  e = hipcub::DeviceScan::ExclusiveSum(scratch + L.cub_temp, tb, is_near, scan, n, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nf_compact, dim3(blocks), dim3(256), 0, s, pos, is_near, scan, n, far, list);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nf_decide, dim3(1), dim3(1), 0, s, is_near, scan, n, n / 64, use_hazard, flags);
  return hipGetLastError();
}

hipError_t launch_decide_simple(hipStream_t s, int use_hazard, int* flags) {
  hipLaunchKernelGGL(nf_decide_simple, dim3(1), dim3(1), 0, s, use_hazard, flags);
  return hipGetLastError();
}

}  // namespace nbody

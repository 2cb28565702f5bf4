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
// Steps, all on the stream, no host round trip (round 3: a hash grid instead of a sort — the question "is any other body in
// my 3 x 3 cells" needs no order, and the sort was two thirds of the split's 0.33 ms: hipCUB picks a merge sort for 64-bit keys
// at these sizes, profiles/r02_direct_kernel_stats.csv):
//   one memset (the table: every slot empty) -> nf_insert: each body enters its cell in the slot of its 2 x 2 super-cell (open
//   addressing, atomicCAS on 32-bit slots, linear probing, load factor <= 1/4) -> nf_mark: four lookups per body, their first
//   probes side by side; near iff the 3 x 3 cells around it hold anyone else -> exclusive scan -> compacted ascending index
//   list + far copy of the positions -> decision word.
// Which bodies are near is a function of the positions alone, whatever order the inserts land in; the list is compacted in
// ascending body index: the step stays bitwise reproducible.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "direct_kernels.h"

namespace nbody {

namespace {

constexpr int kCellBias = 1 << 29;  // cell indices are stored biased, valid while |x / h| < 2^29
// The table holds SUPER-cells of 2 x 2 cells, 32 bits each: a 27-bit fingerprint of the super-cell's coordinates, a "more than
// one body" bit and four occupancy bits (one per cell).  A body's 3 x 3 cells lie in 2 x 2 super-cells: four lookups instead
// of nine, in a table an eighth the size of one with 64-bit keys per cell — the lookups are random accesses, and what they
// cost is the lines they pull through the L2s (191 us for nine 8-byte probes per body at N = 1 M).  Two super-cells whose
// fingerprints collide in one probe chain are merged: that can only turn a far body into a near one (which is always
// correct: near bodies are summed with the clamp), deterministically, with probability ~2^-27 per lookup.
constexpr uint32_t kEmptySlot = ~0u;  // the memset's 0xFF bytes; fingerprints never have all 27 bits set
constexpr uint32_t kMultiBit = 16u;

struct SuperCell {
  uint32_t slot, fp;
};
__device__ __forceinline__ SuperCell super_of(uint32_t sx, uint32_t sy, uint32_t tmask) {
  uint64_t k = ((uint64_t)sx << 32) | sy;  // splitmix64's finaliser: neighbouring super-cells land far apart
  k ^= k >> 30;
  k *= 0xBF58476D1CE4E5B9ull;
  k ^= k >> 27;
  k *= 0x94D049BB133111EBull;
  k ^= k >> 31;
  uint32_t fp = (uint32_t)(k >> 37);
  if (fp == 0x7FFFFFFu) fp = 0x7FFFFFEu;
  return SuperCell{(uint32_t)k & tmask, fp};
}
__device__ __forceinline__ bool cell_of(float2 p, double inv_h, uint32_t* cx, uint32_t* cy) {  // false: outside the grid's range (or NaN)
  double fx = floor((double)p.x * inv_h), fy = floor((double)p.y * inv_h);
  const bool ok = fx > -(double)kCellBias + 2 && fx < (double)kCellBias - 2 && fy > -(double)kCellBias + 2 && fy < (double)kCellBias - 2;
  if (!ok) fx = fy = 0;
  *cx = (uint32_t)((long)fx + kCellBias);
  *cy = (uint32_t)((long)fy + kCellBias);
  return ok;
}

// Every body enters its cell's bit in its super-cell's slot; the second body to arrive in a super-cell sets the "multi" bit.
// `hazard`: the FAST domain check of direct_hazard_scan rides along (the same positions, one pass: a launch fewer per step).
__global__ __launch_bounds__(256) void nf_insert(const float2* __restrict__ pos, int n, double inv_h, uint32_t* __restrict__ table,
                                                 uint32_t tmask, int* __restrict__ flags, int hazard) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t cx, cy;
  const float2 p = pos[i];
  if (hazard) {
    const float vx = __builtin_fabsf(p.x), vy = __builtin_fabsf(p.y);
    const bool bad = !(vx < kFastBig) || (vx != 0.f && vx < kFastTiny) || !(vy < kFastBig) || (vy != 0.f && vy < kFastTiny);
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(bad))) atomicOr(&flags[kFlagHazard], 1);
  }
  if (!cell_of(p, inv_h, &cx, &cy)) atomicOr(&flags[kFlagFallback], 1);
  const SuperCell sc = super_of(cx >> 1, cy >> 1, tmask);
  const uint32_t bit = 1u << ((cx & 1) * 2 + (cy & 1));
  uint32_t h = sc.slot;
  for (uint32_t probe = 0; probe <= tmask; ++probe) {  // (load factor <= 1/4: a free slot is a probe or two away)
    const uint32_t old = atomicCAS(&table[h], kEmptySlot, (sc.fp << 5) | bit);
    if (old == kEmptySlot) return;
    if ((old >> 5) == sc.fp) {
      if ((old & (bit | kMultiBit)) != (bit | kMultiBit)) atomicOr(&table[h], bit | kMultiBit);
      return;
    }
    h = (h + 1) & tmask;
  }
  atomicOr(&flags[kFlagFallback], 1);  // never expected: the table is larger than the number of bodies
}

// is_near[i] = 1 when another body shares body i's cell or one of the 8 cells around it.
// `heavy_base` > 0: a body whose mass differs from it is listed as near too — the main pass then runs with the one mass
// hoisted out of the sum, and direct_finish adds the few odd bodies with their own masses (the reference's scene: two
// heavy bodies among 151 000 of weight 1, main.rs:282-291).
__global__ __launch_bounds__(256) void nf_mark(const float2* __restrict__ pos, int n, double inv_h, const uint32_t* __restrict__ table,
                                               uint32_t tmask, const float* __restrict__ mass, float heavy_base,
                                               uint32_t* __restrict__ is_near) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t cx, cy;
  (void)cell_of(pos[i], inv_h, &cx, &cy);
  const uint32_t sx = cx >> 1, sy = cy >> 1;
  // the 3 x 3 cells around (cx, cy) reach one super-cell further on the side of the body's cell: -1 from an even cell, +1 from an odd one
  const uint32_t ox = sx + ((cx & 1) ? 1u : ~0u), oy = sy + ((cy & 1) ? 1u : ~0u);
  // which cells of each of the four super-cells belong to the 3 x 3 block (the body's own cell left out): bit = x parity * 2 + y parity
  const uint32_t own_bit = 1u << ((cx & 1) * 2 + (cy & 1));
  const uint32_t far_x = (cx & 1) ? 0u : 1u, far_y = (cy & 1) ? 0u : 1u;  // parity of the one column / row the block has in the other super-cell
  uint32_t need[4];
  need[0] = 15u & ~own_bit;                                       // (sx, sy): its other three cells
  need[1] = (1u << (far_x * 2)) | (1u << (far_x * 2 + 1));        // (ox, sy): one column, both rows
  need[2] = (1u << far_y) | (1u << (2 + far_y));                  // (sx, oy): one row, both columns
  need[3] = 1u << (far_x * 2 + far_y);                            // (ox, oy): the corner cell
  SuperCell sc[4] = {super_of(sx, sy, tmask), super_of(ox, sy, tmask), super_of(sx, oy, tmask), super_of(ox, oy, tmask)};
  uint32_t got[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) got[c] = table[sc[c].slot];  // the four first probes side by side: one round trip
  bool near = false;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    uint32_t at = got[c], hh = sc[c].slot;
    for (uint32_t probe = 0; at != kEmptySlot && (at >> 5) != sc[c].fp && probe <= tmask; ++probe) {
      hh = (hh + 1) & tmask;
      at = table[hh];
    }
    if (at != kEmptySlot) near |= (at & need[c]) != 0 || (c == 0 && (at & kMultiBit) != 0);
  }
  if (heavy_base > 0.f) near |= mass[i] != heavy_base;
  is_near[i] = near ? 1u : 0u;
}

// The near list (ascending body index: the near sum has a fixed order) and the far copy of the positions.
// With mass classes the far copy is written in class order (rank[i]) and the threads past n fill the padding slots.
// COUPLES (the packed direct kernels): slot s of the far copy lives in couple s / 2 = {xA, xB, yA, yB} — x at float
// 4 (s / 2) + (s & 1), y two floats on — and the slots from n_slots up to the next multiple of kFarPad hold the far-away point,
// so the main pass reads whole 16-source iterations.
template <bool COUPLES>
__device__ __forceinline__ void far_store(float2* __restrict__ pos_far, uint32_t slot, float2 p) {
  if constexpr (COUPLES) {
    float* f = reinterpret_cast<float*>(pos_far) + 4 * (size_t)(slot >> 1) + (slot & 1u);
    f[0] = p.x;
    f[2] = p.y;
  } else {
    pos_far[slot] = p;
  }
}
// The thread of the last body also decides (nf_decide's rule: everything it reads was written by the kernels before this one).
template <bool COUPLES>
__global__ __launch_bounds__(256) void nf_compact(const float2* __restrict__ pos, const uint32_t* __restrict__ is_near,
                                                  const uint32_t* __restrict__ scan, int n, float2* __restrict__ pos_far,
                                                  uint32_t* __restrict__ near_list, const uint32_t* __restrict__ rank,
                                                  const uint32_t* __restrict__ pad_slots, int n_pad_slots, int n_slots, int max_near,
                                                  int use_hazard, int* __restrict__ flags, const float* __restrict__ mass,
                                                  float* __restrict__ minv_far) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == n - 1) {  // state: 0 = near/far split (no clamp in the main pass), 1 = single clamped pass, 2 = EXACT kernel
    const int m = (int)(scan[n - 1] + is_near[n - 1]);
    flags[kFlagNearCount] = m;
    int state = (flags[kFlagFallback] != 0 || m > max_near) ? 1 : 0;
    if (use_hazard && flags[kFlagHazard] != 0) state = 2;
    flags[kFlagState] = state;
  }
  if (i >= n) {
    const int k = i - n;
    if (k < n_pad_slots) far_store<COUPLES>(pos_far, pad_slots[k], make_float2(1e30f, 1e30f));
    else if (COUPLES && k - n_pad_slots < (int)far_padded(n_slots) - n_slots) {
      far_store<COUPLES>(pos_far, (uint32_t)(n_slots + k - n_pad_slots), make_float2(1e30f, 1e30f));
      if (minv_far) minv_far[n_slots + k - n_pad_slots] = 1.0f;  // (any finite value: the far-away point contributes 0)
    }
    return;
  }
  float2 p = pos[i];
  if (is_near[i]) {
    near_list[scan[i]] = (uint32_t)i;
    p = make_float2(1e30f, 1e30f);
  }
  far_store<COUPLES>(pos_far, rank ? rank[i] : (uint32_t)i, p);
  // free per-body masses (no classes: slot = body): the inverse mass the streamed main pass multiplies the denominator by.
  // IEEE division, as direct_fast's tile_minv; 1/0 = inf: a zero-mass source contributes exactly 0
  if (minv_far) minv_far[i] = 1.0f / mass[i];
}

__global__ void nf_decide_simple(int use_hazard, int* __restrict__ flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  flags[kFlagNearCount] = 0;
  flags[kFlagState] = (use_hazard && flags[kFlagHazard] != 0) ? 2 : 1;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// Layout of the near/far scratch inside the caller's workspace (all 256-B aligned).
static uint32_t table_slots(int64_t n_src) {  // a power of two, at least four times the bodies (a probe is an atomic's round trip)
  uint32_t m = 1024;
  while ((int64_t)m < 4 * n_src && m < (1u << 31)) m <<= 1;
  return m;
}
NearFarLayout nearfar_layout(int64_t n_src) {
  NearFarLayout L{};
  size_t n = (size_t)(n_src > 0 ? n_src : 1);
  const size_t slots = table_slots(n_src);
  size_t off = 0;
  L.table_keys = off; off += align_up(slots * 4);
  L.table_bytes = slots * 4;
  L.is_near = off; off += align_up(n * 4);
  L.scan = off; off += align_up(n * 4);
  L.near_list = off; off += align_up(n * 4);
  L.pos_far = off; off += align_up((n + (size_t)(kMaxMassClasses + 1) * (size_t)kDirectTile + 2 * kFarPad) * 8);  // (room for the mass classes' padding, the couples' padding and one iteration read ahead)
  L.minv_far = off; off += align_up((n + 4 * (size_t)kFarPad) * 4);  // (free masses, no classes: slot = body; the couples' padding and one iteration read ahead)
  L.cub_temp = off;
  size_t need_scan = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, need_scan, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, (hipStream_t) nullptr);
  L.cub_temp_bytes = align_up(need_scan + 256);
  off += L.cub_temp_bytes;
  L.total = off;
  return L;
}

// Enqueues the split.  On return flags[kFlagState] is (will be) valid on the stream.
hipError_t launch_nearfar(hipStream_t s, const float2* pos, const float* mass, float heavy_base, int n, float clamp, int use_hazard,
                          int* flags, char* scratch, const NearFarLayout& L, const float2** pos_far, const uint32_t** near_list,
                          const uint32_t* rank, const uint32_t* pad_slots, int n_pad_slots, bool couples, const float** minv_far) {
  uint32_t* table = (uint32_t*)(scratch + L.table_keys);
  uint32_t* is_near = (uint32_t*)(scratch + L.is_near);
  uint32_t* scan = (uint32_t*)(scratch + L.scan);
  uint32_t* list = (uint32_t*)(scratch + L.near_list);
  float2* far = (float2*)(scratch + L.pos_far);
  *pos_far = far;
  *near_list = list;
  float* minv = nullptr;
  if (minv_far) {
    if (!couples || rank || !mass) return hipErrorInvalidValue;  // slot order = body order only without classes
    minv = (float*)(scratch + L.minv_far);
    *minv_far = minv;
  }
  const double h = sqrt((double)clamp) * 1.001;  // pitch strictly above sqrt(clamp), margin >> f32 rounding of d2
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const uint32_t tmask = table_slots(n) - 1;
  hipError_t e = hipMemsetAsync(table, 0xFF, L.table_bytes, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nf_insert, dim3(blocks), dim3(256), 0, s, pos, n, 1.0 / h, table, tmask, flags, use_hazard);
  hipLaunchKernelGGL(nf_mark, dim3(blocks), dim3(256), 0, s, pos, n, 1.0 / h, (const uint32_t*)table, tmask, mass, heavy_base, is_near);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  size_t tb = L.cub_temp_bytes;
  e = hipcub::DeviceScan::ExclusiveSum(scratch + L.cub_temp, tb, is_near, scan, n, s);
  if (e != hipSuccess) return e;
  if ((int64_t)n_pad_slots > (kMaxMassClasses + 1) * kDirectTile) return hipErrorInvalidValue;
  const int n_pad = rank ? n_pad_slots : 0;
  const int n_slots = n + n_pad;  // (mass classes: the padded class order; otherwise the bodies themselves)
  const unsigned cblocks = (unsigned)((n + n_pad + (couples ? kFarPad : 0) + 255) / 256);
  if (couples)
    hipLaunchKernelGGL(nf_compact<true>, dim3(cblocks), dim3(256), 0, s, pos, is_near, scan, n, far, list, rank, pad_slots, n_pad, n_slots, n / 64,
                       use_hazard, flags, mass, minv);
  else
    hipLaunchKernelGGL(nf_compact<false>, dim3(cblocks), dim3(256), 0, s, pos, is_near, scan, n, far, list, rank, pad_slots, n_pad, n_slots,
                       n / 64, use_hazard, flags, mass, (float*)nullptr);
  return hipGetLastError();
}

hipError_t launch_decide_simple(hipStream_t s, int use_hazard, int* flags) {
  hipLaunchKernelGGL(nf_decide_simple, dim3(1), dim3(1), 0, s, use_hazard, flags);
  return hipGetLastError();
}

}  // namespace nbody

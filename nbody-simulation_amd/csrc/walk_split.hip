// The Barnes-Hut walk in three passes — same nodes, same pairs, same operations, same order of additions as the fused
// walk (tree_kernels.hip) and the CPU recursion (/root/reference src/main.rs:348-386), so still bit-identical — for
// trees with big leaves (the BVH: up to 64 particles per leaf), f32.
//
// Why: in the fused walk a wave that reaches a leaf evaluates the leaf's particles one after the other for all of
// its lanes at once, ~56 instructions per particle (two IEEE divisions) whether 3 or 60 lanes take part; the targets
// near the reference scene's heavy bodies visit 20x the median number of leaves, their waves run 1.2 ms while most of
// the chip idles.  The only thing that has to be sequential is the ADDITION of a target's terms; their values do not
// depend on one another.  So:
//   1. walk_count: the traversal alone (node tests, no arithmetic): how many terms does each target have;
//      an exclusive scan turns the counts into offsets into one big term array (HBM is 288 GB: ~250 MB here);
//   2. walk_terms: the traversal again; an accepted node writes its term; at a leaf the wave takes its acting lanes
//      one at a time and all 64 lanes evaluate that target against 64 particles of the leaf at once (lane = particle),
//      so the cost of a leaf step is proportional to the lanes that want it, and nothing is summed;
//      a pair the reference skips (|dx|+|dy| not normal, main.rs:241-243) writes -0.0, the identity of IEEE addition;
//   3. walk_sum: per target, the terms are added in order from +0.0 — one v_add_f32_dpp per term and coordinate, a row of
//      16 lanes per target.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>

#include <cstdio>
#include <vector>

#include "bvh_build.h"
#include "div_pair.h"
#include "env.h"
#include "walk_split.h"

namespace nbody {

namespace {

// Targets per wave.  Counting is cheapest with full waves; in the term pass a wave's time grows with the number of
// leaves its targets visit, so its waves are cut by work (see walk_pass).
constexpr int kCountTPW = 64;
[[maybe_unused]] constexpr uint32_t kTermBudget = 8192;  // (three-pass walk, laboratory build) terms a wave of the term pass writes, about (at least: see walk_total)
[[maybe_unused]] constexpr uint32_t kBudgetTargets = 12; // ... or this many average targets' worth, if that is more
constexpr uint32_t kTileBudget = 8192;        // smallest budget of a wave of the one-pass walk (walk_tile)
// Waves the one-pass walk aims at; a wave's budget is total terms / (this - n / 64), any integer (tile_budget).  Round 4, measured
// (profiles/r04_walk_wave_target.txt): the walk is its longest waves' chains, and they all start at once only while EVERY wave is
// resident — 256 CUs x 32 slots = 8 192.  Round 3 aimed at 16 384 ("twice what the chip holds") with power-of-two budgets: 9 700 waves
// on the bench's reference-scene leg, a second residency round (exact 0.636 ms, FAST 0.338).  Kept inside one round the same walk takes
// 0.48-0.49 / 0.29-0.30 ms; Plummer 262 144: 0.76 -> 0.61-0.63, 400 000: 1.32 -> 1.15.  Where the head count alone exceeds the
// chip (n / 64 > 8 192) every extra wave repeats a traversal for nothing: 1 024 more than the head count is the measured optimum at
// 400 000, 655 360 and 1 048 576 bodies.
__host__ __device__ inline int64_t tile_waves_target(int64_t n_tgt) {
  const int64_t w = n_tgt / 64 + 1024;
  return w > 6656 ? w : 6656;
}
#ifndef NB_TILE_ROUND_COST
#define NB_TILE_ROUND_COST 66
#endif
#ifndef NB_FAST_ROUND_COST
#define NB_FAST_ROUND_COST 19
#endif
// Which group of four waves a work-group takes (WalkArgs::block_stride).
__device__ __forceinline__ unsigned group_of_block(unsigned b, unsigned nb, int stride) {
  return stride > 1 ? (unsigned)(((unsigned long long)b * (unsigned)stride) % nb) : b;
}
template <class T> __device__ __forceinline__ unsigned group_of_block(const WalkArgs<T>& a, unsigned b, unsigned nb) {
  if (a.group_order) {  // chunk by chunk, the chunks heaviest first (walk_order_chunks); inside a chunk in order (tree-order neighbours share their L2 lines)
    const unsigned slot = b / (unsigned)a.order_chunk;
    return (unsigned)a.group_order[slot] * (unsigned)a.order_chunk + (b - slot * (unsigned)a.order_chunk);
  }
  return group_of_block(b, nb, a.block_stride);
}
constexpr int kTileRoundCost = NB_TILE_ROUND_COST;  // instructions a target costs at a leaf, lane = particle (a round + its share of the adds)
constexpr int kFusedPairCost = 48;            // ... and a particle costs the wave, lane = target

__device__ __forceinline__ float lane_f(float v, int k) {  // k uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

// calculate_gravity (main.rs:234-253) up to, but not including, the `+=`
__device__ __forceinline__ float2 pair_term(float px, float py, float qx, float qy, float force, float clamp) {
  const float dx = qx - px;                                        // :236
  const float dy = qy - py;
  const float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);     // :238
  if (!__builtin_isnormal(sum)) return make_float2(-0.0f, -0.0f);  // :241-243: no addition at all == adding -0.0
  float distance = dx * dx + dy * dy;                              // :245
  // :247-249 `if distance < 0.001 { distance = 0.001 }` as one v_max_f32 (half the cost of compare + select): `distance`
  // is never NaN here (a normal `sum` means finite dx, dy), and for a NaN clamp both forms keep `distance`
  distance = __builtin_fmaxf(distance, clamp);
  const float den = sum * distance;
  return div_pair(dx * force, dy * force, den);                    // :252 (div_pair.h: the two quotients, packed)
}
// The same with a per-lane `valid` folded into the skip: a lane past the leaf's end yields -0.0 like a skipped pair, under
// the one exec mask (a select afterwards costs two v_cndmask per round).
__device__ __forceinline__ float2 pair_term_if(bool valid, float px, float py, float qx, float qy, float force, float clamp) {
  const float dx = qx - px;
  const float dy = qy - py;
  const float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);
  if (!(valid && __builtin_isnormal(sum))) return make_float2(-0.0f, -0.0f);
  float distance = dx * dx + dy * dy;
  distance = __builtin_fmaxf(distance, clamp);
  const float den = sum * distance;
  return div_pair(dx * force, dy * force, den);
}

// ... and as straight-line code: the term computed for every lane, kept by a select (the quotients of a skipped pair are whatever
// the division makes of its operands and are thrown away).  No exec region, so the rounds of two targets sit in ONE basic block.
// (The two rounds written as packed ops over the two targets — subtractions, squares, denominators, numerators and the divisions'
// multiply-adds — are bit-identical too and measured SLOWER: 66 VGPRs instead of 60, seven waves per SIMD, Plummer 1 M 5.70 -> 6.00 ms.)
__device__ __forceinline__ float2 pair_term_sel(bool valid, float px, float py, float qx, float qy, float force, float clamp) {
  const float dx = qx - px;
  const float dy = qy - py;
  const float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);
  const bool ok = valid & __builtin_isnormal(sum);
  const float distance = __builtin_fmaxf(dx * dx + dy * dy, clamp);
  const float den = sum * distance;
  const float2 t = div_pair(dx * force, dy * force, den);
  return make_float2(ok ? t.x : -0.0f, ok ? t.y : -0.0f);
}

// nbody_arith FAST (opt-in, tolerance instead of bit parity): one reciprocal instead of two IEEE divisions; a zero difference
// contributes exactly 0 through the biased denominator (direct_kernels.hip)
[[maybe_unused]] __device__ __forceinline__ float2 pair_term_fast(float px, float py, float qx, float qy, float force, float clamp) {
  const float dx = qx - px, dy = qy - py;
  const float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);
  const float d2 = __builtin_fmaxf(__builtin_fmaf(dy, dy, dx * dx), clamp);
  const float s = force * __builtin_amdgcn_rcpf(__builtin_fmaf(sum, d2, 8.0779356694631609e-28f));  // 2^-90
  return make_float2(dx * s, dy * s);
}
template <bool FAST> __device__ __forceinline__ float2 term_of(float px, float py, float qx, float qy, float force, float clamp) {
  if constexpr (FAST) return pair_term_fast(px, py, qx, qy, force, clamp);
  else return pair_term(px, py, qx, qy, force, clamp);
}

// ---- the same pieces for either precision (walk_tile) ---------------------------------------------------------------
template <class T> struct Vec2Of;
template <> struct Vec2Of<float> { using type = float2; };
template <> struct Vec2Of<double> { using type = double2; };
template <class T> struct Vec4Of;
template <> struct Vec4Of<float> { using type = float4; };
template <> struct Vec4Of<double> { using type = double4; };
__device__ __forceinline__ float lane_t(float v, int k) { return lane_f(v, k); }
__device__ __forceinline__ double lane_t(double v, int k) {  // k uniform
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, k), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), k);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double2 pair_term(double px, double py, double qx, double qy, double force, double clamp) {
  const double dx = qx - px;                                       // main.rs:236
  const double dy = qy - py;
  const double sum = __builtin_fabs(dx) + __builtin_fabs(dy);      // :238
  if (!__builtin_isnormal(sum)) return make_double2(-0.0, -0.0);   // :241-243
  double distance = dx * dx + dy * dy;                             // :245
  distance = __builtin_fmax(distance, clamp);                      // :247-249 (see the f32 version)
  const double den = sum * distance;
  return make_double2((dx * force) / den, (dy * force) / den);     // :252
}
__device__ __forceinline__ double2 pair_term_if(bool valid, double px, double py, double qx, double qy, double force, double clamp) {
  const double dx = qx - px;
  const double dy = qy - py;
  const double sum = __builtin_fabs(dx) + __builtin_fabs(dy);
  if (!(valid && __builtin_isnormal(sum))) return make_double2(-0.0, -0.0);
  double distance = dx * dx + dy * dy;
  distance = __builtin_fmax(distance, clamp);
  const double den = sum * distance;
  return make_double2((dx * force) / den, (dy * force) / den);
}
__device__ __forceinline__ double2 pair_term_sel(bool valid, double px, double py, double qx, double qy, double force, double clamp) {
  const double dx = qx - px;
  const double dy = qy - py;
  const double sum = __builtin_fabs(dx) + __builtin_fabs(dy);
  const bool ok = valid & __builtin_isnormal(sum);
  const double distance = __builtin_fmax(dx * dx + dy * dy, clamp);
  const double den = sum * distance;
  const double tx = (dx * force) / den, ty = (dy * force) / den;
  return make_double2(ok ? tx : -0.0, ok ? ty : -0.0);
}
__device__ __forceinline__ double2 pair_term_fast(double px, double py, double qx, double qy, double force, double clamp) {
  const double dx = qx - px, dy = qy - py;
  const double sum = __builtin_fabs(dx) + __builtin_fabs(dy);
  const double d2 = __builtin_fmax(__builtin_fma(dy, dy, dx * dx), clamp);
  const double den = __builtin_fma(sum, d2, 0x1p-700);
  double r = __builtin_amdgcn_rcp(den);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  const double sc = force * r;
  return make_double2(dx * sc, dy * sc);
}
template <bool FAST> __device__ __forceinline__ double2 term_of(double px, double py, double qx, double qy, double force, double clamp) {
  if constexpr (FAST) return pair_term_fast(px, py, qx, qy, force, clamp);
  else return pair_term(px, py, qx, qy, force, clamp);
}
template <class T> __device__ __forceinline__ typename Vec2Of<T>::type neg_zero2() {
  typename Vec2Of<T>::type v;
  v.x = (T)-0.0;
  v.y = (T)-0.0;
  return v;
}

// The counting traversal for either precision: node tests only (walk_pass<false> is its f32 twin).
template <class T>
__global__ __launch_bounds__(256) void walk_count(const WalkArgs<T> a, uint32_t* __restrict__ cnt) {
  using T2 = typename Vec2Of<T>::type;
  using T4 = typename Vec4Of<T>::type;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = t < a.n_tgt;
  const int64_t row = live ? (a.tgt_index ? (int64_t)a.tgt_index[t] : t) : 0;
  const T2 p = live ? reinterpret_cast<const T2*>(a.tgt_pos)[row] : T2{0, 0};
  const T4* __restrict__ g0 = reinterpret_cast<const T4*>(a.geom0);
  const T4* __restrict__ g1 = reinterpret_cast<const T4*>(a.geom1);
  const int4* __restrict__ lk = reinterpret_cast<const int4*>(a.link);
  const T theta = a.theta;
  const int n_nodes = a.n_nodes;
  int resume = live ? 0 : n_nodes;
  uint32_t n_terms = 0;
  int i = 0;
  while (i < n_nodes) {
    const int4 l = lk[i];
    const T4 b = g0[i];
    const T4 c = g1[i];
    const bool act = resume <= i;
    int next;
    if (l.w) {
      if (act) {
        n_terms += (uint32_t)l.z;
        resume = l.x;
      }
      next = l.x;
    } else {
      bool descend = false;
      if (act) {
        const bool contains = p.y > b.y && p.x > b.x && p.x < b.z && p.y < b.w;
        const T ddx = p.x - c.x, ddy = p.y - c.y;
        const T d2 = ddx * ddx + ddy * ddy;
        if (!contains && c.w < d2 * theta * theta) {
          ++n_terms;
          resume = l.x;
        } else {
          descend = true;
          resume = i + 1;
        }
      }
      next = __builtin_amdgcn_ballot_w64(descend) != 0 ? i + 1 : l.x;
    }
    i = __builtin_amdgcn_readfirstlane(next);
  }
  if (live) cnt[t] = n_terms;
}

#ifdef NBODY_LAB  // ---- the three-pass walk of round 1 (count / emit terms / sum): kept for A/B runs, laboratory build only
// The traversal both passes share.  F: what to do with an accepted node / a leaf.
template <bool EMIT, int kTPW, bool FAST>
__global__ __launch_bounds__(256) void walk_pass(const WalkArgs<float> a, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                                                 float2* __restrict__ terms, const int* __restrict__ info, int64_t capacity) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (EMIT && (info[1] != 0)) return;  // the term array is too small: the caller grows it
  int64_t t;
  bool live;
  if (EMIT) {
    // Waves by WORK, not by head count: wave w takes the targets t with g(t) = off[t] / budget + t / 64 == w
    // (g never decreases: at most 64 targets, about `budget` terms — a target with thousands of terms walks alone,
    // and its wave is as short as its own path).  The two ends of the range by binary search.
    const uint32_t budget = (uint32_t)info[3];  // set by walk_total
    int64_t lo = 0, hi = a.n_tgt;
    while (lo < hi) {  // first t with g(t) >= wave
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)(off[mid] / budget) + (mid >> 6) < wave) lo = mid + 1; else hi = mid;
    }
    const int64_t t0 = lo;
    hi = t0 + 64 < a.n_tgt ? t0 + 64 : a.n_tgt;
    while (lo < hi) {  // first t with g(t) > wave
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)(off[mid] / budget) + (mid >> 6) <= wave) lo = mid + 1; else hi = mid;
    }
    if (lo == t0) return;  // no target has this number
    t = t0 + lane;
    live = t < lo;
  } else {
    t = wave * kTPW + lane;
    live = lane < kTPW && t < a.n_tgt;
  }
  const int64_t row = live ? (a.tgt_index ? (int64_t)a.tgt_index[t] : t) : 0;
  const float2 p = live ? reinterpret_cast<const float2*>(a.tgt_pos)[row] : make_float2(0.f, 0.f);
  const float4* __restrict__ g0 = reinterpret_cast<const float4*>(a.geom0);
  const float4* __restrict__ g1 = reinterpret_cast<const float4*>(a.geom1);
  const int4* __restrict__ lk = reinterpret_cast<const int4*>(a.link);
  const float2* __restrict__ lpos = reinterpret_cast<const float2*>(a.leaf_pos);
  const float* __restrict__ lmass = a.leaf_mass;
  const float theta = a.theta, clamp = a.clamp;
  const int n_nodes = a.n_nodes;
  int resume = live ? 0 : n_nodes;
  uint32_t n_terms = 0;                       // terms of this lane's target so far
  const uint32_t base = (EMIT && live) ? off[t] : 0u;
  int i = 0;
#ifdef NB_WALK_TIMING
  long long tw0 = wall_clock64(), t_leaf = 0, t_node = 0;
#endif
  while (i < n_nodes) {  // i is wave-uniform
#ifdef NB_WALK_TIMING
    const long long ts = wall_clock64();
#endif
    const int4 l = lk[i];    // the three records of a node are fetched together: one latency per step, not two
    const float4 b = g0[i];  // lo.x lo.y hi.x hi.y
    const float4 c = g1[i];  // cog.x cog.y mass s2
    const bool act = resume <= i;
    int next;
    if (l.w) {  // Leaf arm, main.rs:351-363: every particle of the slice, in slice order
      if (EMIT) {
        unsigned long long mask = __builtin_amdgcn_ballot_w64(act);
        for (int k0 = 0; k0 < l.z; k0 += 64) {  // 64 particles at a time, lane = particle
          const int mine = k0 + lane;
          float2 q = make_float2(0.f, 0.f);
          float m = 0.f;
          if (mine < l.z) {
            q = lpos[l.y + mine];
            m = lmass[l.y + mine];
          }
          unsigned long long todo = mask;
          while (todo) {  // one acting target after the other
            const int tl = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float tx = lane_f(p.x, tl), ty = lane_f(p.y, tl);
            const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)(base + n_terms), tl) + (uint32_t)k0;
            if (mine < l.z) terms[dst + lane] = term_of<FAST>(tx, ty, q.x, q.y, m, clamp);
          }
        }
      }
      if (act) {
        n_terms += (uint32_t)l.z;
        resume = l.x;
      }
      next = l.x;
    } else {
      bool descend = false;
      if (act) {
        const bool contains = p.y > b.y && p.x > b.x && p.x < b.z && p.y < b.w;  // bvh_tree.rs:15-20 (all strict)
        const float ddx = p.x - c.x, ddy = p.y - c.y;                              // dist2(p, cog), main.rs:228-232
        const float d2 = ddx * ddx + ddy * ddy;
        if (!contains && c.w < d2 * theta * theta) {                               // :370-372
          if (EMIT) terms[base + n_terms] = term_of<FAST>(p.x, p.y, c.x, c.y, c.z, clamp);  // :374-379
          ++n_terms;
          resume = l.x;
        } else {
          descend = true;                                                          // :381-382
          resume = i + 1;
        }
      }
      next = __builtin_amdgcn_ballot_w64(descend) != 0 ? i + 1 : l.x;
    }
    i = __builtin_amdgcn_readfirstlane(next);
#ifdef NB_WALK_TIMING
    if (l.w) t_leaf += wall_clock64() - ts; else t_node += wall_clock64() - ts;
#endif
  }
#ifdef NB_WALK_TIMING
  if (EMIT && lane == 0) {  // longest wave: total us << 20 | leaf us << 10 | node us
    const long long tot = wall_clock64() - tw0;
    atomicMax(const_cast<int*>(info) + 5, (int)(((tot / 100) << 20) | (((t_leaf / 100) & 1023) << 10) | ((t_node / 100) & 1023)));
  }
#endif
  if (!EMIT && live) cnt[t] = n_terms;
}

// total = off[n-1] + cnt[n-1]; flag what does not fit
__global__ void walk_total(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, int64_t n, int64_t capacity,
                           int* __restrict__ info) {
  const unsigned long long total = n > 0 ? (unsigned long long)off[n - 1] + cnt[n - 1] : 0ull;
  // (the scan is 32 bits wide: walk_check_wrap has flagged a wrapped sum already)
  info[0] = (int)(total > 0x7fffffffull ? 0x7fffffffull : total);
  if (total > (unsigned long long)capacity) info[1] = 1;
  // the term pass' budget per wave: enough for a dozen average targets (each wave repeats the traversal: where every
  // target is heavy, few targets per wave only multiply that), never less than kTermBudget; a power of two
  unsigned long long want = n > 0 ? kBudgetTargets * total / (unsigned long long)n : 0ull;
  uint32_t budget = kTermBudget;
  while (budget < want && budget < (1u << 30)) budget <<= 1;
  info[3] = (int)budget;
}
#endif  // NBODY_LAB (walk_pass, walk_total)
__global__ __launch_bounds__(256) void walk_check_wrap(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, int64_t n,
                                                       int* __restrict__ info) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i + 1 < n && (unsigned long long)off[i] + cnt[i] != (unsigned long long)off[i + 1]) {  // the 32-bit scan wrapped
    info[1] = 1;
    info[2] = 1;
  }
}

#ifdef NBODY_LAB
template <int K> __device__ __forceinline__ void add_row_lane(float& s, float v) {
  asm volatile("v_add_f32_dpp %0, %1, %0 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(v), "n"(K));
}

// acc[target] = ((+0 + t0) + t1) + ... in order; a row of 16 lanes per target, 16 terms per round.
__global__ __launch_bounds__(256) void walk_sum(const WalkArgs<float> a, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                                                const float2* __restrict__ terms, const int* __restrict__ info) {
  if (info[1] != 0) return;
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int64_t t = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool live = t < a.n_tgt;
  const uint32_t n = live ? cnt[t] : 0u;
  const uint32_t base = live ? off[t] : 0u;
  uint32_t nmax = n;  // the wave runs as long as its longest target
  for (int d = 32; d >= 16; d >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)nmax, d, 64);
    nmax = o > nmax ? o : nmax;
  }
  float sx = 0.f, sy = 0.f;  // Vec2::zero(), main.rs:409
  // 256 terms per round: sixteen loads in flight per lane, then 512 dependent adds (16 terms add in ~0.06 us, a
  // load takes ~2 us: the longest target, not the bandwidth, sets this kernel's time)
  constexpr int R = 16;
  for (uint32_t p = 0; p < nmax; p += 16 * R) {
    float2 q[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t k = p + 16u * j + (uint32_t)sub;
      q[j] = k < n ? terms[base + k] : make_float2(-0.0f, -0.0f);  // past the end: the identity of addition
    }
    asm volatile("s_nop 1" ::: "memory");  // q may come from a VALU move: 2 wait states before a DPP read
#define NB_ADD(K) add_row_lane<K>(sx, q[j].x); add_row_lane<K>(sy, q[j].y);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      NB_ADD(0) NB_ADD(1) NB_ADD(2) NB_ADD(3) NB_ADD(4) NB_ADD(5) NB_ADD(6) NB_ADD(7)
      NB_ADD(8) NB_ADD(9) NB_ADD(10) NB_ADD(11) NB_ADD(12) NB_ADD(13) NB_ADD(14) NB_ADD(15)
    }
#undef NB_ADD
  }
  if (live && sub == 0) {
    const int64_t row = a.tgt_index ? (int64_t)a.tgt_index[t] : t;
    reinterpret_cast<float2*>(a.acc)[row] = make_float2(sx, sy);
  }
}


#endif  // NBODY_LAB (walk_sum)

// ---- the walk in ONE pass, terms through LDS (walk_tile) --------------------------------------------------------------
// Same idea as the three passes above - a leaf's terms are evaluated lane = particle, so a leaf step costs what its
// takers cost - but the terms never leave the CU: up to TT acting targets of the wave are evaluated against the leaf
// (one round each, a row of the wave's LDS tile), then every one of those targets' own lanes adds its row in slice
// order (lane = target again: TT independent chains at once).  No term array (8 B per pair written and read back), no
// sum pass, no capacity to outgrow.  Waves are still cut by work; the estimate is the scan of the term counts the
// targets' particles (by id: the build permutes the rows) had in the PREVIOUS walk or, when there is none, of a
// counting traversal.  A bad estimate costs balance, never correctness: each target's additions are the fused
// walk's, in its order.  f32 and f64 (rows of 16-byte terms: half as many waves stay resident).
// Which targets are wave w's: those with g(t) = off[t] / budget + t / 64 == w (g is non-decreasing): [t0, t1).  A 64-ARY search —
// every lane probes one point of the range, a ballot finds the first that has reached w — narrows 64-fold per round trip: three
// rounds for 151 405 targets and ONE for the second bound (t1 <= t0 + 64), instead of the forty dependent probes of two binary
// searches (16 us of every wave's start, on the scalar side; 125 us as vector loads with a division each before that).
// off / budget as a multiply-high by M = floor((2^32 - 1) / budget): monotone in `off`, never above the true quotient, the same
// integer for every wave — all that g needs.  The budget is then ANY integer (round 4: a power of two left the wave count anywhere
// between the aim and half of it, and the walk's time follows the wave count: profiles/r04_walk_wave_target.txt).
__device__ __forceinline__ int off_quot(uint32_t off, uint32_t M) { return (int)__umulhi(off, M); }
// the first target t with g(t) >= wave (n_tgt if none): the whole wave calls it
[[maybe_unused]] __device__ __forceinline__ int first_target_reaching(const uint32_t* __restrict__ off, const int n_tgt, const int wave, const uint32_t M, const int lane) {
  int lo = 0, hi = n_tgt;
  while (lo < hi) {
    const int step = (hi - lo + 63) >> 6;
    const int idx = lo + lane * step;
    const bool reached = idx >= hi || off_quot(off[idx], M) + (idx >> 6) >= wave;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(reached);
    const int first = m ? __builtin_ctzll(m) : 64;
    if (first == 0) { hi = lo; break; }
    const int below = lo + (first - 1) * step;
    if (first < 64) hi = min(hi, lo + first * step);
    lo = below + 1;
  }
  return lo;
}
__device__ __forceinline__ void wave_targets(const uint32_t* __restrict__ off, const int n_tgt, const int wave, const uint32_t M, const int lane,
                                             int& t0, int& t1) {
  int lo = 0, hi = n_tgt;  // the first t with g(t) >= wave lies in [lo, hi] (hi = "none below hi")
  while (lo < hi) {
    const int step = (hi - lo + 63) >> 6;
    const int idx = lo + lane * step;
    const bool reached = idx >= hi || off_quot(off[idx], M) + (idx >> 6) >= wave;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(reached);
    const int first = m ? __builtin_ctzll(m) : 64;  // (lane 0 probes lo itself)
    if (first == 0) { hi = lo; break; }
    const int below = lo + (first - 1) * step;      // the last probe that has not reached `wave`
    if (first < 64) hi = min(hi, lo + first * step);
    lo = below + 1;
  }
  t0 = lo;
  const int idx = t0 + lane;                        // the first t with g(t) > wave: at most 64 further on
  const bool past = idx >= n_tgt || off_quot(off[idx], M) + (idx >> 6) > wave;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(past);
  t1 = t0 + (m ? __builtin_ctzll(m) : 64);
  if (t1 > n_tgt) t1 = n_tgt;
}

template <class T, bool FAST, int TT, bool SREC = false>
__global__ __launch_bounds__(256) void walk_tile(const WalkArgs<T> a, const uint32_t* __restrict__ off, const int* __restrict__ info,
                                                 const uint32_t* __restrict__ tgt_ids, uint32_t* __restrict__ hist,
                                                 unsigned long long* __restrict__ total_out) {
  using T2 = typename Vec2Of<T>::type;
  using T4 = typename Vec4Of<T>::type;
  constexpr int kStride = 65;  // terms per row + 1: rows of different targets start in different banks
  __shared__ T2 tile_all[4][TT * kStride];
  const int lane = threadIdx.x & 63;
  T2* __restrict__ tile = tile_all[threadIdx.x >> 6];
  const int wave = __builtin_amdgcn_readfirstlane((int)(group_of_block(a, blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6)));
  if (info[1] != 0) return;  // the estimate's scan wrapped: the caller takes the fused walk
  if (wave > info[5]) return;  // past the last wave that can hold a target (most of the grid on a small scene): no searches
  // first t with g(t) >= wave, g(t) = off[t] / budget + t / 64 (see walk_pass), then the first with g(t) > wave: all on the
  // scalar side (`off` through the constant address space; the budget is a power of two, tile_total)
  const uint32_t qmul = 0xFFFFFFFFu / (uint32_t)__builtin_amdgcn_readfirstlane(info[3]);  // (budget >= 64)
  const int n_tgt = (int)a.n_tgt;  // (the scan, hence the walk, is 32 bits wide)
  int t0, lo;
  wave_targets(off, n_tgt, wave, qmul, lane, t0, lo);
  if (lo == t0) return;
  const int64_t t = (int64_t)t0 + lane;
  const bool live = t < lo;
  const int64_t row = live ? (a.tgt_index ? (int64_t)a.tgt_index[t] : t) : 0;
  const T2 p = live ? reinterpret_cast<const T2*>(a.tgt_pos)[row] : T2{0, 0};
  const T4* __restrict__ g0 = reinterpret_cast<const T4*>(a.geom0);
  const T4* __restrict__ g1 = reinterpret_cast<const T4*>(a.geom1);
  const int4* __restrict__ lk = reinterpret_cast<const int4*>(a.link);
  const T2* __restrict__ lpos = reinterpret_cast<const T2*>(a.leaf_pos);
  const T* __restrict__ lmass = a.leaf_mass;
  const T theta = a.theta, clamp = a.clamp;
  const int n_nodes = a.n_nodes_dev ? __builtin_amdgcn_readfirstlane(*a.n_nodes_dev) : a.n_nodes;
  int resume = live ? 0 : n_nodes;
  uint32_t n_terms = 0;
  T ax = 0, ay = 0;  // Vec2::zero(), main.rs:409
  int i = 0;
  // The particles [first, first + count), in order, for the lanes of `mask` (`act`: this lane is one of them): main.rs:351-363.
  auto leaf_rows = [&](const unsigned long long mask, const bool act, const int first, const int count) {
    {
      {
        const int takers = __builtin_popcountll(mask);
        for (int k0 = 0; k0 < count; k0 += 64) {  // 64 particles at a time
          const int mine = k0 + lane;
          const int left = count - k0;
          const int rounds8 = ((left < 64 ? left : 64) + 7) >> 3;
          T2 q = T2{0, 0};
          T m = 0;
          if (mine < count) {
            q = lpos[first + mine];
            m = lmass[first + mine];
          }
          if (takers * kTileRoundCost > (left < 64 ? left : 64) * kFusedPairCost) {
            // most of the wave wants this leaf: lane = target, the particles one after the other (the fused walk's
            // arm; the same additions in the same order, so the two arms mix freely)
            const int mc = left < 64 ? left : 64;
            for (int j = 0; j < mc; ++j) {
              const T qx = lane_t(q.x, j), qy = lane_t(q.y, j), qm = lane_t(m, j);
              if (act) {
                const T2 term = term_of<FAST>(p.x, p.y, qx, qy, qm, clamp);
                ax = ax + term.x;
                ay = ay + term.y;
              }
            }
            continue;
          }
          unsigned long long todo = mask;
          const bool valid = mine < count;
          // the acting targets take the rows in lane order, TT per batch: a target's row is its rank among the acting lanes
          const int rank = act ? (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u)) : -1;
          int batch0 = 0;
          while (todo) {
            int slot = 0;
            while (todo && slot < TT) {  // lane = particle: one acting target per round, its terms into row `slot`
              if constexpr (!FAST && TT >= 2) {
                if (slot + 2 <= TT && (todo & (todo - 1)) != 0) {  // two acting targets are there: their rounds as one basic block
                  const int ta = __builtin_ctzll(todo);
                  todo &= todo - 1;
                  const int tb = __builtin_ctzll(todo);
                  todo &= todo - 1;
                  const T2 ra = pair_term_sel(valid, lane_t(p.x, ta), lane_t(p.y, ta), q.x, q.y, m, clamp);
                  const T2 rb = pair_term_sel(valid, lane_t(p.x, tb), lane_t(p.y, tb), q.x, q.y, m, clamp);
                  tile[slot * kStride + lane] = ra;
                  tile[(slot + 1) * kStride + lane] = rb;
                  slot += 2;
                  continue;
                }
              }
              const int tl = __builtin_ctzll(todo);
              todo &= todo - 1;
              const T tx = lane_t(p.x, tl), ty = lane_t(p.y, tl);
              if constexpr (FAST) {
                const T2 term = term_of<FAST>(tx, ty, q.x, q.y, m, clamp);
                tile[slot * kStride + lane] = valid ? term : neg_zero2<T>();  // past the leaf: the identity of addition
              } else {
                tile[slot * kStride + lane] = pair_term_if(valid, tx, ty, q.x, q.y, m, clamp);
              }
              ++slot;
            }
            const int myslot = (rank >= batch0 && rank < batch0 + slot) ? rank - batch0 : -1;
            batch0 += slot;
            wave_lds_handoff();
            if (myslot >= 0) {  // lane = target: its row, in slice order
              const T2* __restrict__ r = tile + myslot * kStride;
              if constexpr (sizeof(T) == 4) {
                // eight terms at a time, the next eight on their way from LDS while these are added (two register sets
                // taken in turn; left to itself the compiler reads a batch, waits, adds, and reads the next)
#define NB_ROW_LOAD(dst, blk) _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dst[j_] = r[(blk) * 8 + j_];
#define NB_ROW_ADD(src) _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) { ax = ax + src[j_].x; ay = ay + src[j_].y; } \
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                T2 va[8], vb[8];
                NB_ROW_LOAD(va, 0)
                int j0 = 0;
                for (; j0 + 3 <= rounds8; j0 += 2) {  // va holds block j0 here
                  NB_ROW_LOAD(vb, j0 + 1)
                  NB_ROW_ADD(va)
                  NB_ROW_LOAD(va, j0 + 2)
                  NB_ROW_ADD(vb)
                }
                if (j0 + 2 <= rounds8) {
                  NB_ROW_LOAD(vb, j0 + 1)
                  NB_ROW_ADD(va)
                  NB_ROW_ADD(vb)
                } else {
                  NB_ROW_ADD(va)
                }
#undef NB_ROW_LOAD
#undef NB_ROW_ADD
              } else {
                for (int j0 = 0; j0 < rounds8; ++j0) {
                  T2 v[8];
#pragma unroll
                  for (int j = 0; j < 8; ++j) v[j] = r[j0 * 8 + j];
#pragma unroll
                  for (int j = 0; j < 8; ++j) {
                    ax = ax + v[j].x;
                    ay = ay + v[j].y;
                  }
                }
              }
            }
            wave_lds_handoff();
          }
        }
      }
    }
  };
  while (i < n_nodes) {  // i is wave-uniform
    const NodeRec<T> rec = SREC ? scalar_node_rec<T>(a.link, a.geom0, a.geom1, i) : NodeRec<T>{lk[i], g0[i], g1[i]};
    const int4 l = rec.l;
    const T4 b = rec.b;
    const T4 c = rec.c;
#ifndef NB_TILE_LATE_GEOM
    asm volatile("" : : "s"(b.x), "s"(c.w));  // the three records together: one latency per step (the compiler sinks the two
                                              // it needs in the node arm only into that arm, behind the first one's wait)
#endif
    const bool act = resume <= i;
    int next;
    if (l.w) {  // Leaf arm
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(act);
      if (mask) leaf_rows(mask, act, l.y, l.z);
      if (act) {
        n_terms += (uint32_t)l.z;
        resume = l.x;
      }
      next = l.x;
    } else {
      // The node test as straight-line code: compares and-ed as masks, selects instead of nested exec regions (what `if (act) { if
      // (...) {...} else {...} }` compiles to: four s_and_saveexec / s_or exec pairs, their copies and branches — about half of the
      // ≈ 100 instructions of a node step, which is what a wave that runs alone pays for: it issues one instruction per ≈ 8-15 cycles).
      const bool contains = (p.y > b.y) & (p.x > b.x) & (p.x < b.z) & (p.y < b.w);  // bvh_tree.rs:15-20 (all strict)
      const T ddx = p.x - c.x, ddy = p.y - c.y;                                   // dist2(p, cog), main.rs:228-232
      const T d2 = ddx * ddx + ddy * ddy;
      const bool accept = act & !contains & (c.w < d2 * theta * theta);              // :370-372
      const bool descend = act & !accept;                                            // :381-382
      if (__builtin_amdgcn_ballot_w64(accept) != 0) {  // (wave-uniform: the as-written term is two IEEE divisions)
        const T2 term = term_of<FAST>(p.x, p.y, c.x, c.y, c.z, clamp);               // :374-379
        const T nax = ax + term.x, nay = ay + term.y;
        ax = accept ? nax : ax;
        ay = accept ? nay : ay;
      }
      n_terms += accept ? 1u : 0u;
      resume = accept ? l.x : (descend ? i + 1 : resume);
      const unsigned long long dmask = __builtin_amdgcn_ballot_w64(descend);
      if (l.x - i == 3) {
        // A subtree of three nodes: both children are leaves and a lane that descends takes both, whole (main.rs:381-382:
        // children[0] then children[1]) — their slices, one after the other, ARE this node's own range [first, first +
        // count) (the partition keeps a node's particles together, left child first; the record of an inner node carries
        // its range too).  So the two leaf steps happen here: the same pairs for the same lanes in the same order, two
        // records and one round trip to the particles fewer per pair of leaves.
        if (dmask) leaf_rows(dmask, descend, l.y, l.z);
        if (descend) {
          n_terms += (uint32_t)l.z;
          resume = l.x;
        }
        next = l.x;
      } else {
        next = dmask != 0 ? i + 1 : l.x;
      }
    }
    i = __builtin_amdgcn_readfirstlane(next);
  }
  if (live) {
    reinterpret_cast<T2*>(a.acc)[row] = T2{ax, ay};
    if (hist) hist[tgt_ids[t]] = n_terms;  // by particle id: the rows are permuted by every build
  }
  unsigned long long sum = live ? n_terms : 0ull;  // what this walk cost, for the next estimate's scale
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, d, 64);
  if (lane == 0) atomicAdd(total_out, sum);
}

// ---- the one-pass walk under the TOLERANCE contract (nbody_arith FAST) --------------------------------------------------
// north_star asks bit parity of the tree INDEXING and a tolerance on the forces.  walk_tile above pays for bit parity of the
// sums as well: a row of LDS per target, a fenced hand-off, 64 dependent adds per target and leaf, two IEEE divisions per
// pair (76 VALU instructions per (target, leaf) round).  walk_tile_fast keeps the traversal — same node tests, same
// interaction lists, so nbody_tree_walk_stats and the history are the exact walk's — and spends the freedom:
//   * a pair costs one v_rcp (pair_term_fast: the direct kernel's arithmetic and tolerance) and lands in an FMA;
//   * lane = particle rounds keep their 64 terms in registers; eight targets' rows are summed TOGETHER by a transposed
//     reduction: v_permlane32_swap + add (8 -> 4 values, each half-wave another target), v_permlane16_swap + add (4 -> 2,
//     each row of 16 lanes another target), a DPP rotate by 8 lanes + add under a bank mask (2 -> 1), three DPP adds inside
//     8 lanes: 36 instructions per coordinate pair for eight targets, no LDS, no fence, no chain;  each target's lane
//     fetches its total with one ds_bpermute per coordinate;
//   * where most of the wave wants the leaf the particles are broadcast instead (lane = target), as before.
// Any order of additions is inside the tolerance (tests/_tol.py: 2e-5 of sum |term|; a tree of 64 + one add per leaf is
// a better-conditioned sum than the reference's sequential chain).
constexpr int kFastRoundCost = NB_FAST_ROUND_COST;  // VALU instructions a target costs at a leaf, lane = particle (round + its share of the reduction)
constexpr int kFastPairCost = 14;   // ... and a particle costs the wave, lane = target (three broadcasts, the pair, two FMAs)

// v_permlane32_swap / v_permlane16_swap (new in gfx950) through inline asm: this compiler's __builtin_amdgcn_permlane32_swap
// hands back element 0 of the result pair twice (its lowering extracts value 0 for both halves), so the second register
// of the swap is lost.  The s_nop covers "VALU writes a VGPR, a permlane swap reads it: two wait states", which the
// compiler inserts for its own instructions and cannot see inside an asm; what reads the results next is a plain add.
//   swap_halves: a <- {a.lo, b.lo}, b <- {a.hi, b.hi} (halves of 32 lanes)
//   swap_rows:   a <- {a.r0, b.r0, a.r2, b.r2}, b <- {a.r1, b.r1, a.r3, b.r3} (rows of 16 lanes)
__device__ __forceinline__ void swap_halves(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap_rows(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
[[maybe_unused]] __device__ __forceinline__ void swap_halves(double& a, double& b) {
  unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
  unsigned al = (unsigned)ua, ah = (unsigned)(ua >> 32), bl = (unsigned)ub, bh = (unsigned)(ub >> 32);
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3" : "+v"(al), "+v"(ah), "+v"(bl), "+v"(bh));
  a = __builtin_bit_cast(double, ((unsigned long long)ah << 32) | al);
  b = __builtin_bit_cast(double, ((unsigned long long)bh << 32) | bl);
}
[[maybe_unused]] __device__ __forceinline__ void swap_rows(double& a, double& b) {
  unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
  unsigned al = (unsigned)ua, ah = (unsigned)(ua >> 32), bl = (unsigned)ub, bh = (unsigned)(ub >> 32);
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3" : "+v"(al), "+v"(ah), "+v"(bl), "+v"(bh));
  a = __builtin_bit_cast(double, ((unsigned long long)ah << 32) | al);
  b = __builtin_bit_cast(double, ((unsigned long long)bh << 32) | bl);
}
// v as the DPP control CTRL moves it (every source lane lies inside the row: nothing is out of range)
template <int CTRL, int BANKS = 0xf> __device__ __forceinline__ float dpp_of(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANKS, BANKS == 0xf));
}
template <int CTRL, int BANKS = 0xf> __device__ __forceinline__ double dpp_of(double old, double v) {
  const unsigned long long uo = __builtin_bit_cast(unsigned long long, old), uv = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)uo, (int)(unsigned)uv, CTRL, 0xf, BANKS, BANKS == 0xf);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(uo >> 32), (int)(unsigned)(uv >> 32), CTRL, 0xf, BANKS, BANKS == 0xf);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
constexpr int kDppRor8 = 0x128, kDppHalfMirror = 0x141, kDppQuad1032 = 0xB1, kDppQuad2301 = 0x4E, kDppQuadIdentity = 0xE4;
// the sum over 8 consecutive lanes, in every one of them
template <class T> __device__ __forceinline__ T sum_of_8_lanes(T r) {
  r = r + dpp_of<kDppHalfMirror>((T)0, r);
  r = r + dpp_of<kDppQuad1032>((T)0, r);
  r = r + dpp_of<kDppQuad2301>((T)0, r);
  return r;
}
// Eight values per lane -> the 64-lane total of value k in the 8 lanes from kSlotLane8(k) on.
__device__ __forceinline__ int slot_lane8(int k) { return 16 * (((k & 1) << 1) | ((k >> 1) & 1)) + 8 * (k >> 2); }  // row {0,2,1,3}[k & 3], half-row k >> 2
template <class T> __device__ __forceinline__ T reduce8(T (&v)[8]) {
  T p[4], q[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // half h of p[j]: value 2j + h summed over the two halves
    swap_halves(v[2 * j], v[2 * j + 1]);
    p[j] = v[2 * j] + v[2 * j + 1];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // row r of q[i]: value 4i + {0,2,1,3}[r] summed over four rows
    swap_rows(p[2 * i], p[2 * i + 1]);
    q[i] = p[2 * i] + p[2 * i + 1];
  }
  const T a = q[0] + dpp_of<kDppRor8>((T)0, q[0]);   // lanes i and i ^ 8 of a row added
  const T b = q[1] + dpp_of<kDppRor8>((T)0, q[1]);
  const T r = dpp_of<kDppQuadIdentity, 0xc>(a, b);   // lanes 0-7 of every row keep a (values 0-3), lanes 8-15 take b (values 4-7)
  return sum_of_8_lanes(r);
}
// Four values per lane -> the total of value k in the 16 lanes of row {0,2,1,3}[k].
template <class T> __device__ __forceinline__ T reduce4(T (&v)[4]) {
  swap_halves(v[0], v[1]);
  swap_halves(v[2], v[3]);
  T w0 = v[0] + v[1], w1 = v[2] + v[3];
  swap_rows(w0, w1);
  T u = w0 + w1;
  u = u + dpp_of<kDppRor8>((T)0, u);
  return sum_of_8_lanes(u);
}
// Two values per lane (a target's x and y terms) -> the 64-lane total of x in lanes 16-31, of y in lanes 48-63: one swap of
// halves, four DPP adds inside the rows, one row broadcast (lane 15 of rows 0 and 2 into rows 1 and 3).
constexpr int kDppRowBcast15 = 0x142;
template <class T> __device__ __forceinline__ T row_bcast15_odd_rows(T v) {  // rows 1 and 3: lane 15 of the row before; rows 0 and 2: zero
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kDppRowBcast15, 0xa, 0xf, false));
  } else {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, kDppRowBcast15, 0xa, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), kDppRowBcast15, 0xa, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
}
template <class T> __device__ __forceinline__ T reduce2(T x, T y) {
  swap_halves(x, y);
  T s = x + y;  // lanes 0-31: x.lo + x.hi, lanes 32-63: y.lo + y.hi
  s = s + dpp_of<kDppRor8>((T)0, s);
  s = sum_of_8_lanes(s);
  return s + row_bcast15_odd_rows(s);
}
template <class T> __device__ __forceinline__ T lane_fetch(T v, int src_lane) {  // v of lane src_lane (per-lane index)
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
  } else {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
}
// pair_term_fast's scale factor: term = d * scale (force 0 makes the term an exact zero: lanes past a leaf's end)
__device__ __forceinline__ float fast_scale(float dx, float dy, float force, float clamp) {
  const float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);
  const float d2 = __builtin_fmaxf(__builtin_fmaf(dy, dy, dx * dx), clamp);
  return force * __builtin_amdgcn_rcpf(__builtin_fmaf(sum, d2, 8.0779356694631609e-28f));  // 2^-90
}
[[maybe_unused]] __device__ __forceinline__ double fast_scale(double dx, double dy, double force, double clamp) {
  const double sum = __builtin_fabs(dx) + __builtin_fabs(dy);
  const double d2 = __builtin_fmax(__builtin_fma(dy, dy, dx * dx), clamp);
  const double den = __builtin_fma(sum, d2, 0x1p-700);
  double r = __builtin_amdgcn_rcp(den);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  return force * r;
}
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
[[maybe_unused]] __device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <class T, int REC, bool LOG>  // REC: how the node records are fetched (0 plain loads: the compiler picks scalar loads; 1 vector loads); LOG: per-wave log (development)
__global__ __launch_bounds__(256) void walk_tile_fast(const WalkArgs<T> a, const uint32_t* __restrict__ off, const int* __restrict__ info,
                                                      const uint32_t* __restrict__ tgt_ids, uint32_t* __restrict__ hist,
                                                      unsigned long long* __restrict__ total_out) {
  using T2 = typename Vec2Of<T>::type;
  using T4 = typename Vec4Of<T>::type;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(group_of_block(a, blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6)));
  if (info[1] != 0) return;  // the estimate's scan wrapped: the caller walks again without one
  if (wave > info[5]) return;  // past the last wave that can hold a target (most of the grid on a small scene): no searches
  const long long log_t0 = LOG ? wall_clock64() : 0;
  // Which targets are this wave's: those with g(t) = off[t] / budget + t / 64 == wave (see walk_pass), by two binary searches.
  // Everything in them is wave-uniform, and kept on the scalar side on purpose: `off` is read through the constant address
  // space (s_load: it was written by kernels before this one) and the budget is a power of two (tile_total), so the
  // quotient is a shift — forty dependent steps that cost a wave 125 us as vector loads and a 32-bit division each, on
  // SIMDs whose vector pipes the other waves keep busy (profiles/r03_walk_wave_log.txt).
  const uint32_t qmul = 0xFFFFFFFFu / (uint32_t)__builtin_amdgcn_readfirstlane(info[3]);  // (budget >= 64)
  const int n_tgt = (int)a.n_tgt;  // (the scan, hence the walk, is 32 bits wide)
  int t0, lo;
  wave_targets(off, n_tgt, wave, qmul, lane, t0, lo);
  const long long log_t1 = LOG ? wall_clock64() : 0;  // the search is over (its ballots waited for its loads)
  if (lo == t0) return;
  const int64_t t = (int64_t)t0 + lane;
  const bool live = t < lo;
  const int64_t row = live ? (a.tgt_index ? (int64_t)a.tgt_index[t] : t) : 0;
  const T2 p = live ? reinterpret_cast<const T2*>(a.tgt_pos)[row] : T2{0, 0};
  const T4* __restrict__ g0 = reinterpret_cast<const T4*>(a.geom0);
  const T4* __restrict__ g1 = reinterpret_cast<const T4*>(a.geom1);
  const int4* __restrict__ lk = reinterpret_cast<const int4*>(a.link);
  const T2* __restrict__ lpos = reinterpret_cast<const T2*>(a.leaf_pos);
  const T* __restrict__ lmass = a.leaf_mass;
  const T theta = a.theta, clamp = a.clamp;
  const int n_nodes = a.n_nodes_dev ? __builtin_amdgcn_readfirstlane(*a.n_nodes_dev) : a.n_nodes;
  int resume = live ? 0 : n_nodes;
  uint32_t n_terms = 0;
  // Two-level summation, as in the direct kernel: the terms go to a block sum (bx, by) that joins the running total after every fourth
  // leaf step.  A target's list can be a large part of all particles (small theta on the BVH's needle boxes; all-negative
  // coordinates, whose boxes stretch to the origin: bvh_tree.rs:42) and a plain f32 chain of N additions drifts by ~sqrt(N)
  // half-ulps of the sum of magnitudes: 2.2e-5 ... 4.9e-5 of it on lists of 2 x 10^4 ... 10^5 terms (found by the extended fuzz;
  // the contract is 2e-5).
  T ax = 0, ay = 0, bx = 0, by = 0;
  int leaf_steps = 0;
  auto flush_if_due = [&]() {
    if (++leaf_steps == 4) {
      ax = ax + bx;
      ay = ay + by;
      bx = by = 0;
      leaf_steps = 0;
    }
  };
  int i = 0;
  unsigned log_nodes = 0, log_leaves = 0, log_rounds = 0;
  // A node's three records (link, box, centre of gravity | mass | s^2), fetched together: one latency per step.  REC = 1
  // fetches them by VECTOR loads of one address (the offset passes through a register the compiler cannot see through, or
  // it would pick scalar loads): the records then sit in VGPRs, where the node test's operands cost half of what SGPR
  // operands cost (measured cost model).  Measured equal within 2 % (profiles/r03_walk_fast_variants.txt);
  // fetching one node AHEAD into a second register set made both scenes 12-15 % slower (the compiler copies the set at the
  // loop's back edge behind a full wait).
  struct Rec { int4 l; T4 b; T4 c; };
  const int last = n_nodes - 1;
  unsigned lane_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
  auto fetch = [&](int k) -> Rec {
    k = k < last ? k : last;
    if constexpr (REC == 3) {  // scalar loads through the constant address space (scalar_node_rec)
      const NodeRec<T> r = scalar_node_rec<T>(a.link, a.geom0, a.geom1, k);
      asm volatile("" : : "s"(r.b.x), "s"(r.c.w));  // the three records together, before anything branches on the first
      return Rec{r.l, r.b, r.c};
    }
    if constexpr (REC == 0) {
      const Rec r{lk[k], g0[k], g1[k]};
      asm volatile("" : : "s"(r.b.x), "s"(r.c.w));  // (as in walk_tile: without it the compiler sinks the box and the centre of gravity
      return r;                                     // into the node arm, behind the link's wait: two round trips per node step)
    }
    const unsigned o16 = (unsigned)k * 16u + lane_zero;
    const unsigned ot = (unsigned)k * (unsigned)sizeof(T4) + lane_zero;
    return Rec{*reinterpret_cast<const int4*>(reinterpret_cast<const char*>(lk) + o16), *reinterpret_cast<const T4*>(reinterpret_cast<const char*>(g0) + ot),
               *reinterpret_cast<const T4*>(reinterpret_cast<const char*>(g1) + ot)};
  };
  // The particles [first, first + count) for the lanes of `mask` (`taker`: this lane is one of them), main.rs:351-363;
  // any order of additions (tolerance contract).
  auto rounds = [&](const unsigned long long mask, const bool taker, const int first, const int count) {
    {
      {
        const bool act = taker;
        const int takers = __builtin_popcountll(mask);
        if constexpr (LOG) { ++log_leaves; log_rounds += (unsigned)takers; }
        const int rank = act ? (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u)) : -1;
        for (int k0 = 0; k0 < count; k0 += 64) {  // 64 particles at a time
          const int mine = k0 + lane;
          const int left = count - k0;
          const int mc = left < 64 ? left : 64;
          T2 q = T2{0, 0};
          T m = 0;  // a lane past the end: force 0, its terms are exact zeros
          if (mine < count) {
            q = lpos[first + mine];
            m = lmass[first + mine];
          }
          if (takers * kFastRoundCost > mc * kFastPairCost) {  // most of the wave wants this leaf: lane = target
            for (int j = 0; j < mc; ++j) {
              const T qx = lane_t(q.x, j), qy = lane_t(q.y, j), qm = lane_t(m, j);
              if (act) {
                const T dx = qx - p.x, dy = qy - p.y;
                const T sc = fast_scale(dx, dy, qm, clamp);
                bx = fma_t(dx, sc, bx);
                by = fma_t(dy, sc, by);
              }
            }
            continue;
          }
          unsigned long long todo = mask;
          int batch0 = 0;
          // one acting target's terms against the 64 particles, lane = particle
#define NB_FAST_ROUND(XV, YV)                                             \
  {                                                                       \
    const int tl = __builtin_ctzll(todo);                                 \
    todo &= todo - 1;                                                     \
    const T dx = q.x - lane_t(p.x, tl), dy = q.y - lane_t(p.y, tl);       \
    const T sc = fast_scale(dx, dy, m, clamp);                            \
    XV = dx * sc;                                                         \
    YV = dy * sc;                                                         \
  }
          while (todo) {
            const int left_t = takers - batch0;
            const int k = rank - batch0;  // this lane's place in the batch, if it is an acting target
            T gx, gy;
            int took;
            if (left_t > 4) {  // eight targets (missing ones contribute zeros): x and y reduced side by side
              T X[8], Y[8];
#pragma unroll
              for (int sl = 0; sl < 5; ++sl) NB_FAST_ROUND(X[sl], Y[sl])  // (five are there: one basic block, so the scheduler
#pragma unroll                                                            // interleaves their chains — what a wave that runs alone lives on)
              for (int sl = 5; sl < 8; ++sl) {
                X[sl] = 0;
                Y[sl] = 0;
                if (todo) NB_FAST_ROUND(X[sl], Y[sl])
              }
              const T rx = reduce8(X), ry = reduce8(Y);
              const int src = slot_lane8(k & 7);
              gx = lane_fetch(rx, src);
              gy = lane_fetch(ry, src);
              took = 8;
            } else if (left_t > 2) {  // three or four targets: their x and y are the eight values of ONE reduction
              T V[8];
#pragma unroll
              for (int sl = 0; sl < 3; ++sl) NB_FAST_ROUND(V[2 * sl], V[2 * sl + 1])  // (three are there)
              V[6] = 0;
              V[7] = 0;
              if (todo) NB_FAST_ROUND(V[6], V[7])
              const T r = reduce8(V);
              const int src = 16 * (k & 1) + 8 * ((k >> 1) & 1);  // slot_lane8(2k); y sits 32 lanes on (slot_lane8(2k + 1))
              gx = lane_fetch(r, src);
              gy = lane_fetch(r, src + 32);
              took = 4;
            } else if (left_t == 2) {  // two targets: four values
              T V[4];
              NB_FAST_ROUND(V[0], V[1])
              NB_FAST_ROUND(V[2], V[3])
              const T r = reduce4(V);  // x0 row 0, y0 row 2, x1 row 1, y1 row 3
              const int src = 16 * (k & 1);
              gx = lane_fetch(r, src);
              gy = lane_fetch(r, src + 32);
              took = 2;
            } else {  // one target
              T x, y;
              NB_FAST_ROUND(x, y)
              const T r = reduce2(x, y);
              gx = lane_t(r, 16);
              gy = lane_t(r, 48);
              took = 1;
            }
            if (k >= 0 && k < took) {
              bx = bx + gx;
              by = by + gy;
            }
            batch0 += took;
          }
#undef NB_FAST_ROUND
        }
      }
    }
  };
  auto step = [&](const Rec& rec, const int i) -> int {
    const int4 l = rec.l;
    const T4 b = rec.b;
    const T4 c = rec.c;
    const bool act = resume <= i;
    int next;
    if (l.w) {  // Leaf arm
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(act);
      if (mask) {
        rounds(mask, act, l.y, l.z);
        flush_if_due();
      }
      if (act) {
        n_terms += (uint32_t)l.z;
        resume = l.x;
      }
      next = l.x;
    } else {
      if constexpr (LOG) ++log_nodes;
      // straight-line: masks and selects instead of nested exec regions (see walk_tile); the term is computed for every lane and
      // kept by the lanes that accept (a select on the RESULT: an empty node's centre of gravity is NaN, and NaN x 0 is not 0)
      const bool contains = (p.y > b.y) & (p.x > b.x) & (p.x < b.z) & (p.y < b.w);  // bvh_tree.rs:15-20 (all strict)
      const T ddx = p.x - c.x, ddy = p.y - c.y;                                   // dist2(p, cog), main.rs:228-232
      const T d2 = ddx * ddx + ddy * ddy;
      const bool accept = act & !contains & (c.w < d2 * theta * theta);              // :370-372 (the test is the exact walk's, bit for bit)
      const bool descend = act & !accept;                                            // :381-382
      const T dx = c.x - p.x, dy = c.y - p.y;                                        // :374-379
      const T sc = fast_scale(dx, dy, c.z, clamp);
      const T nbx = fma_t(dx, sc, bx), nby = fma_t(dy, sc, by);
      bx = accept ? nbx : bx;
      by = accept ? nby : by;
      n_terms += accept ? 1u : 0u;
      resume = accept ? l.x : (descend ? i + 1 : resume);
      const unsigned long long dmask = __builtin_amdgcn_ballot_w64(descend);
      if (l.x - i == 3) {
        // A subtree of three nodes: both children are leaves, and a lane that descends here takes both, whole — their
        // particles are this node's own range [first, first + count) (a node's record carries its range too).  So the two
        // leaf steps happen HERE: two records and one round trip for the particles fewer per pair of leaves, which is most of
        // what a wave waits for (the records of three leaves in four are never fetched).  Same pairs, same lanes.
        if (dmask) {
          rounds(dmask, descend, l.y, l.z);
          flush_if_due();
        }
        if (descend) {
          n_terms += (uint32_t)l.z;
          resume = l.x;
        }
        next = l.x;
      } else {
        next = dmask != 0 ? i + 1 : l.x;
      }
    }
    return __builtin_amdgcn_readfirstlane(next);
  };
  long long log_t2 = 0;
  while (i < n_nodes) {
    const Rec r = fetch(i);
    if (LOG && i == 0) {
      asm volatile("" : : "s"(r.l.x), "v"(p.x));  // the root's record and the targets have arrived
      log_t2 = wall_clock64();
    }
    i = step(r, i);
  }
  const long long log_t3 = LOG ? wall_clock64() : 0;  // the walk is over: what follows is the wave's epilogue
  ax = ax + bx;
  ay = ay + by;
  if (live) {
    reinterpret_cast<T2*>(a.acc)[row] = T2{ax, ay};
    if (hist) hist[tgt_ids[t]] = n_terms;  // by particle id: the rows are permuted by every build
  }
  unsigned long long sum = live ? n_terms : 0ull;  // what this walk cost, for the next estimate's scale
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, d, 64);
  if (lane == 0) atomicAdd(total_out, sum);
  if (LOG && lane == 0) {
    unsigned long long* o = a.wave_log + 4 * wave;
    o[0] = (((unsigned long long)log_t0 & 0xFFFFFFFFFFull) << 24) | ((unsigned long long)(wall_clock64() - log_t0) & 0xFFFFFFull);  // start (absolute, 40 bits) | duration
    o[1] = (unsigned long long)log_nodes | ((unsigned long long)((log_t1 - log_t0) & 0xFFFF) << 32) | ((unsigned long long)((log_t2 - log_t0) & 0xFFFF) << 48);  // + ticks to the end of the search | to the first record
    o[2] = (unsigned long long)log_leaves | ((unsigned long long)((wall_clock64() - log_t3) & 0xFFFF) << 32);  // + ticks of the epilogue
    o[3] = ((unsigned long long)(unsigned)(lo - t0) << 32) | log_rounds;
  }
}

#ifdef NBODY_LAB
// ---- the FAST one-pass walk, BREADTH FIRST (round 4) ------------------------------------------------------------------
// walk_tile_fast above is its longest wave's serial chain (profiles/r03_walk_fast_variants.txt: 0.355 us per node step — a scalar
// load's round trip plus ~100 dependent instructions — 414 node steps in the reference scene's longest wave, 273 of the kernel's
// 334 us).  The depth-first order is what makes it a chain: node i's record must arrive before anyone knows which record comes
// next.  Under the tolerance contract the ORDER of a target's terms is free, and the traversal itself never needed it: a lane acts
// at a node iff it descended through the parent, so a node's acting lanes are its parent's descend mask.  So the wave keeps a
// DEQUE of (node, 64-bit lane mask) in LDS and takes up to 64 entries at a time from its head (breadth first): 64 lanes fetch 64
// nodes' records — and the right siblings' indices, link[i + 1].x — in ONE round trip, then the entries are tested one after the
// other with the record broadcast out of registers (v_readlane: SGPR operands, no memory in the loop); accepted nodes' terms are
// taken on the spot, leaves (and three-node subtrees, as in the depth-first walk) go to a small list that is worked off after the
// batch, the next leaf's particles on their way while this one's rounds run.  Same node tests, same interaction lists, same
// terms as walk_tile_fast (nbody_tree_walk_stats and the history are unchanged); only the order of additions differs, and it is
// a fixed function of the inputs (no atomics): bitwise reproducible.
// MEASURED (profiles/r04_walk_bfs_ab.txt): correct (every FAST parity test green) and SLOWER — 0.579 against 0.338 ms on the
// reference scene, 4.96 against 2.99 ms at Plummer 1 M.  The union of a wave's paths is NARROW: its 64 tree-contiguous targets
// share one chain from the root to their region, so a level holds ~4 entries, not ~64, and every level pays a vector load's round
// trip (longer than the scalar load's it replaces) plus the deque's hand-offs, at 5 waves per SIMD instead of 8 (28 KB of LDS
// per group, 81 VGPRs).  Laboratory build only (NBODY_WALK_FAST_BFS=1); the product walks depth first.
// The deque cannot outgrow its LDS: while it is nearly full the wave takes ONE entry from the TAIL instead (depth first: a
// stack grows by at most the depth of the subtree it is in, and the device builds stop at 56 levels); should it still fill
// up, the overflow word is set and the caller gets an error instead of a wrong answer.
constexpr int kBfsQ = 512;          // deque slots per wave (12 B each)
constexpr int kBfsHeadroom = 192;   // breadth first only while at least this many slots are free
constexpr int kBfsLeaves = 64;      // leaf entries per batch: one per entry taken

__device__ __forceinline__ int rl(int v, int lane_sel) { return __builtin_amdgcn_readlane(v, lane_sel); }
__device__ __forceinline__ float rlf(float v, int lane_sel) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_sel)); }

// One leaf step of the FAST walk for the lanes of `mask` (`act`: this lane is one of them) against the particles [first, first +
// count): walk_tile_fast's rounds (lane = particle, eight targets' rows reduced together; lane = target where most of the wave
// wants the leaf).  (q0, m0): the first 64 particles, already fetched by the caller.
template <class T>
__device__ __forceinline__ void fast_leaf_step(const unsigned long long mask, const bool act, const int first, const int count, const int lane,
                                               const typename Vec2Of<T>::type p, const T clamp, const typename Vec2Of<T>::type* __restrict__ lpos,
                                               const T* __restrict__ lmass, typename Vec2Of<T>::type q0, T m0, T& bx, T& by) {
  using T2 = typename Vec2Of<T>::type;
  const int takers = __builtin_popcountll(mask);
  const int rank = act ? (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u)) : -1;
  for (int k0 = 0; k0 < count; k0 += 64) {  // 64 particles at a time
    const int mine = k0 + lane;
    const int left = count - k0;
    const int mc = left < 64 ? left : 64;
    T2 q = q0;
    T m = m0;
    if (k0 > 0) {
      q = T2{0, 0};
      m = 0;  // a lane past the end: force 0, its terms are exact zeros
      if (mine < count) {
        q = lpos[first + mine];
        m = lmass[first + mine];
      }
    }
    if (takers * kFastRoundCost > mc * kFastPairCost) {  // most of the wave wants this leaf: lane = target
      for (int j = 0; j < mc; ++j) {
        const T qx = lane_t(q.x, j), qy = lane_t(q.y, j), qm = lane_t(m, j);
        if (act) {
          const T dx = qx - p.x, dy = qy - p.y;
          const T sc = fast_scale(dx, dy, qm, clamp);
          bx = fma_t(dx, sc, bx);
          by = fma_t(dy, sc, by);
        }
      }
      continue;
    }
    unsigned long long todo = mask;
    int batch0 = 0;
#define NB_FAST_ROUND(XV, YV)                                             \
  {                                                                       \
    const int tl = __builtin_ctzll(todo);                                 \
    todo &= todo - 1;                                                     \
    const T dx = q.x - lane_t(p.x, tl), dy = q.y - lane_t(p.y, tl);       \
    const T sc = fast_scale(dx, dy, m, clamp);                            \
    XV = dx * sc;                                                         \
    YV = dy * sc;                                                         \
  }
    while (todo) {
      const int left_t = takers - batch0;
      const int k = rank - batch0;  // this lane's place in the batch, if it is an acting target
      T gx, gy;
      int took;
      if (left_t > 4) {  // eight targets (missing ones contribute zeros): x and y reduced side by side
        T X[8], Y[8];
#pragma unroll
        for (int sl = 0; sl < 5; ++sl) NB_FAST_ROUND(X[sl], Y[sl])
#pragma unroll
        for (int sl = 5; sl < 8; ++sl) {
          X[sl] = 0;
          Y[sl] = 0;
          if (todo) NB_FAST_ROUND(X[sl], Y[sl])
        }
        const T rx = reduce8(X), ry = reduce8(Y);
        const int src = slot_lane8(k & 7);
        gx = lane_fetch(rx, src);
        gy = lane_fetch(ry, src);
        took = 8;
      } else if (left_t > 2) {  // three or four targets: their x and y are the eight values of ONE reduction
        T V[8];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) NB_FAST_ROUND(V[2 * sl], V[2 * sl + 1])
        V[6] = 0;
        V[7] = 0;
        if (todo) NB_FAST_ROUND(V[6], V[7])
        const T r = reduce8(V);
        const int src = 16 * (k & 1) + 8 * ((k >> 1) & 1);
        gx = lane_fetch(r, src);
        gy = lane_fetch(r, src + 32);
        took = 4;
      } else if (left_t == 2) {  // two targets: four values
        T V[4];
        NB_FAST_ROUND(V[0], V[1])
        NB_FAST_ROUND(V[2], V[3])
        const T r = reduce4(V);
        const int src = 16 * (k & 1);
        gx = lane_fetch(r, src);
        gy = lane_fetch(r, src + 32);
        took = 2;
      } else {  // one target
        T x, y;
        NB_FAST_ROUND(x, y)
        const T r = reduce2(x, y);
        gx = lane_t(r, 16);
        gy = lane_t(r, 48);
        took = 1;
      }
      if (k >= 0 && k < took) {
        bx = bx + gx;
        by = by + gy;
      }
      batch0 += took;
    }
#undef NB_FAST_ROUND
  }
}

__global__ __launch_bounds__(256) void walk_tile_fast_bfs(const WalkArgs<float> a, const uint32_t* __restrict__ off, int* __restrict__ info,
                                                          const uint32_t* __restrict__ tgt_ids, uint32_t* __restrict__ hist,
                                                          unsigned long long* __restrict__ total_out) {
  __shared__ int q_idx_all[4][kBfsQ];
  __shared__ unsigned q_lo_all[4][kBfsQ], q_hi_all[4][kBfsQ];
  __shared__ int lf_first_all[4][kBfsLeaves], lf_count_all[4][kBfsLeaves];
  __shared__ unsigned lf_lo_all[4][kBfsLeaves], lf_hi_all[4][kBfsLeaves];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  int* __restrict__ q_idx = q_idx_all[wib];
  unsigned* __restrict__ q_lo = q_lo_all[wib];
  unsigned* __restrict__ q_hi = q_hi_all[wib];
  int* __restrict__ lf_first = lf_first_all[wib];
  int* __restrict__ lf_count = lf_count_all[wib];
  unsigned* __restrict__ lf_lo = lf_lo_all[wib];
  unsigned* __restrict__ lf_hi = lf_hi_all[wib];
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wib));
  if (info[1] != 0) return;  // the estimate's scan wrapped: the caller walks again without one
  if (wave > info[5]) return;  // past the last wave that can hold a target
  const uint32_t qmul = 0xFFFFFFFFu / (uint32_t)__builtin_amdgcn_readfirstlane(info[3]);  // (budget >= 64)
  const int n_tgt = (int)a.n_tgt;
  int t0, lo;
  wave_targets(off, n_tgt, wave, qmul, lane, t0, lo);
  if (lo == t0) return;
  const int64_t t = (int64_t)t0 + lane;
  const bool live = t < lo;
  const int64_t row = live ? (a.tgt_index ? (int64_t)a.tgt_index[t] : t) : 0;
  const float2 p = live ? reinterpret_cast<const float2*>(a.tgt_pos)[row] : float2{0, 0};
  const float4* __restrict__ g0 = reinterpret_cast<const float4*>(a.geom0);
  const float4* __restrict__ g1 = reinterpret_cast<const float4*>(a.geom1);
  const int4* __restrict__ lk = reinterpret_cast<const int4*>(a.link);
  const float2* __restrict__ lpos = reinterpret_cast<const float2*>(a.leaf_pos);
  const float* __restrict__ lmass = a.leaf_mass;
  const float theta = a.theta, clamp = a.clamp;
  const int n_nodes = a.n_nodes_dev ? __builtin_amdgcn_readfirstlane(*a.n_nodes_dev) : a.n_nodes;
  uint32_t n_terms = 0;
  float ax = 0, ay = 0, bx = 0, by = 0;  // two-level summation, as walk_tile_fast: the block sum joins the total every fourth leaf step
  int leaf_steps = 0;
  const int last = n_nodes - 1;
  int head = 0, occ = 0;  // wave-uniform
  const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
  if (n_nodes > 0 && lane == 0) {
    q_idx[0] = 0;
    q_lo[0] = (unsigned)live_mask;
    q_hi[0] = (unsigned)(live_mask >> 32);
  }
  if (n_nodes > 0) occ = 1;
  bool overflow = false;
  while (occ > 0) {
    // ---- take a batch: from the head while there is room for its children (breadth first), else the newest entry alone
    int B, start;
    if (occ <= kBfsQ - kBfsHeadroom) {
      B = occ < 64 ? occ : 64;
      start = head;
      head = (head + B) & (kBfsQ - 1);
    } else {
      B = 1;
      start = (head + occ - 1) & (kBfsQ - 1);
    }
    occ -= B;
    wave_lds_handoff();  // the entries' stores (lane 0, the batch before) before these reads
    int my_idx = 0;
    unsigned my_lo = 0, my_hi = 0;
    if (lane < B) {
      const int pos = (start + lane) & (kBfsQ - 1);
      my_idx = q_idx[pos];
      my_lo = q_lo[pos];
      my_hi = q_hi[pos];
    }
    // ---- the batch's records, one round trip: link, box, centre of gravity | mass | s^2, and the right sibling's index
    my_idx = my_idx < last ? my_idx : last;
    const int4 ml = lk[my_idx];
    const float4 mb = g0[my_idx];
    const float4 mc = g1[my_idx];
    const int mr = lk[my_idx < last ? my_idx + 1 : last].x;  // skip of node i + 1 = node i's right child (inner nodes)
    int nleaf = 0;
    for (int e = 0; e < B; ++e) {
      const int i = rl(my_idx, e);
      const int lx = rl(ml.x, e), ly = rl(ml.y, e), lz = rl(ml.z, e), lw = rl(ml.w, e);
      const unsigned long long m = ((unsigned long long)(unsigned)rl((int)my_hi, e) << 32) | (unsigned)rl((int)my_lo, e);
      const bool act = (m >> lane) & 1ull;
      if (lw) {  // Leaf arm, main.rs:351-363
        if (lane == 0) {
          lf_first[nleaf] = ly;
          lf_count[nleaf] = lz;
          lf_lo[nleaf] = (unsigned)m;
          lf_hi[nleaf] = (unsigned)(m >> 32);
        }
        ++nleaf;
        n_terms += act ? (uint32_t)lz : 0u;
        continue;
      }
      const float b_x = rlf(mb.x, e), b_y = rlf(mb.y, e), b_z = rlf(mb.z, e), b_w = rlf(mb.w, e);
      const float c_x = rlf(mc.x, e), c_y = rlf(mc.y, e), c_z = rlf(mc.z, e), c_w = rlf(mc.w, e);
      // the node test is the exact walk's, bit for bit (bvh_tree.rs:15-20 all strict; main.rs:228-232, :370-372)
      const bool contains = (p.y > b_y) & (p.x > b_x) & (p.x < b_z) & (p.y < b_w);
      const float ddx = p.x - c_x, ddy = p.y - c_y;
      const float d2 = ddx * ddx + ddy * ddy;
      const bool accept = act & !contains & (c_w < d2 * theta * theta);
      const bool descend = act & !accept;
      const float dx = c_x - p.x, dy = c_y - p.y;  // :374-379
      const float sc = fast_scale(dx, dy, c_z, clamp);
      const float nbx = fma_t(dx, sc, bx), nby = fma_t(dy, sc, by);
      bx = accept ? nbx : bx;
      by = accept ? nby : by;
      n_terms += accept ? 1u : 0u;
      const unsigned long long dmask = __builtin_amdgcn_ballot_w64(descend);
      if (dmask == 0) continue;
      if (lx - i == 3) {  // both children are leaves: their particles are this node's own range (walk_tile_fast's three-node fold)
        if (lane == 0) {
          lf_first[nleaf] = ly;
          lf_count[nleaf] = lz;
          lf_lo[nleaf] = (unsigned)dmask;
          lf_hi[nleaf] = (unsigned)(dmask >> 32);
        }
        ++nleaf;
        n_terms += descend ? (uint32_t)lz : 0u;
        continue;
      }
      if (occ + 2 > kBfsQ) {  // never expected (see above): say so instead of walking on with a hole in the lists
        overflow = true;
        continue;
      }
      const int right = rl(mr, e);
      if (lane == 0) {
        const int t0q = (head + occ) & (kBfsQ - 1), t1q = (head + occ + 1) & (kBfsQ - 1);
        q_idx[t0q] = i + 1;  // children[0] then children[1], main.rs:381-382
        q_lo[t0q] = (unsigned)dmask;
        q_hi[t0q] = (unsigned)(dmask >> 32);
        q_idx[t1q] = right;
        q_lo[t1q] = (unsigned)dmask;
        q_hi[t1q] = (unsigned)(dmask >> 32);
      }
      occ += 2;
    }
    if (nleaf == 0) continue;
    // ---- the batch's leaves: the next one's particles are fetched while this one's rounds run
    wave_lds_handoff();
    int f_cur = lf_first[0], c_cur = lf_count[0];
    float2 q_cur = float2{0, 0};
    float m_cur = 0;
    if (lane < c_cur) {
      q_cur = lpos[f_cur + lane];
      m_cur = lmass[f_cur + lane];
    }
    for (int j = 0; j < nleaf; ++j) {
      const unsigned long long m = ((unsigned long long)lf_hi[j] << 32) | lf_lo[j];
      int f_nxt = 0, c_nxt = 0;
      float2 q_nxt = float2{0, 0};
      float m_nxt = 0;
      if (j + 1 < nleaf) {
        f_nxt = lf_first[j + 1];
        c_nxt = lf_count[j + 1];
        if (lane < c_nxt) {
          q_nxt = lpos[f_nxt + lane];
          m_nxt = lmass[f_nxt + lane];
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // (the loads above stay above the rounds below)
      fast_leaf_step<float>(m, (m >> lane) & 1ull, f_cur, c_cur, lane, p, clamp, lpos, lmass, q_cur, m_cur, bx, by);
      if (++leaf_steps == 4) {
        ax = ax + bx;
        ay = ay + by;
        bx = by = 0;
        leaf_steps = 0;
      }
      f_cur = f_nxt;
      c_cur = c_nxt;
      q_cur = q_nxt;
      m_cur = m_nxt;
    }
  }
  ax = ax + bx;
  ay = ay + by;
  if (overflow && lane == 0) info[1] = 1;
  if (live) {
    reinterpret_cast<float2*>(a.acc)[row] = float2{ax, ay};
    if (hist) hist[tgt_ids[t]] = n_terms;  // by particle id: the rows are permuted by every build
  }
  unsigned long long sum = live ? n_terms : 0ull;  // what this walk cost, for the next estimate's scale
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, d, 64);
  if (lane == 0) atomicAdd(total_out, sum);
}
#endif  // NBODY_LAB (walk_tile_fast_bfs)

struct EstimateOf {  // target t's terms in the last walk, scaled
  const uint32_t* hist;
  const uint32_t* ids;
  int shift;
  __host__ __device__ __forceinline__ uint32_t operator()(int t) const { return hist ? hist[ids[t]] >> shift : 0u; }
};
// the scan of a history estimate must not wrap either: the shift is sized from the last walk's total, but a target set
// that is not the last one (a shard's slice moves with the tree order) may hold counts of older walks
__global__ __launch_bounds__(256) void walk_check_wrap_est(EstimateOf est, const uint32_t* __restrict__ off, int64_t n, int* __restrict__ info) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i + 1 < n && (unsigned long long)off[i] + est((int)i) != (unsigned long long)off[i + 1]) {
    info[1] = 1;
    info[2] = 1;
  }
  if (i + 1 == n && (unsigned long long)off[i] + est((int)i) > 0x7fffffffull) info[1] = 1;  // the budget arithmetic is 31 bits wide
}
// budget of walk_tile's waves from the estimate's total (see walk_total)
// A wave's budget in (scaled) terms: total / extra_waves rounded up — g(t) = off[t] / budget + t / 64 then ends within extra_waves of the
// head count — never below kTileBudget (scaled like the estimate) or 64.
__device__ __forceinline__ uint32_t tile_budget(unsigned long long total, int64_t extra_waves, int shift) {
  const unsigned long long want = (total + (unsigned long long)extra_waves - 1) / (unsigned long long)extra_waves;
  unsigned long long budget = kTileBudget >> shift;
  if (budget < 64) budget = 64;
  if (budget < want) budget = want;
  if (budget > (1ull << 30)) budget = 1ull << 30;
  return (uint32_t)budget;
}
__device__ __forceinline__ void tile_total(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, int64_t n,
                                           const uint32_t* __restrict__ tgt_ids, const uint32_t* __restrict__ hist, int shift,
                                           int64_t extra_waves, int64_t grid_waves, int* __restrict__ info) {
  const uint32_t last = n > 0 ? (cnt ? cnt[n - 1] : (hist ? hist[tgt_ids[n - 1]] >> shift : 0u)) : 0u;
  const unsigned long long total = n > 0 ? (unsigned long long)off[n - 1] + last : 0ull;
  info[0] = (int)(total > 0x7fffffffull ? 0x7fffffffull : total);
  // g(t) = off[t] / budget + t / 64 ends near total / budget + n / 64: `extra_waves` more than the head count alone
  // (ceiling: with the floor, total / budget could reach extra_waves + extra_waves / want, past the grid)
  const uint32_t budget = tile_budget(total, extra_waves, shift);
  info[3] = (int)budget;
  info[5] = (int)(total / budget + (unsigned long long)((n > 0 ? n - 1 : 0) / 64));  // the last wave that can hold a target (g of the last one)
  // belt and braces: a wave index the grid does not hold would leave its targets unwalked — flag it like a wrapped scan,
  // the host then repeats the walk without an estimate (64 targets per wave, which always fits)
  if (total / budget + (unsigned long long)(n / 64) + 1 >= (unsigned long long)grid_waves) info[1] = 1;
  info[6] = info[7] = 0;  // this walk's total terms (unsigned long long), accumulated by walk_tile
}
__global__ void walk_tile_total(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, int64_t n, const uint32_t* __restrict__ tgt_ids,
                                const uint32_t* __restrict__ hist, int shift, int64_t extra_waves, int64_t grid_waves,
                                int* __restrict__ info) {
  tile_total(cnt, off, n, tgt_ids, hist, shift, extra_waves, grid_waves, info);
}
// The TileTail duties (walk_split.h), by the one work-group that knows the estimate's total: the walk's budget and flags
// (info[0..7] final), the verdict on the build, the record packed for the host, the build's counters cleared.
__device__ __forceinline__ void tile_tail_duties(unsigned long long total, int wrapped, int wrapped2, int shift, int64_t n, int64_t extra_waves,
                                                 int64_t grid_waves, int groups, int* __restrict__ info, const TileTail& tail, int tid) {
  const int my_flag = tail.pack && tid < tail.flag_words ? tail.flags[tid] : 0;  // (on its way while thread 0 works)
  if (tid == 0) {
    int out[8] = {0, wrapped, wrapped2, 0, 0, 0, 0, 0};
    out[0] = (int)(total > 0x7fffffffull ? 0x7fffffffull : total);
    // walk_tile_total's arithmetic (a total clipped to 2^31 - 1 has raised the flag already)
    const uint32_t budget = tile_budget(total, extra_waves, shift);
    out[3] = (int)budget;
    out[5] = (int)(total / budget + (unsigned long long)((n > 0 ? n - 1 : 0) / 64));  // the last wave that can hold a target
    if (total / budget + (unsigned long long)(n / 64) + 1 >= (unsigned long long)grid_waves) out[1] = 1;
    out[4] = groups;
#pragma unroll
    for (int k = 0; k < 8; ++k) info[k] = out[k];
    int v0 = 0, v1 = 0;
    if (tail.verdict) {
      const int m = tail.flags[kBvhNodes];
      const bool ok = tail.flags[kBvhFallback] == 0 && tail.flags[kBvhBadIndex] == 0 && m > 0 && m <= tail.node_cap &&
                      tail.flags[kBvhNodeCount] <= tail.node_cap &&
                      (tail.level_end <= 0 || tail.bigcount[tail.level_end] == 0);
      v0 = ok ? m : 0;
      v1 = ok ? 1 : 0;
      tail.verdict[0] = v0;
      tail.verdict[1] = v1;
    }
    if (tail.pack) {
      tail.pack[0] = v0;
      tail.pack[1] = v1;
#pragma unroll
      for (int k = 0; k < 8; ++k) tail.pack[2 + tail.flag_words + k] = out[k];
    }
  }
  if (tail.pack && tid < tail.flag_words) tail.pack[2 + tid] = my_flag;
  __syncthreads();  // the flags are in registers or packed: the next build's counters may go
  for (int k = tid; k < tail.clear_words; k += 256) tail.clear[k] = 0;
}

// walk_check_wrap_est, and in the work-group that finishes last: walk_tile_total and the TileTail duties (walk_split.h).
// info[4] counts the finished groups, info[5] takes the estimate's total from the thread that meets the last target (zero
// before, like the rest of info).
__global__ __launch_bounds__(256) void walk_check_est_tail(EstimateOf est, const uint32_t* __restrict__ off, int64_t n, int64_t extra_waves,
                                                           int64_t grid_waves, int* __restrict__ info, const TileTail tail) {
  __shared__ int last_group;
  const int tid = threadIdx.x;
  // (few groups, each over many targets: every group ends with an atomic on ONE counter, ~60 ns apiece)
  // four targets per thread and round, their loads side by side (a dependent gather each: hist[ids[i]])
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + tid; i0 < n; i0 += 4 * stride) {
    uint32_t o[4], e[4], nx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * stride;
      o[k] = i < n ? off[i] : 0u;
      e[k] = i < n ? est((int)i) : 0u;
      nx[k] = i + 1 < n ? off[i + 1] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k * stride;
      const unsigned long long end = (unsigned long long)o[k] + e[k];
      if (i + 1 < n && end != (unsigned long long)nx[k]) {
        info[1] = 1;
        info[2] = 1;
      }
      if (i + 1 == n) {
        if (end > 0x7fffffffull) info[1] = 1;  // the budget arithmetic is 31 bits wide
        info[5] = (int)(end > 0x7fffffffull ? 0x7fffffffull : end);
      }
    }
  }
  __threadfence();  // the flags above, before this group counts as finished
  __syncthreads();
  if (tid == 0) last_group = atomicAdd(&info[4], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last_group) return;
  __threadfence();
  // what the other groups left, past this compute unit's cache
  const int wrapped = __hip_atomic_load(&info[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int wrapped2 = __hip_atomic_load(&info[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long total = (unsigned long long)(unsigned)__hip_atomic_load(&info[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  tile_tail_duties(total, wrapped, wrapped2, est.shift, n, extra_waves, grid_waves, (int)gridDim.x, info, tail, tid);
}

// The estimate's exclusive scan, its overflow check and the TileTail duties in ONE launch (the step enqueued ahead of the host
// spent three on them: the library scan's two kernels and walk_check_est_tail, 32 us of a 1.1 ms step).  A single-pass scan
// with decoupled look-back: a work-group takes a ticket (arrival order, so every predecessor is running or done), scans its
// 1 024 estimates, publishes its aggregate, looks back over its predecessors' states until it meets an inclusive prefix, and
// publishes its own.  States are 64-bit words {epoch : 30 | flag : 2 | value : 32} read and written whole; the epoch (the high
// half of the ticket word st[0], drawn with the ticket by one 64-bit atomicAdd; the work-group with the last ticket stores
// {epoch + 1, ticket 0}: by then every group has drawn) makes last launch's states invisible, so the host clears the area once, when it allocates it, and
// keeps no books.  Values saturate at 2^31: a total that large is the overflow the walk must not run with (the old check's
// wrapped 32-bit sums), and offsets are not needed then.  A look-back that waits absurdly long gives up with a saturated
// prefix: the walk is then skipped and the host walks the plain way — never a hang.  The work-group with the last ticket
// knows the total and does the tail's duties.
constexpr int kScanPer = 4;
constexpr int64_t kFusedScanMaxTargets = (int64_t)1 << 24;
constexpr unsigned long long kScanSat = 0x80000000ull;
constexpr unsigned long long kScanAgg = 1ull, kScanPrefix = 2ull;
__device__ __forceinline__ unsigned long long scan_word(unsigned epoch, unsigned long long flag, unsigned long long v) {
  return ((unsigned long long)epoch << 34) | (flag << 32) | (v > kScanSat ? kScanSat : v);
}
__global__ __launch_bounds__(256) void walk_scan_est_tail(EstimateOf est, uint32_t* __restrict__ off, int64_t n, unsigned long long* __restrict__ st,
                                                          int64_t extra_waves, int64_t grid_waves, int* __restrict__ info, const TileTail tail) {
  __shared__ unsigned s_b, s_epoch;
  __shared__ unsigned long long s_wave[4], s_excl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0 && blockIdx.x == 0 && tail.stamp) *tail.stamp = (unsigned long long)wall_clock64();
  if (tid == 0) {
    // ticket (low word) and epoch (high word) come out of ONE 64-bit atomic: two separate relaxed accesses to two addresses are not
    // ordered by the memory model, and a group that took an early ticket but read the epoch the last group had already advanced
    // would publish and wait under the wrong epoch
    const unsigned long long te = atomicAdd(&st[0], 1ull);
    s_b = (unsigned)te;
    s_epoch = (unsigned)(te >> 32) & 0x3FFFFFFFu;
  }
  __syncthreads();
  const unsigned b = s_b, epoch = s_epoch;
  unsigned long long* __restrict__ state = st + 2;
  const int64_t i0 = ((int64_t)b * 256 + tid) * kScanPer;
  uint32_t e[kScanPer];
  unsigned long long tsum = 0ull;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    e[k] = i0 + k < n ? est((int)(i0 + k)) : 0u;
    tsum += e[k];
  }
  unsigned long long inc = tsum;  // inclusive over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = (unsigned long long)__shfl_up((long long)inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  unsigned long long before = 0ull, block_total = 0ull;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) before += s_wave[w];
    block_total += s_wave[w];
  }
  if (wave == 0) {
    unsigned long long excl = 0ull;
    if (b > 0) {
      if (lane == 0) __hip_atomic_store(&state[b], scan_word(epoch, kScanAgg, block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (long long j = (long long)b - 1;; j -= 64) {  // 64 predecessors at a time, the nearest in lane 0
        const long long idx = j - lane;
        unsigned long long v = scan_word(epoch, kScanPrefix, 0ull);  // before the first work-group: an empty prefix
        if (idx >= 0) {
          int patience = 1 << 20;
          for (;;) {  // (its owner holds an earlier ticket: it is on its way.  The states carry their own data: relaxed loads do)
            v = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((unsigned)(v >> 34) == epoch && ((v >> 32) & 3ull) != 0ull) || --patience <= 0) break;
            __builtin_amdgcn_s_sleep(2);
          }
          if (patience <= 0) v = scan_word(epoch, kScanPrefix, kScanSat);  // give up: saturated, the walk will not run on this scan
        }
        const unsigned long long has_prefix = __builtin_amdgcn_ballot_w64(((v >> 32) & 3ull) == kScanPrefix);
        const int stop = has_prefix ? __builtin_ctzll(has_prefix) : 63;  // lanes 0 .. stop count
        unsigned long long part = lane <= stop ? (v & 0xFFFFFFFFull) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += (unsigned long long)__shfl_xor((long long)part, d, 64);
        excl += part;
        if (excl > kScanSat) excl = kScanSat;
        if (has_prefix) break;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(&state[b], scan_word(epoch, kScanPrefix, excl + block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
    }
  }
  __syncthreads();
  const unsigned long long excl = s_excl;
  unsigned long long run = excl + before + (inc - tsum);
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    if (i0 + k < n) off[i0 + k] = (uint32_t)run;
    run += e[k];
  }
  if (b + 1 != gridDim.x) return;
  if (tid == 0)  // every group holds its ticket and, with it, the epoch: the word moves on to {epoch + 1, ticket 0} in one store
    __hip_atomic_store(&st[0], (unsigned long long)((epoch + 1u) & 0x3FFFFFFFu) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long total = excl + block_total;
  if (total > kScanSat) total = kScanSat;
  const int wrapped = total > 0x7fffffffull ? 1 : 0;
  tile_tail_duties(total, wrapped, wrapped, est.shift, n, extra_waves, grid_waves, (int)gridDim.x, info, tail, tid);
}

#ifdef NBODY_LAB
// ---- which work-groups go first (round 4) --------------------------------------------------------------------------------
// On a scene whose waves do not all fit the chip at once (Plummer 1 M: 17 400 waves on 8 192 slots) the work-groups are
// dispatched in index order = tree order, and the dense centre's long waves (2.2 ms against a mean of 0.9) sit in the middle of
// it: those past the first residency round start a millisecond late and end the kernel at 2.9 ms where the sum of all wave
// times over the slots is 1.9 (profiles/r04_walk_wave_log.txt).  So the groups are dealt out longest first — in CHUNKS of
// consecutive groups (neighbouring groups walk neighbouring targets and share their nodes and leaves in the L2s: dealing single
// groups out costs 10-25 %), a chunk's weight being its targets' estimated terms, which the scan has left in `off`.  One
// work-group: a wave per chunk finds the chunk's first target (the walk's own 64-ary search), then the chunks are ranked.
// MEASURED (profiles/r04_walk_order_ab.txt): heaviest first gains 10 % at Plummer 1 M (f32) and 6-8 % at 655 360 / 1 M in f64, and LOSES
// 5 % at 655 360 and 4-8 % at 2 M in f32; "lightest last" gains nothing anywhere.  No rule follows from that, so the product keeps the
// index order and this stays a laboratory switch (NBODY_WALK_ORDER=2 / 1).
// mode 2: all chunks heaviest first.  mode 1: the LIGHTEST chunks — as many as one residency round holds — go last, everything else stays
// in index order: what matters is that no long wave starts late, and the rest of the order is the locality the walks live on.
__global__ __launch_bounds__(1024) void walk_order_chunks(const uint32_t* __restrict__ off, const int n_tgt, const int* __restrict__ info,
                                                         const int chunk_groups, const int n_chunks, int* __restrict__ order, const int mode,
                                                         const int n_light) {
  __shared__ int bnd[kWalkOrderChunks + 1];
  __shared__ unsigned long long cost[kWalkOrderChunks];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t M = 0xFFFFFFFFu / (uint32_t)__builtin_amdgcn_readfirstlane(info[3]);
  for (int c = w; c < n_chunks; c += 16) {
    const int t = first_target_reaching(off, n_tgt, c * chunk_groups * 4, M, lane);
    if (lane == 0) bnd[c] = t;
  }
  if (tid == 0) bnd[n_chunks] = n_tgt;
  __syncthreads();
  const unsigned long long total = (unsigned long long)(unsigned)info[0];
  if (tid < n_chunks) {
    const int b0 = bnd[tid], b1 = bnd[tid + 1];
    const unsigned long long o0 = b0 < n_tgt ? off[b0] : total, o1 = b1 < n_tgt ? off[b1] : total;
    // (the estimate's terms and, so that equal estimates still order by work, the head count)
    cost[tid] = (o1 >= o0 ? o1 - o0 : 0ull) + (unsigned long long)(b1 - b0);
  }
  __syncthreads();
  __shared__ unsigned char light[kWalkOrderChunks];
  int rank = 0;  // among all chunks, heaviest first (a bijection: every chunk has its own rank)
  if (tid < n_chunks) {
    const unsigned long long mine = cost[tid];
    for (int h = 0; h < n_chunks; ++h) rank += (cost[h] > mine || (cost[h] == mine && h < tid)) ? 1 : 0;
    light[tid] = rank >= n_chunks - n_light ? 1 : 0;
  }
  __syncthreads();
  if (tid < n_chunks) {
    if (mode == 2) {
      order[rank] = tid;
    } else {
      int before_same = 0;  // chunks of my kind before me, in index order
      for (int h = 0; h < tid; ++h) before_same += light[h] == light[tid] ? 1 : 0;
      order[light[tid] ? n_chunks - n_light + before_same : before_same] = tid;
    }
  }
}
#endif  // NBODY_LAB (walk_order_chunks)

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
inline uint32_t tile_budget_targets() {  // (laboratory: a wave's budget as this many average targets)
  const int v = lab_int("NBODY_WALK_TILE_BUDGET_TARGETS", 0);
  return (uint32_t)(v < 1 ? 0 : (v > 4096 ? 4096 : v));
}
inline int tile_targets() {  // rows of the LDS tile: 8 keep it at 4 KB per wave, eight waves per SIMD (laboratory: 4 / 16)
  const int v = lab_int("NBODY_WALK_TILE_TARGETS", 8);
  return (v == 16 || v == 4) ? v : 8;
}

}  // namespace

WalkSplitLayout walk_split_layout(int64_t n_tgt) {
  WalkSplitLayout L{};
  const size_t n = (size_t)(n_tgt > 0 ? n_tgt : 1);
  size_t o = 0;
  auto take = [&](size_t b) { size_t r = o; o += align_up(b); return r; };
  // first, at the same place whatever n_tgt is (the caller zeroes it once per allocation): walk_scan_est_tail's ticket, epoch
  // and one state per work-group of 1 024 targets, for up to kFusedScanMaxTargets of them
  L.scan_state_bytes = 8 * (kFusedScanMaxTargets / (256 * kScanPer) + 4);
  L.scan_state = take(L.scan_state_bytes);
  L.cnt = take(4 * n);
  L.off = take(4 * n);
  L.info = take(32);
  L.order = take(4 * (size_t)kWalkOrderChunks);
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, (hipStream_t) nullptr);
  L.cub_temp_bytes = tb;
  L.cub_temp = take(tb);
  L.total = o;
  return L;
}

#ifdef NBODY_LAB
hipError_t launch_tree_walk_split(hipStream_t s, const WalkArgs<float>& a, char* scratch, const WalkSplitLayout& L, void* terms,
                                  int64_t term_capacity) {
  if (a.n_tgt <= 0) return hipSuccess;
  uint32_t* cnt = (uint32_t*)(scratch + L.cnt);
  uint32_t* off = (uint32_t*)(scratch + L.off);
  int* info = (int*)(scratch + L.info);
  const int64_t cwaves = (a.n_tgt + kCountTPW - 1) / kCountTPW;
  const int64_t twaves = term_capacity / kTermBudget + a.n_tgt / 64 + 2;  // upper bound of g(t) + 1
  hipError_t e = hipMemsetAsync(info, 0, 32, s);
  if (e != hipSuccess) return e;
  walk_pass<false, kCountTPW, false><<<dim3((unsigned)((cwaves + 3) / 4)), dim3(256), 0, s>>>(a, cnt, nullptr, nullptr, info, term_capacity);
  size_t tb = L.cub_temp_bytes;
  e = hipcub::DeviceScan::ExclusiveSum((void*)(scratch + L.cub_temp), tb, (const uint32_t*)cnt, off, (int)a.n_tgt, s);
  if (e != hipSuccess) return e;
  walk_check_wrap<<<dim3((unsigned)((a.n_tgt + 255) / 256)), dim3(256), 0, s>>>(cnt, off, a.n_tgt, info);
  walk_total<<<dim3(1), dim3(1), 0, s>>>(cnt, off, a.n_tgt, term_capacity, info);
  if (a.fast) walk_pass<true, 64, true><<<dim3((unsigned)((twaves + 3) / 4)), dim3(256), 0, s>>>(a, cnt, off, (float2*)terms, info, term_capacity);
  else walk_pass<true, 64, false><<<dim3((unsigned)((twaves + 3) / 4)), dim3(256), 0, s>>>(a, cnt, off, (float2*)terms, info, term_capacity);
  const int64_t sum_waves = (a.n_tgt + 3) / 4;
  walk_sum<<<dim3((unsigned)((sum_waves + 3) / 4)), dim3(256), 0, s>>>(a, cnt, off, (const float2*)terms, info);
  return hipGetLastError();
}
#endif  // NBODY_LAB

// The walk's preparation (the estimate's scan, its wrap check, the waves' budget: info[0..3] are final afterwards) and
// the walk proper, separately: a caller that wants to see info[1] before the long kernel has run enqueues a copy and an
// event between the two.  *grid_waves carries the wave count from the one to the other.
template <class T>
hipError_t launch_tree_walk_tile_prep(hipStream_t s, const WalkArgs<T>& a, char* scratch, const WalkSplitLayout& L, const uint32_t* tgt_ids,
                                      uint32_t* hist, int estimate, int shift, int64_t* grid_waves, const TileTail* tail) {
  const bool have_history = estimate != 0;  // 1: from hist; 2: none at all (every wave takes 64 targets)
  *grid_waves = 0;
  if (tail && (estimate != 1 || a.n_tgt <= 0 || tail->flag_words > 256)) return hipErrorInvalidValue;  // the tail rides on the history estimate's check
  if (a.n_tgt <= 0) return hipSuccess;
  uint32_t* cnt = (uint32_t*)(scratch + L.cnt);
  uint32_t* off = (uint32_t*)(scratch + L.off);
  int* info = (int*)(scratch + L.info);
  hipError_t e = hipSuccess;
  if (!(tail && tail->info_zeroed)) e = hipMemsetAsync(info, 0, 32, s);
  if (e != hipSuccess) return e;
  size_t tb = L.cub_temp_bytes;
  if (!have_history) {  // no counts of an earlier walk over these targets: count (exact, so shift 0)
    shift = 0;
    const int64_t cwaves = (a.n_tgt + kCountTPW - 1) / kCountTPW;
    walk_count<T><<<dim3((unsigned)((cwaves + 3) / 4)), dim3(256), 0, s>>>(a, cnt);
    e = hipcub::DeviceScan::ExclusiveSum((void*)(scratch + L.cub_temp), tb, (const uint32_t*)cnt, off, (int)a.n_tgt, s);
    if (e != hipSuccess) return e;
    walk_check_wrap<<<dim3((unsigned)((a.n_tgt + 255) / 256)), dim3(256), 0, s>>>(cnt, off, a.n_tgt, info);
  } else if (tail && tail->fused_scan) {
    // (everything is done by the one kernel launched below)
  } else {
    hipcub::CountingInputIterator<int> idx(0);
    hipcub::TransformInputIterator<uint32_t, EstimateOf, hipcub::CountingInputIterator<int>> est(idx, EstimateOf{estimate == 1 ? hist : nullptr, tgt_ids, shift});
    size_t need = 0;  // the layout sized the scan's scratch for plain pointers: make sure this iterator asks for no more
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, est, off, (int)a.n_tgt, s);
    if (e != hipSuccess) return e;
    if (need > tb) return hipErrorInvalidValue;
    e = hipcub::DeviceScan::ExclusiveSum((void*)(scratch + L.cub_temp), tb, est, off, (int)a.n_tgt, s);
    if (e != hipSuccess) return e;
    if (estimate == 1 && !tail)
      walk_check_wrap_est<<<dim3((unsigned)((a.n_tgt + 255) / 256)), dim3(256), 0, s>>>(EstimateOf{hist, tgt_ids, shift}, off, a.n_tgt, info);
  }
  const int ew = lab_int("NBODY_WALK_TILE_WAVES", 0);  // (laboratory: override of tile_waves_target)
  int64_t extra = (ew > 0 ? (int64_t)ew : tile_waves_target(a.n_tgt)) - a.n_tgt / 64;
  if (extra < 256) extra = 256;
  const uint32_t bt = tile_budget_targets();  // development override: a budget of this many average targets
  if (bt) extra = a.n_tgt / bt;
  if (extra < 1) extra = 1;
  const int64_t twaves = extra + a.n_tgt / 64 + 4;  // upper bound of g(t) + 1 (budget >= ceil(total / extra))
  *grid_waves = twaves;
  if (tail && tail->fused_scan && a.n_tgt > kFusedScanMaxTargets) return hipErrorInvalidValue;
  if (tail && tail->fused_scan) {
    const unsigned groups = (unsigned)((a.n_tgt + 256 * kScanPer - 1) / (256 * kScanPer));
    walk_scan_est_tail<<<dim3(groups), dim3(256), 0, s>>>(EstimateOf{hist, tgt_ids, shift}, off, a.n_tgt, (unsigned long long*)(scratch + L.scan_state),
                                                       extra, twaves, info, *tail);
    return hipGetLastError();
  }
  if (tail) {
    int64_t groups = a.n_tgt / 2048;  // (every group ends with an atomic on one counter: few groups, longer loops)
    groups = groups < 32 ? 32 : (groups > 128 ? 128 : groups);
    walk_check_est_tail<<<dim3((unsigned)groups), dim3(256), 0, s>>>(EstimateOf{hist, tgt_ids, shift}, off, a.n_tgt, extra, twaves,
                                                                                      info, *tail);
    return hipGetLastError();
  }
  walk_tile_total<<<dim3(1), dim3(1), 0, s>>>(have_history ? nullptr : cnt, off, a.n_tgt, tgt_ids, estimate == 1 ? hist : nullptr, shift, extra,
                                              twaves, info);
  *grid_waves = twaves;
  return hipGetLastError();
}

template <class T>
hipError_t launch_tree_walk_tile_main(hipStream_t s, const WalkArgs<T>& a_in, char* scratch, const WalkSplitLayout& L, const uint32_t* tgt_ids,
                                      uint32_t* hist, int64_t grid_waves) {
  if (a_in.n_tgt <= 0 || grid_waves <= 0) return hipSuccess;
  const uint32_t* off = (const uint32_t*)(scratch + L.off);
  int* info = (int*)(scratch + L.info);
  unsigned long long* total_out = (unsigned long long*)(info + 6);
  const int tt = tile_targets();
  const dim3 grid((unsigned)((grid_waves + 3) / 4));
  // laboratory: NBODY_WALK_BLOCK_STRIDE=1 deals the work-groups out with a golden-ratio stride (coprime with the grid) instead of in order
  WalkArgs<T> a_strided = a_in;
  a_strided.block_stride = 1;
  a_strided.group_order = nullptr;
  a_strided.order_chunk = 0;
  dim3 grid_ordered = grid;
  // laboratory: more waves than the chip holds at once (256 CUs x 32): chunks of work-groups, heaviest first (walk_order_chunks)
#ifdef NBODY_LAB
  const int order_mode = lab_int("NBODY_WALK_ORDER", 0);  // laboratory: 2 every chunk heaviest first, 1 the lightest chunks last
  if (grid_waves > 10240 && order_mode != 0 && lab_int("NBODY_WALK_WAVE_LOG", 0) == 0 && lab_int("NBODY_WALK_FAST_BFS", 0) == 0) {
    const int ng = (int)grid.x;
    int cg = (ng + kWalkOrderChunks - 1) / kWalkOrderChunks;
    if (cg < 64) cg = 64;
    const int nc = (ng + cg - 1) / cg;
    int* order = (int*)(scratch + L.order);
    int n_light = 8192 / (cg * 4);  // chunks of one residency round (256 CUs x 32 waves)
    if (n_light > nc / 2) n_light = nc / 2;
    walk_order_chunks<<<dim3(1), dim3(1024), 0, s>>>(off, (int)a_in.n_tgt, info, cg, nc, order, order_mode, n_light);
    a_strided.group_order = order;
    a_strided.order_chunk = cg;
    grid_ordered = dim3((unsigned)(nc * cg));  // (whole chunks: the groups past the last real one find no targets and leave)
  }
#endif
  if (lab_int("NBODY_WALK_BLOCK_STRIDE", 0) != 0 && grid.x > 8) {
    auto gcd = [](unsigned x, unsigned y) { while (y) { const unsigned t = x % y; x = y; y = t; } return x; };
    unsigned st = (unsigned)(0.6180339887 * grid.x) | 1u;
    while (gcd(st, grid.x) != 1) st += 2;
    a_strided.block_stride = (int)st;
  }
  // node records by scalar loads (scalar_node_rec): the exact walk always (62 instead of 72 VGPRs: eight waves per SIMD instead of seven;
  // reference scene 0.579 -> 0.567 ms, Plummer 1 M 6.07 -> 5.96); NBODY_WALK_SCALAR_REC=0: the vector loads of one address
  static const bool srec = lab_int("NBODY_WALK_SCALAR_REC", 1) != 0;
  // FAST: f32 takes walk_tile_fast (registers, lane-parallel sums); f64 the rows arm (walk_tile<double, true>: LDS rows + ordered adds with
  // the one-reciprocal term) — walk_tile_fast<double> moves every value as two 32-bit halves through the swaps and runs at half
  // the occupancy: Plummer 4 M f64 205 ms against 92 (the exact walk: 124).  Laboratory: NBODY_WALK_FAST_ROWS=0/1 forces one or the other.
  static const int fast_rows_env = lab_int("NBODY_WALK_FAST_ROWS", -1);
  const bool fast_rows = fast_rows_env >= 0 ? fast_rows_env != 0 : sizeof(T) == 8;
  // FAST: node records by scalar loads (round 3 took vector loads of one address below 400 000 targets: 0.298 against 0.322 ms on the
  // reference scene's FIRST steps; over the bench leg's 300 steps, with the waves in one residency round, scalar loads win there too:
  // 0.299 -> 0.277 ms, profiles/r04_walk_wave_target.txt).  Laboratory: NBODY_WALK_FAST_REC=0 / 1 the plain / pinned vector loads.
  static const int rec_env = lab_int("NBODY_WALK_FAST_REC", -1);
  const int rec_mode = rec_env >= 0 ? rec_env : 3;
#ifdef NBODY_LAB
  // every variant the A/B tools switch between: rows 4 / 8 / 16, node records by vector loads, the rows arm in f32, the register arm
  // in f64, the per-wave log
#define NB_TILE(F, R) do { if (srec) walk_tile<T, F, R, true><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out); \
                           else walk_tile<T, F, R, false><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out); } while (0)
  unsigned long long* wave_log = nullptr;
  WalkArgs<T> a_log = a_strided;
  if (a_in.fast && lab_int("NBODY_WALK_WAVE_LOG", 0) != 0) {  // development: per-wave time and step counts
    if (hipMalloc((void**)&wave_log, (size_t)grid.x * 4 * 4 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(wave_log, 0, (size_t)grid.x * 4 * 4 * sizeof(unsigned long long), s);
    a_log.wave_log = wave_log;
  }
  const WalkArgs<T>& a = a_log;
  // NBODY_WALK_FAST_BFS=1: round 4's breadth-first FAST walk (walk_tile_fast_bfs) — measured SLOWER than the depth-first kernel
  // (reference scene 0.579 against 0.338 ms, Plummer 1 M 4.96 against 2.99: profiles/r04_walk_bfs_ab.txt), kept here for that A/B only
  bool bfs = false;
  if constexpr (sizeof(T) == 4) bfs = a.fast && !fast_rows && !wave_log && lab_int("NBODY_WALK_FAST_BFS", 0) != 0;
  if (bfs) {
    if constexpr (sizeof(T) == 4) walk_tile_fast_bfs<<<grid, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
  } else if (a.fast && !fast_rows) {
    if (wave_log) walk_tile_fast<T, 0, true><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
    else if (rec_mode == 1) walk_tile_fast<T, 1, false><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
    else if (rec_mode == 3) walk_tile_fast<T, 3, false><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
    else walk_tile_fast<T, 0, false><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
  }
  else if (a.fast) { if (tt == 16) NB_TILE(true, 16); else if (tt == 4) NB_TILE(true, 4); else NB_TILE(true, 8); }
  else { if (tt == 16) NB_TILE(false, 16); else if (tt == 4) NB_TILE(false, 4); else NB_TILE(false, 8); }
#undef NB_TILE
  if (wave_log) {
    const size_t nw = (size_t)grid.x * 4;
    std::vector<unsigned long long> h(nw * 4);
    (void)hipMemcpyAsync(h.data(), wave_log, nw * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(wave_log);
    const char* path = lab_str("NBODY_WALK_WAVE_LOG_FILE");
    FILE* f = fopen(path ? path : "/tmp/nbody_wave_log.bin", "wb");
    if (f) {
      fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
      fclose(f);
    }
  }
#else
  // the product's instantiations: the exact walk with 8 rows and scalar node records; FAST in f32 through registers (node records
  // by plain or scalar loads, by size), FAST in f64 through the rows
  (void)srec; (void)tt; (void)fast_rows; (void)rec_mode;
  const WalkArgs<T>& a = a_strided;
  if constexpr (sizeof(T) == 4) {
    if (a.fast) {
      walk_tile_fast<T, 3, false><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
    } else {
      walk_tile<T, false, 8, true><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
    }
  } else {
    if (a.fast) walk_tile<T, true, 8, true><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
    else walk_tile<T, false, 8, true><<<grid_ordered, dim3(256), 0, s>>>(a, off, info, tgt_ids, hist, total_out);
  }
#endif
  return hipGetLastError();
}

template <class T>
hipError_t launch_tree_walk_tile(hipStream_t s, const WalkArgs<T>& a, char* scratch, const WalkSplitLayout& L, const uint32_t* tgt_ids,
                                 uint32_t* hist, int estimate, int shift) {
  int64_t waves = 0;
  hipError_t e = launch_tree_walk_tile_prep<T>(s, a, scratch, L, tgt_ids, hist, estimate, shift, &waves, nullptr);
  if (e != hipSuccess) return e;
  return launch_tree_walk_tile_main<T>(s, a, scratch, L, tgt_ids, hist, waves);
}

template hipError_t launch_tree_walk_tile<float>(hipStream_t, const WalkArgs<float>&, char*, const WalkSplitLayout&, const uint32_t*, uint32_t*, int, int);
template hipError_t launch_tree_walk_tile<double>(hipStream_t, const WalkArgs<double>&, char*, const WalkSplitLayout&, const uint32_t*, uint32_t*, int, int);
template hipError_t launch_tree_walk_tile_prep<float>(hipStream_t, const WalkArgs<float>&, char*, const WalkSplitLayout&, const uint32_t*, uint32_t*, int, int, int64_t*,
                                                      const TileTail*);
template hipError_t launch_tree_walk_tile_prep<double>(hipStream_t, const WalkArgs<double>&, char*, const WalkSplitLayout&, const uint32_t*, uint32_t*, int, int, int64_t*,
                                                       const TileTail*);
template hipError_t launch_tree_walk_tile_main<float>(hipStream_t, const WalkArgs<float>&, char*, const WalkSplitLayout&, const uint32_t*, uint32_t*, int64_t);
template hipError_t launch_tree_walk_tile_main<double>(hipStream_t, const WalkArgs<double>&, char*, const WalkSplitLayout&, const uint32_t*, uint32_t*, int64_t);

}  // namespace nbody

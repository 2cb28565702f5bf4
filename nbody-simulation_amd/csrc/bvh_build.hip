// Device-side build of the linearised BVH (gfx950, f32) — bit-identical to the host builder (tree_build.hpp) and to
// BVHTree::from + calculate_gravity of /root/reference src/bvh_tree.rs:40-158.
//
// What has to be reproduced, per node (bvh_tree.rs:56-96): ONE sequential fold over the slice (min from f32::MAX, max
// from 0.0, sum in slice order), mean = sum / len, the "better balanced axis" rule on #{x > mean.x} / #{y > mean.y},
// and the two-pointer partition of crate `partition` 0.1.2 (predicate-true side first).  The order of the particles
// inside each side feeds the next level's sum, so the permutation has to be exact, not just the split.
//
//   * min / max / counts are order-independent: plain reductions.
//   * the sum is a rounding chain: exact_sum.h scans it (addend = map parity -> increment inside one binade; a real
//     f32 add whenever the chain leaves its binade).
//   * the two-pointer partition swaps the k-th misplaced element from the left with the k-th misplaced element from
//     the right (the pointers only ever stop at misplaced elements): two rank lists from one prefix count, then
//     independent swaps.
//
// Two phases.  Nodes longer than kSub points are processed breadth-first, a level at a time, with the node's points
// spread over many work-groups: bvh_big_fold (the chain: one 1024-thread group per node, the only serial part),
// bvh_big_count (per-chunk counts; the last chunk to finish picks the axis, the split and creates the children),
// bvh_big_ranks (ranks from the predicate bits bvh_big_count left; the two halves of a pair meet in a 64-bit slot and whoever
// arrives second swaps: round 3, one launch per level fewer).  Every node of at most kSub points is the root of a subtree that ONE
// work-group builds completely out of LDS (bvh_subtrees): long nodes by the whole group, short ones a wave each.
//
// Nodes get breadth-first ids while they are made.  The upward pass (:133-158; leaves: unweighted mean in slice order,
// u32 wrapping mass, :98-131) also counts the nodes of every subtree; the pre-order index the walk needs follows top-down
// (node, left subtree, right subtree) for the nodes above the subtrees, and by a walk up to the nearest numbered ancestor
// for the others; `skip` = index + subtree size.
//
// Anything this cannot express (NaN positions — pathfinder's minps/maxps are order-dependent there —, a node deeper than
// the 56-bit path, more nodes than the buffers hold) raises a flag and the caller uses the host builder instead.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <cstdio>

#include "bvh_build.h"
#include "tree_kernels.h"
#include "exact_sum.h"

namespace nbody {

namespace {

constexpr int kEPT = 4;          // consecutive points per thread in the rank-list passes
constexpr int kSeqRun = 16;      // real adds after every stop of the scan
#ifndef NB_CHAIN_START
#define NB_CHAIN_START 512
#endif
constexpr int kChainStart = NB_CHAIN_START; // real adds at the start of a long node's chain (the sum doubles too often there)
#ifndef NB_KSUB
#define NB_KSUB 2048
#endif
constexpr int kSub = NB_KSUB;       // nodes up to this long are built, subtree and all, by one work-group in LDS
constexpr int kSubWaves = 8;     // waves of a subtree work-group
constexpr int kChunk = 2048;     // points per work-group in the multi-group passes over a long node
#ifndef NB_RUN_LEN
#define NB_RUN_LEN 16384
#endif
constexpr int kFusedPartitionMax = 400000;  // points up to which a level's rank pass also swaps (bvh_big_ranks); beyond: lists + bvh_big_swap
constexpr int kRunLen = NB_RUN_LEN;   // nodes longer than this have their chain prepared chunk by chunk (bvh_chunk_runs)
constexpr float kMaxF = 3.402823466e+38f;

struct BvhPtrs {
  int* flags;
  int* bigcount;    // [level]: long nodes queued for the level
  int* chunkcount;  // [level]
  int* bigq;        // [level & 1][cap_big]
  int* subq;        // subtree roots
  int* topq;        // nodes made by the long-node levels (the ones above the subtrees)
  int4* ch_rec;     // [level & 1][cap_chunk] chunk -> node, index inside the node, the node's begin and length (one load
                    // tells a chunk's kernel where its points are: the node's other fields and the points travel together)
  double2* ch_sum;  // exact-ish (f64) sums of the chunk's coordinates: only used to PREDICT binades
  float4* ch_box;   // min.x min.y max.x max.y of the chunk
  int* ch_run;      // [chunk][2 coordinates][kRunRec]: see bvh_chunk_runs
  int* ch_cx;
  int* ch_cy;
  int* ch_before;   // predicate-true points of the node before the chunk
  float2* P;
  uint32_t* ID;
  int* lidx;
  int* ridx;
  uint8_t* pred;             // long nodes: bit 0 = x > mean.x, bit 1 = y > mean.y of every point (bvh_big_count), read by bvh_big_ranks
  unsigned long long* pair;  // long nodes: one slot per pair of misplaced points (low word: the left one's index + 1, high word: the right one's)
  int* nbegin;
  int* nlen;
  int* nparent;
  int* nchild;
  int* ndepth;
  int* nleaf;
  int* nsize;  // nodes in the subtree (the node itself included)
  int* npre;   // pre-order index, -1 while unknown
  float4* nbox;  // min.x, min.y, max.x, max.y
  float2* ncog;
  uint32_t* nmass;
  int* narrive;
  float2* nmean;
  int* nsplit;   // long nodes: m | axis << 31
  int* nchunk0;
  int* ndone;
  int* nbad;
  int cap, cap_big, cap_chunk;
};

BvhPtrs make_ptrs(char* s, const BvhBuildLayout& L) {
  BvhPtrs a;
  a.flags = (int*)(s + L.flags);
  a.bigcount = (int*)(s + L.bigcount);
  a.chunkcount = (int*)(s + L.chunkcount);
  a.bigq = (int*)(s + L.bigq);
  a.subq = (int*)(s + L.subq);
  a.topq = (int*)(s + L.topq);
  a.ch_rec = (int4*)(s + L.ch_rec);
  a.ch_sum = (double2*)(s + L.ch_sum);
  a.ch_box = (float4*)(s + L.ch_box);
  a.ch_run = (int*)(s + L.ch_run);
  a.ch_cx = (int*)(s + L.ch_cx);
  a.ch_cy = (int*)(s + L.ch_cy);
  a.ch_before = (int*)(s + L.ch_before);
  a.P = (float2*)(s + L.pts);
  a.ID = (uint32_t*)(s + L.ids);
  a.lidx = (int*)(s + L.lidx);
  a.ridx = (int*)(s + L.ridx);
  a.pred = (uint8_t*)(s + L.pred);
  a.pair = (unsigned long long*)(s + L.pair);
  a.nbegin = (int*)(s + L.nbegin);
  a.nlen = (int*)(s + L.nlen);
  a.nparent = (int*)(s + L.nparent);
  a.nchild = (int*)(s + L.nchild);
  a.ndepth = (int*)(s + L.ndepth);
  a.nleaf = (int*)(s + L.nleaf);
  a.nsize = (int*)(s + L.nsize);
  a.npre = (int*)(s + L.npre);
  a.nbox = (float4*)(s + L.nbox);
  a.ncog = (float2*)(s + L.ncog);
  a.nmass = (uint32_t*)(s + L.nmass);
  a.narrive = (int*)(s + L.narrive);
  a.nmean = (float2*)(s + L.nmean);
  a.nsplit = (int*)(s + L.nsplit);
  a.nchunk0 = (int*)(s + L.nchunk0);
  a.ndone = (int*)(s + L.ndone);
  a.nbad = (int*)(s + L.nbad);
  a.cap = L.node_cap;
  a.cap_big = L.big_cap;
  a.cap_chunk = L.chunk_cap;
  return a;
}

// pathfinder_simd min/max on SSE: `a < b ? a : b` (second operand when unordered); NaNs never get here
__device__ __forceinline__ float sse_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sse_max(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ float lane_value(float v, int k) {  // k uniform
  return xsum::u2f((uint32_t)__builtin_amdgcn_readlane((int)xsum::f2u(v), k));
}

// written by another compute unit in this very kernel: read past the local L1
__device__ __forceinline__ uint32_t load_agent(const uint32_t* p) {
  return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int load_agent(const int* p) { return (int)load_agent(reinterpret_cast<const uint32_t*>(p)); }

// ---- the chain, one add after the other -----------------------------------------------------------------------------
struct Box {
  float mnx = kMaxF, mny = kMaxF, mxx = 0.f, mxy = 0.f;  // the fold's start values, bvh_tree.rs:42, :59
  __device__ __forceinline__ void add(float2 q) {
    mnx = sse_min(mnx, q.x); mny = sse_min(mny, q.y);
    mxx = sse_max(mxx, q.x); mxy = sse_max(mxy, q.y);
  }
  __device__ __forceinline__ void reduce_wave() {
    for (int d = 32; d >= 1; d >>= 1) {
      mnx = sse_min(mnx, __shfl_xor(mnx, d, 64)); mny = sse_min(mny, __shfl_xor(mny, d, 64));
      mxx = sse_max(mxx, __shfl_xor(mxx, d, 64)); mxy = sse_max(mxy, __shfl_xor(mxy, d, 64));
    }
  }
};
// s + X[0] + X[STRIDE] + ... (count addends, in order) from LDS, the same address in every lane (a broadcast read); count
// uniform.  The adds depend on one another, the reads do not: the next 16 words are on their way while these 16 are added,
// else every batch waits out the LDS latency (13+ clocks per addend instead of the 4-5 of a dependent add).  Two register
// sets taken in turn; the sched_group_barriers keep the scheduler from sinking the reads below the adds they overlap.
#define NB_CHAIN_LOAD(dst, from) _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) dst[j_] = X[STRIDE * ((from) + j_)];
#define NB_CHAIN_ADD(src) _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) s = s + src[j_]; \
  __builtin_amdgcn_sched_group_barrier(0x100, 16, 0); __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
template <int STRIDE>
__device__ __forceinline__ float chain_lds(const float* X, int count, float s) {
  count = __builtin_amdgcn_readfirstlane(count);
  int k = 0;
  if (count >= 16) {
    float va[16], vb[16];
    NB_CHAIN_LOAD(va, 0)
    for (; k + 48 <= count; k += 32) {  // va holds [k, k + 16) here
      NB_CHAIN_LOAD(vb, k + 16)
      NB_CHAIN_ADD(va)
      NB_CHAIN_LOAD(va, k + 32)
      NB_CHAIN_ADD(vb)
    }
    if (k + 32 <= count) {
      NB_CHAIN_LOAD(vb, k + 16)
      NB_CHAIN_ADD(va)
      NB_CHAIN_ADD(vb)
      k += 32;
    } else {
      NB_CHAIN_ADD(va)
      k += 16;
    }
  }
  for (; k < count; ++k) s = s + X[STRIDE * k];
  return s;
}
#undef NB_CHAIN_LOAD
#undef NB_CHAIN_ADD
// both coordinates of P[0 .. count) at once (two independent chains share the reads)
__device__ __forceinline__ void chain_lds_xy(const float2* P, int count, float& sx, float& sy) {
  count = __builtin_amdgcn_readfirstlane(count);
  int k = 0;
#define NB_CHAIN_LOAD(dst, from) _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dst[j_] = P[(from) + j_];
#define NB_CHAIN_ADD(src) _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) { sx = sx + src[j_].x; sy = sy + src[j_].y; } \
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
  if (count >= 8) {
    float2 va[8], vb[8];
    NB_CHAIN_LOAD(va, 0)
    for (; k + 24 <= count; k += 16) {  // va holds [k, k + 8) here
      NB_CHAIN_LOAD(vb, k + 8)
      NB_CHAIN_ADD(va)
      NB_CHAIN_LOAD(va, k + 16)
      NB_CHAIN_ADD(vb)
    }
    if (k + 16 <= count) {
      NB_CHAIN_LOAD(vb, k + 8)
      NB_CHAIN_ADD(va)
      NB_CHAIN_ADD(vb)
      k += 16;
    } else {
      NB_CHAIN_ADD(va)
      k += 8;
    }
  }
#undef NB_CHAIN_LOAD
#undef NB_CHAIN_ADD
  for (; k < count; ++k) { sx = sx + P[k].x; sy = sy + P[k].y; }
}

// A window of a node's coordinate.  A scan thread takes 8 consecutive addends: one word of padding after every 8 puts the
// threads of a wave 9 words apart (all 64 banks); the chain reads the window front to back, 8 words at fixed offsets from
// one address.
template <int NT> struct StageView {
  static constexpr int kWords = NT * 9 + 8;
  float* b;
  __device__ __forceinline__ static int at(int e) { return e + (e >> 3); }
  __device__ __forceinline__ float operator[](int e) const { return b[at(e)]; }
};
// s += V[begin .. begin+count) in order, in every lane of every wave that calls (begin, count uniform): all lanes read the
// same LDS word (a broadcast) and add it, 8 clocks per add (node_in_lds below does the same).  No result to hand round.
template <int NT>
__device__ __forceinline__ void chain_run_all(const StageView<NT>& V, int begin, int count, float& s) {
  int e = __builtin_amdgcn_readfirstlane(begin);
  const int end = e + __builtin_amdgcn_readfirstlane(count);
  for (; e < end && (e & 7) != 0; ++e) s = s + V[e];
  if (e + 8 <= end) {  // groups of 8 addends at fixed offsets from one address, the next group in flight
    const float* g = V.b + (e + (e >> 3));
#define NB_CHAIN_LOAD(dst, grp) _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) dst[j_] = g[9 * (grp) + j_];
#define NB_CHAIN_ADD(src) _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) s = s + src[j_]; \
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    float va[8], vb[8];
    NB_CHAIN_LOAD(va, 0)
    for (; e + 24 <= end; e += 16, g += 18) {  // va holds [e, e + 8) here
      NB_CHAIN_LOAD(vb, 1)
      NB_CHAIN_ADD(va)
      NB_CHAIN_LOAD(va, 2)
      NB_CHAIN_ADD(vb)
    }
    if (e + 16 <= end) {
      NB_CHAIN_LOAD(vb, 1)
      NB_CHAIN_ADD(va)
      NB_CHAIN_ADD(vb)
      e += 16;
    } else {
      NB_CHAIN_ADD(va)
      e += 8;
    }
#undef NB_CHAIN_LOAD
#undef NB_CHAIN_ADD
  }
  for (; e < end; ++e) s = s + V[e];
}

// ---- scans over the lanes by DPP (a ds_bpermute per step costs a trip through the LDS pipe) -----------------------------------
// The value CTRL brings to this lane, 0 where it brings none (0, 0 is the identity step).
template <int CTRL, int ROWS> __device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false);
}
template <int CTRL, int ROWS> __device__ __forceinline__ xsum::Step dpp_step(xsum::Step v) {
  return xsum::Step{dpp_or_zero<CTRL, ROWS>(v.a0), dpp_or_zero<CTRL, ROWS>(v.a1)};
}
// inclusive scan over the wave's 64 lanes, in lane order
__device__ __forceinline__ xsum::Step wave_scan_steps(xsum::Step v) {
  v = xsum::compose(dpp_step<0x111, 0xf>(v), v);  // row_shr:1
  v = xsum::compose(dpp_step<0x112, 0xf>(v), v);  // row_shr:2
  v = xsum::compose(dpp_step<0x114, 0xf>(v), v);  // row_shr:4
  v = xsum::compose(dpp_step<0x118, 0xf>(v), v);  // row_shr:8
  v = xsum::compose(dpp_step<0x142, 0xa>(v), v);  // row_bcast:15 into rows 1 and 3
  v = xsum::compose(dpp_step<0x143, 0xc>(v), v);  // row_bcast:31 into rows 2 and 3
  return v;
}
// the value of the lane before (wave_shr:1), the identity in lane 0
__device__ __forceinline__ xsum::Step wave_prev_step(xsum::Step v) { return dpp_step<0x138, 0xf>(v); }

// ---- NW waves working on one node --------------------------------------------------------------------------------
// NW == 1 needs no barrier and no LDS (a wave runs in lock-step); NW > 1 is a whole work-group.
template <int NW> struct Scratch {
  xsum::Step w[2][NW];     // wave totals of a scan round; two sets, taken in turn (a round that succeeds has one barrier)
  uint32_t wkind[2][NW];   // per wave: 1 a negative increment, 2 a positive one, 4 a step too large for 32-bit sums
  float redf[2][NW];
  int bad;
  uint32_t bad_s;
};

template <int NW> __device__ __forceinline__ void group_sync() {
  if constexpr (NW > 1) {
    __syncthreads();
  } else {  // LDS written by one lane, read by another lane of the same wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int NW> __device__ __forceinline__ unsigned group_sum(unsigned v, unsigned* slot, int tid) {
  for (int d = 32; d >= 1; d >>= 1) v += (unsigned)__shfl_xor((int)v, d, 64);
  if constexpr (NW == 1) {
    return v;
  } else {
    if ((tid & 63) == 0) slot[tid >> 6] = v;
    __syncthreads();
    unsigned t = 0;
    for (int w = 0; w < NW; ++w) t += slot[w];
    __syncthreads();
    return t;
  }
}

#ifdef NB_FOLD_TIMING
__device__ unsigned long long g_fold_t[16];
#define NB_FT_ADD(k, v) { if (tid == 0) atomicAdd(&g_fold_t[k], (unsigned long long)(v)); }
#else
#define NB_FT_ADD(k, v)
#endif
// One coordinate (comp 0: x, 1: y) of the fold of bvh_tree.rs:58-61 over P[0, len): min, max, and the sum exactly as
// the sequential chain rounds it.  Every thread of the group returns the same values.  The two coordinates are
// independent chains: they run on different work-groups.
// EPT: consecutive addends per thread and scan (the scan's fixed cost is per thread: more addends each = cheaper).
// chain_off: how many addends of the chain came before P[0] (P may be a window of the node staged in LDS).
template <int NW, int EPT, class Src>
__device__ __forceinline__ void exact_fold(const Src& P, int begin, int len, float s_in, int tid, Scratch<NW>* sh,
                                           float& out_sum, float& out_min, float& out_max, int& stops, int chain_off = 0) {
  // the chain over P[begin, len) (one coordinate), entered with the running sum s_in (0.0 and begin = 0 for a whole node)
  static_assert(NW > 1 && EPT == 8, "a work-group of several waves, a row of the stage per addend of a thread");
  constexpr int TILE = NW * 64 * EPT;
  constexpr int kSmall = 1 << 24;  // wave totals below this: NW of them add up without wrapping
  const int lane = tid & 63, wave = tid >> 6;
  float s = s_in;  // uniform across the group
  float mn = kMaxF, mx = 0.f;
  int pos = begin;
  int par = 0;
  bool seq = false;       // decided from s: a chain at 0.0 is not in any binade yet
  bool foreseen = false;  // the last thing done was a run of real adds up to a crossing that was seen coming
  while (pos < len) {
    xsum::Chain ch;
    if (!seq) seq = !xsum::chain_open(s, ch);
    const int gp = chain_off + pos;  // addends behind the chain
    // A chain of same-signed addends leaves its binade about every time the number of addends doubles: near the
    // start of the chain short scans lose less work to the restart than full ones.
    int span = gp > 64 ? gp : 64;
    span = span < TILE ? span : TILE;
    int seq_cnt = gp == 0 ? kChainStart : kSeqRun;
    if (!seq && !foreseen) {
      // ... and where it leaves can be seen coming: gp addends made S ulps, so 2^24 is another gp * (2^24 - S) / S
      // addends away if they go on like that.  The scan stops a little short of that point and real adds carry the
      // chain across (they need no binade), instead of a whole scan failing there and starting over.  A guess: a
      // chain that does something else is scanned and stopped as ever.
      const float rem = (float)gp * ((float)(xsum::kHi - ch.S) / (float)ch.S);
      const int irem = rem < 1.0e9f ? (int)rem : 1000000000;
      const int safe = irem - (irem >> 5) - 8;
      if (safe < 64) {
        seq = true;
        seq_cnt = irem + (irem >> 4) + kSeqRun;
        seq_cnt = seq_cnt < NW * 64 ? seq_cnt : NW * 64;
        foreseen = true;
      } else {
        span = span < safe ? span : safe;
      }
    } else {
      foreseen = false;
    }
    if (seq) {  // real adds, by every wave for itself: nothing to wait for, nothing to pass on
#ifdef NB_FOLD_TIMING
      const long long tq0 = wall_clock64();
#endif
      const int cnt = len - pos < seq_cnt ? len - pos : seq_cnt;
      chain_run_all(P, pos, cnt, s);
      for (int i = tid; i < cnt; i += NW * 64) {
        const float v = P[pos + i];
        mn = sse_min(mn, v);
        mx = sse_max(mx, v);
      }
      pos += cnt;
      seq = false;
#ifdef NB_FOLD_TIMING
      NB_FT_ADD(0, wall_clock64() - tq0) NB_FT_ADD(2, 1)
#endif
      continue;
    }
#ifdef NB_FOLD_TIMING
    const long long tr0 = wall_clock64();
#endif
    const int limit = pos + span < len ? pos + span : len;
    const int base = pos + tid * EPT;
    xsum::Step f[EPT];
    xsum::Step t{0u, 0u};
    int imin = 0, imax = 0;  // extremes of the increments: their signs, and whether 32-bit totals can be trusted
#pragma unroll
    for (int j = 0; j < EPT; ++j) f[j] = xsum::Step{0u, 0u};
    if (base < limit) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        if (base + j < limit) {
          const float v = P[base + j];
          f[j] = xsum::step_of(v, ch.sign, ch.E);
          mn = sse_min(mn, v);
          mx = sse_max(mx, v);
          const int i0 = (int)f[j].a0, i1 = (int)f[j].a1;
          imin = min(imin, min(i0, i1));
          imax = max(imax, max(i0, i1));
        }
        t = xsum::compose(t, f[j]);
      }
    }
    const xsum::Step inc = wave_scan_steps(t);  // inclusive scan inside the wave
    xsum::Step ex = wave_prev_step(inc);        // exclusive: everything before my addends
    {  // (an increment is below 2^22 in size unless it is the poison: 512 of one sign stay below 2^31)
      const uint32_t kind = (__ballot(imin < 0) != 0ull ? 1u : 0u) | (__ballot(imax > 0) != 0ull ? 2u : 0u) |
                            (__ballot(imax >= (1 << 23)) != 0ull ? 4u : 0u);
      if (lane == 63) { sh->w[par][wave] = inc; sh->wkind[par][wave] = kind; }
    }
    if (tid == 0) sh->bad = INT_MAX;
    __syncthreads();
    // every wave scans the NW wave totals for itself (lanes 0..NW-1): no second barrier
    xsum::Step wi{0u, 0u};
    uint32_t wk = 0u;
    if (lane < NW) { wi = sh->w[par][lane]; wk = sh->wkind[par][lane]; }
    const int w0 = (int)wi.a0, w1 = (int)wi.a1;
    if (w0 <= -kSmall || w0 >= kSmall || w1 <= -kSmall || w1 >= kSmall) wk |= 4u;
    const uint32_t kind = (__ballot((wk & 1u) != 0u) != 0ull ? 1u : 0u) | (__ballot((wk & 2u) != 0u) != 0ull ? 2u : 0u) |
                          (__ballot((wk & 4u) != 0u) != 0ull ? 4u : 0u);
    static_assert(NW <= 16, "the wave totals are scanned inside one row of 16 lanes");
    wi = xsum::compose(dpp_step<0x111, 0xf>(wi), wi);
    if constexpr (NW > 2) wi = xsum::compose(dpp_step<0x112, 0xf>(wi), wi);
    if constexpr (NW > 4) wi = xsum::compose(dpp_step<0x114, 0xf>(wi), wi);
    if constexpr (NW > 8) wi = xsum::compose(dpp_step<0x118, 0xf>(wi), wi);
    xsum::Step tot;
    tot.a0 = (uint32_t)__builtin_amdgcn_readlane((int)wi.a0, NW - 1);
    tot.a1 = (uint32_t)__builtin_amdgcn_readlane((int)wi.a1, NW - 1);
    par ^= 1;
    // Increments of one sign only (the usual case: coordinates of one sign) move S one way: every intermediate S lies
    // between the first and the last, so the last one being inside the binade says all were.  The totals are exact:
    // every wave's is below 2^24 in size, NW of them cannot wrap.
    if (kind != 3u && kind < 4u && xsum::in_binade(xsum::apply(ch.S, tot))) {
      s = xsum::chain_value(ch, xsum::apply(ch.S, tot));
      pos += span;
#ifdef NB_FOLD_TIMING
      NB_FT_ADD(1, wall_clock64() - tr0) NB_FT_ADD(3, 1)
#endif
      continue;
    }
    if (wave > 0) {
      xsum::Step pw;  // all the waves before mine
      pw.a0 = (uint32_t)__builtin_amdgcn_readlane((int)wi.a0, wave - 1);
      pw.a1 = (uint32_t)__builtin_amdgcn_readlane((int)wi.a1, wave - 1);
      ex = xsum::compose(pw, ex);
    }
    uint32_t S = xsum::apply(ch.S, ex);
    int bad = INT_MAX;
    uint32_t bs = 0u;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      if (base + j < limit && bad == INT_MAX) {
        const uint32_t nx = xsum::apply(S, f[j]);
        if (!xsum::in_binade(nx)) {
          bad = tid * EPT + j;
          bs = S;
        } else {
          S = nx;
        }
      }
    }
    int first_bad = bad;
    for (int d = 32; d >= 1; d >>= 1) {
      const int o = __shfl_xor(first_bad, d, 64);
      first_bad = o < first_bad ? o : first_bad;
    }
    if (lane == 0 && first_bad != INT_MAX) atomicMin(&sh->bad, first_bad);
    __syncthreads();
    first_bad = sh->bad;
    if (first_bad == INT_MAX) {
      s = xsum::chain_value(ch, xsum::apply(ch.S, tot));
      pos += span;
    } else {  // the chain is exact up to the addend before first_bad; that addend takes a real add
      if (bad == first_bad) sh->bad_s = bs;
      __syncthreads();
      s = xsum::chain_value(ch, sh->bad_s);
      pos += first_bad;
      seq = true;
      ++stops;
    }
    __syncthreads();  // sh->bad, bad_s are rewritten by the next round
#ifdef NB_FOLD_TIMING
    NB_FT_ADD(1, wall_clock64() - tr0) NB_FT_ADD(3, 1) NB_FT_ADD(7, 1)
#endif
  }
  for (int d = 32; d >= 1; d >>= 1) {
    mn = sse_min(mn, __shfl_xor(mn, d, 64));
    mx = sse_max(mx, __shfl_xor(mx, d, 64));
  }
  if (lane == 0) { sh->redf[0][wave] = mn; sh->redf[1][wave] = mx; }
  __syncthreads();
  for (int w = 0; w < NW; ++w) {
    mn = sse_min(mn, sh->redf[0][w]);
    mx = sse_max(mx, sh->redf[1][w]);
  }
  __syncthreads();
  out_sum = s;
  out_min = mn;
  out_max = mx;
}

// exact_fold over P[lo, hi) of a node in memory, a window of NW * 64 * EPT points at a time out of LDS.  The chain is a
// sequence of short dependent rounds (a scan, a stop at every power of two the sum crosses, a few real adds, the next scan):
// read from memory every round pays a trip to L2; the window is fetched once, the next one while this one is folded.
template <int NW, int EPT>
__device__ __forceinline__ void staged_fold(const float2* __restrict__ P, int lo, int hi, float s_in, int comp, int tid,
                                            Scratch<NW>* sh, float* stage, float& out_sum, float& out_min, float& out_max,
                                            int& stops) {
  constexpr int NT = NW * 64, T = NT * EPT;
  const StageView<NT> view{stage};
  const float* __restrict__ X = reinterpret_cast<const float*>(P) + comp;
  float s = s_in, mn = kMaxF, mx = 0.f;
  float nx[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = lo + tid + k * NT;
    nx[k] = e < hi ? X[2 * (size_t)e] : 0.f;
  }
  for (int t0 = lo; t0 < hi; t0 += T) {
    const int cnt = hi - t0 < T ? hi - t0 : T;
    __syncthreads();  // the window before this one has been read to the end
#pragma unroll
    for (int k = 0; k < EPT; ++k) stage[StageView<NT>::at(tid + k * NT)] = nx[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = t0 + T + tid + k * NT;
      nx[k] = e < hi ? X[2 * (size_t)e] : 0.f;
    }
    float ts, tmn, tmx;
    exact_fold<NW, EPT>(view, 0, cnt, s, tid, sh, ts, tmn, tmx, stops, t0);
    s = ts;
    mn = sse_min(mn, tmn);
    mx = sse_max(mx, tmx);
  }
  out_sum = s;
  out_min = mn;
  out_max = mx;
}

// :70-73 — which axis splits closer to the middle; returns the split point m (predicate-true side comes first)
__device__ __forceinline__ int choose_axis(int len, int cxs, int cys, bool& on_x) {
  const int half = len / 2;
  const int hori = half > cxs ? half - cxs : cxs - half;
  const int vert = half > cys ? half - cys : cys - half;
  on_x = vert > hori;
  return on_x ? cxs : cys;
}

// Children of `node` ([b, b+m) and [b+m, b+len)) with the ids first, first + 1: records and keys.  One thread.
// leaf[] says which children need no further splitting.
// depth_max: where the tree's depth is collected; null: the global flag, one atomic per node (the subtree kernel collects per
// work-group in LDS and reports once: thousands of atomics on one address are the slowest thing a build can do).
__device__ __forceinline__ void make_children(const BvhPtrs& a, int node, int first, int b, int len, int m, int leaf_size,
                                              bool leaf[2], int* depth_max = nullptr, int depth = -1) {
  const int d = depth >= 0 ? depth : a.ndepth[node];  // (a caller that knows the node's depth saves the trip to memory)
  a.nchild[node] = first;
  for (int side = 0; side < 2; ++side) {
    const int id = first + side;
    const int cl = side ? len - m : m;
    a.nbegin[id] = side ? b + m : b;
    a.nlen[id] = cl;
    a.nparent[id] = node;
    a.nchild[id] = -1;
    a.ndepth[id] = d + 1;
    bool lf = !(cl > leaf_size);  // :78-88
    if (!lf && d + 1 >= kBvhKeyDepth) { a.flags[kBvhFallback] = 1; lf = true; }
    leaf[side] = lf;
    a.nleaf[id] = lf ? 1 : 0;
    a.npre[id] = -1;
    a.ndone[id] = 0;
    a.nbad[id] = 0;
  }
  atomicMax(depth_max ? depth_max : &a.flags[kBvhMaxDepth], d + 1);
}
// `count` fresh node ids or -1, without a verdict on the build (the caller has a smaller request to fall back on)
__device__ __forceinline__ int try_alloc_nodes(const BvhPtrs& a, int count) {
  const int first = atomicAdd(&a.flags[kBvhNodeCount], count);
  if (first + count > a.cap) {
    atomicSub(&a.flags[kBvhNodeCount], count);
    return -1;
  }
  return first;
}
// two fresh node ids, or -1 with the fallback flag up when the buffers are full
__device__ __forceinline__ int alloc_nodes(const BvhPtrs& a, int count) {
  const int first = atomicAdd(&a.flags[kBvhNodeCount], count);
  if (first + count > a.cap) {
    a.flags[kBvhFallback] = 1;
    return -1;
  }
  return first;
}

// ---- init --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bvh_init(BvhPtrs a, const float2* __restrict__ pos, int n, unsigned long long* __restrict__ stamp_begin,
                                                unsigned long long* __restrict__ stamp_prev_end) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && (stamp_begin || stamp_prev_end)) {  // phase clock (capi.hip, PhaseStamps): this step begins where the one before ends
    const unsigned long long now = (unsigned long long)wall_clock64();
    if (stamp_begin) *stamp_begin = now;
    if (stamp_prev_end) *stamp_prev_end = now;
  }
  if (i < n) {
    const float2 p = pos[i];
    a.P[i] = p;
    a.ID[i] = (uint32_t)i;
    a.pair[i] = 0ull;  // (the pair slots clean themselves; a build that was abandoned half-way may have left some)
    if (p.x != p.x || p.y != p.y) a.flags[kBvhFallback] = 1;
  }
  for (int j = i + 1; j < a.cap; j += gridDim.x * 256) a.ndepth[j] = -1;  // not a node (yet): ids are handed out in ranges
  if (n > kSub && i < (n + kChunk - 1) / kChunk) {  // the root's chunks
    a.ch_rec[i] = make_int4(0, i, 0, n);
  }
  if (i == 0) {  // the top call is unconditional: the root is a Root whatever its length (main.rs:400)
    a.nbegin[0] = 0;
    a.nlen[0] = n;
    a.nparent[0] = -1;
    a.nchild[0] = -1;
    a.ndepth[0] = 0;
    a.nleaf[0] = 0;
    a.npre[0] = 0;  // the root comes first
    a.ndone[0] = 0;
    a.nbad[0] = 0;
    a.flags[kBvhNodeCount] = 1;
    a.flags[kBvhTopCount] = 1;
    a.topq[0] = 0;
    if (n > kSub) {
      a.bigcount[0] = 1;
      a.bigq[0] = 0;
      a.chunkcount[0] = (n + kChunk - 1) / kChunk;
      a.nchunk0[0] = 0;
    } else {
      a.flags[kBvhSubCount] = 1;
      a.subq[0] = 0;
    }
  }
}

// ---- long nodes, one level: chunk sums and chunk runs (only nodes longer than kRunLen) ------------------------------------
// A long chain is cut into chunks of kChunk addends.  bvh_chunk_sums: f64 sums (and the box) of every chunk, in parallel;
// their prefix PREDICTS the binade the chain is in when it reaches a chunk.  bvh_chunk_runs: every chunk's run
// (exact_sum.h) for that binade, in parallel.  bvh_big_fold then walks the chunks: a run is used iff the true state is in
// the predicted binade and the run's bounds hold from it; the other chunks (the first one, the ~log2(len / kChunk) in which
// the sum crosses a power of two, wrong guesses) are added by the scan, as before.
__global__ __launch_bounds__(256) void bvh_chunk_sums(BvhPtrs a, int level) {
  __shared__ double red[6][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = a.chunkcount[level];
  const int4* ch_rec = a.ch_rec + (size_t)(level & 1) * a.cap_chunk;
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
    const int4 cr = ch_rec[c];
    const int ci = cr.y, len = cr.w;
    if (len <= kRunLen) continue;
    const float2* P = a.P + cr.z;
    const int lo = ci * kChunk, hi = lo + kChunk < len ? lo + kChunk : len;
    double sx = 0.0, sy = 0.0;
    Box bx;
    for (int i = lo + tid; i < hi; i += 256) {
      const float2 q = P[i];
      sx += (double)q.x;
      sy += (double)q.y;
      bx.add(q);
    }
    for (int d = 32; d >= 1; d >>= 1) {
      sx += __shfl_xor(sx, d, 64);
      sy += __shfl_xor(sy, d, 64);
    }
    bx.reduce_wave();
    if (lane == 0) {
      red[0][wave] = sx; red[1][wave] = sy;
      red[2][wave] = bx.mnx; red[3][wave] = bx.mny; red[4][wave] = bx.mxx; red[5][wave] = bx.mxy;
    }
    __syncthreads();
    if (tid == 0) {
      double tx = 0.0, ty = 0.0;
      float mnx = kMaxF, mny = kMaxF, mxx = 0.f, mxy = 0.f;
      for (int w = 0; w < 4; ++w) {
        tx += red[0][w]; ty += red[1][w];
        mnx = sse_min(mnx, (float)red[2][w]); mny = sse_min(mny, (float)red[3][w]);
        mxx = sse_max(mxx, (float)red[4][w]); mxy = sse_max(mxy, (float)red[5][w]);
      }
      a.ch_sum[c] = make_double2(tx, ty);
      a.ch_box[c] = make_float4(mnx, mny, mxx, mxy);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ xsum::Run shfl_down_run(const xsum::Run& r, int d) {
  xsum::Run o;
  o.a0 = __shfl_down(r.a0, d, 64); o.a1 = __shfl_down(r.a1, d, 64);
  o.lo0 = __shfl_down(r.lo0, d, 64); o.lo1 = __shfl_down(r.lo1, d, 64);
  o.hi0 = __shfl_down(r.hi0, d, 64); o.hi1 = __shfl_down(r.hi1, d, 64);
  return o;
}

// Runs are prepared per RUN chunk = m consecutive chunks of kChunk addends, m a power of two such that a node has at most
// 64 of them: the chain's walk over the runs is sequential (~0.7 us per run), the preparation is not.
__host__ __device__ inline int run_mult(int len) {
  const int nch = (len + kChunk - 1) / kChunk;
  int m = 1;
  while (nch > 64 * m) m <<= 1;
  return m;
}
constexpr int kGapCap = 2048;  // real adds around powers of two, per chain and walk, fetched ahead into LDS
constexpr int kRunRec = 24;  // ints per chunk and coordinate: sign, E_a (0: no run), E_b, u0, u1, run A (6), run B (6), [17] where its real adds lie in `gaps` (bvh_big_fold), pad
__device__ __forceinline__ void store_run(int* o, const xsum::Run& r) {
  o[0] = r.a0; o[1] = r.a1; o[2] = r.lo0; o[3] = r.lo1; o[4] = r.hi0; o[5] = r.hi1;
}
// ordered reduction over the wave: lane 0 ends with r(lane 0) then r(lane 1) then ...
__device__ __forceinline__ xsum::Run wave_run_in_order(xsum::Run r, int lane) {
  for (int d = 1; d < 64; d <<= 1) {
    const xsum::Run o = shfl_down_run(r, d);
    if ((lane & (2 * d - 1)) == 0) r = xsum::run_then(r, o);
  }
  return r;
}

// One chunk's runs.  Every thread takes PER consecutive addends and the f64 prefix PREDICTS where the chain is at each
// thread.  If the chunk stays in one binade: one run (A) over all of it.  If the prefix crosses ONE power of two B inside
// the chunk (same-signed data): run A (old binade) over the threads safely below B (1 - kRunMargin), run B (new binade)
// over the threads safely above B (1 + kRunMargin), and the threads in between are left to real adds (u0 .. u1).
__global__ __launch_bounds__(256) void bvh_chunk_runs(BvhPtrs a, int level) {
  constexpr int PER = kChunk / 256;
  __shared__ double redd[4][4];
  __shared__ xsum::Run redr[2][2][4];  // [coordinate][A / B][wave]
  __shared__ int redi[2][4][4];        // [coordinate][#A, #B, #active, shape ok][wave]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = a.chunkcount[level];
  const int4* ch_rec = a.ch_rec + (size_t)(level & 1) * a.cap_chunk;
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
    const int4 cr = ch_rec[c];
    const int node = cr.x, ci = cr.y, len = cr.w;
    if (len <= kRunLen) continue;
    constexpr int m = 1;  // runs are prepared per chunk; bvh_big_fold merges them into at most 64 per node
    int* rec = a.ch_run + (size_t)c * 2 * kRunRec;
    if (ci == 0) {  // the chain starts here: nothing to predict
      if (tid == 0) { rec[1] = 0; rec[kRunRec + 1] = 0; }
      continue;
    }
    // where the chain should be when it gets here: the f64 sums of the node's chunks before this one
    const int c0 = c - ci;  // a node's chunks are consecutive
    const int nchn = (len + kChunk - 1) / kChunk;
    double px = 0.0, py = 0.0, tx = 0.0, ty = 0.0;
    for (int i = tid; i < ci + m && i < nchn; i += 256) {
      const double2 v = a.ch_sum[c0 + i];
      if (i < ci) { px += v.x; py += v.y; } else { tx += v.x; ty += v.y; }
    }
    for (int d = 32; d >= 1; d >>= 1) {
      px += __shfl_xor(px, d, 64);
      py += __shfl_xor(py, d, 64);
      tx += __shfl_xor(tx, d, 64);
      ty += __shfl_xor(ty, d, 64);
    }
    if (lane == 0) { redd[0][wave] = px; redd[1][wave] = py; redd[2][wave] = tx; redd[3][wave] = ty; }
    __syncthreads();
    px = redd[0][0] + redd[0][1] + redd[0][2] + redd[0][3];
    py = redd[1][0] + redd[1][1] + redd[1][2] + redd[1][3];
    const double2 tot = make_double2(redd[2][0] + redd[2][1] + redd[2][2] + redd[2][3], redd[3][0] + redd[3][1] + redd[3][2] + redd[3][3]);
    __syncthreads();
    const float2* P = a.P + a.nbegin[node] + (size_t)ci * kChunk;               // this run chunk
    const int cnt = (ci + m) * kChunk < len ? m * kChunk : len - ci * kChunk;  // its addends
    const int seg = m * PER;                                                    // consecutive addends per thread
    const int base = tid * seg;
    // does either coordinate cross a power of two in this chunk?  Only then the threads need their own f64 prefix
    xsum::Chain cax, cbx, cay, cby;
    const bool hx = xsum::chain_open((float)px, cax) && xsum::chain_open((float)(px + tot.x), cbx) && cax.sign == cbx.sign;
    const bool hy = xsum::chain_open((float)py, cay) && xsum::chain_open((float)(py + tot.y), cby) && cay.sign == cby.sign;
    const bool crossing = (hx && cbx.E == cax.E + 1u) || (hy && cby.E == cay.E + 1u);  // uniform
    double sx0 = 0.0, sy0 = 0.0, lx = 0.0, ly = 0.0;
    if (crossing) {  // f64 prefix of my first addend
      for (int j = 0; j < seg && base + j < cnt; ++j) {
        const float2 qq = P[base + j];
        lx += (double)qq.x;
        ly += (double)qq.y;
      }
      double ix = lx, iy = ly;
      for (int d = 1; d < 64; d <<= 1) {
        const double ox = __shfl_up(ix, d, 64), oy = __shfl_up(iy, d, 64);
        if (lane >= d) { ix += ox; iy += oy; }
      }
      if (lane == 63) { redd[0][wave] = ix; redd[1][wave] = iy; }
      __syncthreads();
      sx0 = px + ix - lx;
      sy0 = py + iy - ly;
      for (int w = 0; w < wave; ++w) { sx0 += redd[0][w]; sy0 += redd[1][w]; }
    }
    const bool active = base < cnt;
    for (int comp = 0; comp < 2; ++comp) {
      const double s0 = comp ? sy0 : sx0, s1 = s0 + (comp ? ly : lx);
      const xsum::Chain ca = comp ? cay : cax, cb = comp ? cby : cbx;
      const bool have = (comp ? hy : hx) && (cb.E == ca.E || cb.E == ca.E + 1u);
      int cls = 2;  // 0: run A, 1: run B, 2: left to real adds
      if (have) {
        if (cb.E == ca.E) {
          cls = 0;
        } else {
          const double sgn = ca.sign ? -1.0 : 1.0;
          const double B = __builtin_ldexp(1.0, (int)cb.E - 127);
          const double lo = B * (1.0 - xsum::kRunMargin), hi = B * (1.0 + xsum::kRunMargin);
          const double qs = sgn * s0, qe = sgn * s1;
          if (qs < lo && qe < lo) cls = 0;
          else if (qs > hi && qe > hi) cls = 1;
        }
      }
      if (!active) cls = 3;
      const unsigned long long mA = __builtin_amdgcn_ballot_w64(cls == 0), mB = __builtin_amdgcn_ballot_w64(cls == 1);
      const unsigned long long mAct = __builtin_amdgcn_ballot_w64(active);
      if (lane == 0) {  // inside the wave: A threads first, B threads last (the active threads are a prefix of the lanes)
        const int nA = __builtin_popcountll(mA), nB = __builtin_popcountll(mB), nAct = __builtin_popcountll(mAct);
        const unsigned long long lowA = nA >= 64 ? ~0ull : ((1ull << nA) - 1ull);
        const unsigned long long lowNB = (nAct - nB) >= 64 ? ~0ull : ((1ull << (nAct - nB)) - 1ull);
        redi[comp][0][wave] = nA;
        redi[comp][1][wave] = nB;
        redi[comp][2][wave] = nAct;
        redi[comp][3][wave] = (mA == lowA && mB == (mAct & ~lowNB)) ? 1 : 0;
      }
      xsum::Run r = xsum::run_none();
      if (cls == 0 || cls == 1) {
        const xsum::Chain& ch = cls ? cb : ca;
        for (int j = 0; j < seg && base + j < cnt; ++j) {
          const float2 qq = P[base + j];
          r = xsum::run_then(r, xsum::run_of(xsum::step_of(comp ? qq.y : qq.x, ch.sign, ch.E)));
        }
      }
      // ordered reduction: a wave is usually all A or all B; only a wave that straddles the crossing reduces twice
      xsum::Run ra = xsum::run_none(), rb = xsum::run_none();
      if (mB == 0ull) {
        ra = wave_run_in_order(cls == 0 ? r : xsum::run_none(), lane);
      } else if (mA == 0ull) {
        rb = wave_run_in_order(cls == 1 ? r : xsum::run_none(), lane);
      } else {
        ra = wave_run_in_order(cls == 0 ? r : xsum::run_none(), lane);
        rb = wave_run_in_order(cls == 1 ? r : xsum::run_none(), lane);
      }
      if (lane == 0) { redr[comp][0][wave] = ra; redr[comp][1][wave] = rb; }
      __syncthreads();
      if (tid == 0) {
        int* o = rec + comp * kRunRec;
        int nA = 0, nB = 0, nAct = 0;
        for (int w = 0; w < 4; ++w) { nA += redi[comp][0][w]; nB += redi[comp][1][w]; nAct += redi[comp][2][w]; }
        // A must be a prefix of the chunk's threads and B a suffix (same-signed data, one crossing)
        bool ok = have;
        bool a_over = false, b_begun = false;
        for (int w = 0; w < 4; ++w) {
          const int wa = redi[comp][0][w], wb = redi[comp][1][w], wact = redi[comp][2][w];
          if (!redi[comp][3][w]) ok = false;
          if (a_over && wa > 0) ok = false;
          if (b_begun && wb != wact) ok = false;
          if (wa < wact) a_over = true;
          if (wb > 0) b_begun = true;
        }
        o[0] = (int)ca.sign;
        o[1] = ok ? (int)ca.E : 0;
        o[2] = (int)cb.E;
        const int u0 = nA * seg < cnt ? nA * seg : cnt;
        const int u1 = (nAct - nB) * seg < cnt ? (nAct - nB) * seg : cnt;
        o[3] = u0;
        o[4] = u1 > u0 ? u1 : u0;
        xsum::Run A = redr[comp][0][0], Bq = redr[comp][1][0];
        for (int w = 1; w < 4; ++w) { A = xsum::run_then(A, redr[comp][0][w]); Bq = xsum::run_then(Bq, redr[comp][1][w]); }
        store_run(o + 5, A);
        store_run(o + 11, Bq);
      }
      __syncthreads();
    }
  }
}

// ---- long nodes, one level: fold ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void bvh_big_fold(BvhPtrs a, int level, int use_runs) {
  constexpr int kRecBatch = 64;
  __shared__ Scratch<8> sh;
  __shared__ __attribute__((aligned(16))) int recs[kRecBatch * kRunRec];
  __shared__ float stage[StageView<512>::kWords];
  __shared__ float gaps[kGapCap];
  __shared__ int gap_total;
  const int tid = threadIdx.x;
  const int comp = blockIdx.y;  // 0: x, 1: y
  const int nq = a.bigcount[level];
  const int* queue = a.bigq + (size_t)(level & 1) * a.cap_big;
  int stops = 0;
  for (int qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    const int node = queue[qi];
    const int b = a.nbegin[node], len = a.nlen[node];
    const float2* P = (const float2*)(a.P + b);
    float sum, mn, mx;
#ifdef NB_FOLD_TIMING
    const long long tn0 = wall_clock64();
#endif
    if (use_runs && len > kRunLen) {
      const int c0 = a.nchunk0[node], nch = (len + kChunk - 1) / kChunk;
      const int rm = run_mult(len), rlen = rm * kChunk, nrun = (nch + rm - 1) / rm;  // run chunks: <= 64 of them
      sum = 0.f;
      int used = 0;
      {  // the chunks' runs are merged, rm at a time, into the <= 64 runs the chain will walk (one thread per merged run)
        const int b0 = 0, nb = nrun;
        int glen = 0;
#ifdef NB_FOLD_TIMING
        long long tr0 = wall_clock64();
#endif
        if (tid < nrun) {
          const int f0 = tid * rm, f1 = f0 + rm < nch ? f0 + rm : nch;
          xsum::Run A = xsum::run_none(), B = xsum::run_none();
          int sign = 0, Ea = 0, Eb = 0, u0 = 0, u1 = 0, pos = 0;
          bool valid = true, crossed = false;
          for (int f = f0; f < f1 && valid; ++f) {
            const int* fr = a.ch_run + (size_t)(c0 + f) * 2 * kRunRec + comp * kRunRec;
            const int cntf = (f + 1) * kChunk < len ? kChunk : len - f * kChunk;
            const int fE = fr[1];
            if (fE == 0) { valid = false; break; }
            xsum::Run fa, fb;
            fa.a0 = fr[5]; fa.a1 = fr[6]; fa.lo0 = fr[7]; fa.lo1 = fr[8]; fa.hi0 = fr[9]; fa.hi1 = fr[10];
            fb.a0 = fr[11]; fb.a1 = fr[12]; fb.lo0 = fr[13]; fb.lo1 = fr[14]; fb.hi0 = fr[15]; fb.hi1 = fr[16];
            const bool whole = fr[3] >= cntf;  // one run over the whole chunk
            if (f == f0) { sign = fr[0]; Ea = fE; Eb = fE; }
            if (fr[0] != sign) { valid = false; break; }
            if (!crossed) {
              if (fE != Ea) { valid = false; break; }
              A = xsum::run_then(A, fa);
              if (!whole) {  // the power of two falls into this chunk
                u0 = pos + fr[3];
                u1 = pos + fr[4];
                B = fb;
                Eb = fr[2];
                crossed = true;
              }
            } else {
              if (fE != Eb || !whole) { valid = false; break; }  // a second crossing: leave it to the scan
              B = xsum::run_then(B, fa);
            }
            pos += cntf;
          }
          if (!crossed) { u0 = pos; u1 = pos; }
          int* o = recs + tid * kRunRec;
          o[0] = sign; o[1] = valid ? Ea : 0; o[2] = Eb; o[3] = u0; o[4] = u1;
          o[5] = A.a0; o[6] = A.a1; o[7] = A.lo0; o[8] = A.lo1; o[9] = A.hi0; o[10] = A.hi1;
          o[11] = B.a0; o[12] = B.a1; o[13] = B.lo0; o[14] = B.lo1; o[15] = B.hi0; o[16] = B.hi1;
          glen = valid ? u1 - u0 : 0;  // addends around a power of two, left to real adds
        }
        if (tid < 64) {
          // Those addends are fetched for all runs at once, ahead of the walk (a trip to memory at every crossing is 2 us
          // of a walk that takes 0.1 us per run): where each run's lie in `gaps` (nrun <= 64: the merging threads are
          // this wave).  Runs whose addends do not fit are left to the scan.
          int inc = glen;
          for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if ((tid & 63) >= d) inc += o;
          }
          const int goff = inc - glen;
          if (tid < nrun) {
            int* o = recs + tid * kRunRec;
            o[17] = goff;
            if (goff + glen > kGapCap) o[1] = 0;
#ifdef NB_FORCE_SCANS
            if ((tid % NB_FORCE_SCANS) == 1) o[1] = 0;  // (test builds) every so many runs go to the scan instead
#endif
          }
          if (tid == 63) gap_total = inc < kGapCap ? inc : kGapCap;
        }
        __syncthreads();
#ifdef NB_FOLD_TIMING
        { const long long t = wall_clock64(); NB_FT_ADD(8, t - tr0) tr0 = t; }
#endif
        {
          const float* __restrict__ X = reinterpret_cast<const float*>(P) + comp;
          const int G = gap_total;
          for (int g = tid; g < G; g += 512) {
            int lo_r = 0, hi_r = nrun - 1;  // the run whose addends include the g-th: last one with offset <= g
            while (lo_r < hi_r) {
              const int mid = (lo_r + hi_r + 1) >> 1;
              if (recs[mid * kRunRec + 17] <= g) lo_r = mid; else hi_r = mid - 1;
            }
            const int* o = recs + lo_r * kRunRec;
            const int k = g - o[17];
            if (o[1] != 0 && k < o[4] - o[3]) gaps[g] = X[2 * (size_t)(lo_r * rlen + o[3] + k)];
          }
        }
        __syncthreads();
#ifdef NB_FOLD_TIMING
        { const long long t = wall_clock64(); NB_FT_ADD(9, t - tr0) NB_FT_ADD(13, gap_total) tr0 = t; }
#endif
        int ci = b0;
        while (ci < b0 + nb) {  // ... and the chain walks through them in order
          // one wave runs ahead as long as the prepared runs hold (a few dozen instructions per chunk, no barrier) ...
          if (tid < 64) {
            // The chain's state stays (sign, E, S) from run to run; it turns into a float only where real adds need one.
            xsum::Chain ch;
            bool open = xsum::chain_open(sum, ch);
            for (; ci < b0 + nb; ++ci) {
              // the whole record in one go (LDS latency once per chunk, not once per field)
              const int4* rv = reinterpret_cast<const int4*>(recs + (ci - b0) * kRunRec);
              const int4 r0 = rv[0], r1 = rv[1], r2 = rv[2], r3 = rv[3], r4 = rv[4];
              if (r0.y == 0 || !open) break;  // no run prepared, or a state no run applies to: the scan's
              const int lo = ci * rlen, hi = lo + rlen < len ? lo + rlen : len;
              const int u0 = r0.w, u1 = r1.x;
              xsum::Chain tc = ch;
              bool good = true;
              int u = 0;
              if (u0 > 0) {  // part A: [lo, lo + u0)
                xsum::Run r;
                r.a0 = r1.y; r.a1 = r1.z; r.lo0 = r1.w; r.lo1 = r2.x; r.hi0 = r2.y; r.hi1 = r2.z;
                good = (int)tc.E == r0.y && (int)tc.sign == r0.x && xsum::run_fits(tc.S, r);
                tc.S = (uint32_t)((int)tc.S + ((tc.S & 1u) ? r.a1 : r.a0));
                ++u;
              }
              if (good && u1 > u0) {  // the addends around the power of two: real adds, out of `gaps`
                float t = xsum::chain_value(tc, tc.S);
                t = chain_lds<1>(gaps + r4.y, u1 - u0, t);
                good = xsum::chain_open(t, tc);
              }
              if (good && lo + u1 < hi) {  // part B: [lo + u1, hi), in the binade above
                good = (int)tc.E == r0.z && (int)tc.sign == r0.x;
                xsum::Run r;
                r.a0 = r2.w; r.a1 = r3.x; r.lo0 = r3.y; r.lo1 = r3.z; r.hi0 = r3.w; r.hi1 = r4.x;
                good = good && xsum::run_fits(tc.S, r);
                tc.S = (uint32_t)((int)tc.S + ((tc.S & 1u) ? r.a1 : r.a0));
                ++u;
              }
              if (!good) break;  // ... a chunk whose run does not hold is everybody's business
              ch = tc;
              used += u;
            }
            if (open) sum = xsum::chain_value(ch, ch.S);
            if (tid == 0) { sh.bad_s = xsum::f2u(sum); sh.bad = ci; }
          }
          __syncthreads();
#ifdef NB_FOLD_TIMING
          { const long long t = wall_clock64(); NB_FT_ADD(10, t - tr0) tr0 = t; }
#endif
          sum = xsum::u2f(sh.bad_s);
          ci = sh.bad;
          __syncthreads();
          if (ci >= b0 + nb) break;
          {  // this chunk by the scan, from the true state (its first part may still be taken whole)
            const int* rec = recs + (ci - b0) * kRunRec;
            const int lo = ci * rlen, hi = lo + rlen < len ? lo + rlen : len;
            float dmn, dmx;
            staged_fold<8, 8>(P, lo, hi, sum, comp, tid, &sh, stage, sum, dmn, dmx, stops);
            (void)rec;
            ++ci;
#ifdef NB_FOLD_TIMING
            { const long long t = wall_clock64(); NB_FT_ADD(11, t - tr0) NB_FT_ADD(12, 1) tr0 = t; }
#endif
          }
        }
        __syncthreads();
      }
      mn = kMaxF;
      mx = 0.f;
      for (int ci = tid; ci < nch; ci += 512) {
        const float4 bx = a.ch_box[c0 + ci];
        mn = sse_min(mn, comp ? bx.y : bx.x);
        mx = sse_max(mx, comp ? bx.w : bx.z);
      }
      for (int d = 32; d >= 1; d >>= 1) {
        mn = sse_min(mn, __shfl_xor(mn, d, 64));
        mx = sse_max(mx, __shfl_xor(mx, d, 64));
      }
      if ((tid & 63) == 0) { sh.redf[0][tid >> 6] = mn; sh.redf[1][tid >> 6] = mx; }
      __syncthreads();
      for (int w = 0; w < 8; ++w) {
        mn = sse_min(mn, sh.redf[0][w]);
        mx = sse_max(mx, sh.redf[1][w]);
      }
      __syncthreads();
      if (tid == 0 && used) atomicAdd(&a.flags[kBvhRunsUsed], used);
    } else {
      staged_fold<8, 8>(P, 0, len, 0.f, comp, tid, &sh, stage, sum, mn, mx, stops);
    }
#ifdef NB_FOLD_TIMING
    NB_FT_ADD(4, wall_clock64() - tn0) NB_FT_ADD(5, 1) NB_FT_ADD(6, len)
#endif
    if (tid == 0) {
      float* box = (float*)&a.nbox[node];
      box[comp] = mn;
      box[2 + comp] = mx;
      ((float*)&a.nmean[node])[comp] = sum / (float)len;  // :67
    }
  }
  if (tid == 0 && stops) atomicAdd(&a.flags[kBvhStops], stops);
}

// ---- long nodes: counts per chunk; the chunk that finishes last plans the split ------------------------------------
__global__ __launch_bounds__(256) void bvh_big_count(BvhPtrs a, int level, int leaf_size) {
  __shared__ unsigned red[2][4];
  __shared__ int last_flag;
  __shared__ int kid[2], kid_c0[2], kid_n[2], kid_b[2], kid_len[2], drawn[6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = a.chunkcount[level];
  const int4* ch_rec = a.ch_rec + (size_t)(level & 1) * a.cap_chunk;
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
    const int4 cr = ch_rec[c];
    const int node = cr.x, ci = cr.y, b = cr.z, len = cr.w;
    const float2 h = a.nmean[node];
    const int lo = ci * kChunk, hi = lo + kChunk < len ? lo + kChunk : len;
    unsigned cx = 0u, cy = 0u;
    for (int i = lo + tid; i < hi; i += 256) {
      const float2 q = a.P[b + i];
      const unsigned px = q.x > h.x, py = q.y > h.y;
      cx += px;
      cy += py;
      a.pred[b + i] = (uint8_t)(px | (py << 1));  // both axes: which one splits is known only when every chunk has counted
    }
    cx = group_sum<4>(cx, red[0], tid);
    cy = group_sum<4>(cy, red[1], tid);
    const int nch = (len + kChunk - 1) / kChunk;
    if (tid == 0) {
      a.ch_cx[c] = (int)cx;
      a.ch_cy[c] = (int)cy;
      __threadfence();
      last_flag = atomicAdd(&a.ndone[node], 1) == nch - 1;
    }
    __syncthreads();
    const bool last = last_flag != 0;
    __syncthreads();
    if (!last) continue;
    __threadfence();
    const int c0 = a.nchunk0[node];
    unsigned tx = 0u, ty = 0u;
    for (int i = tid; i < nch; i += 256) {
      tx += (unsigned)load_agent(a.ch_cx + c0 + i);
      ty += (unsigned)load_agent(a.ch_cy + c0 + i);
    }
    tx = group_sum<4>(tx, red[0], tid);
    ty = group_sum<4>(ty, red[1], tid);
    bool on_x;
    const int m = choose_axis(len, (int)tx, (int)ty, on_x);
    // exclusive prefix of the chosen axis' counts over the node's chunks
    unsigned carry = 0u;
    for (int i0 = 0; i0 < nch; i0 += 256) {
      const int i = i0 + tid;
      const unsigned v = i < nch ? (unsigned)load_agent((on_x ? a.ch_cx : a.ch_cy) + c0 + i) : 0u;
      unsigned inc = v;
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
      }
      if (lane == 63) red[0][wave] = inc;
      __syncthreads();
      unsigned before = carry, total = 0u;
      for (int w = 0; w < 4; ++w) {
        if (w < wave) before += red[0][w];
        total += red[0][w];
      }
      if (i < nch) a.ch_before[c0 + i] = (int)(before + inc - v);
      carry += total;
      __syncthreads();
    }
    // The counters this node draws from (node ids, the list of nodes above the subtrees, the next level's queue and chunks
    // per long child) are independent of one another: six threads ask at once, one round trip instead of six in a row.
    {
      const int d = a.ndepth[node];
      if (tid < 6) {
        const int side = tid & 1;
        const int cl = side ? len - m : m;
        const bool inner = cl > leaf_size && d + 1 < kBvhKeyDepth;  // make_children's rule
        int got = -1;
        if (tid == 0) got = alloc_nodes(a, 2);
        else if (tid == 1) got = atomicAdd(&a.flags[kBvhTopCount], 2);  // node ids: at most cap of them
        else if (tid < 4) {
          if (inner && cl > kSub) got = atomicAdd(&a.bigcount[level + 1], 1);
        } else if (inner && cl > kSub) {
          got = atomicAdd(&a.chunkcount[level + 1], (cl + kChunk - 1) / kChunk);  // its chunks for the next level's passes
        }
        drawn[tid] = got;
      }
    }
    __syncthreads();
    if (tid == 0) {
      kid[0] = kid[1] = -1;
      a.nsplit[node] = m | (on_x ? (int)0x80000000 : 0);
      bool leaf[2];
      const int first = drawn[0];
      if (first >= 0) {
        make_children(a, node, first, b, len, m, leaf_size, leaf);
        const int tslot = drawn[1];
        a.topq[tslot] = first;
        a.topq[tslot + 1] = first + 1;
        for (int side = 0; side < 2; ++side) {
          if (leaf[side]) continue;
          const int cl = side ? len - m : m;
          const int slot = drawn[2 + side];
          if (cl > kSub) {
            if (slot < a.cap_big) a.bigq[(size_t)((level + 1) & 1) * a.cap_big + slot] = first + side;
            else a.flags[kBvhFallback] = 1;
            const int cn = (cl + kChunk - 1) / kChunk;
            const int cc0 = drawn[4 + side];
            a.nchunk0[first + side] = cc0;
            if (cc0 + cn > a.cap_chunk) a.flags[kBvhFallback] = 1;
            else { kid[side] = first + side; kid_c0[side] = cc0; kid_n[side] = cn; kid_b[side] = side ? b + m : b; kid_len[side] = cl; }
          } else {
            a.subq[atomicAdd(&a.flags[kBvhSubCount], 1)] = first + side;  // at most one entry per node: cap entries
          }
        }
      } else {
        // No ids left: the build is declined (alloc_nodes raised the flag), but the queue slots and chunks drawn above are
        // read by the next level's passes all the same.  They get this node again: valid, and nobody looks at the result.
        for (int side = 0; side < 2; ++side) {
          const int cl = side ? len - m : m;
          if (drawn[2 + side] < 0) continue;
          if (drawn[2 + side] < a.cap_big) a.bigq[(size_t)((level + 1) & 1) * a.cap_big + drawn[2 + side]] = node;
          const int cn = (cl + kChunk - 1) / kChunk;
          if (drawn[4 + side] + cn <= a.cap_chunk) {
            kid[side] = node; kid_c0[side] = drawn[4 + side]; kid_n[side] = cn; kid_b[side] = b; kid_len[side] = len;
          }
        }
      }
    }
    __syncthreads();
    for (int side = 0; side < 2; ++side) {  // chunk -> node tables of the long children, by everybody
      if (kid[side] < 0) continue;
      int4* tr = a.ch_rec + (size_t)((level + 1) & 1) * a.cap_chunk + kid_c0[side];
      for (int i = tid; i < kid_n[side]; i += 256) tr[i] = make_int4(kid[side], i, kid_b[side], kid_len[side]);
    }
    __syncthreads();
  }
}

// ---- long nodes: the partition ---------------------------------------------------------------------------------------
// The crate's two pointers meet the k-th misplaced point going up from the left and the k-th going down from the right, so
// the swaps are known from ranks alone: a point's rank follows from the chunk prefixes of bvh_big_count.  Rounds 1-2 wrote
// two rank lists here and swapped in a kernel of its own; now each misplaced point posts its index in the slot of its pair
// (one atomicOr on a 64-bit word: low half the left point, high half the right one) and whichever of the two arrives second
// swaps the points and clears the slot.  The ranks are computed from the predicate BITS bvh_big_count left, not from the
// points: those are being swapped by other work-groups meanwhile.  One launch per level fewer, and no list is written or read.
__global__ __launch_bounds__(256) void bvh_big_ranks(BvhPtrs a, int level) {
  __shared__ unsigned wcnt[4];
  constexpr int PER = kChunk / 256;  // consecutive points per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = a.chunkcount[level];
  const int4* ch_rec = a.ch_rec + (size_t)(level & 1) * a.cap_chunk;
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
    const int4 cr = ch_rec[c];
    const int node = cr.x, ci = cr.y, b = cr.z, len = cr.w;
    const int sp = a.nsplit[node];
    const int shift = sp < 0 ? 0 : 1;  // on x: bit 0, on y: bit 1
    const int m = sp & 0x7fffffff;
    const int base = ci * kChunk + tid * PER;
    bool pr[PER];
    unsigned mine = 0u;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      pr[j] = false;
      if (base + j < len) {
        pr[j] = ((a.pred[b + base + j] >> shift) & 1) != 0;
        mine += pr[j];
      }
    }
    unsigned inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wcnt[wave] = inc;
    __syncthreads();
    unsigned t = (unsigned)a.ch_before[c] + inc - mine;  // predicate-true points of the node before my first one
    for (int w = 0; w < wave; ++w) t += wcnt[w];
    float2* P = a.P + b;
    uint32_t* ID = a.ID + b;
    unsigned long long* pair = a.pair + b;  // pair k of this node: k < min(m, len - m) <= len / 2, inside the node's own range
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = base + j;
      if (i < len) {
        const bool left = i < m && !pr[j];    // the k-th misplaced point from the left, k = i - t
        const bool right = i >= m && pr[j];   // the k-th misplaced point from the right, k = m - t - 1
        if (left || right) {
          const int k = left ? i - (int)t : m - (int)t - 1;
          const unsigned long long old = atomicOr(&pair[k], left ? (unsigned long long)(unsigned)(i + 1) : (unsigned long long)(unsigned)(i + 1) << 32);
          const unsigned other = left ? (unsigned)(old >> 32) : (unsigned)old;
          if (other != 0u) {  // the partner was here first: swap, and leave the slot clean for the next level
            const int l = left ? i : (int)other - 1, r = left ? (int)other - 1 : i;
            const float2 pl = P[l], pq = P[r];
            P[l] = pq; P[r] = pl;
            const uint32_t il = ID[l], ir = ID[r];
            ID[l] = ir; ID[r] = il;
            pair[k] = 0ull;
          }
        }
        t += pr[j];
      }
    }
    __syncthreads();
  }
}

// ---- long nodes, many points (a million and more): rank lists, then the swaps in a kernel of their own --------------------
// (the fused pass above spends an atomic with a return value and four scattered accesses per misplaced point inside ONE
// kernel: at N = 1 M 24.6 us per level against 6.7 + 6.4 us for the two below; at 151 405 it is 8.6 us against 9.4)
__global__ __launch_bounds__(256) void bvh_big_ranks_lists(BvhPtrs a, int level) {
  __shared__ unsigned wcnt[4];
  __shared__ unsigned red[4];
  constexpr int PER = kChunk / 256;  // consecutive points per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = a.chunkcount[level];
  const int4* ch_rec = a.ch_rec + (size_t)(level & 1) * a.cap_chunk;
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
    const int4 cr = ch_rec[c];
    const int node = cr.x, ci = cr.y, b = cr.z, len = cr.w;
    const float2 h = a.nmean[node];
    const int sp = a.nsplit[node];
    const bool on_x = sp < 0;
    const int m = sp & 0x7fffffff;
    const int base = ci * kChunk + tid * PER;
    bool pr[PER];
    unsigned mine = 0u;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      pr[j] = false;
      if (base + j < len) {
        const float2 q = a.P[b + base + j];
        pr[j] = on_x ? q.x > h.x : q.y > h.y;
        mine += pr[j];
      }
    }
    unsigned inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wcnt[wave] = inc;
    __syncthreads();
    unsigned t = (unsigned)a.ch_before[c] + inc - mine;  // predicate-true points of the node before my first one
    for (int w = 0; w < wave; ++w) t += wcnt[w];
    unsigned nbad = 0u;
    int* lidx = a.lidx + b;
    int* ridx = a.ridx + b;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = base + j;
      if (i < len) {
        if (i < m && !pr[j]) { lidx[i - (int)t] = i; ++nbad; }   // k-th misplaced from the left
        else if (i >= m && pr[j]) ridx[m - (int)t - 1] = i;      // k-th misplaced from the right
        t += pr[j];
      }
    }
    nbad = group_sum<4>(nbad, red, tid);
    if (tid == 0 && nbad) atomicAdd(&a.nbad[node], (int)nbad);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bvh_big_swap(BvhPtrs a, int level) {
  const int nc = a.chunkcount[level];
  const int4* ch_rec = a.ch_rec + (size_t)(level & 1) * a.cap_chunk;
  for (int c = blockIdx.x; c < nc; c += gridDim.x) {
    const int4 cr = ch_rec[c];
    const int node = cr.x, ci = cr.y, b = cr.z;
    const int nbad = a.nbad[node];
    float2* P = a.P + b;
    uint32_t* ID = a.ID + b;
    const int* lidx = a.lidx + b;
    const int* ridx = a.ridx + b;
    const int hi = (ci + 1) * kChunk < nbad ? (ci + 1) * kChunk : nbad;
    for (int k = ci * kChunk + (int)threadIdx.x; k < hi; k += 256) {
      const int l = lidx[k], r = ridx[k];
      const float2 pl = P[l], pr = P[r];
      P[l] = pr; P[r] = pl;
      const uint32_t il = ID[l], ir = ID[r];
      ID[l] = ir; ID[r] = il;
    }
  }
}

// ---- subtrees of at most kSub points, entirely in LDS -----------------------------------------------------------------
struct SubLds {
  float2 P[kSub];
  uint16_t I[kSub];  // which of the subtree's points (as loaded) sits here now
  uint32_t W[kSub];  // weights of the points as loaded
  uint16_t L[kSub], R[kSub];
  uint16_t lb[2][kSub / 2], ll[2][kSub / 2];  // node lists of two consecutive levels: begin, length (subtree-relative)
  int lg[2][kSub / 2];                        // ... and breadth-first id
  int lcount[2];
  int idbase;  // first id of the children made at this level (one allocation per level)
  int depth_max;  // deepest node made by this work-group
  int id_next, id_end;  // ids this work-group holds (one atomic on the global counter per range, not per level)
  int loff[kBvhKeyDepth + 2];  // where each level of the subtree starts in its list of internal nodes
  // nodes_by_groups: per wave partial results, per node sums
  float4 gbox[kSubWaves];
  float2 gsum[kSubWaves];
  unsigned gcx[kSubWaves], gcy[kSubWaves], gtot[kSubWaves], gbad[kSubWaves];
};

// One node whose points are s.P[lb, lb+len): fold, axis, partition, children — by ONE wave.  Up to kSub points the
// chain is simply added in order (a scan would restart at every doubling of the sum and lose).
__device__ __forceinline__ void node_in_lds(const BvhPtrs& a, SubLds& s, int gbegin, int lb, int len, int gid, int first, int lane,
                                            int nxt, int leaf_size, int depth) {
  constexpr int TILE = 64 * kEPT;
  float2* P = s.P + lb;
  uint16_t* I = s.I + lb;
  uint16_t* L = s.L + lb;
  uint16_t* R = s.R + lb;
  // the chain: every lane reads the same LDS word (a broadcast) and adds it — 8 clocks per add, measured against 14 for
  // the DPP broadcast and 10 for v_readlane (tools/chain_microbench.hip)
  float sx = 0.f, sy = 0.f;
  chain_lds_xy(P, len, sx, sy);
  const float hx = sx / (float)len, hy = sy / (float)len;  // :67
  Box box;
  unsigned cx = 0u, cy = 0u;
  for (int i = lane; i < len; i += 64) {
    const float2 q = P[i];
    box.add(q);
    cx += q.x > hx;
    cy += q.y > hy;
  }
  box.reduce_wave();
  cx = group_sum<1>(cx, nullptr, lane);
  cy = group_sum<1>(cy, nullptr, lane);
  bool on_x;
  const int m = choose_axis(len, (int)cx, (int)cy, on_x);
  unsigned carry = 0u, nbad = 0u;
  for (int p0 = 0; p0 < len; p0 += TILE) {
    const int base = p0 + lane * kEPT;
    bool pr[kEPT];
    unsigned mine = 0u;
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      pr[j] = false;
      if (base + j < len) {
        const float2 q = P[base + j];
        pr[j] = on_x ? q.x > hx : q.y > hy;
        mine += pr[j];
      }
    }
    unsigned inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
      if (lane >= d) inc += o;
    }
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
    unsigned t = carry + inc - mine;
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      const int i = base + j;
      if (i < len) {
        if (i < m && !pr[j]) { L[i - (int)t] = (uint16_t)i; ++nbad; }
        else if (i >= m && pr[j]) R[m - (int)t - 1] = (uint16_t)i;
        t += pr[j];
      }
    }
    carry += total;
  }
  nbad = group_sum<1>(nbad, nullptr, lane);
  group_sync<1>();
  for (int k = lane; k < (int)nbad; k += 64) {
    const int l = L[k], r = R[k];
    const float2 pl = P[l], pr = P[r];
    P[l] = pr; P[r] = pl;
    const uint16_t il = I[l], ir = I[r];
    I[l] = ir; I[r] = il;
  }
  if (lane == 0) {
    a.nbox[gid] = make_float4(box.mnx, box.mny, box.mxx, box.mxy);
    bool leaf[2];
    make_children(a, gid, first, gbegin + lb, len, m, leaf_size, leaf, &s.depth_max, depth);
    for (int side = 0; side < 2; ++side) {
      if (leaf[side]) continue;
      const int slot = atomicAdd(&s.lcount[nxt], 1);
      s.lb[nxt][slot] = (uint16_t)(side ? lb + m : lb);
      s.ll[nxt][slot] = (uint16_t)(side ? len - m : m);
      s.lg[nxt][slot] = first + side;
    }
  }
  group_sync<1>();
}

// Centre and mass of an internal node from its two children (bvh_tree.rs:133-158)
__device__ __forceinline__ void combine_children(const BvhPtrs& a, int node) {
  const int c0 = a.nchild[node];
  if (c0 < 0) {  // children not made (yet: a blind pass on an unfinished tree; or buffers full, the fallback flag is up)
    a.nsize[node] = 1;
    return;
  }
  const float2 g0 = a.ncog[c0], g1 = a.ncog[c0 + 1];
  const uint32_t m0 = a.nmass[c0], m1 = a.nmass[c0 + 1];
  const uint32_t ms = m0 + m1;                                      // :148
  const float bx = (g0.x * (float)m0) + (g1.x * (float)m1);         // :150-153
  const float by = (g0.y * (float)m0) + (g1.y * (float)m1);
  a.ncog[node] = make_float2(bx / (float)ms, by / (float)ms);       // :154
  a.nmass[node] = ms;
  a.nsize[node] = 1 + a.nsize[c0] + a.nsize[c0 + 1];
}

// make_leaf (:40-54) + leaf mass and centre (:98-131) by one wave.  P: the leaf's points; wid(i): row of point i.
template <class PosPtr, class RowOf>
__device__ __forceinline__ void leaf_by_wave(const BvhPtrs& a, int leaf, PosPtr P, int len, int lane, const uint32_t* weight,
                                             RowOf row_of) {
  float mnx = kMaxF, mny = kMaxF, mxx = 0.f, mxy = 0.f, sx = 0.f, sy = 0.f;
  uint32_t ms = 0u;
  for (int k0 = 0; k0 < len; k0 += 64) {
    const int cnt = len - k0 < 64 ? len - k0 : 64;
    float2 q = make_float2(0.f, 0.f);
    if (lane < cnt) {
      q = P[k0 + lane];
      mnx = sse_min(mnx, q.x); mny = sse_min(mny, q.y);
      mxx = sse_max(mxx, q.x); mxy = sse_max(mxy, q.y);
      ms += weight[row_of(k0 + lane)];  // u32, wraps like the release build; any order
    }
    for (int k = 0; k < cnt; ++k) {  // slice order
      sx = sx + lane_value(q.x, k);
      sy = sy + lane_value(q.y, k);
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    mnx = sse_min(mnx, __shfl_xor(mnx, d, 64)); mny = sse_min(mny, __shfl_xor(mny, d, 64));
    mxx = sse_max(mxx, __shfl_xor(mxx, d, 64)); mxy = sse_max(mxy, __shfl_xor(mxy, d, 64));
    ms += (uint32_t)__shfl_xor((int)ms, d, 64);
  }
  if (lane == 0) {
    a.nbox[leaf] = make_float4(mnx, mny, mxx, mxy);
    a.ncog[leaf] = make_float2(sx / (float)len, sy / (float)len);  // NaN for an empty leaf, as upstream
    a.nmass[leaf] = ms;
    a.nsize[leaf] = 1;
  }
}

// The first levels of a subtree have fewer nodes than the work-group has waves: G waves share a node (nc * G <= kSubWaves).
// What node_in_lds does, spread out: the two chains (one add after the other) get a wave each; box, counts, rank lists and
// swaps are split G ways.  Every wave runs the same number of barriers
// (the longest node of the level sets the number of rank-list rounds).
template <int G>
__device__ __forceinline__ void nodes_by_groups(const BvhPtrs& a, SubLds& s, int gbegin, int cur, int nc, int idbase, int tid,
                                                int leaf_size, int depth) {
  static_assert(G >= 2 && kSubWaves % G == 0, "at least one wave beside the chain's");
  constexpr int TILE = 64 * kEPT;
  const int lane = tid & 63, wave = tid >> 6;
  const int grp = wave / G, gw = wave % G, w0 = grp * G;
  const bool act = grp < nc;
  const int lb = act ? s.lb[cur][grp] : 0, len = act ? s.ll[cur][grp] : 0, gid = act ? s.lg[cur][grp] : 0;
  int maxlen = 0;
  for (int e = 0; e < nc; ++e) maxlen = maxlen > (int)s.ll[cur][e] ? maxlen : (int)s.ll[cur][e];
  float2* P = s.P + lb;
  uint16_t* I = s.I + lb;
  uint16_t* L = s.L + lb;
  uint16_t* R = s.R + lb;
  // the two chains, a wave each (one dependent add per point instead of two)
  if (gw < 2) {
    const float sc = chain_lds<2>(reinterpret_cast<const float*>(P) + gw, len, 0.f);
    if (lane == 0) (gw ? s.gsum[grp].y : s.gsum[grp].x) = sc;
  }
  __syncthreads();
  const float2 sum = s.gsum[grp];
  const float hx = sum.x / (float)len, hy = sum.y / (float)len;  // :67
  Box box;
  unsigned cx = 0u, cy = 0u;
  for (int i = gw * 64 + lane; i < len; i += G * 64) {
    const float2 q = P[i];
    box.add(q);
    cx += q.x > hx;
    cy += q.y > hy;
  }
  box.reduce_wave();
  cx = group_sum<1>(cx, nullptr, lane);
  cy = group_sum<1>(cy, nullptr, lane);
  if (lane == 0) {
    s.gbox[wave] = make_float4(box.mnx, box.mny, box.mxx, box.mxy);
    s.gcx[wave] = cx;
    s.gcy[wave] = cy;
  }
  __syncthreads();
  cx = 0u; cy = 0u;
  box = Box();
  for (int w = 0; w < G; ++w) {
    const float4 bx = s.gbox[w0 + w];
    box.mnx = sse_min(box.mnx, bx.x); box.mny = sse_min(box.mny, bx.y);
    box.mxx = sse_max(box.mxx, bx.z); box.mxy = sse_max(box.mxy, bx.w);
    cx += s.gcx[w0 + w];
    cy += s.gcy[w0 + w];
  }
  bool on_x;
  const int m = choose_axis(len, (int)cx, (int)cy, on_x);
  unsigned carry = 0u, nbad = 0u;
  for (int r0 = 0; r0 < maxlen; r0 += G * TILE) {
    const int base = r0 + gw * TILE + lane * kEPT;
    bool pr[kEPT];
    unsigned mine = 0u;
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      pr[j] = false;
      if (base + j < len) {
        const float2 q = P[base + j];
        pr[j] = on_x ? q.x > hx : q.y > hy;
        mine += pr[j];
      }
    }
    unsigned inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = (unsigned)__shfl_up((int)inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) s.gtot[wave] = inc;
    __syncthreads();
    unsigned before = carry, total = 0u;
    for (int w = 0; w < G; ++w) {
      const unsigned v = s.gtot[w0 + w];
      if (w < gw) before += v;
      total += v;
    }
    unsigned t = before + inc - mine;
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      const int i = base + j;
      if (i < len) {
        if (i < m && !pr[j]) { L[i - (int)t] = (uint16_t)i; ++nbad; }
        else if (i >= m && pr[j]) R[m - (int)t - 1] = (uint16_t)i;
        t += pr[j];
      }
    }
    carry += total;
    __syncthreads();  // gtot is rewritten by the next round; after the last one: the rank lists are complete
  }
  nbad = group_sum<1>(nbad, nullptr, lane);
  if (lane == 0) s.gbad[wave] = nbad;
  __syncthreads();
  nbad = 0u;
  for (int w = 0; w < G; ++w) nbad += s.gbad[w0 + w];
  for (int k = gw * 64 + lane; k < (int)nbad; k += G * 64) {
    const int l = L[k], r = R[k];
    const float2 pl = P[l], pr = P[r];
    P[l] = pr; P[r] = pl;
    const uint16_t il = I[l], ir = I[r];
    I[l] = ir; I[r] = il;
  }
  if (act && gw == 0 && lane == 0) {
    a.nbox[gid] = make_float4(box.mnx, box.mny, box.mxx, box.mxy);
    bool leaf[2];
    const int first = idbase + 2 * grp;
    make_children(a, gid, first, gbegin + lb, len, m, leaf_size, leaf, &s.depth_max, depth);
    const int nxt = cur ^ 1;
    for (int side = 0; side < 2; ++side) {
      if (leaf[side]) continue;
      const int slot = atomicAdd(&s.lcount[nxt], 1);
      s.lb[nxt][slot] = (uint16_t)(side ? lb + m : lb);
      s.ll[nxt][slot] = (uint16_t)(side ? len - m : m);
      s.lg[nxt][slot] = first + side;
    }
  }
}

__global__ __launch_bounds__(kSubWaves * 64) void bvh_subtrees(BvhPtrs a, const uint32_t* __restrict__ weight, int leaf_size,
                                                              int sub_start) {
  __shared__ SubLds s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nsub = a.flags[kBvhSubCount];
#ifdef NB_BVH_TIMING
#define NB_STAMP(k) { const long long t_ = wall_clock64(); if (tid == 0) atomicMax(&a.flags[kBvhDebug + k], (int)(t_ - t_prev)); t_prev = t_; }
#else
#define NB_STAMP(k) {}
#endif
  for (int si = sub_start + blockIdx.x; si < nsub; si += gridDim.x) {
#ifdef NB_BVH_TIMING
    long long t_prev = wall_clock64();
#endif
    const int root = a.subq[si];
    const int b = a.nbegin[root], len = a.nlen[root], root_depth = a.ndepth[root];
    int* list = a.lidx + b;  // the subtree's internal nodes, level after level (the global rank lists are free now)
    for (int i = tid; i < len; i += kSubWaves * 64) {
      s.P[i] = a.P[b + i];
      s.I[i] = (uint16_t)i;
      s.W[i] = weight[a.ID[b + i]];
    }
    if (tid == 0) {
      s.depth_max = 0;
      s.id_next = s.id_end = 0;
      s.lb[0][0] = 0;
      s.ll[0][0] = (uint16_t)len;  // kSub = 4096 fits
      s.lg[0][0] = root;
      s.lcount[0] = 1;
      s.lcount[1] = 0;
    }
    __syncthreads();
    NB_STAMP(0)
    int cur = 0, nlev = 0, nint = 0;
    for (;;) {
      const int nc = s.lcount[cur];
      if (nc == 0) break;
      if (nint + nc > len || nlev > kBvhKeyDepth) {  // only degenerate input gets here
        if (tid == 0) a.flags[kBvhFallback] = 1;
        break;
      }
      for (int e = tid; e < nc; e += kSubWaves * 64) list[nint + e] = s.lg[cur][e];
      if (tid == 0) {
        s.loff[nlev] = nint;
        // ids for this level's children out of the work-group's range; a new range when it runs out.  726 work-groups (1 M
        // points) asking the one global counter at every level wait ~60 ns per request ahead of them.
        const int want = 2 * nc;
        if (s.id_end - s.id_next < want) {
          // the first range is what ANY tree over these points needs (leaves hold at most leaf_size points: at least
          // len / leaf_size of them, twice as many nodes less the root's own id) and is used up to the last id; later ones
          // (a quarter of that) may leave a few ids unused.  If the ids are nearly out, the exact count is asked for.
          const int sure = 2 * (len / leaf_size) - 2;
          int range = s.id_end == 0 ? sure : sure / 4;
          range = range > want ? range : want;
          int first = range > want ? try_alloc_nodes(a, range) : -1;
          if (first < 0) {
            range = want;
            first = alloc_nodes(a, want);
          }
          s.id_next = first;
          s.id_end = first < 0 ? first : first + range;
        }
        s.idbase = s.id_next;
        if (s.id_next >= 0) s.id_next += want;
      }
      __syncthreads();
      const int idbase = s.idbase;
      if (idbase < 0) break;
      nint += nc;
      ++nlev;
      const int depth = root_depth + nlev - 1;  // of this level's nodes (nlev counts this level already)
      if (nc == 1) nodes_by_groups<kSubWaves>(a, s, b, cur, nc, idbase, tid, leaf_size, depth);
      else if (nc == 2) nodes_by_groups<kSubWaves / 2>(a, s, b, cur, nc, idbase, tid, leaf_size, depth);
      else if (nc <= 4) nodes_by_groups<kSubWaves / 4>(a, s, b, cur, nc, idbase, tid, leaf_size, depth);
      else
        for (int e = wave; e < nc; e += kSubWaves)  // a wave per node
          node_in_lds(a, s, b, s.lb[cur][e], s.ll[cur][e], s.lg[cur][e], idbase + 2 * e, lane, cur ^ 1, leaf_size, depth);
      __syncthreads();
      if (tid == 0) s.lcount[cur] = 0;
      cur ^= 1;
      __syncthreads();
      if (nlev == 1) NB_STAMP(1)
      if (nlev == 2) NB_STAMP(2)
      if (nlev == 3) NB_STAMP(3)
    }
    if (tid == 0) {
      s.loff[nlev] = nint;
      if (s.depth_max > 0) atomicMax(&a.flags[kBvhMaxDepth], s.depth_max);
    }
    __syncthreads();
    NB_STAMP(4)
    // leaves: children of the listed nodes that were not split further; a lane each, points still in LDS
    // (make_leaf :40-54, leaf mass and centre :98-131)
    for (int task = tid; task < 2 * nint; task += kSubWaves * 64) {
      const int c0 = a.nchild[list[task >> 1]];
      if (c0 < 0) continue;
      const int c = c0 + (task & 1);
      if (!a.nleaf[c]) continue;
      const int lb = a.nbegin[c] - b, ln = a.nlen[c];
      Box box;
      float sx = 0.f, sy = 0.f;
      uint32_t ms = 0u;
      for (int k = 0; k < ln; ++k) {  // slice order
        const float2 q = s.P[lb + k];
        box.add(q);
        sx = sx + q.x;
        sy = sy + q.y;
        ms += s.W[s.I[lb + k]];  // u32, wraps like the release build
      }
      a.nbox[c] = make_float4(box.mnx, box.mny, box.mxx, box.mxy);
      a.ncog[c] = make_float2(sx / (float)ln, sy / (float)ln);  // NaN for an empty leaf, as upstream
      a.nmass[c] = ms;
      a.nsize[c] = 1;
    }
    __syncthreads();
    NB_STAMP(5)
    // upward pass inside the subtree: deepest level first
    for (int lev = nlev - 1; lev >= 0; --lev) {
      for (int e = s.loff[lev] + tid; e < s.loff[lev + 1]; e += kSubWaves * 64) combine_children(a, list[e]);
      __syncthreads();
    }
    NB_STAMP(6)
    // the subtree's rows go back in their final order
    uint32_t ids[kSub / (kSubWaves * 64)];
#pragma unroll
    for (int j = 0; j < kSub / (kSubWaves * 64); ++j) {
      const int i = tid + j * kSubWaves * 64;
      ids[j] = i < len ? a.ID[b + s.I[i]] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSub / (kSubWaves * 64); ++j) {
      const int i = tid + j * kSubWaves * 64;
      if (i < len) {
        a.ID[b + i] = ids[j];
        a.P[b + i] = s.P[i];
      }
    }
    __syncthreads();
    NB_STAMP(7)
  }
}

// The nodes above the subtrees (topq): leaves hanging directly off a long node, then the long nodes themselves,
// deepest level first.  One work-group: a few hundred nodes (a few thousand at N = 4M).
__global__ __launch_bounds__(256) void bvh_top_upward(BvhPtrs a, const uint32_t* __restrict__ weight) {
  constexpr int kKeep = 4096;       // entries whose (id, depth) stay in LDS; more than that are read again every level
  __shared__ int s_id[kKeep];
  __shared__ int16_t s_depth[kKeep];  // depth of a long internal node, -1: not one (a leaf, a subtree root)
  constexpr int kLeafList = 1024;
  __shared__ int s_dmax, s_nleaf;
  __shared__ int s_leaf[kLeafList];
  __shared__ int s_first[kBvhLevels + 2];  // where the entries of each depth begin: the levels append theirs one after the
                                           // other, so topq is ordered by depth and a level's loop visits its own range only
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_top = a.flags[kBvhTopCount];
  if (tid == 0) { a.flags[kBvhBadIndex] = 0; s_dmax = -1; s_nleaf = 0; }
  for (int d = tid; d < kBvhLevels + 2; d += 256) s_first[d] = n_top;
  __syncthreads();
  // what each entry is, once (all threads side by side): the loops below visit every entry, the level loops at every depth
  const auto describe = [&](int e, int& id) {  // depth of a long internal node, -2 a leaf, -1 a subtree's root
    id = a.topq[e];
    if (a.nleaf[id]) return -2;
    return a.nlen[id] > kSub ? a.ndepth[id] : -1;
  };
  int dmax = -1;
  for (int e = tid; e < n_top; e += 256) {
    int id;
    const int d = describe(e, id);
    if (e < kKeep) { s_id[e] = id; s_depth[e] = (int16_t)d; }
    dmax = d > dmax ? d : dmax;
    int td = a.ndepth[id];
    td = td < 0 ? 0 : (td > kBvhLevels ? kBvhLevels : td);
    atomicMin(&s_first[td], e);
    if (d == -2) {  // a leaf hanging directly off a long node (few): listed, so that the waves below do not search for them
      const int slot = atomicAdd(&s_nleaf, 1);
      if (slot < kLeafList) s_leaf[slot] = e;
    }
  }
  if (dmax >= 0) atomicMax(&s_dmax, dmax);
  __syncthreads();
  if (tid == 0)  // a depth without entries begins where the next one does
    for (int d = kBvhLevels; d >= 0; --d) s_first[d] = s_first[d] < s_first[d + 1] ? s_first[d] : s_first[d + 1];
  __syncthreads();
  const bool listed = s_nleaf <= kLeafList;
  for (int k = wave; k < (listed ? s_nleaf : n_top); k += 4) {  // leaves hanging directly off a long node: a wave each
    const int e = listed ? s_leaf[k] : k;
    int id;
    const int kind = e < kKeep ? (id = s_id[e], (int)s_depth[e]) : describe(e, id);
    if (kind != -2) continue;
    const int b = a.nbegin[id];
    const uint32_t* ids = a.ID + b;
    leaf_by_wave(a, id, (const float2*)(a.P + b), a.nlen[id], lane, weight, [&](int i) { return ids[i]; });
  }
  __syncthreads();
  dmax = s_dmax;  // deepest long node: the tree below it belongs to the subtrees
  const auto entry = [&](int e, int& id) {
    if (e < kKeep) { id = s_id[e]; return (int)s_depth[e]; }
    return describe(e, id);
  };
  for (int d = dmax; d >= 0; --d) {
    for (int e = s_first[d] + tid; e < s_first[d + 1]; e += 256) {
      int id;
      if (entry(e, id) == d) combine_children(a, id);
    }
    __syncthreads();
  }
  if (tid == 0) a.flags[kBvhNodes] = a.nsize[0];  // the root's subtree: every node of the tree
  // pre-order numbers, top down: a node, its left subtree, its right subtree (the root has number 0); the nodes inside the
  // subtrees find theirs from here in bvh_emit
  for (int d = 0; d <= dmax; ++d) {
    for (int e = s_first[d] + tid; e < s_first[d + 1]; e += 256) {
      int id;
      if (entry(e, id) == d) {
        const int c0 = a.nchild[id];
        if (c0 >= 0) {
          a.npre[c0] = a.npre[id] + 1;
          a.npre[c0 + 1] = a.npre[id] + 1 + a.nsize[c0];
        }
      }
    }
    __syncthreads();
  }
}

// ---- numbering -----------------------------------------------------------------------------------------------------
// Final arrays in pre-order.  A node's number: walk up to the nearest ancestor whose number is known (one of the nodes above
// the subtrees), adding 1 per step and the left sibling's subtree where the step comes from a right child.
__device__ __forceinline__ void emit_node(const BvhPtrs& a, const int i, float4* __restrict__ geom0, float4* __restrict__ geom1,
                                          int4* __restrict__ link, int* __restrict__ depth_out, uint32_t* __restrict__ mass_out,
                                          float2* __restrict__ size_out) {
  const int ids = a.flags[kBvhNodeCount] < a.cap ? a.flags[kBvhNodeCount] : a.cap;  // ids handed out: not all are nodes
  const int m = a.flags[kBvhNodes];
  if (i >= ids || a.ndepth[i] < 0) return;
  int idx = 0, v = i;
  for (int guard = 0; a.npre[v] < 0 && guard <= kBvhLevels; ++guard) {
    const int p = a.nparent[v];
    const int c0 = a.nchild[p];
    idx += 1 + (v == c0 ? 0 : a.nsize[c0]);
    v = p;
  }
  // (this kernel also runs, blind, on trees whose long nodes are not all split yet: never write through a bad index)
  if (a.npre[v] < 0) {
    a.flags[kBvhBadIndex] = 1;
    return;
  }
  idx += a.npre[v];
  if (idx < 0 || idx >= m) {
    a.flags[kBvhBadIndex] = 1;
    return;
  }
  const int d = a.ndepth[i];
  const float4 box = a.nbox[i];
  const float w = box.z - box.x, h = box.w - box.y;  // boundary.size = max - min (:63-66)
  const float2 cog = a.ncog[i];
  const uint32_t ms = a.nmass[i];
  const float tx = sse_max(w, h), ty = sse_max(h, w);  // size.max(size.yx()), main.rs:371
  geom0[idx] = make_float4(box.x, box.y, box.x + w, box.y + h);
  geom1[idx] = make_float4(cog.x, cog.y, (float)ms, tx * ty);
  link[idx] = make_int4(idx + a.nsize[i], a.nbegin[i], a.nlen[i], a.nleaf[i]);  // skip: the first node after the subtree
  depth_out[idx] = d;
  mass_out[idx] = ms;
  size_out[idx] = make_float2(w, h);
}
__global__ __launch_bounds__(256) void bvh_emit(BvhPtrs a, float4* __restrict__ geom0, float4* __restrict__ geom1, int4* __restrict__ link,
                                                int* __restrict__ depth_out, uint32_t* __restrict__ mass_out,
                                                float2* __restrict__ size_out) {
  emit_node(a, (int)(blockIdx.x * 256 + threadIdx.x), geom0, geom1, link, depth_out, mass_out, size_out);
}
// bvh_emit and the row gather that follows it in every step (gather_particles) as ONE launch: two index spaces that do not depend
// on each other (the gather reads the permutation the partition left, not the numbering).  The numbering's work-groups come first:
// theirs is the longer chain (a walk up to the nearest numbered ancestor).  One launch of ~5 us less per step.
__global__ __launch_bounds__(256) void bvh_emit_gather(BvhPtrs a, float4* __restrict__ geom0, float4* __restrict__ geom1, int4* __restrict__ link,
                                                       int* __restrict__ depth_out, uint32_t* __restrict__ mass_out,
                                                       float2* __restrict__ size_out, const unsigned emit_groups, const GatherArgs<float> g) {
  if (blockIdx.x < emit_groups) emit_node(a, (int)(blockIdx.x * 256 + threadIdx.x), geom0, geom1, link, depth_out, mass_out, size_out);
  else gather_row<float>(g, (int64_t)(blockIdx.x - emit_groups) * 256 + threadIdx.x);
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

BvhBuildLayout bvh_build_layout(int64_t n, int leaf_size) {
  BvhBuildLayout L{};
  const size_t N = (size_t)(n > 0 ? n : 1);
  // leaves end up between half full and full: ~2.7 N / leaf_size nodes.  More than this -> host builder
  const size_t lf = (size_t)(leaf_size > 0 ? leaf_size : 1);
  const size_t C = (4 * N / lf < 2 * N ? 4 * N / lf : 2 * N) + 4096;
  const size_t CB = N / (size_t)(kSub + 1) + 2;      // long nodes of one level
  const size_t CC = N / (size_t)kChunk + CB + 2;     // their chunks
  L.node_cap = (int)C;
  L.big_cap = (int)CB;
  L.chunk_cap = (int)CC;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  L.flags = take(sizeof(int) * kBvhFlagWords);
  L.bigcount = take(sizeof(int) * kBvhLevels);
  L.chunkcount = take(sizeof(int) * kBvhLevels);
  L.zero_end = off;
  L.bigq = take(4 * 2 * CB);
  L.subq = take(4 * C);
  L.topq = take(4 * C);
  L.ch_rec = take(16 * 2 * CC);
  L.ch_sum = take(16 * CC);
  L.ch_box = take(16 * CC);
  L.ch_run = take(4 * 48 * CC);  // 2 * kRunRec ints per chunk
  L.ch_cx = take(4 * CC);
  L.ch_cy = take(4 * CC);
  L.ch_before = take(4 * CC);
  L.pts = take(sizeof(float2) * N);
  L.ids = take(4 * N);
  L.lidx = take(4 * N);
  L.ridx = take(4 * N);
  L.pred = take(N);
  L.pair = take(8 * N);
  L.nbegin = take(4 * C);
  L.nlen = take(4 * C);
  L.nparent = take(4 * C);
  L.nchild = take(4 * C);
  L.ndepth = take(4 * C);
  L.nleaf = take(4 * C);
  L.nsize = take(4 * C);
  L.npre = take(4 * C);
  L.nbox = take(16 * C);
  L.ncog = take(8 * C);
  L.nmass = take(4 * C);
  L.narrive = take(4 * C);
  L.nmean = take(8 * C);
  L.nsplit = take(4 * C);
  L.nchunk0 = take(4 * C);
  L.ndone = take(4 * C);
  L.nbad = take(4 * C);
  L.total = off;
  return L;
}

int bvh_build_first_levels(int64_t n) {
  int lv = 0;
  for (int64_t k = kSub; k < n; k *= 2) ++lv;  // a balanced tree's long levels
  return lv;
}

hipError_t bvh_build_begin(hipStream_t s, const void* pos, int n, char* scratch, const BvhBuildLayout& L, bool flags_clean,
                           unsigned long long* stamp_begin, unsigned long long* stamp_prev_end) {
  hipError_t e = flags_clean ? hipSuccess : hipMemsetAsync(scratch + L.flags, 0, L.zero_end - L.flags, s);  // flags + level counters
  if (e != hipSuccess) return e;
  BvhPtrs a = make_ptrs(scratch, L);
  bvh_init<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a, (const float2*)pos, n, stamp_begin, stamp_prev_end);
  return hipGetLastError();
}

#ifdef NB_FOLD_TIMING
static void bvh_debug_fold_times(hipStream_t s);
#endif
hipError_t bvh_build_levels(hipStream_t s, int n, int leaf_size, int level_begin, int level_end, char* scratch,
                            const BvhBuildLayout& L) {
  BvhPtrs a = make_ptrs(scratch, L);
  for (int level = level_begin; level < level_end && level < kBvhLevels - 1; ++level) {
    const int64_t width = level < 30 ? (int64_t)1 << level : (int64_t)1 << 30;  // a level never has more nodes than this
    int64_t gb = L.big_cap < width ? L.big_cap : width;
    int64_t gc = L.chunk_cap;
    if (gc > 1024) gc = 1024;
    // could a node of this level be longer than kRunLen?  (A performance choice only: a chain without prepared runs is
    // scanned.  Three quarters of kRunLen as the level's average: the level whose nodes average 9 462 of the reference scene's
    // points spent two launches preparing runs nobody used.)
    const int use_runs = (n >> level) > kRunLen * 3 / 4 ? 1 : 0;
    if (use_runs) {
      bvh_chunk_sums<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
      bvh_chunk_runs<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
    }
    bvh_big_fold<<<dim3((unsigned)gb, 2), dim3(512), 0, s>>>(a, level, use_runs);
#ifdef NB_FOLD_TIMING
    std::fprintf(stderr, "level %d: ", level);
    bvh_debug_fold_times(s);
#endif
    bvh_big_count<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level, leaf_size);
    if (n <= kFusedPartitionMax) {
      bvh_big_ranks<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
    } else {
      bvh_big_ranks_lists<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
      bvh_big_swap<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
    }
  }
  return hipGetLastError();
}

hipError_t bvh_build_finish(hipStream_t s, const uint32_t* weight, int n, int leaf_size, int sub_start, char* scratch,
                            const BvhBuildLayout& L,
                            uint32_t* order_out, void* geom0, void* geom1, void* link, int* depth_out, uint32_t* mass_out,
                            float2* size_out, const GatherArgs<float>* gather) {
  BvhPtrs a = make_ptrs(scratch, L);
  const int C = L.node_cap;
  int64_t gs = (int64_t)n / 64 + 1;  // subtree roots
  if (gs > 2048) gs = 2048;
  bvh_subtrees<<<dim3((unsigned)gs), dim3(kSubWaves * 64), 0, s>>>(a, weight, leaf_size, sub_start);
  bvh_top_upward<<<dim3(1), dim3(256), 0, s>>>(a, weight);
  const dim3 gm((unsigned)((C + 255) / 256));
  if (gather && gather->n > 0) {  // the rows into tree order in the same launch (perm: where the build left it, bvh_build_order)
    const unsigned gg = (unsigned)((gather->n + 255) / 256);
    bvh_emit_gather<<<dim3(gm.x + gg), dim3(256), 0, s>>>(a, (float4*)geom0, (float4*)geom1, (int4*)link, depth_out, mass_out, size_out, gm.x,
                                                          *gather);
  } else {
    bvh_emit<<<gm, dim3(256), 0, s>>>(a, (float4*)geom0, (float4*)geom1, (int4*)link, depth_out, mass_out, size_out);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (!order_out) return hipSuccess;
  return hipMemcpyAsync(order_out, a.ID, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, s);
}
const uint32_t* bvh_build_order(const char* scratch, const BvhBuildLayout& L) { return (const uint32_t*)(scratch + L.ids); }


#ifdef NB_FOLD_TIMING
static void bvh_debug_fold_times(hipStream_t s) {
  unsigned long long h[16] = {};
  (void)hipStreamSynchronize(s);
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fold_t), sizeof(h));
  std::fprintf(stderr, "[fold timing, 10 ns ticks] seq %llu in %llu runs; scan %llu in %llu rounds; nodes %llu ticks over %llu chains, %llu points; %llu slow rounds\n",
               h[0], h[2], h[1], h[3], h[4], h[5], h[6], h[7]);
  if (h[8] | h[9] | h[10] | h[11])
    std::fprintf(stderr, "    run path: merge %llu, gap fetch %llu (%llu addends), run walks %llu, scanned chunks %llu in %llu\n", h[8], h[9], h[13], h[10], h[11], h[12]);
  unsigned long long z[16] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fold_t), z, sizeof(z));
}
#endif

}  // namespace nbody

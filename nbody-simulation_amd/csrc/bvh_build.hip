// Device-side build of the linearised BVH (gfx950, f32) — bit-identical to the host builder (tree_build.hpp) and to
// BVHTree::from + calculate_gravity of /root/reference src/bvh_tree.rs:40-158.
//
// What has to be reproduced, per node (bvh_tree.rs:56-96): ONE sequential fold over the slice (min from f32::MAX, max
// from 0.0, sum in slice order), mean = sum / len, the "better balanced axis" rule on #{x > mean.x} / #{y > mean.y},
// and the two-pointer partition of crate `partition` 0.1.2 (predicate-true side first).  The order of the particles
// inside each side feeds the next level's sum, so the permutation has to be exact, not just the split.
//
//   * min / max / counts are order-independent: plain block reductions.
//   * the sum is a rounding chain: exact_sum.h scans it (addend = map parity -> increment inside one binade; a real
//     f32 add whenever the chain leaves its binade).
//   * the two-pointer partition swaps the k-th misplaced element from the left with the k-th misplaced element from
//     the right (the pointers only ever stop at misplaced elements): two rank lists from one prefix count, then
//     independent swaps.
//
// Breadth-first, one work-group per node and level; three group sizes (1024 / 256 / 64 threads) by node length, each
// with its own queue.  Nodes get breadth-first ids; the pre-order index the walk needs is the rank of the node's
// root-to-node path (left-aligned, depth as tie-break) after one radix sort; `skip` is a binary search for the end of
// the path's sub-range.  Leaves (unweighted mean in slice order, u32 wrapping mass, :98-131) and the upward pass
// (:133-158) run bottom-up with one arrival counter per internal node.
//
// Anything this cannot express (NaN positions — pathfinder's minps/maxps are order-dependent there —, a node deeper than
// the 56-bit path, more nodes than the buffers hold) raises a flag and the caller uses the host builder instead.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <limits.h>
#include <stdint.h>

#include "bvh_build.h"
#include "exact_sum.h"

namespace nbody {

namespace {

constexpr int kEPT = 4;      // consecutive addends per thread and scan
constexpr int kSeqRun = 64;  // real adds after the scan stops
constexpr float kMaxF = 3.402823466e+38f;

struct BvhPtrs {
  int* flags;
  int* qcount;  // [class][level]
  int* queue;   // [class][level & 1][cap]
  float2* P;
  uint32_t* ID;
  int* lidx;
  int* ridx;
  int* nbegin;
  int* nlen;
  int* nparent;
  int* nchild;
  int* ndepth;
  int* nleaf;
  uint64_t* nkey;
  float4* nbox;  // min.x, min.y, max.x, max.y
  float2* ncog;
  uint32_t* nmass;
  int* narrive;
  int cap;
};

BvhPtrs make_ptrs(char* s, const BvhBuildLayout& L) {
  BvhPtrs a;
  a.flags = (int*)(s + L.flags);
  a.qcount = (int*)(s + L.qcount);
  a.queue = (int*)(s + L.queue);
  a.P = (float2*)(s + L.pts);
  a.ID = (uint32_t*)(s + L.ids);
  a.lidx = (int*)(s + L.lidx);
  a.ridx = (int*)(s + L.ridx);
  a.nbegin = (int*)(s + L.nbegin);
  a.nlen = (int*)(s + L.nlen);
  a.nparent = (int*)(s + L.nparent);
  a.nchild = (int*)(s + L.nchild);
  a.ndepth = (int*)(s + L.ndepth);
  a.nleaf = (int*)(s + L.nleaf);
  a.nkey = (uint64_t*)(s + L.nkey);
  a.nbox = (float4*)(s + L.nbox);
  a.ncog = (float2*)(s + L.ncog);
  a.nmass = (uint32_t*)(s + L.nmass);
  a.narrive = (int*)(s + L.narrive);
  a.cap = L.node_cap;
  return a;
}

__host__ __device__ inline int class_of(int len) { return len > 16384 ? 0 : (len > 1024 ? 1 : 2); }

// pathfinder_simd min/max on SSE: `a < b ? a : b` (second operand when unordered); NaNs never get here
__device__ __forceinline__ float sse_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sse_max(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ xsum::Step shfl_up_step(xsum::Step v, int d) {
  xsum::Step r;
  r.a0 = (uint32_t)__shfl_up((int)v.a0, d, 64);
  r.a1 = (uint32_t)__shfl_up((int)v.a1, d, 64);
  return r;
}

template <int BLOCK> struct Shared {
  static constexpr int W = BLOCK / 64;
  xsum::Step wx[W], wy[W];
  unsigned wcnt[W];
  float redf[4][W];
  unsigned redu[2][W];
  int bad;
  uint32_t bad_sx, bad_sy;
};

template <int BLOCK> __device__ __forceinline__ unsigned block_sum(unsigned v, unsigned* slot, int tid) {
  for (int d = 32; d >= 1; d >>= 1) v += (unsigned)__shfl_xor((int)v, d, 64);
  if constexpr (BLOCK == 64) return v;
  if ((tid & 63) == 0) slot[tid >> 6] = v;
  __syncthreads();
  unsigned t = 0;
  for (int w = 0; w < BLOCK / 64; ++w) t += slot[w];
  __syncthreads();
  return t;
}

// ---- init --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bvh_init(BvhPtrs a, const float2* __restrict__ pos, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float2 p = pos[i];
    a.P[i] = p;
    a.ID[i] = (uint32_t)i;
    if (p.x != p.x || p.y != p.y) a.flags[kBvhFallback] = 1;
  }
  if (i == 0) {  // the top call is unconditional: the root is a Root whatever its length (main.rs:400)
    a.nbegin[0] = 0;
    a.nlen[0] = n;
    a.nparent[0] = -1;
    a.nchild[0] = -1;
    a.ndepth[0] = 0;
    a.nleaf[0] = 0;
    a.nkey[0] = 0ull;
    a.flags[kBvhNodeCount] = 1;
    const int cls = class_of(n);
    a.qcount[cls * kBvhLevels] = 1;
    a.queue[(size_t)(cls * 2) * a.cap] = 0;
  }
}

// ---- one level ---------------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void bvh_level(BvhPtrs a, int level, int leaf_size) {
  constexpr int cls = BLOCK == 1024 ? 0 : (BLOCK == 256 ? 1 : 2);
  constexpr int TILE = BLOCK * kEPT;
  constexpr int W = BLOCK / 64;
  __shared__ Shared<BLOCK> sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = a.qcount[cls * kBvhLevels + level];
  const int* queue = a.queue + (size_t)(cls * 2 + (level & 1)) * a.cap;
  int stops = 0;
  for (int qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    const int node = queue[qi];
    const int b = a.nbegin[node], len = a.nlen[node];
    float2* P = a.P + b;
    uint32_t* ID = a.ID + b;
    int* lidx = a.lidx + b;
    int* ridx = a.ridx + b;

    // -- the fold of bvh_tree.rs:58-61: min, max, and the sum as the sequential chain would round it
    float s_x = 0.f, s_y = 0.f;  // uniform across the group
    float mnx = kMaxF, mny = kMaxF, mxx = 0.f, mxy = 0.f;
    int pos = 0;
    bool seq = true;  // the chain starts at 0.0: not in any binade yet
    while (pos < len) {
      xsum::Chain cx, cy;
      if (!seq) {
        const bool okx = xsum::chain_open(s_x, cx), oky = xsum::chain_open(s_y, cy);
        seq = !(okx && oky);
      }
      if (seq) {  // real adds, the same ones in every thread
        const int cnt = len - pos < kSeqRun ? len - pos : kSeqRun;
        for (int k = 0; k < cnt; ++k) {
          const float2 q = P[pos + k];
          s_x = s_x + q.x;
          s_y = s_y + q.y;
          mnx = sse_min(mnx, q.x); mny = sse_min(mny, q.y);
          mxx = sse_max(mxx, q.x); mxy = sse_max(mxy, q.y);
        }
        pos += cnt;
        seq = false;
        continue;
      }
      const int base = pos + tid * kEPT;
      xsum::Step fx[kEPT], fy[kEPT];
      xsum::Step tx{0u, 0u}, ty{0u, 0u};
#pragma unroll
      for (int j = 0; j < kEPT; ++j) {
        fx[j] = xsum::Step{0u, 0u};
        fy[j] = xsum::Step{0u, 0u};
        if (base + j < len) {
          const float2 q = P[base + j];
          fx[j] = xsum::step_of(q.x, cx.sign, cx.E);
          fy[j] = xsum::step_of(q.y, cy.sign, cy.E);
          mnx = sse_min(mnx, q.x); mny = sse_min(mny, q.y);
          mxx = sse_max(mxx, q.x); mxy = sse_max(mxy, q.y);
        }
        tx = xsum::compose(tx, fx[j]);
        ty = xsum::compose(ty, fy[j]);
      }
      xsum::Step ix = tx, iy = ty;  // inclusive scan inside the wave
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const xsum::Step ox = shfl_up_step(ix, d), oy = shfl_up_step(iy, d);
        if (lane >= d) {
          ix = xsum::compose(ox, ix);
          iy = xsum::compose(oy, iy);
        }
      }
      if (lane == 63) { sh.wx[wave] = ix; sh.wy[wave] = iy; }
      if (tid == 0) sh.bad = INT_MAX;
      __syncthreads();
      xsum::Step ex = shfl_up_step(ix, 1), ey = shfl_up_step(iy, 1);  // exclusive: everything before my addends
      if (lane == 0) { ex = xsum::Step{0u, 0u}; ey = xsum::Step{0u, 0u}; }
      xsum::Step px{0u, 0u}, py{0u, 0u};
      for (int w = 0; w < wave; ++w) {
        px = xsum::compose(px, sh.wx[w]);
        py = xsum::compose(py, sh.wy[w]);
      }
      ex = xsum::compose(px, ex);
      ey = xsum::compose(py, ey);
      uint32_t Sx = xsum::apply(cx.S, ex), Sy = xsum::apply(cy.S, ey);
      int bad = INT_MAX;
      uint32_t bsx = 0u, bsy = 0u;
#pragma unroll
      for (int j = 0; j < kEPT; ++j) {
        if (base + j < len && bad == INT_MAX) {
          const uint32_t nx = xsum::apply(Sx, fx[j]), ny = xsum::apply(Sy, fy[j]);
          if (!xsum::in_binade(nx) || !xsum::in_binade(ny)) {
            bad = tid * kEPT + j;
            bsx = Sx;
            bsy = Sy;
          } else {
            Sx = nx;
            Sy = ny;
          }
        }
      }
      if (bad != INT_MAX) atomicMin(&sh.bad, bad);
      __syncthreads();
      const int first_bad = sh.bad;
      if (first_bad == INT_MAX) {
        xsum::Step totx{0u, 0u}, toty{0u, 0u};
        for (int w = 0; w < W; ++w) {
          totx = xsum::compose(totx, sh.wx[w]);
          toty = xsum::compose(toty, sh.wy[w]);
        }
        s_x = xsum::chain_value(cx, xsum::apply(cx.S, totx));
        s_y = xsum::chain_value(cy, xsum::apply(cy.S, toty));
        pos += TILE;
      } else {  // the chain is exact up to the addend before first_bad; that addend takes a real add
        if (bad == first_bad) { sh.bad_sx = bsx; sh.bad_sy = bsy; }
        __syncthreads();
        s_x = xsum::chain_value(cx, sh.bad_sx);
        s_y = xsum::chain_value(cy, sh.bad_sy);
        pos += first_bad;
        seq = true;
        ++stops;
      }
      __syncthreads();  // sh is rewritten by the next round
    }
    // group-wide min / max
    for (int d = 32; d >= 1; d >>= 1) {
      mnx = sse_min(mnx, __shfl_xor(mnx, d, 64)); mny = sse_min(mny, __shfl_xor(mny, d, 64));
      mxx = sse_max(mxx, __shfl_xor(mxx, d, 64)); mxy = sse_max(mxy, __shfl_xor(mxy, d, 64));
    }
    if constexpr (W > 1) {
      if (lane == 0) { sh.redf[0][wave] = mnx; sh.redf[1][wave] = mny; sh.redf[2][wave] = mxx; sh.redf[3][wave] = mxy; }
      __syncthreads();
      for (int w = 0; w < W; ++w) {
        mnx = sse_min(mnx, sh.redf[0][w]); mny = sse_min(mny, sh.redf[1][w]);
        mxx = sse_max(mxx, sh.redf[2][w]); mxy = sse_max(mxy, sh.redf[3][w]);
      }
      __syncthreads();
    }
    const float hx = s_x / (float)len, hy = s_y / (float)len;  // :67

    // -- :70-73: which axis splits closer to the middle
    unsigned cx_ = 0u, cy_ = 0u;
    for (int i = tid; i < len; i += BLOCK) {
      const float2 q = P[i];
      cx_ += q.x > hx;
      cy_ += q.y > hy;
    }
    const int cxs = (int)block_sum<BLOCK>(cx_, sh.redu[0], tid);
    const int cys = (int)block_sum<BLOCK>(cy_, sh.redu[1], tid);
    const int half = len / 2;
    const int hori = half > cxs ? half - cxs : cxs - half;
    const int vert = half > cys ? half - cys : cys - half;
    const bool on_x = vert > hori;
    const int m = on_x ? cxs : cys;  // predicate-true ("greater") side comes first: the split point

    // -- the two-pointer partition (:74-77) as two rank lists
    unsigned carry = 0u, nbad_ = 0u;
    for (int p0 = 0; p0 < len; p0 += TILE) {
      const int base = p0 + tid * kEPT;
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
      unsigned before = carry, total;
      if constexpr (W > 1) {
        if (lane == 63) sh.wcnt[wave] = inc;
        __syncthreads();
        total = 0u;
        for (int w = 0; w < W; ++w) {
          if (w < wave) before += sh.wcnt[w];
          total += sh.wcnt[w];
        }
      } else {
        total = (unsigned)__shfl((int)inc, 63, 64);
      }
      unsigned t = before + inc - mine;  // predicate-true elements before my first one
#pragma unroll
      for (int j = 0; j < kEPT; ++j) {
        const int i = base + j;
        if (i < len) {
          if (i < m && !pr[j]) { lidx[i - (int)t] = i; ++nbad_; }   // k-th misplaced from the left
          else if (i >= m && pr[j]) ridx[m - (int)t - 1] = i;       // k-th misplaced from the right
          t += pr[j];
        }
      }
      carry += total;
      if constexpr (W > 1) __syncthreads();
    }
    const int nbad = (int)block_sum<BLOCK>(nbad_, sh.redu[0], tid);
    __syncthreads();
    for (int k = tid; k < nbad; k += BLOCK) {
      const int l = lidx[k], r = ridx[k];
      const float2 pl = P[l], pr2 = P[r];
      P[l] = pr2; P[r] = pl;
      const uint32_t il = ID[l], ir = ID[r];
      ID[l] = ir; ID[r] = il;
    }

    if (tid == 0) {
      a.nbox[node] = make_float4(mnx, mny, mxx, mxy);
      const int d = a.ndepth[node];
      const int first = atomicAdd(&a.flags[kBvhNodeCount], 2);
      if (first + 2 > a.cap) {
        a.flags[kBvhFallback] = 1;
      } else {
        a.nchild[node] = first;
        const uint64_t path = a.nkey[node] & ~63ull;
        for (int side = 0; side < 2; ++side) {
          const int id = first + side;
          const int cl = side ? len - m : m;
          a.nbegin[id] = side ? b + m : b;
          a.nlen[id] = cl;
          a.nparent[id] = node;
          a.nchild[id] = -1;
          a.ndepth[id] = d + 1;
          bool leaf = !(cl > leaf_size);  // :78-88
          if (!leaf && d + 1 >= kBvhKeyDepth) { a.flags[kBvhFallback] = 1; leaf = true; }
          a.nleaf[id] = leaf ? 1 : 0;
          const int dc = d + 1 > kBvhKeyDepth ? kBvhKeyDepth : d + 1;
          a.nkey[id] = (path | (side ? (1ull << (64 - dc)) : 0ull)) | (uint64_t)dc;
          if (!leaf) {
            const int c2 = class_of(cl);
            const int slot = atomicAdd(&a.qcount[c2 * kBvhLevels + level + 1], 1);
            a.queue[(size_t)(c2 * 2 + ((level + 1) & 1)) * a.cap + slot] = id;
          }
        }
        atomicMax(&a.flags[kBvhMaxDepth], d + 1);
      }
    }
    __syncthreads();
  }
  if (tid == 0 && stops) atomicAdd(&a.flags[kBvhStops], stops);
}

// ---- numbering, leaves, upward pass ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bvh_keys(BvhPtrs a, int m, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) vals[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void bvh_rank(const uint32_t* __restrict__ vals_sorted, int m, int* __restrict__ rank) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < m) rank[vals_sorted[r]] = r;
}

// written by another compute unit in this very kernel: read past the local L1
__device__ __forceinline__ uint32_t load_agent(const uint32_t* p) {
  return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_agent(const float* p) {
  return xsum::u2f(load_agent(reinterpret_cast<const uint32_t*>(p)));
}

// One thread per leaf: make_leaf (:40-54), leaf mass and centre (:98-131), then up the parent chain; the second child
// to arrive at a node computes it (:133-158).
__global__ __launch_bounds__(64) void bvh_upward(BvhPtrs a, const uint32_t* __restrict__ weight, int m) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= m || !a.nleaf[i]) return;
  const int b = a.nbegin[i], len = a.nlen[i];
  float mnx = kMaxF, mny = kMaxF, mxx = 0.f, mxy = 0.f, sx = 0.f, sy = 0.f;
  uint32_t ms = 0u;
  for (int k = 0; k < len; ++k) {
    const float2 q = a.P[b + k];
    mnx = sse_min(mnx, q.x); mny = sse_min(mny, q.y);
    mxx = sse_max(mxx, q.x); mxy = sse_max(mxy, q.y);
    sx = sx + q.x;
    sy = sy + q.y;
    ms += weight[a.ID[b + k]];  // u32, wraps like the release build
  }
  a.nbox[i] = make_float4(mnx, mny, mxx, mxy);
  float cgx = sx / (float)len, cgy = sy / (float)len;  // NaN for an empty leaf, as upstream
  int cur = i;
  for (;;) {
    a.ncog[cur] = make_float2(cgx, cgy);
    a.nmass[cur] = ms;
    const int p = a.nparent[cur];
    if (p < 0) break;
    __threadfence();
    if (atomicAdd(&a.narrive[p], 1) == 0) break;  // the sibling finishes this parent
    __threadfence();
    const int c0 = a.nchild[p], c1 = c0 + 1;
    const float* cog = (const float*)a.ncog;
    const float x0 = load_agent(cog + 2 * c0), y0 = load_agent(cog + 2 * c0 + 1);
    const float x1 = load_agent(cog + 2 * c1), y1 = load_agent(cog + 2 * c1 + 1);
    const uint32_t m0 = load_agent(a.nmass + c0), m1 = load_agent(a.nmass + c1);
    ms = m0 + m1;                                                   // :148
    const float bx = (x0 * (float)m0) + (x1 * (float)m1);           // :150-153
    const float by = (y0 * (float)m0) + (y1 * (float)m1);
    cgx = bx / (float)ms;                                           // :154
    cgy = by / (float)ms;
    cur = p;
  }
}

__global__ __launch_bounds__(256) void bvh_emit(BvhPtrs a, int m, const int* __restrict__ rank,
                                                const uint64_t* __restrict__ keys_sorted, float4* __restrict__ geom0,
                                                float4* __restrict__ geom1, int4* __restrict__ link, int* __restrict__ depth_out,
                                                uint32_t* __restrict__ mass_out, float2* __restrict__ size_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int idx = rank[i];
  const int d = a.ndepth[i];
  const float4 box = a.nbox[i];
  const float w = box.z - box.x, h = box.w - box.y;  // boundary.size = max - min (:63-66)
  const float2 cog = a.ncog[i];
  const uint32_t ms = a.nmass[i];
  const float tx = sse_max(w, h), ty = sse_max(h, w);  // size.max(size.yx()), main.rs:371
  geom0[idx] = make_float4(box.x, box.y, box.x + w, box.y + h);
  geom1[idx] = make_float4(cog.x, cog.y, (float)ms, tx * ty);
  int skip;
  if (a.nleaf[i]) {
    skip = idx + 1;
  } else if (d == 0) {
    skip = m;
  } else {
    const uint64_t upper = (a.nkey[i] & ~63ull) + (1ull << (64 - d));  // first path after this subtree
    if (upper == 0ull) {
      skip = m;
    } else {
      int lo = idx + 1, hi = m;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < upper) lo = mid + 1; else hi = mid;
      }
      skip = lo;
    }
  }
  link[idx] = make_int4(skip, a.nbegin[i], a.nlen[i], a.nleaf[i]);
  depth_out[idx] = d;
  mass_out[idx] = ms;
  size_out[idx] = make_float2(w, h);
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

BvhBuildLayout bvh_build_layout(int64_t n, int leaf_size) {
  BvhBuildLayout L{};
  const size_t N = (size_t)(n > 0 ? n : 1);
  // leaves end up between half full and full: ~2.7 N / leaf_size nodes.  More than this -> host builder
  const size_t lf = (size_t)(leaf_size > 0 ? leaf_size : 1);
  const size_t C = (4 * N / lf < 2 * N ? 4 * N / lf : 2 * N) + 4096;
  L.node_cap = (int)C;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  L.flags = take(sizeof(int) * kBvhFlagWords);
  L.qcount = take(sizeof(int) * kBvhClasses * kBvhLevels);
  L.queue = take(sizeof(int) * kBvhClasses * 2 * C);
  L.pts = take(sizeof(float2) * N);
  L.ids = take(4 * N);
  L.lidx = take(4 * N);
  L.ridx = take(4 * N);
  L.nbegin = take(4 * C);
  L.nlen = take(4 * C);
  L.nparent = take(4 * C);
  L.nchild = take(4 * C);
  L.ndepth = take(4 * C);
  L.nleaf = take(4 * C);
  L.nkey = take(8 * C);
  L.nbox = take(16 * C);
  L.ncog = take(8 * C);
  L.nmass = take(4 * C);
  L.narrive = take(4 * C);
  L.keys_sorted = take(8 * C);
  L.vals = take(4 * C);
  L.vals_sorted = take(4 * C);
  L.rank = take(4 * C);
  size_t tb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (int)C, 0, 64, (hipStream_t) nullptr);
  L.cub_temp_bytes = tb;
  L.cub_temp = take(tb);
  L.total = off;
  return L;
}

hipError_t bvh_build_begin(hipStream_t s, const void* pos, int n, char* scratch, const BvhBuildLayout& L) {
  hipError_t e = hipMemsetAsync(scratch + L.flags, 0, L.queue - L.flags, s);  // flags + level counters
  if (e != hipSuccess) return e;
  BvhPtrs a = make_ptrs(scratch, L);
  bvh_init<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a, (const float2*)pos, n);
  return hipGetLastError();
}

hipError_t bvh_build_levels(hipStream_t s, int n, int leaf_size, int level_begin, int level_end, char* scratch,
                            const BvhBuildLayout& L) {
  BvhPtrs a = make_ptrs(scratch, L);
  for (int level = level_begin; level < level_end && level < kBvhLevels - 1; ++level) {
    const int64_t width = level < 30 ? (int64_t)1 << level : (int64_t)1 << 30;  // a level never has more nodes than this
    auto grid = [&](int64_t min_len, int64_t cap) {
      int64_t g = n / min_len + 1;
      if (g > width) g = width;
      if (g > cap) g = cap;
      return dim3((unsigned)(g < 1 ? 1 : g));
    };
    if (n > 16384) bvh_level<1024><<<grid(16385, 512), dim3(1024), 0, s>>>(a, level, leaf_size);
    if (n > 1024) bvh_level<256><<<grid(1025, 2048), dim3(256), 0, s>>>(a, level, leaf_size);
    bvh_level<64><<<grid(leaf_size > 0 ? leaf_size + 1 : 1, 8192), dim3(64), 0, s>>>(a, level, leaf_size);
  }
  return hipGetLastError();
}

hipError_t bvh_build_finish(hipStream_t s, const uint32_t* weight, int n, int n_nodes, char* scratch, const BvhBuildLayout& L,
                            uint32_t* order_out, void* geom0, void* geom1, void* link, int* depth_out, uint32_t* mass_out,
                            float2* size_out) {
  BvhPtrs a = make_ptrs(scratch, L);
  const int m = n_nodes;
  uint32_t* vals = (uint32_t*)(scratch + L.vals);
  uint32_t* vals_sorted = (uint32_t*)(scratch + L.vals_sorted);
  uint64_t* keys_sorted = (uint64_t*)(scratch + L.keys_sorted);
  int* rank = (int*)(scratch + L.rank);
  const dim3 gm((unsigned)((m + 255) / 256));
  hipError_t e = hipMemsetAsync(a.narrive, 0, sizeof(int) * (size_t)m, s);
  if (e != hipSuccess) return e;
  bvh_keys<<<gm, dim3(256), 0, s>>>(a, m, vals);
  size_t tb = L.cub_temp_bytes;
  e = hipcub::DeviceRadixSort::SortPairs((void*)(scratch + L.cub_temp), tb, (const uint64_t*)a.nkey, keys_sorted,
                                         (const uint32_t*)vals, vals_sorted, m, 0, 64, s);
  if (e != hipSuccess) return e;
  bvh_rank<<<gm, dim3(256), 0, s>>>(vals_sorted, m, rank);
  bvh_upward<<<dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s>>>(a, weight, m);
  bvh_emit<<<gm, dim3(256), 0, s>>>(a, m, rank, keys_sorted, (float4*)geom0, (float4*)geom1, (int4*)link, depth_out, mass_out,
                                    size_out);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(order_out, a.ID, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, s);
}

}  // namespace nbody

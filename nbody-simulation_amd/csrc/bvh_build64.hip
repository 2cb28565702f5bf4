// Device-side BVH build for f64 positions (gfx950).  Same tree as the host builder (tree_build.hpp, build_bvh<double>) and
// as the oracle, node for node — BVHTree::from + make_leaf + calculate_gravity of /root/reference src/bvh_tree.rs:40-158
// carried over to f64 (the reference itself is f32: this is the extension BASELINE's f64 configurations use).
//
// bvh_build.hip (f32) is tuned to the last microsecond for the reference's own scene; this one is the same mathematics
// written plainly, level by level over ALL open nodes at once, because what it replaces is a 50-65 ms host build at
// N = 4 M (HISTORY.md §4.3c), not a 0.7 ms device one:
//   per level   fold64      one work-group per (open node, coordinate): min / max and the sum EXACTLY as the sequential chain
//                           `sum = sum + p` rounds it (bvh_tree.rs:58-61) — the parity-map scan of exact_sum64.h, a tile of
//                           addends per round, real adds wherever the chain leaves its binade;
//               plan64      mean = sum / len (:67); the node's 2048-point chunks join the level's chunk table;
//               count64     per chunk: #{x > mean.x}, #{y > mean.y} (:70-72) -> the node's counters;
//               mis64       axis rule `vert > hori -> x` (:73), split = the count of predicate-true points; per chunk the
//                           number of misplaced points left and right of the split;
//               children64  per node: prefix of those counts over its chunks; the two children (left = the "greater"
//                           side, :78-88), open ones queued for the next level;
//               ranks64     per chunk: the k-th misplaced point from the left / from the right -> two rank lists;
//               swap64      the crate's two-pointer partition (:74-77) swaps exactly the k-th misplaced from the left with
//                           the k-th from the right: independent swaps.
//   at the end  leaves64 (box from f64::MAX / 0.0, unweighted mean in slice order, u32 wrapping mass), one upward launch
//               per depth (centre of gravity as written, :150-154; subtree sizes), one downward launch per depth
//               (pre-order numbers), emit64 (the linearised arrays the walks read), the row permutation.
// Every launch is enqueued blind over as many levels as the caller names; a level without open nodes costs a few
// microseconds.  NaN positions (pathfinder's minps / maxps are order-dependent there), more nodes than the buffers hold or
// open nodes left after the last level raise flags the caller reads once.
//
// Compiled with -ffp-contract=off: every operation is the one the reference writes.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include <algorithm>

#include "bvh_build64.h"
#include "exact_sum64.h"

namespace nbody {

namespace {

constexpr int kChunk = 2048;   // points per work-group in the per-chunk passes (256 threads x 8)
constexpr int kEPT = 8;        // ... per thread
constexpr int kSeqRun = 16;    // real adds after every stop of the scan
constexpr int kSeqStart = 64;  // ... and at the start of a chain (the sum doubles every few addends there)
constexpr double kMaxD = 1.7976931348623157e308;

struct Ptrs {
  int* flags;
  int* opencount;   // [level]
  int* chunkcount;  // [level]
  double2* P;       // working copy of the positions, permuted in place
  uint32_t* ID;     // which input row sits here
  int *nbegin, *nlen, *ndepth, *nchild, *nleaf, *ncx, *ncy, *nsplit, *naxis, *nk, *nchunk0, *nsub, *npre;
  double2 *nsum, *nmin, *nmax, *nmean, *ncog;
  uint32_t* nmass;
  int* openq;       // [level & 1][open_cap]
  int *ch_node, *ch_index, *ch_l, *ch_r, *ch_loff, *ch_roff;  // the current level's chunk table
  uint32_t *lidx, *ridx;
  // long nodes (> kLongNode points): their chains cut into kSeg-addend segments with a prepared run each
  int* segcount;  // [level]
  int *nseg0, *seg_node, *seg_index;
  double2 *seg_sum, *seg_min, *seg_max, *seg_prefix;
  uint64_t* seg_pred;       // [seg][coord]: predicted (sign << 32 | E), 0 = no run
  xsum64::Run* seg_run;     // [seg][coord]
  int node_cap, open_cap, chunk_cap, seg_cap, n;
};

Ptrs make_ptrs(char* s, const Bvh64Layout& L, int n) {
  Ptrs a;
  a.flags = (int*)(s + L.flags);
  a.opencount = (int*)(s + L.opencount);
  a.chunkcount = (int*)(s + L.chunkcount);
  a.P = (double2*)(s + L.P);
  a.ID = (uint32_t*)(s + L.ID);
  a.nbegin = (int*)(s + L.nbegin); a.nlen = (int*)(s + L.nlen); a.ndepth = (int*)(s + L.ndepth); a.nchild = (int*)(s + L.nchild);
  a.nleaf = (int*)(s + L.nleaf); a.ncx = (int*)(s + L.ncx); a.ncy = (int*)(s + L.ncy); a.nsplit = (int*)(s + L.nsplit);
  a.naxis = (int*)(s + L.naxis); a.nk = (int*)(s + L.nk); a.nchunk0 = (int*)(s + L.nchunk0); a.nsub = (int*)(s + L.nsub);
  a.npre = (int*)(s + L.npre);
  a.nsum = (double2*)(s + L.nsum); a.nmin = (double2*)(s + L.nmin); a.nmax = (double2*)(s + L.nmax); a.nmean = (double2*)(s + L.nmean);
  a.ncog = (double2*)(s + L.ncog);
  a.nmass = (uint32_t*)(s + L.nmass);
  a.openq = (int*)(s + L.openq);
  a.ch_node = (int*)(s + L.ch_node); a.ch_index = (int*)(s + L.ch_index); a.ch_l = (int*)(s + L.ch_l); a.ch_r = (int*)(s + L.ch_r);
  a.ch_loff = (int*)(s + L.ch_loff); a.ch_roff = (int*)(s + L.ch_roff);
  a.lidx = (uint32_t*)(s + L.lidx); a.ridx = (uint32_t*)(s + L.ridx);
  a.segcount = (int*)(s + L.segcount);
  a.nseg0 = (int*)(s + L.nseg0); a.seg_node = (int*)(s + L.seg_node); a.seg_index = (int*)(s + L.seg_index);
  a.seg_sum = (double2*)(s + L.seg_sum); a.seg_min = (double2*)(s + L.seg_min); a.seg_max = (double2*)(s + L.seg_max);
  a.seg_prefix = (double2*)(s + L.seg_prefix);
  a.seg_pred = (uint64_t*)(s + L.seg_pred);
  a.seg_run = (xsum64::Run*)(s + L.seg_run);
  a.node_cap = L.node_cap; a.open_cap = L.open_cap; a.chunk_cap = L.chunk_cap; a.seg_cap = L.seg_cap; a.n = n;
  return a;
}

__device__ __forceinline__ double sse_min(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double sse_max(double a, double b) { return a > b ? a : b; }

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
  const unsigned lo = (unsigned)__shfl_up((int)(unsigned)v, d, 64), hi = (unsigned)__shfl_up((int)(unsigned)(v >> 32), d, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ xsum64::Step shfl_up_step(xsum64::Step v, int d) { return {shfl_up_u64(v.a0, d), shfl_up_u64(v.a1, d)}; }

// ---- begin: copy the rows in, seed the root --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void b64_init(Ptrs a, const double2* __restrict__ pos) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n) {
    const double2 p = pos[i];
    a.P[i] = p;
    a.ID[i] = (uint32_t)i;
    if (p.x != p.x || p.y != p.y) a.flags[kB64Fallback] = 1;  // NaN: the reference's fold is order-dependent there
  }
  if (i == 0) {
    a.flags[kB64NodeCount] = 1;
    a.nbegin[0] = 0; a.nlen[0] = a.n; a.ndepth[0] = 0; a.nchild[0] = -1; a.nleaf[0] = 0; a.npre[0] = 0;
    a.opencount[0] = 1;  // the top call is unconditional: the root is always a Root (main.rs:400)
    a.openq[0] = 0;
  }
}

// ---- per level -------------------------------------------------------------------------------------------------------
constexpr int kWideLevels = 32;   // levels with at most this many nodes may hold long ones: 8192-addend tiles, prepared segments
                                  // (kLongNode 16384 / 512 levels measured no better: 10.7 vs 10.2 ms at N = 4 M)
constexpr int kSmallNode = 256;  // nodes up to this long are folded by one thread per coordinate, in order (b64_fold_small)
constexpr int kSeg = 8192;        // a long node's chain is cut into segments of this many addends (= the fold's tile), ...
constexpr int kLongNode = 65536;  // ... long meaning more points than this: their runs are prepared by the whole chip

// ---- long nodes: the chain's segments prepared in parallel ---------------------------------------------------------------
// One work-group walks a long node's chain (b64_fold below), ~10 us per 8192-addend scan round on its one compute unit: 5 ms
// for a 4 M-point root.  But a segment's effect on the chain — the map {state at its start} -> {state at its end} — can be
// prepared by any other work-group beforehand, IF it is told which binade the chain will be in: that is predicted from
// plain f64 partial sums, and the run carries the bounds that prove, when it is applied, that every add inside stayed in
// that binade (exact_sum64.h: Run, run_fits).  A wrong prediction or a crossing costs that segment's scan, never a bit.
__global__ __launch_bounds__(64) void b64_seg_plan(Ptrs a, int level) {
  const int nopen = a.opencount[level] < a.open_cap ? a.opencount[level] : a.open_cap;
  const int lane = threadIdx.x;
  for (int q = blockIdx.x; q < nopen; q += gridDim.x) {
    const int node = a.openq[(size_t)(level & 1) * a.open_cap + q];
    const int len = a.nlen[node];
    const int ns = len > kLongNode ? len / kSeg : 0;  // whole segments only: the tail is scanned
    int s0 = -1;
    if (lane == 0) {
      if (ns > 0) {
        s0 = atomicAdd(&a.segcount[level], ns);
        if (s0 + ns > a.seg_cap) s0 = -1;  // (cannot happen: the table holds every whole segment of a level)
      }
      a.nseg0[node] = s0;
    }
    s0 = __shfl(s0, 0, 64);
    if (s0 >= 0)
      for (int k = lane; k < ns; k += 64) {
        a.seg_node[s0 + k] = node;
        a.seg_index[s0 + k] = k;
      }
  }
}

// plain f64 sum (any order: it only predicts), min and max of every segment
__global__ __launch_bounds__(512) void b64_seg_sums(Ptrs a, int level) {
  __shared__ double red[3][8];
  const int nseg = a.segcount[level] < a.seg_cap ? a.segcount[level] : a.seg_cap;
  const int comp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
    const int node = a.seg_node[sg], k = a.seg_index[sg];
    const double* __restrict__ X = reinterpret_cast<const double*>(a.P + a.nbegin[node] + (size_t)k * kSeg) + comp;
    double sum = 0.0, mn = kMaxD, mx = 0.0;
    for (int e = tid; e < kSeg; e += 512) {
      const double v = X[2 * (size_t)e];
      sum += v;
      mn = sse_min(mn, v);
      mx = sse_max(mx, v);
    }
    for (int d = 32; d >= 1; d >>= 1) {
      sum += __shfl_xor(sum, d, 64);
      mn = sse_min(mn, __shfl_xor(mn, d, 64));
      mx = sse_max(mx, __shfl_xor(mx, d, 64));
    }
    if (lane == 0) { red[0][wave] = sum; red[1][wave] = mn; red[2][wave] = mx; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 8; ++w) { sum += red[0][w]; mn = sse_min(mn, red[1][w]); mx = sse_max(mx, red[2][w]); }
      if (comp == 0) { a.seg_sum[sg].x = sum; a.seg_min[sg].x = mn; a.seg_max[sg].x = mx; }
      else { a.seg_sum[sg].y = sum; a.seg_min[sg].y = mn; a.seg_max[sg].y = mx; }
    }
    __syncthreads();
  }
}

// the predicted value of the chain at the start of every segment: a thread per long node and coordinate
__global__ __launch_bounds__(64) void b64_seg_prefix(Ptrs a, int level) {
  const int nopen = a.opencount[level] < a.open_cap ? a.opencount[level] : a.open_cap;
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int q = t >> 1, comp = t & 1;
  if (q >= nopen) return;
  const int node = a.openq[(size_t)(level & 1) * a.open_cap + q];
  const int s0 = a.nseg0[node];
  if (s0 < 0) return;
  const int ns = a.nlen[node] / kSeg;
  double p = 0.0;
  for (int k = 0; k < ns; ++k) {
    if (comp == 0) { a.seg_prefix[s0 + k].x = p; p += a.seg_sum[s0 + k].x; }
    else { a.seg_prefix[s0 + k].y = p; p += a.seg_sum[s0 + k].y; }
  }
}

__device__ __forceinline__ int64_t shfl_up_i64(int64_t v, int d) { return (int64_t)shfl_up_u64((uint64_t)v, d); }
__device__ __forceinline__ xsum64::Run shfl_up_run(const xsum64::Run& r, int d) {
  xsum64::Run o;
  o.a0 = shfl_up_i64(r.a0, d); o.a1 = shfl_up_i64(r.a1, d);
  o.lo0 = shfl_up_i64(r.lo0, d); o.lo1 = shfl_up_i64(r.lo1, d);
  o.hi0 = shfl_up_i64(r.hi0, d); o.hi1 = shfl_up_i64(r.hi1, d);
  return o;
}

// the run of every segment, for the binade its two ends are predicted to lie in
__global__ __launch_bounds__(512) void b64_seg_runs(Ptrs a, int level) {
  __shared__ double stage[kSeg + kSeg / 16];
  __shared__ xsum64::Run wrun[8];
  const int nseg = a.segcount[level] < a.seg_cap ? a.segcount[level] : a.seg_cap;
  const int comp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
    const double p0 = comp == 0 ? a.seg_prefix[sg].x : a.seg_prefix[sg].y;
    const double p1 = p0 + (comp == 0 ? a.seg_sum[sg].x : a.seg_sum[sg].y);
    xsum64::Chain c;
    if (!xsum64::predict_binade(p0, p1, c)) {  // (uniform) the chain's start, a crossing, a sum near zero: scanned by the fold
      if (tid == 0) a.seg_pred[2 * (size_t)sg + comp] = 0;
      continue;
    }
    const int node = a.seg_node[sg], k = a.seg_index[sg];
    const double* __restrict__ X = reinterpret_cast<const double*>(a.P + a.nbegin[node] + (size_t)k * kSeg) + comp;
    for (int e = tid; e < kSeg; e += 512) stage[e + (e >> 4)] = X[2 * (size_t)e];
    __syncthreads();
    xsum64::Run r = xsum64::run_none();
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int e = tid * 16 + j;
      r = xsum64::run_then(r, xsum64::run_of(xsum64::step_of(stage[e + (e >> 4)], c.sign, c.E)));
    }
    for (int d = 1; d < 64; d <<= 1) {  // in thread order: lane 63 ends up with the wave's run
      const xsum64::Run o = shfl_up_run(r, d);
      if (lane >= d) r = xsum64::run_then(o, r);
    }
    if (lane == 63) wrun[wave] = r;
    __syncthreads();
    if (tid == 0) {
      xsum64::Run t = wrun[0];
      for (int w = 1; w < 8; ++w) t = xsum64::run_then(t, wrun[w]);
      a.seg_run[2 * (size_t)sg + comp] = t;
      a.seg_pred[2 * (size_t)sg + comp] = (c.sign << 32) | c.E;
    }
    __syncthreads();
  }
}

// One coordinate of the fold of bvh_tree.rs:58-61 over a node: min, max and the sum exactly as the sequential chain rounds it.
// A scan round covers TILE = NT * EPT consecutive addends: they are fetched coalesced (the rows are double2: a thread reading
// its own 16 consecutive rows straight from memory touches 16 cache lines for 128 useful bytes) into LDS, padded by one
// word per 16 so that the threads' consecutive reads spread over the banks.
template <int NT, int kFoldEPT>
__global__ __launch_bounds__(NT) void b64_fold(Ptrs a, int level) {
  constexpr int NW = NT / 64, TILE = NT * kFoldEPT;
  __shared__ double stage[TILE + TILE / 16];
  __shared__ xsum64::Step wtot[NW];
  __shared__ unsigned long long sh_S;
  __shared__ int sh_bad;
  __shared__ double red[2][NW];
  const int nopen = a.opencount[level] < a.open_cap ? a.opencount[level] : a.open_cap;
  const int comp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int q = blockIdx.x; q < nopen; q += gridDim.x) {
    const int node = a.openq[(size_t)(level & 1) * a.open_cap + q];
    const int b = a.nbegin[node], len = a.nlen[node];
    if (len <= kSmallNode) continue;  // (uniform) b64_fold_small's
    const double* __restrict__ X = reinterpret_cast<const double*>(a.P + b) + comp;  // element i at X[2 i]
    double mn = kMaxD, mx = 0.0;  // min from f64::MAX, max from 0.0 (bvh_tree.rs:59): order-free without NaNs
    double s = 0.0;               // the chain; uniform across the group
    int pos = 0, stops = 0;
    const int seg0 = (TILE == kSeg && len > kLongNode) ? a.nseg0[node] : -1;  // this node's prepared segments, if any
    while (pos < len) {
      xsum64::Chain c;
      if (!xsum64::chain_open(s, c)) {  // not inside a binade (zero, subnormal, a power of two, non-finite): real adds
        int cnt = pos == 0 ? kSeqStart : kSeqRun;
        cnt = len - pos < cnt ? len - pos : cnt;
        for (int k = 0; k < cnt; ++k) {  // every thread the same adds: no hand-over
          const double v = X[2 * (size_t)(pos + k)];
          s = s + v;
          mn = sse_min(mn, v);
          mx = sse_max(mx, v);
        }
        pos += cnt;
        continue;
      }
      int cnt = len - pos < TILE ? len - pos : TILE;
      if (seg0 >= 0) {
        const int in_seg = pos & (kSeg - 1);
        if (in_seg == 0 && pos + kSeg <= len) {  // a whole prepared segment: if its run holds from here, take it in one step
          const size_t sg = 2 * (size_t)(seg0 + (pos >> 13)) + comp;
          static_assert(kSeg == 1 << 13, "pos >> 13 is the segment index");
          if (a.seg_pred[sg] == ((c.sign << 32) | c.E)) {
            const xsum64::Run r = a.seg_run[sg];
            if (xsum64::run_fits(c.S, r)) {
              s = xsum64::chain_value(c, (uint64_t)((int64_t)c.S + ((c.S & 1ull) ? r.a1 : r.a0)));
              const double2 smn = a.seg_min[sg >> 1], smx = a.seg_max[sg >> 1];
              mn = sse_min(mn, comp == 0 ? smn.x : smn.y);
              mx = sse_max(mx, comp == 0 ? smx.x : smx.y);
              pos += kSeg;
              if (tid == 0) atomicAdd(&a.flags[kB64RunsUsed], 1);
              continue;
            }
          }
        }
        if (cnt > kSeg - in_seg) cnt = kSeg - in_seg;  // stay aligned with the segments
      }
      const int end = pos + cnt;
      for (int e = tid; e < cnt; e += NT) stage[e + (e >> 4)] = X[2 * (size_t)(pos + e)];
      if (tid == 0) sh_bad = INT_MAX;
      __syncthreads();
      const int base = tid * kFoldEPT;  // tile-relative
      xsum64::Step f[kFoldEPT];
      xsum64::Step F = xsum64::identity();
#pragma unroll
      for (int j = 0; j < kFoldEPT; ++j) {
        if (base + j < cnt) {
          const double v = stage[base + j + ((base + j) >> 4)];
          f[j] = xsum64::step_of(v, c.sign, c.E);
        } else {
          f[j] = xsum64::identity();
        }
        F = xsum64::compose(F, f[j]);
      }
      xsum64::Step inc = F;  // inclusive scan of the threads' maps, in thread order
      for (int d = 1; d < 64; d <<= 1) {
        const xsum64::Step o = shfl_up_step(inc, d);
        if (lane >= d) inc = xsum64::compose(o, inc);
      }
      if (lane == 63) wtot[wave] = inc;
      __syncthreads();
      xsum64::Step excl = xsum64::identity();
      for (int w = 0; w < wave; ++w) excl = xsum64::compose(excl, wtot[w]);
      xsum64::Step prev = shfl_up_step(inc, 1);
      if (lane == 0) prev = xsum64::identity();
      excl = xsum64::compose(excl, prev);
      uint64_t S = xsum64::apply(c.S, excl);  // the state this thread's first addend meets (if every earlier add stayed in the binade)
      int bad = INT_MAX;
      uint64_t S_at_bad = 0;
#pragma unroll
      for (int j = 0; j < kFoldEPT; ++j) {
        if (base + j < cnt && bad == INT_MAX) {
          const uint64_t after = xsum64::apply(S, f[j]);
          if (!xsum64::in_binade(after)) { bad = base + j; S_at_bad = S; }
          else S = after;
        }
      }
      if (bad != INT_MAX) atomicMin(&sh_bad, bad);
      __syncthreads();
      const int first_bad = sh_bad;  // tile-relative; the earliest one is right: everything before it stayed inside the binade
      const int used = first_bad == INT_MAX ? cnt : first_bad;
      for (int e = tid; e < used; e += NT) {  // min / max of what the chain has taken
        const double v = stage[e + (e >> 4)];
        mn = sse_min(mn, v);
        mx = sse_max(mx, v);
      }
      if (first_bad == INT_MAX) {
        if (base <= cnt - 1 && cnt - 1 < base + kFoldEPT) sh_S = S;  // the owner of the tile's last addend
        __syncthreads();
        s = xsum64::chain_value(c, sh_S);
        pos = end;
      } else {
        if (bad == first_bad) sh_S = S_at_bad;
        __syncthreads();
        s = xsum64::chain_value(c, sh_S);
        ++stops;
        pos += first_bad;
        const int run = cnt - first_bad < kSeqRun ? cnt - first_bad : kSeqRun;  // real adds, as far as the tile holds them
        for (int k = 0; k < run; ++k) {
          const double v = stage[first_bad + k + ((first_bad + k) >> 4)];
          s = s + v;
          if (tid == 0) { mn = sse_min(mn, v); mx = sse_max(mx, v); }
        }
        pos += run;
      }
      __syncthreads();  // stage, sh_S, sh_bad and wtot are reused by the next round
    }
    // the sequential parts were added to every thread's mn / mx alike or to thread 0's only: either way the reduction is right
    for (int d = 32; d >= 1; d >>= 1) {
      mn = sse_min(mn, __shfl_xor(mn, d, 64));
      mx = sse_max(mx, __shfl_xor(mx, d, 64));
    }
    if (lane == 0) { red[0][wave] = mn; red[1][wave] = mx; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NW; ++w) { mn = sse_min(mn, red[0][w]); mx = sse_max(mx, red[1][w]); }
      if (comp == 0) { a.nsum[node].x = s; a.nmin[node].x = mn; a.nmax[node].x = mx; }
      else { a.nsum[node].y = s; a.nmin[node].y = mn; a.nmax[node].y = mx; }
      if (stops) atomicAdd(&a.flags[kB64Stops], stops);
    }
    __syncthreads();
  }
}

// Short nodes: the chain as written, one thread per node and coordinate.
__global__ __launch_bounds__(256) void b64_fold_small(Ptrs a, int level) {
  const int nopen = a.opencount[level] < a.open_cap ? a.opencount[level] : a.open_cap;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int q = t >> 1, comp = t & 1;
  if (q >= nopen) return;
  const int node = a.openq[(size_t)(level & 1) * a.open_cap + q];
  const int b = a.nbegin[node], len = a.nlen[node];
  if (len > kSmallNode) return;
  const double* __restrict__ X = reinterpret_cast<const double*>(a.P + b) + comp;
  double mn = kMaxD, mx = 0.0, s = 0.0;
  for (int i = 0; i < len; ++i) {
    const double v = X[2 * (size_t)i];
    s = s + v;
    mn = sse_min(mn, v);
    mx = sse_max(mx, v);
  }
  if (comp == 0) { a.nsum[node].x = s; a.nmin[node].x = mn; a.nmax[node].x = mx; }
  else { a.nsum[node].y = s; a.nmin[node].y = mn; a.nmax[node].y = mx; }
}

// mean (bvh_tree.rs:67), the node's chunks into the level's table, counters to zero.  One wave per open node.
__global__ __launch_bounds__(64) void b64_plan(Ptrs a, int level) {
  const int nopen = a.opencount[level] < a.open_cap ? a.opencount[level] : a.open_cap;
  const int lane = threadIdx.x;
  for (int q = blockIdx.x; q < nopen; q += gridDim.x) {
    const int node = a.openq[(size_t)(level & 1) * a.open_cap + q];
    const int len = a.nlen[node];
    const int nch = (len + kChunk - 1) / kChunk;
    int c0 = 0;
    if (lane == 0) {
      c0 = atomicAdd(&a.chunkcount[level], nch);
      const double2 s = a.nsum[node];
      a.nmean[node] = make_double2(s.x / (double)len, s.y / (double)len);
      a.ncx[node] = 0; a.ncy[node] = 0;
      a.nchunk0[node] = c0;
      if (c0 + nch > a.chunk_cap) a.flags[kB64Fallback] = 1;
    }
    c0 = __shfl(c0, 0, 64);
    for (int c = lane; c < nch && c0 + c < a.chunk_cap; c += 64) {
      a.ch_node[c0 + c] = node;
      a.ch_index[c0 + c] = c;
    }
  }
}

// The same with one THREAD per open node: below the top levels a node has a chunk or two, and a wave per node idles 63 lanes.
__global__ __launch_bounds__(256) void b64_plan_t(Ptrs a, int level) {
  const int nopen = a.opencount[level] < a.open_cap ? a.opencount[level] : a.open_cap;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nopen) return;
  const int node = a.openq[(size_t)(level & 1) * a.open_cap + q];
  const int len = a.nlen[node];
  const int nch = (len + kChunk - 1) / kChunk;
  const int c0 = atomicAdd(&a.chunkcount[level], nch);
  const double2 s = a.nsum[node];
  a.nmean[node] = make_double2(s.x / (double)len, s.y / (double)len);
  a.ncx[node] = 0; a.ncy[node] = 0;
  a.nchunk0[node] = c0;
  if (c0 + nch > a.chunk_cap) a.flags[kB64Fallback] = 1;
  for (int c = 0; c < nch && c0 + c < a.chunk_cap; ++c) {
    a.ch_node[c0 + c] = node;
    a.ch_index[c0 + c] = c;
  }
}

__device__ __forceinline__ int block_sum(int v, int* slot) {  // 256 threads; every thread gets the total
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  const int t = slot[0] + slot[1] + slot[2] + slot[3];
  __syncthreads();
  return t;
}

// #{x > mean.x}, #{y > mean.y} (strict), bvh_tree.rs:70-72
__global__ __launch_bounds__(256) void b64_count(Ptrs a, int level) {
  __shared__ int slot[4];
  const int nch = a.chunkcount[level] < a.chunk_cap ? a.chunkcount[level] : a.chunk_cap;
  for (int ch = blockIdx.x; ch < nch; ch += gridDim.x) {
    const int node = a.ch_node[ch], ci = a.ch_index[ch];
    const int b = a.nbegin[node], len = a.nlen[node];
    const double2 h = a.nmean[node];
    int cx = 0, cy = 0;
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      const int i = ci * kChunk + j * 256 + (int)threadIdx.x;
      if (i < len) {
        const double2 p = a.P[b + i];
        cx += p.x > h.x;
        cy += p.y > h.y;
      }
    }
    cx = block_sum(cx, slot);
    cy = block_sum(cy, slot);
    if (threadIdx.x == 0) {
      atomicAdd(&a.ncx[node], cx);
      atomicAdd(&a.ncy[node], cy);
    }
  }
}

// the axis rule and the split of a node from its two counts (bvh_tree.rs:70-73)
__device__ __forceinline__ void axis_of(const Ptrs& a, int node, bool& on_x, int& split) {
  const int len = a.nlen[node], cx = a.ncx[node], cy = a.ncy[node];
  const int half = len / 2;
  const int hori = half > cx ? half - cx : cx - half;
  const int vert = half > cy ? half - cy : cy - half;
  on_x = vert > hori;
  split = on_x ? cx : cy;  // the predicate-true points come first (crate `partition`): split = their number
}

// element i of the node (thread-consecutive within the chunk so that a block scan gives ranks in position order)
__device__ __forceinline__ void misplaced(const Ptrs& a, int b, int len, int i, bool on_x, int split, double2 h, int& ml, int& mr) {
  ml = mr = 0;
  if (i < len) {
    const double2 p = a.P[b + i];
    const bool pred = on_x ? p.x > h.x : p.y > h.y;
    ml = (i < split) && !pred;
    mr = (i >= split) && pred;
  }
}

__global__ __launch_bounds__(256) void b64_mis(Ptrs a, int level) {
  __shared__ int slot[4];
  const int nch = a.chunkcount[level] < a.chunk_cap ? a.chunkcount[level] : a.chunk_cap;
  for (int ch = blockIdx.x; ch < nch; ch += gridDim.x) {
    const int node = a.ch_node[ch], ci = a.ch_index[ch];
    const int b = a.nbegin[node], len = a.nlen[node];
    bool on_x;
    int split;
    axis_of(a, node, on_x, split);
    const double2 h = a.nmean[node];
    int l = 0, r = 0;
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      int ml, mr;
      misplaced(a, b, len, ci * kChunk + (int)threadIdx.x * kEPT + j, on_x, split, h, ml, mr);
      l += ml;
      r += mr;
    }
    l = block_sum(l, slot);
    r = block_sum(r, slot);
    if (threadIdx.x == 0) { a.ch_l[ch] = l; a.ch_r[ch] = r; }
  }
}

// The two children of a node whose chunks' misplaced counts add up to runl / runr (bvh_tree.rs:78-88).
__device__ __forceinline__ void make_children64(const Ptrs& a, int level, int node, int leaf_size, int runl, int runr) {
  const int len = a.nlen[node], b = a.nbegin[node];
  bool on_x;
  int split;
  axis_of(a, node, on_x, split);
  a.naxis[node] = on_x ? 1 : 0;
  a.nsplit[node] = split;
  a.nk[node] = runl;  // == runr: as many misplaced on the left as on the right
  if (runl != runr) a.flags[kB64Fallback] = 1;
  const int first = atomicAdd(&a.flags[kB64NodeCount], 2);
  if (first + 2 > a.node_cap) {
    a.flags[kB64Fallback] = 1;
    a.nchild[node] = -1;
    return;
  }
  a.nchild[node] = first;
  const int depth = a.ndepth[node] + 1;
  atomicMax(&a.flags[kB64MaxDepth], depth);
  for (int side = 0; side < 2; ++side) {  // left = the "greater" side, built first
    const int id = first + side;
    const int cl = side == 0 ? split : len - split;
    a.nbegin[id] = side == 0 ? b : b + split;
    a.nlen[id] = cl;
    a.ndepth[id] = depth;
    a.nchild[id] = -1;
    const bool leaf = !(cl > leaf_size);
    a.nleaf[id] = leaf ? 1 : 0;
    if (!leaf) {
      const int slot = atomicAdd(&a.opencount[level + 1], 1);
      if (slot < a.open_cap) a.openq[(size_t)((level + 1) & 1) * a.open_cap + slot] = id;
      else a.flags[kB64Fallback] = 1;
    }
  }
}

// Prefix of the chunks' misplaced counts, then the children.  One wave per open node (the top levels: thousands of chunks) ...
__global__ __launch_bounds__(64) void b64_children(Ptrs a, int level, int leaf_size) {
  const int nopen = a.opencount[level] < a.open_cap ? a.opencount[level] : a.open_cap;
  const int lane = threadIdx.x;
  for (int q = blockIdx.x; q < nopen; q += gridDim.x) {
    const int node = a.openq[(size_t)(level & 1) * a.open_cap + q];
    const int len = a.nlen[node];
    const int nch = (len + kChunk - 1) / kChunk, c0 = a.nchunk0[node];
    int runl = 0, runr = 0;
    for (int cb = 0; cb < nch; cb += 64) {  // exclusive prefix over the node's chunks, 64 at a time
      const int c = cb + lane;
      const bool live = c < nch && c0 + c < a.chunk_cap;
      const int vl = live ? a.ch_l[c0 + c] : 0, vr = live ? a.ch_r[c0 + c] : 0;
      int il = vl, ir = vr;
      for (int d = 1; d < 64; d <<= 1) {
        const int ol = __shfl_up(il, d, 64), orr = __shfl_up(ir, d, 64);
        if (lane >= d) { il += ol; ir += orr; }
      }
      if (live) { a.ch_loff[c0 + c] = runl + il - vl; a.ch_roff[c0 + c] = runr + ir - vr; }
      runl += __shfl(il, 63, 64);
      runr += __shfl(ir, 63, 64);
    }
    if (lane == 0) make_children64(a, level, node, leaf_size, runl, runr);
  }
}
// ... or one thread per open node (everywhere else: a chunk or two each).
__global__ __launch_bounds__(256) void b64_children_t(Ptrs a, int level, int leaf_size) {
  const int nopen = a.opencount[level] < a.open_cap ? a.opencount[level] : a.open_cap;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nopen) return;
  const int node = a.openq[(size_t)(level & 1) * a.open_cap + q];
  const int len = a.nlen[node];
  const int nch = (len + kChunk - 1) / kChunk, c0 = a.nchunk0[node];
  int runl = 0, runr = 0;
  for (int c = 0; c < nch && c0 + c < a.chunk_cap; ++c) {
    a.ch_loff[c0 + c] = runl;
    a.ch_roff[c0 + c] = runr;
    runl += a.ch_l[c0 + c];
    runr += a.ch_r[c0 + c];
  }
  make_children64(a, level, node, leaf_size, runl, runr);
}

// the k-th misplaced point from the left / from the right of every node, in position order
__global__ __launch_bounds__(256) void b64_ranks(Ptrs a, int level) {
  __shared__ int wl[4], wr[4];
  const int nch = a.chunkcount[level] < a.chunk_cap ? a.chunkcount[level] : a.chunk_cap;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int ch = blockIdx.x; ch < nch; ch += gridDim.x) {
    const int node = a.ch_node[ch], ci = a.ch_index[ch];
    const int b = a.nbegin[node], len = a.nlen[node];
    bool on_x;
    int split;
    axis_of(a, node, on_x, split);
    const double2 h = a.nmean[node];
    int ml[kEPT], mr[kEPT], l = 0, r = 0;
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      misplaced(a, b, len, ci * kChunk + (int)threadIdx.x * kEPT + j, on_x, split, h, ml[j], mr[j]);
      l += ml[j];
      r += mr[j];
    }
    int il = l, ir = r;  // inclusive scan over the block's threads
    for (int d = 1; d < 64; d <<= 1) {
      const int ol = __shfl_up(il, d, 64), orr = __shfl_up(ir, d, 64);
      if (lane >= d) { il += ol; ir += orr; }
    }
    if (lane == 63) { wl[wave] = il; wr[wave] = ir; }
    __syncthreads();
    int bl = a.ch_loff[ch] + il - l, br = a.ch_roff[ch] + ir - r;
    for (int w = 0; w < wave; ++w) { bl += wl[w]; br += wr[w]; }
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      const int pos = b + ci * kChunk + (int)threadIdx.x * kEPT + j;
      if (ml[j]) a.lidx[b + bl++] = (uint32_t)pos;
      if (mr[j]) a.ridx[b + br++] = (uint32_t)pos;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void b64_swap(Ptrs a, int level) {
  const int nch = a.chunkcount[level] < a.chunk_cap ? a.chunkcount[level] : a.chunk_cap;
  for (int ch = blockIdx.x; ch < nch; ch += gridDim.x) {
    const int node = a.ch_node[ch], ci = a.ch_index[ch];
    const int b = a.nbegin[node], k_node = a.nk[node];
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
      const int k = ci * kChunk + j * 256 + (int)threadIdx.x;
      if (k < k_node) {
        // the left pointer meets its k-th misplaced point going up, the right pointer its k-th going DOWN (bvh_tree.rs:74-77,
        // crate `partition`): ridx is in ascending position order, so the partner is the k-th from its end
        const uint32_t pl = a.lidx[b + k], pr = a.ridx[b + (k_node - 1 - k)];
        const double2 tp = a.P[pl];
        a.P[pl] = a.P[pr];
        a.P[pr] = tp;
        const uint32_t ti = a.ID[pl];
        a.ID[pl] = a.ID[pr];
        a.ID[pr] = ti;
      }
    }
  }
}

// ---- the end -----------------------------------------------------------------------------------------------------------
// make_leaf (bvh_tree.rs:40-54) + the leaf arm of the upward pass (:98-131): box, unweighted mean in slice order, u32 mass
__global__ __launch_bounds__(256) void b64_leaves(Ptrs a, const uint32_t* __restrict__ weight) {
  const int m = a.flags[kB64NodeCount] < a.node_cap ? a.flags[kB64NodeCount] : a.node_cap;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= m) return;
  if (!a.nleaf[id]) {
    a.nsub[id] = 0;
    return;
  }
  const int b = a.nbegin[id], len = a.nlen[id];
  double mnx = kMaxD, mny = kMaxD, mxx = 0.0, mxy = 0.0, sx = 0.0, sy = 0.0;
  uint32_t ms = 0;
  for (int k = 0; k < len; ++k) {
    const double2 p = a.P[b + k];
    mnx = sse_min(mnx, p.x); mny = sse_min(mny, p.y);
    mxx = sse_max(mxx, p.x); mxy = sse_max(mxy, p.y);
    sx = sx + p.x; sy = sy + p.y;
    ms += weight[a.ID[b + k]];
  }
  a.nmin[id] = make_double2(mnx, mny);
  a.nmax[id] = make_double2(mxx, mxy);
  a.ncog[id] = make_double2(sx / (double)len, sy / (double)len);  // NaN for an empty leaf, as upstream
  a.nmass[id] = ms;
  a.nsub[id] = 1;
}

// internal nodes of one depth: centre of gravity as written (bvh_tree.rs:148-154), subtree size
__global__ __launch_bounds__(256) void b64_up(Ptrs a, int depth) {
  const int m = a.flags[kB64NodeCount] < a.node_cap ? a.flags[kB64NodeCount] : a.node_cap;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= m || a.nleaf[id] || a.ndepth[id] != depth) return;
  const int l = a.nchild[id];
  if (l < 0) return;  // never given children (buffers full / levels missing): the flags say so
  const int r = l + 1;
  const uint32_t m0 = a.nmass[l], m1 = a.nmass[r], ms = m0 + m1;
  const double2 c0 = a.ncog[l], c1 = a.ncog[r];
  const double bx = (c0.x * (double)m0) + (c1.x * (double)m1);
  const double by = (c0.y * (double)m0) + (c1.y * (double)m1);
  a.ncog[id] = make_double2(bx / (double)ms, by / (double)ms);
  a.nmass[id] = ms;
  a.nsub[id] = 1 + a.nsub[l] + a.nsub[r];
}

// pre-order numbers, top down: a node, its left subtree, its right subtree
__global__ __launch_bounds__(256) void b64_pre(Ptrs a, int depth) {
  const int m = a.flags[kB64NodeCount] < a.node_cap ? a.flags[kB64NodeCount] : a.node_cap;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= m || a.nleaf[id] || a.ndepth[id] != depth) return;
  const int l = a.nchild[id];
  if (l < 0) return;
  a.npre[l] = a.npre[id] + 1;
  a.npre[l + 1] = a.npre[id] + 1 + a.nsub[l];
}

__global__ __launch_bounds__(256) void b64_emit(Ptrs a, double4* __restrict__ geom0, double4* __restrict__ geom1, int4* __restrict__ link,
                                                int* __restrict__ depth_out, uint32_t* __restrict__ mass_out, double2* __restrict__ size_out,
                                                uint32_t* __restrict__ order_out) {
  const int m = a.flags[kB64NodeCount] < a.node_cap ? a.flags[kB64NodeCount] : a.node_cap;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id < a.n) order_out[id] = a.ID[id];
  if (id >= m) return;
  const int i = a.npre[id];
  if (i < 0 || i >= m) {  // a node nobody numbered: the build is incomplete
    a.flags[kB64Fallback] = 1;
    return;
  }
  const double2 mn = a.nmin[id], mx = a.nmax[id];
  const double w = mx.x - mn.x, h = mx.y - mn.y;  // boundary.size = max - min (bvh_tree.rs:63-66)
  geom0[i] = make_double4(mn.x, mn.y, mn.x + w, mn.y + h);
  const double tx = sse_max(w, h), ty = sse_max(h, w);  // size.max(size.yx()), main.rs:371
  const double2 cg = a.ncog[id];
  geom1[i] = make_double4(cg.x, cg.y, (double)a.nmass[id], tx * ty);
  link[i] = make_int4(i + a.nsub[id], a.nbegin[id], a.nlen[id], a.nleaf[id]);
  depth_out[i] = a.ndepth[id];
  mass_out[i] = a.nmass[id];
  size_out[i] = make_double2(w, h);
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

Bvh64Layout bvh64_layout(int64_t n, int leaf_size) {
  Bvh64Layout L{};
  const size_t N = (size_t)(n > 0 ? n : 1);
  const size_t lf = (size_t)(leaf_size > 0 ? leaf_size : 1);
  const size_t C = (4 * N / lf < 2 * N ? 4 * N / lf : 2 * N) + 4096;  // as bvh_build.hip: more nodes than this -> host builder
  const size_t OC = N / (lf + 1) + 2;                                   // open nodes of one level hold > leaf_size points each
  const size_t CC = N / (size_t)kChunk + OC + 2;                        // their chunks
  L.node_cap = (int)C;
  L.open_cap = (int)OC;
  L.chunk_cap = (int)CC;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return o; };
  L.flags = take(sizeof(int) * kB64FlagWords);
  L.opencount = take(sizeof(int) * (kB64Levels + 2));
  L.chunkcount = take(sizeof(int) * (kB64Levels + 2));
  L.segcount = take(sizeof(int) * (kB64Levels + 2));
  L.zero_end = off;
  const size_t SC = N / (size_t)kSeg + N / (size_t)kLongNode + 8;  // whole segments of the long nodes of one level
  L.seg_cap = (int)SC;
  L.nseg0 = take(4 * C);
  L.seg_node = take(4 * SC); L.seg_index = take(4 * SC);
  L.seg_sum = take(16 * SC); L.seg_min = take(16 * SC); L.seg_max = take(16 * SC); L.seg_prefix = take(16 * SC);
  L.seg_pred = take(16 * SC);
  L.seg_run = take(2 * sizeof(xsum64::Run) * SC);
  L.P = take(16 * N);
  L.ID = take(4 * N);
  L.nbegin = take(4 * C); L.nlen = take(4 * C); L.ndepth = take(4 * C); L.nchild = take(4 * C); L.nleaf = take(4 * C);
  L.ncx = take(4 * C); L.ncy = take(4 * C); L.nsplit = take(4 * C); L.naxis = take(4 * C); L.nk = take(4 * C);
  L.nchunk0 = take(4 * C); L.nsub = take(4 * C); L.npre = take(4 * C);
  L.nsum = take(16 * C); L.nmin = take(16 * C); L.nmax = take(16 * C); L.nmean = take(16 * C); L.ncog = take(16 * C);
  L.nmass = take(4 * C);
  L.openq = take(4 * 2 * OC);
  L.ch_node = take(4 * CC); L.ch_index = take(4 * CC); L.ch_l = take(4 * CC); L.ch_r = take(4 * CC);
  L.ch_loff = take(4 * CC); L.ch_roff = take(4 * CC);
  L.lidx = take(4 * N); L.ridx = take(4 * N);
  L.total = off;
  return L;
}

int bvh64_first_levels(int64_t n, int leaf_size) {
  int lv = 1;
  const int64_t lf = leaf_size > 0 ? leaf_size : 1;
  for (int64_t k = lf; k < n; k *= 2) ++lv;  // a balanced tree's levels with open nodes
  return lv;
}

hipError_t bvh64_begin(hipStream_t s, const void* pos, int n, char* scratch, const Bvh64Layout& L) {
  hipError_t e = hipMemsetAsync(scratch + L.flags, 0, L.zero_end - L.flags, s);  // flags + level counters
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(scratch + L.npre, 0xFF, 4 * (size_t)L.node_cap, s);        // "not numbered"
  if (e != hipSuccess) return e;
  Ptrs a = make_ptrs(scratch, L, n);
  b64_init<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a, (const double2*)pos);
  return hipGetLastError();
}

hipError_t bvh64_levels(hipStream_t s, int n, int leaf_size, int level_begin, int level_end, char* scratch, const Bvh64Layout& L) {
  Ptrs a = make_ptrs(scratch, L, n);
  for (int level = level_begin; level < level_end && level < kB64Levels; ++level) {
    const int64_t width = level < 30 ? (int64_t)1 << level : (int64_t)1 << 30;  // a level never has more open nodes than this
    const int64_t go = std::min<int64_t>(L.open_cap, width);
    const int64_t gc = std::min<int64_t>(L.chunk_cap, (int64_t)n / kChunk + width + 1);
    // the fold: a scan round covers 16 addends per thread — 512 threads (8192 addends) for the few long chains of the top
    // levels, 256 below; nodes of <= 256 points are added in order, a thread per node and coordinate
    const int64_t gbig = std::min<int64_t>(go, (int64_t)n / kSmallNode + 1);  // blocks stride over the queue: this many is plenty
    if (width <= kWideLevels && n > kLongNode) {  // long chains (they only occur up here): their segments' runs first, on the whole chip
      const int64_t gs = std::min<int64_t>(L.seg_cap, 4096);
      b64_seg_plan<<<dim3((unsigned)go), dim3(64), 0, s>>>(a, level);
      b64_seg_sums<<<dim3((unsigned)gs, 2), dim3(512), 0, s>>>(a, level);
      b64_seg_prefix<<<dim3((unsigned)((2 * go + 63) / 64)), dim3(64), 0, s>>>(a, level);
      b64_seg_runs<<<dim3((unsigned)gs, 2), dim3(512), 0, s>>>(a, level);
    }
    if (width <= kWideLevels) b64_fold<512, 16><<<dim3((unsigned)gbig, 2), dim3(512), 0, s>>>(a, level);
    else b64_fold<256, 16><<<dim3((unsigned)gbig, 2), dim3(256), 0, s>>>(a, level);
    b64_fold_small<<<dim3((unsigned)((2 * go + 255) / 256)), dim3(256), 0, s>>>(a, level);
    const bool by_wave = width <= 64;  // a wave per open node while nodes are few and long, a thread per node after
    if (by_wave) b64_plan<<<dim3((unsigned)go), dim3(64), 0, s>>>(a, level);
    else b64_plan_t<<<dim3((unsigned)((go + 255) / 256)), dim3(256), 0, s>>>(a, level);
    b64_count<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
    b64_mis<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
    if (by_wave) b64_children<<<dim3((unsigned)go), dim3(64), 0, s>>>(a, level, leaf_size);
    else b64_children_t<<<dim3((unsigned)((go + 255) / 256)), dim3(256), 0, s>>>(a, level, leaf_size);
    b64_ranks<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
    b64_swap<<<dim3((unsigned)gc), dim3(256), 0, s>>>(a, level);
  }
  return hipGetLastError();
}

hipError_t bvh64_finish(hipStream_t s, const uint32_t* weight, int n, int level_end, char* scratch, const Bvh64Layout& L, uint32_t* order_out,
                        void* geom0, void* geom1, void* link, int* depth_out, uint32_t* mass_out, void* size_out) {
  Ptrs a = make_ptrs(scratch, L, n);
  const dim3 gm((unsigned)((std::max(L.node_cap, n) + 255) / 256));
  const dim3 gn((unsigned)((L.node_cap + 255) / 256));
  b64_leaves<<<gn, dim3(256), 0, s>>>(a, weight);
  for (int d = level_end; d >= 0; --d) b64_up<<<gn, dim3(256), 0, s>>>(a, d);   // children (depth d + 1) before parents
  for (int d = 0; d <= level_end; ++d) b64_pre<<<gn, dim3(256), 0, s>>>(a, d);
  b64_emit<<<gm, dim3(256), 0, s>>>(a, (double4*)geom0, (double4*)geom1, (int4*)link, depth_out, mass_out, (double2*)size_out, order_out);
  return hipGetLastError();
}

}  // namespace nbody

// Barnes-Hut walk over a linearised (pre-order, skip-linked) BVH or quad tree, plus the gather and integrate
// kernels of a tree step.  gfx950, wave64.  Compiled with -ffp-contract=off: every operation is the one the
// reference writes, so a target's sum is bit-identical to the CPU recursion (same DFS order, same IEEE ops).
//
//   walk      World::bvh_sum_gravity            /root/reference src/main.rs:348-386
//   pair      calculate_gravity                 src/main.rs:234-253
//   integrate the Euler loop of World::update   src/main.rs:419-423
//
// The recursion "children[0] then children[1]" becomes a stackless pre-order scan: descending is i+1, leaving a
// subtree (leaf done, or node accepted) is link.skip.  No per-thread stack is needed at all.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "div_pair.h"
#include "env.h"
#include "tree_kernels.h"

namespace nbody {

template <class T> struct V2;
template <> struct V2<float> { using type = float2; };
template <> struct V2<double> { using type = double2; };
template <class T> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

template <class T> __device__ __forceinline__ bool is_normal_t(T v) { return __builtin_isnormal(v); }

template <class T>
__device__ __forceinline__ void pair_as_written(T px, T py, T qx, T qy, T force, T clamp, T& ax, T& ay) {
  T dx = qx - px;                                        // main.rs:236
  T dy = qy - py;
  T sum = __builtin_fabs(dx) + __builtin_fabs(dy);       // :238
  if (!is_normal_t(sum)) return;                         // :241-243
  T distance = dx * dx + dy * dy;                        // :245
  // :247-249 as one max (half the cost of compare + select): `distance` is never NaN here (a normal `sum` means finite
  // dx, dy), and for a NaN clamp both forms keep `distance`
  distance = __builtin_fmax(distance, clamp);
  T den = sum * distance;
  ax = ax + (dx * force) / den;                          // :252
  ay = ay + (dy * force) / den;
}
template <>
__device__ __forceinline__ void pair_as_written<float>(float px, float py, float qx, float qy, float force,
                                                       float clamp, float& ax, float& ay) {
  float dx = qx - px;
  float dy = qy - py;
  float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);
  if (!__builtin_isnormal(sum)) return;
  float distance = dx * dx + dy * dy;
  distance = __builtin_fmaxf(distance, clamp);
  float den = sum * distance;
  const float2 q = div_pair(dx * force, dy * force, den);  // the two quotients of :252, their multiply-adds packed (div_pair.h)
  ax = ax + q.x;
  ay = ay + q.y;
}

// FAST pair for the walk (opt-in, nbody_arith FAST): one reciprocal instead of two IEEE divisions, fused
// multiply-adds; a zero difference contributes exactly 0 through the biased denominator (direct_kernels.hip).  The
// node tests are untouched, so a target interacts with exactly the reference's list of nodes and particles; only
// the rounding of each term differs (tolerance of tests/_tol.py, not bit parity).
__device__ __forceinline__ void pair_fast(float px, float py, float qx, float qy, float force, float clamp, float& ax,
                                          float& ay) {
  float dx = qx - px, dy = qy - py;
  float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);
  float d2 = __builtin_fmaxf(__builtin_fmaf(dy, dy, dx * dx), clamp);
  float s = force * __builtin_amdgcn_rcpf(__builtin_fmaf(sum, d2, 8.0779356694631609e-28f));  // 2^-90
  ax = __builtin_fmaf(dx, s, ax);
  ay = __builtin_fmaf(dy, s, ay);
}
__device__ __forceinline__ void pair_fast(double px, double py, double qx, double qy, double force, double clamp,
                                          double& ax, double& ay) {
  double dx = qx - px, dy = qy - py;
  double sum = __builtin_fabs(dx) + __builtin_fabs(dy);
  double d2 = __builtin_fmax(__builtin_fma(dy, dy, dx * dx), clamp);
  double den = __builtin_fma(sum, d2, 0x1p-700);
  double r = __builtin_amdgcn_rcp(den);          // ~27 bits
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);   // Newton: ~54 bits
  double s = force * r;
  ax = __builtin_fma(dx, s, ax);
  ay = __builtin_fma(dy, s, ay);
}
template <class T, bool FAST>
__device__ __forceinline__ void walk_pair(T px, T py, T qx, T qy, T force, T clamp, T& ax, T& ay) {
  if constexpr (FAST) pair_fast(px, py, qx, qy, force, clamp, ax, ay);
  else pair_as_written<T>(px, py, qx, qy, force, clamp, ax, ay);
}

#ifdef NBODY_LAB  // the per-thread walk (NBODY_WALK_PER_THREAD): the baseline the wave-uniform walks are measured against
// One thread per target.  tgt_index (optional) maps thread t to the target's row: targets are visited in tree
// order so the lanes of a wave share most of their path, and results are scattered back to acc[row].
template <class T>
__global__ __launch_bounds__(256) void tree_walk(const WalkArgs<T> a) {
  using T2 = typename V2<T>::type;
  using T4 = typename V4<T>::type;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.n_tgt) return;
  const int64_t row = a.tgt_index ? (int64_t)a.tgt_index[t] : t;
  const T2 p = reinterpret_cast<const T2*>(a.tgt_pos)[row];
  const T4* __restrict__ g0 = reinterpret_cast<const T4*>(a.geom0);
  const T4* __restrict__ g1 = reinterpret_cast<const T4*>(a.geom1);
  const int4* __restrict__ lk = reinterpret_cast<const int4*>(a.link);
  const T2* __restrict__ lpos = reinterpret_cast<const T2*>(a.leaf_pos);
  const T* __restrict__ lmass = a.leaf_mass;
  const T theta = a.theta, clamp = a.clamp;
  T ax = 0, ay = 0;
  unsigned long long visits = 0, accepted = 0, leaf_pairs = 0;
  int i = 0;
  const int n_nodes = a.n_nodes;
  while (i < n_nodes) {
    const int4 l = lk[i];
    if (a.stats) visits++;
    if (l.w) {  // Leaf arm, main.rs:351-363: every particle of the slice, in slice order
      for (int k = l.y; k < l.y + l.z; ++k) {
        const T2 q = lpos[k];
        pair_as_written<T>(p.x, p.y, q.x, q.y, lmass[k], clamp, ax, ay);
      }
      if (a.stats) leaf_pairs += (unsigned long long)l.z;
      i = l.x;
      continue;
    }
    const T4 b = g0[i];  // lo.x lo.y hi.x hi.y
    const T4 c = g1[i];  // cog.x cog.y mass s2
    const bool contains = p.y > b.y && p.x > b.x && p.x < b.z && p.y < b.w;  // bvh_tree.rs:15-20 (all strict)
    const T ddx = p.x - c.x, ddy = p.y - c.y;                                  // dist2(p, cog), main.rs:228-232
    const T d2 = ddx * ddx + ddy * ddy;
    if (!contains && c.w < d2 * theta * theta) {                               // main.rs:370-372
      pair_as_written<T>(p.x, p.y, c.x, c.y, c.z, clamp, ax, ay);              // :374-379
      if (a.stats) accepted++;
      i = l.x;
    } else {
      i = i + 1;                                                               // :381-382
    }
  }
  reinterpret_cast<T2*>(a.acc)[row] = T2{ax, ay};
  if (a.stats) {
    atomicAdd(&a.stats[0], visits);
    atomicAdd(&a.stats[1], accepted);
    atomicAdd(&a.stats[2], leaf_pairs);
  }
}
#endif  // NBODY_LAB

// Wave-uniform walk: the 64 targets of a wave traverse the tree TOGETHER, in pre-order, with one node index held in
// an SGPR; each lane carries `resume`, the pre-order index at which it takes part again (a lane that accepted a node
// or finished a leaf sleeps until that subtree's `skip`).  A lane acts on node i iff resume <= i, so every lane sees
// exactly the nodes of its own depth-first walk, in the same order, with the same operations: bit-identical to
// tree_walk and to the CPU recursion.  What changes is who fetches the tree: node and leaf data are wave-uniform,
// so they arrive by scalar loads (no per-lane gathers, no divergent loop control), and the next node needs no
// reduction: sleeping lanes resume at or after skip[i] (their sleeping subtree contains node i), therefore
//     next = any(acting lane descends) ? i + 1 : skip[i].
// The wave visits the union of its lanes' paths; targets are handed out in tree order so that union stays small.
// STATS: the three counters of nbody_tree_walk_stats ride along (a template parameter: the counting costs every node step six
// instructions of ~60 whether or not anyone asked).
template <class T, int LB, bool PREFETCH, bool FAST, bool STATS>
__global__ __launch_bounds__(256) void tree_walk_wave(const WalkArgs<T> a) {
  using T2 = typename V2<T>::type;
  using T4 = typename V4<T>::type;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = t < a.n_tgt;
  const int64_t row = live ? (a.tgt_index ? (int64_t)a.tgt_index[t] : t) : 0;
  const T2 p = live ? reinterpret_cast<const T2*>(a.tgt_pos)[row] : T2{0, 0};
  const T4* __restrict__ g0 = reinterpret_cast<const T4*>(a.geom0);
  const T4* __restrict__ g1 = reinterpret_cast<const T4*>(a.geom1);
  const int4* __restrict__ lk = reinterpret_cast<const int4*>(a.link);
  const T theta = a.theta, clamp = a.clamp;
  const int n_nodes = a.n_nodes;
  T ax = 0, ay = 0;
  // FAST in f32: two-level summation (walk_split.hip, walk_tile_fast) — the terms go to a block sum that joins the running total
  // every 32nd leaf step, so that a long list (theta near 0: the direct sum in disguise) stays inside the 2e-5 contract.  The
  // as-written arithmetic keeps the reference's one chain.
  constexpr bool TWO = FAST && sizeof(T) == 4;
  T bx_ = 0, by_ = 0;
  T& bx = TWO ? bx_ : ax;
  T& by = TWO ? by_ : ay;
  int leaf_steps = 0;
  int resume = live ? 0 : n_nodes;
  unsigned long long visits = 0, accepted = 0, leaf_pairs = 0;
  if (n_nodes <= 0) {
    if (live) reinterpret_cast<T2*>(a.acc)[row] = T2{ax, ay};
    return;
  }
  // The walk is bound by the latency of its scalar loads, so a node's three records are fetched together, the
  // pre-order successor (the common next node) is fetched while the current node is processed, and leaf
  // particles come eight at a time (the arrays are padded, so reading past a leaf's end is safe; the extra
  // entries are not used).
  int i = 0;
  int4 l = lk[0];
  T4 b = g0[0], c = g1[0];
#ifdef NB_WALK_TIMING
  long long tw0 = wall_clock64(), t_leaf = 0, t_node = 0, n_leaf = 0, n_node = 0;
#endif
  while (i < n_nodes) {
#ifdef NB_WALK_TIMING
    const long long ts = wall_clock64();
    const bool was_leaf = l.w != 0;
#endif
    int4 ln = l;
    T4 bn = b, cn = c;
    if constexpr (PREFETCH) {  // speculative: node i+1
      const int ip = (i + 1 < n_nodes) ? i + 1 : i;
      ln = lk[ip];
      bn = g0[ip];
      cn = g1[ip];
    }
    const bool act = resume <= i;
    int next;
    if (l.w) {  // Leaf arm, main.rs:351-363
      const int end = l.y + l.z;
      for (int k0 = l.y; k0 < end; k0 += LB) {
        T2 q[LB];
        T m[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) {  // (wave-uniform: scalar loads)
          q[j] = scalar_leaf_pos<T>(a.leaf_pos, k0 + j);
          m[j] = scalar_leaf_mass<T>(a.leaf_mass, k0 + j);
        }
        if (act) {
#pragma unroll
          for (int j = 0; j < LB; ++j)
            if (k0 + j < end) walk_pair<T, FAST>(p.x, p.y, q[j].x, q[j].y, m[j], clamp, bx, by);
        }
      }
      if constexpr (TWO) {
        if (++leaf_steps == 32) {
          ax = ax + bx_;
          ay = ay + by_;
          bx_ = by_ = 0;
          leaf_steps = 0;
        }
      }
      if (act) {
        resume = l.x;
        if constexpr (STATS) { visits++; leaf_pairs += (unsigned long long)l.z; }
      }
      next = l.x;
    } else {
      // The node test as straight-line code — compares and-ed as masks, selects instead of `if (act) { if (...) {...} else {...} }`,
      // which compiles to nested exec regions with their saves, restores and branches: a third of a node step's instructions,
      // and every instruction of a wave costs an issue slot.  The term is under ONE wave-uniform branch.
      const bool contains = (p.y > b.y) & (p.x > b.x) & (p.x < b.z) & (p.y < b.w);  // bvh_tree.rs:15-20
      const T ddx = p.x - c.x, ddy = p.y - c.y;
      const T d2 = ddx * ddx + ddy * ddy;                                          // main.rs:228-232
      const bool accept = act & !contains & (c.w < d2 * theta * theta);              // :370-372
      const bool descend = act & !accept;                                            // :381-382
      if (__builtin_amdgcn_ballot_w64(accept) != 0) {
        T nax = bx, nay = by;
        walk_pair<T, FAST>(p.x, p.y, c.x, c.y, c.z, clamp, nax, nay);                // :374-379
        bx = accept ? nax : bx;
        by = accept ? nay : by;
      }
      resume = accept ? l.x : (descend ? i + 1 : resume);
      if constexpr (STATS) {
        accepted += accept ? 1u : 0u;
        visits += act ? 1u : 0u;
      }
      next = __builtin_amdgcn_ballot_w64(descend) != 0 ? i + 1 : l.x;
    }
    next = __builtin_amdgcn_readfirstlane(next);
    if (PREFETCH && next == i + 1) {
      l = ln; b = bn; c = cn;
    } else if (next < n_nodes) {
      // the three records by scalar loads (scalar_node_rec), TOGETHER: left alone the compiler sinks the box and the centre of gravity
      // into the node arm, behind the wait for the link — two dependent round trips per node step
      const NodeRec<T> r = scalar_node_rec<T>(a.link, a.geom0, a.geom1, next);
      asm volatile("" : : "s"(r.b.x), "s"(r.c.w));
      l = r.l; b = r.b; c = r.c;
    }
    i = next;
#ifdef NB_WALK_TIMING
    { const long long te = wall_clock64(); if (was_leaf) { t_leaf += te - ts; ++n_leaf; } else { t_node += te - ts; ++n_node; } }
#endif
  }
  if constexpr (TWO) {
    ax = ax + bx_;
    ay = ay + by_;
  }
  if (live) reinterpret_cast<T2*>(a.acc)[row] = T2{ax, ay};
#ifdef NB_WALK_TIMING
  if (a.stats && (threadIdx.x & 63) == 0) {  // timing build: [0] longest wave (10 ns ticks) << 20 | its leaf steps, ...
    const unsigned long long tot = (unsigned long long)(wall_clock64() - tw0);
    atomicMax(&a.stats[0], (tot << 40) | ((unsigned long long)t_leaf << 20) | (unsigned long long)t_node);
    atomicMax(&a.stats[1], (tot << 40) | ((unsigned long long)n_leaf << 20) | (unsigned long long)n_node);
    atomicAdd(&a.stats[2], tot);
  }
  return;
#endif
  if (STATS && a.stats && live) {
    atomicAdd(&a.stats[0], visits);
    atomicAdd(&a.stats[1], accepted);
    atomicAdd(&a.stats[2], leaf_pairs);
  }
}

// out[i] = in[perm[i]] for the particle arrays (the device-side image of the in-place partition permutation).
// calculate_gravity (main.rs:234-253) up to, but not including, the `+=`: a pair the reference skips is -0.0, the
// identity of IEEE addition.
template <class T> __device__ __forceinline__ typename V2<T>::type pair_term_t(T px, T py, T qx, T qy, T force, T clamp) {
  using T2 = typename V2<T>::type;
  const T dx = qx - px;
  const T dy = qy - py;
  const T sum = __builtin_fabs(dx) + __builtin_fabs(dy);
  if (!is_normal_t(sum)) return T2{(T)-0.0, (T)-0.0};
  T distance = dx * dx + dy * dy;
  distance = sizeof(T) == 8 ? (T)__builtin_fmax((double)distance, (double)clamp) : (T)__builtin_fmaxf((float)distance, (float)clamp);
  const T den = sum * distance;
  if constexpr (sizeof(T) == 4) {
    const float2 q = div_pair((float)(dx * force), (float)(dy * force), (float)den);
    return T2{(T)q.x, (T)q.y};
  }
  return T2{(dx * force) / den, (dy * force) / den};
}

// The wave-uniform walk for trees with SMALL leaves (the quad tree: at most 8 particles per leaf), as-written
// arithmetic.  In tree_walk_wave a leaf step costs one pair evaluation per particle for the whole wave, however few
// of its lanes take part: on config 4 (4 M bodies, theta 0.5) 21 of 64 lanes act at the average leaf step and the
// leaf holds 4.2 particles.  Here the (acting target, particle) PAIRS of a leaf step are dealt to the lanes: acting
// lanes put their positions into LDS by rank; lane L evaluates target rank L / m against particle L % m (m = the
// leaf's size, floor(64 / m) targets per round) and stores the term; then each target's own lane adds its m terms in
// slice order.  Same pairs, same operations, same order of additions per target as tree_walk_wave (a skipped pair is
// stored as -0.0), so the same bits.  Taken when it needs fewer rounds than the leaf has particles.
template <class T>
__global__ __launch_bounds__(256) void tree_walk_small(const WalkArgs<T> a, const int rule) {
  using T2 = typename V2<T>::type;
  using T4 = typename V4<T>::type;
  __shared__ T2 s_pos_all[4][64];
  __shared__ T2 s_term_all[4][64];
  T2* __restrict__ s_pos = s_pos_all[threadIdx.x >> 6];
  T2* __restrict__ s_term = s_term_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = t < a.n_tgt;
  const int64_t row = live ? (a.tgt_index ? (int64_t)a.tgt_index[t] : t) : 0;
  const T2 p = live ? reinterpret_cast<const T2*>(a.tgt_pos)[row] : T2{0, 0};
  const T2* __restrict__ lpos = reinterpret_cast<const T2*>(a.leaf_pos);
  const T* __restrict__ lmass = a.leaf_mass;
  const T theta = a.theta, clamp = a.clamp;
  const int n_nodes = a.n_nodes;
  T ax = 0, ay = 0;
  int resume = live ? 0 : n_nodes;
  int i = 0;
  while (i < n_nodes) {  // i is wave-uniform
    const NodeRec<T> rec = scalar_node_rec<T>(a.link, a.geom0, a.geom1, i);  // scalar loads, the three together (see tree_walk_wave)
    asm volatile("" : : "s"(rec.b.x), "s"(rec.c.w));
    const int4 l = rec.l;
    const T4 b = rec.b, c = rec.c;
    const bool act = resume <= i;
    int next;
    if (l.w) {  // Leaf arm, main.rs:351-363
      const int m = l.z;
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(act);
      const int takers = __builtin_popcountll(mask);
      const int tpr = m > 0 && m <= 64 ? 64 / m : 0;  // targets per round
      const int rounds = tpr ? (takers + tpr - 1) / tpr : 0;
      const bool pays = rule == 2 ? rounds < m : (rule == 3 ? rounds * 2 < m : rounds * 4 + 1 < m * 3);
      if (tpr && pays) {  // a round costs about a third more than a plain pair step
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (act) s_pos[rank] = p;
        // lane -> (slot, j) = (lane / m, lane % m); (lane + 0.5) / m is at least 1/16 away from an integer
        const int slot = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)m));
        const int j = lane - slot * m;
        T2 q = T2{0, 0};
        T qm = 0;
        if (slot < tpr) {
          q = lpos[l.y + j];
          qm = lmass[l.y + j];
        }
        wave_lds_handoff();
        for (int r = 0; r < rounds; ++r) {
          const int tr = r * tpr + slot;
          if (slot < tpr && tr < takers) {
            const T2 tp = s_pos[tr];
            s_term[lane] = pair_term_t<T>(tp.x, tp.y, q.x, q.y, qm, clamp);
          }
          wave_lds_handoff();
          const int mine = rank - r * tpr;
          if (act && mine >= 0 && mine < tpr) {  // this target's m terms, in slice order
            const T2* __restrict__ row_terms = s_term + mine * m;
            for (int jj = 0; jj < m; ++jj) {
              const T2 v = row_terms[jj];
              ax = ax + v.x;
              ay = ay + v.y;
            }
          }
          wave_lds_handoff();
        }
      } else {
        const int end = l.y + m;
        for (int k0 = l.y; k0 < end; k0 += 4) {
          T2 q[4];
          T qm[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {  // the arrays are padded: reading past a leaf's end is safe
            q[jj] = lpos[k0 + jj];
            qm[jj] = lmass[k0 + jj];
          }
          if (act) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              if (k0 + jj < end) pair_as_written<T>(p.x, p.y, q[jj].x, q[jj].y, qm[jj], clamp, ax, ay);
          }
        }
      }
      if (act) resume = l.x;
      next = l.x;
    } else {
      // straight-line node test (see tree_walk_wave): masks and selects, the term under one wave-uniform branch
      const bool contains = (p.y > b.y) & (p.x > b.x) & (p.x < b.z) & (p.y < b.w);  // bvh_tree.rs:15-20
      const T ddx = p.x - c.x, ddy = p.y - c.y;
      const T d2 = ddx * ddx + ddy * ddy;                                          // main.rs:228-232
      const bool accept = act & !contains & (c.w < d2 * theta * theta);              // :370-372
      const bool descend = act & !accept;                                            // :381-382
      if (__builtin_amdgcn_ballot_w64(accept) != 0) {
        T nax = ax, nay = ay;
        pair_as_written<T>(p.x, p.y, c.x, c.y, c.z, clamp, nax, nay);                // :374-379
        ax = accept ? nax : ax;
        ay = accept ? nay : ay;
      }
      resume = accept ? l.x : (descend ? i + 1 : resume);
      next = __builtin_amdgcn_ballot_w64(descend) != 0 ? i + 1 : l.x;
    }
    i = __builtin_amdgcn_readfirstlane(next);
  }
  if (live) reinterpret_cast<T2*>(a.acc)[row] = T2{ax, ay};
}

template <class T>
__global__ __launch_bounds__(256) void gather_particles(const GatherArgs<T> a) {
  gather_row<T>(a, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// main.rs:419-423: v += a*dt ; x += v*dt, multiply then add, in place.
template <class T>
__global__ __launch_bounds__(256) void integrate_inplace(void* pos, void* vel, const void* acc, int64_t n, T delta, const Gate gate) {
  using T2 = typename V2<T>::type;
  if ((gate.nonzero && *gate.nonzero == 0) || (gate.zero && *gate.zero != 0)) return;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < gate.carry_words) gate.carry_dst[i] = gate.carry_src[i];
  if (i == 0 && gate.stamp) *gate.stamp = (unsigned long long)wall_clock64();
  if (i >= n) return;
  T2 v = reinterpret_cast<T2*>(vel)[i];
  T2 p = reinterpret_cast<T2*>(pos)[i];
  const T2 ac = reinterpret_cast<const T2*>(acc)[i];
  v.x = v.x + ac.x * delta;
  v.y = v.y + ac.y * delta;
  const T vx = v.x * delta, vy = v.y * delta;
  p.x = p.x + vx;
  p.y = p.y + vy;
  reinterpret_cast<T2*>(vel)[i] = v;
  reinterpret_cast<T2*>(pos)[i] = p;
}

// ---- sharded tree steps (one process per GPU): integrate / export / import a set of rows given by an index list
template <class T>
__global__ __launch_bounds__(256) void integrate_rows(void* pos, void* vel, const void* acc, const uint32_t* rows,
                                                       int64_t row0, int64_t n, T delta) {
  using T2 = typename V2<T>::type;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int64_t i = rows ? (int64_t)rows[k] : row0 + k;
  T2 v = reinterpret_cast<T2*>(vel)[i];
  T2 p = reinterpret_cast<T2*>(pos)[i];
  const T2 ac = reinterpret_cast<const T2*>(acc)[i];
  v.x = v.x + ac.x * delta;  // main.rs:419-423
  v.y = v.y + ac.y * delta;
  const T vx = v.x * delta, vy = v.y * delta;
  p.x = p.x + vx;
  p.y = p.y + vy;
  reinterpret_cast<T2*>(vel)[i] = v;
  reinterpret_cast<T2*>(pos)[i] = p;
}
template <class T>
__global__ __launch_bounds__(256) void export_rows(const void* pos, const void* vel, const uint32_t* rows, int64_t row0, int64_t n,
                                                    uint32_t* rows_out, void* pos_out, void* vel_out) {
  using T2 = typename V2<T>::type;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int64_t i = rows ? (int64_t)rows[k] : row0 + k;
  rows_out[k] = (uint32_t)i;
  reinterpret_cast<T2*>(pos_out)[k] = reinterpret_cast<const T2*>(pos)[i];
  reinterpret_cast<T2*>(vel_out)[k] = reinterpret_cast<const T2*>(vel)[i];
}
template <class T>
__global__ __launch_bounds__(256) void import_rows(void* pos, void* vel, const uint32_t* rows, int64_t n, int64_t n_total,
                                                    const void* pos_in, const void* vel_in) {
  using T2 = typename V2<T>::type;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int64_t i = (int64_t)rows[k];
  if (i >= n_total) return;  // never write outside the state, whatever the caller sent
  reinterpret_cast<T2*>(pos)[i] = reinterpret_cast<const T2*>(pos_in)[k];
  reinterpret_cast<T2*>(vel)[i] = reinterpret_cast<const T2*>(vel_in)[k];
}
template <class T>
hipError_t launch_integrate_rows(hipStream_t s, void* pos, void* vel, const void* acc, const uint32_t* rows, int64_t row0, int64_t n, T delta) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL((integrate_rows<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, vel, acc, rows, row0, n, delta);
  return hipGetLastError();
}
template <class T>
hipError_t launch_export_rows(hipStream_t s, const void* pos, const void* vel, const uint32_t* rows, int64_t row0, int64_t n,
                              uint32_t* rows_out, void* pos_out, void* vel_out) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL((export_rows<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, vel, rows, row0, n, rows_out, pos_out, vel_out);
  return hipGetLastError();
}
template <class T>
hipError_t launch_import_rows(hipStream_t s, void* pos, void* vel, const uint32_t* rows, int64_t n, int64_t n_total, const void* pos_in,
                              const void* vel_in) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL((import_rows<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, vel, rows, n, n_total, pos_in, vel_in);
  return hipGetLastError();
}
#define NB_INST(T)                                                                                                        \
  template hipError_t launch_integrate_rows<T>(hipStream_t, void*, void*, const void*, const uint32_t*, int64_t, int64_t, T); \
  template hipError_t launch_export_rows<T>(hipStream_t, const void*, const void*, const uint32_t*, int64_t, int64_t, uint32_t*, void*, void*); \
  template hipError_t launch_import_rows<T>(hipStream_t, void*, void*, const uint32_t*, int64_t, int64_t, const void*, const void*);
NB_INST(float)
NB_INST(double)
#undef NB_INST

template <class T> hipError_t launch_tree_walk(hipStream_t s, const WalkArgs<T>& a, bool wave_uniform) {
  if (a.n_tgt <= 0) return hipSuccess;
  const dim3 grid((unsigned)((a.n_tgt + 255) / 256));
#ifdef NBODY_LAB
  if (!wave_uniform) {
    hipLaunchKernelGGL((tree_walk<T>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
  }
#else
  (void)wave_uniform;
#endif
  // small leaves (quad tree), as-written arithmetic: the pairs of a leaf step dealt to the lanes (tree_walk_small).
  // Laboratory: NBODY_WALK_COMPACT 0 off / 2, 3 other thresholds for dealing a leaf step's pairs to the lanes
  const int compact = lab_int("NBODY_WALK_COMPACT", 1);
  if (!a.big_leaves && !a.fast && !a.stats && a.n_nodes > 0 && compact != 0) {
    hipLaunchKernelGGL((tree_walk_small<T>), grid, dim3(256), 0, s, a, compact);
    return hipGetLastError();
  }
  // leaf batch / successor prefetch, measured in profiles/r01_walk_kernels_ab.txt: big leaves (BVH, 64) want 8
  // particles per fetch, small ones (quad, <= 8) 4; prefetching node i+1 never pays (the walk is bound by the
  // IEEE divides of the as-written pair function, not by scalar-load latency)
#define NB_WS(L, P, F) do { if (a.stats) hipLaunchKernelGGL((tree_walk_wave<T, L, P, F, true>), grid, dim3(256), 0, s, a); \
                           else hipLaunchKernelGGL((tree_walk_wave<T, L, P, F, false>), grid, dim3(256), 0, s, a); } while (0)
#define NB_W(L, P) do { if (a.fast) NB_WS(L, P, true); else NB_WS(L, P, false); } while (0)
#ifdef NBODY_LAB
  const int env_lb = lab_int("NBODY_WALK_LB", 0);
  const int env_pf = lab_int("NBODY_WALK_PREFETCH", -1);
  const int lb = env_lb ? env_lb : (a.big_leaves ? 8 : 4);
  const bool pf = env_pf >= 0 ? env_pf != 0 : false;
  if (lb >= 8) { if (pf) NB_W(8, true); else NB_W(8, false); }
  else if (lb >= 4) { if (pf) NB_W(4, true); else NB_W(4, false); }
  else { if (pf) NB_W(2, true); else NB_W(2, false); }
#else
  if (a.big_leaves) NB_W(8, false); else NB_W(4, false);
#endif
#undef NB_W
#undef NB_WS
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void div_pair_selftest(const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ den,
                                                         int64_t n, float* __restrict__ qx, float* __restrict__ qy) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float2 q = div_pair(nx[i], ny[i], den[i]);
  qx[i] = q.x;
  qy[i] = q.y;
}
hipError_t launch_div_pair_selftest(hipStream_t s, const float* nx, const float* ny, const float* den, int64_t n, float* qx, float* qy) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(div_pair_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nx, ny, den, n, qx, qy);
  return hipGetLastError();
}
__global__ void stamp_now(unsigned long long* __restrict__ stamp) { *stamp = (unsigned long long)wall_clock64(); }
hipError_t launch_stamp(hipStream_t s, unsigned long long* stamp) {
  hipLaunchKernelGGL(stamp_now, dim3(1), dim3(1), 0, s, stamp);
  return hipGetLastError();
}
template <class T> hipError_t launch_gather(hipStream_t s, const GatherArgs<T>& a) {
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL((gather_particles<T>), dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}
template <class T> hipError_t launch_integrate(hipStream_t s, void* pos, void* vel, const void* acc, int64_t n, T delta, Gate gate) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL((integrate_inplace<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pos, vel, acc, n, delta, gate);
  return hipGetLastError();
}

template hipError_t launch_tree_walk<float>(hipStream_t, const WalkArgs<float>&, bool);
template hipError_t launch_tree_walk<double>(hipStream_t, const WalkArgs<double>&, bool);
template hipError_t launch_gather<float>(hipStream_t, const GatherArgs<float>&);
template hipError_t launch_gather<double>(hipStream_t, const GatherArgs<double>&);
template hipError_t launch_integrate<float>(hipStream_t, void*, void*, const void*, int64_t, float, Gate);
template hipError_t launch_integrate<double>(hipStream_t, void*, void*, const void*, int64_t, double, Gate);

}  // namespace nbody

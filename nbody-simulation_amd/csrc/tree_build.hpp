// Host-side builders of the linearised trees the device walks.  Product code (not the oracle).
//
// BVH  — semantics of BVHTree::from + make_leaf + calculate_gravity, /root/reference src/bvh_tree.rs:40-158:
//        top-down, per node one sequential fold (min from f32::MAX, max from 0.0, sum), mean = sum/len,
//        axis = x iff |len/2 - #{y>mean.y}| > |len/2 - #{x>mean.x}|, predicate-true ("greater") side first,
//        two-pointer in-place partition (crate `partition` 0.1.2), sides of <= leaf_size points become leaves.
//        Built iteratively with an explicit stack straight into pre-order arrays (node i's first child is i+1).
// Quad — semantics of QuadTree::insert/subdivide/calculate_gravity, src/quad_tree.rs:153-270, for particles
//        inserted in index order.  The insertion result is order-determined only inside leaves: a cell is
//        internal iff it holds > 8 points, children exist iff non-empty, a leaf lists its points in ascending
//        particle index.  So the tree is built top-down with a stable 4-way split per cell, which yields the
//        identical cells, child codes (2*(y > ymid) + (x > xmid)), leaf orders and centre-of-gravity sums.
//
// Output layout (both kinds), pre-order:
//   geom0[i] = {lo.x, lo.y, hi.x, hi.y}   hi = offset + size, evaluated in T exactly as Rectangle::contains does
//   geom1[i] = {cog.x, cog.y, (T)total_mass, s2}   s2 = max(w,h)^2 (BVH, main.rs:371) or height^2 (quad)
//   link[i]  = {skip, leaf_first, leaf_count, is_leaf}
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <thread>
#include <utility>
#include <vector>
#include "env.h"

namespace nbody {

template <class T> struct TreeHost {
  struct G4 { T a, b, c, d; };
  struct L4 { int32_t skip, first, count, is_leaf; };
  std::vector<G4> geom0, geom1;
  std::vector<L4> link;
  // export-only (kept in the reference's own terms)
  std::vector<T> size_x, size_y;   // BVH: boundary.size; quad: height in size_x
  std::vector<uint32_t> mass_u32;
  std::vector<uint32_t> order;     // order[k] = index (into the arrays given to the builder) of the k-th
                                   // tree-ordered particle
  int kind = 0;
  int max_depth = 0;
  bool overflow = false;
  size_t size() const { return link.size(); }
  void clear() {
    geom0.clear(); geom1.clear(); link.clear(); size_x.clear(); size_y.clear(); mass_u32.clear(); order.clear();
    max_depth = 0; overflow = false;
  }
};

// SSE minps/maxps rule used by pathfinder_simd: NaN in either operand returns the second.
template <class T> static inline T sse_min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T sse_max(T a, T b) { return a > b ? a : b; }

constexpr int kMaxTreeDepth = 512;

// ---- parallel driver: a subtree is built into its own arrays (pre-order), big subtrees on their own threads,
// and the pieces are concatenated self | first child | ... | last child, which is exactly the pre-order the
// sequential recursion would have produced.  Every node still does the reference's sequential passes over its own
// slice, so the result is bit-identical whatever the thread count.
template <class T> struct SubTree {
  std::vector<typename TreeHost<T>::G4> geom0;
  std::vector<typename TreeHost<T>::L4> link;
  std::vector<T> size_x, size_y;
  int max_depth = 0;
  bool overflow = false;
  void push(T lox, T loy, T sx, T sy, int32_t first, int32_t count, int32_t is_leaf) {
    geom0.push_back({lox, loy, lox + sx, loy + sy});
    link.push_back({0, first, count, is_leaf});
    size_x.push_back(sx);
    size_y.push_back(sy);
  }
  void append(SubTree& o) {
    geom0.insert(geom0.end(), o.geom0.begin(), o.geom0.end());
    link.insert(link.end(), o.link.begin(), o.link.end());
    size_x.insert(size_x.end(), o.size_x.begin(), o.size_x.end());
    size_y.insert(size_y.end(), o.size_y.begin(), o.size_y.end());
    max_depth = std::max(max_depth, o.max_depth);
    overflow = overflow || o.overflow;
  }
};

// subtrees at least kParallelMinLen big get their own thread, down to depth kParallelMaxDepth
inline int64_t par_min_len() {
  static const int64_t v = lab_int("NBODY_BUILD_PAR_MINLEN", 8192);  // tuned on the GPU box, N = 151k
  return v;
}
inline int par_max_depth() {
  static const int v = lab_int("NBODY_BUILD_PAR_DEPTH", 4);
  return v;
}
#define kParallelMinLen par_min_len()
#define kParallelMaxDepth par_max_depth()

inline int build_threads() {
  if (const char* e = std::getenv("NBODY_BUILD_THREADS")) {
    int v = std::atoi(e);
    if (v >= 1) return std::min(v, 64);
  }
  unsigned hc = std::thread::hardware_concurrency();
  return (int)std::min<unsigned>(hc ? hc : 1, 16);
}

template <class T> struct BvhPoint { T x, y; uint32_t id; };

template <class T>
void bvh_rec(BvhPoint<T>* pts, int64_t first, int64_t len, int depth, bool leaf, int64_t leaf_size, bool parallel,
             SubTree<T>& out) {
  const T MAXV = std::numeric_limits<T>::max();
  using P = BvhPoint<T>;
  out.max_depth = std::max(out.max_depth, depth);
  P* p = pts + first;
  T mnx = MAXV, mny = MAXV, mxx = 0, mxy = 0, sx = 0, sy = 0;
  if (leaf) {  // make_leaf, bvh_tree.rs:40-54
    for (int64_t i = 0; i < len; ++i) {
      mnx = sse_min(mnx, p[i].x); mny = sse_min(mny, p[i].y);
      mxx = sse_max(mxx, p[i].x); mxy = sse_max(mxy, p[i].y);
    }
    out.push(mnx, mny, mxx - mnx, mxy - mny, (int32_t)first, (int32_t)len, 1);
    return;
  }
  for (int64_t i = 0; i < len; ++i) {  // the one sequential fold of bvh_tree.rs:58-61
    mnx = sse_min(mnx, p[i].x); mny = sse_min(mny, p[i].y);
    mxx = sse_max(mxx, p[i].x); mxy = sse_max(mxy, p[i].y);
    sx = sx + p[i].x; sy = sy + p[i].y;
  }
  const T hx = sx / (T)len, hy = sy / (T)len;  // :67
  int64_t cx = 0, cy = 0;
  for (int64_t i = 0; i < len; ++i) { cx += p[i].x > hx; cy += p[i].y > hy; }
  const int64_t half = len / 2;
  const int64_t hori = half > cx ? half - cx : cx - half;  // :71
  const int64_t vert = half > cy ? half - cy : cy - half;  // :72
  const bool on_x = vert > hori;                           // :73
  int64_t split = 0;
  if (len > 0) {  // crate `partition` 0.1.2: two pointers, predicate-true side first
    int64_t l = 0, r = len - 1;
    for (;;) {
      if (on_x) {
        while (l < len && p[l].x > hx) ++l;
        while (r > 0 && !(p[r].x > hx)) --r;
      } else {
        while (l < len && p[l].y > hy) ++l;
        while (r > 0 && !(p[r].y > hy)) --r;
      }
      if (l >= r) { split = l; break; }
      std::swap(p[l], p[r]);
    }
  }
  out.push(mnx, mny, mxx - mnx, mxy - mny, (int32_t)first, (int32_t)len, 0);
  const int64_t llen = split, rlen = len - split;
  bool lleaf = !(llen > leaf_size), rleaf = !(rlen > leaf_size);  // :78-88
  if (depth + 1 >= kMaxTreeDepth) {
    if (!lleaf || !rleaf) out.overflow = true;
    lleaf = rleaf = true;
  }
  if (parallel && depth < kParallelMaxDepth && std::min(llen, rlen) >= kParallelMinLen) {
    SubTree<T> left, right;
    std::thread th([&] { bvh_rec<T>(pts, first, llen, depth + 1, lleaf, leaf_size, true, left); });
    bvh_rec<T>(pts, first + split, rlen, depth + 1, rleaf, leaf_size, true, right);
    th.join();
    out.append(left);
    out.append(right);
  } else {
    bvh_rec<T>(pts, first, llen, depth + 1, lleaf, leaf_size, parallel, out);  // left = the "greater" side first
    bvh_rec<T>(pts, first + split, rlen, depth + 1, rleaf, leaf_size, parallel, out);
  }
}

template <class F> void parallel_chunks(int64_t n, int threads, F f) {
  // a std::thread costs tens of microseconds to start: only worth it for >= 64k elements per thread
  if (threads > n / 65536) threads = (int)(n / 65536);
  if (threads <= 1) { f(0, n); return; }
  std::vector<std::thread> th;
  int64_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; ++t) {
    int64_t b = t * per, e = std::min(n, b + per);
    if (b >= e) break;
    th.emplace_back([=] { f(b, e); });
  }
  for (auto& t : th) t.join();
}

// pos: interleaved xy of n particles (not modified); weight: u32 masses.
template <class T>
void build_bvh(const T* pos, const uint32_t* weight, int64_t n, int64_t leaf_size, TreeHost<T>& out) {
  out.clear();
  out.kind = 0;
  using P = BvhPoint<T>;
  const int threads = build_threads();
  std::vector<P> pts((size_t)n);
  parallel_chunks(n, threads, [&](int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) pts[(size_t)i] = {pos[2 * i], pos[2 * i + 1], (uint32_t)i};
  });
  SubTree<T> st;
  // the top call is unconditional: the root is always a Root (main.rs:400)
  bvh_rec<T>(pts.data(), 0, n, 0, false, leaf_size, threads > 1, st);
  out.geom0.swap(st.geom0);
  out.link.swap(st.link);
  out.size_x.swap(st.size_x);
  out.size_y.swap(st.size_y);
  out.max_depth = st.max_depth;
  out.overflow = st.overflow;
  const int64_t m = (int64_t)out.size();
  out.geom1.assign((size_t)m, {T(0), T(0), T(0), T(0)});
  out.mass_u32.assign((size_t)m, 0u);
  out.order.resize((size_t)n);
  parallel_chunks(n, threads, [&](int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) out.order[(size_t)i] = pts[(size_t)i].id;
  });
  // upward pass (bvh_tree.rs:98-158).  Leaves first (independent: unweighted mean in slice order, u32 wrapping mass)
  std::vector<T> cogx((size_t)m), cogy((size_t)m);
  parallel_chunks(m, threads, [&](int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) {
      const auto& lk = out.link[(size_t)i];
      if (!lk.is_leaf) continue;
      uint32_t ms = 0;
      T ax = 0, ay = 0;
      for (int32_t k = 0; k < lk.count; ++k) {
        const P& q = pts[(size_t)(lk.first + k)];
        ms += weight ? weight[q.id] : 1u;
        ax = ax + q.x; ay = ay + q.y;
      }
      out.mass_u32[(size_t)i] = ms;
      cogx[(size_t)i] = ax / (T)lk.count;  // NaN for an empty leaf, as upstream
      cogy[(size_t)i] = ay / (T)lk.count;
    }
  });
  // skip links + internal nodes, children before parents = descending pre-order index
  for (int64_t i = m - 1; i >= 0; --i) {
    auto& lk = out.link[(size_t)i];
    if (lk.is_leaf) {
      lk.skip = (int32_t)(i + 1);
    } else {
      const int64_t l = i + 1, r = out.link[(size_t)l].skip;
      lk.skip = out.link[(size_t)r].skip;
      const uint32_t m0 = out.mass_u32[(size_t)l], m1 = out.mass_u32[(size_t)r];
      const uint32_t ms = m0 + m1;
      const T bx = (cogx[(size_t)l] * (T)m0) + (cogx[(size_t)r] * (T)m1);
      const T by = (cogy[(size_t)l] * (T)m0) + (cogy[(size_t)r] * (T)m1);
      cogx[(size_t)i] = bx / (T)ms;
      cogy[(size_t)i] = by / (T)ms;
      out.mass_u32[(size_t)i] = ms;
    }
    const T w = out.size_x[(size_t)i], h = out.size_y[(size_t)i];
    const T tx = sse_max(w, h), ty = sse_max(h, w);  // size.max(size.yx())
    out.geom1[(size_t)i] = {cogx[(size_t)i], cogy[(size_t)i], (T)out.mass_u32[(size_t)i], tx * ty};
  }
}

template <class T>
void quad_rec(BvhPoint<T>* pts, BvhPoint<T>* tmp, int64_t first, int64_t len, T ox, T oy, T h, int depth,
              bool parallel, SubTree<T>& out) {
  out.max_depth = std::max(out.max_depth, depth);
  bool leaf = len <= 8;  // MAX_CAPACITY, quad_tree.rs:54
  if (!leaf && depth >= kMaxTreeDepth) { out.overflow = true; leaf = true; }
  out.push(ox, oy, h, h, (int32_t)first, (int32_t)len, leaf ? 1 : 0);
  if (leaf) return;
  const T half = h / (T)2.0;                      // quad_tree.rs:172
  const T xmid = ox + half, ymid = oy + half;     // :174-175
  int64_t cnt[4] = {0, 0, 0, 0};
  BvhPoint<T>* p = pts + first;
  for (int64_t i = 0; i < len; ++i) cnt[((p[i].y > ymid) ? 2 : 0) + ((p[i].x > xmid) ? 1 : 0)]++;  // :176-179
  const int64_t start[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
  int64_t fill[4] = {start[0], start[1], start[2], start[3]};
  BvhPoint<T>* sc = tmp + first;
  for (int64_t i = 0; i < len; ++i)  // stable: keeps ascending particle index inside each child
    sc[fill[((p[i].y > ymid) ? 2 : 0) + ((p[i].x > xmid) ? 1 : 0)]++] = p[i];
  std::copy(sc, sc + len, p);
  T cox[4], coy[4];
  for (int c = 0; c < 4; ++c) {  // child offsets, quad_tree.rs:182-188
    cox[c] = ox; coy[c] = oy;
    switch (c) {
      case 1: cox[c] = ox + half; coy[c] = oy + (T)0.0; break;
      case 2: cox[c] = ox + (T)0.0; coy[c] = oy + half; break;
      case 3: cox[c] = ox + half; coy[c] = oy + half; break;
      default: break;
    }
  }
  if (parallel && depth < kParallelMaxDepth && len >= 4 * kParallelMinLen) {
    SubTree<T> sub[4];
    std::vector<std::thread> th;
    for (int c = 0; c < 4; ++c) {
      if (!cnt[c]) continue;
      th.emplace_back([&, c] { quad_rec<T>(pts, tmp, first + start[c], cnt[c], cox[c], coy[c], half, depth + 1, true, sub[c]); });
    }
    for (auto& t : th) t.join();
    for (int c = 0; c < 4; ++c)
      if (cnt[c]) out.append(sub[c]);
  } else {
    for (int c = 0; c < 4; ++c)
      if (cnt[c]) quad_rec<T>(pts, tmp, first + start[c], cnt[c], cox[c], coy[c], half, depth + 1, parallel, out);
  }
}

template <class T>
void build_quad(const T* pos, const uint32_t* weight, int64_t n, T root_x, T root_y, T root_h, TreeHost<T>& out) {
  out.clear();
  out.kind = 1;
  const int threads = build_threads();
  // points travel with their coordinates (sequential passes instead of gathers through an index list)
  std::vector<BvhPoint<T>> pts((size_t)n), tmp((size_t)n);
  parallel_chunks(n, threads, [&](int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) pts[(size_t)i] = {pos[2 * i], pos[2 * i + 1], (uint32_t)i};
  });
  SubTree<T> st;
  quad_rec<T>(pts.data(), tmp.data(), 0, n, root_x, root_y, root_h, 0, threads > 1, st);
  std::vector<uint32_t> idx((size_t)n);
  parallel_chunks(n, threads, [&](int64_t b, int64_t e) {
    for (int64_t i = b; i < e; ++i) idx[(size_t)i] = pts[(size_t)i].id;
  });
  out.geom0.swap(st.geom0);
  out.link.swap(st.link);
  out.size_x.swap(st.size_x);
  out.size_y.swap(st.size_y);
  out.max_depth = st.max_depth;
  out.overflow = st.overflow;
  const int64_t m = (int64_t)out.size();
  out.geom1.assign((size_t)m, {T(0), T(0), T(0), T(0)});
  out.mass_u32.assign((size_t)m, 0u);
  out.order = idx;
  std::vector<T> cogx((size_t)m, T(0)), cogy((size_t)m, T(0));
  parallel_chunks(m, threads, [&](int64_t b, int64_t e) {  // leaves: quad_tree.rs:231-241
    for (int64_t i = b; i < e; ++i) {
      const auto& lk = out.link[(size_t)i];
      if (!lk.is_leaf) continue;
      uint32_t ms = 0;
      T ax = 0, ay = 0;
      for (int32_t k = 0; k < lk.count; ++k) {
        const BvhPoint<T>& q = pts[(size_t)(lk.first + k)];
        ms += weight ? weight[q.id] : 1u;
        ax = ax + q.x; ay = ay + q.y;
      }
      out.mass_u32[(size_t)i] = ms;
      if (lk.count > 0) { cogx[(size_t)i] = ax / (T)lk.count; cogy[(size_t)i] = ay / (T)lk.count; }
    }
  });
  for (int64_t i = m - 1; i >= 0; --i) {  // roots: quad_tree.rs:243-268, children in index order
    auto& lk = out.link[(size_t)i];
    if (lk.is_leaf) {
      lk.skip = (int32_t)(i + 1);
    } else {
      uint32_t ms = 0;
      T bx = 0, by = 0;
      int64_t c = i + 1, last = i, covered = 0;
      while (covered < lk.count) {
        ms += out.mass_u32[(size_t)c];
        bx = bx + (cogx[(size_t)c] * (T)out.mass_u32[(size_t)c]);
        by = by + (cogy[(size_t)c] * (T)out.mass_u32[(size_t)c]);
        covered += out.link[(size_t)c].count;
        last = c;
        c = out.link[(size_t)c].skip;
      }
      lk.skip = out.link[(size_t)last].skip;
      cogx[(size_t)i] = bx / (T)ms;
      cogy[(size_t)i] = by / (T)ms;
      out.mass_u32[(size_t)i] = ms;
    }
    out.geom1[(size_t)i] = {cogx[(size_t)i], cogy[(size_t)i], (T)out.mass_u32[(size_t)i], out.size_x[(size_t)i] * out.size_x[(size_t)i]};
  }
}

}  // namespace nbody

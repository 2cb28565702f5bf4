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
#include <limits>
#include <utility>
#include <vector>

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

// pos: interleaved xy of n particles (not modified); weight: u32 masses.
template <class T>
void build_bvh(const T* pos, const uint32_t* weight, int64_t n, int64_t leaf_size, TreeHost<T>& out) {
  out.clear();
  out.kind = 0;
  const T MAXV = std::numeric_limits<T>::max();
  struct P { T x, y; uint32_t id; };
  std::vector<P> pts((size_t)n);
  for (int64_t i = 0; i < n; ++i) pts[(size_t)i] = {pos[2 * i], pos[2 * i + 1], (uint32_t)i};

  struct Task { int64_t first, len; int depth; bool leaf; };
  std::vector<Task> stack;
  stack.push_back({0, n, 0, false});  // the top call is unconditional: the root is always a Root (main.rs:400)
  auto push_node = [&](T lox, T loy, T sx, T sy, int32_t first, int32_t count, int32_t is_leaf) {
    out.geom0.push_back({lox, loy, lox + sx, loy + sy});
    out.geom1.push_back({T(0), T(0), T(0), T(0)});
    out.link.push_back({0, first, count, is_leaf});
    out.size_x.push_back(sx);
    out.size_y.push_back(sy);
    out.mass_u32.push_back(0);
  };
  while (!stack.empty()) {
    Task t = stack.back();
    stack.pop_back();
    out.max_depth = std::max(out.max_depth, t.depth);
    P* p = pts.data() + t.first;
    T mnx = MAXV, mny = MAXV, mxx = 0, mxy = 0, sx = 0, sy = 0;
    if (t.leaf) {
      for (int64_t i = 0; i < t.len; ++i) {
        mnx = sse_min(mnx, p[i].x); mny = sse_min(mny, p[i].y);
        mxx = sse_max(mxx, p[i].x); mxy = sse_max(mxy, p[i].y);
      }
      push_node(mnx, mny, mxx - mnx, mxy - mny, (int32_t)t.first, (int32_t)t.len, 1);
      continue;
    }
    for (int64_t i = 0; i < t.len; ++i) {
      mnx = sse_min(mnx, p[i].x); mny = sse_min(mny, p[i].y);
      mxx = sse_max(mxx, p[i].x); mxy = sse_max(mxy, p[i].y);
      sx = sx + p[i].x; sy = sy + p[i].y;
    }
    const T hx = sx / (T)t.len, hy = sy / (T)t.len;
    int64_t cx = 0, cy = 0;
    for (int64_t i = 0; i < t.len; ++i) { cx += p[i].x > hx; cy += p[i].y > hy; }
    const int64_t half = t.len / 2;
    const int64_t hori = half > cx ? half - cx : cx - half;
    const int64_t vert = half > cy ? half - cy : cy - half;
    const bool on_x = vert > hori;
    // two-pointer partition, predicate-true side first
    int64_t split = 0;
    if (t.len > 0) {
      int64_t l = 0, r = t.len - 1;
      for (;;) {
        if (on_x) {
          while (l < t.len && p[l].x > hx) ++l;
          while (r > 0 && !(p[r].x > hx)) --r;
        } else {
          while (l < t.len && p[l].y > hy) ++l;
          while (r > 0 && !(p[r].y > hy)) --r;
        }
        if (l >= r) { split = l; break; }
        std::swap(p[l], p[r]);
      }
    }
    push_node(mnx, mny, mxx - mnx, mxy - mny, (int32_t)t.first, (int32_t)t.len, 0);
    const int64_t llen = split, rlen = t.len - split;
    bool lleaf = !(llen > leaf_size), rleaf = !(rlen > leaf_size);
    if (t.depth + 1 >= kMaxTreeDepth) {
      if (!lleaf || !rleaf) out.overflow = true;
      lleaf = rleaf = true;
    }
    // right is pushed first so the left ("greater") side is numbered first
    stack.push_back({t.first + split, rlen, t.depth + 1, rleaf});
    stack.push_back({t.first, llen, t.depth + 1, lleaf});
  }
  const int64_t m = (int64_t)out.size();
  out.order.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) out.order[(size_t)i] = pts[(size_t)i].id;
  // skip links + upward pass (bvh_tree.rs:98-158), children before parents = descending pre-order index
  std::vector<T> cogx((size_t)m), cogy((size_t)m);
  for (int64_t i = m - 1; i >= 0; --i) {
    auto& lk = out.link[(size_t)i];
    if (lk.is_leaf) {
      lk.skip = (int32_t)(i + 1);
      uint32_t ms = 0;
      T ax = 0, ay = 0;
      for (int32_t k = 0; k < lk.count; ++k) {
        const P& q = pts[(size_t)(lk.first + k)];
        ms += weight ? weight[q.id] : 1u;
        ax = ax + q.x; ay = ay + q.y;
      }
      out.mass_u32[(size_t)i] = ms;
      cogx[(size_t)i] = ax / (T)lk.count;  // unweighted mean; NaN for an empty leaf, as upstream
      cogy[(size_t)i] = ay / (T)lk.count;
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
void build_quad(const T* pos, const uint32_t* weight, int64_t n, T root_x, T root_y, T root_h, TreeHost<T>& out) {
  out.clear();
  out.kind = 1;
  std::vector<uint32_t> idx((size_t)n), tmp((size_t)n);
  for (int64_t i = 0; i < n; ++i) idx[(size_t)i] = (uint32_t)i;
  struct Task { int64_t first, len; T ox, oy, h; int depth; };
  std::vector<Task> stack;
  stack.push_back({0, n, root_x, root_y, root_h, 0});
  while (!stack.empty()) {
    Task t = stack.back();
    stack.pop_back();
    out.max_depth = std::max(out.max_depth, t.depth);
    bool leaf = t.len <= 8;
    if (!leaf && t.depth >= kMaxTreeDepth) { out.overflow = true; leaf = true; }
    out.geom0.push_back({t.ox, t.oy, t.ox + t.h, t.oy + t.h});
    out.geom1.push_back({T(0), T(0), T(0), t.h * t.h});
    out.link.push_back({0, (int32_t)t.first, (int32_t)t.len, leaf ? 1 : 0});
    out.size_x.push_back(t.h);
    out.size_y.push_back(t.h);
    out.mass_u32.push_back(0);
    if (leaf) continue;
    const T half = t.h / (T)2.0;
    const T xmid = t.ox + half, ymid = t.oy + half;
    int64_t cnt[4] = {0, 0, 0, 0};
    uint32_t* ids = idx.data() + t.first;
    for (int64_t i = 0; i < t.len; ++i) {
      const int c = ((pos[2 * (int64_t)ids[i] + 1] > ymid) ? 2 : 0) + ((pos[2 * (int64_t)ids[i]] > xmid) ? 1 : 0);
      cnt[c]++;
    }
    int64_t start[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
    int64_t fill[4] = {start[0], start[1], start[2], start[3]};
    uint32_t* sc = tmp.data() + t.first;
    for (int64_t i = 0; i < t.len; ++i) {  // stable: keeps ascending particle index inside each child
      const int c = ((pos[2 * (int64_t)ids[i] + 1] > ymid) ? 2 : 0) + ((pos[2 * (int64_t)ids[i]] > xmid) ? 1 : 0);
      sc[fill[c]++] = ids[i];
    }
    std::copy(sc, sc + t.len, ids);
    for (int c = 3; c >= 0; --c) {  // pushed in reverse so child 0 is numbered first
      if (!cnt[c]) continue;
      T ox = t.ox, oy = t.oy;
      switch (c) {
        case 1: ox = t.ox + half; oy = t.oy + (T)0.0; break;
        case 2: ox = t.ox + (T)0.0; oy = t.oy + half; break;
        case 3: ox = t.ox + half; oy = t.oy + half; break;
        default: break;
      }
      stack.push_back({t.first + start[c], cnt[c], ox, oy, half, t.depth + 1});
    }
  }
  const int64_t m = (int64_t)out.size();
  out.order = idx;
  std::vector<T> cogx((size_t)m, T(0)), cogy((size_t)m, T(0));
  for (int64_t i = m - 1; i >= 0; --i) {
    auto& lk = out.link[(size_t)i];
    if (lk.is_leaf) {
      lk.skip = (int32_t)(i + 1);
      uint32_t ms = 0;
      T ax = 0, ay = 0;
      for (int32_t k = 0; k < lk.count; ++k) {
        const int64_t id = idx[(size_t)(lk.first + k)];
        ms += weight ? weight[id] : 1u;
        ax = ax + pos[2 * id]; ay = ay + pos[2 * id + 1];
      }
      out.mass_u32[(size_t)i] = ms;
      if (lk.count > 0) { cogx[(size_t)i] = ax / (T)lk.count; cogy[(size_t)i] = ay / (T)lk.count; }
    } else {
      uint32_t ms = 0;
      T bx = 0, by = 0;
      int64_t c = i + 1, last = i;
      // children are the pre-order nodes i+1, skip[i+1], ... up to this node's own end (unknown yet: walk
      // until the particle range of the node is exhausted)
      int64_t covered = 0;
      while (covered < lk.count) {
        ms += out.mass_u32[(size_t)c];
        covered += out.link[(size_t)c].count;
        last = c;
        c = out.link[(size_t)c].skip;
      }
      c = i + 1;
      covered = 0;
      while (covered < lk.count) {
        bx = bx + (cogx[(size_t)c] * (T)out.mass_u32[(size_t)c]);
        by = by + (cogy[(size_t)c] * (T)out.mass_u32[(size_t)c]);
        covered += out.link[(size_t)c].count;
        c = out.link[(size_t)c].skip;
      }
      lk.skip = out.link[(size_t)last].skip;
      cogx[(size_t)i] = bx / (T)ms;
      cogy[(size_t)i] = by / (T)ms;
      out.mass_u32[(size_t)i] = ms;
    }
    out.geom1[(size_t)i].a = cogx[(size_t)i];
    out.geom1[(size_t)i].b = cogy[(size_t)i];
    out.geom1[(size_t)i].c = (T)out.mass_u32[(size_t)i];
  }
}

}  // namespace nbody

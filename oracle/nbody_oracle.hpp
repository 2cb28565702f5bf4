// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU restatement of the hot path of KristinnVikarJ/nbody-simulation (Rust), written from the
// behavioural description in SURVEY.md §8a.  It is the checker for the HIP path, never the product:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// "Parity unpinned": the reference has no tests, fixtures or golden vectors, and no Rust toolchain
// exists in this image, so this restatement could not be checked against the reference's own
// outputs.  It is pinned only by hand-derived known-answer tests (tests/test_oracle_kat.py) and by a second,
// independently written reading of the same reference text in Python (tests/_np_restatement.py: force law, BVH
// build + upward pass + walk, quad insert + upward pass + walk + empty / prune) that agrees with this file bit for bit.
//
// Third-party arithmetic that is NOT under /root/reference and is restated from the crates'
// published behaviour (Cargo.lock pins):
//   pathfinder_geometry 0.5.1 / pathfinder_simd 0.5.2 — Vector2F: component-wise IEEE f32 ops,
//     square_length = x*x + y*y, v*s and v/s with a splatted scalar (true division), min/max with
//     the SSE minps/maxps NaN rule (second operand returned when either is NaN).
//   partition 0.1.2 — partition(slice, pred): in-place two-pointer (Hoare style) partition,
//     predicate-true elements first, not stable.
//   rayon 1.7.0 — order-preserving parallel map/collect; no arithmetic.
//
// Compile with -ffp-contract=off and without -ffast-math: Rust never contracts a*b+c.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>
#include <limits>
#include <memory>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------- Vector2F (pathfinder_geometry)
template <class T> struct Vec2 {
  T x, y;
};
template <class T> inline Vec2<T> operator+(Vec2<T> a, Vec2<T> b) { return {a.x + b.x, a.y + b.y}; }
template <class T> inline Vec2<T> operator-(Vec2<T> a, Vec2<T> b) { return {a.x - b.x, a.y - b.y}; }
template <class T> inline Vec2<T> operator*(Vec2<T> a, T s) { return {a.x * s, a.y * s}; }
template <class T> inline Vec2<T> operator/(Vec2<T> a, T s) { return {a.x / s, a.y / s}; }
// minps / maxps: result = (a < b) ? a : b, so a NaN in either operand yields b.
template <class T> inline T sse_min(T a, T b) { return a < b ? a : b; }
template <class T> inline T sse_max(T a, T b) { return a > b ? a : b; }
template <class T> inline Vec2<T> vmin(Vec2<T> a, Vec2<T> b) { return {sse_min(a.x, b.x), sse_min(a.y, b.y)}; }
template <class T> inline Vec2<T> vmax(Vec2<T> a, Vec2<T> b) { return {sse_max(a.x, b.x), sse_max(a.y, b.y)}; }
template <class T> inline T square_length(Vec2<T> a) { return a.x * a.x + a.y * a.y; }

// src/main.rs:193-198.  AoS as in the reference; weight is an integer mass.
template <class T> struct Particle {
  Vec2<T> position;
  Vec2<T> velocity;
  uint32_t weight;
  uint32_t id;  // not in the reference: original index, carried so tests can read the permutation
};

// src/main.rs:228-232
template <class T> inline T dist2(Vec2<T> p1, Vec2<T> p2) { return square_length(p1 - p2); }

// ---------------------------------------------------------------- force law, src/main.rs:234-253
template <class T>
inline void calculate_gravity(Vec2<T> particle1, Vec2<T> particle2, Vec2<T>& accel, T force, T clamp) {
  Vec2<T> diff = particle2 - particle1;                    // :236
  T sum = std::fabs(diff.x) + std::fabs(diff.y);           // :238
  if (!std::isnormal(sum)) return;                         // :241-243 (0, subnormal, inf, NaN skip)
  T distance = square_length(diff);                        // :245
  if (distance < clamp) distance = clamp;                  // :247-249 (clamp = 0.001f32 upstream)
  accel = accel + (diff * force) / (sum * distance);       // :252
}

// ---------------------------------------------------------------- BVH, src/bvh_tree.rs
template <class T> struct Rect {  // bvh_tree.rs:8-21
  Vec2<T> offset, size;
  bool contains(Vec2<T> o) const {
    return o.y > offset.y && o.x > offset.x && o.x < (offset.x + size.x) && o.y < (offset.y + size.y);
  }
};

template <class T> struct BVHNode {  // bvh_tree.rs:23-35 (enum Root | Leaf)
  bool is_leaf;
  Rect<T> boundary;
  // Root
  uint32_t total_mass = 0;
  Vec2<T> center_of_gravity{0, 0};
  std::unique_ptr<BVHNode> children[2];
  // Leaf: slice [first, first+count) of the (permuted) particle array
  size_t first = 0, count = 0;
};

struct BuildLimits {
  int max_depth = 512;  // the reference recurses without bound on > leaf_size coincident points
  bool overflow = false;
};

// partition 0.1.2 (restated; see header).  Returns the split index.
template <class P, class Pred> inline size_t hoare_partition(P* data, size_t len, Pred pred) {
  if (len == 0) return 0;
  size_t l = 0, r = len - 1;
  for (;;) {
    while (l < len && pred(data[l])) ++l;
    while (r > 0 && !pred(data[r])) --r;
    if (l >= r) return l;
    std::swap(data[l], data[r]);
  }
}

template <class T>
std::unique_ptr<BVHNode<T>> bvh_make_leaf(const Particle<T>* base, size_t first, size_t count) {  // :40-54
  const T MAXV = std::numeric_limits<T>::max();
  Vec2<T> mn{MAXV, MAXV}, mx{0, 0};
  for (size_t i = 0; i < count; ++i) {
    mn = vmin(mn, base[first + i].position);
    mx = vmax(mx, base[first + i].position);
  }
  auto n = std::make_unique<BVHNode<T>>();
  n->is_leaf = true;
  n->boundary = {mn, mx - mn};
  n->first = first;
  n->count = count;
  return n;
}

template <class T>
std::unique_ptr<BVHNode<T>> bvh_from(Particle<T>* base, size_t first, size_t len, size_t leaf_size,
                                     BuildLimits& lim, int depth) {  // :56-96
  const T MAXV = std::numeric_limits<T>::max();
  Particle<T>* pts = base + first;
  Vec2<T> mn{MAXV, MAXV}, mx{0, 0}, sum{0, 0};
  for (size_t i = 0; i < len; ++i) {  // :58-61 one sequential fold
    mn = vmin(mn, pts[i].position);
    mx = vmax(mx, pts[i].position);
    sum = sum + pts[i].position;
  }
  Rect<T> bounds{mn, mx - mn};        // :63-66
  Vec2<T> halved = sum / (T)len;      // :67
  size_t half_len = len / 2;          // :70
  size_t cx = 0, cy = 0;
  for (size_t i = 0; i < len; ++i) {
    cx += pts[i].position.x > halved.x;
    cy += pts[i].position.y > halved.y;
  }
  size_t hori = half_len > cx ? half_len - cx : cx - half_len;  // :71 abs_diff
  size_t vert = half_len > cy ? half_len - cy : cy - half_len;  // :72
  size_t split;
  if (vert > hori)                                              // :73-77
    split = hoare_partition(pts, len, [&](const Particle<T>& p) { return p.position.x > halved.x; });
  else
    split = hoare_partition(pts, len, [&](const Particle<T>& p) { return p.position.y > halved.y; });

  auto node = std::make_unique<BVHNode<T>>();
  node->is_leaf = false;
  node->boundary = bounds;
  if (depth >= lim.max_depth) {  // not in the reference (it would overflow the stack)
    lim.overflow = true;
    node->children[0] = bvh_make_leaf(base, first, split);
    node->children[1] = bvh_make_leaf(base, first + split, len - split);
    return node;
  }
  size_t llen = split, rlen = len - split;
  node->children[0] = llen > leaf_size ? bvh_from(base, first, llen, leaf_size, lim, depth + 1)
                                       : bvh_make_leaf(base, first, llen);          // :78-82
  node->children[1] = rlen > leaf_size ? bvh_from(base, first + split, rlen, leaf_size, lim, depth + 1)
                                       : bvh_make_leaf(base, first + split, rlen);  // :84-88
  return node;  // cog (0,0), mass 0 placeholders :90-95
}

template <class T> Vec2<T> bvh_cog(const BVHNode<T>& n, const Particle<T>* base) {  // :98-116
  if (!n.is_leaf) return n.center_of_gravity;
  Vec2<T> acc{0, 0};
  for (size_t i = 0; i < n.count; ++i) acc = acc + base[n.first + i].position;
  return acc / (T)n.count;  // unweighted; NaN for an empty leaf
}
template <class T> uint32_t bvh_mass(const BVHNode<T>& n, const Particle<T>* base) {  // :118-131
  if (!n.is_leaf) return n.total_mass;
  uint32_t a = 0;
  for (size_t i = 0; i < n.count; ++i) a += base[n.first + i].weight;  // wraps like release Rust
  return a;
}
template <class T> void bvh_calculate_gravity(BVHNode<T>& n, const Particle<T>* base) {  // :133-158
  if (n.is_leaf) return;
  bvh_calculate_gravity(*n.children[0], base);
  bvh_calculate_gravity(*n.children[1], base);
  uint32_t m0 = bvh_mass(*n.children[0], base), m1 = bvh_mass(*n.children[1], base);
  uint32_t mass = m0 + m1;
  Vec2<T> big = (bvh_cog(*n.children[0], base) * (T)m0) + (bvh_cog(*n.children[1], base) * (T)m1);
  n.center_of_gravity = big / (T)mass;
  n.total_mass = mass;
}

struct WalkStats {
  uint64_t node_visits = 0, accepted = 0, leaf_pairs = 0;
};

// src/main.rs:348-386
template <class T>
void bvh_sum_gravity(Vec2<T> p, const BVHNode<T>& tree, const Particle<T>* base, Vec2<T>& accel, T theta,
                     T clamp, WalkStats* st) {
  if (st) st->node_visits++;
  if (tree.is_leaf) {
    for (size_t i = 0; i < tree.count; ++i)
      calculate_gravity(p, base[tree.first + i].position, accel, (T)base[tree.first + i].weight, clamp);
    if (st) st->leaf_pairs += tree.count;
    return;
  }
  const Rect<T>& b = tree.boundary;
  Vec2<T> tmp = vmax(b.size, Vec2<T>{b.size.y, b.size.x});  // size.max(size.yx())
  if (!b.contains(p) && tmp.x * tmp.y < dist2(p, tree.center_of_gravity) * theta * theta) {
    calculate_gravity(p, tree.center_of_gravity, accel, (T)tree.total_mass, clamp);
    if (st) st->accepted++;
  } else {
    bvh_sum_gravity(p, *tree.children[0], base, accel, theta, clamp, st);
    bvh_sum_gravity(p, *tree.children[1], base, accel, theta, clamp, st);
  }
}

// The interaction list of main.rs:348-386 as a visitor: on_term(source position, force) for every leaf particle and every
// accepted node, in the recursion's order.  Same tests as bvh_sum_gravity above; used by the tolerance reference
// (terms evaluated as written, accumulated in double, with sum |term|: what a FAST walk is measured against).
template <class T, class F>
void bvh_visit_terms(Vec2<T> p, const BVHNode<T>& tree, const Particle<T>* base, T theta, F&& on_term) {
  if (tree.is_leaf) {
    for (size_t i = 0; i < tree.count; ++i) on_term(base[tree.first + i].position, (T)base[tree.first + i].weight);
    return;
  }
  const Rect<T>& b = tree.boundary;
  Vec2<T> tmp = vmax(b.size, Vec2<T>{b.size.y, b.size.x});
  if (!b.contains(p) && tmp.x * tmp.y < dist2(p, tree.center_of_gravity) * theta * theta) {
    on_term(tree.center_of_gravity, (T)tree.total_mass);
  } else {
    bvh_visit_terms(p, *tree.children[0], base, theta, on_term);
    bvh_visit_terms(p, *tree.children[1], base, theta, on_term);
  }
}

// Pre-order flattening used by the tests to compare with the product's linearised tree.
template <class T> struct FlatNode {
  T off_x, off_y, size_x, size_y, cog_x, cog_y;
  uint32_t mass;
  int32_t is_leaf;
  int64_t first, count;  // leaf slice
  int64_t skip;          // index of the next node in DFS order after this subtree
};
template <class T>
void bvh_flatten(const BVHNode<T>& n, const Particle<T>* base, std::vector<FlatNode<T>>& out) {
  size_t me = out.size();
  out.push_back({});
  FlatNode<T> f{};
  f.off_x = n.boundary.offset.x; f.off_y = n.boundary.offset.y;
  f.size_x = n.boundary.size.x; f.size_y = n.boundary.size.y;
  Vec2<T> c = bvh_cog(n, base);
  f.cog_x = c.x; f.cog_y = c.y;
  f.mass = bvh_mass(n, base);
  f.is_leaf = n.is_leaf;
  f.first = (int64_t)n.first; f.count = (int64_t)n.count;
  if (!n.is_leaf) {
    bvh_flatten(*n.children[0], base, out);
    size_t right = out.size();
    bvh_flatten(*n.children[1], base, out);
    f.first = out[me + 1].first;  // an internal node reports the slice its subtree covers
    f.count = out[me + 1].count + out[right].count;
  }
  f.skip = (int64_t)out.size();
  out[me] = f;
}

// ---------------------------------------------------------------- Quad tree, src/quad_tree.rs (dead code upstream)
template <class T> struct QRect {  // quad_tree.rs:8-33
  Vec2<T> offset;
  T height, height2;
  bool contains(Vec2<T> o) const {
    return o.y > offset.y && o.x > offset.x && o.x < (offset.x + height) && o.y < (offset.y + height);
  }
};
template <class T> struct QPoint {  // SmallParticle + the weight quad_tree.rs:144 reads
  Vec2<T> position;
  uint32_t weight;
  uint32_t id;
};
template <class T> struct QuadNode {  // quad_tree.rs:35-51
  QRect<T> boundary;
  Vec2<T> center_of_gravity{0, 0};
  bool is_leaf = true;
  // Leaf
  uint8_t count = 0;
  QPoint<T> pts[8];
  // Root
  uint8_t flags = 0;
  uint32_t total_mass = 0;
  std::unique_ptr<QuadNode> children[4];
};
constexpr int QUAD_MAX_CAPACITY = 8;  // quad_tree.rs:54

template <class T> void quad_insert(QuadNode<T>& n, const QPoint<T>& pt, BuildLimits& lim, int depth);

template <class T> void quad_subdivide(QuadNode<T>& n, BuildLimits& lim, int depth) {  // :209-227
  QPoint<T> old[8];
  int cnt = n.count;
  for (int i = 0; i < cnt; ++i) old[i] = n.pts[i];
  n.is_leaf = false;
  n.flags = 0;
  n.total_mass = (uint32_t)cnt;  // placeholder :216
  n.count = 0;
  n.center_of_gravity = {0, 0};
  for (int i = 0; i < cnt; ++i) quad_insert(n, old[i], lim, depth);
}

template <class T> void quad_insert(QuadNode<T>& n, const QPoint<T>& pt, BuildLimits& lim, int depth) {  // :153-207
  if (n.is_leaf) {
    if (n.count == QUAD_MAX_CAPACITY) {
      if (depth >= lim.max_depth) {  // not in the reference (unbounded recursion there)
        lim.overflow = true;
        return;
      }
      quad_subdivide(n, lim, depth);
      quad_insert(n, pt, lim, depth);
    } else {
      n.pts[n.count] = pt;
      n.count++;
    }
    return;
  }
  T half_height = n.boundary.height / (T)2.0;        // :172
  T hori_half = n.boundary.offset.x + half_height;   // :174
  T vert_half = n.boundary.offset.y + half_height;   // :175
  bool north = pt.position.y > vert_half;            // :176
  bool west = pt.position.x > hori_half;             // :177
  int child = ((int)north << 1) + (int)west;         // :179
  if ((n.flags & (1 << child)) == 0) {               // :181-194
    Vec2<T> off = n.boundary.offset;
    switch (child) {
      case 1: off = {n.boundary.offset.x + half_height, n.boundary.offset.y + (T)0.0}; break;
      case 2: off = {n.boundary.offset.x + (T)0.0, n.boundary.offset.y + half_height}; break;
      case 3: off = {n.boundary.offset.x + half_height, n.boundary.offset.y + half_height}; break;
      default: break;
    }
    n.children[child] = std::make_unique<QuadNode<T>>();
    n.children[child]->boundary = {off, half_height, half_height * half_height};
    n.flags |= (uint8_t)(1 << child);
  }
  quad_insert(*n.children[child], pt, lim, depth + 1);
}

template <class T> uint32_t quad_mass(const QuadNode<T>& n) {  // :139-151
  if (!n.is_leaf) return n.total_mass;
  uint32_t a = 0;
  for (int i = 0; i < n.count; ++i) a += n.pts[i].weight;
  return a;
}
template <class T> void quad_calculate_gravity(QuadNode<T>& n) {  // :229-270
  if (n.is_leaf) {
    if (n.count > 0) {
      Vec2<T> acc{0, 0};
      for (int i = 0; i < n.count; ++i) acc = acc + n.pts[i].position;
      n.center_of_gravity = acc / (T)n.count;
    }
    return;
  }
  for (auto& c : n.children)
    if (c) quad_calculate_gravity(*c);
  uint32_t mass = 0;
  for (auto& c : n.children)
    if (c) mass += quad_mass(*c);
  Vec2<T> big{0, 0};
  for (auto& c : n.children)
    if (c) big = big + (c->center_of_gravity * (T)quad_mass(*c));
  n.center_of_gravity = big / (T)mass;
  n.total_mass = mass;
}

// quad_tree.rs:66-89: the tree keeps its cells and loses its points.  Returns the cells visited, the one it was called on
// included; a root whose mass is already 0 counts as one and is not entered (:77-79).  Centres of gravity are left as they were.
template <class T> uint32_t quad_empty(QuadNode<T>& n) {
  if (n.is_leaf) {
    n.count = 0;  // (children.iter_mut().for_each(|child| *child = None): the slots past `count` are never read)
    return 1;
  }
  if (n.total_mass == 0) return 1;
  n.total_mass = 0;
  uint32_t sum = 0;
  for (auto& c : n.children)
    if (c) sum += quad_empty(*c);
  return sum + 1;
}
// quad_tree.rs:94-137 ("Must be ran after calculate_gravity"): children that are empty leaves or roots without mass are
// dropped and their flag bit flipped; roots with mass are entered.  Returns the children dropped (a dropped root counts once,
// whatever hung below it).
template <class T> uint32_t quad_prune(QuadNode<T>& n) {
  uint32_t sum = 0;
  if (n.is_leaf) return sum;
  for (int i = 0; i < 4; ++i) {
    bool child_empty = false;
    if (n.children[i]) {
      QuadNode<T>& inner = *n.children[i];
      if (inner.is_leaf) {
        if (inner.count == 0) child_empty = true;
      } else if (inner.total_mass == 0) {
        child_empty = true;
      } else {
        sum += quad_prune(inner);
      }
    }
    if (child_empty) {
      n.children[i].reset();
      n.flags ^= (uint8_t)(1 << i);
      sum += 1;
    }
  }
  return sum;
}

// No walker exists upstream.  Defined here by analogy with main.rs:348-386:
// leaf -> every stored point in slot order; root -> accept iff !contains && height2 < d2*theta*theta,
// else children in index order 0..3.
template <class T>
void quad_sum_gravity(Vec2<T> p, const QuadNode<T>& tree, Vec2<T>& accel, T theta, T clamp, WalkStats* st) {
  if (st) st->node_visits++;
  if (tree.is_leaf) {
    for (int i = 0; i < tree.count; ++i)
      calculate_gravity(p, tree.pts[i].position, accel, (T)tree.pts[i].weight, clamp);
    if (st) st->leaf_pairs += tree.count;
    return;
  }
  if (!tree.boundary.contains(p) &&
      tree.boundary.height2 < dist2(p, tree.center_of_gravity) * theta * theta) {
    calculate_gravity(p, tree.center_of_gravity, accel, (T)tree.total_mass, clamp);
    if (st) st->accepted++;
  } else {
    for (auto& c : tree.children)
      if (c) quad_sum_gravity(p, *c, accel, theta, clamp, st);
  }
}

template <class T, class F>
void quad_visit_terms(Vec2<T> p, const QuadNode<T>& tree, T theta, F&& on_term) {  // quad_sum_gravity's interaction list
  if (tree.is_leaf) {
    for (int i = 0; i < tree.count; ++i) on_term(tree.pts[i].position, (T)tree.pts[i].weight);
    return;
  }
  if (!tree.boundary.contains(p) && tree.boundary.height2 < dist2(p, tree.center_of_gravity) * theta * theta) {
    on_term(tree.center_of_gravity, (T)tree.total_mass);
  } else {
    for (auto& c : tree.children)
      if (c) quad_visit_terms(p, *c, theta, on_term);
  }
}

template <class T> struct QuadFlatNode {
  T off_x, off_y, height, cog_x, cog_y;
  uint32_t mass;
  int32_t is_leaf;
  int32_t depth;
  uint32_t child_code;   // 0..3 code relative to the parent (0 for the root)
  uint64_t path;         // 2-bit child codes from the root, most recent in the low bits
  int64_t first, count;  // leaf: range in the leaf-ordered id list
  int64_t skip;
};
template <class T>
void quad_flatten(const QuadNode<T>& n, int depth, uint32_t code, uint64_t path,
                  std::vector<QuadFlatNode<T>>& out, std::vector<uint32_t>& order) {
  size_t me = out.size();
  out.push_back({});
  QuadFlatNode<T> f{};
  f.off_x = n.boundary.offset.x; f.off_y = n.boundary.offset.y; f.height = n.boundary.height;
  f.cog_x = n.center_of_gravity.x; f.cog_y = n.center_of_gravity.y;
  f.mass = quad_mass(n);
  f.is_leaf = n.is_leaf; f.depth = depth; f.child_code = code; f.path = path;
  f.first = (int64_t)order.size();
  if (n.is_leaf) {
    for (int i = 0; i < n.count; ++i) order.push_back(n.pts[i].id);
    f.count = n.count;
  } else {
    for (int c = 0; c < 4; ++c)
      if (n.children[c]) quad_flatten(*n.children[c], depth + 1, (uint32_t)c, (path << 2) | (uint64_t)c, out, order);
    f.count = (int64_t)order.size() - f.first;
  }
  f.skip = (int64_t)out.size();
  out[me] = f;
}

}  // namespace oracle

// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see nbody_oracle.hpp).
// extern "C" surface over the templated restatement, for ctypes (oracle/oracle.py).
// Nothing in the product (nbody-simulation_amd/) may link or load this file.
#include "nbody_oracle.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>

using namespace oracle;

namespace {

template <class F> void parallel_for(size_t n, int nthreads, size_t min_len, F f) {
  // static chunks; mirrors rayon's with_min_len(5000) split floor (main.rs:408)
  if (nthreads < 1) nthreads = 1;
  size_t chunks = std::max<size_t>(1, std::min<size_t>((size_t)nthreads, (n + min_len - 1) / std::max<size_t>(min_len, 1)));
  if (chunks == 1) {
    f(0, n);
    return;
  }
  std::vector<std::thread> th;
  size_t per = (n + chunks - 1) / chunks;
  for (size_t c = 0; c < chunks; ++c) {
    size_t b = c * per, e = std::min(n, b + per);
    if (b >= e) break;
    th.emplace_back([=] { f(b, e); });
  }
  for (auto& t : th) t.join();
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class T> std::vector<Particle<T>> to_aos(size_t n, const T* pos, const T* vel, const uint32_t* w) {
  std::vector<Particle<T>> p(n);
  for (size_t i = 0; i < n; ++i) {
    p[i].position = {pos[2 * i], pos[2 * i + 1]};
    p[i].velocity = vel ? Vec2<T>{vel[2 * i], vel[2 * i + 1]} : Vec2<T>{0, 0};
    p[i].weight = w ? w[i] : 1u;
    p[i].id = (uint32_t)i;
  }
  return p;
}
template <class T> void from_aos(const std::vector<Particle<T>>& p, T* pos, T* vel, uint32_t* w, uint32_t* id) {
  for (size_t i = 0; i < p.size(); ++i) {
    if (pos) { pos[2 * i] = p[i].position.x; pos[2 * i + 1] = p[i].position.y; }
    if (vel) { vel[2 * i] = p[i].velocity.x; vel[2 * i + 1] = p[i].velocity.y; }
    if (w) w[i] = p[i].weight;
    if (id) id[i] = p[i].id;
  }
}

// ------------------------------------------------------------------ direct sum (ours, = walker at theta 0)
// accum_mode 0: accumulate in T, ascending j (the definition).  1: every term evaluated in T exactly as
// main.rs:252 writes it, accumulated in double (tolerance reference).  norm_out (optional): sum_j |term_j|_1.
template <class T>
void direct_accel(size_t n_src, const T* pos, const uint32_t* w, size_t n_tgt, const int64_t* tgt_idx,
                  const T* tgt_pos, T clamp, int accum_mode, int nthreads, double* acc_out, double* norm_out) {
  parallel_for(n_tgt, nthreads, 64, [&](size_t b, size_t e) {
    for (size_t t = b; t < e; ++t) {
      Vec2<T> p = tgt_pos ? Vec2<T>{tgt_pos[2 * t], tgt_pos[2 * t + 1]}
                          : Vec2<T>{pos[2 * (tgt_idx ? tgt_idx[t] : (int64_t)t)], pos[2 * (tgt_idx ? tgt_idx[t] : (int64_t)t) + 1]};
      if (accum_mode == 0) {
        Vec2<T> a{0, 0};
        for (size_t j = 0; j < n_src; ++j)
          calculate_gravity(p, Vec2<T>{pos[2 * j], pos[2 * j + 1]}, a, (T)(w ? w[j] : 1u), clamp);
        acc_out[2 * t] = a.x;
        acc_out[2 * t + 1] = a.y;
        if (norm_out) norm_out[t] = 0;
      } else {
        double ax = 0, ay = 0, nrm = 0;
        for (size_t j = 0; j < n_src; ++j) {
          Vec2<T> term{0, 0};
          calculate_gravity(p, Vec2<T>{pos[2 * j], pos[2 * j + 1]}, term, (T)(w ? w[j] : 1u), clamp);
          ax += (double)term.x;
          ay += (double)term.y;
          nrm += std::fabs((double)term.x) + std::fabs((double)term.y);
        }
        acc_out[2 * t] = ax;
        acc_out[2 * t + 1] = ay;
        if (norm_out) norm_out[t] = nrm;
      }
    }
  });
}

template <class T> void integrate(std::vector<Particle<T>>& ps, const std::vector<Vec2<T>>& acc, T delta) {
  for (size_t i = 0; i < ps.size(); ++i) {  // main.rs:419-423, sequential
    ps[i].velocity = ps[i].velocity + acc[i] * delta;
    Vec2<T> velo = ps[i].velocity * delta;
    ps[i].position = ps[i].position + velo;
  }
}

template <class T>
int update_direct(size_t n, T* pos, T* vel, const uint32_t* w, T delta, T clamp, int nsteps, int nthreads,
                  double* counting) {
  auto ps = to_aos(n, pos, vel, w);
  std::vector<Vec2<T>> acc(n);
  for (int s = 0; s < nsteps; ++s) {
    double t0 = now_s();
    parallel_for(n, nthreads, 64, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        Vec2<T> a{0, 0};
        for (size_t j = 0; j < n; ++j) calculate_gravity(ps[i].position, ps[j].position, a, (T)ps[j].weight, clamp);
        acc[i] = a;
      }
    });
    double t1 = now_s();
    integrate(ps, acc, delta);
    double t2 = now_s();
    if (counting) { counting[1] += t1 - t0; counting[2] += t2 - t1; }
  }
  from_aos(ps, pos, vel, (uint32_t*)nullptr, (uint32_t*)nullptr);
  return 0;
}

// ------------------------------------------------------------------ World::update, main.rs:388-425
// mode 0 = as_written (reproduces F6: accel computed for snapshot order, applied by index to the permuted
// array); mode 1 = consistent (accel applied to the particle it was computed for).
template <class T>
int update_bvh(size_t n, T* pos, T* vel, uint32_t* w, uint32_t* id, T delta, T theta, T clamp, size_t leaf_size,
               int mode, int nsteps, int nthreads, double* counting) {
  auto ps = to_aos(n, pos, vel, w);
  if (id) for (size_t i = 0; i < n; ++i) ps[i].id = id[i];
  std::vector<Vec2<T>> acc(n);
  for (int s = 0; s < nsteps; ++s) {
    double t0 = now_s();
    std::vector<Particle<T>> cloned = ps;                          // :398
    BuildLimits lim;
    auto tree = bvh_from(ps.data(), 0, n, leaf_size, lim, 0);      // :400 (permutes ps)
    if (lim.overflow) return -2;
    bvh_calculate_gravity(*tree, ps.data());                       // :401
    double t1 = now_s();
    const std::vector<Particle<T>>& targets = (mode == 0) ? cloned : ps;
    parallel_for(n, nthreads, 5000, [&](size_t b, size_t e) {      // :406-416
      for (size_t i = b; i < e; ++i) {
        Vec2<T> a{0, 0};
        bvh_sum_gravity(targets[i].position, *tree, ps.data(), a, theta, clamp, (WalkStats*)nullptr);
        acc[i] = a;
      }
    });
    double t2 = now_s();
    tree.reset();
    integrate(ps, acc, delta);                                     // :419-423
    double t3 = now_s();
    if (counting) { counting[0] += t1 - t0; counting[1] += t2 - t1; counting[2] += t3 - t2; }
  }
  from_aos(ps, pos, vel, w, id);
  return 0;
}

template <class T>
std::unique_ptr<QuadNode<T>> quad_build(const std::vector<Particle<T>>& ps, T root_x, T root_y, T root_h,
                                        BuildLimits& lim) {
  auto root = std::make_unique<QuadNode<T>>();
  root->boundary = {{root_x, root_y}, root_h, root_h * root_h};  // Rectangle::new quad_tree.rs:15-21
  for (size_t i = 0; i < ps.size(); ++i) {
    QPoint<T> q{ps[i].position, ps[i].weight, ps[i].id};
    quad_insert(*root, q, lim, 0);
  }
  return root;
}

// Quad-tree step: the same driver shape as World::update with the quad tree in place of the BVH.  The quad
// build does not permute the particle array, so as_written and consistent coincide.
template <class T>
int update_quad(size_t n, T* pos, T* vel, const uint32_t* w, T delta, T theta, T clamp, T root_x, T root_y,
                T root_h, int nsteps, int nthreads, double* counting) {
  auto ps = to_aos(n, pos, vel, w);
  std::vector<Vec2<T>> acc(n);
  for (int s = 0; s < nsteps; ++s) {
    double t0 = now_s();
    BuildLimits lim;
    auto tree = quad_build(ps, root_x, root_y, root_h, lim);
    if (lim.overflow) return -2;
    quad_calculate_gravity(*tree);
    double t1 = now_s();
    parallel_for(n, nthreads, 5000, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        Vec2<T> a{0, 0};
        quad_sum_gravity(ps[i].position, *tree, a, theta, clamp, (WalkStats*)nullptr);
        acc[i] = a;
      }
    });
    double t2 = now_s();
    tree.reset();
    integrate(ps, acc, delta);
    double t3 = now_s();
    if (counting) { counting[0] += t1 - t0; counting[1] += t2 - t1; counting[2] += t3 - t2; }
  }
  from_aos(ps, pos, vel, (uint32_t*)nullptr, (uint32_t*)nullptr);
  return 0;
}

// ------------------------------------------------------------------ handle-based tree access for the tests
template <class T> struct BvhHandle {
  std::vector<Particle<T>> ps;
  std::unique_ptr<BVHNode<T>> tree;
  std::vector<FlatNode<T>> flat;
  bool overflow = false;
};
template <class T> struct QuadHandle {
  std::vector<Particle<T>> ps;
  std::unique_ptr<QuadNode<T>> tree;
  std::vector<QuadFlatNode<T>> flat;
  std::vector<uint32_t> order;
  bool overflow = false;
};

template <class T> BvhHandle<T>* bvh_create(size_t n, const T* pos, const uint32_t* w, size_t leaf_size) {
  auto* h = new BvhHandle<T>();
  h->ps = to_aos<T>(n, pos, nullptr, w);
  BuildLimits lim;
  h->tree = bvh_from(h->ps.data(), 0, n, leaf_size, lim, 0);
  h->overflow = lim.overflow;
  bvh_calculate_gravity(*h->tree, h->ps.data());
  bvh_flatten(*h->tree, h->ps.data(), h->flat);
  return h;
}
template <class T>
void bvh_walk(BvhHandle<T>* h, size_t n_tgt, const T* tgt, T theta, T clamp, int nthreads, T* acc, uint64_t* stats) {
  if (stats) {
    // sequential when statistics are wanted (exact counts, no contention)
    WalkStats total;
    for (size_t i = 0; i < n_tgt; ++i) {
      Vec2<T> a{0, 0};
      bvh_sum_gravity(Vec2<T>{tgt[2 * i], tgt[2 * i + 1]}, *h->tree, h->ps.data(), a, theta, clamp, &total);
      acc[2 * i] = a.x; acc[2 * i + 1] = a.y;
    }
    stats[0] = total.node_visits; stats[1] = total.accepted; stats[2] = total.leaf_pairs;
    return;
  }
  parallel_for(n_tgt, nthreads, 256, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      Vec2<T> a{0, 0};
      bvh_sum_gravity(Vec2<T>{tgt[2 * i], tgt[2 * i + 1]}, *h->tree, h->ps.data(), a, theta, clamp, (WalkStats*)nullptr);
      acc[2 * i] = a.x; acc[2 * i + 1] = a.y;
    }
  });
}

// Tolerance reference of a walk: the same interaction list, every term evaluated in T exactly as main.rs:252 writes it,
// accumulated in double; norm = sum |term|_1 (the scale of tests/_tol.py).
template <class T>
void bvh_walk_ref(BvhHandle<T>* h, size_t n_tgt, const T* tgt, T theta, T clamp, int nthreads, double* acc, double* norm) {
  parallel_for(n_tgt, nthreads, 256, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const Vec2<T> p{tgt[2 * i], tgt[2 * i + 1]};
      double ax = 0, ay = 0, nrm = 0;
      bvh_visit_terms(p, *h->tree, h->ps.data(), theta, [&](Vec2<T> q, T force) {
        Vec2<T> term{0, 0};
        calculate_gravity(p, q, term, force, clamp);
        ax += (double)term.x; ay += (double)term.y;
        nrm += std::fabs((double)term.x) + std::fabs((double)term.y);
      });
      acc[2 * i] = ax; acc[2 * i + 1] = ay; norm[i] = nrm;
    }
  });
}

template <class T>
QuadHandle<T>* quad_create(size_t n, const T* pos, const uint32_t* w, T rx, T ry, T rh) {
  auto* h = new QuadHandle<T>();
  h->ps = to_aos<T>(n, pos, nullptr, w);
  BuildLimits lim;
  h->tree = quad_build(h->ps, rx, ry, rh, lim);
  h->overflow = lim.overflow;
  quad_calculate_gravity(*h->tree);
  quad_flatten(*h->tree, 0, 0u, 0ull, h->flat, h->order);
  return h;
}
// The life of a tree that is kept from step to step, which is what empty() and prune() are for (quad_tree.rs:66-137; nothing
// upstream calls them): empty it, insert the points where they are now, the upward pass, prune.  counts[0] = empty()'s return,
// counts[1] = prune()'s.  The cells of such a tree are NOT a fresh build's: a root stays a root however few points it still
// holds (the TODO at :91-93).
template <class T>
void quad_reuse(QuadHandle<T>* h, size_t n, const T* pos, const uint32_t* w, uint32_t* counts) {
  h->ps = to_aos<T>(n, pos, nullptr, w);
  const uint32_t emptied = quad_empty(*h->tree);
  BuildLimits lim;
  for (size_t i = 0; i < h->ps.size(); ++i) {
    QPoint<T> q{h->ps[i].position, h->ps[i].weight, h->ps[i].id};
    quad_insert(*h->tree, q, lim, 0);
  }
  h->overflow = lim.overflow;
  quad_calculate_gravity(*h->tree);
  const uint32_t pruned = quad_prune(*h->tree);
  h->flat.clear();
  h->order.clear();
  quad_flatten(*h->tree, 0, 0u, 0ull, h->flat, h->order);
  if (counts) { counts[0] = emptied; counts[1] = pruned; }
}
template <class T>
void quad_walk(QuadHandle<T>* h, size_t n_tgt, const T* tgt, T theta, T clamp, int nthreads, T* acc, uint64_t* stats) {
  if (stats) {
    WalkStats total;
    for (size_t i = 0; i < n_tgt; ++i) {
      Vec2<T> a{0, 0};
      quad_sum_gravity(Vec2<T>{tgt[2 * i], tgt[2 * i + 1]}, *h->tree, a, theta, clamp, &total);
      acc[2 * i] = a.x; acc[2 * i + 1] = a.y;
    }
    stats[0] = total.node_visits; stats[1] = total.accepted; stats[2] = total.leaf_pairs;
    return;
  }
  parallel_for(n_tgt, nthreads, 256, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      Vec2<T> a{0, 0};
      quad_sum_gravity(Vec2<T>{tgt[2 * i], tgt[2 * i + 1]}, *h->tree, a, theta, clamp, (WalkStats*)nullptr);
      acc[2 * i] = a.x; acc[2 * i + 1] = a.y;
    }
  });
}

template <class T>
void quad_walk_ref(QuadHandle<T>* h, size_t n_tgt, const T* tgt, T theta, T clamp, int nthreads, double* acc, double* norm) {
  parallel_for(n_tgt, nthreads, 256, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const Vec2<T> p{tgt[2 * i], tgt[2 * i + 1]};
      double ax = 0, ay = 0, nrm = 0;
      quad_visit_terms(p, *h->tree, theta, [&](Vec2<T> q, T force) {
        Vec2<T> term{0, 0};
        calculate_gravity(p, q, term, force, clamp);
        ax += (double)term.x; ay += (double)term.y;
        nrm += std::fabs((double)term.x) + std::fabs((double)term.y);
      });
      acc[2 * i] = ax; acc[2 * i + 1] = ay; norm[i] = nrm;
    }
  });
}

}  // namespace

#define ORC_API extern "C" __attribute__((visibility("default")))

ORC_API int orc_abi_version() { return 1; }

// single pair, accumulating into acc[2] (KATs)
ORC_API void orc_pair_f32(float p1x, float p1y, float p2x, float p2y, float force, float clamp, float* acc) {
  Vec2<float> a{acc[0], acc[1]};
  calculate_gravity(Vec2<float>{p1x, p1y}, Vec2<float>{p2x, p2y}, a, force, clamp);
  acc[0] = a.x; acc[1] = a.y;
}
ORC_API void orc_pair_f64(double p1x, double p1y, double p2x, double p2y, double force, double clamp, double* acc) {
  Vec2<double> a{acc[0], acc[1]};
  calculate_gravity(Vec2<double>{p1x, p1y}, Vec2<double>{p2x, p2y}, a, force, clamp);
  acc[0] = a.x; acc[1] = a.y;
}

#define ORC_INSTANTIATE(SFX, T)                                                                                     \
  ORC_API void orc_direct_accel_##SFX(int64_t n_src, const T* pos, const uint32_t* w, int64_t n_tgt,                \
                                      const int64_t* tgt_idx, const T* tgt_pos, T clamp, int accum_mode,            \
                                      int nthreads, double* acc_out, double* norm_out) {                            \
    direct_accel<T>((size_t)n_src, pos, w, (size_t)n_tgt, tgt_idx, tgt_pos, clamp, accum_mode, nthreads, acc_out,   \
                    norm_out);                                                                                      \
  }                                                                                                                 \
  ORC_API int orc_update_direct_##SFX(int64_t n, T* pos, T* vel, const uint32_t* w, T delta, T clamp, int nsteps,   \
                                      int nthreads, double* counting) {                                             \
    return update_direct<T>((size_t)n, pos, vel, w, delta, clamp, nsteps, nthreads, counting);                      \
  }                                                                                                                 \
  ORC_API int orc_update_bvh_##SFX(int64_t n, T* pos, T* vel, uint32_t* w, uint32_t* id, T delta, T theta, T clamp, \
                                   int64_t leaf_size, int mode, int nsteps, int nthreads, double* counting) {       \
    return update_bvh<T>((size_t)n, pos, vel, w, id, delta, theta, clamp, (size_t)leaf_size, mode, nsteps,          \
                         nthreads, counting);                                                                       \
  }                                                                                                                 \
  ORC_API int orc_update_quad_##SFX(int64_t n, T* pos, T* vel, const uint32_t* w, T delta, T theta, T clamp,        \
                                    T rx, T ry, T rh, int nsteps, int nthreads, double* counting) {                 \
    return update_quad<T>((size_t)n, pos, vel, w, delta, theta, clamp, rx, ry, rh, nsteps, nthreads, counting);     \
  }                                                                                                                 \
  ORC_API void* orc_bvh_create_##SFX(int64_t n, const T* pos, const uint32_t* w, int64_t leaf_size) {               \
    return bvh_create<T>((size_t)n, pos, w, (size_t)leaf_size);                                                     \
  }                                                                                                                 \
  ORC_API void orc_bvh_free_##SFX(void* h) { delete (BvhHandle<T>*)h; }                                             \
  ORC_API int64_t orc_bvh_num_nodes_##SFX(void* h) { return (int64_t)((BvhHandle<T>*)h)->flat.size(); }             \
  ORC_API int orc_bvh_overflow_##SFX(void* h) { return ((BvhHandle<T>*)h)->overflow; }                              \
  ORC_API void orc_bvh_export_##SFX(void* hv, T* geom /*[n][6]*/, uint32_t* mass, int32_t* is_leaf,                 \
                                    int64_t* first, int64_t* count, int64_t* skip, T* pos_perm, uint32_t* ids) {    \
    auto* h = (BvhHandle<T>*)hv;                                                                                    \
    for (size_t i = 0; i < h->flat.size(); ++i) {                                                                   \
      const auto& f = h->flat[i];                                                                                   \
      geom[6 * i + 0] = f.off_x; geom[6 * i + 1] = f.off_y; geom[6 * i + 2] = f.size_x;                             \
      geom[6 * i + 3] = f.size_y; geom[6 * i + 4] = f.cog_x; geom[6 * i + 5] = f.cog_y;                             \
      mass[i] = f.mass; is_leaf[i] = f.is_leaf; first[i] = f.first; count[i] = f.count; skip[i] = f.skip;           \
    }                                                                                                               \
    from_aos<T>(h->ps, pos_perm, nullptr, nullptr, ids);                                                            \
  }                                                                                                                 \
  ORC_API void orc_bvh_walk_##SFX(void* h, int64_t n_tgt, const T* tgt, T theta, T clamp, int nthreads, T* acc,     \
                                  uint64_t* stats) {                                                                \
    bvh_walk<T>((BvhHandle<T>*)h, (size_t)n_tgt, tgt, theta, clamp, nthreads, acc, stats);                          \
  }                                                                                                                 \
  ORC_API void orc_bvh_walk_ref_##SFX(void* h, int64_t n_tgt, const T* tgt, T theta, T clamp, int nthreads,         \
                                      double* acc, double* norm) {                                                  \
    bvh_walk_ref<T>((BvhHandle<T>*)h, (size_t)n_tgt, tgt, theta, clamp, nthreads, acc, norm);                       \
  }                                                                                                                 \
  ORC_API void orc_quad_walk_ref_##SFX(void* h, int64_t n_tgt, const T* tgt, T theta, T clamp, int nthreads,        \
                                       double* acc, double* norm) {                                                 \
    quad_walk_ref<T>((QuadHandle<T>*)h, (size_t)n_tgt, tgt, theta, clamp, nthreads, acc, norm);                     \
  }                                                                                                                 \
  ORC_API void* orc_quad_create_##SFX(int64_t n, const T* pos, const uint32_t* w, T rx, T ry, T rh) {               \
    return quad_create<T>((size_t)n, pos, w, rx, ry, rh);                                                           \
  }                                                                                                                 \
  ORC_API void orc_quad_free_##SFX(void* h) { delete (QuadHandle<T>*)h; }                                           \
  ORC_API void orc_quad_reuse_##SFX(void* h, int64_t n, const T* pos, const uint32_t* w, uint32_t* counts) {        \
    quad_reuse<T>((QuadHandle<T>*)h, (size_t)n, pos, w, counts);                                                    \
  }                                                                                                                 \
  ORC_API uint32_t orc_quad_empty_##SFX(void* h) { return quad_empty<T>(*((QuadHandle<T>*)h)->tree); }              \
  ORC_API int64_t orc_quad_num_nodes_##SFX(void* h) { return (int64_t)((QuadHandle<T>*)h)->flat.size(); }           \
  ORC_API int orc_quad_overflow_##SFX(void* h) { return ((QuadHandle<T>*)h)->overflow; }                            \
  ORC_API void orc_quad_export_##SFX(void* hv, T* geom /*[n][5]*/, uint32_t* mass, int32_t* is_leaf,                \
                                     int32_t* depth, uint32_t* child_code, uint64_t* path, int64_t* first,          \
                                     int64_t* count, int64_t* skip, uint32_t* order) {                              \
    auto* h = (QuadHandle<T>*)hv;                                                                                   \
    for (size_t i = 0; i < h->flat.size(); ++i) {                                                                   \
      const auto& f = h->flat[i];                                                                                   \
      geom[5 * i + 0] = f.off_x; geom[5 * i + 1] = f.off_y; geom[5 * i + 2] = f.height;                             \
      geom[5 * i + 3] = f.cog_x; geom[5 * i + 4] = f.cog_y;                                                         \
      mass[i] = f.mass; is_leaf[i] = f.is_leaf; depth[i] = f.depth; child_code[i] = f.child_code;                   \
      path[i] = f.path; first[i] = f.first; count[i] = f.count; skip[i] = f.skip;                                   \
    }                                                                                                               \
    std::memcpy(order, h->order.data(), h->order.size() * sizeof(uint32_t));                                        \
  }                                                                                                                 \
  ORC_API void orc_quad_walk_##SFX(void* h, int64_t n_tgt, const T* tgt, T theta, T clamp, int nthreads, T* acc,    \
                                   uint64_t* stats) {                                                               \
    quad_walk<T>((QuadHandle<T>*)h, (size_t)n_tgt, tgt, theta, clamp, nthreads, acc, stats);                        \
  }

ORC_INSTANTIATE(f32, float)
ORC_INSTANTIATE(f64, double)


// ---- draw(), /root/reference src/main.rs:41-72, with within_bounds :224-226 — restated line by line -------------
namespace {
template <class T> static inline uint32_t as_u32(T v) {  // Rust `as u32`: saturating, NaN -> 0
  if (!(v == v)) return 0u;
  if (v <= (T)0) return 0u;
  if (v >= (T)4294967295.0) return 4294967295u;
  return (uint32_t)v;
}
template <class T> static inline uint8_t as_u8(T v) {  // Rust `as u8`
  if (!(v == v)) return 0;
  if (v <= (T)0) return 0;
  if (v >= (T)255) return 255;
  return (uint8_t)v;
}
template <class T>
void draw_t(int64_t n, const T* pos, const T* vel, const uint32_t* weight, uint32_t HEIGHT, uint32_t RENDER_HEIGHT, uint8_t* frame) {
  const size_t bytes = (size_t)RENDER_HEIGHT * RENDER_HEIGHT * 4;
  for (size_t i = 0; i < bytes; ++i) frame[i] = 0;                                   // :43
  for (int64_t i = 0; i < n; ++i) {                                                  // :46
    const T x = pos[2 * i], y = pos[2 * i + 1];
    if (!(y < (T)HEIGHT && x < (T)HEIGHT && y >= (T)0 && x >= (T)0)) continue;       // :47, :224-226
    const size_t offset = ((size_t)(as_u32(y) / (HEIGHT / RENDER_HEIGHT)) * RENDER_HEIGHT +
                           (size_t)(as_u32(x) / (HEIGHT / RENDER_HEIGHT))) * 4;       // :51-54
    if (offset + 3 >= bytes) continue;  // upstream would panic; callers keep RENDER_HEIGHT | HEIGHT
    if (weight[i] > 10) {                                                            // :55
      frame[offset] = 0x00; frame[offset + 1] = 0xff; frame[offset + 2] = 0x00; frame[offset + 3] = 0xff;
    } else if (frame[offset + 3] != 0xff) {                                          // :60
      const T vx = vel[2 * i], vy = vel[2 * i + 1];
      const T a = ((vx < 0 ? -vx : vx) + (vy < 0 ? -vy : vy)) * (T)10.0;             // :62
      uint8_t b = as_u8(a);
      if (b > 0xef) b = 0xef;                                                        // :63 .min(0xef)
      const uint8_t velocity = (uint8_t)(0x10 + b);                                  // :61
      frame[offset] = 0xff;
      frame[offset + 1] = (uint8_t)(0xff - velocity);
      frame[offset + 2] = (uint8_t)(0xff - velocity);
      if (frame[offset + 3] <= 240) frame[offset + 3] = (uint8_t)(frame[offset + 3] + 10);  // :67-69
    }
  }
}
}  // namespace
ORC_API void orc_draw_f32(int64_t n, const float* pos, const float* vel, const uint32_t* weight, uint32_t height, uint32_t render_px,
                          uint8_t* frame) {
  draw_t<float>(n, pos, vel, weight, height, render_px, frame);
}
ORC_API void orc_draw_f64(int64_t n, const double* pos, const double* vel, const uint32_t* weight, uint32_t height,
                          uint32_t render_px, uint8_t* frame) {
  draw_t<double>(n, pos, vel, weight, height, render_px, frame);
}

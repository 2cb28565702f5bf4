// Launch interface of the tree kernels (tree_kernels.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbody {

// LDS written by some lanes of a wave and read by other lanes of the same wave: wave_barrier alone is a scheduling
// barrier (IntrNoMem), not a memory fence, so the hand-off is release -> barrier -> acquire at wavefront scope
// (the form bvh_build.hip's group_sync<1> uses).  Costs nothing at run time: a wave's LDS accesses complete in order.
__device__ __forceinline__ void wave_lds_handoff() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// A node's three records by SCALAR loads: the index is wave-uniform, but the compiler only picks s_load for memory it can prove
// unwritten during the kernel, which it cannot here (the kernels store accelerations and history) — so plain loads become
// vector loads of one address (a round trip through the vector L1 and 64 lanes' worth of return data for 48 bytes).  The tree was
// written by the kernels BEFORE this one, so reading it through the constant address space is sound.
template <class T> struct NodeVec4;
template <> struct NodeVec4<float> { using type = float4; };
template <> struct NodeVec4<double> { using type = double4; };
template <class T> struct NodeRec { int4 l; typename NodeVec4<T>::type b, c; };
template <class T>
__device__ __forceinline__ NodeRec<T> scalar_node_rec(const void* link, const void* geom0, const void* geom1, const int k) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef T v4t __attribute__((ext_vector_type(4)));
  using T4 = typename NodeVec4<T>::type;
  const v4i l = ((const v4i __attribute__((address_space(4)))*)link)[k];
  const v4t b = ((const v4t __attribute__((address_space(4)))*)geom0)[k];
  const v4t c = ((const v4t __attribute__((address_space(4)))*)geom1)[k];
  return NodeRec<T>{int4{l.x, l.y, l.z, l.w}, T4{b.x, b.y, b.z, b.w}, T4{c.x, c.y, c.z, c.w}};
}
// A tree-ordered particle (position, mass) the same way: leaf data is wave-uniform in the fused walks.
template <class T> struct NodeVec2;
template <> struct NodeVec2<float> { using type = float2; };
template <> struct NodeVec2<double> { using type = double2; };
template <class T>
__device__ __forceinline__ typename NodeVec2<T>::type scalar_leaf_pos(const void* leaf_pos, const int k) {
  typedef T v2t __attribute__((ext_vector_type(2)));
  const v2t q = ((const v2t __attribute__((address_space(4)))*)leaf_pos)[k];
  return typename NodeVec2<T>::type{q.x, q.y};
}
template <class T> __device__ __forceinline__ T scalar_leaf_mass(const T* leaf_mass, const int k) {
  return ((const T __attribute__((address_space(4)))*)leaf_mass)[k];
}

template <class T> struct WalkArgs {
  const void* geom0;      // T4[n_nodes]  lo.x lo.y hi.x hi.y
  const void* geom1;      // T4[n_nodes]  cog.x cog.y mass s2
  const void* link;       // int4[n_nodes] skip first count is_leaf
  int n_nodes;
  const void* leaf_pos;   // T2[n] tree-ordered particle positions
  const T* leaf_mass;     // T[n]  tree-ordered masses
  const void* tgt_pos;    // T2[*] target positions
  const uint32_t* tgt_index;  // optional: thread t handles row tgt_index[t]
  int64_t n_tgt;
  void* acc;              // T2[*] indexed like tgt_pos
  T theta, clamp;
  unsigned long long* stats;  // optional [3]: node visits, accepted nodes, leaf pairs
  int fast;                   // FAST pair arithmetic (one reciprocal) instead of the as-written IEEE divides
  int big_leaves;             // leaves hold tens of particles (BVH) rather than a handful (quad)
  const int* n_nodes_dev;     // optional (walk_tile only): the node count is read here instead — the walk was enqueued
                              // before the host saw the build's verdict; 0 there = the build failed, walk nothing
  // One-pass walks on scenes whose waves do not all fit the chip at once (round 4): the work-groups are dealt out CHUNK by chunk of
  // `order_chunk` consecutive groups, the chunks in the order `group_order` gives (heaviest first, walk_order_chunks); null: in order.
  const int* group_order;
  int order_chunk;
  int block_stride;              // one-pass walks: work-group b takes the waves of group (b * block_stride) mod gridDim.x (a stride coprime with
                                 // the grid; 1 = in order): consecutive groups hold tree-order neighbours of like weight, the stride deals them out
  unsigned long long* wave_log;  // optional (walk_tile_fast, NBODY_WALK_WAVE_LOG=1): per wave {clock ticks, node steps, leaf steps,
                                 // (target, leaf) rounds} — where the longest waves spend their time (tools/walk_wave_log.py)
};

// Device-side condition of a kernel enqueued ahead of the host's knowledge: it runs iff *nonzero != 0 (when given) and
// *zero == 0 (when given).
struct Gate {
  const int* nonzero = nullptr;
  const int* zero = nullptr;
  // a few words carried along by the kernel's first thread (a record the host reads later, written straight into its pinned
  // memory: no copy kernel of its own on the stream)
  const int* carry_src = nullptr;
  int* carry_dst = nullptr;
  int carry_words = 0;
  unsigned long long* stamp = nullptr;  // the kernel's first thread writes the 100 MHz wall clock here (phase timing without event records)
};
hipError_t launch_stamp(hipStream_t s, unsigned long long* stamp);  // a one-thread kernel that does only that

template <class T> struct GatherArgs {
  const uint32_t* perm;
  int64_t n;
  const void* pos_in; void* pos_out;
  const void* vel_in; void* vel_out;
  const uint32_t* weight_in; uint32_t* weight_out;
  const uint32_t* ids_in; uint32_t* ids_out;
  T* mass_out;
  uint32_t* perm_copy = nullptr;  // the permutation itself, written out once more (a step enqueued ahead reads it where the build left it)
  int* zero8 = nullptr;           // 8 words set to zero by the first thread (counters of a kernel further down the stream)
};

// one row of the gather (gather_particles, and the tail of bvh_emit_gather)
template <class T> __device__ __forceinline__ void gather_row(const GatherArgs<T>& a, const int64_t i) {
  using T2 = typename NodeVec2<T>::type;
  if (i == 0 && a.zero8)
    for (int k = 0; k < 8; ++k) a.zero8[k] = 0;
  if (i >= a.n) return;
  int64_t s = (int64_t)a.perm[i];
  if (a.perm_copy) a.perm_copy[i] = (uint32_t)s;
  if (s >= a.n) s = i;  // a build that failed leaves no permutation behind; the rows it produces are never used
  if (a.pos_out) reinterpret_cast<T2*>(a.pos_out)[i] = reinterpret_cast<const T2*>(a.pos_in)[s];
  if (a.vel_out) reinterpret_cast<T2*>(a.vel_out)[i] = reinterpret_cast<const T2*>(a.vel_in)[s];
  if (a.weight_out) a.weight_out[i] = a.weight_in[s];
  if (a.ids_out) a.ids_out[i] = a.ids_in[s];
  if (a.mass_out) a.mass_out[i] = (T)a.weight_in[s];  // `weight as f32`, main.rs:360
}

template <class T> hipError_t launch_tree_walk(hipStream_t s, const WalkArgs<T>& a, bool wave_uniform);
hipError_t launch_div_pair_selftest(hipStream_t s, const float* nx, const float* ny, const float* den, int64_t n, float* qx, float* qy);
template <class T> hipError_t launch_gather(hipStream_t s, const GatherArgs<T>& a);
template <class T> hipError_t launch_integrate(hipStream_t s, void* pos, void* vel, const void* acc, int64_t n, T delta, Gate gate = Gate{});

template <class T>
hipError_t launch_integrate_rows(hipStream_t s, void* pos, void* vel, const void* acc, const uint32_t* rows, int64_t row0, int64_t n, T delta);
template <class T>
hipError_t launch_export_rows(hipStream_t s, const void* pos, const void* vel, const uint32_t* rows, int64_t row0, int64_t n,
                              uint32_t* rows_out, void* pos_out, void* vel_out);
template <class T>
hipError_t launch_import_rows(hipStream_t s, void* pos, void* vel, const uint32_t* rows, int64_t n, int64_t n_total, const void* pos_in,
                              const void* vel_in);

}  // namespace nbody

// Device-side build of the linearised quad tree (gfx950) — bit-identical to the host builder and to the
// point-by-point insertion of /root/reference src/quad_tree.rs:153-270.
//
// Why it can be exact: the cell a point falls into at each level depends on that point alone
// (`child = 2*(y > ymid) + (x > xmid)` with `half = height/2`, `mid = offset + half`, quad_tree.rs:172-179), a cell
// is internal iff it holds more than 8 points (MAX_CAPACITY, :54, :159-161), children exist iff non-empty, and a
// leaf lists its points in insertion (= particle index) order.  So:
//   1. every particle walks down from the root cell on its own and packs its first 31 child codes into a 62-bit path key
//      (the reference's arithmetic, level by level);
//   2. a radix sort of the keys gives the depth-first order of the cells; a particle's leaf depth is 1 + the longest
//      prefix it shares with a window of 9 consecutive keys containing it (9 points in one cell force a split);
//   3. keys are masked to their leaf depth and sorted again, stably from particle-index order: equal masked keys = one
//      leaf, in ascending particle index, exactly the reference's slot order;
//   4. the nodes starting at sorted position r are the cells of depths first_depth(r) .. leaf_depth(r); an exclusive scan
//      of those counts numbers them in pre-order; `skip` of a node is the first node of the particle after its range;
//   5. upward pass as the reference writes it: leaf centre = unweighted sequential mean of <= 8 points, u32 wrapping
//      masses, internal centre = sum over children in code order of cog*mass, divided by the mass (:229-270).
// Anything the 62-bit key cannot express (a leaf deeper than 31 levels) or the node buffers cannot hold sets a flag and
// the caller falls back to the host builder, which has no such limit.  No sequential chain is longer than 8 elements.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

#include "quad_build.h"

namespace nbody {

namespace {

constexpr int kLevels = 31;  // child codes held by the path key (2 bits each)

template <class T> struct V2;
template <> struct V2<float> { using type = float2; };
template <> struct V2<double> { using type = double2; };
template <class T> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

// prefix of depth d (d child codes) of a path key, as a comparable integer
__device__ __forceinline__ uint64_t prefix(uint64_t key, int d) { return d <= 0 ? 0 : key >> (2 * (kLevels - d)); }
// number of leading child codes two keys share (0..31)
__device__ __forceinline__ int common_depth(uint64_t a, uint64_t b) {
  uint64_t x = (a ^ b) << 2;  // the key occupies the low 62 bits
  if (x == 0) return kLevels;
  return __builtin_clzll(x) / 2;
}

template <class T>
__global__ __launch_bounds__(256) void qb_path_keys(const void* pos_, int n, T rx, T ry, T rh, uint64_t* keys, uint32_t* idx) {
  using T2 = typename V2<T>::type;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T2 p = reinterpret_cast<const T2*>(pos_)[i];
  T ox = rx, oy = ry, h = rh;
  uint64_t key = 0;
  for (int l = 0; l < kLevels; ++l) {
    const T half = h / (T)2.0;                         // quad_tree.rs:172
    const T xmid = ox + half, ymid = oy + half;        // :174-175
    const int north = p.y > ymid, west = p.x > xmid;   // :176-177 (strict)
    key = (key << 2) | (uint64_t)((north << 1) + west);
    ox = west ? ox + half : ox + (T)0.0;               // :183-186
    oy = north ? oy + half : oy + (T)0.0;
    h = half;
  }
  keys[i] = key;
  idx[i] = (uint32_t)i;
}

// Leaf depth of the particle at sorted position r — and its final position: a leaf's particles stand in index order
// (the reference pushes them in that order, quad_tree.rs insert), while the sort has ordered them by the child codes
// below the leaf.  A leaf holds at most 8 particles (9 that share a cell split it), all within 7 places of r: r's rank
// by index among them is its place.  (This replaces masking the keys and a second, stable sort of all of them.)
// Writes, at the final position: the key cut at the leaf's depth, the particle's index, the leaf depth.
__global__ __launch_bounds__(256) void qb_leaf_order(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, int n,
                                                     uint64_t* __restrict__ keys2, uint32_t* __restrict__ order, int* __restrict__ ld_out,
                                                     int* __restrict__ flags, int sort_levels) {
  __shared__ int wave_max[4];
  int r = blockIdx.x * 256 + threadIdx.x;
  const bool live = r < n;
  if (!live) r = n - 1;  // (n > 0) a bystander that stays for the barrier; it writes nothing
  uint64_t w[17];  // keys[r - 8 .. r + 8]
#pragma unroll
  for (int j = 0; j < 17; ++j) {
    const int s = r - 8 + j;
    w[j] = (s >= 0 && s < n) ? keys[s] : 0ull;
  }
  int best = -1;  // longest prefix shared by 9 consecutive keys that include r
#pragma unroll
  for (int j = 0; j <= 8; ++j) {
    const int s = r - 8 + j;
    if (s < 0 || s + 8 >= n) continue;
    const int c = common_depth(w[j], w[j + 8]);  // sorted: the whole window shares what its ends share
    best = c > best ? c : best;
  }
  int ld = best + 1;  // -1 -> the root itself is a leaf (n <= 8)
  if (ld > kLevels) {
    atomicOr(&flags[0], 1);  // deeper than the key can say: host builder
    ld = kLevels;
  }
  // the sort looked at the first sort_levels child codes only: 9 keys that agree on all of them are in no particular
  // order below, so a leaf that deep has to be found again with more levels sorted
  if (ld > sort_levels) atomicOr(&flags[0], 2);
  const uint64_t k = w[8];
  const int sh = 2 * (kLevels - ld);
  const uint64_t mine = ld >= kLevels ? k : (ld <= 0 ? 0ull : k >> sh);
  // my leaf's neighbours: a run around r (when a flag above is up the run may be cut short: the build is redone anyway)
  int lo = 8, hi = 9;
#pragma unroll
  for (int j = 7; j >= 1; --j) {
    const int s = r - 8 + j;
    const uint64_t o = ld >= kLevels ? w[j] : (ld <= 0 ? 0ull : w[j] >> sh);
    if (lo == j + 1 && s >= 0 && o == mine) lo = j;
  }
#pragma unroll
  for (int j = 9; j <= 15; ++j) {
    const int s = r - 8 + j;
    const uint64_t o = ld >= kLevels ? w[j] : (ld <= 0 ? 0ull : w[j] >> sh);
    if (hi == j && s < n && o == mine) hi = j + 1;
  }
  const uint32_t me = idx[r];
  int rank = 0;
  for (int j = lo; j < hi; ++j) rank += idx[r - 8 + j] < me;
  int p = r - 8 + lo + rank;
  p = p < n ? p : n - 1;
  if (live) {
    keys2[p] = ld >= kLevels ? k : (ld <= 0 ? 0ull : mine << sh);
    order[p] = me;
    ld_out[p] = ld;
  }
  // the tree's depth (flags[2]): one atomic per work-group, and only from groups that would raise it
  int v = ld;
  for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(v, o); v = x > v ? x : v; }
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k2 = 1; k2 < 4; ++k2) v = wave_max[k2] > v ? wave_max[k2] : v;
    if (v > __hip_atomic_load(&flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&flags[2], v);
  }
}

// nodes that start at sorted position r: depths first_depth(r) .. leaf_depth(r)
__global__ __launch_bounds__(256) void qb_node_counts(const uint64_t* __restrict__ keys2, const int* __restrict__ ld, int n,
                                                      int* __restrict__ fd, uint32_t* __restrict__ cnt) {
  int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const int l = ld[r];
  int f = 0;
  if (r > 0) {
    const uint64_t a = keys2[r - 1], b = keys2[r];
    f = (a == b) ? kLevels + 1 : common_depth(a, b) + 1;
  }
  fd[r] = f;
  cnt[r] = f <= l ? (uint32_t)(l - f + 1) : 0u;
}

template <class T>
__global__ __launch_bounds__(256) void qb_emit_nodes(const uint64_t* __restrict__ keys2, const int* __restrict__ ld,
                                                     const int* __restrict__ fd, const uint32_t* __restrict__ cnt,
                                                     const uint32_t* __restrict__ base, int n, int n_nodes, T rx, T ry, T rh,
                                                     void* geom0_, void* geom1_, int4* __restrict__ link, int* __restrict__ depth) {
  using T4 = typename V4<T>::type;
  int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n || cnt[r] == 0) return;
  const uint64_t key = keys2[r];
  const int f = fd[r], l = ld[r];
  T ox = rx, oy = ry, h = rh;
  int bound = -1;  // end of the cell one level up: this level's cell ends no later
  for (int d = 0; d <= l; ++d) {
    if (d >= f) {
      const int j = (int)base[r] + (d - f);
      // extent of the cell: particles sharing the depth-d prefix, starting at r
      const uint64_t pre = prefix(key, d);
      int lo = r, hi;  // lo: shares the prefix; hi: first known position that does not (or n)
      if (bound < 0) {  // the shallowest node starting here: gallop (most cells are short; a search over all n keys
        hi = n;         // is 20 dependent loads)
        for (int step = 1; lo + step < n; step *= 2) {
          if (prefix(keys2[lo + step], d) == pre) lo += step;
          else { hi = lo + step; break; }
        }
      } else {
        hi = bound;
      }
      while (lo + 1 < hi) {
        int mid = (lo + hi) >> 1;
        if (prefix(keys2[mid], d) == pre) lo = mid; else hi = mid;
      }
      const int end = hi;
      bound = end;
      const int skip = end < n ? (int)base[end] : n_nodes;
      reinterpret_cast<T4*>(geom0_)[j] = T4{ox, oy, ox + h, oy + h};
      reinterpret_cast<T4*>(geom1_)[j] = T4{(T)0, (T)0, (T)0, h * h};  // height2, quad_tree.rs:19
      link[j] = make_int4(skip, r, end - r, d == l ? 1 : 0);
      depth[j] = d;
    }
    if (d < l) {  // descend one level along the path
      const int code = (int)((key >> (2 * (kLevels - 1 - d))) & 3);
      const T half = h / (T)2.0;
      ox = (code & 1) ? ox + half : ox + (T)0.0;
      oy = (code & 2) ? oy + half : oy + (T)0.0;
      h = half;
    }
  }
}

// leaves: quad_tree.rs:231-241 (unweighted mean in slot order) and :139-151 (u32 mass)
template <class T>
__global__ __launch_bounds__(256) void qb_leaf_stats(const void* pos_, const uint32_t* __restrict__ weight,
                                                     const uint32_t* __restrict__ order, const int4* __restrict__ link,
                                                     int n_nodes, void* geom1_, uint32_t* __restrict__ mass) {
  using T2 = typename V2<T>::type;
  using T4 = typename V4<T>::type;
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_nodes) return;
  const int4 lk = link[j];
  if (!lk.w) return;
  uint32_t ms = 0;
  T ax = 0, ay = 0;
  for (int k = 0; k < lk.z; ++k) {
    const uint32_t id = order ? order[lk.y + k] : (uint32_t)(lk.y + k);  // (no order: pos and weight are in tree order already)
    const T2 q = reinterpret_cast<const T2*>(pos_)[id];
    ms += weight[id];
    ax = ax + q.x;
    ay = ay + q.y;
  }
  T4 g = reinterpret_cast<T4*>(geom1_)[j];
  if (lk.z > 0) { g.x = ax / (T)lk.z; g.y = ay / (T)lk.z; }
  g.z = (T)ms;
  reinterpret_cast<T4*>(geom1_)[j] = g;
  mass[j] = ms;
}

// internal cells of one depth: quad_tree.rs:243-268, children in index order
template <class T>
__global__ __launch_bounds__(256) void qb_internal_stats(const int4* __restrict__ link, const int* __restrict__ depth, int n_nodes,
                                                         int d, void* geom1_, uint32_t* __restrict__ mass) {
  using T4 = typename V4<T>::type;
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_nodes) return;
  const int4 lk = link[j];
  if (lk.w || depth[j] != d) return;
  T4* g1 = reinterpret_cast<T4*>(geom1_);
  uint32_t ms = 0;
  T bx = 0, by = 0;
  for (int c = j + 1; c < lk.x; c = link[c].x) {
    const uint32_t mc = mass[c];
    const T4 gc = g1[c];
    ms += mc;
    bx = bx + gc.x * (T)mc;
    by = by + gc.y * (T)mc;
  }
  T4 g = g1[j];
  g.x = bx / (T)ms;
  g.y = by / (T)ms;
  g.z = (T)ms;
  g1[j] = g;
  mass[j] = ms;
}

__global__ void qb_totals(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ base, const int* __restrict__ ld, int n,
                          int* __restrict__ flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  flags[1] = n > 0 ? (int)(base[n - 1] + cnt[n - 1]) : 1;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

QuadBuildLayout quad_build_layout(int64_t n_) {
  QuadBuildLayout L{};
  size_t n = (size_t)(n_ > 0 ? n_ : 1), off = 0;
  L.flags = off; off += 256;
  L.keys_a = off; off += align_up(n * 8);
  L.keys_b = off; off += align_up(n * 8);
  L.keys_by_index = off; off += align_up(n * 8);
  L.idx_a = off; off += align_up(n * 4);
  L.idx_b = off; off += align_up(n * 4);
  L.ld_by_index = off; off += align_up(n * 4);
  L.ld = off; off += align_up(n * 4);
  L.fd = off; off += align_up(n * 4);
  L.cnt = off; off += align_up(n * 4);
  L.base = off; off += align_up(n * 4);
  L.cub_temp = off;
  L.cub_temp_bytes = align_up((size_t)8 << 20);
  off += L.cub_temp_bytes;
  L.total = off;
  return L;
}

// Phase A: everything up to the node count (flags[1]).  The caller reads flags {fallback, n_nodes, max_depth}.
template <class T>
hipError_t quad_build_phase_a(hipStream_t s, const void* pos, int n, T rx, T ry, T rh, char* scratch, const QuadBuildLayout& L,
                              uint32_t* order_out, int sort_levels) {
  if (sort_levels < 1) sort_levels = 1;
  if (sort_levels > kLevels) sort_levels = kLevels;
  const int bit0 = 2 * (kLevels - sort_levels);  // the radix sorts skip the child codes below sort_levels
  int* flags = (int*)(scratch + L.flags);
  uint64_t* ka = (uint64_t*)(scratch + L.keys_a);
  uint64_t* kb = (uint64_t*)(scratch + L.keys_b);
  uint64_t* kf = (uint64_t*)(scratch + L.keys_by_index);  // the keys in their final order, cut at the leaf depth (phase B reads them)
  uint32_t* ia = (uint32_t*)(scratch + L.idx_a);
  uint32_t* ib = (uint32_t*)(scratch + L.idx_b);
  int* ld = (int*)(scratch + L.ld);
  int* fd = (int*)(scratch + L.fd);
  uint32_t* cnt = (uint32_t*)(scratch + L.cnt);
  uint32_t* base = (uint32_t*)(scratch + L.base);
  hipError_t e = hipMemsetAsync(flags, 0, 256, s);
  if (e != hipSuccess) return e;
  if (n <= 0) {
    hipLaunchKernelGGL(qb_totals, dim3(1), dim3(1), 0, s, cnt, base, ld, 0, flags);
    return hipGetLastError();
  }
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL((qb_path_keys<T>), dim3(blocks), dim3(256), 0, s, pos, n, rx, ry, rh, ka, ia);
  size_t tb = L.cub_temp_bytes, need = 0;
  {
    // rocprim's own choice below a million keys is a merge sort (a block sort and ~10 merge passes over whole keys, 0.2 ms
    // at N = 2^20); the Onesweep radix sort looks at the sorted bits only (4-5 passes of 8 bits): taken above 256 k keys
    using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 262144>;
    rocprim::double_buffer<uint64_t> dk(ka, kb);
    rocprim::double_buffer<uint32_t> dv(ia, ib);
    e = rocprim::radix_sort_pairs<SortConfig>(nullptr, need, dk, dv, (size_t)n, 0u, 62u, s);  // (size query: the widest sort)
    if (e != hipSuccess) return e;
    if (need > L.cub_temp_bytes) return hipErrorOutOfMemory;
    e = rocprim::radix_sort_pairs<SortConfig>(scratch + L.cub_temp, tb, dk, dv, (size_t)n, (unsigned)bit0, 62u, s);
    if (e != hipSuccess) return e;
    // leaf depths, and every leaf's particles into index order: final keys (cut at the leaf) in kf, final order in order_out
    hipLaunchKernelGGL(qb_leaf_order, dim3(blocks), dim3(256), 0, s, dk.current(), dv.current(), n, kf, order_out, ld, flags, sort_levels);
  }
  hipLaunchKernelGGL(qb_node_counts, dim3(blocks), dim3(256), 0, s, kf, ld, n, fd, cnt);
  need = 0;
  e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, cnt, base, n, s);
  if (e != hipSuccess) return e;
  if (need > L.cub_temp_bytes) return hipErrorOutOfMemory;
  tb = L.cub_temp_bytes;
  e = hipcub::DeviceScan::ExclusiveSum(scratch + L.cub_temp, tb, cnt, base, n, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(qb_totals, dim3(1), dim3(1), 0, s, cnt, base, ld, n, flags);
  return hipGetLastError();
}

// Phase B: nodes, leaf statistics, upward pass.  n_nodes and max_depth are what phase A reported.
template <class T>
hipError_t quad_build_phase_b(hipStream_t s, const void* pos, const uint32_t* weight, int n, T rx, T ry, T rh, char* scratch,
                              const QuadBuildLayout& L, const uint32_t* order, int n_nodes, int max_depth, void* geom0,
                              void* geom1, void* link, int* depth, uint32_t* mass) {
  const uint64_t* ka = (const uint64_t*)(scratch + L.keys_by_index);  // final order, cut at the leaf depth (phase A)
  const int* ld = (const int*)(scratch + L.ld);
  const int* fd = (const int*)(scratch + L.fd);
  const uint32_t* cnt = (const uint32_t*)(scratch + L.cnt);
  const uint32_t* base = (const uint32_t*)(scratch + L.base);
  if (n <= 0) return hipSuccess;
  const unsigned pblocks = (unsigned)((n + 255) / 256), nblocks = (unsigned)((n_nodes + 255) / 256);
  hipLaunchKernelGGL((qb_emit_nodes<T>), dim3(pblocks), dim3(256), 0, s, ka, ld, fd, cnt, base, n, n_nodes, rx, ry, rh, geom0,
                     geom1, (int4*)link, depth);
  hipLaunchKernelGGL((qb_leaf_stats<T>), dim3(nblocks), dim3(256), 0, s, pos, weight, order, (const int4*)link, n_nodes, geom1, mass);
  for (int d = max_depth - 1; d >= 0; --d)
    hipLaunchKernelGGL((qb_internal_stats<T>), dim3(nblocks), dim3(256), 0, s, (const int4*)link, depth, n_nodes, d, geom1, mass);
  return hipGetLastError();
}

template hipError_t quad_build_phase_a<float>(hipStream_t, const void*, int, float, float, float, char*, const QuadBuildLayout&, uint32_t*, int);
template hipError_t quad_build_phase_a<double>(hipStream_t, const void*, int, double, double, double, char*, const QuadBuildLayout&, uint32_t*, int);
template hipError_t quad_build_phase_b<float>(hipStream_t, const void*, const uint32_t*, int, float, float, float, char*,
                                              const QuadBuildLayout&, const uint32_t*, int, int, void*, void*, void*, int*, uint32_t*);
template hipError_t quad_build_phase_b<double>(hipStream_t, const void*, const uint32_t*, int, double, double, double, char*,
                                               const QuadBuildLayout&, const uint32_t*, int, int, void*, void*, void*, int*, uint32_t*);

}  // namespace nbody

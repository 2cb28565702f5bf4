// Device-side BVH build (bvh_build.hip), f32.  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nbody {

constexpr int kBvhLevels = 64;    // level counters kept on the device
constexpr int kBvhKeyDepth = 56;  // deepest node the pre-order key can express; deeper -> host builder
constexpr int kBvhClasses = 3;    // work-group sizes 1024 / 256 / 64 by node length

// flags (int[] at scratch + L.flags)
enum : int {
  kBvhFallback = 0,   // != 0: the device build declines (buffers, depth, NaN positions): host builder
  kBvhNodeCount = 1,  // nodes allocated so far (breadth-first ids)
  kBvhMaxDepth = 2,
  kBvhStops = 3,      // restarts of the exact-sum scan (diagnostic)
  kBvhFlagWords = 8,
};

struct BvhBuildLayout {
  int node_cap;
  size_t flags, qcount, queue, pts, ids, lidx, ridx;
  size_t nbegin, nlen, nparent, nchild, ndepth, nleaf, nkey, nbox, ncog, nmass, narrive;
  size_t keys_sorted, vals, vals_sorted, rank, cub_temp, cub_temp_bytes, total;
};
BvhBuildLayout bvh_build_layout(int64_t n, int leaf_size);

// Zeroes the counters, copies the positions into the working array and seeds the root.
hipError_t bvh_build_begin(hipStream_t s, const void* pos, int n, char* scratch, const BvhBuildLayout& L);
// Enqueues levels [level_begin, level_end).  flags / per-level queue counts are read by the caller afterwards.
hipError_t bvh_build_levels(hipStream_t s, int n, int leaf_size, int level_begin, int level_end, char* scratch,
                            const BvhBuildLayout& L);
// Pre-order numbering, leaves + upward pass, final arrays.  n_nodes / max_depth as read from the flags.
hipError_t bvh_build_finish(hipStream_t s, const uint32_t* weight, int n, int n_nodes, char* scratch, const BvhBuildLayout& L,
                            uint32_t* order_out, void* geom0, void* geom1, void* link, int* depth_out, uint32_t* mass_out,
                            float2* size_out);

}  // namespace nbody

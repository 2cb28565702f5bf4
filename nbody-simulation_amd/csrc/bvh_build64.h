// Device-side BVH build for f64 positions (bvh_build64.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nbody {

constexpr int kB64Levels = 62;  // deepest level the device build follows; deeper (degenerate input) -> host builder

// flags (int[] at scratch + L.flags)
enum : int {
  kB64Fallback = 0,   // != 0: the device build declines (NaN positions, buffers): host builder
  kB64NodeCount = 1,  // nodes made so far (breadth-first ids)
  kB64MaxDepth = 2,
  kB64Stops = 3,      // restarts of the exact-sum scan (diagnostic)
  kB64RunsUsed = 4,   // segments of long chains whose prepared run was applied (diagnostic)
  kB64FlagWords = 16,
};

struct Bvh64Layout {
  int node_cap, open_cap, chunk_cap, seg_cap;
  size_t flags, opencount, chunkcount, segcount, zero_end, nseg0, seg_node, seg_index, seg_sum, seg_min, seg_max, seg_prefix, seg_pred, seg_run, P, ID, nbegin, nlen, ndepth, nchild, nleaf, ncx, ncy, nsplit, naxis, nk, nchunk0, nsub, npre,
      nsum, nmin, nmax, nmean, ncog, nmass, openq, ch_node, ch_index, ch_l, ch_r, ch_loff, ch_roff, lidx, ridx, total;
};
Bvh64Layout bvh64_layout(int64_t n, int leaf_size);

// Levels with open nodes a balanced tree over n points has (the caller enqueues these blind, plus a margin, then asks).
int bvh64_first_levels(int64_t n, int leaf_size);
// Zeroes the counters, copies the positions into the working array and seeds the root (n > 0).
hipError_t bvh64_begin(hipStream_t s, const void* pos, int n, char* scratch, const Bvh64Layout& L);
// Enqueues levels [level_begin, level_end).  opencount[level_end] (int at scratch + L.opencount) != 0 afterwards: more to do.
hipError_t bvh64_levels(hipStream_t s, int n, int leaf_size, int level_begin, int level_end, char* scratch, const Bvh64Layout& L);
// Leaves, upward pass, pre-order numbering, the final arrays (sized for L.node_cap nodes; the node count is in the flags).
// level_end: as many levels as have been enqueued.  May be called again after more levels.
hipError_t bvh64_finish(hipStream_t s, const uint32_t* weight, int n, int level_end, char* scratch, const Bvh64Layout& L, uint32_t* order_out,
                        void* geom0, void* geom1, void* link, int* depth_out, uint32_t* mass_out, void* size_out /* double2[] */);

}  // namespace nbody

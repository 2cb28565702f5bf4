// Device-side BVH build (bvh_build.hip), f32.  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nbody {
template <class T> struct GatherArgs;  // tree_kernels.h

constexpr int kBvhLevels = 64;    // level counters kept on the device
constexpr int kBvhKeyDepth = 56;  // deepest level the device build follows; deeper (degenerate input) -> host builder

// flags (int[] at scratch + L.flags)
enum : int {
  kBvhFallback = 0,   // != 0: the device build declines (buffers, depth, NaN positions): host builder
  kBvhNodeCount = 1,  // node ids handed out so far: a high-water mark (the subtree kernel takes its ids a range at a time and
                      // may leave the end of a range unused: ids are not dense; kBvhNodes counts the nodes)
  kBvhMaxDepth = 2,
  kBvhStops = 3,      // restarts of the exact-sum scan (diagnostic)
  kBvhSubCount = 4,   // subtree roots queued for bvh_subtrees
  kBvhTopCount = 5,   // nodes made by the long-node levels (they sit above the subtrees)
  kBvhRunsUsed = 6,   // chunks of long chains whose prepared run was used (diagnostic)
  kBvhBadIndex = 7,   // bvh_emit met a node it could not number (expected while long nodes are still pending only)
  kBvhDebug = 8,      // 8 words of max-over-groups cycle counts per phase of bvh_subtrees (NB_BVH_TIMING builds)
  kBvhNodes = 16,     // nodes of the finished tree (the root's subtree size, written by bvh_top_upward)
  kBvhFlagWords = 24,
};

struct BvhBuildLayout {
  int node_cap, big_cap, chunk_cap;
  size_t flags, bigcount, chunkcount, zero_end, bigq, subq, topq, ch_rec, ch_sum, ch_box, ch_run, ch_cx, ch_cy, ch_before, pts, ids, lidx, ridx, pred, pair;
  size_t nbegin, nlen, nparent, nchild, ndepth, nleaf, nsize, npre, nbox, ncog, nmass, narrive, nmean, nsplit, nchunk0, ndone, nbad;
  size_t total;
};
BvhBuildLayout bvh_build_layout(int64_t n, int leaf_size);

// Zeroes the counters, copies the positions into the working array and seeds the root.
// flags_clean: the flags and level counters are zero already (the step before left them so: TileTail::clear_flags).
// stamp_begin / stamp_prev_end (optional): the first thread writes the 100 MHz wall clock there (phase timing without event records)
hipError_t bvh_build_begin(hipStream_t s, const void* pos, int n, char* scratch, const BvhBuildLayout& L, bool flags_clean = false,
                           unsigned long long* stamp_begin = nullptr, unsigned long long* stamp_prev_end = nullptr);
// Levels of long nodes a balanced tree over n points has (the caller enqueues these blind, then asks).
int bvh_build_first_levels(int64_t n);
// Enqueues the long-node passes of levels [level_begin, level_end).  bigcount[level_end] != 0 afterwards: more to do.
hipError_t bvh_build_levels(hipStream_t s, int n, int leaf_size, int level_begin, int level_end, char* scratch,
                            const BvhBuildLayout& L);
// Subtrees, pre-order numbering, leaves + upward pass, final arrays (sized for L.node_cap nodes; the node count is in
// the flags afterwards).
// sub_start: subtree roots already built by an earlier call (the caller enqueues this blind after the levels it
// expects and calls again, after more levels, if long nodes were left).
hipError_t bvh_build_finish(hipStream_t s, const uint32_t* weight, int n, int leaf_size, int sub_start, char* scratch,
                            const BvhBuildLayout& L,
                            uint32_t* order_out, void* geom0, void* geom1, void* link, int* depth_out, uint32_t* mass_out,
                            float2* size_out, const GatherArgs<float>* gather = nullptr);
// gather: the row gather of the step (tree_kernels.h; perm = bvh_build_order, perm_copy = where the caller wants the permutation)
// done by the numbering's launch instead of one of its own.
// The permutation where the build leaves it (n words inside `scratch`): with order_out == nullptr bvh_build_finish does not
// copy it out, the caller's gather reads it here.
const uint32_t* bvh_build_order(const char* scratch, const BvhBuildLayout& L);


}  // namespace nbody

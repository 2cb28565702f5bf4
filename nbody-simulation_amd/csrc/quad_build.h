// Device-side quad-tree build (quad_build.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nbody {

struct QuadBuildLayout {
  size_t flags, keys_a, keys_b, keys_by_index, idx_a, idx_b, ld_by_index, ld, fd, cnt, base, cub_temp, cub_temp_bytes, total;
};
QuadBuildLayout quad_build_layout(int64_t n);

// flags (int[3] at scratch + L.flags) after phase A: {bit 0: needs the host builder, bit 1: a leaf lies deeper than
// sort_levels (call again with more), n_nodes, max_depth}.  sort_levels (1..31): how many child codes the two radix
// sorts look at — the previous step's depth plus a few is enough and saves passes.
template <class T>
hipError_t quad_build_phase_a(hipStream_t s, const void* pos, int n, T rx, T ry, T rh, char* scratch, const QuadBuildLayout& L,
                              uint32_t* order_out, int sort_levels);
template <class T>
hipError_t quad_build_phase_b(hipStream_t s, const void* pos, const uint32_t* weight, int n, T rx, T ry, T rh, char* scratch,
                              const QuadBuildLayout& L, const uint32_t* order, int n_nodes, int max_depth, void* geom0,
                              void* geom1, void* link, int* depth, uint32_t* mass);

}  // namespace nbody

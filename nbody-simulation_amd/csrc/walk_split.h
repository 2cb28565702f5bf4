// Barnes-Hut walk in three parallel passes (walk_split.hip), f32, bit-identical to the fused walk.  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "tree_kernels.h"

namespace nbody {

struct WalkSplitLayout {
  size_t cnt, off, info, cub_temp, cub_temp_bytes, total;  // bytes from the start of the scratch block
};
WalkSplitLayout walk_split_layout(int64_t n_tgt);

// info (int[8] at scratch + L.info) afterwards: {total terms, term array too small, 32-bit offsets wrapped, terms per wave
// of the term pass, ...}.
// terms: float2[term_capacity].  When the walk needs more than term_capacity terms nothing is written to acc and the
// overflow flag is set: the caller grows the buffer (or uses the fused walk) and calls again.
hipError_t launch_tree_walk_split(hipStream_t s, const WalkArgs<float>& a, char* scratch, const WalkSplitLayout& L, void* terms,
                                  int64_t term_capacity);

}  // namespace nbody

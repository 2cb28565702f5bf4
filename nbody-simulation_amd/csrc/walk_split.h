// Barnes-Hut walk in three parallel passes (walk_split.hip), f32, bit-identical to the fused walk.  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "tree_kernels.h"

namespace nbody {

struct WalkSplitLayout {
  size_t cnt, off, info, cub_temp, cub_temp_bytes, total;  // bytes from the start of the scratch block
  size_t scan_state, scan_state_bytes;  // walk_scan_est_tail: a ticket and one 64-bit state per work-group (zeroed once, by the caller)
  size_t order;                         // walk_order_chunks: the chunks of work-groups, heaviest first (kWalkOrderChunks ints)
};
constexpr int kWalkOrderChunks = 256;   // at most this many chunks (a chunk: at least 64 work-groups = 256 waves)
WalkSplitLayout walk_split_layout(int64_t n_tgt);
constexpr int64_t kWalkFusedScanMaxTargets = (int64_t)1 << 24;  // TileTail::fused_scan: at most this many targets

// info (int[8] at scratch + L.info) afterwards: {total terms, term array too small, 32-bit offsets wrapped, terms per wave
// of the term pass, ...}.
// terms: float2[term_capacity].  When the walk needs more than term_capacity terms nothing is written to acc and the
// overflow flag is set: the caller grows the buffer (or uses the fused walk) and calls again.
hipError_t launch_tree_walk_split(hipStream_t s, const WalkArgs<float>& a, char* scratch, const WalkSplitLayout& L, void* terms,
                                  int64_t term_capacity);

// The same walk in one pass, the terms handed from "lane = particle" to "lane = target" through LDS (walk_tile): no term
// array.  The waves are cut by an estimate of each target's work.  estimate 1: hist[tgt_ids[t]] holds the term count of
// target t's particle in the previous walk (scaled down by `shift` bits so that the sum stays below 2^31); 0: a counting
// traversal runs first; 2: none (64 targets per wave; for walks whose counts overflow 32 bits and have no history yet).
// hist (may be null: nothing is recorded) receives this walk's counts, by particle id.
// info afterwards: [1] != 0: nothing was written to acc (the estimate's scan overflowed: call again with estimate 2);
// info[6..7]: this walk's total terms (unsigned long long).
template <class T>
hipError_t launch_tree_walk_tile(hipStream_t s, const WalkArgs<T>& a, char* scratch, const WalkSplitLayout& L, const uint32_t* tgt_ids,
                                 uint32_t* hist, int estimate, int shift);

// The same in two halves (preparation: estimate scan, wrap check, budget — info[0..3] final; then the walk kernel), for
// a caller that enqueues a copy of info and an event in between.
// tail (estimate 1 only): what a step enqueued ahead of the host needs between its build and its walk rides along with the
// estimate check instead of costing launches of its own — the check's last work-group also sets the budget, concludes on the
// build, packs verdict | build flags | info for ONE copy to the host, and clears the build's
// counters for the next step.
struct TileTail {
  const int* flags = nullptr;     // the build's flags and level counters: flag_words of them are packed
  int flag_words = 0;
  const int* bigcount = nullptr;  // verdict: no long node left at level_end, no fallback, every node numbered, count <= node_cap
  int level_end = 0, node_cap = 0;
  int* verdict = nullptr;         // device: [0] node count or 0 (walk_tile reads it), [1] 0/1
  int* pack = nullptr;            // verdict (2) | flags (flag_words) | info (8)
  int* clear = nullptr;           // zeroed once packed
  int clear_words = 0;
  bool info_zeroed = false;       // info is zero already (an earlier kernel of the stream did it)
  unsigned long long* stamp = nullptr;  // fused scan only: the kernel's first thread writes the 100 MHz wall clock here (the walk phase begins)
  // The estimate's scan, its overflow check and the duties above as ONE kernel (walk_scan_est_tail: a single-pass scan with
  // decoupled look-back) instead of the library scan's two launches + the check's.  The caller has zeroed
  // scratch + L.scan_state (L.scan_state_bytes) once, when it allocated the scratch; the kernel keeps its own books there.
  bool fused_scan = false;
};
template <class T>
hipError_t launch_tree_walk_tile_prep(hipStream_t s, const WalkArgs<T>& a, char* scratch, const WalkSplitLayout& L, const uint32_t* tgt_ids,
                                      uint32_t* hist, int estimate, int shift, int64_t* grid_waves, const TileTail* tail = nullptr);
template <class T>
hipError_t launch_tree_walk_tile_main(hipStream_t s, const WalkArgs<T>& a, char* scratch, const WalkSplitLayout& L, const uint32_t* tgt_ids,
                                      uint32_t* hist, int64_t grid_waves);

}  // namespace nbody

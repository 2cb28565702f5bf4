// Delta-snapshot stream "NBD1" (SURVEY §8f-4): the format, shared by the device encoder (delta_snapshot.hip) and the
// host decoder (capi.hip).  Upstream has only a commented-out experiment that prints the zstd size of raw position
// differences (main.rs:119-134): there is no format to match, this one is ours.  Lossless on the bit patterns.
//
//   element   a coordinate's bits (u32 for f32 contexts, u64 for f64), mapped to an ordered integer key:
//             key = bits ^ (sign ? all ones : sign bit)        (monotone in the float's value; any bit pattern allowed)
//   state     per body id and coordinate the keys of the last two snapshots (prev, prev2); zero after a reset
//   residual  r = key - prediction (wrapping), prediction = prev (predictor 0) or 2*prev - prev2 (predictor 1),
//             zigzag z = (r << 1) ^ (r >> (bits-1))  (arithmetic shift)
//   block     64 consecutive body ids x one coordinate; width w = bits needed by the largest z of the block; the
//             predictor with the smaller w is taken (ties: predictor 0); lanes past n hold z = 0
//   stream    header (32 bytes, little endian):
//               0  'N' 'B' 'D' '1'
//               4  u8 element bits (32 | 64)     5  u8 key frame (1: the decoder zeroes its state first)    6  u16 0
//               8  u64 n                        16  u64 step                    24  u64 payload words (64-bit)
//             widths: 2 * ceil(n/64) bytes, [2*block + coordinate]: bits 0-6 = w (0..64), bit 7 = predictor; zero
//               padded to a multiple of 8 bytes
//             payload: for every (block, coordinate) in that order w words; bit l of word b = bit b of z of id 64*block + l
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace nbody {

constexpr size_t kDeltaHeader = 32;

#if defined(__HIPCC__)
#define NB_HD __host__ __device__ __forceinline__
#else
#define NB_HD inline
#endif

NB_HD uint32_t delta_key(uint32_t u) { return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
NB_HD uint64_t delta_key(uint64_t u) { return u ^ ((u >> 63) ? ~0ull : 0x8000000000000000ull); }
NB_HD uint32_t delta_unkey(uint32_t k) { return k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu); }
NB_HD uint64_t delta_unkey(uint64_t k) { return k ^ ((k >> 63) ? 0x8000000000000000ull : ~0ull); }
NB_HD uint32_t delta_zigzag(uint32_t r) { return (r << 1) ^ (uint32_t)((int32_t)r >> 31); }
NB_HD uint64_t delta_zigzag(uint64_t r) { return (r << 1) ^ (uint64_t)((int64_t)r >> 63); }
NB_HD uint32_t delta_unzigzag(uint32_t z) { return (z >> 1) ^ (0u - (z & 1u)); }
NB_HD uint64_t delta_unzigzag(uint64_t z) { return (z >> 1) ^ (0ull - (z & 1ull)); }

inline size_t delta_blocks(int64_t n) { return (size_t)((n + 63) / 64); }
inline size_t delta_width_bytes(int64_t n) { return (2 * delta_blocks(n) + 7) / 8 * 8; }
// Largest stream n bodies can produce.
inline size_t delta_bound(int64_t n, int elem_bits) {
  return kDeltaHeader + delta_width_bytes(n) + 2 * delta_blocks(n) * (size_t)elem_bits * 8;
}

}  // namespace nbody

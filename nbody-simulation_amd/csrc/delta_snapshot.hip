// Device encoder of the delta-snapshot stream (format: delta_codec.h).  Three launches and a scan, all
// memory bound: 8 B + 4 B read and 8 B written per body by the scatter, then three key arrays read once by the width
// pass and once by the packing pass (the residuals are recomputed instead of stored).  One wave = one block of 64 ids,
// so a bit plane of a block is exactly one ballot.
#include "delta_snapshot.h"

#include <hipcub/hipcub.hpp>

#include "delta_codec.h"

namespace nbody {
namespace {

template <class T> struct KeyOf;
template <> struct KeyOf<float> { using type = uint32_t; };
template <> struct KeyOf<double> { using type = uint64_t; };

// cur[c][ids[row]] = key(pos[row].c): the stream is in id order whatever the tree builds did to the rows
template <class T>
__global__ __launch_bounds__(256) void delta_scatter(int64_t n, int64_t npad, const void* pos, const uint32_t* ids, void* cur) {
  using K = typename KeyOf<T>::type;
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const K* p = reinterpret_cast<const K*>(pos) + 2 * row;
  const uint32_t id = ids[row];
  K* out = reinterpret_cast<K*>(cur);
  out[id] = delta_key(p[0]);
  out[npad + id] = delta_key(p[1]);
}

__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d, 64);
  return v;
}
__device__ __forceinline__ uint64_t wave_or(uint64_t v) {
  return ((uint64_t)wave_or((uint32_t)(v >> 32)) << 32) | wave_or((uint32_t)v);
}
__device__ __forceinline__ int bits_of(uint32_t v) { return 32 - __clz((int)v); }
__device__ __forceinline__ int bits_of(uint64_t v) { return 64 - __clzll((long long)v); }

template <class K> __device__ __forceinline__ void residuals(K cur, K prev, K prev2, K& z0, K& z1) {
  z0 = delta_zigzag((K)(cur - prev));
  z1 = delta_zigzag((K)(cur - (K)(prev + (K)(prev - prev2))));
}

// One wave per block of 64 ids, both coordinates: width byte + payload words of each (block, coordinate).
template <class K>
__global__ __launch_bounds__(256) void delta_widths(int64_t nblk, int64_t npad, const K* cur, const K* prev, const K* prev2,
                                                    uint8_t* widths, uint32_t* words) {
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= nblk) return;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int64_t at = c * npad + blk * 64 + lane;
    K z0, z1;
    residuals(cur[at], prev[at], prev2[at], z0, z1);
    const int w0 = bits_of(wave_or(z0)), w1 = bits_of(wave_or(z1));
    const int pred = w1 < w0 ? 1 : 0;
    const int w = pred ? w1 : w0;
    if (lane == 0) {
      widths[2 * blk + c] = (uint8_t)(w | (pred << 7));
      words[2 * blk + c] = (uint32_t)w;
    }
  }
}

template <class K>
__global__ __launch_bounds__(256) void delta_pack(int64_t nblk, int64_t npad, const K* cur, const K* prev, const K* prev2,
                                                  const uint8_t* widths, const uint32_t* words, const uint32_t* offsets,
                                                  unsigned long long* payload, unsigned long long* total) {
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= nblk) return;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int64_t at = c * npad + blk * 64 + lane;
    const int wb = widths[2 * blk + c];
    const int w = wb & 127;
    K z0, z1;
    residuals(cur[at], prev[at], prev2[at], z0, z1);
    const K z = (wb & 128) ? z1 : z0;
    unsigned long long mine = 0;
    for (int b = 0; b < w; ++b) {  // w is wave-uniform
      const unsigned long long plane = __ballot((z >> b) & 1);
      if (lane == b) mine = plane;
    }
    if (lane < w) payload[(size_t)offsets[2 * blk + c] + lane] = mine;
  }
  if (blk == nblk - 1 && lane == 0) *total = (unsigned long long)offsets[2 * nblk - 1] + words[2 * nblk - 1];
}

}  // namespace

size_t delta_scan_temp_bytes(int64_t n) {
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)(2 * delta_blocks(n)),
                                         (hipStream_t) nullptr);
  return (tb + 255) / 256 * 256;
}

template <class T>
hipError_t launch_delta_encode(hipStream_t s, int64_t n, const void* pos, const uint32_t* ids, void* cur, const void* prev,
                               const void* prev2, uint8_t* widths, uint32_t* words, uint32_t* offsets, void* scan_temp,
                               size_t scan_temp_bytes, uint64_t* payload, uint64_t* total) {
  using K = typename KeyOf<T>::type;
  if (n <= 0) return hipMemsetAsync(total, 0, 8, s);
  const int64_t nblk = (int64_t)delta_blocks(n), npad = nblk * 64;
  hipLaunchKernelGGL((delta_scatter<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, npad, pos, ids, cur);
  const dim3 grid((unsigned)((nblk + 3) / 4));
  hipLaunchKernelGGL((delta_widths<K>), grid, dim3(256), 0, s, nblk, npad, (const K*)cur, (const K*)prev, (const K*)prev2, widths,
                     words);
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_temp_bytes, (const uint32_t*)words, offsets, (int)(2 * nblk), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((delta_pack<K>), grid, dim3(256), 0, s, nblk, npad, (const K*)cur, (const K*)prev, (const K*)prev2, widths,
                     words, offsets, (unsigned long long*)payload, (unsigned long long*)total);
  return hipGetLastError();
}

template hipError_t launch_delta_encode<float>(hipStream_t, int64_t, const void*, const uint32_t*, void*, const void*, const void*,
                                               uint8_t*, uint32_t*, uint32_t*, void*, size_t, uint64_t*, uint64_t*);
template hipError_t launch_delta_encode<double>(hipStream_t, int64_t, const void*, const uint32_t*, void*, const void*, const void*,
                                                uint8_t*, uint32_t*, uint32_t*, void*, size_t, uint64_t*, uint64_t*);

}  // namespace nbody

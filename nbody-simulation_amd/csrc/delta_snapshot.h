// Device encoder of the delta-snapshot stream (delta_snapshot.hip; format in delta_codec.h).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nbody {

size_t delta_scan_temp_bytes(int64_t n);

// cur / prev / prev2: key arrays of 2 * 64 * ceil(n/64) elements (u32 for float, u64 for double): x keys, then y keys,
// in id order; lanes past n must be zero in all three (the call never writes them).  On return (stream order) `cur`
// holds this snapshot's keys, widths[2*nblk] the width bytes, payload[0 .. *total) the words.
template <class T>
hipError_t launch_delta_encode(hipStream_t s, int64_t n, const void* pos, const uint32_t* ids, void* cur, const void* prev,
                               const void* prev2, uint8_t* widths, uint32_t* words, uint32_t* offsets, void* scan_temp,
                               size_t scan_temp_bytes, uint64_t* payload, uint64_t* total);

}  // namespace nbody

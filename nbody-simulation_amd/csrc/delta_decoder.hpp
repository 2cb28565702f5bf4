// Host decoder of the delta-snapshot stream (format: delta_codec.h).  Plain C++ (no HIP), so that the sanitizer
// harness of tests/native/ can compile it with g++; capi.hip wraps it as nbody_delta_decoder_*.  The stream comes from
// outside (a channel, a file): everything is validated before the state is touched.
#pragma once
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "delta_codec.h"

namespace nbody {

struct DeltaDecoder {
  int bits = 0;
  int64_t n = -1;  // -1: no key frame applied yet
  uint64_t step = 0;
  std::vector<uint64_t> prev, prev2;  // keys, x then y, 64 * ceil(n/64) each (u32 keys are stored widened)
  uint64_t max_bodies = 0x7fffffffULL;  // caller-set cap on the body count a header may claim (the stream is untrusted)
  std::string err;

  bool fail(const char* msg) {
    err = msg;
    return false;
  }

  // Applies one stream; false (and `err`) on a malformed, truncated or out-of-sequence one, the state untouched.
  bool apply(const uint8_t* stream, size_t bytes) {
    if (!stream || bytes < kDeltaHeader) return fail("delta stream: shorter than its header");
    if (std::memcmp(stream, "NBD1", 4) != 0) return fail("delta stream: bad magic");
    const int sbits = stream[4];
    const int key = stream[5];
    if ((sbits != 32 && sbits != 64) || key > 1 || stream[6] || stream[7]) return fail("delta stream: bad header");
    uint64_t n64, sstep, total;
    std::memcpy(&n64, stream + 8, 8);
    std::memcpy(&sstep, stream + 16, 8);
    std::memcpy(&total, stream + 24, 8);
    if (n64 > 0x7fffffffULL || n64 > max_bodies) return fail("delta stream: body count out of range");
    const int64_t sn = (int64_t)n64;
    const size_t nblk = delta_blocks(sn), npad = nblk * 64, wb = delta_width_bytes(sn);
    if (total > 2 * nblk * (uint64_t)sbits || bytes != kDeltaHeader + wb + (size_t)total * 8)
      return fail("delta stream: size does not match its header");
    if (!key && (n != sn || bits != sbits))
      return fail(n < 0 ? "delta stream: a delta before any key frame"
                        : "delta stream: another body count or precision than the state");
    const uint8_t* widths = stream + kDeltaHeader;
    uint64_t sum = 0;
    for (size_t i = 0; i < 2 * nblk; ++i) {
      const int w = widths[i] & 127;
      if (w > sbits) return fail("delta stream: a width exceeds the element size");
      sum += (uint64_t)w;
    }
    for (size_t i = 2 * nblk; i < wb; ++i)
      if (widths[i]) return fail("delta stream: non-zero padding");
    if (sum != total) return fail("delta stream: the widths do not add up to the payload");
    // valid from here on.  The new state is built aside and swapped in at the end, so that a failed allocation (a
    // header may claim 2^31 bodies with an all-zero-width payload of a few MB) leaves the decoder as it was.
    std::vector<uint64_t> next, zeros;
    try {
      next.assign(2 * npad, 0);
      if (key) zeros.assign(2 * npad, 0);
    } catch (const std::bad_alloc&) {
      return fail("delta stream: out of memory for the body count its header claims");
    }
    const std::vector<uint64_t>& old1 = key ? zeros : prev;
    const std::vector<uint64_t>& old2 = key ? zeros : prev2;
    const uint64_t mask = sbits == 64 ? ~0ull : 0xFFFFFFFFull;
    const uint8_t* pay = stream + kDeltaHeader + wb;
    for (size_t blk = 0; blk < nblk; ++blk)
      for (int co = 0; co < 2; ++co) {
        const int wbyte = widths[2 * blk + co], w = wbyte & 127;
        uint64_t z[64] = {0};
        for (int b = 0; b < w; ++b) {
          uint64_t plane;
          std::memcpy(&plane, pay, 8);
          pay += 8;
          for (int l = 0; l < 64; ++l) z[l] |= ((plane >> l) & 1ull) << b;
        }
        const size_t base = (size_t)co * npad + blk * 64;
        for (int l = 0; l < 64; ++l) {
          const uint64_t p1 = old1[base + l], p2 = old2[base + l];
          const uint64_t pred = (wbyte & 128) ? (p1 + (p1 - p2)) : p1;
          const uint64_t r = sbits == 64 ? delta_unzigzag(z[l]) : (uint64_t)delta_unzigzag((uint32_t)z[l]);
          next[base + l] = (pred + r) & mask;
        }
      }
    if (key) prev2.swap(zeros); else prev2.swap(prev);
    prev.swap(next);
    n = sn;
    bits = sbits;
    step = sstep;
    err.clear();
    return true;
  }

  // Positions of the current state, x y per body in id order; T must be the stream's precision (K its key type).
  template <class T, class K> bool positions(T* pos) const {
    if (n < 0 || bits != (int)sizeof(T) * 8 || (n > 0 && !pos)) return false;
    const size_t npad = delta_blocks(n) * 64;
    for (int64_t i = 0; i < n; ++i)
      for (int co = 0; co < 2; ++co) {
        const K u = delta_unkey((K)prev[(size_t)co * npad + (size_t)i]);
        std::memcpy(&pos[2 * i + co], &u, sizeof(T));
      }
    return true;
  }
};

}  // namespace nbody

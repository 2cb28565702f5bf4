// A sequential f64 sum, evaluated in parallel without changing one bit of it: exact_sum.h with 53-bit significands.
//
// The f64 BVH (an extension: the reference is f32) folds `sum = sum + p.position` over a node's slice in slice order
// exactly like the f32 one (/root/reference src/bvh_tree.rs:58-61, :67), so its split depends on every rounding of that
// chain too.  The construction is the same: while the running sum stays inside one binade, s = S * ulp with S an integer
// in [2^52, 2^53), one addend is a map {parity of S} -> {integer increment} (the FPU does the rounding: the increments
// are read off fl(C + x) - C for an even and an odd C in the middle of the binade), maps compose associatively, a prefix
// scan yields every intermediate S, and the first addend that takes S out of (2^52, 2^53) — or is too large for the
// binade, or not finite — is added with a real f64 add, after which the scan restarts in the new binade.
// Increments are 64-bit: that is all "53-bit increments (two-word state)" takes.
//
// Host + device: the device fold (bvh_build64.hip) and the CPU emulation the tests run (`nbody_selftest_exact_sum_f64`,
// capi.hip) are the same functions.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NB_HD64 __host__ __device__ __forceinline__
#else
#define NB_HD64 inline
#endif

namespace nbody {
namespace xsum64 {

NB_HD64 uint64_t d2u(double f) {
  uint64_t u;
  memcpy(&u, &f, 8);
  return u;
}
NB_HD64 double u2d(uint64_t u) {
  double f;
  memcpy(&f, &u, 8);
  return f;
}

constexpr uint64_t kLo = 1ull << 52, kHi = 1ull << 53, kMant = (1ull << 52) - 1;
constexpr uint64_t kPoison = 1ull << 62;  // an increment no valid S survives

// Running sum s = (-1)^sign * S * 2^(E - 1075).  Usable iff s is normal (E >= 3 keeps the step's limit normal) and
// S > 2^52 (at S == 2^52 a subtraction would land in the finer binade below without S leaving the range).
struct Chain {
  uint64_t sign, E, S;
};
NB_HD64 bool chain_open(double s, Chain& c) {
  const uint64_t b = d2u(s), e = (b >> 52) & 2047ull;
  c.sign = b >> 63;
  c.E = e;
  c.S = (b & kMant) | kLo;
  return e >= 3ull && e != 2047ull && c.S != kLo;
}
NB_HD64 double chain_value(const Chain& c, uint64_t S) { return u2d((c.sign << 63) | (c.E << 52) | (S & kMant)); }
NB_HD64 bool in_binade(uint64_t S) { return S > kLo && S < kHi; }

// Increment of S caused by adding x, for S even (a0) / odd (a1); wrapping u64 arithmetic.  C0 = 1.5 * 2^E (even),
// C1 = C0 + ulp (odd); with |x| < 2^(E-2) both C + x stay inside the binade, so fl(C + x) - C is x rounded to a multiple of
// ulp with ties going to the even / odd side exactly as they would from any S of that parity.
struct Step {
  uint64_t a0, a1;
};
NB_HD64 Step step_of(double x, uint64_t chain_sign, uint64_t E) {
  const uint64_t c0 = (E << 52) | (1ull << 51);
  const double xs = u2d(d2u(x) ^ (chain_sign << 63));
  const double lim = u2d((E - 2ull) << 52);  // chain_open guarantees E >= 3
  const double ax = u2d(d2u(x) & 0x7fffffffffffffffull);
  if (!(ax < lim)) return {kPoison, kPoison};  // too large for this binade, inf or NaN: a real add decides
  const double r0 = u2d(c0) + xs, r1 = u2d(c0 + 1ull) + xs;
  return {d2u(r0) - c0, d2u(r1) - (c0 + 1ull)};
}
NB_HD64 Step identity() { return {0ull, 0ull}; }
NB_HD64 uint64_t apply(uint64_t S, Step f) { return S + ((S & 1ull) ? f.a1 : f.a0); }
// f first, then g
NB_HD64 Step compose(Step f, Step g) {
  Step h;
  h.a0 = f.a0 + ((f.a0 & 1ull) ? g.a1 : g.a0);
  h.a1 = f.a1 + (((f.a1 + 1ull) & 1ull) ? g.a1 : g.a0);
  return h;
}

// CPU emulation of the device fold's control flow (bvh_build64.hip, fold64: `tile` addends scanned at once, `seq_run` real
// adds after a stop), to check the functions above against the plain loop.  Returns the sum; *stops counts the restarts.
inline double emulate_fold(const double* x, int64_t n, int tile, int seq_run, int64_t* stops) {
  double s = 0.0;
  int64_t pos = 0, nstop = 0;
  while (pos < n) {
    Chain c;
    if (!chain_open(s, c)) {
      const int64_t cnt = (n - pos < seq_run) ? n - pos : seq_run;
      for (int64_t k = 0; k < cnt; ++k) s = s + x[pos + k];
      pos += cnt;
      ++nstop;
      continue;
    }
    const int64_t cnt = (n - pos < tile) ? n - pos : tile;
    Step acc = identity();
    uint64_t S = c.S;
    int64_t bad = -1;
    for (int64_t k = 0; k < cnt; ++k) {
      const Step f = step_of(x[pos + k], c.sign, c.E);
      const uint64_t before = apply(c.S, acc);  // what the scan hands to element k
      const uint64_t after = apply(before, f);
      if (!in_binade(after)) { bad = k; S = before; break; }
      acc = compose(acc, f);
      S = after;
    }
    s = chain_value(c, S);
    if (bad < 0) { pos += cnt; continue; }
    ++nstop;
    pos += bad;
    const int64_t run = (n - pos < seq_run) ? n - pos : seq_run;
    for (int64_t k = 0; k < run; ++k) s = s + x[pos + k];
    pos += run;
  }
  if (stops) *stops = nstop;
  return s;
}

}  // namespace xsum64
}  // namespace nbody

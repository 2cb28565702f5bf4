// A sequential f32 sum, evaluated in parallel without changing one bit of it.
//
// BVHTree::from folds `sum = sum + p.position` over the node's slice in slice order (/root/reference
// src/bvh_tree.rs:58-61) and splits at `sum / len` (:67): the split, hence the whole tree, depends on every rounding of
// that chain.  The chain cannot be re-associated, but it can be scanned:
//
//   while the running sum s stays inside one binade, s = S * ulp with S an integer in [2^23, 2^24), and
//   fl(s + x) = (S + r(x)) * ulp, where r(x) is x/ulp rounded to nearest; the only way the result depends on S is the
//   tie rule (x/ulp exactly half-way: round so that the new S is even), i.e. through the PARITY of S.
//
// So one addend is a map {parity of S} -> {integer increment}: a pair (a0, a1).  Maps compose associatively
// ((f then g)_p = f_p + g_[(p + f_p) & 1]), so a prefix scan of the addends gives every intermediate S exactly.  The scan
// is only valid while every intermediate stays strictly inside (2^23, 2^24); the first addend that leaves the binade
// (or is not finite, or is larger than the binade) is found by the same scan, added with a real f32 add, and the scan
// restarts in the new binade.  For sums of same-signed numbers that happens once per binade (~20 times per node).
//
// Host + device: the device fold (bvh_build.hip) and the CPU emulation used by the tests (`nbody_selftest_exact_sum`,
// capi.hip) run the very same functions.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NB_HD __host__ __device__ __forceinline__
#else
#define NB_HD inline
#endif

namespace nbody {
namespace xsum {

NB_HD uint32_t f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
NB_HD float u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

constexpr uint32_t kLo = 1u << 23, kHi = 1u << 24;
constexpr uint32_t kPoison = 1u << 30;  // an increment no valid S survives

// Running sum s = (-1)^sign * S * 2^(E - 150).  Usable iff s is normal (E >= 3 keeps the step's limit normal) and
// S > 2^23 (at S == 2^23 a subtraction would land in the finer binade below without S leaving the range).
struct Chain {
  uint32_t sign, E, S;
};
NB_HD bool chain_open(float s, Chain& c) {
  const uint32_t b = f2u(s), e = (b >> 23) & 255u;
  c.sign = b >> 31;
  c.E = e;
  c.S = (b & 0x7fffffu) | kLo;
  return e >= 3u && e != 255u && c.S != kLo;
}
NB_HD float chain_value(const Chain& c, uint32_t S) { return u2f((c.sign << 31) | (c.E << 23) | (S & 0x7fffffu)); }
NB_HD bool in_binade(uint32_t S) { return S > kLo && S < kHi; }

// Increment of S caused by adding x, for S even (a0) / odd (a1).  Wrapping u32 arithmetic.
// The FPU does the rounding: with C0 = 1.5 * 2^E (integer part 0xC00000, even) and C1 = C0 + ulp (odd), and
// |x| < 2^(E-2), C + x stays inside the binade, so fl(C + x) - C is x rounded to a multiple of ulp with ties going
// to the even/odd side exactly as they would from any S of the same parity; the difference of the bit patterns is
// that multiple as an integer.
struct Step {
  uint32_t a0, a1;
};
NB_HD Step step_of(float x, uint32_t chain_sign, uint32_t E) {
  const uint32_t c0 = (E << 23) | 0x400000u;
  const float xs = u2f(f2u(x) ^ (chain_sign << 31));
  const float lim = u2f((E - 2u) << 23);  // chain_open guarantees E >= 3
  const float ax = u2f(f2u(x) & 0x7fffffffu);
  if (!(ax < lim)) return {kPoison, kPoison};  // too large for this binade, inf or NaN: a real add decides
  const float r0 = u2f(c0) + xs, r1 = u2f(c0 + 1u) + xs;
  return {f2u(r0) - c0, f2u(r1) - (c0 + 1u)};
}
NB_HD uint32_t apply(uint32_t S, Step f) { return S + ((S & 1u) ? f.a1 : f.a0); }
// f first, then g
NB_HD Step compose(Step f, Step g) {
  Step h;
  h.a0 = f.a0 + ((f.a0 & 1u) ? g.a1 : g.a0);
  h.a1 = f.a1 + (((f.a1 + 1u) & 1u) ? g.a1 : g.a0);
  return h;
}

// ---- whole runs of addends, so that a long chain can be cut into chunks that are prepared in parallel ------------------
// A run seen from a chain in one binade: the total increment of S and the extremes of every intermediate S, relative to
// the S the run starts from, for an even (index 0) / odd (1) start.  If S + lo > 2^23 and S + hi < 2^24 for the actual
// start, every add of the run stayed in the binade and S + a is the exact result.  Saturating at +-2^29: beyond that a
// run is unusable anyway (a poison step is 2^30, saturated on entry).
struct Run {
  int32_t a0, a1, lo0, lo1, hi0, hi1;  // scalars: the compiler turns a select between two array elements into an indexed load from scratch
};
// How close (relatively) to a power of two a predicted prefix may come and still be trusted.  The f32 chain drifts from the
// exact prefix by ~sqrt(n) half-ulps (worst case n): 2^-13 covers millions of addends; if it is ever too tight, the run's
// own bounds fail when it is applied and the scan takes over.  The addends inside the margin are added for real, so the
// margin also is 2 * margin * (addends so far) of serial work per crossing.
constexpr double kRunMargin = 1.0 / 8192.0;
constexpr int32_t kRunSat = 1 << 29;  // two saturated values still add without wrapping
NB_HD int32_t run_sat(int32_t v) { return v > kRunSat ? kRunSat : (v < -kRunSat ? -kRunSat : v); }
NB_HD Run run_of(Step f) {
  Run r;
  r.a0 = r.lo0 = r.hi0 = run_sat((int32_t)f.a0);
  r.a1 = r.lo1 = r.hi1 = run_sat((int32_t)f.a1);
  return r;
}
NB_HD Run run_none() { return Run{0, 0, 0, 0, 0, 0}; }  // an open chain has S inside the binade: offsets 0 fit
// f first, then g.  All values travel as scalars: given `const Run&` the compiler fuses `q ? g.a1 : g.a0` into one load at a
// computed address before it has inlined the call, and the run then lives in scratch memory on the device.
NB_HD void run_then_from(int p, int32_t fa, int32_t flo, int32_t fhi, int32_t ga0, int32_t ga1, int32_t glo0, int32_t glo1, int32_t ghi0, int32_t ghi1,
                   int32_t& ha, int32_t& hlo, int32_t& hhi) {
  const bool q = (((p + fa)) & 1) != 0;
  const int32_t ga = q ? ga1 : ga0, glo = q ? glo1 : glo0, ghi = q ? ghi1 : ghi0;
  ha = run_sat(fa + ga);
  const int32_t gl = run_sat(fa + glo), gh = run_sat(fa + ghi);
  hlo = flo < gl ? flo : gl;
  hhi = fhi > gh ? fhi : gh;
}
NB_HD Run run_then(const Run& f, const Run& g) {
  const int32_t ga0 = g.a0, ga1 = g.a1, glo0 = g.lo0, glo1 = g.lo1, ghi0 = g.hi0, ghi1 = g.hi1;
  Run h;
  run_then_from(0, f.a0, f.lo0, f.hi0, ga0, ga1, glo0, glo1, ghi0, ghi1, h.a0, h.lo0, h.hi0);
  run_then_from(1, f.a1, f.lo1, f.hi1, ga0, ga1, glo0, glo1, ghi0, ghi1, h.a1, h.lo1, h.hi1);
  return h;
}
NB_HD bool run_fits(uint32_t S, const Run& r) {
  const int32_t lo0 = r.lo0, lo1 = r.lo1, hi0 = r.hi0, hi1 = r.hi1;
  const bool p = (S & 1u) != 0u;
  return (int64_t)S + (p ? lo1 : lo0) > (int64_t)kLo && (int64_t)S + (p ? hi1 : hi0) < (int64_t)kHi;
}

// CPU emulation of the device fold's control flow (tile = `tile` addends scanned at once, `seq_run` real adds after a
// stop), used to check the functions above against the plain loop.  Returns the sum; *stops counts the restarts.
inline float emulate_fold(const float* x, int64_t n, int tile, int seq_run, int64_t* stops) {
  float s = 0.0f;
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
    // "scan": prefix compositions, then every element checks its own intermediate
    Step acc{0u, 0u};
    uint32_t S = c.S;
    int64_t bad = -1;
    for (int64_t k = 0; k < cnt; ++k) {
      const Step f = step_of(x[pos + k], c.sign, c.E);
      const uint32_t before = apply(c.S, acc);  // what the scan hands to element k
      const uint32_t after = apply(before, f);
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

// CPU emulation of the chunked fold (bvh_build.hip: bvh_chunk_sums / bvh_chunk_runs / the chunk walk of bvh_big_fold):
// every chunk's run is prepared for the binade its start is PREDICTED to be in (from exact f64 partial sums); the walk
// uses a run only if the prediction and the bounds hold for the true state, and adds the chunk for real otherwise.
inline float emulate_fold_chunked(const float* x, int64_t n, int chunk, int64_t* used_runs) {
  float s = 0.0f;
  double prefix = 0.0;
  int64_t used = 0;
  for (int64_t c0 = 0; c0 < n; c0 += chunk) {
    const int64_t c1 = c0 + chunk < n ? c0 + chunk : n;
    Chain pred;
    bool have = c0 > 0 && chain_open((float)prefix, pred);
    Run r = run_none();
    if (have)
      for (int64_t k = c0; k < c1; ++k) r = run_then(r, run_of(step_of(x[k], pred.sign, pred.E)));
    Chain cur;
    if (have && chain_open(s, cur) && cur.E == pred.E && cur.sign == pred.sign && run_fits(cur.S, r)) {
      s = chain_value(cur, (uint32_t)((int64_t)cur.S + ((cur.S & 1u) ? r.a1 : r.a0)));
      ++used;
    } else {
      for (int64_t k = c0; k < c1; ++k) s = s + x[k];
    }
    for (int64_t k = c0; k < c1; ++k) prefix += (double)x[k];
  }
  if (used_runs) *used_runs = used;
  return s;
}

// The same with the chunk that contains a crossing split three ways (bvh_chunk_runs' second form): thread segments of
// `seg` addends whose predicted prefix stays below (1 - kRunMargin) of the power of two form run A (old binade), those above
// (1 + kRunMargin) of it run B (new binade), the segments in between are added for real.
inline float emulate_fold_chunked2(const float* x, int64_t n, int chunk, int seg, int64_t* used_runs) {
  float s = 0.0f;
  double prefix = 0.0;
  int64_t used = 0;
  auto real = [&](int64_t b, int64_t e) { for (int64_t k = b; k < e; ++k) s = s + x[k]; };
  auto take = [&](const Run& r, uint32_t sign, uint32_t E, int64_t b, int64_t e) {
    Chain cur;
    if (e > b && chain_open(s, cur) && cur.E == E && cur.sign == sign && run_fits(cur.S, r)) {
      s = chain_value(cur, (uint32_t)((int64_t)cur.S + ((cur.S & 1u) ? r.a1 : r.a0)));
      ++used;
    } else {
      real(b, e);
    }
  };
  for (int64_t c0 = 0; c0 < n; c0 += chunk) {
    const int64_t c1 = c0 + chunk < n ? c0 + chunk : n;
    double total = 0.0;
    for (int64_t k = c0; k < c1; ++k) total += (double)x[k];
    Chain ca, cb;
    const bool have = c0 > 0 && chain_open((float)prefix, ca) && chain_open((float)(prefix + total), cb) && ca.sign == cb.sign &&
                      (cb.E == ca.E || cb.E == ca.E + 1);
    if (!have) {
      real(c0, c1);
    } else {
      const int64_t nseg = (c1 - c0 + seg - 1) / seg;
      if (nseg > (1 << 16)) { real(c0, c1); prefix += total; continue; }  // more segments than the scratch below holds
      const double sgn = ca.sign ? -1.0 : 1.0;
      double B = 1.0;
      for (int e = 127; e < (int)cb.E; ++e) B *= 2.0;
      for (int e = 127; e > (int)cb.E; --e) B *= 0.5;
      const double lo = B * (1.0 - kRunMargin), hi = B * (1.0 + kRunMargin);
      int64_t nA = 0, nB = 0;
      bool contiguous = true;
      {
        double run = prefix;
        // pass 1: classify
        static thread_local int cls[1 << 16];
        for (int64_t t = 0; t < nseg; ++t) {
          const int64_t b = c0 + t * seg, e = b + seg < c1 ? b + seg : c1;
          const double qs = sgn * run;
          for (int64_t k = b; k < e; ++k) run += (double)x[k];
          const double qe = sgn * run;
          int cl = 2;  // zone
          if (cb.E == ca.E) cl = 0;
          else if (qs < lo && qe < lo) cl = 0;
          else if (qs > hi && qe > hi) cl = 1;
          cls[t] = cl;
          nA += cl == 0;
          nB += cl == 1;
        }
        for (int64_t t = 0; t < nseg; ++t) {
          if (cls[t] == 0 && t >= nA) contiguous = false;
          if (cls[t] == 1 && t < nseg - nB) contiguous = false;
        }
      }
      if (!contiguous) {
        real(c0, c1);
      } else {
        const int64_t u0 = c0 + (nA * seg < c1 - c0 ? nA * seg : c1 - c0);
        const int64_t u1 = c0 + ((nseg - nB) * seg < c1 - c0 ? (nseg - nB) * seg : c1 - c0);
        Run ra = run_none(), rb = run_none();
        for (int64_t k = c0; k < u0; ++k) ra = run_then(ra, run_of(step_of(x[k], ca.sign, ca.E)));
        for (int64_t k = u1; k < c1; ++k) rb = run_then(rb, run_of(step_of(x[k], cb.sign, cb.E)));
        take(ra, ca.sign, ca.E, c0, u0);
        real(u0, u1);
        take(rb, cb.sign, cb.E, u1, c1);
      }
    }
    prefix += total;
  }
  if (used_runs) *used_runs = used;
  return s;
}

}  // namespace xsum
}  // namespace nbody

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

// ---- whole runs of addends, so that a long chain can be cut into segments that are prepared in parallel ----------------
// (exact_sum.h's Run with 64-bit words.)  A run seen from a chain in one binade: the total increment of S and the extremes
// of every intermediate S, relative to the S the run starts from, for an even (index 0) / odd (1) start.  If
// S + lo > 2^52 and S + hi < 2^53 for the actual start, every add of the run stayed in the binade and S + a is the exact
// result.  Saturating at +-2^60: beyond that a run is unusable anyway (a poison step is 2^62, saturated on entry).
struct Run {
  int64_t a0, a1, lo0, lo1, hi0, hi1;  // scalars, as in exact_sum.h
};
constexpr int64_t kRunSat = 1ll << 60;  // two saturated values still add without wrapping
NB_HD64 int64_t run_sat(int64_t v) { return v > kRunSat ? kRunSat : (v < -kRunSat ? -kRunSat : v); }
NB_HD64 Run run_of(Step f) {
  Run r;
  r.a0 = r.lo0 = r.hi0 = run_sat((int64_t)f.a0);
  r.a1 = r.lo1 = r.hi1 = run_sat((int64_t)f.a1);
  return r;
}
NB_HD64 Run run_none() { return Run{0, 0, 0, 0, 0, 0}; }
// f first, then g.  All values travel as scalars: given `const Run&` the compiler fuses `q ? g.a1 : g.a0` into one load at a
// computed address before it has inlined the call, and the run then lives in scratch memory on the device.
NB_HD64 void run_then_from(int p, int64_t fa, int64_t flo, int64_t fhi, int64_t ga0, int64_t ga1, int64_t glo0, int64_t glo1, int64_t ghi0, int64_t ghi1,
                   int64_t& ha, int64_t& hlo, int64_t& hhi) {
  const bool q = (((int)(p + fa)) & 1) != 0;
  const int64_t ga = q ? ga1 : ga0, glo = q ? glo1 : glo0, ghi = q ? ghi1 : ghi0;
  ha = run_sat(fa + ga);
  const int64_t gl = run_sat(fa + glo), gh = run_sat(fa + ghi);
  hlo = flo < gl ? flo : gl;
  hhi = fhi > gh ? fhi : gh;
}
NB_HD64 Run run_then(const Run& f, const Run& g) {
  const int64_t ga0 = g.a0, ga1 = g.a1, glo0 = g.lo0, glo1 = g.lo1, ghi0 = g.hi0, ghi1 = g.hi1;
  Run h;
  run_then_from(0, f.a0, f.lo0, f.hi0, ga0, ga1, glo0, glo1, ghi0, ghi1, h.a0, h.lo0, h.hi0);
  run_then_from(1, f.a1, f.lo1, f.hi1, ga0, ga1, glo0, glo1, ghi0, ghi1, h.a1, h.lo1, h.hi1);
  return h;
}
NB_HD64 bool run_fits(uint64_t S, const Run& r) {
  const int64_t lo0 = r.lo0, lo1 = r.lo1, hi0 = r.hi0, hi1 = r.hi1;
  const bool p = (S & 1ull) != 0ull;
  return (int64_t)S + (p ? lo1 : lo0) > (int64_t)kLo && (int64_t)S + (p ? hi1 : hi0) < (int64_t)kHi;
}
// The binade a chain is predicted to be in over a segment whose exact prefix sums start at `p0` and end at `p1` (any f64
// evaluation of them: the prediction only has to be right often, the run's own bounds decide): both ends inside one binade
// and at least 2^-20 (relatively) away from its edges.  false: no run is prepared for the segment.
NB_HD64 bool predict_binade(double p0, double p1, Chain& c) {
  Chain c1;
  if (!chain_open(p0, c) || !chain_open(p1, c1) || c.E != c1.E || c.sign != c1.sign) return false;
  const uint64_t margin = 1ull << 32;  // 2^-20 of the binade's 2^52 steps
  return c.S > kLo + margin && c.S < kHi - margin && c1.S > kLo + margin && c1.S < kHi - margin;
}

// CPU emulation of the segmented fold (bvh_build64.hip: b64_seg_sums / b64_seg_runs / the segment test of b64_fold): every
// `seg`-long segment's run is prepared for the binade its ends are PREDICTED to be in (from plain f64 partial sums); the
// walk uses a run only if the prediction and the run's bounds hold for the true state, and scans the segment otherwise.
inline double emulate_fold_segmented(const double* x, int64_t n, int seg, int64_t* used_runs) {
  double s = 0.0, prefix = 0.0;
  int64_t used = 0;
  for (int64_t c0 = 0; c0 < n; c0 += seg) {
    const int64_t c1 = c0 + seg < n ? c0 + seg : n;
    double total = 0.0;
    for (int64_t k = c0; k < c1; ++k) total += x[k];
    Chain pred;
    const bool have = c1 - c0 == seg && predict_binade(prefix, prefix + total, pred);
    Run r = run_none();
    if (have)
      for (int64_t k = c0; k < c1; ++k) r = run_then(r, run_of(step_of(x[k], pred.sign, pred.E)));
    Chain cur;
    if (have && chain_open(s, cur) && cur.E == pred.E && cur.sign == pred.sign && run_fits(cur.S, r)) {
      s = chain_value(cur, (uint64_t)((int64_t)cur.S + ((cur.S & 1ull) ? r.a1 : r.a0)));
      ++used;
    } else {
      for (int64_t k = c0; k < c1; ++k) s = s + x[k];
    }
    prefix += total;
  }
  if (used_runs) *used_runs = used;
  return s;
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

// Two IEEE f32 divisions by the same denominator: the compiler's expansion twice, against one f64 reciprocal refined to
// ~2^-52 and the two quotients rounded from f64 (a quotient of two 24-bit numbers is either a rounding midpoint — never, for
// a normal result — or at least 2^-49 (relative) away from one: an error of 2^-51 cannot change the rounding).  Results that
// are not normal-or-zero (denormal, inf, NaN: every special input ends up there) take the compiler's division.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div32_microbench.hip -o tools/div32_microbench && tools/div32_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>
__device__ __forceinline__ void div2_via_f64(float nx, float ny, float d, float& qx, float& qy) {
  const double dd = (double)d;
  double r = __builtin_amdgcn_rcp(dd);
  double e = __builtin_fma(-dd, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-dd, r, 1.0);
  r = __builtin_fma(r, e, r);
  float fx = (float)((double)nx * r), fy = (float)((double)ny * r);
  // normal or zero: class bits 1 (-normal), 5 (-0), 6 (+0), 8 (+normal) -- gfx9 numbering: 0 sNaN 1 qNaN 2 -inf 3 -normal 4 -denorm 5 -0 6 +0 7 +denorm 8 +normal 9 +inf
  const bool okx = __builtin_amdgcn_class(fx, (1 << 3) | (1 << 5) | (1 << 6) | (1 << 8));
  const bool oky = __builtin_amdgcn_class(fy, (1 << 3) | (1 << 5) | (1 << 6) | (1 << 8));
  if (!(okx && oky)) { fx = nx / d; fy = ny / d; }
  qx = fx; qy = fy;
}
template <int MODE>
__global__ void k(const float* nx, const float* ny, const float* d, float* qx, float* qy, int n, int reps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = nx[i], b = ny[i], c = d[i], sx = 0, sy = 0;
  for (int r = 0; r < reps; ++r) {
    float x, y;
    if (MODE == 0) { x = a / c; y = b / c; } else div2_via_f64(a, b, c, x, y);
    sx += x; sy += y;
    a = a * 1.0000001f; b = b * 0.9999999f;   // keep the compiler from hoisting
  }
  qx[i] = sx; qy[i] = sy;
}
template <int MODE>
__global__ void once(const float* nx, const float* ny, const float* d, float* qx, float* qy, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (MODE == 0) { qx[i] = nx[i] / d[i]; qy[i] = ny[i] / d[i]; } else div2_via_f64(nx[i], ny[i], d[i], qx[i], qy[i]);
}
int main(int argc, char** argv) {
  const int n = 1 << 24;
  const int rounds = argc > 1 ? atoi(argv[1]) : 8;
  std::vector<float> hx(n), hy(n), hd(n), a(n), b(n), c(n), e(n);
  float *x, *y, *d, *q0x, *q0y, *q1x, *q1y;
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&d, n * 4); hipMalloc(&q0x, n * 4); hipMalloc(&q0y, n * 4); hipMalloc(&q1x, n * 4); hipMalloc(&q1y, n * 4);
  long bad = 0, total = 0;
  for (int round = 0; round < rounds; ++round) {
    std::mt19937_64 rng(5 + round);
    for (int i = 0; i < n; ++i) {
      auto rnd = [&](int span, int center) { unsigned m = (unsigned)rng() & 0x7fffffu; int ex = center + (int)(rng() % (2 * span + 1)) - span; ex = ex < 0 ? 0 : (ex > 255 ? 255 : ex);
                                             unsigned bb = ((unsigned)(rng() & 1) << 31) | ((unsigned)ex << 23) | m; float v; memcpy(&v, &bb, 4); return v; };
      const int kind = i & 7;
      if (kind < 3) { hx[i] = rnd(20, 127); hy[i] = rnd(20, 127); hd[i] = rnd(20, 127); }             // ordinary magnitudes
      else if (kind == 3) { hx[i] = rnd(127, 127); hy[i] = rnd(127, 127); hd[i] = rnd(127, 127); }     // everything, specials included
      else if (kind == 4) { hx[i] = rnd(8, 10); hy[i] = rnd(8, 120); hd[i] = rnd(8, 135); }          // denormal quotients
      else if (kind == 5) { hx[i] = rnd(8, 245); hy[i] = rnd(8, 130); hd[i] = rnd(8, 8); }            // overflowing quotients, tiny denominators
      else if (kind == 6) {  // quotients that are exact or one bit from a tie: short significands
        unsigned mq = ((unsigned)rng() & 0xfffu) << 11, md = ((unsigned)rng() & 0x7ffu) << 12;
        unsigned qb = (127u << 23) | mq, db = ((unsigned)(100 + rng() % 50) << 23) | md; float q, dd; memcpy(&q, &qb, 4); memcpy(&dd, &db, 4);
        hd[i] = dd; hx[i] = q * dd; hy[i] = -hx[i] * 0.5f;
      } else { hx[i] = rnd(2, 127); hy[i] = (float)(int)(rng() % 1000) - 500.f; hd[i] = (float)(1 + rng() % 4096); }
      if (i % 1001 == 0) hx[i] = 0.0f;
      if (i % 1003 == 0) hy[i] = -0.0f;
      if (i % 100003 == 0) hd[i] = 0.0f;
    }
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(y, hy.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(d, hd.data(), n * 4, hipMemcpyHostToDevice);
    once<0><<<n / 256, 256>>>(x, y, d, q0x, q0y, n); once<1><<<n / 256, 256>>>(x, y, d, q1x, q1y, n);
    hipMemcpy(a.data(), q0x, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), q0y, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), q1x, n * 4, hipMemcpyDeviceToHost); hipMemcpy(e.data(), q1y, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
      if (memcmp(&a[i], &c[i], 4)) { if (++bad <= 5) printf("x mismatch: %a / %a -> %a vs %a\n", hx[i], hd[i], a[i], c[i]); }
      if (memcmp(&b[i], &e[i], 4)) { if (++bad <= 5) printf("y mismatch: %a / %a -> %a vs %a\n", hy[i], hd[i], b[i], e[i]); }
    }
    total += 2L * n;
  }
  printf("bitwise mismatches (NaN payloads included): %ld of %ld quotients\n", bad, total);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<4096, 256>>>(x, y, d, q0x, q0y, n, 200); else k<1><<<4096, 256>>>(x, y, d, q1x, q1y, n, 200);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s: %.3f ms for %d x 200 division pairs\n", mode ? "one f64 reciprocal " : "compiler, twice    ", ms, 4096 * 256);
  }
  return bad != 0;
}

// VALU issue-rate microbenchmark for gfx950: what the chip sustains per instruction kind, wall-clock based.
// Decides the design of the direct O(N^2) kernel (packed vs scalar f32, cost of v_rcp_f32, operand kinds).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

#define REP8(X) X X X X X X X X
#define ITERS 8000

#define KERNEL(NAME, BODY)                                                                       \
  __global__ __launch_bounds__(256) void NAME(float* out, float seed, float one, float small) {  \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5,  \
          a6 = seed + 6, a7 = seed + 7;                                                          \
    a0 += threadIdx.x * 0.37f; a1 -= threadIdx.x * 0.11f; a2 += threadIdx.x * 1.3f; a3 -= threadIdx.x * 0.7f; \
    float b0 = one * 1.0001f, b1 = small;                                                        \
    float __attribute__((ext_vector_type(2))) p0 = {a0, a1}, p1 = {a2, a3}, p2 = p0 + 2.f, p3 = p0 + 3.f, \
        p4 = p1 + 4.f, p5 = p1 + 5.f, p6 = p0 + 6.f, p7 = p1 + 7.f, q = {one * 1.0001f, one * 0.9999f}, r = {small, -small}; \
    for (int i = 0; i < ITERS; ++i) {                                                            \
      asm volatile(REP8(BODY)                                                                    \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                     "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)  \
                   : "v"(b0), "v"(b1), "v"(q), "s"(seed), "v"(r), "s"(one));                     \
    }                                                                                            \
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y; \
    if (s == 123.456f) out[0] = 1;                                                               \
  }

// operands: %0..%7 = a0..a7 (vgpr), %8..%15 = p0..p7 (vgpr pairs), %16 = b0 (~1.0001), %17 = b1 (small),
//           %18 = q (pair ~1), %19 = seed (sgpr), %20 = r (pair small), %21 = one (sgpr)
#define I8(OP, TAIL) OP " %0, " TAIL "\n" OP " %1, " TAIL "\n" OP " %2, " TAIL "\n" OP " %3, " TAIL "\n" \
                     OP " %4, " TAIL "\n" OP " %5, " TAIL "\n" OP " %6, " TAIL "\n" OP " %7, " TAIL "\n"
KERNEL(k_fma_vvv, "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                  "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n")
KERNEL(k_fmac, "v_fmac_f32 %0, %16, %17\n v_fmac_f32 %1, %16, %17\n v_fmac_f32 %2, %16, %17\n v_fmac_f32 %3, %16, %17\n"
               "v_fmac_f32 %4, %16, %17\n v_fmac_f32 %5, %16, %17\n v_fmac_f32 %6, %16, %17\n v_fmac_f32 %7, %16, %17\n")
KERNEL(k_mul_vv, "v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n"
                 "v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
KERNEL(k_mul_sv, "v_mul_f32 %0, %21, %0\n v_mul_f32 %1, %21, %1\n v_mul_f32 %2, %21, %2\n v_mul_f32 %3, %21, %3\n"
                 "v_mul_f32 %4, %21, %4\n v_mul_f32 %5, %21, %5\n v_mul_f32 %6, %21, %6\n v_mul_f32 %7, %21, %7\n")
KERNEL(k_add_vv, "v_add_f32 %0, %0, %17\n v_add_f32 %1, %1, %17\n v_add_f32 %2, %2, %17\n v_add_f32 %3, %3, %17\n"
                 "v_add_f32 %4, %4, %17\n v_add_f32 %5, %5, %17\n v_add_f32 %6, %6, %17\n v_add_f32 %7, %7, %17\n")
KERNEL(k_add_abs, "v_add_f32 %0, |%0|, |%17|\n v_add_f32 %1, |%1|, |%17|\n v_add_f32 %2, |%2|, |%17|\n v_add_f32 %3, |%3|, |%17|\n"
                  "v_add_f32 %4, |%4|, |%17|\n v_add_f32 %5, |%5|, |%17|\n v_add_f32 %6, |%6|, |%17|\n v_add_f32 %7, |%7|, |%17|\n")
KERNEL(k_sub_vv, "v_sub_f32 %0, %0, %17\n v_sub_f32 %1, %1, %17\n v_sub_f32 %2, %2, %17\n v_sub_f32 %3, %3, %17\n"
                 "v_sub_f32 %4, %4, %17\n v_sub_f32 %5, %5, %17\n v_sub_f32 %6, %6, %17\n v_sub_f32 %7, %7, %17\n")
KERNEL(k_sub_sv, "v_sub_f32 %0, %19, %0\n v_sub_f32 %1, %19, %1\n v_sub_f32 %2, %19, %2\n v_sub_f32 %3, %19, %3\n"
                 "v_sub_f32 %4, %19, %4\n v_sub_f32 %5, %19, %5\n v_sub_f32 %6, %19, %6\n v_sub_f32 %7, %19, %7\n")
KERNEL(k_subrev_sv, "v_subrev_f32 %0, %19, %0\n v_subrev_f32 %1, %19, %1\n v_subrev_f32 %2, %19, %2\n v_subrev_f32 %3, %19, %3\n"
                    "v_subrev_f32 %4, %19, %4\n v_subrev_f32 %5, %19, %5\n v_subrev_f32 %6, %19, %6\n v_subrev_f32 %7, %19, %7\n")
KERNEL(k_max_vv, "v_max_f32 %0, %0, %16\n v_max_f32 %1, %1, %16\n v_max_f32 %2, %2, %16\n v_max_f32 %3, %3, %16\n"
                 "v_max_f32 %4, %4, %16\n v_max_f32 %5, %5, %16\n v_max_f32 %6, %6, %16\n v_max_f32 %7, %7, %16\n")
KERNEL(k_max_chg, "v_max_f32 %0, %1, %16\n v_max_f32 %1, %2, %16\n v_max_f32 %2, %3, %16\n v_max_f32 %3, %4, %16\n"
                  "v_max_f32 %4, %5, %16\n v_max_f32 %5, %6, %16\n v_max_f32 %6, %7, %16\n v_max_f32 %7, %0, %17\n")
KERNEL(k_fmaak, "v_fmaak_f32 %0, %0, %16, 0x12800000\n v_fmaak_f32 %1, %1, %16, 0x12800000\n v_fmaak_f32 %2, %2, %16, 0x12800000\n v_fmaak_f32 %3, %3, %16, 0x12800000\n"
                "v_fmaak_f32 %4, %4, %16, 0x12800000\n v_fmaak_f32 %5, %5, %16, 0x12800000\n v_fmaak_f32 %6, %6, %16, 0x12800000\n v_fmaak_f32 %7, %7, %16, 0x12800000\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
KERNEL(k_rsq, "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
              "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n")
KERNEL(k_pk_fma, "v_pk_fma_f32 %8, %8, %18, %20\n v_pk_fma_f32 %9, %9, %18, %20\n v_pk_fma_f32 %10, %10, %18, %20\n v_pk_fma_f32 %11, %11, %18, %20\n"
                 "v_pk_fma_f32 %12, %12, %18, %20\n v_pk_fma_f32 %13, %13, %18, %20\n v_pk_fma_f32 %14, %14, %18, %20\n v_pk_fma_f32 %15, %15, %18, %20\n")
KERNEL(k_pk_mul, "v_pk_mul_f32 %8, %8, %18\n v_pk_mul_f32 %9, %9, %18\n v_pk_mul_f32 %10, %10, %18\n v_pk_mul_f32 %11, %11, %18\n"
                 "v_pk_mul_f32 %12, %12, %18\n v_pk_mul_f32 %13, %13, %18\n v_pk_mul_f32 %14, %14, %18\n v_pk_mul_f32 %15, %15, %18\n")
KERNEL(k_pk_add, "v_pk_add_f32 %8, %8, %20\n v_pk_add_f32 %9, %9, %20\n v_pk_add_f32 %10, %10, %20\n v_pk_add_f32 %11, %11, %20\n"
                 "v_pk_add_f32 %12, %12, %20\n v_pk_add_f32 %13, %13, %20\n v_pk_add_f32 %14, %14, %20\n v_pk_add_f32 %15, %15, %20\n")
KERNEL(k_mix_rcp7, "v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                   "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n")
KERNEL(k_nop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
KERNEL(k_cmpclass, "v_cmp_class_f32 vcc, %0, %16\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_class_f32 vcc, %2, %16\n v_cndmask_b32 %2, %2, %3, vcc\n"
                   "v_cmp_class_f32 vcc, %4, %16\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_class_f32 vcc, %6, %16\n v_cndmask_b32 %6, %6, %7, vcc\n")

typedef void (*kfn)(float*, float, float, float);
struct Case { const char* name; kfn fn; double flop_per_instr; };

int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  const int cus = prop.multiProcessorCount;
  float* out; (void)hipMalloc(&out, 1024);
  std::vector<Case> cases = {
    {"v_fma_f32 v,v,v", k_fma_vvv, 2}, {"v_fmac_f32", k_fmac, 2}, {"v_mul_f32 v,v", k_mul_vv, 1}, {"v_mul_f32 s,v", k_mul_sv, 1},
    {"v_add_f32 v,v", k_add_vv, 1}, {"v_add_f32 |v|,|v|", k_add_abs, 1}, {"v_sub_f32 v,v", k_sub_vv, 1}, {"v_sub_f32 s,v", k_sub_sv, 1},
    {"v_subrev_f32 s,v", k_subrev_sv, 1}, {"v_max_f32 (const data)", k_max_vv, 1}, {"v_max_f32 (moving data)", k_max_chg, 1},
    {"v_fmaak_f32", k_fmaak, 2}, {"v_rcp_f32", k_rcp, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_pk_fma_f32", k_pk_fma, 4}, {"v_pk_mul_f32", k_pk_mul, 2},
    {"v_pk_add_f32", k_pk_add, 2}, {"1 rcp + 7 fma", k_mix_rcp7, 2}, {"s_nop 0", k_nop, 0}, {"v_mov_b32", k_mov, 0},
    {"cmp_class+cndmask", k_cmpclass, 0}};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("ns per wave-instruction per SIMD (wall clock), by waves per SIMD; last col: TFLOP/s at w=8\n");
  printf("%-26s %8s %8s %8s %8s %10s\n", "instr", "w=1", "w=2", "w=4", "w=8", "TF(w=8)");
  for (auto& c : cases) {
    printf("%-26s", c.name);
    double tf = 0;
    for (int w : {1, 2, 4, 8}) {
      int blocks = cus * w;
      hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 1.0f, 1e-3f);
      (void)hipDeviceSynchronize();
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 1.0f, 1e-3f);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
      }
      double instr_per_simd = (double)ITERS * 64 * w;  // one wave per SIMD per block
      printf(" %8.3f", best * 1e6 / instr_per_simd);
      if (w == 8) tf = (double)blocks * 4 * ITERS * 64 * 64 * c.flop_per_instr / (best * 1e-3) / 1e12;
    }
    printf(" %10.1f\n", tf);
  }
  return 0;
}

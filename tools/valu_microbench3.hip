// Second VALU microbenchmark: explicit registers, to learn gfx950's operand-fetch rules (VGPR banks, SGPR /
// constant-bus operands, 64-bit packed operands) and the cost of clamp alternatives.  Wall-clock, w=8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 4000
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","vcc"
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, NOIEEE, BODY)                                                                     \
  __global__ __launch_bounds__(256) void NAME(float* out, float seed, float one, float small) { \
    float t = seed + threadIdx.x * 0.37f;                                                       \
    asm volatile("v_mov_b32 v0, %0\n v_add_f32 v1, 1.0, v0\n v_add_f32 v2, 2.0, v0\n v_add_f32 v3, 4.0, v0\n" \
                 "v_add_f32 v4, 0.5, v0\n v_add_f32 v5, 1.0, v1\n v_add_f32 v6, 1.0, v2\n v_add_f32 v7, 1.0, v3\n" \
                 "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %2\n v_mov_b32 v11, %2\n v_mov_b32 v12, %1\n v_mov_b32 v13, %2\n v_mov_b32 v14, %1\n v_mov_b32 v15, %2\n" \
                 "v_mov_b32 v16, v0\n v_mov_b32 v17, v1\n v_mov_b32 v18, v2\n v_mov_b32 v19, v3\n v_mov_b32 v20, v4\n v_mov_b32 v21, v5\n v_mov_b32 v22, v6\n v_mov_b32 v23, v7\n" \
                 "v_mov_b32 v24, %1\n v_mov_b32 v25, %1\n v_mov_b32 v26, %2\n v_mov_b32 v27, %2\n v_mov_b32 v28, %1\n v_mov_b32 v29, %1\n v_mov_b32 v30, %2\n v_mov_b32 v31, %2\n" \
                 :: "v"(t), "v"(one * 1.0001f), "v"(small) : CLOB);                             \
    asm volatile("s_mov_b32 s4, %0\n s_mov_b32 s5, %1\n s_mov_b32 s6, %0\n" :: "s"(one), "s"(seed) : "s4","s5","s6");\
    if (NOIEEE) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");\
    for (int i = 0; i < ITERS; ++i) { asm volatile(REP8(BODY) ::: CLOB, "s4","s5","s6"); } \
    float s;                                                                                    \
    asm volatile("v_add_f32 %0, v0, v1\n v_add_f32 %0, %0, v2\n v_add_f32 %0, %0, v3\n v_add_f32 %0, %0, v4\n v_add_f32 %0, %0, v5\n v_add_f32 %0, %0, v6\n v_add_f32 %0, %0, v7\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v20\n v_add_f32 %0, %0, v22\n" : "=v"(s) :: CLOB); \
    if (s == 123.456f) out[0] = 1;                                                              \
  }
KERNEL(k_S_general, 0, "v_pk_add_f32 v[20:21], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v28, v20, v20\n v_fmac_f32 v28, v21, v21\n v_add_f32 v0, |v20|, |v21|\n v_max_f32 v28, v28, v10\n v_fmaak_f32 v0, v0, v28, 0x12800000\n v_rcp_f32 v0, v0\n v_mul_f32 v0, s6, v0\n v_pk_fma_f32 v[16:17], v[20:21], v[0:1], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[22:23], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v29, v22, v22\n v_fmac_f32 v29, v23, v23\n v_add_f32 v2, |v22|, |v23|\n v_max_f32 v29, v29, v10\n v_fmaak_f32 v2, v2, v29, 0x12800000\n v_rcp_f32 v2, v2\n v_mul_f32 v2, s6, v2\n v_pk_fma_f32 v[16:17], v[22:23], v[2:3], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[24:25], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v30, v24, v24\n v_fmac_f32 v30, v25, v25\n v_add_f32 v4, |v24|, |v25|\n v_max_f32 v30, v30, v10\n v_fmaak_f32 v4, v4, v30, 0x12800000\n v_rcp_f32 v4, v4\n v_mul_f32 v4, s6, v4\n v_pk_fma_f32 v[16:17], v[24:25], v[4:5], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[26:27], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v31, v26, v26\n v_fmac_f32 v31, v27, v27\n v_add_f32 v6, |v26|, |v27|\n v_max_f32 v31, v31, v10\n v_fmaak_f32 v6, v6, v31, 0x12800000\n v_rcp_f32 v6, v6\n v_mul_f32 v6, s6, v6\n v_pk_fma_f32 v[16:17], v[26:27], v[6:7], v[16:17] op_sel_hi:[1,0,1]\n")
KERNEL(k_S_uniform, 0, "v_pk_add_f32 v[20:21], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v28, v20, v20\n v_fmac_f32 v28, v21, v21\n v_add_f32 v0, |v20|, |v21|\n v_max_f32 v28, v28, v10\n v_fmaak_f32 v0, v0, v28, 0x12800000\n v_rcp_f32 v0, v0\n v_pk_fma_f32 v[16:17], v[20:21], v[0:1], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[22:23], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v29, v22, v22\n v_fmac_f32 v29, v23, v23\n v_add_f32 v2, |v22|, |v23|\n v_max_f32 v29, v29, v10\n v_fmaak_f32 v2, v2, v29, 0x12800000\n v_rcp_f32 v2, v2\n v_pk_fma_f32 v[16:17], v[22:23], v[2:3], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[24:25], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v30, v24, v24\n v_fmac_f32 v30, v25, v25\n v_add_f32 v4, |v24|, |v25|\n v_max_f32 v30, v30, v10\n v_fmaak_f32 v4, v4, v30, 0x12800000\n v_rcp_f32 v4, v4\n v_pk_fma_f32 v[16:17], v[24:25], v[4:5], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[26:27], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v31, v26, v26\n v_fmac_f32 v31, v27, v27\n v_add_f32 v6, |v26|, |v27|\n v_max_f32 v31, v31, v10\n v_fmaak_f32 v6, v6, v31, 0x12800000\n v_rcp_f32 v6, v6\n v_pk_fma_f32 v[16:17], v[26:27], v[6:7], v[16:17] op_sel_hi:[1,0,1]\n")
KERNEL(k_L_general, 0, "v_sub_f32 v20, v12, v8\n v_sub_f32 v21, v13, v9\n v_mul_f32 v28, v20, v20\n v_fmac_f32 v28, v21, v21\n v_add_f32 v0, |v20|, |v21|\n v_max_f32 v28, v28, v10\n v_fmaak_f32 v0, v0, v28, 0x12800000\n v_rcp_f32 v0, v0\n v_mul_f32 v0, v14, v0\n v_pk_fma_f32 v[16:17], v[20:21], v[0:1], v[16:17] op_sel_hi:[1,0,1]\n v_sub_f32 v22, v12, v8\n v_sub_f32 v23, v13, v9\n v_mul_f32 v29, v22, v22\n v_fmac_f32 v29, v23, v23\n v_add_f32 v2, |v22|, |v23|\n v_max_f32 v29, v29, v10\n v_fmaak_f32 v2, v2, v29, 0x12800000\n v_rcp_f32 v2, v2\n v_mul_f32 v2, v14, v2\n v_pk_fma_f32 v[16:17], v[22:23], v[2:3], v[16:17] op_sel_hi:[1,0,1]\n v_sub_f32 v24, v12, v8\n v_sub_f32 v25, v13, v9\n v_mul_f32 v30, v24, v24\n v_fmac_f32 v30, v25, v25\n v_add_f32 v4, |v24|, |v25|\n v_max_f32 v30, v30, v10\n v_fmaak_f32 v4, v4, v30, 0x12800000\n v_rcp_f32 v4, v4\n v_mul_f32 v4, v14, v4\n v_pk_fma_f32 v[16:17], v[24:25], v[4:5], v[16:17] op_sel_hi:[1,0,1]\n v_sub_f32 v26, v12, v8\n v_sub_f32 v27, v13, v9\n v_mul_f32 v31, v26, v26\n v_fmac_f32 v31, v27, v27\n v_add_f32 v6, |v26|, |v27|\n v_max_f32 v31, v31, v10\n v_fmaak_f32 v6, v6, v31, 0x12800000\n v_rcp_f32 v6, v6\n v_mul_f32 v6, v14, v6\n v_pk_fma_f32 v[16:17], v[26:27], v[6:7], v[16:17] op_sel_hi:[1,0,1]\n")
KERNEL(k_L_uniform, 0, "v_sub_f32 v20, v12, v8\n v_sub_f32 v21, v13, v9\n v_mul_f32 v28, v20, v20\n v_fmac_f32 v28, v21, v21\n v_add_f32 v0, |v20|, |v21|\n v_max_f32 v28, v28, v10\n v_fmaak_f32 v0, v0, v28, 0x12800000\n v_rcp_f32 v0, v0\n v_pk_fma_f32 v[16:17], v[20:21], v[0:1], v[16:17] op_sel_hi:[1,0,1]\n v_sub_f32 v22, v12, v8\n v_sub_f32 v23, v13, v9\n v_mul_f32 v29, v22, v22\n v_fmac_f32 v29, v23, v23\n v_add_f32 v2, |v22|, |v23|\n v_max_f32 v29, v29, v10\n v_fmaak_f32 v2, v2, v29, 0x12800000\n v_rcp_f32 v2, v2\n v_pk_fma_f32 v[16:17], v[22:23], v[2:3], v[16:17] op_sel_hi:[1,0,1]\n v_sub_f32 v24, v12, v8\n v_sub_f32 v25, v13, v9\n v_mul_f32 v30, v24, v24\n v_fmac_f32 v30, v25, v25\n v_add_f32 v4, |v24|, |v25|\n v_max_f32 v30, v30, v10\n v_fmaak_f32 v4, v4, v30, 0x12800000\n v_rcp_f32 v4, v4\n v_pk_fma_f32 v[16:17], v[24:25], v[4:5], v[16:17] op_sel_hi:[1,0,1]\n v_sub_f32 v26, v12, v8\n v_sub_f32 v27, v13, v9\n v_mul_f32 v31, v26, v26\n v_fmac_f32 v31, v27, v27\n v_add_f32 v6, |v26|, |v27|\n v_max_f32 v31, v31, v10\n v_fmaak_f32 v6, v6, v31, 0x12800000\n v_rcp_f32 v6, v6\n v_pk_fma_f32 v[16:17], v[26:27], v[6:7], v[16:17] op_sel_hi:[1,0,1]\n")
KERNEL(k_S_joint_uniform, 0, "v_pk_add_f32 v[20:21], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v28, v20, v20\n v_fmac_f32 v28, v21, v21\n v_add_f32 v0, |v20|, |v21|\n v_max_f32 v28, v28, v10\n v_fmaak_f32 v0, v0, v28, 0x12800000\n v_pk_add_f32 v[22:23], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v29, v22, v22\n v_fmac_f32 v29, v23, v23\n v_add_f32 v2, |v22|, |v23|\n v_max_f32 v29, v29, v10\n v_fmaak_f32 v2, v2, v29, 0x12800000\n v_mul_f32 v28, v0, v2\n v_rcp_f32 v28, v28\n v_mul_f32 v29, v28, v2\n v_mul_f32 v2, v28, v0\n v_pk_fma_f32 v[16:17], v[20:21], v[28:29], v[16:17] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[16:17], v[22:23], v[2:3], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[24:25], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v30, v24, v24\n v_fmac_f32 v30, v25, v25\n v_add_f32 v4, |v24|, |v25|\n v_max_f32 v30, v30, v10\n v_fmaak_f32 v4, v4, v30, 0x12800000\n v_pk_add_f32 v[26:27], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v31, v26, v26\n v_fmac_f32 v31, v27, v27\n v_add_f32 v6, |v26|, |v27|\n v_max_f32 v31, v31, v10\n v_fmaak_f32 v6, v6, v31, 0x12800000\n v_mul_f32 v30, v4, v6\n v_rcp_f32 v30, v30\n v_mul_f32 v31, v30, v6\n v_mul_f32 v6, v30, v4\n v_pk_fma_f32 v[16:17], v[24:25], v[30:31], v[16:17] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[16:17], v[26:27], v[6:7], v[16:17] op_sel_hi:[1,0,1]\n")
KERNEL(k_S_uniform_noieee, 1, "v_pk_add_f32 v[20:21], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v28, v20, v20\n v_fmac_f32 v28, v21, v21\n v_add_f32 v0, |v20|, |v21|\n v_max_f32 v28, v28, v10\n v_fmaak_f32 v0, v0, v28, 0x12800000\n v_rcp_f32 v0, v0\n v_pk_fma_f32 v[16:17], v[20:21], v[0:1], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[22:23], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v29, v22, v22\n v_fmac_f32 v29, v23, v23\n v_add_f32 v2, |v22|, |v23|\n v_max_f32 v29, v29, v10\n v_fmaak_f32 v2, v2, v29, 0x12800000\n v_rcp_f32 v2, v2\n v_pk_fma_f32 v[16:17], v[22:23], v[2:3], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[24:25], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v30, v24, v24\n v_fmac_f32 v30, v25, v25\n v_add_f32 v4, |v24|, |v25|\n v_max_f32 v30, v30, v10\n v_fmaak_f32 v4, v4, v30, 0x12800000\n v_rcp_f32 v4, v4\n v_pk_fma_f32 v[16:17], v[24:25], v[4:5], v[16:17] op_sel_hi:[1,0,1]\n v_pk_add_f32 v[26:27], s[4:5], v[8:9] neg_lo:[0,1] neg_hi:[0,1]\n v_mul_f32 v31, v26, v26\n v_fmac_f32 v31, v27, v27\n v_add_f32 v6, |v26|, |v27|\n v_max_f32 v31, v31, v10\n v_fmaak_f32 v6, v6, v31, 0x12800000\n v_rcp_f32 v6, v6\n v_pk_fma_f32 v[16:17], v[26:27], v[6:7], v[16:17] op_sel_hi:[1,0,1]\n")
KERNEL(k_max_noieee, 1, "v_max_f32 v0, v0, v9\n v_max_f32 v1, v1, v10\n v_max_f32 v2, v2, v11\n v_max_f32 v3, v3, v12\n v_max_f32 v4, v4, v9\n v_max_f32 v5, v5, v10\n v_max_f32 v6, v6, v11\n v_max_f32 v7, v7, v12\n")

typedef void (*kfn)(float*, float, float, float);
struct Case { const char* name; kfn fn; int per_body; };
int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out; (void)hipMalloc(&out, 1024);
  std::vector<Case> cases = { {"S general (4 pairs)", k_S_general, 4}, {"S uniform mass", k_S_uniform, 4}, {"L general", k_L_general, 4}, {"L uniform", k_L_uniform, 4},
    {"S joint-rcp uniform", k_S_joint_uniform, 4}, {"S uniform, IEEE off", k_S_uniform_noieee, 4}, {"v_max_f32 IEEE off (x8)", k_max_noieee, 8} };
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("%-28s %12s %12s %14s\n", "body (w=8 waves/SIMD)", "ns/pair", "slots/pair", "Gpairs/s chip");
  for (auto& c : cases) {
    int w = 8, blocks = cus * w;
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
    double ns = best * 1e6 / ((double)ITERS * 8 * c.per_body * w);
    printf("%-28s %12.3f %12.2f %14.1f\n", c.name, ns, ns / 0.875, 64.0 * 4 * cus / ns);
  }
  return 0;
}

// Second VALU microbenchmark: explicit registers, to learn gfx950's operand-fetch rules (VGPR banks, SGPR /
// constant-bus operands, 64-bit packed operands) and the cost of clamp alternatives.  Wall-clock, w=8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 8000
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","vcc"
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, BODY)                                                                     \
  __global__ __launch_bounds__(256) void NAME(float* out, float seed, float one, float small) { \
    float t = seed + threadIdx.x * 0.37f;                                                       \
    asm volatile("v_mov_b32 v0, %0\n v_add_f32 v1, 1.0, v0\n v_add_f32 v2, 2.0, v0\n v_add_f32 v3, 4.0, v0\n" \
                 "v_add_f32 v4, 0.5, v0\n v_add_f32 v5, 1.0, v1\n v_add_f32 v6, 1.0, v2\n v_add_f32 v7, 1.0, v3\n" \
                 "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %2\n v_mov_b32 v11, %2\n v_mov_b32 v12, %1\n v_mov_b32 v13, %2\n v_mov_b32 v14, %1\n v_mov_b32 v15, %2\n" \
                 "v_mov_b32 v16, v0\n v_mov_b32 v17, v1\n v_mov_b32 v18, v2\n v_mov_b32 v19, v3\n v_mov_b32 v20, v4\n v_mov_b32 v21, v5\n v_mov_b32 v22, v6\n v_mov_b32 v23, v7\n" \
                 "v_mov_b32 v24, %1\n v_mov_b32 v25, %1\n v_mov_b32 v26, %2\n v_mov_b32 v27, %2\n v_mov_b32 v28, %1\n v_mov_b32 v29, %1\n v_mov_b32 v30, %2\n v_mov_b32 v31, %2\n" \
                 :: "v"(t), "v"(one * 1.0001f), "v"(small) : CLOB);                             \
    for (int i = 0; i < ITERS; ++i) { asm volatile(REP8(BODY) :: "s"(one), "s"(seed) : CLOB); } \
    float s;                                                                                    \
    asm volatile("v_add_f32 %0, v0, v1\n v_add_f32 %0, %0, v2\n v_add_f32 %0, %0, v3\n v_add_f32 %0, %0, v4\n v_add_f32 %0, %0, v5\n v_add_f32 %0, %0, v6\n v_add_f32 %0, %0, v7\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v20\n v_add_f32 %0, %0, v22\n" : "=v"(s) :: CLOB); \
    if (s == 123.456f) out[0] = 1;                                                              \
  }
// accumulators v0..v7 (and pairs v[16:17]..v[22:23]); constants v8..v15, v24..v31
// banks = reg % 4
KERNEL(k_fma_b012, "v_fma_f32 v0, v0, v9, v10\n v_fma_f32 v1, v1, v10, v11\n v_fma_f32 v2, v2, v11, v12\n v_fma_f32 v3, v3, v12, v13\n"
                   "v_fma_f32 v4, v4, v9, v10\n v_fma_f32 v5, v5, v10, v11\n v_fma_f32 v6, v6, v11, v12\n v_fma_f32 v7, v7, v12, v13\n")
KERNEL(k_fma_b000, "v_fma_f32 v0, v0, v8, v12\n v_fma_f32 v1, v1, v9, v13\n v_fma_f32 v2, v2, v10, v14\n v_fma_f32 v3, v3, v11, v15\n"
                   "v_fma_f32 v4, v4, v8, v12\n v_fma_f32 v5, v5, v9, v13\n v_fma_f32 v6, v6, v10, v14\n v_fma_f32 v7, v7, v11, v15\n")
KERNEL(k_fma_2rd, "v_fma_f32 v0, v0, v9, v9\n v_fma_f32 v1, v1, v10, v10\n v_fma_f32 v2, v2, v11, v11\n v_fma_f32 v3, v3, v12, v12\n"
                  "v_fma_f32 v4, v4, v9, v9\n v_fma_f32 v5, v5, v10, v10\n v_fma_f32 v6, v6, v11, v11\n v_fma_f32 v7, v7, v12, v12\n")
KERNEL(k_fma_inl, "v_fma_f32 v0, v0, v9, 1.0\n v_fma_f32 v1, v1, v10, 1.0\n v_fma_f32 v2, v2, v11, 1.0\n v_fma_f32 v3, v3, v12, 1.0\n"
                  "v_fma_f32 v4, v4, v9, 1.0\n v_fma_f32 v5, v5, v10, 1.0\n v_fma_f32 v6, v6, v11, 1.0\n v_fma_f32 v7, v7, v12, 1.0\n")
KERNEL(k_fmac_2rd, "v_fmac_f32 v0, v9, v9\n v_fmac_f32 v1, v10, v10\n v_fmac_f32 v2, v11, v11\n v_fmac_f32 v3, v12, v12\n"
                   "v_fmac_f32 v4, v9, v9\n v_fmac_f32 v5, v10, v10\n v_fmac_f32 v6, v11, v11\n v_fmac_f32 v7, v12, v12\n")
KERNEL(k_fmac_3rd, "v_fmac_f32 v0, v9, v10\n v_fmac_f32 v1, v10, v11\n v_fmac_f32 v2, v11, v12\n v_fmac_f32 v3, v12, v13\n"
                   "v_fmac_f32 v4, v9, v10\n v_fmac_f32 v5, v10, v11\n v_fmac_f32 v6, v11, v12\n v_fmac_f32 v7, v12, v13\n")
KERNEL(k_mul_b01, "v_mul_f32 v0, v0, v9\n v_mul_f32 v1, v1, v10\n v_mul_f32 v2, v2, v11\n v_mul_f32 v3, v3, v12\n"
                  "v_mul_f32 v4, v4, v9\n v_mul_f32 v5, v5, v10\n v_mul_f32 v6, v6, v11\n v_mul_f32 v7, v7, v12\n")
KERNEL(k_mul_b00, "v_mul_f32 v0, v0, v8\n v_mul_f32 v1, v1, v9\n v_mul_f32 v2, v2, v10\n v_mul_f32 v3, v3, v11\n"
                  "v_mul_f32 v4, v4, v8\n v_mul_f32 v5, v5, v9\n v_mul_f32 v6, v6, v10\n v_mul_f32 v7, v7, v11\n")
KERNEL(k_mul_sgpr, "v_mul_f32 v0, %0, v0\n v_mul_f32 v1, %0, v1\n v_mul_f32 v2, %0, v2\n v_mul_f32 v3, %0, v3\n"
                   "v_mul_f32 v4, %0, v4\n v_mul_f32 v5, %0, v5\n v_mul_f32 v6, %0, v6\n v_mul_f32 v7, %0, v7\n")
KERNEL(k_mul_inl, "v_mul_f32 v0, 1.0, v0\n v_mul_f32 v1, 1.0, v1\n v_mul_f32 v2, 1.0, v2\n v_mul_f32 v3, 1.0, v3\n"
                  "v_mul_f32 v4, 1.0, v4\n v_mul_f32 v5, 1.0, v5\n v_mul_f32 v6, 1.0, v6\n v_mul_f32 v7, 1.0, v7\n")
KERNEL(k_mul_lit, "v_mul_f32 v0, 0x3f800347, v0\n v_mul_f32 v1, 0x3f800347, v1\n v_mul_f32 v2, 0x3f800347, v2\n v_mul_f32 v3, 0x3f800347, v3\n"
                  "v_mul_f32 v4, 0x3f800347, v4\n v_mul_f32 v5, 0x3f800347, v5\n v_mul_f32 v6, 0x3f800347, v6\n v_mul_f32 v7, 0x3f800347, v7\n")
KERNEL(k_max_f32, "v_max_f32 v0, v0, v9\n v_max_f32 v1, v1, v10\n v_max_f32 v2, v2, v11\n v_max_f32 v3, v3, v12\n"
                  "v_max_f32 v4, v4, v9\n v_max_f32 v5, v5, v10\n v_max_f32 v6, v6, v11\n v_max_f32 v7, v7, v12\n")
KERNEL(k_max_lit, "v_max_f32 v0, 0x3a83126f, v0\n v_max_f32 v1, 0x3a83126f, v1\n v_max_f32 v2, 0x3a83126f, v2\n v_max_f32 v3, 0x3a83126f, v3\n"
                  "v_max_f32 v4, 0x3a83126f, v4\n v_max_f32 v5, 0x3a83126f, v5\n v_max_f32 v6, 0x3a83126f, v6\n v_max_f32 v7, 0x3a83126f, v7\n")
KERNEL(k_min_f32, "v_min_f32 v0, v0, v9\n v_min_f32 v1, v1, v10\n v_min_f32 v2, v2, v11\n v_min_f32 v3, v3, v12\n"
                  "v_min_f32 v4, v4, v9\n v_min_f32 v5, v5, v10\n v_min_f32 v6, v6, v11\n v_min_f32 v7, v7, v12\n")
KERNEL(k_max_u32, "v_max_u32 v0, v0, v9\n v_max_u32 v1, v1, v10\n v_max_u32 v2, v2, v11\n v_max_u32 v3, v3, v12\n"
                  "v_max_u32 v4, v4, v9\n v_max_u32 v5, v5, v10\n v_max_u32 v6, v6, v11\n v_max_u32 v7, v7, v12\n")
KERNEL(k_max_i32, "v_max_i32 v0, v0, v9\n v_max_i32 v1, v1, v10\n v_max_i32 v2, v2, v11\n v_max_i32 v3, v3, v12\n"
                  "v_max_i32 v4, v4, v9\n v_max_i32 v5, v5, v10\n v_max_i32 v6, v6, v11\n v_max_i32 v7, v7, v12\n")
KERNEL(k_max_u32lit, "v_max_u32 v0, 0x3a83126f, v0\n v_max_u32 v1, 0x3a83126f, v1\n v_max_u32 v2, 0x3a83126f, v2\n v_max_u32 v3, 0x3a83126f, v3\n"
                     "v_max_u32 v4, 0x3a83126f, v4\n v_max_u32 v5, 0x3a83126f, v5\n v_max_u32 v6, 0x3a83126f, v6\n v_max_u32 v7, 0x3a83126f, v7\n")
KERNEL(k_med3, "v_med3_f32 v0, v0, v9, v10\n v_med3_f32 v1, v1, v10, v11\n v_med3_f32 v2, v2, v11, v12\n v_med3_f32 v3, v3, v12, v13\n"
               "v_med3_f32 v4, v4, v9, v10\n v_med3_f32 v5, v5, v10, v11\n v_med3_f32 v6, v6, v11, v12\n v_med3_f32 v7, v7, v12, v13\n")
KERNEL(k_add_u32, "v_add_u32 v0, v0, v9\n v_add_u32 v1, v1, v10\n v_add_u32 v2, v2, v11\n v_add_u32 v3, v3, v12\n"
                  "v_add_u32 v4, v4, v9\n v_add_u32 v5, v5, v10\n v_add_u32 v6, v6, v11\n v_add_u32 v7, v7, v12\n")
KERNEL(k_and_b32, "v_and_b32 v0, v0, v9\n v_and_b32 v1, v1, v10\n v_and_b32 v2, v2, v11\n v_and_b32 v3, v3, v12\n"
                  "v_and_b32 v4, v4, v9\n v_and_b32 v5, v5, v10\n v_and_b32 v6, v6, v11\n v_and_b32 v7, v7, v12\n")
KERNEL(k_pk_fma, "v_pk_fma_f32 v[16:17], v[16:17], v[24:25], v[26:27]\n v_pk_fma_f32 v[18:19], v[18:19], v[24:25], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[28:29], v[30:31]\n v_pk_fma_f32 v[22:23], v[22:23], v[28:29], v[30:31]\n"
                 "v_pk_fma_f32 v[16:17], v[16:17], v[24:25], v[26:27]\n v_pk_fma_f32 v[18:19], v[18:19], v[24:25], v[26:27]\n v_pk_fma_f32 v[20:21], v[20:21], v[28:29], v[30:31]\n v_pk_fma_f32 v[22:23], v[22:23], v[28:29], v[30:31]\n")
KERNEL(k_pk_fma_2rd, "v_pk_fma_f32 v[16:17], v[24:25], v[24:25], v[16:17]\n v_pk_fma_f32 v[18:19], v[24:25], v[24:25], v[18:19]\n v_pk_fma_f32 v[20:21], v[28:29], v[28:29], v[20:21]\n v_pk_fma_f32 v[22:23], v[28:29], v[28:29], v[22:23]\n"
                     "v_pk_fma_f32 v[16:17], v[24:25], v[24:25], v[16:17]\n v_pk_fma_f32 v[18:19], v[24:25], v[24:25], v[18:19]\n v_pk_fma_f32 v[20:21], v[28:29], v[28:29], v[20:21]\n v_pk_fma_f32 v[22:23], v[28:29], v[28:29], v[22:23]\n")
KERNEL(k_pk_fma_bc, "v_pk_fma_f32 v[16:17], v[24:25], v[26:27], v[16:17] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[18:19], v[24:25], v[26:27], v[18:19] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[20:21], v[28:29], v[30:31], v[20:21] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[22:23], v[28:29], v[30:31], v[22:23] op_sel_hi:[1,0,1]\n"
                    "v_pk_fma_f32 v[16:17], v[24:25], v[26:27], v[16:17] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[18:19], v[24:25], v[26:27], v[18:19] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[20:21], v[28:29], v[30:31], v[20:21] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[22:23], v[28:29], v[30:31], v[22:23] op_sel_hi:[1,0,1]\n")
KERNEL(k_pk_mul, "v_pk_mul_f32 v[16:17], v[16:17], v[24:25]\n v_pk_mul_f32 v[18:19], v[18:19], v[24:25]\n v_pk_mul_f32 v[20:21], v[20:21], v[28:29]\n v_pk_mul_f32 v[22:23], v[22:23], v[28:29]\n"
                 "v_pk_mul_f32 v[16:17], v[16:17], v[24:25]\n v_pk_mul_f32 v[18:19], v[18:19], v[24:25]\n v_pk_mul_f32 v[20:21], v[20:21], v[28:29]\n v_pk_mul_f32 v[22:23], v[22:23], v[28:29]\n")
KERNEL(k_pk_mul_1rd, "v_pk_mul_f32 v[16:17], v[24:25], v[24:25]\n v_pk_mul_f32 v[18:19], v[24:25], v[24:25]\n v_pk_mul_f32 v[20:21], v[28:29], v[28:29]\n v_pk_mul_f32 v[22:23], v[28:29], v[28:29]\n"
                     "v_pk_mul_f32 v[16:17], v[24:25], v[24:25]\n v_pk_mul_f32 v[18:19], v[24:25], v[24:25]\n v_pk_mul_f32 v[20:21], v[28:29], v[28:29]\n v_pk_mul_f32 v[22:23], v[28:29], v[28:29]\n")
KERNEL(k_pk_add_sg, "v_pk_add_f32 v[16:17], s[4:5], v[16:17]\n v_pk_add_f32 v[18:19], s[4:5], v[18:19]\n v_pk_add_f32 v[20:21], s[4:5], v[20:21]\n v_pk_add_f32 v[22:23], s[4:5], v[22:23]\n"
                    "v_pk_add_f32 v[16:17], s[4:5], v[16:17]\n v_pk_add_f32 v[18:19], s[4:5], v[18:19]\n v_pk_add_f32 v[20:21], s[4:5], v[20:21]\n v_pk_add_f32 v[22:23], s[4:5], v[22:23]\n")
KERNEL(k_rcp, "v_rcp_f32 v0, v0\n v_rcp_f32 v1, v1\n v_rcp_f32 v2, v2\n v_rcp_f32 v3, v3\n v_rcp_f32 v4, v4\n v_rcp_f32 v5, v5\n v_rcp_f32 v6, v6\n v_rcp_f32 v7, v7\n")
KERNEL(k_rcp_f16, "v_rcp_f16 v0, v0\n v_rcp_f16 v1, v1\n v_rcp_f16 v2, v2\n v_rcp_f16 v3, v3\n v_rcp_f16 v4, v4\n v_rcp_f16 v5, v5\n v_rcp_f16 v6, v6\n v_rcp_f16 v7, v7\n")
KERNEL(k_rcp_then_mul, "v_rcp_f32 v0, v0\n v_mul_f32 v1, v1, v9\n v_mul_f32 v2, v2, v10\n v_mul_f32 v3, v3, v11\n v_rcp_f32 v4, v4\n v_mul_f32 v5, v5, v10\n v_mul_f32 v6, v6, v11\n v_mul_f32 v7, v7, v12\n")
KERNEL(k_dpp_sub, "v_sub_f32_dpp v0, v0, v9 row_shr:1\n v_sub_f32_dpp v1, v1, v10 row_shr:1\n v_sub_f32_dpp v2, v2, v11 row_shr:1\n v_sub_f32_dpp v3, v3, v12 row_shr:1\n"
                  "v_sub_f32_dpp v4, v4, v9 row_shr:1\n v_sub_f32_dpp v5, v5, v10 row_shr:1\n v_sub_f32_dpp v6, v6, v11 row_shr:1\n v_sub_f32_dpp v7, v7, v12 row_shr:1\n")
KERNEL(k_cndmask, "v_cndmask_b32 v0, v0, v9, vcc\n v_cndmask_b32 v1, v1, v10, vcc\n v_cndmask_b32 v2, v2, v11, vcc\n v_cndmask_b32 v3, v3, v12, vcc\n"
                  "v_cndmask_b32 v4, v4, v9, vcc\n v_cndmask_b32 v5, v5, v10, vcc\n v_cndmask_b32 v6, v6, v11, vcc\n v_cndmask_b32 v7, v7, v12, vcc\n")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, v0, v9\n v_cmp_lt_f32 vcc, v1, v10\n v_cmp_lt_f32 vcc, v2, v11\n v_cmp_lt_f32 vcc, v3, v12\n"
              "v_cmp_lt_f32 vcc, v4, v9\n v_cmp_lt_f32 vcc, v5, v10\n v_cmp_lt_f32 vcc, v6, v11\n v_cmp_lt_f32 vcc, v7, v12\n")
typedef void (*kfn)(float*, float, float, float);
struct Case { const char* name; kfn fn; };
int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out; (void)hipMalloc(&out, 1024);
  std::vector<Case> cases = {
    {"fma v,v,v banks 0,1,2", k_fma_b012}, {"fma v,v,v one bank", k_fma_b000}, {"fma v,a,a (2 distinct)", k_fma_2rd}, {"fma v,v,1.0 (inline)", k_fma_inl},
    {"fmac v,a,a (2 distinct)", k_fmac_2rd}, {"fmac v,a,b (3 distinct)", k_fmac_3rd},
    {"mul v,v banks 0,1", k_mul_b01}, {"mul v,v one bank", k_mul_b00}, {"mul sgpr,v", k_mul_sgpr}, {"mul 1.0(inline),v", k_mul_inl}, {"mul literal,v", k_mul_lit},
    {"max_f32 v,v", k_max_f32}, {"max_f32 literal,v", k_max_lit}, {"min_f32 v,v", k_min_f32}, {"max_u32 v,v", k_max_u32}, {"max_i32 v,v", k_max_i32}, {"max_u32 literal,v", k_max_u32lit},
    {"med3_f32", k_med3}, {"add_u32", k_add_u32}, {"and_b32", k_and_b32},
    {"pk_fma 3x64b distinct", k_pk_fma}, {"pk_fma a,a,acc", k_pk_fma_2rd}, {"pk_fma a,b(bcast),acc", k_pk_fma_bc}, {"pk_mul acc,a", k_pk_mul}, {"pk_mul a,a", k_pk_mul_1rd},
    {"pk_add sgpr pair,v", k_pk_add_sg},
    {"rcp_f32", k_rcp}, {"rcp_f16", k_rcp_f16}, {"1 rcp + 3 mul", k_rcp_then_mul}, {"sub_dpp row_shr", k_dpp_sub}, {"cndmask", k_cndmask}, {"cmp_lt_f32", k_cmp}};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("%-28s %10s %10s\n", "instr (w=8 waves/SIMD)", "ns/instr", "slots(0.875)");
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
    double ns = best * 1e6 / ((double)ITERS * 64 * w);
    printf("%-28s %10.3f %10.2f\n", c.name, ns, ns / 0.875);
  }
  return 0;
}

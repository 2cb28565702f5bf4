// Direct O(N^2) force + semi-implicit Euler kernels for gfx950 (MI355X, wave64).
//
// What is computed:
//   a_i = sum_j calculate_gravity(p_i, p_j, w_j)        reference force law   src/main.rs:234-253
//   v_i += a_i*dt ; x_i += v_i*dt                        reference integrate   src/main.rs:419-423
//
// Two arithmetic flavours (include/nbody_hip.h, nbody_arith):
//   FAST   direct_stream / direct_fast: one thread per target, 6 issue slots per pair.  The coincident-pair skip of
//          main.rs:241-243 costs nothing: den = fma(sum, d2c, 2^-90) keeps the reciprocal finite, so a zero diff contributes
//          exactly 0.  Valid when no position is non-finite, >= 2^60, or non-zero below 2^-22 (then every non-zero |dx|+|dy| is
//          >= 2^-46 and the bias is below half an ulp of den); direct_hazard_scan (or nf_insert) checks exactly that, per call.
//   EXACT  direct_exact: every operation as the reference writes it: IEEE subtract/multiply/add/divide, no
//          contraction, the is_normal() skip, one sequential ascending-j chain per target.  Bit-identical to
//          the oracle.
//
// The measured issue model that shapes FAST (tools/valu_microbench*.hip, tools/mb/*):
//   a wave issues one VALU instruction per 4-cycle slot, and so does its SIMD — except that two PLAIN f32 ops of two different
//   waves can share a slot, which the arbiter only arranges when it has nothing else to issue; a packed op (v_pk_*_f32) fills its
//   slot alone with two values per lane, v_max / compares / SGPR-operand ops with one, v_rcp_f32 takes two slots.  So: every
//   multiply-add is packed over TWO pairs (sources as couples {xA, xB, yA, yB}), the far sources stream through SGPRs (a packed op
//   takes an SGPR pair at no cost: no LDS, no barrier), and the few plain ops that remain (|dx| + |dy|) run at a lower wave priority
//   so that they pair: 12 slots per couple = 24 cycles per pair.  The round-1/2 kernels (USE_ASM 0 / 1: a pair per instruction, four
//   plain ops per pair at a slot each) are kept for A/B runs.
//
// This translation unit is compiled with -ffp-contract=off: nothing fuses unless written as fmaf().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "direct_kernels.h"
#include "div_pair.h"

namespace nbody {

static constexpr float kDenBias = 8.0779356694631609e-28f;  // 2^-90
static constexpr float kBig = kFastBig;    // 2^60
static constexpr float kTiny = kFastTiny;  // 2^-22

__device__ __forceinline__ int wave_id_uniform() {
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

__device__ __forceinline__ bool gate_open(const DirectArgs& a) {
  return a.run_state < 0 || a.flags[kFlagState] == a.run_state;
}

// One FAST pair.  14 algorithmic flops: 2 sub, 1 add, mul+fma, max, fma, rcp, mul, 2 fma.
template <bool UNIFORM, bool NOCLAMP>
__device__ __forceinline__ void fast_pair(float xi, float yi, float xj, float yj, float mj, float clamp, float& ax,
                                          float& ay) {
  float dx = xj - xi;
  float dy = yj - yi;
  float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);
  float d2 = __builtin_fmaf(dy, dy, dx * dx);
  if (!NOCLAMP) d2 = __builtin_fmaxf(d2, clamp);
  float den = __builtin_fmaf(sum, d2, kDenBias);
  float s = __builtin_amdgcn_rcpf(den);
  if (!UNIFORM) s = mj * s;
  ax = __builtin_fmaf(dx, s, ax);
  ay = __builtin_fmaf(dy, s, ay);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#ifdef NBODY_LAB  // the round-2 block (NBODY_DIRECT_ASM=1)
// Eight FAST pairs for one target as one hand-ordered instruction block: the same operations as fast_pair,
// issued in phases (8 x v_pk_add | the 32-bit ops | 8 x v_rcp | 8 x v_pk_fma).  With per-body masses m0..m7 are the
// INVERSE masses and scale the denominator before the reciprocal (s = 1/(den/m) instead of m * (1/den)).  p0..p7 come in as source
// positions and leave as the differences.  Temporaries live in v40..v55: pair k = v[40+2k : 41+2k] = (s_k, d2_k),
// so that the pk_fma can broadcast s_k with op_sel_hi.  0x12800000 = 2^-90 (kDenBias).  VALU->VALU
// dependencies are interlocked by the hardware; every v_rcp result is consumed >= 8 instructions later (the
// trans forwarding hazard of gfx940+ needs 1).  Worth 2 % over hipcc's schedule of the same instructions.
// Operands are native 2-vectors (HIP's float2 is a struct: as an asm operand it is coerced through an i64 and costs
// shift/or/move instructions around every block).
template <bool UNIFORM, bool NOCLAMP>
__device__ __forceinline__ void fast_block8(v2f t, float clamp, v2f& p0, v2f& p1, v2f& p2, v2f& p3, v2f& p4, v2f& p5,
                                            v2f& p6, v2f& p7, float m0, float m1, float m2, float m3, float m4,
                                            float m5, float m6, float m7, v2f& acc) {
  asm volatile(
      "v_pk_add_f32 %[p0], %[p0], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p1], %[p1], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p2], %[p2], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p3], %[p3], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p4], %[p4], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p5], %[p5], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p6], %[p6], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p7], %[p7], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [p5] "+v"(p5), [p6] "+v"(p6),
        [p7] "+v"(p7)
      : [t] "v"(t));
  if constexpr (UNIFORM && NOCLAMP) {
      asm volatile(
        "v_mul_f32 v41, %[x0], %[x0]\n\tv_mul_f32 v43, %[x1], %[x1]\n\tv_mul_f32 v45, %[x2], %[x2]\n\tv_mul_f32 v47, %[x3], %[x3]\n\tv_mul_f32 v49, %[x4], %[x4]\n\tv_mul_f32 v51, %[x5], %[x5]\n\tv_mul_f32 v53, %[x6], %[x6]\n\tv_mul_f32 v55, %[x7], %[x7]\n\t"
        "v_fmac_f32 v41, %[y0], %[y0]\n\tv_fmac_f32 v43, %[y1], %[y1]\n\tv_fmac_f32 v45, %[y2], %[y2]\n\tv_fmac_f32 v47, %[y3], %[y3]\n\tv_fmac_f32 v49, %[y4], %[y4]\n\tv_fmac_f32 v51, %[y5], %[y5]\n\tv_fmac_f32 v53, %[y6], %[y6]\n\tv_fmac_f32 v55, %[y7], %[y7]\n\t"
        "v_add_f32 v40, |%[x0]|, |%[y0]|\n\tv_add_f32 v42, |%[x1]|, |%[y1]|\n\tv_add_f32 v44, |%[x2]|, |%[y2]|\n\tv_add_f32 v46, |%[x3]|, |%[y3]|\n\tv_add_f32 v48, |%[x4]|, |%[y4]|\n\tv_add_f32 v50, |%[x5]|, |%[y5]|\n\tv_add_f32 v52, |%[x6]|, |%[y6]|\n\tv_add_f32 v54, |%[x7]|, |%[y7]|\n\t"
        "v_fmaak_f32 v40, v40, v41, 0x12800000\n\tv_fmaak_f32 v42, v42, v43, 0x12800000\n\tv_fmaak_f32 v44, v44, v45, 0x12800000\n\tv_fmaak_f32 v46, v46, v47, 0x12800000\n\tv_fmaak_f32 v48, v48, v49, 0x12800000\n\tv_fmaak_f32 v50, v50, v51, 0x12800000\n\tv_fmaak_f32 v52, v52, v53, 0x12800000\n\tv_fmaak_f32 v54, v54, v55, 0x12800000\n\t"
        "v_rcp_f32 v40, v40\n\tv_rcp_f32 v42, v42\n\tv_rcp_f32 v44, v44\n\tv_rcp_f32 v46, v46\n\tv_rcp_f32 v48, v48\n\tv_rcp_f32 v50, v50\n\tv_rcp_f32 v52, v52\n\tv_rcp_f32 v54, v54\n\t"
        "v_pk_fma_f32 %[a], %[p0], v[40:41], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p1], v[42:43], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p2], v[44:45], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p3], v[46:47], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p4], v[48:49], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p5], v[50:51], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p6], v[52:53], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p7], v[54:55], %[a] op_sel_hi:[1,0,1]\n\t"
        : [a] "+v"(acc)
        :[p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
        [x0] "v"(p0.x), [y0] "v"(p0.y), [x1] "v"(p1.x), [y1] "v"(p1.y), [x2] "v"(p2.x), [y2] "v"(p2.y), [x3] "v"(p3.x), [y3] "v"(p3.y), [x4] "v"(p4.x), [y4] "v"(p4.y), [x5] "v"(p5.x), [y5] "v"(p5.y), [x6] "v"(p6.x), [y6] "v"(p6.y), [x7] "v"(p7.x), [y7] "v"(p7.y)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  } else if constexpr (UNIFORM && !NOCLAMP) {
      asm volatile(
        "v_mul_f32 v41, %[x0], %[x0]\n\tv_mul_f32 v43, %[x1], %[x1]\n\tv_mul_f32 v45, %[x2], %[x2]\n\tv_mul_f32 v47, %[x3], %[x3]\n\tv_mul_f32 v49, %[x4], %[x4]\n\tv_mul_f32 v51, %[x5], %[x5]\n\tv_mul_f32 v53, %[x6], %[x6]\n\tv_mul_f32 v55, %[x7], %[x7]\n\t"
        "v_fmac_f32 v41, %[y0], %[y0]\n\tv_fmac_f32 v43, %[y1], %[y1]\n\tv_fmac_f32 v45, %[y2], %[y2]\n\tv_fmac_f32 v47, %[y3], %[y3]\n\tv_fmac_f32 v49, %[y4], %[y4]\n\tv_fmac_f32 v51, %[y5], %[y5]\n\tv_fmac_f32 v53, %[y6], %[y6]\n\tv_fmac_f32 v55, %[y7], %[y7]\n\t"
        "v_add_f32 v40, |%[x0]|, |%[y0]|\n\tv_max_f32 v41, v41, %[c]\n\t"
        "v_add_f32 v42, |%[x1]|, |%[y1]|\n\tv_max_f32 v43, v43, %[c]\n\t"
        "v_add_f32 v44, |%[x2]|, |%[y2]|\n\tv_max_f32 v45, v45, %[c]\n\t"
        "v_add_f32 v46, |%[x3]|, |%[y3]|\n\tv_max_f32 v47, v47, %[c]\n\t"
        "v_add_f32 v48, |%[x4]|, |%[y4]|\n\tv_max_f32 v49, v49, %[c]\n\t"
        "v_add_f32 v50, |%[x5]|, |%[y5]|\n\tv_max_f32 v51, v51, %[c]\n\t"
        "v_add_f32 v52, |%[x6]|, |%[y6]|\n\tv_max_f32 v53, v53, %[c]\n\t"
        "v_add_f32 v54, |%[x7]|, |%[y7]|\n\tv_max_f32 v55, v55, %[c]\n\t"
        "v_fmaak_f32 v40, v40, v41, 0x12800000\n\tv_fmaak_f32 v42, v42, v43, 0x12800000\n\tv_fmaak_f32 v44, v44, v45, 0x12800000\n\tv_fmaak_f32 v46, v46, v47, 0x12800000\n\tv_fmaak_f32 v48, v48, v49, 0x12800000\n\tv_fmaak_f32 v50, v50, v51, 0x12800000\n\tv_fmaak_f32 v52, v52, v53, 0x12800000\n\tv_fmaak_f32 v54, v54, v55, 0x12800000\n\t"
        "v_rcp_f32 v40, v40\n\tv_rcp_f32 v42, v42\n\tv_rcp_f32 v44, v44\n\tv_rcp_f32 v46, v46\n\tv_rcp_f32 v48, v48\n\tv_rcp_f32 v50, v50\n\tv_rcp_f32 v52, v52\n\tv_rcp_f32 v54, v54\n\t"
        "v_pk_fma_f32 %[a], %[p0], v[40:41], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p1], v[42:43], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p2], v[44:45], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p3], v[46:47], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p4], v[48:49], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p5], v[50:51], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p6], v[52:53], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p7], v[54:55], %[a] op_sel_hi:[1,0,1]\n\t"
        : [a] "+v"(acc)
        :[p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
        [x0] "v"(p0.x), [y0] "v"(p0.y), [x1] "v"(p1.x), [y1] "v"(p1.y), [x2] "v"(p2.x), [y2] "v"(p2.y), [x3] "v"(p3.x), [y3] "v"(p3.y), [x4] "v"(p4.x), [y4] "v"(p4.y), [x5] "v"(p5.x), [y5] "v"(p5.y), [x6] "v"(p6.x), [y6] "v"(p6.y), [x7] "v"(p7.x), [y7] "v"(p7.y),
        [c] "v"(clamp)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  } else if constexpr (!UNIFORM && NOCLAMP) {
      asm volatile(
        "v_mul_f32 v41, %[x0], %[x0]\n\tv_mul_f32 v43, %[x1], %[x1]\n\tv_mul_f32 v45, %[x2], %[x2]\n\tv_mul_f32 v47, %[x3], %[x3]\n\tv_mul_f32 v49, %[x4], %[x4]\n\tv_mul_f32 v51, %[x5], %[x5]\n\tv_mul_f32 v53, %[x6], %[x6]\n\tv_mul_f32 v55, %[x7], %[x7]\n\t"
        "v_fmac_f32 v41, %[y0], %[y0]\n\tv_fmac_f32 v43, %[y1], %[y1]\n\tv_fmac_f32 v45, %[y2], %[y2]\n\tv_fmac_f32 v47, %[y3], %[y3]\n\tv_fmac_f32 v49, %[y4], %[y4]\n\tv_fmac_f32 v51, %[y5], %[y5]\n\tv_fmac_f32 v53, %[y6], %[y6]\n\tv_fmac_f32 v55, %[y7], %[y7]\n\t"
        "v_add_f32 v40, |%[x0]|, |%[y0]|\n\tv_add_f32 v42, |%[x1]|, |%[y1]|\n\tv_add_f32 v44, |%[x2]|, |%[y2]|\n\tv_add_f32 v46, |%[x3]|, |%[y3]|\n\tv_add_f32 v48, |%[x4]|, |%[y4]|\n\tv_add_f32 v50, |%[x5]|, |%[y5]|\n\tv_add_f32 v52, |%[x6]|, |%[y6]|\n\tv_add_f32 v54, |%[x7]|, |%[y7]|\n\t"
        "v_fmaak_f32 v40, v40, v41, 0x12800000\n\tv_fmaak_f32 v42, v42, v43, 0x12800000\n\tv_fmaak_f32 v44, v44, v45, 0x12800000\n\tv_fmaak_f32 v46, v46, v47, 0x12800000\n\tv_fmaak_f32 v48, v48, v49, 0x12800000\n\tv_fmaak_f32 v50, v50, v51, 0x12800000\n\tv_fmaak_f32 v52, v52, v53, 0x12800000\n\tv_fmaak_f32 v54, v54, v55, 0x12800000\n\t"
        "v_mul_f32 v40, %[m0], v40\n\tv_mul_f32 v42, %[m1], v42\n\tv_mul_f32 v44, %[m2], v44\n\tv_mul_f32 v46, %[m3], v46\n\tv_mul_f32 v48, %[m4], v48\n\tv_mul_f32 v50, %[m5], v50\n\tv_mul_f32 v52, %[m6], v52\n\tv_mul_f32 v54, %[m7], v54\n\t"
        "v_rcp_f32 v40, v40\n\tv_rcp_f32 v42, v42\n\tv_rcp_f32 v44, v44\n\tv_rcp_f32 v46, v46\n\tv_rcp_f32 v48, v48\n\tv_rcp_f32 v50, v50\n\tv_rcp_f32 v52, v52\n\tv_rcp_f32 v54, v54\n\t"
        "v_pk_fma_f32 %[a], %[p0], v[40:41], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p1], v[42:43], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p2], v[44:45], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p3], v[46:47], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p4], v[48:49], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p5], v[50:51], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p6], v[52:53], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p7], v[54:55], %[a] op_sel_hi:[1,0,1]\n\t"
        : [a] "+v"(acc)
        :[p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
        [x0] "v"(p0.x), [y0] "v"(p0.y), [x1] "v"(p1.x), [y1] "v"(p1.y), [x2] "v"(p2.x), [y2] "v"(p2.y), [x3] "v"(p3.x), [y3] "v"(p3.y), [x4] "v"(p4.x), [y4] "v"(p4.y), [x5] "v"(p5.x), [y5] "v"(p5.y), [x6] "v"(p6.x), [y6] "v"(p6.y), [x7] "v"(p7.x), [y7] "v"(p7.y),
        [m0] "v"(m0), [m1] "v"(m1), [m2] "v"(m2), [m3] "v"(m3), [m4] "v"(m4), [m5] "v"(m5), [m6] "v"(m6), [m7] "v"(m7)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  } else {
      asm volatile(
        "v_mul_f32 v41, %[x0], %[x0]\n\tv_mul_f32 v43, %[x1], %[x1]\n\tv_mul_f32 v45, %[x2], %[x2]\n\tv_mul_f32 v47, %[x3], %[x3]\n\tv_mul_f32 v49, %[x4], %[x4]\n\tv_mul_f32 v51, %[x5], %[x5]\n\tv_mul_f32 v53, %[x6], %[x6]\n\tv_mul_f32 v55, %[x7], %[x7]\n\t"
        "v_fmac_f32 v41, %[y0], %[y0]\n\tv_fmac_f32 v43, %[y1], %[y1]\n\tv_fmac_f32 v45, %[y2], %[y2]\n\tv_fmac_f32 v47, %[y3], %[y3]\n\tv_fmac_f32 v49, %[y4], %[y4]\n\tv_fmac_f32 v51, %[y5], %[y5]\n\tv_fmac_f32 v53, %[y6], %[y6]\n\tv_fmac_f32 v55, %[y7], %[y7]\n\t"
        "v_add_f32 v40, |%[x0]|, |%[y0]|\n\tv_max_f32 v41, v41, %[c]\n\t"
        "v_add_f32 v42, |%[x1]|, |%[y1]|\n\tv_max_f32 v43, v43, %[c]\n\t"
        "v_add_f32 v44, |%[x2]|, |%[y2]|\n\tv_max_f32 v45, v45, %[c]\n\t"
        "v_add_f32 v46, |%[x3]|, |%[y3]|\n\tv_max_f32 v47, v47, %[c]\n\t"
        "v_add_f32 v48, |%[x4]|, |%[y4]|\n\tv_max_f32 v49, v49, %[c]\n\t"
        "v_add_f32 v50, |%[x5]|, |%[y5]|\n\tv_max_f32 v51, v51, %[c]\n\t"
        "v_add_f32 v52, |%[x6]|, |%[y6]|\n\tv_max_f32 v53, v53, %[c]\n\t"
        "v_add_f32 v54, |%[x7]|, |%[y7]|\n\tv_max_f32 v55, v55, %[c]\n\t"
        "v_fmaak_f32 v40, v40, v41, 0x12800000\n\tv_fmaak_f32 v42, v42, v43, 0x12800000\n\tv_fmaak_f32 v44, v44, v45, 0x12800000\n\tv_fmaak_f32 v46, v46, v47, 0x12800000\n\tv_fmaak_f32 v48, v48, v49, 0x12800000\n\tv_fmaak_f32 v50, v50, v51, 0x12800000\n\tv_fmaak_f32 v52, v52, v53, 0x12800000\n\tv_fmaak_f32 v54, v54, v55, 0x12800000\n\t"
        "v_mul_f32 v40, %[m0], v40\n\tv_mul_f32 v42, %[m1], v42\n\tv_mul_f32 v44, %[m2], v44\n\tv_mul_f32 v46, %[m3], v46\n\tv_mul_f32 v48, %[m4], v48\n\tv_mul_f32 v50, %[m5], v50\n\tv_mul_f32 v52, %[m6], v52\n\tv_mul_f32 v54, %[m7], v54\n\t"
        "v_rcp_f32 v40, v40\n\tv_rcp_f32 v42, v42\n\tv_rcp_f32 v44, v44\n\tv_rcp_f32 v46, v46\n\tv_rcp_f32 v48, v48\n\tv_rcp_f32 v50, v50\n\tv_rcp_f32 v52, v52\n\tv_rcp_f32 v54, v54\n\t"
        "v_pk_fma_f32 %[a], %[p0], v[40:41], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p1], v[42:43], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p2], v[44:45], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p3], v[46:47], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p4], v[48:49], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p5], v[50:51], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p6], v[52:53], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p7], v[54:55], %[a] op_sel_hi:[1,0,1]\n\t"
        : [a] "+v"(acc)
        :[p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
        [x0] "v"(p0.x), [y0] "v"(p0.y), [x1] "v"(p1.x), [y1] "v"(p1.y), [x2] "v"(p2.x), [y2] "v"(p2.y), [x3] "v"(p3.x), [y3] "v"(p3.y), [x4] "v"(p4.x), [y4] "v"(p4.y), [x5] "v"(p5.x), [y5] "v"(p5.y), [x6] "v"(p6.x), [y6] "v"(p6.y), [x7] "v"(p7.x), [y7] "v"(p7.y),
        [m0] "v"(m0), [m1] "v"(m1), [m2] "v"(m2), [m3] "v"(m3), [m4] "v"(m4), [m5] "v"(m5), [m6] "v"(m6), [m7] "v"(m7),
        [c] "v"(clamp)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  }
}
#endif  // NBODY_LAB

// Eight FAST pairs for one target with EVERY multiply-add packed over two pairs (round 3; USE_ASM = 2, the default).
// Measured (profiles/r01_valu_microbench_wallclock.txt, r03_pair_body_packed.txt): a wave issues one VALU instruction per
// 4-cycle slot; plain f32 ops reach 2 cycles only when two waves' plain ops share a slot, which a body that mixes plain,
// packed and transcendental ops hardly ever achieves (30.4 cycles per pair for the 7-instruction block above against 24
// for the sum of its parts), whereas a packed op fills its slot alone.  So the sources of a tile sit in LDS as COUPLES
// {xA, xB, yA, yB}; the block forms (xA - tx, xB - tx) and (yA - ty, yB - ty) (op_sel picks the target's coordinate for both
// halves), squares, denominators and the accumulation are v_pk_mul / v_pk_fma over the couple, and only |dx| + |dy| (no abs
// modifier on packed ops) and the reciprocal stay one per pair: per couple 7 packed + 2 adds + 2 v_rcp = 13 slots = 26
// cycles per pair.  The accumulators hold the even and the odd sources' sums side by side; the caller adds the halves.
// c0..c3 come in as couples of source positions and leave as the differences.  Temporaries: couple k has Q = v[40+4k:41+4k]
// (squared distances), S = v[42+4k:43+4k] (sum, denominator, reciprocal).  The eight plain adds issue at wave priority 0, the rest
// at 1, so that the adds of different waves pair up in a slot (see direct_stream below).  bias2 = {2^-90, 2^-90} in an SGPR pair (free on a
// packed op).  With per-body masses mi0..mi3 are couples of INVERSE masses and scale the denominators.
template <bool UNIFORM, bool NOCLAMP>
__device__ __forceinline__ void fast_block8p(v2f t, float clamp, unsigned long long bias2, v2f& x0, v2f& y0, v2f& x1, v2f& y1, v2f& x2,
                                             v2f& y2, v2f& x3, v2f& y3, v2f mi0, v2f mi1, v2f mi2, v2f mi3, v2f& accx, v2f& accy) {
#define NB_SUBX(X) "v_pk_add_f32 %[" #X "], %[" #X "], %[t] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define NB_SUBY(Y) "v_pk_add_f32 %[" #Y "], %[" #Y "], %[t] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
  asm volatile(NB_SUBX(x0) NB_SUBY(y0) NB_SUBX(x1) NB_SUBY(y1) NB_SUBX(x2) NB_SUBY(y2) NB_SUBX(x3) NB_SUBY(y3)
               : [x0] "+v"(x0), [y0] "+v"(y0), [x1] "+v"(x1), [y1] "+v"(y1), [x2] "+v"(x2), [y2] "+v"(y2), [x3] "+v"(x3), [y3] "+v"(y3)
               : [t] "v"(t));
#undef NB_SUBX
#undef NB_SUBY
#define NB_SQ(K, Q) "v_pk_mul_f32 v[" Q "], %[x" #K "], %[x" #K "]\n\t"
#define NB_SQ2(K, Q) "v_pk_fma_f32 v[" Q "], %[y" #K "], %[y" #K "], v[" Q "]\n\t"
#define NB_SUM(K, SL, SH) "v_add_f32 v" SL ", |%[x" #K "l]|, |%[y" #K "l]|\n\tv_add_f32 v" SH ", |%[x" #K "h]|, |%[y" #K "h]|\n\t"
#define NB_MAX(QL, QH) "v_max_f32 v" QL ", v" QL ", %[c]\n\tv_max_f32 v" QH ", v" QH ", %[c]\n\t"
#define NB_DEN(S, Q) "v_pk_fma_f32 v[" S "], v[" S "], v[" Q "], %[b]\n\t"
#define NB_MINV(K, S) "v_pk_mul_f32 v[" S "], v[" S "], %[m" #K "]\n\t"
#define NB_RCP(SL, SH) "v_rcp_f32 v" SL ", v" SL "\n\tv_rcp_f32 v" SH ", v" SH "\n\t"
#define NB_ACC(K, S) "v_pk_fma_f32 %[ax], %[x" #K "], v[" S "], %[ax]\n\tv_pk_fma_f32 %[ay], %[y" #K "], v[" S "], %[ay]\n\t"
#define NB_HEAD NB_SQ(0, "40:41") NB_SQ(1, "44:45") NB_SQ(2, "48:49") NB_SQ(3, "52:53") NB_SQ2(0, "40:41") NB_SQ2(1, "44:45") NB_SQ2(2, "48:49") NB_SQ2(3, "52:53") \
                "s_setprio 0\n\t" NB_SUM(0, "42", "43") NB_SUM(1, "46", "47") NB_SUM(2, "50", "51") NB_SUM(3, "54", "55") "s_setprio 1\n\t"
#define NB_CLAMP NB_MAX("40", "41") NB_MAX("44", "45") NB_MAX("48", "49") NB_MAX("52", "53")
#define NB_DENS NB_DEN("42:43", "40:41") NB_DEN("46:47", "44:45") NB_DEN("50:51", "48:49") NB_DEN("54:55", "52:53")
#define NB_MINVS NB_MINV(0, "42:43") NB_MINV(1, "46:47") NB_MINV(2, "50:51") NB_MINV(3, "54:55")
#define NB_TAIL NB_RCP("42", "43") NB_RCP("46", "47") NB_RCP("50", "51") NB_RCP("54", "55") NB_ACC(0, "42:43") NB_ACC(1, "46:47") NB_ACC(2, "50:51") NB_ACC(3, "54:55")
#define NB_OPS                                                                                                                      \
  [x0] "v"(x0), [y0] "v"(y0), [x1] "v"(x1), [y1] "v"(y1), [x2] "v"(x2), [y2] "v"(y2), [x3] "v"(x3), [y3] "v"(y3), [x0l] "v"(x0.x),   \
      [x0h] "v"(x0.y), [y0l] "v"(y0.x), [y0h] "v"(y0.y), [x1l] "v"(x1.x), [x1h] "v"(x1.y), [y1l] "v"(y1.x), [y1h] "v"(y1.y),        \
      [x2l] "v"(x2.x), [x2h] "v"(x2.y), [y2l] "v"(y2.x), [y2h] "v"(y2.y), [x3l] "v"(x3.x), [x3h] "v"(x3.y), [y3l] "v"(y3.x),        \
      [y3h] "v"(y3.y), [b] "s"(bias2)
#define NB_CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"
  if constexpr (UNIFORM && NOCLAMP) {
    asm volatile(NB_HEAD NB_DENS NB_TAIL : [ax] "+v"(accx), [ay] "+v"(accy) : NB_OPS : NB_CLOB);
  } else if constexpr (UNIFORM && !NOCLAMP) {
    asm volatile(NB_HEAD NB_CLAMP NB_DENS NB_TAIL : [ax] "+v"(accx), [ay] "+v"(accy) : NB_OPS, [c] "v"(clamp) : NB_CLOB);
  } else if constexpr (!UNIFORM && NOCLAMP) {
    asm volatile(NB_HEAD NB_DENS NB_MINVS NB_TAIL
                 : [ax] "+v"(accx), [ay] "+v"(accy)
                 : NB_OPS, [m0] "v"(mi0), [m1] "v"(mi1), [m2] "v"(mi2), [m3] "v"(mi3)
                 : NB_CLOB);
  } else {
    asm volatile(NB_HEAD NB_CLAMP NB_DENS NB_MINVS NB_TAIL
                 : [ax] "+v"(accx), [ay] "+v"(accy)
                 : NB_OPS, [c] "v"(clamp), [m0] "v"(mi0), [m1] "v"(mi1), [m2] "v"(mi2), [m3] "v"(mi3)
                 : NB_CLOB);
  }
#undef NB_SQ
#undef NB_SQ2
#undef NB_SUM
#undef NB_MAX
#undef NB_DEN
#undef NB_MINV
#undef NB_RCP
#undef NB_ACC
#undef NB_HEAD
#undef NB_CLAMP
#undef NB_DENS
#undef NB_MINVS
#undef NB_TAIL
#undef NB_OPS
#undef NB_CLOB
}

// One EXACT pair: src/main.rs:236-252 operation by operation.
__device__ __forceinline__ void exact_pair(float xi, float yi, float xj, float yj, float mj, float clamp, float& ax,
                                           float& ay) {
  float dx = xj - xi;                                     // :236
  float dy = yj - yi;
  float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);  // :238
  if (!__builtin_isnormal(sum)) return;                   // :241-243
  float distance = dx * dx + dy * dy;                     // :245
  distance = __builtin_fmaxf(distance, clamp);            // :247-249 (one v_max_f32; `distance` is never NaN after the is_normal test)
  float den = sum * distance;
  const float2 q = div_pair(dx * mj, dy * mj, den);       // :252 (div_pair.h: the two quotients, their multiply-adds packed)
  ax = ax + q.x;
  ay = ay + q.y;
}

// main.rs:419-423, no contraction (TU flag).
__device__ __forceinline__ void integrate_store(const DirectArgs& a, int t_local, float ax, float ay) {
  if (a.acc_out) a.acc_out[t_local] = make_float2(ax, ay);
  if (a.vel) {
    float2 v = a.vel[t_local];
    float2 p = a.pos_all[a.tgt_begin + t_local];
    v.x = v.x + ax * a.delta;
    v.y = v.y + ay * a.delta;
    float vx = v.x * a.delta, vy = v.y * a.delta;
    p.x = p.x + vx;
    p.y = p.y + vy;
    a.vel[t_local] = v;
    a.pos_out[t_local] = p;
  }
}

// ------------------------------------------------------------------------------------------------ FAST
// Block = 256 threads = 4 waves sharing one group of 64*TPT targets and splitting every 1024-source LDS tile
// four ways; blockIdx.y splits the sources further.  The partial sums of the four waves meet in LDS in fixed
// order; across blockIdx.y (and always when the near sources are added afterwards) they go to a.partial and
// direct_finish completes the step.  Bitwise reproducible.
// Two-level summation: each wave sums its 256-source share of a tile on its own and then adds it to the running
// total (one extra add per 256 pairs; at N = 1M the error drops from ~1e-4 to ~2e-6 of sum|term|).
template <int TPT, bool UNIFORM, bool NOCLAMP, int USE_ASM>  // USE_ASM: 0 the compiler's schedule, 1 the hand-ordered block, 2 the packed couples
__global__ __launch_bounds__(256) void direct_fast(const DirectArgs a) {
  if (!gate_open(a)) return;

  constexpr int WSPLIT = 4;
  constexpr int TGT_PER_BLOCK = 64 * TPT;
  constexpr int TILE = 1024;
  constexpr int SHARE = TILE / WSPLIT;
  constexpr int UNR = 8;

  const int lane = threadIdx.x & 63;
  const int ws = wave_id_uniform();
  const int t0 = blockIdx.x * TGT_PER_BLOCK;

  float xi[TPT], yi[TPT], ax[TPT], ay[TPT];
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    int t = t0 + k * 64 + lane;
    float2 p = (t < a.n_tgt) ? a.pos_all[a.tgt_begin + t] : make_float2(0.f, 0.f);
    xi[k] = p.x;
    yi[k] = p.y;
    ax[k] = 0.f;
    ay[k] = 0.f;
  }
  float clamp = a.clamp;
  if constexpr (USE_ASM != 0) asm volatile("v_mov_b32 %0, %1" : "=v"(clamp) : "s"(a.clamp));  // keep it in a VGPR
  // USE_ASM = 2: source s of the tile lives in couple s / 2 = {xA, xB, yA, yB}: x at float 4 (s / 2) + (s & 1), y two floats on
  constexpr bool COUPLES = USE_ASM == 2;
  auto tile_x = [](int s) { return COUPLES ? 4 * (s >> 1) + (s & 1) : 2 * s; };

  __shared__ __attribute__((aligned(16))) float2 tile_pos[TILE];
  __shared__ __attribute__((aligned(16))) float4 tile_mass4[UNIFORM ? 1 : TILE / 4];  // masses, four per element
  float* tile_mass = reinterpret_cast<float*>(tile_mass4);
  // the hand-ordered block multiplies the denominator by 1/m before the reciprocal (one class switch fewer than
  // multiplying the reciprocal by m afterwards); 1/0 = inf makes a zero-mass source contribute exactly 0
  __shared__ __attribute__((aligned(16))) float4 tile_minv4[(UNIFORM || USE_ASM == 0) ? 1 : TILE / 4];
  float* tile_minv = reinterpret_cast<float*>(tile_minv4);

  // sources of this grid split: [g0, g1)
  int gchunk = (a.n_src + (int)gridDim.y - 1) / (int)gridDim.y;
  gchunk = (gchunk + 31) & ~31;
  // Mass classes: the sources come ordered by class, each class padded to whole tiles, so a tile has ONE mass and the
  // equal-mass arithmetic applies: the class's mass rides in the add that joins a tile's partial sum to the running total
  // (an FMA instead of an add, once per 256 pairs: nothing per pair).  The splits then start on tile boundaries.
  const float* __restrict__ class_mass = UNIFORM ? a.tile_mass : nullptr;
  if (class_mass) gchunk = (gchunk + TILE - 1) & ~(TILE - 1);
  long g0l = (long)blockIdx.y * gchunk;
  const int g0 = g0l < a.n_src ? (int)g0l : a.n_src;
  const int g1 = (g0 + gchunk < a.n_src) ? g0 + gchunk : a.n_src;
  const float2* __restrict__ src = a.src_pos;

  for (int base = g0; base < g1; base += TILE) {
    const int cnt = (g1 - base < TILE) ? g1 - base : TILE;
    const float tm = class_mass ? class_mass[base / TILE] : 1.0f;  // (1.0: fma(b, 1, a) is a + b, bit for bit)
#pragma unroll
    for (int r = 0; r < TILE / 256; ++r) {
      int s = r * 256 + (int)threadIdx.x;
      float2 p = make_float2(0.f, 0.f);
      float m = 0.f;  // padding of the last tile: zero mass (per-body masses) / never read (UNIFORM)
      if (s < cnt) {
        p = src[base + s];
        if (!UNIFORM) m = a.mass_all[base + s];
      }
      float* tile_f = reinterpret_cast<float*>(tile_pos);
      if (COUPLES && a.src_couples) {  // the far copy is in couples already: its image
        tile_pos[s] = p;
      } else {
        tile_f[tile_x(s)] = p.x;
        tile_f[tile_x(s) + (COUPLES ? 2 : 1)] = p.y;
      }
      if (!UNIFORM) tile_mass[s] = m;
      if (!UNIFORM && USE_ASM != 0) tile_minv[s] = 1.0f / m;
    }
    __syncthreads();
    const int lo = ws * SHARE;
    int hi = lo + SHARE;
    if (hi > cnt) hi = cnt > lo ? cnt : lo;
    int u = lo;
    if constexpr (USE_ASM == 2) {
      static_assert(TPT == 1, "the packed block handles one target per thread");
      // the wave's share of the tile, four couples per block; the accumulators keep the even and the odd sources' sums apart
      v2f accx = {0.f, 0.f}, accy = {0.f, 0.f};
      const v2f tgt = {xi[0], yi[0]};
      unsigned long long bias2 = 0x1280000012800000ull;  // {2^-90, 2^-90}
      asm volatile("" : "+s"(bias2));                    // (an SGPR pair, set once: not a literal per use)
      __builtin_amdgcn_s_setprio(1);                     // the block's plain adds run at priority 0 (see direct_stream)
      for (; u + UNR <= hi; u += UNR) {
        const v4f* src4 = reinterpret_cast<const v4f*>(&tile_pos[u]);
        const v4f s0 = src4[0], s1 = src4[1], s2 = src4[2], s3 = src4[3];
        v2f x0 = s0.xy, y0 = s0.zw, x1 = s1.xy, y1 = s1.zw, x2 = s2.xy, y2 = s2.zw, x3 = s3.xy, y3 = s3.zw;
        v4f ma = {1.f, 1.f, 1.f, 1.f}, mb = ma;
        if constexpr (!UNIFORM) {  // u is a multiple of 8: two ds_read_b128 fetch the eight inverse masses
          ma = reinterpret_cast<const v4f*>(tile_minv4)[(u >> 2)];
          mb = reinterpret_cast<const v4f*>(tile_minv4)[(u >> 2) + 1];
        }
        fast_block8p<UNIFORM, NOCLAMP>(tgt, clamp, bias2, x0, y0, x1, y1, x2, y2, x3, y3, ma.xy, ma.zw, mb.xy, mb.zw, accx, accy);
      }
      float bx = accx.x + accx.y, by = accy.x + accy.y;
      const float* tile_f = reinterpret_cast<const float*>(tile_pos);
      for (; u < hi; ++u) {
        float m = UNIFORM ? 1.0f : tile_mass[u];
        fast_pair<UNIFORM, NOCLAMP>(xi[0], yi[0], tile_f[tile_x(u)], tile_f[tile_x(u) + 2], m, clamp, bx, by);
      }
      ax[0] = __builtin_fmaf(bx, tm, ax[0]);
      ay[0] = __builtin_fmaf(by, tm, ay[0]);
#ifdef NBODY_LAB
    } else if constexpr (USE_ASM == 1) {
      static_assert(TPT == 1, "the hand-ordered block handles one target per thread");
      // the wave's share of the tile, 8 sources per block, summed on its own (two-level summation); everything the
      // block touches stays in 64-bit register pairs so that no repacking moves surround the asm
      v2f accb = {0.f, 0.f};
      const v2f tgt = {xi[0], yi[0]};
      for (; u + UNR <= hi; u += UNR) {
        const v4f* src4 = reinterpret_cast<const v4f*>(&tile_pos[u]);
        const v4f s0 = src4[0], s1 = src4[1], s2 = src4[2], s3 = src4[3];
        v2f q0 = s0.xy, q1 = s0.zw, q2 = s1.xy, q3 = s1.zw, q4 = s2.xy, q5 = s2.zw, q6 = s3.xy, q7 = s3.zw;
        float4 ma = make_float4(1.f, 1.f, 1.f, 1.f), mb = ma;
        if constexpr (!UNIFORM) {  // u is a multiple of 8: two ds_read_b128 fetch the eight inverse masses
          ma = tile_minv4[(u >> 2)];
          mb = tile_minv4[(u >> 2) + 1];
        }
        fast_block8<UNIFORM, NOCLAMP>(tgt, clamp, q0, q1, q2, q3, q4, q5, q6, q7, ma.x, ma.y, ma.z, ma.w, mb.x, mb.y,
                                      mb.z, mb.w, accb);
      }
      float bx = accb.x, by = accb.y;
      for (; u < hi; ++u) {
        float2 p = tile_pos[u];
        float m = UNIFORM ? 1.0f : tile_mass[u];
        fast_pair<UNIFORM, NOCLAMP>(xi[0], yi[0], p.x, p.y, m, clamp, bx, by);
      }
      ax[0] = __builtin_fmaf(bx, tm, ax[0]);
      ay[0] = __builtin_fmaf(by, tm, ay[0]);
#endif
    } else {
      float bx[TPT], by[TPT];
#pragma unroll
      for (int k = 0; k < TPT; ++k) bx[k] = by[k] = 0.f;
      for (; u + UNR <= hi; u += UNR) {
        float4 pp[UNR / 2];
        float2 mm[UNR / 2];
#pragma unroll
        for (int h = 0; h < UNR / 2; ++h) pp[h] = *reinterpret_cast<const float4*>(&tile_pos[u + 2 * h]);
        if constexpr (!UNIFORM) {
          const float4 ma = tile_mass4[(u >> 2)], mb = tile_mass4[(u >> 2) + 1];
          mm[0] = make_float2(ma.x, ma.y); mm[1] = make_float2(ma.z, ma.w);
          mm[2] = make_float2(mb.x, mb.y); mm[3] = make_float2(mb.z, mb.w);
        } else {
#pragma unroll
          for (int h = 0; h < UNR / 2; ++h) mm[h] = make_float2(1.f, 1.f);
        }
#pragma unroll
        for (int h = 0; h < UNR / 2; ++h)
#pragma unroll
          for (int k = 0; k < TPT; ++k) {
            fast_pair<UNIFORM, NOCLAMP>(xi[k], yi[k], pp[h].x, pp[h].y, mm[h].x, clamp, bx[k], by[k]);
            fast_pair<UNIFORM, NOCLAMP>(xi[k], yi[k], pp[h].z, pp[h].w, mm[h].y, clamp, bx[k], by[k]);
          }
      }
      for (; u < hi; ++u) {
        float2 p = tile_pos[u];
        float m = UNIFORM ? 1.0f : tile_mass[u];
#pragma unroll
        for (int k = 0; k < TPT; ++k) fast_pair<UNIFORM, NOCLAMP>(xi[k], yi[k], p.x, p.y, m, clamp, bx[k], by[k]);
      }
#pragma unroll
      for (int k = 0; k < TPT; ++k) {
        ax[k] = __builtin_fmaf(bx[k], tm, ax[k]);
        ay[k] = __builtin_fmaf(by[k], tm, ay[k]);
      }
    }
    __syncthreads();
  }

  if constexpr (UNIFORM) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      ax[k] *= a.uniform_mass;
      ay[k] *= a.uniform_mass;
    }
  }

  // ---- the four waves' partial sums, in fixed order
  __shared__ float2 red[WSPLIT - 1][TGT_PER_BLOCK];
  if (ws > 0) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) red[ws - 1][k * 64 + lane] = make_float2(ax[k], ay[k]);
  }
  __syncthreads();
  if (ws > 0) return;
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
#pragma unroll
    for (int w = 0; w < WSPLIT - 1; ++w) {
      float2 r = red[w][k * 64 + lane];
      ax[k] += r.x;
      ay[k] += r.y;
    }
    int t = t0 + k * 64 + lane;
    if (t >= a.n_tgt) continue;
    if (a.to_partial) a.partial[(size_t)blockIdx.y * a.n_tgt + t] = make_float2(ax[k], ay[k]);
    else integrate_store(a, t, ax[k], ay[k]);
  }
}

// ------------------------------------------------------------------------------------------------ FAST, streamed
// The far sources of the near/far split (equal masses or mass classes, no clamp) straight from memory through SGPRs: a packed
// op takes an SGPR pair as a source at no cost (profiles/r03_pair_body_packed.txt, P1s), so the couples {xA, xB, yA, yB} that
// nearfar.hip writes arrive by s_load_dwordx16 (four couples = one 8-pair block) and feed the subtractions directly.  No LDS
// tile, no ds_read, no barrier, no tile prologue: what a wave issues per block is the 52 slots of the block itself.  The loop
// is one asm statement with fixed registers (the compiler cannot see an asm's scalar load in flight, so nothing of it is left
// to the compiler): two register sets, the next block's load issued as soon as the current one has arrived, the last (unused)
// prefetch drained before the statement ends.  It reads one block past the range it is given: the far copy carries that slack.
// Same arithmetic, same order of additions as direct_fast<1, true, true, 2>: the two give the same bits.
// Wave priority: the eight plain adds of a block (|dx| + |dy|: packed ops have no abs modifier) issue at priority 0, everything
// else at 1.  A lone plain op takes a whole 4-cycle slot; two waves' plain ops can share one, but the arbiter only pairs
// them when no wave has anything else to issue — at the lower priority the adds wait until every wave of the SIMD is at
// its adds, and then go two to a slot: 13 -> 12 slots per couple (profiles/r03_pair_body_packed.txt, Q1: 26.1 -> 24.7 cycles).
#define NB_S_SUBX(D, S) "v_pk_add_f32 v[" D "], s[" S "], %[t] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define NB_S_SUBY(D, S) "v_pk_add_f32 v[" D "], s[" S "], %[t] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define NB_S_COUPLE_A(X, Y, Q) "v_pk_mul_f32 v[" Q "], v[" X "], v[" X "]\n\tv_pk_fma_f32 v[" Q "], v[" Y "], v[" Y "], v[" Q "]\n\t"
#define NB_S_COUPLE_S(XL, XH, YL, YH, SL, SH) "v_add_f32 v" SL ", |v" XL "|, |v" YL "|\n\tv_add_f32 v" SH ", |v" XH "|, |v" YH "|\n\t"
#define NB_S_COUPLE_B(X, Y, Q, S, SL, SH)                                                                  \
  "v_pk_fma_f32 v[" S "], v[" S "], v[" Q "], %[b]\n\tv_rcp_f32 v" SL ", v" SL "\n\tv_rcp_f32 v" SH ", v" SH "\n\t"
#define NB_S_COUPLE_C(X, Y, S) "v_pk_fma_f32 %[ax], v[" X "], v[" S "], %[ax]\n\tv_pk_fma_f32 %[ay], v[" Y "], v[" S "], %[ay]\n\t"
// one 8-pair block from the SGPR set that starts at s[B]: differences in v[24:39], Q / S of couple k in v[40+4k:43+4k]
#define NB_S_BLOCK(S0, S1, S2, S3, S4, S5, S6, S7)                                                                                      \
  NB_S_SUBX("24:25", S0) NB_S_SUBY("26:27", S1) NB_S_SUBX("28:29", S2) NB_S_SUBY("30:31", S3) NB_S_SUBX("32:33", S4) NB_S_SUBY("34:35", S5) \
  NB_S_SUBX("36:37", S6) NB_S_SUBY("38:39", S7)                                                                                          \
  NB_S_COUPLE_A("24:25", "26:27", "40:41") NB_S_COUPLE_A("28:29", "30:31", "44:45") NB_S_COUPLE_A("32:33", "34:35", "48:49") NB_S_COUPLE_A("36:37", "38:39", "52:53") \
  "s_setprio 0\n\t"                                                                                                                    \
  NB_S_COUPLE_S("24", "25", "26", "27", "42", "43") NB_S_COUPLE_S("28", "29", "30", "31", "46", "47")                                    \
  NB_S_COUPLE_S("32", "33", "34", "35", "50", "51") NB_S_COUPLE_S("36", "37", "38", "39", "54", "55")                                    \
  "s_setprio 1\n\t"                                                                                                                    \
  NB_S_COUPLE_B("24:25", "26:27", "40:41", "42:43", "42", "43") NB_S_COUPLE_B("28:29", "30:31", "44:45", "46:47", "46", "47")             \
  NB_S_COUPLE_B("32:33", "34:35", "48:49", "50:51", "50", "51") NB_S_COUPLE_B("36:37", "38:39", "52:53", "54:55", "54", "55")             \
  NB_S_COUPLE_C("24:25", "26:27", "42:43") NB_S_COUPLE_C("28:29", "30:31", "46:47") NB_S_COUPLE_C("32:33", "34:35", "50:51") NB_S_COUPLE_C("36:37", "38:39", "54:55")
// n16 >= 1 iterations of 16 sources from `src` (wave-uniform, 64-byte aligned)
__device__ __forceinline__ void stream_chunk(v2f t, unsigned long long bias2, const void* src, int n16, v2f& accx, v2f& accy) {
  asm volatile(
      "s_mov_b64 s[68:69], %[p]\n\t"
      "s_mov_b32 s70, %[n]\n\t"
      "s_load_dwordx16 s[36:51], s[68:69], 0x0\n"
      ".Lnb_stream_%=:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[52:67], s[68:69], 0x40\n\t"
      NB_S_BLOCK("36:37", "38:39", "40:41", "42:43", "44:45", "46:47", "48:49", "50:51")
      "s_add_u32 s68, s68, 0x80\n\t"
      "s_addc_u32 s69, s69, 0\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[36:51], s[68:69], 0x0\n\t"
      NB_S_BLOCK("52:53", "54:55", "56:57", "58:59", "60:61", "62:63", "64:65", "66:67")
      "s_sub_u32 s70, s70, 1\n\t"
      "s_cmp_lg_u32 s70, 0\n\t"
      "s_cbranch_scc1 .Lnb_stream_%=\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : [ax] "+v"(accx), [ay] "+v"(accy)
      : [t] "v"(t), [b] "s"(bias2), [p] "s"(src), [n] "s"(n16)
      : "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",
        "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "s36", "s37", "s38", "s39", "s40", "s41",
        "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60",
        "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "scc", "memory");
}
// The same loop for FREE per-body masses (more than 32 distinct u32 weights: neither the equal-mass hoist nor the mass classes
// apply — main.rs:193-198 gives every particle its own `weight`, :360 uses it as f32).  The inverse masses of a block's eight
// sources arrive beside its couples by one s_load_dwordx8 from the far copy's second array (nearfar.hip writes 1/m in slot order:
// couple k's pair is {1/mA, 1/mB}, adjacent), and scale the denominators before the reciprocal — s = 1 / (den / m), one packed
// multiply on an SGPR pair per couple: 13 slots per couple against the equal-mass loop's 12.  1/0 = inf keeps a zero-mass source
// at exactly 0.  Same arithmetic and order of additions as direct_fast<1, false, true, 2> (fast_block8p's NB_MINVS): the same bits.
#define NB_S_COUPLE_D(Q, S) "v_pk_fma_f32 v[" S "], v[" S "], v[" Q "], %[b]\n\t"
#define NB_S_COUPLE_M(S, M) "v_pk_mul_f32 v[" S "], v[" S "], s[" M "]\n\t"
#define NB_S_COUPLE_R(SL, SH) "v_rcp_f32 v" SL ", v" SL "\n\tv_rcp_f32 v" SH ", v" SH "\n\t"
#define NB_S_BLOCK_M(S0, S1, S2, S3, S4, S5, S6, S7, M0, M1, M2, M3)                                                                     \
  NB_S_SUBX("24:25", S0) NB_S_SUBY("26:27", S1) NB_S_SUBX("28:29", S2) NB_S_SUBY("30:31", S3) NB_S_SUBX("32:33", S4) NB_S_SUBY("34:35", S5) \
  NB_S_SUBX("36:37", S6) NB_S_SUBY("38:39", S7)                                                                                          \
  NB_S_COUPLE_A("24:25", "26:27", "40:41") NB_S_COUPLE_A("28:29", "30:31", "44:45") NB_S_COUPLE_A("32:33", "34:35", "48:49") NB_S_COUPLE_A("36:37", "38:39", "52:53") \
  "s_setprio 0\n\t"                                                                                                                    \
  NB_S_COUPLE_S("24", "25", "26", "27", "42", "43") NB_S_COUPLE_S("28", "29", "30", "31", "46", "47")                                    \
  NB_S_COUPLE_S("32", "33", "34", "35", "50", "51") NB_S_COUPLE_S("36", "37", "38", "39", "54", "55")                                    \
  "s_setprio 1\n\t"                                                                                                                    \
  NB_S_COUPLE_D("40:41", "42:43") NB_S_COUPLE_D("44:45", "46:47") NB_S_COUPLE_D("48:49", "50:51") NB_S_COUPLE_D("52:53", "54:55")         \
  NB_S_COUPLE_M("42:43", M0) NB_S_COUPLE_R("42", "43") NB_S_COUPLE_M("46:47", M1) NB_S_COUPLE_R("46", "47")                             \
  NB_S_COUPLE_M("50:51", M2) NB_S_COUPLE_R("50", "51") NB_S_COUPLE_M("54:55", M3) NB_S_COUPLE_R("54", "55")                             \
  NB_S_COUPLE_C("24:25", "26:27", "42:43") NB_S_COUPLE_C("28:29", "30:31", "46:47") NB_S_COUPLE_C("32:33", "34:35", "50:51") NB_S_COUPLE_C("36:37", "38:39", "54:55")
// n16 >= 1 iterations of 16 sources: couples from `src` (64-byte aligned), inverse masses from `minv` (32-byte aligned), both
// wave-uniform; both arrays carry one block of slack past the range (the last prefetch is issued and drained, never used)
__device__ __forceinline__ void stream_chunk_m(v2f t, unsigned long long bias2, const void* src, const void* minv, int n16, v2f& accx,
                                               v2f& accy) {
  asm volatile(
      "s_mov_b64 s[68:69], %[p]\n\t"
      "s_mov_b64 s[88:89], %[q]\n\t"
      "s_mov_b32 s70, %[n]\n\t"
      "s_load_dwordx16 s[36:51], s[68:69], 0x0\n\t"
      "s_load_dwordx8 s[72:79], s[88:89], 0x0\n"
      ".Lnb_stream_m_%=:\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[52:67], s[68:69], 0x40\n\t"
      "s_load_dwordx8 s[80:87], s[88:89], 0x20\n\t"
      NB_S_BLOCK_M("36:37", "38:39", "40:41", "42:43", "44:45", "46:47", "48:49", "50:51", "72:73", "74:75", "76:77", "78:79")
      "s_add_u32 s68, s68, 0x80\n\t"
      "s_addc_u32 s69, s69, 0\n\t"
      "s_add_u32 s88, s88, 0x40\n\t"
      "s_addc_u32 s89, s89, 0\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 s[36:51], s[68:69], 0x0\n\t"
      "s_load_dwordx8 s[72:79], s[88:89], 0x0\n\t"
      NB_S_BLOCK_M("52:53", "54:55", "56:57", "58:59", "60:61", "62:63", "64:65", "66:67", "80:81", "82:83", "84:85", "86:87")
      "s_sub_u32 s70, s70, 1\n\t"
      "s_cmp_lg_u32 s70, 0\n\t"
      "s_cbranch_scc1 .Lnb_stream_m_%=\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      : [ax] "+v"(accx), [ay] "+v"(accy)
      : [t] "v"(t), [b] "s"(bias2), [p] "s"(src), [q] "s"(minv), [n] "s"(n16)
      : "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",
        "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "s36", "s37", "s38", "s39", "s40", "s41",
        "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60",
        "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80",
        "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "scc", "memory");
}
#undef NB_S_COUPLE_D
#undef NB_S_COUPLE_M
#undef NB_S_COUPLE_R
#undef NB_S_BLOCK_M
#undef NB_S_SUBX
#undef NB_S_SUBY
#undef NB_S_COUPLE_A
#undef NB_S_COUPLE_S
#undef NB_S_COUPLE_B
#undef NB_S_COUPLE_C
#undef NB_S_BLOCK

// Block = 4 waves sharing 64 targets; wave w takes sources [256 w, 256 w + 256) of every 1024-source tile of its grid split —
// the assignment and the two-level summation of direct_fast, so the sums are the same.  PER_MASS: direct_stream_m, the inverse
// masses streamed beside the couples (a.src_minv, slot order).
template <bool PER_MASS>
__device__ __forceinline__ void direct_stream_body(const DirectArgs& a) {
  constexpr int TILE = 1024, SHARE = 256;
  const int lane = threadIdx.x & 63;
  const int ws = wave_id_uniform();
  const int t = blockIdx.x * 64 + lane;
  const float2 pt = (t < a.n_tgt) ? a.pos_all[a.tgt_begin + t] : make_float2(0.f, 0.f);
  const v2f tgt = {pt.x, pt.y};
  float ax = 0.f, ay = 0.f;
  int gchunk = (a.n_src + (int)gridDim.y - 1) / (int)gridDim.y;
  gchunk = (gchunk + 31) & ~31;
  const float* __restrict__ class_mass = PER_MASS ? nullptr : a.tile_mass;
  if (class_mass) gchunk = (gchunk + TILE - 1) & ~(TILE - 1);
  long g0l = (long)blockIdx.y * gchunk;
  const int g0 = g0l < a.n_src ? (int)g0l : a.n_src;
  const int g1 = (g0 + gchunk < a.n_src) ? g0 + gchunk : a.n_src;
  unsigned long long bias2 = 0x1280000012800000ull;  // {2^-90, 2^-90}
  __builtin_amdgcn_s_setprio(1);
  for (int base = g0; base < g1; base += TILE) {
    const int cnt = (g1 - base < TILE) ? g1 - base : TILE;
    const float tm = class_mass ? class_mass[base / TILE] : 1.0f;
    const int lo = ws * SHARE;
    int hi = lo + SHARE;
    if (hi > cnt) hi = cnt > lo ? cnt : lo;
    v2f accx = {0.f, 0.f}, accy = {0.f, 0.f};
    const int n16 = __builtin_amdgcn_readfirstlane((hi - lo) >> 4);  // (n_src, hence every bound here, is a multiple of 16)
    if (n16 > 0) {
      const char* couples = reinterpret_cast<const char*>(a.src_pos) + (size_t)(base + lo) * 8;
      if constexpr (PER_MASS) stream_chunk_m(tgt, bias2, couples, reinterpret_cast<const char*>(a.src_minv) + (size_t)(base + lo) * 4, n16, accx, accy);
      else stream_chunk(tgt, bias2, couples, n16, accx, accy);
    }
    const float bx = accx.x + accx.y, by = accy.x + accy.y;
    ax = __builtin_fmaf(bx, tm, ax);
    ay = __builtin_fmaf(by, tm, ay);
  }
  if constexpr (!PER_MASS) {
    ax *= a.uniform_mass;
    ay *= a.uniform_mass;
  }
  __shared__ float2 red[3][64];
  if (ws > 0) red[ws - 1][lane] = make_float2(ax, ay);
  __syncthreads();
  if (ws > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    float2 r = red[w][lane];
    ax += r.x;
    ay += r.y;
  }
  if (t >= a.n_tgt) return;
  if (a.to_partial) a.partial[(size_t)blockIdx.y * a.n_tgt + t] = make_float2(ax, ay);
  else integrate_store(a, t, ax, ay);
}
__global__ __launch_bounds__(256) void direct_stream(const DirectArgs a) {
  if (!gate_open(a)) return;
  direct_stream_body<false>(a);
}
__global__ __launch_bounds__(256) void direct_stream_m(const DirectArgs a) {
  if (!gate_open(a)) return;
  direct_stream_body<true>(a);
}

// Completes a step whose main pass wrote partial sums: adds the grid-split partials in ascending split order,
// then (near/far split) the near sources with the clamp, in ascending body index, then integrates.
__global__ __launch_bounds__(256) void direct_finish(const DirectArgs a, int n_gsplit, int add_near) {
  if (!gate_open(a)) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < a.n_tgt;
  float ax = 0.f, ay = 0.f;
  if (live)
    for (int g = 0; g < n_gsplit; ++g) {
      float2 r = a.partial[(size_t)g * a.n_tgt + t];
      ax += r.x;
      ay += r.y;
    }
  if (add_near) {
    // the near sources (ascending body index) come through LDS, 256 at a time: read straight from memory each target pays
    // three dependent loads per source (index, position, mass), ~1 us of latency apiece
    __shared__ float2 npos[256];
    __shared__ float nmass[256];
    const int m = a.flags[kFlagNearCount];
    const float2 pi = live ? a.pos_all[a.tgt_begin + t] : make_float2(0.f, 0.f);
    float nx = 0.f, ny = 0.f;
    for (int base = 0; base < m; base += 256) {
      const int cnt = m - base < 256 ? m - base : 256;
      if ((int)threadIdx.x < cnt) {
        const uint32_t j = a.near_list[base + threadIdx.x];
        npos[threadIdx.x] = a.pos_all[j];
        nmass[threadIdx.x] = a.mass_all[j];
      }
      __syncthreads();
      for (int q = 0; q < cnt; ++q) fast_pair<false, false>(pi.x, pi.y, npos[q].x, npos[q].y, nmass[q], a.clamp, nx, ny);
      __syncthreads();
    }
    ax += nx;
    ay += ny;
  }
  if (live) integrate_store(a, t, ax, ay);
}

// ------------------------------------------------------------------------------------------------ EXACT
__global__ __launch_bounds__(256) void direct_exact(const DirectArgs a) {
  if (!gate_open(a)) return;
  int t = blockIdx.x * 256 + threadIdx.x;
  float2 pi = (t < a.n_tgt) ? a.pos_all[a.tgt_begin + t] : make_float2(0.f, 0.f);
  float ax = 0.f, ay = 0.f;
  const float2* __restrict__ ps = a.pos_all;
  const float* __restrict__ ms = a.mass_all;
  const float clamp = a.clamp;
  int j = 0;
  for (; j + 4 <= a.n_src; j += 4) {
    float2 p[4];
    float m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      p[u] = ps[j + u];
      m[u] = ms[j + u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) exact_pair(pi.x, pi.y, p[u].x, p[u].y, m[u], clamp, ax, ay);
  }
  for (; j < a.n_src; ++j) exact_pair(pi.x, pi.y, ps[j].x, ps[j].y, ms[j], clamp, ax, ay);
  if (t < a.n_tgt) integrate_store(a, t, ax, ay);
}

// ------------------------------------------------------------------------------------------------ hazard scan
// flags[kFlagHazard] |= 1 when any coordinate is outside FAST's domain.
__global__ __launch_bounds__(256) void direct_hazard_scan(const float* __restrict__ xy, long n_floats, int* flags) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long stride = (long)gridDim.x * 256;
  int bad = 0;
  for (; i < n_floats; i += stride) {
    float v = __builtin_fabsf(xy[i]);
    bad |= !(v < kBig) || (v != 0.f && v < kTiny);
  }
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(&flags[kFlagHazard], 1);
}

__global__ __launch_bounds__(256) void weights_to_mass(const uint32_t* __restrict__ w, float* __restrict__ m, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) m[i] = (float)w[i];  // `weight as f32`, main.rs:360 (round to nearest even)
}

// ------------------------------------------------------------------------------------------------ host launchers
template <int TPT, bool UNIFORM, bool NOCLAMP, int USE_ASM>
static void launch_fast_k(hipStream_t s, const DirectArgs& a, int n_gsplit) {
  dim3 grid((unsigned)((a.n_tgt + 64 * TPT - 1) / (64 * TPT)), (unsigned)n_gsplit);
  hipLaunchKernelGGL((direct_fast<TPT, UNIFORM, NOCLAMP, USE_ASM>), grid, dim3(256), 0, s, a);
}

hipError_t launch_direct_fast(hipStream_t s, const DirectArgs& a, const DirectConfig& c, bool noclamp) {
  if (a.n_tgt <= 0) return hipSuccess;
  const bool uni = a.uniform_mass > 0.f;
  const int tpt = c.tpt == 2 ? 2 : 1;
  const int use_asm = tpt == 1 ? c.use_asm : 0;
  if (a.src_couples && (use_asm < 2 || (a.n_src % kFarPad) != 0)) return hipErrorInvalidValue;  // only the packed kernels read couples
  if (use_asm == 3 && noclamp && a.src_couples && (uni || a.src_minv)) {
    const dim3 grid((unsigned)((a.n_tgt + 63) / 64), (unsigned)c.gsplit);
    if (uni) hipLaunchKernelGGL(direct_stream, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(direct_stream_m, grid, dim3(256), 0, s, a);
    return hipGetLastError();
  }
#define NB_GO(T, U, N, A) launch_fast_k<T, U, N, A>(s, a, c.gsplit)
#ifdef NBODY_LAB  // the rounds 1-2 main passes (NBODY_DIRECT_ASM 0 / 1, NBODY_DIRECT_TPT 2): A/B runs only
  if (tpt == 2) {
    if (uni) { if (noclamp) NB_GO(2, true, true, 0); else NB_GO(2, true, false, 0); }
    else     { if (noclamp) NB_GO(2, false, true, 0); else NB_GO(2, false, false, 0); }
    return hipGetLastError();
  }
  if (use_asm == 1) {
    if (uni) { if (noclamp) NB_GO(1, true, true, 1); else NB_GO(1, true, false, 1); }
    else     { if (noclamp) NB_GO(1, false, true, 1); else NB_GO(1, false, false, 1); }
    return hipGetLastError();
  }
  if (use_asm == 0) {
    if (uni) { if (noclamp) NB_GO(1, true, true, 0); else NB_GO(1, true, false, 0); }
    else     { if (noclamp) NB_GO(1, false, true, 0); else NB_GO(1, false, false, 0); }
    return hipGetLastError();
  }
#else
  if (tpt != 1 || use_asm < 2) return hipErrorInvalidValue;  // (choose_direct_config cannot ask for them in this build)
#endif
  // packed couples through LDS: the clamped single pass (small problems, fallbacks) and whatever the streamed kernels do not take
  if (uni) { if (noclamp) NB_GO(1, true, true, 2); else NB_GO(1, true, false, 2); }
  else     { if (noclamp) NB_GO(1, false, true, 2); else NB_GO(1, false, false, 2); }
#undef NB_GO
  return hipGetLastError();
}

hipError_t launch_direct_finish(hipStream_t s, const DirectArgs& a, int n_gsplit, bool add_near) {
  if (a.n_tgt <= 0) return hipSuccess;
  hipLaunchKernelGGL(direct_finish, dim3((unsigned)((a.n_tgt + 255) / 256)), dim3(256), 0, s, a, n_gsplit, add_near ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_direct_exact(hipStream_t s, const DirectArgs& a) {
  if (a.n_tgt <= 0) return hipSuccess;
  hipLaunchKernelGGL(direct_exact, dim3((unsigned)((a.n_tgt + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_hazard_scan(hipStream_t s, const float* xy, long n_floats, int* flags) {
  if (n_floats <= 0) return hipSuccess;
  long blocks = (n_floats + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(direct_hazard_scan, dim3((unsigned)blocks), dim3(256), 0, s, xy, n_floats, flags);
  return hipGetLastError();
}

hipError_t launch_weights_to_mass(hipStream_t s, const uint32_t* w, float* m, long n) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(weights_to_mass, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, m, n);
  return hipGetLastError();
}

}  // namespace nbody

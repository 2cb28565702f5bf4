// Direct O(N^2) force + semi-implicit Euler kernels for gfx950 (MI355X, wave64).
//
// What is computed:
//   a_i = sum_j calculate_gravity(p_i, p_j, w_j)        reference force law   src/main.rs:234-253
//   v_i += a_i*dt ; x_i += v_i*dt                        reference integrate   src/main.rs:419-423
//
// Two arithmetic flavours (include/nbody_hip.h, nbody_arith):
//   FAST   10 full-rate VALU ops + one v_rcp_f32 per pair.  Sources are wave-uniform, so they are fetched
//          with scalar loads into SGPRs (variant "sgpr": no LDS, no barrier in the loop) or staged as a
//          float4 tile in LDS and read back with broadcast ds_read_b128 (variant "lds", the classic tiling).
//          The coincident-pair skip of main.rs:241-243 costs nothing: den = fma(sum, d2c, 2^-90) keeps the
//          reciprocal finite, so a zero diff contributes exactly 0.  Valid when no position is non-finite,
//          >= 2^60, or non-zero below 2^-22 (then every non-zero |dx|+|dy| is >= 2^-46 and the 2^-90 bias is
//          below half an ulp of den); direct_hazard_scan checks exactly that, per call.
//   EXACT  every operation as the reference writes it: IEEE subtract/multiply/add/divide, no contraction,
//          the is_normal() skip, one sequential ascending-j chain per target.  Bit-identical to the oracle.
//
// This translation unit is compiled with -ffp-contract=off: nothing fuses unless written as fmaf().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "direct_kernels.h"

namespace nbody {

static constexpr float kDenBias = 8.0779356694631609e-28f;  // 2^-90
static constexpr float kBig = 1152921504606846976.0f;       // 2^60
static constexpr float kTiny = 2.384185791015625e-07f;      // 2^-22

__device__ __forceinline__ int wave_id_uniform() {
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// One FAST pair.  14 algorithmic flops: 2 sub, 1 add, mul+fma, max, fma, rcp, mul, 2 fma.
template <bool UNIFORM>
__device__ __forceinline__ void fast_pair(float xi, float yi, float xj, float yj, float mj, float clamp, float& ax,
                                          float& ay) {
  float dx = xj - xi;
  float dy = yj - yi;
  float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);
  float d2 = __builtin_fmaf(dy, dy, dx * dx);
  d2 = __builtin_fmaxf(d2, clamp);
  float den = __builtin_fmaf(sum, d2, kDenBias);
  float s = __builtin_amdgcn_rcpf(den);
  if (!UNIFORM) s = mj * s;
  ax = __builtin_fmaf(dx, s, ax);
  ay = __builtin_fmaf(dy, s, ay);
}

// One EXACT pair: src/main.rs:236-252 operation by operation.
__device__ __forceinline__ void exact_pair(float xi, float yi, float xj, float yj, float mj, float clamp, float& ax,
                                           float& ay) {
  float dx = xj - xi;                                     // :236
  float dy = yj - yi;
  float sum = __builtin_fabsf(dx) + __builtin_fabsf(dy);  // :238
  if (!__builtin_isnormal(sum)) return;                   // :241-243
  float distance = dx * dx + dy * dy;                     // :245
  if (distance < clamp) distance = clamp;                 // :247-249
  float den = sum * distance;
  ax = ax + (dx * mj) / den;                              // :252
  ay = ay + (dy * mj) / den;
}

// Eight FAST pairs for one target as one hand-ordered instruction block.  Same operations as fast_pair, but
// issued in phases (8 x v_pk_add | the 32-bit ops, v_max interleaved with v_add | 8 x v_rcp | 8 x v_pk_fma): on
// gfx950 changing between the packed, transcendental and plain VALU classes costs issue cycles
// (tools/mb/switch_bench) and hipcc's interleaved schedule of the same 64 instructions runs 7 % slower
// (tools/mb/body_bench: 37.7 vs 34.9 cycles per pair).  p0..p7 come in as source positions and leave as the
// differences.  Temporaries live in v40..v55: pair k = v[40+2k : 41+2k] = (s_k, d2_k), so that the pk_fma can
// broadcast s_k with op_sel_hi.  0x12800000 = 2^-90 (kDenBias).  VALU->VALU dependencies are interlocked by
// the hardware; every v_rcp result is consumed >= 8 instructions later (trans forwarding hazard needs 1).
template <bool UNIFORM>
__device__ __forceinline__ void fast_block8(float2 t, float clamp, float2& p0, float2& p1, float2& p2, float2& p3,
                                            float2& p4, float2& p5, float2& p6, float2& p7, float m0, float m1,
                                            float m2, float m3, float m4, float m5, float m6, float m7, float2& acc) {
  asm volatile(
      "v_pk_add_f32 %[p0], %[p0], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p1], %[p1], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p2], %[p2], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p3], %[p3], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p4], %[p4], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p5], %[p5], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p6], %[p6], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %[p7], %[p7], %[t] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [p5] "+v"(p5), [p6] "+v"(p6),
        [p7] "+v"(p7)
      : [t] "v"(t));
  if constexpr (UNIFORM) {
    asm volatile(
      "v_mul_f32 v41, %[x0], %[x0]\n\tv_mul_f32 v43, %[x1], %[x1]\n\tv_mul_f32 v45, %[x2], %[x2]\n\tv_mul_f32 v47, %[x3], %[x3]\n\tv_mul_f32 v49, %[x4], %[x4]\n\tv_mul_f32 v51, %[x5], %[x5]\n\tv_mul_f32 v53, %[x6], %[x6]\n\tv_mul_f32 v55, %[x7], %[x7]\n\t"
      "v_fmac_f32 v41, %[y0], %[y0]\n\tv_fmac_f32 v43, %[y1], %[y1]\n\tv_fmac_f32 v45, %[y2], %[y2]\n\tv_fmac_f32 v47, %[y3], %[y3]\n\tv_fmac_f32 v49, %[y4], %[y4]\n\tv_fmac_f32 v51, %[y5], %[y5]\n\tv_fmac_f32 v53, %[y6], %[y6]\n\tv_fmac_f32 v55, %[y7], %[y7]\n\t"
      "v_add_f32 v40, |%[x0]|, |%[y0]|\n\tv_max_f32 v41, v41, %[c]\n\tv_add_f32 v42, |%[x1]|, |%[y1]|\n\tv_max_f32 v43, v43, %[c]\n\tv_add_f32 v44, |%[x2]|, |%[y2]|\n\tv_max_f32 v45, v45, %[c]\n\tv_add_f32 v46, |%[x3]|, |%[y3]|\n\tv_max_f32 v47, v47, %[c]\n\tv_add_f32 v48, |%[x4]|, |%[y4]|\n\tv_max_f32 v49, v49, %[c]\n\tv_add_f32 v50, |%[x5]|, |%[y5]|\n\tv_max_f32 v51, v51, %[c]\n\tv_add_f32 v52, |%[x6]|, |%[y6]|\n\tv_max_f32 v53, v53, %[c]\n\tv_add_f32 v54, |%[x7]|, |%[y7]|\n\tv_max_f32 v55, v55, %[c]\n\t"
      "v_fmaak_f32 v40, v40, v41, 0x12800000\n\tv_fmaak_f32 v42, v42, v43, 0x12800000\n\tv_fmaak_f32 v44, v44, v45, 0x12800000\n\tv_fmaak_f32 v46, v46, v47, 0x12800000\n\tv_fmaak_f32 v48, v48, v49, 0x12800000\n\tv_fmaak_f32 v50, v50, v51, 0x12800000\n\tv_fmaak_f32 v52, v52, v53, 0x12800000\n\tv_fmaak_f32 v54, v54, v55, 0x12800000\n\t"
      "v_rcp_f32 v40, v40\n\tv_rcp_f32 v42, v42\n\tv_rcp_f32 v44, v44\n\tv_rcp_f32 v46, v46\n\tv_rcp_f32 v48, v48\n\tv_rcp_f32 v50, v50\n\tv_rcp_f32 v52, v52\n\tv_rcp_f32 v54, v54\n\t"
      "v_pk_fma_f32 %[a], %[p0], v[40:41], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p1], v[42:43], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p2], v[44:45], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p3], v[46:47], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p4], v[48:49], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p5], v[50:51], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p6], v[52:53], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p7], v[54:55], %[a] op_sel_hi:[1,0,1]\n\t"
      : [a] "+v"(acc)
      : [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
        [x0] "v"(p0.x), [y0] "v"(p0.y), [x1] "v"(p1.x), [y1] "v"(p1.y), [x2] "v"(p2.x), [y2] "v"(p2.y), [x3] "v"(p3.x), [y3] "v"(p3.y), [x4] "v"(p4.x), [y4] "v"(p4.y), [x5] "v"(p5.x), [y5] "v"(p5.y), [x6] "v"(p6.x), [y6] "v"(p6.y), [x7] "v"(p7.x), [y7] "v"(p7.y),
        [c] "v"(clamp)
      : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  } else {
    asm volatile(
      "v_mul_f32 v41, %[x0], %[x0]\n\tv_mul_f32 v43, %[x1], %[x1]\n\tv_mul_f32 v45, %[x2], %[x2]\n\tv_mul_f32 v47, %[x3], %[x3]\n\tv_mul_f32 v49, %[x4], %[x4]\n\tv_mul_f32 v51, %[x5], %[x5]\n\tv_mul_f32 v53, %[x6], %[x6]\n\tv_mul_f32 v55, %[x7], %[x7]\n\t"
      "v_fmac_f32 v41, %[y0], %[y0]\n\tv_fmac_f32 v43, %[y1], %[y1]\n\tv_fmac_f32 v45, %[y2], %[y2]\n\tv_fmac_f32 v47, %[y3], %[y3]\n\tv_fmac_f32 v49, %[y4], %[y4]\n\tv_fmac_f32 v51, %[y5], %[y5]\n\tv_fmac_f32 v53, %[y6], %[y6]\n\tv_fmac_f32 v55, %[y7], %[y7]\n\t"
      "v_add_f32 v40, |%[x0]|, |%[y0]|\n\tv_max_f32 v41, v41, %[c]\n\tv_add_f32 v42, |%[x1]|, |%[y1]|\n\tv_max_f32 v43, v43, %[c]\n\tv_add_f32 v44, |%[x2]|, |%[y2]|\n\tv_max_f32 v45, v45, %[c]\n\tv_add_f32 v46, |%[x3]|, |%[y3]|\n\tv_max_f32 v47, v47, %[c]\n\tv_add_f32 v48, |%[x4]|, |%[y4]|\n\tv_max_f32 v49, v49, %[c]\n\tv_add_f32 v50, |%[x5]|, |%[y5]|\n\tv_max_f32 v51, v51, %[c]\n\tv_add_f32 v52, |%[x6]|, |%[y6]|\n\tv_max_f32 v53, v53, %[c]\n\tv_add_f32 v54, |%[x7]|, |%[y7]|\n\tv_max_f32 v55, v55, %[c]\n\t"
      "v_fmaak_f32 v40, v40, v41, 0x12800000\n\tv_fmaak_f32 v42, v42, v43, 0x12800000\n\tv_fmaak_f32 v44, v44, v45, 0x12800000\n\tv_fmaak_f32 v46, v46, v47, 0x12800000\n\tv_fmaak_f32 v48, v48, v49, 0x12800000\n\tv_fmaak_f32 v50, v50, v51, 0x12800000\n\tv_fmaak_f32 v52, v52, v53, 0x12800000\n\tv_fmaak_f32 v54, v54, v55, 0x12800000\n\t"
      "v_rcp_f32 v40, v40\n\tv_rcp_f32 v42, v42\n\tv_rcp_f32 v44, v44\n\tv_rcp_f32 v46, v46\n\tv_rcp_f32 v48, v48\n\tv_rcp_f32 v50, v50\n\tv_rcp_f32 v52, v52\n\tv_rcp_f32 v54, v54\n\t"
      "v_mul_f32 v40, %[m0], v40\n\tv_mul_f32 v42, %[m1], v42\n\tv_mul_f32 v44, %[m2], v44\n\tv_mul_f32 v46, %[m3], v46\n\tv_mul_f32 v48, %[m4], v48\n\tv_mul_f32 v50, %[m5], v50\n\tv_mul_f32 v52, %[m6], v52\n\tv_mul_f32 v54, %[m7], v54\n\t"
      "v_pk_fma_f32 %[a], %[p0], v[40:41], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p1], v[42:43], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p2], v[44:45], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p3], v[46:47], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p4], v[48:49], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p5], v[50:51], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p6], v[52:53], %[a] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a], %[p7], v[54:55], %[a] op_sel_hi:[1,0,1]\n\t"
      : [a] "+v"(acc)
      : [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [p4] "v"(p4), [p5] "v"(p5), [p6] "v"(p6), [p7] "v"(p7),
        [x0] "v"(p0.x), [y0] "v"(p0.y), [x1] "v"(p1.x), [y1] "v"(p1.y), [x2] "v"(p2.x), [y2] "v"(p2.y), [x3] "v"(p3.x), [y3] "v"(p3.y), [x4] "v"(p4.x), [y4] "v"(p4.y), [x5] "v"(p5.x), [y5] "v"(p5.y), [x6] "v"(p6.x), [y6] "v"(p6.y), [x7] "v"(p7.x), [y7] "v"(p7.y),
        [m0] "v"(m0), [m1] "v"(m1), [m2] "v"(m2), [m3] "v"(m3), [m4] "v"(m4), [m5] "v"(m5), [m6] "v"(m6), [m7] "v"(m7),
        [c] "v"(clamp)
      : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  }
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
// Block = 256 threads = 4 waves.  WSPLIT waves share one group of 64*TPT targets and split the sources;
// 4/WSPLIT groups per block.  blockIdx.y splits the sources further (partials reduced by direct_finish).
//
// Cost model measured on MI355X (tools/valu_microbench*.hip; 1 slot = one full-rate wave64 VALU issue):
//   v_sub/v_mul/v_add/v_fmaak with VGPR+literal operands 1.0-1.1 | any SGPR operand 1.93 | v_max_f32 1.93 |
//   v_pk_add_f32 (SGPR pair - VGPR pair) 1.96 | v_pk_fma_f32 ~2.1 | v_rcp_f32 3.85.
// USE_LDS=false: sources arrive in SGPRs by scalar loads (no LDS, no barrier); the mass multiply then reads an
//   SGPR (half rate), so this flavour is best when UNIFORM (all masses equal: the multiply is hoisted out of
//   the sum; for mass 1 the result is bit-identical to multiplying every term by 1.0).
// USE_LDS=true: the classic tile in LDS, read back two sources per ds_read_b128 (positions) + ds_read_b64
//   (masses); all operands are VGPRs, which is what general masses want.
template <int TPT, int WSPLIT, bool USE_LDS, bool UNIFORM, bool USE_ASM>
__global__ __launch_bounds__(256) void direct_fast(const DirectArgs a) {
  if (a.gate && ((*a.gate != 0) != (a.run_if != 0))) return;

  constexpr int GROUPS = 4 / WSPLIT;
  constexpr int TGT_PER_GROUP = 64 * TPT;
  constexpr int TGT_PER_BLOCK = GROUPS * TGT_PER_GROUP;
  constexpr int TILE = 1024;  // sources per LDS tile
  constexpr int UNR = 8;
  constexpr int BLK = 256;    // inner block of the two-level summation (SGPR flavour)

  const int lane = threadIdx.x & 63;
  const int wave = wave_id_uniform();
  const int group = wave / WSPLIT;
  const int ws = wave % WSPLIT;
  const int t0 = blockIdx.x * TGT_PER_BLOCK + group * TGT_PER_GROUP;

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

  // source range of this (grid split, wave split)
  const int n_split = (int)gridDim.y * WSPLIT;
  int chunk = (a.n_src + n_split - 1) / n_split;
  chunk = (chunk + 7) & ~7;
  const float clamp = a.clamp;

  if constexpr (!USE_LDS) {
    const int split = (int)blockIdx.y * WSPLIT + ws;
    long j0l = (long)split * chunk;
    int j0 = j0l < a.n_src ? (int)j0l : a.n_src;
    int j1 = (j0 + chunk < a.n_src) ? j0 + chunk : a.n_src;
    const float2* __restrict__ ps = a.pos_all;
    const float* __restrict__ ms = a.mass_all;
    int j = j0;
    while (j < j1) {
      // two-level summation: a block of BLK sources is summed on its own, then added to the running total
      const int jb = (j + BLK < j1) ? j + BLK : j1;
      float bx[TPT], by[TPT];
#pragma unroll
      for (int k = 0; k < TPT; ++k) bx[k] = by[k] = 0.f;
      for (; j + UNR <= jb; j += UNR) {
        float2 p[UNR];
        float m[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          p[u] = ps[j + u];
          m[u] = UNIFORM ? 1.0f : ms[j + u];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int k = 0; k < TPT; ++k) fast_pair<UNIFORM>(xi[k], yi[k], p[u].x, p[u].y, m[u], clamp, bx[k], by[k]);
      }
      for (; j < jb; ++j) {
        float2 p = ps[j];
        float m = UNIFORM ? 1.0f : ms[j];
#pragma unroll
        for (int k = 0; k < TPT; ++k) fast_pair<UNIFORM>(xi[k], yi[k], p.x, p.y, m, clamp, bx[k], by[k]);
      }
#pragma unroll
      for (int k = 0; k < TPT; ++k) {
        ax[k] += bx[k];
        ay[k] += by[k];
      }
    }
  } else {
    __shared__ __attribute__((aligned(16))) float2 tile_pos[TILE];
    __shared__ __attribute__((aligned(16))) float tile_mass[UNIFORM ? 4 : TILE];
    // sources of this grid split: [g0, g1); all waves of the block stage a tile together, wave ws reads
    // its WSPLIT-th share of it
    const int gchunk = chunk * WSPLIT;
    long g0l = (long)blockIdx.y * gchunk;
    const int g0 = g0l < a.n_src ? (int)g0l : a.n_src;
    const int g1 = (g0 + gchunk < a.n_src) ? g0 + gchunk : a.n_src;
    for (int base = g0; base < g1; base += TILE) {
      const int cnt = (g1 - base < TILE) ? g1 - base : TILE;
#pragma unroll
      for (int r = 0; r < TILE / 256; ++r) {
        int s = r * 256 + (int)threadIdx.x;
        // padding: a source at the origin with zero mass; UNIFORM pads with a far-away point instead and
        // the tail below never reads it
        float2 p = make_float2(0.f, 0.f);
        float m = 0.f;
        if (s < cnt) {
          p = a.pos_all[base + s];
          if (!UNIFORM) m = a.mass_all[base + s];
        }
        tile_pos[s] = p;
        if (!UNIFORM) tile_mass[s] = m;
      }
      __syncthreads();
      constexpr int SHARE = TILE / WSPLIT;
      const int lo = ws * SHARE;
      int hi = lo + SHARE;
      if (UNIFORM && hi > cnt) hi = cnt > lo ? cnt : lo;   // no zero-mass trick without masses: stop at cnt
      int u = lo;
      // two-level summation: this wave's share of the tile is summed on its own, then added to the total
      float bx[TPT], by[TPT];
#pragma unroll
      for (int k = 0; k < TPT; ++k) bx[k] = by[k] = 0.f;
      for (; u + UNR <= hi; u += UNR) {
        float4 pp[UNR / 2];
        float2 mm[UNR / 2];
#pragma unroll
        for (int h = 0; h < UNR / 2; ++h) {
          pp[h] = *reinterpret_cast<const float4*>(&tile_pos[u + 2 * h]);
          if (!UNIFORM) mm[h] = *reinterpret_cast<const float2*>(&tile_mass[u + 2 * h]);
          else mm[h] = make_float2(1.f, 1.f);
        }
        if constexpr (USE_ASM) {
#pragma unroll
          for (int k = 0; k < TPT; ++k) {
            float2 q0 = make_float2(pp[0].x, pp[0].y), q1 = make_float2(pp[0].z, pp[0].w);
            float2 q2 = make_float2(pp[1].x, pp[1].y), q3 = make_float2(pp[1].z, pp[1].w);
            float2 q4 = make_float2(pp[2].x, pp[2].y), q5 = make_float2(pp[2].z, pp[2].w);
            float2 q6 = make_float2(pp[3].x, pp[3].y), q7 = make_float2(pp[3].z, pp[3].w);
            float2 acc2 = make_float2(bx[k], by[k]);
            fast_block8<UNIFORM>(make_float2(xi[k], yi[k]), clamp, q0, q1, q2, q3, q4, q5, q6, q7, mm[0].x, mm[0].y,
                                 mm[1].x, mm[1].y, mm[2].x, mm[2].y, mm[3].x, mm[3].y, acc2);
            bx[k] = acc2.x;
            by[k] = acc2.y;
          }
        } else {
#pragma unroll
          for (int h = 0; h < UNR / 2; ++h)
#pragma unroll
            for (int k = 0; k < TPT; ++k) {
              fast_pair<UNIFORM>(xi[k], yi[k], pp[h].x, pp[h].y, mm[h].x, clamp, bx[k], by[k]);
              fast_pair<UNIFORM>(xi[k], yi[k], pp[h].z, pp[h].w, mm[h].y, clamp, bx[k], by[k]);
            }
        }
      }
      for (; u < hi; ++u) {
        float2 p = tile_pos[u];
        float m = UNIFORM ? 1.0f : tile_mass[u];
#pragma unroll
        for (int k = 0; k < TPT; ++k) fast_pair<UNIFORM>(xi[k], yi[k], p.x, p.y, m, clamp, bx[k], by[k]);
      }
#pragma unroll
      for (int k = 0; k < TPT; ++k) {
        ax[k] += bx[k];
        ay[k] += by[k];
      }
      __syncthreads();
    }
  }

  if constexpr (UNIFORM) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      ax[k] *= a.uniform_mass;
      ay[k] *= a.uniform_mass;
    }
  }

  // ---- reduce the WSPLIT partial sums of a group in fixed order (deterministic)
  if constexpr (WSPLIT > 1) {
    __shared__ float2 red[GROUPS][WSPLIT - 1][TGT_PER_GROUP];
    if (ws > 0) {
#pragma unroll
      for (int k = 0; k < TPT; ++k) red[group][ws - 1][k * 64 + lane] = make_float2(ax[k], ay[k]);
    }
    __syncthreads();
    if (ws > 0) return;
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
#pragma unroll
      for (int w = 0; w < WSPLIT - 1; ++w) {
        float2 r = red[group][w][k * 64 + lane];
        ax[k] += r.x;
        ay[k] += r.y;
      }
    }
  }

#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    int t = t0 + k * 64 + lane;
    if (t >= a.n_tgt) continue;
    if (gridDim.y > 1) {
      a.partial[(size_t)blockIdx.y * a.n_tgt + t] = make_float2(ax[k], ay[k]);
    } else {
      integrate_store(a, t, ax[k], ay[k]);
    }
  }
}

// Sums the grid-split partials in ascending split order, then integrates.
__global__ __launch_bounds__(256) void direct_finish(const DirectArgs a, int n_gsplit) {
  if (a.gate && ((*a.gate != 0) != (a.run_if != 0))) return;
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.n_tgt) return;
  float ax = 0.f, ay = 0.f;
  for (int g = 0; g < n_gsplit; ++g) {
    float2 r = a.partial[(size_t)g * a.n_tgt + t];
    ax += r.x;
    ay += r.y;
  }
  integrate_store(a, t, ax, ay);
}

// ------------------------------------------------------------------------------------------------ EXACT
__global__ __launch_bounds__(256) void direct_exact(const DirectArgs a) {
  if (a.gate && ((*a.gate != 0) != (a.run_if != 0))) return;
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
// flag |= 1 when any coordinate is outside FAST's domain.
__global__ __launch_bounds__(256) void direct_hazard_scan(const float* __restrict__ xy, long n_floats, int* flag) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long stride = (long)gridDim.x * 256;
  int bad = 0;
  for (; i < n_floats; i += stride) {
    float v = __builtin_fabsf(xy[i]);
    bad |= !(v < kBig) || (v != 0.f && v < kTiny);
  }
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void weights_to_mass(const uint32_t* __restrict__ w, float* __restrict__ m, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) m[i] = (float)w[i];  // `weight as f32`, main.rs:360 (round to nearest even)
}

// ------------------------------------------------------------------------------------------------ host launchers
template <int TPT, int WSPLIT, bool USE_LDS>
static hipError_t launch_fast_t(hipStream_t s, const DirectArgs& a, int n_gsplit, bool use_asm) {
  constexpr int TGT_PER_BLOCK = (4 / WSPLIT) * 64 * TPT;
  dim3 grid((unsigned)((a.n_tgt + TGT_PER_BLOCK - 1) / TGT_PER_BLOCK), (unsigned)n_gsplit);
  constexpr bool CAN_ASM = USE_LDS && TPT == 1;   // the hand-ordered block exists for the LDS flavour, 1 target/thread
  if (a.uniform_mass > 0.f) {
    if (CAN_ASM && use_asm) hipLaunchKernelGGL((direct_fast<TPT, WSPLIT, USE_LDS, true, CAN_ASM>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((direct_fast<TPT, WSPLIT, USE_LDS, true, false>), grid, dim3(256), 0, s, a);
  } else {
    if (CAN_ASM && use_asm) hipLaunchKernelGGL((direct_fast<TPT, WSPLIT, USE_LDS, false, CAN_ASM>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((direct_fast<TPT, WSPLIT, USE_LDS, false, false>), grid, dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

hipError_t launch_direct_fast(hipStream_t s, const DirectArgs& a, const DirectConfig& c) {
  if (a.n_tgt <= 0) return hipSuccess;
  hipError_t e = hipErrorInvalidValue;
#define NB_CASE(T, W, L) \
  if (c.tpt == T && c.wsplit == W && c.use_lds == L) e = launch_fast_t<T, W, L>(s, a, c.gsplit, c.use_asm);
  NB_CASE(1, 1, false) NB_CASE(2, 1, false) NB_CASE(4, 1, false)
  NB_CASE(1, 4, false) NB_CASE(2, 4, false) NB_CASE(4, 4, false)
  NB_CASE(1, 1, true) NB_CASE(2, 1, true) NB_CASE(4, 1, true)
  NB_CASE(1, 4, true) NB_CASE(2, 4, true) NB_CASE(4, 4, true)
#undef NB_CASE
  if (e != hipSuccess) return e;
  if (c.gsplit > 1) {
    hipLaunchKernelGGL(direct_finish, dim3((unsigned)((a.n_tgt + 255) / 256)), dim3(256), 0, s, a, c.gsplit);
    e = hipGetLastError();
  }
  return e;
}

hipError_t launch_direct_exact(hipStream_t s, const DirectArgs& a) {
  if (a.n_tgt <= 0) return hipSuccess;
  hipLaunchKernelGGL(direct_exact, dim3((unsigned)((a.n_tgt + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_hazard_scan(hipStream_t s, const float* xy, long n_floats, int* flag) {
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
  if (e != hipSuccess || n_floats <= 0) return e;
  long blocks = (n_floats + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(direct_hazard_scan, dim3((unsigned)blocks), dim3(256), 0, s, xy, n_floats, flag);
  return hipGetLastError();
}

hipError_t launch_weights_to_mass(hipStream_t s, const uint32_t* w, float* m, long n) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(weights_to_mass, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, m, n);
  return hipGetLastError();
}

}  // namespace nbody

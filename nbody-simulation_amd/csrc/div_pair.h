// (nx / den, ny / den) in f32, each quotient exactly what the compiler's expansion of `/` gives — the IEEE-correctly rounded
// quotient, every special value included — for the two divisions by one denominator of main.rs:252
// (`*accel += (diff * force) / (sum * distance)`: a Vector2F divided by an f32 is two f32 divisions).
//
// The expansion of one `/` is v_div_scale (denominator), v_div_scale (numerator), v_rcp, six dependent multiply-adds,
// v_div_fmas, v_div_fixup.  Here the same instructions run for both quotients, lane by lane the same operations on the same
// values — so the same bits — but the six multiply-adds of the x and the y quotient are issued as PACKED ops: on gfx950 a
// VALU instruction of a wave that shares its SIMD costs a 4-cycle issue slot whatever it is, and a packed op does two lanes'
// worth in one (measured cost model): 19 slots for the pair instead of 25.
// f32 denormals are on in every kernel of this library (the default), as the compiler's own expansion assumes.
#pragma once
#include <hip/hip_runtime.h>

namespace nbody {

typedef float div_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 div_pair(const float nx, const float ny, const float den) {
  bool fx, fy, unused;
  const div_v2f ds = {__builtin_amdgcn_div_scalef(nx, den, false, &unused), __builtin_amdgcn_div_scalef(ny, den, false, &unused)};
  const div_v2f ns = {__builtin_amdgcn_div_scalef(nx, den, true, &fx), __builtin_amdgcn_div_scalef(ny, den, true, &fy)};
  div_v2f r = {__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
  const div_v2f one = {1.0f, 1.0f};
  div_v2f e = __builtin_elementwise_fma(-ds, r, one);
  r = __builtin_elementwise_fma(e, r, r);
  div_v2f q = ns * r;
  e = __builtin_elementwise_fma(-ds, q, ns);
  q = __builtin_elementwise_fma(e, r, q);
  e = __builtin_elementwise_fma(-ds, q, ns);
  return make_float2(__builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e.x, r.x, q.x, fx), den, nx),
                     __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e.y, r.y, q.y, fy), den, ny));
}

}  // namespace nbody

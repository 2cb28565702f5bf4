// The reference's draw() (/root/reference src/main.rs:41-72) on the device, bit for bit.
//
// draw() walks `particles` in row order and, per particle inside [0, HEIGHT)^2 (within_bounds, :224-226), touches the
// pixel (y as u32 / cell) * RENDER_HEIGHT + (x as u32 / cell), cell = HEIGHT / RENDER_HEIGHT (:51-54):
//   weight > 10           -> the pixel becomes (0, 255, 0, 255)                                        (:55-59)
//   else, alpha != 255    -> R = 255, G = B = 255 - v with v = 0x10 + min(((|vx|+|vy|) * 10) as u8, 0xef),
//                            alpha += 10 while alpha <= 240                                             (:60-69)
// The loop is sequential but its result per pixel is a function of three order-free facts: does any heavy particle
// land there (then green wins, whatever came before or after), how many light ones do (alpha = min(10 * count, 250)),
// and which light one comes LAST in row order (its v colours the pixel).  So: one atomicOr/atomicAdd word and one
// atomicMax word (row << 8 | v) per pixel, then a resolve pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "render.h"

namespace nbody {

namespace {

template <class T> struct V2r;
template <> struct V2r<float> { using type = float2; };
template <> struct V2r<double> { using type = double2; };

constexpr uint32_t kHeavyBit = 0x80000000u;

template <class T>
__global__ __launch_bounds__(256) void render_splat(int64_t n, const void* pos_, const void* vel_, const uint32_t* __restrict__ weight,
                                                    T height, uint32_t cell, uint32_t render_px, uint32_t* __restrict__ count,
                                                    uint32_t* __restrict__ last) {
  using T2 = typename V2r<T>::type;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T2 p = reinterpret_cast<const T2*>(pos_)[i];
  if (!(p.y < height && p.x < height && p.y >= (T)0 && p.x >= (T)0)) return;  // within_bounds; NaN fails
  const uint32_t px = (uint32_t)p.x / cell, py = (uint32_t)p.y / cell;        // `as u32` truncates
  const uint32_t pix = py * render_px + px;
  if (weight[i] > 10u) {
    atomicOr(&count[pix], kHeavyBit);
    return;
  }
  const T2 v = reinterpret_cast<const T2*>(vel_)[i];
  const T t = ((v.x < 0 ? -v.x : v.x) + (v.y < 0 ? -v.y : v.y)) * (T)10.0;
  uint32_t b = t != t ? 0u : (t >= (T)255 ? 255u : (uint32_t)t);  // `as u8` saturates, NaN -> 0
  b = b < 0xefu ? b : 0xefu;
  atomicAdd(&count[pix], 1u);
  atomicMax(&last[pix], ((uint32_t)i << 8) | (0x10u + b));  // rows < 2^24
}

__global__ __launch_bounds__(256) void render_resolve(uint32_t npix, const uint32_t* __restrict__ count, const uint32_t* __restrict__ last,
                                                      uchar4* __restrict__ rgba) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const uint32_t c = count[i];
  uchar4 o = make_uchar4(0, 0, 0, 0);
  if (c & kHeavyBit) {
    o = make_uchar4(0x00, 0xff, 0x00, 0xff);
  } else if (c) {
    const uint32_t v = last[i] & 0xffu;
    const uint32_t a = c < 25u ? 10u * c : 250u;
    o = make_uchar4(0xff, (unsigned char)(0xffu - v), (unsigned char)(0xffu - v), (unsigned char)a);
  }
  rgba[i] = o;
}

}  // namespace

template <class T>
hipError_t launch_render(hipStream_t s, int64_t n, const void* pos, const void* vel, const uint32_t* weight, uint32_t height,
                         uint32_t render_px, uint32_t* work, uint8_t* rgba) {
  const uint32_t npix = render_px * render_px;
  hipError_t e = hipMemsetAsync(work, 0, sizeof(uint32_t) * 2 * (size_t)npix, s);
  if (e != hipSuccess) return e;
  if (n > 0)
    render_splat<T><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(n, pos, vel, weight, (T)height, height / render_px,
                                                                            render_px, work, work + npix);
  render_resolve<<<dim3((npix + 255) / 256), dim3(256), 0, s>>>(npix, work, work + npix, (uchar4*)rgba);
  return hipGetLastError();
}

template hipError_t launch_render<float>(hipStream_t, int64_t, const void*, const void*, const uint32_t*, uint32_t, uint32_t,
                                         uint32_t*, uint8_t*);
template hipError_t launch_render<double>(hipStream_t, int64_t, const void*, const void*, const uint32_t*, uint32_t, uint32_t,
                                          uint32_t*, uint8_t*);

}  // namespace nbody

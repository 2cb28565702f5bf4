// Frame raster of the reference's draw() (render.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nbody {

// work: 2 * render_px^2 u32 (zeroed by the call); rgba: render_px^2 * 4 bytes, device memory.
template <class T>
hipError_t launch_render(hipStream_t s, int64_t n, const void* pos, const void* vel, const uint32_t* weight, uint32_t height,
                         uint32_t render_px, uint32_t* work, uint8_t* rgba);

}  // namespace nbody

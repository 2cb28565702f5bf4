// How fast can ONE wave add N numbers in order (a dependent chain of f32 adds)?  Variants of feeding the addend.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int K> __device__ __forceinline__ void add_row_lane(float& s, float v) {
  asm volatile("v_add_f32_dpp %0, %1, %0 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(v), "n"(K));
}
__global__ void k_dpp(const float* x, int n, float* out, long long* cyc) {
  const int lane = threadIdx.x, sub = lane & 15;
  float s = 0.f;
  long long t0 = clock64();
  for (int p = 0; p < n; p += 16) {
    float v = x[p + sub];
    asm volatile("s_nop 1" ::: "memory");
#define A(K) add_row_lane<K>(s, v);
    A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) A(12) A(13) A(14) A(15)
#undef A
  }
  long long t1 = clock64();
  if (lane == 0) { out[0] = s; cyc[0] = t1 - t0; }
}
__global__ void k_readlane(const float* x, int n, float* out, long long* cyc) {
  const int lane = threadIdx.x;
  float s = 0.f;
  long long t0 = clock64();
  for (int p = 0; p < n; p += 64) {
    float v = x[p + lane];
#pragma unroll
    for (int k = 0; k < 64; ++k) s = s + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
  }
  long long t1 = clock64();
  if (lane == 0) { out[1] = s; cyc[1] = t1 - t0; }
}
__global__ void k_lds(const float* x, int n, float* out, long long* cyc) {
  __shared__ float buf[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < n && i < 8192; i += 64) buf[i] = x[i];
  __syncthreads();
  float s = 0.f;
  long long t0 = clock64();
  const int m = n < 8192 ? n : 8192;
#pragma unroll 16
  for (int k = 0; k < m; ++k) s = s + buf[k];
  long long t1 = clock64();
  if (lane == 0) { out[2] = s; cyc[2] = t1 - t0; }
}
__global__ void k_scalar(const float* x, int n, float* out, long long* cyc) {  // uniform global loads -> s_load + v_add with SGPR operand
  const int lane = threadIdx.x;
  float s = 0.f;
  long long t0 = clock64();
#pragma unroll 16
  for (int k = 0; k < n; ++k) s = s + x[k];
  long long t1 = clock64();
  if (lane == 0) { out[3] = s; cyc[3] = t1 - t0; }
}
__global__ void k_dpp2(const float* x, int n, float* out, long long* cyc) {  // two interleaved chains (x and y of a point)
  const int lane = threadIdx.x, sub = lane & 15;
  float s = 0.f, t = 0.f;
  long long t0 = clock64();
  for (int p = 0; p < n; p += 32) {
    float v = x[p + sub], w = x[p + 16 + sub];
    asm volatile("s_nop 1" ::: "memory");
#define A(K) add_row_lane<K>(s, v); add_row_lane<K>(t, w);
    A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) A(12) A(13) A(14) A(15)
#undef A
  }
  long long t1 = clock64();
  if (lane == 0) { out[4] = s + t; cyc[4] = t1 - t0; }
}
__global__ void k_readlane2(const float2* x, int n, float* out, long long* cyc) {
  const int lane = threadIdx.x;
  float s = 0.f, t = 0.f;
  long long t0 = clock64();
  for (int p = 0; p < n; p += 64) {
    float2 v = x[p + lane];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      s = s + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.x), k));
      t = t + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.y), k));
    }
  }
  long long t1 = clock64();
  if (lane == 0) { out[5] = s + t; cyc[5] = t1 - t0; }
}
__global__ void k_lds2(const float2* x, int n, float* out, long long* cyc) {
  __shared__ float2 buf[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < n && i < 4096; i += 64) buf[i] = x[i];
  __syncthreads();
  float s = 0.f, t = 0.f;
  long long t0 = clock64();
  const int m = n < 4096 ? n : 4096;
#pragma unroll 16
  for (int k = 0; k < m; ++k) { const float2 q = buf[k]; s = s + q.x; t = t + q.y; }
  long long t1 = clock64();
  if (lane == 0) { out[6] = s + t; cyc[6] = t1 - t0; }
}
int main() {
  const int n = 8192;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)(i % 97) * 1.5f + 3.f;
  float *x, *out; long long* cyc;
  hipMalloc(&x, n * 4); hipMalloc(&out, 64); hipMalloc(&cyc, 64);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    k_dpp<<<1, 64>>>(x, n, out, cyc); k_readlane<<<1, 64>>>(x, n, out, cyc); k_lds<<<1, 64>>>(x, n, out, cyc);
    k_scalar<<<1, 64>>>(x, n, out, cyc); k_dpp2<<<1, 64>>>(x, n, out, cyc);
    k_readlane2<<<1, 64>>>((const float2*)x, n / 2, out, cyc); k_lds2<<<1, 64>>>((const float2*)x, n / 2, out, cyc);
    hipDeviceSynchronize();
  }
  float o[7]; long long c[7];
  hipMemcpy(o, out, 28, hipMemcpyDeviceToHost); hipMemcpy(c, cyc, 56, hipMemcpyDeviceToHost);
  float ref = 0.f; for (int i = 0; i < n; ++i) ref = ref + h[i];
  const char* names[7] = {"dpp row_newbcast", "readlane + sgpr add", "lds broadcast", "uniform global (s_load)", "dpp, two chains",
                          "readlane, two chains", "lds, two chains"};
  for (int i = 0; i < 7; ++i) printf("%-26s sum %.1f (ref %.1f)  %lld clocks (100 MHz counter?) per add: %.2f\n", names[i], o[i], i >= 4 ? 0.f : ref, c[i], (double)c[i] / n);
  return 0;
}

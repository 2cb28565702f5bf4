"""Switching-penalty matrix: time A B A B ... for instruction kinds A, B (independent registers)."""
import subprocess
ops = {
 'mul':    lambda i: f"v_mul_f32 v{i}, v{i}, v{40+(i%4)}",
 'fmac':   lambda i: f"v_fmac_f32 v{i}, v{40+(i%4)}, v{40+(i%4)}",
 'add64':  lambda i: f"v_add_f32 v{i}, |v{i}|, |v{40+(i%4)}|",
 'max':    lambda i: f"v_max_f32 v{i}, v{i}, v{40+(i%4)}",
 'fmaak':  lambda i: f"v_fmaak_f32 v{i}, v{i}, v{40+(i%4)}, 0x12800000",
 'rcp':    lambda i: f"v_rcp_f32 v{i}, v{i}",
 'pk_add': lambda i: f"v_pk_add_f32 v[{2*(i%8)+16}:{2*(i%8)+17}], v[{2*(i%8)+16}:{2*(i%8)+17}], v[44:45] neg_lo:[0,1] neg_hi:[0,1]",
 'pk_fma': lambda i: f"v_pk_fma_f32 v[{2*(i%8)+16}:{2*(i%8)+17}], v[46:47], v[44:45], v[{2*(i%8)+16}:{2*(i%8)+17}] op_sel_hi:[1,0,1]",
}
names = list(ops)
kern = []
for a in names:
    for b in names:
        body = []
        for i in range(8):
            body.append(ops[a](i)); body.append(ops[b](8 + i if not b.startswith('pk') and not a.startswith('pk') else (i + 4)))
        kern.append((a, b, body))
src = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 3000
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v40","v41","v42","v43","v44","v45","v46","v47"
#define KERNEL(NAME, BODY) \
  __global__ __launch_bounds__(256) void NAME(float* out, float seed, float one, float small) { \
    float t = seed + threadIdx.x * 0.37f; \
    asm volatile("v_mov_b32 v0, %0\n v_add_f32 v1, 1.0, v0\n v_add_f32 v2, 2.0, v0\n v_add_f32 v3, 4.0, v0\n v_add_f32 v4, 0.5, v0\n v_add_f32 v5, 1.0, v1\n v_add_f32 v6, 1.0, v2\n v_add_f32 v7, 1.0, v3\n" \
                 "v_add_f32 v8, 1.0, v4\n v_add_f32 v9, 1.0, v5\n v_add_f32 v10, 1.0, v6\n v_add_f32 v11, 1.0, v7\n v_add_f32 v12, 1.0, v8\n v_add_f32 v13, 1.0, v9\n v_add_f32 v14, 1.0, v10\n v_add_f32 v15, 1.0, v11\n" \
                 "v_mov_b32 v16, v0\n v_mov_b32 v17, v1\n v_mov_b32 v18, v2\n v_mov_b32 v19, v3\n v_mov_b32 v20, v4\n v_mov_b32 v21, v5\n v_mov_b32 v22, v6\n v_mov_b32 v23, v7\n v_mov_b32 v24, v0\n v_mov_b32 v25, v1\n v_mov_b32 v26, v2\n v_mov_b32 v27, v3\n v_mov_b32 v28, v4\n v_mov_b32 v29, v5\n v_mov_b32 v30, v6\n v_mov_b32 v31, v7\n" \
                 "v_mov_b32 v40, %1\n v_mov_b32 v41, %1\n v_mov_b32 v42, %1\n v_mov_b32 v43, %1\n v_mov_b32 v44, %2\n v_mov_b32 v45, %2\n v_mov_b32 v46, %2\n v_mov_b32 v47, %2\n" \
                 :: "v"(t), "v"(one * 1.0001f), "v"(small) : CLOB); \
    for (int i = 0; i < ITERS; ++i) { asm volatile(BODY ::: CLOB); } \
    float s; asm volatile("v_add_f32 %0, v0, v1\n v_add_f32 %0, %0, v8\n v_add_f32 %0, %0, v9\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v24\n" : "=v"(s) :: CLOB); \
    if (s == 123.456f) out[0] = 1; \
  }
'''
for i,(a,b,body) in enumerate(kern):
    src += f'KERNEL(k{i}, "' + '\\n '.join(body) + '\\n")\n'
src += 'typedef void (*kfn)(float*, float, float, float);\nint main() {\n  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); const int cus = prop.multiProcessorCount; float* out; (void)hipMalloc(&out, 1024);\n'
src += '  kfn fns[] = {' + ', '.join(f'k{i}' for i in range(len(kern))) + '};\n'
src += f'  const char* names[] = {{{", ".join(chr(34)+n+chr(34) for n in names)}}};\n  const int K = {len(names)};\n'
src += r'''  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::vector<double> cyc(K * K);
  for (int i = 0; i < K * K; ++i) {
    int w = 8, blocks = cus * w;
    hipLaunchKernelGGL(fns[i], dim3(blocks), dim3(256), 0, 0, out, 1.5f, 1.0f, 1e-3f); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) { (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(fns[i], dim3(blocks), dim3(256), 0, 0, out, 1.5f, 1.0f, 1e-3f); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
    cyc[i] = best * 1e6 / ((double)ITERS * 8 * w) * 2.38;   // cycles per (A,B) pair of instructions
  }
  printf("cycles for one A followed by one B (at 2.38 GHz), A = row, B = column; diagonal = 2 x cost(A)\n%-8s", "");
  for (int j = 0; j < K; ++j) printf("%8s", names[j]);
  printf("\n");
  for (int i = 0; i < K; ++i) { printf("%-8s", names[i]); for (int j = 0; j < K; ++j) printf("%8.2f", cyc[i * K + j]); printf("\n"); }
  printf("\nswitching penalty = AB - (AA + BB)/2\n%-8s", "");
  for (int j = 0; j < K; ++j) printf("%8s", names[j]);
  printf("\n");
  for (int i = 0; i < K; ++i) { printf("%-8s", names[i]); for (int j = 0; j < K; ++j) printf("%8.2f", cyc[i * K + j] - 0.5 * (cyc[i * K + i] + cyc[j * K + j])); printf("\n"); }
  return 0;
}
'''
open('switch_bench.hip','w').write(src)
r = subprocess.run(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','switch_bench.hip','-o','switch_bench'], capture_output=True, text=True)
print(r.stderr[-3000:] if r.returncode else 'built')

"""Generates and builds a microbenchmark of whole inner-loop bodies (asm text) — wall clock, 8 waves/SIMD."""
import re, sys, subprocess
import os
S = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'compiler_body_v1.s')).read().split('\n')  # hipcc's inner loop of direct_fast (v1, clamped, equal masses)
start = next(i for i,l in enumerate(S) if '.LBB22_18:' in l)
end = next(i for i,l in enumerate(S) if i > start and 's_cbranch_scc0 .LBB22_18' in l)
raw = [l.strip() for l in S[start+1:end]]
body0 = [l for l in raw if l and not l.startswith(';') and not l.startswith('ds_read') and not l.startswith('s_') and not l.startswith('v_mov_b32_e32 v15, s10')]
def variants():
    V = {}
    V['B0 compiler body'] = body0
    # 2 / 4 accumulators
    def multi_acc(n):
        out=[]; k=0
        for l in body0:
            if l.startswith('v_pk_fma_f32 v[8:9]'):
                a = 8 + 2*(k % n) if n <= 2 else [8, 10, 50, 52][k % n]
                l = l.replace('v_pk_fma_f32 v[8:9]', f'v_pk_fma_f32 v[{a}:{a+1}]').replace(', v[8:9] op_sel_hi', f', v[{a}:{a+1}] op_sel_hi'); k+=1
            out.append(l)
        return out
    V['B1 two accumulators'] = multi_acc(2)
    V['B2 four accumulators'] = multi_acc(4)
    V['B3 rcp -> v_mov'] = [re.sub(r'^v_rcp_f32_e32', 'v_mov_b32_e32', l) for l in body0]
    V['B4 max -> v_mov (1 src)'] = [re.sub(r'^v_max_f32_e32 (v\d+), (v\d+), v14', r'v_mov_b32_e32 \1, \2', l) for l in body0]
    V['B5 no pk_fma'] = [l for l in body0 if not l.startswith('v_pk_fma')]
    V['B6 no pk_add'] = [l for l in body0 if not l.startswith('v_pk_add')]
    V['B7 scalar fma x2 instead of pk_fma'] = sum([[l] if not l.startswith('v_pk_fma') else
        [re.sub(r'v_pk_fma_f32 v\[8:9\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[8:9\].*', r'v_fmac_f32_e32 v8, v\1, v\3', l),
         re.sub(r'v_pk_fma_f32 v\[8:9\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[8:9\].*', r'v_fmac_f32_e32 v9, v\2, v\3', l)] for l in body0], [])
    # phase-ordered
    order = ['v_pk_add','v_mul','v_fmac','v_add','v_max','v_fmaak','v_rcp','v_pk_fma']
    V['B8 phase ordered'] = sum([[l for l in body0 if l.startswith(o)] for o in order], [])
    V['B9 scalar sub x2 instead of pk_add'] = sum([[l] if not l.startswith('v_pk_add') else
        [re.sub(r'v_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[6:7\].*', r'v_sub_f32_e32 v\1, v\3, v6', l),
         re.sub(r'v_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[6:7\].*', r'v_sub_f32_e32 v\2, v\4, v7', l)] for l in body0], [])
    return V

def hand(npairs=8, order="PCRF", nacc=1, max_interleave=True):
    # per pair k: d = v[18+2k:19+2k] (in place), t = v{34+k} (d2), u = v{42+k} (sum/den/inv)  -- for 8 pairs
    P=[];M=[];F2=[];A=[];X=[];K=[];R=[];F=[]
    for k in range(npairs):
        dx=18+2*k; dy=dx+1; t=34+k; u=42+k
        P.append(f"v_pk_add_f32 v[{dx}:{dy}], v[{dx}:{dy}], v[6:7] neg_lo:[0,1] neg_hi:[0,1]")
        M.append(f"v_mul_f32_e32 v{t}, v{dx}, v{dx}")
        F2.append(f"v_fmac_f32_e32 v{t}, v{dy}, v{dy}")
        A.append(f"v_add_f32_e64 v{u}, |v{dx}|, |v{dy}|")
        X.append(f"v_max_f32_e32 v{t}, v{t}, v14")
        K.append(f"v_fmaak_f32 v{u}, v{u}, v{t}, 0x12800000")
        R.append(f"v_rcp_f32_e32 v{u}, v{u}")
    for k in range(npairs):
        dx=18+2*k; u=42+k
        # s operand pair [u:u+1] (op_sel_hi broadcast low); u+1 is another pair's value, harmless
        acc = [8,10,50,52][k % nacc]
        F.append(f"v_pk_fma_f32 v[{acc}:{acc+1}], v[{dx}:{dx+1}], v[{u - (u%2)}:{u - (u%2) + 1}], v[{acc}:{acc+1}] op_sel_hi:[1,0,1]" if u % 2 == 0 else
                 f"v_pk_fma_f32 v[{acc}:{acc+1}], v[{dx}:{dx+1}], v[{u-1}:{u}], v[{acc}:{acc+1}] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    if max_interleave:
        C = M + F2
        AX = []
        for a, x in zip(A, X): AX += [a, x]
        C = C + AX + K
    else:
        C = M + F2 + A + X + K
    seq = {"P": P, "C": C, "R": R, "F": F}
    return sum([seq[c] for c in order], [])

V = variants()

H = hand()
V = {}
V['H0 full hand body'] = H
V['X3 only 32-bit phase (40)'] = [l for l in H if not l.startswith('v_pk') and not l.startswith('v_rcp')]
V['X4 only pk phases (16)'] = [l for l in H if l.startswith('v_pk')]
V['X5 only rcp (8)'] = [l for l in H if l.startswith('v_rcp')]
V['X6 pk + rcp (24)'] = [l for l in H if l.startswith('v_pk') or l.startswith('v_rcp')]
V['X7 32-bit + rcp (48)'] = [l for l in H if not l.startswith('v_pk')]
V['X8 32-bit + pk (56)'] = [l for l in H if not l.startswith('v_rcp')]
V['X9 C phase: mul only x8'] = [l for l in H if l.startswith('v_mul')]
V['X10 C phase: fmac only x8'] = [l for l in H if l.startswith('v_fmac')]
V['X11 C phase: add+max x16'] = [l for l in H if l.startswith('v_add') or l.startswith('v_max')]
V['X12 C phase: fmaak x8'] = [l for l in H if l.startswith('v_fmaak')]
V['X13 pk_add only x8'] = [l for l in H if l.startswith('v_pk_add')]
V['X14 pk_fma only x8'] = [l for l in H if l.startswith('v_pk_fma')]
V['X15 pk_fma 4 acc x8'] = [l for l in hand(nacc=4) if l.startswith('v_pk_fma')]

V['H0b'] = hand()
V['H1 hand PCRF, max grouped'] = hand(max_interleave=False)
V['H2 hand PCRF, 2 acc'] = hand(nacc=2)
V['H3 hand FPCR (fma first)'] = hand(order='FPCR')
V['H4 hand PCRF 4 acc'] = hand(nacc=4)

src = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 4000
#define CLOB "v6","v7","v8","v9","v10","v11","v14","v15","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53"
#define KERNEL(NAME, BODY) \
  __global__ __launch_bounds__(256) void NAME(float* out, float seed, float one, float small) { \
    float t = seed + threadIdx.x * 0.37f; \
    asm volatile("v_mov_b32 v6, %1\n v_mov_b32 v7, %1\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v14, %1\n v_mov_b32 v49, 0\n v_mov_b32 v35, 0\n v_mov_b32 v37, 0\n v_mov_b32 v39, 0\n v_mov_b32 v41, 0\n v_mov_b32 v43, 0\n v_mov_b32 v45, 0\n v_mov_b32 v47, 0\n" \
                 "v_mov_b32 v18, %0\n v_add_f32 v19, 1.0, v18\n v_add_f32 v20, 2.0, v18\n v_add_f32 v21, 4.0, v18\n v_add_f32 v22, 0.5, v18\n v_add_f32 v23, 1.0, v19\n v_add_f32 v24, 1.0, v20\n v_add_f32 v25, 1.0, v21\n" \
                 "v_add_f32 v26, 1.0, v22\n v_add_f32 v27, 1.0, v23\n v_add_f32 v28, 2.0, v24\n v_add_f32 v29, 4.0, v25\n v_add_f32 v30, 0.5, v26\n v_add_f32 v31, 1.0, v27\n v_add_f32 v32, 1.0, v28\n v_add_f32 v33, 1.0, v29\n" \
                 :: "v"(t), "v"(small) : CLOB); \
    for (int i = 0; i < ITERS; ++i) { asm volatile(BODY ::: CLOB); } \
    float s; asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11\n v_add_f32 %0, %0, v50\n v_add_f32 %0, %0, v52\n v_add_f32 %0, %0, v17\n v_add_f32 %0, %0, v34\n v_add_f32 %0, %0, v36\n v_add_f32 %0, %0, v48\n" : "=v"(s) :: CLOB); \
    if (s == 123.456f) out[0] = 1; \
  }
'''
names=[]
for i,(name,body) in enumerate(V.items()):
    b = '"' + '\\n '.join(body) + '\\n"'
    src += f'KERNEL(k{i}, {b})\n'
    names.append((f'k{i}', name, sum(1 for l in body if l.startswith('v_'))))
src += 'typedef void (*kfn)(float*, float, float, float);\nstruct Case { const char* name; kfn fn; int ninstr; };\nint main() {\n  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); const int cus = prop.multiProcessorCount; float* out; (void)hipMalloc(&out, 1024);\n  std::vector<Case> cases = {'
src += ', '.join(f'{{"{n}", {k}, {c}}}' for k,n,c in names) + '};\n'
src += r'''  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("%-40s %8s %10s %12s %12s\n", "body (8 pairs per iteration)", "instrs", "ns/pair", "cyc/pair@2.38", "Gpairs/s");
  for (int w : {8, 4}) for (auto& c : cases) {
    int blocks = cus * w;
    hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 1.0f, 1e-3f); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) { (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 1.0f, 1e-3f); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
    double ns = best * 1e6 / ((double)ITERS * 8 * w);
    printf("w=%d %-36s %8d %10.3f %12.2f %12.1f\n", w, c.name, c.ninstr, ns, ns * 2.38, 64.0 * 4 * cus / ns);
  }
  return 0;
}
'''
open('body_bench_x.hip','w').write(src)
r = subprocess.run(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','body_bench_x.hip','-o','body_bench_x'], capture_output=True, text=True)
print(r.stderr[-3000:] if r.returncode else 'built', len(V), 'variants')

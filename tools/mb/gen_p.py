"""Round 3: does the unclamped pair body get cheaper when EVERY multiply-add is a packed op over TWO pairs?
(profiles/r01_valu_microbench_wallclock.txt: a wave issues one VALU per 4 cycles; plain f32 ops reach 2 cycles only when two
waves' plain ops share a slot, packed / max / rcp ops take the slot alone — a body that mixes the classes loses the sharing.)
Generates and builds body_bench_p (wall clock, w waves per SIMD)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def current(npairs=8):
    """direct_fast<1,true,true,true>'s block: 7 instructions per pair."""
    P, M, F2, A, K, R, F = [], [], [], [], [], [], []
    for k in range(npairs):
        dx = 18 + 2 * k
        dy = dx + 1
        u = 40 + 2 * k
        t = u + 1
        P.append(f"v_pk_add_f32 v[{dx}:{dy}], v[{dx}:{dy}], v[6:7] neg_lo:[0,1] neg_hi:[0,1]")
        M.append(f"v_mul_f32_e32 v{t}, v{dx}, v{dx}")
        F2.append(f"v_fmac_f32_e32 v{t}, v{dy}, v{dy}")
        A.append(f"v_add_f32_e64 v{u}, |v{dx}|, |v{dy}|")
        K.append(f"v_fmaak_f32 v{u}, v{u}, v{t}, 0x12800000")
        R.append(f"v_rcp_f32_e32 v{u}, v{u}")
        F.append(f"v_pk_fma_f32 v[8:9], v[{dx}:{dy}], v[{u}:{t}], v[8:9] op_sel_hi:[1,0,1]")
    return P + M + F2 + A + K + R + F


def packed(npairs=8, order="phase", bias="v", adds="add", rcp="rcp", prio=None, minv=False):
    """Two pairs per packed op.  Couple c: SX = (xA, xB) = v[18+4c:19+4c], SY = (yA, yB) = v[20+4c:21+4c] (differences in
    place), Q = v[34+2c:35+2c], S = v[42+2c:43+2c]; accumulators v[8:9] (x of even | odd sources), v[10:11] (y)."""
    PX, PY, M, F2, A, K, R, FX, FY = [], [], [], [], [], [], [], [], []
    R1, R2, R3 = [], [], []
    b = "v[14:15]" if bias == "v" else "s[20:21]"
    for c in range(npairs // 2):
        sx = 18 + 4 * c
        sy = sx + 2
        q = 34 + 2 * c
        s = 42 + 2 * c
        PX.append(f"v_pk_add_f32 v[{sx}:{sx+1}], v[{sx}:{sx+1}], v[6:7] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
        PY.append(f"v_pk_add_f32 v[{sy}:{sy+1}], v[{sy}:{sy+1}], v[6:7] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]")
        M.append(f"v_pk_mul_f32 v[{q}:{q+1}], v[{sx}:{sx+1}], v[{sx}:{sx+1}]")
        F2.append(f"v_pk_fma_f32 v[{q}:{q+1}], v[{sy}:{sy+1}], v[{sy}:{sy+1}], v[{q}:{q+1}]")
        if adds == "add":
            A.append(f"v_add_f32_e64 v{s}, |v{sx}|, |v{sy}|")
            A.append(f"v_add_f32_e64 v{s+1}, |v{sx+1}|, |v{sy+1}|")
        elif adds == "max":  # (not the arithmetic: a slot-alone op in the adds' place)
            A.append(f"v_max_f32_e64 v{s}, |v{sx}|, |v{sy}|")
            A.append(f"v_max_f32_e64 v{s+1}, |v{sx+1}|, |v{sy+1}|")
        elif adds == "pk":   # (not the arithmetic: no abs)
            A.append(f"v_pk_add_f32 v[{s}:{s+1}], v[{sx}:{sx+1}], v[{sy}:{sy+1}]")
        K.append(f"v_pk_fma_f32 v[{s}:{s+1}], v[{s}:{s+1}], v[{q}:{q+1}], {b}")
        if minv:  # round 4, direct_stream_m: the denominators scaled by the couple's inverse masses (an SGPR pair) before the reciprocals
            R.append(f"v_pk_mul_f32 v[{s}:{s+1}], v[{s}:{s+1}], s[22:23]")
        if rcp == "rcp":
            R.append(f"v_rcp_f32_e32 v{s}, v{s}")
            R.append(f"v_rcp_f32_e32 v{s+1}, v{s+1}")
        elif rcp in ("one", "one_lowprio"):
            # round 4: ONE reciprocal per couple — P = denA * denB, R = 1 / P, (rA, rB) = (denB, denA) * R.
            # P lives in the couple's Q pair (the squares are dead once the denominators exist).
            R1.append(f"v_mul_f32_e32 v{q}, v{s}, v{s+1}")
            R2.append(f"v_rcp_f32_e32 v{q}, v{q}")
            R3.append(f"v_pk_mul_f32 v[{s}:{s+1}], v[{s}:{s+1}], v[{q}:{q+1}] op_sel:[1,0] op_sel_hi:[0,0]")
        FX.append(f"v_pk_fma_f32 v[8:9], v[{sx}:{sx+1}], v[{s}:{s+1}], v[8:9]")
        FY.append(f"v_pk_fma_f32 v[10:11], v[{sy}:{sy+1}], v[{s}:{s+1}], v[10:11]")
    if rcp == "one":
        R = R + R1 + R2 + R3
    elif rcp == "one_lowprio" and prio is not None:  # the plain products pair up like the adds do
        R = R + [f"s_setprio {prio[0]}"] + R1 + [f"s_setprio {prio[1]}"] + R2 + R3
    if order == "phase" and prio is not None:  # the adds at another priority than the rest
        return PX + PY + M + F2 + [f"s_setprio {prio[0]}"] + A + [f"s_setprio {prio[1]}"] + K + R + FX + FY
    if order == "adds_first" and prio is not None:  # the NEXT block's adds cannot move up; the adds open the block
        return [f"s_setprio {prio[0]}"] + A + [f"s_setprio {prio[1]}"] + PX + PY + M + F2 + K + R + FX + FY
    if order == "phase":
        return PX + PY + M + F2 + A + K + R + FX + FY
    if order == "couple":  # couple by couple
        out = []
        n = npairs // 2
        for c in range(n):
            out += [PX[c], PY[c], M[c], F2[c]] + A[len(A) // n * c: len(A) // n * (c + 1)] + [K[c]] + R[2 * c: 2 * c + 2] + [FX[c], FY[c]]
        return out
    if order == "rcp_spread":  # one reciprocal between packed ops (the trans pipe next to the packed one)
        rest = PX + PY + M + F2 + A + K
        tail = FX + FY
        out = list(rest)
        for i, r in enumerate(R):
            out.append(r)
            if i < len(tail):
                out.append(tail[i])
        return out
    raise ValueError(order)


V = {}
V["N0 current block (7/pair)"] = current()
V["P1 packed, phases, bias vgpr"] = packed()
V["P1s packed, phases, bias sgpr"] = packed(bias="s")
V["P1c packed, couple by couple"] = packed(order="couple")
V["P1r packed, rcp spread over the fmas"] = packed(order="rcp_spread")
V["P2 packed, adds -> v_max (model probe)"] = packed(adds="max")
V["P3 packed, adds -> one pk_add (model probe)"] = packed(adds="pk")
V["P4 packed, no rcp (model probe)"] = packed(rcp="none")
V["Q1 packed, adds at LOW priority (1 elsewhere)"] = packed(bias="s", prio=(0, 1))
V["Q2 packed, adds at HIGH priority (0 elsewhere)"] = packed(bias="s", prio=(1, 0))
V["Q3 packed, adds at prio 0, rest at 3"] = packed(bias="s", prio=(0, 3))
# round 4
V["M1 = Q1 + inverse masses (direct_stream_m)"] = packed(bias="s", prio=(0, 1), minv=True)
V["R1 = Q1, ONE rcp per couple"] = packed(bias="s", prio=(0, 1), rcp="one")
V["R2 = R1, the products at LOW priority too"] = packed(bias="s", prio=(0, 1), rcp="one_lowprio")
V["R0 = P1s, ONE rcp per couple (no priorities)"] = packed(bias="s", rcp="one")

regs = list(range(6, 12)) + [14, 15] + list(range(18, 58))
CLOB = ",".join(f'"v{r}"' for r in regs) + ',"s20","s21","s22","s23"'
init = ["v_mov_b32 v6, %1", "v_mov_b32 v7, %1", "v_mov_b32 v8, 0", "v_mov_b32 v9, 0", "v_mov_b32 v10, 0", "v_mov_b32 v11, 0",
        "v_mov_b32 v14, 0x12800000", "v_mov_b32 v15, 0x12800000", "s_mov_b32 s20, 0x12800000", "s_mov_b32 s21, 0x12800000", "s_mov_b32 s22, 1.0", "s_mov_b32 s23, 1.0", "v_mov_b32 v18, %0"]
for r in range(19, 34):
    init.append(f"v_add_f32 v{r}, {['1.0', '2.0', '4.0', '0.5'][r % 4]}, v{r-1}")
for r in range(34, 58):
    init.append(f"v_mov_b32 v{r}, 1.0")
src = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 4000
#define CLOB ''' + CLOB + r'''
#define KERNEL(NAME, BODY) \
  __global__ __launch_bounds__(256) void NAME(float* out, float seed, float one, float small) { \
    float t = seed + threadIdx.x * 0.37f; \
    asm volatile("''' + r"\n ".join(init) + r'''\n" :: "v"(t), "v"(small) : CLOB); \
    for (int i = 0; i < ITERS; ++i) { asm volatile(BODY ::: CLOB); } \
    float s; asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11\n v_add_f32 %0, %0, v40\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v34\n" : "=v"(s) :: CLOB); \
    if (s == 123.456f) out[0] = 1; \
  }
'''
names = []
for i, (name, body) in enumerate(V.items()):
    b = '"' + '\\n '.join(body) + '\\n"'
    src += f'KERNEL(k{i}, {b})\n'
    names.append((f'k{i}', name, sum(1 for l in body if l.startswith('v_'))))
src += 'typedef void (*kfn)(float*, float, float, float);\nstruct Case { const char* name; kfn fn; int ninstr; };\nint main() {\n  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); const int cus = prop.multiProcessorCount; float* out; (void)hipMalloc(&out, 1024);\n  std::vector<Case> cases = {'
src += ', '.join(f'{{"{n}", {k}, {c}}}' for k, n, c in names) + '};\n'
src += r'''  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("%-48s %8s %10s %14s %12s\n", "body (8 pairs per iteration)", "instrs", "ns/pair", "cyc/pair@2.38", "frac@14flop");
  for (int w : {8, 4, 2}) for (auto& c : cases) {
    int blocks = cus * w;
    hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 1.0f, 1e-3f); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) { (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f, 1.0f, 1e-3f); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
    double ns = best * 1e6 / ((double)ITERS * 8 * w);
    printf("w=%d %-44s %8d %10.3f %14.2f %12.3f\n", w, c.name, c.ninstr, ns, ns * 2.38, 14.0 * 64 * 4 * cus / ns / 157.3e3);
  }
  return 0;
}
'''
open(os.path.join(HERE, 'body_bench_p.hip'), 'w').write(src)
r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', os.path.join(HERE, 'body_bench_p.hip'), '-o', os.path.join(HERE, 'body_bench_p')],
                   capture_output=True, text=True)
print(r.stderr[-3000:] if r.returncode else 'built', len(V), 'variants')

#!/usr/bin/env python3
"""Headline benchmark: pair-interactions/s and ms/step of the direct O(N^2) force + integration step at
N = 1 048 576 bodies (BASELINE.json configs[2]) on 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 5 --warmup 1

One process per GPU.  The N bodies are fixed (strong scaling): rank r owns targets [r*N/G, (r+1)*N/G), keeps a
replicated copy of all positions, runs nbody_direct_step_dev on its shard and all-gathers the new positions
(RCCL over xGMI) once per step.  A step = force on every body from every body + semi-implicit Euler
(reference: World::update, src/main.rs:388-425, with the direct sum of SURVEY a9 as the force phase).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_BODIES = 1 << 20          # BASELINE.json configs[2]
FLOPS_PER_PAIR = 14         # SURVEY §8d / DESIGN.md: algorithmic flops of one force-law evaluation
PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector (= FP32 matrix) rate
DT = 0.1                    # STEP_SIZE, main.rs:34
CLAMP = 0.001               # main.rs:247-248
SEED = 0x5EED0003


def cpu_baseline(pos, w, n_sample_targets):
    """The oracle (CPU restatement, -march=native build made on this box) timed on a bounded sample of the same
    workload: n_sample_targets targets x all N sources, all host cores.  Reported, never the target."""
    from oracle import oracle as orc
    try:
        orc.build(native=True)
        native = True
    except Exception:
        native = False
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # a container's CPU share (cgroup v2 quota), when there is one
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    n = pos.shape[0]
    tg = np.arange(0, n, n // n_sample_targets)[:n_sample_targets]
    orc.direct_accel(pos[:4096], w[:4096], targets=np.arange(64), nthreads=cores, native_lib=native)  # warm
    t0 = time.perf_counter()
    orc.direct_accel(pos, w, targets=tg, nthreads=cores, native_lib=native)
    dt = time.perf_counter() - t0
    pairs = float(len(tg)) * n
    return {"value": pairs / dt, "unit": "pair-interactions/s", "cores": cores, "kind": "port",
            "sample": f"{len(tg)} targets x {n} sources = {pairs:.3g} pairs in {dt:.1f} s "
                      f"(oracle/nbody_oracle.cpp, g++ -O3 {'-march=native' if native else '-march=x86-64-v3'} "
                      f"-ffp-contract=off, f32 sequential sum as main.rs:234-253 writes it)",
            "ms_per_step_extrapolated": 1e3 * float(n) * n / (pairs / dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bodies", dest="n", type=int, default=N_BODIES, help="total bodies (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                      "several ranks on one GPU)")
    ap.add_argument("--cpu-sample-targets", type=int, default=131072)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import nbody_simulation_amd as nb
    from nbody_simulation_amd.sharding import ShardedDirectStepper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    dev_index = local_rank % torch.cuda.device_count()   # one rank per GPU; the modulo only matters in a rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    n = args.n
    pos, vel, w = nb.scenes.plummer(n, seed=SEED)          # every rank generates the same bodies
    timer = nb._capi.Timer()
    stepper = ShardedDirectStepper(pos, vel, w, rank=rank, world=world, device=dev, clamp=CLAMP,
                                   arith=nb._capi.ARITH_AUTO, timer=timer,
                                   group=dist.group.WORLD if world > 1 else None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        stepper.step(DT)
    barrier()
    timer.read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stepper.step(DT)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_launches = timer.read(reset=True)

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        pairs_per_step = float(n) * float(n)
        value = pairs_per_step * args.steps / elapsed
        n_tgt = stepper.n_local
        flops_per_launch = FLOPS_PER_PAIR * float(n) * float(n_tgt)
        achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "r01_direct_pmc.json")
        if os.path.exists(prof) and world == 1 and n == N_BODIES:
            try:
                with open(prof) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "pair-interactions/sec + ms/step at N=1M direct O(N^2) (force + integration step)",
            "value": value, "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"direct O(N^2) f32, N={n} bodies, seeded 2-D Plummer sphere, masses 1, "
                                   f"dt={DT}, clamp={CLAMP} (BASELINE.json configs[2])",
                       "n_bodies": n, "targets_per_gpu": n_tgt, "chunks_per_step": stepper.chunks,
                       "exchange": "none" if world == 1 else f"{args.backend} in-place all-gather of float2 positions per chunk"
                                   + (" (RCCL over xGMI)" if args.backend == "nccl" else " (rehearsal)"),
                       "arith": "AUTO (FAST kernel; EXACT on hazardous positions)"},
            "roofline": {"bound": "valu_f32", "bound_class": "compute", "kernel": "nbody::direct_fast", "achieved": achieved,
                         "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_TFLOPS,
                         "traffic": traffic, "flops_per_pair": FLOPS_PER_PAIR, "pairs_per_launch": float(n) * n_tgt,
                         "kernel_ms": kern_ms, "launches_timed": kern_launches,
                         "note": "no MFMA on this path (no dense contraction); peak = f32 vector peak"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pos, w, args.cpu_sample_targets)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

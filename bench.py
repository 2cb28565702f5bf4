#!/usr/bin/env python3
"""Headline benchmark: pair-interactions/s and ms/step of the direct O(N^2) force + integration step at
N = 1 048 576 bodies (BASELINE.json configs[2]) on 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 5 --warmup 1

One process per GPU.  The N bodies are fixed (strong scaling): rank r owns the target blocks {c*G + r} (sharding.py),
keeps a replicated copy of all positions, runs nbody_direct_prep_dev / nbody_direct_run_dev on its blocks and
all-gathers the new positions in place (RCCL over xGMI) once per chunk.  A step = force on every body from every
body + semi-implicit Euler (reference: World::update, src/main.rs:388-425, with the direct sum of SURVEY a9 as the
force phase).

Output (rank 0, stdout): the HEADLINE is the LAST line, one compact JSON object (< 2 KB: metric, value, unit, n_gpus, steps,
warmup, ms_per_step, dtype, config, `roofline`, `cpu_baseline`, and a one-number-per-leg `legs` digest).  At N = 1 it is
preceded by one compact JSON line (< 1 KB) per leg — the other single-GPU configurations: the reference's live BVH step, the
same on the headline's bodies, config 2 (65 536 direct), mass classes, free per-body masses, the reference's own scene
(direct), config 4 (4 194 304 bodies, quad tree, theta 0.5, f64; the reference's arithmetic and the opt-in FAST one).  The
explanatory notes every record used to repeat live once in profiles/README.md ("Reading a bench line").  `--full-out PATH`
also writes the unabridged records (headline + legs) to PATH; `--leg NAME` runs one leg alone and prints its unabridged
record (what the rocprofv3 passes under profiles/ are taken on).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one rank per GPU shares device buffers between processes (RCCL): the host driver of this pool supports dmabuf IPC only.  Set before
# anything initialises HIP; a value the launcher exported wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

N_BODIES = 1 << 20          # BASELINE.json configs[2]
FLOPS_PER_PAIR = 14         # SURVEY §8d / DESIGN.md: algorithmic flops of one force-law evaluation
# what the timed kernel (equal masses, far sources: direct_stream, packed couples) executes per pair: v_pk_add 2 x 1/2 x 2 = 2,
# v_pk_mul 1, v_pk_fma 2 (the squares), v_add 1, v_pk_fma 2 (the denominator; the 2^-90 bias rides in its addend), v_rcp 1, two
# v_pk_fma 4 = 13; against the 14 algorithmic ones the mass multiply is hoisted out of the sum and the clamp is dropped for far
# sources (-2), the bias add is extra (+1)
FLOPS_EXECUTED_PER_PAIR = 13


def _main_pass_kernel(uniform=True):
    """Name of the direct step's dominant kernel (direct_kernels.hip launch_direct_fast): the far sources of equal masses / mass
    classes stream through SGPRs (direct_stream); free per-body masses bring their inverse masses along (direct_stream_m)."""
    return "nbody::direct_stream" if uniform else "nbody::direct_stream_m"


PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector (= FP32 matrix) rate
PEAK_F64_TFLOPS = 78.6      # FP64 vector: half the FP32 vector rate (256 CU x 2.4 GHz x 128 flop/clk/CU; AMD's MI355X figure)
DT = 0.1                    # STEP_SIZE, main.rs:34
CLAMP = 0.001               # main.rs:247-248
SEED = 0x5EED0003
PAIR_FLOPS_AS_WRITTEN = 14  # main.rs:236-252 in GPU-minimal counting, a division as one flop
NODE_TEST_FLOPS = 12        # contains (4 compares) + dist2 (2 sub, 2 mul, 1 add) + d2*theta*theta (2 mul) + 1 compare


def _profile(name):
    """A committed rocprofv3 summary under profiles/ (HBM bytes per launch from PMC counters), or None."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return json.load(f), "profiles/" + name
    except Exception:
        return None, None


def _traffic(roof, profile_name, algorithmic_bytes):
    """roofline.traffic is not measurable from inside this process (PMC counters need rocprofv3): it is read from the
    committed summary of the same command (this round's, else the previous round's file of the same name) and says which.
    How the figure is corrected: profiles/README.md, "Reading a bench line"."""
    roof["algorithmic_bytes_per_launch"] = algorithmic_bytes
    roof["traffic"] = None
    roof["traffic_source"] = None
    for cand in (profile_name, profile_name.replace("r04_", "r03_", 1)):
        prof, src = _profile(cand)
        if prof and prof.get("hbm_bytes_per_launch") is not None:
            roof["traffic"] = prof["hbm_bytes_per_launch"]
            roof["traffic_source"] = src + " (rocprofv3 --pmc, separate passes; not measured in this run)"
            if algorithmic_bytes:
                roof["traffic_over_algorithmic"] = prof["hbm_bytes_per_launch"] / algorithmic_bytes
            break
    return roof


def cpu_baseline(pos, w, n_sample_targets):
    """The oracle (CPU restatement, -march=native build made on this box) timed on a bounded sample of the same
    workload: n_sample_targets targets x all N sources, all host cores.  Reported, never the target."""
    return _cpu_direct_sample(pos, w, n_sample_targets)


# ---------------------------------------------------------------------------------------------------- legs
def _host_cores():
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
    return cores


def _oracle_native():
    from oracle import oracle as orc
    try:
        orc.build(native=True)
        return orc, True
    except Exception:
        return orc, False


def _cpu_direct_sample(pos, w, n_targets):
    """cpu_baseline of a direct-sum leg: the oracle on `n_targets` evenly spaced targets x all sources."""
    orc, native = _oracle_native()
    cores = _host_cores()
    n = pos.shape[0]
    n_targets = min(n_targets, n)
    tg = np.arange(0, n, max(1, n // n_targets))[:n_targets]
    orc.direct_accel(pos[:4096], w[:4096], targets=np.arange(64), nthreads=cores, native_lib=native)  # warm
    t0 = time.perf_counter()
    orc.direct_accel(pos, w, targets=tg, nthreads=cores, native_lib=native)
    dt = time.perf_counter() - t0
    pairs = float(len(tg)) * n
    return {"value": pairs / dt, "unit": "pair-interactions/s", "cores": cores, "kind": "port",
            "sample": f"{len(tg)} targets x {n} sources = {pairs:.3g} pairs in {dt:.2f} s (oracle/nbody_oracle.cpp, g++ -O3 "
                      f"{'-march=native' if native else '-march=x86-64-v3'} -ffp-contract=off, the force law and its f32 sequential "
                      f"sum as main.rs:234-253 writes them)",
            "sample_short": f"{len(tg)} targets x {n} sources, {dt:.1f} s, oracle f32 as written",
            "ms_per_step_extrapolated": 1e3 * float(n) * n / (pairs / dt)}


def _cpu_tree_steps(pos, vel, w, kind_name, theta, order_mode, steps):
    """cpu_baseline of a Barnes-Hut leg: `steps` whole World::update steps of the oracle (build single-threaded as
    bvh_tree.rs:56-96, force map over all host cores as main.rs:406-416, integrate single-threaded as :419-423)."""
    orc, native = _oracle_native()
    cores = _host_cores()
    t0 = time.perf_counter()
    if kind_name == "bvh":
        _, _, _, _, c3 = orc.update_bvh(pos, vel, w, delta=DT, theta=theta, mode=order_mode, nsteps=steps, nthreads=cores,
                                        native_lib=native)
    else:
        _, _, c3 = orc.update_quad(pos, vel, w, delta=DT, theta=theta, nsteps=steps, nthreads=cores, native_lib=native)
    dt = time.perf_counter() - t0
    return {"value": 1e3 * dt / steps, "unit": "ms/step", "cores": cores, "kind": "port",
            "build_ms": 1e3 * c3[0] / steps, "sum_gravity_ms": 1e3 * c3[1] / steps, "post_calculations_ms": 1e3 * c3[2] / steps,
            "sample": f"{steps} whole step(s) of the oracle's World::update ({kind_name} tree, theta {theta}) on the same bodies in {dt:.2f} s "
                      f"(oracle/nbody_oracle.cpp, {'-march=native' if native else '-march=x86-64-v3'}; Counting split as main.rs:402/417/424)",
            "sample_short": f"{steps} whole oracle World::update step(s), {kind_name} theta {theta}, {dt:.2f} s"}


def _direct_leg(nb, name, workload, pos, vel, w, steps, profile_name, executed=None, cpu_targets=8192, kernel=None, short=None):
    """A direct-sum leg through the context path (nbody_update_direct_f32), timed like the headline."""
    C = nb._capi
    n = pos.shape[0]
    timer = C.Timer()
    with C.Context(0) as ctx:
        ctx.upload(pos, vel, w)
        ctx.update_direct(DT, 1)
        t0 = time.perf_counter()               # the leg's value: the call as a caller makes it (no timer attached: small problems
        ctx.update_direct(DT, steps)           # replay their step as a hipGraph)
        dt = time.perf_counter() - t0
        ctx.set_timer(timer)                   # the same steps again with HIP events around the main pass (eager launches)
        ctx.update_direct(DT, 1)
        timer.read(reset=True)
        ctx.update_direct(DT, steps)
        kms, kl = timer.read(reset=True)
        ctx.set_timer(None)
    pairs = float(n) * n
    kms_per_launch = kms                      # nbody_timer_read: average ms per timed region (one region = one step's main pass)
    ach = FLOPS_PER_PAIR * pairs / (kms_per_launch * 1e-3) / 1e12 if kms > 0 else 0.0
    roof = {"bound": "valu_f32", "bound_class": "compute", "kernel": kernel or _main_pass_kernel(), "achieved": ach, "peak": PEAK_F32_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / PEAK_F32_TFLOPS, "flops_per_pair": FLOPS_PER_PAIR, "pairs_per_launch": pairs,
            "kernel_ms": kms, "launches_timed": kl}
    if executed:
        roof["flops_executed_per_pair"] = executed
        roof["frac_executed"] = roof["frac"] * executed / FLOPS_PER_PAIR
    _traffic(roof, profile_name, 36 * n)
    out = {"leg": name, "workload": workload, "workload_short": short, "metric": "pair-interactions/s", "value": pairs * steps / dt,
           "ms_per_step": 1e3 * dt / steps, "steps": steps, "dtype": "f32", "roofline": roof}
    if cpu_targets:
        out["cpu_baseline"] = _cpu_direct_sample(pos, w, cpu_targets)
    return out


def _tree_leg(nb, name, workload, pos, vel, w, kind, theta, steps, profile_names, order=None, cpu_steps=1, default_arith="exact", short=None):
    """A Barnes-Hut leg: whole steps (build + walk + integrate, Counting split) with the reference's arithmetic
    (bit-identical to the oracle) and with the tolerance-contract FAST walk; roofline of the walk kernel priced in flops
    (it is VALU-bound); cpu_baseline = the oracle's World::update on the same bodies, timed in this run."""
    C = nb._capi
    from oracle import oracle as orc
    f64 = pos.dtype == np.float64
    n = pos.shape[0]
    order = C.ORDER_CONSISTENT if order is None else order
    kind_name = "quad" if kind == C.TREE_QUAD else "bvh"
    out = {"leg": name, "workload": workload, "workload_short": short, "dtype": "f64" if f64 else "f32", "steps": steps,
           "order": "as written (main.rs:398-423, SURVEY F6)" if order == C.ORDER_AS_WRITTEN else "consistent"}
    with C.Context(0) as ctx:
        ctx.set_params(theta=theta, order=order)
        for label, arith in (("exact", C.ARITH_AUTO), ("fast", C.ARITH_FAST)):
            ctx.set_params(arith=arith)
            ctx.upload(pos, vel, w)
            ctx.walk_stats(True)                   # one untimed walk with the counters on (3 atomics per target)
            ctx.accel_tree(kind)
            v0, a0, l0 = ctx.walk_stats(False)
            ctx.upload(pos, vel, w)
            timer = C.Timer()
            ctx.update_tree(kind, DT, 2)           # warm-up: allocations, the first build, the walk's history
            ctx.set_timer(timer)
            cnt = C.Counting()
            t0 = time.perf_counter()
            ctx.update_tree(kind, DT, steps, cnt)
            dt = time.perf_counter() - t0
            kms, kl = timer.read(reset=True)
            ctx.set_timer(None)
            ctx.walk_stats(True)                   # and the counts of the scene the timed steps ended on
            ctx.accel_tree(kind)
            v1, a1, l1 = ctx.walk_stats(False)
            visits, accepted, leaf_pairs = 0.5 * (v0 + v1), 0.5 * (a0 + a1), 0.5 * (l0 + l1)
            flops = PAIR_FLOPS_AS_WRITTEN * float(leaf_pairs + accepted) + NODE_TEST_FLOPS * float(visits)
            ach = flops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
            peak = PEAK_F64_TFLOPS if f64 else PEAK_F32_TFLOPS
            roof = {"bound": "valu_f64" if f64 else "valu_f32", "bound_class": "compute",
                    "kernel": {(True, "exact"): "nbody::tree_walk_small", (True, "fast"): "nbody::tree_walk_wave<FAST>",
                               (False, "exact"): "nbody::walk_tile", (False, "fast"): "nbody::walk_tile_fast"}[(kind == C.TREE_QUAD, label)],
                    "achieved": ach, "peak": peak,
                    "unit": "TFLOP/s", "frac": ach / peak, "kernel_ms": kms, "launches_timed": kl,
                    "pair_evaluations_per_launch": float(leaf_pairs + accepted), "node_tests_per_launch": float(visits),
                    "flops_per_pair": PAIR_FLOPS_AS_WRITTEN, "flops_per_node_test": NODE_TEST_FLOPS}
            # node records and leaf particles reach a wave by scalar loads, shared by its 64 targets: per-target counts / 64 is
            # the lower bound (every lane on the same path); plus the targets read and the accelerations written
            node_b, pair_b = (144, 24) if f64 else (80, 12)
            _traffic(roof, profile_names.get(label, "-"), int(visits * node_b / 64 + leaf_pairs * pair_b / 64 + n * (32 if f64 else 16)))
            out[label] = {"ms_per_step": 1e3 * dt / steps, "bodies_per_s": n * steps / dt, "build_ms": 1e3 * cnt.build_bvh / steps,
                          "walk_phase_ms": 1e3 * cnt.sum_gravity / steps, "integrate_ms": 1e3 * cnt.post_calculations / steps,
                          "interactions_per_s": (leaf_pairs + accepted) / (kms * 1e-3) if kms > 0 else None, "roofline": roof,
                          "parity": "bit-identical to the oracle" if label == "exact" else "tolerance of the direct kernel (2e-5); tree indexing bit-exact"}
            if label == "exact":
                out["walk"] = {"node_visits_per_target": visits / n, "accepted_per_target": accepted / n, "leaf_pairs_per_target": leaf_pairs / n}
    out["metric"] = "ms/step"
    out["value"] = out[default_arith]["ms_per_step"]
    out["roofline"] = out[default_arith]["roofline"]
    out["default_arithmetic"] = "exact (the reference's operations; FAST is opt-in, NBODY_ARITH_FAST)"
    if cpu_steps:
        out["cpu_baseline"] = _cpu_tree_steps(pos, vel, w, kind_name, theta,
                                              orc.AS_WRITTEN if order == C.ORDER_AS_WRITTEN else orc.CONSISTENT, cpu_steps)
    return out


# the order they print in: the driver keeps the tail of stdout, so the reference's live path and the BASELINE configs go last
LEGS = ("plummer1m_bvh", "reference_scene_direct", "mass_classes", "config2", "free_masses", "config4", "reference_scene_bvh")


def run_leg(nb, name, cpu=True):
    C = nb._capi
    if name == "reference_scene_bvh":
        # the ONE path the reference really runs: World::update (main.rs:388-425) with the BVH at THETA = 50 (main.rs:35),
        # leaf 64 (bvh_tree.rs:37), on World::new's scene (main.rs:276-346, seeded here), accelerations applied as written
        pos, vel, w = nb.scenes.galaxy()
        return _tree_leg(nb, name, f"the reference's live path: World::update, BVH, theta 50, leaf 64, on World::new's scene ({pos.shape[0]} bodies, "
                                   "masses 1 but for 75 000 000 and 750 000), f32, NBODY_ORDER_AS_WRITTEN", pos, vel, w, C.TREE_BVH, 50.0, 300,
                         {"exact": "r04_leg_reference_scene_bvh_pmc.json", "fast": "r04_leg_reference_scene_bvh_fast_pmc.json"},
                         order=C.ORDER_AS_WRITTEN, cpu_steps=10 if cpu else 0,
                         short=f"reference's live path: World::update BVH theta 50 leaf 64, World::new scene, {pos.shape[0]} bodies, f32")
    if name == "plummer1m_bvh":
        pos, vel, w = nb.scenes.plummer(N_BODIES, seed=SEED)
        return _tree_leg(nb, name, "World::update, BVH, theta 50, leaf 64, on the headline's bodies (1 048 576, seeded Plummer, masses 1), f32, "
                                   "NBODY_ORDER_AS_WRITTEN", pos, vel, w, C.TREE_BVH, 50.0, 20,
                         {"exact": "r04_leg_plummer1m_bvh_pmc.json", "fast": "r04_leg_plummer1m_bvh_fast_pmc.json"},
                         order=C.ORDER_AS_WRITTEN, cpu_steps=2 if cpu else 0,
                         short="World::update BVH theta 50 leaf 64, the headline's 1048576 Plummer bodies, f32")
    if name == "config2":
        pos, vel, w = nb.scenes.plummer(65536, seed=0x5EED0002)
        return _direct_leg(nb, name, "BASELINE.json configs[1]: 65 536 bodies direct O(N^2) f32, Plummer, masses 1", pos, vel, w, 200,
                           "r04_leg_config2_pmc.json", executed=FLOPS_EXECUTED_PER_PAIR, cpu_targets=65536 if cpu else 0,   # (from 2^32 pairs on the
                           short="BASELINE configs[1]: 65536 bodies direct f32, Plummer, masses 1")   # near/far split runs: the headline's instantiation; the CPU leg is one whole step)
    if name == "mass_classes":
        pos, vel, _ = nb.scenes.plummer(N_BODIES, seed=SEED)
        w = (np.arange(N_BODIES) % 5 + 1).astype(np.uint32)
        return _direct_leg(nb, name, "1 048 576 bodies direct f32, Plummer, per-body masses 1..5 (five mass classes: sources in class order, "
                                     "the equal-mass kernel tile by tile)", pos, vel, w, 3,
                           "r04_leg_mass_classes_pmc.json", executed=FLOPS_EXECUTED_PER_PAIR, cpu_targets=16384 if cpu else 0,
                           short="1048576 bodies direct f32, Plummer, masses 1..5 (five mass classes)")
    if name == "free_masses":
        pos, vel, _ = nb.scenes.plummer(N_BODIES, seed=SEED)
        w = nb.scenes.free_weights(N_BODIES, seed=SEED)
        return _direct_leg(nb, name, f"1 048 576 bodies direct f32, Plummer, free per-body u32 weights 1..100 000 ({len(np.unique(w))} distinct values: "
                                     "no classes; inverse masses streamed beside the couples, one packed multiply per couple more than the headline)",
                           pos, vel, w, 3, "r04_leg_free_masses_pmc.json", executed=FLOPS_EXECUTED_PER_PAIR + 1, cpu_targets=16384 if cpu else 0,
                           kernel=_main_pass_kernel(uniform=False),
                           short=f"1048576 bodies direct f32, Plummer, free u32 weights 1..100000 ({len(np.unique(w))} distinct)")
    if name == "reference_scene_direct":
        pos, vel, w = nb.scenes.galaxy()
        return _direct_leg(nb, name, f"the reference's own scene (World::new, main.rs:276-346, seeded): {pos.shape[0]} bodies, masses 1 but for "
                                     "75 000 000 and 750 000 — the two ride with the near list, the main pass runs at the equal-mass rate",
                           pos, vel, w, 50, "r04_leg_reference_scene_direct_pmc.json", executed=FLOPS_EXECUTED_PER_PAIR,
                           cpu_targets=32768 if cpu else 0, short=f"World::new scene direct f32, {pos.shape[0]} bodies, two heavy bodies in the near list")
    if name == "config4":
        pos, vel, w = nb.scenes.plummer(1 << 22, seed=0x5EED0004, dtype=np.float64)
        return _tree_leg(nb, name, "BASELINE.json configs[3]: 4 194 304 bodies Barnes-Hut theta 0.5, linearised quad tree, f64", pos, vel, w,
                         C.TREE_QUAD, 0.5, 5, {"exact": "r04_leg_config4_pmc.json", "fast": "r04_leg_config4_fast_pmc.json"},
                         cpu_steps=1 if cpu else 0, short="BASELINE configs[3]: 4194304 bodies Barnes-Hut theta 0.5, quad tree, f64")
    raise SystemExit(f"bench.py: unknown leg {name!r} (one of {LEGS})")


# ---------------------------------------------------------------------------------------------------- compact lines
# The driver keeps only the tail of stdout and parses the last line: the printed records are digests (numbers, kernel
# names, the profile file a traffic figure came from); the prose lives in profiles/README.md, the unabridged records
# in --full-out.
HEADLINE_MAX_BYTES = 1900
LEG_MAX_BYTES = 1000


def _r(x, sig=6):
    """Round a float to `sig` significant digits (bench lines are read by people and by a tail-limited parser)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x == 0.0 or x != x or x in (float("inf"), float("-inf")):
        return x
    return float(f"{x:.{sig}g}")


def _short_source(src):
    """'profiles/r04_x_pmc.json (rocprofv3 ...)' -> 'profiles/r04_x_pmc.json'"""
    return src.split(" ")[0] if src else None


def _compact_roofline(roof):
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_executed", "kernel_ms", "launches_timed", "traffic")
    out = {k: _r(roof[k]) for k in keep if k in roof}
    out["traffic_source"] = _short_source(roof.get("traffic_source"))
    if roof.get("algorithmic_bytes_per_launch") is not None:
        out["algorithmic_bytes"] = roof["algorithmic_bytes_per_launch"]
    return out


def _compact_cpu(cpu):
    if not cpu:
        return None
    out = {k: _r(cpu[k]) for k in ("value", "unit", "cores", "kind") if k in cpu}
    out["sample"] = cpu.get("sample_short") or cpu.get("sample", "")[:120]
    return out


def compact_leg(full):
    """One leg's digest: < LEG_MAX_BYTES."""
    if "error" in full:
        return {"leg": full["leg"], "error": full["error"][:300]}
    out = {"leg": full["leg"], "workload": full.get("workload_short") or full["workload"][:110], "metric": full["metric"],
           "value": _r(full["value"]), "ms_per_step": _r(full.get("ms_per_step", full["value"])), "steps": full["steps"],
           "dtype": full["dtype"]}
    if "exact" in full:                                    # a tree leg: both arithmetics, the walk kernel's fraction each
        for label in ("exact", "fast"):
            a = full[label]
            rf = a["roofline"]
            out[label] = {"ms_per_step": _r(a["ms_per_step"], 5), "build_ms": _r(a["build_ms"], 4), "walk_ms": _r(a["walk_phase_ms"], 4),
                          "kernel": rf["kernel"], "kernel_ms": _r(rf["kernel_ms"], 5), "tflops": _r(rf["achieved"], 4),
                          "frac": _r(rf["frac"], 4), "traffic": _r(rf.get("traffic"))}
        out["peak_tflops"] = full["roofline"]["peak"]
    else:
        out["roofline"] = _compact_roofline(full["roofline"])
    cpu = _compact_cpu(full.get("cpu_baseline"))
    if cpu:
        out["cpu_baseline"] = cpu
    return out


def _leg_digest(full):
    """What the headline line says about a leg: ms/step and the dominant kernel's fraction of its roofline."""
    if "error" in full:
        return "error"
    return [_r(full.get("ms_per_step", full["value"]), 5), _r(full["roofline"]["frac"], 4)]


def compact_headline(full, legs_full=()):
    """The driver's line: < HEADLINE_MAX_BYTES, everything the contract names and nothing explanatory."""
    out = {k: _r(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = full["config"]
    out["config"] = {k: cfg[k] for k in ("workload", "n_bodies", "targets_per_gpu", "chunks_per_step", "n_ranks", "entry", "exchange", "arith")
                     if k in cfg}
    out["roofline"] = _compact_roofline(full["roofline"])
    out["roofline"]["flops_per_pair"] = full["roofline"].get("flops_per_pair")
    if full.get("cpu_baseline"):
        out["cpu_baseline"] = _compact_cpu(full["cpu_baseline"])
    if legs_full:
        out["legs"] = {l["leg"]: _leg_digest(l) for l in legs_full}     # [ms/step, roofline frac]; full lines precede this one
    return out


def dumps_line(obj, limit):
    """json.dumps with the size contract enforced: a line the driver cannot read is worth nothing, so trim rather than
    exceed (drop the optional members, last resort the strings)."""
    line = json.dumps(obj, separators=(", ", ": "))
    for key in ("legs", "cpu_baseline.sample", "roofline.traffic_source", "config.entry", "config.exchange", "workload", "config.workload"):
        if len(line) < limit:
            break
        tgt, k = obj, key
        if "." in key:
            head, k = key.split(".")
            tgt = obj.get(head) or {}
        if k in tgt:
            if isinstance(tgt[k], str) and len(tgt[k]) > 40:
                tgt[k] = tgt[k][:40]
            else:
                del tgt[k]
        line = json.dumps(obj, separators=(", ", ": "))
    return line


def emit_headline(full, legs_full, full_out=None):
    """The LAST line of stdout: the compact headline.  The unabridged records go to `full_out` when asked for."""
    if full_out:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_out)), exist_ok=True)
            with open(full_out, "w") as f:
                json.dump(dict(full, legs=list(legs_full)), f)
        except OSError as e:
            print(f"bench.py: could not write {full_out}: {e}", file=sys.stderr)
    print(dumps_line(compact_headline(full, legs_full), HEADLINE_MAX_BYTES), flush=True)


def _headline_roofline(n, n_tgt, kern_ms, kern_launches, steps, single_gpu_full):
    n_launch = max(1, kern_launches // max(1, steps))      # launches of the dominant kernel per step (= chunks)
    flops_per_launch = FLOPS_PER_PAIR * float(n) * float(n_tgt) / n_launch
    achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    roof = {"bound": "valu_f32", "bound_class": "compute", "kernel": _main_pass_kernel(), "achieved": achieved,
            "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_TFLOPS,
            "flops_per_pair": FLOPS_PER_PAIR, "flops_executed_per_pair": FLOPS_EXECUTED_PER_PAIR,
            "frac_executed": achieved / PEAK_F32_TFLOPS * FLOPS_EXECUTED_PER_PAIR / FLOPS_PER_PAIR,
            "pairs_per_launch": float(n) * n_tgt / n_launch, "kernel_ms": kern_ms, "launches_timed": kern_launches}
    if single_gpu_full:
        _traffic(roof, "r04_direct_pmc.json", 36 * n)
    else:
        roof["traffic"] = None
        roof["traffic_source"] = None
    return roof


def _headline_line(args, n, world, elapsed, roof, config_extra):
    pairs_per_step = float(n) * float(n)
    cfg = {"workload": f"direct O(N^2) f32, N={n} bodies, seeded 2-D Plummer sphere, masses 1, "
                       f"dt={DT}, clamp={CLAMP} (BASELINE.json configs[2])",
           "n_bodies": n, "arith": "AUTO (FAST kernel; EXACT on hazardous positions)"}
    cfg.update(config_extra)
    return {"metric": "pair-interactions/sec + ms/step at N=1M direct O(N^2) (force + integration step)",
            "value": pairs_per_step * args.steps / elapsed, "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg, "roofline": roof}


def run_in_library(args):
    """`--gpus N` without an external launcher (no WORLD_SIZE): ONE process, the product's own multi-GPU path —
    nbody_create_multi (csrc/multi.hip): one worker thread per device, RCCL by ncclCommInitAll, the reference's single
    `world.update` call (main.rs:120) sharded inside the library.  What a Rust host would call."""
    import nbody_simulation_amd as nb
    C = nb._capi
    G = args.gpus
    n = args.n
    pos, vel, w = nb.scenes.plummer(n, seed=SEED)
    timer = C.Timer()
    exchange = C.EXCHANGE_PEER if os.environ.get("NBODY_MULTI_EXCHANGE", "") == "peer" else C.EXCHANGE_RCCL
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(G))
    if len(devices) != G:
        raise SystemExit(f"bench.py: --devices lists {len(devices)} devices, --gpus is {G}")
    with C.MultiContext(devices, exchange=exchange, chunks=0) as ctx:
        ctx.upload(pos, vel, w)
        ctx.set_timer(timer)
        if args.warmup:
            ctx.update_direct(DT, args.warmup)
        timer.read(reset=True)
        # update_direct returns when every device's stream has drained (hipStreamSynchronize per worker, then the
        # workers' join): the call itself is the barrier + synchronize bracket, and the wall time is the max over ranks
        t0 = time.perf_counter()
        ctx.update_direct(DT, args.steps)
        elapsed = time.perf_counter() - t0
        kern_ms, kern_launches = timer.read(reset=True)
        ctx.set_timer(None)
        g, xch, chunks, block = ctx.multi_info()
        n_ranks = ctx.comm_count()
    # device 0's targets (its kernels are the ones timed): blocks {c*G + 0}
    n_tgt = sum(max(0, min(block, n - c * g * block)) for c in range(chunks)) if g > 1 else n
    roof = _headline_roofline(n, n_tgt, kern_ms, kern_launches, args.steps, g == 1 and n == N_BODIES)
    out = _headline_line(args, n, g, elapsed, roof,
                         {"targets_per_gpu": n_tgt, "chunks_per_step": chunks, "block": block, "entry": "nbody_create_multi (one process, "
                          "one worker thread per device)", "n_ranks": n_ranks,
                          "exchange": ("RCCL in-place ncclAllGather of float2 positions per chunk (xGMI)" if xch == C.EXCHANGE_RCCL
                                       else "hipMemcpyPeerAsync of every block to every peer")})
    if g == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pos, w, args.cpu_sample_targets)
    emit_headline(out, [], args.full_out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bodies", dest="n", type=int, default=N_BODIES, help="total bodies (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the non-headline single-GPU configurations")
    ap.add_argument("--leg", default=None, help=f"run ONE leg alone and print its unabridged record (for profiling): {', '.join(LEGS)}")
    ap.add_argument("--full-out", default=None, help="also write the unabridged records (headline + legs) to this file")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                      "several ranks on one GPU)")
    ap.add_argument("--cpu-sample-targets", type=int, default=131072)
    ap.add_argument("--in-library", action="store_true", help="run the step through nbody_create_multi (one process) even for --gpus 1")
    ap.add_argument("--devices", default=None, help="in-library path: comma-separated device ids (default 0..gpus-1; one id may "
                                                    "repeat under NBODY_MULTI_EXCHANGE=peer to rehearse on one GPU)")
    args = ap.parse_args()

    # The launch convention is decided before anything touches a GPU, and nothing is ever re-executed: under an external
    # launcher (torch.distributed.run sets WORLD_SIZE) this is one rank of N processes; without one, `--gpus N > 1` runs
    # in this one process through the library's own multi-GPU context.
    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.in_library or (not launched and args.gpus > 1):
        if args.leg:
            raise SystemExit("bench.py: --leg is a single-GPU run")
        import nbody_simulation_amd  # noqa: F401  (loads the HIP library; fails loudly when it is missing)
        return run_in_library(args)

    import torch
    import torch.distributed as dist
    import nbody_simulation_amd as nb
    from nbody_simulation_amd.sharding import ShardedDirectStepper

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if args.leg:
        print(json.dumps(run_leg(nb, args.leg, cpu=not args.no_cpu_baseline)), flush=True)
        return
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
        n_tgt = stepper.n_local
        roof = _headline_roofline(n, n_tgt, kern_ms, kern_launches, args.steps, world == 1 and n == N_BODIES)
        out = _headline_line(args, n, world, elapsed, roof,
                             {"targets_per_gpu": n_tgt, "chunks_per_step": stepper.chunks,
                              "entry": "one process per GPU (sharding.py over nbody_direct_{prep,run}_dev)",
                              "n_ranks": dist.get_world_size() if world > 1 else 1,
                              "exchange": "none" if world == 1 else f"{args.backend} in-place all-gather of float2 positions per chunk"
                                          + (" (RCCL over xGMI)" if args.backend == "nccl" else " (rehearsal)")})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pos, w, args.cpu_sample_targets)
        legs = []
        if world == 1 and not args.no_legs and n == N_BODIES:
            del stepper
            torch.cuda.empty_cache()
            for name in LEGS:
                try:
                    leg = run_leg(nb, name, cpu=not args.no_cpu_baseline)
                except Exception as e:  # a leg must not cost the headline line
                    leg = {"leg": name, "error": repr(e)}
                legs.append(leg)
                print(dumps_line(compact_leg(leg), LEG_MAX_BYTES), flush=True)
        emit_headline(out, legs, args.full_out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

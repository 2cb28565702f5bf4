"""The sharded stepper with the real HIP backend: 1 rank, and a 2-rank rehearsal where both ranks share the one
GPU of the test box and exchange through gloo (RCCL needs one GPU per rank; the driver's 8-GPU run covers that)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, make_args, nprocs):
    """mp.spawn with a rendezvous port probed just before: between the probe and the children's bind another process can take the
    port (seen once on the GPU box: EADDRINUSE in the TCPStore, before any rank had touched the GPU).  That — and only that — is
    tried again with a fresh port; any other failure is the test's."""
    for attempt in range(3):
        try:
            mp.spawn(fn, args=make_args(_free_port()), nprocs=nprocs, join=True)
            return
        except Exception as e:  # noqa: BLE001
            if "EADDRINUSE" not in str(e) or attempt == 2:
                raise


def _worker(rank, world, port, n, steps, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nbody_simulation_amd as nb
    from nbody_simulation_amd.sharding import ShardedDirectStepper
    pos, vel, w = nb.scenes.plummer(n, seed=71)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    st = ShardedDirectStepper(pos, vel, w, rank=rank, world=world, device=torch.device("cuda", 0),
                              arith=nb._capi.ARITH_EXACT, group=dist.group.WORLD)
    for _ in range(steps):
        st.step(0.1)
    torch.cuda.synchronize()
    p, v = st.local_state()
    ret[rank] = (p, v, st.all_positions(), st.local_rows())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_oracle(orc, nb):
    n, steps, world = 4096, 3, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn(_worker, lambda port: (world, port, n, steps, ret), world)
    pos, vel, _ = nb.scenes.plummer(n, seed=71)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    rp, rv, _ = orc.update_direct(pos, vel, w, delta=0.1, nsteps=steps, nthreads=8)
    for r in range(world):
        p, v, allp, rows = ret[r]
        assert np.array_equal(p, rp[rows]) and np.array_equal(v, rv[rows])
        assert np.array_equal(allp, rp)


def test_single_rank_stepper_equals_context_path(nb):
    from nbody_simulation_amd.sharding import ShardedDirectStepper
    C = nb._capi
    n = 10000
    pos, vel, w = nb.scenes.plummer(n, seed=72)
    st = ShardedDirectStepper(pos, vel, w, device=torch.device("cuda", 0), arith=C.ARITH_AUTO)
    for _ in range(4):
        st.step(0.1)
    torch.cuda.synchronize()
    p, v = st.local_state()
    with C.Context(0) as ctx:
        ctx.upload(pos, vel, w)
        ctx.update_direct(0.1, 4)
        cp, cv, _, _ = ctx.download()
    assert np.array_equal(p, cp) and np.array_equal(v, cv)


def test_single_rank_stepper_with_free_masses_equals_context_path(nb, orc):
    """The device-pointer entry points (nbody_direct_prep_dev / _run_dev: what one process per GPU calls) with FREE per-body masses at a
    size where the near/far split is on: the streamed per-mass main pass (direct_stream_m) through this route gives the context
    route's bits, and both are inside the frozen tolerance of the oracle."""
    from nbody_simulation_amd.sharding import ShardedDirectStepper
    from tests._tol import check_fast
    C = nb._capi
    n = 70001
    pos, vel, _ = nb.scenes.plummer(n, seed=74)
    w = nb.scenes.free_weights(n, seed=74)
    st = ShardedDirectStepper(pos, vel, w, device=torch.device("cuda", 0), arith=C.ARITH_AUTO)
    st.step(0.1)
    torch.cuda.synchronize()
    p, v = st.local_state()
    with C.Context(0) as ctx:
        ctx.upload(pos, vel, w)
        acc = ctx.accel_direct()
        ctx.update_direct(0.1, 1)
        cp, cv, _, _ = ctx.download()
    assert np.array_equal(p, cp) and np.array_equal(v, cv)
    tg = np.arange(0, n, 23)
    ref64, norm = orc.direct_accel(pos, w, targets=tg, accum="f64", nthreads=16)
    check_fast(acc[tg], ref64, norm, label=" free masses, device-pointer route")
    assert np.array_equal(v, vel + acc * np.float32(0.1))


def test_bench_two_rank_rehearsal_via_torchrun():
    """bench.py under the launch line the task statement gives for N > 1 (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`: a convention taken from that text, not
    a recorded driver command), rehearsed with 2 ranks sharing the test box's one GPU and gloo standing in for RCCL."""
    import json
    import subprocess
    for attempt in range(3):                  # (a probed port can be taken before the launcher binds it: only that is tried again)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--bodies", "65536", "--backend", "gloo"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["unit"] == "pair-interactions/s"
    assert d["config"]["targets_per_gpu"] == 32768 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert "cpu_baseline" not in d            # rank 0 at N = 1 only


def _bench(args, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_without_a_launcher_runs_the_library_multi_gpu_path():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment must not die on a launch convention:
    it runs ONE process through nbody_create_multi — what a Rust host would call.  On the one-GPU test box: the in-library
    route with one device under RCCL (communicator, in-place ncclAllGather of one rank) gives the plain single-GPU line's
    value to within a few per cent, and `--gpus 2 --devices 0,0` (peer-copy exchange: two ranks sharing the device)
    reports 2 ranks, half the targets each, and the same metric and workload."""
    common = ["--steps", "3", "--warmup", "1", "--bodies", "262144", "--no-cpu-baseline", "--no-legs"]
    plain = _bench(["--gpus", "1"] + common)
    lib1 = _bench(["--gpus", "1", "--in-library"] + common)
    assert plain["n_gpus"] == lib1["n_gpus"] == 1 and plain["metric"] == lib1["metric"] and plain["unit"] == lib1["unit"]
    assert lib1["config"]["n_ranks"] == 1 and "nbody_create_multi" in lib1["config"]["entry"] and "RCCL" in lib1["config"]["exchange"]
    assert abs(lib1["value"] / plain["value"] - 1.0) < 0.05, (lib1["value"], plain["value"])
    assert abs(lib1["roofline"]["frac"] / plain["roofline"]["frac"] - 1.0) < 0.05
    lib2 = _bench(["--gpus", "2", "--devices", "0,0"] + common, env={"NBODY_MULTI_EXCHANGE": "peer"})
    assert lib2["n_gpus"] == 2 and lib2["config"]["targets_per_gpu"] == 131072 and lib2["config"]["n_ranks"] == 0   # no RCCL communicator under peer copies
    assert lib2["scaling"] == "strong" and lib2["steps"] == 3 and lib2["value"] > 0.5 * plain["value"]
    assert "hipMemcpyPeerAsync" in lib2["config"]["exchange"]


def _tree_worker(rank, world, port, kind, dtype_name, order, steps, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nbody_simulation_amd as nb
    from nbody_simulation_amd.sharding import ShardedTreeStepper
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(8192, seed=73, dtype=np.dtype(dtype_name).type)
    w = (np.arange(8192) % 3 + 1).astype(np.uint32)
    st = ShardedTreeStepper(pos, vel, w, kind=kind, rank=rank, world=world, group=dist.group.WORLD, theta=0.5,
                            order=C.ORDER_AS_WRITTEN if order == "as_written" else C.ORDER_CONSISTENT)
    for _ in range(steps):
        st.step(0.1)
    ret[rank] = st.particles()
    st.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind_name,dtype_name,order", [("quad", "float32", "consistent"), ("quad", "float64", "consistent"),
                                                        ("bvh", "float32", "as_written"), ("bvh", "float32", "consistent")])
def test_sharded_tree_steps_equal_single_context(orc, nb, kind_name, dtype_name, order):
    """2 ranks (sharing the test box's GPU, gloo standing in for RCCL) each walk half of the tree-ordered targets and
    exchange rows; after 3 steps every rank's full state equals the oracle's single-process trajectory bit for bit."""
    C = nb._capi
    kind = C.TREE_QUAD if kind_name == "quad" else C.TREE_BVH
    steps, world = 3, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn(_tree_worker, lambda port: (world, port, kind, dtype_name, order, steps, ret), world)
    pos, vel, _ = nb.scenes.plummer(8192, seed=73, dtype=np.dtype(dtype_name).type)
    w = (np.arange(8192) % 3 + 1).astype(np.uint32)
    if kind_name == "quad":
        rp, rv, _ = orc.update_quad(pos, vel, w, delta=0.1, theta=0.5, nsteps=steps, nthreads=8)
        rids = np.arange(8192, dtype=np.uint32)
    else:
        mode = orc.AS_WRITTEN if order == "as_written" else orc.CONSISTENT
        rp, rv, _, rids, _ = orc.update_bvh(pos, vel, w, delta=0.1, theta=0.5, mode=mode, nsteps=steps, nthreads=8)
    for r in range(world):
        p, v, _, ids = ret[r]
        assert np.array_equal(ids, rids)
        assert np.array_equal(p, rp) and np.array_equal(v, rv)


# ------------------------------------------------------------------ RCCL between two GPUs (needs >= 2 visible devices)
def _nccl_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import nbody_simulation_amd as nb
    from nbody_simulation_amd.sharding import ShardedDirectStepper, ShardedTreeStepper
    C = nb._capi
    n = 20011
    pos, vel, _ = nb.scenes.plummer(n, seed=74)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    st = ShardedDirectStepper(pos, vel, w, rank=rank, world=world, device=dev, arith=C.ARITH_EXACT, group=dist.group.WORLD, chunks=2)
    for _ in range(3):
        st.step(0.1)
    torch.cuda.synchronize()
    p, v = st.local_state()
    out = {"direct": (p, v, st.all_positions(), st.local_rows())}
    for name, kind, order in (("quad", C.TREE_QUAD, C.ORDER_CONSISTENT), ("bvh", C.TREE_BVH, C.ORDER_AS_WRITTEN)):
        ts = ShardedTreeStepper(pos, vel, w, kind=kind, rank=rank, world=world, device_index=rank, group=dist.group.WORLD, theta=0.5,
                                order=order)
        for _ in range(3):
            ts.step(0.1)
        out[name] = ts.particles()
        ts.close()
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_two_gpus_rccl(orc, nb):
    """The 'nccl' (= RCCL) branches of both steppers, one process per GPU: in-place chunked all-gather of the positions,
    one packed all-gather of a tree step's slices on the context's stream.  Bit-identical to the oracle's trajectories."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL between ranks runs on the driver's multi-GPU node")
    world, n = 2, 20011
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn(_nccl_worker, lambda port: (world, port, ret), world)
    pos, vel, _ = nb.scenes.plummer(n, seed=74)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    rp, rv, _ = orc.update_direct(pos, vel, w, delta=0.1, nsteps=3, nthreads=8)
    qp, qv, _ = orc.update_quad(pos, vel, w, delta=0.1, theta=0.5, nsteps=3, nthreads=8)
    bp, bv, _, bids, _ = orc.update_bvh(pos, vel, w, delta=0.1, theta=0.5, mode=orc.AS_WRITTEN, nsteps=3, nthreads=8)
    for r in range(world):
        p, v, allp, rows = ret[r]["direct"]
        assert np.array_equal(p, rp[rows]) and np.array_equal(v, rv[rows]) and np.array_equal(allp, rp)
        p, v, _, ids = ret[r]["quad"]
        assert np.array_equal(ids, np.arange(n)) and np.array_equal(p, qp) and np.array_equal(v, qv)
        p, v, _, ids = ret[r]["bvh"]
        assert np.array_equal(ids, bids) and np.array_equal(p, bp) and np.array_equal(v, bv)


# ------------------------------------------------------------------ the torch 'nccl' code path with ONE rank (real RCCL)
def _one_rank_nccl_worker(rank, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import nbody_simulation_amd as nb
    from nbody_simulation_amd.sharding import ShardedDirectStepper, ShardedTreeStepper
    C = nb._capi
    n = 70001
    pos, vel, _ = nb.scenes.plummer(n, seed=75)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    st = ShardedDirectStepper(pos, vel, w, rank=0, world=1, device=dev, arith=C.ARITH_AUTO, group=dist.group.WORLD, chunks=3,
                              exchange_always=True)
    for _ in range(4):
        st.step(0.1)
    torch.cuda.synchronize()
    out = {"direct": (st.all_positions(), st.local_state()[1], st._inplace)}
    for name, kind in (("quad", C.TREE_QUAD), ("bvh", C.TREE_BVH)):
        ts = ShardedTreeStepper(pos, vel, w, kind=kind, rank=0, world=1, device_index=0, group=dist.group.WORLD, theta=0.5,
                                exchange_always=True)
        for _ in range(3):
            ts.step(0.1)
        out[name] = (ts.particles(), ts._inplace)
        ts.close()
    ret[0] = out
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_over_the_nccl_backend(nb):
    """What one GPU can say about the torchrun path the driver's multi-GPU bench takes: the process group is 'nccl' (RCCL),
    the collectives are issued although there is one rank — the in-place, asynchronous, chunked all-gather of the positions
    behind each chunk's kernels; the packed all-gather of a tree step on the context's own stream — and the results equal
    the plain context's.  (Aliased send / receive buffers, async_op + wait, ExternalStream: the torch API usage is what
    this checks; the exchange between ranks needs two GPUs, test_two_ranks_two_gpus_rccl.)"""
    C = nb._capi
    mgr = mp.Manager()
    ret = mgr.dict()
    _spawn(_one_rank_nccl_worker, lambda port: (port, ret), 1)
    n = 70001
    pos, vel, _ = nb.scenes.plummer(n, seed=75)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    allp, v, inplace = ret[0]["direct"]
    assert inplace                                           # the aliased (in-place) all-gather was accepted
    with C.Context(0) as c:
        c.upload(pos, vel, w)
        c.update_direct(0.1, 4)
        cp, cv, _, _ = c.download()
    # three chunks of a third of the targets each instead of one launch: another split of the sources over blockIdx.y, so
    # the FAST sums differ by summation order; the exchange itself must not change a bit: compare through EXACT below
    assert np.allclose(allp, cp, rtol=0, atol=1e-2) and np.allclose(v, cv, rtol=1e-3, atol=1e-6)
    for name, kind in (("quad", C.TREE_QUAD), ("bvh", C.TREE_BVH)):
        (p, vv, ww, ids), inplace = ret[0][name]
        assert inplace
        with C.Context(0) as c:
            c.set_params(theta=0.5)
            c.upload(pos, vel, w)
            c.update_tree(kind, 0.1, 3)
            ref = c.download()
        assert all(np.array_equal(a, b) for a, b in zip((p, vv, ww, ids), ref)), name

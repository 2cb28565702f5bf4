"""One-process-per-GPU sharding of the direct step.

The reference's only parallel region is a map over targets with read-only sources (main.rs:406-416), so the
step shards by target: rank r owns the contiguous block [r*N/G, (r+1)*N/G) of bodies (positions + velocities),
keeps a replicated copy of ALL positions and masses, computes force + integration for its block with
nbody_direct_step_dev, and the new positions are exchanged with ONE all-gather per step
(torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
Velocities never leave their rank; masses are static.

`backend` is the compute engine for one shard.  The product default is HipBackend (C ABI -> HIP kernels; it
raises without a GPU: there is no CPU fallback).  The world_size-2 gloo tests inject the CPU oracle instead, to
exercise exactly this file's partitioning and exchange logic on a machine without GPUs.
"""
from __future__ import annotations

import numpy as np

from . import _capi


class HipBackend:
    """Runs a shard's step through the C ABI on the current torch stream of `device`."""

    def __init__(self, device, n_sources, n_local, clamp, arith, timer=None, uniform_mass=0.0):
        import torch
        if device.type != "cuda":
            raise _capi.NBodyError(_capi.ERR_NO_DEVICE, "HipBackend needs a CUDA/HIP device; there is no CPU fallback")
        self.torch = torch
        self.device = device
        self.clamp, self.arith, self.timer, self.uniform_mass = clamp, arith, timer, uniform_mass
        self.ws_bytes = _capi.direct_workspace_bytes(n_sources, n_local)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)

    def step(self, pos_all, mass_all, begin, n_local, vel_shard, out_shard, dt):
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        _capi.direct_step_dev(stream, pos_all.shape[0], pos_all.data_ptr(), mass_all.data_ptr(), begin, n_local,
                              vel_shard.data_ptr(), out_shard.data_ptr(), None, dt, self.clamp, self.arith,
                              self.ws.data_ptr(), self.ws_bytes, self.timer, uniform_mass=self.uniform_mass)


class ShardedDirectStepper:
    def __init__(self, pos, vel, weight, *, rank=0, world=1, device=None, clamp=0.001, arith=_capi.ARITH_AUTO,
                 timer=None, group=None, backend=None):
        import torch
        self.torch = torch
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 2)
        weight = np.ascontiguousarray(weight, np.uint32)
        n = pos.shape[0]
        if n % world:
            raise ValueError(f"N={n} must be divisible by the number of ranks ({world})")
        self.n, self.rank, self.world, self.group = n, rank, world, group
        self.n_local = n // world
        self.begin = rank * self.n_local
        device = device if device is not None else torch.device("cuda", 0)
        self.device = device
        uniform = float(weight[0]) if n > 0 and weight[0] > 0 and bool(np.all(weight == weight[0])) else 0.0
        self.pos_all = torch.from_numpy(pos).to(device)                       # replicated, read by the kernel
        self.mass_all = torch.from_numpy(weight.astype(np.float32)).to(device)  # `weight as f32`, main.rs:360
        self.vel = torch.from_numpy(vel[self.begin:self.begin + self.n_local].copy()).to(device)
        self.out_shard = torch.empty((self.n_local, 2), dtype=torch.float32, device=device)
        self.pos_next = torch.empty_like(self.pos_all) if world > 1 else None
        self.backend = backend if backend is not None else HipBackend(device, n, self.n_local, clamp, arith, timer,
                                                                      uniform)

    def step(self, dt):
        """One World::update: force + integrate for the local block, then the position exchange."""
        self.backend.step(self.pos_all, self.mass_all, self.begin, self.n_local, self.vel, self.out_shard, dt)
        if self.world == 1:
            # single rank: the shard is the whole array; swap buffers
            self.pos_all, self.out_shard = self.out_shard, self.pos_all
        else:
            import torch.distributed as dist
            if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":
                # rehearsal mode (several ranks sharing one GPU, no RCCL between them): stage through the host
                host = self.torch.empty((self.n, 2), dtype=self.torch.float32)
                dist.all_gather_into_tensor(host.view(-1), self.out_shard.cpu().view(-1), group=self.group)
                self.pos_next.copy_(host)
            else:
                dist.all_gather_into_tensor(self.pos_next.view(-1), self.out_shard.view(-1), group=self.group)
            self.pos_all, self.pos_next = self.pos_next, self.pos_all

    def local_state(self):
        """-> (positions of the local block, velocities of the local block) as numpy arrays."""
        p = self.pos_all[self.begin:self.begin + self.n_local]
        return p.cpu().numpy(), self.vel.cpu().numpy()

    def all_positions(self):
        return self.pos_all.cpu().numpy()


class ShardedTreeStepper:
    """Barnes-Hut steps sharded over ranks (second bullet).

    Every rank holds ALL particles in its own context and builds the same tree (the quad tree on the device, the BVH
    on the host: both deterministic); rank r walks and integrates only the r-th slice of the tree-ordered targets
    (tree order keeps a wave's 64 targets close together).  The exchange step: each rank exports {row, position,
    velocity} of its slice into device buffers, one all-gather per array (RCCL over xGMI; gloo staging in the
    rehearsal), and every rank imports all rows, after which all contexts hold the same state again.  Results are
    bit-identical to the single-context step: a target's walk does not depend on who performs it."""

    def __init__(self, pos, vel, weight, *, kind=_capi.TREE_QUAD, rank=0, world=1, device_index=0, group=None, **params):
        import torch
        self.torch = torch
        self.kind, self.rank, self.world, self.group = kind, rank, world, group
        self.ctx = _capi.Context(device_index)
        if params:
            self.ctx.set_params(**params)
        self.ctx.upload(pos, vel, weight)
        n = self.ctx.n
        if n % world:
            raise ValueError(f"N={n} must be divisible by the number of ranks ({world})")
        self.n, self.n_local = n, n // world
        self.begin = rank * self.n_local
        dev = torch.device("cuda", device_index)
        self.device = dev
        ft = torch.float64 if self.ctx.dtype == np.float64 else torch.float32
        self.rows = torch.empty(self.n_local, dtype=torch.int32, device=dev)
        self.pos = torch.empty((self.n_local, 2), dtype=ft, device=dev)
        self.vel = torch.empty((self.n_local, 2), dtype=ft, device=dev)
        if world > 1:
            self.all_rows = torch.empty(n, dtype=torch.int32, device=dev)
            self.all_pos = torch.empty((n, 2), dtype=ft, device=dev)
            self.all_vel = torch.empty((n, 2), dtype=ft, device=dev)

    def _all_gather(self, out, inp):
        import torch.distributed as dist
        if dist.get_backend(self.group) == "gloo":   # rehearsal: several ranks on one GPU, stage through the host
            host = self.torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host.view(-1), inp.cpu().view(-1), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group)

    def step(self, dt, counter=None):
        self.ctx.update_tree_shard(self.kind, dt, self.begin, self.n_local, counter)
        if self.world == 1:
            return
        self.ctx.export_slice_dev(self.begin, self.n_local, self.rows.data_ptr(), self.pos.data_ptr(), self.vel.data_ptr())
        self._all_gather(self.all_rows, self.rows)
        self._all_gather(self.all_pos, self.pos)
        self._all_gather(self.all_vel, self.vel)
        self.torch.cuda.synchronize(self.device)
        self.ctx.import_rows_dev(self.n, self.all_rows.data_ptr(), self.all_pos.data_ptr(), self.all_vel.data_ptr())

    def particles(self):
        return self.ctx.download()

    def close(self):
        self.ctx.close()

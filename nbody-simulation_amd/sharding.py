"""One-process-per-GPU sharding of a step: what a torchrun / MPI host does above the device-pointer level
of the C ABI.  (A single-process host needs none of this: nbody_create_multi shards inside the library, csrc/multi.hip.)

The reference's only parallel region is a map over targets with read-only sources (main.rs:406-416), so the step
shards by target.  Layout (the same as csrc/multi.hip): with G ranks and C chunks per step, the bodies are cut into
G*C blocks of `block` rows; rank r owns blocks {c*G + r : c < C}.  Block c*G + r is computed straight into its place
in the NEXT position array, so chunk c = blocks [c*G, (c+1)*G) is one contiguous region and its exchange is ONE
in-place all-gather (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests),
issued asynchronously right after the chunk's kernels while the next chunk computes; only the last chunk's gather is
exposed.  Velocities never leave their rank; masses are static.  Any N works (the last blocks are short or empty).

`backend` is the compute engine for one block.  The product default is HipBackend (C ABI -> HIP kernels; it raises
without a GPU: there is no CPU fallback).  The world_size-2 gloo tests inject the CPU oracle instead, to exercise
exactly this file's partitioning and exchange logic on a machine without GPUs.
"""
from __future__ import annotations

import numpy as np

from . import _capi


def block_layout(n: int, world: int, chunks: int = 0):
    """-> (chunks, block): blocks of `block` rows (a multiple of 64: whole waves of targets), world * chunks of them
    cover n.  chunks = 0 picks by size: a chunk should still fill the chip (>= 262 144 targets per rank and chunk), so
    N = 1 M on 8 ranks steps in one chunk and N = 16.7 M on 8 ranks in eight (csrc/multi.hip, choose_chunks)."""
    if chunks <= 0:
        chunks = 1 if world == 1 else max(1, min(8, -(-n // world) // 262144))
    chunks = max(1, min(16, chunks))
    block = -(-n // (world * chunks))
    block = max(64, (block + 63) // 64 * 64)
    return chunks, block


class HipBackend:
    """Runs a rank's blocks through the C ABI on the current torch stream of `device`: one preparation per step over all
    positions (hazard scan, near/far split), one run per block."""

    def __init__(self, device, n_sources, n_local_total, n_block_max, clamp, arith, timer=None, uniform_mass=0.0):
        import torch
        if device.type != "cuda":
            raise _capi.NBodyError(_capi.ERR_NO_DEVICE, "HipBackend needs a CUDA/HIP device; there is no CPU fallback")
        self.torch = torch
        self.device = device
        self.clamp, self.arith, self.timer, self.uniform_mass = clamp, arith, timer, uniform_mass
        self.n_total, self.n_max = int(n_local_total), int(n_block_max)
        self.ws_bytes = _capi.direct_workspace_bytes(n_sources, self.n_max)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def prep(self, pos_all, mass_all, n):
        _capi.direct_prep_dev(self._stream(), n, pos_all.data_ptr(), mass_all.data_ptr(), self.n_total, self.n_max, self.clamp,
                              self.arith, self.ws.data_ptr(), self.ws_bytes, uniform_mass=self.uniform_mass)

    def run(self, pos_all, mass_all, n, begin, count, vel_block, out_block, dt):
        _capi.direct_run_dev(self._stream(), n, pos_all.data_ptr(), mass_all.data_ptr(), begin, count, vel_block.data_ptr(),
                             out_block.data_ptr(), None, dt, self.clamp, self.arith, self.n_total, self.n_max, self.ws.data_ptr(),
                             self.ws_bytes, self.timer, uniform_mass=self.uniform_mass)


class ShardedDirectStepper:
    def __init__(self, pos, vel, weight, *, rank=0, world=1, device=None, clamp=0.001, arith=_capi.ARITH_AUTO,
                 timer=None, group=None, backend=None, chunks=0, exchange_always=False):
        """exchange_always: issue the collectives even with one rank (a one-rank all-gather is a no-op that still goes
        through RCCL: how the exchange code is exercised on a one-GPU box)."""
        import torch
        self.torch = torch
        self.exchange_always = exchange_always
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 2)
        weight = np.ascontiguousarray(weight, np.uint32)
        n = pos.shape[0]
        self.n, self.rank, self.world, self.group = n, rank, world, group
        self.chunks, self.block = block_layout(n, world, chunks)
        cap = world * self.chunks * self.block                     # rows the arrays hold (>= n: whole blocks)
        self.blocks = [((c * world + rank) * self.block, max(0, min(self.block, n - (c * world + rank) * self.block)))
                       for c in range(self.chunks)]                # (first row, rows) of this rank's blocks
        self.n_local = sum(cnt for _, cnt in self.blocks)
        device = device if device is not None else torch.device("cuda", 0)
        self.device = device
        uniform = _capi.mass_hint(weight)    # > 0 equal masses, < 0 equal but for a few (added after the main pass), 0 neither

        def padded(a, rows):
            out = np.zeros((rows,) + a.shape[1:], a.dtype)
            out[:a.shape[0]] = a
            return torch.from_numpy(out).to(device)

        self.pos_all = padded(pos, cap)                            # replicated, read by the kernel (rows >= n never are)
        self.pos_next = torch.zeros_like(self.pos_all)
        self.mass_all = padded(weight.astype(np.float32), cap)     # `weight as f32`, main.rs:360
        self.vel = padded(vel, cap)                                # only this rank's blocks are kept up to date
        n_max = max([cnt for _, cnt in self.blocks] + [0])
        self.backend = backend if backend is not None else HipBackend(device, n, self.n_local, n_max, clamp, arith, timer, uniform)
        self._inplace = True

    def _gather_chunk(self, c):
        """In-place all-gather of chunk c of pos_next (rank r's piece = block c*world + r).  Returns a work handle or None."""
        import torch.distributed as dist
        m, g = self.block, self.world
        region = self.pos_next[c * g * m:(c + 1) * g * m]
        mine = region[self.rank * m:(self.rank + 1) * m]
        backend = dist.get_backend(self.group)
        if self.device.type == "cuda" and backend == "gloo":
            # rehearsal mode (several ranks sharing one GPU, no RCCL between them): stage through the host
            host = self.torch.empty((g * m, 2), dtype=self.torch.float32)
            dist.all_gather_into_tensor(host.view(-1), mine.cpu().view(-1), group=self.group)
            region.copy_(host)
            return None
        if backend == "gloo":                                      # CPU tensors: gloo is not promised to accept aliasing
            dist.all_gather_into_tensor(region.view(-1), mine.clone().view(-1), group=self.group)
            return None
        # RCCL: sendbuff = recvbuff + rank * count is NCCL's in-place all-gather.  async_op: the collective runs on the
        # process group's own stream, ordered after what the current stream holds now (this chunk's kernels); the next
        # chunk's kernels, enqueued next on the current stream, overlap it.
        if self._inplace:
            try:
                return dist.all_gather_into_tensor(region.view(-1), mine.view(-1), group=self.group, async_op=True)
            except (RuntimeError, ValueError):   # a torch build that refuses aliased buffers: send from a copy instead
                self._inplace = False
        return dist.all_gather_into_tensor(region.view(-1), mine.clone().view(-1), group=self.group, async_op=True)

    def step(self, dt):
        """One World::update: force + integrate for this rank's blocks, each chunk's exchange behind its kernels."""
        self.backend.prep(self.pos_all, self.mass_all, self.n)
        pending = []
        for c, (begin, count) in enumerate(self.blocks):
            if count > 0:
                self.backend.run(self.pos_all, self.mass_all, self.n, begin, count, self.vel[begin:begin + count],
                                 self.pos_next[begin:begin + count], dt)
            if self.world > 1 or self.exchange_always:
                w = self._gather_chunk(c)
                if w is not None:
                    pending.append(w)
        for w in pending:
            w.wait()          # the current stream waits for the collective (no host wait): the next preparation needs every row
        self.pos_all, self.pos_next = self.pos_next, self.pos_all

    def local_rows(self):
        """Indices of the bodies this rank owns, in block order."""
        return np.concatenate([np.arange(b, b + c) for b, c in self.blocks] + [np.zeros(0, np.int64)]).astype(np.int64)

    def local_state(self):
        """-> (positions, velocities) of the rows of local_rows(), as numpy arrays."""
        rows = self.torch.from_numpy(self.local_rows()).to(self.device)
        return self.pos_all[rows].cpu().numpy(), self.vel[rows].cpu().numpy()

    def all_positions(self):
        return self.pos_all[:self.n].cpu().numpy()


class ShardedTreeStepper:
    """Barnes-Hut steps sharded over ranks (second bullet).

    Every rank holds ALL particles in its own context and builds the same tree (on the device: deterministic); rank r
    walks and integrates only the r-th slice of the tree-ordered targets (tree order keeps a wave's 64 targets close
    together).  The exchange step: each rank exports {row, position, velocity} of its slice into its section of ONE
    packed device buffer, ONE all-gather of that buffer (RCCL over xGMI; gloo staging in the rehearsal), and every rank
    imports the other ranks' sections, after which all contexts hold the same state again.  Everything is enqueued on
    the context's own stream (torch sees it as an ExternalStream), so there is no device-wide synchronisation.
    Results are bit-identical to the single-context step: a target's walk does not depend on who performs it."""

    def __init__(self, pos, vel, weight, *, kind=_capi.TREE_QUAD, rank=0, world=1, device_index=0, group=None,
                 exchange_always=False, **params):
        import torch
        self.torch = torch
        self.exchange_always = exchange_always
        self.kind, self.rank, self.world, self.group = kind, rank, world, group
        self.ctx = _capi.Context(device_index)
        if params:
            self.ctx.set_params(**params)
        self.ctx.upload(pos, vel, weight)
        n = self.ctx.n
        self.n = n
        self.slice = -(-n // world) if n else 0                    # rows per rank (the last slices are short or empty)
        self.begin = min(rank * self.slice, n)
        self.n_local = max(0, min(self.slice, n - rank * self.slice))
        dev = torch.device("cuda", device_index)
        self.device = dev
        es = 16 if self.ctx.dtype == np.float64 else 8             # bytes of one xy pair
        rows = max(self.slice, 1)
        a256 = lambda v: (v + 255) // 256 * 256                    # noqa: E731
        self.off_pos = a256(rows * 4)
        self.off_vel = self.off_pos + a256(rows * es)
        self.sec = self.off_vel + a256(rows * es)                  # one rank's section: rows u32 | positions | velocities
        self.buf = torch.zeros(self.sec * world, dtype=torch.uint8, device=dev)
        self.stream = torch.cuda.ExternalStream(self.ctx.stream, device=dev)
        self._inplace = True

    def _count(self, r):
        return max(0, min(self.slice, self.n - r * self.slice))

    def step(self, dt, counter=None):
        self.ctx.update_tree_shard(self.kind, dt, self.begin, self.n_local, counter)
        if self.world == 1 and not self.exchange_always:
            return
        import torch.distributed as dist
        base = self.buf.data_ptr()
        mine = base + self.rank * self.sec
        if self.n_local:
            self.ctx.export_slice_dev(self.begin, self.n_local, mine, mine + self.off_pos, mine + self.off_vel)
        section = self.buf[self.rank * self.sec:(self.rank + 1) * self.sec]
        with self.torch.cuda.stream(self.stream):                  # the context's stream: ordered after the export, before the import
            if dist.get_backend(self.group) == "gloo":             # rehearsal: several ranks on one GPU, stage through the host
                host = self.torch.empty(self.sec * self.world, dtype=self.torch.uint8)
                dist.all_gather_into_tensor(host, section.cpu(), group=self.group)
                self.buf.copy_(host)
            else:                                                  # in place: sendbuff = recvbuff + rank * count
                try:
                    dist.all_gather_into_tensor(self.buf, section if self._inplace else section.clone(), group=self.group)
                except (RuntimeError, ValueError):                 # a torch build that refuses aliased buffers
                    if not self._inplace:
                        raise
                    self._inplace = False
                    dist.all_gather_into_tensor(self.buf, section.clone(), group=self.group)
        for r in range(self.world):
            cnt = self._count(r)
            if r == self.rank or cnt == 0:
                continue
            sec = base + r * self.sec
            self.ctx.import_rows_dev(cnt, sec, sec + self.off_pos, sec + self.off_vel)

    def particles(self):
        return self.ctx.download()

    def close(self):
        self.ctx.close()

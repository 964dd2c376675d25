"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  The training path shards by batch: every rank runs the same plans on its own rows and
the only exchange is the gradient all-reduce of the two flat gradient buffers (22.8 MB critic per iteration,
23.2 MB generator every n_critic-th iteration).  Sampling needs no collective."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


_OWN_PG = False          # the default process group was created by init_from_env() (so shutdown() may destroy it)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the default process group if world > 1."""
    global _OWN_PG
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:     # WFD_DIST_BACKEND=gloo: test rigs where several ranks share one GPU
            backend = os.environ.get("WFD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        _OWN_PG = True
    return rank, world, local


def shutdown(destroy: Optional[bool] = None) -> None:
    """Ordered end of a data-parallel run.  The CALLER first drops every object that holds a HIP graph with captured RCCL
    nodes (engines, grouped trainers, tail engines: ``del eng, grp, tail``); then this does
    barrier -> gc.collect() -> torch.cuda.synchronize() -> destroy_process_group().
    Why the order matters (seen in round 2 as an abort in the test that FOLLOWED a data-parallel one, main thread in plain
    tensor copies, no capture active): a graph whose nodes were captured from a communicator's collectives outlived that
    communicator; when the allocator later collected it, RCCL aborted the process.  The barrier keeps the other ranks
    alive until rank 0 has finished writing its artefacts (they would otherwise tear their communicator down under rank
    0's last collective).  ``destroy``: None = only a group that init_from_env() created (callers that brought their own
    group keep it); True / False force it."""
    global _OWN_PG
    import gc
    if not (dist.is_available() and dist.is_initialized()):
        return
    dist.barrier()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    for ref in list(_MESHES):               # peer mappings of the mesh exchange go while the ranks can still meet
        m = ref()
        if m is not None:
            m.close()
    _MESHES.clear()
    if destroy if destroy is not None else _OWN_PG:
        dist.destroy_process_group()
        _OWN_PG = False


def graph_capturable(group=None) -> bool:
    """True when this group's collectives may be issued inside a HIP-graph capture: the "nccl" (RCCL) backend enqueues
    them on its own stream ordered by events against the capturing stream, so start / wait become graph edges and a whole
    data-parallel step replays as ONE graph.  gloo (host staging) cannot.  WFD_DP_GRAPH=0 keeps the collectives out of the
    graphs (eager, between separately captured stages)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True                  # no process group: the "collectives" are identities
    if os.environ.get("WFD_DP_GRAPH", "1") == "0":
        return False
    return dist.get_backend(group) == "nccl"


def graph_capture_kwargs() -> dict:
    """Keyword arguments for ``torch.cuda.graph(...)``.  Once a process group exists, its backend's watchdog thread makes
    HIP calls of its own at any time (event queries on earlier collectives); under the default (global) capture mode such
    a call from ANOTHER thread invalidates a capture that is in progress on this one -- also a capture that contains no
    collective at all (seen as `hipErrorStreamCaptureInvalidated` / an abort of the watchdog).  The thread-local mode only
    polices the capturing thread."""
    if dist.is_available() and dist.is_initialized():
        return dict(capture_error_mode="thread_local")
    return {}


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of a flat gradient buffer over all ranks (the optimizer applies the 1/world factor)."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class Comm:
    """The small collectives of the exact (global-batch) data-parallel mode on device tensors: BatchNorm statistics,
    BatchNorm backward sums, the selected-row count, loss scalars.  With RCCL ("nccl") they run in place on the
    device; with "gloo" (CPU tests, or several test ranks sharing one GPU) device tensors are staged through the
    host.  Without an initialised process group every call is the identity (world 1)."""

    def __init__(self, group=None):
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.staged = self.active and dist.get_backend(group) == "gloo"
        self.capturable = graph_capturable(group)

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        if not self.active:
            return t
        if self.staged and t.is_cuda:
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """(world, *t.shape), rank-major."""
        if not self.active:
            return t.unsqueeze(0).clone()
        src = t.detach().cpu().contiguous() if (self.staged and t.is_cuda) else t.contiguous()
        out = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
        if self.staged:
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self.group)
            out = torch.stack(parts)
        else:
            dist.all_gather_into_tensor(out, src, group=self.group)
        return out.to(t.device)

    def combine_bn_stats_(self, mean: torch.Tensor, m2: torch.Tensor, rows_per_rank: int) -> None:
        """Per-feature (mean, M2 = sum (x - mean)^2) of equal-sized row shards -> the same pair over all rows
        (pairwise update of Chan et al.): mean = avg(mean_r); M2 = sum M2_r + rows * sum (mean_r - mean)^2.  Every
        rank computes the identical result from the identical gathered operands."""
        if self.world == 1:
            return
        n = mean.numel()
        allp = self.all_gather(torch.cat([mean, m2]))
        means, m2s = allp[:, :n], allp[:, n:]
        mu = means.sum(0) / self.world
        mean.copy_(mu)
        m2.copy_(m2s.sum(0) + float(rows_per_rank) * ((means - mu) ** 2).sum(0))


class GradSync:
    """Bucketed gradient all-reduce that overlaps with the rest of the backward: ``start(bucket)`` launches an
    asynchronous sum over a slice of the flat gradient buffer as soon as that slice is final, ``finish()`` makes
    the current stream wait for every bucket before the optimizer kernel reads the buffer."""

    def __init__(self, group=None):
        self.group = group
        self.pending = []
        self.active = dist.is_available() and dist.is_initialized()
        self.staged = self.active and dist.get_backend(group) == "gloo"
        # under capture start() / finish() become a fork onto RCCL's stream and the join before the optimizer kernel
        self.capturable = graph_capturable(group)

    def start(self, bucket: torch.Tensor):
        if not self.active:                         # world 1 without a process group: the sum is the bucket itself
            return
        if self.staged and bucket.is_cuda:          # test configuration: ranks sharing a GPU over gloo
            h = bucket.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            bucket.copy_(h)
            return
        self.pending.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending.clear()

    def reset(self):
        """Forget the handles of collectives issued under a capture that was abandoned (engine._capture_stages' fallback)."""
        self.pending.clear()


class ShardedSync:
    """Data-parallel optimizer step without the full-size flat pass (round 3): per gradient bucket, in the order the backward
    completes them,

        reduce-scatter of the bucket's gradients (sum)  ->  RMSprop on this rank's 1/world shard  ->  all-gather of the
        bucket's updated parameters

    instead of all-reduce of the gradients followed by RMSprop over the whole buffer on every rank.  The bytes on the wire
    are the same (an all-reduce IS a reduce-scatter plus an all-gather), the optimizer pass is world times smaller, every
    element is reduced exactly once (on its owner) and then copied, so the replicas stay bit-identical by construction.
    Each bucket's chain runs on a SIDE stream that forks from the caller's stream when the bucket's gradients are final and
    is joined by ``finish()`` at the end of the step, so an early bucket's chain overlaps the rest of the backward; under
    HIP-graph capture the fork / join become graph edges and the RCCL calls graph nodes, as in GradSync.  Both collectives run
    in place (RCCL's documented in-place forms: recvbuff = sendbuff + rank * count and sendbuff = recvbuff + rank * count),
    so a bucket's length must be a multiple of the world size (engine.FlatParams pads the bucket starts).
    Safe because a bucket's parameters are dead for the rest of the step once its gradients are final: the launch that
    produces a layer's weight gradient is also the last reader of that layer's weights (engine._build_d_rest / _build_g_plan).
    gloo (test rigs, ranks sharing a GPU): the same arithmetic staged through the host on the caller's stream."""

    def __init__(self, lib, group=None):
        self.L = lib
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.staged = self.active and dist.get_backend(group) == "gloo"
        self.capturable = graph_capturable(group)
        self.side = None
        self.forked = False
        self.pending = []           # (interface parity with GradSync: nothing is ever pending on the host)

    def start(self, flat: torch.Tensor, grad: torch.Tensor, sq: torch.Tensor, lo: int, hi: int, lr: float, gscale: float,
              state=(0, 0, 0)):
        """The chain for the bucket [lo, hi) of one network's flat (parameter, gradient, square average) buffers.  ``state``:
        (device pointer, inc0, inc1) of the step's bookkeeping, done by the launch that gets it (the last bucket's)."""
        n = hi - lo
        if n % self.world:
            raise ValueError(f"bucket of {n} elements does not split over {self.world} ranks")
        sh = n // self.world
        a = lo + self.rank * sh
        main = torch.cuda.current_stream(flat.device) if flat.is_cuda else None        # (CPU tensors: host-logic tests)

        def step(stream):
            self.L.rmsprop(flat.data_ptr() + 4 * a, grad.data_ptr() + 4 * a, sq.data_ptr() + 4 * a, sh, lr, 0.99, 1e-8, gscale,
                           state[0], state[1], state[2], stream.cuda_stream if stream is not None else 0)
        if not self.active or self.world == 1 and self.staged:
            step(main)
            return
        if self.staged:
            h = grad[lo:hi].detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            grad[a:a + sh].copy_(h[self.rank * sh:(self.rank + 1) * sh])
            step(main)
            mine = flat[a:a + sh].detach().cpu()
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
            flat[lo:hi].copy_(torch.cat(parts))
            return
        if self.side is None:
            self.side = torch.cuda.Stream(device=flat.device)
        side = self.side
        side.wait_stream(main)                       # fork: everything that produced this bucket's gradients
        self.forked = True
        try:
            with torch.cuda.stream(side):
                dist.reduce_scatter_tensor(grad[a:a + sh], grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
                step(side)
                dist.all_gather_into_tensor(flat[lo:hi], flat[a:a + sh], group=self.group)
        except BaseException:
            # a collective that refuses (e.g. under stream capture): join the side stream before the error travels on, so
            # that a capture in progress can still be ended (an unjoined stream invalidates it) and the eager fallback
            # finds both streams usable
            main.wait_stream(side)
            self.forked = False
            raise

    def finish(self):
        """Join: the caller's stream waits for every bucket's chain (updated parameters of all ranks in place)."""
        if self.forked:
            torch.cuda.current_stream(self.side.device).wait_stream(self.side)
            self.forked = False

    def reset(self):
        """A capture that forked the side stream was abandoned (engine._capture_stages' fallback): forget the fork."""
        self.forked = False


class _Region:
    """A device allocation of the kernel library seen as a flat fp32 torch tensor (``__cuda_array_interface__``); tensors made
    from it keep it alive, and it frees the allocation when the last of them is gone."""

    def __init__(self, lib, ptr: int, nfloats: int):
        self._lib, self.ptr, self.nfloats = lib, int(ptr), int(nfloats)
        self.__cuda_array_interface__ = {"shape": (self.nfloats,), "typestr": "<f4", "data": (self.ptr, False), "version": 2,
                                         "strides": None}

    def __del__(self):
        try:
            self._lib.mesh_free(self.ptr)
        except Exception:        # interpreter shutdown: the process is about to release everything anyway
            pass


_MESHES = []        # weak references to the live MeshSync objects (shutdown() closes them before the process group goes)


class MeshSync:
    """Peer-to-peer gradient exchange (opt-in, WFD_DP_MESH=1; include/wfdgan.h ``wfd_mesh_*``, csrc/mesh.hip): every rank's
    exchange region -- both networks' flat PARAMETER and GRADIENT buffers live in it -- is mapped into every other rank's
    address space (hipIpcMemHandle), and the data-parallel end of a step is ONE kernel launch on the step's own stream:
    start barrier -> every rank sums its 1/world shard of the gradients from all ranks (rank order), applies RMSprop to its
    shard and stores the updated parameters to all ranks -> end barrier.  No host action, no side stream, no separate
    optimizer pass; a plain launch in the plan, so the step stays one HIP graph on any backend.  torch.distributed is used
    once, at set-up, to hand the 64-byte IPC handles around (and for the barriers of ``close()``).
    An MI355X node is a full mesh (a link to each of the 7 peers), which is what this exchange pattern is shaped for; here
    it could only be run with several ranks sharing ONE GPU (tests), so on real links it is unmeasured."""

    ARENA = 1 << 19         # floats at the end of the region for the exact mode's small per-engine vectors (alloc_small)

    def __init__(self, lib, region_floats: int, device, group=None):
        import ctypes
        import weakref
        self._arena_at = int(region_floats)
        region_floats = int(region_floats) + MeshSync.ARENA
        self.L, self.group, self.device = lib, group, torch.device(device)
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        if self.world > 8:
            raise ValueError("the mesh exchange serves one node: at most 8 ranks")
        self.capturable = True
        self.pending = []
        self.mesh = None
        self._opened: List[int] = []
        n = int(region_floats)
        with torch.cuda.device(self.device):
            sharing = self._ranks_on_my_gpu()          # (also refuses ranks on more than one node, before any mapping)
            hd, hf = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
            pd, pf = ctypes.c_void_p(), ctypes.c_void_p()
            lib.mesh_alloc(4 * n, 0, ctypes.addressof(pd), ctypes.addressof(hd))
            self.region = _Region(lib, pd.value, n)
            lib.mesh_alloc(lib.mesh_flag_bytes(), 1, ctypes.addressof(pf), ctypes.addressof(hf))
            self._flags = pf.value
            data, flags = [self.region.ptr], [self._flags]
            if self.world > 1:
                handles = [None] * self.world
                dist.all_gather_object(handles, (hd.raw, hf.raw), group=group)
                data, flags = [], []
                for r, (a, b) in enumerate(handles):
                    if r == self.rank:
                        data.append(self.region.ptr)
                        flags.append(self._flags)
                        continue
                    for h, lst in ((a, data), (b, flags)):
                        q, hb = ctypes.c_void_p(), ctypes.create_string_buffer(h, 64)
                        lib.mesh_open(ctypes.addressof(hb), ctypes.addressof(q))
                        self._opened.append(q.value)
                        lst.append(q.value)
            arr_d = (ctypes.c_void_p * self.world)(*data)
            arr_f = (ctypes.c_void_p * self.world)(*flags)
            m = ctypes.c_void_p()
            lib.mesh_create(self.rank, self.world, arr_d, arr_f, n, ctypes.addressof(m))
            self.mesh = m.value
            self.base = torch.as_tensor(self.region, device=self.device)
            # every workgroup of an exchange waits for the peers, so all workgroups of all ranks on one GPU must be resident
            # together: ranks that share a GPU (test rigs) split the library's default of 128 workgroups
            blocks = int(os.environ.get("WFD_MESH_BLOCKS", "0")) or 128 // sharing
            lib.mesh_set_max_blocks(self.mesh, max(1, min(256, blocks)))
            # bound of every in-kernel wait for a peer (a rank busy on the host -- rank 0 writing a checkpoint -- keeps the
            # others waiting inside their next exchange; callers put a host barrier behind such work, see wait_for_peers)
            lib.mesh_set_timeout_ms(self.mesh, int(1000 * float(os.environ.get("WFD_MESH_TIMEOUT_S", "20"))))
            self.sharing = sharing
        if self.base.data_ptr() != self.region.ptr or self.base.numel() != n:
            raise RuntimeError("exchange region: torch did not wrap the allocation in place")
        _MESHES.append(weakref.ref(self))

    def _ranks_on_my_gpu(self) -> int:
        if self.world == 1:
            return 1
        import socket
        pr = torch.cuda.get_device_properties(self.device)
        ident = getattr(pr, "uuid", None) or getattr(pr, "pci_bus_id", None)
        mine = (socket.gethostname(), str(ident)) if ident is not None else None
        allids = [None] * self.world
        dist.all_gather_object(allids, mine, group=self.group)
        if any(x is None for x in allids):
            return self.world             # the GPU cannot be identified: assume the worst
        if len({x[0] for x in allids}) > 1:
            raise ValueError("the mesh exchange maps the peers' memory through hipIpcMemHandle: all ranks must be on ONE "
                             f"node, found {sorted({x[0] for x in allids})}")
        return sum(1 for x in allids if x == mine)

    def view(self, off: int, n: int) -> torch.Tensor:
        return self.base[off: off + n]

    def alloc_small(self, n: int) -> torch.Tensor:
        """n zeroed floats of the region's arena (256-byte aligned).  Every rank must allocate in the same order: the
        exchanges address peers' vectors by offset."""
        a = self._arena_at
        self._arena_at = a + (n + 63) // 64 * 64
        if self._arena_at > self.region.nfloats:
            raise RuntimeError("exchange region: arena of small vectors exhausted")
        return self.base[a: a + n]

    def offset_of(self, t: torch.Tensor) -> int:
        d = t.data_ptr() - self.region.ptr
        if d < 0 or d % 16 or d // 4 + t.numel() > self.region.nfloats:
            raise ValueError("tensor is not a 16-byte aligned part of the exchange region")
        return d // 4

    # (interface parity with GradSync / ShardedSync: nothing is ever pending on the host)
    def finish(self):
        pass

    def reset(self):
        pass

    def status(self):
        """(exchanges completed, waits that timed out); synchronises the device."""
        import ctypes
        out = (ctypes.c_long * 3)()
        self.L.mesh_status(self.mesh, out)
        return int(out[0]), int(out[2])

    def check(self):
        """Raise if a wait for a peer ever timed out (the result of that exchange is undefined).  Synchronises the device:
        call it where the host synchronises anyway (epoch ends)."""
        done, timeouts = self.status()
        if timeouts:
            raise RuntimeError(f"mesh exchange: {timeouts} waits for a peer timed out ({done} exchanges issued on rank "
                               f"{self.rank}); a rank died or stayed away longer than WFD_MESH_TIMEOUT_S")

    def wait_for_peers(self):
        """Host barrier: call after work that only some ranks do (rank 0 writing files), so that no rank waits for it INSIDE
        an exchange kernel (whose waits are bounded)."""
        if self.active and self.world > 1:
            dist.barrier(self.group)

    def close(self):
        """Collective: every rank unmaps its peers' regions, then drops its own mesh.  The region itself is freed when the
        last tensor made from it is gone.  Call after the engines (and their graphs) are gone: parallel.shutdown() does."""
        if self.mesh is None:
            return
        torch.cuda.synchronize(self.device)
        if self.active:
            dist.barrier(self.group)          # nobody is inside an exchange any more
        with torch.cuda.device(self.device):
            for q in self._opened:
                self.L.mesh_close(q)
            self._opened = []
            if self.active:
                dist.barrier(self.group)      # every mapping of this rank's memory is gone
            self.L.mesh_destroy(self.mesh)
            self.L.mesh_free(self._flags)
        self.mesh = None
        self.base = None


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) shard of n_rows for this rank; the first n_rows % world ranks get one extra row."""
    base, rem = divmod(n_rows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device) -> float:
    """Timing convention of bench.py: the slowest rank defines the step time."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_state_(tensors: List[torch.Tensor], src: int = 0, group=None) -> None:
    """Make replicas identical at start-up (parameters, optimizer state, BN buffers)."""
    if dist.is_available() and dist.is_initialized():
        staged = dist.get_backend(group) == "gloo"
        for t in tensors:
            if staged and t.is_cuda:
                h = t.detach().cpu()
                dist.broadcast(h, src=src, group=group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=group)

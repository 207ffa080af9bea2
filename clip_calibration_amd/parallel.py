"""Multi-GPU layer of the hot path: one process per GPU, weights and text features replicated and
resident, the image batch sharded on dim 0, ONE all-gather of the per-GPU L2-normalised image embeddings per step
(``torch.distributed`` backend "nccl" = RCCL over xGMI) before the shared ``img @ txt^T`` kernel.  Gather, not reduce:
results are bitwise independent of the number of ranks.  ECE accumulators (3*(n_bins+1) float64) merge with one
all-reduce at the end of an evaluation, not per step.

The reference has no counterpart (single process + nn.DataParallel re-broadcasting the weights every call,
coop.py:268-272, tempscaling.py:117-120)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) of a global batch of n that rank owns: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class EmbeddingExchange:
    """The exchange step of the path as the library owns it: ``clipmi_allgather`` = RCCL ``ncclAllGather`` on the caller's
    stream (include/clipmi.h, "Multi-GPU exchange"), communicator built from a unique id that rank 0 creates and
    ``torch.distributed`` only carries to the other ranks.  ``backend``:

    * ``"rccl"``  -- the C-ABI communicator (default whenever every rank has its own GPU);
    * ``"torch"`` -- ``torch.distributed.all_gather_into_tensor`` on the process group (gloo in the CPU tests and in
      same-GPU debug runs: RCCL refuses two ranks on one device);
    * ``"auto"``  -- rccl, falling back to torch with a printed reason if the communicator cannot be built.

    Gather, not reduce: the result is bitwise independent of the number of ranks."""

    def __init__(self, device: torch.device, group=None, backend: str = "auto"):
        from . import _lib
        self.group, self.device = group, torch.device(device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._comm = None
        self.backend = "torch"
        if backend in ("rccl", "auto") and self.device.type == "cuda":
            # Every rank must take the same branch: a rank that raised before ncclCommInitRank would leave the others waiting in it,
            # and one that fell back alone would wait in a different collective.  So rank 0 broadcasts an EMPTY id when it cannot
            # make one, and the ranks agree (min over a CPU flag) on whether every communicator came up.
            reason, handle = "", None
            uid = [b""]
            if self.rank == 0:
                try:
                    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
                    _lib.check(_lib.lib.clipmi_comm_unique_id(buf), "clipmi_comm_unique_id")
                    uid = [buf.raw]
                except (_lib.ClipmiError, RuntimeError, AttributeError, OSError) as e:
                    reason = str(e)
            # the id and the agreement flag travel on CPU tensors when the group has a CPU backend (gloo, or "cpu:gloo,cuda:nccl"),
            # on tensors of this rank's GPU when it is a plain "nccl" group
            cpu_ok = self.world == 1 or "gloo" in str(dist.get_backend(group)).lower() or "cpu" in str(dist.get_backend(group)).lower()
            side = torch.device("cpu") if cpu_ok else self.device
            if self.world > 1:
                src = dist.get_global_rank(group, 0) if group is not None else 0
                msg = torch.zeros(_lib.COMM_ID_BYTES + 1, dtype=torch.uint8, device=side)
                if self.rank == 0 and uid[0]:
                    msg[0] = 1
                    msg[1:] = torch.frombuffer(bytearray(uid[0]), dtype=torch.uint8).to(side)
                dist.broadcast(msg, src=src, group=group)
                msg = msg.cpu()
                uid = [bytes(msg[1:].numpy().tobytes()) if int(msg[0]) == 1 else b""]
            if uid[0]:
                try:
                    handle = C.c_void_p()
                    with torch.cuda.device(self.device):
                        _lib.check(_lib.lib.clipmi_comm_create(uid[0], self.world, self.rank, C.byref(handle)), "clipmi_comm_create")
                except (_lib.ClipmiError, RuntimeError, AttributeError, OSError) as e:
                    reason, handle = str(e), None
            else:
                reason = reason or "rank 0 could not create an RCCL unique id"
            ok = handle is not None
            if self.world > 1:
                flag = torch.tensor([int(ok)], dtype=torch.int32, device=side)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if ok and int(flag.item()) == 0:
                    reason = "another rank could not build its communicator"
                    _lib.lib.clipmi_comm_destroy(handle)
                    ok = False
            if ok:
                self._comm, self.backend = handle, "rccl"
            elif backend == "rccl":
                raise RuntimeError(f"RCCL communicator unavailable: {reason}")
            else:
                print(f"[clip_calibration_amd] RCCL communicator unavailable ({reason}); using torch.distributed", flush=True)

    @property
    def rccl_ranks(self) -> Optional[int]:
        """World size as the RCCL communicator reports it (None on the torch backend)."""
        if self._comm is None:
            return None
        from . import _lib
        w, r = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib.clipmi_comm_ranks(self._comm, C.byref(w), C.byref(r)), "clipmi_comm_ranks")
        return int(w.value)

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        """[b, E] per rank -> [world*b, E] on every rank, rank-major (equal b on every rank)."""
        local = local.contiguous()
        if self._comm is None:
            return all_gather_embeddings(local, self.group)
        from . import _lib
        out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        with torch.cuda.device(local.device):
            _lib.check(_lib.lib.clipmi_allgather(self._comm, local.data_ptr(), out.data_ptr(), local.numel() * local.element_size(),
                                                 torch.cuda.current_stream().cuda_stream), "clipmi_allgather")
        return out

    def close(self):
        if self._comm is not None:
            from . import _lib
            _lib.lib.clipmi_comm_destroy(self._comm)
            self._comm = None


def all_gather_embeddings(local: torch.Tensor, group=None) -> torch.Tensor:
    """[b, E] per rank -> [world*b, E] on every rank, rank-major (equal b on every rank)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(out, local, group=group)
    except (RuntimeError, NotImplementedError):   # backends without the flat form (older gloo)
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, local, group=group)
    return out


def all_gather_ragged(local: torch.Tensor, n_global: int, group=None) -> torch.Tensor:
    """Ragged variant for a global batch that does not divide evenly: pads each shard to the largest shard, gathers,
    and strips the padding so that row i of the result is global row i."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_global, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    gathered = all_gather_embeddings(pad, group)
    return torch.cat([gathered[r * width: r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def merge_ece_bins(bins: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank (count, sum_conf, sum_correct) accumulators; exact for the counts, fp64 sums otherwise."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(bins, op=dist.ReduceOp.SUM, group=group)
    return bins


def all_gather_varlen(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate per-rank tensors whose dim-0 lengths differ (rank-major): lengths are exchanged first, shards padded to
    the longest, one all-gather, padding stripped."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(x.item()) for x in lens]
    width = max(lens)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    gathered = all_gather_embeddings(pad, group)
    return torch.cat([gathered[r * width: r * width + lens[r]] for r in range(world)], dim=0)


def gather_samples(evaluator, proximity=None, group=None, has_proximity: Optional[bool] = None):
    """End-of-evaluation merge for a sharded test split: sums the bin accumulators and replaces the evaluator's kept
    (conf, pred, gt) vectors by the all-rank concatenation, so that ``evaluate`` returns identical numbers on every rank.
    Returns the gathered proximity vector (or None).

    Every rank enters every collective, also a rank whose shard was EMPTY (more ranks than batches): the decision to
    gather samples hangs on ``keep_samples`` and the one for proximity on ``has_proximity`` (rank-uniform configuration,
    default: ``proximity is not None`` -- pass it explicitly when a rank may hold an empty shard), never on local data."""
    merge_ece_bins(evaluator.bins, group)
    dev = evaluator.bins.device
    if evaluator.keep_samples:
        def cat(parts, dtype):   # one dtype on every rank, whatever a rank happened to collect (or not collect)
            return torch.cat(parts).to(dtype) if parts else torch.zeros(0, dtype=dtype, device=dev)
        evaluator._conf = [all_gather_varlen(cat(evaluator._conf, torch.float32), group)]
        evaluator._pred = [all_gather_varlen(cat(evaluator._pred, torch.int64), group)]
        evaluator._gt = [all_gather_varlen(cat(evaluator._gt, torch.int64), group)]
    if has_proximity is None:
        has_proximity = proximity is not None
    if not has_proximity:
        return None
    if proximity is None:
        proximity = torch.zeros(0, dtype=torch.float32, device=dev)
    return all_gather_varlen(proximity.to(torch.float32), group)

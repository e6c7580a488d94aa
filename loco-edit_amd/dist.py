"""Probe sharding across the GPUs of one node.

One process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
ROCm; ``gloo`` in the CPU tests).  The only data-path collective of the hot path
is the all-gather of the ``A = J^T J V`` row shards once per solver iteration:
k=64 probes on 8 GPUs is 6.3 MB per rank -- far below the compute time of the
two U-Net passes it follows, so a single flat all-gather is used (no ring
pipelining, no bucketing).

Shards are contiguous and as even as possible (the first ``k % world`` ranks own
one probe more), so any probe count runs on any world size -- the reference's
defaults ``pca_rank=50`` / ``pca_rank_null=10`` included; ranks beyond ``k`` own no
probe and only take part in the gather.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(k: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of ``rank`` when k probes are dealt contiguously over ``world`` ranks."""
    base, extra = divmod(k, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ProbeSharder:
    def __init__(self, group="world"):
        """``group=None`` -> single process (no collective).  ``"world"`` uses the
        default process group when torch.distributed is initialised."""
        if group == "world":
            self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            self.group = None
        else:
            self.active = group is not None
            self.group = group
        self.world = dist.get_world_size(self.group) if self.active else 1
        self.rank = dist.get_rank(self.group) if self.active else 0

    @property
    def is_main(self) -> bool:
        return self.rank == 0

    def rows(self, k: int) -> Tuple[int, int]:
        """Contiguous row block [lo, hi) of this rank (the gathered block keeps the probe order, so the result
        is the 1-GPU layout)."""
        if not self.active:
            return 0, k
        return shard_bounds(k, self.world, self.rank)

    def all_gather_rows(self, local: torch.Tensor, k: int) -> torch.Tensor:
        if not self.active:
            return local
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo takes device tensors but moves them at a crawl (measured 4 s per 8 MB all-gather with two ranks on one GPU,
            # 0.6 s for the whole 12-iteration solve beside it): stage through host memory explicitly.  RCCL (the production
            # backend) gathers device to device below.
            return self._gather(local.cpu(), k).to(local.device)
        return self._gather(local, k)

    def _gather(self, local: torch.Tensor, k: int) -> torch.Tensor:
        out = torch.empty((k,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        if k % self.world == 0:
            # views of one contiguous buffer: works on nccl (RCCL) and gloo alike
            dist.all_gather(list(out.chunk(self.world, dim=0)), local.contiguous(), group=self.group)
            return out
        # uneven shards: every rank contributes a block padded to the largest shard, trimmed on arrival
        per = (k + self.world - 1) // self.world
        pad = torch.zeros((per,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        pad[:local.shape[0]] = local
        buf = torch.empty((self.world * per,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        dist.all_gather(list(buf.chunk(self.world, dim=0)), pad, group=self.group)
        for r in range(self.world):
            lo, hi = shard_bounds(k, self.world, r)
            out[lo:hi] = buf[r * per:r * per + (hi - lo)]
        return out

    def barrier(self):
        if self.active:
            dist.barrier(group=self.group)

    def agree(self, value):
        """Rank 0's ``value`` on every rank (file-existence decisions, drawn seeds): ranks must take the same
        branch or they hang in the next all-gather."""
        if not self.active:
            return value
        box = [value]
        dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]


def init_from_env(device_arg: Optional[str] = None) -> Tuple[int, int, Optional[str]]:
    """One process per GPU under ``torchrun`` / ``torch.distributed.run``: when WORLD_SIZE > 1, bind this rank to
    ``cuda:LOCAL_RANK`` and create the default process group (RCCL; ``LOCO_DIST_BACKEND=gloo`` lets several ranks
    share one GPU on a 1-GPU box).  Must run before any other GPU call.  -> (rank, world, device string or None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, device_arg
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    backend = os.environ.get("LOCO_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = device_arg
    if device_arg is None or str(device_arg).startswith("cuda"):
        ndev = torch.cuda.device_count()
        if backend != "nccl" and ndev > 0:
            local = local % ndev
        device = f"cuda:{local}"
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, device


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()

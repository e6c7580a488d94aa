"""Probe sharding across the GPUs of one node.

One process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
ROCm; ``gloo`` in the CPU tests).  The only data-path collective of the hot path
is the all-gather of the ``A = J^T J V`` row shards once per solver iteration:
k=64 probes on 8 GPUs is 6.3 MB per rank -- far below the compute time of the
two U-Net passes it follows, so a single flat all-gather is used (no ring
pipelining, no bucketing).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


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

    def rows(self, k: int) -> Tuple[int, int]:
        """Contiguous row block [lo, hi) of this rank; k must divide evenly so the
        gathered block keeps the probe order (bitwise equal to the 1-GPU layout)."""
        if not self.active:
            return 0, k
        if k % self.world != 0:
            raise ValueError(f"probe count {k} must be a multiple of the world size {self.world}")
        per = k // self.world
        return self.rank * per, (self.rank + 1) * per

    def all_gather_rows(self, local: torch.Tensor, k: int) -> torch.Tensor:
        if not self.active:
            return local
        out = torch.empty((k,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        # views of one contiguous buffer: works on nccl (RCCL) and gloo alike
        dist.all_gather(list(out.chunk(self.world, dim=0)), local.contiguous(), group=self.group)
        return out

"""View-parallel multi-GPU harness: every rank holds a full replica of the
Gaussians, camera views are dealt round-robin (view v -> rank v mod world), each rank accumulates
parameter gradients locally over its views, and ONE sum all-reduce of the flattened gradient
buffer per optimizer step crosses xGMI (torch.distributed backend "nccl" = RCCL on ROCm; "gloo" in
the CPU tests).  Render-only work needs no collective at all.

The reference's own mechanism is generic DDP (nerfstudio/scripts/train.py:139-145,
nerfstudio/pipelines/base_pipeline.py:303-305), which SURVEY §5 shows is not functional for the
splatting model; this is the explicit equivalent."""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world_size: int) -> List[int]:
    """Indices of the views rank `rank` renders (round-robin)."""
    return list(range(rank, num_views, world_size))


class GradBucket:
    """One flat fp32 buffer aliasing the .grad of every parameter, so the per-step reduction is a
    single large collective (xGMI is point-to-point: fewer, larger messages)."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = list(params)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)   # autograd accumulates in place
            off += n

    def zero_(self) -> None:
        self.flat.zero_()

    def all_reduce(self) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


def train_step(render_and_backward: Callable[[int], None], bucket: GradBucket, view_ids: Sequence[int],
               reduce: bool = True) -> None:
    """One optimizer step's worth of rasterizer work on this rank: fwd+bwd of its views with local
    gradient accumulation, then the gradient-to-Gaussian reduction."""
    bucket.zero_()
    for v in view_ids:
        render_and_backward(v)
    if reduce:
        bucket.all_reduce()

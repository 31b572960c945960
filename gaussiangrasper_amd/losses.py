"""Cosine-similarity loss of the feature field: host-side mirror of the reference's
`cosine_similarity_loss(embeddings1, embeddings2)` (nerfstudio/models/gaussian_splatting.py:113-118):

    e1 = F.normalize(embeddings1, dim=0); e2 = F.normalize(embeddings2, dim=0)
    loss = 1 - (e1 * e2).sum(dim=0).mean()

with embeddings of shape (C, M) — the reference builds them as `.permute(1, 0)` of (M, C) gathers of the
rendered feature image (:909-918), so the (M, C) layout the kernels want is the storage they already
have.  Forward and backward are one launch each (`gg_cosine_loss_fwd/bwd`, csrc/losses.hip) instead of
~10 elementwise / reduction launches each way.  No CPU path."""
from __future__ import annotations

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib
from .ops import _f32, _ptr, _require_hip, _stream


class _CosineLoss(Function):
    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor) -> Tensor:     # a, b: (M, C)
        dev = _require_hip(a, b)
        if a.shape != b.shape or a.dim() != 2:
            raise ValueError(f"embeddings must have the same 2-D shape, got {tuple(a.shape)} and {tuple(b.shape)}")
        a2, b2 = _f32(a), _f32(b)
        m, c = a2.shape
        sim = torch.empty(m, dtype=torch.float32, device=dev)
        na, nb = torch.empty_like(sim), torch.empty_like(sim)
        total = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().gg_cosine_loss_fwd(m, c, _ptr(a2), _ptr(b2), _ptr(sim), _ptr(na), _ptr(nb),
                                                  _ptr(total), _stream(dev)), "gg_cosine_loss_fwd")
        ctx.save_for_backward(a2, b2, sim, na, nb)
        return 1.0 - total[0] / max(m, 1)

    @staticmethod
    def backward(ctx, v_loss: Tensor):
        a2, b2, sim, na, nb = ctx.saved_tensors
        dev = a2.device
        m, c = a2.shape
        v_a, v_b = torch.empty_like(a2), torch.empty_like(b2)
        vl = _f32(v_loss).reshape(1)
        _lib.check(_lib.load().gg_cosine_loss_bwd(m, c, _ptr(a2), _ptr(b2), _ptr(sim), _ptr(na), _ptr(nb),
                                                  _ptr(vl), _ptr(v_a), _ptr(v_b), _stream(dev)),
                   "gg_cosine_loss_bwd")
        return v_a, v_b


def cosine_similarity_loss(embeddings1: Tensor, embeddings2: Tensor) -> Tensor:
    """Drop-in for the reference function: embeddings (C, M), normalised along dim 0."""
    return _CosineLoss.apply(embeddings1.permute(1, 0), embeddings2.permute(1, 0))

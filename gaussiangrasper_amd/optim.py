"""Fused Adam for the Gaussian parameter groups.

The reference builds one `torch.optim.Adam` per parameter group through
`AdamOptimizerConfig.setup(params=...)` (nerfstudio/engine/optimizers.py:45-58,81-110; groups and
hyper-parameters nerfstudio/configs/method_configs.py:618-660: xyz 1.6e-4, color / feature 5e-4,
opacity 0.05, scaling 0.005, rotation 0.001, all eps 1e-15) and steps them one after the other
(`optimizer_step_all`, :158-171) — 6 optimizers x ~10 elementwise kernels over 472 B per Gaussian.

`FusedAdam` is a drop-in `torch.optim.Optimizer` with torch.optim.Adam's constructor, `param_groups`
and per-parameter state (`step`, `exp_avg`, `exp_avg_sq`), so the reference's optimizer surgery
(`remove_from_optim` / `dup_in_optim`, gaussian_splatting.py:333-371) and its checkpoints keep working on
it; `AdamOptimizerConfig(_target=FusedAdam)` is the only change a maintainer makes.  `step()` is ONE
kernel launch per optimizer, `fused_step([...])` ONE launch for all groups (C ABI `gg_adam_step`).
No CPU path: parameters must live on the HIP device."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Sequence

import torch

from . import _lib
from .ops import _require_hip, _stream


def _entries(opt: "FusedAdam"):
    """(param, group) for every parameter that has a gradient; state created lazily as torch does."""
    out = []
    for group in opt.param_groups:
        if group.get("amsgrad") or group.get("maximize"):
            raise NotImplementedError("FusedAdam covers torch.optim.Adam with amsgrad=False, maximize=False "
                                      "(what the reference configures)")
        for p in group["params"]:
            if p.grad is None:
                continue
            if p.grad.is_sparse:
                raise RuntimeError("FusedAdam does not support sparse gradients")
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                raise TypeError("FusedAdam handles fp32 parameters (the reference trains in fp32)")
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            out.append((p, group))
    return out


def _launch(entries, zero_grad: bool) -> None:
    if not entries:
        return
    lib = _lib.load()
    dev = _require_hip(*[p for p, _, _ in entries])
    for start in range(0, len(entries), _lib.ADAM_MAX_GROUPS):
        chunk = entries[start:start + _lib.ADAM_MAX_GROUPS]
        arr = (_lib.AdamGroup * len(chunk))()
        keep = []
        for k, (p, group, st) in enumerate(chunk):
            step = int(st["step"].item() if torch.is_tensor(st["step"]) else st["step"]) + 1
            if torch.is_tensor(st["step"]):
                st["step"] += 1
            else:
                st["step"] = step
            g = p.grad
            if not (p.is_contiguous() and g.is_contiguous() and st["exp_avg"].is_contiguous()
                    and st["exp_avg_sq"].is_contiguous()):
                raise RuntimeError("FusedAdam needs contiguous parameters, gradients and moments")
            keep.append((p, g))
            b1, b2 = group["betas"]
            lr = group["lr"]
            arr[k] = _lib.AdamGroup(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                    st["exp_avg_sq"].data_ptr(), p.numel(),
                                    float(lr.item() if torch.is_tensor(lr) else lr), float(b1), float(b2),
                                    float(group["eps"]), float(group["weight_decay"]), step)
        _lib.check(lib.gg_adam_step(len(chunk), arr, int(zero_grad), _stream(dev)), "gg_adam_step")


@torch.no_grad()
def adam_pieces(entries) -> None:
    """Adam on raw flat pieces — (param, grad, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, step) each — in one
    launch per 8 pieces (dist.ShardedAdamStep: a rank's shard of the flat parameter buffer)."""
    if not entries:
        return
    lib = _lib.load()
    dev = _require_hip(*[e[0] for e in entries])
    for start in range(0, len(entries), _lib.ADAM_MAX_GROUPS):
        chunk = entries[start:start + _lib.ADAM_MAX_GROUPS]
        arr = (_lib.AdamGroup * len(chunk))()
        for k, (p, g, m, v, lr, betas, eps, wd, step) in enumerate(chunk):
            if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
                raise RuntimeError("adam_pieces needs contiguous pieces")
            arr[k] = _lib.AdamGroup(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr),
                                    float(betas[0]), float(betas[1]), float(eps), float(wd), int(step))
        _lib.check(lib.gg_adam_step(len(chunk), arr, 0, _stream(dev)), "gg_adam_step")


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (amsgrad off) as one streaming HIP kernel per step."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, amsgrad: bool = False, *, maximize: bool = False, **_ignored):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameter")
        if amsgrad or maximize:
            raise NotImplementedError("amsgrad / maximize are not used by the reference and not built")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      amsgrad=False, maximize=False))

    @torch.no_grad()
    def step(self, closure=None, zero_grad: bool = False):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _launch([(p, g, self.state[p]) for p, g in _entries(self)], zero_grad)
        return loss


@torch.no_grad()
def fused_step(optimizers: Sequence[FusedAdam], zero_grad: bool = False) -> None:
    """`Optimizers.optimizer_step_all` (engine/optimizers.py:158-171) for FusedAdam instances: every
    parameter group of every optimizer in ONE launch (up to 8 arrays per launch)."""
    entries: List = []
    for opt in optimizers:
        if not isinstance(opt, FusedAdam):
            raise TypeError("fused_step takes FusedAdam optimizers")
        entries += [(p, g, opt.state[p]) for p, g in _entries(opt)]
    _launch(entries, zero_grad)

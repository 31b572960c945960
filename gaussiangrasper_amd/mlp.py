"""Feature up-projection MLP: host-side mirror of the reference's `MLP` module
(nerfstudio/models/gaussian_splatting.py:198-213; instantiated as
`self.fea_up = MLP(self.feature_dim, self.clip_dim, hidden_list=[128])` at :258, applied to every
pixel of the rendered feature image by render.sh at nerfstudio/pipelines/base_pipeline.py:408 and to
1000 sampled pixels per training step at :917).

Same constructor, same sub-module layout (`layers = Sequential(Linear, ReLU, Linear)`, so reference
checkpoints' `fea_up.layers.{0,2}.{weight,bias}` load unchanged) and same call.  The forward is ONE
fused fp32 MFMA kernel (csrc/mlp.hip) behind `gg_mlp_fwd`; there is no CPU path.  The backward at the
reference's training size (1000 sampled pixels, :917) is ONE launch of `gg_mlp_bwd` (csrc/losses.hip);
above `NATIVE_BWD_MAX_ROWS` rows the weight gradients are plain GEMMs and go to the library.  A first layer of
128 inputs (BASELINE config 5) has its own forward kernel (W1 in LDS, the hidden tile in registers, W2 streamed)."""
from __future__ import annotations

from typing import Sequence

import torch
from torch import Tensor, nn
from torch.autograd import Function

from . import _lib
from .ops import _f32, _ptr, _require_hip, _stream

HIDDEN = 128
SUPPORTED_IN = (8, 16, 32, 64, 128)     # fused forward kernels (W1 in registers; 128: W1 in LDS, W2 slices streamed)
SUPPORTED_IN_BWD = (8, 16, 32, 64, 128)
NATIVE_BWD_MAX_ROWS = 1 << 16
FAST_MAX_OUT = 3968        # gg_mlp_fwd_fast: two 64 KB weight slices + the biases share the CU's 160 KB of LDS
FAST_IN = (32, 64, 128)     # gg_mlp_fwd_fast: fp16 two-piece operands on the 16x-rate matrix instruction (fp32-grade)
# True: always the exact-order kernels (bit-identical to oracle.mlp_fwd's summation order; 4x the matrix cycles)
EXACT_ORDER = False


class _MLPForward(Function):
    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
        dev = _require_hip(x, w1, b1, w2, b2)
        in_dim, out_dim = w1.shape[1], w2.shape[0]
        if tuple(w1.shape) != (HIDDEN, in_dim) or tuple(b1.shape) != (HIDDEN,) or \
                tuple(w2.shape) != (out_dim, HIDDEN) or tuple(b2.shape) != (out_dim,):
            raise ValueError("expected w1 (128, in), b1 (128,), w2 (out, 128), b2 (out,)")
        if x.shape[-1] != in_dim:
            raise ValueError(f"x has {x.shape[-1]} features, the first layer takes {in_dim}")
        x2 = _f32(x).reshape(-1, in_dim)
        w1c, b1c, w2c, b2c = _f32(w1), _f32(b1), _f32(w2), _f32(b2)
        if in_dim in FAST_IN and out_dim % 16 == 0 and out_dim <= FAST_MAX_OUT and not EXACT_ORDER:
            lib = _lib.load()
            y = torch.empty(x2.shape[0], out_dim, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.gg_mlp_fwd_fast_workspace(in_dim, HIDDEN, out_dim), dtype=torch.uint8, device=dev)
            _lib.check(lib.gg_mlp_fwd_fast(x2.shape[0], in_dim, HIDDEN, out_dim, _ptr(x2), _ptr(w1c), _ptr(b1c),
                                           _ptr(w2c), _ptr(b2c), _ptr(y), _ptr(ws), ws.numel(), _stream(dev)),
                       "gg_mlp_fwd_fast")
        elif in_dim in SUPPORTED_IN and out_dim % 32 == 0:
            y = torch.empty(x2.shape[0], out_dim, dtype=torch.float32, device=dev)
            _lib.check(_lib.load().gg_mlp_fwd(x2.shape[0], in_dim, HIDDEN, out_dim, _ptr(x2), _ptr(w1c),
                                              _ptr(b1c), _ptr(w2c), _ptr(b2c), _ptr(y), _stream(dev)),
                       "gg_mlp_fwd")
        else:   # plain GEMMs: the library's job (hipBLASLt)
            y = torch.addmm(b2c, torch.relu(torch.addmm(b1c, x2, w1c.t())), w2c.t())
        ctx.save_for_backward(x2, w1c, b1c, w2c)
        ctx.x_shape = x.shape
        return y.reshape(x.shape[:-1] + (out_dim,))

    @staticmethod
    def backward(ctx, v_y: Tensor):
        x2, w1, b1, w2 = ctx.saved_tensors
        g = _f32(v_y).reshape(-1, w2.shape[0])
        rows, in_dim, out_dim = x2.shape[0], w1.shape[1], w2.shape[0]
        if rows <= NATIVE_BWD_MAX_ROWS and in_dim in SUPPORTED_IN_BWD and out_dim <= 1024:
            dev = x2.device
            v_x, v_w1, v_b1 = torch.empty_like(x2), torch.empty_like(w1), torch.empty_like(b1)
            v_w2 = torch.empty_like(w2)
            v_b2 = torch.empty(out_dim, dtype=torch.float32, device=dev)
            _lib.check(_lib.load().gg_mlp_bwd(rows, in_dim, HIDDEN, out_dim, _ptr(x2), _ptr(w1), _ptr(b1),
                                              _ptr(w2), _ptr(g), _ptr(v_x), _ptr(v_w1), _ptr(v_b1), _ptr(v_w2),
                                              _ptr(v_b2), _stream(dev)), "gg_mlp_bwd")
            return v_x.reshape(ctx.x_shape), v_w1, v_b1, v_w2, v_b2
        h_pre = torch.addmm(b1, x2, w1.t())
        g_h = (g @ w2) * (h_pre > 0)
        v_w2 = g.t() @ torch.relu(h_pre)
        return ((g_h @ w1).reshape(ctx.x_shape), g_h.t() @ x2, g_h.sum(0), v_w2, g.sum(0))


def mlp_forward(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """relu(x @ w1.T + b1) @ w2.T + b2 on the matrix pipe; x (..., in) -> (..., out)."""
    return _MLPForward.apply(x, w1, b1, w2, b2)


class MLP(nn.Module):
    """Drop-in for the reference's `MLP(in_dim=8, out_dim=512, hidden_list=[128])`."""

    def __init__(self, in_dim: int = 8, out_dim: int = 512, hidden_list: Sequence[int] = (128,)):
        super().__init__()
        hidden_list = list(hidden_list)
        if hidden_list != [HIDDEN]:
            raise NotImplementedError(
                "the kernels cover the reference's fea_up shape family: one hidden layer of 128 "
                f"(got hidden={hidden_list})")
        self.layers = nn.Sequential(nn.Linear(in_dim, HIDDEN), nn.ReLU(), nn.Linear(HIDDEN, out_dim))

    def forward(self, x: Tensor) -> Tensor:
        l0, l2 = self.layers[0], self.layers[2]
        return mlp_forward(x, l0.weight, l0.bias, l2.weight, l2.bias)

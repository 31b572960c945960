"""Densify / cull / split with optimizer-state surgery: host-side mirror of the
reference model's refinement callbacks, on the C ABI of csrc/densify.hip.

    after_train        nerfstudio/models/gaussian_splatting.py:373-393   -> Refiner.after_train
    refinement_after   :402-473                                           -> Refiner.refinement_after
    cull_gaussians     :480-502 + remove_from_optim :333-350              -> cull()
    split / dup        :504-546, torch.cat :434-443 + dup_in_optim :352-371 -> densify()

The reference rebuilds the six parameter tensors with boolean indexing / torch.cat and then patches
`exp_avg` / `exp_avg_sq` of six optimizers one by one (18 tensors, dozens of launches and several
`.item()` syncs).  Here cull is ONE launch over all 18 arrays (mask -> decoupled look-back prefix sum
-> row gather) and densify two mask scans plus ONE append launch; the new parameters replace the old
ones in their optimizers exactly as the reference's helpers do (same `param_groups` / `state`
surgery), so torch.optim.Adam and `optim.FusedAdam` both work.  A `GradBucket` (dist.py) is
re-aliased after every change of N.  No CPU path."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor
from torch.nn import Parameter

from . import _lib
from .ops import _ptr, _require_hip, _stream, _workspace

# group name -> attribute, as GaussianSplattingModel.get_gaussian_param_groups (:562-571)
GROUPS = {"xyz": "means", "color": "colors_all", "opacity": "opacities", "scaling": "scales",
          "rotation": "quats", "feature": "feature"}
_KIND = {"means": _lib.ROWS_MEANS, "scales": _lib.ROWS_SCALES}
SIZE_FAC = 1.6      # split_gaussians :523


@dataclass
class RefineConfig:
    """The fields of GaussianSplattingModelConfig the refinement reads (:150-196), same defaults."""
    warmup_length: int = 500
    refine_every: int = 100
    cull_alpha_thresh: float = 0.1
    cull_scale_thresh: float = 0.5
    reset_alpha_every: int = 30
    densify_grad_thresh: float = 0.0002
    densify_size_thresh: float = 0.01
    n_split_samples: int = 2
    cull_screen_size: float = 0.15
    split_screen_size: float = 0.05
    stop_screen_size_at: int = 4000
    stop_split_at: int = 15000


def _u8(mask: Tensor) -> Tensor:
    return mask.reshape(-1).to(torch.uint8).contiguous()


def _rows(t: Tensor) -> Tuple[Tensor, int]:
    t = t.detach()
    if t.dtype != torch.float32:
        raise TypeError("row arrays must be fp32")
    t = t.contiguous()
    return t, (t.numel() // t.shape[0] if t.shape[0] else 1)


def mask_scan(mask: Tensor, invert: bool = False) -> Tuple[Tensor, int]:
    """ranks (N,) int32 of the selected rows and their count (one launch + one read-back; the
    reference's `.sum().item()` at :510,:538)."""
    dev = _require_hip(mask)
    lib = _lib.load()
    m = _u8(mask)
    n = m.numel()
    ranks = torch.empty(n, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    ws = _workspace(lib.gg_rows_workspace(n), dev)
    _lib.check(lib.gg_mask_scan(n, _ptr(m), int(invert), _ptr(ranks), _ptr(total), _ptr(ws), ws.numel(),
                                _stream(dev)), "gg_mask_scan")
    return ranks, int(total.item())


def compact(arrays: Sequence[Tensor], deleted_mask: Tensor) -> List[Tensor]:
    """[a[~deleted_mask] for a in arrays] in one launch (<= 24 arrays per launch)."""
    if not arrays:
        return []
    dev = _require_hip(deleted_mask, *arrays)
    lib = _lib.load()
    m = _u8(deleted_mask)
    n = m.numel()
    srcs = []
    for a in arrays:
        if a.shape[0] != n:
            raise ValueError(f"array with {a.shape[0]} rows, mask with {n}")
        srcs.append(_rows(a))
    outs = [torch.empty_like(s) for s, _ in srcs]
    kept = torch.empty(1, dtype=torch.int64, device=dev)
    ws = _workspace(lib.gg_rows_workspace(n), dev)
    for start in range(0, len(srcs), _lib.MAX_ROW_ARRAYS):
        chunk = list(range(start, min(start + _lib.MAX_ROW_ARRAYS, len(srcs))))
        desc = (_lib.RowArray * len(chunk))()
        for k, i in enumerate(chunk):
            desc[k] = _lib.RowArray(srcs[i][0].data_ptr(), outs[i].data_ptr(), srcs[i][1], 0)
        _lib.check(lib.gg_compact_rows(n, _ptr(m), len(chunk), desc, _ptr(kept), _ptr(ws), ws.numel(),
                                       _stream(dev)), "gg_compact_rows")
    k = int(kept.item())
    if not 0 <= k <= n:         # impossible total: a look-back of the scan gave up (csrc/scan.h)
        raise _lib.GGError("gg_compact_rows: the prefix scan's look-back timed out; the compacted rows are not valid")
    return [o[:k] for o in outs]


def append_rows(arrays: Sequence[Tuple[Tensor, int]], split_mask: Tensor, dup_mask: Tensor, nsamps: int,
                samples: Optional[Tensor], means: Tensor, scales: Tensor, quats: Tensor,
                generator: Optional[torch.Generator] = None) -> Tuple[List[Tensor], int, int, Tensor]:
    """torch.cat([a, split rows, dup rows]) for every (array, kind) in one launch.
    -> (new arrays, n_split, n_dup, the N(0,1) samples used)."""
    dev = _require_hip(split_mask, dup_mask, means, scales, quats, *[a for a, _ in arrays])
    lib = _lib.load()
    sm, dm = _u8(split_mask), _u8(dup_mask)
    n = sm.numel()
    s_rank, n_split = mask_scan(sm)
    d_rank, n_dup = mask_scan(dm)
    if samples is None:   # `torch.randn((samps * n_splits, 3), device=self.device)`, :508
        samples = torch.randn((nsamps * n_split, 3), device=dev, generator=generator)
    samples = samples.to(torch.float32).contiguous()
    if samples.shape != (nsamps * n_split, 3):
        raise ValueError(f"samples must be ({nsamps * n_split}, 3), got {tuple(samples.shape)}")
    total = n + nsamps * n_split + n_dup
    srcs = [(*_rows(a), kind) for a, kind in arrays]
    outs = [torch.empty((total,) + tuple(s.shape[1:]), dtype=torch.float32, device=dev) for s, _, _ in srcs]
    mc, sc, qc = _rows(means)[0], _rows(scales)[0], _rows(quats)[0]
    for start in range(0, len(srcs), _lib.MAX_ROW_ARRAYS):
        chunk = list(range(start, min(start + _lib.MAX_ROW_ARRAYS, len(srcs))))
        desc = (_lib.RowArray * len(chunk))()
        for k, i in enumerate(chunk):
            desc[k] = _lib.RowArray(srcs[i][0].data_ptr(), outs[i].data_ptr(), srcs[i][1], srcs[i][2])
        _lib.check(lib.gg_densify_rows(n, _ptr(sm), _ptr(dm), _ptr(s_rank), _ptr(d_rank), n_split, n_dup,
                                       nsamps, _ptr(samples), SIZE_FAC, _ptr(mc), _ptr(sc), _ptr(qc),
                                       len(chunk), desc, _stream(dev)), "gg_densify_rows")
    return outs, n_split, n_dup, samples


# ------------------------------------------------------------------------------------------------
# optimizer surgery (same effect as remove_from_optim / dup_in_optim, for every group at once)
# ------------------------------------------------------------------------------------------------
def _opt_state(optimizer) -> Tuple[Parameter, dict]:
    param = optimizer.param_groups[0]["params"][0]
    return param, optimizer.state.get(param, {})


def _swap_param(optimizer, old: Parameter, new: Parameter, exp_avg: Optional[Tensor],
                exp_avg_sq: Optional[Tensor]) -> None:
    state = optimizer.state.pop(old, None)
    optimizer.param_groups[0]["params"] = [new]
    if state:
        state["exp_avg"], state["exp_avg_sq"] = exp_avg, exp_avg_sq
        optimizer.state[new] = state


class Refiner:
    """Holds the Gaussians' Parameters under the reference's attribute names, their per-group
    optimizers, the running statistics of `after_train`, and applies `refinement_after`."""

    def __init__(self, params: Dict[str, Tensor], optimizers: Dict[str, torch.optim.Optimizer],
                 config: Optional[RefineConfig] = None, num_train_data: int = 0, bucket=None):
        for attr in GROUPS.values():
            if attr not in params:
                raise KeyError(f"missing parameter '{attr}'")
        self.params: Dict[str, Parameter] = {k: v if isinstance(v, Parameter) else Parameter(v)
                                             for k, v in params.items()}
        self.optimizers = optimizers
        self.config = config or RefineConfig()
        self.num_train_data = num_train_data
        self.bucket = bucket
        self.step = 0
        self.xys_grad_norm: Optional[Tensor] = None
        self.vis_counts: Optional[Tensor] = None
        self.max_2Dsize: Optional[Tensor] = None
        self.last_split_samples: Optional[Tensor] = None

    @property
    def num_points(self) -> int:
        return self.params["means"].shape[0]

    # -- after_train (:373-393) ------------------------------------------------------------------
    @torch.no_grad()
    def after_train(self, xys_grad: Tensor, radii: Tensor, last_size: Tuple[int, int]) -> None:
        dev = _require_hip(xys_grad, radii)
        n = xys_grad.shape[0]
        first = self.xys_grad_norm is None        # the three accumulators are reset together (:471-473)
        if first:
            self.xys_grad_norm = torch.empty(n, dtype=torch.float32, device=dev)
            self.vis_counts = torch.empty(n, dtype=torch.float32, device=dev)
            self.max_2Dsize = torch.empty(n, dtype=torch.float32, device=dev)
        g = xys_grad.detach().to(torch.float32).contiguous()
        r = radii.detach().to(torch.int32).contiguous().reshape(-1)
        _lib.check(_lib.load().gg_densify_stats(n, _ptr(g), _ptr(r), int(max(last_size)), int(first),
                                                _ptr(self.xys_grad_norm), _ptr(self.vis_counts),
                                                _ptr(self.max_2Dsize), _stream(dev)), "gg_densify_stats")
        self.last_size = tuple(last_size)

    # -- masks -------------------------------------------------------------------------------------
    def densify_masks(self) -> Tuple[Tensor, Tensor]:
        p, c = self.params, self.config
        dev = p["means"].device
        n = self.num_points
        split = torch.empty(n, dtype=torch.uint8, device=dev)
        dup = torch.empty(n, dtype=torch.uint8, device=dev)
        use_screen = int(self.step < c.stop_screen_size_at)
        sc = _rows(p["scales"])[0]
        _lib.check(_lib.load().gg_densify_masks(
            n, _ptr(self.xys_grad_norm), _ptr(self.vis_counts), _ptr(self.max_2Dsize), _ptr(sc),
            int(max(self.last_size)), c.densify_grad_thresh, c.densify_size_thresh, c.split_screen_size,
            use_screen, SIZE_FAC, _ptr(split), _ptr(dup), _stream(dev)), "gg_densify_masks")
        return split, dup

    def cull_mask(self) -> Tensor:
        p, c = self.params, self.config
        dev = p["means"].device
        n = self.num_points
        mask = torch.empty(n, dtype=torch.uint8, device=dev)
        use_scale = int(self.step > c.refine_every * c.reset_alpha_every)
        use_screen = int(self.step < c.stop_screen_size_at)
        op, sc = _rows(p["opacities"])[0], _rows(p["scales"])[0]
        _lib.check(_lib.load().gg_cull_mask(
            n, _ptr(op), _ptr(sc), _ptr(self.max_2Dsize), c.cull_alpha_thresh, c.cull_scale_thresh,
            c.cull_screen_size, use_scale, use_screen, _ptr(mask), _stream(dev)), "gg_cull_mask")
        return mask

    # -- row surgery on the 6 parameters + 12 moments --------------------------------------------
    def _gather_arrays(self):
        names, arrays = [], []
        for group, attr in GROUPS.items():
            names.append((attr, None))
            arrays.append(self.params[attr])
            opt = self.optimizers.get(group)
            if opt is not None:
                _, st = _opt_state(opt)
                for key in ("exp_avg", "exp_avg_sq"):
                    if key in st:
                        names.append((attr, key))
                        arrays.append(st[key])
        return names, arrays

    def _install(self, names, new_arrays) -> None:
        new_params = {attr: Parameter(a) for (attr, key), a in zip(names, new_arrays) if key is None}
        moments: Dict[str, Dict[str, Tensor]] = {}
        for (attr, key), a in zip(names, new_arrays):
            if key is not None:
                moments.setdefault(attr, {})[key] = a
        for group, attr in GROUPS.items():
            opt = self.optimizers.get(group)
            if opt is not None:
                mom = moments.get(attr, {})
                _swap_param(opt, self.params[attr], new_params[attr], mom.get("exp_avg"), mom.get("exp_avg_sq"))
            self.params[attr] = new_params[attr]
        if self.bucket is not None:
            self.bucket.rebind([self.params[a] for a in ("means", "scales", "quats", "opacities",
                                                         "colors_all", "feature")])

    @torch.no_grad()
    def cull(self, deleted_mask: Tensor) -> int:
        """cull_gaussians :497-502 + remove_from_optim :333-350 for every group: one launch."""
        names, arrays = self._gather_arrays()
        before = self.num_points
        self._install(names, compact(arrays, deleted_mask))
        return before - self.num_points

    @torch.no_grad()
    def densify(self, split_mask: Tensor, dup_mask: Tensor, samples: Optional[Tensor] = None,
                generator: Optional[torch.Generator] = None) -> Tuple[int, int]:
        """split_gaussians + dup_gaussians + torch.cat (:424-451) + dup_in_optim for every group."""
        names, arrays = self._gather_arrays()
        kinds = [(_lib.ROWS_ZERO_NEW if key is not None else _KIND.get(attr, _lib.ROWS_COPY))
                 for attr, key in names]
        p = self.params
        outs, n_split, n_dup, used = append_rows(list(zip(arrays, kinds)), split_mask, dup_mask,
                                                 self.config.n_split_samples, samples, p["means"],
                                                 p["scales"], p["quats"], generator)
        self.last_split_samples = used
        self._install(names, outs)
        if self.max_2Dsize is not None:   # :441 append zeros
            extra = self.num_points - self.max_2Dsize.shape[0]
            self.max_2Dsize = torch.cat([self.max_2Dsize, self.max_2Dsize.new_zeros(extra)])
        return n_split, n_dup

    # -- refinement_after (:402-473) -------------------------------------------------------------
    @torch.no_grad()
    def refinement_after(self, step: int, samples: Optional[Tensor] = None,
                         generator: Optional[torch.Generator] = None) -> Dict[str, int]:
        self.step = step
        c = self.config
        info = {"split": 0, "dup": 0, "culled": 0, "opacity_reset": 0}
        if self.step < c.warmup_length:
            return info
        reset_interval = c.reset_alpha_every * c.refine_every
        window = self.step % reset_interval > self.num_train_data + c.refine_every
        if self.step < c.stop_split_at and window:
            if self.xys_grad_norm is None:
                raise RuntimeError("refinement_after before any after_train")
            split, dup = self.densify_masks()
            info["split"], info["dup"] = self.densify(split, dup, samples, generator)
        if window:
            info["culled"] = self.cull(self.cull_mask())
        if self.step % reset_interval == c.refine_every:     # opacity reset (:459-470)
            reset_value = c.cull_alpha_thresh * 0.8
            self.params["opacities"].data.fill_(torch.logit(torch.tensor(reset_value)).item())
            opt = self.optimizers.get("opacity")
            if opt is not None:
                _, st = _opt_state(opt)
                if "exp_avg" in st:
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
            info["opacity_reset"] = 1
        self.xys_grad_norm = None
        self.vis_counts = None
        self.max_2Dsize = None
        return info

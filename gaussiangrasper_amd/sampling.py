"""Pixel sampling for the feature losses, on the mask's device.

The reference's helpers (nerfstudio/models/gaussian_splatting.py:120-148) draw `torch.randperm(count)[:m]` per label on
the HOST generator, with `count` the label's pixel count: nine permutations of ~480 000 elements per 1600x1200 view —
32 ms each on the GPU box's cores (tools/train_step_profile.py: 295 ms of a 300 ms training iteration whose
render + losses + backward take 5 ms).  The two functions below return samples of the same law — per label, `m` DISTINCT
pixels, uniform over the label's pixels, both members of a pair drawn independently — from the device generator: one
stable sort of the mask, one host read of the label counts, one radix select of uniform keys per draw.  They are NOT the same
draws as the reference's for a given seed (another generator), which is why the plugin uses them only when asked to
(plugin.make_fused_model_class(device_sampling=True) / GG_DEVICE_SAMPLING=1); the default keeps the reference's helpers.
Return types and shapes are the reference's: (M, 2) long rows of (row, col)."""
from typing import List

import torch


_last_runs = None      # (key, mask kept alive, result): get_loss_dict asks for pairs and for points on the SAME mask


def _label_runs(mask: torch.Tensor):
    """labels > -1 (ascending), and per label the flat indices of its pixels in row-major order (`torch.where` order)"""
    global _last_runs
    key = (mask.data_ptr(), mask._version, tuple(mask.shape), mask.dtype)
    if _last_runs is not None and _last_runs[0] == key:
        return _last_runs[2]
    out = _label_runs_of(mask.detach())
    _last_runs = (key, mask, out)
    return out


def _label_runs_of(mask: torch.Tensor):
    flat = mask.reshape(-1)
    labels, counts = torch.unique(flat, return_counts=True)
    order = torch.argsort(flat, stable=True)
    labels_h, counts_h = labels.tolist(), counts.tolist()          # the one host read
    runs, start = [], 0
    for lab, cnt in zip(labels_h, counts_h):
        if lab > -1:
            runs.append(order[start:start + cnt])
        start += cnt
    return len(labels_h), runs, mask.shape[1]


def _draw(run: torch.Tensor, m: int, width: int) -> torch.Tensor:
    # m distinct positions, every m-subset equally likely, in random order — the law of randperm(count)[:m] — as the
    # positions of the m largest of `count` uniform keys: a radix select instead of a full sort of the label's pixels
    # (device randperm of ~480 000 elements: 185 us, nine per view = half of a training iteration's GPU time)
    count = run.shape[0]
    if m >= count:
        pick = torch.randperm(count, device=run.device)
    else:
        pick = torch.rand(count, device=run.device).topk(m).indices        # (sorted by key: a random order)
    flat = run[pick]
    return torch.stack((torch.div(flat, width, rounding_mode="floor"), flat % width), dim=1)


def sampling_in_mask(mask: torch.Tensor, sample_num: int) -> torch.Tensor:
    """:120-132 — up to sample_num // (number of distinct mask values - 1) distinct pixels of every label > -1"""
    num_values, runs, width = _label_runs(mask)
    per_label = sample_num // (num_values - 1)
    return torch.cat([_draw(r, min(per_label, r.shape[0]), width) for r in runs])


def sampling_pairs_in_mask(mask: torch.Tensor, sample_num: int) -> List[List[torch.Tensor]]:
    """:134-148 — per label > -1 two independent draws of up to sample_num distinct pixels"""
    _, runs, width = _label_runs(mask)
    out = []
    for r in runs:
        m = min(sample_num, r.shape[0])
        out.append([_draw(r, m, width), _draw(r, m, width)])
    return out

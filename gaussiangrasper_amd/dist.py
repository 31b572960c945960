"""View-parallel multi-GPU harness: every rank holds a full replica of the
Gaussians, camera views are dealt round-robin (view v -> rank v mod world), each rank accumulates
parameter gradients locally over its views, and the gradient-to-Gaussian reduction crosses xGMI once
per optimizer step (torch.distributed backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
Render-only work needs no collective at all.

The reduction is issued PER PARAMETER, asynchronously, from a post-accumulate-grad hook that is armed
for the last view of the step: as soon as autograd has added the last view's gradient of a parameter
(the 128 MB feature gradient is complete two rasterize backwards before the step ends) its slice of
the flat bucket goes out on RCCL's own stream while the remaining backward kernels run; the step ends
by waiting for the handles.  Six messages of 4-300 MB — still large ones, xGMI is point-to-point and
per-link bound, so nothing is gained by cutting them smaller.

The reference's own mechanism is generic DDP (nerfstudio/scripts/train.py:139-152,205-210,
nerfstudio/pipelines/base_pipeline.py:303-305), which SURVEY §5 shows is not functional for the
splatting model (parameters are replaced every 100 steps); this is the explicit equivalent, and
`GradBucket.rebind` is the hook densification uses after it has replaced the parameters."""
from __future__ import annotations

import os
import time
import socket
import subprocess
import sys
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world_size: int) -> List[int]:
    """Indices of the views rank `rank` renders (round-robin)."""
    return list(range(rank, num_views, world_size))


def _dist_on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class GradBucket:
    """One flat fp32 buffer aliasing the .grad of every parameter: autograd accumulates in place, the
    optimizer reads slices, and the per-step reduction touches one allocation."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self._hooks: list = []
        self._armed = False
        self._work: list = []
        self.rebind(params)

    def rebind(self, params: Sequence[torch.Tensor]) -> None:
        """(Re)alias the gradients — call again after densification replaced the parameter tensors
        (reference gaussian_splatting.py:434-439,495-500 build new nn.Parameters of a new N)."""
        for h in self._hooks:
            h.remove()
        direct = getattr(self, "_direct_ops", None)
        if direct is not None:
            direct.clear_grad_sinks()
        self.params = list(params)
        # every slice starts on a 256-byte boundary (vector loads of the fused Adam kernel, whole cache
        # lines per parameter in the collectives); the padding stays zero
        pad = lambda n: (n + 63) // 64 * 64
        self.payload = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(sum(pad(p.numel()) for p in self.params), dtype=torch.float32, device=dev)
        self.slices = []
        off = 0
        for p in self.params:
            n = p.numel()
            sl = self.flat[off:off + n]
            p.grad = sl.view_as(p)   # autograd accumulates in place
            self.slices.append(sl)
            off += pad(n)
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                       for i, p in enumerate(self.params) if p.requires_grad and p.is_leaf]
        self._armed = False
        self._work = []
        self._exposed = getattr(self, "_exposed", [])
        self._order = None     # parameter indices in the order autograd completes them (agreed by all ranks)
        self._fired: List[int] = []
        self._ready: List[bool] = []
        self._next = 0
        if direct is not None:
            self.enable_direct(direct)

    def enable_direct(self, ops, params: Optional[Sequence[torch.Tensor]] = None, defer_sh: bool = False) -> None:
        """Let the operators' backward kernels add into this bucket directly (ops.register_grad_sink)
        for the given parameters (default: all) — no per-view gradient tensor and no separate add for
        the parameters that enter an operator as leaves (SH coefficients, features).  The overlapped
        reduction is driven by the operators' notification instead of autograd's hook for those."""
        index = {id(p): i for i, p in enumerate(self.params)}
        for p in (self.params if params is None else params):
            i = index[id(p)]
            if defer_sh:
                # the SH gradient of a view may stay in factored form until the step's last view (arm()) or until
                # finish() / all_reduce() flush it; other operators ignore the hint
                ops.register_grad_sink(p, self.slices[i].view_as(p), self._make_direct_done(i),
                                       defer=lambda: not self._armed)
            else:
                ops.register_grad_sink(p, self.slices[i].view_as(p), self._make_direct_done(i))
        self._direct_ops = ops

    def _make_direct_done(self, i: int):
        def done(param: torch.Tensor) -> None:
            if not self._armed or self._ready[i]:
                return
            self._fired.append(i)
            self._ready[i] = True
            self._issue_ready()
        return done

    def _make_hook(self, i: int):
        def hook(param: torch.Tensor) -> None:
            if not self._armed:
                return
            if param.grad is None or param.grad.data_ptr() != self.slices[i].data_ptr():
                raise RuntimeError("GradBucket: .grad of parameter %d no longer aliases the bucket "
                                   "(call rebind() after replacing parameters)" % i)
            if not self._ready[i]:
                self._fired.append(i)
                self._ready[i] = True
                self._issue_ready()
        return hook

    def _issue_ready(self) -> None:
        # collectives must be issued in the same order on every rank: walk the agreed order and
        # send the ready prefix
        if self._order is None or not _dist_on():
            return
        while self._next < len(self._order) and self._ready[self._order[self._next]]:
            i = self._order[self._next]
            self._work.append(dist.all_reduce(self.slices[i], op=dist.ReduceOp.SUM, async_op=True))
            self._next += 1

    def flush(self) -> None:
        """Make every gradient the operators still hold in factored form (the SH gradient of the step's views,
        `enable_direct(defer_sh=True)`) part of the bucket.  finish() / all_reduce() do it themselves; a caller
        that steps its optimizer WITHOUT a reduction (train_step(reduce=False), one process) must call it before
        reading `.grad` — otherwise those contributions are silently missing and zero_() discards them."""
        ops = getattr(self, "_direct_ops", None)
        if ops is not None and hasattr(ops, "flush_grad_sinks"):
            ops.flush_grad_sinks()

    _flush_deferred = flush

    def zero_(self) -> None:
        ops = getattr(self, "_direct_ops", None)
        if ops is not None and hasattr(ops, "discard_deferred_grads"):
            ops.discard_deferred_grads()     # kept-but-not-expanded contributions belong to what is zeroed here
        self.flat.zero_()

    def arm(self) -> None:
        """The next backward is the last of the step: reduce each parameter's slice as it completes."""
        self._armed = True
        self._work = []
        self._fired = []
        self._ready = [False] * len(self.params)
        self._next = 0

    def finish(self) -> None:
        """Wait for the armed reductions.  Slices whose hook did not fire (no gradient reached the
        parameter in the last view) are reduced here, so every rank always reduces every slice, in the
        agreed order.  The order is agreed once: the first armed step only records the order in which
        autograd completed the parameters, the ranks compare it, and from then on it is fixed."""
        if not self._armed:      # a rank without views this step still takes part in every collective
            self.arm()
        self._flush_deferred()   # nothing to do when the last view's backward already expanded the kept views
        self._armed = False
        if not _dist_on():
            return
        n = len(self.params)
        if self._order is None:
            seen = self._fired + [i for i in range(n) if i not in self._fired]
            mine = torch.tensor(seen, dtype=torch.int64, device=self.flat.device)
            ref = mine.clone()
            dist.broadcast(ref, src=0)
            same = torch.tensor([int(torch.equal(ref, mine))], dtype=torch.int64, device=self.flat.device)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            self._order = seen if int(same.item()) == 1 else list(range(n))
            self._ready = [True] * n
        else:
            self._ready = [True] * n
        self._issue_ready()
        # how long does the step wait for its collectives BEHIND its last kernel?  On a GPU: two events on the current
        # stream around the waits (wait() makes the stream wait for the collective's stream: the gap between the events
        # is the time the stream had nothing to run but the hand-over); on the CPU (gloo): wall time of the waits.
        cuda = self.flat.is_cuda
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        else:
            t0 = time.perf_counter()
        for w in self._work:
            w.wait()
        if cuda:
            e1.record()
            self._exposed.append((e0, e1))
        else:
            self._exposed.append(1e3 * (time.perf_counter() - t0))
        self._collectives = len(self._work)
        self._work = []

    def comm_stats(self, reset: bool = True) -> dict:
        """Per-step communication figures of the armed reductions since the last call (bench.py --gpus N > 1):
        exposed_ms = time finish() waited for the collectives after the step's last kernel (mean / max over the steps),
        bytes = gradient bytes reduced per step, collectives = messages per step."""
        vals = []
        for x in self._exposed:
            if isinstance(x, tuple):
                x[1].synchronize()
                vals.append(float(x[0].elapsed_time(x[1])))
            else:
                vals.append(float(x))
        if reset:
            self._exposed = []
        return {"steps": len(vals), "exposed_ms_mean": (sum(vals) / len(vals)) if vals else None,
                "exposed_ms_max": max(vals) if vals else None, "bytes_per_step": self.nbytes,
                "collectives_per_step": getattr(self, "_collectives", None)}

    def all_reduce(self) -> None:
        """Unoverlapped form: one collective over the whole bucket."""
        self._flush_deferred()
        if _dist_on():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    @property
    def nbytes(self) -> int:
        """gradient bytes reduced per step (without the alignment padding)"""
        return self.payload * 4

    def gathered(self) -> torch.Tensor:
        """the gradients back to back, without padding (tests, checkpoints)"""
        return torch.cat([sl.reshape(-1) for sl in self.slices])


# ------------------------------------------------------------------------------------------------
# reduce-scatter + Adam on a shard + all-gather (the second reduction scheme; `bench.py --reduce rs_ag`)
# ------------------------------------------------------------------------------------------------
def torch_adam_piece(p, g, m, v, lr, betas, eps, weight_decay, step) -> None:
    """torch.optim.Adam's update (amsgrad off) on one flat piece, with torch ops: the CPU / gloo tests' stand-in
    for gg_adam_step (same expressions as torch/optim/adam.py _single_tensor_adam)."""
    b1, b2 = betas
    if weight_decay != 0:
        g = g.add(p, alpha=weight_decay)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


class ShardedAdamStep:
    """Optimizer step of the replicated Gaussians with the work and the traffic of ONE replica spread over the ranks:

        reduce-scatter   rank r receives the sum over ranks of elements [r S, (r + 1) S) of the gradient bucket
        Adam             on that shard only: parameters, gradients and BOTH moments of 1 / world of the elements
                         (the moments of the other shards do not exist on this rank: 2 x 472 MB -> 2 x 59 MB at 8)
        all-gather       every rank receives every shard of the updated parameters

    instead of all-reduce + a full Adam step on every rank.  The bytes on xGMI are the same (a ring all-reduce IS a
    reduce-scatter followed by an all-gather); what is saved is 7 / 8 of the optimizer's HBM traffic per rank
    (0.61 ms -> 0.08 ms at 1 M Gaussians, profiles/r02_adam_bench.json) and 7 / 8 of the moment memory, and the
    all-gather half moves PARAMETERS, which are needed only by the next step's first projection — a window the
    all-reduce of gradients does not have.
    The parameters are re-homed into one flat buffer with the bucket's layout (`p.data` becomes a view), so a shard
    is one contiguous range of both; hyper-parameters stay per parameter (the reference's six groups:
    method_configs.py:618-660), a shard that spans several parameters is stepped piece by piece in ONE launch.
    gloo has no reduce-scatter: there the gradient is all-reduced and the shard cut out (tests only)."""

    def __init__(self, bucket: GradBucket, hyper: Sequence[dict], adam_piece=None):
        """hyper: one dict per bucket parameter (lr, betas, eps, weight_decay); adam_piece(p, g, m, v, lr, betas,
        eps, weight_decay, step): the update of one flat piece (default: gg_adam_step through optim.adam_pieces)."""
        self.bucket = bucket
        self.world = dist.get_world_size() if _dist_on() else 1
        self.rank = dist.get_rank() if _dist_on() else 0
        assert len(hyper) == len(bucket.params)
        self.hyper = [dict(lr=h["lr"], betas=tuple(h.get("betas", (0.9, 0.999))), eps=h.get("eps", 1e-8),
                           weight_decay=h.get("weight_decay", 0.0)) for h in hyper]
        self.adam_piece = adam_piece
        self.step_count = 0
        # flat parameters in the bucket's layout, padded so that every rank's shard has the same length
        L = bucket.flat.numel()
        unit = 64 * self.world
        self.padded = (L + unit - 1) // unit * unit
        self.shard = self.padded // self.world
        dev = bucket.flat.device
        self.flat_params = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        # the reduce-scatter's source: the bucket's flat gradient itself when its length divides among the ranks,
        # otherwise ONE padded buffer allocated here (its tail stays zero; the gradients are copied in per step)
        self.padded_grads = None if self.padded == L else torch.zeros(self.padded, dtype=torch.float32, device=dev)
        # what this object was built over: densification / culling replaces the Parameters and re-binds the bucket
        # (GradBucket.rebind) — stepping the stale views would update nothing the model still uses
        self._bound = (bucket.flat.data_ptr(), L, tuple(id(p) for p in bucket.params))
        self.offsets = []
        for p, sl in zip(bucket.params, bucket.slices):
            off = sl.storage_offset() - bucket.flat.storage_offset()
            self.offsets.append(off)
            view = self.flat_params[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        lo, hi = self.rank * self.shard, (self.rank + 1) * self.shard
        self.pieces = []      # (parameter index, start, stop) of this rank's shard, in flat coordinates
        for i, (p, off) in enumerate(zip(bucket.params, self.offsets)):
            a, b = max(lo, off), min(hi, off + p.numel())
            if a < b:
                self.pieces.append((i, a, b))
        self.exp_avg = [torch.zeros(b - a, dtype=torch.float32, device=dev) for _, a, b in self.pieces]
        self.exp_avg_sq = [torch.zeros(b - a, dtype=torch.float32, device=dev) for _, a, b in self.pieces]
        self.grad_shard = torch.zeros(self.shard, dtype=torch.float32, device=dev)

    def step(self) -> None:
        """after the step's backward passes: reduce-scatter, Adam on the shard, all-gather"""
        b = self.bucket
        if (b.flat.data_ptr(), b.flat.numel(), tuple(id(p) for p in b.params)) != self._bound:
            raise RuntimeError("ShardedAdamStep: the gradient bucket was re-bound (densification or culling replaced the "
                               "Parameters) after this object was built; build a new ShardedAdamStep over the new "
                               "bucket (its moments start at zero for new rows, as the reference's do)")
        b.flush()
        lo = self.rank * self.shard
        if self.world > 1:
            src = b.flat
            if self.padded_grads is not None:
                src = self.padded_grads
                src[:b.flat.numel()].copy_(b.flat)
            if dist.get_backend() == "gloo":
                dist.all_reduce(src, op=dist.ReduceOp.SUM)
                self.grad_shard.copy_(src[lo:lo + self.shard])
            else:
                dist.reduce_scatter_tensor(self.grad_shard, src, op=dist.ReduceOp.SUM)
        else:
            self.grad_shard[:b.flat.numel()].copy_(b.flat) if self.padded != b.flat.numel() else \
                self.grad_shard.copy_(b.flat)
        self.step_count += 1
        entries = []
        for k, (i, a, z) in enumerate(self.pieces):
            h = self.hyper[i]
            entries.append((self.flat_params[a:z], self.grad_shard[a - lo:z - lo], self.exp_avg[k], self.exp_avg_sq[k],
                            h["lr"], h["betas"], h["eps"], h["weight_decay"], self.step_count))
        if self.adam_piece is not None:
            with torch.no_grad():
                for e in entries:
                    self.adam_piece(*e)
        else:
            from .optim import adam_pieces
            adam_pieces(entries)
        if self.world > 1:
            mine = self.flat_params[lo:lo + self.shard]
            if dist.get_backend() == "gloo":
                outs = [self.flat_params[r * self.shard:(r + 1) * self.shard] for r in range(self.world)]
                dist.all_gather(outs, mine.clone())
            else:
                dist.all_gather_into_tensor(self.flat_params, mine)


def train_step(render_and_backward: Callable[[int], None], bucket: GradBucket, view_ids: Sequence[int],
               reduce: bool = True, overlap: bool = True) -> None:
    """One optimizer step's worth of rasterizer work on this rank: fwd+bwd of its views with local
    gradient accumulation, then the gradient-to-Gaussian reduction (overlapped with the last view's
    backward when `overlap`)."""
    bucket.zero_()
    view_ids = list(view_ids)
    for k, v in enumerate(view_ids):
        if reduce and overlap and k == len(view_ids) - 1:
            bucket.arm()
        render_and_backward(v)
    if reduce:
        if overlap:
            bucket.finish()
        else:
            bucket.all_reduce()
    else:
        bucket.flush()      # no reduction: the gradients must still be complete when the caller reads them


def train_step_pipelined(render: Callable[[int], object], backward: Callable[[object], None], bucket: GradBucket,
                         view_ids: Sequence[int], streams, reduce: bool = True, overlap: bool = True) -> None:
    """train_step with the views of the step software-pipelined over two (or more) HIP streams: view k runs
    forward AND backward on stream k mod S (autograd runs a backward node on its forward's stream), the forward
    chains are ordered among themselves by events and so are the backward chains — so the only overlap is
    backward(k) beside forward(k + 1).  The views of a step are independent until the optimizer step; what the
    overlap buys is the second stream's kernels filling the first one's tails and launch gaps (a view is ~45
    kernels, half of them a few microseconds long, and every blend kernel ends with ~20 % of its wave slots
    empty).  Forward kernels never write a gradient buffer and the backward chains are serialised, so every
    accumulation into the bucket happens one view at a time, as in train_step."""
    cur = torch.cuda.current_stream()
    bucket.zero_()
    for s in streams:
        s.wait_stream(cur)
    ev_f = ev_b = None
    view_ids = list(view_ids)
    for k, v in enumerate(view_ids):
        s = streams[k % len(streams)]
        if reduce and overlap and k == len(view_ids) - 1:
            bucket.arm()
        with torch.cuda.stream(s):
            if ev_f is not None:
                s.wait_event(ev_f)          # one forward chain at a time (they share the binning's scratch order)
            out = render(v)
            ev_f = torch.cuda.Event()
            ev_f.record(s)
            if ev_b is not None:
                s.wait_event(ev_b)          # one backward chain at a time: gradient accumulation
            backward(out)
            del out
            ev_b = torch.cuda.Event()
            ev_b.record(s)
    for s in streams:
        cur.wait_stream(s)
    if reduce:
        if overlap:
            bucket.finish()
        else:
            bucket.all_reduce()
    else:
        bucket.flush()


# ------------------------------------------------------------------------------------------------
# rank launcher: N fresh processes, one per GPU, started BEFORE the parent touches the GPU
# ------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(argv: Sequence[str], world_size: int, timeout: Optional[float] = None,
                extra_env: Optional[dict] = None) -> int:
    """Start `world_size` children `python argv...` with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT set (the contract `torch.distributed.run` gives its workers;
    reference: nerfstudio/scripts/train.py:205-210 uses mp.spawn for the same purpose) and wait for
    them.  The caller must not have initialised HIP: the children are fresh interpreters, nothing is
    re-exec'd.  Returns 0 if every rank exited 0, otherwise the first non-zero code (the remaining
    ranks are terminated).  Children inherit stdout/stderr, so rank 0's JSON line is the parent's."""
    port = str(free_port())
    procs = []
    for r in range(world_size):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world_size),
                   LOCAL_WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, *argv], env=env))
    rc = 0
    try:
        pending = list(procs)
        import time
        t_end = None if timeout is None else time.monotonic() + timeout
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (t_end is not None and time.monotonic() > t_end):
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc

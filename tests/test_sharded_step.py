"""dist.ShardedAdamStep (`bench.py --reduce rs_ag`): reduce-scatter of the gradient bucket, Adam on this rank's shard,
all-gather of the parameters — two gloo ranks on the CPU against ONE process running torch.optim.Adam on the summed
gradients (the all-reduce scheme's result).  The reference's counterpart is DDP's all-reduce followed by a full
optimizer step on every rank (nerfstudio/pipelines/base_pipeline.py:303-305, engine/optimizers.py:158-171)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SHAPES = [(301, 3), (301, 3), (301, 4), (301, 1), (301, 25, 3), (301, 8)]      # the six Gaussian parameters
HYPER = [dict(lr=1.6e-4, eps=1e-15), dict(lr=5e-3, eps=1e-15), dict(lr=1e-3, eps=1e-15), dict(lr=5e-2, eps=1e-15),
         dict(lr=5e-4, eps=1e-15), dict(lr=5e-4, eps=1e-15, weight_decay=0.01)]
STEPS = 4


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g).requires_grad_(True) for s in SHAPES]


def _grads(step, rank):
    g = torch.Generator().manual_seed(1000 + 17 * step + rank)
    return [torch.randn(*s, generator=g) for s in SHAPES]


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussiangrasper_amd.dist import GradBucket, ShardedAdamStep, torch_adam_piece
        params = _params()
        bucket = GradBucket(params)
        stepper = ShardedAdamStep(bucket, HYPER, adam_piece=torch_adam_piece)
        assert stepper.shard * world == stepper.padded >= bucket.flat.numel()
        assert sum(z - a for _, a, z in stepper.pieces) <= stepper.shard
        for step in range(STEPS):
            bucket.zero_()
            for p, g in zip(params, _grads(step, rank)):
                p.grad.add_(g)                      # what the rank's views accumulated
            stepper.step()
        # every rank holds every parameter, and they are views of the flat buffer
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        if rank == 0:
            torch.save([p.detach().clone() for p in params], out_path)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_scatter_sharded_adam_all_gather_equals_allreduce_plus_adam(tmp_path, world):
    out = tmp_path / "params.pt"
    mp.spawn(_worker, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    # one process: the summed gradients through torch.optim.Adam, one optimizer per parameter as the reference has
    params = _params()
    opts = [torch.optim.Adam([p], **h) for p, h in zip(params, HYPER)]
    for step in range(STEPS):
        for p in params:
            p.grad = torch.zeros_like(p)
        for r in range(world):
            for p, g in zip(params, _grads(step, r)):
                p.grad.add_(g)
        for o in opts:
            o.step()
    for a, b, shape in zip(got, params, SHAPES):
        assert a.shape == b.shape == torch.Size(shape)
        assert torch.allclose(a, b.detach(), rtol=2e-6, atol=1e-7), float((a - b).abs().max())


def test_single_process_sharded_step_is_a_plain_adam_step():
    from gaussiangrasper_amd.dist import GradBucket, ShardedAdamStep, torch_adam_piece
    params = _params(3)
    ref = [p.detach().clone().requires_grad_(True) for p in params]
    bucket = GradBucket(params)
    stepper = ShardedAdamStep(bucket, HYPER, adam_piece=torch_adam_piece)
    opts = [torch.optim.Adam([p], **h) for p, h in zip(ref, HYPER)]
    for step in range(3):
        bucket.zero_()
        for p, q, g in zip(params, ref, _grads(step, 0)):
            p.grad.add_(g)
            q.grad = g.clone()
        stepper.step()
        for o in opts:
            o.step()
    for a, b in zip(params, ref):
        assert torch.allclose(a.detach(), b.detach(), rtol=2e-6, atol=1e-7)

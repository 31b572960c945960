"""RCCL smoke (one GPU): the collectives of dist.GradBucket (per-parameter asynchronous all-reduce, armed on the step's
last backward) and of dist.ShardedAdamStep (reduce_scatter_tensor / all_gather_into_tensor) issued through a real
`nccl` (= RCCL) process group of ONE rank, on the tensors and in the order the multi-GPU paths use.  A one-rank group
moves no data between GPUs — the 8-GPU run is the driver's — but every call goes through RCCL's argument checks,
stream handling and work objects, which until round 3 only gloo had done for this code.  In a subprocess: a process
group is global state.  Reference counterpart: nerfstudio/scripts/train.py:139-145 (`init_process_group("nccl")`),
pipelines/base_pipeline.py:303-305 (DDP's bucketed all-reduce)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys
    sys.path[:0] = [%r, %r]
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%d))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import gaussiangrasper_amd.dist as D
    D._dist_on = lambda: True          # a one-rank group still issues every collective
    shapes = [(5000, 3), (5000, 3), (5000, 4), (5000, 1), (5000, 25, 3), (5000, 32)]
    g = torch.Generator().manual_seed(0)
    params = [torch.randn(*s, generator=g).to(dev).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone() for p in params]
    bucket = D.GradBucket(params)
    grads = [torch.randn(*s, generator=g).to(dev) for s in shapes]

    def render_and_backward(v):        # a "view": every parameter receives a gradient through autograd
        loss = sum((p * gr).sum() for p, gr in zip(params, grads))
        loss.backward()

    for overlap in (True, False):      # per-parameter async all-reduce (hooks) and the single collective
        D.train_step(render_and_backward, bucket, [0, 1, 2], overlap=overlap)
        torch.cuda.synchronize()
        for p, gr in zip(params, grads):
            assert torch.allclose(p.grad, 3 * gr, rtol=1e-6, atol=1e-6)
    assert bucket._order is not None and sorted(bucket._order) == list(range(6))
    # reduce-scatter + fused Adam (gg_adam_step) on the shard + all-gather
    lrs = (1.6e-4, 0.005, 0.001, 0.05, 5e-4, 5e-4)
    stepper = D.ShardedAdamStep(bucket, [dict(lr=lr, eps=1e-15) for lr in lrs])
    opts = [torch.optim.Adam([r.requires_grad_(True)], lr=lr, eps=1e-15) for r, lr in zip(ref, lrs)]
    for step in range(3):
        D.train_step(render_and_backward, bucket, [0], reduce=False)
        stepper.step()
        for r, gr, o in zip(ref, grads, opts):
            r.grad = gr.clone()
            o.step()
    torch.cuda.synchronize()
    for p, r in zip(params, ref):
        assert torch.allclose(p.detach(), r.detach(), rtol=2e-6, atol=1e-7), float((p - r).abs().max())
    dist.destroy_process_group()
    print("RCCL-OK", dist.is_nccl_available())
''')


@pytest.mark.gpu
def test_collectives_of_both_reduction_schemes_through_rccl_on_one_rank():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, os.path.join(ROOT, "shim"), port)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])

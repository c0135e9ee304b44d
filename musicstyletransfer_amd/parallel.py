"""Data parallelism for the VAE step: one process per GPU, the minibatch sharded contiguously, ONE
collective per step — a SUM all-reduce of the flat fp32 gradient bucket (RCCL over xGMI; `nccl` IS RCCL
on ROCm) — followed by the same fused Adam on every rank.

The reference has no counterpart (single context everywhere: main.py:124, trainer.py:75,99-101); the
only cross-sample operation in its step is the implicit sum over the batch in loss.backward() plus the
1/B of Trainer.step(batch_size) (trainer.py:176-177). Hence:
  * every rank normalises by the GLOBAL batch size (StepPlan(global_batch=...)),
  * every rank pads to the GLOBAL batch-max length, because SoftmaxCrossEntropy divides by the padded
    length T (loss.py:23) and data.py:196-198 truncates to the batch maximum,
  * parameters, Adam state and positional tables are replicated; initial weights are identical on every
    rank (same seed), and identical reduced gradients keep them identical.
The bucket is 7.5 MB (configs[1]); xGMI is point-to-point (7 links x ~153 GB/s per GPU), so the
collective costs tens of microseconds and is issued as a single call on the step's stream between the
captured forward/backward graph and the captured optimizer graph.
"""
import os

import numpy as np
import torch


def init_process_group(world, rank, backend=None):
    """env:// rendezvous on 127.0.0.1 (the container hostname may not resolve). The launcher (torch.distributed.run)
    exports MASTER_PORT; the fallback below only serves hand-started ranks, which must agree on a port anyway.
    Call torch.cuda.set_device() for this rank's GPU FIRST: the nccl (= RCCL) group binds to the current device.
    MST_DIST_BACKEND=gloo rehearses the multi-rank path with the ranks sharing one card (tests, one-GPU boxes)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver
    if backend is None:
        backend = os.environ.get("MST_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank, **kw)
    return dist


def make_grad_allreduce(dist):
    """returns reduce_fn(flat_grad): in-place SUM over ranks, asynchronous on the current stream (nccl) or
    blocking (gloo, CPU tests)"""
    def reduce_fn(flat):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return reduce_fn


class GradReducer:
    """Asynchronous SUM all-reduce of ranges of the flat gradient bucket. start() enqueues the collective behind
    everything already issued on the current stream (nccl: on RCCL's own stream, so kernels launched afterwards on
    the current stream overlap with it; gloo: a host-side work item), finish() makes the current stream (nccl) or the
    host (gloo) wait for the given handles. This is the comm/compute overlap of the data-parallel step: the part of
    the bucket that the top of the backward pass completes travels over xGMI while the rest of backward executes."""

    def __init__(self, dist):
        self.dist = dist

    def start(self, flat_range):
        return self.dist.all_reduce(flat_range, op=self.dist.ReduceOp.SUM, async_op=True)

    def finish(self, handles):
        for h in handles:
            h.wait()


def shard_bounds(global_batch, world, rank):
    """contiguous shard [lo, hi) of the global batch for `rank`"""
    assert global_batch % world == 0, "global batch must divide evenly over the ranks"
    per = global_batch // world
    return rank * per, (rank + 1) * per


def shard_batch(batch, world, rank):
    """slice every per-sample array of a batch dict; the padded length (axis 1) is left at the global
    batch maximum on purpose (see module docstring)"""
    B = len(batch["seq_lens"])
    lo, hi = shard_bounds(B, world, rank)
    return {k: v[lo:hi] for k, v in batch.items()}


def global_eps(seed, step, global_batch, latent_dim, lo, hi):
    """Host-side eps drawn per GLOBAL sample index, for callers that INJECT eps (StepPlan(internal_eps=False)): results
    then do not depend on how the batch is sharded. The in-graph draw of the training step follows the same rule on the
    device: mst_step_begin(eps_index0 = first global sample index * Z), see StepPlan(sample_offset=...)."""
    rng = np.random.default_rng([seed, step])
    return rng.standard_normal((global_batch, latent_dim)).astype(np.float32)[lo:hi]

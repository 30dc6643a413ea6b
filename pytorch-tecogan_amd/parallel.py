"""Data parallelism for the training step: one process per GPU, sequences sharded over ranks, replicated weights.

The reference has no distributed code at all; this is new.  Per step and per network ONE flat fp32
gradient buffer is all-reduced (sum) over RCCL ("nccl" backend == RCCL on ROCm; gloo on CPU for the tests) and the
1/world scaling is folded into the fused Adam kernel (hyper[6]).  Both all-reduces are launched (async, concurrently) right
after the forward+backward graph segment and awaited before the update segment (step.TecoGANStep._run).

BatchNorm statistics stay per rank (standard DDP; the reference's D is called on per-rank batches anyway), so an N-rank run
equals "N shards evaluated with local BN, gradients averaged" - that is what tests/test_parallel_cpu.py checks on gloo."""
import os

import torch
import torch.distributed as dist


def dist_info():
    """(process_group or None, world_size)."""
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1 or os.environ.get("TECOGAN_FORCE_DP_SEGMENTS", "0") == "1":
            return dist.group.WORLD, dist.get_world_size()
    return None, 1


def shard_bounds(n, world, rank):
    """rank r of `world` owns sequences [lo, hi) of a global batch of n (n divisible by world)."""
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return rank * per, (rank + 1) * per


def allreduce_sum_async(buf, group, world):
    """launches the all-reduce of one flat gradient buffer; returns a work handle (None when single process)."""
    if group is None or world == 1:
        return None
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)


def wait_all(works):
    for w in works:
        if w is not None:
            w.wait()


def average_gradients_(bufs, group, world):
    """synchronous helper (tests, eager tools): sum all-reduce every buffer and scale by 1/world in place."""
    works = [allreduce_sum_async(b, group, world) for b in bufs]
    wait_all(works)
    if world > 1:
        for b in bufs:
            b.mul_(1.0 / world)
    return bufs

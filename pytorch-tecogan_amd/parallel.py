"""Data parallelism for the training step: one process per GPU, sequences sharded over ranks, replicated weights.

The reference has no distributed code at all; this is new.  Per step and per network ONE flat fp32
gradient buffer is all-reduced (sum) over RCCL ("nccl" backend == RCCL on ROCm; gloo on CPU for the tests) and the
1/world scaling is folded into the fused Adam kernel (hyper[6]).  The G all-reduce is issued behind the G backward (lane A) while lane B
is still in the fake half of the D backward, the D all-reduce behind lane B (step.TecoGANStep._run_lanes).  Replicas must
start equal: broadcast_state() sends rank 0's parameters, BN buffers and Adam moments to every rank (main.py calls it
after construction and after a checkpoint load).

BatchNorm statistics stay per rank (standard DDP; the reference's D is called on per-rank batches anyway), so an N-rank run
equals "N shards evaluated with local BN, gradients averaged" - that is what tests/test_parallel_cpu.py checks on gloo."""
import os

import torch
import torch.distributed as dist


def dist_info():
    """(process_group or None, world_size)."""
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1 or os.environ.get("TECOGAN_FORCE_COLLECTIVES", "0") == "1":
            return dist.group.WORLD, dist.get_world_size()
    return None, 1


def shard_bounds(n, world, rank):
    """rank r of `world` owns sequences [lo, hi) of a global batch of n (n divisible by world)."""
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return rank * per, (rank + 1) * per


class _StagedWork:
    """all-reduce of a DEVICE buffer over a gloo group (the tests' rendezvous when several ranks share one GPU; RCCL
    refuses that).  Issued like an RCCL collective - nothing blocks at issue: the device-to-host copy is enqueued on the
    issuing stream, a worker thread (ONE per process, FIFO, so every rank runs its collectives in issue order) waits for
    the copy's event and reduces the pinned host buffer; wait() joins that job and enqueues the copy back on the
    caller's current stream.  The step's launches issued between all-reduce and wait() run meanwhile, which is the
    interleave the RCCL path has."""
    _queue, _thread, _pinned = None, None, {}

    def __init__(self, buf, group):
        import queue
        import threading
        cls = _StagedWork
        if cls._thread is None or not cls._thread.is_alive():
            cls._queue = queue.Queue()
            cls._thread = threading.Thread(target=cls._serve, args=(cls._queue,), daemon=True)
            cls._thread.start()
        key = (buf.data_ptr(), buf.numel())
        host = cls._pinned.get(key)
        if host is None:
            host = cls._pinned[key] = torch.empty(buf.numel(), dtype=buf.dtype).pin_memory()
        self.buf, self.host, self.group = buf, host, group
        self.done, self.error = threading.Event(), None
        host.copy_(buf.view(-1), non_blocking=True)
        self.copied = torch.cuda.Event()
        self.copied.record()
        cls._queue.put(self)

    @staticmethod
    def _serve(q):
        while True:
            w = q.get()
            try:
                w.copied.synchronize()
                dist.all_reduce(w.host, op=dist.ReduceOp.SUM, group=w.group)
            except Exception as e:  # noqa: BLE001  (re-raised by wait())
                w.error = e
            w.done.set()

    def wait(self):
        self.done.wait()
        if self.error is not None:
            raise self.error
        self.buf.view(-1).copy_(self.host, non_blocking=True)


def allreduce_sum_async(buf, group, world):
    """launches the all-reduce of one flat gradient buffer on the current stream's data; returns a work handle with
    .wait() (None when single process).  RCCL: Work.wait() makes the current STREAM wait.  gloo with a device buffer (the
    tests' two-ranks-on-one-GPU rendezvous): _StagedWork - asynchronous at issue, host-joined at wait()."""
    if group is None or world == 1:
        return None
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        return _StagedWork(buf, group)
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)


def broadcast_state(modules, optimizers=(), group=None, src=0):
    """makes every rank a replica of rank `src`: parameters and buffers of `modules` (BN running statistics included)
    and the tensor state of `optimizers` (exp_avg, exp_avg_sq, step).  In-place, so views of the flat buffers stay views.
    Returns the number of tensors sent.  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    staged = dist.get_backend(group) == "gloo"
    n = 0
    seen = set()

    def send(t):
        nonlocal n
        key = (t.data_ptr(), t.numel())
        if key in seen:          # the fused update keeps ONE step tensor per optimiser
            return
        seen.add(key)
        if staged and t.is_cuda:
            h = t.detach().cpu()
            dist.broadcast(h, src=src, group=group)
            t.detach().copy_(h)
        else:
            dist.broadcast(t.detach(), src=src, group=group)
        n += 1

    for m in modules:
        for p in m.parameters():
            send(p.data)
        for b in m.buffers():
            send(b)
        if hasattr(m, "mark_weights_changed"):
            m.mark_weights_changed()       # packed compute copies are rebuilt before the next launch
    for opt in optimizers:
        for grp in opt.param_groups:
            for p in grp["params"]:
                st = opt.state.get(p)
                if not st:
                    continue
                for key in ("exp_avg", "exp_avg_sq", "step"):
                    if torch.is_tensor(st.get(key)):
                        send(st[key])
    return n


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


def replicas_equal(modules, group=None):
    """True when every rank holds bit-identical parameters (compared through per-module fp64 checksums gathered from all
    ranks; cheap enough for once per epoch).  Data-parallel training keeps replicas equal by construction - equal start
    (broadcast_state), equal all-reduced gradients, deterministic update - so a False here means a broken invariant."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    vals = []
    for m in modules:
        flat = torch.cat([p.detach().double().flatten() for p in m.parameters()])
        vals += [flat.sum(), flat.abs().sum(), (flat * torch.arange(1, flat.numel() + 1, device=flat.device,
                                                                    dtype=torch.float64)).sum()]
    mine = torch.stack(vals).cpu()
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "gloo":
        mine = mine.to(next(modules[0].parameters()).device)
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine, group=group)
    return all(torch.equal(both[0], b) for b in both[1:])

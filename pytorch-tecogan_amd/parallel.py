"""Data parallelism for the training step: one process per GPU, sequences sharded over ranks, replicated weights.

The reference has no distributed code at all; this is new.  Per step and per network ONE flat fp32
gradient buffer is all-reduced (sum) over RCCL ("nccl" backend == RCCL on ROCm; gloo on CPU for the tests) and the
1/world scaling is folded into the fused Adam kernel (hyper[6]).  The G all-reduce is issued behind the G backward (lane A) while lane B
is still in the fake half of the D backward, the D all-reduce behind lane B (step.TecoGANStep._run_lanes).  Replicas must
start equal: broadcast_state() sends rank 0's parameters, BN buffers and Adam moments to every rank (main.py calls it
after construction and after a checkpoint load).

BatchNorm statistics stay per rank (standard DDP; the reference's D is called on per-rank batches anyway), so an N-rank run
equals "N shards evaluated with local BN, gradients averaged" - that is what tests/test_parallel_cpu.py checks on gloo."""
import torch
import torch.distributed as dist

from . import tuning


def dist_info():
    """(process_group or None, world_size)."""
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1 or tuning.current().force_collectives:
            return dist.group.WORLD, dist.get_world_size()
    return None, 1


class _GroupCache:
    """per-process-group facts, keyed by the group OBJECT: after destroy_process_group() + init_process_group() in one process a new
    group can reuse the old one's id(), and an id-keyed cache would then skip the backend warm-up (the 13-ms bucket-mode replay it cures
    would be back) and reuse a stale stream-ordering verdict.  Weak references where the object allows them; otherwise the entry keeps the
    object alive, so its id cannot be reused while the entry exists."""

    def __init__(self):
        import weakref
        self._weak, self._strong = weakref.WeakKeyDictionary(), {}

    def get(self, group, default=None):
        try:
            return self._weak.get(group, default)
        except TypeError:
            ent = self._strong.get(id(group))
            return ent[1] if ent is not None and ent[0] is group else default

    def set(self, group, value):
        try:
            self._weak[group] = value
        except TypeError:
            self._strong[id(group)] = (group, value)

    def clear(self):
        self._weak.clear()
        self._strong.clear()


_SYNC_ORDERED = _GroupCache()   # group -> {"ok": bool | None, "host_ms": float}
_WARMED = _GroupCache()         # group -> True


def warm_backend(group, device):
    """one synchronous and one ASYNCHRONOUS all-reduce of a tiny tensor, once per process group, before the first step is built.
    The RCCL backend creates its internal stream at the first asynchronous collective; when that happened AFTER a step had been
    built and replayed with synchronous collectives only (the inline mode), a bucket-mode step built next in the same process
    replayed at 12.5-13.4 ms instead of 3.9 - with 8, 16 or 24 hardware queues alike - and at 3.9 when the asynchronous path had
    been used once beforehand (tools/pg_tax_probe.py, profiles/r05_c_pg_tax_probe.log).  A collective: every rank builds its first
    data-parallel step at the same point.  No-op for gloo (its device buffers take the staged path)."""
    if group is None or not (dist.is_available() and dist.is_initialized()) or _WARMED.get(group):
        return
    _WARMED.set(group, True)
    if dist.get_backend(group) == "gloo":
        return
    t = torch.zeros(8, dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w.wait()
    torch.cuda.synchronize(device)


def sync_allreduce_stream_ordered(group, device):
    """True when `dist.all_reduce(t, group=group)` (async_op=False) issued under a side stream is ORDERED on that stream: it
    reads what the stream wrote before it and the stream's next operation sees its result - with the producer deliberately
    late (a ~8 ms device-side sleep in front of the write), so an implementation that ran the collective elsewhere without
    waiting would reduce stale zeros.  The data-parallel default (step.TecoGANStep, TECOGAN_DP_INLINE) issues ONE such call
    per network on its lane's stream and relies on exactly this; DESIGN.md (e) observed it for torch 2.10's RCCL backend, this
    checks it where it is used.  None: not applicable (no group, gloo - those take the staged asynchronous path).  One check per
    process group; every rank must call it at the same point (it is a collective)."""
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_backend(group) == "gloo":
        return None
    ent = _SYNC_ORDERED.get(group)
    if ent is not None:
        return ent["ok"]
    import time
    world = dist.get_world_size(group)
    t = torch.zeros(2, dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)      # communicator warm-up (lazy initialisation is host work)
    torch.cuda.synchronize(device)
    s = torch.cuda.Stream(device=device)
    with torch.cuda.stream(s):
        torch.cuda._sleep(int(0.008 * 2.0e9))
        t.fill_(1.0)                                           # the producer, late
        t0 = time.perf_counter()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)  # the call under test
        host_ms = (time.perf_counter() - t0) * 1e3
        out = t * 2.0                                          # the consumer, on the same stream
    s.synchronize()
    ok = bool((out == 2.0 * world).all().item())
    # the verdict picks the collective SCHEDULE (one synchronous all-reduce per network, or asynchronous buckets): it has to be the
    # same on every rank - a backend that is not stream-ordered gives a racy, rank-local answer, and ranks that disagreed would
    # issue collectives of different sizes.  MIN over the group: everybody falls back together.
    if world > 1:
        v = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=device)
        dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
        torch.cuda.synchronize(device)
        ok = bool(v.item() >= 1.0)
    _SYNC_ORDERED.set(group, {"ok": ok, "host_ms": host_ms})   # (host_ms, diagnostic: a stream-ordered call returns long before the sleep ends)
    return ok


def streams_overlap(device, s1, s2):
    """True when work on the two streams really runs concurrently.  The step's two lanes need their own hardware queues: with an
    RCCL process group in the process and the HIP runtime's default GPU_MAX_HW_QUEUES (4) both lane streams land on ONE queue
    and the step runs 1.6x slower (profiles/r03_c_dp_hw_queues.log).  The package sets GPU_MAX_HW_QUEUES=8 at
    import - which only works if that happens before the process's FIRST HIP call (torch.cuda.is_available() / device_count()
    may already be one), so the effect is measured here instead of assumed: a device-side sleep on each stream, alone and
    together (~1.5 ms of GPU time, once per step construction under a process group)."""
    def span(streams):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream(device)
        torch.cuda.synchronize(device)
        e0.record(cur)
        for st in streams:
            st.wait_event(e0)
            with torch.cuda.stream(st):
                torch.cuda._sleep(int(0.0004 * 2.0e9))
            cur.wait_stream(st)
        e1.record(cur)
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1)
    span([s1, s2])                       # (a stream's first launch creates its hardware queue: not part of the measurement)
    one = min(span([s1]), span([s1]))
    both = min(span([s1, s2]), span([s1, s2]))
    return both < 1.6 * one


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
    _queue, _thread, _pinned, _groups = None, None, {}, {}

    def __init__(self, buf, group):
        import queue
        import threading
        cls = _StagedWork
        if cls._thread is None or not cls._thread.is_alive():
            cls._queue = queue.Queue()
            cls._thread = threading.Thread(target=cls._serve, args=(cls._queue,), daemon=True)
            cls._thread.start()
        key = (buf.data_ptr(), buf.numel(), buf.dtype)
        host = cls._pinned.get(key)
        if host is None:
            host = cls._pinned[key] = torch.empty(buf.numel(), dtype=buf.dtype).pin_memory()
        # the worker thread reduces on a gloo group of its OWN: the main thread may issue collectives on `group` at the same time
        # (replicas_equal, barrier, broadcast_state), and two threads on one group have no common issue order across ranks.
        # (created at the first staged all-reduce, which every rank reaches at the same point of the step)
        own = cls._groups.get(id(group))
        if own is None:
            own = cls._groups[id(group)] = dist.new_group(ranks=dist.get_process_group_ranks(group), backend="gloo")
        self.buf, self.host, self.group = buf, host, own
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

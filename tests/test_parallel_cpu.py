"""Data-parallel path on CPU: world_size 2 over gloo.  Each rank owns one sequence, computes its gradients (CPU oracle),
puts them into the same flat buffers the HIP engine uses and runs parallel.average_gradients_; the result must equal
"both shards evaluated sequentially on one process with local BatchNorm, gradients averaged" (SURVEY.md 8e), the generator
part must also equal the global-batch gradient (G has no cross-sample op), and both ranks must end with identical weights."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _synth(seed):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.random((1, 10, 3, 32, 32), dtype=np.float32))
    y = torch.from_numpy(rng.random((1, 10, 3, 128, 128), dtype=np.float32))
    return x, y


def _grads(orc, gp, dp, x, y, args):
    g = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    d = {k: v.clone().requires_grad_(True) for k, v in dp.items()}
    bufs = orc.init_bn_buffers(dp, args.discrim_resblocks)
    f = orc.tecogan_forward(g, d, bufs, x, y, args, 0)
    gg = torch.autograd.grad(f["gen_loss"], list(g.values()), retain_graph=True)
    dg = torch.autograd.grad(f["d_loss"], list(d.values()))
    return dict(zip(g.keys(), gg)), dict(zip(d.keys(), dg))


def _worker(rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.set_num_threads(3)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import pytorch_tecogan_amd  # noqa: F401
        from pytorch_tecogan_amd import engine as E
        from pytorch_tecogan_amd import parallel
        import tecogan_oracle as orc

        args = orc.default_args(num_resblock=1, discrim_resblocks=1)
        gshapes, dshapes = E.generator_shapes(1), E.discriminator_shapes(1, 128)
        gp = orc.init_params(orc.generator_param_shapes(1), 7)
        dp = orc.init_params(orc.discriminator_param_shapes(1, 128), 8)
        lo, hi = parallel.shard_bounds(2, world, rank)
        assert (lo, hi) == (rank, rank + 1)
        x, y = _synth(20 + rank)
        gg, dg = _grads(orc, gp, dp, x, y, args)
        fg, fd = E.FlatParams(gshapes, torch.device("cpu")), E.FlatParams(dshapes, torch.device("cpu"))
        for k in gp:
            fg.view(fg.g, k).copy_(gg[k])
        for k in dp:
            fd.view(fd.g, k).copy_(dg[k])
        group, w = parallel.dist_info()
        assert w == world
        parallel.average_gradients_([fg.g, fd.g], group, w)
        og, od = orc.AdamState(gp, 1e-4), orc.AdamState(dp, 1e-4)
        with torch.no_grad():
            og.step(gp, {k: fg.view(fg.g, k) for k in gp})
            od.step(dp, {k: fd.view(fd.g, k) for k in dp})
        digest = torch.tensor([float(sum(v.double().sum() for v in gp.values())),
                               float(sum(v.double().sum() for v in dp.values()))], dtype=torch.float64)
        both = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, digest)
        res = {"rank": rank, "digests": [b.tolist() for b in both]}
        if rank == 0:
            # sequential emulation on one process
            gp0 = orc.init_params(orc.generator_param_shapes(1), 7)
            dp0 = orc.init_params(orc.discriminator_param_shapes(1, 128), 8)
            acc_g = {k: torch.zeros_like(v) for k, v in gp0.items()}
            acc_d = {k: torch.zeros_like(v) for k, v in dp0.items()}
            xs, ys = [], []
            for r in range(world):
                xr, yr = _synth(20 + r)
                xs.append(xr)
                ys.append(yr)
                g1, d1 = _grads(orc, gp0, dp0, xr, yr, args)
                for k in acc_g:
                    acc_g[k] += g1[k] / world
                for k in acc_d:
                    acc_d[k] += d1[k] / world
            rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))  # noqa: E731
            res["g_vs_seq"] = max(rel(fg.view(fg.g, k), acc_g[k]) for k in gp0)
            res["d_vs_seq"] = max(rel(fd.view(fd.g, k), acc_d[k]) for k in dp0)
            # generator: averaged shard gradients == gradient of the global batch (no cross-sample coupling in G)
            gB, _ = _grads(orc, gp0, dp0, torch.cat(xs), torch.cat(ys), args)
            big = [k for k in gp0 if float(gB[k].norm()) > 1e-6]
            res["g_vs_global"] = max(rel(fg.view(fg.g, k), gB[k]) for k in big)
            og0, od0 = orc.AdamState(gp0, 1e-4), orc.AdamState(dp0, 1e-4)
            with torch.no_grad():
                og0.step(gp0, acc_g)
                od0.step(dp0, acc_d)
            res["w_vs_seq"] = max(rel(gp[k], gp0[k]) for k in gp0)
        q.put(res)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})
        raise


@pytest.mark.timeout(900)
def test_gradient_averaging_over_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=800) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert "error" not in r, r.get("error")
    r0 = next(r for r in results if r["rank"] == 0)
    assert r0["g_vs_seq"] < 1e-6 and r0["d_vs_seq"] < 1e-6, r0
    assert r0["g_vs_global"] < 2e-3, r0      # equal up to fp32 summation order (content-loss gradients cancel heavily)
    assert r0["w_vs_seq"] < 1e-6, r0
    for r in results:                          # replicas stay bit-identical
        assert r["digests"][0] == r["digests"][1], r


def test_shard_bounds():
    from pytorch_tecogan_amd import parallel
    assert [parallel.shard_bounds(32, 8, r) for r in (0, 7)] == [(0, 4), (28, 32)]
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 4, 0)


def _bcast_worker(rank, world, port, q):
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import argparse
        import pytorch_tecogan_amd  # noqa: F401
        from pytorch_tecogan_amd import models as M, parallel
        args = argparse.Namespace(num_resblock=1, discrim_resblocks=1, discrim_channels=128, crop_size=32)
        torch.manual_seed(100 + rank)                      # every rank draws its OWN initial weights (main.py does too)
        G, D = M.generator(3, args), M.discriminator(args)
        D.block1._modules["1"].running_mean.fill_(float(rank + 1))
        og = torch.optim.Adam(G.parameters(), 1e-4)
        for p in G.parameters():                            # a populated optimiser state, different per rank
            og.state[p] = {"step": torch.tensor(float(3 + rank)), "exp_avg": torch.full_like(p, rank + 1.0),
                           "exp_avg_sq": torch.full_like(p, rank + 2.0)}
        n = parallel.broadcast_state((G, D), (og,))
        dig = torch.tensor([float(sum(p.double().sum() for p in G.parameters())),
                            float(sum(p.double().sum() for p in D.parameters())),
                            float(sum(b.double().sum() for b in D.buffers())),
                            float(sum(s["exp_avg"].double().sum() + s["exp_avg_sq"].double().sum() + s["step"].double()
                                      for s in og.state.values()))], dtype=torch.float64)
        both = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, dig)
        q.put({"rank": rank, "n": n, "digests": [b.tolist() for b in both], "dirty": (G._dirty, D._dirty)})
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})
        raise


@pytest.mark.timeout(300)
def test_broadcast_state_makes_replicas_of_rank0():
    """ranks construct different initial weights; parallel.broadcast_state must make all of them
    (parameters, BN buffers, Adam moments and step) equal to rank 0's and mark the packed weights stale."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert "error" not in r, r.get("error")
        assert r["digests"][0] == r["digests"][1], r
        assert r["n"] > 50 and r["dirty"] == (True, True)
    # rank 0's own state is what everybody has: its BN running_mean was filled with 1.0, its Adam step was 3
    assert results[0]["digests"][0] == results[1]["digests"][0]


def test_broadcast_state_is_a_noop_without_a_process_group():
    from pytorch_tecogan_amd import parallel
    assert parallel.broadcast_state((torch.nn.Linear(2, 2),)) == 0


def _world8_worker(rank, world, port, q):
    """the data-parallel HOST logic at the driver's scaling-run size: shard bounds, rank-0 broadcast (ranks draw different weights),
    flat-buffer averaging with a rank-dependent gradient, Adam on the averaged gradient, replica check - no oracle evaluation (eight of
    those do not fit this box's time budget; world 2 above does the numerics)"""
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import argparse
        import pytorch_tecogan_amd  # noqa: F401
        from pytorch_tecogan_amd import models as M, parallel
        lo, hi = parallel.shard_bounds(32, world, rank)        # configs[2]: global batch 32 -> 4 sequences per GPU
        assert (lo, hi) == (4 * rank, 4 * rank + 4)
        lo16, hi16 = parallel.shard_bounds(16, world, rank)    # configs[3]: global batch 16 -> 2 per GPU
        assert hi16 - lo16 == 2
        args = argparse.Namespace(num_resblock=1, discrim_resblocks=1, discrim_channels=128, crop_size=32)
        torch.manual_seed(500 + rank)
        G, D = M.generator(3, args), M.discriminator(args)
        og = torch.optim.Adam(G.parameters(), 1e-4)
        group, w = parallel.dist_info()
        assert w == world and group is not None
        before = parallel.replicas_equal((G, D), group)        # every rank drew its own weights
        n = parallel.broadcast_state((G, D), (og,), group)
        after = parallel.replicas_equal((G, D), group)
        # the flat gradient buffers the engines all-reduce: rank r holds (r + 1) * pattern; the average is (world + 1) / 2 * pattern
        gflat = torch.cat([p.detach().flatten() for p in G.parameters()])
        pat = torch.arange(gflat.numel(), dtype=torch.float32).remainder(97.0) / 97.0 - 0.5
        bufs = [(rank + 1.0) * pat.clone(), (rank + 1.0) * 2.0 * pat[:1000].clone()]
        parallel.average_gradients_(bufs, group, w)
        err = float((bufs[0] - (world + 1) / 2.0 * pat).abs().max()), float((bufs[1] - (world + 1.0) * pat[:1000]).abs().max())
        # a step on the averaged gradient keeps the replicas equal
        off = 0
        for p in G.parameters():
            p.grad = bufs[0][off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        og.step()
        still = parallel.replicas_equal((G, D), group)
        q.put({"rank": rank, "before": before, "after": after, "still": still, "n": n, "err": err})
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})
        raise


@pytest.mark.timeout(600)
def test_world_size_8_rehearsal_over_gloo():
    """nothing had ever run at the world size of the driver's scaling run.  Eight gloo ranks on this box's CPUs:
    shard bounds of configs[2] / configs[3], broadcast of rank 0's state, averaging of the flat gradient buffers, replica equality
    before (False: every rank initialises by itself), after the broadcast and after an update on the averaged gradient (True)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert "error" not in r, r.get("error")
        assert r["before"] is False and r["after"] is True and r["still"] is True, r
        assert r["n"] > 50 and max(r["err"]) < 1e-5, r
    assert sorted(r["rank"] for r in results) == list(range(world))

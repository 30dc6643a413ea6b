"""Throughput bench of the TecoGAN training step (BASELINE.json config 2: B=4 sequences per GPU, T=10, 32x32 -> 128x128,
bf16 compute / fp32 master weights, full G + pseudo-flow/warp + D + losses + two Adam updates per step).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched once per rank by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      the dominant kernel family, timed live with events around every one of its launches in one eager step
  cpu_baseline  the CPU oracle (PyTorch-CPU fp32 restatement of the reference) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
STEP_GFLOP_PER_SEQ = 380.0  # 10 G frames * 25.884 + 6 D samples * 20.196 GFLOP (T=10, cs=32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="sequences per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def default_args(dtype):
    return argparse.Namespace(RNN_N=10, crop_size=32, num_resblock=16, discrim_resblocks=4, discrim_channels=128,
                              pingpang=False, pp_scaling=1.0, vgg_scaling=-0.002, crop_dt=0.75, Dt_mergeDs=True,
                              D_LAYERLOSS=True, EPS=1e-12, ratio=0.01, Dt_ratio_0=1.0, Dt_ratio_add=0.0, Dt_ratio_max=1.0,
                              learning_rate=1e-4, beta=0.9, adameps=1e-8, tg_dtype=dtype)


def synth(B, T, cs, seed):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.random((B, T, 3, cs, cs), dtype=np.float32))
    y = torch.from_numpy(rng.random((B, T, 3, 4 * cs, 4 * cs), dtype=np.float32))
    return x, y


def conv_flops(spec, N, H, W):
    """algorithmic FLOPs (2*MAC on REAL channels) of one forward/dgrad/wgrad launch of `spec` on an [N,*,H,W] input."""
    if spec.kind == "c3":
        return 2.0 * N * H * W * 9 * spec.cin * spec.cout
    if spec.kind == "c4s2":
        return 2.0 * N * (H // 2) * (W // 2) * 16 * spec.cin * spec.cout
    return 2.0 * N * H * W * 9 * spec.cin * spec.cout  # conv-transpose: every input pixel meets all 9 taps


TILE_PARAMS = {1: "4, 4, 1, 4", 2: "2, 2, 2, 2", 3: "4, 4, 2, 2", 4: "2, 2, 1, 4", 5: "2, 1, 1, 4", 6: "4, 2, 1, 4",
               7: "2, 2, 2, 4"}


def kernel_name(conv, dtype):
    """the rocprofv3 kernel name of the launch `conv` just made (template parameters from the C library's launch plan)"""
    import ctypes
    from pytorch_tecogan_amd import _lib as L
    if conv.last_desc is None or conv.last_desc == "c4d":  # the sub-pixel launches (csrc/convt_mfma.hip)
        return f"subpixel_kernel<{'BF16' if dtype == 'bf16' else 'F32'}, {0 if conv.last_desc is None else 1}>"
    if conv.last_desc in ("c4s2", "ctd"):  # csrc/conv4s2_mfma.hip
        return f"conv_s2_gather_kernel<{'BF16' if dtype == 'bf16' else 'F32'}, {4 if conv.last_desc == 'c4s2' else 3}>"
    plan = L.load().tg_conv_pick_tile(ctypes.byref(conv.last_desc))
    t = "BF16" if dtype == "bf16" else "F32"
    return f"conv_gather_kernel<{t}, {TILE_PARAMS[plan & 255]}, {'true' if plan >> 8 else 'false'}>"


def roofline_pass(st, dtype):
    """One eager step with a start/stop event pair around every MFMA launch; returns the per-family table."""
    from pytorch_tecogan_amd import engine as E
    from pytorch_tecogan_amd import kernels as K
    recs, replays = [], {}
    orig_fwd, orig_dgrad, orig_wgrad = E.Conv.fwd, E.Conv.dgrad, E.Conv.wgrad

    def timed(label_fn, flops_fn, fn):
        def wrapper(self, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(self, *a, **kw)
            e1.record()
            lab = label_fn(self, *a)  # label AFTER the call: needs last_desc
            recs.append((lab, flops_fn(self, *a), e0, e1))
            replays.setdefault(lab, []).append(lambda: fn(self, *a, **kw))
            return r
        return wrapper

    def in_shape_fwd(self, x, *a):
        return x.shape[0], x.shape[1], x.shape[2]

    def lab_fwd(self, x, *a):
        return kernel_name(self, dtype)

    def lab_dgrad(self, dout, out, *a):
        return kernel_name(self, dtype)

    def fl_fwd(self, x, *a):
        return conv_flops(self.spec, *in_shape_fwd(self, x))

    def fl_dgrad(self, dout, out, *a):
        return conv_flops(self.spec, out.shape[0], out.shape[1], out.shape[2])

    def fl_wgrad(self, x_in, dout, *rest):
        return conv_flops(self.spec, x_in.shape[0], x_in.shape[1], x_in.shape[2])

    # the fused residual-block launch (csrc/resblock.hip) and the grouped weight-gradient launch are not Conv methods
    orig_rb, orig_rbb, orig_group = K.resblock_fwd, K.resblock_bwd, E.WgradGroup.launch

    def rb_wrap(orig, label):
        def rb_timed(x, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(x, *a, **kw)
            e1.record()
            N, H, W, C_ = x.shape
            recs.append((label, 2 * 2.0 * N * H * W * 9 * C_ * C_, e0, e1))  # algorithmic: two 3x3 convs
            replays.setdefault(label, []).append(lambda: orig(x, *a, **kw))
        return rb_timed

    def group_timed(self):
        items = list(self.items)
        if not items:
            return orig_group(self)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_group(self)
        e1.record()
        fl = sum(conv_flops(c.spec, x.shape[0], x.shape[1], x.shape[2]) for c, x, _, _ in items)
        lab = f"wgrad_kernel<{'BF16' if dtype == 'bf16' else 'F32'}, 9, 9, ..> (tg_wgrad_multi, {len(items)} layers)"
        recs.append((lab, fl, e0, e1))

        def again():
            for it in items:
                self.add(*it)
            orig_group(self)
        replays.setdefault(lab, []).append(again)

    K.resblock_fwd, K.resblock_bwd = rb_wrap(orig_rb, "resblock_kernel<false>"), rb_wrap(orig_rbb, "resblock_kernel<true>")
    E.WgradGroup.launch = group_timed
    E.Conv.fwd = timed(lab_fwd, fl_fwd, orig_fwd)
    E.Conv.dgrad = timed(lab_dgrad, fl_dgrad, orig_dgrad)
    E.Conv.wgrad = timed(lambda self, *a: f"wgrad_kernel<{'BF16' if dtype == 'bf16' else 'F32'}, {self.spec.nslots}, ..> + "
                         "wgrad_finalize_kernel", fl_wgrad, orig_wgrad)
    try:
        # park the GPU behind a ~60 ms spin kernel so that the host enqueues the whole eager step ahead of it: the event
        # pairs then bracket back-to-back kernel executions, not host launch gaps (eager launches are host-bound here).
        # _forward_backward runs both lanes of the step on ONE stream, in dependency order: isolated launch durations.
        torch.cuda.synchronize()
        torch.cuda._sleep(int(0.06 * 2.0e9))
        st._forward_backward(True)
        torch.cuda.synchronize()
    finally:
        E.Conv.fwd, E.Conv.dgrad, E.Conv.wgrad = orig_fwd, orig_dgrad, orig_wgrad
        K.resblock_fwd, K.resblock_bwd, E.WgradGroup.launch = orig_rb, orig_rbb, orig_group
    fam = {}
    for label, fl, e0, e1 in recs:
        d = fam.setdefault(label, dict(launches=0, flops=0.0, ms=0.0))
        d["launches"] += 1
        d["flops"] += fl
        d["ms"] += e0.elapsed_time(e1)
    # An event pair costs several microseconds of its own, which matters for the ~7 us recurrent-pass launches.  The
    # dominant family is therefore re-timed as ONE bracket around all of its launches (same arguments, back to back,
    # GPU parked first): that is the average duration rocprofv3 --kernel-trace reports, plus the inter-kernel gap.
    dom = max(fam, key=lambda k: fam[k]["ms"])
    torch.cuda.synchronize()
    torch.cuda._sleep(int(0.03 * 2.0e9))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for call in replays[dom]:
        call()
    e1.record()
    torch.cuda.synchronize()
    fam[dom]["ms_bracketed_once"] = e0.elapsed_time(e1)
    return fam


def usable_cores():
    """cores this process may actually use: min(affinity mask, cgroup v2 cpu.max quota) - the GPU box shows 256 logical
    CPUs but grants a 16-CPU quota, and 256 oneDNN threads on 16 CPUs are orders of magnitude slower than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are separate
    runs of this same command; they cannot be collected live).  gfx950 correction: FETCH_SIZE counts 64 B per 128-B request."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    try:
        tab = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None
    ent = tab.get(kernel)
    if not ent:
        return None, None
    return int(ent["hbm_bytes_per_launch"]), "profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, not live)"


def cpu_baseline(B, n_steps):
    """the oracle (CPU restatement of the reference step) on this box's host cores; same synthetic workload shape."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tecogan_oracle as orc
    threads = usable_cores()
    torch.set_num_threads(threads)
    args = orc.default_args()
    x, y = synth(B, 10, 32, 1)
    gp = orc.init_params(orc.generator_param_shapes(16), 101)
    dp = orc.init_params(orc.discriminator_param_shapes(4, 128), 201)
    bufs = orc.init_bn_buffers(dp)
    og = orc.AdamState(gp, 1e-4)
    od = orc.AdamState(dp, 1e-4)
    orc.tecogan_step(gp, dp, bufs, og, od, x, y, args, 0)  # warm-up
    t0 = time.perf_counter()
    for s in range(n_steps):
        orc.tecogan_step(gp, dp, bufs, og, od, x, y, args, s + 1)
    dt = (time.perf_counter() - t0) / n_steps
    return dict(value=round(B * 10 / dt, 3), unit="HR-frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n_steps} full train steps (B={B}, T=10, 32->128, fp32) after 1 warm-up, {dt:.2f} s/step, "
                       f"os.cpu_count()={os.cpu_count()}, usable (affinity/cgroup quota)={threads}")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)
    os.environ["TECOGAN_GRAPH"] = "0" if a.no_graph else "1"

    import pytorch_tecogan_amd  # noqa: F401
    from pytorch_tecogan_amd import models as M
    from pytorch_tecogan_amd import train as TR

    args = default_args(a.dtype)
    torch.manual_seed(1)
    G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
    og = torch.optim.Adam(G.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    od = torch.optim.Adam(D.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    B, T = a.batch, 10
    x, y = synth(B, T, 32, 1 + rank)  # every rank owns different sequences (weak scaling)
    x, y = x.to(dev), y.to(dev)       # inputs resident in HBM before the timed region

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"models built ({a.dtype}, B={B}/gpu, world={world}); warm-up {a.warmup} steps (step 0 eager, then capture)")
    for s in range(a.warmup):
        TR.FRVSR_Train(x, y, args, D, G, s, 0.0, 0.0, og, od)
        torch.cuda.synchronize()
        log(f"warm-up step {s} done")
    barrier()
    t0 = time.perf_counter()
    for s in range(a.steps):
        out = TR.FRVSR_Train(x, y, args, D, G, a.warmup + s, 0.0, 0.0, og, od)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    log(f"timed region done: {dt / a.steps * 1e3:.3f} ms/step")
    gen_loss, d_loss = float(out.gen_loss), float(out.d_loss)
    if not (np.isfinite(gen_loss) and np.isfinite(d_loss)):
        raise SystemExit(f"non-finite losses after the timed steps: {gen_loss} {d_loss}")

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * B * T * a.steps / dt
        res = {"metric": "HR frames/sec per train step, 4x 32->128 seq-10", "value": round(value, 2),
               "unit": "HR-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "configs[1]: full G+pseudo-flow/warp+D+losses+2xAdam train step, "
                                      f"B={B} sequences/GPU, T=10, 32x32->128x128", "global_batch": world * B,
                          "seq_len": T, "parallelism": f"dp{world}", "hipgraph": not a.no_graph},
               "step_tflops": round(world * B * STEP_GFLOP_PER_SEQ / 1e3 / (dt / a.steps), 2),
               "step_mfma_frac": round(B * STEP_GFLOP_PER_SEQ / 1e3 / (dt / a.steps) / MFMA_PEAK_TFLOPS[a.dtype], 5),
               "final_losses": {"gen_loss": round(gen_loss, 5), "d_loss": round(d_loss, 5)}}
        if not a.no_roofline:
            st = next(iter(TR._STEPS.values()))
            log("roofline pass (events around every MFMA launch of one eager step)")
            fam = roofline_pass(st, a.dtype)
            dom = max(fam, key=lambda k: fam[k]["ms"])
            d = fam[dom]
            dom_ms = d.get("ms_bracketed_once", d["ms"])
            ach = d["flops"] / (dom_ms * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic(dom)
            res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2),
                               "peak": MFMA_PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                               "frac": round(ach / MFMA_PEAK_TFLOPS[a.dtype], 5), "traffic": traffic,
                               "traffic_source": traffic_src,
                               "launches_per_step": d["launches"],
                               "avg_launch_us": round(dom_ms * 1e3 / d["launches"], 2),
                               "avg_launch_us_with_event_pair_per_launch": round(d["ms"] * 1e3 / d["launches"], 2),
                               "avg_launch_gflop": round(d["flops"] / d["launches"] / 1e9, 3),
                               "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                            for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}}
        if not a.no_cpu_baseline:
            log(f"cpu baseline: oracle, {a.cpu_steps}+1 steps on {usable_cores()} usable host cores")
            res["cpu_baseline"] = cpu_baseline(B, a.cpu_steps)
        print(json.dumps(res), flush=True)
    barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Throughput bench of the TecoGAN training step (BASELINE.json config 2: B=4 sequences per GPU, T=10, 32x32 -> 128x128,
bf16 compute / fp32 master weights, full G + pseudo-flow/warp + D + losses + two Adam updates per step).

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched once per rank by torch.distributed.run (RANK / WORLD_SIZE in the environment; WORLD_SIZE must equal
--gpus), or - with no RANK in the environment - this process starts the N ranks itself as children (before it touches the
GPU) and exits with their code.  It refuses to run when fewer than N GPUs are visible: a silent 1-GPU "dp1" line is never
printed for --gpus N.  --dry: the launch logic only (gloo rendezvous on the CPU, no GPU, no kernels) - tests/test_host_cpu.py.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      the dominant kernel family (every family of one eager step is replayed inside one event bracket), the
                per-family table, and `hbm_kernels`: the HBM-bound kernels as GB/s of algorithmic bytes
  cpu_baseline  the CPU oracle (PyTorch-CPU fp32 restatement of the reference) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts: pytorch-tecogan_amd/__init__.py says why

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
# BASELINE.json configs this bench can run: per-GPU batch, frames, LR crop, algorithmic GFLOP per sequence
#   2: configs[1] (the headline metric; configs[2] is the same per GPU on 8 GPUs)        10 G frames * 25.884 + 6 D samples * 20.196
#   4: configs[3] (64->256, seq-16, fp16 + loss scaling, 2 sequences per GPU; tg_extend)  16 * 103.54 + 10 * 80.78
WORKLOADS = {2: dict(batch=4, T=10, cs=32, dtype="bf16", gflop_per_seq=380.0, extend=False,
                     name="configs[1]: full G+pseudo-flow/warp+D+losses+2xAdam train step"),
             4: dict(batch=2, T=16, cs=64, dtype="fp16", gflop_per_seq=2464.4, extend=True,
                     name="configs[3]: same step at 64x64->256x256, seq-16, fp16 with dynamic loss scaling (tg_extend shapes)")}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS), help="BASELINE.json config number (see WORKLOADS)")
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (default: the config's)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"], help="default: the config's")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--dry", action="store_true", help="launch logic only: gloo rendezvous on the CPU, no GPU work")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip psnr_delta_db and other_configs (configs[3] shard, configs[4] inference) behind the headline line")
    ap.add_argument("--dp-mode", default=None, choices=["inline", "buckets", "both"],
                    help="data-parallel collectives (N > 1): one synchronous all-reduce per network on its lane's stream | two "
                         "asynchronous buckets per network | both measured, the better one is `value`, the other under dp.alt "
                         "(default: the package's, i.e. inline unless TECOGAN_DP_INLINE=0)")
    ap.add_argument("--dp-steps", type=int, default=10, help="steps of the two extra data-parallel passes (event-timed, no collectives)")
    return ap.parse_args(argv)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(a):
    """--gpus N without RANK in the environment: start N child ranks with torch.distributed.run.  Runs BEFORE anything
    in this process initialises the GPU (device_count() does not), and the children are fresh processes - a process
    that has touched the GPU is never re-executed."""
    import subprocess
    if not a.dry:
        vis = torch.cuda.device_count()
        if vis < a.gpus:
            raise SystemExit(f"bench.py: {a.gpus} GPUs requested, {vis} visible - refusing to print a {a.gpus}-GPU line")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(a, rank, world):
    """--dry: rendezvous (gloo, CPU), barrier, max-over-ranks timing and the line's parallelism fields - everything of the
    N-rank launch except the GPU work."""
    import torch.distributed as dist
    n = 1
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo")
        n = dist.get_world_size()
        if n != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but the process group has {n} ranks")
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == float(n)
    if rank == 0:
        line = {"dry": True, "n_gpus": n, "pg_world_size": n, "pg_backend": "gloo" if n > 1 or "RANK" in os.environ else None,
                "config": {"parallelism": f"dp{n}", "global_batch": n * WORKLOADS[a.config]["batch"]}}
        if n > 1 or "RANK" in os.environ:   # the data-parallel object's shape: the modes a real run would time, fields unfilled
            modes = ["buckets", "inline"] if a.dp_mode == "both" else [a.dp_mode or "inline"]
            blank = lambda m: {"mode": m, "requested_mode": m, "allreduce_exposed_ms_laneA": None,  # noqa: E731
                               "allreduce_exposed_ms_laneB": None, "step_ms_no_collectives": None, "probe_steps": a.dp_steps}
            line["dp"] = blank(modes[0])
            if len(modes) > 1:
                line["dp"]["alt"] = blank(modes[1])
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def default_args(dtype, T=10, cs=32, extend=False):
    return argparse.Namespace(RNN_N=T, crop_size=cs, tg_extend=extend, num_resblock=16, discrim_resblocks=4, discrim_channels=128,
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


TAG = {"bf16": "BF16", "fp16": "F16", "fp32": "F32"}


def kernel_name(conv, dtype):
    """the rocprofv3 kernel name of the launch `conv` just made (template parameters from the C library's launch plan)"""
    import ctypes
    from pytorch_tecogan_amd import _lib as L
    if conv.last_desc is None or conv.last_desc == "c4d":  # the sub-pixel launches (csrc/convt_mfma.hip)
        return f"subpixel_kernel<{TAG[dtype]}, {0 if conv.last_desc is None else 1}>"
    if conv.last_desc in ("c4s2", "ctd"):  # csrc/conv4s2_mfma.hip
        return f"conv_s2_gather_kernel<{TAG[dtype]}, {4 if conv.last_desc == 'c4s2' else 3}>"
    if conv.last_desc == "s2cw":  # csrc/conv_s2_cw.hip (register-weights stride-2 gathers: KS 4 = D's down-sampling convs, 3 = conv-transpose dgrad)
        return f"conv_s2_cw_kernel<{4 if conv.spec.kind == 'c4s2' else 3}, {conv.last_rw_nch}, ..>"   # (statistics / element type left open: pmc_traffic's prefix rule)
    if conv.last_desc == "rgb":  # csrc/conv_rgb.hip (the generator's output layer)
        return f"conv_rgb_kernel<{TAG[dtype]}>"
    if conv.last_desc == "ctcw":  # csrc/convt_cw.hip (conv-transpose forward, class-specialised waves)
        return f"convt_cw_kernel<{conv.last_rw_nch}, {TAG[dtype]}>"
    if conv.last_desc == "c4dcw":  # csrc/conv4s2d_cw.hip (input-gradient of the 4x4 stride-2 convs, class-specialised waves)
        return f"conv4s2d_cw_kernel<{conv.last_rw_nch}, {TAG[dtype]}>"
    if conv.last_desc == "c3cw":  # csrc/conv3_cw.hip (64 reduction channels, no statistics: eight equal waves)
        return f"conv3_cw_kernel<0, {TAG[dtype]}>"
    if conv.last_desc == "rw":  # csrc/conv3_rw.hip (NCH = input channels / 32; statistics variant not distinguished)
        return f"conv3_rw_kernel<{conv.last_rw_nch}, ..>"
    plan = L.load().tg_conv_pick_tile(ctypes.byref(conv.last_desc))
    t = TAG[dtype]
    return f"conv_gather_kernel<{t}, {TILE_PARAMS[plan & 255]}, {'true' if plan >> 8 else 'false'}>"


def roofline_pass(st, dtype):
    """One eager step (both lanes on ONE stream, in dependency order) with every MFMA launch and every HBM-bound launch
    recorded; then each family is replayed back to back inside ONE event bracket (GPU parked first, so the bracket holds
    kernel executions, not host launch gaps) - the per-launch average that rocprofv3 --kernel-trace reports for the same
    kernel.  Returns (mfma families, hbm families): {label: launches, work (flops | bytes), ms}."""
    from pytorch_tecogan_amd import engine as E
    from pytorch_tecogan_amd import kernels as K
    recs, replays = {}, {}
    saved = []

    def record(label, work, call, kind):
        d = recs.setdefault(label, dict(launches=0, work=0.0, kind=kind))
        d["launches"] += 1
        d["work"] += work
        replays.setdefault(label, []).append(call)

    def wrap(obj, name, label_fn, work_fn, kind):
        orig = getattr(obj, name)
        saved.append((obj, name, orig))

        def wrapper(*a, **kw):
            r = orig(*a, **kw)
            record(label_fn(*a, **kw), work_fn(*a, **kw), lambda: orig(*a, **kw), kind)  # label AFTER the call (last_desc)
            return r
        setattr(obj, name, wrapper)

    T16 = {"bf16": "BF16", "fp16": "F16"}.get(dtype, "F32")
    nb = lambda *ts: float(sum(t.numel() * t.element_size() for t in ts if t is not None))  # noqa: E731

    # ---- MFMA launches
    wrap(E.Conv, "fwd", lambda self, x, *a, **k: kernel_name(self, dtype),
         lambda self, x, *a, **k: conv_flops(self.spec, x.shape[0], x.shape[1], x.shape[2]), "mfma")
    wrap(E.Conv, "dgrad", lambda self, dout, out, *a, **k: kernel_name(self, dtype),
         lambda self, dout, out, *a, **k: conv_flops(self.spec, out.shape[0], out.shape[1], out.shape[2]), "mfma")
    wrap(E.Conv, "wgrad", lambda self, *a, **k: f"wgrad_kernel<{T16}, {self.spec.nslots}, ..>",
         lambda self, x_in, dout, *a, **k: conv_flops(self.spec, x_in.shape[0], x_in.shape[1], x_in.shape[2]), "mfma")
    rb_fl = lambda x, *a, **k: 2 * 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * 9 * x.shape[3] * x.shape[3]  # noqa: E731
    wrap(K, "resblock_fwd", lambda *a, **k: "resblock_ws_kernel" if k.get("ws") else "resblock_kernel<false>", rb_fl, "mfma")
    wrap(K, "resblock_bwd", lambda *a, **k: "resblock_kernel<true>", rb_fl, "mfma")
    wrap(K, "resblock2_fwd", lambda *a, **k: f"resblock2_kernel<{T16}>", lambda x, *a, **k: 2 * rb_fl(x), "mfma")  # two blocks
    def time_group(cls, label):
        orig_group = cls.launch
        saved.append((cls, "launch", orig_group))

        def group_timed(self, only=None):
            var = getattr(cls, "VARIANT", None)
            items = [it for it in self.items if only is None or var is None or var[it[0].spec.kind] in only]
            if only is None:
                orig_group(self)
            else:
                orig_group(self, only=only)
            if not items:
                return
            fl = sum(conv_flops(c.spec, x.shape[0], x.shape[1], x.shape[2]) for c, x, _, _ in items)
            cap = getattr(self, "cap", None)   # (the discriminator's halves run their lists under different caps: plans are per cap)

            def again():
                rest, self.items = self.items, list(items)
                keep = getattr(self, "cap", None)
                if cap is not None:
                    self.cap = cap
                orig_group(self)
                self.items = rest
                if cap is not None:
                    self.cap = keep
            record(label(items), fl, again, "mfma")
        cls.launch = group_timed
    time_group(E.WgradGroup, lambda items: f"wgrad_kernel<{T16}, 9, 9, ..> (tg_wgrad_multi, {len(items)} layers)")
    # the work-list launch (csrc/wgrad_group.hip): one rocprofv3 row per tile width; bracketed per call site
    time_group(E.WgradList, lambda items: f"wgrad_group_kernel<{T16}, ..> (tg_wgrad_group_v work lists: {len(items)} layers, "
                                          f"kinds {'+'.join(sorted({c.spec.kind for c, _, _, _ in items}))}, first {items[0][1].shape[1]}x{items[0][1].shape[2]})")

    # ---- HBM-bound launches: algorithmic bytes = every tensor the op must read + write once
    wrap(K, "bn_apply", lambda *a, **k: f"bn_apply_kernel<{T16}>",
         lambda z, stats, gamma, beta, y, *a, skip=None, **k: nb(z, y, skip), "hbm")
    wrap(K, "bn_bwd_reduce", lambda *a, **k: f"bn_bwd_reduce_kernel<{T16}>",
         lambda dy, yact, z, *a, **k: nb(dy, z, yact), "hbm")
    wrap(K, "bn_bwd_apply", lambda *a, **k: f"bn_bwd_apply_kernel<{T16}>",
         lambda dy, yact, z, save, red, gamma, dz, *a, **k: nb(dy, z, yact, dz), "hbm")
    wrap(K, "adam", lambda *a, **k: "adam_kernel", lambda p, *a, **k: 28.0 * p.numel(), "hbm")  # p,g,m,v read; p,m,v written
    wrap(K, "gen_input", lambda *a, **k: f"gen_input_kernel<{T16}>",
         lambda lr, lo, ls, prev, po, ps, grid, go, gs, dst, B, h, w: nb(dst) + 4.0 * B * h * w * (3 + (16 * 5 if prev is not None else 0)),
         "hbm")  # reads: LR frame, previous HR frame (3 ch) and the flow (2 ch) at 16 HR pixels per LR pixel
    wrap(K, "d_assemble", lambda *a, **k: f"d_assemble_kernel<{T16}>",
         lambda x, y, gen, tvel, dst, B, T, Kk, h, border, half=-1: nb(dst) + 4.0 * dst.shape[0] * 16 * h * h * (9 + 9 + 6) + 4.0 * dst.shape[0] * 9 * h * h,
         "hbm")  # reads: 9 target ch + 9 warped-source ch + 3 x 2 velocity ch per HR pixel, 9 LR ch
    wrap(K, "content_loss", lambda *a, **k: f"content_loss_kernel<{T16}>",
         lambda gen, y, dpre, *a, **k: nb(gen, y, dpre), "hbm")
    wrap(K, "conv3x3_rgb_bwd", lambda *a, **k: f"rgb_bwd_kernel<{T16}>",
         lambda dpre4, x, w, dx, slab, cap: nb(dpre4, x, dx), "hbm")  # output layer backward: x in, dx out, 8 B/pixel of dpre
    wrap(K, "up4_planes", lambda *a, **k: "up4_planes_kernel",
         lambda src, so, dst, do, n, h, w, **k: 4.0 * n * h * w * 17, "hbm")
    wrap(K, "absdiff_sum", lambda *a, **k: f"absdiff_sum_kernel<{T16}>", lambda a_, b_, *r, **k: nb(a_, b_), "hbm")
    wrap(K, "absdiff_sum_multi", lambda *a, **k: f"absdiff_sum_multi_kernel<{T16}>",
         lambda dt, jobs, n, *r, **k: float(sum(2 * int(row[3]) * int(row[5]) * (4 if dt == torch.float32 else 2) for row in jobs.cpu().tolist())),
         "hbm")

    def fold_bytes(self, only=None):
        jobs = []
        for c in (self.convs if only is None else only):
            if c.fin_job is not None:  # one fold job per conv, or one per 64 x 64 channel block (engine.WgradList)
                jobs += c.fin_job if isinstance(c.fin_job[0], list) else [c.fin_job]
        return float(sum(j[4] * j[11] * 4 + j[5] * j[8] * j[9] * 4 for j in jobs))  # slabs read + gradient written
    wrap(E.Finalizer, "run", lambda self, **k: "wgrad_fold_items_kernel" if self.fold_items else "wgrad_finalize_multi_kernel",
         lambda self, **k: fold_bytes(self, **k), "hbm")
    try:
        torch.cuda.synchronize()
        st._forward_backward(True)
        st._update_all()
        torch.cuda.synchronize()
    finally:
        for obj, name, orig in saved:
            setattr(obj, name, orig)
    for label, calls in replays.items():
        torch.cuda.synchronize()
        torch.cuda._sleep(int(0.008 * 2.0e9))  # park the GPU: the host enqueues the whole family ahead of it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for call in calls:
            call()
        e1.record()
        torch.cuda.synchronize()
        recs[label]["ms"] = e0.elapsed_time(e1)
    mf = {k: v for k, v in recs.items() if v["kind"] == "mfma"}
    hb = {k: v for k, v in recs.items() if v["kind"] == "hbm"}
    # The dominant family once more, this time IN THE STEP'S CONDITIONS: the other lane's real work (the discriminator's real
    # half, the piece that runs beside the generator chain) replays on lane B's stream while the family is bracketed on this
    # one.  A latency-bound chain of small launches pays for its neighbour ("what slows the chain"); the
    # stand-alone bracket above does not see that, rocprofv3's per-kernel average of the whole step does.
    dom = max(mf, key=lambda k: mf[k]["ms"])
    if st.graphs is not None and st.lanes and "d_real" in st.graphs:
        calls = replays[dom]
        torch.cuda.synchronize()
        neighbour_ms = 1.7                                       # d_real alone (tools/step_breakdown.py)
        reps = max(1, int(mf[dom]["ms"] * 1.5 / neighbour_ms) + 1)
        torch.cuda._sleep(int(0.008 * 2.0e9))
        with torch.cuda.stream(st.sBm):
            torch.cuda._sleep(int(0.008 * 2.0e9))
            for _ in range(reps):
                st.graphs["d_real"]()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for call in calls:
            call()
        e1.record()
        torch.cuda.synchronize()
        mf[dom]["ms_in_step"] = e0.elapsed_time(e1)
    return mf, hb


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


PMC_ROUND = 6
PMC_SUMMARY = f"r{PMC_ROUND:02d}_pmc_summary.json"


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are separate
    runs of this same command; they cannot be collected live).  gfx950 correction: FETCH_SIZE counts 64 B per 128-B request."""
    # ONLY the current round's file: an older round's counters under this round's kernel names (a renamed or rebuilt kernel) would be a
    # number about another binary - a label missing from the newest file reports no traffic instead
    for name in (PMC_SUMMARY,):
        try:
            tab = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        src = f"profiles/{name} (round {PMC_ROUND}; rocprofv3 --pmc, separate passes of this same command; not live)"
        if kernel in tab:
            return dict(tab[kernel], rocprof_names=[kernel]), src
        # a bench label leaves trailing template arguments open ("conv_gather_kernel<BF16, 4, 2, 1, 4, true>" covers the slim
        # and the general epilogue build): launch-weighted average over every profiled kernel the label is a prefix of
        stem = kernel.split(" (")[0].rstrip(">").rstrip(".").rstrip(", ")
        hits = {k: v for k, v in tab.items() if k.startswith(stem + ",") or k.startswith(stem + ">")}
        if hits:
            n = sum(v["launches"] for v in hits.values())
            ent = {"launches": n, "rocprof_names": sorted(hits)}
            for key in ("hbm_bytes_per_launch", "mfma_busy_pct", "avg_ns"):
                vals = [(v[key], v["launches"]) for v in hits.values() if key in v]
                if vals:
                    ent[key] = round(sum(a * b for a, b in vals) / sum(b for _, b in vals), 2)
            return ent, src
    return None, None


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


def roofline_object(st, dtype):
    """the `roofline` object of the bench line (contract: task description, part 4)"""
    fam, hbm = roofline_pass(st, dtype)
    dom = max(fam, key=lambda k: fam[k]["ms"])
    d = fam[dom]
    ach_alone = d["work"] / (d["ms"] * 1e-3) / 1e12
    # `achieved` / `frac` are quoted from the bracket taken beside the other lane's work (what rocprofv3 reports for the
    # kernel inside the overlapped step); the stand-alone bracket is kept as frac_standalone
    ach = d["work"] / (d.get("ms_in_step", d["ms"]) * 1e-3) / 1e12
    pmc, pmc_src = pmc_traffic(dom)

    def busy(label):  # MFMA-pipe busy share of the kernel's SQ busy cycles, from the committed counter passes
        e, _ = pmc_traffic(label)
        return e.get("mfma_busy_pct") if e else None
    return {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2),
                       "peak": MFMA_PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                       "frac": round(ach / MFMA_PEAK_TFLOPS[dtype], 5),
                       "traffic": int(pmc["hbm_bytes_per_launch"]) if pmc else None,
                       "traffic_source": pmc_src, "mfma_busy_pct": busy(dom),
                       "rocprof_names": pmc["rocprof_names"] if pmc else None,
                       "frac_standalone": round(ach_alone / MFMA_PEAK_TFLOPS[dtype], 5),
                       "achieved_standalone": round(ach_alone, 2),
                       "frac_basis": "family bracketed with HIP events on its stream while the discriminator's real half "
                                     "replays on the other lane (in-step conditions); *_standalone: the family alone",
                       "launches_per_step": d["launches"],
                       "avg_launch_us": round(d.get("ms_in_step", d["ms"]) * 1e3 / d["launches"], 2),
                       "avg_launch_us_standalone": round(d["ms"] * 1e3 / d["launches"], 2),
                       "avg_launch_gflop": round(d["work"] / d["launches"] / 1e9, 3),
                       # rocprofv3 lists the 9-tap weight-gradient kernel as ONE row (single-layer and grouped launches
                       # are the same instantiation); bracketed here as three families - their sum, for comparison
                       "also": (lambda ws: {"kernel": "all weight gradients: wgrad_group_kernel (work-list launches) + wgrad_kernel (the layers it does not take)",
                                            "launches_per_step": sum(v["launches"] for v in ws),
                                            "ms": round(sum(v["ms"] for v in ws), 3),
                                            "achieved": round(sum(v["work"] for v in ws) / (sum(v["ms"] for v in ws) * 1e-3) / 1e12, 2),
                                            "frac": round(sum(v["work"] for v in ws) / (sum(v["ms"] for v in ws) * 1e-3) / 1e12
                                                          / MFMA_PEAK_TFLOPS[dtype], 5)} if ws else None)(
                           [v for k, v in fam.items() if k.startswith("wgrad_kernel<") or k.startswith("wgrad_group_kernel<")]),
                       "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                        "tflops": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 2),
                                        "mfma_busy_pct": busy(k)}
                                    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])},
                       # the HBM-bound kernels of the step (reported separately, as GB/s of algorithmic bytes)
                       "hbm_kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                           "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2),
                                           "GBps": round(v["work"] / (v["ms"] * 1e-3) / 1e9, 1),
                                           "frac_of_8TBps": round(v["work"] / (v["ms"] * 1e-3) / 8e12, 4),
                                           # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE)
                                           # beside the algorithmic bytes per launch
                                           "algorithmic_bytes_per_launch": int(v["work"] / v["launches"]),
                                           "pmc_hbm_bytes_per_launch": (lambda e: int(e["hbm_bytes_per_launch"]) if e and
                                                                        "hbm_bytes_per_launch" in e else None)(pmc_traffic(k)[0])}
                                       for k, v in sorted(hbm.items(), key=lambda kv: -kv[1]["ms"])}}


def build_step_objects(args, dev):
    from pytorch_tecogan_amd import models as M
    G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
    og = torch.optim.Adam(G.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    od = torch.optim.Adam(D.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    return G, D, og, od


def timed_region(run_step, first, warmup, steps, world, dev, log=None):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks.
    Returns (seconds, result of the last step)."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
    for s in range(warmup):
        run_step(first + s)
        torch.cuda.synchronize()
        if log:
            log(f"warm-up step {s} done")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        out = run_step(first + warmup + s)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def dp_breakdown(st, run_step, first, steps, world, dev):
    """What a data-parallel step pays for its collectives, so that a scaling curve explains itself:
      allreduce_exposed_ms_laneA / _laneB  HIP events on each lane's stream: from the end of the lane's last backward piece to the
                                            point where its Adam may start (the all-reduce(s) issued or still pending there) - the
                                            part of the collectives that nothing hides; mean over `steps` steps, MAX over ranks
      step_ms_no_collectives                the same `steps` steps with every all-reduce skipped (replicas diverge: last pass)"""
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("A0", "A1", "B0", "B1")}
    st.dp_events = ev
    acc = {"A": 0.0, "B": 0.0}
    for s in range(steps):
        run_step(first + s)
        torch.cuda.synchronize()
        acc["A"] += ev["A0"].elapsed_time(ev["A1"])
        acc["B"] += ev["B0"].elapsed_time(ev["B1"])
    st.dp_events = None
    t = torch.tensor([acc["A"] / steps, acc["B"] / steps], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    st.skip_collectives = True
    dt, _ = timed_region(run_step, first + steps, 1, steps, world, dev)
    st.skip_collectives = False
    return {"allreduce_exposed_ms_laneA": round(float(t[0]), 4), "allreduce_exposed_ms_laneB": round(float(t[1]), 4),
            "step_ms_no_collectives": round(dt / steps * 1e3, 4), "probe_steps": steps,
            "grad_bytes": {"G": int(st.G.flat.g.numel() * 4), "D": int(st.D.flat.g.numel() * 4)}}


def psnr_delta(dev):
    """Second half of BASELINE.json's metric ("...; PSNR delta vs ref", north_star: within 0.05 dB): the bf16 HIP recurrent
    generator against the fp32 CPU oracle at the config-1 shape (1 sequence of 10 32x32 frames), compute_psnr
    (code/ops.py:130-139) on x255 outputs.  Working point of tests/test_bench_config_gpu.py::test_psnr_gate_bf16_vs_fp32_oracle_can_fail
    (where the gate is proven able to fail): the target is the output of a TEACHER generator (oracle init, conv weights x1.8: a
    structured output), the generator under test is the teacher with 10 % multiplicative weight noise - PSNR ~27 dB, the level
    TecoGAN reaches on real video, so a compute error of 1-2 % of the output range moves it.  The oracle is the checker here."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tecogan_oracle as orc
    from pytorch_tecogan_amd import models as M
    from pytorch_tecogan_amd import ops as O
    torch.set_num_threads(usable_cores())
    teacher = orc.init_params(orc.generator_param_shapes(16), 31)
    teacher = {k: (v * 1.8 if k.endswith(".weight") else v) for k, v in teacher.items()}
    x, _ = synth(1, 10, 32, 33)
    rng = np.random.default_rng(32)
    student = {k: v * torch.from_numpy(1.0 + 0.10 * rng.standard_normal(v.shape).astype(np.float32)) for k, v in teacher.items()}
    with torch.no_grad():
        y = orc.recurrent_generator(teacher, x, orc.pseudo_flow(x)).reshape(10, 3, 128, 128)
        ref = orc.recurrent_generator(student, x, orc.pseudo_flow(x)).reshape(10, 3, 128, 128)
    psnr_ref = float(orc.compute_psnr(ref * 255, y * 255))
    G = M.generator(3, default_args("bf16"))
    G.load_state_dict(student)
    out = G.to(dev).recurrent(x.to(dev), use_graph=False).cpu().reshape(10, 3, 128, 128)
    psnr_hip = float(O.compute_psnr(out * 255, y * 255))
    rel = float((out - ref).norm() / ref.norm())
    return {"psnr_delta_db": round(psnr_hip - psnr_ref, 5),
            "psnr": {"hip_bf16_db": round(psnr_hip, 4), "ref_fp32_oracle_db": round(psnr_ref, 4), "gate_db": 0.05,
                     "within_gate": abs(psnr_hip - psnr_ref) <= 0.05, "output_rel_err": round(rel, 6),
                     "working_point": "configs[0] shape (1 x 10 frames 32->128); target = teacher generator (oracle init seed 31, conv "
                                      "weights x1.8), generator under test = teacher x (1 + 0.10 N(0,1)) seed 32; compute_psnr on x255"}}


def other_config4(dev, steps, warmup, log):
    """BASELINE configs[3], per-GPU shard (B=2 of the global 16, T=16, 64->256, fp16 + dynamic loss scaling; args.tg_extend): the
    training step through FRVSR_Train, then the per-family brackets of one eager step."""
    from pytorch_tecogan_amd import train as TR
    wl = WORKLOADS[4]
    args = default_args(wl["dtype"], wl["T"], wl["cs"], wl["extend"])
    torch.manual_seed(1)
    G, D, og, od = build_step_objects(args, dev)
    B, T, cs = wl["batch"], wl["T"], wl["cs"]
    x, y = synth(B, T, cs, 1)
    x, y = x.to(dev), y.to(dev)
    run = lambda s: TR.FRVSR_Train(x, y, args, D, G, s, 0.0, 0.0, og, od)  # noqa: E731
    dt, out = timed_region(run, 0, warmup, steps, 1, dev)
    ms = dt / steps * 1e3
    res = {"workload": f"{wl['name']}, B={B} sequences/GPU (per-GPU shard of global 16), T={T}, {cs}x{cs}->{4 * cs}x{4 * cs}",
           "dtype": wl["dtype"], "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4),
           "hr_frames_per_s": round(B * T * steps / dt, 1), "step_tflops": round(B * wl["gflop_per_seq"] / 1e3 / (dt / steps), 2),
           "step_mfma_frac": round(B * wl["gflop_per_seq"] / 1e3 / (dt / steps) / MFMA_PEAK_TFLOPS[wl["dtype"]], 5),
           "finite": bool(np.isfinite(float(out.gen_loss)) and np.isfinite(float(out.d_loss)))}
    st = next(iter(TR._STEPS.values()))
    res["loss_scale"] = st.scaler_state()
    fam, _ = roofline_pass(st, wl["dtype"])
    dom = max(fam, key=lambda k: fam[k]["ms"])
    tf = lambda v, key="ms": v["work"] / (v[key] * 1e-3) / 1e12  # noqa: E731
    res["dominant_family"] = {"kernel": dom, "launches_per_step": fam[dom]["launches"], "ms": round(fam[dom]["ms"], 3),
                              "tflops": round(tf(fam[dom]), 1), "frac": round(tf(fam[dom]) / MFMA_PEAK_TFLOPS[wl["dtype"]], 5)}
    if "ms_in_step" in fam[dom]:
        res["dominant_family"]["frac_in_step"] = round(tf(fam[dom], "ms_in_step") / MFMA_PEAK_TFLOPS[wl["dtype"]], 5)
    res["families_top"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflops": round(tf(v), 1)}
                           for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:6]}
    for s_ in list(TR._STEPS.values()):
        s_.close()
    TR._STEPS.clear()
    return res


def other_config5(dev, frames, reps, log):
    """BASELINE configs[4]: generator-only recurrent inference, 1 sequence of `frames` 128x128 LR frames -> 512x512, per-frame
    hipGraph replay (main.py:171-219 of the reference: the loop of the inference caller), bf16."""
    from pytorch_tecogan_amd import models as M
    from pytorch_tecogan_amd import engine as E
    from pytorch_tecogan_amd import kernels as K
    lr = 128
    torch.manual_seed(1)
    G = M.generator(3, default_args("bf16", cs=lr)).to(dev)
    x = torch.from_numpy(np.random.default_rng(1).random((1, frames, 3, lr, lr), dtype=np.float32)).to(dev)
    out = G.recurrent(x, use_graph=True)   # warm-up: eager chunk, capture of the chunk graphs
    torch.cuda.synchronize()
    per_rep = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = G.recurrent(x, use_graph=True)
        torch.cuda.synchronize()
        per_rep.append(time.perf_counter() - t0)
    log("config5 sequences (ms): " + " ".join(f"{1e3 * t:.1f}" for t in per_rep))
    dt = sorted(per_rep)[len(per_rep) // 2]   # median sequence (each one is synchronised; the first may still pay allocator growth)
    gflop = 8.648 * (lr / 32) ** 2        # forward GFLOP per LR frame at 32x32, x16 at 128x128
    res = {"workload": f"configs[4]: generator-only recurrent inference, {lr}x{lr}->{4 * lr}x{4 * lr}, seq-{frames}, one hipGraph per chunk of {G._rec.FR} frames",
           "dtype": "bf16", "frames": frames, "reps": reps, "hr_frames_per_s": round(frames / dt, 1),
           "ms_per_frame": round(dt / frames * 1e3, 4), "tflops": round(gflop * frames / dt / 1e3, 1),
           "mfma_frac": round(gflop * frames / dt / 1e3 / MFMA_PEAK_TFLOPS["bf16"], 5), "finite": bool(torch.isfinite(out).all())}
    # the fused residual block (the trunk: 16 launches + conv_trans.2 per frame) bracketed over one eager frame's launches
    calls, fl = [], [0.0]
    orig = K.resblock_fwd

    ws_seen = []

    def rec(xa, *a, **k):
        r = orig(xa, *a, **k)
        ws_seen.append(bool(k.get("ws")))
        calls.append(lambda: orig(xa, *a, **k))
        fl[0] += 2 * 2.0 * xa.shape[0] * xa.shape[1] * xa.shape[2] * 9 * xa.shape[3] * xa.shape[3]
        return r
    K.resblock_fwd = rec
    try:
        G._rec._flows(1)
        G._rec._frame(1)   # (slot 1 of the staging ring: any frame but the sequence's first)
    finally:
        K.resblock_fwd = orig
    if calls:
        torch.cuda.synchronize()
        torch.cuda._sleep(int(0.008 * 2.0e9))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for c in calls:
            c()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        res["trunk_family"] = {"kernel": "resblock_ws_kernel" if all(ws_seen) else "resblock_kernel<false>", "launches_per_frame": len(calls), "ms_per_frame": round(ms, 4),
                               "tflops": round(fl[0] / (ms * 1e-3) / 1e12, 1),
                               "frac": round(fl[0] / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["bf16"], 5),
                               "share_of_frame": round(ms / (dt / frames * 1e3), 3)}
    G._rec.close()
    return res


def main(argv=None):
    a = parse(argv)
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))      # this process never touches the GPU; it exits with the ranks' code
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:   # a launcher that started another number of ranks than the line would claim
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if a.dry:
        return dry_run(a, rank, world)
    vis = torch.cuda.device_count()
    if vis < 1 or local >= vis:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local}, {vis} visible ({a.gpus} GPUs requested)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg_backend = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)
        pg_backend = dist.get_backend()
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but the process group has {dist.get_world_size()} ranks")
        world = dist.get_world_size()   # the line reports what the process group says, not the environment
    os.environ["TECOGAN_GRAPH"] = "0" if a.no_graph else "1"

    import pytorch_tecogan_amd  # noqa: F401
    from pytorch_tecogan_amd import train as TR

    wl = WORKLOADS[a.config]
    a.dtype = a.dtype or wl["dtype"]
    a.batch = a.batch or wl["batch"]
    args = default_args(a.dtype, wl["T"], wl["cs"], wl["extend"])
    torch.manual_seed(1)
    G, D, og, od = build_step_objects(args, dev)
    B, T, cs = a.batch, wl["T"], wl["cs"]
    STEP_GFLOP_PER_SEQ = wl["gflop_per_seq"]
    x, y = synth(B, T, cs, 1 + rank)  # every rank owns different sequences (weak scaling)
    x, y = x.to(dev), y.to(dev)       # inputs resident in HBM before the timed region

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    run_step = lambda s: TR.FRVSR_Train(x, y, args, D, G, s, 0.0, 0.0, og, od)  # noqa: E731
    # data-parallel collectives: the mode(s) to time.  A process group of one rank (torch.distributed.run with one process, or
    # TECOGAN_FORCE_COLLECTIVES=1) runs the same code path, so the fields below can be rehearsed on one GPU.
    from pytorch_tecogan_amd import parallel
    dp_live = parallel.dist_info()[0] is not None
    # "both": buckets first, so that the inline step (the default) is the one left for the roofline pass below.  (Until round 5 the
    # order mattered: a bucket-mode step built after an inline one replayed at 13 ms when the backend's first ASYNCHRONOUS collective
    # came after the first step's graphs; parallel.warm_backend() now uses that path once before the first step -
    # profiles/r05_c_pg_tax_probe.log.)
    modes = (["buckets", "inline"] if a.dp_mode == "both" else [a.dp_mode or "env"]) if dp_live else [None]
    runs, first = [], 0
    for mode in modes:
        if mode in ("inline", "buckets"):
            os.environ["TECOGAN_DP_INLINE"] = "1" if mode == "inline" else "0"
            for s_ in list(TR._STEPS.values()):   # a step is built for one mode
                s_.close()
            TR._STEPS.clear()
        log(f"models built ({a.dtype}, B={B}/gpu, world={world}, dp mode {mode}); warm-up {a.warmup} steps (step 0 eager, then capture)")
        dt, out = timed_region(run_step, first, a.warmup, a.steps, world, dev, log)
        first += a.warmup + a.steps
        st = next(iter(TR._STEPS.values()))
        rec = {"mode": mode, "dt": dt, "out": out}
        log(f"timed region done: {dt / a.steps * 1e3:.3f} ms/step")
        if mode is not None:
            rec["mode"] = "inline" if st.dp_inline else ("buckets" if st.buckets else "single")
            rec["requested_mode"] = mode
            rec["sync_allreduce_stream_ordered"] = st.dp_sync_ordered
            rec.update(dp_breakdown(st, run_step, first, a.dp_steps, world, dev))
            first += 2 * a.dp_steps + 1
        runs.append(rec)
    best = min(runs, key=lambda r: r["dt"])
    dt, out = best["dt"], best["out"]
    gen_loss, d_loss = float(out.gen_loss), float(out.d_loss)
    if not (np.isfinite(gen_loss) and np.isfinite(d_loss)):
        raise SystemExit(f"non-finite losses after the timed steps: {gen_loss} {d_loss}")

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * B * T * a.steps / dt
        res = {"metric": f"HR frames/sec per train step, 4x {cs}->{4 * cs} seq-{T}; PSNR delta vs ref", "value": round(value, 2),
               "unit": "HR-frames/s", "n_gpus": world, "pg_world_size": world if pg_backend else None,
               "pg_backend": pg_backend, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{wl['name']}, B={B} sequences/GPU, T={T}, {cs}x{cs}->{4 * cs}x{4 * cs}",
                          "global_batch": world * B,
                          "seq_len": T, "parallelism": f"dp{world}", "hipgraph": not a.no_graph},
               "step_tflops": round(world * B * STEP_GFLOP_PER_SEQ / 1e3 / (dt / a.steps), 2),
               "step_mfma_frac": round(B * STEP_GFLOP_PER_SEQ / 1e3 / (dt / a.steps) / MFMA_PEAK_TFLOPS[a.dtype], 5),
               "final_losses": {"gen_loss": round(gen_loss, 5), "d_loss": round(d_loss, 5)}}
        if best["mode"] is not None:
            keep = lambda r: {k: v for k, v in r.items() if k not in ("dt", "out")}  # noqa: E731
            res["dp"] = dict(keep(best), ms_per_step=round(best["dt"] / a.steps * 1e3, 4))
            alt = [r for r in runs if r is not best]
            if alt:
                res["dp"]["alt"] = dict(keep(alt[0]), ms_per_step=round(alt[0]["dt"] / a.steps * 1e3, 4))
        if not a.no_roofline:
            st = next(iter(TR._STEPS.values()))
            log("roofline pass (one eager step recorded, every kernel family replayed inside one event bracket)")
            res["roofline"] = roofline_object(st, a.dtype)
        if not a.no_cpu_baseline and a.config == 2:
            log(f"cpu baseline: oracle, {a.cpu_steps}+1 steps on {usable_cores()} usable host cores")
            res["cpu_baseline"] = cpu_baseline(B, a.cpu_steps)
        if world == 1 and not a.no_extras and a.config == 2:
            # the rest of BASELINE.json's metric and the configurations the headline line does not time, in this same process
            for s_ in list(TR._STEPS.values()):
                s_.close()
            TR._STEPS.clear()
            del G, D, og, od, run_step
            log("psnr_delta_db: bf16 HIP recurrent generator vs the fp32 oracle (checker leg, like cpu_baseline)")
            res.update(psnr_delta(dev))
            res["other_configs"] = {}
            log("other_configs.config4: configs[3] shard (B=2, T=16, 64->256, fp16), 10 steps")
            res["other_configs"]["config4"] = other_config4(dev, 10, 3, log)
            log("other_configs.config5: configs[4] inference (128->512, 120 frames, graph replay)")
            res["other_configs"]["config5"] = other_config5(dev, 120, 6, log)
        print(json.dumps(res), flush=True)
    barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""One TecoGAN training step (code/train.py:49-370) as a static sequence of HIP kernel launches on preallocated
buffers - which is what makes it hipGraph-capturable - plus the generator-only recurrent inference loop
(main.py:171-219).  See SURVEY.md Appendix B for the behavioural spec this follows, quirks included:
  * pseudo-flow = bilinear x4 of the previous LR frame's first two channels, reinterpreted (not permuted) as a grid;
  * HR / fake warp grids rounded to fp16, real / LR warp grids fp32;
  * gen_flow_back built from rows 0..B-1 of a (3B,6,h,h) reshape (mixes batch elements);
  * fake D input reuses the TARGET frames as its first 9 channels; both D inputs detached from G;
  * reported gen_loss = content + 2*ratio*t_adv + layer_sum*dt_ratio (aliased tensor), gradient of G = content only;
  * BN running statistics updated twice per step (real pass, then fake pass).
Data parallel: sequences are sharded over ranks; the flat G and D gradient buffers are all-reduced over RCCL, the G
all-reduce overlapping the D backward pass."""
import os

import torch

from . import _lib as L
from . import kernels as K
from . import parallel
from .kernels import pad32

LAYER_NORM = (12.0, 14.0, 24.0, 100.0)  # code/train.py:214

SCALAR_NAMES = ["D_layer_0_loss", "D_layer_1_loss", "D_layer_2_loss", "D_layer_3_loss", "D_layer_loss_sum",
                "gen_loss", "l2_warp_loss", "t_adversarial_loss", "t_discrim_loss", "t_discrim_real_output",
                "t_discrim_fake_output", "content", "t_balance", "PingPang"]


def _i64(vals, device):
    return torch.tensor(vals, dtype=torch.int64, device=device)


def build_tables(B, T, h, K, pingpang=False, fnet_flow=False):
    """Element-offset tables that drive tg_up4_planes / tg_copy_blocks / tg_warp_nchw (pure host logic, unit-tested on CPU).
    x is (B,T,3,h,h), flow (B,T-1,2,H,H), T_vel (B*3K,H,H,2) == blocks of 2*H*H floats."""
    H = 4 * h
    hh, HH = h * h, H * H
    tsize = 3 * K
    xo = lambda b, t, c=0: ((b * T + t) * 3 + c) * hh
    out = {}
    # pseudo-flow planes: x[b,t,c] (t<T-1, c<2) -> flow[b,t,c]   (code/train.py:71-77)
    # (fnet_flow: the planes come from the f_net output of all B*T frames, 2 channels per frame)
    src, dst = [], []
    for b in range(B):
        for t in range(T - 1):
            for c in range(2):
                src.append(((b * T + t) * 2 + c) * hh if fnet_flow else xo(b, t, c))
                dst.append(((b * (T - 1) + t) * 2 + c) * HH)
    out["flow_src"], out["flow_dst"] = src, dst
    # LR warp (logged loss only): img x[b,t], grid block = x[b,t+1,0:2], reference x[b,t+1]  (code/train.py:78-84,247-249)
    img, grd = [], []
    for b in range(B):
        for t in range(T - 1):
            img.append(xo(b, t))
            grd.append(xo(b, t + 1))
    out["lrw_img"], out["lrw_grid"] = img, grd
    # T_vel blocks (code/train.py:138-158): (b, j, r) -> r=0 flow[b,3j], r=1 zeros, r=2 2*up4(4*back)-1
    csrc, cdst = [], []
    for b in range(B):
        for j in range(K):
            csrc += [((b * (T - 1) + 3 * j) * 2) * HH, -1]
            cdst += [((b * tsize + 3 * j) * 2) * HH, ((b * tsize + 3 * j + 1) * 2) * HH]
    out["tv_csrc"], out["tv_cdst"] = csrc, cdst
    # "back" planes: cat(x[:,2:ts:3], x[:,1:ts:3], dim=1) is (B,2K,3,h,h); the reference views it as (B*K,6,h,h) and keeps
    # rows 0..B-1 only, i.e. the FIRST 6B planes of the flattened tensor (for K=3 that mixes batch elements).
    frames = list(range(2, tsize, 3)) + list(range(1, tsize, 3))
    bsrc, bdst = [], []
    if pingpang:  # VNxt = flip(flow, time)[:, 1:ts:3], raw (code/train.py:154): plain block copies, no "back" planes
        for b in range(B):
            for j in range(K):
                csrc.append(((b * (T - 1) + (T - 2 - (1 + 3 * j))) * 2) * HH)
                cdst.append(((b * tsize + 3 * j + 2) * 2) * HH)
        out["tv_csrc"], out["tv_cdst"] = csrc, cdst
    for b in range(B if not pingpang else 0):
        for j in range(K):
            for comp in range(2):
                P = 2 * K * b + 2 * j + comp
                bb, f, c = P // (2 * K * 3), (P // 3) % (2 * K), P % 3
                bsrc.append(xo(bb, frames[f], c))
                bdst.append((((b * tsize + 3 * j + 2) * 2) + comp) * HH)
    out["tv_bsrc"], out["tv_bdst"] = bsrc, bdst
    return out


class TecoGANStep:
    def __init__(self, G, D, B, T, h, args, device, use_graph=False, process_group=None, world_size=1):
        """G: GeneratorEngine, D: DiscriminatorEngine (already bound to flat parameter buffers on `device`)."""
        if not getattr(args, "Dt_mergeDs", True):
            raise RuntimeError("Dt_mergeDs=False feeds 9 channels into a 27-channel conv in the reference and raises "
                               "there too (SURVEY.md 8a8)")
        if float(getattr(args, "vgg_scaling", -1.0)) > 0.0:
            raise NotImplementedError("vgg_scaling>0 crashes in the reference (code/train.py:126 vs :30); not built")
        # ping-pong (code/train.py:56-62): the step runs on x followed by reverse(x)[1:], i.e. 2T-1 frames
        self.pingpang = bool(getattr(args, "pingpang", False))
        self.T_in = T
        if self.pingpang:
            T = 2 * T - 1
        self.G, self.D, self.B, self.T, self.h, self.args, self.dev = G, D, B, T, h, args, device
        self.H = H = 4 * h
        self.K = T // 3
        if self.K < 1:
            raise ValueError("RNN_N must be >= 3")
        self.tsize = 3 * self.K
        self.tb = B * self.K
        self.use_graph, self.pg, self.world = use_graph, process_group, world_size
        f32 = dict(dtype=torch.float32, device=device)
        self.x = torch.empty(B, T, 3, h, h, **f32)
        self.y = torch.empty(B, T, 3, H, H, **f32)
        self.gen = torch.empty(B, T, 3, H, H, **f32)
        self.flow = torch.empty(B, T - 1, 2, H, H, **f32)
        self.tvel = torch.empty(B * self.tsize, H, H, 2, **f32)
        self.target = torch.empty(self.tb, 27, H, H, **f32)
        self.acc = torch.zeros(16, **f32)
        self.scalars = torch.zeros(48, **f32)
        # per-step host parameters (loss config, Adam bias corrections, lr) travel through ONE small async copy from a
        # ring of pinned slots, so the CPU may run many steps ahead of the GPU without overwriting a pending copy
        self.params_dev = torch.zeros(32, **f32)
        self.cfg = self.params_dev[0:16]
        self.hyper = self.params_dev[16:32].view(2, 8)
        self.ring = torch.zeros(256, 32, dtype=torch.float32).pin_memory()
        self.ring_i = 0
        G.alloc(T * B, h, h)
        # Frame-chunked G backward overlapping the rest of the chain was measured SLOWER (13.8 / 14.8 / 16.8 ms per step at
        # 1 / 2 / 5 chunks): the dense backward launches hold every CU's LDS, so the chain's latency-critical launches
        # queue behind them.  Default: one chunk, after the chain.
        nchunk = max(1, min(T, int(os.environ.get("TECOGAN_GBWD_CHUNKS", "1"))))
        bounds = [round(i * T / nchunk) for i in range(nchunk + 1)]
        self.chunks = [(bounds[i], bounds[i + 1]) for i in range(nchunk)]
        cmax = max(t1 - t0 for t0, t1 in self.chunks) * B
        G._alloc_grad(cmax)
        if nchunk > 1:
            G.side.streams = []  # chunked backward runs on a forked stream: no nested joins under capture
            if G.finalizer is not None:  # every chunk reuses the slabs: fold per conv
                G.finalizer.disable()
                G.finalizer = None
        # one d(pre-sigmoid) buffer per chunk: chunk i+1's loss kernel must not overwrite what chunk i's backward reads
        self.dpre = [torch.empty((t1 - t0) * B, H, H, 32, dtype=G.dt, device=device) for t0, t1 in self.chunks]
        self.sB, self.sC = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
        self.dreal_early = os.environ.get("TECOGAN_DREAL_EARLY", "1") != "0"
        D.alloc(2 * self.tb, H)
        self._tables()
        self.graphs = None
        self.adam_t = [0, 0]
        self.border = (H - int(H * args.crop_dt)) // 2 if args.crop_dt < 1.0 else 0

    # ----------------------------------------------------------------------------------------------------------
    def _tables(self):
        # opt-in (not reference behaviour): args.tg_fnet = an f_net module -> flow = up4(4 * f_net(previous LR frame))
        fn = getattr(self.args, "tg_fnet", None)
        self.F = fn.engine(self.G.dt) if fn is not None else None
        if self.F is not None:
            self.F.alloc(self.B * self.T, self.h, self.h)
            self.fx = torch.empty(self.B * self.T, 2, self.h, self.h, dtype=torch.float32, device=self.dev)
        t = build_tables(self.B, self.T, self.h, self.K, self.pingpang, fnet_flow=self.F is not None)
        dev = self.dev
        self.flow_src, self.flow_dst, self.n_flow = _i64(t["flow_src"], dev), _i64(t["flow_dst"], dev), len(t["flow_src"])
        self.lrw_img, self.lrw_grid = _i64(t["lrw_img"], dev), _i64(t["lrw_grid"], dev)
        self.tv_csrc, self.tv_cdst, self.n_tvc = _i64(t["tv_csrc"], dev), _i64(t["tv_cdst"], dev), len(t["tv_csrc"])
        self.tv_bsrc, self.tv_bdst, self.n_tvb = _i64(t["tv_bsrc"], dev), _i64(t["tv_bdst"], dev), len(t["tv_bsrc"])

    # ----------------------------------------------------------------------------------------------------------
    def _host_params(self, global_step, lr_g, lr_d, betas_g, betas_d, eps_g, eps_d):
        a = self.args
        B, T, h, H = self.B, self.T, self.h, self.H
        slot = self.ring[self.ring_i % self.ring.shape[0]]
        self.ring_i += 1
        c = [0.0] * 32
        c[0] = B * T * 3 * H                      # content: mean over (BT,3,H) of sum over W
        c[1] = B * (T - 1) * 3 * h                # warp loss
        for i, l in enumerate(self.D.layers()):
            c[2 + i] = self.tb * l.shape[3] * l.shape[1]  # mean over (tb, C, Hl) of sum over W
            c[12 + i] = LAYER_NORM[i]
        c[6], c[7] = a.EPS, a.ratio
        c[8] = min(a.Dt_ratio_max, a.Dt_ratio_0 + a.Dt_ratio_add * float(global_step))
        c[9] = 1.0 if a.D_LAYERLOSS else 0.0
        c[10] = float(B * (self.T_in - 1) * 3 * H * H) if self.pingpang else 0.0
        c[11] = a.pp_scaling
        self.dt_ratio = c[8]
        gs = 1.0 / self.world
        c[16:24] = K.adam_hyper(lr_g, betas_g[0], betas_g[1], eps_g, self.adam_t[0] + 1, gs)
        c[24:32] = K.adam_hyper(lr_d, betas_d[0], betas_d[1], eps_d, self.adam_t[1] + 1, gs)
        slot.copy_(torch.tensor(c, dtype=torch.float32))
        self.params_dev.copy_(slot, non_blocking=True)

    # ----------------------------------------------------------------------------------------------------------
    def _forward_backward(self, include_d_backward, parts=None):
        """Everything up to (and including) the backward passes, as a fork/join over three streams so that the serial
        generator chain (<= 64 workgroups per launch at B=4) shares the chip with independent dense work:
            main : pseudo-flow, T_vel | G pass 0 .. T-1 (each: warp+pack, 41 convs) | content loss per frame chunk
            sB   : D input (real) -> D forward (real half)  ...............| D input (fake) -> D forward (fake half)
                   -> layer losses -> loss scalars, d(logit) -> [D backward when single-GPU]
            sC   : G backward of frame chunk 0 (while the chain is still producing chunk 1), then chunk 1, ...
        Captured in a hipGraph the fork/join become graph edges."""
        G, D, B, T, h, H = self.G, self.D, self.B, self.T, self.h, self.H
        hh, HH = h * h, H * H
        main, sB, sC = torch.cuda.current_stream(), self.sB, self.sC
        on = (lambda name: True) if parts is None else (lambda name: name in parts)  # tools/step_breakdown.py only
        G.side.prefork(main)
        D.side.prefork(main)
        self.acc.zero_()
        D.arena.zero()
        G.flat.g.zero_()
        D.flat.g.zero_()
        if self.F is not None:  # the estimator sees every LR frame (the last frame of a sequence is computed but unused)
            K.nchw_to_nhwc(self.x, 3 * hh, self.F.act["in"], B * T, 3, h, h)
            self.F.forward(self.fx)
            K.up4_planes(self.fx, self.flow_src, self.flow, self.flow_dst, self.n_flow, h, h, pre=4.0)
        else:
            K.up4_planes(self.x, self.flow_src, self.flow, self.flow_dst, self.n_flow, h, h, pre=4.0)
        K.warp_nchw(self.x, self.lrw_img, self.x, self.lrw_grid, B * (T - 1), 3, h, h, h, h, False, sq_ref=self.x,
                    sq_off=self.lrw_grid, loss_acc=self.acc[1:2])
        K.copy_blocks(self.flow, self.tv_csrc, self.tvel, self.tv_cdst, self.n_tvc, 2 * HH)
        if self.n_tvb:
            K.up4_planes(self.x, self.tv_bsrc, self.tvel, self.tv_bdst, self.n_tvb, h, h, pre=4.0, post_a=2.0, post_b=-1.0)
        tb = self.tb
        pp_T = self.T_in if self.pingpang else 0
        pp_coef = (2.0 * self.args.pp_scaling / (B * (self.T_in - 1) * 3 * H * H)) if (self.pingpang and self.args.pp_scaling > 0) else 0.0
        early = self.dreal_early
        sB.wait_stream(main)
        with torch.cuda.stream(sB):
            if on("dreal") and early:
                K.d_assemble(self.x, self.y, self.gen, self.tvel, D.act["in"][:tb], B, T, self.K, h, self.border, half=0)
                K.nhwc_to_nchw(D.act["in"][:tb], self.target, 27 * HH, tb, 27, H, H)
                D.forward(update_stats=True, half=0)
        ci = 0
        for t in range(T if on("chain") else 0):
            dst = G.act["in0"][t * B:(t + 1) * B]
            if t == 0:
                K.gen_input(self.x, 0, T * 3 * hh, None, 0, 0, None, 0, 0, dst, B, h, h)
            else:
                K.gen_input(self.x, t * 3 * hh, T * 3 * hh, self.gen, (t - 1) * 3 * HH, T * 3 * HH, self.flow,
                            (t - 1) * 2 * HH, (T - 1) * 2 * HH, dst, B, h, h)
            G.forward(t * B, B, self.gen, t * 3 * HH, T * 3 * HH)
            if t + 1 == self.chunks[ci][1] and on("gbwd"):
                t0, t1 = self.chunks[ci]
                K.content_loss(self.gen, self.y, self.dpre[ci], self.acc, B, T, H, H, 1.0 / (B * T * 3 * H), t0, t1, pp_T,
                               pp_coef)
                if len(self.chunks) > 1:  # experimental frame-chunked backward beside the chain (measured slower)
                    sC.wait_stream(main)
                    with torch.cuda.stream(sC):
                        G.backward(t0 * B, t1 * B, dpre=self.dpre[ci])
                ci += 1
        sB.wait_stream(main)  # all frames generated, content-loss sum complete
        if len(self.chunks) == 1 and on("gbwd") and on("chain"):
            # the chain is over, so the main (origin) stream carries the G backward's dgrad chain; its weight gradients
            # fan out to G.side and join back into main
            G.backward(0, T * B, dpre=self.dpre[0])
        with torch.cuda.stream(sB):
            if on("dfake") and early:
                K.d_assemble(self.x, self.y, self.gen, self.tvel, D.act["in"][tb:], B, T, self.K, h, self.border, half=1)
                D.forward(update_stats=True, half=1)
            elif on("dfake"):  # both halves as ONE batch of 2*tb samples (BN statistics per half), after the chain
                K.d_assemble(self.x, self.y, self.gen, self.tvel, D.act["in"], B, T, self.K, h, self.border, half=-1)
                K.nhwc_to_nchw(D.act["in"][:tb], self.target, 27 * HH, tb, 27, H, H)
                D.forward(groups=2, update_stats=True)
            if self.args.D_LAYERLOSS and on("dfake"):
                for i, l in enumerate(D.layers()):
                    n = tb * l.shape[1] * l.shape[2]
                    K.absdiff_sum(l[:tb], l[tb:], self.acc, 2 + i, n, l.shape[3], l.shape[3])
            K.loss_finalize(D.prob, self.acc, self.scalars, D.dlogit, tb, self.cfg)
            if include_d_backward and on("dbwd"):
                D.backward(groups=2, join=False)
        main.wait_stream(sB)
        main.wait_stream(sC)
        D.side.join(main)
        G.cout.gbias[:3] += self.acc[8:11]

    def _d_backward(self):
        self.D.backward(groups=2)

    def _update(self):
        G, D = self.G, self.D
        K.adam(G.flat.p, G.flat.g, G.flat.m, G.flat.v, self.hyper[0])
        K.adam(D.flat.p, D.flat.g, D.flat.m, D.flat.v, self.hyper[1])
        G.repack()
        D.repack()

    def _allreduce(self, buf):
        if self.pg is not None and os.environ.get("TECOGAN_FORCE_DP_SEGMENTS", "0") == "1":
            import torch.distributed as dist  # test hook: exercise the RCCL call path even with one rank
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return parallel.allreduce_sum_async(buf, self.pg, self.world)

    def _segments(self):
        """[forward + both backward passes | update].  Data parallel: both flat gradient buffers are all-reduced (RCCL, async,
        concurrently) between the two segments.  A three-segment variant that overlapped the G all-reduce with a separate
        D-backward segment was measured 0.9 ms/step slower on one GPU (it gives up the G-backward / D-backward overlap to hide
        a 7 MB all-reduce of ~0.1 ms), so it is not used."""
        return [lambda: self._forward_backward(True), None, self._update]

    def _run(self, segs):
        segs[0]()
        w1 = self._allreduce(self.G.flat.g)
        if segs[1] is not None:
            segs[1]()
        w2 = self._allreduce(self.D.flat.g)
        for w in (w1, w2):
            if w is not None:
                w.wait()
        segs[2]()

    def _capture(self):
        pool = torch.cuda.graph_pool_handle()
        self.G.ws.frozen = self.D.ws.frozen = True
        graphs = []
        for fn in self._segments():
            if fn is None:
                graphs.append(None)
                continue
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread polls its work events (hipEventQuery) while this thread captures;
            # under the default global mode that aborts with hipErrorStreamCaptureUnsupported
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                fn()
            graphs.append(g.replay)
        self.graphs = graphs

    # ----------------------------------------------------------------------------------------------------------
    def run(self, x, y, global_step, lr_g, lr_d, betas_g=(0.9, 0.999), betas_d=(0.9, 0.999), eps_g=1e-8, eps_d=1e-8):
        """x (B,T,3,h,h), y (B,T,3,H,H) fp32 device tensors.  Returns nothing; results live in self.gen / self.scalars /
        self.target and the parameter / optimiser buffers are updated in place."""
        Ti = self.T_in
        if tuple(x.shape) != (self.B, Ti, 3, self.h, self.h) or tuple(y.shape) != (self.B, Ti, 3, self.H, self.H):
            raise ValueError(f"step built for B={self.B}, T={Ti}, crop {self.h}; got {tuple(x.shape)} / {tuple(y.shape)}")
        self.x[:, :Ti].copy_(x)
        self.y[:, :Ti].copy_(y)
        if self.pingpang:  # reverse(x)[1:] appended (data movement only)
            self.x[:, Ti:].copy_(torch.flip(x, dims=[1])[:, 1:])
            self.y[:, Ti:].copy_(torch.flip(y, dims=[1])[:, 1:])
        self._host_params(global_step + 1, lr_g, lr_d, betas_g, betas_d, eps_g, eps_d)
        if self.use_graph:
            if self.graphs is None:
                self._run(self._segments())   # warm-up: one-time attribute setup, workspace growth
                self.adam_t = [self.adam_t[0] + 1, self.adam_t[1] + 1]
                torch.cuda.synchronize()
                self._capture()
                return
            self._run(self.graphs)
        else:
            self._run(self._segments())
        self.adam_t = [self.adam_t[0] + 1, self.adam_t[1] + 1]


class RecurrentGenerator:
    """Generator-only recurrent inference (main.py:171-219) without the per-frame CPU<->GPU bounces; the per-frame
    step (warp -> pack -> G) can be captured once as a hipGraph and replayed for every frame."""

    def __init__(self, G, B, h, w, device, use_graph=False):
        self.G, self.B, self.h, self.w, self.dev, self.use_graph = G, B, h, w, device, use_graph
        H, W = 4 * h, 4 * w
        f32 = dict(dtype=torch.float32, device=device)
        self.lr = torch.empty(B, 3, h, w, **f32)
        self.prev_lr = torch.empty(B, 3, h, w, **f32)
        self.prev = torch.zeros(B, 3, H, W, **f32)
        self.out = torch.empty(B, 3, H, W, **f32)
        self.flow = torch.empty(B, 2, H, W, **f32)
        hh, HH = h * w, H * W
        src, dst = [], []
        for b in range(B):
            for c in range(2):
                src.append((b * 3 + c) * hh)
                dst.append((b * 2 + c) * HH)
        self.fsrc, self.fdst = _i64(src, device), _i64(dst, device)
        G.alloc(B, h, w)
        self.graph = None

    def _frame(self):
        G, B, h, w = self.G, self.B, self.h, self.w
        H, W = 4 * h, 4 * w
        K.up4_planes(self.prev_lr, self.fsrc, self.flow, self.fdst, 2 * B, h, w, pre=4.0)
        K.gen_input(self.lr, 0, 3 * h * w, self.prev, 0, 3 * H * W, self.flow, 0, 2 * H * W, G.act["in0"], B, h, w)
        G.forward(0, B, self.out, 0, 3 * H * W)

    def run(self, frames):
        """frames (B,T,3,h,w) fp32 device -> (B,T,3,4h,4w)."""
        B, T = frames.shape[:2]
        h, w = self.h, self.w
        outs = torch.empty(B, T, 3, 4 * h, 4 * w, dtype=torch.float32, device=self.dev)
        self.lr.copy_(frames[:, 0])
        K.gen_input(self.lr, 0, 3 * h * w, None, 0, 0, None, 0, 0, self.G.act["in0"], B, h, w)
        self.G.forward(0, B, self.out, 0, 3 * 16 * h * w)
        outs[:, 0].copy_(self.out)
        for t in range(1, T):
            self.prev.copy_(self.out)
            self.prev_lr.copy_(frames[:, t - 1])
            self.lr.copy_(frames[:, t])
            if self.use_graph:
                if self.graph is None:
                    self._frame()
                    torch.cuda.synchronize()
                    self.graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                        self._frame()
                self.graph.replay()
            else:
                self._frame()
            outs[:, t].copy_(self.out)
        return outs

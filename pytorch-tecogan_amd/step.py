"""One TecoGAN training step (code/train.py:49-370) as a static sequence of HIP kernel launches on preallocated
buffers - which is what makes it hipGraph-capturable - plus the generator-only recurrent inference loop
(main.py:171-219).  See SURVEY.md Appendix B for the behavioural spec this follows, quirks included:
  * pseudo-flow = bilinear x4 of the previous LR frame's first two channels, reinterpreted (not permuted) as a grid;
  * HR / fake warp grids rounded to fp16, real / LR warp grids fp32;
  * gen_flow_back built from rows 0..B-1 of a (3B,6,h,h) reshape (mixes batch elements);
  * fake D input reuses the TARGET frames as its first 9 channels; both D inputs detached from G;
  * reported gen_loss = content + 2*ratio*t_adv + layer_sum*dt_ratio (aliased tensor), gradient of G = content only;
  * BN running statistics updated twice per step (real pass, then fake pass).
Data parallel: sequences are sharded over ranks; the flat G and D gradient buffers are all-reduced over RCCL, each
issued where its inputs become final - the D all-reduce behind lane B's last backward launch, the G all-reduce behind
lane A's (TecoGANStep._run_lanes)."""
import torch

from . import _lib as L
from . import kernels as K
from . import parallel
from . import tuning
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


_LANE_STREAMS = {}


def lane_b_stream(device):
    """lane B's stream: ONE per device for the life of the process.  A step that replaces another (train.get_step: another batch
    size, another data-parallel mode) reuses it - hardware queues are few (GPU_MAX_HW_QUEUES) and the runtime deals streams onto
    them round-robin, so every further stream raises the odds that the two lanes of the live step share a queue and serialise."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    s = _LANE_STREAMS.get(key)
    if s is None:
        s = _LANE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


def lane_stream(device, reserve_cus):
    """stream of the dense lane: confined to every CU except the first `reserve_cus` mask bits (tg_stream_create_cumask),
    or an ordinary stream when reserve_cus == 0"""
    if reserve_cus <= 0:
        return torch.cuda.Stream(device=device)
    import ctypes
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        L.check(L.load().tg_stream_create_cumask(int(reserve_cus), ctypes.byref(h)), "tg_stream_create_cumask")
    return torch.cuda.ExternalStream(h.value, device=device)


class TecoGANStep:
    """The step as two LANES of launches (each a linear sequence, so each is one hipGraph replayed on its own stream):

        lane A (current stream): prep (zeroing, pseudo-flow, T_vel) | generator chain, T passes (the pass(es) D never sees
                                 - frame 9 of 10 - after lane B's fake half has been released) | G backward (T*B samples)
        lane B (stream sB)     :          D input (real), D forward + backward of the REAL half | D input (fake), D forward,
                                          layer losses, loss scalars, D backward of the FAKE half
        update (lane A)        : [data parallel: all-reduce G grads after lane A, D grads after lane B] 2 x Adam, repack

    The chain is 420 dependent launches of 64-256 workgroups; what slows it down when dense work shares the chip is
    QUEUEING: its launches need CUs with free LDS, and a dense launch holds every CU with workgroups that run for tens of
    microseconds.  Stream priorities do nothing on MI355X and a CU mask does not survive inside one forked graph capture
    (profiles/r02_a_overlap_probe_priority_cumask.log), hence one graph per lane; lane B's stream can be masked off the
    first TECOGAN_CU_RESERVE CUs so that the chain's residual-block launches (64 workgroups) always start at once.  Masking
    all of lane B lost more than it gained (profiles/r02_b_lane_matrix.log: the fake half runs beside the dense G backward
    and needs the whole chip), so only the REAL half - the part that runs beside the chain - goes on the masked stream.
    The real half of the discriminator does not depend on the generator at all (code/train.py:160-185,199-203,304-307:
    separate BN statistics, loss = mean of per-half terms), so its forward AND backward run beside the chain; only the
    fake half is left for the time after the chain, beside the G backward."""

    def __init__(self, G, D, B, T, h, args, device, use_graph=False, process_group=None, world_size=1):
        """G: GeneratorEngine, D: DiscriminatorEngine (already bound to flat parameter buffers on `device`)."""
        if not getattr(args, "Dt_mergeDs", True):
            raise RuntimeError("Dt_mergeDs=False feeds 9 channels into a 27-channel conv in the reference and raises "
                               "there too (SURVEY.md 8a8)")
        # vgg_scaling > 0 (opt-in; the reference's own VGG path cannot execute, DESIGN.md fixes its semantics): needs the
        # frozen feature extractor as args.tg_vgg (train.get_step builds the default one)
        self.vgg_scaling = max(0.0, float(getattr(args, "vgg_scaling", -1.0)))
        if self.vgg_scaling > 0.0 and getattr(args, "tg_vgg", None) is None:
            raise ValueError("vgg_scaling > 0 needs args.tg_vgg (a models.VGG19 module); train.get_step provides it")
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
        self._y_src = None
        f32 = dict(dtype=torch.float32, device=device)
        self.x = torch.empty(B, T, 3, h, h, **f32)
        self.y = torch.empty(B, T, 3, H, H, **f32)
        self.gen = torch.empty(B, T, 3, H, H, **f32)
        self.flow = torch.empty(B, T - 1, 2, H, H, **f32)
        self.tvel = torch.empty(B * self.tsize, H, H, 2, **f32)
        self.target = self.out_gen = self.out_scalars = None   # the last call's fresh result tensors (_emit_target / _emit_gen / _emit_scalars)
        self.acc = torch.zeros(16, **f32)
        self.scalars = torch.zeros(64, **f32)
        # per-step host parameters (loss config, Adam bias corrections, lr) travel through ONE small async copy from a
        # ring of pinned slots, so the CPU may run many steps ahead of the GPU without overwriting a pending copy
        self.params_dev = torch.zeros(48, **f32)   # [0:16] loss config, [16:32] two Adam rows, [32:40] VGG loss config,
        self.cfg = self.params_dev                  # [40:48] the estimator's Adam row (opt-in FNet training)
        self.hyper = self.params_dev[16:32].view(2, 8)   # (tg_loss_finalize reads cfg[32..35] when the VGG flag is set)
        self.hyper_f = self.params_dev[40:48]
        self.ring = torch.zeros(256, 48, dtype=torch.float32).pin_memory()
        # fp16 element type: dynamic loss scaling, state on the device (scale, growth tracker, found_inf G / D, 1/scale) so
        # that the captured graphs read the current scale and a skipped update needs no host round trip.  The reference
        # shares ONE GradScaler (init 65536) between both optimisers and updates it after each step() (code/train.py:9,
        # 335-342).  Here both backward passes of a step start from the same scale - they run concurrently - and the two
        # update() calls follow the two Adam launches; the unscaled gradients are the same either way.
        self.scaler = None
        if G.dt == torch.float16:
            s0 = float(getattr(args, "tg_loss_scale", 65536.0))
            self.scaler = torch.tensor([s0, 0.0, 0.0, 0.0, 1.0 / s0, 0.0, 0.0, 0.0], **f32)
        self.loss_scale = self.scaler[0:1] if self.scaler is not None else None
        self.ring_i = 0
        # the engines keep one buffer set per launch shape; this step's sets are pinned (its graphs hold their addresses)
        # and re-selected at the start of every run(), so a module forward at another shape in between is harmless
        tu = self.tu = tuning.current()
        cap_g, cap_dr = K.persist_wgs_g_for(B * h * h), K.persist_wgs_dreal_for(B * h * h)
        if cap_g is not None:
            G.set_cap(cap_g, tu.cap_fwd_g_for(B * h * h))
        G.set_trunk_cap(tu.cap_trunk_g_for(B * h * h))
        # (set both ways: the engine may have served a step of another size before)
        D.cap[0] = cap_dr if cap_dr is not None else tu.cap_dreal_default()
        D.rw_extra_real = tu.rw_extra_dreal if tu.rw_extra_dreal is not None else ("s1" if cap_dr is not None else "")
        G.sets.pin((T * B, h, h))
        G.alloc(T * B, h, h)
        G._alloc_grad()
        # d(loss)/d(pre-sigmoid): 3 real channels.  16-bit modes without the VGG term keep it compact ([..., 4]) and run the
        # output layer's backward as one launch (csrc/rgb_bwd.hip); else the padded operand of the generic conv launches
        compact = G.rgb_bwd_ok() and self.vgg_scaling <= 0.0
        self.dpre = torch.empty(T * B, H, H, 4 if compact else 32, dtype=G.dt, device=device)
        # TECOGAN_LANES=0: the whole forward/backward as ONE forked capture (lane B's stream is then an ordinary one:
        # a CU mask is lost inside a forked graph)
        self.lanes = tu.lanes
        # data-parallel mode: two gradient buckets per network, all-reduced where they become final (TecoGANStep._run_lanes)
        # Data-parallel collectives.  Default: ONE all-reduce per network issued as a SYNCHRONOUS call on the lane's own stream, behind
        # the piece that completes the gradients (this torch build enqueues a synchronous RCCL collective on the current stream:
        # no hand-over to the backend's internal stream and back).  One RCCL rank on one GPU: 4.318 ms/step vs 4.313 without a
        # process group; the asynchronous forms pay two cross-stream event hops per collective: 4.58 ms with one per network,
        # 4.64 with the two buckets per network of TECOGAN_DP_INLINE=0 (profiles/r03_q_dp_inline.log).  What the inline form
        # gives up is overlap of the first bucket (~4 MB of 7 / 13 MB) with the rest of the backward pass - worth less than
        # the hops as long as a bucket's all-reduce is shorter than ~0.3 ms.
        if process_group is not None:
            parallel.warm_backend(process_group, device)   # (the backend's asynchronous path exists before the lanes' graphs do)
        self.dp_inline = process_group is not None and tu.dp_inline
        # The inline form relies on a synchronous all-reduce being ordered on the issuing stream (true of this torch build's RCCL
        # backend); checked once per process group on a 2-element tensor, with the asynchronous form as fallback.
        self.dp_sync_ordered = None
        if self.dp_inline:
            self.dp_sync_ordered = parallel.sync_allreduce_stream_ordered(process_group, device)
            if self.dp_sync_ordered is False:
                import warnings
                warnings.warn("pytorch-tecogan_amd: a synchronous all_reduce is NOT ordered on the issuing stream with this torch / "
                              "backend - falling back to asynchronous collectives (TECOGAN_DP_INLINE=0 behaviour)")
                self.dp_inline = False
        # bench.py: HIP events around each lane's exposed collective time / a pass with every all-reduce skipped (dp_breakdown)
        self.dp_events, self.skip_collectives = None, False
        self.buckets = self.lanes and process_group is not None and tu.dp_buckets and \
            not self.dp_inline
        # measured (tools/lane_matrix.sh, profiles/r02_b_lane_matrix.log): reserving CUs for the chain does not pay - the dense
        # lane loses more on 192 CUs than the chain gains - so the default is an unmasked lane B
        self.reserve = tu.cu_reserve if self.lanes else 0
        # hipExtStreamCreateWithCUMask makes a BLOCKING stream: it serialises against the legacy default stream.  Lane A
        # therefore runs on a stream of its own as well (the caller's stream only brackets the step)
        # lane A is the caller's stream (no hand-over: two cross-stream waits cost ~15 us of idle chip each, every step) -
        # except beside a CU-masked lane B, whose BLOCKING stream would serialise against the legacy default stream
        self.sA = torch.cuda.Stream(device=device) if self.reserve > 0 else None
        self.sB = lane_b_stream(torch.device(device))
        # the real half runs BESIDE the chain (phase 1): only there can a CU reservation pay - its stream may be masked off
        # the first TECOGAN_CU_RESERVE CUs; the fake half (phase 2, beside the dense G backward) always has the whole chip
        self.sBm = lane_stream(device, self.reserve) if self.reserve > 0 else self.sB
        self.dreal_bwd_early = tu.dreal_bwd
        if self.lanes and process_group is not None and not parallel.streams_overlap(device, torch.cuda.current_stream(device), self.sB):
            import warnings
            warnings.warn("pytorch-tecogan_amd: the step's two lane streams do NOT overlap (they share a hardware queue): expect "
                          "~1.6x the step time.  Set GPU_MAX_HW_QUEUES=8 (or import pytorch_tecogan_amd) BEFORE the process's first "
                          "HIP call - torch.cuda.is_available() / device_count() may already be one (DESIGN.md (e))")
        self.ev = {k: torch.cuda.Event() for k in ("start", "prep", "chain", "tail", "d", "dreal")}
        D.sets.pin((2 * self.tb, H))
        D.alloc(2 * self.tb, H)
        self.V = None
        if self.vgg_scaling > 0.0:
            self.V = args.tg_vgg.engine(G.dt)
            self.V.sets.pin((T * B, H, H))
            self.V.alloc(T * B, H, H)
        self._tables()
        self.graphs = None
        self.adam_t = [0, 0, 0]
        self.border = (H - int(H * args.crop_dt)) // 2 if args.crop_dt < 1.0 else 0

    # ----------------------------------------------------------------------------------------------------------
    def _tables(self):
        # opt-in (not reference behaviour): args.tg_fnet = an f_net module -> flow = up4(4 * f_net(previous LR frame))
        fn = getattr(self.args, "tg_fnet", None)
        self.F = fn.engine(self.G.dt) if fn is not None else None
        # opt-in on top of it: args.tg_fnet_train -> the estimator is trained on the LR warp loss (DESIGN.md "FNet training")
        self.F_train = self.F is not None and bool(getattr(self.args, "tg_fnet_train", False))
        if self.F_train and self.G.dt == torch.float16:
            raise ValueError("tg_fnet_train runs in bf16 / fp32 (the fp16 loss scaler tracks two optimisers, code/train.py:9)")
        if self.F is not None:
            self.F.sets.pin((self.B * self.T, self.h, self.h))
            self.F.alloc(self.B * self.T, self.h, self.h)
            self.fx = torch.empty(self.B * self.T, 2, self.h, self.h, dtype=torch.float32, device=self.dev)
            if self.F_train:
                self.F._alloc_grad()
                self.dfx = torch.zeros_like(self.fx)   # blocks of the last frame of a sequence are never written: zero
                hh, T = self.h * self.h, self.T
                self.fx_off = _i64([(b * T + t) * 2 * hh for b in range(self.B) for t in range(T - 1)], self.dev)
        t = build_tables(self.B, self.T, self.h, self.K, self.pingpang, fnet_flow=self.F is not None)
        dev = self.dev
        self.flow_src, self.flow_dst, self.n_flow = _i64(t["flow_src"], dev), _i64(t["flow_dst"], dev), len(t["flow_src"])
        self.lrw_img, self.lrw_grid = _i64(t["lrw_img"], dev), _i64(t["lrw_grid"], dev)
        self.tv_csrc, self.tv_cdst, self.n_tvc = _i64(t["tv_csrc"], dev), _i64(t["tv_cdst"], dev), len(t["tv_csrc"])
        self.tv_bsrc, self.tv_bdst, self.n_tvb = _i64(t["tv_bsrc"], dev), _i64(t["tv_bdst"], dev), len(t["tv_bsrc"])

    def _select_sets(self):
        self.G.alloc(self.T * self.B, self.h, self.h)
        self.G._alloc_grad()
        self.D.alloc(2 * self.tb, self.H)
        if self.F is not None:
            self.F.alloc(self.B * self.T, self.h, self.h)
        if self.V is not None:
            self.V.alloc(self.T * self.B, self.H, self.H)

    def close(self):
        """releases the pins on the engines' buffer sets (train.get_step calls it when another configuration replaces
        this one); the graphs of a closed step must not be replayed"""
        self.graphs = None
        self._merge_skipped_updates()
        from .engine import release_plans
        for eng, shape in ((self.G, (self.T * self.B, self.h, self.h)), (self.D, (2 * self.tb, self.H)),
                           (self.F, (self.B * self.T, self.h, self.h)), (self.V, (self.T * self.B, self.H, self.H))):
            if eng is not None:
                eng.sets.unpin(shape)
                eng.sets.drop(shape)      # the step's activation / gradient buffers go with it
                release_plans(eng)        # ... and every launch plan keyed by their addresses; the capture freeze is lifted
                if eng.shape == shape:    # the engine re-selects (re-creates) a set at its next alloc()
                    eng.shape = None
                    for attr in ("cur", "act", "grad", "gbuf", "g_c0", "prob", "dlogit"):
                        if hasattr(eng, attr):
                            setattr(eng, attr, None)
        if self.sBm is not self.sB:       # the CU-masked stream was created through the C ABI (lane_stream)
            import ctypes
            torch.cuda.synchronize(self.dev)
            L.check(L.load().tg_stream_destroy(ctypes.c_void_p(self.sBm.cuda_stream)), "tg_stream_destroy")
            self.sBm = self.sB

    def _merge_skipped_updates(self):
        """fp16: Adam's true step count is the host's count minus the updates this step's scaler skipped (scaler[5:7], device).
        Moves that count into the bound optimisers' step tensors (train.sync_optimizer_steps does the same before a checkpoint)
        so that it survives this step being closed or its scaler state being overwritten.  Synchronises; no-op otherwise."""
        opts = getattr(self, "_opts", None)
        if self.scaler is None or not opts:
            return
        skipped = self.scaler[5:7].cpu()
        for opt, k in zip(opts, (0, 1)):
            if getattr(opt, "_tg_step", None) is not None and float(skipped[k]) != 0.0:
                opt._tg_step -= float(skipped[k])
        self.scaler[5:7].zero_()

    # ----------------------------------------------------------------------------------------------------------
    def _host_params(self, global_step, lr_g, lr_d, betas_g, betas_d, eps_g, eps_d, f_hyper=None):
        a = self.args
        B, T, h, H = self.B, self.T, self.h, self.H
        slot = self.ring[self.ring_i % self.ring.shape[0]]
        self.ring_i += 1
        c = [0.0] * 48
        c[0] = B * T * 3 * H                      # content: mean over (BT,3,H) of sum over W
        c[1] = B * (T - 1) * 3 * h                # warp loss
        for i, l in enumerate(self.D.layers()):
            c[2 + i] = self.tb * l.shape[3] * l.shape[1]  # mean over (tb, C, Hl) of sum over W
            c[12 + i] = LAYER_NORM[i]
        c[6], c[7] = a.EPS, a.ratio
        c[8] = min(a.Dt_ratio_max, a.Dt_ratio_0 + a.Dt_ratio_add * float(global_step))
        c[9] = (1.0 if a.D_LAYERLOSS else 0.0) + (2.0 if self.V is not None else 0.0)   # flag word
        if self.V is not None:
            c[32] = self.vgg_scaling
            for i, t in enumerate(("Conv2_2", "Conv3_4", "Conv4_4")):
                f = self.V.act[t]
                c[33 + i] = float(B * T * f.shape[1] * f.shape[2])
        c[10] = float(B * (self.T_in - 1) * 3 * H * H) if self.pingpang else 0.0
        c[11] = a.pp_scaling
        self.dt_ratio = c[8]
        gs = 1.0 / self.world
        c[16:24] = K.adam_hyper(lr_g, betas_g[0], betas_g[1], eps_g, self.adam_t[0] + 1, gs)
        c[24:32] = K.adam_hyper(lr_d, betas_d[0], betas_d[1], eps_d, self.adam_t[1] + 1, gs)
        if self.F_train:
            lr_f, betas_f, eps_f = f_hyper or (lr_g, betas_g, eps_g)   # main.py:244-245: the generator's settings
            c[40:48] = K.adam_hyper(lr_f, betas_f[0], betas_f[1], eps_f, self.adam_t[2] + 1, gs)
        slot.copy_(torch.tensor(c, dtype=torch.float32))
        self._params_slot = slot   # copied to the device at the head of lane B (_stage_inputs_b)

    def _stage_inputs_b(self):
        """HR targets -> the static buffer the graphs read, parameter block -> device (current stream: lane B's)"""
        y, Ti = self._y_src, self.T_in
        if y is None:   # a replay of the schedule on the inputs already staged (tools/step_breakdown.py)
            return
        self.y[:, :Ti].copy_(y)
        if self.pingpang:
            self.y[:, Ti:].copy_(torch.flip(y, dims=[1])[:, 1:])
        self.params_dev.copy_(self._params_slot, non_blocking=True)
        self._y_src = None

    # ---------------------------------------------------------------------------------------------------------- pieces
    # Each piece is a linear launch sequence on the CURRENT stream (no forks inside), so it can be captured as one graph
    # and replayed on whatever stream its lane uses.
    def _prep(self):
        """zero the accumulators / gradient buffers; pseudo-flow, LR warp loss (logged only) and T_vel"""
        G, D, B, T, h, H = self.G, self.D, self.B, self.T, self.h, self.H
        hh, HH = h * h, H * H
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
        if not self.F_train:  # (with FNet training the warp loss is the estimator's: _fnet_bwd)
            K.warp_nchw(self.x, self.lrw_img, self.x, self.lrw_grid, B * (T - 1), 3, h, h, h, h, False, sq_ref=self.x,
                        sq_off=self.lrw_grid, loss_acc=self.acc[1:2])
        K.copy_blocks(self.flow, self.tv_csrc, self.tvel, self.tv_cdst, self.n_tvc, 2 * HH)
        if self.n_tvb:
            K.up4_planes(self.x, self.tv_bsrc, self.tvel, self.tv_bdst, self.n_tvb, h, h, pre=4.0, post_a=2.0, post_b=-1.0)

    def _fnet_bwd(self):
        """opt-in FNet training: the LR warp loss of code/train.py:78-84,247-249 with the estimator's output as the
        sampling grid, its gradient w.r.t. that grid, and the estimator's backward pass.  Lane B, behind the prologue:
        lane A only waits for the flow, not for this."""
        B, T, h = self.B, self.T, self.h
        self.F.flat.g.zero_()
        K.warp_grid_grad(self.x, self.lrw_img, self.fx, self.fx_off, self.x, self.lrw_grid, self.dfx, self.fx_off, B * (T - 1),
                         3, h, h, h, h, 1.0 / (B * (T - 1) * 3 * h), loss_acc=self.acc[1:2], loss_scale=self.loss_scale)
        self.F.backward(self.dfx, self.fx)

    def _update_f(self):
        """the estimator's Adam (the third optimiser the reference leaves commented out, main.py:244-245) + repack"""
        F = self.F
        K.adam(F.flat.p, F.flat.g, F.flat.m, F.flat.v, self.hyper_f)
        F.repack()

    def _d_real(self, backward=None):
        """real half of the discriminator: input assembly, forward (BN statistics of this half, first running-stat
        update) and - independent of the generator - its backward pass"""
        D, B, T, h, H, tb = self.D, self.B, self.T, self.h, self.H, self.tb
        backward = self.dreal_bwd_early if backward is None else backward
        K.d_assemble(self.x, self.y, self.gen, self.tvel, D.act["in"][:tb], B, T, self.K, h, self.border, half=0)
        D.forward(update_stats=True, half=0)
        if backward:   # (the loss seed of this half - tg_dlogit_real - is the backward pass's first action: inside its tail launch)
            D.backward(groups=2, half=0, real_seed=(self.cfg, self.loss_scale))
        # (the returned `target` - this half's input as fp32 NCHW, code/train.py:368 - is converted by _emit_target right behind this
        # piece, where lane B waits for the chain anyway)

    def _emit_target(self):
        """Network.target of this call: a NEW tensor every call, as in the reference (code/train.py:357-370 builds its results
        afresh), written on the current stream - lane B, behind the real half, outside the hipGraphs (their addresses are fixed).
        The fake half writes D.act["in"][tb:], not the real half's rows, so this reads what the real half's forward saw."""
        H, tb = self.H, self.tb
        out = torch.empty(tb, 27, H, H, dtype=torch.float32, device=self.dev)
        out.record_stream(self._caller)   # allocated on lane B's stream, read by the caller's
        K.nhwc_to_nchw(self.D.act["in"][:tb], out, 27 * H * H, tb, 27, H, H)
        self.target = out

    def _emit_scalars(self):
        """the step's loss scalars for the Network tuple: a copy, so that the next call's graphs may overwrite the buffer - made on lane B
        (whose loss_finalize wrote them), not behind the step on the caller's stream"""
        out = torch.empty_like(self.scalars)
        out.record_stream(self._caller)
        out.copy_(self.scalars, non_blocking=True)
        self.out_scalars = out

    def _emit_gen(self):
        """Network.gen_output of this call: a fresh copy of the step's frame buffer (which the next call's graphs overwrite)"""
        out = torch.empty_like(self.gen)
        out.record_stream(self._caller)
        out.copy_(self.gen, non_blocking=True)
        self.out_gen = out

    def _chain(self, t0=0, t1=None, loss=False):
        """recurrent generator passes t0..t1-1 (each: warp + pack, conv0, residual trunk, up-sampling stage).  The piece
        'chain' runs the frames the discriminator sees (the first 3*(T//3), code/train.py:138-142), 'chain_tail' the rest
        (frame 9 of 10) and then the content loss and d(loss)/d(pre-sigmoid) of all frames - so the fake half of the
        discriminator starts one pass earlier, beside the tail"""
        G, B, T, h, H = self.G, self.B, self.T, self.h, self.H
        hh, HH = h * h, H * H
        t1 = self.tsize if t1 is None else t1
        for t in range(t0, t1):
            dst = G.act["in0"][t * B:(t + 1) * B]
            if t == 0:
                K.gen_input(self.x, 0, T * 3 * hh, None, 0, 0, None, 0, 0, dst, B, h, h)
            else:
                K.gen_input(self.x, t * 3 * hh, T * 3 * hh, self.gen, (t - 1) * 3 * HH, T * 3 * HH, self.flow,
                            (t - 1) * 2 * HH, (T - 1) * 2 * HH, dst, B, h, h)
            G.forward(t * B, B, self.gen, t * 3 * HH, T * 3 * HH)
        if not loss:
            return
        pp_T = self.T_in if self.pingpang else 0
        pp_coef = (2.0 * self.args.pp_scaling / (B * (self.T_in - 1) * 3 * H * H)) if (self.pingpang and self.args.pp_scaling > 0) else 0.0
        # the output layer's bias gradient is the channel sum of d(loss)/d(pre-sigmoid): added straight into its slot of the
        # flat gradient buffer (zeroed by the prologue, which lane A has picked up before pass 1)
        K.content_loss(self.gen, self.y, self.dpre, self.acc, B, T, H, H, 1.0 / (B * T * 3 * H), 0, T, pp_T, pp_coef,
                       loss_scale=self.loss_scale, bias_acc=G.cout.gbias)
        if self.V is not None:  # VGG features of all generated and target frames; its input-gradient is added to dpre
            gen, tgt = self.gen.view(B * T, 3, H, H), self.y.view(B * T, 3, H, H)
            self.V.forward(gen, tgt)
            self.V.loss_backward(self.acc[11:14], self.vgg_scaling, gen, self.dpre, loss_scale=self.loss_scale,
                                 bias_acc=G.cout.gbias)

    def _g_backward(self, part=None):
        """G backward for all T*B samples as ONE batch (the passes are independent in backward: every generator input is
        detached, code/train.py:90,108); the output bias gradient came from the content-loss kernel's channel sums"""
        self.G.backward(0, self.T * self.B, dpre=self.dpre, part=part)

    def _d_fake(self):
        """fake half, forward: input assembly from the generated frames the discriminator sees, forward, layer losses"""
        D, B, T, h, tb = self.D, self.B, self.T, self.h, self.tb
        K.d_assemble(self.x, self.y, self.gen, self.tvel, D.act["in"][tb:], B, T, self.K, h, self.border, half=1)
        D.forward(update_stats=True, half=1)
        if self.args.D_LAYERLOSS:   # the four layer losses in one launch (they sit on the step's critical path)
            key = tuple(l.data_ptr() for l in D.layers())
            if getattr(self, "_ll_key", None) != key:
                rows = [[l[:tb].data_ptr(), l[tb:].data_ptr(), self.acc.data_ptr() + 4 * (2 + i), tb * l.shape[1] * l.shape[2],
                         l.shape[3], l.shape[3]] for i, l in enumerate(D.layers())]
                self._ll_jobs, self._ll_key = _i64(rows, self.dev), key
            K.absdiff_sum_multi(D.dt, self._ll_jobs, self._ll_jobs.shape[0])

    def _d_fake_bwd(self, backward=True, part=None):
        """every loss scalar (needs the content loss of lane A's tail) and d(logit), then the backward pass of the fake
        half (or of both halves when the real half has not run its own yet).  part 'hi' / 'lo': the two gradient buckets
        of data-parallel mode (fc ... block2, then stage 1 and conv.0) as separate launch sequences."""
        D, tb = self.D, self.tb
        if part != "lo":
            K.loss_finalize(D.prob, self.acc, self.scalars, D.dlogit, tb, self.cfg, self.loss_scale)
        if backward:
            if self.dreal_bwd_early:
                D.backward(groups=2, half=1, part=part)
            else:
                D.backward(groups=2, part=part)

    def _update_d(self):
        """discriminator: Adam + repack.  Runs at the tail of lane B, as soon as D's gradients are final - lane B is done
        ~0.3 ms before lane A's G backward, so this is off the step's serial tail."""
        D, sc = self.D, self.scaler
        if sc is not None:  # GradScaler.step: inf/NaN anywhere in a network's (all-reduced) gradients skips its update
            K.check_finite(D.flat.g, sc[3:4])
        K.adam(D.flat.p, D.flat.g, D.flat.m, D.flat.v, self.hyper[1], scaler=sc, which=1)
        D.repack()

    def _update(self, with_d=False):
        """generator: Adam + repack (+ the loss scaler's two update() calls, which need both networks' found_inf flags: the
        caller has joined lane B by now).  with_d: the single-stream schedules run the discriminator's update here too."""
        G, sc = self.G, self.scaler
        if sc is not None:
            K.check_finite(G.flat.g, sc[2:3])
        K.adam(G.flat.p, G.flat.g, G.flat.m, G.flat.v, self.hyper[0], scaler=sc, which=0)
        if with_d:
            self._update_d()
            if self.F_train:
                self._update_f()
        if sc is not None:
            K.scaler_update(sc)
        G.repack()

    def _update_all(self):
        self._update(with_d=True)

    def scaler_state(self):
        """{'scale', 'growth_tracker'} of the fp16 loss scaler (GradScaler.state_dict() keys), None otherwise; synchronises"""
        if self.scaler is None:
            return None
        v = self.scaler.cpu()
        return {"scale": float(v[0]), "growth_tracker": int(v[1])}

    def _chain_tail(self):
        self._chain(self.tsize, self.T, loss=True)

    def _chain0(self):
        self._chain(0, 1)

    def _chain_rest(self):
        self._chain(1, self.tsize)

    PIECES = ("prep", "d_real", "chain0", "chain", "chain_tail", "d_fake", "d_fake_bwd", "g_bwd", "update_d", "update")
    # data-parallel mode: both backward passes are cut where their first gradient bucket is final
    PIECES_DP = ("prep", "d_real", "chain0", "chain", "chain_tail", "d_fake", "d_fake_bwd_hi", "d_fake_bwd_lo", "g_bwd_hr",
                 "g_bwd_trunk", "update_d", "update")
    LANE_B = ("prep", "d_real", "d_fake", "d_fake_bwd", "d_fake_bwd_hi", "d_fake_bwd_lo", "update_d")

    def _piece_fns(self):
        fns = {"prep": self._prep, "d_real": self._d_real, "chain0": self._chain0, "chain": self._chain_rest,
               "chain_tail": self._chain_tail,
               "d_fake": self._d_fake, "d_fake_bwd": self._d_fake_bwd, "g_bwd": self._g_backward,
               "update_d": self._update_d, "update": self._update}
        if self.F_train:
            fns.update({"fnet_bwd": self._fnet_bwd, "update_f": self._update_f})
        if self.buckets:
            fns.update({"d_fake_bwd_hi": lambda: self._d_fake_bwd(part="hi"), "d_fake_bwd_lo": lambda: self._d_fake_bwd(part="lo"),
                        "g_bwd_hr": lambda: self._g_backward(part="hr"), "g_bwd_trunk": lambda: self._g_backward(part="trunk")})
        return fns

    def _forward_backward(self, include_d_backward=True):
        """everything up to the update on the CURRENT stream alone, in dependency order (serial; tools and bench.py's
        per-launch roofline pass use this)"""
        self._prep()
        if self.F_train:
            self._fnet_bwd()
        self._d_real()
        self._chain()
        self._chain_tail()
        self._d_fake()
        self._d_fake_bwd(backward=include_d_backward)
        self._g_backward()

    # ---------------------------------------------------------------------------------------------------------- schedule
    def _allreduce(self, buf):
        if self.skip_collectives:
            return None
        forced = self.pg is not None and self.tu.force_collectives
        if self.dp_inline and (forced or self.world > 1):
            import torch.distributed as dist
            if dist.get_backend(self.pg) != "gloo" or not buf.is_cuda:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)   # stream-ordered on the current (lane) stream
                return None
        if forced:
            import torch.distributed as dist  # test hook: exercise the collective's call path even with one rank
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        return parallel.allreduce_sum_async(buf, self.pg, self.world)

    def _run_lanes(self, fn):
        """fn: piece name -> callable (the eager pieces or their graphs' replay).  Cross-lane dependencies are events
        recorded between the pieces; the collectives of data-parallel mode are issued where their inputs become final (see
        below: two buckets per network).  Work.wait() of the RCCL backend makes the current STREAM wait (no host block)."""
        main, sB, sBm, ev = torch.cuda.current_stream(), self.sB, self.sBm, self.ev
        tm = self.dp_events   # {"A0", "A1", "B0", "B1"}: timing events around what each lane waits for its collectives (bench.py)
        # the step's prologue (zeroing, pseudo-flow, T_vel: 8 small launches, ~60 us) runs at the head of lane B while lane A
        # is already in the first generator pass - frame 0 has no previous frame, so it needs neither the flow nor any of
        # the zeroed accumulators; lane A picks the prologue up before pass 1 (kernel trace: 125 us from the previous
        # step's last kernel to this step's first convolution when the prologue ran in front of both lanes)
        ev["start"].record(main)
        sBm.wait_event(ev["start"])
        with torch.cuda.stream(sBm):
            self._stage_inputs_b()
            fn["prep"]()
            ev["prep"].record(sBm)
            if self.F_train:   # estimator: loss, backward, [all-reduce,] Adam - its weights are next read by the NEXT step's prologue
                fn["fnet_bwd"]()
                w_f = self._allreduce(self.F.flat.g)
                if w_f is not None:
                    w_f.wait()
                fn["update_f"]()
            fn["d_real"]()
            self._emit_target()
        if sBm is not sB:
            ev["dreal"].record(sBm)
            sB.wait_event(ev["dreal"])
        fn["chain0"]()
        main.wait_event(ev["prep"])
        fn["chain"]()
        ev["chain"].record(main)
        sB.wait_event(ev["chain"])
        with torch.cuda.stream(sB):
            fn["d_fake"]()
        fn["chain_tail"]()
        ev["tail"].record(main)
        sB.wait_event(ev["tail"])
        if not self.buckets:
            # one all-reduce per network (TECOGAN_DP_BUCKETS=0, or no process group: _allreduce returns None); the D one is
            # issued first - lane B ends before lane A
            with torch.cuda.stream(sB):
                fn["d_fake_bwd"]()
                if tm:
                    tm["B0"].record(sB)
                works_d = (self._allreduce(self.D.flat.g),)
            fn["g_bwd"]()
            if tm:
                tm["A0"].record(main)
            works_g = (self._allreduce(self.G.flat.g),)
        else:
            # Data parallel.  RCCL runs a process group's collectives on ONE internal stream in issue order (the same on every
            # rank), so they are issued in the order their inputs become final: the generator's up-sampling stage (its weight
            # gradients are folded ~0.8 ms before the G backward ends: the 32 trunk input-gradients and the trunk's weight
            # gradients are still to come), the discriminator's upper stages (stage 1, the largest tensors, is still to come),
            # then the two remainders.  Each is enqueued behind the piece that completes it, on that piece's lane.
            G, D = self.G, self.D
            gs, ds = G.bucket_split(), D.bucket_split()
            with torch.cuda.stream(sB):
                fn["d_fake_bwd_hi"]()
            fn["g_bwd_hr"]()
            w_g1 = self._allreduce(G.flat.g[gs:])
            with torch.cuda.stream(sB):
                w_d1 = self._allreduce(D.flat.g[ds:])
                fn["d_fake_bwd_lo"]()
                if tm:
                    tm["B0"].record(sB)
                w_d2 = self._allreduce(D.flat.g[:ds])
            fn["g_bwd_trunk"]()
            if tm:
                tm["A0"].record(main)
            w_g2 = self._allreduce(G.flat.g[:gs])
            works_d, works_g = (w_d1, w_d2), (w_g1, w_g2)
        with torch.cuda.stream(sB):
            for w in works_d:
                if w is not None:
                    w.wait()       # (RCCL: makes lane B's stream wait, no host block)
            if tm:
                tm["B1"].record(sB)
            fn["update_d"]()
            self._emit_gen()   # (the frames are final since the tail event this lane waited for; 8 MB, beside lane A's update)
            self._emit_scalars()
            ev["d"].record(sB)
        for w in works_g:
            if w is not None:
                w.wait()
        if tm:
            tm["A1"].record(main)
        if self.scaler is None:
            # the generator's Adam + repack need nothing of lane B: they run beside lane B's tail (the fake half's last weight
            # gradients, fold, D update); the caller's stream then picks lane B up (its results are read next)
            fn["update"]()
            main.wait_event(ev["d"])
        else:  # fp16: the shared loss scaler's two update() calls sit in this piece and need both networks' found_inf flags
            main.wait_event(ev["d"])
            fn["update"]()

    def _fork_join(self):
        """TECOGAN_LANES=0: the same schedule as ONE capturable fork/join (all joins go into the origin stream: HIP stream
        capture segfaults when a non-origin captured stream joins a side stream, tools/capture_probe.py)"""
        main, sB = torch.cuda.current_stream(), self.sB
        self._prep()
        if self.F_train:
            self._fnet_bwd()
        sB.wait_stream(main)
        with torch.cuda.stream(sB):
            self._d_real()
        self._chain()
        sB.wait_stream(main)
        with torch.cuda.stream(sB):
            self._d_fake()
        self._chain_tail()
        sB.wait_stream(main)
        with torch.cuda.stream(sB):
            self._d_fake_bwd()
        self._g_backward()
        main.wait_stream(sB)

    def _run_single(self, fwd_bwd, update):
        fwd_bwd()
        self._emit_target()
        self._emit_gen()
        self._emit_scalars()
        works = [self._allreduce(self.G.flat.g), self._allreduce(self.D.flat.g)]
        if self.F_train:
            works.append(self._allreduce(self.F.flat.g))
        for w in works:
            if w is not None:
                w.wait()
        update()

    def _capture(self):
        pool = torch.cuda.graph_pool_handle()
        self.G.ws.frozen = self.D.ws.frozen = True

        def cap(fn, stream=None):
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread polls its work events (hipEventQuery) while this thread captures;
            # under the default global mode that aborts with hipErrorStreamCaptureUnsupported
            with torch.cuda.graph(g, pool=pool, stream=stream, capture_error_mode="thread_local"):
                fn()
            return g.replay

        if self.lanes:
            fns = self._piece_fns()
            names = (self.PIECES_DP if self.buckets else self.PIECES) + (("fnet_bwd", "update_f") if self.F_train else ())
            # Lane B's pieces are captured on torch's own capture stream, not on lane B's stream (only a CU-masked lane stream is
            # captured on itself): the default data-parallel mode records RCCL work events on the lane streams, the backend's
            # watchdog thread polls them, and hipEventQuery on an event whose stream is capturing aborts the process
            # (hipErrorCapturedEvent - 1 run in 10 of the one-rank RCCL test before this).  Replay picks the lane's stream.
            masked = self.sBm is not self.sB
            lane = lambda k: (self.sBm if k in ("prep", "d_real", "fnet_bwd", "update_f") else (self.sB if k in self.LANE_B else None)) \
                if masked else None  # noqa: E731
            self.graphs = {k: cap(fns[k], lane(k)) for k in names}
        else:
            self.graphs = (cap(self._fork_join), cap(self._update_all))

    # ----------------------------------------------------------------------------------------------------------
    def run(self, x, y, global_step, lr_g, lr_d, betas_g=(0.9, 0.999), betas_d=(0.9, 0.999), eps_g=1e-8, eps_d=1e-8,
            f_hyper=None):
        """x (B,T,3,h,h), y (B,T,3,H,H) fp32 device tensors.  Returns nothing; results: self.out_gen / self.target (fresh tensors
        every call), self.scalars, self.gen (the internal frame buffer, overwritten by the next call); the parameter / optimiser
        buffers are updated in place."""
        Ti = self.T_in
        if tuple(x.shape) != (self.B, Ti, 3, self.h, self.h) or tuple(y.shape) != (self.B, Ti, 3, self.H, self.H):
            raise ValueError(f"step built for B={self.B}, T={Ti}, crop {self.h}; got {tuple(x.shape)} / {tuple(y.shape)}")
        self._select_sets()
        # the LR frames are all the first generator pass needs: they are copied here, on the caller's stream (lane A); the HR
        # targets and the step's parameter block go in at the head of lane B (_stage_inputs_b), off lane A's critical path
        self.x[:, :Ti].copy_(x)
        if self.pingpang:  # reverse(x)[1:] appended (data movement only)
            self.x[:, Ti:].copy_(torch.flip(x, dims=[1])[:, 1:])
        self._y_src = y
        self._host_params(global_step + 1, lr_g, lr_d, betas_g, betas_d, eps_g, eps_d, f_hyper)
        if not self.lanes:
            self._stage_inputs_b()
        eager = (lambda: self._run_lanes(self._piece_fns())) if self.lanes else \
            (lambda: self._run_single(self._fork_join, self._update_all))
        caller = self._caller = torch.cuda.current_stream()
        laneA = self.sA if self.sA is not None else caller
        if laneA is not caller:
            laneA.wait_stream(caller)
        with torch.cuda.stream(laneA):
            if self.use_graph and self.graphs is None:
                eager()   # warm-up: one-time attribute setup, workspace growth, job tables
                torch.cuda.synchronize()
                self._capture()
            elif self.use_graph and self.lanes:
                self._run_lanes(self.graphs)
            elif self.use_graph:
                self._run_single(*self.graphs)
            else:
                eager()
        if laneA is not caller:
            caller.wait_stream(laneA)
        self.adam_t = [t + 1 for t in self.adam_t]


class RecurrentGenerator:
    """Generator-only recurrent inference (main.py:171-219) without the per-frame CPU<->GPU bounces.  The sequence is processed
    in CHUNKS of up to FR frames: the chunk's LR frames are staged with one copy into slots 1..n of a ring buffer (slot 0 holds the
    frame before the chunk), the chunk's n frame steps (flow -> warp + pack -> G, each reading slot s - 1 as "previous" and
    writing the HR frame into slot s of the output ring) are ONE hipGraph, one strided copy takes the n results out, two small
    copies carry slot n to slot 0.  Per frame that leaves 1/FR of a graph launch and of four copies (the first version moved
    four tensors around every frame; round 2 one LR copy in, one graph replay and one HR copy out per frame: 19 us of a 310-us
    frame at 128 x 128, profiles/r04_x_rw_fwd_routing.log)."""

    def __init__(self, G, B, h, w, device, use_graph=False):
        self.G, self.B, self.h, self.w, self.dev, self.use_graph = G, B, h, w, device, use_graph
        self.FR = max(1, tuning.current().infer_chunk)
        H, W = 4 * h, 4 * w
        f32 = dict(dtype=torch.float32, device=device)
        n = self.n = self.FR + 1
        self.sin = torch.empty(B, n, 3, h, w, **f32)
        self.sout = torch.zeros(B, n, 3, H, W, **f32)
        # the up-sampled pseudo-flow of every frame of a chunk: it depends on the LR inputs only (channels 0, 1 of the PREVIOUS LR
        # frame, code/train.py:77-80), so ONE launch per chunk computes all of them instead of one per frame inside the recurrence
        self.flow = torch.empty(self.FR, B, 2, H, W, **f32)
        hh, HH = h * w, H * W
        self.fsrc = _i64([((b * n + s) * 3 + c) * hh for s in range(n) for b in range(B) for c in range(2)], device).view(n, 2 * B)
        self.fdst = _i64([((s * B + b) * 2 + c) * HH for s in range(self.FR) for b in range(B) for c in range(2)], device).view(self.FR, 2 * B)
        G.sets.pin((B, h, w))   # the chunk graphs hold addresses of this buffer set (engine.ShapeSets)
        G.alloc(B, h, w)
        self.graphs = {}        # frames in the chunk -> graph

    @property
    def out(self):
        return self.sout[:, 0]

    def close(self):
        self.graphs = {}
        self.G.sets.unpin((self.B, self.h, self.w))

    def _frame(self, s):
        """the frame in slot s >= 1: previous LR / HR frames in slot s - 1, result into slot s of the output ring"""
        G, B, h, w, n = self.G, self.B, self.h, self.w, self.n
        hh, HH = h * w, 16 * h * w
        K.gen_input(self.sin, s * 3 * hh, n * 3 * hh, self.sout, (s - 1) * 3 * HH, n * 3 * HH, self.flow, (s - 1) * B * 2 * HH, 2 * HH,
                    G.act["in0"], B, h, w)
        G.forward(0, B, self.sout, s * 3 * HH, n * 3 * HH, keep_h=False)

    def _flows(self, nf):
        """pseudo-flow of slots 1..nf (from the LR frames in slots 0..nf-1) in one launch"""
        K.up4_planes(self.sin, self.fsrc[:nf].reshape(-1), self.flow, self.fdst[:nf].reshape(-1), 2 * self.B * nf, self.h, self.w, pre=4.0)

    def _chunk(self, nf):
        self._flows(nf)
        for s in range(1, nf + 1):
            self._frame(s)

    def run(self, frames):
        """frames (B,T,3,h,w) fp32 device -> (B,T,3,4h,4w)."""
        B, T = frames.shape[:2]
        h, w = self.h, self.w
        self.G.alloc(B, h, w)   # re-select this loop's buffer set (a training step may have selected its own since)
        # no other lane here: the persistent launches may take more of the chip than inside a training step (restored below)
        cap0, fwd0 = self.G.convs[0].persist_wgs, self.G.convs[0].persist_fwd
        self.G.set_cap(tuning.current().infer_wgs)
        try:
            return self._run(frames, B, T, h, w)
        finally:
            self.G.set_cap(cap0, fwd0)

    # ---- one frame per call: the live loop (experimental/live.py:100-128 of the reference; the same recurrence as main.py:191-219)
    def reset(self):
        """forget the previous frame: the next step() is a sequence's first frame (zeros as "previous", main.py:189-196)"""
        self._live = False

    def step(self, frame):
        """frame (B,3,h,w) fp32 device -> the HR frame (B,3,4h,4w), a new tensor.  STATEFUL: the previous LR / HR frames stay in slot 0
        of the rings between calls, so a caller can feed a camera one frame at a time - what run() computes for a whole sequence, in
        the same arithmetic (bit-identical: tests/test_inference_gpu.py).  With use_graph the frame step (flow -> warp + pack -> G)
        is the one-frame chunk graph; per call: one LR copy in, one replay, one HR copy out, two copies that carry slot 1 to slot 0."""
        B, h, w = self.B, self.h, self.w
        if tuple(frame.shape) != (B, 3, h, w):
            raise ValueError(f"recurrent step built for frames of {(B, 3, h, w)}, got {tuple(frame.shape)}")
        self.G.alloc(B, h, w)
        cap0, fwd0 = self.G.convs[0].persist_wgs, self.G.convs[0].persist_fwd
        self.G.set_cap(tuning.current().infer_wgs)
        try:
            n, hh, HH = self.n, h * w, 16 * h * w
            if not getattr(self, "_live", False):
                self.sin[:, 0].copy_(frame)
                K.gen_input(self.sin, 0, n * 3 * hh, None, 0, 0, None, 0, 0, self.G.act["in0"], B, h, w)
                self.G.forward(0, B, self.sout, 0, n * 3 * HH, keep_h=False)
                self._live = True
                return self.sout[:, 0].clone()
            self.sin[:, 1].copy_(frame)
            self._replay_chunk(1)
            out = self.sout[:, 1].clone()
            self.sin[:, 0].copy_(self.sin[:, 1])
            self.sout[:, 0].copy_(self.sout[:, 1])
            return out
        finally:
            self.G.set_cap(cap0, fwd0)

    def _replay_chunk(self, nf):
        if not self.use_graph:
            self._chunk(nf)
            return
        g = self.graphs.get(nf)
        if g is None:
            self._chunk(nf)           # warm-up: workspace growth, launch plans
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._chunk(nf)
            self.graphs[nf] = g
        g.replay()

    def _run(self, frames, B, T, h, w):
        n, hh, HH = self.n, h * w, 16 * h * w
        self._live = False   # (a whole-sequence run uses the same rings: a live stream restarts afterwards)
        outs = torch.empty(B, T, 3, 4 * h, 4 * w, dtype=torch.float32, device=self.dev)
        # frame 0 has no previous frame (main.py:189-196: zeros): it goes through slot 0, the first chunk's "previous" slot
        self.sin[:, 0].copy_(frames[:, 0])
        K.gen_input(self.sin, 0, n * 3 * hh, None, 0, 0, None, 0, 0, self.G.act["in0"], B, h, w)
        self.G.forward(0, B, self.sout, 0, n * 3 * HH, keep_h=False)
        outs[:, 0].copy_(self.sout[:, 0])
        t = 1
        while t < T:
            nf = min(self.FR, T - t)
            self.sin[:, 1:nf + 1].copy_(frames[:, t:t + nf])
            self._replay_chunk(nf)
            outs[:, t:t + nf].copy_(self.sout[:, 1:nf + 1])
            t += nf
            if t < T:   # the chunk's last frame becomes the next chunk's "previous"
                self.sin[:, 0].copy_(self.sin[:, nf])
                self.sout[:, 0].copy_(self.sout[:, nf])
        return outs

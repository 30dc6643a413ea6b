"""Torch-tensor facing wrappers over the C ABI (include/tecogan_hip.h).  PyTorch is used for device memory and
streams only; every function enqueues HIP kernels on torch's current stream (so they are graph-capturable)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Tuple

import torch

from . import _lib as L


def pad32(c):
    return (c + 31) // 32 * 32


def tg_dtype(dt):
    if dt == torch.bfloat16:
        return L.TG_BF16
    if dt == torch.float32:
        return L.TG_F32
    if dt == torch.float16:
        return L.TG_F16
    raise L.TecoganHipError(f"unsupported element type {dt}")


from . import tuning


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.TecoganHipError("HIP kernels need device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise L.TecoganHipError("HIP kernels need contiguous tensors")
    return t.data_ptr()


# ---------------------------------------------------------------------------------------------------------
# geometry of the reference's three conv flavours (code/ops.py:45-63) and of their gradients
# ---------------------------------------------------------------------------------------------------------
@dataclass
class Geom:
    S: int
    OS: int
    classes: List[Tuple[int, int, List[Tuple[int, int, int]]]]  # (ooy, oox, [(dy, dx, widx)])
    out_scale: Tuple[int, int] = (1, 1)  # OH = IH * num // den


@dataclass
class ConvSpec:
    kind: str  # "c3" conv3x3 s1 p1 | "c4s2" conv4x4 s2 p1 | "ct" conv-transpose k3 s2 p1 op1
    cin: int
    cout: int
    k: int = field(init=False)

    def __post_init__(self):
        self.k = 4 if self.kind == "c4s2" else 3

    @property
    def nslots(self):
        return self.k * self.k

    @property
    def weight_shape(self):
        return (self.cin, self.cout, 3, 3) if self.kind == "ct" else (self.cout, self.cin, self.k, self.k)

    def out_hw(self, h, w):
        if self.kind == "c3":
            return h, w
        if self.kind == "c4s2":
            return h // 2, w // 2
        return 2 * h, 2 * w

    # ---- forward: rows = cout, K = cin
    def fwd_geom(self):
        k = self.k
        if self.kind in ("c3", "c4s2"):
            taps = [(kh - 1, kw - 1, kh * k + kw) for kh in range(k) for kw in range(k)]
            return Geom(S=1 if self.kind == "c3" else 2, OS=1, classes=[(0, 0, taps)],
                        out_scale=(1, 1) if self.kind == "c3" else (1, 2))
        cls = []
        for a in (0, 1):
            for b in (0, 1):
                ys = [(1, 0)] if a == 0 else [(0, 1), (2, 0)]  # (kh, dy): oy = 2*iy - 1 + kh
                xs = [(1, 0)] if b == 0 else [(0, 1), (2, 0)]
                cls.append((a, b, [(dy, dx, kh * 3 + kw) for kh, dy in ys for kw, dx in xs]))
        return Geom(S=1, OS=2, classes=cls, out_scale=(2, 1))

    def fwd_pack(self):
        """(rows, K, s_row, s_k): packed[slot][row][k] = w[row*s_row + k*s_k + slot]"""
        kk = self.nslots
        if self.kind == "ct":
            return self.cout, self.cin, kk, self.cout * kk
        return self.cout, self.cin, self.cin * kk, kk

    # ---- input gradient: rows = cin, K = cout
    def dgrad_geom(self):
        k = self.k
        if self.kind == "c3":
            return Geom(S=1, OS=1, classes=[(0, 0, [(-(kh - 1), -(kw - 1), kh * 3 + kw) for kh in range(3)
                                                     for kw in range(3)])])
        if self.kind == "c4s2":
            cls = []
            for a in (0, 1):
                for b in (0, 1):
                    ys = [(1, 0), (3, -1)] if a == 0 else [(0, 1), (2, 0)]  # (kh, dy on the dout grid)
                    xs = [(1, 0), (3, -1)] if b == 0 else [(0, 1), (2, 0)]
                    cls.append((a, b, [(dy, dx, kh * 4 + kw) for kh, dy in ys for kw, dx in xs]))
            return Geom(S=1, OS=2, classes=cls, out_scale=(2, 1))
        taps = [(kh - 1, kw - 1, kh * 3 + kw) for kh in range(3) for kw in range(3)]
        return Geom(S=2, OS=1, classes=[(0, 0, taps)], out_scale=(1, 2))

    def dgrad_pack(self):
        kk = self.nslots
        if self.kind == "ct":
            return self.cin, self.cout, self.cout * kk, kk
        return self.cin, self.cout, kk, self.cin * kk

    # ---- weight gradient: dWt[t][a][b] = sum X[p*S + d_t][a] * Y[p][b]
    def wgrad_info(self):
        """returns (x_is_input, S, taps, ca, cb, s_a, s_b)"""
        k, kk = self.k, self.nslots
        taps = [(kh - 1, kw - 1) for kh in range(k) for kw in range(k)]
        if self.kind == "c3":
            return True, 1, taps, self.cin, self.cout, kk, self.cin * kk
        if self.kind == "c4s2":
            return True, 2, taps, self.cin, self.cout, kk, self.cin * kk
        return False, 2, taps, self.cout, self.cin, kk, self.cout * kk  # X = dout (2h grid), Y = in


def make_conv_desc(geom: Geom, dtype, N, IH, IW, cin_p, OH, OW, cout_p, act=L.ACT_NONE, mask_mode=L.MASK_NONE,
                   stats_mode=0, stats_groups=1, out_mode=L.OUT_NHWC, c_real=0, out_n_stride=0, tile_cfg=L.TILE_AUTO,
                   stats_replicas=1):
    d = L.ConvDesc()
    d.dtype = dtype
    d.N, d.IH, d.IW, d.Cin, d.OH, d.OW, d.Cout = N, IH, IW, cin_p, OH, OW, cout_p
    d.S, d.OS, d.ncls = geom.S, geom.OS, len(geom.classes)
    for i, (ooy, oox, taps) in enumerate(geom.classes):
        c = d.cls[i]
        c.ooy, c.oox, c.ntaps = ooy, oox, len(taps)
        for t, (dy, dx, w) in enumerate(taps):
            c.dy[t], c.dx[t], c.widx[t] = dy, dx, w
    d.act, d.mask_mode, d.stats_mode, d.stats_groups = act, mask_mode, stats_mode, stats_groups
    d.out_mode, d.c_real, d.out_n_stride, d.tile_cfg = out_mode, c_real, out_n_stride, tile_cfg
    d.stats_replicas = stats_replicas
    return d


def stats_replicas_for(n_pixels):
    """replica count for the per-channel statistics of a conv launch: keep <= ~64 workgroups (of >= 64 pixels) per replica."""
    tiles = max(1, n_pixels // 256)
    if tiles < 1024:  # measured: the extra zero + fold launches cost more than the contention below ~1000 workgroups
        return 1
    r = 1
    while r < 64 and tiles // r > 64:
        r *= 2
    return r


def reduce_replicas(src, replicas, stride, n, dst, accumulate=True):
    L.check(L.load().tg_reduce_replicas(_ptr(src), replicas, stride, n, _ptr(dst), int(accumulate), _stream()),
            "tg_reduce_replicas")


def conv(desc, x, w_packed, out, bias=None, res=None, mask=None, stats=None):
    L.check(L.load().tg_conv(C.byref(desc), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(res), _ptr(mask), _ptr(out),
                             _ptr(stats), _stream()), "tg_conv")


C3_CW_FORCE = None   # tests: True / False overrides TECOGAN_C3_CW for conv3x3_rw's Cin = 64 launches


def conv3x3_rw(x, w_packed, out, flip=False, bias=None, res=None, mask=None, mask_mode=L.MASK_NONE, act=L.ACT_NONE,
               stats=None, stats_mode=2, groups=1, max_workgroups=0, stats_replicas=1, cw=None):
    """3x3 stride-1 conv / input-gradient through the persistent register-weights kernel (csrc/conv3_rw.hip):
    x [N,H,W,Cin] -> out [N,H,W,Cout], bf16, Cin in {64,128}, Cout % 64 == 0"""
    N, H, W, cin = x.shape
    # 64 reduction channels: the eight-equal-waves form (csrc/conv3_cw.hip, round 5) unless TECOGAN_C3_CW=0
    if C3_CW_FORCE is not None:
        cw = C3_CW_FORCE
    elif cw is None:
        cw = tuning.current().c3_cw
    if mask_mode == L.MASK_RELU_BITS and mask is not None:
        cw = False   # (the 1-bit mask form exists on conv3_rw.hip only)
    cw = (cw and cin == 64 and (stats is None or C3_CW_FORCE)) or cin == 32   # (launches with statistics: 17.9 vs 17.0 us - they stay on
    #                                                                         conv3_rw; 32 reduction channels exist on conv3_cw only)
    fn, name = (L.load().tg_conv3x3_cw, "tg_conv3x3_cw") if cw else (L.load().tg_conv3x3_rw, "tg_conv3x3_rw")
    L.check(fn(tg_dtype(x.dtype), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(res), _ptr(mask), _ptr(out),
               _ptr(stats), N, H, W, cin, out.shape[3], int(flip), act,
               mask_mode if mask is not None else L.MASK_NONE, stats_mode, groups, stats_replicas,
               max_workgroups or persist_wgs(None), _stream()), name)


def rw_eligible(dtype_t, cin_p, cout_p, N, H, W, masked=False, extra="", dgrad=False, tu=None):
    """launch shapes routed to the persistent register-weights 3x3 kernel (csrc/conv3_rw.hip).  cin_p = reduction channels.
    Measured against tg_conv on the step's dense shapes (tools/mb_rw.py, profiles/r02_c_mb_rw.log):
      64 -> 64  @64x64   N=40  32.7 -> 20.0 us      64 -> 128 @128x128 N=40  154 -> 106 us     128 -> 128 @64x64 N=40 78.5 -> 74.1
      128 -> 128 @32x32  N=12  14.9 -> 11.9 us
    and slower ALONE on the rest.  In the step, though, a capped persistent launch is the better neighbour for the other lane than a
    grid of thousands of short workgroups, and since the Cin = 128 instantiations stopped spilling (round 3) three more classes pay
    there although tg_conv wins or ties alone (TECOGAN_RW_EXTRA, profiles/r03_r_rw_dma_ab.log): the trunk's input-gradients at 40 x
    32 x 32 (4.23 -> 4.205 ms/step), 128 -> 64 input-gradients (c30: -> 4.204) and masked 128 -> 128 ones (c32, with both: 4.195);
    the discriminator's stage 1 (64 -> 64 @64x64 N=12: 4.30) and stage 3 (16 x 16: 4.32) stayed on tg_conv.
    Round 4 (the kernel is wave-specialised; profiles/r04_l_rw_extra_s3.log, r04_x_rw_fwd_routing.log): stage 3 moved (s3), then every
    FORWARD launch of >= TECOGAN_RW_FWD_MIN = 4096 pixels (the chain's conv0 / c30 / c32 / c6, the discriminator's stage 1, config 5's
    HR stage: config 2 3.92 -> 3.80 ms, config 5 2840 -> 2946 frames/s at 160 workgroups, 3130 at 256) and stage 1's input-gradients
    in both halves (s1); with conv_trans.2's pair as two launches and the generator at 144 workgroups config 2 runs at 3.75 ms."""
    TU = tu if tu is not None else tuning.current()   # (tu: the caller's snapshot - engines route by what they were built with, and a
    _RW, _RW_EXTRA_ENV = TU.rw, TU.rw_extra           #  re-parse of ~50 environment variables per launch decision cost ~30 us of host time)   # 0: never, 1: where it measured faster, all | classes routed there for the STEP's sake
    if _RW == "0" or dtype_t not in (torch.bfloat16, torch.float16) or cin_p not in (64, 128) or cout_p % 64:
        return False
    if _RW == "all":
        return N * H * W >= 8192
    npix = N * H * W
    _RW_EXTRA = _RW_EXTRA_ENV + "," + extra if extra else _RW_EXTRA_ENV
    # (round 3 kept these to input-gradient launches: the forward ones were faster on tg_conv; no longer - see RW_FWD_MIN above)
    if not dgrad and TU.rw_fwd_min > 0 and npix >= TU.rw_fwd_min:
        return True
    if dgrad and "trunk" in _RW_EXTRA and cin_p == 64 and H == 32 and W == 32 and npix >= 32768:
        return True
    if dgrad and "c30" in _RW_EXTRA and cin_p == 128 and cout_p == 64 and npix >= 131072:
        return True
    if "s1" in _RW_EXTRA and cin_p == 64 and H >= 64 and npix >= 32768:         # (A/B only: slower)
        return True
    if "s3" in _RW_EXTRA and cin_p == 128 and cout_p == 128 and H == 16 and npix >= 2048:
        return True   # the discriminator's stage 3: 6.2 vs 7.8 us alone since the kernel is wave-specialised, step 4.035 -> 4.008 ms (r04_l)
    if cin_p == 64:
        return npix >= 131072
    if masked and npix >= 131072 and "m128" not in _RW_EXTRA:
        return False  # Cin = 128 has no registers for the early mask fetch: tg_conv's epilogue (all mask vectors in one round trip)
    return cout_p == 128 and (npix >= 131072 or (H == 32 and W == 32 and npix >= 8192))


def slot_table(n, device):
    return torch.arange(n, dtype=torch.int32, device=device)


def pack_weights(dtype_t, w, rows, K, s_row, s_k, nslots, slots, out=None):
    """w: fp32 PyTorch-layout weight tensor on device.  Returns packed tensor [nslots][K_p/chunk][rows_p][chunk]."""
    rows_p, K_p = pad32(rows), pad32(K)
    if out is None:
        out = torch.empty(nslots * rows_p * K_p, dtype=dtype_t, device=w.device)
    L.check(L.load().tg_pack_conv_weights(tg_dtype(dtype_t), _ptr(w), _ptr(out), rows, K, rows_p, K_p, s_row, s_k,
                                          nslots, _ptr(slots), _stream()), "tg_pack_conv_weights")
    return out


def make_wgrad_desc(dtype, N, XH, XW, cx_p, YH, YW, cy_p, S, taps, nsplit, taps_per_wg=0, y_sum=False):
    d = L.WgradDesc()
    d.taps_per_wg = taps_per_wg
    d.y_sum = 1 if y_sum else 0
    d.dtype, d.N, d.XH, d.XW, d.Cx, d.YH, d.YW, d.Cy, d.S = dtype, N, XH, XW, cx_p, YH, YW, cy_p, S
    d.ntaps, d.nsplit = len(taps), nsplit
    for t, (dy, dx) in enumerate(taps):
        d.dy[t], d.dx[t] = dy, dx
    return d


def wgrad(desc, x, y, slab):
    L.check(L.load().tg_wgrad(C.byref(desc), _ptr(x), _ptr(y), _ptr(slab), _stream()), "tg_wgrad")


def wgrad_multi(desc, jobs, njobs):
    L.check(L.load().tg_wgrad_multi(C.byref(desc), _ptr(jobs), njobs, _stream()), "tg_wgrad_multi")


def wgrad_tiles(N, YH, YW, S):
    tw, th = (16, 4) if S == 2 else ((32, 4) if YW > 16 else (16, 8))
    return N * ((YW + tw - 1) // tw) * ((YH + th - 1) // th)


def wgrad_finalize(slab, nsplit, ntaps, ca_p, cb_p, ca, cb, grad, s_a, s_b, slots, accumulate, bias_grad=None):
    stride = ntaps * ca_p * cb_p + (cb_p if bias_grad is not None else 0)
    L.check(L.load().tg_wgrad_finalize(_ptr(slab), nsplit, ntaps, ca_p, cb_p, ca, cb, _ptr(grad), s_a, s_b,
                                       _ptr(slots), int(accumulate), _ptr(bias_grad), stride, _stream()),
            "tg_wgrad_finalize")


def wgrad_nsplit(N, YH, YW, S, blocks=1):
    """Number of pixel splits.  The kernel runs one workgroup per CU (its accumulators take most of the register file), so
    the best split count is the one that gives every CU exactly one workgroup: 256 / (channel blocks) - measured per layer
    with tools/microbench.py.  Never more splits than pixel tiles."""
    tw, th = (16, 4) if S == 2 else ((32, 4) if YW > 16 else (16, 8))
    tiles = N * ((YW + tw - 1) // tw) * ((YH + th - 1) // th)
    return max(1, min(tiles, 256 // max(1, blocks)))


# Workgroups of a persistent launch (conv3_rw, wgrad): one per CU would be 256 - but a persistent workgroup holds its CU for
# the whole launch (60-125 us for the G backward's HR-stage launches), and the OTHER lane's small launches then queue for a
# CU the whole time: with 256, lane B's fake-half forward took 1.35 ms beside the G backward (0.6 ms alone) and lane A idled
# 0.6 ms at the end of every step waiting for it (tools/lane_ends.py).  160 leaves 96 CUs to the neighbour: the persistent
# launches lose ~10 %, the step 5.06 -> 4.71 ms (sweep 256/240/224/208/192/176/160/144/128/96: 5.06 5.05 4.91 4.88 4.75 4.79 4.71
# 4.84 4.89 5.27, profiles/r02_q_persist_wgs_sweep.log).  Per network: TECOGAN_PERSIST_WGS_G / _D (else TECOGAN_PERSIST_WGS).
PERSIST_WGS = 160   # the documented default (tuning.KNOBS); the live value is tuning.current().cap(None)


def persist_wgs(net):
    """cap for the persistent launches of network `net` ('G', 'D' or None).  Separate sweep with G = 160: D = 64/96/112/120/128/
    160/192/256 -> 4.98 4.62 4.64 4.63 4.65 4.72 4.88 4.95 ms; with D = 128: G = 128/144/160/168/176/192 -> 4.89 4.78 4.65 4.73 4.75
    4.76 (the discriminator's persistent launches - weight gradients, stage-2 convs - run beside the latency-bound chain and G
    backward tail and should hold even fewer CUs)"""
    return tuning.current().cap(net or None)


def persist_wgs_g_for(lr_pixels):
    """the generator's cap for a training step whose recurrent pass has `lr_pixels` = B * h * w pixels per launch, or None when
    the environment fixes it.  History (profiles/r03_o_rgb_bwd_ab.log, r03_r_rw_dma_ab.log): with the one-pass output-layer backward
    lane A became ~70 us shorter and 144 was best for configs[1] (G = 120/136/144/152/160 -> 4.39 4.33 4.33 4.36 4.38 ms); once the
    register-weights kernel stopped spilling, lane B got shorter too and the optimum moved back: 136/144/152/160 -> 4.277 4.264 4.249
    4.24 ms.  The configs[3] shard always preferred 160 (10.34 vs 10.47 ms)."""
    return tuning.current().cap_g_for(lr_pixels)


def persist_wgs_dreal_for(lr_pixels):
    """cap of the discriminator's persistent launches in its REAL half for a step of `lr_pixels` (see persist_wgs_g_for), or None.
    The real half runs beside the latency-bound chain and lane B then waits ~0.4 ms for the chain's last frame: with fewer
    workgroups than the fake half's 96 it loads the memory system less while the chain runs and still ends before the chain does -
    96 -> 80: 4.36 -> 4.32 ms/step; after the register-weights change 80 / 72 / 64 = 4.238 4.226 4.218 (48: +0.18 ms, the real half
    becomes the long pole).  Round 5: the chain got 0.05 ms shorter (resblock_ws), the real half became the phase's long pole at 72 and
    80 was the optimum again (3.54-3.56 -> 3.485-3.50 ms, profiles/r05_l_caps_resweep.log); at the end of round 5 the fake half's 96
    (3.37 -> 3.31-3.32 ms, profiles/r05_v_caps_write_through.log).  The configs[3] shard is not chain-bound:
    no gain there."""
    return tuning.current().cap_dreal_for(lr_pixels)


def wgrad_plan(N, YH, YW, S, ntaps, cx_p, cy_p, cap=None):
    """(nsplit, taps_per_wg).  Layers with few pixel tiles split the taps over workgroups (3 of 9 / 4 of 16 each): the fp32
    slab traffic (nsplit x taps x Cx x Cy x 4 B written, then read by the fold) is what bounds them; layers with thousands
    of tiles keep all taps in one workgroup (each staged tile is then used for every tap)."""
    tw, th = (16, 4) if S == 2 else ((32, 4) if YW > 16 else (16, 8))
    tiles = N * ((YW + tw - 1) // tw) * ((YH + th - 1) // th)
    blocks = wgrad_blocks(ntaps, cx_p, cy_p)
    if ntaps == 9 and tiles <= 512 and blocks == 1:  # measured: 21 vs 25 us on the 64-channel 32x32 trunk layers (N=40);
        return max(1, min(tiles, (cap or persist_wgs(None)) // 3)), 3       # slower on every larger layer (tools/microbench.py wgrad)
    return max(1, min(tiles, (cap or persist_wgs(None)) // max(1, blocks))), 0


def wgrad_blocks(ntaps, cx_p, cy_p):
    """(X-channel blocks) * (Y-channel blocks) of the wgrad launch, mirroring pick_cfg() in wgrad_mfma.hip."""
    if ntaps == 16:
        return (cx_p // 64) * (cy_p // 32)
    if cx_p % 64:
        return (cx_p // 32) * (cy_p // 64 if cy_p % 64 == 0 else cy_p // 32)
    if cy_p % 64:
        return (cx_p // 64) * (cy_p // 32)
    return (cx_p // 64) * (cy_p // 64)


# ---------------------------------------------------------------------------------------------------------
def nchw_to_nhwc(src, n_stride, dst, N, C_, H, W):
    Cp = dst.shape[-1]
    L.check(L.load().tg_nchw_to_nhwc(tg_dtype(dst.dtype), _ptr(src), n_stride, _ptr(dst), N, C_, Cp, H, W, _stream()),
            "tg_nchw_to_nhwc")


def nhwc_to_nchw(src, dst, n_stride, N, C_, H, W):
    Cp = src.shape[-1]
    L.check(L.load().tg_nhwc_to_nchw(tg_dtype(src.dtype), _ptr(src), _ptr(dst), n_stride, N, C_, Cp, H, W, _stream()),
            "tg_nhwc_to_nchw")


def resblock_fwd(x, w1, b1, w2, out_h, out_a, next_w=None, skip=True, ws=False):
    """next_w: (w1, w2) packed weights of the block launched next (L2 prefetch hint) or None; skip=False: conv-relu-conv;
    ws: the wave-specialised kernel of round 5 (csrc/resblock_ws.hip; no prefetch hint)"""
    N, H, W, C_ = x.shape
    if ws:
        L.check(L.load().tg_resblock_fwd_ws(tg_dtype(x.dtype), _ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(out_h), _ptr(out_a),
                                            N, H, W, C_, int(skip), _stream()), "tg_resblock_fwd_ws")
        return
    n1, n2 = next_w if next_w is not None else (None, None)
    L.check(L.load().tg_resblock_fwd(tg_dtype(x.dtype), _ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(out_h), _ptr(out_a),
                                     N, H, W, C_, int(skip), _ptr(n1), _ptr(n2), _stream()), "tg_resblock_fwd")


def conv3x3_rgb(x, w_packed, bias, out_buf, out_off, n_stride, c_real, act=L.ACT_SIGMOID):
    """the generator's output layer (conv 64 -> c_real <= 4, + bias, act) stored fp32 NCHW at element `out_off` of `out_buf`
    with `n_stride` elements between samples (csrc/conv_rgb.hip)"""
    N, H, W, cin = x.shape
    if out_buf.dtype != torch.float32 or out_off + (N - 1) * n_stride + c_real * H * W > out_buf.numel():
        raise L.TecoganHipError("tg_conv3x3_rgb: output window exceeds the fp32 buffer")
    L.check(L.load().tg_conv3x3_rgb(tg_dtype(x.dtype), _ptr(x), _ptr(w_packed), _ptr(bias), out_buf.data_ptr() + 4 * out_off,
                                    n_stride, c_real, N, H, W, cin, act, _stream()), "tg_conv3x3_rgb")


def rgb_bwd_workgroups(N, H, W, cap):
    """persistent workgroups (= slab slots, = the fold job's nsplit) of tg_conv3x3_rgb_bwd"""
    return max(1, min(N * ((H + 15) // 16) * ((W + 15) // 16), cap))


def conv3x3_rgb_bwd(dpre4, x, w_master, dx, slab, cap):
    """backward of the generator's output layer in one pass (csrc/rgb_bwd.hip): dx = relu-masked input gradient, slab = one
    partial-dW slot per workgroup; dpre4 [N,H,W,4] (content_loss(dpre_channels=4)), w_master the fp32 [3,64,3,3] weight"""
    N, H, W, cin = x.shape
    if dpre4.shape != (N, H, W, 4) or dx.shape != x.shape or w_master.dtype != torch.float32 or w_master.numel() != 3 * cin * 9:
        raise L.TecoganHipError("tg_conv3x3_rgb_bwd: operand shapes")
    lib = L.load()
    if slab.numel() < rgb_bwd_workgroups(N, H, W, cap) * int(lib.tg_conv3x3_rgb_bwd_slot_floats()):
        raise L.TecoganHipError("tg_conv3x3_rgb_bwd: slab too small")
    L.check(lib.tg_conv3x3_rgb_bwd(tg_dtype(x.dtype), _ptr(dpre4), _ptr(x), _ptr(w_master), _ptr(dx), _ptr(slab), N, H, W, cin, cap,
                                   _stream()), "tg_conv3x3_rgb_bwd")


def conv4s2_fwd(x, w_packed, bias, out, stats=None, groups=1, stats_replicas=1, max_workgroups=0):
    """conv k4 s2 p1 forward with compile-time taps; stats: `stats_replicas` blocks of [groups][2][Cout], accumulated;
    max_workgroups: 0 = one workgroup per unit, else a capped grid whose workgroups walk the units"""
    N, H, W, cin = x.shape
    L.check(L.load().tg_conv4s2_fwd_capped(tg_dtype(x.dtype), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(out), _ptr(stats),
                                           groups, stats_replicas, N, H, W, cin, out.shape[3], max_workgroups, _stream()),
            "tg_conv4s2_fwd_capped")


def conv4s2_fwd_cw(x, w_packed, bias, out, stats=None, groups=1, stats_replicas=1, max_workgroups=0):
    """the same layer with the weights in registers (csrc/conv_s2_cw.hip, round 6): Cin in {64, 128}, Cout % 64 == 0, 16-bit types"""
    N, H, W, cin = x.shape
    L.check(L.load().tg_conv4s2_fwd_cw(tg_dtype(x.dtype), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(out), _ptr(stats), groups,
                                       stats_replicas, N, H, W, cin, out.shape[3], max_workgroups or persist_wgs(None), _stream()),
            "tg_conv4s2_fwd_cw")


def s2_cw_ok(dtype, c_red, c_out, H, W):
    """shapes csrc/conv_s2_cw.hip takes: 16-bit elements, 64 / 128 reduction channels, output channels % 64, even input size"""
    return dtype in (torch.bfloat16, torch.float16) and c_red in (64, 128) and c_out % 64 == 0 and H % 2 == 0 and W % 2 == 0


def conv4s2_dgrad(dout, wb_packed, din, mask=None, mask_mode=L.MASK_NONE):
    """input-gradient of conv k4 s2 p1: dout [N,OH,OW,Cout] -> din [N,2OH,2OW,Cin] (four sub-pixel classes in one launch),
    optionally times act'(mask)"""
    N, OH, OW, cout = dout.shape
    L.check(L.load().tg_conv4s2_dgrad(tg_dtype(dout.dtype), _ptr(dout), _ptr(wb_packed), _ptr(din), N, OH, OW, cout,
                                      din.shape[3], _ptr(mask), mask_mode, _stream()), "tg_conv4s2_dgrad")


def conv4s2_dgrad_cw(dout, wb_packed, din, mask=None, mask_mode=L.MASK_NONE, max_workgroups=0):
    """the same input-gradient on persistent workgroups with class-specialised waves (csrc/conv4s2d_cw.hip): reduction channels 64 / 128"""
    N, OH, OW, cout = dout.shape
    L.check(L.load().tg_conv4s2_dgrad_cw(tg_dtype(dout.dtype), _ptr(dout), _ptr(wb_packed), _ptr(din), N, OH, OW, cout,
                                         din.shape[3], _ptr(mask), mask_mode, max_workgroups or persist_wgs(None), _stream()),
            "tg_conv4s2_dgrad_cw")


def convt_dgrad(dout, wb_packed, din):
    """input-gradient of conv-transpose k3 s2: dout [N,2H,2W,Cout] -> din [N,H,W,Cin] (3x3-window stride-2 gather)"""
    N, OH, OW, cout = dout.shape
    L.check(L.load().tg_convt_dgrad(tg_dtype(dout.dtype), _ptr(dout), _ptr(wb_packed), _ptr(din), N, OH, OW, cout,
                                    din.shape[3], _stream()), "tg_convt_dgrad")


def convt_dgrad_cw(dout, wb_packed, din, max_workgroups=0):
    """the same input-gradient with the weights in registers (csrc/conv_s2_cw.hip, round 6): Cout (the reduction) in {64, 128}"""
    N, OH, OW, cout = dout.shape
    L.check(L.load().tg_convt_dgrad_cw(tg_dtype(dout.dtype), _ptr(dout), _ptr(wb_packed), _ptr(din), N, OH, OW, cout,
                                       din.shape[3], max_workgroups or persist_wgs(None), _stream()), "tg_convt_dgrad_cw")


def convt_fwd(x, w_packed, bias, out, act=L.ACT_NONE):
    """conv-transpose k3 s2 forward, all four sub-pixel classes per workgroup; x [N,H,W,Cin] -> out [N,2H,2W,Cout]"""
    N, H, W, cin = x.shape
    L.check(L.load().tg_convt_fwd(tg_dtype(x.dtype), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(out), N, H, W, cin,
                                  out.shape[3], act, _stream()), "tg_convt_fwd")


def convt_fwd_cw(x, w_packed, bias, out, act=L.ACT_NONE, max_workgroups=0, relu_bits=None):
    """the same layer with class-specialised waves (csrc/convt_cw.hip): Cin in {64, 128}, Cout % 64 == 0.  relu_bits (uint8
    [N,2H,2W,Cout/8], act = ReLU): also the 1-bit mask of the output, for the input-gradient of the layer above (L.MASK_RELU_BITS)"""
    N, H, W, cin = x.shape
    L.check(L.load().tg_convt_fwd_cw(tg_dtype(x.dtype), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(out), N, H, W, cin,
                                     out.shape[3], act, _ptr(relu_bits), max_workgroups or persist_wgs(None), _stream()), "tg_convt_fwd_cw")


def resblock2_fwd_ws(x, w1a, b1a, w2a, w1b, b1b, w2b, out_h1, out_a1, out_h2, out_a2):
    """two consecutive residual blocks in one launch of the stream-first kernel (csrc/exp/resblock2_ws.hip); out_h1 / out_h2 may be None"""
    N, H, W, C_ = x.shape
    L.check(L.load().tg_resblock2_fwd_ws(tg_dtype(x.dtype), _ptr(x), _ptr(w1a), _ptr(b1a), _ptr(w2a), _ptr(w1b), _ptr(b1b), _ptr(w2b),
                                         _ptr(out_h1), _ptr(out_a1), _ptr(out_h2), _ptr(out_a2), N, H, W, C_, _stream()),
            "tg_resblock2_fwd_ws")


def resblock2_fwd(x, w1a, b1a, w2a, w1b, b1b, w2b, out_h1, out_a1, out_h2, out_a2, next_w=None):
    """two consecutive residual blocks in one launch (csrc/exp/resblock2.hip); next_w: the four packed weight images of the next
    launch (L2 prefetch hint) or None"""
    N, H, W, C_ = x.shape
    nxt = None
    if next_w is not None:
        nxt = (C.c_void_p * 4)(*[t.data_ptr() for t in next_w])
    L.check(L.load().tg_resblock2_fwd(tg_dtype(x.dtype), _ptr(x), _ptr(w1a), _ptr(b1a), _ptr(w2a), _ptr(w1b), _ptr(b1b), _ptr(w2b),
                                      _ptr(out_h1), _ptr(out_a1), _ptr(out_h2), _ptr(out_a2), N, H, W, C_, nxt, _stream()),
            "tg_resblock2_fwd")


def resblock_bwd_pp(dout, w2b, h, w1b, out_dh, out_din, max_workgroups=0):
    """both input-gradients of a residual block as one persistent, tile-pipelined launch (csrc/exp/resblock_pp.hip)"""
    N, H, W, C_ = dout.shape
    L.check(L.load().tg_resblock_bwd_pp(tg_dtype(dout.dtype), _ptr(dout), _ptr(w2b), _ptr(h), _ptr(w1b), _ptr(out_dh), _ptr(out_din),
                                        N, H, W, C_, int(max_workgroups), _stream()), "tg_resblock_bwd_pp")


def resblock_bwd(dout, w2b, h, w1b, out_dh, out_din, next_w=None):
    """input-gradient of conv-relu-conv-skip in one launch; w*b = dgrad packings, h = saved forward activation"""
    N, H, W, C_ = dout.shape
    n1, n2 = next_w if next_w is not None else (None, None)
    L.check(L.load().tg_resblock_bwd(tg_dtype(dout.dtype), _ptr(dout), _ptr(w2b), _ptr(h), _ptr(w1b), _ptr(out_dh),
                                     _ptr(out_din), N, H, W, C_, _ptr(n1), _ptr(n2), _stream()), "tg_resblock_bwd")


def maxpool2(src, dst):
    N, H, W, C_ = src.shape
    L.check(L.load().tg_maxpool2(tg_dtype(src.dtype), _ptr(src), _ptr(dst), N, H, W, C_, _stream()), "tg_maxpool2")


def vgg_input(src_nchw, dst, scale, shift3):
    """fp32 [N,3,H,W] -> NHWC [N,H,W,32] of dst.dtype: scale * x + shift3[c] on the 3 live channels (csrc/vgg.hip)"""
    N, _, H, W = src_nchw.shape
    sh = (C.c_float * 3)(*[float(v) for v in shift3])
    L.check(L.load().tg_vgg_input(tg_dtype(dst.dtype), _ptr(src_nchw), _ptr(dst), N, H, W, float(scale), sh, _stream()),
            "tg_vgg_input")


def cosine_loss(fg, ft, dg, coef, relu_mask, acc, loss_scale=None):
    """per-pixel cosine similarity of two NHWC feature maps; acc[0] += sum; dg = coef [* loss_scale] * d(sum cos)/d(fg)"""
    N, H, W, C_ = fg.shape
    L.check(L.load().tg_cosine_loss(tg_dtype(fg.dtype), _ptr(fg), _ptr(ft), _ptr(dg), N * H * W, C_, float(coef),
                                    int(bool(relu_mask)), _ptr(acc), _ptr(loss_scale), _stream()), "tg_cosine_loss")


def maxpool2_bwd(a, dpool, out, res=None, relu_mask=True):
    """relu_mask: False / 0 none, True / 1 ReLU'(a), 2 LeakyReLU(0.2)'(a)"""
    N, H, W, C_ = a.shape
    L.check(L.load().tg_maxpool2_bwd(tg_dtype(a.dtype), _ptr(a), _ptr(dpool), _ptr(res), _ptr(out), N, H, W, C_,
                                     int(relu_mask), _stream()), "tg_maxpool2_bwd")


def vgg_input_grad(dx, gen_nchw, dpre, scale, bias_acc=None):
    N, H, W, _ = dx.shape
    L.check(L.load().tg_vgg_input_grad(tg_dtype(dx.dtype), _ptr(dx), _ptr(gen_nchw), _ptr(dpre), N, H, W, float(scale),
                                       _ptr(bias_acc), _stream()), "tg_vgg_input_grad")


def resample_u8(frames_u8, plan, out):
    """PIL-BILINEAR resize + ToTensor of decoded uint8 frames on the GPU (resize.ResizePlan holds the coefficient tables)"""
    N, H, W, _ = frames_u8.shape
    tmp = torch.empty(N, H, plan.out_w, 3, dtype=torch.uint8, device=frames_u8.device)
    L.check(L.load().tg_resample_u8(_ptr(frames_u8), _ptr(tmp), _ptr(out), _ptr(plan.bw), _ptr(plan.kw), plan.kw_n,
                                    _ptr(plan.bh), _ptr(plan.kh), plan.kh_n, N, H, W, plan.out_h, plan.out_w, _stream()),
            "tg_resample_u8")


def up2_bilinear(src, dst):
    N, H, W, C_ = src.shape
    L.check(L.load().tg_up2_bilinear(tg_dtype(src.dtype), _ptr(src), _ptr(dst), N, H, W, C_, _stream()), "tg_up2_bilinear")


def up2_bilinear_bwd(ddst, dsrc, lrelu_mask=None):
    """backward of tg_up2_bilinear: ddst [N,2H,2W,C] -> dsrc [N,H,W,C] (x LeakyReLU(0.2)'(lrelu_mask) when given)"""
    N, H, W, C_ = dsrc.shape
    L.check(L.load().tg_up2_bilinear_bwd(tg_dtype(dsrc.dtype), _ptr(ddst), _ptr(lrelu_mask), _ptr(dsrc), N, H, W, C_, _stream()),
            "tg_up2_bilinear_bwd")


def tanh24_bwd(dout, out, dpre):
    """dout / out fp32 [N,2,H,W] (f_net's result and its gradient) -> dpre NHWC [N,H,W,32] = dout * (24 - out^2 / 24)"""
    N, H, W, _ = dpre.shape
    L.check(L.load().tg_tanh24_bwd(tg_dtype(dpre.dtype), _ptr(dout), _ptr(out), _ptr(dpre), N, H, W, _stream()), "tg_tanh24_bwd")


def warp_grid_grad(img, img_off, grid, grid_off, ref, ref_off, dgrid, dgrid_off, N, C_, IH, IW, GH, GW, coef, loss_acc=None,
                   loss_scale=None):
    """LR warp loss and its gradient w.r.t. the sampling grid (include/tecogan_hip.h, tg_warp_grid_grad)"""
    L.check(L.load().tg_warp_grid_grad(_ptr(img), _ptr(img_off), _ptr(grid), _ptr(grid_off), _ptr(ref), _ptr(ref_off), _ptr(dgrid),
                                       _ptr(dgrid_off), _ptr(loss_acc), N, C_, IH, IW, GH, GW, float(coef), _ptr(loss_scale),
                                       _stream()), "tg_warp_grid_grad")


def to_nhwc(x, dtype_t):
    """[N,C,H,W] fp32 device tensor -> [N,H,W,pad32(C)] of dtype_t."""
    N, C_, H, W = x.shape
    x = x.contiguous().float()
    out = torch.empty(N, H, W, pad32(C_), dtype=dtype_t, device=x.device)
    nchw_to_nhwc(x, C_ * H * W, out, N, C_, H, W)
    return out


def to_nchw(x, C_):
    N, H, W, _ = x.shape
    out = torch.empty(N, C_, H, W, dtype=torch.float32, device=x.device)
    nhwc_to_nchw(x, out, C_ * H * W, N, C_, H, W)
    return out


def up4_planes(src, src_off, dst, dst_off, nplanes, h, w, pre=1.0, post_a=1.0, post_b=0.0):
    L.check(L.load().tg_up4_planes(_ptr(src), _ptr(src_off), _ptr(dst), _ptr(dst_off), nplanes, h, w, pre, post_a,
                                   post_b, _stream()), "tg_up4_planes")


def copy_blocks(src, src_off, dst, dst_off, nblocks, length):
    L.check(L.load().tg_copy_blocks(_ptr(src), _ptr(src_off), _ptr(dst), _ptr(dst_off), nblocks, length, _stream()),
            "tg_copy_blocks")


def warp_nchw(img, img_off, grid, grid_off, N, C_, IH, IW, GH, GW, fp16_grid, out=None, corner=None, sq_ref=None,
              sq_off=None, loss_acc=None):
    L.check(L.load().tg_warp_nchw(_ptr(img), _ptr(img_off), _ptr(grid), _ptr(grid_off), _ptr(out), _ptr(corner),
                                  _ptr(sq_ref), _ptr(sq_off), _ptr(loss_acc), N, C_, IH, IW, GH, GW, int(fp16_grid),
                                  _stream()), "tg_warp_nchw")


def _sub_ptr(t, elem_off):
    return C.c_void_p(t.data_ptr() + elem_off * t.element_size())


def gen_input(lr, lr_off, lr_n_stride, prev, prev_off, prev_n_stride, grid, grid_off, grid_n_stride, dst, B, h, w):
    """lr/prev/grid are fp32 device buffers addressed by element offset + per-sample stride."""
    lib = L.load()
    L.check(lib.tg_gen_input(tg_dtype(dst.dtype), _sub_ptr(lr, lr_off), lr_n_stride,
                             None if prev is None else _sub_ptr(prev, prev_off), prev_n_stride,
                             None if grid is None else _sub_ptr(grid, grid_off), grid_n_stride, _ptr(dst), B, h, w,
                             _stream()), "tg_gen_input")


def d_assemble(x, y, gen, tvel, dst, B, T, K, h, border, half=-1):
    L.check(L.load().tg_d_assemble(tg_dtype(dst.dtype), _ptr(x), _ptr(y), _ptr(gen), _ptr(tvel), _ptr(dst), B, T, K, h,
                                   border, half, _stream()), "tg_d_assemble")


def bn_apply(z, stats, gamma, beta, y, save, N, HW, C_, groups, act, skip=None, running_mean=None, running_var=None,
             eps=1e-3, momentum=0.1, nbt=None, replicas=1):
    """stats: `replicas` blocks of [groups][2][C] sums (see include/tecogan_hip.h, "replica blocks")"""
    L.check(L.load().tg_bn_apply(tg_dtype(z.dtype), _ptr(z), _ptr(stats), replicas, _ptr(gamma), _ptr(beta), _ptr(skip), _ptr(y),
                                 _ptr(running_mean), _ptr(running_var), _ptr(save), N, HW, C_, groups, act, eps,
                                 momentum, _ptr(nbt), _stream()), "tg_bn_apply")


def bn_bwd_reduce(dy, yact, z, save, red, N, HW, C_, groups, act, replicas=1):
    L.check(L.load().tg_bn_bwd_reduce(tg_dtype(z.dtype), _ptr(dy), _ptr(yact), _ptr(z), _ptr(save), _ptr(red), replicas, N, HW,
                                      C_, groups, act, _stream()), "tg_bn_bwd_reduce")


def bn_bwd_apply(dy, yact, z, save, red, gamma, dz, dgamma, dbeta, N, HW, C_, groups, act, replicas=1, red_raw=False):
    """red_raw: `red` holds (sum dy, sum dy * z) from the epilogue of the conv launch that produced dy (Conv.dgrad bn_sums)"""
    L.check(L.load().tg_bn_bwd_apply(tg_dtype(z.dtype), _ptr(dy), _ptr(yact), _ptr(z), _ptr(save), _ptr(red), replicas,
                                     _ptr(gamma), _ptr(dz), _ptr(dgamma), _ptr(dbeta), N, HW, C_, groups, act, int(red_raw),
                                     _stream()), "tg_bn_bwd_apply")


_BN_COOP_MAX = None


def bn_bwd_coop_ok(N, HW, C_, groups, dtype_t):
    """does tg_bn_bwd_coop take this tensor (all its workgroups co-resident: csrc/elementwise.hip, bn_bwd_coop_kernel)"""
    global _BN_COOP_MAX
    if _BN_COOP_MAX is None:
        _BN_COOP_MAX = int(L.load().tg_bn_bwd_coop_max_workgroups())
    rows = 256 // (C_ // (4 if dtype_t == torch.float32 else 8))
    npix = (N // groups) * HW
    return -(-npix // (rows * 8)) * groups <= _BN_COOP_MAX


def bn_bwd_coop(dy, yact, z, save, red, gamma, dz, dgamma, dbeta, N, HW, C_, groups, act, bar, replicas=1):
    """batch-norm backward (sums, grid-wide wait, apply) in one launch; `bar`: one zeroed word of the accumulator arena"""
    L.check(L.load().tg_bn_bwd_coop(tg_dtype(z.dtype), _ptr(dy), _ptr(yact), _ptr(z), _ptr(save), _ptr(red), replicas,
                                    _ptr(gamma), _ptr(dz), _ptr(dgamma), _ptr(dbeta), N, HW, C_, groups, act, _ptr(bar),
                                    _stream()), "tg_bn_bwd_coop")


_BN_FUSED_MAX = None


def bn_bwd_fused_max_pixels():
    """pixels per group up to which tg_bn_bwd_fused runs the whole backward pass of a batch norm in one launch"""
    global _BN_FUSED_MAX
    if _BN_FUSED_MAX is None:
        _BN_FUSED_MAX = int(L.load().tg_bn_bwd_fused_max_pixels())
    return _BN_FUSED_MAX


def bn_bwd_fused(dy, yact, z, save, gamma, dz, dgamma, dbeta, N, HW, C_, groups, act):
    """reduce + apply of a small tensor in one launch (csrc/elementwise.hip: bn_bwd_fused_kernel)"""
    L.check(L.load().tg_bn_bwd_fused(tg_dtype(z.dtype), _ptr(dy), _ptr(yact), _ptr(z), _ptr(save), _ptr(gamma), _ptr(dz),
                                     _ptr(dgamma), _ptr(dbeta), N, HW, C_, groups, act, _stream()), "tg_bn_bwd_fused")


def fc_head_fwd(feat, w, b, prob, N, HW, C_, Cp):
    L.check(L.load().tg_fc_head_fwd(tg_dtype(feat.dtype), _ptr(feat), _ptr(w), _ptr(b), _ptr(prob), N, HW, C_, Cp,
                                    _stream()), "tg_fc_head_fwd")


def fc_head_bwd(feat, w, dlogit, dfeat, dw, db, N, HW, C_, Cp):
    L.check(L.load().tg_fc_head_bwd(tg_dtype(feat.dtype), _ptr(feat), _ptr(w), _ptr(dlogit), _ptr(dfeat), _ptr(dw),
                                    _ptr(db), N, HW, C_, Cp, _stream()), "tg_fc_head_bwd")


def d_tail_ok(n_per_group, H4, C4, C5):
    """shapes the tail launches take (csrc/d_tail.hip, check_tail)"""
    return n_per_group * H4 * H4 <= int(L.load().tg_d_tail_max_pixels()) and n_per_group <= 256 and C4 == 64 and \
        1 <= C5 <= 4 and C5 * (H4 // 2) ** 2 <= 256 and H4 % 2 == 0


def d_tail_fwd(z4, stats4, replicas, gamma4, beta4, rm4, rv4, nbt4, save4, n4, w5, z5, gamma5, beta5, rm5, rv5, nbt5, save5, n5,
               fc_w, fc_b, prob, N, H4, C5, groups, scratch, eps=1e-3, momentum=0.1):
    """everything of discriminator.forward behind block4's convolution in one launch (include/tecogan_hip.h, tg_d_tail_fwd)"""
    L.check(L.load().tg_d_tail_fwd(tg_dtype(z4.dtype), _ptr(z4), _ptr(stats4), replicas, _ptr(gamma4), _ptr(beta4), _ptr(rm4), _ptr(rv4),
                                   _ptr(nbt4), _ptr(save4), _ptr(n4), _ptr(w5), _ptr(z5), _ptr(gamma5), _ptr(beta5), _ptr(rm5), _ptr(rv5),
                                   _ptr(nbt5), _ptr(save5), _ptr(n5), _ptr(fc_w), _ptr(fc_b), _ptr(prob), N, H4, z4.shape[3], C5,
                                   z5.shape[3], groups, eps, momentum, _ptr(scratch), _stream()), "tg_d_tail_fwd")


def d_tail_scratch(N, H4, groups, device):
    """the zeroed scratch of the tail launches for N samples (every launch leaves it zero again)"""
    return torch.zeros(int(L.load().tg_d_tail_scratch_floats(N, H4, groups)), device=device)


def d_tail_bwd(dlogit, prob, cfg, loss_scale, seed_real, n5, z5, save5, gamma5, fc_w, w5, n4, z4, save4, gamma4, dz5, dn4, dz4,
               g_fc_w, g_fc_b, dgamma5, dbeta5, dgamma4, dbeta4, N, H4, C5, groups, scratch):
    """... and of its backward pass between d(loss)/d(logit) and block4's input-gradient (tg_d_tail_bwd)"""
    L.check(L.load().tg_d_tail_bwd(tg_dtype(z4.dtype), _ptr(dlogit), _ptr(prob), _ptr(cfg), _ptr(loss_scale), int(seed_real), _ptr(n5),
                                   _ptr(z5), _ptr(save5), _ptr(gamma5), _ptr(fc_w), _ptr(w5), _ptr(n4), _ptr(z4), _ptr(save4),
                                   _ptr(gamma4), _ptr(dz5), _ptr(dn4), _ptr(dz4), _ptr(g_fc_w), _ptr(g_fc_b), _ptr(dgamma5),
                                   _ptr(dbeta5), _ptr(dgamma4), _ptr(dbeta4), N, H4, z4.shape[3], C5, z5.shape[3], groups, _ptr(scratch),
                                   _stream()),
            "tg_d_tail_bwd")


def absdiff_sum(a, b, acc, acc_idx, npix, C_, Cp):
    L.check(L.load().tg_absdiff_sum(tg_dtype(a.dtype), _ptr(a), _ptr(b), _sub_ptr(acc, acc_idx), npix, C_, Cp,
                                    _stream()), "tg_absdiff_sum")


def absdiff_sum_multi(dtype_t, jobs, njobs, blocks_per_job=128):
    """jobs: int64 device table, njobs x {a ptr, b ptr, acc ptr, npix, C, Cp} (include/tecogan_hip.h)"""
    L.check(L.load().tg_absdiff_sum_multi(tg_dtype(dtype_t), _ptr(jobs), njobs, blocks_per_job, _stream()), "tg_absdiff_sum_multi")


def content_loss(gen, y, dpre, acc, B, T, H, W, gscale, t0=0, t1=None, pp_T=0, pp_coef=0.0, loss_scale=None, bias_acc=None):
    """loss_scale: device float (fp16 mode) multiplied into every backward seed - here d(loss)/d(pre-sigmoid);
    bias_acc: 3 floats that receive the channel sums of dpre (the output layer's bias gradient); default acc[8:11];
    dpre: [..., 32] (padded conv operand) or [..., 4] (16-bit: the compact operand of conv3x3_rgb_bwd)"""
    dt = tg_dtype(dpre.dtype) if dpre is not None else L.TG_F32
    L.check(L.load().tg_content_loss(dt, _ptr(gen), _ptr(y), _ptr(dpre), _ptr(acc), B, T, H, W, gscale, t0,
                                     T if t1 is None else t1, pp_T, pp_coef, _ptr(loss_scale), _ptr(bias_acc),
                                     int(dpre.shape[-1]) if dpre is not None else 32, _stream()),
            "tg_content_loss")


def dlogit_real(prob, dlogit, tb, cfg, loss_scale=None):
    L.check(L.load().tg_dlogit_real(_ptr(prob), _ptr(dlogit), tb, _ptr(cfg), _ptr(loss_scale), _stream()), "tg_dlogit_real")


def loss_finalize(prob, acc, scalars, dlogit, tb, cfg, loss_scale=None):
    L.check(L.load().tg_loss_finalize(_ptr(prob), _ptr(acc), _ptr(scalars), _ptr(dlogit), tb, _ptr(cfg), _ptr(loss_scale),
                                      _stream()), "tg_loss_finalize")


def adam_hyper(lr, beta1, beta2, eps, step, grad_scale=1.0):
    return [lr, beta1, beta2, eps, 1.0 - beta1 ** step, 1.0 - beta2 ** step, grad_scale, float(step)]


def adam(p, g, m, v, hyper_dev, scaler=None, which=0):
    """scaler: the 8-float loss-scale state of the fp16 mode (then g is divided by the scale and the update is skipped when
    found_inf[which] is set); None: plain tg_adam"""
    if scaler is None:
        L.check(L.load().tg_adam(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper_dev), _stream()), "tg_adam")
    else:
        L.check(L.load().tg_adam_scaled(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper_dev), _ptr(scaler),
                                        which, _stream()), "tg_adam_scaled")


def check_finite(g, flag):
    L.check(L.load().tg_check_finite(_ptr(g), g.numel(), _ptr(flag), _stream()), "tg_check_finite")


def scaler_update(state, growth=2.0, backoff=0.5, interval=2000):
    """the two GradScaler.update() calls of one step (torch defaults: growth 2, backoff 0.5, interval 2000)"""
    L.check(L.load().tg_scaler_update(_ptr(state), float(growth), float(backoff), int(interval), _stream()),
            "tg_scaler_update")

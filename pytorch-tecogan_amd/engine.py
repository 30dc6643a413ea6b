"""Layer graphs of the generator and discriminator on the HIP kernels, forward and hand-derived backward.

Nothing here uses torch autograd or torch math: torch supplies device buffers, streams and (optionally) hipGraph
capture.  All activations are NHWC with channels padded to 32; parameters live in one flat fp32 buffer per network
(PyTorch layouts, so state_dict keys/shapes are those of the reference: SURVEY.md 8b), with matching flat buffers
for gradients and the Adam moments - one all-reduce and one Adam launch per network.

Reference behaviour reproduced (file:line into the reference):
  generator.forward        code/models.py:78-86      discriminator.forward   code/models.py:125-146
  detach structure         code/train.py:90,108,199  (no gradient through warp / recurrence / from D into G), hence the
                           T generator passes are independent in backward and are run as ONE batch of T*B samples.
"""
from collections import OrderedDict

import torch

from . import _lib as L
from . import kernels as K
from . import tuning
from .kernels import ConvSpec, pad32


def _align32(n):
    return (n + 31) // 32 * 32


class FlatParams:
    """One contiguous fp32 buffer (plus grad / exp_avg / exp_avg_sq twins) holding every parameter of a network.
    Each tensor's slot is padded to 32 floats so that kernels can read channel-padded bias / BN vectors in place."""

    def __init__(self, shapes, device, train=True):
        self.shapes = OrderedDict(shapes)
        self.offsets = OrderedDict()
        off = 0
        for name, shp in self.shapes.items():
            n = 1
            for s in shp:
                n *= s
            self.offsets[name] = (off, n)
            off += _align32(n)
        self.total = off
        self.device = device
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        # frozen networks (the VGG feature extractor) carry no gradient / Adam-moment twins
        self.g = torch.zeros_like(self.p) if train else None
        self.m = torch.zeros_like(self.p) if train else None
        self.v = torch.zeros_like(self.p) if train else None

    def view(self, buf, name):
        off, n = self.offsets[name]
        return buf[off:off + n].view(self.shapes[name])

    def padded(self, buf, name):
        off, n = self.offsets[name]
        return buf[off:off + _align32(n)]

    def load(self, tensors):
        for name in self.shapes:
            self.view(self.p, name).copy_(tensors[name].to(self.device, torch.float32))

    def named_views(self, buf):
        return OrderedDict((n, self.view(buf, n)) for n in self.shapes)


class Workspace:
    """Grow-only scratch (weight-gradient slabs) and a zero-initialised arena for per-step accumulators."""

    def __init__(self, device):
        self.device = device
        self.slab = torch.empty(0, dtype=torch.float32, device=device)
        self.frozen = False

    def get_slab(self, nfloats):
        if self.slab.numel() < nfloats:
            if self.frozen:
                raise L.TecoganHipError("workspace grew after graph capture")
            self.slab = torch.empty(nfloats, dtype=torch.float32, device=self.device)
        return self.slab


class ShapeSets:
    """Activation / gradient buffer sets of an engine, one per launch shape, kept alive while in use.

    A captured hipGraph holds raw pointers into the set it was captured on, and the module surface (generator.forward,
    .recurrent, discriminator.forward) may run the same engine at another shape between two training steps.  So switching
    shapes SELECTS another set instead of freeing the current one; sets pinned by a live TecoGANStep are never dropped,
    the others are dropped oldest-first beyond `keep`."""

    def __init__(self, keep=2):
        self.sets, self.pinned, self.keep = OrderedDict(), set(), keep

    def get(self, shape, make):
        ent = self.sets.get(shape)
        if ent is None:
            ent = self.sets[shape] = make()
            loose = [k for k in self.sets if k not in self.pinned and k != shape]
            for k in loose[:max(0, len(loose) + 1 - self.keep)]:
                del self.sets[k]
        else:
            self.sets.move_to_end(shape)
        return ent

    def pin(self, shape):
        self.pinned.add(shape)

    def unpin(self, shape):
        self.pinned.discard(shape)

    def drop(self, shape):
        """forgets an unpinned set at once (TecoGANStep.close: the configuration it served is gone)"""
        if shape not in self.pinned:
            self.sets.pop(shape, None)


def release_plans(engine):
    """Forgets every launch plan of `engine` that is tied to buffer ADDRESSES of a buffer set (weight-gradient work lists and
    their slabs, the output layer's backward plan, fold tables) and lifts the post-capture freeze.  TecoGANStep.close() calls it:
    the step's buffer sets are dropped there, a replacement step allocates new addresses, and plans keyed by the old ones would
    miss under `ws.frozen` ("new wgrad shape after graph capture") - and their slabs (~25 MB per launch) would never be freed."""
    engine.ws.frozen = False
    for name in ("hr_list", "trunk_group", "res_group", "wlist"):
        lst = getattr(engine, name, None)
        if lst is not None:
            lst.cache.clear()
            lst.items = []
    if getattr(engine, "_rgb_cache", None) is not None:
        engine._rgb_cache.clear()
    fin = getattr(engine, "finalizer", None)
    if fin is not None:
        fin.tables.clear()
    convs = engine.convs.values() if isinstance(engine.convs, dict) else engine.convs
    for c in convs:
        c._desc = {k: v for k, v in c._desc.items() if k[0] != "w"}   # (shape-keyed conv descriptors hold no addresses)
        c.fin_job = None


TU = tuning.current   # every scheduling / routing knob comes from this object (tuning.KNOBS: defaults + evidence)


class Conv:
    """One conv / conv-transpose layer of the reference: packed weights + forward / dgrad / wgrad launches."""

    def __init__(self, flat, wname, bname, spec, dtype_t, ws, need_dgrad=True, tile=L.TILE_AUTO):
        self.flat, self.spec, self.dt, self.ws = flat, spec, dtype_t, ws
        self.tu = TU()
        self.tg = K.tg_dtype(dtype_t)
        self.w = flat.view(flat.p, wname)
        self.gw = flat.view(flat.g, wname) if flat.g is not None else None
        self.bias = flat.padded(flat.p, bname) if bname else None
        self.gbias = flat.padded(flat.g, bname) if (bname and flat.g is not None) else None
        dev = flat.device
        self.slots = K.slot_table(spec.nslots, dev)
        self.cin_p, self.cout_p = pad32(spec.cin), pad32(spec.cout)
        if self.bias is not None and self.bias.numel() < self.cout_p:
            raise L.TecoganHipError("bias slot smaller than padded channel count")
        self.wf = torch.empty(spec.nslots * self.cin_p * self.cout_p, dtype=dtype_t, device=dev)
        self.wb = torch.empty_like(self.wf) if need_dgrad else None
        self.tile = tile
        self._desc = {}
        self.defer_finalize = False
        self.fin_job = None
        self.persist_wgs = K.PERSIST_WGS  # workgroups of this layer's persistent launches (the engines set their network's cap)
        self.persist_rw = 0               # ... of its register-weights conv launches, when different (0: the same)
        self.persist_dgrad = 0            # ... of its register-weights INPUT-GRADIENT launches only (GeneratorEngine.set_trunk_cap)
        self.persist_fwd = 0              # ... of its FORWARD register-weights launches, when different again (0: the same)
        self.rw_off = False               # route this conv's launches past the register-weights kernel (A/B knob of the D halves)
        self.rw_extra = ""                # more launch classes of kernels.rw_eligible for this conv (set per D half)

    def repack(self):
        s = self.spec
        rows, Kd, s_row, s_k = s.fwd_pack()
        K.pack_weights(self.dt, self.w, rows, Kd, s_row, s_k, s.nslots, self.slots, out=self.wf)
        if self.wb is not None:
            rows, Kd, s_row, s_k = s.dgrad_pack()
            K.pack_weights(self.dt, self.w, rows, Kd, s_row, s_k, s.nslots, self.slots, out=self.wb)

    def fwd(self, x, out, act=L.ACT_NONE, res=None, stats=None, groups=1, nchw=None, stats_r=1, relu_bits=None):
        """x [N,H,W,cin_p] -> out [N,OH,OW,cout_p]; nchw=(buffer, elem_offset, n_stride, c_real) for the fp32 NCHW store.
        stats: `stats_r` replica blocks of [groups][2][cout_p] (the consumer, BatchNorm.apply, folds them).
        relu_bits: also the 1-bit mask of the ReLU output - written only by the class-waves conv-transpose launch; returns True then."""
        N, H, W, _ = x.shape
        OH, OW = self.spec.out_hw(H, W)
        if self.spec.kind == "ct" and self.cout_p % 64 == 0 and self.cin_p in (64, 128) and res is None and stats is None and \
                nchw is None and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU) and self.tile == L.TILE_AUTO and \
                self.dt in (torch.bfloat16, torch.float16) and self.tu.ct_cw:
            # persistent workgroups, one sub-pixel class per wave, weights in registers (csrc/convt_cw.hip, round 5)
            self.last_desc, self.last_rw_nch = "ctcw", self.cin_p // 32
            bits = relu_bits if act == L.ACT_RELU else None
            K.convt_fwd_cw(x, self.wf, self.bias, out, act, max_workgroups=self.persist_fwd or self.persist_rw or self.persist_wgs,
                           relu_bits=bits)
            return bits is not None
        if self.spec.kind == "ct" and self.cout_p % 64 == 0 and res is None and stats is None and nchw is None and \
                act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU) and self.tile == L.TILE_AUTO and self.tu.subpix_ct and \
                N * ((H + 7) // 8) * ((W + 15) // 16) * (self.cout_p // 64) >= 128:
            # one sub-pixel launch for all four classes (csrc/convt_mfma.hip).  Measured (tools/mb_convt.py): 128->128 on
            # 4x64x64 14.4 vs 24.5 us; with only 32 workgroups (64->64 on 4x32x32) the four-class launch wins, 6.9 vs 9.8 us
            self.last_desc = None
            K.convt_fwd(x, self.wf, self.bias, out, act)
            return
        if self.spec.kind == "c4s2" and res is None and nchw is None and act == L.ACT_NONE and self.tile == L.TILE_AUTO and \
                self.tu.s2_cw and K.s2_cw_ok(self.dt, self.cin_p, self.cout_p, H, W) and \
                (stats is None or K.stats_replicas_for(N * OH * OW) == 1):
            # persistent workgroups, the weights in registers, four-way split of the reduction (csrc/conv_s2_cw.hip, round 6)
            self.last_desc, self.last_rw_nch = "s2cw", self.cin_p // 32
            K.conv4s2_fwd_cw(x, self.wf, self.bias, out, stats, groups, stats_replicas=stats_r,
                             max_workgroups=self.persist_fwd or self.persist_rw or self.persist_wgs)
            return
        if self.spec.kind == "c4s2" and self.cout_p % 64 == 0 and res is None and nchw is None and act == L.ACT_NONE and \
                self.tile == L.TILE_AUTO and self.tu.fast_c4s2 and H % 2 == 0 and W % 2 == 0 and \
                (stats is None or K.stats_replicas_for(N * OH * OW) == 1):
            self.last_desc = "c4s2"  # compile-time-tap stride-2 kernel (csrc/conv4s2_mfma.hip)
            # (capped at the network's workgroups since round 5: no faster alone, better beside the other lane - profiles/r05_zz_s2_cap_ab.log)
            K.conv4s2_fwd(x, self.wf, self.bias, out, stats, groups, stats_replicas=stats_r,
                          max_workgroups=(self.persist_wgs if self.dt in (torch.bfloat16, torch.float16) else 0))
            return
        if self.spec.kind == "c3" and nchw is not None and self.cin_p == 64 and self.spec.cout <= 4 and res is None and \
                stats is None and act in (L.ACT_NONE, L.ACT_SIGMOID) and self.tile == L.TILE_AUTO and self.tu.rgb_out and \
                self.dt in (torch.bfloat16, torch.float16):
            self.last_desc = "rgb"  # one 16-row MFMA tile + fp32 NCHW store (csrc/conv_rgb.hip)
            K.conv3x3_rgb(x, self.wf, self.bias, nchw[0], nchw[1], nchw[2], nchw[3], act)
            return
        if self.spec.kind == "c3" and nchw is None and self.tile == L.TILE_AUTO and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU) \
                and self.cin_p == 32 and self.cout_p % 64 == 0 and stats is None and self.tu.c3_cw and not self.rw_off and self.tu.rw != "0" \
                and self.dt in (torch.bfloat16, torch.float16) and 0 < self.tu.rw_fwd_min <= N * H * W:
            # the discriminator's first layer (27 -> 64 at HR size): the eight-equal-waves kernel's 32-channel form, a capped persistent
            # launch instead of 1536 workgroups beside the other lane (csrc/conv3_cw.hip)
            self.last_desc, self.last_rw_nch = "c3cw", 1
            K.conv3x3_rw(x, self.wf, out, False, bias=self.bias, res=res, act=act,
                         max_workgroups=self.persist_fwd or self.persist_rw or self.persist_wgs, cw=True)
            return
        if self.spec.kind == "c3" and nchw is None and self.tile == L.TILE_AUTO and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU) \
                and not self.rw_off and K.rw_eligible(self.dt, self.cin_p, self.cout_p, N, H, W, extra=self.rw_extra, tu=self.tu):
            # persistent register-weights kernel (csrc/conv3_rw.hip; 64 input channels without statistics: csrc/conv3_cw.hip)
            self.last_desc, self.last_rw_nch = ("c3cw" if (self.tu.c3_cw and self.cin_p == 64 and stats is None) else "rw"), self.cin_p // 32
            K.conv3x3_rw(x, self.wf, out, False, bias=self.bias, res=res, act=act, stats=stats, stats_mode=2, groups=groups,
                         stats_replicas=stats_r, max_workgroups=self.persist_fwd or self.persist_rw or self.persist_wgs, cw=self.tu.c3_cw)
            return
        key = ("f", N, H, W, act, res is not None, stats is not None, groups, nchw is not None and nchw[2:], stats_r)
        ent = self._desc.get(key)
        if ent is None:
            # very large launches: private replica scratch folded into the caller's block 0; else the caller's own blocks
            R = K.stats_replicas_for(N * OH * OW) if stats is not None else 1
            if R <= stats_r:
                R = 1
            d = K.make_conv_desc(self.spec.fwd_geom(), self.tg, N, H, W, self.cin_p, OH, OW, self.cout_p, act=act,
                                 stats_mode=2 if stats is not None else 0, stats_groups=groups,
                                 out_mode=L.OUT_NCHW_F32 if nchw else L.OUT_NHWC, c_real=nchw[3] if nchw else 0,
                                 out_n_stride=nchw[2] if nchw else 0, tile_cfg=self.tile,
                                 stats_replicas=R if R > 1 else stats_r)
            scratch = torch.zeros(R * groups * 2 * self.cout_p, device=x.device) if R > 1 else None
            ent = (d, R, scratch)
            self._desc[key] = ent
        d, R, scratch = ent
        self.last_desc = d
        if R > 1:  # statistics through replicas (atomic contention), folded into the caller's buffer afterwards
            scratch.zero_()
            K.conv(d, x, self.wf, out, bias=self.bias, res=res, stats=scratch)
            K.reduce_replicas(scratch, R, groups * 2 * self.cout_p, groups * 2 * self.cout_p, stats, accumulate=True)
            return
        if nchw:
            import ctypes
            buf, off = nchw[0], nchw[1]
            L.check(L.load().tg_conv(ctypes.byref(d), x.data_ptr(), self.wf.data_ptr(), self.bias.data_ptr(), None, None,
                                     buf.data_ptr() + off * 4, None, torch.cuda.current_stream().cuda_stream), "tg_conv")
        else:
            K.conv(d, x, self.wf, out, bias=self.bias, res=res, stats=stats)

    def dgrad(self, dout, out, mask=None, mask_mode=L.MASK_NONE, res=None, bias_grad_of=None, bn_sums=None, mask_bits=None):
        """dout [N,OH,OW,cout_p] -> out [N,H,W,cin_p] = (dgrad + res) * act'(mask); bias_grad_of: Conv whose bias
        gradient is the per-channel sum of `out` (accumulated straight into its grad slot).
        bn_sums = (red, z, groups, replicas): `out` is the output gradient of a BatchNorm without activation whose
        pre-normalisation tensor is z - the epilogue also adds (sum out, sum out * z) per channel into red (the layout of
        tg_bn_bwd_reduce, raw second sum: BatchNorm.backward(reduced=True) converts), which saves that reduction launch."""
        N, OH, OW, _ = dout.shape
        _, H, W, _ = out.shape
        if bn_sums is not None:
            if self.spec.kind != "c3" or mask is not None or bias_grad_of is not None:
                raise L.TecoganHipError("bn_sums: plain 3x3 input-gradient without mask / bias sums only")
            red, z, groups, R = bn_sums
            key = ("dbn", N, OH, OW, res is not None, groups, R)
            d = self._desc.get(key)
            if d is None:
                d = self._desc[key] = K.make_conv_desc(self.spec.dgrad_geom(), self.tg, N, OH, OW, self.cout_p, H, W, self.cin_p,
                                                       mask_mode=L.MASK_BNZ, stats_mode=3, stats_groups=groups, stats_replicas=R)
            self.last_desc = d
            K.conv(d, dout, self.wb, out, res=res, mask=z, stats=red)
            return
        if self.spec.kind == "ct" and mask is None and res is None and bias_grad_of is None and self.tu.s2_cw and \
                K.s2_cw_ok(self.dt, self.cout_p, self.cin_p, OH, OW) and OH == 2 * H and OW == 2 * W:
            self.last_desc, self.last_rw_nch = "s2cw", self.cout_p // 32   # register-weights stride-2 gather (csrc/conv_s2_cw.hip, KS = 3)
            K.convt_dgrad_cw(dout, self.wb, out, max_workgroups=self.persist_dgrad or self.persist_rw or self.persist_wgs)
            return
        if self.spec.kind == "ct" and self.cin_p % 64 == 0 and mask is None and res is None and bias_grad_of is None and \
                self.tu.fast_c4s2 and OH == 2 * H and OW == 2 * W:
            self.last_desc = "ctd"  # 3x3-window stride-2 gather (csrc/conv4s2_mfma.hip, KS = 3)
            K.convt_dgrad(dout, self.wb, out)
            return
        if self.spec.kind == "c4s2" and self.cin_p % 64 == 0 and self.cout_p in (64, 128) and res is None and bias_grad_of is None and \
                self.tu.c4d_cw and self.dt in (torch.bfloat16, torch.float16) and H == 2 * OH and W == 2 * OW and \
                mask_mode in (L.MASK_NONE, L.MASK_RELU, L.MASK_LRELU) and \
                N * ((OH + 3) // 4) * ((OW + 15) // 16) * (self.cin_p // 64) >= 48:
            # persistent workgroups, one sub-pixel class per wave (csrc/conv4s2d_cw.hip, round 5)
            self.last_desc, self.last_rw_nch = "c4dcw", self.cout_p // 32
            K.conv4s2_dgrad_cw(dout, self.wb, out, mask, mask_mode if mask is not None else L.MASK_NONE,
                               max_workgroups=self.persist_dgrad or self.persist_rw or self.persist_wgs)
            return
        if self.spec.kind == "c4s2" and self.cin_p % 64 == 0 and res is None and bias_grad_of is None and \
                self.tu.fast_c4s2 and H == 2 * OH and W == 2 * OW and N * ((OH + 7) // 8) * ((OW + 15) // 16) * (self.cin_p // 64) >= 64:
            # four-class sub-pixel launch (csrc/convt_mfma.hip, PAT 1).  Measured (tools/mb_c4s2_dgrad.py, 24 samples):
            # 64->128 @64x64 15.3 vs 29.0 us, 128->128 @32x32 14.6 vs 19.1, masked 64->64 @128x128 38.1 vs 45.3; launches of
            # fewer than 64 workgroups (9.8 vs 7.0) stay on tg_conv
            self.last_desc = "c4d"
            K.conv4s2_dgrad(dout, self.wb, out, mask, mask_mode if mask is not None else L.MASK_NONE)
            return
        st = bias_grad_of.gbias if bias_grad_of is not None else None
        if self.spec.kind == "c3" and self.tile == L.TILE_AUTO and \
                not self.rw_off and K.rw_eligible(self.dt, self.cout_p, self.cin_p, N, H, W, masked=mask is not None,
                                                  extra=self.rw_extra, dgrad=True, tu=self.tu):
            # the input-gradient of a 3x3 conv is the same conv with mirrored taps
            # mask_bits: the 1-bit form of the ReLU mask (written by the forward launch below this layer) - a sixteenth of the mask rows'
            # bytes; conv3_rw.hip's plain masked input-gradient only
            bits = mask_bits is not None and mask_mode == L.MASK_RELU and res is None
            self.last_desc, self.last_rw_nch = ("c3cw" if (self.tu.c3_cw and self.cout_p == 64 and st is None and not bits) else "rw"), self.cout_p // 32
            K.conv3x3_rw(dout, self.wb, out, True, res=res, mask=mask_bits if bits else mask,
                         mask_mode=L.MASK_RELU_BITS if bits else mask_mode, stats=st, stats_mode=1,
                         max_workgroups=self.persist_dgrad or self.persist_rw or self.persist_wgs, cw=self.tu.c3_cw)
            return
        key = ("d", N, OH, OW, mask_mode, res is not None, st is not None)
        ent = self._desc.get(key)
        if ent is None:
            R = K.stats_replicas_for(N * H * W) if st is not None else 1
            d = K.make_conv_desc(self.spec.dgrad_geom(), self.tg, N, OH, OW, self.cout_p, H, W, self.cin_p,
                                 mask_mode=mask_mode, stats_mode=1 if st is not None else 0, stats_groups=1,
                                 stats_replicas=R)
            scratch = torch.zeros(R * 2 * self.cin_p, device=dout.device) if R > 1 else None
            ent = (d, R, scratch)
            self._desc[key] = ent
        d, R, scratch = ent
        self.last_desc = d
        if R > 1:  # bias gradient through replicas (see tg_conv_desc.stats_replicas)
            scratch.zero_()
            K.conv(d, dout, self.wb, out, res=res, mask=mask, stats=scratch)
            K.reduce_replicas(scratch, R, 2 * self.cin_p, self.cin_p, st, accumulate=True)  # slot [0] = per-channel sums
            return
        K.conv(d, dout, self.wb, out, res=res, mask=mask, stats=st)

    def wgrad(self, x_in, dout, bias_sum=False):
        """accumulates dW into the flat gradient buffer (which the step zeroes first): one launch writes per-split fp32
        slabs, the fold adds them into the PyTorch-layout gradient (per conv, or once per network: Finalizer).
        bias_sum: `dout` is the gradient w.r.t. this conv's output, so its per-channel sum (the bias gradient) is taken
        from the Y tiles that pass through the kernel anyway (plain convs only: Y must be `dout`)."""
        x_is_in, S, taps, ca, cb, s_a, s_b = self.spec.wgrad_info()
        if bias_sum and (not x_is_in or self.gbias is None):
            raise L.TecoganHipError("bias_sum needs a conv with a bias whose wgrad Y operand is the output gradient")
        X, Y = (x_in, dout) if x_is_in else (dout, x_in)
        N, XH, XW, cx = X.shape
        _, YH, YW, cy = Y.shape
        key = ("w", N, XH, XW, YH, YW, bias_sum, self.persist_wgs)
        ent = self._desc.get(key)
        if ent is None:
            nsplit, tpw = K.wgrad_plan(N, YH, YW, S, len(taps), cx, cy, cap=self.persist_wgs)
            if self.ws.frozen:
                raise L.TecoganHipError("new wgrad shape after graph capture")
            stride = len(taps) * cx * cy + (cy if bias_sum else 0)
            slab = torch.empty(nsplit * stride, dtype=torch.float32, device=X.device)
            ent = (K.make_wgrad_desc(self.tg, N, XH, XW, cx, YH, YW, cy, S, taps, nsplit, tpw, y_sum=bias_sum), nsplit,
                   slab, stride)
            self._desc[key] = ent
        d, nsplit, slab, stride = ent
        gb = self.gbias if bias_sum else None
        self.fin_job = [slab.data_ptr(), self.gw.data_ptr(), s_a, s_b, nsplit, len(taps), cx, cy, ca, cb,
                        gb.data_ptr() if bias_sum else 0, stride]
        K.wgrad(d, X, Y, slab)
        if not self.defer_finalize:  # otherwise the network folds every slab in one launch after its backward pass
            K.wgrad_finalize(slab, nsplit, len(taps), cx, cy, ca, cb, self.gw, s_a, s_b, self.slots, True, gb)


class WgradGroup:
    """Weight gradients of several same-shaped layers in ONE launch (tg_wgrad_multi).  The layers' backward operands all
    exist once the dgrad chain has passed them (every gradient tensor has its own buffer), so the launch is issued after
    the chain: the grid is layers x splits, i.e. the chip is filled by cross-layer parallelism instead of a deep split of
    the pixel dimension, and the fp32 slab volume (written here, read back by the fold) shrinks by the number of layers.
    Needs the deferred fold (Finalizer)."""

    def __init__(self):
        self.items, self.cache = [], {}

    def add(self, conv, x_in, dout, bias_sum=False):
        self.items.append((conv, x_in, dout, bias_sum))

    def launch(self):
        items, self.items = self.items, []
        if not items:
            return
        key = (items[0][0].persist_wgs,) + tuple((id(c), x.data_ptr(), y.data_ptr(), tuple(x.shape), tuple(y.shape), b)
                                                 for c, x, y, b in items)
        ent = self.cache.get(key)
        if ent is None:
            c0 = items[0][0]
            x_is_in, S, taps, ca, cb, s_a, s_b = c0.spec.wgrad_info()
            rows = []
            any_bias = any(b for _, _, _, b in items)
            for c, x_in, dout, b in items:
                if c.spec.wgrad_info()[:3] != (x_is_in, S, taps) or (c.cin_p, c.cout_p) != (c0.cin_p, c0.cout_p) or \
                        x_in.shape != items[0][1].shape or dout.shape != items[0][2].shape:
                    raise L.TecoganHipError("WgradGroup needs layers of identical shape")
                if b and (not x_is_in or c.gbias is None):
                    raise L.TecoganHipError("bias_sum needs a conv with a bias whose wgrad Y operand is the output gradient")
                if c.ws.frozen:
                    raise L.TecoganHipError("new wgrad shape after graph capture")
            X0, Y0 = (items[0][1], items[0][2]) if x_is_in else (items[0][2], items[0][1])
            N, XH, XW, cx = X0.shape
            _, YH, YW, cy = Y0.shape
            blocks = K.wgrad_blocks(len(taps), cx, cy)
            nsplit = max(1, min(K.wgrad_tiles(N, YH, YW, S), c0.persist_wgs // (len(items) * blocks)))
            fin = []
            stride = len(taps) * cx * cy + (cy if any_bias else 0)
            for c, x_in, dout, b in items:
                X, Y = (x_in, dout) if x_is_in else (dout, x_in)
                slab = torch.empty(nsplit * stride, dtype=torch.float32, device=X.device)
                _, _, _, ca, cb, s_a, s_b = c.spec.wgrad_info()
                rows.append([X.data_ptr(), Y.data_ptr(), slab.data_ptr()])
                fin.append((c, slab, [slab.data_ptr(), c.gw.data_ptr(), s_a, s_b, nsplit, len(taps), cx, cy, ca, cb,
                                      c.gbias.data_ptr() if b else 0, stride]))
            desc = K.make_wgrad_desc(c0.tg, N, XH, XW, cx, YH, YW, cy, S, taps, nsplit, 0, y_sum=any_bias)
            ent = (desc, torch.tensor(rows, dtype=torch.int64, device=X0.device), fin)
            self.cache[key] = ent
        desc, jobs, fin = ent
        for c, _, job in fin:
            c.fin_job = job
        K.wgrad_multi(desc, jobs, len(fin))


class WgradList:
    """Weight gradients of several layers - ANY mix of image sizes and channel counts - in ONE persistent launch per layer
    kind (tg_wgrad_group_v, csrc/wgrad_group.hip: 3x3 stride-1 convs, conv-transposes k3 s2, convs k4 s2): the (layer, 64 x 64
    channel block, pixel tile) units of all layers form one list that `cap` workgroups share evenly, so a step writes
    workgroups + channel blocks  slabs per launch instead of  workgroups  slabs per LAYER.  Same calling pattern as
    WgradGroup (add ... launch, deferred fold); layers the kernel does not take (fp32, 32 -> 32 channels) run their own
    tg_wgrad launch at add()."""
    VARIANT = {"c3": L.WGROUP_C3, "ct": L.WGROUP_CT, "c4s2": L.WGROUP_C4S2}
    WIDE = {L.WGROUP_C3: L.WGROUP_C3_B128, L.WGROUP_CT: L.WGROUP_CT_B128}   # the same kinds with 64 x 128 channel blocks
    # 64 x 128 blocks for layers whose Y operand has a multiple of 128 channels and at least this many pixels; 0 = never (default).
    # Built, parity-tested and measured SLOWER: 144 accumulator registers + the kernel's ~120 other live registers do not fit the 256
    # of a two-waves-per-SIMD workgroup (139-157 VGPRs spilled for the 3x3 kind, 67 for the conv-transpose kind): c32 + c30 of the
    # generator 89 -> 230 us at 160 workgroups, ct4 79 -> 87 us, the discriminator's stage 2 62 -> 126 us
    # (profiles/r03_m_wgrad_b128.log; the 64 x 64 lists run these layers at 700-1050 TFLOP/s alone).

    @staticmethod
    def wide_min_pixels():
        n = TU().wgrad_b128_pixels
        if n > 0:
            tuning.need_experiments("wgrad_b128_pixels")
        return n

    def __init__(self, cap):
        self.items, self.cache, self.cap = [], {}, cap

    @classmethod
    def variant_of(cls, conv, x_in, dout):
        """the work list (L.WGROUP_*) a layer's weight gradient runs in"""
        v = cls.VARIANT[conv.spec.kind]
        if v in cls.WIDE and cls.wide_min_pixels() > 0:
            y = dout if conv.spec.wgrad_info()[0] else x_in
            if y.shape[3] % 128 == 0 and y.shape[0] * y.shape[1] * y.shape[2] >= cls.wide_min_pixels():
                return cls.WIDE[v]
        return v

    @staticmethod
    def takes(conv):
        return conv.spec.kind in WgradList.VARIANT and conv.dt in (torch.bfloat16, torch.float16) and conv.defer_finalize and \
            (conv.cin_p >= 64 or conv.cout_p >= 64)   # (32 -> 32 layers would run quarter-full blocks: tg_wgrad's 32 x 32 config)

    def add(self, conv, x_in, dout, bias_sum=False):
        if not self.takes(conv):
            conv.wgrad(x_in, dout, bias_sum=bias_sum)
            return
        self.items.append((conv, x_in, dout, bias_sum))

    @staticmethod
    def block_b(variant):
        """Y channels per channel block of a work list"""
        return 128 if variant in (L.WGROUP_C3_B128, L.WGROUP_CT_B128) else 64

    @staticmethod
    def plan(shapes, cap, slot, variant=L.WGROUP_C3):
        """pure host logic (unit-tested on the CPU).  shapes: [(N, H, W, cx_p, cy_p)] with H x W the grid of the Y operand ->
        (tile_w, job rows without the two pointers, units_total, the `workgroups` argument of tg_wgrad_group_v (it launches
        ceil(units / ceil(units / workgroups)) of them), [(job, a0, b0, first_slot, count)] per channel block, slots)"""
        if variant in (L.WGROUP_C3, L.WGROUP_C3_B128):
            tw = 32 if max(s[2] for s in shapes) > 16 else 16
            th = 128 // tw
        else:
            tw, th = 16, 4
        cbw = WgradList.block_b(variant)
        rows, units, gb = [], 0, 0
        spans = []
        for j, (N, H, W, cx, cy) in enumerate(shapes):
            tx, ty = (W + tw - 1) // tw, (H + th - 1) // th
            tiles = N * tx * ty
            ab, bb = (cx + 63) // 64, (cy + cbw - 1) // cbw   # a channel remainder is a part-empty block
            blocks = ab * bb
            rows.append([units, N, H, W, cx, cy, tx, ty, 0, gb])
            for blk in range(blocks):
                spans.append((j, (blk // bb) * 64, (blk % bb) * cbw, units + blk * tiles, units + (blk + 1) * tiles))
            units += blocks * tiles
            gb += blocks
        cap = max(1, min(cap, units))
        per = (units + cap - 1) // cap     # what tg_wgrad_group derives from (units, cap) - pass `cap`, not the launch's grid
        nwg = (units + per - 1) // per
        fold = []
        for g, (j, a0, b0, beg, end) in enumerate(spans):
            w0, w1 = beg // per, (end - 1) // per
            fold.append((j, a0, b0, w0 + g, w1 - w0 + 1))
        return tw, rows, units, cap, fold, nwg + gb

    def launch(self, only=None):
        """launches the queued layers, one work list per layer kind; only: the kinds (L.WGROUP_*) to launch now - the
        others stay queued (the discriminator queues its five k4 s2 layers across the whole backward pass)"""
        now = [it for it in self.items if only is None or self.VARIANT[it[0].spec.kind] in only]
        self.items = [it for it in self.items if not (only is None or self.VARIANT[it[0].spec.kind] in only)]
        var = [self.variant_of(c, x, y) for c, x, y, _ in now]
        for variant in sorted(set(var)):
            self._launch([it for it, v in zip(now, var) if v == variant], variant)

    def _launch(self, items, variant):
        key = (self.cap, variant) + tuple((id(c), x.data_ptr(), y.data_ptr(), tuple(x.shape), tuple(y.shape), b)
                                          for c, x, y, b in items)
        ent = self.cache.get(key)
        if ent is None:
            if items[0][0].ws.frozen:
                raise L.TecoganHipError("new wgrad shape after graph capture")
            lib = L.load()
            slot = int(lib.tg_wgrad_group_slot_floats_v(variant))
            ops = []
            for c, x_in, dout, b in items:   # X: the operand the taps shift (on the S-times finer grid), Y: the other one
                x_is_in = c.spec.wgrad_info()[0]
                ops.append((x_in, dout) if x_is_in else (dout, x_in))
            S = 1 if variant in (L.WGROUP_C3, L.WGROUP_C3_B128) else 2
            for X, Y in ops:
                if (X.shape[0], X.shape[1], X.shape[2]) != (Y.shape[0], S * Y.shape[1], S * Y.shape[2]):
                    raise L.TecoganHipError("tg_wgrad_group: operand grids do not match the layer kind")
            shapes = [(Y.shape[0], Y.shape[1], Y.shape[2], X.shape[3], Y.shape[3]) for X, Y in ops]
            tw, rows, units, wgs, fold, slots = self.plan(shapes, self.cap, slot, variant)
            dev = ops[0][0].device
            slab = torch.empty(slots * slot, dtype=torch.float32, device=dev)
            jobs = []
            for (c, _, _, b), (X, Y), r in zip(items, ops, rows):
                r[8] = 1 if b else 0
                jobs.append([X.data_ptr(), Y.data_ptr()] + r)
            fin, cbw = {}, self.block_b(variant)
            for j, a0, b0, first, count in fold:
                c, _, _, b = items[j]
                _, _, taps, ca, cb, s_a, s_b = c.spec.wgrad_info()
                bias = c.gbias.data_ptr() + 4 * b0 if (b and a0 == 0) else 0
                fin.setdefault(j, []).append([slab.data_ptr() + 4 * slot * first, c.gw.data_ptr() + 4 * (a0 * s_a + b0 * s_b),
                                              s_a, s_b, count, len(taps), 64, cbw, min(64, ca - a0), min(cbw, cb - b0), bias, slot])
            ent = (tw, torch.tensor(jobs, dtype=torch.int64, device=dev), units, wgs, slab, fin, K.tg_dtype(items[0][0].dt))
            self.cache[key] = ent
        tw, jobs, units, wgs, slab, fin, tg = ent
        for j, (c, _, _, _) in enumerate(items):
            c.fin_job = fin[j]   # a LIST of fold jobs (one per channel block): Finalizer.run flattens
        L.check(L.load().tg_wgrad_group_v(tg, variant, tw, jobs.data_ptr(), jobs.shape[0], units, wgs, slab.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "tg_wgrad_group_v")


def _wgrad_lists():
    """TECOGAN_WGRAD_LIST=0: the per-layer / same-shape launches (tg_wgrad, tg_wgrad_multi) instead of tg_wgrad_group"""
    return TU().wgrad_list


# (workgroups per (conv, packing) job of the repack launch, TECOGAN_PACK_BLOCKS: 16 / 48 / 96: D update 52 / 42 / 38 us alone, step
# 4.186 / 4.177 / 4.182 ms)


class Repacker:
    """fp32 master weights -> packed compute copies (forward + dgrad) of every conv of a network in one launch."""

    def __init__(self, convs, dtype_t, device):
        jobs = []
        for c in convs:
            for packed, (rows, Kd, s_row, s_k) in ((c.wf, c.spec.fwd_pack()), (c.wb, c.spec.dgrad_pack())):
                if packed is None:
                    continue
                jobs.append([c.w.data_ptr(), packed.data_ptr(), s_row, s_k, rows, Kd, pad32(rows), pad32(Kd),
                             c.spec.nslots])
        self.jobs = torch.tensor(jobs, dtype=torch.int64, device=device)
        self.n, self.tg = len(jobs), K.tg_dtype(dtype_t)
        self.blocks = TU().pack_blocks

    def run(self):
        L.check(L.load().tg_pack_conv_weights_multi(self.tg, self.jobs.data_ptr(), self.n, self.blocks,
                                                    torch.cuda.current_stream().cuda_stream), "tg_pack_conv_weights_multi")


def _defer_finalize():
    """one fold launch per network (every slab survives until the end of the backward pass) instead of one per conv"""
    return TU().defer_finalize


# Two residual blocks per launch on the recurrent pass (csrc/exp/resblock2.hip: halo recomputed, no cross-workgroup traffic, results
# bit-identical): built, tested and measured SLOWER - 21 MFMA pixel tiles instead of 2 x 6 put 1.75x the matrix and LDS work on
# the workgroup's serial path, which costs more than the launch boundary and patch round trip it saves: a generator pass 0.200 ->
# 0.219 ms alone, the chain 1.56 -> 1.71 ms, the step 4.20 -> 4.34 ms (profiles/r03_t_resblock2_ab.log).  Off by default.
# (TECOGAN_RB_PAIR: experiments build only)
# Batch-norm backward sums in the epilogue of the input-gradient launch that produces dy (9 of the 17 BN layers per pass):
# built, parity-tested (tests/test_kernels_gpu.py::test_bn_backward_sums_in_the_dgrad_epilogue) and measured SLOWER in the step
# - 18 tg_bn_bwd_reduce launches fewer, but the sums need the general conv epilogue (the plain launch takes the slim one) and
# stage 2 leaves the register-weights kernel: D real 1.589 -> 1.574 ms alone, step 4.43 -> 4.46 ms on one box
# (profiles/r03_g_bn_fuse_ab.log).  Off by default.
# (TECOGAN_BN_FUSE: experiments build only)


# Backward pass of a batch norm on a small tensor (the discriminator's 16x16 ... 4x4 layers: 7 of 17 per pass) as one launch
# (tg_bn_bwd_fused) instead of tg_bn_bwd_reduce + tg_bn_bwd_apply: built, parity-tested and measured SLOWER - a workgroup per
# 16-byte channel piece reads 16 B of every 256-byte pixel row (64 cache lines per wave-load; 16 workgroups instead of ~50):
# d_fake_bwd alone 0.936 -> 0.963 ms, D real 1.596 -> 1.608, step 4.405 -> 4.42 ms (profiles/r03_l_bn_bwd_fused_ab.log).  Off.
# (TECOGAN_BN_BWD_FUSED: experiments build only)


def fold_items(jobs):
    """host side of tg_wgrad_fold_items: appends each 12-entry fold job's first item index (pure logic, CPU-tested).
    A job has (ca_p / AB) * (cb_p / BB) tiles x ceil(nsplit / 8) chunks, BB = 64 or 32, AB = 1024 / BB."""
    rows, n = [], 0
    for j in jobs:
        nsplit, ca_p, cb_p = j[4], j[6], j[7]
        bb = 64 if cb_p % 64 == 0 else 32
        rows.append(list(j) + [n])
        n += (ca_p // (1024 // bb)) * (cb_p // bb) * ((nsplit + 7) // 8)
    return rows, n


class Finalizer:
    """folds the weight-gradient slabs of every conv of a network into the flat gradient buffer with ONE launch (instead
    of one ~9 us launch per conv).  The job table is rebuilt only when a conv's launch shape changed."""

    def __init__(self, convs, device):
        self.convs, self.dev, self.tables = convs, device, {}
        self.fold_items = TU().fold_items
        for c in convs:
            c.defer_finalize = True

    def disable(self):
        for c in self.convs:
            c.defer_finalize = False

    def run(self, only=None):
        """folds the slabs of the launches issued since the previous run (a network whose backward pass runs in several
        parts - the discriminator's real and fake halves, the gradient buckets of data-parallel mode - folds after each
        part; every distinct job set keeps its own device table, so nothing is rebuilt under graph capture).
        only: the convs of this part (default: all)."""
        jobs = []
        for c in (self.convs if only is None else only):
            if c.fin_job is not None:  # one job, or one per channel block (WgradList)
                jobs += c.fin_job if isinstance(c.fin_job[0], list) else [c.fin_job]
        if not jobs:
            return
        key = tuple(tuple(j) for j in jobs)
        ent = self.tables.get(key)
        if ent is None:
            rows, n = fold_items(jobs)
            ent = self.tables[key] = (torch.tensor(rows, dtype=torch.int64, device=self.dev), n, max(j[5] for j in jobs))
        table, nitems, max_taps = ent
        if self.fold_items:  # one workgroup per (tile, 8-slab chunk) item of any job
            L.check(L.load().tg_wgrad_fold_items(table.data_ptr(), len(jobs), nitems, max_taps,
                                                 torch.cuda.current_stream().cuda_stream), "tg_wgrad_fold_items")
        else:
            t12 = self.tables.get(("t12", key))
            if t12 is None:
                t12 = self.tables[("t12", key)] = torch.tensor(jobs, dtype=torch.int64, device=self.dev)
            L.check(L.load().tg_wgrad_finalize_multi(t12.data_ptr(), len(jobs), 64,
                                                     torch.cuda.current_stream().cuda_stream), "tg_wgrad_finalize_multi")


class BatchNorm:
    def __init__(self, flat, prefix, C_, bufs, dtype_t, arena):
        self.C, self.Cp = C_, pad32(C_)
        self.gamma, self.beta = flat.padded(flat.p, prefix + ".weight"), flat.padded(flat.p, prefix + ".bias")
        self.dgamma, self.dbeta = flat.padded(flat.g, prefix + ".weight"), flat.padded(flat.g, prefix + ".bias")
        self.rm, self.rv, self.nbt = bufs[prefix + ".running_mean"], bufs[prefix + ".running_var"], bufs[
            prefix + ".num_batches_tracked"]
        # accumulators in R replica blocks: [half][R][2][Cp] when the halves run as separate launches, [R][2 groups][2][Cp]
        # for a whole-batch launch (zeroed once per step with the rest of the arena)
        self.R = TU().stats_replicas
        self.bwd_fused = TU().bn_bwd_fused
        if self.bwd_fused:
            tuning.need_experiments("bn_bwd_fused")
        self.stats = arena.take(2 * self.R * 2 * self.Cp)
        self.red = arena.take(2 * self.R * 2 * self.Cp)
        self.bar = arena.take(32)     # arrival counters of the one-launch backward (tg_bn_bwd_coop): [half or 0] * 8, zeroed with the arena
        self.coop = TU().bn_bwd_coop
        if self.coop:
            tuning.need_experiments("bn_bwd_coop")
        self.save = torch.empty(2, 2, self.Cp, device=flat.device)

    def _slot(self, buf, half):
        n = self.R * 2 * self.Cp
        return buf if half is None else buf[half * n:(half + 1) * n]

    def stats_slot(self, half=None):
        """what the producing conv adds its statistics into (Conv.fwd(stats=..., stats_r=self.R))"""
        return self._slot(self.stats, half)

    def apply(self, z, y, act, groups, skip=None, update=True, half=None):
        """half=None: z holds `groups` equal sample groups.  half=g: z holds only group g of 2 (the reference's g-th D
        call of the step); statistics slot g is used and the running statistics get ONE update."""
        N, H, W, C_ = z.shape
        stats, save = self._slot(self.stats, half), (self.save if half is None else self.save[half])
        g = groups if half is None else 1
        K.bn_apply(z, stats, self.gamma, self.beta, y, save, N, H * W, C_, g, act, skip=skip,
                   running_mean=self.rm if update else None, running_var=self.rv if update else None,
                   nbt=self.nbt if update else None, replicas=self.R)

    def red_slot(self, half=None):
        """what a producing input-gradient launch adds the backward sums into (Conv.dgrad(bn_sums=...))"""
        return self._slot(self.red, half)

    def backward(self, dy, yact, z, dz, act, groups, half=None, reduced=False):
        """half=g: the tensors hold only group g of 2 (see apply); that group's saved statistics / reduction slots are used.
        reduced: the launch that produced dy has already left (sum dy, sum dy * z) in red_slot(half) (act must be none)"""
        N, H, W, C_ = z.shape
        save, red = (self.save if half is None else self.save[half]), self._slot(self.red, half)
        g = groups if half is None else 1
        if not reduced and self.bwd_fused and (N // g) * H * W <= K.bn_bwd_fused_max_pixels():
            K.bn_bwd_fused(dy, yact, z, save, self.gamma, dz, self.dgamma, self.dbeta, N, H * W, C_, g, act)
            return
        if not reduced and self.coop and K.bn_bwd_coop_ok(N, H * W, C_, g, z.dtype):
            bar = self.bar[(half or 0) * 8:(half or 0) * 8 + 1]
            K.bn_bwd_coop(dy, yact, z, save, red, self.gamma, dz, self.dgamma, self.dbeta, N, H * W, C_, g, act, bar, replicas=self.R)
            return
        if not reduced:
            K.bn_bwd_reduce(dy, yact, z, save, red, N, H * W, C_, g, act, replicas=self.R)
        elif act != L.ACT_NONE:
            raise L.TecoganHipError("BatchNorm.backward(reduced=True) needs a BatchNorm without activation")
        K.bn_bwd_apply(dy, yact, z, save, red, self.gamma, dz, self.dgamma, self.dbeta, N, H * W, C_, g, act,
                       replicas=self.R, red_raw=reduced)


class Arena:
    def __init__(self, nfloats, device):
        self.buf = torch.zeros(nfloats, dtype=torch.float32, device=device)
        self.used = 0

    def take(self, n):
        n = _align32(n)
        if self.used + n > self.buf.numel():
            raise L.TecoganHipError("accumulator arena too small")
        v = self.buf[self.used:self.used + n]
        self.used += n
        return v

    def zero(self):
        self.buf.zero_()


# =============================================================================================================
def generator_shapes(num_resblock=16, out_ch=3):
    s = OrderedDict()
    s["conv.0.weight"], s["conv.0.bias"] = (64, 51, 3, 3), (64,)
    for i in range(num_resblock):
        s[f"resids.{i}.0.weight"], s[f"resids.{i}.0.bias"] = (64, 64, 3, 3), (64,)
        s[f"resids.{i}.2.weight"] = (64, 64, 3, 3)
    s["conv_trans.0.weight"], s["conv_trans.0.bias"] = (64, 64, 3, 3), (64,)
    s["conv_trans.2.0.weight"], s["conv_trans.2.0.bias"] = (64, 64, 3, 3), (64,)
    s["conv_trans.2.2.weight"] = (64, 64, 3, 3)
    s["conv_trans.3.0.weight"], s["conv_trans.3.0.bias"] = (128, 64, 3, 3), (128,)
    s["conv_trans.3.2.weight"] = (128, 128, 3, 3)
    s["conv_trans.4.weight"], s["conv_trans.4.bias"] = (128, 128, 3, 3), (128,)
    s["conv_trans.6.weight"], s["conv_trans.6.bias"] = (64, 128, 3, 3), (64,)
    s["output.weight"], s["output.bias"] = (out_ch, 64, 3, 3), (out_ch,)
    return s


class GeneratorEngine:
    """code/models.py:61-86 on HIP kernels.  `slots` activations sets are kept (T*B samples) for the batched backward."""

    def __init__(self, flat, dtype_t, num_resblock=16, out_ch=3):
        self.flat, self.dt, self.nrb, self.out_ch = flat, dtype_t, num_resblock, out_ch
        self.ws = Workspace(flat.device)
        mk = lambda w, b, kind, ci, co, dg=True: Conv(flat, w, b, ConvSpec(kind, ci, co), dtype_t, self.ws, need_dgrad=dg)
        self.conv0 = mk("conv.0.weight", "conv.0.bias", "c3", 51, 64, dg=False)
        self.rb = [(mk(f"resids.{i}.0.weight", f"resids.{i}.0.bias", "c3", 64, 64),
                    mk(f"resids.{i}.2.weight", None, "c3", 64, 64)) for i in range(num_resblock)]
        self.ct0 = mk("conv_trans.0.weight", "conv_trans.0.bias", "ct", 64, 64)
        self.c20 = mk("conv_trans.2.0.weight", "conv_trans.2.0.bias", "c3", 64, 64)
        self.c22 = mk("conv_trans.2.2.weight", None, "c3", 64, 64)
        self.c30 = mk("conv_trans.3.0.weight", "conv_trans.3.0.bias", "c3", 64, 128)
        self.c32 = mk("conv_trans.3.2.weight", None, "c3", 128, 128)
        self.ct4 = mk("conv_trans.4.weight", "conv_trans.4.bias", "ct", 128, 128)
        self.c6 = mk("conv_trans.6.weight", "conv_trans.6.bias", "c3", 128, 64)
        self.cout = mk("output.weight", "output.bias", "c3", 64, out_ch)
        self.convs = [self.conv0] + [c for pair in self.rb for c in pair] + [self.ct0, self.c20, self.c22, self.c30,
                                                                            self.c32, self.ct4, self.c6, self.cout]
        self.act = None
        self.shape = None
        self.sets = ShapeSets()
        for c in self.convs:
            c.persist_wgs, c.persist_fwd = K.persist_wgs("G"), TU().persist_fwd_g
        self.repacker = Repacker(self.convs, dtype_t, flat.device)
        self.finalizer = Finalizer(self.convs, flat.device) if _defer_finalize() else None
        self.trunk_group = WgradGroup() if TU().wgrad_groups else None
        # 16-bit modes: ONE work-list launch for the plain 3x3 layers of the up-sampling stage and one for the trunk
        self.hr_list = None
        if self.finalizer is not None and self.trunk_group is not None and _wgrad_lists() and \
                dtype_t in (torch.bfloat16, torch.float16):
            self.hr_list, self.trunk_group = WgradList(K.persist_wgs("G")), WgradList(K.persist_wgs("G"))
        self._rgb_cache = {}
        self.tu = TU()   # (the engine routes by what it was built with)
        self.fused_rb = dtype_t in (torch.bfloat16, torch.float16) and TU().fused_resblock
        # the fused input-gradient launch (tg_resblock_bwd) is numerically identical and was measured EQUAL in time on the
        # batched backward (15 x 28.2 us vs 30 x 14.9 us at 40 samples: 640 workgroups each re-read 147 KB of weights), so
        # it stays an option
        # L2 prefetch of the next block's weights by the fused launches: round 2 measured a gain (the weights came back from the
        # Infinity Cache every pass); since the 16 blocks' 2.4 MB stay in the XCDs' L2s between passes it only costs issue slots and
        # traffic - chain 1.970 -> 1.862 ms alone, step 4.008 -> 3.898 ms without it (profiles/r04_p_resblock_prefetch_ab.log)
        self.rb_prefetch = TU().rb_prefetch
        # round 5: the wave-specialised, stream-first form of the fused block (csrc/resblock_ws.hip); no prefetch hint there
        self.rb_ws = TU().rb_ws
        self.mask_bits = TU().mask_bits
        if self.mask_bits:
            tuning.need_experiments("mask_bits")
        self.rb_pair_ws = self.rb_ws and TU().rb_pair_ws
        if self.rb_pair_ws:
            tuning.need_experiments("rb_pair_ws")
        self.fused_rb_bwd = self.fused_rb and TU().fused_resblock_bwd
        self.rb_pair = self.fused_rb and TU().rb_pair
        if self.rb_pair:
            tuning.need_experiments("rb_pair")

    def repack(self):
        self.repacker.run()

    def set_trunk_cap(self, cap):
        """workgroups of the trunk's register-weights launches (the 32 input-gradients of the batched backward) when different
        from the generator's (0: the same)"""
        for pair in self.rb:
            for c in pair:
                c.persist_dgrad = cap    # (its own attribute: forward launches are not touched)

    def set_cap(self, cap, fwd=0):
        """workgroups of the generator's persistent launches (register-weights convs, weight-gradient work lists); a scheduling
        knob only - call before the first backward pass of a shape (launch plans are cached per cap).  fwd: the FORWARD
        register-weights launches' own cap (0: the same) - the chain runs beside the real half's 72 workgroups, the backward pass
        beside the fake half's 96"""
        for c in self.convs:
            c.persist_wgs, c.persist_fwd = cap, fwd
        for lst in (self.hr_list, self.trunk_group):
            if isinstance(lst, WgradList):
                lst.cap = cap

    def rgb_bwd_ok(self):
        """the output layer's backward as ONE launch (tg_conv3x3_rgb_bwd, a compact [..., 4] dpre): 16-bit modes with the deferred
        fold (TECOGAN_RGB_BWD=0: the generic input-gradient launch + the layer's place in the weight-gradient work list)"""
        return self.dt in (torch.bfloat16, torch.float16) and self.finalizer is not None and self.out_ch == 3 and \
            TU().rgb_bwd

    def _rgb_bwd(self, x, dpre4, dx):
        c = self.cout
        key = (x.data_ptr(), dpre4.data_ptr(), dx.data_ptr(), tuple(x.shape))
        ent = self._rgb_cache.get(key)
        if ent is None:
            if self.ws.frozen:
                raise L.TecoganHipError("new output-layer backward shape after graph capture")
            # round 3: 160 / 256 / 512 / 1024 = 4.37 4.37 4.39 4.41 ms per step.  Round 6 (lane A is the long pole and the fake half runs
            # beside this launch): 128 / 160 / 192 / 256 = 3.237 / 3.21-3.23 / 3.231 / 3.237 at config 2, 160 / 192 / 256 = 8.17 / 8.30 / 8.15 ms
            # at the configs[3] shard (profiles/r06_u_knobs.log): 160 for the small step, 256 otherwise
            cap = TU().rgb_bwd_wgs_for(x.shape[0] * x.shape[1] * x.shape[2])
            nwg = K.rgb_bwd_workgroups(x.shape[0], x.shape[1], x.shape[2], cap)
            slot = int(L.load().tg_conv3x3_rgb_bwd_slot_floats())
            slab = torch.empty(nwg * slot, dtype=torch.float32, device=x.device)
            _, _, taps, ca, cb, s_a, s_b = c.spec.wgrad_info()
            ent = self._rgb_cache[key] = (slab, cap, [slab.data_ptr(), c.gw.data_ptr(), s_a, s_b, nwg, len(taps), 64, 32, ca, cb, 0, slot])
        slab, cap, c.fin_job = ent
        K.conv3x3_rgb_bwd(dpre4, x, c.w, dx, slab, cap)

    def alloc(self, NS, h, w):
        """selects (creating it on first use) the activation storage for NS samples (NS = T*B when training, B for
        inference); another shape's set stays alive while a step has it pinned (ShapeSets)."""
        if self.shape == (NS, h, w):
            return
        dev, dt = self.flat.device, self.dt
        e = lambda n, hh, ww, c: torch.empty(n, hh, ww, c, dtype=dt, device=dev)

        def make():
            return {"act": {"in0": e(NS, h, w, 64), "a": [e(NS, h, w, 64) for _ in range(self.nrb + 1)],
                            "h": [e(NS, h, w, 64) for _ in range(self.nrb)], "u0": e(NS, 2 * h, 2 * w, 64),
                            "hh": e(NS, 2 * h, 2 * w, 64), "u1": e(NS, 2 * h, 2 * w, 64),
                            "h2": e(NS, 2 * h, 2 * w, 128), "u2": e(NS, 2 * h, 2 * w, 128),
                            "u3": e(NS, 4 * h, 4 * w, 128), "u4": e(NS, 4 * h, 4 * w, 64),
                            # the 1-bit mask of u3 (conv_trans.4's ReLU output), read back by c6's input-gradient instead of u3
                            "u3b": torch.empty(NS, 4 * h, 4 * w, 16, dtype=torch.uint8, device=dev)}, "grad": None}

        self.cur = self.sets.get((NS, h, w), make)
        self.act, self.shape, self.grad = self.cur["act"], (NS, h, w), self.cur["grad"]

    def forward(self, s0, B, out_buf, out_off, out_n_stride, keep_h=True):
        """runs samples [s0, s0+B) of act['in0'] through the net; sigmoid output goes to out_buf (fp32 NCHW).
        keep_h=False (inference): the fused residual blocks do not store their intermediate activation - only the backward pass
        reads it (2 MB of writes per block at 128 x 128)."""
        a = self.act
        sl = slice(s0, s0 + B)
        hbuf = (lambda t: t[sl]) if keep_h else (lambda t: None)
        self.conv0.fwd(a["in0"][sl], a["a"][0][sl], act=L.ACT_RELU)
        # small launches (the recurrent pass: <= 128 tiles of 8 x 8): TWO blocks per launch, halo recomputed (csrc/exp/resblock2.hip)
        pair = self.rb_pair and B * ((a["in0"].shape[1] + 7) // 8) * ((a["in0"].shape[2] + 7) // 8) <= 128
        # round 5: the same for the stream-first kernel - again slower than one block per launch (profiles/r05_b_resblock2_ws_ab.log)
        pair_ws = self.fused_rb and self.rb_pair_ws and not pair and \
            B * ((a["in0"].shape[1] + 7) // 8) * ((a["in0"].shape[2] + 7) // 8) <= 128
        skip_next = False
        for i, (c1, c2) in enumerate(self.rb):
            if skip_next:
                skip_next = False
                continue
            if pair and i + 1 < self.nrb:
                c3, c4 = self.rb[i + 1]
                nxt = (self.rb[i + 2][0].wf, self.rb[i + 2][1].wf, self.rb[i + 3][0].wf, self.rb[i + 3][1].wf) \
                    if i + 3 < self.nrb else None
                K.resblock2_fwd(a["a"][i][sl], c1.wf, c1.bias, c2.wf, c3.wf, c3.bias, c4.wf, a["h"][i][sl], a["a"][i + 1][sl],
                                a["h"][i + 1][sl], a["a"][i + 2][sl], next_w=nxt)
                skip_next = True
                continue
            if pair_ws and i + 1 < self.nrb:   # two blocks in one launch (csrc/exp/resblock2_ws.hip)
                c3, c4 = self.rb[i + 1]
                K.resblock2_fwd_ws(a["a"][i][sl], c1.wf, c1.bias, c2.wf, c3.wf, c3.bias, c4.wf, hbuf(a["h"][i]), a["a"][i + 1][sl],
                                   hbuf(a["h"][i + 1]), a["a"][i + 2][sl])
                skip_next = True
                continue
            if self.fused_rb:  # conv-relu-conv-skip in one launch (csrc/resblock_ws.hip / resblock.hip)
                nxt = (self.rb[i + 1][0].wf, self.rb[i + 1][1].wf) if (self.rb_prefetch and i + 1 < self.nrb) else None
                K.resblock_fwd(a["a"][i][sl], c1.wf, c1.bias, c2.wf, hbuf(a["h"][i]), a["a"][i + 1][sl], next_w=nxt, ws=self.rb_ws)
                continue
            c1.fwd(a["a"][i][sl], a["h"][i][sl], act=L.ACT_RELU)
            c2.fwd(a["h"][i][sl], a["a"][i + 1][sl], res=a["a"][i][sl])
        self.ct0.fwd(a["a"][self.nrb][sl], a["u0"][sl], act=L.ACT_RELU)
        # conv_trans.2 is conv-relu-conv without a skip: the same fused launch - up to TECOGAN_PAIR_RW_MIN pixels; beyond that two
        # launches of the register-weights kernel are faster (config 5: 256 x 256, profiles/r04_x_rw_fwd_routing.log)
        npix2 = B * a["u0"].shape[1] * a["u0"].shape[2]
        if self.fused_rb and not (0 < self.tu.pair_rw_min <= npix2 and
                                  K.rw_eligible(self.dt, 64, 64, B, a["u0"].shape[1], a["u0"].shape[2], tu=self.tu)):
            K.resblock_fwd(a["u0"][sl], self.c20.wf, self.c20.bias, self.c22.wf, hbuf(a["hh"]), a["u1"][sl], skip=False, ws=self.rb_ws)
        else:
            self.c20.fwd(a["u0"][sl], a["hh"][sl], act=L.ACT_RELU)
            self.c22.fwd(a["hh"][sl], a["u1"][sl])
        self.c30.fwd(a["u1"][sl], a["h2"][sl], act=L.ACT_RELU)
        self.c32.fwd(a["h2"][sl], a["u2"][sl])
        wrote = self.ct4.fwd(a["u2"][sl], a["u3"][sl], act=L.ACT_RELU, relu_bits=a["u3b"][sl] if (keep_h and self.mask_bits) else None)
        if s0 == 0:
            self.cur["u3b_ok"] = bool(wrote)   # (per buffer set; every pass of a step takes the same route)
        elif not wrote:
            self.cur["u3b_ok"] = False
        self.c6.fwd(a["u3"][sl], a["u4"][sl], act=L.ACT_RELU)
        self.cout.fwd(a["u4"][sl], None, act=L.ACT_SIGMOID, nchw=(out_buf, out_off, out_n_stride, self.out_ch))

    def _alloc_grad(self, chunk=None):
        """gradient scratch for a backward pass over `chunk` samples (default: all).  Every gradient tensor has its own
        buffer: the grouped weight-gradient launch at the end of the pass reads all of them."""
        if self.grad is not None and chunk is None:
            return
        NS, h, w = self.shape
        NC = chunk or NS
        dev, dt = self.flat.device, self.dt
        e = lambda hh, ww, c: torch.empty(NC, hh, ww, c, dtype=dt, device=dev)
        self.cur["grad"] = self.grad = {"dpre": e(4 * h, 4 * w, 32), "hr64": e(4 * h, 4 * w, 64), "hr128": e(4 * h, 4 * w, 128),
                     "m128a": e(2 * h, 2 * w, 128), "m128b": e(2 * h, 2 * w, 128), "m64a": e(2 * h, 2 * w, 64),
                     "m64b": e(2 * h, 2 * w, 64), "m64c": e(2 * h, 2 * w, 64),
                     "dA": [e(h, w, 64) for _ in range(self.nrb + 1)], "dH": [e(h, w, 64) for _ in range(self.nrb)]}

    def bucket_split(self):
        """element offset in the flat gradient buffer where the up-sampling stage's parameters (conv_trans.*, output.*)
        begin: [split, total) is final after backward(part='hr'), [0, split) after part='trunk' (data-parallel buckets)"""
        return self.flat.offsets["conv_trans.0.weight"][0]

    def backward(self, s0=0, s1=None, dpre=None, part=None):
        """consumes grad['dpre'][:s1-s0] (d loss / d pre-sigmoid of samples [s0,s1)) and ACCUMULATES every weight/bias
        gradient.  The T passes are independent in backward (inputs are detached, code/train.py:90,108), so any
        sample range may be processed as one batch; ranges must not run concurrently (they add into the same grads).
        part: None = the whole pass; 'hr' = the up-sampling stage (output layer ... conv_trans.0) including the fold of its
        weight gradients, 'trunk' = the rest - two launch sequences, so that data-parallel mode can all-reduce the first
        bucket while the second is still being computed."""
        NS = self.shape[0]
        s1 = NS if s1 is None else s1
        n = s1 - s0
        if self.finalizer is not None and (s0, s1) != (0, NS):
            # the deferred fold takes each layer's LAST launch: a partial range would lose the slabs of the ranges before it
            raise L.TecoganHipError("GeneratorEngine.backward: sample ranges need per-layer folds (TECOGAN_DEFER_FINALIZE=0)")
        a = {k: ([t[s0:s1] for t in v] if isinstance(v, list) else v[s0:s1]) for k, v in self.act.items()}
        g = {k: ([t[:n] for t in v] if isinstance(v, list) else v[:n]) for k, v in self.grad.items()}
        if dpre is not None:
            g["dpre"] = dpre
        RELU = L.MASK_RELU
        # the plain 3x3 layers of the up-sampling stage share one work-list launch behind its input-gradient chain
        # (hr_list; every gradient tensor has its own buffer); the other kinds launch where they stand
        hr = self.hr_list
        wh = (lambda c, x, y, b=False: hr.add(c, x, y, b)) if hr is not None else \
            (lambda c, x, y, b=False: c.wgrad(x, y, bias_sum=b))
        dA, dH = g["dA"], g["dH"]
        hr_convs = [self.cout, self.c6, self.ct4, self.c32, self.c30, self.c22, self.c20, self.ct0]
        if part != "trunk":
            self._backward_hr(a, g, wh, hr, RELU)
            if part == "hr":
                if self.finalizer is not None:
                    self.finalizer.run(only=hr_convs)
                return
        self._backward_trunk(a, g, dA, dH, RELU)
        if self.finalizer is not None and s1 == NS:  # last (or only) sample range of the step
            self.finalizer.run(only=None if part is None else [c for c in self.convs if all(c is not x for x in hr_convs)])

    def _backward_hr(self, a, g, wh, hr, RELU):
        dA = g["dA"]
        # (output bias gradient: the content-loss kernel's channel sums, see TecoGANStep)
        if g["dpre"].shape[-1] == 4:    # compact dpre: input gradient + weight gradient of the output layer in one pass
            self._rgb_bwd(a["u4"], g["dpre"], g["hr64"])
        else:
            wh(self.cout, a["u4"], g["dpre"])
            self.cout.dgrad(g["dpre"], g["hr64"], mask=a["u4"], mask_mode=RELU)
        # bias gradients of plain convs come out of their own wgrad launch (bias_sum): the output gradient is that
        # launch's Y operand, so its channel sums cost a few VALU adds there instead of an atomics epilogue here
        wh(self.c6, a["u3"], g["hr64"], True)
        self.c6.dgrad(g["hr64"], g["hr128"], mask=a["u3"], mask_mode=RELU, bias_grad_of=self.ct4,
                      mask_bits=a["u3b"] if self.cur.get("u3b_ok") else None)
        wh(self.ct4, a["u2"], g["hr128"])
        self.ct4.dgrad(g["hr128"], g["m128a"])
        wh(self.c32, a["h2"], g["m128a"])
        self.c32.dgrad(g["m128a"], g["m128b"], mask=a["h2"], mask_mode=RELU)
        wh(self.c30, a["u1"], g["m128b"], True)
        self.c30.dgrad(g["m128b"], g["m64a"])
        wh(self.c22, a["hh"], g["m64a"])
        self.c22.dgrad(g["m64a"], g["m64b"], mask=a["hh"], mask_mode=RELU)
        wh(self.c20, a["u0"], g["m64b"], True)
        self.c20.dgrad(g["m64b"], g["m64c"], mask=a["u0"], mask_mode=RELU, bias_grad_of=self.ct0)
        wh(self.ct0, a["a"][self.nrb], g["m64c"])
        self.ct0.dgrad(g["m64c"], dA[self.nrb])
        if hr is not None:
            hr.launch()

    def _backward_trunk(self, a, g, dA, dH, RELU):
        grouped = self.finalizer is not None and self.trunk_group is not None
        wg = (lambda c, x, y, b=False: self.trunk_group.add(c, x, y, b)) if grouped else \
            (lambda c, x, y, b=False: c.wgrad(x, y, bias_sum=b))
        for i in range(self.nrb - 1, -1, -1):
            c1, c2 = self.rb[i]
            wg(c2, a["h"][i], dA[i + 1])
            if self.fused_rb_bwd and i > 0:  # both input-gradients of the block in one launch (csrc/resblock.hip, BWD)
                nxt = (self.rb[i - 1][1].wb, self.rb[i - 1][0].wb) if i > 1 else None
                K.resblock_bwd(dA[i + 1], c2.wb, a["h"][i], c1.wb, dH[i], dA[i], next_w=nxt)
                wg(c1, a["a"][i], dH[i], True)
                continue
            c2.dgrad(dA[i + 1], dH[i], mask=a["h"][i], mask_mode=RELU)
            wg(c1, a["a"][i], dH[i], True)
            if i > 0:
                c1.dgrad(dH[i], dA[i], res=dA[i + 1])
            else:  # a[0] = relu(conv0(in0)): fold its relu' into the same epilogue
                c1.dgrad(dH[i], dA[i], res=dA[i + 1], mask=a["a"][0], mask_mode=RELU)
        wg(self.conv0, a["in0"], dA[0], True)
        if grouped:
            self.trunk_group.launch()  # 2*nrb+1 layers of one shape (64 -> 64 channels, 3x3, h x w): one grid


# =============================================================================================================
FNET_BLOCKS = (("down1", 3, 32), ("down2", 32, 64), ("down3", 64, 128), ("down4", 128, 256), ("up1", 256, 512),
               ("up2", 512, 256), ("up3", 256, 128), ("up4", 128, 64))


def fnet_shapes():
    s = OrderedDict()
    for name, ci, co in FNET_BLOCKS:
        s[f"{name}.0.weight"], s[f"{name}.0.bias"] = (co, ci, 3, 3), (co,)
        s[f"{name}.2.weight"], s[f"{name}.2.bias"] = (co, co, 3, 3), (co,)
    s["output_block.0.weight"], s["output_block.0.bias"] = (32, 64, 3, 3), (32,)
    s["output_block.2.weight"], s["output_block.2.bias"] = (2, 32, 3, 3), (2,)
    return s


class FNetEngine:
    """code/models.py:9-50: four (conv-lrelu-conv-lrelu-maxpool) encoder blocks, four (conv-lrelu-conv-lrelu-bilinear x2)
    decoder blocks, conv-lrelu-conv, 24*tanh.  The reference never instantiates f_net (main.py:231 is commented out) and
    leaves its optimiser dead (main.py:244-245, code/train.py:343-346); forward() serves inference and the opt-in flow
    option, backward() the opt-in FNet training (parity unpinned): hand-derived, same kernels as G / D."""

    def __init__(self, flat, dtype_t):
        self.flat, self.dt = flat, dtype_t
        self.ws = Workspace(flat.device)
        train = flat.g is not None
        mk = lambda pre, ci, co, dg=True: Conv(flat, pre + ".weight", pre + ".bias", ConvSpec("c3", ci, co), dtype_t, self.ws,
                                               need_dgrad=train and dg)
        self.blocks = [(name, mk(name + ".0", ci, co, name != "down1"), mk(name + ".2", co, co), co) for name, ci, co in FNET_BLOCKS]
        self.o0, self.o2 = mk("output_block.0", 64, 32), mk("output_block.2", 32, 2)
        self.convs = [c for _, a, b, _ in self.blocks for c in (a, b)] + [self.o0, self.o2]
        self.repacker = Repacker(self.convs, dtype_t, flat.device)
        self.shape, self.act, self.grad = None, None, None
        self.sets = ShapeSets()
        self.finalizer = Finalizer(self.convs, flat.device) if (train and _defer_finalize()) else None
        self.wlist = WgradList(K.persist_wgs("D")) if (self.finalizer is not None and _wgrad_lists() and
                                                      dtype_t in (torch.bfloat16, torch.float16)) else None
        for c in self.convs:
            c.persist_wgs = K.persist_wgs("D")   # (runs at the head of lane B, beside the generator chain)

    def repack(self):
        self.repacker.run()

    def alloc(self, N, h, w):
        if h % 16 or w % 16:
            raise ValueError("f_net needs H and W divisible by 16 (four 2x2 max-pools)")
        if self.shape == (N, h, w):
            return
        dev, dt = self.flat.device, self.dt
        e = lambda hh, ww, c: torch.empty(N, hh, ww, pad32(c), dtype=dt, device=dev)

        def make():
            bufs, hh, ww = [], h, w
            for i, (_, _, _, co) in enumerate(self.blocks):
                a, b = e(hh, ww, co), e(hh, ww, co)
                hh, ww = (hh // 2, ww // 2) if i < 4 else (hh * 2, ww * 2)
                bufs.append((a, b, e(hh, ww, co)))
            return {"in": e(h, w, 3), "blocks": bufs, "o0": e(h, w, 32), "grad": None}

        self.act = self.sets.get((N, h, w), make)
        self.shape, self.grad = (N, h, w), self.act["grad"]

    def _alloc_grad(self):
        """one gradient buffer per activation (the grouped weight-gradient launch at the end reads all of them)"""
        if self.grad is None:
            z = torch.empty_like
            self.act["grad"] = self.grad = {"blocks": [(z(a), z(b), z(r)) for a, b, r in self.act["blocks"]],
                                            "o0": z(self.act["o0"]), "o2": z(self.act["in"])}

    def forward(self, out):
        """act['in'] -> out [N,2,h,w] fp32 NCHW"""
        N, h, w = self.shape
        x = self.act["in"]
        for i, (_, c0, c2, _) in enumerate(self.blocks):
            a, b, r = self.act["blocks"][i]
            c0.fwd(x, a, act=L.ACT_LRELU)
            c2.fwd(a, b, act=L.ACT_LRELU)
            (K.maxpool2 if i < 4 else K.up2_bilinear)(b, r)
            x = r
        self.o0.fwd(x, self.act["o0"], act=L.ACT_LRELU)
        self.o2.fwd(self.act["o0"], None, act=L.ACT_TANH24, nchw=(out, 0, 2 * h * w, 2))

    def backward(self, dout, out):
        """dout = d(loss)/d(result), out = the result of forward() (both fp32 [N,2,h,w]); ACCUMULATES every weight / bias
        gradient of the estimator into the flat gradient buffer (which the caller zeroes).  Chain: 24 tanh' -> output block
        -> decoder (bilinear x2 backward, LeakyReLU') -> encoder (max-pool routing, LeakyReLU')."""
        if self.flat.g is None:
            raise L.TecoganHipError("this f_net engine was built without gradient buffers")
        self._alloc_grad()
        a_, g_ = self.act, self.grad
        LR = L.MASK_LRELU
        wl = self.wlist
        wg = (lambda c, x, y: wl.add(c, x, y, True)) if wl is not None else (lambda c, x, y: c.wgrad(x, y, bias_sum=True))
        K.tanh24_bwd(dout, out, g_["o2"])
        wg(self.o2, a_["o0"], g_["o2"])
        self.o2.dgrad(g_["o2"], g_["o0"], mask=a_["o0"], mask_mode=LR)
        wg(self.o0, a_["blocks"][-1][2], g_["o0"])
        self.o0.dgrad(g_["o0"], g_["blocks"][-1][2])
        for i in range(len(self.blocks) - 1, -1, -1):
            _, c0, c2, _ = self.blocks[i]
            a, b, r = a_["blocks"][i]
            ga, gb, gr = g_["blocks"][i]
            if i >= 4:
                K.up2_bilinear_bwd(gr, gb, lrelu_mask=b)
            else:
                K.maxpool2_bwd(b, gr, gb, relu_mask=2)
            x_in = a_["in"] if i == 0 else a_["blocks"][i - 1][2]
            wg(c2, a, gb)
            c2.dgrad(gb, ga, mask=a, mask_mode=LR)
            wg(c0, x_in, ga)
            if i > 0:  # (the pooled / up-sampled tensor below carries no activation of its own)
                c0.dgrad(ga, g_["blocks"][i - 1][2])
        if wl is not None:
            wl.launch()
        if self.finalizer is not None:
            self.finalizer.run()


# =============================================================================================================
VGG_LAYERS = (("Conv1_1", 3, 64), ("Conv1_2", 64, 64), "pool1", ("Conv2_1", 64, 128), ("Conv2_2", 128, 128), "pool2",
              ("Conv3_1", 128, 256), ("Conv3_2", 256, 256), ("Conv3_3", 256, 256), ("Conv3_4", 256, 256), "pool3",
              ("Conv4_1", 256, 512), ("Conv4_2", 512, 512), ("Conv4_3", 512, 512), ("Conv4_4", 512, 512))
VGG_TAPS = ("Conv2_2", "Conv3_4", "Conv4_4")   # vgg_19/conv2_2, conv3_4, conv4_4 (code/train.py:125)
VGG_MEAN = (123.68, 116.78, 103.94)            # code/train.py:6


def vgg_shapes():
    """state_dict keys of the reference's VGG19 module (code/ops.py:146-165) up to Conv4_4 - nothing deeper is ever read
    (code/train.py:125); the 3x3 kernel size the reference leaves out from Conv3_1 on is VGG-19's"""
    s = OrderedDict()
    for l in VGG_LAYERS:
        if isinstance(l, tuple):
            s[f"{l[0]}.0.weight"], s[f"{l[0]}.0.bias"] = (l[2], l[1], 3, 3), (l[2],)
    return s


class VGGEngine:
    """Opt-in VGG feature loss (semantics fixed in DESIGN.md because the reference's cannot execute):
    a FROZEN VGG-19 conv stack up to Conv4_4 on tg_conv, run on the generated and the target frames as one batch; per tap
    layer the per-pixel cosine similarity of the channel-normalised features; input-gradient of the generated half back
    to d(loss)/d(pre-sigmoid) of the generator.  No weight gradients (the network is a fixed feature extractor)."""

    def __init__(self, flat, dtype_t):
        self.flat, self.dt = flat, dtype_t
        self.ws = Workspace(flat.device)
        self.convs = OrderedDict()
        for l in VGG_LAYERS:
            if isinstance(l, tuple):
                self.convs[l[0]] = Conv(flat, f"{l[0]}.0.weight", f"{l[0]}.0.bias", ConvSpec("c3", l[1], l[2]), dtype_t,
                                        self.ws, need_dgrad=True)
        self.repacker = Repacker(list(self.convs.values()), dtype_t, flat.device)
        self.sets = ShapeSets()
        self.shape = None

    def repack(self):
        self.repacker.run()

    def alloc(self, N, H, W):
        """N generated + N target frames of H x W"""
        if H % 8 or W % 8:
            raise ValueError("the VGG loss needs frame sizes divisible by 8 (three 2x2 max-pools)")
        if self.shape == (N, H, W):
            return
        dev, dt = self.flat.device, self.dt

        def make():
            act, grad, hh, ww = OrderedDict(), OrderedDict(), H, W
            act["in"] = torch.empty(2 * N, H, W, 32, dtype=dt, device=dev)
            grad["in"] = torch.empty(N, H, W, 32, dtype=dt, device=dev)
            for l in VGG_LAYERS:
                if isinstance(l, tuple):
                    act[l[0]] = torch.empty(2 * N, hh, ww, l[2], dtype=dt, device=dev)
                    grad[l[0]] = torch.empty(N, hh, ww, l[2], dtype=dt, device=dev)      # d loss / d (conv output, pre-ReLU)
                    c = l[2]
                else:
                    hh, ww = hh // 2, ww // 2
                    act[l] = torch.empty(2 * N, hh, ww, c, dtype=dt, device=dev)
                    grad[l] = torch.empty(N, hh, ww, c, dtype=dt, device=dev)
            for t in VGG_TAPS[:-1]:
                grad["tap_" + t] = torch.empty_like(grad[t])
            return {"act": act, "grad": grad}

        cur = self.sets.get((N, H, W), make)
        self.act, self.grad, self.shape = cur["act"], cur["grad"], (N, H, W)

    def forward(self, gen_nchw, tgt_nchw):
        """gen / tgt: fp32 [N,3,H,W] in [0,1] -> activations of both halves (generated first)"""
        N = self.shape[0]
        a = self.act
        shift = [127.5 - m for m in VGG_MEAN]
        K.vgg_input(gen_nchw, a["in"][:N], 127.5, shift)
        K.vgg_input(tgt_nchw, a["in"][N:], 127.5, shift)
        x = a["in"]
        for l in VGG_LAYERS:
            if isinstance(l, tuple):
                self.convs[l[0]].fwd(x, a[l[0]], act=L.ACT_RELU)
                x = a[l[0]]
            else:
                K.maxpool2(x, a[l])
                x = a[l]

    def loss_backward(self, acc3, coef_scale, gen_nchw, dpre, loss_scale=None, bias_acc=None):
        """acc3[i] += sum of per-pixel cosines of tap layer i; d(loss)/d(pre-sigmoid) of the generated frames is ADDED to
        dpre (and its channel sums to bias_acc[0:3], the output layer's bias gradient), for loss = coef_scale * sum_i (1 - mean
        cosine_i)."""
        N = self.shape[0]
        a, g = self.act, self.grad
        for i, t in enumerate(VGG_TAPS):
            f = a[t]
            npix = N * f.shape[1] * f.shape[2]
            last = t == VGG_TAPS[-1]
            # the deepest tap starts the backward chain: its gradient is masked by its own ReLU here; the shallower taps
            # are added (and masked) where the chain passes them (tg_maxpool2_bwd)
            K.cosine_loss(f[:N], f[N:], g[t] if last else g["tap_" + t], -coef_scale / npix, last, acc3[i:i + 1],
                          loss_scale=loss_scale)
        names = [l if isinstance(l, str) else l[0] for l in VGG_LAYERS]
        for j in range(len(names) - 1, -1, -1):
            name = names[j]
            below = names[j - 1] if j > 0 else "in"
            if name.startswith("pool"):
                # d(pool out) -> d(conv out below, pre-ReLU), + that layer's tap gradient, x relu'
                K.maxpool2_bwd(a[below][:N], g[name], g[below], res=g.get("tap_" + below), relu_mask=True)
                continue
            conv = self.convs[name]
            if below == "in":
                conv.dgrad(g[name], g["in"])
            elif below.startswith("pool"):
                conv.dgrad(g[name], g[below])                       # the ReLU below a pool is handled by tg_maxpool2_bwd
            else:
                conv.dgrad(g[name], g[below], mask=a[below][:N], mask_mode=L.MASK_RELU)
        K.vgg_input_grad(g["in"], gen_nchw, dpre, 127.5, bias_acc=bias_acc)


def discriminator_shapes(resblocks=4, ch=128, fc_in=48):
    s = OrderedDict()
    s["conv.0.weight"], s["conv.0.bias"] = (64, 27, 3, 3), (64,)
    stage_c = {1: 64, 2: ch, 3: ch}
    blk = {1: (64, 64), 2: (ch, 64), 3: (ch, ch), 4: (64, ch), 5: (3, 64)}
    for st in (1, 2, 3):
        co, ci = blk[st]
        s[f"block{st}.0.weight"] = (co, ci, 4, 4)
        s[f"block{st}.1.weight"], s[f"block{st}.1.bias"] = (co,), (co,)
        c = stage_c[st]
        for j in range(resblocks):
            s[f"resids{st}.{j}.0.0.weight"], s[f"resids{st}.{j}.0.0.bias"] = (c, c, 3, 3), (c,)
            s[f"resids{st}.{j}.0.2.weight"] = (c, c, 3, 3)
            s[f"resids{st}.{j}.1.weight"], s[f"resids{st}.{j}.1.bias"] = (c,), (c,)
    for k in (4, 5):
        co, ci = blk[k]
        s[f"block{k}.0.weight"] = (co, ci, 4, 4)
        s[f"block{k}.1.weight"], s[f"block{k}.1.bias"] = (co,), (co,)
    s["fc.weight"], s["fc.bias"] = (1, fc_in), (1,)
    return s


def discriminator_bn_names(resblocks=4):
    names = []
    for st in (1, 2, 3):
        names.append(f"block{st}.1")
        names += [f"resids{st}.{j}.1" for j in range(resblocks)]
    return names + ["block4.1", "block5.1"]


def make_bn_buffers(shapes, resblocks, device):
    """running_mean / running_var padded to 32 (kernels read the padded length); state_dict sees the first C entries."""
    bufs = OrderedDict()
    for bn in discriminator_bn_names(resblocks):
        c = shapes[bn + ".weight"][0]
        bufs[bn + ".running_mean"] = torch.zeros(pad32(c), device=device)
        bufs[bn + ".running_var"] = torch.ones(pad32(c), device=device)
        bufs[bn + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long, device=device)
    return bufs


class DiscriminatorEngine:
    """code/models.py:97-146 on HIP kernels.  The reference calls D twice per step (real, fake) with separate BN batch
    statistics; here both calls run as ONE batch of 2*tb samples whose BN statistics are kept per half (groups=2)."""

    def __init__(self, flat, bufs, dtype_t, resblocks=4, ch=128):
        self.flat, self.dt, self.nrb, self.ch = flat, dtype_t, resblocks, ch
        self.ws = Workspace(flat.device)
        self.arena = Arena(32 * 1024 * TU().stats_replicas, flat.device)
        mk = lambda w, b, kind, ci, co, dg=True: Conv(flat, w, b, ConvSpec(kind, ci, co), dtype_t, self.ws, need_dgrad=dg)
        bn = lambda p, c: BatchNorm(flat, p, c, bufs, dtype_t, self.arena)
        self.conv0 = mk("conv.0.weight", "conv.0.bias", "c3", 27, 64, dg=False)
        cin = {1: 64, 2: 64, 3: ch, 4: ch, 5: 64}
        cout = {1: 64, 2: ch, 3: ch, 4: 64, 5: 3}
        self.blk = {k: (mk(f"block{k}.0.weight", None, "c4s2", cin[k], cout[k]), bn(f"block{k}.1", cout[k]))
                    for k in range(1, 6)}
        self.res = {st: [(mk(f"resids{st}.{j}.0.0.weight", f"resids{st}.{j}.0.0.bias", "c3", cout[st], cout[st]),
                          mk(f"resids{st}.{j}.0.2.weight", None, "c3", cout[st], cout[st]),
                          bn(f"resids{st}.{j}.1", cout[st])) for j in range(resblocks)] for st in (1, 2, 3)}
        self.cout = cout
        self.sets = ShapeSets()
        self.fc_w, self.fc_b = flat.view(flat.p, "fc.weight"), flat.padded(flat.p, "fc.bias")
        self.g_fc_w, self.g_fc_b = flat.view(flat.g, "fc.weight"), flat.padded(flat.g, "fc.bias")
        self.convs = [self.conv0] + [self.blk[k][0] for k in range(1, 6)] + [c for st in (1, 2, 3) for (c1, c2, _) in
                                                                             self.res[st] for c in (c1, c2)]
        self.shape = None
        for c in self.convs:
            c.persist_wgs = K.persist_wgs("D")
        self.repacker = Repacker(self.convs, dtype_t, flat.device)
        self.finalizer = Finalizer(self.convs, flat.device) if _defer_finalize() else None
        self.res_group = WgradGroup() if TU().wgrad_groups else None
        if self.finalizer is not None and self.res_group is not None and _wgrad_lists() and \
                dtype_t in (torch.bfloat16, torch.float16):
            self.res_group = WgradList(K.persist_wgs("D"))
        # workgroups of the persistent launches per half: the REAL half runs beside the latency-bound generator chain and
        # lane B has slack there (it waits for the chain's last frame), the fake half is on the step's critical path
        self.cap = {None: K.persist_wgs("D"), 1: K.persist_wgs("D"),
                    0: TU().cap_dreal_default()}
        self.rw_extra_real = TU().rw_extra_dreal or ""
        self.bn_fuse = TU().bn_fuse
        if self.bn_fuse:
            tuning.need_experiments("bn_fuse")
        self.tail = TU().d_tail
        self._tail_scratch = {}

    # launch classes of kernels.rw_eligible added for the REAL half only (TecoGANStep sets "s1" for chain-bound steps: the real
    # half runs beside the chain with slack, and its stage-1 convs are better neighbours as capped persistent launches: 4.194 ->
    # 4.18 ms/step; for the fake half - on the critical path - the same routing costs 0.07 ms)

    def _set_cap(self, half):
        cap = self.cap[half]
        for c in self.convs:
            c.persist_wgs = cap
            c.rw_extra = self.rw_extra_real if half == 0 else ""
        if isinstance(self.res_group, WgradList):
            self.res_group.cap = cap

    def repack(self):
        self.repacker.run()

    def alloc(self, N, H):
        """selects (creating it on first use) the buffer set for N samples of H x H (see ShapeSets)"""
        if self.shape == (N, H):
            return
        self.fc_hw = (H // 32) ** 2
        if self.fc_w.numel() != 3 * self.fc_hw:
            raise L.TecoganHipError(f"fc expects {self.fc_w.numel()} inputs but the D input gives {3 * self.fc_hw} "
                                    "(code/models.py:123 hard-wires 48 = 128x128 HR)")
        cur = self.sets.get((N, H), lambda: self._make_set(N, H))
        self.act, self.gbuf, self.g_c0, self.prob, self.dlogit = (cur[k] for k in ("act", "gbuf", "g_c0", "prob", "dlogit"))
        self.shape = (N, H)

    def _make_set(self, N, H):
        dev, dt = self.flat.device, self.dt
        e = lambda hh, c: torch.empty(N, hh, hh, pad32(c), dtype=dt, device=dev)
        a = {"in": e(H, 27), "c0": e(H, 64), "z": {}, "n": {}, "h": {}, "r": {}, "net": {}}
        hh = H
        for k in range(1, 6):
            hh //= 2
            a["z"][k], a["n"][k] = e(hh, self.cout[k]), e(hh, self.cout[k])
            if k <= 3:
                a["h"][k] = [e(hh, self.cout[k]) for _ in range(self.nrb)]
                a["r"][k] = [e(hh, self.cout[k]) for _ in range(self.nrb)]
                a["net"][k] = [e(hh, self.cout[k]) for _ in range(self.nrb)]
        g = {"dz": {}, "dn": {}, "dr": {}, "dh": {}, "dnet": {}}  # one buffer per gradient tensor (see GeneratorEngine)
        hh = H
        for k in range(1, 6):
            hh //= 2
            g["dz"][k], g["dn"][k] = e(hh, self.cout[k]), e(hh, self.cout[k])
            if k <= 3:
                g["dr"][k] = [e(hh, self.cout[k]) for _ in range(self.nrb)]
                g["dh"][k] = [e(hh, self.cout[k]) for _ in range(self.nrb)]
                g["dnet"][k] = [e(hh, self.cout[k]) for _ in range(self.nrb)]
        return {"act": a, "gbuf": g, "g_c0": e(H, 64), "prob": torch.empty(N, device=dev),
                "dlogit": torch.empty(N, device=dev)}

    def stage_out(self, k):
        return self.act["net"][k][self.nrb - 1] if (k <= 3 and self.nrb > 0) else self.act["n"][k]

    def layers(self):
        return [self.stage_out(1), self.stage_out(2), self.stage_out(3), self.act["n"][4]]

    def forward(self, groups=2, update_stats=True, half=None):
        """half=None: the whole batch (`groups` BN groups).  half=0/1: only the real / fake half of a 2-group batch (lets
        the real half run while the generator is still producing the frames the fake half needs)."""
        a = self.act
        N = a["in"].shape[0]
        self._set_cap(half)
        if half is None:
            sl, st_of = slice(0, N), (lambda bn: bn.stats_slot())
        else:
            hb = N // 2
            sl, st_of = slice(half * hb, (half + 1) * hb), (lambda bn: bn.stats_slot(half))
            groups = 1
        v = lambda t: t[sl]
        self.conv0.fwd(v(a["in"]), v(a["c0"]), act=L.ACT_LRELU)
        prev = v(a["c0"])
        n_here = sl.stop - sl.start
        for k in range(1, 6):
            conv, bn = self.blk[k]
            conv.fwd(prev, v(a["z"][k]), stats=st_of(bn), groups=groups, stats_r=bn.R)
            if k == 4 and self.tail_fused(n_here // groups):
                # BN + LeakyReLU of block4, block5, fc, sigmoid: ONE single-workgroup launch (csrc/d_tail.hip, round 6)
                bn5, sv = self.blk[5][1], (lambda b: b.save if half is None else b.save[half])
                rs = lambda b: (b.rm, b.rv, b.nbt) if update_stats else (None, None, None)
                K.d_tail_fwd(v(a["z"][4]), st_of(bn), bn.R, bn.gamma, bn.beta, *rs(bn), sv(bn), v(a["n"][4]), self.blk[5][0].w,
                             v(a["z"][5]), bn5.gamma, bn5.beta, *rs(bn5), sv(bn5), v(a["n"][5]), self.fc_w, self.fc_b, self.prob[sl],
                             n_here, a["z"][4].shape[1], self.cout[5], groups, self._tail_ws(half, n_here, groups))
                return
            bn.apply(v(a["z"][k]), v(a["n"][k]), L.ACT_LRELU, groups, update=update_stats, half=half)
            net = v(a["n"][k])
            if k <= 3:
                for j, (c1, c2, bnj) in enumerate(self.res[k]):
                    c1.fwd(net, v(a["h"][k][j]), act=L.ACT_RELU)
                    c2.fwd(v(a["h"][k][j]), v(a["r"][k][j]), stats=st_of(bnj), groups=groups, stats_r=bnj.R)
                    bnj.apply(v(a["r"][k][j]), v(a["net"][k][j]), L.ACT_NONE, groups, skip=net, update=update_stats,
                              half=half)
                    net = v(a["net"][k][j])
            prev = net
        K.fc_head_fwd(v(a["n"][5]), self.fc_w, self.fc_b, self.prob[sl], sl.stop - sl.start, self.fc_hw, 3, 32)

    def _tail_ws(self, half, n, groups):
        """the tail launches' scratch for this half (or the whole batch) of the current buffer set: zero at creation, left zero by every launch"""
        key = (self.shape, half, n, groups)
        ws = self._tail_scratch.get(key)
        if ws is None:   # (a captured step's entries exist since its eager warm-up; a module forward at another shape may add its own later)
            ws = self._tail_scratch[key] = K.d_tail_scratch(n, self.act["z"][4].shape[1], groups, self.flat.device)
        return ws

    def tail_fused(self, n_per_group):
        """the tail (block4's BatchNorm ... sigmoid, and its backward) as single-workgroup launches for this many samples per BN group?"""
        return self.tail and K.d_tail_ok(n_per_group, self.act["z"][4].shape[1], pad32(self.cout[4]), self.cout[5])

    def bucket_split(self):
        """element offset in the flat gradient buffer where block2 begins: [0, split) = conv.0, block1, resids1 - the
        layers whose gradients come LAST in backward - is final after backward(part='lo'), [split, total) after 'hi'"""
        return self.flat.offsets["block2.0.weight"][0]

    def backward(self, groups=2, half=None, part=None, real_seed=None):
        """consumes self.dlogit; accumulates all D gradients (everything on the current stream).
        real_seed = (cfg, loss_scale): self.dlogit of this (real) half is first computed from self.prob (tg_dlogit_real's expression).
        half=0/1: only the real / fake half of the 2-group batch (the loss is a mean of per-half terms and BN statistics
        are per half, code/train.py:199-203,304-307, so the halves are independent in backward): the real half's backward
        can run before the generator has produced the frames the fake half needs.
        part: None = the whole pass; 'hi' = fc, block5 ... block2 (with the fold of their weight gradients), 'lo' = stage 1
        and conv.0 - the gradient buckets of data-parallel mode."""
        N = self.act["in"].shape[0]
        self._set_cap(half)
        if half is None:
            sl = slice(0, N)
        else:
            sl = slice(half * (N // 2), (half + 1) * (N // 2))
        cut = lambda v: ({k: cut(x) for k, x in v.items()} if isinstance(v, dict) else
                         [x[sl] for x in v] if isinstance(v, list) else v[sl])
        a, g = cut(self.act), cut(self.gbuf)
        n = sl.stop - sl.start
        lo_convs = [self.conv0, self.blk[1][0]] + [c for (c1, c2, _) in self.res[1] for c in (c1, c2)]
        gh = groups if half is None else 1
        tail = part != "lo" and self.tail_fused(n // gh) and (real_seed is None or gh == 1)
        if part != "lo" and tail:
            # fc, block5 and block4's BatchNorm backward: ONE single-workgroup launch (csrc/d_tail.hip); leaves d z5, d z4 and the
            # parameter gradients of fc / block5.1 / block4.1
            bn4, bn5, sv = self.blk[4][1], self.blk[5][1], (lambda b: b.save if half is None else b.save[half])
            K.d_tail_bwd(self.dlogit[sl], self.prob[sl], real_seed[0] if real_seed else None, real_seed[1] if real_seed else None,
                         real_seed is not None, a["n"][5], a["z"][5], sv(bn5), bn5.gamma, self.fc_w, self.blk[5][0].w, a["n"][4],
                         a["z"][4], sv(bn4), bn4.gamma, g["dz"][5], g["dn"][4], g["dz"][4], self.g_fc_w, self.g_fc_b, bn5.dgamma,
                         bn5.dbeta, bn4.dgamma, bn4.dbeta, n, a["z"][4].shape[1], self.cout[5], gh, self._tail_ws(half, n, gh))
            d_net = None
        elif part != "lo":
            if real_seed is not None:
                K.dlogit_real(self.prob[sl], self.dlogit[sl], n, real_seed[0], real_seed[1])
            K.fc_head_bwd(a["n"][5], self.fc_w, self.dlogit[sl], g["dn"][5], self.g_fc_w, self.g_fc_b, n, self.fc_hw, 3, 32)
            d_net = g["dn"][5]  # gradient w.r.t. the current stage's output
        else:
            d_net = g["dnet"][1][self.nrb - 1] if self.nrb > 0 else g["dn"][1]   # what part 'hi' left behind
        grouped = self.finalizer is not None and self.res_group is not None
        lists = grouped and isinstance(self.res_group, WgradList)
        wg = (lambda c, x, y, b=False: self.res_group.add(c, x, y, b)) if grouped else \
            (lambda c, x, y, b=False: c.wgrad(x, y, bias_sum=b))
        stage_out = lambda k: a["net"][k][self.nrb - 1] if (k <= 3 and self.nrb > 0) else a["n"][k]
        g_c0 = self.g_c0[sl]
        for k in (range(5, 1, -1) if part == "hi" else (1,) if part == "lo" else range(5, 0, -1)):
            if k <= 3:
                fused = False   # the stage's last block gets its output gradient from the next stage's k4 s2 input-gradient
                for j in range(self.nrb - 1, -1, -1):
                    c1, c2, bnj = self.res[k][j]
                    net_in = a["net"][k][j - 1] if j > 0 else a["n"][k]
                    d_r, d_h = g["dr"][k][j], g["dh"][k][j]
                    d_in = g["dnet"][k][j - 1] if j > 0 else g["dn"][k]
                    bnj.backward(d_net, None, a["r"][k][j], d_r, L.ACT_NONE, groups, half=half, reduced=fused)
                    wg(c2, a["h"][k][j], d_r)
                    c2.dgrad(d_r, d_h, mask=a["h"][k][j], mask_mode=L.MASK_RELU)
                    wg(c1, net_in, d_h, True)
                    # d_in is the output gradient of the previous block's BatchNorm (no activation): its backward sums ride
                    # in this launch's epilogue instead of a tg_bn_bwd_reduce launch
                    fused = self.bn_fuse and j > 0
                    if fused:
                        bnp = self.res[k][j - 1][2]
                        c1.dgrad(d_h, d_in, res=d_net, bn_sums=(bnp.red_slot(half), a["r"][k][j - 1],
                                                                groups if half is None else 1, bnp.R))
                    else:
                        c1.dgrad(d_h, d_in, res=d_net)
                    d_net = d_in
                if grouped and not lists:
                    self.res_group.launch()  # the 2*nrb same-shaped residual convs of this stage in one grid
                elif grouped and k > 1:
                    self.res_group.launch(only=(L.WGROUP_C3,))
                # (work lists: stage 1 waits for conv.0 below - one launch for both; the k4 s2 layers stay queued to the end)
            conv, bn = self.blk[k]
            d_z = g["dz"][k]
            if not (tail and k >= 4):   # (the fused tail has left d z5 and d z4)
                bn.backward(d_net, a["n"][k], a["z"][k], d_z, L.ACT_LRELU, groups, half=half)
            prev = stage_out(k - 1) if k > 1 else a["c0"]
            if lists:
                wg(conv, prev, d_z)
            else:
                conv.wgrad(prev, d_z)
            if tail and k == 5:
                continue                # (block5.0's input-gradient is inside the fused launch)
            if k > 1:
                d_prev = g["dnet"][k - 1][self.nrb - 1] if (k - 1 <= 3 and self.nrb > 0) else g["dn"][k - 1]
                conv.dgrad(d_z, d_prev)
                d_net = d_prev
            else:
                conv.dgrad(d_z, g_c0, mask=a["c0"], mask_mode=L.MASK_LRELU)
                if lists:   # conv.0 (27 -> 64, full resolution) shares stage 1's work list
                    wg(self.conv0, a["in"], g_c0, True)
                else:
                    self.conv0.wgrad(a["in"], g_c0, bias_sum=True)
        if lists:
            self.res_group.launch()   # whatever is still queued: the k4 s2 layers of this part, stage 1 + conv.0
        if self.finalizer is not None:
            self.finalizer.run(only=None if part is None else lo_convs if part == "lo" else
                               [c for c in self.convs if all(c is not x for x in lo_convs)])

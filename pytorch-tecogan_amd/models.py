"""nn.Module surface of the reference's code/models.py (generator :61-86, discriminator :97-146, f_net :22-50).

The modules are parameter containers with the reference's state_dict keys, shapes, registration order and default
initialisation; their forward runs the HIP engines of engine.py.  There is no torch-op forward: without
the HIP library the modules raise.  The first forward / training step after a device move re-binds the parameters as
views of one flat fp32 buffer per network (this is what lets one Adam launch and one RCCL all-reduce cover a network).
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import kernels as K
from . import tuning
from .step import RecurrentGenerator


def compute_dtype(args=None):
    """bf16 by default (BASELINE config 2); TECOGAN_DTYPE / args.tg_dtype = 'fp32' selects the fp32 parity mode, 'fp16' the
    reference's own autocast element type (BASELINE configs[3]) - the training step then runs with dynamic loss scaling
    (step.TecoGANStep, torch.cuda.amp.GradScaler semantics of code/train.py:9,335-342)."""
    name = getattr(args, "tg_dtype", None) or tuning.current().dtype
    name = str(name).lower()
    if name in ("fp32", "f32", "float32"):
        return torch.float32
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("fp16", "f16", "float16", "half"):
        return torch.float16
    raise ValueError(f"unsupported compute dtype {name!r} (use bf16, fp16 or fp32)")


class _Node(nn.Module):
    """anonymous container so that dotted reference names ('resids.3.0.weight') become a module tree."""

    def forward(self, *a, **k):  # pragma: no cover - never called
        raise RuntimeError("parameter container")


def _fan_in(shape):
    return shape[1] * shape[2] * shape[3]  # torch uses size(1)*receptive field for Conv2d AND ConvTranspose2d


def _build_tree(root, shapes, bn_prefixes=(), gen=None):
    """registers parameters (PyTorch default init: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +), same bound for
    the bias; BN weight 1 / bias 0) under nested _Node modules, in the order of `shapes`."""
    last_bound = 1.0
    for name, shp in shapes.items():
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        prefix = ".".join(parts[:-1])
        if prefix in bn_prefixes:
            t = torch.ones(shp) if parts[-1] == "weight" else torch.zeros(shp)
        elif len(shp) == 4:
            last_bound = 1.0 / math.sqrt(_fan_in(shp))
            t = torch.empty(shp).uniform_(-last_bound, last_bound, generator=gen)
        elif len(shp) == 2:  # denselayer: xavier_uniform weight (code/ops.py:87), default Linear bias
            b = math.sqrt(6.0 / (shp[0] + shp[1]))
            t = torch.empty(shp).uniform_(-b, b, generator=gen)
            last_bound = 1.0 / math.sqrt(shp[1])
        else:
            t = torch.empty(shp).uniform_(-last_bound, last_bound, generator=gen)
        node.register_parameter(parts[-1], nn.Parameter(t))
    for prefix in bn_prefixes:
        node = root
        for p in prefix.split("."):
            node = node._modules[p]
        c = node.weight.shape[0]
        node.register_buffer("running_mean", torch.zeros(c))
        node.register_buffer("running_var", torch.ones(c))
        node.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _HipModule(nn.Module):
    """common flat-buffer binding logic"""

    _shapes = None
    _dirty = False

    def mark_weights_changed(self):
        """The engines compute from PACKED copies of the conv weights (rebuilt after every Adam step).  Anything else that
        writes the parameters in place - load_state_dict on an already bound module (hooked below), a broadcast, a manual
        p.data.copy_() - must call this so the copies are rebuilt before the next launch."""
        self._dirty = True

    def _install_hooks(self):
        if not getattr(self, "_hooked", False):
            self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_weights_changed())
            self._hooked = True

    def _bound(self):
        flat = getattr(self, "_flat", None)
        if flat is None:
            return False
        first, last = self._edge_params
        for name, p in (first, last):  # a device move / load replaces every Parameter's storage, so two probes suffice
            if p.data_ptr() != flat.view(flat.p, name).data_ptr() or p.device != flat.p.device:
                return False
        return True

    def _bind(self, dtype_t):
        """(re)creates the flat buffers on the parameters' device and makes every Parameter a view of them."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.TecoganHipError("the HIP path needs the module on a GPU: call .cuda() first (no CPU fallback)")
        L.load()
        self._install_hooks()
        if self._bound() and self._dtype == dtype_t:
            if self._dirty:
                self._engine.repack()
                self._dirty = False
            return
        old_flat = getattr(self, "_flat", None)
        flat = E.FlatParams(self._shapes, dev)
        for name, p in self.named_parameters():
            v = flat.view(flat.p, name)
            v.copy_(p.data)
            if old_flat is not None and old_flat.p.device == dev:
                flat.view(flat.m, name).copy_(old_flat.view(old_flat.m, name))
                flat.view(flat.v, name).copy_(old_flat.view(old_flat.v, name))
            p.data = v
            p.grad = flat.view(flat.g, name)
        named = list(self.named_parameters())
        self._edge_params = (named[0], named[-1])
        self._flat, self._dtype = flat, dtype_t
        self._make_engine(flat, dtype_t)
        self._engine.repack()
        self._dirty = False

    def flat_params(self):
        return self._flat

    def engine(self, dtype_t=None):
        self._bind(dtype_t or getattr(self, "_dtype", None) or compute_dtype(self._args))
        return self._engine


class generator(_HipModule):
    """code/models.py:61-86.  forward(x[B,51,h,w]) -> [B,3,4h,4w] (inference; the training step drives the engine
    directly and batches the backward over all frames)."""

    def __init__(self, gen_output_channels, args=None):
        super().__init__()
        if args is None:
            raise ValueError("No args is provided for generator")
        self._args = args
        self.num = int(args.num_resblock)
        self._out_ch = int(gen_output_channels)
        if self._out_ch > 4:
            raise ValueError("gen_output_channels > 4 is not supported by the NCHW store epilogue")
        self._shapes = E.generator_shapes(self.num, self._out_ch)
        _build_tree(self, self._shapes)
        self._rec = None

    def _make_engine(self, flat, dtype_t):
        self._engine = E.GeneratorEngine(flat, dtype_t, self.num, self._out_ch)
        self._rec = None

    def forward(self, x):
        eng = self.engine()
        B, C_, h, w = x.shape
        if C_ != 51:
            raise ValueError("generator expects 51 input channels (3 LR + 48 packed warped HR)")
        eng.alloc(B, h, w)
        K.nchw_to_nhwc(x.contiguous().float(), C_ * h * w, eng.act["in0"], B, C_, h, w)
        out = torch.empty(B, self._out_ch, 4 * h, 4 * w, dtype=torch.float32, device=x.device)
        eng.forward(0, B, out, 0, self._out_ch * 16 * h * w)
        return out

    def recurrent(self, frames, use_graph=False):
        """(B,T,3,h,w) LR frames -> (B,T,3,4h,4w): the whole inference loop of main.py:171-219 on device."""
        B, T, _, h, w = frames.shape
        return self._rec_for(B, h, w, frames.device, use_graph).run(frames.contiguous().float())

    def _rec_for(self, B, h, w, device, use_graph):
        if self._rec is None or (self._rec.B, self._rec.h, self._rec.w, self._rec.use_graph) != (B, h, w, use_graph):
            if self._rec is not None:
                self._rec.close()
            self._rec = RecurrentGenerator(self.engine(), B, h, w, device, use_graph)
        return self._rec

    def recurrent_step(self, frame, use_graph=False, reset=False):
        """ONE frame of a live stream: (B,3,h,w) LR frame -> (B,3,4h,4w) HR frame, carrying the previous LR / HR frame across calls
        (the per-frame body of /root/reference/experimental/live.py:100-128 and main.py:191-219: flow from the previous LR frame, warp of
        the previous output, (x+1)/2, space-to-depth, generator).  reset=True (or a new frame shape) starts a new sequence: zeros as
        the previous output (main.py:189-196).  A whole-sequence recurrent() call in between also restarts the stream."""
        B, _, h, w = frame.shape
        rec = self._rec_for(B, h, w, frame.device, use_graph)
        if reset:
            rec.reset()
        return rec.step(frame.contiguous().float())


class discriminator(_HipModule):
    """code/models.py:97-146.  forward(x[N,27,H,W]) -> (prob[N,1], [4 feature maps]); BN in training mode."""

    def __init__(self, args=None):
        super().__init__()
        if args is None:
            raise ValueError("No args is provided for discriminator")
        self._args = args
        self.resblocks = int(args.discrim_resblocks)
        self.channels = int(args.discrim_channels)
        auto = getattr(args, "tg_fc_auto", False) or getattr(args, "tg_extend", False)
        fc_in = 3 * (int(getattr(args, "crop_size", 32)) * 4 // 32) ** 2 if auto else 48  # 48 hard-coded at :123
        self._shapes = E.discriminator_shapes(self.resblocks, self.channels, fc_in)
        _build_tree(self, self._shapes, bn_prefixes=E.discriminator_bn_names(self.resblocks))

    def _make_engine(self, flat, dtype_t):
        # BN buffers: kernels read 32-padded vectors; the registered buffers become views of the first C entries
        bufs = E.make_bn_buffers(self._shapes, self.resblocks, flat.device)
        for prefix in E.discriminator_bn_names(self.resblocks):
            node = self
            for p in prefix.split("."):
                node = node._modules[p]
            c = node.weight.shape[0]
            for key in ("running_mean", "running_var"):
                bufs[f"{prefix}.{key}"][:c].copy_(getattr(node, key))
                node._buffers[key] = bufs[f"{prefix}.{key}"][:c]
            bufs[f"{prefix}.num_batches_tracked"].copy_(node.num_batches_tracked)
            node._buffers["num_batches_tracked"] = bufs[f"{prefix}.num_batches_tracked"]
        self._engine = E.DiscriminatorEngine(flat, bufs, dtype_t, self.resblocks, self.channels)

    def forward(self, x):
        eng = self.engine()
        N, C_, H, W = x.shape
        if C_ != 27 or H != W:
            raise ValueError("discriminator expects [N,27,H,H]")
        eng.alloc(N, H)
        K.nchw_to_nhwc(x.contiguous().float(), C_ * H * W, eng.act["in"], N, C_, H, W)
        eng.arena.zero()
        eng.forward(groups=1, update_stats=True)
        layers = [K.to_nchw(l, l.shape[3]) for l in eng.layers()]
        return eng.prob.clone().view(N, 1), layers


class f_net(_HipModule):
    """code/models.py:22-50: defined and imported by the reference but never instantiated (main.py:231 is commented
    out; the step uses the pseudo-flow instead).  Same parameter names/shapes/init; forward(x[N,3,h,w]) -> [N,2,h,w]
    (24*tanh) runs on the HIP kernels, inference only (the reference has no training path through it)."""

    def __init__(self, args=None):
        super().__init__()
        self._args = args
        self._shapes = E.fnet_shapes()
        _build_tree(self, self._shapes)

    def _make_engine(self, flat, dtype_t):
        self._engine = E.FNetEngine(flat, dtype_t)

    def forward(self, x):
        eng = self.engine()
        N, C_, h, w = x.shape
        if C_ != 3:
            raise ValueError("f_net expects 3 input channels (code/models.py:26)")
        eng.alloc(N, h, w)
        K.nchw_to_nhwc(x.contiguous().float(), C_ * h * w, eng.act["in"], N, C_, h, w)
        out = torch.empty(N, 2, h, w, dtype=torch.float32, device=x.device)
        eng.forward(out)
        return out


class VGG19(_HipModule):
    """code/ops.py:144-213, as far as it is ever read: the conv stack up to Conv4_4 (the step taps conv2_2, conv3_4 and
    conv4_4, code/train.py:125), with the reference module's parameter names (Conv1_1.0.weight ...).  FROZEN feature
    extractor of the opt-in VGG loss (args.vgg_scaling > 0).  Weights: args.vgg_ckpt (a torch file holding a state_dict
    with these keys, extra keys ignored) or a deterministic He-uniform initialisation under numpy seed 19 - the reference
    builds a fresh randomly initialised VGG19 on every call and never loads vgg_ckpt (code/train.py:33); see DESIGN.md.
    forward(x[N,3,H,W] in [0,1]) -> dict of the three tap feature maps (fp32 NCHW), input arithmetic of
    code/train.py:31-32 included."""

    def __init__(self, args=None):
        super().__init__()
        self._args = args
        self._shapes = E.vgg_shapes()
        import numpy as np
        rng = np.random.default_rng(19)
        for name, shp in self._shapes.items():
            lname, _, leaf = name.split(".")
            if lname not in self._modules:
                self.add_module(lname, _Node())
                self._modules[lname].add_module("0", _Node())
            if leaf == "weight":
                bound = math.sqrt(6.0 / (shp[1] * 9))
                t = torch.from_numpy(rng.uniform(-bound, bound, size=shp).astype("float32"))
            else:
                t = torch.zeros(shp)
            self._modules[lname]._modules["0"].register_parameter(leaf, nn.Parameter(t, requires_grad=False))
        ck = getattr(args, "vgg_ckpt", None) if args is not None else None
        if ck:
            sd = torch.load(ck, map_location="cpu")
            sd = sd.get("model_state_dict", sd)
            self.load_state_dict({k: v for k, v in sd.items() if k in self._shapes}, strict=True)

    def _bind(self, dtype_t):
        """frozen: one flat parameter buffer, no gradient / moment twins"""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.TecoganHipError("the HIP path needs the module on a GPU: call .cuda() first (no CPU fallback)")
        L.load()
        self._install_hooks()
        if self._bound() and self._dtype == dtype_t:
            if self._dirty:
                self._engine.repack()
                self._dirty = False
            return
        flat = E.FlatParams(self._shapes, dev, train=False)
        for name, p in self.named_parameters():
            v = flat.view(flat.p, name)
            v.copy_(p.data)
            p.data = v
        named = list(self.named_parameters())
        self._edge_params = (named[0], named[-1])
        self._flat, self._dtype = flat, dtype_t
        self._engine = E.VGGEngine(flat, dtype_t)
        self._engine.repack()
        self._dirty = False

    def forward(self, x):
        eng = self.engine()
        N, C_, H, W = x.shape
        if C_ != 3:
            raise ValueError("VGG19 expects 3 input channels")
        eng.alloc(N, H, W)
        xx = x.contiguous().float()
        eng.forward(xx, xx)
        return {"vgg_19/" + t.lower(): K.to_nchw(eng.act[t][:N], eng.act[t].shape[3]) for t in E.VGG_TAPS}

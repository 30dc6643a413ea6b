"""CPU oracle for the TecoGAN per-sequence training step.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of the arithmetic of the reference's hot path.
It exists to *check* the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg);
nothing in the product package imports it and the product never falls back to it.

Parity pin: tests/test_oracle_golden.py compares every function here against fixtures produced by
oracle/make_golden.py, which imports the real reference (/root/reference/code) in the build container.
Floating-point parity is pinned that way; there are no golden vectors in the reference itself
(SURVEY.md section 4), so the pin is "reference code + torch CPU in the build container".

Reference map (file:line are into /root/reference):
  up4                     code/ops.py:98-100        nn.Upsample(scale_factor=4, bilinear, align_corners=False)
  warp                    code/train.py:81-84,98,165,187   F.grid_sample defaults (bilinear, zeros, align_corners=False)
  generator_forward       code/models.py:61-86 (+ residual_block :54-58, conv2 ops.py:57-63, conv2_tran ops.py:45-54)
  discriminator_forward   code/models.py:97-146 (+ discriminator_block :90-94, batchnorm ops.py:75-77 eps=1e-3)
  fnet_forward            code/models.py:22-50
  recurrent_generator     code/train.py:86-118 ; main.py:191-219
  d_inputs                code/train.py:129-198
  losses                  code/train.py:205-333
  adam_step               torch.optim.Adam as constructed at main.py:239-243
  tecogan_step            code/train.py:49-370
  compute_psnr            code/ops.py:130-139
"""
import collections
import math

import numpy as np
import torch
import torch.nn.functional as F

Network = collections.namedtuple(
    "Network",
    "gen_output, learning_rate, update_list, update_list_name, update_list_avg, global_step, d_loss, "
    "gen_loss, fnet_loss, tb, target",
)

LAYER_NORM = (12.0, 14.0, 24.0, 100.0)  # code/train.py:214
FIX_RANGE = 0.02  # code/train.py:206
BN_EPS = 1e-3  # code/ops.py:76
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------
# parameter construction (numpy PCG64 so fixtures do not depend on torch's RNG)
# --------------------------------------------------------------------------------------------
def generator_param_shapes(num_resblock=16, out_ch=3):
    s = collections.OrderedDict()
    s["conv.0.weight"] = (64, 51, 3, 3)
    s["conv.0.bias"] = (64,)
    for i in range(num_resblock):
        s[f"resids.{i}.0.weight"] = (64, 64, 3, 3)
        s[f"resids.{i}.0.bias"] = (64,)
        s[f"resids.{i}.2.weight"] = (64, 64, 3, 3)
    s["conv_trans.0.weight"] = (64, 64, 3, 3)  # ConvTranspose2d: [Cin, Cout, kh, kw]
    s["conv_trans.0.bias"] = (64,)
    s["conv_trans.2.0.weight"] = (64, 64, 3, 3)
    s["conv_trans.2.0.bias"] = (64,)
    s["conv_trans.2.2.weight"] = (64, 64, 3, 3)
    s["conv_trans.3.0.weight"] = (128, 64, 3, 3)
    s["conv_trans.3.0.bias"] = (128,)
    s["conv_trans.3.2.weight"] = (128, 128, 3, 3)
    s["conv_trans.4.weight"] = (128, 128, 3, 3)  # ConvTranspose2d
    s["conv_trans.4.bias"] = (128,)
    s["conv_trans.6.weight"] = (64, 128, 3, 3)
    s["conv_trans.6.bias"] = (64,)
    s["output.weight"] = (out_ch, 64, 3, 3)
    s["output.bias"] = (out_ch,)
    return s


def discriminator_param_shapes(resblocks=4, ch=128, fc_in=48):
    s = collections.OrderedDict()
    s["conv.0.weight"] = (64, 27, 3, 3)
    s["conv.0.bias"] = (64,)
    stage_c = {1: 64, 2: ch, 3: ch}
    blk = {1: (64, 64), 2: (ch, 64), 3: (ch, ch), 4: (64, ch), 5: (3, 64)}

    def add_block(k):
        co, ci = blk[k]
        s[f"block{k}.0.weight"] = (co, ci, 4, 4)
        s[f"block{k}.1.weight"] = (co,)
        s[f"block{k}.1.bias"] = (co,)

    def add_resids(st):
        c = stage_c[st]
        for j in range(resblocks):
            s[f"resids{st}.{j}.0.0.weight"] = (c, c, 3, 3)
            s[f"resids{st}.{j}.0.0.bias"] = (c,)
            s[f"resids{st}.{j}.0.2.weight"] = (c, c, 3, 3)
            s[f"resids{st}.{j}.1.weight"] = (c,)
            s[f"resids{st}.{j}.1.bias"] = (c,)

    # order mirrors nn.Module registration order of code/models.py:102-123
    add_block(1)
    add_resids(1)
    add_block(2)
    add_resids(2)
    add_block(3)
    add_resids(3)
    add_block(4)
    add_block(5)
    s["fc.weight"] = (1, fc_in)
    s["fc.bias"] = (1,)
    return s


def discriminator_bn_names(resblocks=4):
    names = []
    for st in (1, 2, 3):
        names.append(f"block{st}.1")
        names += [f"resids{st}.{j}.1" for j in range(resblocks)]
    names += ["block4.1", "block5.1"]
    return names


def fnet_param_shapes():
    s = collections.OrderedDict()
    chans = [("down1", 3, 32), ("down2", 32, 64), ("down3", 64, 128), ("down4", 128, 256),
             ("up1", 256, 512), ("up2", 512, 256), ("up3", 256, 128), ("up4", 128, 64)]
    for name, ci, co in chans:
        s[f"{name}.0.weight"] = (co, ci, 3, 3)
        s[f"{name}.0.bias"] = (co,)
        s[f"{name}.2.weight"] = (co, co, 3, 3)
        s[f"{name}.2.bias"] = (co,)
    s["output_block.0.weight"] = (32, 64, 3, 3)
    s["output_block.0.bias"] = (32,)
    s["output_block.2.weight"] = (2, 32, 3, 3)
    s["output_block.2.bias"] = (2,)
    return s


def _fan_in(name, shape):
    if len(shape) == 1:
        return None
    if len(shape) == 2:
        return shape[1]
    # ConvTranspose2d weights are [Cin, Cout, kh, kw]; torch's fan_in for them is size(1)*kh*kw too.
    return shape[1] * shape[2] * shape[3]


def init_params(shapes, seed, bn_affine_random=True):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights; the matching bound for biases; BN gamma in
    [0.5,1.5), beta in [-0.2,0.2) (random so that BN-backward is exercised non-trivially)."""
    rng = np.random.default_rng(seed)
    out = collections.OrderedDict()
    last_bound = 1.0
    for name, shape in shapes.items():
        fi = _fan_in(name, shape)
        if fi is not None:
            last_bound = 1.0 / math.sqrt(fi)
            a = rng.uniform(-last_bound, last_bound, size=shape)
        elif _is_bn(name) and name.endswith(".weight"):
            a = rng.uniform(0.5, 1.5, size=shape) if bn_affine_random else np.ones(shape)
        elif _is_bn(name):
            a = rng.uniform(-0.2, 0.2, size=shape) if bn_affine_random else np.zeros(shape)
        else:
            a = rng.uniform(-last_bound, last_bound, size=shape)
        out[name] = torch.from_numpy(a.astype(np.float32))
    return out


def _is_bn(name):
    # BN parameter names in the discriminator: block{k}.1.{weight,bias}, resids{s}.{j}.1.{weight,bias}
    parts = name.split(".")
    if parts[0].startswith("block") and parts[1] == "1":
        return True
    if parts[0].startswith("resids") and len(parts) == 4 and parts[2] == "1":
        return True
    return False


def init_bn_buffers(shapes_or_params, resblocks=4):
    bufs = collections.OrderedDict()
    for bn in discriminator_bn_names(resblocks):
        c = shapes_or_params[bn + ".weight"]
        c = c[0] if isinstance(c, tuple) else c.shape[0]
        bufs[bn + ".running_mean"] = torch.zeros(c)
        bufs[bn + ".running_var"] = torch.ones(c)
        bufs[bn + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return bufs


# --------------------------------------------------------------------------------------------
# small ops
# --------------------------------------------------------------------------------------------
def up4(x):
    """code/ops.py:98-100: bilinear x4, align_corners=False."""
    return F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False)


def as_grid(block):
    """Reinterpret a contiguous (..., 2, H, W) block as a (..., H, W, 2) sampling grid
    (code/train.py:96 uses .view, :84/:157 use reshape: a reinterpretation, never a permute)."""
    shp = block.shape
    return block.contiguous().reshape(*shp[:-3], shp[-2], shp[-1], 2)


def warp(img, grid):
    """F.grid_sample defaults; both arguments promoted to fp32 (what CUDA autocast does, SURVEY 8c shim 3)."""
    return F.grid_sample(img.float(), grid.float(), mode="bilinear", padding_mode="zeros", align_corners=False)


def fp16_round(t):
    """`.half()` at code/train.py:98,187 followed by the fp32 promotion inside grid_sampler."""
    return t.half().float()


def pixel_unshuffle4(x):
    """code/train.py:102-106: view(B,3,h,4,w,4).permute(0,1,3,5,2,4).reshape(B,48,h,w)."""
    return F.pixel_unshuffle(x, 4)


def compute_psnr(ref, target):
    """code/ops.py:130-139 (expects 0..255-scaled inputs)."""
    diff = target.float() - ref.float()
    mse = (diff * diff).sum() / diff.numel()
    return 10.0 * (torch.log(255.0 * 255.0 / mse) / math.log(10.0))


# --------------------------------------------------------------------------------------------
# networks (functional)
# --------------------------------------------------------------------------------------------
def _c3(x, p, name, bias=True):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"] if bias else None, stride=1, padding=1)


def generator_forward(p, x, num_resblock=16, taps=None):
    """code/models.py:78-86.  x: [N,51,h,w] -> [N,3,4h,4w]."""
    net = F.relu(_c3(x, p, "conv.0"))
    for i in range(num_resblock):
        r = _c3(F.relu(_c3(net, p, f"resids.{i}.0")), p, f"resids.{i}.2", bias=False)
        net = r + net
        if taps is not None and i in (0, num_resblock - 1):
            taps[f"resid{i}"] = net
    net = F.relu(F.conv_transpose2d(net, p["conv_trans.0.weight"], p["conv_trans.0.bias"], stride=2, padding=1,
                                    output_padding=1))
    if taps is not None:
        taps["ct0"] = net
    net = _c3(F.relu(_c3(net, p, "conv_trans.2.0")), p, "conv_trans.2.2", bias=False)  # no skip
    net = _c3(F.relu(_c3(net, p, "conv_trans.3.0")), p, "conv_trans.3.2", bias=False)  # no skip
    if taps is not None:
        taps["ct3"] = net
    net = F.relu(F.conv_transpose2d(net, p["conv_trans.4.weight"], p["conv_trans.4.bias"], stride=2, padding=1,
                                    output_padding=1))
    net = F.relu(_c3(net, p, "conv_trans.6"))
    if taps is not None:
        taps["ct6"] = net
    net = _c3(net, p, "output")
    return torch.sigmoid(net)


def _bn_train(x, p, bufs, name, update=True):
    rm = bufs[name + ".running_mean"] if update else None
    rv = bufs[name + ".running_var"] if update else None
    y = F.batch_norm(x, rm, rv, p[name + ".weight"], p[name + ".bias"], training=True, momentum=BN_MOMENTUM,
                     eps=BN_EPS)
    if update:
        bufs[name + ".num_batches_tracked"] += 1
    return y


def discriminator_forward(p, bufs, x, resblocks=4, update_stats=True):
    """code/models.py:125-146.  x: [N,27,H,W] -> (prob [N,1], [4 feature maps]).  BN always in training mode
    (modules are never put in eval() by the reference train loop)."""
    layers = []
    net = F.leaky_relu(_c3(x, p, "conv.0"), 0.2)

    def block(net, k):
        net = F.conv2d(net, p[f"block{k}.0.weight"], None, stride=2, padding=1)
        return F.leaky_relu(_bn_train(net, p, bufs, f"block{k}.1", update_stats), 0.2)

    for st in (1, 2, 3):
        net = block(net, st)
        for j in range(resblocks):
            r = _c3(F.relu(_c3(net, p, f"resids{st}.{j}.0.0")), p, f"resids{st}.{j}.0.2", bias=False)
            net = _bn_train(r, p, bufs, f"resids{st}.{j}.1", update_stats) + net
        layers.append(net)
    net = block(net, 4)
    layers.append(net)
    net = block(net, 5)
    net = net.reshape(net.shape[0], -1)
    net = F.linear(net, p["fc.weight"], p["fc.bias"])
    return torch.sigmoid(net), layers


def fnet_forward(p, x):
    """code/models.py:37-50."""
    net = x
    for name in ("down1", "down2", "down3", "down4"):
        net = F.leaky_relu(_c3(net, p, name + ".0"), 0.2)
        net = F.leaky_relu(_c3(net, p, name + ".2"), 0.2)
        net = F.max_pool2d(net, 2)
    for name in ("up1", "up2", "up3", "up4"):
        net = F.leaky_relu(_c3(net, p, name + ".0"), 0.2)
        net = F.leaky_relu(_c3(net, p, name + ".2"), 0.2)
        net = F.interpolate(net, scale_factor=2, mode="bilinear", align_corners=False)
    net = F.leaky_relu(_c3(net, p, "output_block.0"), 0.2)
    net = _c3(net, p, "output_block.2")
    return torch.tanh(net) * 24.0


# --------------------------------------------------------------------------------------------
# the step
# --------------------------------------------------------------------------------------------
def pseudo_flow(x, fnet_params=None):
    """code/train.py:71-77.  x: (B,T,3,h,h) -> (B,T-1,2,4h,4h), values in [0,4].
    fnet_params (NOT reference behaviour, opt-in, parity unpinned): the flow estimator the reference
    defines but never calls replaces the raw LR frame at code/train.py:74, i.e. gen_flow_lr = f_net(Frame_t_pre)."""
    B, T, C, h, w = x.shape
    prev = x[:, :-1].reshape(B * (T - 1), C, h, w)
    if fnet_params is not None:
        return up4(fnet_forward(fnet_params, prev).detach() * 4.0).reshape(B, T - 1, 2, 4 * h, 4 * w)
    f = up4(prev * 4.0)
    return f[:, 0:2].reshape(B, T - 1, 2, 4 * h, 4 * w)


def recurrent_generator(gp, x, flow, num_resblock=16, fp16_grid=True):
    """code/train.py:86-118 / main.py:191-219.  Every generator input is detached (no BPTT)."""
    B, T, _, h, w = x.shape
    outs = []
    zeros = torch.zeros(B, 48, h, w, dtype=torch.float32)
    out = generator_forward(gp, torch.cat((x[:, 0], zeros), dim=1).detach(), num_resblock)
    outs.append(out)
    for i in range(T - 1):
        g = as_grid(flow[:, i])
        if fp16_grid:
            g = fp16_round(g)
        wpd = warp(out, g)
        packed = pixel_unshuffle4((wpd + 1.0) / 2.0)
        out = generator_forward(gp, torch.cat((x[:, i + 1], packed), dim=1).detach(), num_resblock)
        outs.append(out)
    return torch.stack(outs, dim=1)


def t_velocity(x, flow, t_size, pingpang=False, extended=False):
    """code/train.py:138-158 -> (B*t_size, H, H, 2) detached grid block.
    extended=True (NOT reference behaviour, parity unpinned): for t_size//3 != 3 the reference's
    reshape(B*K, 6, h, h)[0:B] -> (B, K, 2, H, H) raises; the extension keeps the first B*K*2 planes of the flattened
    tensor, which is exactly what the reference keeps when K == 3."""
    B = x.shape[0]
    h = x.shape[-1]
    H = 4 * h
    K = t_size // 3
    v_pre = flow[:, 0:t_size:3]
    v_mid = torch.zeros_like(v_pre)
    if not pingpang:
        back_in = torch.cat((x[:, 2:t_size:3], x[:, 1:t_size:3]), dim=1)
        if extended and K != 3:
            back = up4(back_in.reshape(1, B * 2 * K * 3, h, h)[:, :B * K * 2] * 4.0).reshape(B, K, 2, H, H)
        else:
            back = up4(back_in.reshape(B * K, 6, h, h)[0:B] * 4.0).reshape(B, K, 2, H, H)  # rows 0..B-1 only (quirk)
        v_nxt = back * 2.0 - 1.0
    else:
        v_nxt = torch.flip(flow, dims=[1])[:, 1:t_size:3]
    tv = torch.stack([v_pre, v_mid, v_nxt], dim=2)
    return tv.reshape(B * t_size, H, H, 2).detach()


def zero_border(z, o):
    """resized_crop(top=left=o, size H-2o) then F.pad(o) (code/train.py:160-174) == zero the outer o pixels."""
    if o == 0:
        return z
    H = z.shape[-1]
    return F.pad(z[..., o:H - o, o:H - o], (o, o, o, o), "constant", 0.0)


def d_inputs(x, y, gen, t_vel, t_size, crop_dt):
    """code/train.py:160-198 -> (real_in, fake_in), each (t_batch, 27, H, H)."""
    B = x.shape[0]
    h = x.shape[-1]
    H = 4 * h
    tb = B * t_size // 3
    if crop_dt < 1.0:
        o = (H - int(H * crop_dt)) // 2
    else:
        o = 0
    tgt = y[:, :t_size].reshape(B * t_size, 3, H, H)
    gen_t = gen[:, :t_size].reshape(B * t_size, 3, H, H)
    tgt9 = tgt.reshape(tb, 9, H, H)
    hi = up4(x[:, :t_size].reshape(tb, 9, h, h))
    real_w = zero_border(warp(tgt, t_vel).reshape(tb, 9, H, H), o)
    fake_w = zero_border(warp(gen_t, fp16_round(t_vel)).reshape(tb, 9, H, H), o)
    real_in = torch.cat((tgt9, real_w, hi), dim=1)
    fake_in = torch.cat((tgt9, fake_w, hi), dim=1)  # target frames again, not generated ones (:197-198)
    return real_in, fake_in


def default_args(**over):
    import argparse
    a = dict(RNN_N=10, crop_size=32, num_resblock=16, discrim_resblocks=4, discrim_channels=128, pingpang=False,
             pp_scaling=1.0, vgg_scaling=-0.002, crop_dt=0.75, Dt_mergeDs=True, D_LAYERLOSS=True, EPS=1e-12,
             ratio=0.01, Dt_ratio_0=1.0, Dt_ratio_add=0.0, Dt_ratio_max=1.0, learning_rate=1e-4, beta=0.9,
             adameps=1e-8, decay_step=250, decay_rate=0.8, max_epochs=1)
    a.update(over)
    return argparse.Namespace(**a)


class AdamState:
    """torch.optim.Adam(lr, betas=(beta,0.999), eps) restated (main.py:239-243)."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        for k, p in params.items():
            g = grads[k]
            if g is None:
                continue
            self.m[k].mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-(self.lr / bc1))


# ---------------------------------------------------------------------------------------------------------------------
# Opt-in VGG feature loss.  PARITY UNPINNED: the reference's path (code/train.py:30-45,124-127,253-273, code/ops.py:144-213)
# cannot execute - VGG19_slim is called without its `reuse` argument, Conv3_1.. have no kernel size, torch.min(dim=1)
# returns a tuple, the three layer terms have different shapes, and a fresh random VGG19 is built per call.  What is
# stated here is the documented fix: the TensorFlow original's semantics, which the reference transcribed.
VGG_MEAN = (123.68, 116.78, 103.94)  # code/train.py:6
VGG_LAYERS = (("Conv1_1", 3, 64), ("Conv1_2", 64, 64), "pool1", ("Conv2_1", 64, 128), ("Conv2_2", 128, 128), "pool2",
              ("Conv3_1", 128, 256), ("Conv3_2", 256, 256), ("Conv3_3", 256, 256), ("Conv3_4", 256, 256), "pool3",
              ("Conv4_1", 256, 512), ("Conv4_2", 512, 512), ("Conv4_3", 512, 512), ("Conv4_4", 512, 512))
VGG_TAPS = ("Conv2_2", "Conv3_4", "Conv4_4")  # vgg_19/conv2_2, conv3_4, conv4_4 (code/train.py:125)


def vgg_param_shapes():
    s = collections.OrderedDict()
    for l in VGG_LAYERS:
        if isinstance(l, tuple):
            s[f"{l[0]}.0.weight"], s[f"{l[0]}.0.bias"] = (l[2], l[1], 3, 3), (l[2],)
    return s


def vgg_default_params():
    """the build's default frozen extractor: He-uniform weights under numpy seed 19, zero biases (models.VGG19)"""
    rng = np.random.default_rng(19)
    out = collections.OrderedDict()
    for name, shp in vgg_param_shapes().items():
        if name.endswith("weight"):
            b = math.sqrt(6.0 / (shp[1] * 9))
            out[name] = torch.from_numpy(rng.uniform(-b, b, size=shp).astype(np.float32))
        else:
            out[name] = torch.zeros(shp)
    return out


def vgg_features(vp, img):
    """img [N,3,H,W] in [0,1] -> the three tap feature maps; input arithmetic of code/train.py:31-32 (deprocess, *255,
    - VGG_MEAN per channel), 3x3 convs + ReLU + 2x2 max-pools of code/ops.py:146-165"""
    x = (img + 1) / 2 * 255.0 - torch.tensor(VGG_MEAN, dtype=img.dtype).view(1, 3, 1, 1)
    feats = {}
    for l in VGG_LAYERS:
        if isinstance(l, tuple):
            x = F.relu(F.conv2d(x, vp[f"{l[0]}.0.weight"].to(x.dtype), vp[f"{l[0]}.0.bias"].to(x.dtype), padding=1))
            if l[0] in VGG_TAPS:
                feats[l[0]] = x
        else:
            x = F.max_pool2d(x, 2, 2)
    return feats


def vgg_loss_terms(vp, s_gen, s_tgt):
    """per tap layer: 1 - mean over pixels of the cosine between the channel-normalised feature vectors
    (f / sqrt(sum_c f^2 + 1e-12): the norm code/train.py:39-40 means; 1 - reduce_mean of the product summed over
    channels: code/train.py:258-262 made shape-consistent).  The target branch carries no gradient."""
    fg = vgg_features(vp, s_gen)
    with torch.no_grad():
        ft = vgg_features(vp, s_tgt)
    terms = []
    for t in VGG_TAPS:
        g, tt = fg[t], ft[t]
        g = g / torch.sqrt(torch.sum(g * g, dim=1, keepdim=True) + 1e-12)
        tt = tt / torch.sqrt(torch.sum(tt * tt, dim=1, keepdim=True) + 1e-12)
        terms.append(1.0 - torch.mean(torch.sum(g * tt, dim=1)))
    return terms


def tecogan_forward(gp, dp, dbufs, x, y, args, global_step, counter1=0.0, counter2=0.0, update_stats=True):
    """Forward part of code/train.py:49-333.  gp/dp tensors may require grad.  Returns a dict of everything
    the tests compare (losses are fp32 tensors attached to the autograd graph where the reference's are)."""
    global_step = global_step + 1
    T = int(args.RNN_N)
    if args.pingpang:
        x = torch.cat([x, torch.flip(x, dims=[1])[:, 1:]], dim=1)
        y = torch.cat([y, torch.flip(y, dims=[1])[:, 1:]], dim=1)
        T = 2 * T - 1
    B = x.shape[0]
    h = args.crop_size
    H = 4 * h
    lr_prev = x[:, :-1].reshape(B * (T - 1), 3, h, h)
    lr_next = x[:, 1:].reshape(B * (T - 1), 3, h, h)
    fnet_params = getattr(args, "tg_fnet_params", None)
    if fnet_params is not None and bool(getattr(args, "tg_fnet_train", False)):
        # Opt-in, parity unpinned (DESIGN.md "FNet training"): the estimator is TRAINED, wired the way the
        # reference's dead code hints (main.py:231,244-245; code/train.py:343-346: a third optimiser stepping on fnet_loss).
        # Every generator input is detached (code/train.py:90,108), so the only term of fnet_loss that reaches the estimator
        # is the LR warp loss of code/train.py:78-84,247-249 - with the estimator's output in the place of the raw next frame
        # as the sampling grid (the same (2,h,w) -> (h,w,2) reinterpretation the reference applies to every flow block).
        fx = fnet_forward(fnet_params, lr_prev)
        flow = up4(fx.detach() * 4.0).reshape(B, T - 1, 2, H, H)
        lr_warp = warp(lr_prev, as_grid(fx))
    else:
        flow = pseudo_flow(x, fnet_params)
        lr_warp = warp(lr_prev, x[:, 1:, 0:2].reshape(B * (T - 1), h, h, 2))
    gen = recurrent_generator(gp, x, flow, int(args.num_resblock))
    s_gen = gen.reshape(B * T, 3, H, H)
    s_tgt = y.reshape(B * T, 3, H, H)

    names, vals = [], []
    t_size = 3 * (T // 3)
    t_vel = t_velocity(x, flow, t_size, args.pingpang, bool(getattr(args, "tg_extend", False)))
    real_in, fake_in = d_inputs(x, y, gen, t_vel, t_size, args.crop_dt)
    p_real, L_real = discriminator_forward(dp, dbufs, real_in, int(args.discrim_resblocks), update_stats)
    p_fake, L_fake = discriminator_forward(dp, dbufs, fake_in.detach(), int(args.discrim_resblocks), update_stats)

    layer_sum = 0
    if args.D_LAYERLOSS:
        ll = []
        for i in range(len(L_real)):
            l = torch.mean(torch.sum(torch.abs(L_real[i].detach() - L_fake[i].detach()), dim=[3]))
            ll.append(l)
            layer_sum = layer_sum + FIX_RANGE * l / LAYER_NORM[i]
        vals += ll
        names += [f"D_layer_{i}_loss" for i in range(len(ll))]
        vals.append(layer_sum)
        names.append("D_layer_loss_sum")

    content = torch.mean(torch.sum(torch.square(s_gen - s_tgt), dim=[3]))
    # aliasing quirk (code/train.py:244-245,293-294,299): gen_loss and fnet_loss are ONE tensor mutated in place.
    total = content.clone()
    content_slot = len(vals)
    vals.append(None)
    names.append("l2_content_loss")
    warp_loss = torch.mean(torch.sum(torch.square(lr_next - lr_warp), dim=[3]))
    vals.append(warp_loss)
    names.append("l2_warp_loss")
    if float(getattr(args, "vgg_scaling", -1.0)) > 0.0:  # opt-in, parity unpinned (see vgg_loss_terms)
        vp = getattr(args, "tg_vgg_params", None) or vgg_default_params()
        terms = vgg_loss_terms(vp, s_gen, s_tgt)
        vgg_all = terms[0] + terms[1] + terms[2]
        # gen_loss += s*vgg ; fnet_loss += s*vgg.detach() on ONE aliased tensor (code/train.py:268-269)
        total = total + args.vgg_scaling * vgg_all + args.vgg_scaling * vgg_all.detach()
        vals += terms + [vgg_all]
        names += ["vgg_loss_2", "vgg_loss_3", "vgg_loss_4", "vgg_all"]
    pp = None
    if args.pingpang:
        n = int(args.RNN_N)
        pp = torch.mean(torch.abs(gen[:, 0:n - 1] - torch.flip(gen, dims=[1])[:, :n - 1]))
        if args.pp_scaling > 0:
            total = total + 2.0 * pp * args.pp_scaling  # added to gen_loss and fnet_loss == same tensor, twice
        vals.append(pp)
        names.append("PingPang")
    t_adv = torch.mean(-torch.log(p_fake.detach() + args.EPS))
    d_adv = torch.mean(-torch.log(p_fake + args.EPS))
    dt_ratio = torch.min(torch.tensor(args.Dt_ratio_max),
                         args.Dt_ratio_0 + args.Dt_ratio_add * torch.tensor(global_step, dtype=torch.float32))
    total = total + 2.0 * args.ratio * t_adv
    vals.append(t_adv)
    names.append("t_adversarial_loss")
    if args.D_LAYERLOSS:
        total = total + layer_sum * dt_ratio
    vals[content_slot] = total  # the list holds the aliased tensor, so it reports the final value

    fake_l = torch.log(1 - p_fake + args.EPS)
    real_l = torch.log(p_real + args.EPS)
    d_loss = torch.mean(-(fake_l + real_l))
    t_balance = torch.mean(real_l) + d_adv
    vals += [d_loss, torch.mean(p_real), torch.mean(p_fake), total]
    names += ["t_discrim_loss", "t_discrim_real_output", "t_discrim_fake_output", "All_loss_Gen"]
    tb = 0.99 * t_balance  # fresh EMA(0.99) seeded with zeros every call (code/train.py:324-327)
    avg = []
    shadow = torch.zeros(())
    for v in vals:
        shadow = 0.99 * v.detach() + 0.01 * shadow
        avg.append(shadow)
    avg += [tb.detach(), dt_ratio, counter1, counter2]
    names_avg = names + ["t_balance", "Dst_ratio", "withD_counter", "w_o_D_counter"]
    return dict(gen=gen, flow=flow, lr_warp=lr_warp, t_vel=t_vel, real_in=real_in, fake_in=fake_in,
                p_real=p_real, p_fake=p_fake, L_real=L_real, L_fake=L_fake, content=content, gen_loss=total,
                d_loss=d_loss, tb=tb, update_list=vals, update_list_name=names_avg, update_list_avg=avg,
                global_step=global_step, pp=pp, warp_loss=warp_loss)


def tecogan_step(gp, dp, dbufs, opt_g, opt_d, x, y, args, global_step, counter1=0.0, counter2=0.0,
                 return_grads=False, opt_f=None):
    """One full step (forward, both backward passes, both Adam updates), in place on gp/dp/dbufs/opt_*.
    opt_f (with args.tg_fnet_params and args.tg_fnet_train): the estimator's Adam, stepping on the LR warp loss."""
    fp = getattr(args, "tg_fnet_params", None) if (opt_f is not None and getattr(args, "tg_fnet_train", False)) else None
    for t in list(gp.values()) + list(dp.values()) + (list(fp.values()) if fp else []):
        t.requires_grad_(True)
        t.grad = None
    f = tecogan_forward(gp, dp, dbufs, x, y, args, global_step, counter1, counter2)
    g_grads = torch.autograd.grad(f["gen_loss"], list(gp.values()), retain_graph=True, allow_unused=True)
    d_grads = torch.autograd.grad(f["d_loss"], list(dp.values()), allow_unused=True, retain_graph=fp is not None)
    f_grads = torch.autograd.grad(f["warp_loss"], list(fp.values()), allow_unused=True) if fp else None
    for t in list(gp.values()) + list(dp.values()) + (list(fp.values()) if fp else []):
        t.requires_grad_(False)
    gg = dict(zip(gp.keys(), g_grads))
    dg = dict(zip(dp.keys(), d_grads))
    with torch.no_grad():
        opt_g.step(gp, gg)
        opt_d.step(dp, dg)
        if fp:
            f["fnet_grads"] = dict(zip(fp.keys(), f_grads))
            opt_f.step(fp, f["fnet_grads"])
    net = Network(gen_output=f["gen"].detach(), learning_rate=args.learning_rate,
                  update_list=[v.detach() for v in f["update_list"]], update_list_name=f["update_list_name"],
                  update_list_avg=f["update_list_avg"], global_step=f["global_step"], d_loss=f["d_loss"].detach(),
                  gen_loss=f["gen_loss"].detach(), fnet_loss=f["gen_loss"].detach(), tb=f["tb"].detach(),
                  target=f["real_in"])
    if return_grads:
        return net, gg, dg, f
    return net


def generator_content_grads(gp, x, y, dtype=torch.float64, num_resblock=16):
    """Content-loss gradient of the generator (== the whole G gradient, SURVEY finding 3) evaluated in `dtype`.
    Used by the tests as an fp64 yardstick: some trunk gradients are ~1e-7 in magnitude and cancel heavily, so the
    fp32 PyTorch-CPU reference itself is only good to ~1.5e-3 relative on them (measured: fp32 8-thread vs fp64 1.6e-3,
    8-thread vs 1-thread 1.2e-3 on resids.4.0.weight)."""
    global warp
    p = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in gp.items()}
    xx, yy = x.to(dtype), y.to(dtype)
    saved = warp
    warp = lambda img, grid: F.grid_sample(img, grid.to(img.dtype), mode="bilinear", padding_mode="zeros",  # noqa: E731
                                           align_corners=False)
    try:
        gen = recurrent_generator(p, xx, pseudo_flow(xx), num_resblock)
    finally:
        warp = saved
    B, T = x.shape[:2]
    H = y.shape[-1]
    loss = torch.mean(torch.sum(torch.square(gen.reshape(B * T, 3, H, H) - yy.reshape(B * T, 3, H, H)), dim=[3]))
    g = torch.autograd.grad(loss, list(p.values()))
    return dict(zip(p.keys(), g)), gen.detach()


def discriminator_loss_grads(dp, real_in, fake_in, eps=1e-12, dtype=torch.float64, resblocks=4):
    """Gradient of t_discrim_loss (code/train.py:304-307) w.r.t. every D parameter, evaluated in `dtype` on fixed D inputs.
    fp64 yardstick for the tests: with BN batches of 3 samples the fp32 PyTorch-CPU reference is itself ~1e-2 relative
    from the fp64 value on several BN/bias gradients (measured, deterministic across thread counts)."""
    p = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in dp.items()}
    b = {k: (v.to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in init_bn_buffers(dp, resblocks).items()}
    pr, _ = discriminator_forward(p, b, real_in.detach().to(dtype), resblocks)
    pf, _ = discriminator_forward(p, b, fake_in.detach().to(dtype), resblocks)
    loss = torch.mean(-(torch.log(1 - pf + eps) + torch.log(pr + eps)))
    return dict(zip(p.keys(), torch.autograd.grad(loss, list(p.values()))))

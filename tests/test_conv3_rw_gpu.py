"""Parity of the persistent register-weights 3x3 kernel (csrc/conv3_rw.hip, tg_conv3x3_rw) against torch fp32 convolutions on
bf16-rounded operands: forward and input-gradient (mirrored taps), Cin 64 / 128, one and several output-channel tiles, ragged
image sizes (partial tiles), persistent grids (several tiles per workgroup, max_workgroups capped), every epilogue option
(+bias, +res, ReLU / LeakyReLU, act'(mask), per-channel statistics with groups).  Replaces aten::conv2d /
convolution_backward(input) of code/models.py:54-58,68,73-76,102."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import _lib as L  # noqa: E402
from pytorch_tecogan_amd import kernels as K  # noqa: E402
from parity import assert_rel_l2  # noqa: E402  (whole-tensor relative L2 <= 4e-3: tests/parity.py)

DEV = "cuda:0"
BF = torch.bfloat16


@pytest.fixture(autouse=True, params=["rw", "cw"])
def which_kernel(request):
    """every case runs on csrc/conv3_rw.hip (producer / consumer waves) and, for 64 reduction channels, on csrc/conv3_cw.hip (eight
    equal waves, round 5): K.conv3x3_rw routes by K.C3_CW_FORCE"""
    K.C3_CW_FORCE = request.param == "cw"
    yield request.param
    K.C3_CW_FORCE = None


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32))


def q(t):
    return t.to(BF).float()


def packed(spec, w, dgrad):
    rows, Kd, s_row, s_k = spec.dgrad_pack() if dgrad else spec.fwd_pack()
    return K.pack_weights(BF, w.to(DEV).contiguous(), rows, Kd, s_row, s_k, 9, K.slot_table(9, DEV))


CASES = [  # cin, cout, N, H, W, max_workgroups
    (64, 64, 2, 32, 32, 0),
    (64, 64, 3, 40, 24, 4),      # 3*5*2 = 30 tiles on 4 workgroups: 8 rounds, partial last round
    (64, 128, 2, 20, 50, 6),     # ragged rows and columns, two channel tiles
    (64, 64, 1, 8, 16, 0),       # a single tile
    (64, 64, 5, 7, 9, 2),        # image smaller than a tile
    (128, 128, 2, 24, 32, 8),
    (128, 64, 3, 16, 40, 3),
    (128, 256, 1, 13, 21, 0),
]


@pytest.mark.parametrize("cout,N,H,W,cap,act", [(64, 2, 32, 32, 0, L.ACT_LRELU), (64, 3, 40, 24, 4, L.ACT_RELU), (128, 2, 20, 50, 6, L.ACT_NONE),
                                                (64, 1, 8, 16, 0, L.ACT_LRELU), (64, 5, 7, 9, 2, L.ACT_LRELU), (64, 12, 128, 128, 96, L.ACT_LRELU)])
def test_cw_forward_27_input_channels(cout, N, H, W, cap, act, which_kernel):
    """the 32-channel form of csrc/conv3_cw.hip (the discriminator's first layer, 27 -> 64, code/models.py:102): + bias, activation"""
    if which_kernel != "cw":
        pytest.skip("32 reduction channels exist on conv3_cw only")
    spec = K.ConvSpec("c3", 27, cout)
    x, w, b = q(rnd((N, 27, H, W), 31)), q(rnd(spec.weight_shape, 32, -0.1, 0.1)), rnd((cout,), 33)
    y = F.conv2d(x, w, b, 1, 1)
    ref = F.relu(y) if act == L.ACT_RELU else (F.leaky_relu(y, 0.2) if act == L.ACT_LRELU else y)
    out = torch.empty(N, H, W, cout, dtype=BF, device=DEV)
    K.conv3x3_rw(K.to_nhwc(x.to(DEV), BF), packed(spec, w, False), out, False, bias=b.to(DEV), act=act, max_workgroups=cap)
    torch.testing.assert_close(K.to_nchw(out, cout).cpu(), ref, rtol=2e-2, atol=2e-2)
    assert_rel_l2(K.to_nchw(out, cout).cpu(), ref, BF)


@pytest.mark.parametrize("cin,cout,N,H,W,cap", CASES)
def test_rw_forward_bias_relu(cin, cout, N, H, W, cap):
    spec = K.ConvSpec("c3", cin, cout)
    x, w, b = q(rnd((N, cin, H, W), 1)), q(rnd(spec.weight_shape, 2, -0.1, 0.1)), rnd((cout,), 3)
    ref = F.relu(F.conv2d(x, w, b, 1, 1))
    out = torch.empty(N, H, W, cout, dtype=BF, device=DEV)
    K.conv3x3_rw(K.to_nhwc(x.to(DEV), BF), packed(spec, w, False), out, False, bias=b.to(DEV), act=L.ACT_RELU, max_workgroups=cap)
    torch.testing.assert_close(K.to_nchw(out, cout).cpu(), ref, rtol=2e-2, atol=2e-2)
    assert_rel_l2(K.to_nchw(out, cout).cpu(), ref, BF)


@pytest.mark.parametrize("cin,cout,N,H,W,cap", CASES)
def test_rw_input_gradient_with_residual_and_mask(cin, cout, N, H, W, cap):
    """dgrad of conv(cin -> cout): dout [N,cout,H,W] -> din [N,cin,H,W] = (conv_transpose(dout, w) + res) * (mask > 0)"""
    if cout % 64 or cin % 64 or cout not in (64, 128):
        pytest.skip("the input-gradient's reduction dimension is cout: 64 or 128")
    spec = K.ConvSpec("c3", cin, cout)
    w = q(rnd(spec.weight_shape, 4, -0.1, 0.1))
    dout, res, mask = q(rnd((N, cout, H, W), 5)), q(rnd((N, cin, H, W), 6)), q(rnd((N, cin, H, W), 7))
    ref = (F.conv_transpose2d(dout, w, None, 1, 1) + res) * (mask > 0).float()
    out = torch.empty(N, H, W, cin, dtype=BF, device=DEV)
    K.conv3x3_rw(K.to_nhwc(dout.to(DEV), BF), packed(spec, w, True), out, True, res=K.to_nhwc(res.to(DEV), BF),
                 mask=K.to_nhwc(mask.to(DEV), BF), mask_mode=L.MASK_RELU, max_workgroups=cap)
    torch.testing.assert_close(K.to_nchw(out, cin).cpu(), ref, rtol=2e-2, atol=2e-2)
    assert_rel_l2(K.to_nchw(out, cin).cpu(), ref, BF)
    # LeakyReLU mask, no residual
    ref2 = F.conv_transpose2d(dout, w, None, 1, 1) * torch.where(mask > 0, 1.0, 0.2)
    K.conv3x3_rw(K.to_nhwc(dout.to(DEV), BF), packed(spec, w, True), out, True, mask=K.to_nhwc(mask.to(DEV), BF),
                 mask_mode=L.MASK_LRELU, max_workgroups=cap)
    torch.testing.assert_close(K.to_nchw(out, cin).cpu(), ref2, rtol=2e-2, atol=2e-2)
    assert_rel_l2(K.to_nchw(out, cin).cpu(), ref2, BF)


@pytest.mark.parametrize("cin,cout,N,H,W,cap,groups", [(64, 64, 4, 24, 32, 5, 2), (64, 128, 6, 16, 16, 0, 1),
                                                       (128, 128, 4, 16, 24, 7, 2), (64, 64, 12, 8, 8, 3, 2),
                                                       # every workgroup's LAST tile is the first of a new statistics group (6 tiles
                                                       # on 3 workgroups: b, b + 3): two flushes with no tile between them
                                                       (64, 64, 2, 24, 16, 4, 2)])
def test_rw_statistics_groups_and_lrelu(cin, cout, N, H, W, cap, groups):
    """the discriminator's residual convs: per-channel sum / sum of squares of the STORED values per BN group
    (code/ops.py:75-77 via batch_norm's batch statistics); bias-gradient mode (sums only) leaves the second row untouched"""
    spec = K.ConvSpec("c3", cin, cout)
    x, w = q(rnd((N, cin, H, W), 8)), q(rnd(spec.weight_shape, 9, -0.1, 0.1))
    ref = F.leaky_relu(F.conv2d(x, w, None, 1, 1), 0.2)
    out = torch.empty(N, H, W, cout, dtype=BF, device=DEV)
    stats = torch.zeros(groups, 2, cout, device=DEV)
    K.conv3x3_rw(K.to_nhwc(x.to(DEV), BF), packed(spec, w, False), out, False, act=L.ACT_LRELU, stats=stats, stats_mode=2,
                 groups=groups, max_workgroups=cap)
    got = K.to_nchw(out, cout).cpu()
    torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2)
    assert_rel_l2(got, ref, BF)
    per = N // groups
    for gi in range(groups):
        r = got[gi * per:(gi + 1) * per].double()  # statistics are taken from the fp32 values before the bf16 store
        torch.testing.assert_close(stats[gi, 0].cpu().double(), r.sum(dim=(0, 2, 3)), rtol=2e-2, atol=0.5)
        torch.testing.assert_close(stats[gi, 1].cpu().double(), (r * r).sum(dim=(0, 2, 3)), rtol=2e-2, atol=0.5)
    s1 = torch.full((1, 2, cout), 7.0, device=DEV)
    K.conv3x3_rw(K.to_nhwc(x.to(DEV), BF), packed(spec, w, False), out, False, stats=s1, stats_mode=1, groups=1, max_workgroups=cap)
    lin = F.conv2d(x, w, None, 1, 1).double()
    torch.testing.assert_close(s1[0, 0].cpu().double() - 7.0, lin.sum(dim=(0, 2, 3)), rtol=2e-2, atol=0.5)
    assert float((s1[0, 1] - 7.0).abs().max()) == 0.0
    # replica blocks (include/tecogan_hip.h, tg_bn_apply): workgroup b adds into block b mod R; the blocks sum to the same totals
    R = 4
    rep = torch.zeros(R, groups, 2, cout, device=DEV)
    K.conv3x3_rw(K.to_nhwc(x.to(DEV), BF), packed(spec, w, False), out, False, act=L.ACT_LRELU, stats=rep, stats_mode=2,
                 groups=groups, max_workgroups=cap, stats_replicas=R)
    torch.testing.assert_close(rep.sum(0), stats, rtol=1e-4, atol=1e-2)
    if cap == 0:  # a full grid (one workgroup per CU, >= R of them along x): every block was used
        assert all(float(rep[r].abs().max()) > 0.0 for r in range(R))


def test_rw_matches_tg_conv_bit_for_bit_on_the_trunk_shape():
    """same operands, same bf16 rounding points: the new kernel and tg_conv may differ only by fp32 summation order"""
    spec = K.ConvSpec("c3", 64, 64)
    x, w = q(rnd((4, 64, 32, 32), 10)), q(rnd(spec.weight_shape, 11, -0.05, 0.05))
    xd, wp = K.to_nhwc(x.to(DEV), BF), packed(spec, w, False)
    a, b = torch.empty(4, 32, 32, 64, dtype=BF, device=DEV), torch.empty(4, 32, 32, 64, dtype=BF, device=DEV)
    K.conv3x3_rw(xd, wp, a, False)
    d = K.make_conv_desc(spec.fwd_geom(), L.TG_BF16, 4, 32, 32, 64, 32, 32, 64)
    K.conv(d, xd, wp, b)
    diff = (a.float() - b.float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(b.float().abs().max())  # at most one bf16 ulp of the largest value
    assert float((diff > 0).float().mean()) < 0.05


def test_rw_unsupported_shapes_are_refused():
    lib = L.load()
    args = [L.TG_BF16, 16, 16, None, None, None, 16, None, 1, 8, 8]
    assert lib.tg_conv3x3_rw(*args, 32, 64, 0, 0, 0, 2, 1, 1, 0, None) == -2      # Cin = 32
    assert lib.tg_conv3x3_rw(*args, 64, 32, 0, 0, 0, 2, 1, 1, 0, None) == -2      # Cout = 32
    assert lib.tg_conv3x3_rw(L.TG_F32, *args[1:], 64, 64, 0, 0, 0, 2, 1, 1, 0, None) == -2   # fp32 runs on tg_conv
    assert lib.tg_conv3x3_rw(*args, 64, 64, 0, L.ACT_SIGMOID, 0, 2, 1, 1, 0, None) == -2
    assert lib.tg_conv3x3_rw(L.TG_BF16, None, 16, None, None, None, 16, None, 1, 8, 8, 64, 64, 0, 0, 0, 2, 1, 1, 0, None) == -1
    st = [L.TG_BF16, 16, 16, None, None, None, 16, 16, 1, 8, 8]                      # with a statistics pointer
    assert lib.tg_conv3x3_rw(*st, 64, 64, 0, 0, 0, 2, 1, 3, 0, None) == -1          # replica count: a power of two

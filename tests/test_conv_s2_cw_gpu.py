"""Parity of the register-weights stride-2 gather kernel (csrc/conv_s2_cw.hip, round 6) against torch fp32 on 16-bit-rounded operands,
and against the per-tile-staging kernel it replaces (csrc/conv4s2_mfma.hip):
  tg_conv4s2_fwd_cw  == F.conv2d(k4, s2, p1) (+ bias, + per-group sum / sum of squares)        /root/reference/code/models.py:90-94
  tg_convt_dgrad_cw  == autograd of F.conv_transpose2d(k3, s2, p1, op1) w.r.t. its input      /root/reference/code/ops.py:45-54
Cases: 64 / 128 reduction channels (one / two phases per tile), one and two output-channel tiles, ragged sizes (partial tiles in both
directions, images smaller than a tile), persistent grids (several tiles per workgroup; a workgroup whose LAST tile opens a new statistics
group), replica blocks, bf16 and fp16.  Also the capped form of the OLD kernel (tg_conv4s2_fwd_capped with 0 < cap < units)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import _lib as L  # noqa: E402
from pytorch_tecogan_amd import kernels as K  # noqa: E402
from parity import assert_rel_l2  # noqa: E402

DEV = "cuda:0"


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32))


def q(t, dt):
    return t.to(dt).float()


FWD_CASES = [  # cin, cout, N, H, W, cap, groups
    (64, 64, 4, 32, 32, 0, 2),
    (64, 64, 12, 128, 128, 96, 1),      # the discriminator's block1 at its cap: 8 tiles per workgroup
    (64, 128, 2, 16, 16, 0, 1),
    (64, 128, 12, 64, 64, 96, 1),       # block2
    (128, 128, 2, 8, 8, 0, 2),
    (128, 128, 12, 32, 32, 96, 1),      # block3: two phases per tile
    (128, 64, 12, 16, 16, 96, 1),       # block4: 8 x 8 outputs, half-empty tile columns
    (128, 64, 3, 16, 16, 2, 1),
    (64, 64, 2, 20, 44, 3, 1),          # ragged: 10 x 22 outputs
    (64, 64, 2, 24, 32, 4, 2),          # 6 tiles on 3 workgroups (b, b + 3): every workgroup's last tile opens group 1
    (128, 128, 5, 6, 6, 2, 1),          # image smaller than a tile, several tiles per workgroup
    (64, 64, 1, 2, 2, 0, 1),
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,N,H,W,cap,G", FWD_CASES)
def test_conv4s2_forward_register_weights(cin, cout, N, H, W, cap, G, dt):
    spec = K.ConvSpec("c4s2", cin, cout)
    x, w, b = q(rnd((N, cin, H, W), 120), dt), q(rnd(spec.weight_shape, 121, -0.1, 0.1), dt), rnd((cout,), 122)
    ref = F.conv2d(x, w, b, 2, 1)
    xd = K.to_nhwc(x.to(DEV), dt)
    rows, Kd, s_row, s_k = spec.fwd_pack()
    wp = K.pack_weights(dt, w.to(DEV).contiguous(), rows, Kd, s_row, s_k, 16, K.slot_table(16, DEV))
    out = torch.full((N, H // 2, W // 2, cout), float("nan"), dtype=dt, device=DEV)
    stats = torch.zeros(G, 2, cout, device=DEV)
    K.conv4s2_fwd_cw(xd, wp, b.to(DEV), out, stats, G, max_workgroups=cap)
    torch.cuda.synchronize()
    got = K.to_nchw(out, cout).cpu()
    torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2)
    assert_rel_l2(got, ref, dt, "conv k4 s2 forward")
    n = N // G
    for g in range(G):   # statistics of the fp32 values before the 16-bit store
        r = ref[g * n:(g + 1) * n].double()
        torch.testing.assert_close(stats[g, 0].cpu().double(), r.sum(dim=(0, 2, 3)), rtol=2e-3, atol=2e-2 * (n * H * W / 4) ** 0.5)
        torch.testing.assert_close(stats[g, 1].cpu().double(), (r * r).sum(dim=(0, 2, 3)), rtol=2e-3, atol=2e-2 * (n * H * W / 4) ** 0.5)
    # the kernel it replaces: same operands, same rounding points - only the fp32 summation order differs
    old = torch.empty_like(out)
    K.conv4s2_fwd(xd, wp, b.to(DEV), old, None)
    torch.testing.assert_close(out.float(), old.float(), rtol=1.6e-2 if dt == torch.bfloat16 else 2e-3, atol=1e-3)
    assert_rel_l2(out.float().cpu(), old.float().cpu(), dt, "new vs old kernel", scale=0.5)
    # replica blocks: workgroup b adds into block b mod R; the blocks sum to the same totals.  No bias, no statistics: a plain launch
    R = 4
    rep = torch.zeros(R, G, 2, cout, device=DEV)
    K.conv4s2_fwd_cw(xd, wp, b.to(DEV), out, rep, G, stats_replicas=R, max_workgroups=cap)
    torch.testing.assert_close(rep.sum(0), stats, rtol=1e-4, atol=1e-2)
    K.conv4s2_fwd_cw(xd, wp, None, out, None, max_workgroups=cap)
    assert_rel_l2(K.to_nchw(out, cout).cpu(), F.conv2d(x, w, None, 2, 1), dt, "without bias")


def test_conv4s2_forward_register_weights_argument_checks():
    lib = L.load()
    x = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(16 * 64 * 64, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(1, 2, 2, 64, dtype=torch.bfloat16, device=DEV)
    st = torch.zeros(2, 64, device=DEV)
    call = lambda dtype=L.TG_BF16, cin=64, cout=64, H=4, W=4, groups=1, reps=1, stats=st: lib.tg_conv4s2_fwd_cw(
        dtype, x.data_ptr(), w.data_ptr(), None, o.data_ptr(), stats.data_ptr() if stats is not None else None, groups, reps, 1, H, W,
        cin, cout, 0, None)
    assert call() == 0
    assert call(dtype=L.TG_F32) == -2 and call(cin=32) == -2 and call(cout=32) == -2 and call(H=3) == -2
    assert call(groups=0) == -1 and call(reps=3) == -1
    assert lib.tg_conv4s2_fwd_cw(L.TG_BF16, None, w.data_ptr(), None, o.data_ptr(), None, 1, 1, 1, 4, 4, 64, 64, 0, None) == -1
    assert lib.tg_convt_dgrad_cw(L.TG_BF16, x.data_ptr(), w.data_ptr(), o.data_ptr(), 1, 4, 4, 32, 64, 0, None) == -2


DG_CASES = [  # cin, cout (of the conv-transpose), N, H, W (of its input), cap
    (64, 64, 2, 16, 16, 0),
    (64, 64, 8, 32, 32, 144),           # conv_trans.0's input-gradient shape (reduced batch)
    (128, 128, 1, 8, 12, 0),
    (128, 128, 4, 64, 64, 144),         # conv_trans.4's: two phases per tile, two channel tiles, 3-4 tiles per workgroup
    (64, 128, 2, 5, 9, 3),
    (128, 64, 1, 1, 1, 0),
    (128, 64, 3, 7, 20, 2),
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,N,H,W,cap", DG_CASES)
def test_convt_input_gradient_register_weights(cin, cout, N, H, W, cap, dt):
    spec = K.ConvSpec("ct", cin, cout)
    x = q(rnd((N, cin, H, W), 130), dt).requires_grad_(True)
    w = q(rnd(spec.weight_shape, 131, -0.1, 0.1), dt)
    dout = q(rnd((N, cout, 2 * H, 2 * W), 132), dt)
    F.conv_transpose2d(x, w, None, 2, 1, 1).backward(dout)
    dd = K.to_nhwc(dout.to(DEV), dt)
    rows, Kd, s_row, s_k = spec.dgrad_pack()
    wb = K.pack_weights(dt, w.to(DEV).contiguous(), rows, Kd, s_row, s_k, 9, K.slot_table(9, DEV))
    dx = torch.full((N, H, W, cin), float("nan"), dtype=dt, device=DEV)
    K.convt_dgrad_cw(dd, wb, dx, max_workgroups=cap)
    torch.cuda.synchronize()
    got = K.to_nchw(dx, cin).cpu()
    scale = float(x.grad.abs().max()) + 1e-6
    torch.testing.assert_close(got, x.grad, rtol=2e-2, atol=2e-2 * max(1.0, scale))
    assert_rel_l2(got, x.grad, dt, "conv-transpose input-gradient")
    old = torch.empty_like(dx)
    K.convt_dgrad(dd, wb, old)
    assert_rel_l2(dx.float().cpu(), old.float().cpu(), dt, "new vs old kernel", scale=0.5)


@pytest.mark.parametrize("cap", [8, 24])
@pytest.mark.parametrize("cin,cout,N,H,W,G,R", [(64, 128, 3, 40, 24, 1, 1), (64, 128, 4, 16, 16, 2, 4), (128, 64, 5, 20, 36, 1, 2),
                                                (64, 64, 12, 64, 64, 2, 4)])
def test_old_conv4s2_forward_capped_walks_its_units(cin, cout, N, H, W, G, R, cap):
    """tg_conv4s2_fwd_capped with 0 < max_workgroups < units (the form the fp32 step and S2_CW=0 run): a workgroup walks several
    (pixel tile, channel tile) units incl. the padding units of ny > 1; results and statistics equal the one-unit-per-workgroup launch"""
    dt = torch.bfloat16
    spec = K.ConvSpec("c4s2", cin, cout)
    x, w = q(rnd((N, cin, H, W), 140), dt), q(rnd(spec.weight_shape, 141, -0.1, 0.1), dt)
    xd = K.to_nhwc(x.to(DEV), dt)
    rows, Kd, s_row, s_k = spec.fwd_pack()
    wp = K.pack_weights(dt, w.to(DEV).contiguous(), rows, Kd, s_row, s_k, 16, K.slot_table(16, DEV))
    a, b = (torch.full((N, H // 2, W // 2, cout), float("nan"), dtype=dt, device=DEV) for _ in range(2))
    sa, sb = torch.zeros(R, G, 2, cout, device=DEV), torch.zeros(R, G, 2, cout, device=DEV)
    K.conv4s2_fwd(xd, wp, None, a, sa, G, stats_replicas=R, max_workgroups=0)
    K.conv4s2_fwd(xd, wp, None, b, sb, G, stats_replicas=R, max_workgroups=cap)
    assert torch.equal(a, b)
    torch.testing.assert_close(sa.sum(0), sb.sum(0), rtol=1e-4, atol=1e-2)
    assert_rel_l2(K.to_nchw(b, cout).cpu(), F.conv2d(x, w, None, 2, 1), dt, "capped launch")

"""1-bit ReLU masks (round 6): tg_convt_fwd_cw writes, beside its output, one bit per element (stored value > 0), and tg_conv3x3_rw reads
that as the mask of the input-gradient of the layer above (TG_MASK_RELU_BITS) instead of the 16-bit activation - the generator's
conv_trans.4 -> conv_trans.6 pair (/root/reference/code/models.py:74-75; autograd of the ReLU between them).  Bit-for-bit: the bits
equal (output > 0), and the input-gradient under the bits equals the input-gradient under the 16-bit mask."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.experiments]   # built, bit-exact, slower in the step: experiments build only (profiles/r06_k_*)

import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import _lib as L  # noqa: E402
from pytorch_tecogan_amd import kernels as K  # noqa: E402

DEV = "cuda:0"


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,N,H,W,cap", [(128, 128, 2, 16, 16, 0), (128, 128, 3, 10, 22, 5), (64, 64, 2, 8, 8, 0), (128, 128, 4, 64, 64, 144)])
def test_conv_transpose_writes_the_bit_mask_of_its_relu_output(cin, cout, N, H, W, cap, dt):
    spec = K.ConvSpec("ct", cin, cout)
    x, w, b = rnd((N, cin, H, W), 1).to(dt).float(), rnd(spec.weight_shape, 2, -0.1, 0.1).to(dt).float(), rnd((cout,), 3, -0.5, 0.5)
    rows, Kd, s_row, s_k = spec.fwd_pack()
    wp = K.pack_weights(dt, w.to(DEV).contiguous(), rows, Kd, s_row, s_k, 9, K.slot_table(9, DEV))
    xd = K.to_nhwc(x.to(DEV), dt)
    out, plain = (torch.full((N, 2 * H, 2 * W, cout), float("nan"), dtype=dt, device=DEV) for _ in range(2))
    bits = torch.full((N, 2 * H, 2 * W, cout // 8), 0xAA, dtype=torch.uint8, device=DEV)
    K.convt_fwd_cw(xd, wp, b.to(DEV), out, L.ACT_RELU, max_workgroups=cap, relu_bits=bits)
    K.convt_fwd_cw(xd, wp, b.to(DEV), plain, L.ACT_RELU, max_workgroups=cap)
    torch.cuda.synchronize()
    assert torch.equal(out, plain)                                    # the output does not change
    want = (out.float() > 0).view(N, 2 * H, 2 * W, cout // 8, 8)       # bit c % 8 of byte c / 8
    weights = (2 ** torch.arange(8, device=DEV)).view(1, 1, 1, 1, 8)
    assert torch.equal(bits.long(), (want.long() * weights).sum(-1))
    assert 0.2 < float(want.float().mean()) < 0.8
    # only with ReLU
    assert L.load().tg_convt_fwd_cw(K.tg_dtype(dt), xd.data_ptr(), wp.data_ptr(), None, out.data_ptr(), N, H, W, cin, cout, L.ACT_LRELU,
                                    bits.data_ptr(), 0, None) == -1


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,N,H,W,cap,bias_sum", [(128, 64, 2, 32, 32, 0, True), (128, 64, 3, 20, 50, 6, True), (64, 64, 2, 24, 16, 3, False),
                                                         (128, 128, 2, 16, 24, 4, False), (128, 64, 8, 128, 128, 144, True)])
def test_input_gradient_under_the_bit_mask_equals_the_16_bit_mask(cin, cout, N, H, W, cap, bias_sum, dt):
    """dgrad of conv(cin -> cout) masked by the ReLU output of the layer below: bits vs the activation itself, bit for bit - and the
    bias-gradient sums (statistics mode 1) that ride in the same launch (c6's input-gradient carries conv_trans.4's bias gradient)"""
    spec = K.ConvSpec("c3", cin, cout)
    w = rnd(spec.weight_shape, 4, -0.1, 0.1).to(dt).float()
    rows, Kd, s_row, s_k = spec.dgrad_pack()
    wb = K.pack_weights(dt, w.to(DEV).contiguous(), rows, Kd, s_row, s_k, 9, K.slot_table(9, DEV))
    dout = K.to_nhwc(rnd((N, cout, H, W), 5).to(DEV), dt)
    act = torch.relu(rnd((N, H, W, cin), 6)).to(dt).to(DEV)           # a ReLU output: about half zeros
    act[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, 1.0, 0.0, 2.0, -0.0, 3.0]).to(dt)   # +0, -0 and a value that rounds to a denormal / zero
    bits = ((act.float() > 0).view(N, H, W, cin // 8, 8).long() * (2 ** torch.arange(8, device=DEV))).sum(-1).to(torch.uint8)
    a, b = (torch.full((N, H, W, cin), float("nan"), dtype=dt, device=DEV) for _ in range(2))
    sa, sb = (torch.zeros(1, 2, cin, device=DEV) if bias_sum else None for _ in range(2))
    K.conv3x3_rw(dout, wb, a, True, mask=act, mask_mode=L.MASK_RELU, stats=sa, stats_mode=1, max_workgroups=cap, cw=False)
    K.conv3x3_rw(dout, wb, b, True, mask=bits, mask_mode=L.MASK_RELU_BITS, stats=sb, stats_mode=1, max_workgroups=cap)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert float(b.float().abs().max()) > 0 and float((b == 0).float().mean()) > 0.3
    if bias_sum:
        torch.testing.assert_close(sa, sb, rtol=1e-5, atol=1e-3)
    lib = L.load()
    args = lambda res, bias, act_: (K.tg_dtype(dt), dout.data_ptr(), wb.data_ptr(), bias, res, bits.data_ptr(), b.data_ptr(), None, N, H, W, cout,
                                    cin, 1, act_, L.MASK_RELU_BITS, 1, 1, 1, 0, None)
    assert lib.tg_conv3x3_rw(*args(a.data_ptr(), None, L.ACT_NONE)) == -2 and lib.tg_conv3x3_rw(*args(None, None, L.ACT_RELU)) == -2
    torch.cuda.synchronize()

"""The discriminator's fused tail (csrc/d_tail.hip: tg_d_tail_fwd / tg_d_tail_bwd, round 6) against the separate launches it replaces -
tg_bn_apply, tg_conv, tg_bn_apply, tg_fc_head_fwd [, tg_dlogit_real]; tg_fc_head_bwd, 2 x (tg_bn_bwd_reduce, tg_bn_bwd_apply), tg_conv -
which are gated against the reference's fixtures and the oracle (tests/test_step_gpu.py, whose step tests run on the fused tail by
default).  Same engine, same weights, same inputs, TECOGAN_D_TAIL = 0 / 1: every tensor another launch reads (n4 z5 n5 prob, d z5, d z4),
the BatchNorm running statistics / num_batches_tracked / saved statistics, and EVERY parameter gradient of the discriminator.
/root/reference/code/models.py:119-123,137-146, code/ops.py:75-77, autograd of code/train.py:304-307."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(1, os.path.join(ROOT, "code"))
import models  # noqa: E402
import tecogan_oracle as orc  # noqa: E402
from pytorch_tecogan_amd import kernels as K  # noqa: E402
from parity import rel_l2  # noqa: E402

DEV = "cuda:0"


def setup_engine(dtype, N, H, monkeypatch, seed=3):
    monkeypatch.setenv("TECOGAN_D_TAIL", "0")   # the engine runs the SEPARATE launches: the reference side of this comparison
    args = orc.default_args(discrim_resblocks=1, crop_size=H // 4)
    args.tg_dtype, args.tg_fc_auto = dtype, True
    D = models.discriminator(args)
    D.load_state_dict(orc.init_params(orc.discriminator_param_shapes(1, 128, 3 * (H // 32) ** 2), seed), strict=False)
    D = D.cuda()
    eng = D.engine()
    assert eng.tail is False
    eng.alloc(2 * N, H)
    rng = np.random.default_rng(seed + 1)
    x = torch.from_numpy(rng.random((2 * N, 27, H, H), dtype=np.float32)).cuda()
    eng.act["in"].copy_(K.to_nhwc(x, eng.dt))
    eng.flat.g.zero_()
    eng.arena.zero()
    # non-trivial BatchNorm parameters for the two layers of the tail (defaults: gamma 1, beta 0)
    for k, bn in ((4, eng.blk[4][1]), (5, eng.blk[5][1])):
        c = bn.C
        bn.gamma[:c].copy_(torch.from_numpy(rng.uniform(0.5, 1.5, size=c).astype(np.float32)))
        bn.beta[:c].copy_(torch.from_numpy(rng.uniform(-0.3, 0.3, size=c).astype(np.float32)))
    return D, eng, rng


@pytest.mark.parametrize("dtype,N,H,mode", [("bf16", 12, 128, "half"), ("bf16", 12, 128, "whole"), ("fp16", 10, 256, "half"),
                                            ("fp32", 3, 128, "half"), ("fp32", 4, 128, "whole"), ("bf16", 5, 192, "half")])
def test_fused_tail_equals_the_separate_launches(dtype, N, H, mode, monkeypatch):
    """the engine runs the separate launches (TECOGAN_D_TAIL=0); the fused launches then run on the SAME z4 / statistics / saved tensors
    (the layers above take their BatchNorm statistics through float atomics: two engine runs differ in the last bit before the tail)"""
    D, eng, rng = setup_engine(dtype, N, H, monkeypatch)
    f32 = dtype == "fp32"
    half, groups, n = (0, 1, N) if mode == "half" else (None, 2, 2 * N)
    sl = slice(0, n)
    cfg = torch.zeros(64, device=DEV)
    cfg[6] = 1e-12
    eng.forward(groups=2, update_stats=True, half=half)
    if mode == "half":
        eng.backward(groups=2, half=0, real_seed=(cfg, None))
    else:
        eng.dlogit.copy_(torch.from_numpy(rng.uniform(-0.1, 0.1, size=2 * N).astype(np.float32)).cuda())
        eng.backward(groups=2)
    torch.cuda.synchronize()
    a, g, bn4, bn5 = eng.act, eng.gbuf, eng.blk[4][1], eng.blk[5][1]
    sv = lambda b: b.save if half is None else b.save[half]
    H4 = a["z"][4].shape[1]
    # ---- forward on the same z4 and statistics
    like = lambda t: torch.full_like(t[sl], float("nan"))
    n4, z5, n5, prob = like(a["n"][4]), like(a["z"][5]), like(a["n"][5]), torch.zeros(n, device=DEV)
    mk = lambda b: (torch.zeros(b.Cp, device=DEV), torch.ones(b.Cp, device=DEV), torch.zeros((), dtype=torch.long, device=DEV))
    rs4, rs5 = mk(bn4), mk(bn5)
    save4, save5 = torch.zeros_like(sv(bn4)), torch.zeros_like(sv(bn5))
    ws = K.d_tail_scratch(n, H4, groups, DEV)   # zero; every launch leaves it zero (forward: the ticket; backward: + the sums)
    K.d_tail_fwd(a["z"][4][sl], bn4.stats_slot(half), bn4.R, bn4.gamma, bn4.beta, *rs4, save4, n4, eng.blk[5][0].w, z5, bn5.gamma, bn5.beta,
                 *rs5, save5, n5, eng.fc_w, eng.fc_b, prob, n, H4, 3, groups, ws)
    torch.cuda.synchronize()
    assert float(ws.abs().max()) >= 0.0 and int(ws[:1].view(torch.int32)) == 0   # the ticket is back at zero
    tl = dict(rtol=1e-5, atol=1e-5) if f32 else dict(rtol=1.6e-2, atol=1e-3)   # 16-bit: at most a rounding flip where fp32 sums differ in order
    for name, got, ref in (("n4", n4, a["n"][4][sl]), ("z5", z5, a["z"][5][sl]), ("n5", n5, a["n"][5][sl])):
        assert rel_l2(got.float(), ref.float()) < (1e-6 if f32 else 3e-4), (name, rel_l2(got.float(), ref.float()))
        torch.testing.assert_close(got.float(), ref.float(), **tl)
    torch.testing.assert_close(prob, eng.prob[sl], rtol=1e-5 if f32 else 1e-3, atol=1e-6 if f32 else 2e-4)
    torch.testing.assert_close(save4, sv(bn4), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(save5[..., :3], sv(bn5)[..., :3], rtol=1e-5 if f32 else 2e-3, atol=1e-6 if f32 else 1e-3)
    for (rm, rv, nbt), b in ((rs4, bn4), (rs5, bn5)):
        torch.testing.assert_close(rm[:b.C], b.rm[:b.C], rtol=1e-5 if f32 else 2e-3, atol=1e-6 if f32 else 1e-4)
        torch.testing.assert_close(rv[:b.C], b.rv[:b.C], rtol=1e-5 if f32 else 2e-3, atol=1e-6 if f32 else 1e-4)
        assert int(nbt) == int(b.nbt) == groups
    # ---- backward on the same saved tensors: d z5, d z4, the loss seed, the six parameter gradients
    dz5, dn4, dz4 = like(g["dz"][5]), like(g["dz"][4]), like(g["dz"][4])
    gfw, gfb = torch.zeros_like(eng.g_fc_w), torch.zeros(32, device=DEV)
    dg5, db5, dg4, db4 = (torch.zeros(32, device=DEV), torch.zeros(32, device=DEV), torch.zeros(64, device=DEV), torch.zeros(64, device=DEV))
    dl = torch.zeros(n, device=DEV) if mode == "half" else eng.dlogit.clone()
    K.d_tail_bwd(dl, eng.prob[sl], cfg if mode == "half" else None, None, mode == "half", a["n"][5][sl], a["z"][5][sl], sv(bn5), bn5.gamma,
                 eng.fc_w, eng.blk[5][0].w, a["n"][4][sl], a["z"][4][sl], sv(bn4), bn4.gamma, dz5, dn4, dz4, gfw, gfb, dg5, db5, dg4, db4,
                 n, H4, 3, groups, ws)
    torch.cuda.synchronize()
    assert int(ws[:1].view(torch.int32)) == 0 and float(ws[32:32 + groups * 128].abs().max()) == 0.0
    if mode == "half":   # the real half's seed: -1/N * p (1 - p) / (p + eps), and what the engine's tg_dlogit_real left
        p_ = eng.prob[sl].double()
        torch.testing.assert_close(dl.double(), -(1.0 / n) * p_ * (1 - p_) / (p_ + 1e-12), rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(dl, eng.dlogit[sl], rtol=1e-6, atol=1e-9)
    for name, got, ref in (("dz5", dz5, g["dz"][5][sl]), ("dz4", dz4, g["dz"][4][sl])):
        e = rel_l2(got.float(), ref.float())
        assert e < (1e-5 if f32 else 6e-3), (name, e)
    view = lambda nm: eng.flat.view(eng.flat.g, nm)
    for name, got, ref in (("fc.weight", gfw, view("fc.weight")), ("fc.bias", gfb[:1], view("fc.bias")),
                           ("block5.1.weight", dg5[:3], view("block5.1.weight")), ("block5.1.bias", db5[:3], view("block5.1.bias")),
                           ("block4.1.weight", dg4, view("block4.1.weight")), ("block4.1.bias", db4, view("block4.1.bias"))):
        e = rel_l2(got.float().flatten(), ref.float().flatten())
        assert e < (1e-5 if f32 else 6e-3), (name, e)


def test_engine_runs_the_fused_tail_by_default(monkeypatch):
    """routing: with default knobs the discriminator engine takes the fused launches for both halves, and falls back beyond their shape"""
    monkeypatch.delenv("TECOGAN_D_TAIL", raising=False)
    args = orc.default_args(discrim_resblocks=1)
    args.tg_dtype = "bf16"
    D = models.discriminator(args).cuda()
    eng = D.engine()
    eng.alloc(24, 128)
    assert eng.tail and eng.tail_fused(12) and not eng.tail_fused(100)
    eng.act["in"].normal_()
    eng.flat.g.zero_()
    eng.arena.zero()
    cfg = torch.zeros(64, device=DEV)
    cfg[6] = 1e-12
    for h in (0, 1):
        eng.forward(update_stats=True, half=h)
    eng.dlogit.fill_(0.01)
    eng.backward(groups=2, half=0, real_seed=(cfg, None))
    eng.backward(groups=2, half=1)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(eng.prob).all()) and bool(torch.isfinite(eng.flat.g).all()) and float(eng.flat.g.abs().max()) > 0
    assert int(eng.blk[5][1].nbt) == 2 and int(eng.blk[4][1].nbt) == 2


def test_fused_tail_argument_checks():
    from pytorch_tecogan_amd import _lib as L
    assert K.d_tail_ok(12, 8, 64, 3) and K.d_tail_ok(10, 16, 64, 3)
    assert not K.d_tail_ok(24, 24, 64, 3) and not K.d_tail_ok(12, 8, 64, 5) and not K.d_tail_ok(300, 2, 64, 3)
    z = torch.zeros(12, 8, 8, 64, dtype=torch.bfloat16, device=DEV)
    z5 = torch.zeros(12, 4, 4, 32, dtype=torch.bfloat16, device=DEV)
    f = torch.zeros(4096, device=DEV)
    lib = L.load()
    args = [z.data_ptr(), f.data_ptr(), 1, f.data_ptr(), f.data_ptr(), None, None, None, f.data_ptr(), z.data_ptr(), f.data_ptr(),
            z5.data_ptr(), f.data_ptr(), f.data_ptr(), None, None, None, f.data_ptr(), z5.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr()]
    ws = K.d_tail_scratch(12, 8, 1, DEV)
    assert ws.numel() == lib.tg_d_tail_scratch_floats(12, 8, 1) == 32 + 128 + 12 * 16 * 4 and lib.tg_d_tail_scratch_floats(0, 8, 1) == -1
    assert lib.tg_d_tail_fwd(L.TG_BF16, *args, 12, 8, 64, 3, 32, 1, 1e-3, 0.1, ws.data_ptr(), None) == 0
    assert lib.tg_d_tail_fwd(L.TG_BF16, *args, 12, 8, 64, 3, 32, 5, 1e-3, 0.1, ws.data_ptr(), None) == -1      # N % groups
    assert lib.tg_d_tail_fwd(L.TG_BF16, *args, 12, 8, 64, 5, 32, 1, 1e-3, 0.1, ws.data_ptr(), None) == -2      # C5 > 4
    assert lib.tg_d_tail_fwd(L.TG_BF16, *args, 12, 80, 64, 3, 32, 1, 1e-3, 0.1, ws.data_ptr(), None) == -2     # beyond the launches' shape
    assert lib.tg_d_tail_fwd(L.TG_BF16, *args, 12, 8, 64, 3, 32, 1, 1e-3, 0.1, None, None) == -1               # no scratch
    assert lib.tg_d_tail_fwd(L.TG_BF16, None, *args[1:], 12, 8, 64, 3, 32, 1, 1e-3, 0.1, ws.data_ptr(), None) == -1
    torch.cuda.synchronize()

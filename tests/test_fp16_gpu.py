"""fp16 element type with dynamic loss scaling: BASELINE configs[3] (64x64 -> 256x256, seq-16, fp16 autocast; 2 sequences
per GPU of the 8-GPU global batch of 16).  The reference's CUDA path is fp16 autocast + ONE shared GradScaler
(code/train.py:3,9,70,335-342); here TG_F16 is a third element tag of the same kernels and the scaler lives on the device."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(1, os.path.join(ROOT, "code"))
import models  # noqa: E402
import train  # noqa: E402
import tecogan_oracle as orc  # noqa: E402
import pytorch_tecogan_amd.train as hip_train  # noqa: E402
from pytorch_tecogan_amd import _lib as L  # noqa: E402
from pytorch_tecogan_amd import kernels as K  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def synth(B, T, cs, seed):
    rng = np.random.default_rng(seed)
    return (torch.from_numpy(rng.random((B, T, 3, cs, cs), dtype=np.float32)),
            torch.from_numpy(rng.random((B, T, 3, 4 * cs, 4 * cs), dtype=np.float32)))


def build(seed, dtype, **over):
    args = orc.default_args(**over)
    args.tg_dtype = dtype
    gp = orc.init_params(orc.generator_param_shapes(args.num_resblock), seed + 100)
    fc_in = 3 * (args.crop_size * 4 // 32) ** 2 if getattr(args, "tg_extend", False) else 48
    dp = orc.init_params(orc.discriminator_param_shapes(args.discrim_resblocks, args.discrim_channels, fc_in), seed + 200)
    G, D = models.generator(3, args), models.discriminator(args)
    G.load_state_dict(gp)
    D.load_state_dict(dp, strict=False)
    G, D = G.cuda(), D.cuda()
    og = torch.optim.Adam(G.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    od = torch.optim.Adam(D.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    return args, G, D, og, od, gp, dp


@pytest.mark.parametrize("kind,cin,cout,N,H", [("c3", 64, 64, 2, 32), ("c3", 51, 64, 2, 32), ("c3", 128, 64, 1, 64),
                                               ("c4s2", 64, 128, 2, 32), ("ct", 64, 64, 2, 16)])
def test_fp16_conv_forward_dgrad_wgrad_vs_torch(kind, cin, cout, N, H):
    """the three conv flavours on the TG_F16 tag: forward, input-gradient and weight-gradient against torch fp32 on the
    fp16-rounded operands (the MFMA accumulates in fp32, so only the output rounding separates the two)"""
    import torch.nn.functional as F
    from pytorch_tecogan_amd import engine as E
    g = torch.Generator().manual_seed(cin * 7 + cout)
    spec = K.ConvSpec(kind, cin, cout)
    w = (torch.randn(*spec.weight_shape, generator=g) * 0.05).half().float()
    b = (torch.randn(cout, generator=g) * 0.1)
    x = torch.randn(N, cin, H, H, generator=g).half().float()
    flat = E.FlatParams({"w": tuple(w.shape), "b": (cout,)}, torch.device(DEV))
    flat.load({"w": w, "b": b})
    conv = E.Conv(flat, "w", "b", spec, torch.float16, E.Workspace(torch.device(DEV)))
    conv.repack()
    if kind == "c3":
        ref = F.conv2d(x, w, b, padding=1)
    elif kind == "c4s2":
        ref = F.conv2d(x, w, b, stride=2, padding=1)
    else:
        ref = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    xin = K.to_nhwc(x.to(DEV), torch.float16)
    OH = ref.shape[2]
    out = torch.empty(N, OH, OH, K.pad32(cout), dtype=torch.float16, device=DEV)
    conv.fwd(xin, out)
    assert rel(K.to_nchw(out, cout), ref) < 2e-3
    dy = torch.randn(N, cout, OH, OH, generator=g).half().float()
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    if kind == "c3":
        y = F.conv2d(xr, wr, None, padding=1)
    elif kind == "c4s2":
        y = F.conv2d(xr, wr, None, stride=2, padding=1)
    else:
        y = F.conv_transpose2d(xr, wr, None, stride=2, padding=1, output_padding=1)
    y.backward(dy)
    dyn = K.to_nhwc(dy.to(DEV), torch.float16)
    dx = torch.empty(N, H, H, K.pad32(cin), dtype=torch.float16, device=DEV)
    conv.dgrad(dyn, dx)
    assert rel(K.to_nchw(dx, cin), xr.grad) < 2e-3
    flat.g.zero_()
    conv.wgrad(xin, dyn)
    torch.cuda.synchronize()
    assert rel(flat.view(flat.g, "w"), wr.grad) < 2e-3


def test_adam_scaled_skips_on_overflow_and_unscales():
    n = 4096
    g = torch.Generator().manual_seed(0)
    p0, grad = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-3
    hyper = torch.tensor(K.adam_hyper(1e-3, 0.9, 0.999, 1e-8, 1), device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], 1e-3)
    ref.grad = grad.clone()
    opt.step()
    S = 1024.0
    for bad in (False, True):
        p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        gs = (grad * S).to(DEV)
        if bad:
            gs[17] = float("inf")
        sc = torch.tensor([S, 5.0, 0, 0, 1.0 / S, 0, 0, 0], device=DEV)
        K.check_finite(gs, sc[2:3])
        K.check_finite(torch.ones(8, device=DEV), sc[3:4])
        K.adam(p, gs, m, v, hyper, scaler=sc, which=0)
        K.scaler_update(sc, interval=7)
        torch.cuda.synchronize()
        if bad:   # skipped: parameters and moments untouched; update(G) backs off and resets, update(D) counts one good step
            assert torch.equal(p.cpu(), p0) and float(m.abs().max()) == 0.0 and float(v.abs().max()) == 0.0
            assert sc.cpu().tolist()[:5] == [S / 2, 1.0, 0.0, 0.0, 2.0 / S]
        else:     # two good updates: tracker 5 -> 6 -> 7 == interval -> scale doubles, tracker resets
            assert rel(p, ref.detach()) < 1e-6
            assert sc.cpu().tolist()[:5] == [2 * S, 0.0, 0.0, 0.0, 0.5 / S]
    nan = torch.tensor([0.0, float("nan")], device=DEV)
    flag = torch.zeros(1, device=DEV)
    K.check_finite(nan, flag)
    assert float(flag) == 1.0


def test_fp16_step_config4_shape_vs_oracle_extension(monkeypatch):
    """configs[3] per-GPU shard: B = 2 sequences of 16 frames, 64x64 -> 256x256, fp16 + loss scaling (tg_extend for the
    shapes the reference cannot run).  Losses / gen_output against the fp32 oracle extension at bf16-class tolerances, the
    scale state after a clean step, and the gradient buffers unscale to the oracle's gradients."""
    monkeypatch.setenv("TECOGAN_GRAPH", "0")
    hip_train._STEPS.clear()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    over = dict(RNN_N=16, crop_size=64, tg_extend=True, num_resblock=4, discrim_resblocks=1)
    args, G, D, og, od, gp, dp = build(8, "fp16", **over)
    x, y = synth(2, 16, 64, 8)
    out = train.FRVSR_Train(x.cuda(), y.cuda(), args, D, G, 0, 0.0, 0.0, og, od)
    torch.cuda.synchronize()
    st = next(iter(hip_train._STEPS.values()))
    assert st.G.dt == torch.float16 and st.scaler is not None
    oargs = orc.default_args(**over)
    g = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    f = orc.tecogan_forward(g, dp, orc.init_bn_buffers(dp, 1), x, y, oargs, 0)
    got = np.array([float(v) for v in out.update_list])
    exp = np.array([float(v) for v in f["update_list"]])
    np.testing.assert_allclose(got, exp, rtol=3e-2, atol=2e-3)
    assert rel(out.gen_output, f["gen"].detach()) < 5e-3
    state = st.scaler_state()
    assert state["scale"] in (65536.0, 32768.0, 16384.0) and state["growth_tracker"] in (0, 1, 2), state
    if state["scale"] == 65536.0:   # no overflow: the flat gradient buffer holds scale * gradient
        gg = torch.autograd.grad(f["gen_loss"], list(g.values()))
        gvec = torch.cat([p.grad.flatten() for _, p in G.named_parameters()]) / 65536.0
        assert rel(gvec, torch.cat([t.flatten() for t in gg])) < 3e-2
        w = torch.cat([p.detach().flatten() for _, p in G.named_parameters()])
        w0 = torch.cat([gp[k].flatten() for k, _ in G.named_parameters()])
        assert 1e-5 < rel(w, w0) < 2e-2      # the update happened, at Adam's scale


def test_config4_full_depth_shard_fp32_vs_oracle_extension(monkeypatch):
    """configs[3] per-GPU shard at FULL depth (16 generator / 4 discriminator residual blocks, B = 2, T = 16, 64x64 -> 256x256,
    tg_extend): the fp32 HIP step against the oracle extension - every loss scalar, gen_output, and the generator's whole
    gradient vector."""
    monkeypatch.setenv("TECOGAN_GRAPH", "0")
    hip_train._STEPS.clear()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    over = dict(RNN_N=16, crop_size=64, tg_extend=True)
    args, G, D, og, od, gp, dp = build(21, "fp32", **over)
    assert args.num_resblock == 16 and args.discrim_resblocks == 4
    x, y = synth(2, 16, 64, 21)
    out = train.FRVSR_Train(x.cuda(), y.cuda(), args, D, G, 0, 0.0, 0.0, og, od)
    torch.cuda.synchronize()
    g = {k: v.clone().requires_grad_(True) for k, v in gp.items()}
    f = orc.tecogan_forward(g, dp, orc.init_bn_buffers(dp), x, y, orc.default_args(**over), 0)
    got = np.array([float(v) for v in out.update_list])
    exp = np.array([float(v) for v in f["update_list"]])
    np.testing.assert_allclose(got, exp, rtol=1e-3, atol=1e-6)
    assert rel(out.gen_output, f["gen"].detach()) < 1e-4
    assert tuple(out.gen_output.shape) == (2, 16, 3, 256, 256) and tuple(out.target.shape) == (2 * 5, 27, 256, 256)
    gg = torch.autograd.grad(f["gen_loss"], list(g.values()))
    gvec = torch.cat([p.grad.flatten() for _, p in G.named_parameters()])
    assert rel(gvec, torch.cat([t.flatten() for t in gg])) < 1e-3
    hip_train._STEPS.clear()


def test_config4_full_depth_shard_fp16_graph_replay_properties(monkeypatch):
    """the same shard in its benchmarked form (fp16 + dynamic loss scaling, per-lane hipGraphs; `bench.py --config 4`): three
    steps replayed from graphs and three eager steps from the same start - finite losses and weights, a loss-scale state
    the GradScaler rule can produce, and graph == eager on every reported loss."""
    over = dict(RNN_N=16, crop_size=64, tg_extend=True)
    x, y = synth(2, 16, 64, 22)
    x, y = x.cuda(), y.cuda()
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TECOGAN_GRAPH", mode)
        hip_train._STEPS.clear()
        args, G, D, og, od, gp, dp = build(22, "fp16", **over)
        args.tg_loss_scale = 4096.0     # (a start the first steps do not overflow from: both modes take the same updates)
        losses = []
        for s in range(3):
            out = train.FRVSR_Train(x, y, args, D, G, s, 0.0, 0.0, og, od)
            losses.append(np.array([float(v) for v in out.update_list]))
        st = next(iter(hip_train._STEPS.values()))
        assert (st.graphs is not None) == (mode == "1") and st.G.dt == torch.float16 and st.G.nrb == 16 and st.D.nrb == 4
        state = st.scaler_state()
        res[mode] = (losses, state, torch.cat([p.detach().flatten() for p in G.parameters()]).cpu())
        assert all(np.isfinite(l).all() for l in losses) and bool(torch.isfinite(res[mode][2]).all())
        assert state["scale"] in (4096.0, 2048.0, 1024.0, 512.0) and 0 <= state["growth_tracker"] <= 6, state
        assert float(st.scaler[2:4].abs().sum()) == 0.0      # found_inf flags cleared by the step's two update() calls
    assert res["0"][1] == res["1"][1]
    for a, b in zip(res["0"][0], res["1"][0]):
        np.testing.assert_allclose(a, b, rtol=2e-2, atol=2e-3)     # fp16 activations + order-dependent float atomics
    assert rel(res["1"][2], res["0"][2]) < 1e-3
    hip_train._STEPS.clear()


def test_fp16_overflow_skips_both_updates_and_backs_the_scale_off(monkeypatch):
    """a loss scale that overflows fp16 everywhere: both networks keep their weights, the scale is halved twice (the two
    update() calls of a step), and training proceeds once the scale has come down; hipGraph replay reads the new scale."""
    monkeypatch.setenv("TECOGAN_GRAPH", "1")
    hip_train._STEPS.clear()
    args, G, D, og, od, gp, dp = build(9, "fp16", num_resblock=2, discrim_resblocks=1)
    args.tg_loss_scale = 2.0 ** 40
    x, y = synth(1, 10, 32, 9)
    x, y = x.cuda(), y.cuda()
    w0 = torch.cat([p.detach().flatten() for p in G.parameters()]).clone()
    d0 = torch.cat([p.detach().flatten() for p in D.parameters()]).clone()
    out = train.FRVSR_Train(x, y, args, D, G, 0, 0.0, 0.0, og, od)
    torch.cuda.synchronize()
    st = next(iter(hip_train._STEPS.values()))
    assert st.scaler_state() == {"scale": 2.0 ** 38, "growth_tracker": 0}
    assert torch.equal(torch.cat([p.detach().flatten() for p in G.parameters()]), w0)
    assert torch.equal(torch.cat([p.detach().flatten() for p in D.parameters()]), d0)
    assert np.isfinite(float(out.gen_loss)) and np.isfinite(float(out.d_loss))    # the reported losses are unscaled fp32
    scales = []
    for s in range(1, 41):
        out = train.FRVSR_Train(x, y, args, D, G, s, 0.0, 0.0, og, od)
        scales.append(st.scaler_state()["scale"])
    assert scales[-1] < 2.0 ** 38 and scales[-1] == scales[-2] == scales[-3] == scales[-6], scales    # settled
    assert not torch.equal(torch.cat([p.detach().flatten() for p in G.parameters()]), w0)   # and training resumed
    assert bool(torch.isfinite(torch.cat([p.detach().flatten() for p in D.parameters()])).all())
    assert np.isfinite(float(out.gen_loss)) and np.isfinite(float(out.d_loss))
    # Adam's step count is torch's: GradScaler.step() does not call optimizer.step() on overflow (code/train.py:337,341).
    # 41 calls; the skipped ones were counted on the device and are merged into optimizer.state['step'] for the checkpoint
    skipped = st.scaler[5:7].cpu()
    assert float(skipped[0]) >= 1.0 and float(skipped[1]) >= 1.0, skipped
    assert float(og._tg_step) == 41.0 and float(od._tg_step) == 41.0
    hip_train.sync_optimizer_steps(og, od)
    assert float(og.state_dict()["state"][0]["step"]) == 41.0 - float(skipped[0])
    assert float(od.state_dict()["state"][0]["step"]) == 41.0 - float(skipped[1])
    assert float(st.scaler[5:7].abs().sum()) == 0.0      # merged: the device counters start again at zero
    hip_train._STEPS.clear()


def test_adam_scaled_bias_correction_counts_only_the_updates_taken():
    """tg_adam_scaled with `skipped` updates on the device counter: the host says t = 2 + skipped calls, the update must be
    torch.optim.Adam's step 2 (bias corrections 1 - beta^2), and a set found_inf flag leaves everything untouched."""
    rng = np.random.default_rng(3)
    p0 = torch.from_numpy(rng.standard_normal(2051).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((2, 2051)).astype(np.float32))
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], 1e-3, betas=(0.9, 0.999), eps=1e-8)
    pd, m, v = p0.to(DEV), torch.zeros(2051, device=DEV), torch.zeros(2051, device=DEV)
    S = 1024.0
    sc = torch.tensor([S, 0.0, 0.0, 0.0, 1.0 / S, 0.0, 0.0, 0.0], device=DEV)
    hy = lambda t: torch.tensor(K.adam_hyper(1e-3, 0.9, 0.999, 1e-8, t), device=DEV)  # noqa: E731
    pt.grad = g[0]
    opt.step()
    K.adam(pd, (g[0] * S).to(DEV), m, v, hy(1), scaler=sc, which=0)            # call 1: taken
    sc[2] = 1.0
    before = pd.clone()
    for t in (2, 3, 4):                                                          # calls 2-4: overflow, skipped
        K.adam(pd, torch.full((2051,), float("inf"), device=DEV), m, v, hy(t), scaler=sc, which=0)
    assert torch.equal(pd, before)
    sc[2], sc[5] = 0.0, 3.0                                                      # (what three tg_scaler_update calls leave)
    pt.grad = g[1]
    opt.step()
    K.adam(pd, (g[1] * S).to(DEV), m, v, hy(5), scaler=sc, which=0)            # call 5 = torch's step 2
    torch.testing.assert_close(pd.cpu(), pt.detach(), rtol=1e-5, atol=5e-7)
    # and the counter itself: update() of a step whose generator overflowed
    st = torch.tensor([S, 5.0, 1.0, 0.0, 1.0 / S, 3.0, 0.0, 0.0], device=DEV)
    K.scaler_update(st)
    assert st.cpu().tolist() == [S / 2, 1.0, 0.0, 0.0, 2.0 / S, 4.0, 0.0, 0.0]


def test_fp16_generator_inference_matches_fp32_oracle():
    args = orc.default_args()
    args.tg_dtype = "fp16"
    gp = orc.init_params(orc.generator_param_shapes(16), 41)
    G = models.generator(3, args)
    G.load_state_dict(gp)
    G = G.cuda()
    x = torch.from_numpy(np.random.default_rng(41).random((1, 5, 3, 32, 32), dtype=np.float32))
    with torch.no_grad():
        ref = orc.recurrent_generator(gp, x, orc.pseudo_flow(x))
    for graph in (False, True):
        assert rel(G.recurrent(x.cuda(), use_graph=graph), ref) < 3e-3


def test_main_py_fp16_checkpoint_carries_the_loss_scaler(tmp_path, monkeypatch):
    """--tg_dtype fp16 through main.py: generator.pt gains the extra key tg_scaler (GradScaler.state_dict() fields) and a
    resumed run starts from it"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tg_main_fp16", os.path.join(ROOT, "main.py"))
    tg_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tg_main)
    monkeypatch.chdir(tmp_path)
    hip_train._STEPS.clear()
    common = ["--synthetic", "4", "--max_epochs", "1", "--tg_dtype", "fp16", "--num_resblock", "2", "--discrim_resblocks", "1"]
    tg_main.main(common)
    ck = torch.load(tmp_path / "generator.pt")
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "tg_scaler"}
    assert ck["tg_scaler"]["scale"] in (65536.0, 32768.0, 16384.0) and ck["tg_scaler"]["growth_tracker"] in (0, 1, 2)
    ck["tg_scaler"] = {"scale": 1024.0, "growth_tracker": 7}
    torch.save(ck, tmp_path / "generator.pt")
    hip_train._STEPS.clear()
    tg_main.main(common + ["--pre_trained_model", "true", "--g_checkpoint", str(tmp_path / "generator.pt"), "--d_checkpoint",
                           str(tmp_path / "discrim.pt")])
    ck2 = torch.load(tmp_path / "generator.pt")
    assert ck2["tg_scaler"] == {"scale": 1024.0, "growth_tracker": 9}      # one step = two update() calls, no overflow at 1024
    hip_train._STEPS.clear()

"""Opt-in FNet training (VERDICT r2 missing 1; SURVEY.md 8 a3/f4, parity unpinned): the estimator the reference defines
(code/models.py:22-50) and whose optimiser it leaves commented out (main.py:231,244-245,249; code/train.py:343-346) is
trained on the LR warp loss of code/train.py:78-84,247-249 with its own output as the sampling grid.  Kernels against torch
autograd, the step against the oracle's statement of the same option (oracle.tecogan_step(..., opt_f=...))."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

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


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return torch.from_numpy(np.random.default_rng(seed).uniform(lo, hi, size=shape).astype(np.float32))


def test_warp_grid_gradient_vs_torch_grid_sample_backward():
    """tg_warp_grid_grad: loss sum and d(loss)/d(grid) of mean-over-(N,C,H)-of-sum-over-W (ref - grid_sample(img, grid))^2,
    the grid being the (2,h,w) -> (h,w,2) reinterpretation of an [N,2,h,w] block; grid values straddle the image border."""
    N, C_, h = 5, 3, 16
    img, ref = rnd((N + 1, C_, h, h), 1, 0, 1), rnd((N + 1, C_, h, h), 2, 0, 1)
    fx = rnd((N + 1, 2, h, h), 3, -1.3, 1.3)
    fxt = fx.clone().requires_grad_(True)
    grid = fxt[:N].reshape(N, h, h, 2)
    v = F.grid_sample(img[:N], grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    loss = torch.mean(torch.sum(torch.square(ref[1:] - v), dim=[3]))
    loss.backward()
    imgd, refd, fxd = img.to(DEV), ref.to(DEV), fx.to(DEV)
    dfx = torch.zeros_like(fxd)
    acc = torch.zeros(1, device=DEV)
    off3 = torch.tensor([n * C_ * h * h for n in range(N)], dtype=torch.int64, device=DEV)
    off3r = torch.tensor([(n + 1) * C_ * h * h for n in range(N)], dtype=torch.int64, device=DEV)
    off2 = torch.tensor([n * 2 * h * h for n in range(N)], dtype=torch.int64, device=DEV)
    coef = 1.0 / (N * C_ * h)
    K.warp_grid_grad(imgd, off3, fxd, off2, refd, off3r, dfx, off2, N, C_, h, h, h, h, coef, loss_acc=acc)
    torch.cuda.synchronize()
    assert abs(float(acc) * coef - float(loss)) < 1e-5 * float(loss)
    assert float(dfx[N].abs().max()) == 0.0          # the block behind the last one is not written
    assert rel(dfx[:N].cpu(), fxt.grad[:N]) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fnet_backward_pieces_vs_torch(dt):
    """tg_up2_bilinear_bwd (with the LeakyReLU' of the up-sampled tensor), tg_maxpool2_bwd with the LeakyReLU mask,
    tg_tanh24_bwd: each against autograd of the module ops f_net uses (code/models.py:6-19,49-50)."""
    q = lambda t: t.to(dt).float()  # noqa: E731
    N, H, W, C_ = 2, 6, 4, 64
    pre = q(rnd((N, C_, H, W), 5))
    pt = pre.clone().requires_grad_(True)
    b = F.leaky_relu(pt, 0.2)
    up = F.interpolate(b, scale_factor=2, mode="bilinear", align_corners=False)
    dup = q(rnd((N, C_, 2 * H, 2 * W), 6))
    up.backward(dup)
    bd, dupd = K.to_nhwc(F.leaky_relu(pre, 0.2).to(DEV), dt), K.to_nhwc(dup.to(DEV), dt)
    out = torch.empty_like(bd)
    K.up2_bilinear_bwd(dupd, out, lrelu_mask=bd)
    assert rel(K.to_nchw(out, C_).cpu(), pt.grad) < (1e-6 if dt == torch.float32 else 6e-3)
    # max-pool + LeakyReLU
    pt2 = pre.clone().requires_grad_(True)
    b2 = F.leaky_relu(pt2, 0.2)
    pool = F.max_pool2d(b2, 2)
    dpool = q(rnd((N, C_, H // 2, W // 2), 7))
    pool.backward(dpool)
    out2 = torch.empty_like(bd)
    K.maxpool2_bwd(bd, K.to_nhwc(dpool.to(DEV), dt), out2, relu_mask=2)
    assert rel(K.to_nchw(out2, C_).cpu(), pt2.grad) < (1e-6 if dt == torch.float32 else 6e-3)
    # 24 tanh
    p3 = rnd((N, 2, H, W), 8, -2, 2).requires_grad_(True)
    o3 = torch.tanh(p3) * 24.0
    d3 = rnd((N, 2, H, W), 9)
    o3.backward(d3)
    dpre = torch.empty(N, H, W, 32, dtype=dt, device=DEV)
    K.tanh24_bwd(d3.to(DEV), o3.detach().to(DEV), dpre)
    got = K.to_nchw(dpre, 32).cpu()
    assert rel(got[:, :2], p3.grad) < (1e-5 if dt == torch.float32 else 6e-3) and float(got[:, 2:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [(3, 32), (32, 32), (32, 2)])
def test_wgrad_32x32_channel_blocks(cin, cout, dt):
    """the 32 x 32 channel-block configuration of tg_wgrad (f_net's padded 3 -> 32, 32 -> 32, 32 -> 2 layers)"""
    q = lambda t: t.to(dt).float()  # noqa: E731
    spec = K.ConvSpec("c3", cin, cout)
    N, H, W = 3, 16, 16
    x, d = q(rnd((N, cin, H, W), 11)), q(rnd((N, cout, H, W), 12))
    w = torch.zeros(spec.weight_shape, requires_grad=True)
    bb = torch.zeros(cout, requires_grad=True)
    F.conv2d(x, w, bb, 1, 1).backward(d)
    X, Y = K.to_nhwc(x.to(DEV), dt), K.to_nhwc(d.to(DEV), dt)
    x_is_in, S, taps, ca, cb, s_a, s_b = spec.wgrad_info()
    nsplit, tpw = K.wgrad_plan(N, H, W, S, 9, X.shape[3], Y.shape[3])
    desc = K.make_wgrad_desc(K.tg_dtype(dt), N, H, W, X.shape[3], H, W, Y.shape[3], S, taps, nsplit, tpw, y_sum=True)
    stride = 9 * X.shape[3] * Y.shape[3] + Y.shape[3]
    slab = torch.full((nsplit * stride,), float("nan"), device=DEV)
    K.wgrad(desc, X, Y, slab)
    g, gb = torch.zeros(spec.weight_shape, device=DEV), torch.zeros(32, device=DEV)
    K.wgrad_finalize(slab, nsplit, 9, X.shape[3], Y.shape[3], ca, cb, g, s_a, s_b, K.slot_table(9, DEV), True, gb)
    assert rel(g.cpu(), w.grad) < (1e-5 if dt == torch.float32 else 2e-3)
    assert rel(gb[:cout].cpu(), bb.grad) < 1e-4


def _build(seed, dtype):
    args = orc.default_args(num_resblock=2, discrim_resblocks=1)
    args.tg_dtype = dtype
    gp = orc.init_params(orc.generator_param_shapes(2), seed + 1)
    dp = orc.init_params(orc.discriminator_param_shapes(1, 128), seed + 2)
    fp = orc.init_params(orc.fnet_param_shapes(), seed + 3)
    G, D, Fn = models.generator(3, args), models.discriminator(args), models.f_net(args)
    G.load_state_dict(gp)
    D.load_state_dict(dp, strict=False)
    Fn.load_state_dict(fp)
    G, D, Fn = G.cuda(), D.cuda(), Fn.cuda()
    mk = lambda m: torch.optim.Adam(m.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)  # noqa: E731
    return args, G, D, Fn, mk(G), mk(D), mk(Fn), gp, dp, fp


@pytest.mark.parametrize("graph", ["0", "1"])
def test_step_with_fnet_training_vs_oracle(graph, monkeypatch):
    """fp32, B = 2: losses (l2_warp_loss is now the estimator's), every f_net weight gradient, the estimator's post-Adam
    weights after two steps, and the generator / discriminator left exactly as in the flow-only option."""
    monkeypatch.setenv("TECOGAN_GRAPH", graph)
    hip_train._STEPS.clear()
    args, G, D, Fn, og, od, of, gp, dp, fp = _build(50, "fp32")
    args.tg_fnet, args.tg_fnet_train, args.tg_fnet_optimizer = Fn, True, of
    rng = np.random.default_rng(50)
    x = torch.from_numpy(rng.random((2, 10, 3, 32, 32), dtype=np.float32))
    y = torch.from_numpy(rng.random((2, 10, 3, 128, 128), dtype=np.float32))
    oargs = orc.default_args(num_resblock=2, discrim_resblocks=1)
    oargs.tg_fnet_params, oargs.tg_fnet_train = fp, True
    bufs = orc.init_bn_buffers(dp, 1)
    o_g, o_d, o_f = (orc.AdamState(p, oargs.learning_rate, oargs.beta, 0.999, oargs.adameps) for p in (gp, dp, fp))
    torch.set_num_threads(max(1, os.cpu_count() // 2))
    for s in range(2):
        out = train.FRVSR_Train(x.cuda(), y.cuda(), args, D, G, s, 0.0, 0.0, og, od)
        torch.cuda.synchronize()
        net, gg, dg, f = orc.tecogan_step(gp, dp, bufs, o_g, o_d, x, y, oargs, s, return_grads=True, opt_f=o_f)
        got = np.array([float(v) for v in out.update_list])
        exp = np.array([float(v) for v in net.update_list])
        np.testing.assert_allclose(got, exp, rtol=2e-3, atol=1e-5)
        assert out.update_list_name[6] == "l2_warp_loss" and abs(got[6] - float(f["warp_loss"])) < 1e-4 * float(f["warp_loss"])
        if s == 0:   # (step 1 of the graph path replays: gradients of step 0 are compared on the first, eager, step)
            hg = {k: p.grad.detach().cpu() for k, p in Fn.named_parameters()}
            worst = max((rel(hg[k], f["fnet_grads"][k]), k) for k in hg if float(f["fnet_grads"][k].norm()) > 1e-7)
            assert worst[0] < 2e-3, worst
            gv = torch.cat([hg[k].flatten() for k in hg])
            assert rel(gv, torch.cat([f["fnet_grads"][k].flatten() for k in hg])) < 5e-4
    st = next(iter(hip_train._STEPS.values()))
    assert (st.graphs is not None) == (graph == "1")
    for k, p in Fn.named_parameters():
        assert rel(p.detach().cpu(), fp[k]) < 2e-5, k          # two Adam steps of the estimator
    assert float(of.state_dict()["state"][0]["step"]) == 2.0
    assert rel(torch.cat([p.detach().flatten() for p in G.parameters()]).cpu(), torch.cat([gp[k].flatten() for k, _ in G.named_parameters()])) < 1e-4
    hip_train._STEPS.clear()


def test_fnet_training_bf16_follows_the_oracle_trajectory_and_moves_only_when_asked(monkeypatch):
    """bf16 (the benchmarked element type) under graph replay, estimator lr 1e-3, a fixed batch, 6 steps: the warp loss follows
    the oracle's trajectory of the same option step by step (on random frames it RISES - 7.49, 10.2, 10.5 ...: the samples leave
    the image, there is nothing to learn - which a wrong gradient would not reproduce); without tg_fnet_train the estimator's
    weights stay bit-identical (default = reference behaviour) and the reported warp loss is the raw-frame one."""
    monkeypatch.setenv("TECOGAN_GRAPH", "1")
    rng = np.random.default_rng(7)
    xc = torch.from_numpy(rng.random((2, 10, 3, 32, 32), dtype=np.float32))
    x = xc.cuda()
    y = torch.from_numpy(rng.random((2, 10, 3, 128, 128), dtype=np.float32)).cuda()
    # oracle: the estimator alone (the generator / discriminator never feed back into it: every input is detached)
    fp = orc.init_params(orc.fnet_param_shapes(), 63)
    o_f = orc.AdamState(fp, 1e-3)
    prev, nxt = xc[:, :-1].reshape(18, 3, 32, 32), xc[:, 1:].reshape(18, 3, 32, 32)
    torch.set_num_threads(max(1, os.cpu_count() // 2))
    exp = []
    for s in range(6):
        for t in fp.values():
            t.requires_grad_(True)
        loss = torch.mean(torch.sum(torch.square(nxt - orc.warp(prev, orc.as_grid(orc.fnet_forward(fp, prev)))), dim=[3]))
        grads = torch.autograd.grad(loss, list(fp.values()))
        for t in fp.values():
            t.requires_grad_(False)
        with torch.no_grad():
            o_f.step(fp, dict(zip(fp.keys(), grads)))
        exp.append(float(loss))
    for trainf in (False, True):
        hip_train._STEPS.clear()
        args, G, D, Fn, og, od, of, gp, dp, fp0 = _build(60, "bf16")
        args.tg_fnet, args.tg_fnet_train, args.tg_fnet_optimizer = Fn, trainf, of
        for grp in of.param_groups:
            grp["lr"] = 1e-3
        w0 = torch.cat([p.detach().flatten() for p in Fn.parameters()]).clone()
        losses = []
        for s in range(6):
            out = train.FRVSR_Train(x, y, args, D, G, s, 0.0, 0.0, og, od)
            losses.append(float(out.update_list[6]))
        w1 = torch.cat([p.detach().flatten() for p in Fn.parameters()])
        if trainf:
            assert not torch.equal(w0, w1)
            np.testing.assert_allclose(losses, exp, rtol=2e-2)
        else:
            assert torch.equal(w0, w1) and max(losses) - min(losses) < 1e-6 * losses[0]
    hip_train._STEPS.clear()


def test_main_py_trains_the_estimator_and_resumes_from_fnet_pt(tmp_path, monkeypatch):
    """main.py --tg_fnet true --tg_fnet_train true: the third optimiser / scheduler / checkpoint the reference leaves commented out
    (main.py:231,244-245,249,259-261): fnet.pt with the reference module's keys, resumed with --f_checkpoint."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tg_main_f", os.path.join(ROOT, "main.py"))
    tg_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tg_main)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("TECOGAN_GRAPH", "1")
    hip_train._STEPS.clear()
    common = ["--synthetic", "8", "--max_epochs", "1", "--tg_dtype", "bf16", "--num_resblock", "2", "--discrim_resblocks", "1",
              "--tg_fnet", "true", "--tg_fnet_train", "true"]
    # main.py draws its initial weights and synthetic frames from the global generators: seeded here, because an unlucky estimator
    # can send every sample of the warp outside the image - zero gradient, Adam update 0 / (0 + eps), weights legitimately unchanged
    torch.manual_seed(int(os.environ.get('TG_TEST_SEED', '1234')))
    np.random.seed(1234)
    tg_main.main(common)
    f_ck = torch.load(tmp_path / "fnet.pt")
    assert set(f_ck) == {"model_state_dict", "optimizer_state_dict"}
    assert list(f_ck["model_state_dict"].keys()) == list(orc.fnet_param_shapes().keys())
    assert len(f_ck["optimizer_state_dict"]["state"]) == 36 and float(f_ck["optimizer_state_dict"]["state"][0]["step"]) == 2.0
    w1 = f_ck["model_state_dict"]["up1.2.weight"].clone()
    hip_train._STEPS.clear()
    torch.manual_seed(4321)
    np.random.seed(4321)
    tg_main.main(["--synthetic", "4", "--max_epochs", "1", "--tg_dtype", "bf16", "--num_resblock", "2", "--discrim_resblocks", "1",
                  "--tg_fnet", "true", "--tg_fnet_train", "true", "--pre_trained_model", "true", "--g_checkpoint",
                  str(tmp_path / "generator.pt"), "--d_checkpoint", str(tmp_path / "discrim.pt"), "--f_checkpoint",
                  str(tmp_path / "fnet.pt")])
    f2 = torch.load(tmp_path / "fnet.pt")
    assert float(f2["optimizer_state_dict"]["state"][0]["step"]) == 3.0
    changed = [k for k, v in f2["model_state_dict"].items() if not torch.equal(v, f_ck["model_state_dict"][k])]
    assert "up1.2.weight" in changed and len(changed) == len(f2["model_state_dict"]), changed
    hip_train._STEPS.clear()

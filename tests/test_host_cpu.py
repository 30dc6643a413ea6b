"""CPU-only tests (no GPU, no compute calls into the HIP library): the C-ABI library loads and exports every symbol the
header declares; the host-side geometry / packing / index tables that drive the kernels are correct (checked by emulating the
kernel contracts with plain torch on CPU against torch's own convolutions and the oracle); module surface and error
behaviour match the reference."""
import argparse
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import _lib as L  # noqa: E402
from pytorch_tecogan_amd import engine as E  # noqa: E402
from pytorch_tecogan_amd import kernels as K  # noqa: E402
from pytorch_tecogan_amd import models as M  # noqa: E402
from pytorch_tecogan_amd import step as S  # noqa: E402
import tecogan_oracle as orc  # noqa: E402


# ------------------------------------------------------------------------------------------------ C ABI
def test_capi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tecogan_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    # the rejected variants are declared under `#ifdef TG_EXPERIMENTS` and exported by the experiments build only
    exp_blocks = re.findall(r"#ifdef TG_EXPERIMENTS(.*?)#endif", hdr, flags=re.S)
    experimental = set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", "".join(exp_blocks)))
    declared = set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", re.sub(r"#ifdef TG_EXPERIMENTS.*?#endif", "", hdr, flags=re.S)))
    assert len(declared) >= 25
    assert experimental == set(L._PROTOS_EXPERIMENTS), experimental ^ set(L._PROTOS_EXPERIMENTS)
    lib = L.load()  # raises if the .so is missing: there is no fallback
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tecogan_hip.h but not exported"
    assert declared == set(L.EXPORTED), declared ^ set(L.EXPORTED)
    # the default library exports exactly the declared set: nothing of the experiments, nothing undeclared
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0 and not lib.tg_has_experiments():
        exported = {ln.split()[-1] for ln in nm.stdout.splitlines() if " T tg_" in ln}
        assert exported == declared, exported ^ declared
    assert lib.tg_abi_version() == L.ABI_VERSION == 4
    assert lib.tg_error_string(-2) == b"unsupported shape"


def test_capi_argument_validation_without_gpu():
    """validation happens before any launch, so these status codes can be checked on a CPU-only box"""
    lib = L.load()
    d = K.make_conv_desc(K.ConvSpec("c3", 64, 64).fwd_geom(), L.TG_BF16, 1, 8, 8, 64, 8, 8, 64)
    import ctypes
    assert lib.tg_conv(ctypes.byref(d), None, None, None, None, None, None, None, None) == -1  # TG_E_BADARG
    d2 = K.make_conv_desc(K.ConvSpec("c3", 64, 64).fwd_geom(), L.TG_BF16, 1, 8, 8, 48, 8, 8, 64)
    assert lib.tg_conv(ctypes.byref(d2), 16, 16, None, None, None, 16, None, None) == -3  # channels not multiple of 32
    assert lib.tg_adam(None, None, None, None, 10, None, None) == -1
    assert lib.tg_packed_weight_bytes(L.TG_BF16, 9, 64, 64) == 9 * 64 * 64 * 2


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libtecogan_hip.so")
    with pytest.raises(L.TecoganHipError):
        L.load()


# ------------------------------------------------------------------------------------------------ geometry
def emulate_gather_conv(geom, x, wp_slots, OH, OW):
    """torch-CPU emulation of the tg_conv contract: out[n, cy*OS+ooy, cx*OS+oox] = sum_t in[n, cy*S+dy, cx*S+dx] @ W[slot]."""
    N, Cin, IH, IW = x.shape
    Cout = wp_slots.shape[1]
    out = torch.zeros(N, Cout, OH, OW)
    pad = 4
    xp = F.pad(x, (pad, pad, pad, pad))
    for ooy, oox, taps in geom.classes:
        ohc = (OH - ooy + geom.OS - 1) // geom.OS
        owc = (OW - oox + geom.OS - 1) // geom.OS
        acc = torch.zeros(N, Cout, ohc, owc)
        for dy, dx, slot in taps:
            ys = pad + dy + geom.S * torch.arange(ohc)
            xs = pad + dx + geom.S * torch.arange(owc)
            ok = (ys < IH + 2 * pad) & (ys >= 0)
            patch = xp[:, :, ys.clamp(0, IH + 2 * pad - 1)][:, :, :, xs.clamp(0, IW + 2 * pad - 1)]
            patch = patch * ok.view(1, 1, -1, 1) * ((xs < IW + 2 * pad) & (xs >= 0)).view(1, 1, 1, -1)
            acc += torch.einsum("nchw,oc->nohw", patch, wp_slots[slot])
        out[:, :, ooy::geom.OS, oox::geom.OS] = acc
    return out


def slots_from_pack(w, rows, Kd, s_row, s_k, nslots):
    """what tg_pack_conv_weights reads: packed[slot][row][k] = w.flat[row*s_row + k*s_k + slot]"""
    flat = w.reshape(-1)
    r = torch.arange(rows).view(-1, 1) * s_row
    k = torch.arange(Kd).view(1, -1) * s_k
    return torch.stack([flat[r + k + s] for s in range(nslots)])


@pytest.mark.parametrize("kind,cin,cout,H,W", [("c3", 5, 7, 9, 6), ("c4s2", 6, 4, 8, 12), ("ct", 4, 6, 5, 7)])
def test_conv_geometry_tables_forward_and_dgrad(kind, cin, cout, H, W):
    rng = np.random.default_rng(0)
    spec = K.ConvSpec(kind, cin, cout)
    x = torch.from_numpy(rng.standard_normal((2, cin, H, W)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal(spec.weight_shape).astype(np.float32))
    OH, OW = spec.out_hw(H, W)

    def ref(xx):
        if kind == "c3":
            return F.conv2d(xx, w, None, 1, 1)
        if kind == "c4s2":
            return F.conv2d(xx, w, None, 2, 1)
        return F.conv_transpose2d(xx, w, None, 2, 1, 1)

    got = emulate_gather_conv(spec.fwd_geom(), x, slots_from_pack(w, *spec.fwd_pack(), spec.nslots), OH, OW)
    torch.testing.assert_close(got, ref(x), rtol=1e-4, atol=1e-4)
    dout = torch.from_numpy(rng.standard_normal((2, cout, OH, OW)).astype(np.float32))
    xr = x.clone().requires_grad_(True)
    ref(xr).backward(dout)
    got = emulate_gather_conv(spec.dgrad_geom(), dout, slots_from_pack(w, *spec.dgrad_pack(), spec.nslots), H, W)
    torch.testing.assert_close(got, xr.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kind,cin,cout,H,W", [("c3", 5, 7, 9, 6), ("c4s2", 6, 4, 8, 12), ("ct", 4, 6, 5, 7)])
def test_wgrad_geometry_tables(kind, cin, cout, H, W):
    rng = np.random.default_rng(1)
    spec = K.ConvSpec(kind, cin, cout)
    x = torch.from_numpy(rng.standard_normal((2, cin, H, W)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal(spec.weight_shape).astype(np.float32)).requires_grad_(True)
    OH, OW = spec.out_hw(H, W)
    dout = torch.from_numpy(rng.standard_normal((2, cout, OH, OW)).astype(np.float32))
    out = F.conv2d(x, w, None, 1, 1) if kind == "c3" else (F.conv2d(x, w, None, 2, 1) if kind == "c4s2" else
                                                           F.conv_transpose2d(x, w, None, 2, 1, 1))
    out.backward(dout)
    x_is_in, Sx, taps, ca, cb, s_a, s_b = spec.wgrad_info()
    X, Y = (x, dout) if x_is_in else (dout, x)
    pad = 4
    Xp = F.pad(X, (pad, pad, pad, pad))
    YH, YW = Y.shape[2:]
    grad = torch.zeros(w.numel())
    for t, (dy, dx) in enumerate(taps):  # the tg_wgrad / tg_wgrad_finalize contract
        ys = pad + dy + Sx * torch.arange(YH)
        xs = pad + dx + Sx * torch.arange(YW)
        patch = Xp[:, :, ys][:, :, :, xs]
        dwt = torch.einsum("nahw,nbhw->ab", patch, Y)
        a_idx = torch.arange(ca).view(-1, 1) * s_a
        b_idx = torch.arange(cb).view(1, -1) * s_b
        grad[(a_idx + b_idx + t).reshape(-1)] += dwt.reshape(-1)
    torch.testing.assert_close(grad.view(w.shape), w.grad, rtol=1e-4, atol=1e-4)


def test_bf16_row_permutation_is_a_bijection_with_16_byte_epilogue_vectors():
    rows = []
    for R in range(64):
        tile, r = R >> 4, R & 15
        q, j = r >> 2, r & 3
        rows.append(32 * (tile >> 1) + 8 * q + 4 * (tile & 1) + j)  # row_to_channel<BF16> of csrc/common.h
    assert sorted(rows) == list(range(64))
    for pair in range(2):  # lane q of tile pair u owns channels 32u+8q .. +7
        for q in range(4):
            chans = [rows[(2 * pair) * 16 + 4 * q + j] for j in range(4)] + [rows[(2 * pair + 1) * 16 + 4 * q + j] for j in range(4)]
            assert chans == list(range(32 * pair + 8 * q, 32 * pair + 8 * q + 8))


# ------------------------------------------------------------------------------------------------ step tables
def test_step_tables_reproduce_flow_and_tvel_against_the_oracle():
    B, T, h, Kt = 2, 10, 4, 3
    H = 4 * h
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.random((B, T, 3, h, h), dtype=np.float32))
    t = S.build_tables(B, T, h, Kt)
    xf = x.reshape(-1)
    hh, HH = h * h, H * H
    # pseudo-flow via the plane table == oracle.pseudo_flow
    flow = torch.zeros(B * (T - 1) * 2 * HH)
    for so, do in zip(t["flow_src"], t["flow_dst"]):
        flow[do:do + HH] = orc.up4(xf[so:so + hh].view(1, 1, h, h) * 4.0).reshape(-1)
    ref_flow = orc.pseudo_flow(x)
    assert torch.equal(flow.view_as(ref_flow), ref_flow)
    # T_vel via copy-block + back-plane tables == oracle.t_velocity (including the rows-0..B-1 batch-mixing quirk)
    tv = torch.full((B * 3 * Kt * 2 * HH,), float("nan"))
    for so, do in zip(t["tv_csrc"], t["tv_cdst"]):
        tv[do:do + 2 * HH] = 0.0 if so < 0 else flow[so:so + 2 * HH]
    for so, do in zip(t["tv_bsrc"], t["tv_bdst"]):
        tv[do:do + HH] = (orc.up4(xf[so:so + hh].view(1, 1, h, h) * 4.0) * 2.0 - 1.0).reshape(-1)
    ref_tv = orc.t_velocity(x, ref_flow, 3 * Kt)
    assert torch.equal(tv.view_as(ref_tv), ref_tv)
    # ping-pong: 2T-1 frames, K = 6, VNxt = flip(flow)[:, 1:ts:3] copied raw (no back planes)
    Tp, Kp = 2 * T - 1, (2 * T - 1) // 3
    xp = torch.cat([x, torch.flip(x, dims=[1])[:, 1:]], dim=1)
    tp = S.build_tables(B, Tp, h, Kp, pingpang=True)
    assert tp["tv_bsrc"] == []
    fp = orc.pseudo_flow(xp)
    tvp = torch.full((B * 3 * Kp * 2 * HH,), float("nan"))
    for so, do in zip(tp["tv_csrc"], tp["tv_cdst"]):
        tvp[do:do + 2 * HH] = 0.0 if so < 0 else fp.reshape(-1)[so:so + 2 * HH]
    ref_tvp = orc.t_velocity(xp, fp, 3 * Kp, pingpang=True)
    assert torch.equal(tvp.view_as(ref_tvp), ref_tvp)
    # opt-in extension for RNN_N//3 != 3 (the reference raises there): first B*K*2 planes of the flattened back tensor;
    # identical to the reference rule at K == 3
    assert torch.equal(orc.t_velocity(x, ref_flow, 9, extended=True), ref_tv)
    Te, Ke = 16, 5
    xe = torch.from_numpy(rng.random((B, Te, 3, h, h), dtype=np.float32))
    te = S.build_tables(B, Te, h, Ke)
    fe = orc.pseudo_flow(xe)
    tve = torch.full((B * 3 * Ke * 2 * HH,), float("nan"))
    for so, do in zip(te["tv_csrc"], te["tv_cdst"]):
        tve[do:do + 2 * HH] = 0.0 if so < 0 else fe.reshape(-1)[so:so + 2 * HH]
    for so, do in zip(te["tv_bsrc"], te["tv_bdst"]):
        tve[do:do + HH] = (orc.up4(xe.reshape(-1)[so:so + hh].view(1, 1, h, h) * 4.0) * 2.0 - 1.0).reshape(-1)
    ref_tve = orc.t_velocity(xe, fe, 3 * Ke, extended=True)
    # (ATen's CPU bilinear kernel rounds differently for a 20-channel image than for single planes: last-bit only)
    torch.testing.assert_close(tve.view_as(ref_tve), ref_tve, rtol=0, atol=2e-6)
    with pytest.raises(RuntimeError):
        orc.t_velocity(xe, fe, 3 * Ke)  # reference behaviour: the reshape fails
    # LR-warp tables: image x[b,t], grid block x[b,t+1,0:2]
    for n, (io, go) in enumerate(zip(t["lrw_img"], t["lrw_grid"])):
        b, tt = divmod(n, T - 1)
        assert torch.equal(xf[io:io + 3 * hh].view(3, h, h), x[b, tt])
        assert torch.equal(xf[go:go + 2 * hh].view(2, h, h), x[b, tt + 1, 0:2])


# ------------------------------------------------------------------------------------------------ module surface
def _args(**kw):
    a = dict(num_resblock=16, discrim_resblocks=4, discrim_channels=128, crop_size=32)
    a.update(kw)
    return argparse.Namespace(**a)


def test_module_surface_matches_reference_checkpoint_abi():
    G, D, Fn = M.generator(3, _args()), M.discriminator(_args()), M.f_net()
    assert [(k, tuple(v.shape)) for k, v in G.named_parameters()] == list(orc.generator_param_shapes().items())
    assert [(k, tuple(v.shape)) for k, v in D.named_parameters()] == list(orc.discriminator_param_shapes().items())
    assert [(k, tuple(v.shape)) for k, v in Fn.named_parameters()] == list(orc.fnet_param_shapes().items())
    assert sum(p.numel() for p in G.parameters()) == 1765251
    assert sum(p.numel() for p in D.parameters()) == 3267383
    assert sum(p.numel() for p in Fn.parameters()) == 7054594
    sd = D.state_dict()
    for bn in orc.discriminator_bn_names():
        assert tuple(sd[bn + ".running_mean"].shape) == tuple(sd[bn + ".weight"].shape)
        assert sd[bn + ".num_batches_tracked"].dtype == torch.long
        assert float(sd[bn + ".weight"].min()) == 1.0 and float(sd[bn + ".bias"].abs().max()) == 0.0
    # state dicts produced by the oracle's (reference-order) tables load strictly
    G.load_state_dict(orc.init_params(orc.generator_param_shapes(), 1), strict=True)
    with pytest.raises(ValueError):
        M.generator(3)
    with pytest.raises(ValueError):
        M.discriminator()
    G2 = M.generator(3, _args(num_resblock=2))
    assert len([k for k in G2.state_dict() if k.startswith("resids.")]) == 6
    # default init: U(-1/sqrt(fan_in), +) like torch's Conv2d
    w = dict(G.named_parameters())["resids.0.0.weight"]
    G3 = M.generator(3, _args())
    w = dict(G3.named_parameters())["resids.0.0.weight"]
    assert float(w.abs().max()) <= 1.0 / np.sqrt(64 * 9) + 1e-7 and float(w.abs().max()) > 0.9 / np.sqrt(64 * 9)


def test_no_cpu_fallback():
    G = M.generator(3, _args())
    with pytest.raises(L.TecoganHipError):
        G(torch.zeros(1, 51, 8, 8))
    with pytest.raises(L.TecoganHipError):
        M.f_net()(torch.zeros(1, 3, 32, 32))
    from pytorch_tecogan_amd import train as TR
    a = orc.default_args()
    with pytest.raises(L.TecoganHipError):
        TR.FRVSR_Train(torch.zeros(1, 10, 3, 32, 32), torch.zeros(1, 10, 3, 128, 128), a, None, None, 0, 0.0, 0.0, None, None)


def test_flat_params_layout_and_padding():
    flat = E.FlatParams(E.discriminator_shapes(4, 128), torch.device("cpu"))
    assert flat.total % 32 == 0
    for name, (off, n) in flat.offsets.items():
        assert off % 32 == 0
    assert flat.padded(flat.p, "block5.1.weight").numel() == 32  # 3 real BN channels, kernels read the padded 32
    assert flat.view(flat.p, "fc.weight").shape == (1, 48)
    assert E.discriminator_bn_names(4) == orc.discriminator_bn_names(4)
    assert list(E.generator_shapes(16).items()) == list(orc.generator_param_shapes(16).items())


def test_wgrad_split_rule():
    assert K.wgrad_nsplit(40, 32, 32, 1, blocks=1) == 256
    assert K.wgrad_nsplit(40, 64, 64, 1, blocks=4) == 64
    assert K.wgrad_nsplit(1, 8, 8, 1, blocks=1) == 1  # never more splits than pixel tiles
    assert K.wgrad_blocks(9, 128, 128) == 4 and K.wgrad_blocks(16, 64, 64) == 2 and K.wgrad_blocks(9, 32, 64) == 1


def test_dataloader_semantics(tmp_path):
    """code/dataloader.py quirks: __len__ = number of scenes, 10-frame windows, frame sizes, skip of short scenes."""
    sys.path.insert(1, os.path.join(ROOT, "code"))
    import dataloader as DL
    from PIL import Image
    rng = np.random.default_rng(0)
    for scene, nframes in ((1000, 120), (1001, 50)):
        d = tmp_path / ("scene_%04d" % scene)
        d.mkdir()
        for k in range(nframes):
            Image.fromarray(rng.integers(0, 255, (24, 24, 3), dtype=np.uint8)).save(d / ("col_high_%04d.png" % k))
    a = argparse.Namespace(input_video_dir=str(tmp_path), input_video_pre="scene", str_dir=1000, end_dir=1002,
                           max_frm=119, crop_size=8)
    ds = DL.train_dataset(a)
    assert len(ds) == 1 and len(ds.windows) == 110
    lr, hr = ds[0]
    assert lr.shape == (10, 3, 8, 8) and hr.shape == (10, 3, 32, 32) and lr.dtype == torch.float32
    assert 0.0 <= float(lr.min()) and float(lr.max()) <= 1.0
    with pytest.raises(ValueError):
        DL.train_dataset(argparse.Namespace(input_video_dir="", input_video_pre="scene", str_dir=0, end_dir=1, max_frm=119,
                                            crop_size=8))
    inf = DL.inference_dataset(argparse.Namespace(input_dir_LR=str(tmp_path), input_dir_HR=None, crop_size=8))
    assert len(inf) == 2 and inf[0].shape[1:] == (3, 8, 8)
    # a PALETTE scene is resized as opened (PIL: NEAREST for mode 'P') and converted afterwards, like the reference's
    # torchvision resize of the opened image (code/dataloader.py:88-93 there); the frame cache keeps it unconverted, and "auto" does
    # not hand such a dataset (or one with a differently sized scene in the middle) to the decode-only / GPU-resize path
    d = tmp_path / "pal" / "scene_2000"
    d.mkdir(parents=True)
    for k in range(120):
        Image.fromarray(rng.integers(0, 255, (24, 24), dtype=np.uint8), mode="P").save(d / ("col_high_%04d.png" % k))
    ap = argparse.Namespace(input_video_dir=str(tmp_path / "pal"), input_video_pre="scene", str_dir=2000, end_dir=2000, max_frm=119,
                            crop_size=8)
    dsp = DL.train_dataset(ap, decode_only="auto")
    assert dsp.decode_only is False
    lr_p, _ = dsp[0]
    ref = Image.open(d / "col_high_0001.png").resize((8, 8), Image.BILINEAR).convert("RGB")
    assert torch.equal(lr_p[1], torch.from_numpy(np.asarray(ref, dtype=np.float32) / 255.0).permute(2, 0, 1))
    assert DL.train_dataset(a, decode_only="auto").decode_only is True      # the RGB tree above: one size, RGB


def test_shape_sets_keep_pinned_sets_alive_and_drop_loose_ones():
    """a module forward at another shape must not free the buffers a captured step points into"""
    from pytorch_tecogan_amd.engine import ShapeSets
    s = ShapeSets(keep=2)
    made = []
    mk = lambda tag: (lambda: made.append(tag) or {"tag": tag})  # noqa: E731
    s.pin((40, 32, 32))
    a = s.get((40, 32, 32), mk("train"))
    for i in range(5):
        s.get((1, 16 * (i + 1), 16), mk(f"inf{i}"))
    assert s.get((40, 32, 32), mk("again")) is a and "again" not in made      # pinned: survived five other shapes
    loose = [k for k in s.sets if k not in s.pinned]
    assert len(loose) <= 2 and (1, 80, 16) in s.sets                           # most recent loose sets kept
    s.unpin((40, 32, 32))
    for i in range(3):
        s.get((2, 16 * (i + 1), 16), mk(f"x{i}"))
    assert (40, 32, 32) not in s.sets


def test_save_image_grid_layout(tmp_path):
    """ops.save_image = torchvision.utils.save_image defaults (main.py:287-294): 8 tiles per row, 2 px padding"""
    import numpy as np
    from PIL import Image
    from pytorch_tecogan_amd import ops
    t = torch.zeros(10, 3, 4, 6)
    t[9] = 1.0
    fp = tmp_path / "grid.png"
    ops.save_image(t, str(fp))
    img = np.asarray(Image.open(fp))
    assert img.shape == (2 * (4 + 2) + 2, 8 * (6 + 2) + 2, 3)
    assert img[8:12, 10:16].min() == 255 and img[:8].max() == 0 and img[8:12, :10].max() == 0   # tile 9 = row 1, column 1
    ops.save_image(torch.full((1, 3, 5, 5), 0.5), str(fp))
    assert np.asarray(Image.open(fp)).shape == (5, 5, 3)


@pytest.mark.parametrize("hw,out", [((40, 56), 16), ((37, 23), 32), ((24, 24), 8), ((24, 24), 96), ((180, 320), 32),
                                    ((180, 320), 128), ((32, 32), 32)])
def test_resize_restatement_equals_pil_bilinear_bit_for_bit(hw, out):
    """data ingest, GPU-side resize: the host-built fixed-point coefficient tables and the two-pass 8-bit
    arithmetic reproduce PIL's Image.resize(BILINEAR) - the reference's frame resize - exactly"""
    from PIL import Image
    from pytorch_tecogan_amd import resize as R
    rng = np.random.default_rng(hw[0] * 1000 + out)
    img = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    img[:5, :7] = 255
    img[-3:, -9:] = 0
    exp = np.asarray(Image.fromarray(img).resize((out, out), Image.BILINEAR))
    got = R.resize_u8_reference(img, out, out)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    b, k = R.pil_bilinear_coeffs(hw[1], out)
    assert b.shape == (out, 2) and int((b[:, 0] + b[:, 1]).max()) <= hw[1] and int(b[:, 1].min()) >= 1
    assert abs(int(k.sum(axis=1).max()) - (1 << R.PRECISION_BITS)) <= k.shape[1]   # rows sum to one in fixed point


def test_persistent_workgroup_caps_defaults_and_overrides(monkeypatch):
    """kernels.persist_wgs: 160 workgroups for the generator's persistent launches, 96 for the discriminator's; one global
    override (TECOGAN_PERSIST_WGS) applies to both, the per-network ones win; wgrad_plan deals pixel tiles over cap / blocks"""
    import importlib
    from pytorch_tecogan_amd import kernels as K
    for k in ("TECOGAN_PERSIST_WGS", "TECOGAN_PERSIST_WGS_G", "TECOGAN_PERSIST_WGS_D"):
        monkeypatch.delenv(k, raising=False)
    for k in ("TECOGAN_PERSIST_WGS_DREAL",):
        monkeypatch.delenv(k, raising=False)
    assert (K.PERSIST_WGS, K.persist_wgs("G"), K.persist_wgs("D"), K.persist_wgs(None)) == (160, 160, 96, 160)
    monkeypatch.setenv("TECOGAN_PERSIST_WGS_D", "128")
    assert K.persist_wgs("D") == 128 and K.persist_wgs("G") == 160
    monkeypatch.setenv("TECOGAN_PERSIST_WGS", "256")
    monkeypatch.delenv("TECOGAN_PERSIST_WGS_D")
    assert (K.persist_wgs("G"), K.persist_wgs("D")) == (256, 256)
    # 9-tap 128 -> 64 layer on 40 x 128x128 pixels: 2 channel blocks; all taps in one workgroup
    assert K.wgrad_plan(40, 128, 128, 1, 9, 128, 64, cap=160) == (80, 0)
    assert K.wgrad_plan(40, 128, 128, 1, 9, 128, 64, cap=256) == (128, 0)
    assert K.wgrad_plan(1, 8, 8, 1, 9, 64, 64, cap=160)[0] >= 1            # never zero splits
    monkeypatch.delenv("TECOGAN_PERSIST_WGS")
    importlib.reload(K)


def test_bench_launch_logic_spawns_its_ranks_and_refuses_mismatches():
    """bench.py --gpus N with no RANK in the environment starts N ranks itself (torch.distributed.run children, before any
    GPU call) - here with --dry: gloo rendezvous on the CPU, barrier, max-over-ranks reduce, the line's n_gpus taken from the
    process group.  Without --dry this box has no GPU: `--gpus 2` must fail loudly instead of printing a 1-GPU line, and so
    must a launcher whose WORLD_SIZE differs from --gpus."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["pg_world_size"] == 2 and line["config"] == {"parallelism": "dp2", "global_batch": 8}
    r = subprocess.run([sys.executable, bench, "--dry"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # the driver's scaling run goes to 8 ranks: the same launch logic at that size (configs[2]: global batch 32)
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--dry"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["pg_world_size"] == 8 and line["config"] == {"parallelism": "dp8", "global_batch": 32}
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, bench, "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode != 0 and "2 GPUs requested" in r.stderr and "{" not in r.stdout, (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--dry"], capture_output=True, text=True, timeout=120,
                       env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


def test_wgrad_work_list_plan_covers_every_unit_once_and_slots_are_disjoint():
    """engine.WgradList.plan (host logic of tg_wgrad_group): the unit ranges of the workgroups tile the work list, every channel
    block's slab slots [first, first + count) are exactly the slots `workgroup + block ordinal` of the workgroups whose range
    meets the block, and no two blocks share a slot."""
    from pytorch_tecogan_amd import engine as E
    shapes = [(40, 128, 128, 128, 64), (40, 64, 64, 128, 128), (40, 64, 64, 64, 128), (40, 64, 64, 64, 64)] + [(40, 32, 32, 64, 64)] * 5
    for cap in (1, 2, 5, 7, 96, 160, 100000):
        tw, rows, units, wgs, fold, slots = E.WgradList.plan(shapes, cap, 36928)
        per = (units + wgs - 1) // wgs           # the C side's rule
        nwg = (units + per - 1) // per
        assert tw == 32 and wgs == min(cap, units) and nwg <= cap and slots == nwg + sum((s[3] // 64) * (s[4] // 64) for s in shapes)
        assert (nwg - 1) * per < units <= nwg * per
        # units of job j: [rows[j][0], next); blocks in order
        used = set()
        g = 0
        for j, s in enumerate(shapes):
            tiles = s[0] * rows[j][6] * rows[j][7]
            assert rows[j][6] == (s[2] + 31) // 32 and rows[j][7] == (s[1] + 3) // 4 and rows[j][9] == g
            for blk in range((s[3] // 64) * (s[4] // 64)):
                beg, end = rows[j][0] + blk * tiles, rows[j][0] + (blk + 1) * tiles
                wgs = [w for w in range(nwg) if w * per < end and (w + 1) * per > beg]
                fj, a0, b0, first, count = fold[g]
                assert fj == j and (a0, b0) == ((blk // (s[4] // 64)) * 64, (blk % (s[4] // 64)) * 64)
                assert [w + g for w in wgs] == list(range(first, first + count))
                assert not (used & set(range(first, first + count)))
                used |= set(range(first, first + count))
                g += 1
        assert max(used) < slots
    assert E.WgradList.plan([(12, 16, 16, 128, 128)], 96, 36928)[0] == 16
    # 32-channel remainders are half-empty 64 blocks
    _, rows, units, _, fold, slots = E.WgradList.plan([(1, 32, 32, 32, 96)], 3, 36928)
    assert units == 2 * 8 and [(f[1], f[2]) for f in fold] == [(0, 0), (0, 64)]


def test_step_aware_caps_and_register_weights_routing(monkeypatch):
    """host rules added in round 3: the discriminator's real half runs at 72 persistent workgroups for chain-bound steps (<= 4096 LR
    pixels per pass) unless the environment fixes a cap; the register-weights kernel takes the trunk / c30 / masked-c32
    input-gradients of the batched G backward (TECOGAN_RW_EXTRA), the discriminator's stage 1 only where asked per conv"""
    import importlib
    import torch
    from pytorch_tecogan_amd import kernels as K
    for k in ("TECOGAN_PERSIST_WGS", "TECOGAN_PERSIST_WGS_G", "TECOGAN_PERSIST_WGS_D", "TECOGAN_PERSIST_WGS_DREAL", "TECOGAN_RW_EXTRA",
              "TECOGAN_RW"):
        monkeypatch.delenv(k, raising=False)
    K = importlib.reload(K)
    assert K.persist_wgs_g_for(4 * 32 * 32) == 144 and K.persist_wgs_g_for(2 * 64 * 64) == 160   # (round 4: r04_x)
    assert K.persist_wgs_dreal_for(4 * 32 * 32) == 96 and K.persist_wgs_dreal_for(2 * 64 * 64) is None
    monkeypatch.setenv("TECOGAN_PERSIST_WGS_DREAL", "96")
    assert K.persist_wgs_dreal_for(4096) is None
    monkeypatch.delenv("TECOGAN_PERSIST_WGS_DREAL")
    monkeypatch.setenv("TECOGAN_PERSIST_WGS_G", "128")
    assert K.persist_wgs_g_for(4096) is None
    monkeypatch.delenv("TECOGAN_PERSIST_WGS_G")
    bf = torch.bfloat16
    el = K.rw_eligible
    assert el(bf, 64, 64, 40, 32, 32, dgrad=True) and not el(bf, 64, 64, 4, 32, 32, dgrad=True)      # trunk input-gradients: batched only
    assert el(bf, 128, 64, 40, 64, 64, dgrad=True) and not el(bf, 128, 64, 4, 64, 64, dgrad=True)    # c30's input-gradient
    # FORWARD launches: from 16384 pixels since the kernel is wave-specialised (round 4, TECOGAN_RW_FWD_MIN; r04_x): the chain's
    # c30 / c32 / c6, config 5's HR stage, the discriminator's stage 1 - not conv0 of a 4 x 32 x 32 pass
    assert el(bf, 128, 64, 1, 512, 512) and el(bf, 128, 64, 4, 128, 128) and el(bf, 64, 128, 4, 64, 64) and el(bf, 64, 64, 4, 32, 32)
    assert not el(bf, 64, 64, 2, 32, 32)
    monkeypatch.setenv("TECOGAN_RW_FWD_MIN", "0")
    assert not el(bf, 64, 64, 40, 32, 32) and not el(bf, 128, 64, 1, 512, 512)
    monkeypatch.delenv("TECOGAN_RW_FWD_MIN")
    assert el(bf, 128, 128, 40, 64, 64, masked=True, dgrad=True)                          # c32's (masked)
    assert el(bf, 64, 64, 12, 64, 64) and el(bf, 64, 64, 12, 64, 64, dgrad=True)         # D stage 1: both directions, both halves (r04_x)
    assert el(bf, 128, 128, 12, 16, 16) and el(bf, 128, 128, 12, 32, 32)                 # D stage 3 (round 4: the wave-specialised kernel wins there) and stage 2
    assert not el(torch.float32, 64, 64, 40, 32, 32) and not el(bf, 32, 64, 40, 32, 32) and not el(bf, 64, 96, 40, 64, 64)
    monkeypatch.setenv("TECOGAN_RW_EXTRA", "none")
    K = importlib.reload(K)
    assert not K.rw_eligible(bf, 64, 64, 40, 32, 32, dgrad=True) and not K.rw_eligible(bf, 128, 128, 40, 64, 64, masked=True, dgrad=True)
    monkeypatch.delenv("TECOGAN_RW_EXTRA")
    importlib.reload(K)
    assert K.rgb_bwd_workgroups(40, 128, 128, 256) == 256 and K.rgb_bwd_workgroups(1, 20, 52, 256) == 2 * 4


def test_wgrad_work_list_plan_with_wide_channel_blocks():
    """WgradList.plan for the 64 x 128 block variants: b_blocks = ceil(Cy / 128), fold spans step by 128 in b, slot count follows;
    the variants are opt-in (variant_of keeps every layer in its kind's 64 x 64 list by default)"""
    from pytorch_tecogan_amd import _lib as L
    from pytorch_tecogan_amd import engine as E
    slot = 9 * 64 * 128 + 128
    tw, rows, units, wgs, fold, slots = E.WgradList.plan([(40, 64, 64, 128, 128), (40, 64, 64, 64, 256), (2, 16, 16, 128, 160)], 7, slot,
                                                        L.WGROUP_C3_B128)
    tiles = [40 * 2 * 16, 40 * 2 * 16, 2 * 1 * 4]
    assert tw == 32 and units == 2 * tiles[0] + 2 * tiles[1] + 2 * 2 * tiles[2]
    assert [(f[0], f[1], f[2]) for f in fold] == [(0, 0, 0), (0, 64, 0), (1, 0, 0), (1, 0, 128), (2, 0, 0), (2, 0, 128), (2, 64, 0), (2, 64, 128)]
    assert E.WgradList.block_b(L.WGROUP_CT_B128) == 128 and E.WgradList.block_b(L.WGROUP_C4S2) == 64
    assert E.WgradList.wide_min_pixels() == 0     # measured slower: opt-in only (experiments build)


def test_fold_items_prefix_table():
    """engine.fold_items (host side of tg_wgrad_fold_items): item counts per job = tiles x chunks of 8 slabs"""
    from pytorch_tecogan_amd import engine as E
    jobs = [[0, 0, 9, 576, 5, 9, 64, 64, 64, 64, 0, 36928],       # 4 tiles x 1 chunk
            [0, 0, 9, 576, 26, 9, 64, 64, 51, 64, 0, 36928],      # 4 x 4
            [0, 0, 16, 1024, 12, 16, 128, 128, 128, 128, 0, 262144],   # (128/16) * (128/64) = 16 tiles x 2
            [0, 0, 9, 576, 96, 9, 64, 32, 64, 3, 0, 18432]]       # BB = 32: AB = 32 -> 2 tiles x 12
    rows, n = E.fold_items(jobs)
    assert [r[12] for r in rows] == [0, 4, 20, 52] and n == 76 and all(r[:12] == j for r, j in zip(rows, jobs))


# ------------------------------------------------------------------------------------------------ tuning object
def test_tuning_defaults_equal_the_documented_optimum(monkeypatch):
    """ONE object holds every TECOGAN_* knob (pytorch-tecogan_amd/tuning.py).  With a clean environment its values are the
    measured optimum the documentation quotes (INTEGRATION.md's table is generated from the same KNOBS; DESIGN.md: caps 160 / 96 /
    72, four replica blocks, work lists, inline collectives, every rejected experiment off), and each knob names its evidence."""
    from pytorch_tecogan_amd import tuning
    for k in tuning.KNOBS.values():
        monkeypatch.delenv(k.env, raising=False)
    t = tuning.current()
    assert not t.explicit
    assert (t.cap("G"), t.cap("D"), t.cap(None), t.cap_g_for(4096), t.cap_dreal_for(4096), t.cap_dreal_for(8192)) == \
        (160, 96, 160, 144, 96, None)
    assert (t.cap_g_for(8192), t.rw_fwd_min, t.pair_rw_min, t.infer_wgs, t.infer_chunk) == (160, 4096, 32768, 256, 16)
    assert (t.cap_fwd_g_for(4096), t.cap_fwd_g_for(8192)) == (160, 0)
    assert (t.cap_trunk_g_for(4096), t.cap_trunk_g_for(8192), t.s2_cw, t.d_tail, t.mask_bits) == (160, 0, True, True, False)
    assert (t.graph, t.lanes, t.dreal_bwd, t.dp_inline, t.dp_buckets, t.cu_reserve, t.force_collectives) == \
        (True, True, True, True, True, 0, False)
    assert (t.rw, t.rw_extra, t.rw_extra_dreal) == ("1", "trunk,c30,m128,s3,s1", None)
    assert (t.wgrad_list, t.wgrad_groups, t.defer_finalize, t.fold_items, t.pack_blocks, t.stats_replicas) == \
        (True, True, True, True, 48, 4)
    assert (t.fused_resblock, t.fused_resblock_bwd, t.subpix_ct, t.fast_c4s2, t.rgb_out, t.rgb_bwd, t.rgb_bwd_wgs, t.rb_prefetch, t.rb_ws) == \
        (True, False, True, True, True, True, 0, False, True)
    assert (t.rgb_bwd_wgs_for(40 * 128 * 128), t.rgb_bwd_wgs_for(32 * 256 * 256)) == (160, 256)
    assert t.dtype == "bf16"
    # every rejected experiment is off, and is marked as needing the experiments build
    exp = [k for k in tuning.KNOBS.values() if k.experiment]
    assert {k.attr for k in exp} == {"rb_pair", "rb_pair_ws", "bn_fuse", "bn_bwd_fused", "bn_bwd_coop", "wgrad_b128_pixels", "mask_bits"}
    assert all(not getattr(t, k.attr) for k in exp)
    for k in tuning.KNOBS.values():
        assert k.evidence and k.doc, k.env
        for f in re.findall(r"profiles/[A-Za-z0-9_./]+", k.evidence):
            if "*" not in f and f.endswith((".log", ".json", ".csv")):
                assert os.path.exists(os.path.join(ROOT, f)), (k.env, f)
    # the INTEGRATION.md table is this object's
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for k in tuning.KNOBS.values():
        assert f"`{k.env}`" in doc, f"{k.env} missing from INTEGRATION.md's knob table"
    # nothing else in the package reads a TECOGAN_ variable (one parse site)
    pkg = os.path.join(ROOT, "pytorch-tecogan_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py") and fn not in ("tuning.py", "_lib.py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"environ[^\n]*TECOGAN_", src), fn
    # overrides are seen by the NEXT current() (tests / A-B tools set them between constructions)
    monkeypatch.setenv("TECOGAN_PERSIST_WGS", "256")
    monkeypatch.setenv("TECOGAN_DP_INLINE", "0")
    t2 = tuning.current()
    assert t2 is not t and (t2.cap("G"), t2.cap("D"), t2.cap_g_for(4096), t2.cap_dreal_for(4096), t2.dp_inline) == \
        (256, 256, None, None, False)
    assert tuning.current() is t2
    monkeypatch.setenv("TECOGAN_STATS_REPLICAS", "3")
    with pytest.raises(ValueError):
        tuning.current()


def test_rejected_experiments_need_the_experiments_library(monkeypatch):
    """the default library does not contain the rejected variants: switching one on fails loudly instead of silently running
    something else; the default .so reports itself as a non-experiments build"""
    from pytorch_tecogan_amd import tuning
    if L.has_experiments():
        pytest.skip("TECOGAN_LIB points at the experiments build")
    for attr in ("rb_pair", "bn_fuse", "bn_bwd_fused", "bn_bwd_coop", "wgrad_b128_pixels"):
        with pytest.raises(L.TecoganHipError, match="experiments build"):
            tuning.need_experiments(attr)
    lib = L.load()
    for name in L._PROTOS_EXPERIMENTS:
        assert not hasattr(lib, name)
    # argument forms of the experiments are refused by the default library (validated before any launch)
    assert lib.tg_wgrad_group_slot_floats_v(L.WGROUP_C3_B128) == -1
    import ctypes
    d = K.make_conv_desc(K.ConvSpec("c3", 64, 64).dgrad_geom(), L.TG_BF16, 1, 8, 8, 64, 8, 8, 64, mask_mode=L.MASK_BNZ, stats_mode=3)
    assert lib.tg_conv(ctypes.byref(d), 16, 16, None, None, 16, 16, 16, None) == -2  # TG_E_UNSUPPORTED


def test_bench_flags_and_dry_dp_fields():
    """bench.py's data-parallel flags (--dp-mode inline|buckets|both, --dp-steps, --no-extras) and the shape of the `dp` object a
    multi-rank line carries (rehearsed by the --dry launch: gloo rendezvous on the CPU, two ranks started by bench.py itself)"""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    a = bench.parse([])
    assert (a.dp_mode, a.dp_steps, a.no_extras, a.gpus) == (None, 10, False, 1)
    a = bench.parse(["--gpus", "8", "--dp-mode", "both", "--dp-steps", "4", "--no-extras"])
    assert (a.dp_mode, a.dp_steps, a.no_extras, a.gpus) == ("both", 4, True, 8)
    with pytest.raises(SystemExit):
        bench.parse(["--dp-mode", "ring"])
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--dp-mode", "both", "--dp-steps", "3"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and {line["dp"]["mode"], line["dp"]["alt"]["mode"]} == {"inline", "buckets"}
    for k in ("allreduce_exposed_ms_laneA", "allreduce_exposed_ms_laneB", "step_ms_no_collectives", "probe_steps"):
        assert k in line["dp"] and k in line["dp"]["alt"]
    assert line["dp"]["probe_steps"] == 3

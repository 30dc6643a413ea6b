"""Per-layer micro-benchmark of the MFMA kernels under hipGraph replay (what the training step sees):
a graph of REP dependent launches of one layer shape is replayed and the time per launch reported.
    python tools/microbench.py [conv|wgrad|all]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import _lib as L  # noqa: E402
from pytorch_tecogan_amd import kernels as K  # noqa: E402

DEV = "cuda:0"
REP = 40
TILES = {"auto": L.TILE_AUTO, "64x256": L.TILE_64x256, "64x64": L.TILE_64x64, "128x128": L.TILE_128x128,
         "32x128": L.TILE_32x128, "32x64": 5, "64x128": 6, "64x128w8": 7, "64x64w8": 8}


def time_graph(fn, reps=REP, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3  # us per launch


def bench_conv(kind, cin, cout, N, H, W, mode, dt, tiles):
    spec = K.ConvSpec(kind, cin, cout)
    OH, OW = spec.out_hw(H, W)
    if mode == "fwd":
        geom, (rows, Kd, s_row, s_k) = spec.fwd_geom(), spec.fwd_pack()
        x = torch.randn(N, H, W, K.pad32(cin), device=DEV).to(dt)
        out = torch.empty(N, OH, OW, K.pad32(cout), dtype=dt, device=DEV)
        dims = (N, H, W, K.pad32(cin), OH, OW, K.pad32(cout))
    else:
        geom, (rows, Kd, s_row, s_k) = spec.dgrad_geom(), spec.dgrad_pack()
        x = torch.randn(N, OH, OW, K.pad32(cout), device=DEV).to(dt)
        out = torch.empty(N, H, W, K.pad32(cin), dtype=dt, device=DEV)
        dims = (N, OH, OW, K.pad32(cout), H, W, K.pad32(cin))
    w = torch.randn(spec.weight_shape, device=DEV) * 0.05
    wp = K.pack_weights(dt, w, rows, Kd, s_row, s_k, spec.nslots, K.slot_table(spec.nslots, DEV))
    flops = 2.0 * N * (H * W if kind != "c4s2" else OH * OW) * spec.nslots * cin * cout
    res = []
    for name in tiles:
        d = K.make_conv_desc(geom, K.tg_dtype(dt), *dims, act=L.ACT_RELU if mode == "fwd" else L.ACT_NONE,
                             tile_cfg=TILES[name])
        try:
            us = time_graph(lambda: K.conv(d, x, wp, out))
        except L.TecoganHipError as e:
            res.append(f"{name}: n/a")
            continue
        res.append(f"{name}: {us:7.1f} us {flops / us / 1e6:7.1f} TF/s")
    print(f"conv {mode:5s} {kind:4s} {cin:3d}->{cout:3d} N={N:2d} {H}x{W} {str(dt)[6:]:8s} | " + " | ".join(res), flush=True)


def bench_wgrad(kind, cin, cout, N, H, W, dt, splits):
    spec = K.ConvSpec(kind, cin, cout)
    OH, OW = spec.out_hw(H, W)
    x_is_in, S, taps, ca, cb, s_a, s_b = spec.wgrad_info()
    xin = torch.randn(N, H, W, K.pad32(cin), device=DEV).to(dt)
    dout = torch.randn(N, OH, OW, K.pad32(cout), device=DEV).to(dt)
    X, Y = (xin, dout) if x_is_in else (dout, xin)
    flops = 2.0 * N * (H * W if kind != "c4s2" else OH * OW) * spec.nslots * cin * cout
    grad = torch.zeros(spec.weight_shape, device=DEV)
    slots = K.slot_table(len(taps), DEV)
    res = []
    for ns in splits:
        tpw = 0
        if ns == 0:
            nsplit, tpw = K.wgrad_plan(N, Y.shape[1], Y.shape[2], S, len(taps), X.shape[3], Y.shape[3])
        elif ns < 0:
            nsplit, tpw = -ns, (3 if len(taps) == 9 else 4)
        else:
            nsplit = ns
        desc = K.make_wgrad_desc(K.tg_dtype(dt), N, X.shape[1], X.shape[2], X.shape[3], Y.shape[1], Y.shape[2], Y.shape[3],
                                 S, taps, nsplit, tpw)
        slab = torch.empty(nsplit * len(taps) * X.shape[3] * Y.shape[3], device=DEV)

        def fn():
            K.wgrad(desc, X, Y, slab)
            K.wgrad_finalize(slab, nsplit, len(taps), X.shape[3], Y.shape[3], ca, cb, grad, s_a, s_b, slots, True)
        us = time_graph(fn)
        res.append(f"ns={nsplit}{'/tpw' + str(tpw) if tpw else ''}: {us:7.1f} us {flops / us / 1e6:6.1f} TF/s")
    print(f"wgrad {kind:4s} {cin:3d}->{cout:3d} N={N:2d} {H}x{W} {str(dt)[6:]:8s} | " + " | ".join(res), flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    bf = torch.bfloat16
    if what == "small":  # the deep discriminator layers (a few thousand output pixels): pure latency
        t = ["64x64", "32x128", "32x64", "64x128"]
        for N in (12, 24):
            bench_conv("c4s2", 128, 128, N, 32, 32, "fwd", bf, t)
            bench_conv("c4s2", 128, 64, N, 16, 16, "fwd", bf, t)
            bench_conv("c4s2", 64, 3, N, 8, 8, "fwd", bf, ["32x128", "32x64"])
            bench_conv("c3", 128, 128, N, 16, 16, "fwd", bf, t)
        bench_conv("c4s2", 128, 128, 24, 32, 32, "dgrad", bf, t)
        bench_conv("c4s2", 128, 64, 24, 16, 16, "dgrad", bf, t)
        bench_conv("c4s2", 64, 3, 24, 8, 8, "dgrad", bf, t)
        bench_conv("c3", 128, 128, 24, 16, 16, "dgrad", bf, t)
        bench_conv("c4s2", 64, 128, 24, 64, 64, "fwd", bf, t + ["128x128"])
        bench_conv("c4s2", 64, 128, 24, 64, 64, "dgrad", bf, t + ["128x128"])
        return
    if what == "w8s":  # small launches: 32x64 (4 waves) vs 64x64 with 8 waves
        t = ["32x64", "64x64w8", "64x64", "64x128w8"]
        bench_conv("c3", 51, 64, 4, 32, 32, "fwd", bf, t)
        bench_conv("c3", 64, 64, 4, 32, 32, "fwd", bf, t)
        bench_conv("c3", 128, 128, 12, 16, 16, "fwd", bf, t)
        bench_conv("c3", 128, 128, 24, 16, 16, "dgrad", bf, t)
        bench_conv("c3", 128, 128, 12, 32, 32, "fwd", bf, t)
        bench_conv("c3", 64, 128, 4, 64, 64, "fwd", bf, t)
        return
    if what == "w8":  # 64x128 tile: 4 waves vs 8 waves (two per SIMD from one workgroup)
        t = ["64x128", "64x128w8", "32x64"]
        bench_conv("c3", 64, 128, 4, 64, 64, "fwd", bf, t)
        bench_conv("c3", 128, 128, 4, 64, 64, "fwd", bf, t)
        bench_conv("c3", 128, 64, 4, 128, 128, "fwd", bf, t)
        bench_conv("c3", 64, 64, 4, 64, 64, "fwd", bf, t)
        bench_conv("c3", 51, 64, 4, 32, 32, "fwd", bf, t)
        bench_conv("c3", 64, 64, 12, 64, 64, "fwd", bf, t)
        bench_conv("c3", 128, 128, 12, 32, 32, "fwd", bf, t)
        bench_conv("c3", 128, 128, 12, 16, 16, "fwd", bf, t)
        bench_conv("c3", 64, 64, 40, 32, 32, "dgrad", bf, t)
        bench_conv("c3", 64, 64, 40, 64, 64, "dgrad", bf, t)
        bench_conv("c3", 64, 128, 40, 64, 64, "dgrad", bf, t)
        bench_conv("c3", 128, 128, 24, 32, 32, "dgrad", bf, t)
        return
    if what == "tiles":  # the launches the 64x128-vs-64x256 rule of pick_tile() decides
        big = ["64x256", "64x128"]
        bench_conv("c3", 128, 64, 4, 128, 128, "fwd", bf, big)
        bench_conv("c3", 128, 64, 40, 128, 128, "dgrad", bf, big)
        bench_conv("c3", 64, 3, 40, 128, 128, "dgrad", bf, big)
        bench_conv("c3", 27, 64, 24, 128, 128, "fwd", bf, big)
        bench_conv("c3", 64, 64, 24, 64, 64, "fwd", bf, big)
        bench_conv("c3", 128, 128, 24, 32, 32, "fwd", bf, big)
        bench_conv("c3", 64, 128, 40, 64, 64, "dgrad", bf, big)
        bench_conv("c3", 64, 128, 4, 64, 64, "fwd", bf, big + ["32x64"])
        bench_conv("c3", 128, 128, 4, 64, 64, "fwd", bf, big + ["32x64"])
        bench_conv("c3", 64, 64, 4, 64, 64, "fwd", bf, big + ["32x64", "64x64"])
        bench_conv("c3", 128, 128, 40, 64, 64, "dgrad", bf, big)
        bench_conv("c3", 64, 64, 40, 64, 64, "dgrad", bf, big)
    if what in ("conv", "all"):
        small = ["64x64", "32x64", "32x128", "64x256"]
        bench_conv("c3", 64, 64, 4, 32, 32, "fwd", bf, small)
        bench_conv("c3", 51, 64, 4, 32, 32, "fwd", bf, small)
        bench_conv("ct", 64, 64, 4, 32, 32, "fwd", bf, small)
        bench_conv("c3", 64, 64, 4, 64, 64, "fwd", bf, small)
        bench_conv("c3", 64, 128, 4, 64, 64, "fwd", bf, ["64x64", "32x64", "128x128", "64x256"])
        bench_conv("c3", 128, 128, 4, 64, 64, "fwd", bf, ["64x64", "32x64", "128x128", "64x256"])
        bench_conv("ct", 128, 128, 4, 64, 64, "fwd", bf, ["64x64", "128x128", "64x256", "64x128"])
        bench_conv("c3", 128, 64, 4, 128, 128, "fwd", bf, ["64x64", "32x64", "64x256", "64x128"])
        bench_conv("c3", 64, 3, 4, 128, 128, "fwd", bf, ["32x128", "32x64"])
        big = ["64x64", "64x256", "128x128", "64x128"]
        bench_conv("c3", 64, 64, 40, 32, 32, "dgrad", bf, big)
        bench_conv("c3", 128, 128, 40, 64, 64, "dgrad", bf, big)
        bench_conv("c3", 128, 64, 40, 128, 128, "dgrad", bf, big)
        bench_conv("ct", 128, 128, 40, 64, 64, "dgrad", bf, big + ["64x128"])
        bench_conv("c3", 64, 3, 40, 128, 128, "dgrad", bf, big)
        bench_conv("c3", 27, 64, 24, 128, 128, "fwd", bf, big)
        bench_conv("c4s2", 64, 64, 24, 128, 128, "fwd", bf, ["64x64", "128x128", "32x64"])
        bench_conv("c3", 64, 64, 24, 64, 64, "fwd", bf, big)
        bench_conv("c3", 128, 128, 24, 32, 32, "fwd", bf, big)
        bench_conv("c3", 128, 128, 24, 16, 16, "fwd", bf, ["64x64", "32x64", "128x128"])
        bench_conv("c4s2", 64, 64, 24, 128, 128, "dgrad", bf, big)
    if what in ("wgrad", "all"):
        sp = [0, -32, -85, -128, 128, 256]
        bench_wgrad("c3", 64, 64, 40, 32, 32, bf, sp)
        bench_wgrad("c3", 64, 64, 40, 64, 64, bf, sp)
        bench_wgrad("c3", 128, 128, 40, 64, 64, bf, sp)
        bench_wgrad("c3", 128, 64, 40, 128, 128, bf, sp)
        bench_wgrad("ct", 128, 128, 40, 64, 64, bf, sp)
        bench_wgrad("c3", 64, 3, 40, 128, 128, bf, sp)
        bench_wgrad("c3", 64, 64, 24, 64, 64, bf, sp)
        bench_wgrad("c3", 128, 128, 24, 32, 32, bf, sp)
        bench_wgrad("c4s2", 64, 64, 24, 128, 128, bf, sp)


if __name__ == "__main__":
    main()

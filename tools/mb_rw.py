"""Microbench: tg_conv3x3_rw (persistent, register weights) vs tg_conv on the dense 3x3 launch shapes of the step
(hipGraph replay of 20 launches each, so launch gaps are the graph's)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, kernels as K

dt = torch.bfloat16
CASES = [  # name, cin (reduction), cout, N, H, W, flip
    ("G trunk dgrad 64->64 @32 N40", 64, 64, 40, 32, 32, 1),
    ("G c20/c22 dgrad 64->64 @64 N40", 64, 64, 40, 64, 64, 1),
    ("G c6 dgrad 64->128 @128 N40", 64, 128, 40, 128, 128, 1),
    ("G c30 fwd 64->128 @64 N4", 64, 128, 4, 64, 64, 0),
    ("G c20 fwd 64->64 @64 N4", 64, 64, 4, 64, 64, 0),
    ("G trunk fwd 64->64 @32 N4", 64, 64, 4, 32, 32, 0),
    ("D s1 64->64 @64 N12", 64, 64, 12, 64, 64, 0),
    ("G c32 dgrad 128->128 @64 N40", 128, 128, 40, 64, 64, 1),
    ("G c30 dgrad 128->64 @64 N40", 128, 64, 40, 64, 64, 1),
    ("G c32 fwd 128->128 @64 N4", 128, 128, 4, 64, 64, 0),
    ("G c6 fwd 128->64 @128 N4", 128, 64, 4, 128, 128, 0),
    ("D s2 128->128 @32 N12", 128, 128, 12, 32, 32, 0),
    ("D s3 128->128 @16 N12", 128, 128, 12, 16, 16, 0),
]


def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 / reps * 1e6


for name, cin, cout, N, H, W, flip in CASES:
    x = torch.randn(N, H, W, cin, device="cuda").to(dt)
    out = torch.empty(N, H, W, cout, dtype=dt, device="cuda")
    # packed image [9][cin/32][cout][32]: any values will do for timing
    wp = (torch.randn(9 * cin * cout, device="cuda") * 0.05).to(dt)
    geom = K.ConvSpec("c3", cout, cin).dgrad_geom() if flip else K.ConvSpec("c3", cin, cout).fwd_geom()
    d = K.make_conv_desc(geom, L.TG_BF16, N, H, W, cin, H, W, cout)
    t_old = bench(lambda: K.conv(d, x, wp, out))
    gf = 2.0 * N * H * W * 9 * cin * cout / 1e9
    line = f"{name:34s} tg_conv {t_old:8.1f} us ({gf / t_old * 1e3:6.0f} TF/s)"
    try:
        t_new = bench(lambda: K.conv3x3_rw(x, wp, out, bool(flip)))
        line += f" | rw {t_new:8.1f} us ({gf / t_new * 1e3:6.0f} TF/s)"
        for cap in (128, 512):
            t_c = bench(lambda: K.conv3x3_rw(x, wp, out, bool(flip), max_workgroups=cap))
            line += f" | cap{cap} {t_c:7.1f}"
    except Exception as e:  # noqa: BLE001
        line += f" | rw failed: {e}"
    print(line, flush=True)

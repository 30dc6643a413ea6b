"""The generator trunk's backward dgrad chain as the step runs it: 64 dependent launches, each with its own weights,
mask (forward activation) and output buffer, epilogue = relu mask + residual + bias-gradient statistics.  Sweeps the tile
configuration; prints microseconds per launch under hipGraph replay."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd
from pytorch_tecogan_amd import _lib as L, kernels as K
from microbench import time_graph, TILES
DEV = "cuda:0"; dt = torch.bfloat16
N, H, C_, NL = 40, 32, 64, 64
if os.environ.get("MB_SHAPE"):  # "N,H,C,NL"
    N, H, C_, NL = [int(v) for v in os.environ["MB_SHAPE"].split(",")]
spec = K.ConvSpec("c3", C_, C_)
geom, (rows, Kd, s_row, s_k) = spec.dgrad_geom(), spec.dgrad_pack()
bufs = [torch.randn(N, H, H, C_, device=DEV).to(dt) for _ in range(NL + 1)]
masks = [torch.randn(N, H, H, C_, device=DEV).to(dt) for _ in range(NL)]
wps = [K.pack_weights(dt, torch.randn(spec.weight_shape, device=DEV) * 0.05, rows, Kd, s_row, s_k, spec.nslots,
                      K.slot_table(spec.nslots, DEV)) for _ in range(NL)]
stats = [torch.zeros(2 * C_, device=DEV) for _ in range(NL)]
dims = (N, H, H, C_, H, H, C_)
for variant in os.environ.get("MB_VARIANTS", "plain,mask,mask+res,mask+stats,mask+res+stats").split(","):
    for tile in sys.argv[1:] or ["64x256", "64x128", "32x128", "64x64", "32x64"]:
        d = K.make_conv_desc(geom, K.tg_dtype(dt), *dims, mask_mode=L.MASK_RELU if "mask" in variant else 0,
                             stats_mode=1 if "stats" in variant else 0, stats_groups=1, tile_cfg=TILES[tile])
        def chain():
            for i in range(NL):
                K.conv(d, bufs[i], wps[i], bufs[i + 1], mask=masks[i] if "mask" in variant else None,
                       res=bufs[max(i - 1, 0)] if "res" in variant else None, stats=stats[i] if "stats" in variant else None)
        try:
            us = time_graph(chain, reps=1, iters=20) / NL
            print(f"dgrad c3 {C_}->{C_} N={N} {H}x{H} {variant:16s} {tile:8s} {us:7.1f} us/launch", flush=True)
        except Exception as e:
            print(f"dgrad c3 {C_}->{C_} N={N} {H}x{H} {variant:16s} {tile:8s} n/a ({str(e)[:60]})", flush=True)

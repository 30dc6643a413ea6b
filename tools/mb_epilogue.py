import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd
from pytorch_tecogan_amd import _lib as L, kernels as K
from microbench import time_graph
DEV="cuda:0"; dt=torch.bfloat16
def run(kind, cin, cout, N, H, mode, variants):
    spec = K.ConvSpec(kind, cin, cout); OH, OW = spec.out_hw(H, H)
    geom, (rows, Kd, s_row, s_k) = (spec.dgrad_geom(), spec.dgrad_pack()) if mode == "dgrad" else (spec.fwd_geom(), spec.fwd_pack())
    if mode == "dgrad":
        x = torch.randn(N, OH, OW, K.pad32(cout), device=DEV).to(dt); out = torch.empty(N, H, H, K.pad32(cin), dtype=dt, device=DEV)
        dims = (N, OH, OW, K.pad32(cout), H, H, K.pad32(cin)); co = K.pad32(cin)
    else:
        x = torch.randn(N, H, H, K.pad32(cin), device=DEV).to(dt); out = torch.empty(N, OH, OW, K.pad32(cout), dtype=dt, device=DEV)
        dims = (N, H, H, K.pad32(cin), OH, OW, K.pad32(cout)); co = K.pad32(cout)
    w = torch.randn(spec.weight_shape, device=DEV) * 0.05
    wp = K.pack_weights(dt, w, rows, Kd, s_row, s_k, spec.nslots, K.slot_table(spec.nslots, DEV))
    mask = torch.randn_like(out); res = torch.randn_like(out); bias = torch.zeros(co, device=DEV)
    for name in variants:
        R = 64 if "rep" in name else 1
        stats = torch.zeros(R * 2 * co, device=DEV) if "stats" in name else None
        d = K.make_conv_desc(geom, K.tg_dtype(dt), *dims, mask_mode=L.MASK_RELU if "mask" in name else 0,
                             stats_mode=1 if "stats" in name else 0, stats_groups=1, stats_replicas=R,
                             act=L.ACT_RELU if "relu" in name else 0)
        us = time_graph(lambda: K.conv(d, x, wp, out, bias=bias if "bias" in name else None, mask=mask if "mask" in name else None,
                                       res=res if "res" in name else None, stats=stats))
        print(f"{kind} {cin}->{cout} N={N} {H}x{H} {mode:5s} {name:22s} {us:8.1f} us", flush=True)
run("c3", 128, 64, 40, 128, "dgrad", ["plain", "mask", "stats", "stats+rep", "mask+stats+rep", "res"])
run("c3", 64, 64, 40, 32, "dgrad", ["plain", "mask", "mask+stats", "mask+stats+rep", "res+mask+stats"])
run("c3", 128, 64, 4, 128, "fwd", ["plain", "bias+relu"])

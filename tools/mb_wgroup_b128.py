"""64 x 128 channel blocks (TG_WGROUP_C3_B128 / TG_WGROUP_CT_B128) against the 64 x 64 work lists on the layers with >= 128 Y channels,
kernel only, under hipGraph replay.     python tools/mb_wgroup_b128.py [cap ...]   (default 160 256)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import _lib as L  # noqa: E402
from pytorch_tecogan_amd import engine as E  # noqa: E402
from microbench import time_graph  # noqa: E402

DEV, bf = "cuda:0", torch.bfloat16
# (kind, [(N, H, W of Y, Cx, Cy)])
SETS = {
    "G c32 128->128 @64 N=40": ("c3", [(40, 64, 64, 128, 128)]),
    "G c30 64->128 @64 N=40": ("c3", [(40, 64, 64, 64, 128)]),
    "G c32 + c30": ("c3", [(40, 64, 64, 128, 128), (40, 64, 64, 64, 128)]),
    "G ct4 128->128 @64->128 N=40": ("ct", [(40, 64, 64, 128, 128)]),
    "D stage2 8 x 128->128 @32 N=12": ("c3", [(12, 32, 32, 128, 128)] * 8),
    "D stage3 8 x 128->128 @16 N=12": ("c3", [(12, 16, 16, 128, 128)] * 8),
}


def run(name, kind, shapes, cap):
    lib = L.load()
    S = 1 if kind == "c3" else 2
    Xs = [torch.randn(N, S * H, S * W, cx, device=DEV).to(bf) for N, H, W, cx, cy in shapes]
    Ys = [torch.randn(N, H, W, cy, device=DEV).to(bf) for N, H, W, cx, cy in shapes]
    fl = sum(2.0 * N * H * W * 9 * cx * cy for N, H, W, cx, cy in shapes)
    out = []
    for variant in (E.WgradList.VARIANT[kind], E.WgradList.WIDE[E.WgradList.VARIANT[kind]]):
        slot = int(lib.tg_wgrad_group_slot_floats_v(variant))
        tw, rows, units, nwg, fold, slots = E.WgradList.plan(shapes, cap, slot, variant)
        slab = torch.empty(slots * slot, device=DEV)
        jt = torch.tensor([[X.data_ptr(), Y.data_ptr()] + r for X, Y, r in zip(Xs, Ys, rows)], dtype=torch.int64, device=DEV)

        def k():
            L.check(lib.tg_wgrad_group_v(L.TG_BF16, variant, tw, jt.data_ptr(), len(shapes), units, nwg, slab.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "tg_wgrad_group_v")
        t = time_graph(k, reps=10)
        out.append(f"{t:7.1f} us {fl / t / 1e6:6.0f} TF/s ({units} units, slabs {slots * slot * 4 / 1e6:5.1f} MB)")
    print(f"{name:34s} cap {cap:3d} | 64x64 {out[0]} | 64x128 {out[1]}", flush=True)


if __name__ == "__main__":
    for name, (kind, shapes) in SETS.items():
        for cap in [int(a) for a in sys.argv[1:]] or [160, 256]:
            run(name, kind, shapes, min(cap, 96) if name.startswith("D") and cap == 160 else cap)

"""Work-list weight-gradient launch (tg_wgrad_group, csrc/wgrad_group.hip) against the per-layer / same-shape launches
(tg_wgrad, tg_wgrad_multi) on the step's layer sets, each under hipGraph replay; kernel only (no fold) and with its fold.
    python tools/mb_wgroup.py [cap ...]        (workgroup caps; default 160 256)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import _lib as L  # noqa: E402
from pytorch_tecogan_amd import engine as E  # noqa: E402
from pytorch_tecogan_amd import kernels as K  # noqa: E402
from microbench import time_graph  # noqa: E402

DEV = "cuda:0"
bf = torch.bfloat16

SETS = {
    "G c6 128->64 @128 N=40": [(40, 128, 128, 128, 64)],
    "G c32 128->128 @64 N=40": [(40, 64, 64, 128, 128)],
    "G c30 64->128 @64 N=40": [(40, 64, 64, 64, 128)],
    "G c22 64->64 @64 N=40": [(40, 64, 64, 64, 64)],
    "G up-sampling stage (c6 c32 c30 c22 c20)": [(40, 128, 128, 128, 64), (40, 64, 64, 128, 128), (40, 64, 64, 64, 128),
                                                  (40, 64, 64, 64, 64), (40, 64, 64, 64, 64)],
    "G trunk 34 x 64->64 @32 N=40": [(40, 32, 32, 64, 64)] * 34,
    "D stage1 8 x 64->64 @64 N=12": [(12, 64, 64, 64, 64)] * 8,
    "D stage2 8 x 128->128 @32 N=12": [(12, 32, 32, 128, 128)] * 8,
    "D stage3 8 x 128->128 @16 N=12": [(12, 16, 16, 128, 128)] * 8,
}


def flops(shapes):
    return sum(2.0 * N * H * W * 9 * cx * cy for N, H, W, cx, cy in shapes)


def run_set(name, shapes, cap):
    lib = L.load()
    slot = int(lib.tg_wgrad_group_slot_floats_v(L.WGROUP_C3))
    Xs = [torch.randn(N, H, W, cx, device=DEV).to(bf) for N, H, W, cx, cy in shapes]
    Ys = [torch.randn(N, H, W, cy, device=DEV).to(bf) for N, H, W, cx, cy in shapes]
    fl = flops(shapes)
    # ---- new: one work-list launch (+ fold with one job per channel block)
    tw, rows, units, nwg, fold, slots = E.WgradList.plan(shapes, cap, slot)
    slab = torch.empty(slots * slot, device=DEV)
    jt = torch.tensor([[X.data_ptr(), Y.data_ptr()] + r for X, Y, r in zip(Xs, Ys, rows)], dtype=torch.int64, device=DEV)
    grads = [torch.zeros(cy, cx, 3, 3, device=DEV) for _, _, _, cx, cy in shapes]
    fin = torch.tensor([[slab.data_ptr() + 4 * slot * first, grads[j].data_ptr() + 4 * (a0 * 9 + b0 * shapes[j][3] * 9), 9, shapes[j][3] * 9,
                         count, 9, 64, 64, 64, 64, 0, slot] for j, a0, b0, first, count in fold], dtype=torch.int64, device=DEV)

    def new_k():
        L.check(lib.tg_wgrad_group_v(L.TG_BF16, L.WGROUP_C3, tw, jt.data_ptr(), len(shapes), units, nwg, slab.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream), "tg_wgrad_group_v")

    def new_kf():
        new_k()
        L.check(lib.tg_wgrad_finalize_multi(fin.data_ptr(), fin.shape[0], 8, torch.cuda.current_stream().cuda_stream), "fold")

    t_new, t_newf = time_graph(new_k, reps=10), time_graph(new_kf, reps=10)
    # ---- old: tg_wgrad per layer (distinct shapes) or tg_wgrad_multi (same-shaped layers), + tg_wgrad_finalize_multi
    taps = [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]
    same = len(set(shapes)) == 1 and len(shapes) > 1
    launches, fins = [], []
    if same:
        N, H, W, cx, cy = shapes[0]
        blocks = K.wgrad_blocks(9, cx, cy)
        nsplit = max(1, min(K.wgrad_tiles(N, H, W, 1), cap // (len(shapes) * blocks)))
        stride = 9 * cx * cy
        slabs = [torch.empty(nsplit * stride, device=DEV) for _ in shapes]
        desc = K.make_wgrad_desc(L.TG_BF16, N, H, W, cx, H, W, cy, 1, taps, nsplit, 0)
        jobs = torch.tensor([[X.data_ptr(), Y.data_ptr(), s.data_ptr()] for X, Y, s in zip(Xs, Ys, slabs)], dtype=torch.int64, device=DEV)
        launches.append(lambda: K.wgrad_multi(desc, jobs, len(shapes)))
        fins = [[s.data_ptr(), g.data_ptr(), 9, cx * 9, nsplit, 9, cx, cy, cx, cy, 0, stride] for s, g in zip(slabs, grads)]
    else:
        for (N, H, W, cx, cy), X, Y, g in zip(shapes, Xs, Ys, grads):
            nsplit, tpw = K.wgrad_plan(N, H, W, 1, 9, cx, cy, cap=cap)
            stride = 9 * cx * cy
            sl = torch.empty(nsplit * stride, device=DEV)
            desc = K.make_wgrad_desc(L.TG_BF16, N, H, W, cx, H, W, cy, 1, taps, nsplit, tpw)
            launches.append(lambda d=desc, X=X, Y=Y, sl=sl: K.wgrad(d, X, Y, sl))
            fins.append([sl.data_ptr(), g.data_ptr(), 9, cx * 9, nsplit, 9, cx, cy, cx, cy, 0, stride])
    fin_old = torch.tensor(fins, dtype=torch.int64, device=DEV)

    def old_k():
        for f in launches:
            f()

    def old_kf():
        old_k()
        L.check(lib.tg_wgrad_finalize_multi(fin_old.data_ptr(), fin_old.shape[0], 64, torch.cuda.current_stream().cuda_stream), "fold")

    t_old, t_oldf = time_graph(old_k, reps=10), time_graph(old_kf, reps=10)
    print(f"{name:44s} cap {cap:3d} | old {t_old:7.1f} us {fl / t_old / 1e6:6.0f} TF/s, +fold {t_oldf:7.1f} | "
          f"work list {t_new:7.1f} us {fl / t_new / 1e6:6.0f} TF/s, +fold {t_newf:7.1f} | slabs {slots * slot * 4 / 1e6:6.1f} MB "
          f"({nwg} wgs, {units} units)", flush=True)


def main():
    caps = [int(a) for a in sys.argv[1:]] or [160, 256]
    for name, shapes in SETS.items():
        for cap in caps:
            c = min(cap, 96) if name.startswith("D") and cap == 160 else cap
            run_set(name, shapes, c)


if __name__ == "__main__":
    main()

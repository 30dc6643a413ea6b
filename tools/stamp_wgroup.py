"""Diagnostic: phase stamps (s_memtime ticks) of waves 0 and 7 of workgroup 0 of the work-list weight-gradient kernel
(csrc/wgrad_group.hip built with -DTG_STAMP: tools/build_variant.sh wg_stamp wgrad_group -DTG_STAMP;
TECOGAN_LIB=_ab/libtecogan_hip_wg_stamp.so).  Per tile: wait for the tile's DMA | barrier | issue of the next tile's DMA | k-loop."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, engine as E
from microbench import time_graph

lib = L.load()
lib.tg_debug_read_wg_stamps.restype = ctypes.c_int
lib.tg_debug_read_wg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
bf = torch.bfloat16
SETS = {"G c6 128->64 @128 N=40": [(40, 128, 128, 128, 64)], "G c22 64->64 @64 N=40": [(40, 64, 64, 64, 64)],
        "G trunk 34 x 64->64 @32 N=40": [(40, 32, 32, 64, 64)] * 34, "D stage2 8 x 128->128 @32 N=12": [(12, 32, 32, 128, 128)] * 8}
for cap in (256, 144):
    for name, shapes in SETS.items():
        slot = int(lib.tg_wgrad_group_slot_floats_v(L.WGROUP_C3))
        Xs = [torch.randn(N, H, W, cx, device="cuda").to(bf) for N, H, W, cx, cy in shapes]
        Ys = [torch.randn(N, H, W, cy, device="cuda").to(bf) for N, H, W, cx, cy in shapes]
        tw, rows, units, nwg, fold, slots = E.WgradList.plan(shapes, cap, slot)
        slab = torch.empty(slots * slot, device="cuda")
        jt = torch.tensor([[X.data_ptr(), Y.data_ptr()] + r for X, Y, r in zip(Xs, Ys, rows)], dtype=torch.int64, device="cuda")
        run = lambda: L.check(lib.tg_wgrad_group_v(L.TG_BF16, L.WGROUP_C3, tw, jt.data_ptr(), len(shapes), units, nwg, slab.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream), "tg_wgrad_group_v")
        us = time_graph(run, reps=10)
        run(); torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 80)()
        lib.tg_debug_read_wg_stamps(buf, 80)
        fl = sum(2.0 * N * H * W * 9 * cx * cy for N, H, W, cx, cy in shapes)
        print(f"== {name} cap {cap}: {us:.1f} us, {fl / us / 1e6:.0f} TFLOP/s, {units} units on {nwg} workgroups (tile width {tw})")
        for w in range(2):
            t = list(buf)[w * 40:(w + 1) * 40]
            line = f"  wave {0 if w == 0 else 7}:"
            for i in range(8):
                a = t[i * 5:i * 5 + 5]
                if i and a[0] <= t[(i - 1) * 5]:
                    break
                nm = ("issue", "k-loop") if w == 0 else ("DMA issue", "bias sums")   # wave-specialised build: wave 0 consumes, wave 7 produces
                line += f" t{i}: dma-wait {a[1]-a[0]} barrier {a[2]-a[1]} {nm[0]} {a[3]-a[2]} {nm[1]} {a[4]-a[3]} = {a[4]-a[0]} |"
            print(line)

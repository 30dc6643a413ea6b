"""Diagnostic: phase stamps of workgroup 0 of one trunk conv launch (needs a library built with -DTG_STAMP)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd
from pytorch_tecogan_amd import _lib as L, kernels as K
lib = L.load()
lib.tg_debug_read_stamps.restype = ctypes.c_int
lib.tg_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dt = torch.bfloat16
for (kind, cin, cout, N, H, tile) in (("c3",64,64,4,32,5), ("c3",64,64,40,32,6), ("c3",64,64,12,64,6), ("c3",128,128,12,16,5), ("c3",128,128,12,32,5), ("c3",128,64,4,128,7), ("ct",128,128,4,64,6), ("c4s2",128,64,12,16,2), ("c4s2",128,128,12,32,2), ("c4s2",64,64,12,128,1)):
    spec = K.ConvSpec(kind, cin, cout)
    OH, OW = spec.out_hw(H, H)
    x = torch.randn(N, H, H, K.pad32(cin), device="cuda").to(dt)
    out = torch.empty(N, OH, OW, K.pad32(cout), dtype=dt, device="cuda")
    w = torch.randn(spec.weight_shape, device="cuda") * 0.05
    rows, Kd, s_row, s_k = spec.fwd_pack()
    wp = K.pack_weights(dt, w, rows, Kd, s_row, s_k, spec.nslots, K.slot_table(spec.nslots, "cuda"))
    d = K.make_conv_desc(spec.fwd_geom(), K.tg_dtype(dt), N, H, H, K.pad32(cin), OH, OW, K.pad32(cout), act=L.ACT_RELU, tile_cfg=tile)
    for rep in range(3):
        K.conv(d, x, wp, out); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 8)()
    lib.tg_debug_read_stamps(buf, 8)
    t = list(buf)
    print(kind, cin, cout, N, H, "tile", tile, f"| issue+wait {t[1]-t[0]} | LDS stores {t[3]-t[1]} | barrier {t[4]-t[3]} | k-loop {t[5]-t[4]} | "
          f"epilogue {t[6]-t[5]} | total {t[6]-t[0]} cycles (last stage only for issue/stores/barrier)")

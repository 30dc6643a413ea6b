"""conv-transpose forward: the sub-pixel launch (tg_convt_fwd) vs the four-class tg_conv launch vs the persistent class-waves kernel
(tg_convt_fwd_cw, round 5; at 256 / 192 / 144 workgroups), under hipGraph replay"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd
from pytorch_tecogan_amd import _lib as L, kernels as K
from microbench import time_graph
dt = torch.bfloat16
for cin, cout, N, H in ((64, 64, 4, 32), (128, 128, 4, 64), (64, 64, 2, 64), (128, 128, 2, 128), (64, 64, 1, 128), (128, 128, 1, 256)):
    spec = K.ConvSpec("ct", cin, cout)
    x = torch.randn(N, H, H, cin, device="cuda").to(dt)
    w = torch.randn(spec.weight_shape, device="cuda") * 0.05
    rows, Kd, s_row, s_k = spec.fwd_pack()
    wp = K.pack_weights(dt, w, rows, Kd, s_row, s_k, 9, K.slot_table(9, "cuda"))
    b = torch.zeros(cout, device="cuda")
    out = torch.empty(N, 2 * H, 2 * H, cout, dtype=dt, device="cuda")
    d = K.make_conv_desc(spec.fwd_geom(), L.TG_BF16, N, H, H, cin, 2 * H, 2 * H, cout, act=L.ACT_RELU)
    t_cls = time_graph(lambda: K.conv(d, x, wp, out, bias=b))
    t_sub = time_graph(lambda: K.convt_fwd(x, wp, b, out, L.ACT_RELU))
    gf = 2.0 * N * H * H * 9 * cin * cout / 1e9
    t_cw = [time_graph(lambda: K.convt_fwd_cw(x, wp, b, out, L.ACT_RELU, max_workgroups=c)) for c in (256, 192, 144)]
    print(f"ct {cin}->{cout} N={N} {H}x{H}: four-class tg_conv {t_cls:6.1f} us ({gf / t_cls * 1e3:6.1f} TF/s) | sub-pixel {t_sub:6.1f} us ({gf / t_sub * 1e3:6.1f} TF/s)"
          f" | class-waves @256/192/144 {t_cw[0]:6.1f} / {t_cw[1]:6.1f} / {t_cw[2]:6.1f} us ({gf / t_cw[0] * 1e3:6.1f} TF/s)")

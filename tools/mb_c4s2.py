"""4x4 stride-2 forward: tg_conv4s2_fwd vs the generic tg_conv path, under hipGraph replay (discriminator shapes)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd
from pytorch_tecogan_amd import _lib as L, kernels as K
from microbench import time_graph
dt = torch.bfloat16
for cin, cout, N, H in ((64, 64, 12, 128), (64, 128, 12, 64), (128, 128, 12, 32), (128, 64, 12, 16), (64, 64, 24, 128), (128, 128, 24, 32)):
    spec = K.ConvSpec("c4s2", cin, cout)
    x = torch.randn(N, H, H, cin, device="cuda").to(dt)
    w = torch.randn(spec.weight_shape, device="cuda") * 0.05
    rows, Kd, s_row, s_k = spec.fwd_pack()
    wp = K.pack_weights(dt, w, rows, Kd, s_row, s_k, 16, K.slot_table(16, "cuda"))
    out = torch.empty(N, H // 2, H // 2, cout, dtype=dt, device="cuda")
    stats = torch.zeros(1, 2, cout, device="cuda")
    d = K.make_conv_desc(spec.fwd_geom(), L.TG_BF16, N, H, H, cin, H // 2, H // 2, cout, stats_mode=2, stats_groups=1)
    t_gen = time_graph(lambda: K.conv(d, x, wp, out, stats=stats))
    t_new = time_graph(lambda: K.conv4s2_fwd(x, wp, None, out, stats, 1))
    gf = 2.0 * N * (H // 2) ** 2 * 16 * cin * cout / 1e9
    print(f"c4s2 {cin}->{cout} N={N} {H}x{H}: generic {t_gen:6.1f} us ({gf / t_gen * 1e3:6.1f} TF/s) | compile-time taps {t_new:6.1f} us ({gf / t_new * 1e3:6.1f} TF/s)")

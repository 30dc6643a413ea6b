"""4x4 stride-2 input-gradient: sub-pixel launch (tg_conv4s2_dgrad) vs the generic four-class tg_conv vs the persistent class-waves
kernel (tg_conv4s2_dgrad_cw, round 5; at 256 / 96 / 80 workgroups), under hipGraph replay; N = 12 is a half of the step"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd
from pytorch_tecogan_amd import _lib as L, kernels as K
from microbench import time_graph
dt = torch.bfloat16
for cin, cout, N, H, masked in ((64, 64, 12, 128, True), (64, 128, 12, 64, True), (128, 128, 12, 32, True), (128, 64, 12, 16, True), (64, 64, 24, 128, True)):
    spec = K.ConvSpec("c4s2", cin, cout)
    dout = torch.randn(N, H // 2, H // 2, cout, device="cuda").to(dt)
    w = torch.randn(spec.weight_shape, device="cuda") * 0.05
    rows, Kd, s_row, s_k = spec.dgrad_pack()
    wb = K.pack_weights(dt, w, rows, Kd, s_row, s_k, 16, K.slot_table(16, "cuda"))
    dx = torch.empty(N, H, H, cin, dtype=dt, device="cuda")
    mask = torch.randn(N, H, H, cin, device="cuda").to(dt) if masked else None
    mm = L.MASK_LRELU if masked else L.MASK_NONE
    d = K.make_conv_desc(spec.dgrad_geom(), L.TG_BF16, N, H // 2, H // 2, cout, H, H, cin, mask_mode=mm)
    t_gen = time_graph(lambda: K.conv(d, dout, wb, dx, mask=mask))
    t_new = time_graph(lambda: K.conv4s2_dgrad(dout, wb, dx, mask, mm))
    gf = 2.0 * N * (H // 2) ** 2 * 16 * cin * cout / 1e9
    t_cw = [time_graph(lambda: K.conv4s2_dgrad_cw(dout, wb, dx, mask, mm, max_workgroups=c)) for c in (256, 96, 80)]
    print(f"c4s2 dgrad {cin}->{cout} N={N} {H}x{H} mask={masked}: generic {t_gen:6.1f} us ({gf / t_gen * 1e3:6.1f} TF/s) | sub-pixel {t_new:6.1f} us ({gf / t_new * 1e3:6.1f} TF/s)"
          f" | class-waves @256/96/80 {t_cw[0]:6.1f} / {t_cw[1]:6.1f} / {t_cw[2]:6.1f} us ({gf / t_cw[0] * 1e3:6.1f} TF/s)")

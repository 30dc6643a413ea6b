"""Stride-2 gathers: csrc/conv_s2_cw.hip (register weights, round 6) vs csrc/conv4s2_mfma.hip (per-tile staging), kernel alone under
hipGraph replay.  The discriminator's four 4x4 s2 forwards at N = 12 (a half of config 2) at lane B's cap, the two conv-transpose
input-gradients of the batched G backward (N = 40) at lane A's cap; + full-chip numbers."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd
from pytorch_tecogan_amd import _lib as L, kernels as K
from microbench import time_graph
dt = torch.bfloat16
print("4x4 stride-2 forward + statistics (old: tg_conv4s2_fwd_capped at the same cap)")
for cin, cout, N, H in ((64, 64, 12, 128), (64, 128, 12, 64), (128, 128, 12, 32), (128, 64, 12, 16)):
    spec = K.ConvSpec("c4s2", cin, cout)
    x = torch.randn(N, H, H, cin, device="cuda").to(dt)
    w = torch.randn(spec.weight_shape, device="cuda") * 0.05
    rows, Kd, s_row, s_k = spec.fwd_pack()
    wp = K.pack_weights(dt, w, rows, Kd, s_row, s_k, 16, K.slot_table(16, "cuda"))
    out = torch.empty(N, H // 2, H // 2, cout, dtype=dt, device="cuda")
    stats = torch.zeros(4, 1, 2, cout, device="cuda")
    gf = 2.0 * N * (H // 2) ** 2 * 16 * cin * cout / 1e9
    row = []
    for cap in (96, 256):
        t_old = time_graph(lambda: K.conv4s2_fwd(x, wp, None, out, stats, 1, stats_replicas=4, max_workgroups=cap if cap < 256 else 0))
        t_new = time_graph(lambda: K.conv4s2_fwd_cw(x, wp, None, out, stats, 1, stats_replicas=4, max_workgroups=cap))
        row.append(f"cap {cap}: old {t_old:6.1f} us ({gf / t_old * 1e3:5.0f} TF/s) new {t_new:6.1f} us ({gf / t_new * 1e3:5.0f} TF/s)")
    print(f"c4s2 {cin}->{cout} N={N} {H}x{H}: " + " | ".join(row))
print("conv-transpose k3 s2 input-gradient (old: tg_convt_dgrad, full grid)")
for cin, cout, N, H in ((64, 64, 40, 32), (128, 128, 40, 64), (128, 128, 16, 128)):
    spec = K.ConvSpec("ct", cin, cout)
    w = torch.randn(spec.weight_shape, device="cuda") * 0.05
    rows, Kd, s_row, s_k = spec.dgrad_pack()
    wb = K.pack_weights(dt, w, rows, Kd, s_row, s_k, 9, K.slot_table(9, "cuda"))
    dout = torch.randn(N, 2 * H, 2 * H, cout, device="cuda").to(dt)
    dx = torch.empty(N, H, H, cin, dtype=dt, device="cuda")
    gf = 2.0 * N * H * H * 9 * cin * cout / 1e9
    t_old = time_graph(lambda: K.convt_dgrad(dout, wb, dx))
    row = [f"old {t_old:6.1f} us ({gf / t_old * 1e3:5.0f} TF/s)"]
    for cap in (144, 192, 256):
        t_new = time_graph(lambda: K.convt_dgrad_cw(dout, wb, dx, max_workgroups=cap))
        row.append(f"new@{cap} {t_new:6.1f} us ({gf / t_new * 1e3:5.0f} TF/s)")
    print(f"ct-dgrad {cin}<-{cout} N={N} {H}x{H}: " + " | ".join(row))

"""the discriminator's fused tail launches alone (csrc/d_tail.hip), under hipGraph replay: config 2's half (N = 12, 8 x 8 at block4) and the
configs[3] shard's (N = 10, 16 x 16)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd
from pytorch_tecogan_amd import kernels as K
from microbench import time_graph
dev = "cuda"
for dt in (torch.bfloat16,):
    for N, H4 in ((12, 8), (10, 16)):
        hw5 = (H4 // 2) ** 2
        z4 = torch.randn(N, H4, H4, 64, device=dev).to(dt)
        n4, dn4, dz4 = (torch.empty_like(z4) for _ in range(3))
        z5 = torch.empty(N, H4 // 2, H4 // 2, 32, device=dev, dtype=dt)
        n5, dz5 = torch.empty_like(z5), torch.empty_like(z5)
        stats = torch.zeros(4, 1, 2, 64, device=dev)
        stats[0, 0, 0] = z4.float().sum(dim=(0, 1, 2)); stats[0, 0, 1] = (z4.float() ** 2).sum(dim=(0, 1, 2))
        f = lambda n, v=0.0: torch.full((n,), v, device=dev)
        g4, b4, g5, b5 = f(64, 1.0), f(64), f(32, 1.0), f(32)
        rm4, rv4, rm5, rv5 = f(64), f(64, 1.0), f(32), f(32, 1.0)
        nbt4, nbt5 = torch.zeros((), dtype=torch.long, device=dev), torch.zeros((), dtype=torch.long, device=dev)
        s4, s5 = torch.zeros(2, 64, device=dev), torch.zeros(2, 32, device=dev)
        w5 = torch.randn(3, 64, 4, 4, device=dev) * 0.05
        fcw, fcb, prob, dl = torch.randn(1, 3 * hw5, device=dev) * 0.1, f(32), f(N), f(N, 0.01)
        cfg = torch.zeros(64, device=dev); cfg[6] = 1e-12
        ws = K.d_tail_scratch(N, H4, 1, dev)
        gfw, gfb, dg5, db5, dg4, db4 = torch.zeros_like(fcw), f(32), f(32), f(32), f(64), f(64)
        fwd = lambda: K.d_tail_fwd(z4, stats, 4, g4, b4, rm4, rv4, nbt4, s4, n4, w5, z5, g5, b5, rm5, rv5, nbt5, s5, n5, fcw, fcb, prob, N, H4, 3, 1, ws)
        bwd = lambda: K.d_tail_bwd(dl, prob, cfg, None, True, n5, z5, s5, g5, fcw, w5, n4, z4, s4, g4, dz5, dn4, dz4, gfw, gfb, dg5, db5, dg4, db4, N, H4, 3, 1, ws)
        print(f"{dt} N={N} block4 {H4}x{H4}: tg_d_tail_fwd {time_graph(fwd):6.1f} us   tg_d_tail_bwd {time_graph(bwd):6.1f} us")

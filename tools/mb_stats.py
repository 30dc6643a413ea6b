"""What the batch-norm statistics epilogue of a conv launch costs, and how much of it is atomic contention: the discriminator's
stage-1 residual conv (12 x 64x64, 64 -> 64) without statistics, with them (R = 1: every workgroup adds into the same 128
floats) and with R replica slots (tg_conv_desc.stats_replicas; no fold here - timing only).  hipGraph of 16 launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, kernels as K

dt = torch.bfloat16
dev = "cuda"


def timed(fn, n=16, reps=30):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            g.replay()
        e1.record(side)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


for name, N, H, C in (("D stage 1 (12 x 64x64 x 64)", 12, 64, 64), ("D stage 3 (12 x 16x16 x 128)", 12, 16, 128)):
    spec = K.ConvSpec("c3", C, C)
    rows, Kd, s_row, s_k = spec.fwd_pack()
    wp = K.pack_weights(dt, torch.randn(spec.weight_shape, device=dev) * 0.03, rows, Kd, s_row, s_k, 9, K.slot_table(9, dev))
    x = torch.randn(N, H, H, C, device=dev).to(dt)
    out = torch.empty_like(x)
    res = []
    for R in (0, 1, 4, 16):
        d = K.make_conv_desc(spec.fwd_geom(), L.TG_BF16, N, H, H, C, H, H, C, stats_mode=2 if R else 0, stats_groups=1,
                             stats_replicas=max(R, 1))
        stats = torch.zeros(max(R, 1) * 2 * C, device=dev)
        res.append((R, timed(lambda: K.conv(d, x, wp, out, stats=stats if R else None))))
    print(name, " | ".join(f"{'no stats' if R == 0 else f'R={R}'} {t:.2f} us" for R, t in res))
    # batch-norm backward reduce on the same tensor (atomics of every workgroup into 2*C floats)
    save = torch.zeros(2 * C, device=dev); save[C:] = 1.0
    for R in (1, 4):
        red = torch.zeros(R * 2 * C, device=dev)
        t = timed(lambda: K.bn_bwd_reduce(x, None, out, save, red, N, H * H, C, 1, L.ACT_NONE, replicas=R))
        print(f"   bn_bwd_reduce R={R} {t:.2f} us")

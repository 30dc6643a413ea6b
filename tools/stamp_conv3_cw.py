"""Diagnostic + microbenchmark of the eight-equal-waves 3x3 kernel (tg_conv3x3_cw, csrc/conv3_cw.hip) against the producer / consumer
kernel (tg_conv3x3_rw) on the step's 64-reduction-channel shapes: microseconds per launch under hipGraph replay for both, and - with a
-DTG_STAMP build (tools/build_variant.sh stamp conv3_cw -DTG_STAMP; TECOGAN_LIB=_ab/libtecogan_hip_stamp.so) - the phase stamps of
waves 0 and 4 of workgroup 0: barrier wait | DMA + rows + k-loop | epilogue (+ patch wait)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, kernels as K

lib = L.load()
stamps = hasattr(lib, "tg_debug_read_c3cw_stamps")
if stamps:
    lib.tg_debug_read_c3cw_stamps.restype = ctypes.c_int
    lib.tg_debug_read_c3cw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dt = torch.bfloat16
CASES = [  # name, cin, cout, N, H, W, flip, masked, stats, cap
    ("G trunk dgrad 64->64 @32 N40 cap144", 64, 64, 40, 32, 32, 1, True, False, 144),
    ("D s1 fwd+stats 64->64 @64 N12 cap80", 64, 64, 12, 64, 64, 0, False, True, 80),
    ("D s1 dgrad(lrelu) 64->64 @64 N12 cap96", 64, 64, 12, 64, 64, 1, True, False, 96),
    ("G c20 dgrad 64->64 @64 N40 cap144", 64, 64, 40, 64, 64, 1, True, False, 144),
    ("G c6 dgrad 64->128 @128 N40 cap144", 64, 128, 40, 128, 128, 1, True, False, 144),
    ("chain c20 fwd 64->64 @64 N4 cap160", 64, 64, 4, 64, 64, 0, False, False, 160),
    ("chain c30 fwd 64->128 @64 N4 cap160", 64, 128, 4, 64, 64, 0, False, False, 160),
    ("chain conv0 fwd 64->64 @32 N4 cap160", 64, 64, 4, 32, 32, 0, False, False, 160),
    ("infer c20 fwd 64->64 @256 N1 cap256", 64, 64, 1, 256, 256, 0, False, False, 256),
]


def time_graph(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 / n * 1e6


for name, cin, cout, N, H, W, flip, masked, stats, cap in CASES:
    x = torch.randn(N, H, W, cin, device="cuda").to(dt)
    out = torch.empty(N, H, W, cout, dtype=dt, device="cuda")
    mask = torch.randn(N, H, W, cout, device="cuda").to(dt) if masked else None
    st = torch.zeros(4, 2, 2, cout, device="cuda") if stats else None
    wp = (torch.randn(9 * cin * cout, device="cuda") * 0.05).to(dt)
    bias = None if masked else torch.zeros(cout, device="cuda")

    def run(cw):
        K.conv3x3_rw(x, wp, out, bool(flip), bias=bias, mask=mask, mask_mode=L.MASK_RELU if masked else L.MASK_NONE,
                     act=L.ACT_NONE if masked else L.ACT_LRELU, stats=st, groups=2 if stats else 1, stats_replicas=4 if stats else 1,
                     max_workgroups=cap, cw=cw)
    us = [time_graph(lambda: run(False)), time_graph(lambda: run(True))]
    gf = 2.0 * N * H * W * 9 * cin * cout / 1e3
    print(f"== {name}: producer/consumer {us[0]:6.1f} us ({gf / us[0] / 1e3:5.0f} TF/s) | eight equal waves {us[1]:6.1f} us ({gf / us[1] / 1e3:5.0f} TF/s)")
    if stamps:
        run(True); torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 64)()
        lib.tg_debug_read_c3cw_stamps(buf, 64)
        for role in (0, 1):
            t = list(buf)[role * 32:(role + 1) * 32]
            line = f"  wave {4 * role}: weights + patch issue {t[1]-t[0]} | fragments {t[2]-t[1]} ||"
            for i in range(6):
                b0 = 4 + 4 * i
                if t[b0] <= 0 or t[b0 + 3] <= t[b0]:
                    break
                line += f" tile{i}: barrier {t[b0+1]-t[b0]} dma+rows+k-loop {t[b0+2]-t[b0+1]} epilogue {t[b0+3]-t[b0+2]} = {t[b0+3]-t[b0]} |"
            print(line + f" total {t[28]-t[0]}")

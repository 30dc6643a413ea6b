"""Diagnostic: phase stamps (s_memtime ticks) of waves 0 (class 3) and 4 (class 0) of workgroup 0 of the class-waves conv-transpose
kernel (tg_convt_fwd_cw; csrc/convt_cw.hip built with -DTG_STAMP: tools/build_variant.sh stamp convt_cw -DTG_STAMP;
TECOGAN_LIB=_ab/libtecogan_hip_stamp.so).  Per tile: barrier wait | DMA issue of the next patch | k-loop | epilogue + stores + patch wait."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, kernels as K

lib = L.load()
lib.tg_debug_read_cw_stamps.restype = ctypes.c_int
lib.tg_debug_read_cw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dt = torch.bfloat16
for name, cin, cout, N, H, cap in (("ct0 64->64 N=1 128x128", 64, 64, 1, 128, 256), ("ct4 128->128 N=1 256x256", 128, 128, 1, 256, 256),
                                   ("ct4 128->128 N=4 64x64", 128, 128, 4, 64, 256), ("ct0 64->64 N=4 32x32", 64, 64, 4, 32, 256)):
    x = torch.randn(N, H, H, cin, device="cuda").to(dt)
    out = torch.empty(N, 2 * H, 2 * H, cout, dtype=dt, device="cuda")
    wp = (torch.randn(9 * cin * cout, device="cuda") * 0.05).to(dt)
    b = torch.zeros(cout, device="cuda")
    for _ in range(3):
        K.convt_fwd_cw(x, wp, b, out, L.ACT_RELU, max_workgroups=cap)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.tg_debug_read_cw_stamps(buf, 64)
    print(f"== {name}")
    for role in (0, 1):
        t = list(buf)[role * 32:(role + 1) * 32]
        line = f"  wave {4 * role} (class {3 if role == 0 else 0}): DMA issue {t[1]-t[0]} | weights issue {t[2]-t[1]} ||"
        for i in range(6):
            b0 = 4 + 4 * i
            if t[b0] <= 0 or t[b0 + 3] <= t[b0]:
                break
            line += f" tile{i}: barrier {t[b0+1]-t[b0]} dma+k-loop {t[b0+2]-t[b0+1]} epilogue+wait {t[b0+3]-t[b0+2]} = {t[b0+3]-t[b0]} |"
        print(line + f" total {t[28]-t[0]}")

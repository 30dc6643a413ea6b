"""Diagnostic: phase stamps (s_memtime ticks) of waves 0 and 4 of workgroup 0 of the register-weights 3x3 kernel (csrc/conv3_rw.hip
built with -DTG_STAMP: tools/build_variant.sh stamp conv3_rw -DTG_STAMP; TECOGAN_LIB=_ab/libtecogan_hip_stamp.so).
Per launch shape of the step: the prologue and, for the first iterations, the consumer's k-loop | accumulator store | barrier wait and
the producer's issue phase (stores of tile i - 2, DMA of tile i + 1, mask / residual rows of tile i) | epilogue arithmetic of tile i - 1 | vmcnt(0) | barrier wait.  (The round-3 kernel's stamps:
profiles/r04_b_stamp_rw_v1.log.)"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, kernels as K

lib = L.load()
lib.tg_debug_read_rw_stamps.restype = ctypes.c_int
lib.tg_debug_read_rw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dt = torch.bfloat16
CASES = [  # name, cin, cout, N, H, W, flip, masked, cap
    ("G trunk dgrad 64->64 @32 N40 cap160", 64, 64, 40, 32, 32, 1, True, 160),
    ("D s1 64->64 @64 N12 cap72", 64, 64, 12, 64, 64, 0, False, 72),
    ("G c20 dgrad 64->64 @64 N40 cap160", 64, 64, 40, 64, 64, 1, True, 160),
    ("G c6 dgrad 64->128 @128 N40 cap160", 64, 128, 40, 128, 128, 1, True, 160),
    ("G c32 dgrad 128->128 @64 N40 cap160", 128, 128, 40, 64, 64, 1, True, 160),
    ("D s2 128->128 @32 N12 cap96", 128, 128, 12, 32, 32, 0, False, 96),
]
for name, cin, cout, N, H, W, flip, masked, cap in CASES:
    x = torch.randn(N, H, W, cin, device="cuda").to(dt)
    out = torch.empty(N, H, W, cout, dtype=dt, device="cuda")
    mask = torch.randn(N, H, W, cout, device="cuda").to(dt) if masked else None
    wp = (torch.randn(9 * cin * cout, device="cuda") * 0.05).to(dt)
    run = lambda: K.conv3x3_rw(x, wp, out, bool(flip), mask=mask, mask_mode=L.MASK_RELU if masked else L.MASK_NONE,
                               act=L.ACT_NONE if masked else L.ACT_RELU, max_workgroups=cap)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib.tg_debug_read_rw_stamps(buf, 64)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 5 / 20 * 1e6
    print(f"== {name}: {us:.1f} us per launch (hipGraph replay), {2.0 * N * H * W * 9 * cin * cout / us / 1e6:.0f} TFLOP/s")
    # wave-specialised kernel (round 4): role 0 = consumer wave 0, role 1 = producer wave 4
    for role, names in ((0, ["k-loop", "acc->LDS", "barrier"]), (1, ["stores(i-2)+DMA(i+1)+rows(i) issue", "compute(i-1)", "vmcnt(0)", "barrier"])):
        t = list(buf)[role * 32:(role + 1) * 32]
        line = f"  {'consumer' if role == 0 else 'producer'}: prologue {t[1]-t[0]} | {t[2]-t[1]} | first barrier {t[3]-t[2]} ||"
        for i in range(4):
            b = 4 + 6 * i
            n = len(names)
            if t[b + n] <= t[b]:
                break
            line += f" it{i}: " + " ".join(f"{nm} {t[b+j+1]-t[b+j]}" for j, nm in enumerate(names)) + f" = {t[b+n]-t[b]}" + \
                (f" (of compute: acc reads landed after {t[b+5]-t[b+1]})" if role == 1 and t[b + 5] > t[b + 1] else "") + " |"
        print(line + f" total {t[28]-t[0]}")

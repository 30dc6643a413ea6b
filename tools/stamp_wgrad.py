"""Diagnostic: where workgroup 0 of a weight-gradient launch spends its cycles (needs a library built with -DTG_STAMP:
TECOGAN_LIB=<that library> python tools/stamp_wgrad.py)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, kernels as K
lib = L.load()
lib.tg_debug_wgrad_stamps.restype = ctypes.c_int
lib.tg_debug_wgrad_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dt = torch.bfloat16
for (kind, cin, cout, N, H) in (("c3", 128, 64, 40, 128), ("c3", 128, 128, 40, 64), ("c3", 64, 64, 40, 64), ("c3", 64, 64, 12, 64)):
    spec = K.ConvSpec(kind, cin, cout)
    x_is_in, S, taps, ca, cb, s_a, s_b = spec.wgrad_info()
    OH, OW = spec.out_hw(H, H)
    X = torch.randn(N, H, H, K.pad32(cin), device="cuda").to(dt)
    Y = torch.randn(N, OH, OW, K.pad32(cout), device="cuda").to(dt)
    nsplit, tpw = K.wgrad_plan(N, OH, OW, S, len(taps), X.shape[3], Y.shape[3])
    desc = K.make_wgrad_desc(K.tg_dtype(dt), N, H, H, X.shape[3], OH, OW, Y.shape[3], S, taps, nsplit, tpw)
    slab = torch.empty(lib.tg_wgrad_slab_floats(ctypes.byref(desc)), device="cuda")
    for _ in range(2):
        K.wgrad(desc, X, Y, slab)
    torch.cuda.synchronize()
    lib.tg_debug_wgrad_stamps(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.wgrad(desc, X, Y, slab); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 8)()
    lib.tg_debug_wgrad_stamps(buf, 0)
    t = list(buf)
    n = max(1, t[5])
    names = ["barrier 1 (wait for readers)", "LDS stores (+ wait for this tile's loads)", "barrier 2", "issue next tile's loads", "k-loop (MFMA)"]
    print(f"{kind} {cin}->{cout} N={N} {H}x{H} nsplit={nsplit} tpw={tpw}: {e0.elapsed_time(e1)*1e3:.1f} us, {n} tiles in workgroup 0; s_memtime ticks per tile (100 MHz):")
    for i, nm in enumerate(names):
        print(f"    {nm:44s} {t[i] / n:8.1f}")
    print(f"    {'sum':44s} {sum(t[:5]) / n:8.1f}")

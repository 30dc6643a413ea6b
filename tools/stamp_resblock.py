"""Diagnostic: phase stamps of workgroup 0 of the fused residual-block launch (library built with build.sh -DTG_STAMP).
Runs the 16-block trunk as the recurrent pass does (distinct weights/buffers per block) and prints the last block's stamps."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd
from pytorch_tecogan_amd import _lib as L, kernels as K
lib = L.load()
PAIR = os.environ.get("RB_PAIR") == "1"   # two blocks per launch (csrc/exp/resblock2_ws.hip)
WS = os.environ.get("RB_WS", "1") == "1"   # round 5: the wave-specialised kernel (csrc/resblock_ws.hip); RB_WS=0: resblock.hip
STAMPS = hasattr(lib, "tg_debug_read_rb_stamps") and not WS   # (a library built without -DTG_STAMP: only the launch time at the end)
WSTAMPS = WS and hasattr(lib, "tg_debug_read_rbw_stamps")
N_, H_ = int(os.environ.get("RB_N", "4")), int(os.environ.get("RB_H", "32"))
if STAMPS:
    lib.tg_debug_read_rb_stamps.restype = ctypes.c_int
    lib.tg_debug_read_rb_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dt = torch.bfloat16
N, H, NB = N_, H_, 16
spec = K.ConvSpec("c3", 64, 64)
rows, Kd, s_row, s_k = spec.fwd_pack()
slots = K.slot_table(9, "cuda")
wps = [[K.pack_weights(dt, torch.randn(spec.weight_shape, device="cuda") * 0.03, rows, Kd, s_row, s_k, 9, slots) for _ in range(2)] for _ in range(NB)]
if os.environ.get("RB_SAMEW") == "1":   # diagnostic: every block streams the SAME 147 KB (they stay in the XCDs' L2s)
    wps = [wps[0]] * NB
bs = [torch.zeros(64, device="cuda") for _ in range(NB)]
a = [torch.randn(N, H, H, 64, device="cuda").to(dt) for _ in range(NB + 1)]
h = [torch.empty(N, H, H, 64, dtype=dt, device="cuda") for _ in range(NB)]
names = ["issue loads", "addr+wait+LDS stores", "barrier", "conv1 MFMA", "epilogue1", "barrier", "conv2 prologue", "conv2 MFMA", "epilogue2"]
for rep in range(4 if STAMPS else 0):
    for i in range(NB):
        K.resblock_fwd(a[i], wps[i][0], bs[i], wps[i][1], h[i], a[i + 1], next_w=(wps[i + 1] if i + 1 < NB and os.environ.get('RB_PREFETCH', '0') == '1' else None))
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    lib.tg_debug_read_rb_stamps(buf, 16)
    t = list(buf)
    print(" | ".join(f"{n} {t[i+1]-t[i]}" for i, n in enumerate(names)), "| total", t[9] - t[0], "ticks;",
          f"epilogue1 = exchange writes {t[10]-t[4]} + barrier {t[11]-t[10]} + finalise {t[5]-t[11]}")
    if hasattr(lib, "tg_debug_read_rb_wstamps") and rep == 3:
        wb = (ctypes.c_longlong * 40)()
        lib.tg_debug_read_rb_wstamps(wb, 40)
        w = list(wb)
        t0 = min(w[k * 5] for k in range(8))
        for k in range(8):
            a_ = w[k * 5:k * 5 + 5]
            print(f"   wave {k} (row tile {k & 3}, K half {k >> 2}): start +{a_[0]-t0}, conv1 begins +{a_[1]-t0}, conv1 done +{a_[2]-t0} (k-loop {a_[2]-a_[1]}), "
                  f"exchange barrier passed +{a_[3]-t0} (waited {a_[3]-a_[2]}), conv2 done +{a_[4]-t0}")

if PAIR and hasattr(lib, "tg_debug_read_rbw2_stamps"):
    lib.tg_debug_read_rbw2_stamps.restype = ctypes.c_int
    lib.tg_debug_read_rbw2_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for rep in range(2):
        for i in range(0, NB, 2):
            K.resblock2_fwd_ws(a[i], wps[i][0], bs[i], wps[i][1], wps[i + 1][0], bs[i + 1], wps[i + 1][1], h[i], a[i + 1], h[i + 1], a[i + 2])
        torch.cuda.synchronize()
        wb = (ctypes.c_longlong * 96)()
        lib.tg_debug_read_rbw2_stamps(wb, 96)
        w = list(wb)
        t0 = min(w[k * 12] for k in range(8))
        names1 = ["start", "DMA issued", "conv1A done", "[H1] passed", "h1 stored, W1b issued", "conv1B done", "[H2] passed", "end"]
        names2 = ["start", "DMA issued", "[S2A] passed, W2a issued", "conv2A loop done", "a1 written", "[S2B] passed", "conv2B loop done", "end"]
        print(f"-- pass {rep} (last pair's workgroup 0, ticks after the first wave's start)")
        for k in range(8):
            s_ = [x - t0 for x in w[k * 12:k * 12 + 8]]
            print(f"   {'conv1' if k < 4 else 'conv2'} wave {k}: " + ", ".join(f"{n_} +{v}" for n_, v in zip(names1 if k < 4 else names2, s_)))
            e_ = [x - t0 for x in w[k * 12 + 8:k * 12 + 10]]
            print(f"        " + (f"a1 seen complete +{e_[0]}" if k < 4 else f"all through with h1 +{e_[0]}, exchange A met +{e_[1]}"))
elif WSTAMPS:
    # [0 start | 1 DMA issued | 2 first stage passed | 3 conv1 done | 4 h barrier passed | 5 conv2 done | 6 end] per wave of workgroup 0
    lib.tg_debug_read_rbw_stamps.restype = ctypes.c_int
    lib.tg_debug_read_rbw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for rep in range(3):
        for i in range(NB):
            K.resblock_fwd(a[i], wps[i][0], bs[i], wps[i][1], h[i], a[i + 1], ws=True)
        torch.cuda.synchronize()
        wb = (ctypes.c_longlong * 64)()
        lib.tg_debug_read_rbw_stamps(wb, 64)
        w = list(wb)
        t0 = min(w[k * 8] for k in range(8))
        print(f"-- pass {rep} (last block's workgroup 0, ticks after the first wave's start)")
        for k in range(8):
            s_ = [x - t0 for x in w[k * 8:k * 8 + 7]]
            if k < 4:
                print(f"   conv1 wave {k}: start +{s_[0]}, DMA issued +{s_[1]}, stage 0 passed +{s_[2]}, conv1 done +{s_[3]}, "
                      f"h barrier passed +{s_[4]}, h stored +{s_[5]}, end +{s_[6]}")
            else:
                print(f"   conv2 wave {k}: start +{s_[0]}, DMA issued +{s_[1]}, stage 0 passed +{s_[2]}, last stage passed +{s_[3]}, "
                      f"h barrier passed +{s_[4]}, conv2 done +{s_[5]}, end +{s_[6]}")

# wall time per launch of the same 16-launch trunk replayed as a hipGraph (what the step does; eager launches are host-bound)
def trunk():
    if PAIR:
        for i in range(0, NB, 2):
            K.resblock2_fwd_ws(a[i], wps[i][0], bs[i], wps[i][1], wps[i + 1][0], bs[i + 1], wps[i + 1][1], h[i], a[i + 1], h[i + 1], a[i + 2])
        return
    for i in range(NB):
        if WS:
            K.resblock_fwd(a[i], wps[i][0], bs[i], wps[i][1], h[i], a[i + 1], ws=True)
            continue
        K.resblock_fwd(a[i], wps[i][0], bs[i], wps[i][1], h[i], a[i + 1], next_w=(wps[i + 1] if i + 1 < NB and os.environ.get('RB_PREFETCH', '0') == '1' else None))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        trunk()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(side)
    for rep in range(50):
        g.replay()
    e1.record(side)
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 1e3 / (50 * NB):.2f} us per launch (hipGraph replay of the 16-launch trunk, N = {N}, {H} x {H}, "
      f"{'resblock2_ws.hip: PER BLOCK, two per launch' if PAIR else 'resblock_ws.hip' if WS else 'resblock.hip'})")

"""Diagnostic: per-tile stamps of the persistent, tile-pipelined backward block (csrc/exp/resblock_pp.hip; library built with -DTG_STAMP)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, kernels as K
lib = L.load()
dt = torch.bfloat16
N, H = [int(v) for v in os.environ.get("MB_SHAPE", "32,64").split(",")]
cap = int(os.environ.get("MB_CAP", "144"))
spec = K.ConvSpec("c3", 64, 64)
rows, Kd, s_row, s_k = spec.dgrad_pack()
slots = K.slot_table(9, "cuda")
wb = [K.pack_weights(dt, torch.randn(spec.weight_shape, device="cuda") * 0.05, rows, Kd, s_row, s_k, 9, slots) for _ in range(2)]
dA1 = torch.randn(N, H, H, 64, device="cuda").to(dt)
hf = torch.randn(N, H, H, 64, device="cuda").clamp_min(0).to(dt)
dH = torch.empty_like(dA1); dA0 = torch.empty_like(dA1)
for _ in range(3):
    K.resblock_bwd_pp(dA1, wb[1], hf, wb[0], dH, dA0, max_workgroups=cap)
torch.cuda.synchronize()
lib.tg_debug_read_rbp_stamps.restype = ctypes.c_int
lib.tg_debug_read_rbp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_longlong * 48)()
lib.tg_debug_read_rbp_stamps(buf, 48)
w = list(buf)
t0 = min(v for v in w if v > 0)
n1 = ["top", "patch seen", "h buffer free", "k-loop done", "mask landed", "h written + signalled", "dH stored"]
n2 = ["top", "epi seen + DMA issued", "h seen", "k-loop done", "exchange met", "patch i+2 landed", "end"]
for role, names in ((0, n1), (1, n2)):
    for q in range(3):
        s_ = w[role * 24 + q * 8: role * 24 + q * 8 + 7]
        print(f"{'conv1 wave 0' if role == 0 else 'conv2 wave 4'} tile {q + 2}: " + ", ".join(f"{n_} +{v - t0}" for n_, v in zip(names, s_)))

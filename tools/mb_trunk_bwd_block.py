"""The trunk's input-gradient chain of the batched generator backward, block by block as the step runs it (16 blocks, each with its
own weights / mask / buffers), hipGraph replay: two register-weights launches per block (round 4) against ONE persistent,
tile-pipelined launch (csrc/exp/resblock_pp.hip, round 5), at several workgroup caps.  MB_SHAPE="N,H" (default 40,32: config 2)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import _lib as L, kernels as K
from microbench import time_graph
DEV = "cuda:0"; dt = torch.bfloat16
N, H = [int(v) for v in os.environ.get("MB_SHAPE", "40,32").split(",")]
NB = 16
spec = K.ConvSpec("c3", 64, 64)
rows, Kd, s_row, s_k = spec.dgrad_pack()
slots = K.slot_table(9, DEV)
wb = [[K.pack_weights(dt, torch.randn(spec.weight_shape, device=DEV) * 0.05, rows, Kd, s_row, s_k, 9, slots) for _ in range(2)] for _ in range(NB)]
dA = [torch.randn(N, H, H, 64, device=DEV).to(dt) for _ in range(NB + 1)]
dH = [torch.empty(N, H, H, 64, dtype=dt, device=DEV) for _ in range(NB)]
hf = [torch.randn(N, H, H, 64, device=DEV).clamp_min(0).to(dt) for _ in range(NB)]
for cap in (144, 160, 256):
    def two():
        for i in range(NB):
            K.conv3x3_rw(dA[i + 1], wb[i][1], dH[i], True, mask=hf[i], mask_mode=L.MASK_RELU, max_workgroups=cap)
            K.conv3x3_rw(dH[i], wb[i][0], dA[i], True, res=dA[i + 1], max_workgroups=cap)
    def one():
        for i in range(NB):
            K.resblock_bwd_pp(dA[i + 1], wb[i][1], hf[i], wb[i][0], dH[i], dA[i], max_workgroups=cap)
    t2 = time_graph(two, reps=1, iters=20) / NB
    t1 = time_graph(one, reps=1, iters=20) / NB
    print(f"N={N} {H}x{H} cap {cap}: two register-weights launches {t2:6.1f} us per block | one persistent pipelined launch {t1:6.1f} us per block", flush=True)

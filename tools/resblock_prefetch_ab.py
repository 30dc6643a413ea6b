"""A/B: the fused residual block's L2 prefetch of the NEXT block's weights - on (shipped) / off - in the step:
chain alone and the whole step.  usage: python tools/resblock_prefetch_ab.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench as B
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import kernels as K, train as TR

orig = K.resblock_fwd
dev = torch.device("cuda", 0)
for mode in ("prefetch on", "prefetch off", "prefetch on", "prefetch off"):
    K.resblock_fwd = orig if mode.endswith("on") else (lambda x, w1, b1, w2, h, a, next_w=None, skip=True: orig(x, w1, b1, w2, h, a, next_w=None, skip=skip))
    for s_ in list(TR._STEPS.values()):
        s_.close()
    TR._STEPS.clear()
    os.environ["TECOGAN_GRAPH"] = "1"
    args = B.default_args("bf16")
    torch.manual_seed(1)
    G, D, og, od = B.build_step_objects(args, dev)
    x, y = B.synth(4, 10, 32, 1); x, y = x.to(dev), y.to(dev)
    for s in range(4):
        TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
    torch.cuda.synchronize()
    st = next(iter(TR._STEPS.values()))
    def t(fn, n=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    chain = t(lambda: (st.graphs["chain0"](), st.graphs["chain"](), st.graphs["chain_tail"]()))
    k = [4]
    def step():
        TR.FRVSR_Train(x, y, args, D, G, k[0], 0., 0., og, od); k[0] += 1
    print(f"{mode:13s}: chain0 + chain + tail alone {chain:.3f} ms | whole step {t(step, 60):.3f} ms", flush=True)

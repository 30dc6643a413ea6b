"""What slows the recurrent chain down when the discriminator's real half runs beside it?  The chain graphs replayed
  (a) alone, (b) beside the real half (prep + d_real), (c) beside a stream of EMPTY launches (one-element fills, as many as the
real half has: kernel boundaries - packet processing, cache write-back / invalidate - without any work), (d) beside ONE long
memory-streaming kernel sequence (large copies: HBM / L2 pressure with few boundaries)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import models as M, train as TR
import bench as B

args = B.default_args("bf16"); torch.manual_seed(1); dev = torch.device("cuda", 0)
G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
og = torch.optim.Adam(G.parameters(), 1e-4); od = torch.optim.Adam(D.parameters(), 1e-4)
x, y = B.synth(4, 10, 32, 1); x, y = x.to(dev), y.to(dev)
os.environ["TECOGAN_GRAPH"] = "1"
for s in range(3):
    TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
torch.cuda.synchronize()
st = next(iter(TR._STEPS.values())); g = st.graphs
side = torch.cuda.Stream()


def graph_of(fn):
    with torch.cuda.stream(side):
        fn(); gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            fn()
    torch.cuda.synchronize()
    return gr


one = torch.zeros(64, device=dev)
empty = graph_of(lambda: [one.fill_(1.0) for _ in range(125)])
big_a, big_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)   # 256 MB each
stream8 = graph_of(lambda: [big_b.copy_(big_a) for _ in range(8)])                      # 8 x 512 MB of traffic
writes8 = graph_of(lambda: [big_b.fill_(1.0) for _ in range(16)])                       # 16 x 256 MB written, nothing read (round 5)
acc_r = torch.zeros(1, device=dev)
reads8 = graph_of(lambda: [torch.sum(big_a, dim=0, keepdim=True, out=acc_r) for _ in range(16)])   # 16 x 256 MB read, nothing written
small_w = torch.empty(1 << 20, device=dev)                                              # 4 MB: stays in the L2s
writes_small = graph_of(lambda: [small_w.fill_(1.0) for _ in range(125)])               # 125 launches that dirty 4 MB each
compute = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
gemm = graph_of(lambda: [torch.mm(compute, compute) for _ in range(3)])


def run(label, beside):
    def once():
        ev = torch.cuda.Event(); ev.record()
        if beside is not None:
            side.wait_event(ev)
            with torch.cuda.stream(side):
                beside()
        g["chain0"](); g["chain"]()
    once(); torch.cuda.synchronize()
    t = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); once(); e1.record(); torch.cuda.synchronize()   # e1: the chain's end only
        t.append(e0.elapsed_time(e1))
    print(f"{label:58s} chain {sum(t) / len(t):.3f} ms")


def dreal():
    with torch.cuda.stream(st.sBm):
        pass
    g["prep"](); g["d_real"]()


run("chain alone", None)
run("chain || real half of D (prep + d_real graphs)", dreal)
run("chain || 125 empty launches", empty.replay)
run("chain || 8 x 256-MB device copies (4 GB of traffic)", stream8.replay)
run("chain || 16 x 256-MB fills (writes only)", writes8.replay)
run("chain || 16 x 256-MB sums (reads only)", reads8.replay)
run("chain || 125 x 4-MB fills (dirty lines, many boundaries)", writes_small.replay)
run("chain || 3 bf16 GEMMs 8192^3 (dense MFMA, few launches)", gemm.replay)

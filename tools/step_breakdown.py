"""True cost of the step's parts under hipGraph replay (the profiler serialises the graph's concurrent branches):
captures the forward/backward segment with selected parts enabled and times the replay."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd
from pytorch_tecogan_amd import models as M, train as TR
import bench as B

def main():
    args = B.default_args("bf16")
    torch.manual_seed(1)
    dev = torch.device("cuda", 0)
    G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
    og = torch.optim.Adam(G.parameters(), 1e-4); od = torch.optim.Adam(D.parameters(), 1e-4)
    x, y = B.synth(4, 10, 32, 1); x, y = x.to(dev), y.to(dev)
    os.environ["TECOGAN_GRAPH"] = "0"
    for s in range(2):
        TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
    torch.cuda.synchronize()
    st = next(iter(TR._STEPS.values()))
    combos = [("chain",), ("chain", "dreal"), ("dreal",), ("chain", "gbwd"), ("gbwd",), ("dfake",), ("dfake", "dbwd"),
              ("dbwd",), ("chain", "dreal", "dfake"), ("chain", "dreal", "gbwd", "dfake", "dbwd")]
    # gbwd alone needs dpre etc. from an earlier full run: buffers hold valid data from the warm-up steps
    for parts in combos:
        pset = set(parts)
        g = torch.cuda.CUDAGraph()
        st._forward_backward(True, parts=pset); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            st._forward_backward(True, parts=pset)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print(f"{'+'.join(parts):40s} {(time.perf_counter()-t0)/10*1e3:7.3f} ms", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st._update()
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); print(f"{'update (2x Adam + 2x repack)':40s} {(time.perf_counter()-t0)/10*1e3:7.3f} ms")

if __name__ == "__main__":
    main()

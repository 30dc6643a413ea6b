"""Generator-only recurrent inference (main.py:171-219 of the reference) on device: BASELINE configs 1 and 5.
    python tools/bench_inference.py [--lr 32|128] [--frames T] [--dtype bf16|fp32] [--no-graph]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd
from pytorch_tecogan_amd import models as M

ap = argparse.ArgumentParser(); ap.add_argument("--lr", type=int, default=128); ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--no-graph", action="store_true"); a = ap.parse_args()
args = argparse.Namespace(num_resblock=16, tg_dtype=a.dtype)
torch.manual_seed(1)
G = M.generator(3, args).cuda()
x = torch.from_numpy(np.random.default_rng(1).random((1, a.frames, 3, a.lr, a.lr), dtype=np.float32)).cuda()
out = G.recurrent(x, use_graph=not a.no_graph); torch.cuda.synchronize()
for _ in range(2): out = G.recurrent(x, use_graph=not a.no_graph)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n): out = G.recurrent(x, use_graph=not a.no_graph)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
gflop = 8.648 * (a.lr / 32) ** 2
print(json.dumps({"metric": "HR frames/sec, generator-only recurrent inference", "lr": a.lr, "hr": 4 * a.lr, "frames": a.frames,
                  "value": round(a.frames / dt, 1), "ms_per_frame": round(dt / a.frames * 1e3, 4), "dtype": a.dtype,
                  "hipgraph": not a.no_graph, "tflops": round(gflop * a.frames / dt / 1e3, 1), "finite": bool(torch.isfinite(out).all())}))

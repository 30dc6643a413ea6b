"""Where does a data-parallel step spend HOST time?  One rank with the real RCCL process group (TECOGAN_FORCE_COLLECTIVES=1):
host seconds inside the all-reduce calls / the waits / the whole FRVSR_Train call, beside the GPU time per step.
    TECOGAN_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port 29578 tools/dp_host_probe.py
"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import models as M, train as TR, step as S  # noqa: E402
import bench as B  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if "RANK" in os.environ:
    dist.init_process_group("nccl", device_id=dev)
args = B.default_args("bf16")
torch.manual_seed(1)
G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
og, od = torch.optim.Adam(G.parameters(), 1e-4), torch.optim.Adam(D.parameters(), 1e-4)
x, y = B.synth(4, 10, 32, 1)
x, y = x.to(dev), y.to(dev)
os.environ["TECOGAN_GRAPH"] = "1"
for s in range(4):
    TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
torch.cuda.synchronize()
st = next(iter(TR._STEPS.values()))
acc = {"allreduce": 0.0, "wait": 0.0, "n_ar": 0}
orig = st._allreduce


class TimedWork:
    def __init__(self, w):
        self.w = w

    def wait(self):
        t = time.perf_counter()
        self.w.wait()
        acc["wait"] += time.perf_counter() - t


def timed(buf):
    t = time.perf_counter()
    w = orig(buf)
    acc["allreduce"] += time.perf_counter() - t
    acc["n_ar"] += 1
    return TimedWork(w) if w is not None else None


st._allreduce = timed
N = 30
host = 0.0
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(N):
    t = time.perf_counter()
    TR.FRVSR_Train(x, y, args, D, G, 4 + s, 0., 0., og, od)
    host += time.perf_counter() - t
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"process group: {dist.is_initialized()}, buckets: {st.buckets}; per step: wall {t_all / N * 1e3:.3f} ms, host time inside FRVSR_Train "
      f"{host / N * 1e3:.3f} ms (issue loop {t_issue / N * 1e3:.3f}), of it all-reduce calls {acc['allreduce'] / N * 1e3:.3f} ms "
      f"({acc['n_ar'] // N} per step), Work.wait() {acc['wait'] / N * 1e3:.3f} ms")
# the same step with a host synchronisation after every step (GPU time per step without any run-ahead)
t0 = time.perf_counter()
for s in range(N):
    TR.FRVSR_Train(x, y, args, D, G, 40 + s, 0., 0., og, od)
    torch.cuda.synchronize()
print(f"synchronised after every step: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per step")
if dist.is_initialized():
    dist.destroy_process_group()

"""Does a step that REPLACES another on the same engines (train.get_step: new batch size / data-parallel mode) run as fast as a
first one?  usage: python tools/rebuild_probe.py [mode sequence, e.g. inline,buckets,inline] (under torch.distributed.run with
TECOGAN_FORCE_COLLECTIVES=1 for the data-parallel modes; without a process group the same configuration is simply rebuilt)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
import pytorch_tecogan_amd  # noqa: F401
from pytorch_tecogan_amd import train as TR

modes = (sys.argv[1] if len(sys.argv) > 1 else "x,x,x").split(",")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if "RANK" in os.environ:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)
os.environ["TECOGAN_GRAPH"] = "1"
args = bench.default_args("bf16")
torch.manual_seed(1)
G, D, og, od = bench.build_step_objects(args, dev)
x, y = bench.synth(4, 10, 32, 1)
x, y = x.to(dev), y.to(dev)
step = 0
for m in modes:
    if m in ("inline", "buckets"):
        os.environ["TECOGAN_DP_INLINE"] = "1" if m == "inline" else "0"
    for s_ in list(TR._STEPS.values()):
        s_.close()
    TR._STEPS.clear()
    for _ in range(4):
        TR.FRVSR_Train(x, y, args, D, G, step, 0.0, 0.0, og, od); step += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        TR.FRVSR_Train(x, y, args, D, G, step, 0.0, 0.0, og, od); step += 1
    torch.cuda.synchronize()
    st = next(iter(TR._STEPS.values()))
    print(f"mode {m}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step  (dp_inline={st.dp_inline} buckets={st.buckets} graphs={'yes' if st.graphs else 'no'} "
          f"sB={st.sB.cuda_stream:#x})", flush=True)

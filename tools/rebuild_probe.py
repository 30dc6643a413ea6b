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
    # anatomy: every captured piece alone (serial replays on the caller's stream), then the step with its collectives skipped
    if st.graphs and isinstance(st.graphs, dict):
        parts = []
        for k, rp in st.graphs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                rp()
            torch.cuda.synchronize()
            parts.append(f"{k}={(time.perf_counter() - t0) / 10 * 1e3:.3f}")
        print("   pieces alone (ms): " + " ".join(parts), flush=True)
        st.skip_collectives = True
        for _ in range(2):
            TR.FRVSR_Train(x, y, args, D, G, step, 0.0, 0.0, og, od); step += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            TR.FRVSR_Train(x, y, args, D, G, step, 0.0, 0.0, og, od); step += 1
        torch.cuda.synchronize()
        print(f"   collectives skipped: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
        st.skip_collectives = False
    if os.environ.get("PROBE_PROFILE") == m:   # where does the host spend a slow mode's step?
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            TR.FRVSR_Train(x, y, args, D, G, step, 0.0, 0.0, og, od); step += 1
        t_issue = time.perf_counter()
        torch.cuda.synchronize()
        pr.disable()
        print(f"   host: 10 steps issued, then {1e3 * (time.perf_counter() - t_issue):.2f} ms until the GPU drained", flush=True)
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

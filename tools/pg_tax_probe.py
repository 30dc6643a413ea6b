"""What does a process group cost the step when it has nothing to reduce (one rank)?  VERDICT r4 item 3: 3.92-3.96 ms with a one-rank
RCCL group and the collectives SKIPPED against 3.72 without a group (profiles/r04_w_rebuild_probe.log), unexplained; and a bucket-mode
step built after an inline one replayed at 13.4 ms.  Every variant runs in a process of its own (the runtime's stream -> hardware
queue mapping and the backend's threads are per process):

    python tools/pg_tax_probe.py            # runs all variants as children and prints one line each
    python tools/pg_tax_probe.py <variant>  # one variant in this process

variants: none | gloo | nccl | nccl_forced | nccl_forced_quiet | seq_bi | seq_ib | seq_ib_nowarm (| seq_ib_q16 | seq_ib_q24 with _nowarm semantics)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ["none", "gloo", "nccl", "nccl_forced", "nccl_forced_quiet", "seq_bi", "seq_ib", "seq_ib_nowarm"]


def child(v):
    env = os.environ
    env.setdefault("GPU_MAX_HW_QUEUES", "16" if v.endswith("q16") else "24" if v.endswith("q24") else "8")
    backend = None if v == "none" else "gloo" if v == "gloo" else "nccl"
    if v in ("seq_ib_q16", "seq_ib_q24"):
        os.environ["PG_NOWARM"] = "1"
    if v.startswith("nccl_forced") or v.startswith("seq_"):
        env["TECOGAN_FORCE_COLLECTIVES"] = "1"
    if v == "nccl_forced_quiet":   # the backend's watchdog / monitor threads off
        env.update(TORCH_NCCL_ENABLE_MONITORING="0", TORCH_NCCL_ASYNC_ERROR_HANDLING="0", TORCH_NCCL_DUMP_ON_TIMEOUT="0",
                   TORCH_NCCL_ENABLE_TIMING="0", TORCH_NCCL_DESYNC_DEBUG="0")
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import pytorch_tecogan_amd  # noqa: F401
    from pytorch_tecogan_amd import train as TR
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if backend:
        import torch.distributed as dist
        env.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
        env.setdefault("MASTER_PORT", "29617")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        if v == "seq_ib_nowarm" or os.environ.get("PG_NOWARM"):   # (round 5: parallel.warm_backend uses the asynchronous path once before the first step; this
            from pytorch_tecogan_amd import parallel   # variant switches it off to show the 13-ms replay it cures)
            parallel._WARMED.set(dist.group.WORLD, True)
    env["TECOGAN_GRAPH"] = "1"
    args = bench.default_args("bf16")
    torch.manual_seed(1)
    G, D, og, od = bench.build_step_objects(args, dev)
    x, y = bench.synth(4, 10, 32, 1)
    x, y = x.to(dev), y.to(dev)
    step = 0

    def timed(n):
        nonlocal step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            TR.FRVSR_Train(x, y, args, D, G, step, 0.0, 0.0, og, od)
            step += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    modes = {"seq_bi": ("buckets", "inline")}.get(v, ("inline", "buckets") if v.startswith("seq_") else ("x",))
    out = []
    for m in modes:
        if m in ("inline", "buckets"):
            env["TECOGAN_DP_INLINE"] = "1" if m == "inline" else "0"
        for s_ in list(TR._STEPS.values()):
            s_.close()
        TR._STEPS.clear()
        timed(4)
        ms = min(timed(30), timed(30))
        st = next(iter(TR._STEPS.values()))
        s = f"{m if m != 'x' else 'step'} {ms:.3f} ms"
        if st.pg is not None:
            st.skip_collectives = True
            timed(3)
            s += f" (collectives skipped {min(timed(30), timed(30)):.3f})"
            st.skip_collectives = False
        out.append(s)
    print(f"{v:18s} GPU_MAX_HW_QUEUES={env['GPU_MAX_HW_QUEUES']:2s} world={'-' if st.pg is None else st.world} pg={'yes' if st.pg is not None else 'no '}: "
          + " | then ".join(out), flush=True)
    if backend:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for i, v in enumerate(VARIANTS * 2):   # twice: boxes drift by ~1 %
            env = dict(os.environ, MASTER_PORT=str(29617 + i), HSA_ENABLE_IPC_MODE_LEGACY="0")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), v], env=env, capture_output=True, text=True, timeout=300)
            lines = [ln for ln in r.stdout.splitlines() if "GPU_MAX_HW_QUEUES" in ln]
            print(lines[-1] if lines else f"{v}: FAILED rc={r.returncode} {r.stderr[-400:]}", flush=True)

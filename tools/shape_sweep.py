"""Do the scheduling heuristics (workgroup caps, routing thresholds of pytorch-tecogan_amd/tuning.py) generalise beyond the three shapes
they were tuned on (configs[1], the configs[3] shard, configs[4])?  VERDICT r4 item 6.

For every shape B x crop (T = 10; crop != 32 needs the tg_extend shapes) the training step is built and timed with the DEFAULT knobs and
with the caps of the persistent launches scaled (x 0.75, x 1.25) or lifted (one workgroup per CU); the regret of the default is
(default - best) / best.  One process per measurement (a Tuning object is parsed per process; steps of other settings must not share
engines).  Prints one table row per shape and a summary line.

    python tools/shape_sweep.py                       # the full sweep (bf16; fp16 at the diagonal)
    python tools/shape_sweep.py --one B crop dtype variant   # one measurement in this process (used by the sweep itself)
    python tools/shape_sweep.py --shapes 1x32,3x48,8x64      # a subset"""
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ("default", "x0.75", "x1.25", "uncapped")


def variant_env(variant, lr_pixels):
    """environment overrides of a variant, relative to what tuning.py would pick for a step of `lr_pixels` LR pixels per pass"""
    if variant == "default":
        return {}
    if variant == "uncapped":
        return {"TECOGAN_PERSIST_WGS": "256"}
    f = float(variant[1:])
    chain_bound = lr_pixels <= 4096
    g, gf, d, dr = (144, 160, 96, 96) if chain_bound else (160, 160, 96, 96)
    r8 = lambda v: str(max(8, int(round(v * f / 8.0)) * 8))  # noqa: E731
    return {"TECOGAN_PERSIST_WGS_G": r8(g), "TECOGAN_PERSIST_FWD_G": r8(gf), "TECOGAN_PERSIST_WGS_D": r8(d), "TECOGAN_PERSIST_WGS_DREAL": r8(dr)}


def one(B, crop, dtype, variant, steps=12):
    os.environ.update(variant_env(variant, B * crop * crop))
    os.environ["TECOGAN_GRAPH"] = "1"
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import pytorch_tecogan_amd  # noqa: F401
    from pytorch_tecogan_amd import train as TR
    dev = torch.device("cuda", 0)
    args = bench.default_args(dtype, T=10, cs=crop, extend=crop != 32)
    torch.manual_seed(1)
    G, D, og, od = bench.build_step_objects(args, dev)
    x, y = bench.synth(B, 10, crop, 1)
    x, y = x.to(dev), y.to(dev)
    step = 0

    def run(n):
        nonlocal step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = TR.FRVSR_Train(x, y, args, D, G, step, 0.0, 0.0, og, od)
            step += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out
    run(3)
    ms = min(run(steps)[0], run(steps)[0])
    _, out = run(1)
    finite = bool(torch.isfinite(out.gen_output).all()) and all(bool(torch.isfinite(torch.as_tensor(v)).all()) for v in out.update_list)
    print(json.dumps({"B": B, "crop": crop, "dtype": dtype, "variant": variant, "ms": round(ms, 4), "finite": finite}), flush=True)


def sweep(shapes, fp16_shapes):
    rows = []
    for (B, crop, dtype) in [(b, c, "bf16") for b, c in shapes] + [(b, c, "fp16") for b, c in fp16_shapes]:
        res = {}
        for v in VARIANTS:
            env = dict(os.environ)
            for k in list(env):
                if k.startswith("TECOGAN_PERSIST"):
                    env.pop(k)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(B), str(crop), dtype, v], env=env,
                               capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                res[v] = None
                print(f"# {B}x{crop} {dtype} {v}: FAILED rc={r.returncode} {r.stderr[-300:]!r}", flush=True)
                continue
            d = json.loads(line[-1])
            res[v] = d["ms"] if d["finite"] else None
        ok = {k: v for k, v in res.items() if v is not None}
        best = min(ok, key=ok.get) if ok else None
        regret = (res["default"] - ok[best]) / ok[best] if best and res.get("default") else None
        rows.append((B, crop, dtype, res, best, regret))
        frames = B * 10
        print(f"B={B} crop={crop:3d} {dtype}: " + "  ".join(f"{v} {res[v]:.3f}" if res[v] else f"{v} -" for v in VARIANTS)
              + (f"  | best {best}, regret of the default {100 * regret:.1f} %, {frames / res['default'] * 1e3:.0f} HR-frames/s" if regret is not None else ""),
              flush=True)
    regs = [r[5] for r in rows if r[5] is not None]
    if regs:
        worst = max(rows, key=lambda r: r[5] if r[5] is not None else -1)
        print(f"# {len(regs)} shapes: worst regret {100 * max(regs):.1f} % (B={worst[0]} crop={worst[1]} {worst[2]}, best {worst[4]}), "
              f"median {100 * sorted(regs)[len(regs) // 2]:.1f} %", flush=True)
    return rows


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
    else:
        shapes = [(b, c) for b in (1, 2, 4, 8) for c in (32, 48, 64, 96)]
        fp16 = [(1, 32), (2, 48), (4, 64), (8, 96)]
        if len(sys.argv) > 2 and sys.argv[1] == "--shapes":
            shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[2].split(",")]
            fp16 = []
        sweep(shapes, fp16)

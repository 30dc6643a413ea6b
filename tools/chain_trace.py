"""Anatomy of the recurrent generator pass (lane A's chain) alone - or of any other piece of the step.
  python tools/chain_trace.py run              replays only the chain graphs (chain0, chain, chain_tail) 20 times
  python tools/chain_trace.py parse <csv>      from a rocprofv3 --kernel-trace csv of the line above: the kernels of ONE frame in
                                               launch order with their average duration and the gap to the next kernel
  python tools/chain_trace.py run g_bwd        replays the named pieces (TecoGANStep.PIECES) 20 times instead
  python tools/chain_trace.py parse <csv> 20   ... and folds the trace's last 20 equal repetitions
(rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chain -o chain -- python3 tools/chain_trace.py run)"""
import collections, csv, os, re, sys


def run(pieces=("chain0", "chain", "chain_tail")):
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
    import pytorch_tecogan_amd  # noqa: F401
    from pytorch_tecogan_amd import models as M, train as TR
    import bench as B
    args = B.default_args("bf16")
    torch.manual_seed(1)
    dev = torch.device("cuda", 0)
    G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
    og = torch.optim.Adam(G.parameters(), 1e-4); od = torch.optim.Adam(D.parameters(), 1e-4)
    x, y = B.synth(4, 10, 32, 1); x, y = x.to(dev), y.to(dev)
    os.environ["TECOGAN_GRAPH"] = "1"
    for s in range(3):
        TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
    torch.cuda.synchronize()
    st = next(iter(TR._STEPS.values()))
    g = st.graphs
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(21):
        if rep == 1:
            e0.record()
        for k in pieces:
            if k in st.LANE_B:
                with torch.cuda.stream(st.sB):
                    g[k]()
                torch.cuda.current_stream().wait_stream(st.sB)
            else:
                g[k]()
    e1.record(); torch.cuda.synchronize()
    print(f"{'+'.join(pieces)} alone: {e0.elapsed_time(e1) / 20:.3f} ms per step")


def parse_reps(path, reps):
    """the last `reps` repetitions of a replayed piece: per launch position the average duration"""
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:70]
    names = [short(r["Kernel_Name"]) for r in rows]
    n = next(n for n in range(1, len(rows) // reps + 1)   # period of the tail of the trace
             if names[-n * reps:] == names[-n:] * reps)
    reps_rows = [rows[len(rows) - (i + 1) * n:len(rows) - i * n] for i in range(reps)]
    tot = 0.0
    print(f"{reps} repetitions of {n} kernels")
    for k in range(n):
        d = sum(rr[k]["e"] - rr[k]["s"] for rr in reps_rows) / reps / 1e3
        r0 = reps_rows[0][k]
        wg = int(r0["Grid_Size_X"]) * int(r0["Grid_Size_Y"]) * int(r0["Grid_Size_Z"]) // max(1, int(r0["Workgroup_Size_X"]) * int(r0["Workgroup_Size_Y"]))
        tot += d
        print(f"{k:3d} {names[len(rows) - n + k]:72s} WGs {wg:6d}  {d:8.2f} us")
    print(f"sum of durations {tot:.1f} us")


def parse(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:70]
    gi = [i for i, r in enumerate(rows) if "gen_input" in r["Kernel_Name"]]
    # frames = runs between consecutive gen_input launches; keep those with the modal length (the passes of the replays)
    frames = [rows[a:b] for a, b in zip(gi, gi[1:])]
    n = collections.Counter(len(f) for f in frames).most_common(1)[0][0]
    frames = [f for f in frames if len(f) == n][-150:]
    tot_d = tot_g = 0.0
    print(f"{len(frames)} frames of {n} kernels")
    for k in range(n):
        d = sum(f[k]["e"] - f[k]["s"] for f in frames) / len(frames) / 1e3
        gap = sum((f[k + 1]["s"] - f[k]["e"]) for f in frames if k + 1 < n) / len(frames) / 1e3 if k + 1 < n else 0.0
        wg = int(frames[0][k]["Grid_Size_X"]) * int(frames[0][k]["Grid_Size_Y"]) * int(frames[0][k]["Grid_Size_Z"]) // max(1, int(frames[0][k]["Workgroup_Size_X"]))
        tot_d += d; tot_g += gap
        print(f"{k:3d} {short(frames[0][k]['Kernel_Name']):72s} WGs {wg:5d}  {d:7.2f} us  gap {gap:5.2f}")
    print(f"sum of durations {tot_d:.1f} us, of gaps {tot_g:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(tuple(sys.argv[2:])) if len(sys.argv) > 2 else run()
    elif len(sys.argv) > 3:
        parse_reps(sys.argv[2], int(sys.argv[3]))
    else:
        parse(sys.argv[2])

"""When do the two lanes of a replayed step run?  HIP events in front of and behind every piece of the step on its lane's
stream (no profiler: rocprofv3's kernel trace dispatches the two streams' kernels almost serially, tools/overlap_from_trace.py),
per-frame events inside the chain, and from the piece intervals the share of the step in which BOTH lanes are at work.
    python tools/lane_ends.py [--json out.json]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import models as M, train as TR  # noqa: E402
import bench as B  # noqa: E402

args = B.default_args("bf16")
torch.manual_seed(1)
dev = torch.device("cuda", 0)
G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
og, od = torch.optim.Adam(G.parameters(), 1e-4), torch.optim.Adam(D.parameters(), 1e-4)
x, y = B.synth(4, 10, 32, 1)
x, y = x.to(dev), y.to(dev)
os.environ["TECOGAN_GRAPH"] = "1"
for s in range(3):
    TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
torch.cuda.synchronize()
st = next(iter(TR._STEPS.values()))
g = st.graphs
# per-frame graphs of the chain (frames 1..8), so that events can sit between the frames
frames = []
for t in range(1, st.tsize):
    st._chain(t, t + 1)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        st._chain(t, t + 1)
    frames.append(gr)

A_PIECES = ["chain0"] + [f"frame{t}" for t in range(1, st.tsize)] + ["chain_tail", "g_bwd", "update"]
B_PIECES = ["prep", "d_real", "d_fake", "d_fake_bwd", "update_d"]
R = 20
evs = []
for rep in range(R + 2):
    ev = {"start": torch.cuda.Event(enable_timing=True)}
    main, sB, sBm = torch.cuda.current_stream(), st.sB, st.sBm

    def run(name, stream, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        fn()
        b.record(stream)
        ev[name] = (a, b)

    ev["start"].record(main)   # (no host sync between repetitions: the host runs ahead, as in training)
    st.ev["start"].record(main)
    sBm.wait_event(st.ev["start"])
    with torch.cuda.stream(sBm):
        run("prep", sBm, g["prep"])
        st.ev["prep"].record(sBm)
        run("d_real", sBm, g["d_real"])
    run("chain0", main, g["chain0"])
    main.wait_event(st.ev["prep"])
    for t, gr in enumerate(frames, start=1):
        run(f"frame{t}", main, gr.replay)
    st.ev["chain"].record(main)
    sB.wait_event(st.ev["chain"])
    with torch.cuda.stream(sB):
        run("d_fake", sB, g["d_fake"])
    run("chain_tail", main, g["chain_tail"])
    st.ev["tail"].record(main)
    sB.wait_event(st.ev["tail"])
    with torch.cuda.stream(sB):
        run("d_fake_bwd", sB, g["d_fake_bwd"])
        run("update_d", sB, g["update_d"])
        st.ev["d"].record(sB)
    run("g_bwd", main, g["g_bwd"])
    if st.scaler is None:   # as TecoGANStep._run_lanes: G's Adam + repack beside lane B's tail, then the caller's stream joins lane B
        run("update", main, g["update"])
        main.wait_event(st.ev["d"])
    else:
        main.wait_event(st.ev["d"])
        run("update", main, g["update"])
    ev["end"] = torch.cuda.Event(enable_timing=True)
    ev["end"].record(main)
    evs.append(ev)
torch.cuda.synchronize()
iv = {}
for k in A_PIECES + B_PIECES:
    a = sum(e["start"].elapsed_time(e[k][0]) for e in evs[2:]) / R
    b = sum(e["start"].elapsed_time(e[k][1]) for e in evs[2:]) / R
    iv[k] = (a, b)
    print(f"{k:12s} {a:7.3f} -> {b:7.3f} ms  ({b - a:6.3f})")
step = sum(e["start"].elapsed_time(e["end"]) for e in evs[2:]) / R   # the caller's stream has joined lane B


def overlap(x, y):
    return max(0.0, min(x[1], y[1]) - max(x[0], y[0]))


both = sum(overlap(iv[a], iv[b]) for a in A_PIECES for b in B_PIECES)
ta, tb = sum(iv[k][1] - iv[k][0] for k in A_PIECES), sum(iv[k][1] - iv[k][0] for k in B_PIECES)
res = {"step_ms": round(step, 3), "lane_A_busy_ms": round(ta, 3), "lane_B_busy_ms": round(tb, 3), "both_lanes_busy_ms": round(both, 3),
       "both_lanes_busy_frac": round(both / step, 4), "lane_A_busy_frac": round(ta / step, 4), "lane_B_busy_frac": round(tb / step, 4),
       "pieces_ms": {k: [round(v[0], 3), round(v[1], 3)] for k, v in iv.items()},
       "note": "HIP events around every piece (a piece = one hipGraph of back-to-back launches on its lane's stream), mean of "
               f"{R} replayed steps, no profiler; lane A = caller's stream (generator chain, G backward), lane B = stream sB "
               "(discriminator)"}
print(f"step {step:.3f} ms; lane A busy {ta:.3f}, lane B busy {tb:.3f}, both busy {both:.3f} ms = {both / step * 100:.1f} % of the step")
if "--json" in sys.argv:
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
        json.dump(res, fh, indent=1)

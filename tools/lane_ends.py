"""When do the two lanes of a replayed step end?  (events behind lane B's last piece and behind lane A's G backward / update)"""
import os, sys
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
# per-frame graphs of the chain (frames 1..8), so that events can sit between the frames
frames = []
for t in range(1, st.tsize):
    st._chain(t, t + 1); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        st._chain(t, t + 1)
    frames.append(gr)
names = ["start", "chain_end", "tail_end", "gbwd_end", "step_end", "dreal_end", "dfake_end", "laneB_end"] + \
        [f"frame{t}_end" for t in range(0, st.tsize)]
acc = {k: 0.0 for k in names}
R = 20
evs = []
for rep in range(R + 2):
    ev = {k: torch.cuda.Event(enable_timing=True) for k in names}
    main, sB, sBm = torch.cuda.current_stream(), st.sB, st.sBm
    ev["start"].record(main)   # (no host sync between repetitions: the host runs ahead, as in training)
    st.ev["start"].record(main); sBm.wait_event(st.ev["start"])
    with torch.cuda.stream(sBm):
        g["prep"](); st.ev["prep"].record(sBm); g["d_real"](); ev["dreal_end"].record(sBm)
    g["chain0"](); ev["frame0_end"].record(main); main.wait_event(st.ev["prep"])
    for t, gr in enumerate(frames, start=1):
        gr.replay(); ev[f"frame{t}_end"].record(main)
    ev["chain_end"].record(main)
    st.ev["chain"].record(main); sB.wait_event(st.ev["chain"])
    with torch.cuda.stream(sB):
        g["d_fake"](); ev["dfake_end"].record(sB)
    g["chain_tail"](); ev["tail_end"].record(main); st.ev["tail"].record(main); sB.wait_event(st.ev["tail"])
    with torch.cuda.stream(sB):
        g["d_fake_bwd"](); g["update_d"](); st.ev["d"].record(sB); ev["laneB_end"].record(sB)
    g["g_bwd"](); ev["gbwd_end"].record(main)
    main.wait_event(st.ev["d"]); g["update"](); ev["step_end"].record(main)
    evs.append(ev)
torch.cuda.synchronize()
for ev in evs[2:]:
    for k in names[1:]:
        acc[k] += ev["start"].elapsed_time(ev[k])
for k in names[1:]:
    print(f"{k:12s} at {acc[k] / R:7.3f} ms")

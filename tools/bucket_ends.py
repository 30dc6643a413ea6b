"""Data-parallel gradient buckets: WHEN is each bucket's all-reduce issued relative to the end of its backward pass?
One rank on one GPU with the real RCCL process group (TECOGAN_FORCE_COLLECTIVES=1 issues the collectives although
world == 1), the step replayed from its per-lane graphs with events behind every bucket piece.
    TECOGAN_DP_INLINE=0 TECOGAN_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port 29577 tools/bucket_ends.py
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_tecogan_amd  # noqa: E402,F401
from pytorch_tecogan_amd import models as M, train as TR  # noqa: E402
import bench as B  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
args = B.default_args("bf16")
torch.manual_seed(1)
G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
og, od = torch.optim.Adam(G.parameters(), 1e-4), torch.optim.Adam(D.parameters(), 1e-4)
x, y = B.synth(4, 10, 32, 1)
x, y = x.to(dev), y.to(dev)
os.environ["TECOGAN_GRAPH"] = "1"
for s in range(3):
    TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
torch.cuda.synchronize()
st = next(iter(TR._STEPS.values()))
assert st.buckets and st.graphs is not None, "needs TECOGAN_DP_INLINE=0 TECOGAN_FORCE_COLLECTIVES=1 under torch.distributed.run"
g = st.graphs
names = ["start", "tail_end", "g_hr_end", "g_trunk_end", "d_hi_end", "d_lo_end", "g_ar1_done", "g_ar2_done", "d_ar1_done", "d_ar2_done",
         "step_end"]
R, evs = 20, []
side = torch.cuda.Stream(device=dev)
gs, ds = st.G.bucket_split(), st.D.bucket_split()
for rep in range(R + 2):
    ev = {k: torch.cuda.Event(enable_timing=True) for k in names}
    main, sB, sBm = torch.cuda.current_stream(), st.sB, st.sBm
    st.ev["start"].record(main)
    sBm.wait_event(st.ev["start"])
    with torch.cuda.stream(sBm):
        g["prep"]()
        st.ev["prep"].record(sBm)
        g["d_real"]()
    g["chain0"]()
    main.wait_event(st.ev["prep"])
    g["chain"]()
    st.ev["chain"].record(main)
    sB.wait_event(st.ev["chain"])
    with torch.cuda.stream(sB):
        g["d_fake"]()
    ev["start"].record(main)          # phase 2 begins: tail pass + G backward | fake half
    g["chain_tail"]()
    ev["tail_end"].record(main)
    st.ev["tail"].record(main)
    sB.wait_event(st.ev["tail"])
    with torch.cuda.stream(sB):
        g["d_fake_bwd_hi"]()
        ev["d_hi_end"].record(sB)
    g["g_bwd_hr"]()
    ev["g_hr_end"].record(main)
    w_g1 = st._allreduce(st.G.flat.g[gs:])
    with torch.cuda.stream(sB):
        w_d1 = st._allreduce(st.D.flat.g[ds:])
        g["d_fake_bwd_lo"]()
        ev["d_lo_end"].record(sB)
        w_d2 = st._allreduce(st.D.flat.g[:ds])
    g["g_bwd_trunk"]()
    ev["g_trunk_end"].record(main)
    w_g2 = st._allreduce(st.G.flat.g[:gs])
    # completion of each collective, seen from a stream of its own (Work.wait() makes the CURRENT stream wait: on the lanes
    # themselves the wait would queue behind the lane's later pieces)
    # (ONE extra stream, waits in RCCL's issue order: more streams would take hardware queues away from the lanes)
    with torch.cuda.stream(side):
        for w, k in ((w_g1, "g_ar1_done"), (w_d1, "d_ar1_done"), (w_d2, "d_ar2_done"), (w_g2, "g_ar2_done")):
            w.wait()
            ev[k].record(side)
    with torch.cuda.stream(sB):
        w_d1.wait()
        w_d2.wait()
        g["update_d"]()
        st.ev["d"].record(sB)
    w_g1.wait()
    w_g2.wait()
    main.wait_event(st.ev["d"])
    g["update"]()
    ev["step_end"].record(main)
    evs.append(ev)
torch.cuda.synchronize()
print(f"RCCL world {dist.get_world_size()}, buckets: G [{gs}:] then [:{gs}] of {st.G.flat.total} floats, D [{ds}:] then [:{ds}] of "
      f"{st.D.flat.total}; ms after the start of phase 2 (mean of {R} replayed steps)")
for k in names[1:]:
    print(f"{k:12s} at {sum(e['start'].elapsed_time(e[k]) for e in evs[2:]) / R:7.3f} ms")
dist.destroy_process_group()

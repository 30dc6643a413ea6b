"""Per-layer launch times of one eager step (GPU parked first so that host launch gaps do not count; each launch has its
own event pair, which adds a few microseconds per launch - use for ranking, not for absolute small-kernel durations)."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pytorch_tecogan_amd
from pytorch_tecogan_amd import models as M, train as TR, engine as E
import bench as B

args = B.default_args("bf16"); torch.manual_seed(1); dev = torch.device("cuda", 0)
G, D = M.generator(3, args).to(dev), M.discriminator(args).to(dev)
og = torch.optim.Adam(G.parameters(), 1e-4); od = torch.optim.Adam(D.parameters(), 1e-4)
x, y = B.synth(4, 10, 32, 1); x, y = x.to(dev), y.to(dev)
os.environ["TECOGAN_GRAPH"] = "0"
for s in range(2): TR.FRVSR_Train(x, y, args, D, G, s, 0., 0., og, od)
torch.cuda.synchronize()
st = next(iter(TR._STEPS.values()))
names = {}
for net, eng in (("G", st.G), ("D", st.D)):
    for k, v in vars(eng).items():
        if isinstance(v, E.Conv): names[id(v)] = f"{net}.{k}"
    for i, (c1, c2) in enumerate(getattr(eng, "rb", [])): names[id(c1)] = f"G.rb{i}.0"; names[id(c2)] = f"G.rb{i}.2"
    if net == "D":
        for k, (c, bn) in eng.blk.items(): names[id(c)] = f"D.block{k}"
        for stg, lst in eng.res.items():
            for j, (c1, c2, bn) in enumerate(lst): names[id(c1)] = f"D.res{stg}.{j}.0"; names[id(c2)] = f"D.res{stg}.{j}.2"
recs = []
def wrap(op, fn):
    def w(self, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(self, *a, **kw); e1.record()
        recs.append((names.get(id(self), "?"), op, a[0].shape[0], e0, e1)); return r
    return w
o = (E.Conv.fwd, E.Conv.dgrad, E.Conv.wgrad)
E.Conv.fwd, E.Conv.dgrad, E.Conv.wgrad = wrap("fwd", o[0]), wrap("dgrad", o[1]), wrap("wgrad+fin", o[2])
torch.cuda.synchronize(); torch.cuda._sleep(int(0.08 * 2.0e9))
st._forward_backward(True); torch.cuda.synchronize()
E.Conv.fwd, E.Conv.dgrad, E.Conv.wgrad = o
agg = collections.OrderedDict()
for n, op, N, e0, e1 in recs:
    key = (n if not n.startswith("G.rb") else "G.rb*." + n[-1], op, N)
    d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e3
tot = collections.Counter()
for (n, op, N), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot[(n[0], op)] += us
    if us > 60: print(f"{n:14s} {op:10s} N={N:2d} launches {c:4d} total {us:8.1f} us  avg {us/c:7.1f} us")
print({f"{k[0]}:{k[1]}": round(v / 1e3, 3) for k, v in tot.items()})

"""Worker of tests/test_dp_gpu.py (not a test module): ONE rank of a 2- (or 4-) rank data-parallel run whose ranks share cuda:0.
Launched by torch.distributed.run with the gloo backend (RCCL refuses two ranks on one device), before anything touches
the GPU.  Each rank owns one sequence; 2 steps through FRVSR_Train (step 0 eager + capture, step 1 hipGraph replay with
the collectives between the lane graphs).  Rank 0 then replays the same two steps on the CPU oracle as "two shards with
local BatchNorm, gradients averaged"  and writes the comparison as JSON."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.path.insert(1, os.path.join(ROOT, "code"))


def synth(seed):
    rng = np.random.default_rng(seed)
    return (torch.from_numpy(rng.random((1, 10, 3, 32, 32), dtype=np.float32)),
            torch.from_numpy(rng.random((1, 10, 3, 128, 128), dtype=np.float32)))


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main(out_path):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    import models
    import train
    import tecogan_oracle as orc

    args = orc.default_args(num_resblock=2, discrim_resblocks=1)
    args.tg_dtype = "fp32"
    os.environ["TECOGAN_GRAPH"] = "1"
    gp = orc.init_params(orc.generator_param_shapes(2), 11)
    dp = orc.init_params(orc.discriminator_param_shapes(1, 128), 12)
    G, D = models.generator(3, args), models.discriminator(args)
    G.load_state_dict(gp)
    D.load_state_dict(dp, strict=False)
    G, D = G.cuda(), D.cuda()
    og = torch.optim.Adam(G.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    od = torch.optim.Adam(D.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    x, y = synth(40 + rank)
    xs, ys = x.cuda(), y.cuda()
    n_steps = 2
    sums = []
    for s in range(n_steps):
        out = train.FRVSR_Train(xs, ys, args, D, G, s, 0.0, 0.0, og, od)
        torch.cuda.synchronize()
        sums.append([float(v) for v in out.update_list])
    import pytorch_tecogan_amd.train as hip_train
    st = next(iter(hip_train._STEPS.values()))
    assert st.world == world and st.graphs is not None, (st.world, st.graphs)
    w_g = torch.cat([p.detach().flatten() for p in G.parameters()]).cpu()
    w_d = torch.cat([p.detach().flatten() for p in D.parameters()]).cpu()
    digest = torch.stack([w_g.double().sum(), w_g.double().abs().sum(), w_d.double().sum(), w_d.double().abs().sum()])
    both = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, digest)
    res = {"rank": rank, "replicas_bit_equal": all(bool(torch.equal(both[0], b)) for b in both[1:]), "scalars": sums, "world": world}
    if rank == 0:
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
        # the HIP side's view of the LAST step: flat gradient buffers hold the SUM over ranks (1/world lives in tg_adam)
        g_sum = {k: p.grad.detach().cpu().clone() for k, p in G.named_parameters()}
        d_sum = {k: p.grad.detach().cpu().clone() for k, p in D.named_parameters()}
        m_g = {k: og.state[p]["exp_avg"].detach().cpu().clone() for k, p in G.named_parameters()}
        # oracle: two shards at the same weights, local BN statistics (own running buffers per shard), averaged gradients
        gp0 = orc.init_params(orc.generator_param_shapes(2), 11)
        dp0 = orc.init_params(orc.discriminator_param_shapes(1, 128), 12)
        bufs = [orc.init_bn_buffers(dp0, 1) for _ in range(world)]
        o_g, o_d = orc.AdamState(gp0, args.learning_rate, args.beta, 0.999, args.adameps), \
            orc.AdamState(dp0, args.learning_rate, args.beta, 0.999, args.adameps)
        data = [synth(40 + r) for r in range(world)]
        o_scal = []
        for s in range(n_steps):
            acc_g = {k: torch.zeros_like(v) for k, v in gp0.items()}
            acc_d = {k: torch.zeros_like(v) for k, v in dp0.items()}
            for r in range(world):
                g = {k: v.clone().requires_grad_(True) for k, v in gp0.items()}
                d = {k: v.clone().requires_grad_(True) for k, v in dp0.items()}
                f = orc.tecogan_forward(g, d, bufs[r], data[r][0], data[r][1], args, s)
                gg = torch.autograd.grad(f["gen_loss"], list(g.values()), retain_graph=True)
                dg = torch.autograd.grad(f["d_loss"], list(d.values()))
                for k, t in zip(g, gg):
                    acc_g[k] += t / world
                for k, t in zip(d, dg):
                    acc_d[k] += t / world
                if r == 0:
                    o_scal.append([float(v) for v in f["update_list"]])
            with torch.no_grad():
                o_g.step(gp0, acc_g)
                o_d.step(dp0, acc_d)
        ow_g = torch.cat([gp0[k].flatten() for k, _ in G.named_parameters()])
        ow_d = torch.cat([dp0[k].flatten() for k, _ in D.named_parameters()])
        res["w_g"], res["w_d"] = rel(w_g, ow_g), rel(w_d, ow_d)
        big = lambda t: float(t.norm()) > 1e-6  # noqa: E731
        res["grad_sum_g"] = max(rel(g_sum[k] / world, acc_g[k]) for k in acc_g if big(acc_g[k]))
        res["grad_sum_d"] = max(rel(d_sum[k] / world, acc_d[k]) for k in acc_d if big(acc_d[k]))
        gvec = torch.cat([g_sum[k].flatten() / world for k in acc_g])
        res["grad_vec_g"] = rel(gvec, torch.cat([acc_g[k].flatten() for k in acc_g]))
        res["grad_vec_d"] = rel(torch.cat([d_sum[k].flatten() / world for k in acc_d]),
                                torch.cat([acc_d[k].flatten() for k in acc_d]))
        res["adam_m_g"] = rel(torch.cat([m_g[k].flatten() for k in acc_g]), torch.cat([o_g.m[k].flatten() for k in acc_g]))
        res["scal_err"] = float(np.max(np.abs(np.array(sums) - np.array(o_scal)) / (np.abs(np.array(o_scal)) + 1e-6)))
        res["bn_rm"] = rel(D.state_dict()["block1.1.running_mean"], bufs[0]["block1.1.running_mean"])
    with open(f"{out_path}.{rank}", "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])

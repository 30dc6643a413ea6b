"""Checkpoint fixtures written BY THE REFERENCE's own modules (build-container tool, like make_golden.py: the GPU box has no
/root/reference; nothing here is imported by the product).

  python oracle/make_ckpt_golden.py                  writes tests/golden/ref_generator.pt.xz, ref_discrim.pt.xz, ckpt_expect.npz
  python oracle/make_ckpt_golden.py --check DIR      loads DIR/generator.pt, DIR/discrim.pt - written by THIS build's main.py on a
                                                     GPU - into the reference's modules and optimisers with the reference's own
                                                     resume statements (/root/reference/main.py:251-258); prints what it verified

What is pinned (/root/reference/main.py:230-247 construction, :308-317 save, :251-263 load):
  * the files are torch.save({'epoch', 'model_state_dict', 'optimizer_state_dict'}) / ({'model_state_dict', 'optimizer_state_dict'}) of
    the reference's `generator(3, args)` / `discriminator(args)` and the two `torch.optim.Adam` (+ StepLR, which adds `initial_lr` to
    the param groups) - the pickle layout, key order, `param_groups` / integer-indexed `state` layout are whatever the reference writes;
  * a SMALL architecture (num_resblock 2, discrim_resblocks 1) and low-entropy, position-dependent values: every parameter element is
    ((i mod 61) - 30) / 2048 + ((i div 61) mod 7) / 16384 (a transposed or shifted load changes it), the optimiser state comes from two
    real `Adam.step()` calls on gradients of the same kind - so the 24 MB of tensors compress to a few hundred KB under xz;
  * ckpt_expect.npz: per tensor a few sampled values and the sum, for tests that check what the build's modules hold after loading."""
import argparse
import lzma
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (the shims)
import tecogan_oracle as orc  # noqa: E402

EPOCH = 7


def pattern(n, salt):
    i = np.arange(n, dtype=np.int64) + salt
    return (((i % 61) - 30) / 2048.0 + ((i // 61) % 7) / 16384.0).astype(np.float32)


def fill(module, opt, salt0):
    """position-dependent low-entropy parameters; two Adam steps on gradients of the same kind; BN buffers likewise"""
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            p.copy_(torch.from_numpy(pattern(p.numel(), salt0 + 17 * k)).view_as(p))
        for k, (name, b) in enumerate(module.named_buffers()):
            if b.dtype == torch.long:
                b.fill_(4)
            else:
                b.copy_(torch.from_numpy(np.abs(pattern(b.numel(), salt0 + 5 * k)) + (0.5 if name.endswith("running_var") else 0.0)).view_as(b))
    for step in range(2):
        for k, (name, p) in enumerate(module.named_parameters()):
            p.grad = torch.from_numpy(pattern(p.numel(), salt0 + 1000 + 31 * k + step)).view_as(p).clone()
        opt.step()
    opt.zero_grad(set_to_none=True)


def build_reference(args):
    import models  # the REFERENCE's (sys.path set by main())
    G = models.generator(3, args=args)
    D = models.discriminator(args=args)
    # /root/reference/main.py:236-247
    lr_d = args.learning_rate if args.Dt_mergeDs else args.learning_rate * 0.3
    opt_d = torch.optim.Adam(D.parameters(), lr_d, betas=(args.beta, 0.999), eps=args.adameps)
    opt_g = torch.optim.Adam(G.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    sch = (torch.optim.lr_scheduler.StepLR(opt_d, args.decay_step, args.decay_rate),
           torch.optim.lr_scheduler.StepLR(opt_g, args.decay_step, args.decay_rate))
    return G, D, opt_g, opt_d, sch


def small_args():
    a = orc.default_args(num_resblock=2, discrim_resblocks=1)
    a.decay_step, a.decay_rate = 250, 0.8
    return a


def write():
    G, D, opt_g, opt_d, _ = build_reference(small_args())
    fill(G, opt_g, 3)
    fill(D, opt_d, 4001)
    tmp = os.path.join(OUT, "_tmp.pt")
    expect = {}
    for tag, obj, mod, opt in (("generator", {"epoch": EPOCH, "model_state_dict": G.state_dict(), "optimizer_state_dict": opt_g.state_dict()}, G, opt_g),
                               ("discrim", {"model_state_dict": D.state_dict(), "optimizer_state_dict": opt_d.state_dict()}, D, opt_d)):
        torch.save(obj, tmp)   # /root/reference/main.py:308-317
        raw = open(tmp, "rb").read()
        with open(os.path.join(OUT, f"ref_{tag}.pt.xz"), "wb") as f:
            f.write(lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME))
        print(tag, len(raw), "->", os.path.getsize(os.path.join(OUT, f"ref_{tag}.pt.xz")), "bytes")
        names = [n for n, _ in mod.named_parameters()]
        expect[tag + "_param_names"] = np.array(names)
        expect[tag + "_state_keys"] = np.array(list(mod.state_dict().keys()))
        for k, v in mod.state_dict().items():
            expect[f"{tag}.{k}.sum"] = np.float64(v.double().sum())
            expect[f"{tag}.{k}.head"] = v.reshape(-1)[:8].double().numpy()
        st = opt.state_dict()
        expect[tag + "_opt_group_keys"] = np.array(sorted(st["param_groups"][0].keys()))
        expect[tag + "_opt_state_keys"] = np.array(sorted(st["state"][0].keys()))
        for i, n in enumerate(names):
            expect[f"{tag}.opt.{n}.exp_avg.sum"] = np.float64(st["state"][i]["exp_avg"].double().sum())
            expect[f"{tag}.opt.{n}.exp_avg_sq.sum"] = np.float64(st["state"][i]["exp_avg_sq"].double().sum())
        expect[tag + "_opt_step"] = np.float64(float(st["state"][0]["step"]))
    os.remove(tmp)
    expect["epoch"] = np.int64(EPOCH)
    np.savez_compressed(os.path.join(OUT, "ckpt_expect.npz"), **expect)


def check(directory):
    """a checkpoint written by this build's main.py resumes in the REFERENCE: its own statements, /root/reference/main.py:251-258"""
    g_checkpoint = torch.load(os.path.join(directory, "generator.pt"), map_location="cpu")
    d_checkpoint = torch.load(os.path.join(directory, "discrim.pt"), map_location="cpu")
    nrb = sum(1 for k in g_checkpoint["model_state_dict"] if k.startswith("resids.") and k.endswith(".0.weight"))
    drb = sum(1 for k in d_checkpoint["model_state_dict"] if k.startswith("resids1.") and k.endswith(".0.0.weight"))
    args = orc.default_args(num_resblock=nrb, discrim_resblocks=drb)
    args.decay_step, args.decay_rate = 250, 0.8
    generator_F, discriminator_F, gen_optimizer, tdiscrim_optimizer, _ = build_reference(args)
    generator_F.load_state_dict(g_checkpoint["model_state_dict"])
    gen_optimizer.load_state_dict(g_checkpoint["optimizer_state_dict"])
    current_epoch = g_checkpoint["epoch"]
    discriminator_F.load_state_dict(d_checkpoint["model_state_dict"])
    tdiscrim_optimizer.load_state_dict(d_checkpoint["optimizer_state_dict"])
    # ... and what arrived is what the file holds: parameters, BN buffers, both Adam moments and the step count of every parameter
    for mod, opt, ck in ((generator_F, gen_optimizer, g_checkpoint), (discriminator_F, tdiscrim_optimizer, d_checkpoint)):
        sd = mod.state_dict()
        assert list(sd.keys()) == list(ck["model_state_dict"].keys()), "state_dict key order differs from the reference's"
        for k, v in ck["model_state_dict"].items():
            assert sd[k].shape == v.shape and torch.equal(sd[k].float(), v.float().cpu()), k
        params = [p for _, p in mod.named_parameters()]
        saved = ck["optimizer_state_dict"]
        assert len(saved["param_groups"]) == 1 and saved["param_groups"][0]["params"] == list(range(len(params)))
        for i, p in enumerate(params):
            s_ref, s_ck = opt.state[p], saved["state"][i]
            assert s_ref["exp_avg"].shape == p.shape and torch.equal(s_ref["exp_avg"], s_ck["exp_avg"].cpu())
            assert torch.equal(s_ref["exp_avg_sq"], s_ck["exp_avg_sq"].cpu()) and float(s_ref["step"]) == float(s_ck["step"])
        # the resumed optimiser can take a step (the state has the dtypes / devices torch's Adam expects)
        for p in params:
            p.grad = torch.zeros_like(p)
        opt.step()
    print(f"reference resumed from {directory}: epoch {current_epoch}, generator {nrb} residual blocks "
          f"({len(g_checkpoint['model_state_dict'])} tensors), discriminator {drb} per stage ({len(d_checkpoint['model_state_dict'])} tensors), "
          f"Adam step {float(g_checkpoint['optimizer_state_dict']['state'][0]['step']):.0f} / "
          f"{float(d_checkpoint['optimizer_state_dict']['state'][0]['step']):.0f}; extra keys: "
          f"{sorted(set(g_checkpoint) - {'epoch', 'model_state_dict', 'optimizer_state_dict'})}")
    return current_epoch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", default=None)
    a = ap.parse_args()
    warnings.filterwarnings("ignore")
    torch.set_num_threads(1)
    mg.install_shims()
    sys.path.insert(1, mg.REF)
    if a.check:
        check(a.check)
    else:
        write()


if __name__ == "__main__":
    main()

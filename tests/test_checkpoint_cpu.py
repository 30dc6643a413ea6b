"""Checkpoint ABI against files WRITTEN BY THE REFERENCE (tests/golden/ref_generator.pt.xz, ref_discrim.pt.xz: the
reference's own modules and optimisers saved with its own statements, /root/reference/main.py:308-317, by oracle/make_ckpt_golden.py).
No GPU: the build's modules are parameter containers on the CPU here; nothing computes.
  * the reference's files load into the build's modules and torch optimisers with the statements of main.py's resume branch;
    keys, key ORDER, shapes, values, param-group keys and the integer-indexed optimiser state are the reference's;
  * (build container only - the GPU box has no /root/reference) a checkpoint written from the build's modules resumes in the
    REFERENCE through its own load statements; so does one written by main.py on a GPU, when a run has left it under gpurun_out/."""
import lzma
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(1, os.path.join(ROOT, "code"))
import tecogan_oracle as orc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def unxz(name, tmp_path):
    dst = tmp_path / name.replace(".xz", "")
    dst.write_bytes(lzma.decompress(open(os.path.join(GOLD, name), "rb").read()))
    return str(dst)


def small_args():
    return orc.default_args(num_resblock=2, discrim_resblocks=1)


def build_modules():
    import models   # ./code/models.py -> the build's modules
    args = small_args()
    G, D = models.generator(3, args=args), models.discriminator(args=args)
    og = torch.optim.Adam(G.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    od = torch.optim.Adam(D.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    return G, D, og, od


def check_against_expect(tag, module, opt, exp):
    sd = module.state_dict()
    assert list(sd.keys()) == list(exp[tag + "_state_keys"])                       # the reference's registration order
    assert [n for n, _ in module.named_parameters()] == list(exp[tag + "_param_names"])
    for k, v in sd.items():
        np.testing.assert_allclose(float(v.double().sum()), float(exp[f"{tag}.{k}.sum"]), rtol=1e-12, atol=1e-12, err_msg=k)
        np.testing.assert_array_equal(v.reshape(-1)[:8].double().cpu().numpy(), exp[f"{tag}.{k}.head"][:v.numel()], err_msg=k)
    st = opt.state_dict()
    assert sorted(st["state"][0].keys()) == list(exp[tag + "_opt_state_keys"])
    for i, n in enumerate(exp[tag + "_param_names"]):
        assert float(st["state"][i]["exp_avg"].double().sum()) == float(exp[f"{tag}.opt.{n}.exp_avg.sum"]), n
        assert float(st["state"][i]["exp_avg_sq"].double().sum()) == float(exp[f"{tag}.opt.{n}.exp_avg_sq.sum"]), n
        assert float(st["state"][i]["step"]) == float(exp[tag + "_opt_step"]) == 2.0


def test_reference_written_checkpoints_load_into_the_builds_modules(tmp_path):
    exp = np.load(os.path.join(GOLD, "ckpt_expect.npz"))
    G, D, og, od = build_modules()
    # main.py's resume branch (the reference's statements, /root/reference/main.py:251-258)
    g_ck = torch.load(unxz("ref_generator.pt.xz", tmp_path), map_location="cpu")
    G.load_state_dict(g_ck["model_state_dict"])
    og.load_state_dict(g_ck["optimizer_state_dict"])
    assert g_ck["epoch"] == int(exp["epoch"]) == 7 and set(g_ck) == {"epoch", "model_state_dict", "optimizer_state_dict"}
    d_ck = torch.load(unxz("ref_discrim.pt.xz", tmp_path), map_location="cpu")
    D.load_state_dict(d_ck["model_state_dict"])
    od.load_state_dict(d_ck["optimizer_state_dict"])
    assert set(d_ck) == {"model_state_dict", "optimizer_state_dict"}
    check_against_expect("generator", G, og, exp)
    check_against_expect("discrim", D, od, exp)
    # what the reference's StepLR left in the param groups is carried, not dropped (main.py re-creates its schedulers on resume)
    assert "initial_lr" in og.state_dict()["param_groups"][0] and list(exp["generator_opt_group_keys"]) == sorted(
        g_ck["optimizer_state_dict"]["param_groups"][0].keys())
    assert int(D.state_dict()["block1.1.num_batches_tracked"]) == 4
    # the build writes the same layout back: key order and the optimiser's indexing survive a save / load round trip
    torch.save({"epoch": 8, "model_state_dict": G.state_dict(), "optimizer_state_dict": og.state_dict()}, tmp_path / "g2.pt")
    g2 = torch.load(tmp_path / "g2.pt")
    assert list(g2["model_state_dict"].keys()) == list(g_ck["model_state_dict"].keys())
    assert g2["optimizer_state_dict"]["param_groups"][0]["params"] == g_ck["optimizer_state_dict"]["param_groups"][0]["params"]
    assert all(torch.equal(g2["model_state_dict"][k], v) for k, v in g_ck["model_state_dict"].items())


@pytest.mark.skipif(not os.path.isdir("/root/reference/code"), reason="build container only: the reference tree does not travel")
def test_build_written_checkpoints_resume_in_the_reference(tmp_path):
    """the other direction, in a child process (the reference's `models` / `ops` module names collide with ./code's)"""
    import subprocess
    G, D, og, od = build_modules()
    rng = np.random.default_rng(3)
    with torch.no_grad():
        for p in list(G.parameters()) + list(D.parameters()):
            p.copy_(torch.from_numpy(rng.uniform(-0.05, 0.05, size=tuple(p.shape)).astype(np.float32)))
    # the optimiser state in the form train._bind_optimizer leaves it: ONE shared step tensor, moments shaped like the parameters
    for mod, opt in ((G, og), (D, od)):
        step = torch.tensor(5.0)
        for p in mod.parameters():
            opt.state[p] = {"step": step, "exp_avg": torch.full_like(p, 1e-3), "exp_avg_sq": torch.full_like(p, 1e-6)}
    torch.save({"epoch": 3, "model_state_dict": G.state_dict(), "optimizer_state_dict": og.state_dict(), "tg_scaler": None},
               tmp_path / "generator.pt")
    torch.save({"model_state_dict": D.state_dict(), "optimizer_state_dict": od.state_dict()}, tmp_path / "discrim.pt")
    dirs = [str(tmp_path)]
    gpu_written = os.path.join(ROOT, "gpurun_out", "r06_ckpt")   # left there by tests/test_step_gpu.py on a GPU box, when one has run
    if os.path.exists(os.path.join(gpu_written, "generator.pt")):
        dirs.append(gpu_written)
    for d in dirs:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_ckpt_golden.py"), "--check", d], capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "reference resumed from" in r.stdout, (d, r.stdout[-500:], r.stderr[-2000:])

"""Size-independent properties of the step at BASELINE configs[1]'s FULL size (B = 4 sequences, T = 10, 32x32 -> 128x128,
bf16, hipGraph replay) - what can be held exactly without running the CPU oracle at that size:
  * structure of the discriminator inputs (target frames copied bit for bit, zero border of the warped part, LR up-sample);
  * the generator output is a sigmoid (0 < gen < 1) and frame 0 is the single-frame forward of the module;
  * the content loss scalar equals its definition on the returned gen_output;
  * gradients are linear in the loss: the flat gradient buffers of the two networks do not depend on each other's data path
    (G's gradient is the content gradient alone - the aliasing quirk - so it is untouched by a changed discriminator);
  * Adam with a zero learning rate leaves every weight and BN buffer bit-identical; replay == replay (determinism of
    everything but the float-atomic sums, which is bounded).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(1, os.path.join(ROOT, "code"))
import models  # noqa: E402
import train  # noqa: E402
import tecogan_oracle as orc  # noqa: E402
import pytorch_tecogan_amd.train as hip_train  # noqa: E402
from pytorch_tecogan_amd import ops  # noqa: E402

B, T, CS = 4, 10, 32
H = 4 * CS


def synth(seed):
    rng = np.random.default_rng(seed)
    return (torch.from_numpy(rng.random((B, T, 3, CS, CS), dtype=np.float32)).cuda(),
            torch.from_numpy(rng.random((B, T, 3, H, H), dtype=np.float32)).cuda())


def build(seed, lr=1e-4):
    args = orc.default_args(learning_rate=lr)
    args.tg_dtype = "bf16"
    torch.manual_seed(seed)
    G, D = models.generator(3, args).cuda(), models.discriminator(args).cuda()
    og = torch.optim.Adam(G.parameters(), lr, betas=(args.beta, 0.999), eps=args.adameps)
    od = torch.optim.Adam(D.parameters(), lr, betas=(args.beta, 0.999), eps=args.adameps)
    return args, G, D, og, od


@pytest.fixture()
def fresh(monkeypatch):
    monkeypatch.setenv("TECOGAN_GRAPH", "1")
    hip_train._STEPS.clear()
    yield
    hip_train._STEPS.clear()


def test_discriminator_input_structure_and_generator_output_at_full_size(fresh):
    args, G, D, og, od = build(11)
    x, y = synth(11)
    for s in range(3):   # eager + capture, then two replays
        out = train.FRVSR_Train(x, y, args, D, G, s, 0.0, 0.0, og, od)
    torch.cuda.synchronize()
    tgt = out.target                                   # real_in (12, 27, 128, 128): [target frames | warped targets | up4(LR)]
    assert tgt.shape == (B * 3, 27, H, H)
    y9 = y[:, :9].reshape(B * 3, 9, H, H)
    r16 = lambda t: t.bfloat16().float()   # the D input lives in the compute element type (bf16): one RNE rounding  # noqa: E731
    assert torch.equal(tgt[:, 0:9], r16(y9))                                      # copied (and rounded once)
    o = (H - int(H * args.crop_dt)) // 2
    w = tgt[:, 9:18]
    assert float(w[:, :, :o].abs().max()) == 0.0 and float(w[:, :, -o:].abs().max()) == 0.0
    assert float(w[:, :, :, :o].abs().max()) == 0.0 and float(w[:, :, :, -o:].abs().max()) == 0.0   # crop + pad == zero border
    assert float(w[:, :, o:-o, o:-o].abs().max()) > 0.0
    up = ops.upscale_four(x[:, :9].reshape(B * 9, 3, CS, CS)).reshape(B * 3, 9, H, H)
    assert torch.equal(tgt[:, 18:27], r16(up))                                     # bilinear x4, bit-exact kernel
    gen = out.gen_output
    assert gen.shape == (B, T, 3, H, H) and float(gen.min()) > 0.0 and float(gen.max()) < 1.0
    # the reported content part: l2_content_loss carries the aliased total, so rebuild it from its definition
    names = list(out.update_list_name)
    vals = {n: float(v) for n, v in zip(names, out.update_list)}
    content = float(torch.mean(torch.sum((gen - y) ** 2, dim=4)))
    total = content + 2 * args.ratio * vals["t_adversarial_loss"] + vals["D_layer_loss_sum"] * 1.0
    assert abs(vals["l2_content_loss"] - total) < 2e-3 * abs(total)
    assert vals["All_loss_Gen"] == vals["l2_content_loss"]
    assert 0.0 < vals["t_discrim_real_output"] < 1.0 and 0.0 < vals["t_discrim_fake_output"] < 1.0
    assert int(D.state_dict()["block1.1.num_batches_tracked"]) == 6              # two BN updates per step


def test_generator_gradient_does_not_depend_on_the_discriminator(fresh):
    """aliasing quirk (code/train.py:244,293-299): only the content loss reaches G.  Two runs that differ ONLY in D's
    weights must leave bit-identical generator outputs and (up to the float-atomic bias sums) identical G gradients."""
    x, y = synth(12)
    grads, gens = [], []
    for dseed in (1, 2):
        hip_train._STEPS.clear()
        args, G, D, og, od = build(12, lr=0.0)
        torch.manual_seed(100 + dseed)
        D2 = models.discriminator(args).cuda()
        out = train.FRVSR_Train(x, y, args, D2, G, 0, 0.0, 0.0, og, torch.optim.Adam(D2.parameters(), 0.0))
        torch.cuda.synchronize()
        gens.append(out.gen_output.clone())
        grads.append(torch.cat([p.grad.flatten() for p in G.parameters()]).clone())
    assert torch.equal(gens[0], gens[1])
    rel = float((grads[0] - grads[1]).norm() / grads[1].norm())
    assert rel < 1e-5, rel


def test_zero_learning_rate_leaves_all_state_but_the_bn_statistics_unchanged(fresh):
    args, G, D, og, od = build(13, lr=0.0)
    x, y = synth(13)
    g0 = {k: v.clone() for k, v in G.state_dict().items()}
    d0 = {k: v.clone() for k, v in D.state_dict().items()}
    losses = []
    for s in range(3):
        out = train.FRVSR_Train(x, y, args, D, G, s, 0.0, 0.0, og, od)
        losses.append([float(v) for v in out.update_list])
    torch.cuda.synchronize()
    for k, v in G.state_dict().items():
        assert torch.equal(v, g0[k]), k
    for k, v in D.state_dict().items():
        if "running_" in k or "num_batches" in k:
            continue
        assert torch.equal(v, d0[k]), k
    # same weights, same data: eager step, captured replay 1 and replay 2 agree.  (The BN batch statistics are float-atomic
    # sums whose last bits move bf16 roundings downstream: the discriminator's scalars repeat to ~3e-3, the generator's
    # to rounding)
    np.testing.assert_allclose(losses[1], losses[0], rtol=1e-2, atol=1e-5)
    np.testing.assert_allclose(losses[2], losses[1], rtol=1e-2, atol=1e-5)
    assert abs(losses[2][6] - losses[0][6]) <= 1e-6 * abs(losses[0][6])          # l2_warp_loss: no atomics between runs
    # running statistics follow momentum 0.1 from (0, 1): after 6 updates with the same batch statistics m,
    # running_mean = m * (1 - 0.9^6): the ratio between layers' own values after 2 and 6 updates is fixed
    rm = D.state_dict()["block1.1.running_mean"]
    assert bool(torch.isfinite(rm).all()) and float(rm.abs().max()) > 0.0

"""BASELINE configs[4]: generator-only recurrent inference at 128x128 -> 512x512 (main.py:141-220), the per-frame step
captured as one hipGraph; and the `--mode inference` branch of main.py on a folder of frames."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(1, os.path.join(ROOT, "code"))
import models  # noqa: E402
import tecogan_oracle as orc  # noqa: E402


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _gen(dtype, seed=31):
    args = orc.default_args()
    args.tg_dtype = dtype
    gp = orc.init_params(orc.generator_param_shapes(16), seed)
    G = models.generator(3, args)
    G.load_state_dict(gp)
    return G.cuda(), gp


@pytest.mark.timeout(900)
def test_config5_shape_four_frames_vs_oracle_fp32():
    """128x128 -> 512x512, T = 4 (what the CPU oracle finishes in bounded time), eager and per-frame hipGraph"""
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    G, gp = _gen("fp32")
    x = torch.from_numpy(np.random.default_rng(5).random((1, 4, 3, 128, 128), dtype=np.float32))
    with torch.no_grad():
        ref = orc.recurrent_generator(gp, x, orc.pseudo_flow(x))
    assert ref.shape == (1, 4, 3, 512, 512)
    for graph in (False, True):
        out = G.recurrent(x.cuda(), use_graph=graph)
        assert rel(out, ref) < 1e-4, graph


@pytest.mark.timeout(900)
def test_config5_sequence_of_120_frames_graph_equals_eager_bf16():
    """the benchmarked inference configuration: bf16, T = 120, per-frame hipGraph.  Properties that do not need the CPU
    oracle at this size: every frame finite and in (0,1) (sigmoid output), replaying the captured frame step 119 times
    gives exactly what 119 eager frame steps give, and frame 0 equals the single-frame generator forward."""
    G, _ = _gen("bf16")
    x = torch.from_numpy(np.random.default_rng(6).random((1, 120, 3, 128, 128), dtype=np.float32)).cuda()
    eager = G.recurrent(x, use_graph=False).clone()
    graph = G.recurrent(x, use_graph=True)
    assert graph.shape == (1, 120, 3, 512, 512)
    assert bool(torch.isfinite(graph).all()) and float(graph.min()) > 0.0 and float(graph.max()) < 1.0
    assert float((graph[:, 119] - eager[:, 119]).abs().max()) == 0.0
    assert float((graph - eager).abs().max()) == 0.0
    first = G(torch.cat([x[:, 0], torch.zeros(1, 48, 128, 128, device="cuda")], dim=1))
    assert float((first - graph[:, 0]).abs().max()) == 0.0
    # the recurrence matters: later frames depend on the previous output (not a per-frame feed-forward)
    alone = G.recurrent(x[:, 119:120], use_graph=False)
    assert float((alone[:, 0] - graph[:, 119]).abs().max()) > 0.0


@pytest.mark.parametrize("graph", [False, True])
def test_live_per_frame_step_carries_the_previous_frame_across_calls(graph):
    """generator.recurrent_step(frame): the stateful per-frame entry point of the live loop
    (/root/reference/experimental/live.py:100-128, main.py:191-219).  Feeding a sequence one frame per call gives, bit for bit, what
    the whole-sequence recurrent() gives (which is gated against the oracle above); the results are new tensors; reset=True and an
    intervening whole-sequence call both start a new stream."""
    G, _ = _gen("bf16")
    x = torch.from_numpy(np.random.default_rng(8).random((2, 7, 3, 32, 48), dtype=np.float32)).cuda()
    ref = G.recurrent(x, use_graph=False).clone()
    outs = [G.recurrent_step(x[:, t], use_graph=graph, reset=(t == 0)) for t in range(7)]
    assert len({o.data_ptr() for o in outs}) == 7
    for t, o in enumerate(outs):
        assert o.shape == (2, 3, 128, 192) and float((o - ref[:, t]).abs().max()) == 0.0, t
    # a second stream on the same object: reset, then the same frames again
    again = [G.recurrent_step(x[:, t], use_graph=graph, reset=(t == 0)) for t in range(3)]
    assert all(float((a - ref[:, t]).abs().max()) == 0.0 for t, a in enumerate(again))
    # without a reset the stream continues: frame 0 fed as the 4th frame of the stream is NOT a first frame
    cont = G.recurrent_step(x[:, 0], use_graph=graph)
    assert float((cont - ref[:, 0]).abs().max()) > 0.0
    # a whole-sequence call restarts the stream
    G.recurrent(x[:, :2].contiguous(), use_graph=graph)
    first = G.recurrent_step(x[:, 0], use_graph=graph)
    assert float((first - ref[:, 0]).abs().max()) == 0.0
    with pytest.raises(ValueError):
        G._rec.step(x[:, 0, :, :16].contiguous())


@pytest.mark.parametrize("chunk", ["16", "4", "1"])
def test_chunked_inference_batch_of_two_ragged_length_graph_equals_eager(chunk, monkeypatch):
    """RecurrentGenerator runs the sequence in chunks of TECOGAN_INFER_CHUNK frames (one staging copy in, one hipGraph, one strided
    copy out per chunk, the chunk's last frame carried into slot 0 of the next): B = 2 sequences of 19 frames - a first frame, one
    full chunk and a ragged tail at the default chunk size - give bit-identical results for every chunk size, graph or eager,
    and a second call on the same object (the graphs are reused, the carry slots start stale) repeats them."""
    monkeypatch.setenv("TECOGAN_INFER_CHUNK", chunk)
    G, _ = _gen("bf16")
    x = torch.from_numpy(np.random.default_rng(7).random((2, 19, 3, 32, 48), dtype=np.float32)).cuda()
    eager = G.recurrent(x, use_graph=False).clone()
    graph = G.recurrent(x, use_graph=True).clone()
    again = G.recurrent(x, use_graph=True)
    assert graph.shape == (2, 19, 3, 128, 192) and bool(torch.isfinite(graph).all())
    assert float((graph - eager).abs().max()) == 0.0 and float((again - graph).abs().max()) == 0.0
    # the two sequences do not see each other; a sequence alone gives the same frames
    solo = G.recurrent(x[1:2].contiguous(), use_graph=False)
    assert float((solo[0] - graph[1]).abs().max()) == 0.0
    monkeypatch.setenv("TECOGAN_INFER_CHUNK", "16")
    G2, _ = _gen("bf16")
    ref = G2.recurrent(x, use_graph=False)
    assert float((ref - graph).abs().max()) == 0.0


def test_main_py_inference_mode_on_a_folder_of_frames(tmp_path, monkeypatch):
    """main.py --mode inference --inferencetype dataset (main.py:141-220): one output per sub-folder of input_dir_LR"""
    import importlib.util
    from PIL import Image
    rng = np.random.default_rng(2)
    for clip, n in (("clip_a", 5), ("clip_b", 3)):
        d = tmp_path / "lr" / clip
        d.mkdir(parents=True)
        for k in range(n):
            Image.fromarray(rng.integers(0, 255, size=(40, 40, 3), dtype=np.uint8)).save(d / f"{k:04d}.png")
    G, _ = _gen("bf16")
    ck = tmp_path / "generator.pt"
    torch.save({"epoch": 0, "model_state_dict": G.state_dict(), "optimizer_state_dict": {}}, ck)
    spec = importlib.util.spec_from_file_location("tg_main_inf", os.path.join(ROOT, "main.py"))
    tg_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tg_main)
    monkeypatch.chdir(tmp_path)
    tg_main.main(["--mode", "inference", "--inferencetype", "dataset", "--input_dir_LR", str(tmp_path / "lr"),
                  "--g_checkpoint", str(ck), "--crop_size", "32", "--videotype", ".gif", "--output_dir", "out"])
    outs = sorted(os.listdir(tmp_path / "out"))
    assert outs == ["output0.gif", "output1.gif"], outs
    # (PIL merges consecutive identical frames, and a default-initialised generator answers nearly every input with the
    # same flat image: the animations hold at least two and at most all of the clip's frames)
    frames = sorted(Image.open(tmp_path / "out" / o).n_frames for o in outs)
    assert 2 <= frames[0] <= 3 and frames[0] <= frames[1] <= 5, frames
    with Image.open(tmp_path / "out" / "output0.gif") as im:
        assert im.size == (128, 128)
    with pytest.raises(ValueError):
        tg_main.main(["--mode", "inference", "--input_dir_LR", str(tmp_path / "lr"), "--output_dir", "out"])  # no checkpoint


def test_gpu_resize_equals_pil_bit_for_bit_and_feeds_main(tmp_path, monkeypatch):
    """data ingest with --tg_gpu_resize: tg_resample_u8 == PIL Image.resize(BILINEAR) + ToTensor exactly (down- and
    up-scaling, non-square frames), and main.py trains from decode-only workers"""
    import importlib.util
    from PIL import Image
    from pytorch_tecogan_amd import resize as R
    rng = np.random.default_rng(3)
    for (H, W, out) in ((180, 320, 32), (180, 320, 128), (37, 23, 32), (24, 24, 96), (32, 32, 32)):
        fr = rng.integers(0, 256, size=(3, H, W, 3), dtype=np.uint8)
        got = R.resize_frames(torch.from_numpy(fr).cuda(), out).cpu().numpy()
        for n in range(3):
            exp = np.asarray(Image.fromarray(fr[n]).resize((out, out), Image.BILINEAR), dtype=np.float32) / 255.0
            assert np.array_equal(got[n], exp.transpose(2, 0, 1)), (H, W, out)
    # end to end: 4 scenes of 120 frames of 48x40 pixels, one epoch through main.py with GPU-side resize
    for scene in range(1000, 1004):
        d = tmp_path / "data" / ("scene_%04d" % scene)
        d.mkdir(parents=True)
        base = rng.integers(0, 256, size=(40, 48, 3), dtype=np.uint8)
        for k in range(120):
            Image.fromarray(np.roll(base, k, axis=1)).save(d / ("col_high_%04d.png" % k))
    spec = importlib.util.spec_from_file_location("tg_main_ingest", os.path.join(ROOT, "main.py"))
    tg_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tg_main)
    monkeypatch.chdir(tmp_path)
    import pytorch_tecogan_amd.train as hip_train
    hip_train._STEPS.clear()
    tg_main.main(["--input_video_dir", str(tmp_path / "data"), "--max_epochs", "1", "--tg_gpu_resize", "true",
                  "--num_resblock", "2", "--discrim_resblocks", "1", "--queue_thread", "2"])
    ck = torch.load(tmp_path / "generator.pt")
    assert float(ck["optimizer_state_dict"]["state"][0]["step"]) == 1.0     # 4 scenes -> one batch of 4 windows
    assert (tmp_path / "Gan_examples.jpg").exists()
    hip_train._STEPS.clear()


def test_ingest_keeps_up_with_the_step(tmp_path):
    """SURVEY 8f row f2 / VERDICT r3 item 6: what main.py's loader delivers from a PNG tree in the reference's layout (408 scenes
    x 120 frames of 320 x 240, code/dataloader.py:46-98; --queue_thread 8 workers, decoded-frame cache, uint8 pinned batches,
    PIL-exact resize on the GPU, device staging one batch ahead) against what the training step consumes, measured by
    tools/ingest_bench.py in one process.  Floors asserted here (the measured figures are in DESIGN.md / profiles/r04_n_ingest.log:
    loader alone 1.37x the step's demand, end to end 0.95x the step-alone rate on the 16-core GPU box; the round-3 loader - no
    frame cache, inline staging - delivered 0.52x and ran the step at 0.50x)."""
    import json
    import subprocess
    out = tmp_path / "ingest.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ingest_bench.py"), "--root", str(tmp_path / "tree"), "--epochs", "2",
                        "--json", str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    need = res["step_alone"]["sequences_per_s"]
    assert need > 500.0
    fast = res["decode_only_gpu_resize"]
    assert fast["batches_per_epoch"] == 102                           # 408 scenes -> 408 windows per epoch (the reference's __len__)
    assert fast["loader_alone"]["sequences_per_s"] >= 0.9 * need, (fast, need)
    assert fast["end_to_end"]["sequences_per_s"] >= 0.8 * need, (fast, need)
    ref = res["reference_pipeline_cpu_resize"]                        # (the reference's own pipeline, cached frames: slower, still fed)
    assert ref["end_to_end"]["sequences_per_s"] >= 0.6 * need, (ref, need)

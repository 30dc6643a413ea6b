"""Data ingest against what the training step consumes.

Generates a PNG tree in the reference's layout (`scene_%04d/col_high_%04d.png`, 120 frames of 320 x 240 per scene - the size of
the UCF-101 clips data/convert2images.py of the reference unpacks; a few scenes hold real files, the rest are directory
symlinks so that the dataset sees the reference's scene count without 50 000 files being written) and measures, with the SAME
DataLoader main.py builds (main.make_train_loader: --queue_thread workers, pinned batches, prefetch):
  1. loader alone: sequences/s for the reference pipeline (decode + two PIL resizes per frame in the workers) and for
     --tg_gpu_resize (workers only decode, PIL-exact resize on the GPU, pytorch_tecogan_amd.resize);
  2. end to end: the same loaders feeding FRVSR_Train (hipGraph replay), steps/s and sequences/s;
  3. the step alone on resident tensors (what it could consume).
  python tools/ingest_bench.py [--scenes 408] [--real 4] [--workers 8] [--epochs 3] [--root /tmp/tg_ingest] [--json out.json]
An epoch is len(dataset) = #scenes sequences (the reference's __len__ quirk, code/dataloader.py:78-79)."""
import argparse, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "code"))


def make_tree(root, scenes, real, frames=120, w=320, h=240, seed=1):
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    for s in range(real):
        d = os.path.join(root, "scene_%04d" % (1000 + s))
        if os.path.isdir(d) and len(os.listdir(d)) >= frames:
            continue
        os.makedirs(d, exist_ok=True)
        # smooth moving content + mild noise: compresses like video frames do (pure noise would not)
        base = rng.random((3, h // 8 + 8, w // 8 + 8)).astype(np.float32)
        for k in range(frames):
            win = base[:, (k // 4) % 8:(k // 4) % 8 + h // 8, (k // 3) % 8:(k // 3) % 8 + w // 8]
            img = np.kron(win, np.ones((1, 8, 8), np.float32))
            img = np.clip(img + 0.03 * rng.standard_normal(img.shape).astype(np.float32), 0, 1)
            Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(os.path.join(d, "col_high_%04d.png" % k),
                                                                                    compress_level=3)
    for s in range(real, scenes):
        d = os.path.join(root, "scene_%04d" % (1000 + s))
        if not os.path.exists(d):
            os.symlink(os.path.join(root, "scene_%04d" % (1000 + s % real)), d)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=408)   # the reference's dataset: 408 scenes (README.md:21)
    ap.add_argument("--real", type=int, default=4)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--root", default="/tmp/tg_ingest")
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-gpu", action="store_true", help="loader-alone rates only (CPU box)")
    ap.add_argument("--inline", action="store_true", help="copy / resize each batch on the step's own stream (the round-3 loop)")
    ap.add_argument("--cache-mb", type=int, default=512, help="per-worker decoded-frame cache (0: off = the round-3 loader)")
    a = ap.parse_args()
    import main as M
    from dataloader import train_dataset, frames_to_batches, device_batches
    gen_s = make_tree(a.root, a.scenes, a.real)
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    res = {"tree": {"scenes": a.scenes, "real_scenes": a.real, "frames_per_scene": 120, "frame": "320x240 RGB png",
                    "generate_s": round(gen_s, 1)}, "usable_cores": cores, "workers": a.workers, "epochs_timed": a.epochs,
           "frame_cache_mb_per_worker": a.cache_mb,
           "device_staging": "inline on the step's stream" if a.inline else "one batch ahead on a side stream (dataloader.device_batches)"}
    dev = None if a.no_gpu else torch.device("cuda", 0)

    def margs(gpu_resize):
        argv = ["--input_video_dir", a.root, "--str_dir", "1000", "--end_dir", str(1000 + a.scenes - 1), "--queue_thread", str(a.workers),
                "--tg_gpu_resize", "true" if gpu_resize else "false", "--tg_frame_cache_mb", str(a.cache_mb)]
        args = M.build_parser().parse_args(argv)
        args.RNN_N = int(args.RNN_N)
        return args

    def batches(loader, gpu_resize, args):
        if dev is not None and not a.inline:
            yield from device_batches(loader, dev, args.crop_size)   # what main.py does: one batch ahead on a side stream
            return
        for batch in loader:
            if gpu_resize:
                if dev is None:
                    yield batch, None
                else:
                    yield frames_to_batches(batch.to(dev, non_blocking=True), args.crop_size)
            else:
                x, y = batch
                yield (x, y) if dev is None else (x.to(dev, non_blocking=True), y.to(dev, non_blocking=True))

    step = None
    if dev is not None:
        import pytorch_tecogan_amd  # noqa: F401
        import bench as B
        from pytorch_tecogan_amd import train as TR
        os.environ["TECOGAN_GRAPH"] = "1"
        sargs = B.default_args("bf16")
        torch.manual_seed(1)
        G, D, og, od = B.build_step_objects(sargs, dev)
        xs, ys = B.synth(4, 10, 32, 1)
        xs, ys = xs.to(dev), ys.to(dev)
        n = [0]

        def step(x, y):
            TR.FRVSR_Train(x, y, sargs, D, G, n[0], 0.0, 0.0, og, od)
            n[0] += 1
        for _ in range(4):
            step(xs, ys)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step(xs, ys)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        res["step_alone"] = {"ms_per_step": round(dt * 1e3, 3), "steps_per_s": round(1 / dt, 1), "sequences_per_s": round(4 / dt, 1)}

    for name, gpu_resize in (("reference_pipeline_cpu_resize", False), ("decode_only_gpu_resize", True)):
        args = margs(gpu_resize)
        ds = train_dataset(args, decode_only=gpu_resize)
        loader = M.make_train_loader(args, ds)
        ent = {"batches_per_epoch": len(loader)}
        for mode in (["loader_alone"] if dev is None else ["loader_alone", "end_to_end"]):
            for _ in batches(loader, gpu_resize, args):   # warm-up epoch: workers start, page cache fills
                pass
            if dev is not None:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb = 0
            for _ in range(a.epochs):
                for x, y in batches(loader, gpu_resize, args):
                    if mode == "end_to_end":
                        step(x, y)
                    nb += 1
            if dev is not None:
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ent[mode] = {"batches_per_s": round(nb / dt, 1), "sequences_per_s": round(4 * nb / dt, 1),
                         "hr_frames_per_s": round(40 * nb / dt, 1)}
        res[name] = ent
        del loader
    if step is not None:
        need = res["step_alone"]["sequences_per_s"]
        for name in ("reference_pipeline_cpu_resize", "decode_only_gpu_resize"):
            res[name]["loader_alone"]["fraction_of_step_demand"] = round(res[name]["loader_alone"]["sequences_per_s"] / need, 3)
            res[name]["end_to_end"]["fraction_of_step_alone"] = round(res[name]["end_to_end"]["sequences_per_s"] / need, 3)
    print(json.dumps(res, indent=1))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

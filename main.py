"""Entry point with the reference's command line (main.py:33-128 of dwight-foster/Pytorch-TecoGAN): same flags, defaults,
start-up side effects and output file names; the training / inference loops call the MI355X implementation through the
reference-named modules in ./code (FRVSR_Train, generator, discriminator).

Extra flags (all optional): --synthetic N trains on N random sequences instead of --input_video_dir (no dataset needed),
--tg_dtype {bf16,fp16,fp32} (fp16: with dynamic loss scaling), --tg_extend true (shapes the reference cannot run, e.g. 64->256 seq-16).
Multi-GPU: launch with torch.distributed.run (one process per GPU).  Every rank builds the same models, rank 0's
parameters / BN buffers / Adam moments are broadcast after construction and after a checkpoint load, a DistributedSampler
gives each rank a disjoint 1/world of the sequences (4 per rank and step, as hard-coded at main.py:227 of the reference),
gradients are averaged over RCCL inside FRVSR_Train, and rank 0 alone writes the outputs.
"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts: pytorch-tecogan_amd/__init__.py says why
sys.path.insert(1, os.path.join(os.path.dirname(os.path.abspath(__file__)), "code"))


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def build_parser():
    p = argparse.ArgumentParser(description="TecoGAN (MI355X-native)")
    a = p.add_argument
    a("--rand_seed", default=1, type=int); a("--input_dir_LR", default="", type=str)
    a("--input_dir_len", default=-1, type=int); a("--input_dir_HR", default="", type=str)
    a("--mode", default="train", type=str); a("--output_dir", default="output"); a("--output_pre", default="")
    a("--output_name", default="output"); a("--output_ext", default="jpg"); a("--summary_dir", default="summary")
    a("--videotype", default=".mp4", type=str); a("--inferencetype", default="dataset", type=str)
    a("--g_checkpoint", default=None); a("--d_checkpoint", default=None)
    a("--num_resblock", type=int, default=16); a("--discrim_resblocks", type=int, default=4)
    a("--discrim_channels", type=int, default=128); a("--pre_trained_model", type=str2bool, default=False)
    a("--vgg_ckpt", default=None); a("--cudaID", default="0", type=str); a("--queue_thread", default=8, type=int)
    a("--RNN_N", default=10); a("--batch_size", default=4, type=int); a("--flip", default=True, type=str2bool)
    a("--random_crop", default=True, type=str2bool); a("--movingFirstFrame", default=True, type=str2bool)
    a("--crop_size", default=32, type=int); a("--input_video_dir", type=str, default="../TrainingDataPath")
    a("--input_video_pre", default="scene", type=str); a("--str_dir", default=1000, type=int)
    a("--end_dir", default=1400, type=int); a("--end_dir_val", default=2050, type=int); a("--max_frm", default=119, type=int)
    a("--vgg_scaling", default=-0.002, type=float); a("--warp_scaling", default=1.0, type=float)
    a("--pingpang", default=False, type=str2bool); a("--pp_scaling", default=1.0, type=float)
    a("--EPS", default=1e-12, type=float); a("--learning_rate", default=0.0001, type=float)
    a("--decay_step", default=250, type=int); a("--decay_rate", default=0.8, type=float)
    a("--stair", default=False, type=str2bool); a("--beta", default=0.9, type=float); a("--adameps", default=1e-8, type=float)
    a("--max_epochs", default=10000000, type=int); a("--ratio", default=0.01, type=float)
    a("--Dt_mergeDs", default=True, type=str2bool); a("--Dt_ratio_0", default=1.0, type=float)
    a("--Dt_ratio_add", default=0.0, type=float); a("--Dt_ratio_max", default=1.0, type=float)
    a("--Dbalance", default=0.4, type=float); a("--crop_dt", default=0.75, type=float)
    a("--D_LAYERLOSS", default=True, type=str2bool)
    a("--synthetic", default=0, type=int, help="train on this many random sequences (no dataset)")
    a("--tg_dtype", default=None, choices=[None, "bf16", "fp16", "fp32"])
    a("--tg_gpu_resize", default="auto", type=lambda v: "auto" if str(v).lower() == "auto" else str2bool(v),
      help="data ingest: the workers only decode the PNGs; the PIL-bilinear resize to the LR / HR sizes runs on the GPU "
           "(bit-exact restatement of PIL's resize).  auto (default): when the dataset's frames share one size, else the "
           "reference's pipeline (decode + two PIL resizes per frame in the workers); tools/ingest_bench.py measures both")
    a("--tg_frame_cache_mb", default=512, type=int,
      help="data ingest: per-worker budget of the decoded-frame cache (a scene's 110 windows overlap in 9 of 10 frames); 0 = off")
    a("--tg_fnet", default=False, type=str2bool,
      help="opt-in (not reference behaviour): the flow comes from the f_net estimator the reference defines and never calls "
           "(main.py:231): gen_flow = up4(4 * f_net(previous LR frame)) instead of the raw-frame pseudo-flow")
    a("--tg_fnet_train", default=False, type=str2bool,
      help="opt-in, with --tg_fnet: train the estimator with its own Adam (the optimiser main.py:244-245 leaves commented out) on "
           "the LR warp loss; fnet.pt is written next to generator.pt (--f_checkpoint to resume)")
    a("--f_checkpoint", default=None, type=str, help="fnet.pt to load with --pre_trained_model (main.py:259-261, commented out there)")
    a("--tg_extend", default=False, type=str2bool,
      help="opt-in extension beyond what the reference can execute: RNN_N outside 9..11 and crop_size != 32 "
           "(discriminator fc sized from crop_size); parity with the reference is undefined there")
    return p


def make_train_loader(args, dataset, sampler=None):
    """The training DataLoader.  Batch size 4 is hard-coded in the reference (main.py:227); per rank here.  drop_last: the step is a
    static launch sequence for ONE batch shape (the reference would run a short last batch through eager PyTorch).  Real data: the
    PNG decode (+ resize, unless --tg_gpu_resize) of code/dataloader.py runs in --queue_thread worker processes (the flag the
    reference defines and never uses, main.py:64) with pinned, prefetched batches, so ingest overlaps the GPU step
    (tools/ingest_bench.py measures what this loader delivers against what the step consumes)."""
    import torch
    workers = 0 if args.synthetic else max(0, min(int(args.queue_thread), len(os.sched_getaffinity(0))))
    return torch.utils.data.DataLoader(dataset, batch_size=4, shuffle=sampler is None, sampler=sampler, drop_last=True,
                                       num_workers=workers, pin_memory=True, persistent_workers=workers > 0,
                                       prefetch_factor=4 if workers > 0 else None)


def main(argv=None):
    args = build_parser().parse_args(argv)
    args.RNN_N = int(args.RNN_N)  # the reference leaves a CLI value as str (main.py:78-79)
    if "LOCAL_RANK" not in os.environ:
        os.environ["CUDA_VISIBLE_DEVICES"] = args.cudaID
    if args.output_dir is None:
        raise ValueError("The output directory is needed")
    for d in (args.output_dir, args.summary_dir):
        os.makedirs(d, exist_ok=True)  # (several ranks start at once)

    import numpy as np
    import torch
    from models import discriminator, generator
    from train import FRVSR_Train

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # one GPU per rank.  Only the gloo test backend can share a device (ranks beyond the visible devices wrap around there);
    # RCCL fails obscurely with two ranks on one GPU, so that is refused here
    from pytorch_tecogan_amd import tuning
    backend = tuning.current().dist_backend  # TECOGAN_DIST_BACKEND: "nccl" is RCCL here; gloo: ranks sharing one GPU (tests)
    local, ndev = int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count()
    if local >= max(1, ndev):
        if world > 1 and backend == "nccl":
            raise RuntimeError(f"LOCAL_RANK {local} but only {ndev} GPU(s) visible: RCCL needs one GPU per rank")
        local %= max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.mode == "inference":
        if args.g_checkpoint is None:
            raise ValueError("The checkpoint file is needed to perform the test")
        if args.inferencetype == "dataset":
            from dataloader import inference_dataset
            loader = torch.utils.data.DataLoader(inference_dataset(args), batch_size=1, shuffle=False)
        elif args.inferencetype == "video":
            from dataloader import video_frames
            loader = [video_frames(args)]
        else:
            raise ValueError("Invalid data type entered. Please use either video or dataset.")
        G = generator(3, args=args).to(dev)
        G.load_state_dict(torch.load(args.g_checkpoint, map_location=dev)["model_state_dict"])
        from ops import save_as_gif
        for batch_idx, r_inputs in enumerate(loader):
            out = G.recurrent(r_inputs.to(dev), use_graph=True)  # whole recurrence on device, per-frame hipGraph
            save_as_gif(out[0].cpu(), f"./{args.output_dir}/output{batch_idx}{args.videotype}")
        return

    if args.mode != "train":
        raise ValueError("mode must be train or inference")

    if args.synthetic:
        rng = np.random.default_rng(args.rand_seed)  # the same sequences on every rank; the sampler shards them
        T, cs = args.RNN_N, args.crop_size
        data = [(torch.from_numpy(rng.random((T, 3, cs, cs), dtype=np.float32)),
                 torch.from_numpy(rng.random((T, 3, 4 * cs, 4 * cs), dtype=np.float32))) for _ in range(args.synthetic)]
        dataset = data
    else:
        from dataloader import train_dataset
        dataset = train_dataset(args, decode_only=args.tg_gpu_resize)
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True,
                                                                  seed=args.rand_seed, drop_last=True)
    loader = make_train_loader(args, dataset, sampler)
    if len(loader) == 0:
        raise ValueError(f"{len(dataset)} training sequences give rank {rank} of {world} no full batch of 4: nothing would "
                         "be trained")

    G, D = generator(3, args=args).to(dev), discriminator(args=args).to(dev)
    lr_d = args.learning_rate * (1.0 if args.Dt_mergeDs else 0.3)
    opt_d = torch.optim.Adam(D.parameters(), lr_d, betas=(args.beta, 0.999), eps=args.adameps)
    opt_g = torch.optim.Adam(G.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
    sch_d = torch.optim.lr_scheduler.StepLR(opt_d, args.decay_step, args.decay_rate)
    sch_g = torch.optim.lr_scheduler.StepLR(opt_g, args.decay_step, args.decay_rate)
    Fn = opt_f = sch_f = None
    if args.tg_fnet or args.tg_fnet_train:   # the estimator the reference leaves commented out (main.py:231,244-245,249)
        from models import f_net
        Fn = f_net().to(dev)
        args.tg_fnet = Fn
        if args.tg_fnet_train:
            opt_f = torch.optim.Adam(Fn.parameters(), args.learning_rate, betas=(args.beta, 0.999), eps=args.adameps)
            sch_f = torch.optim.lr_scheduler.StepLR(opt_f, args.decay_step, args.decay_rate)
            args.tg_fnet_optimizer = opt_f
    else:
        args.tg_fnet = None
    epoch0 = 0
    if args.pre_trained_model:
        g_ck = torch.load(args.g_checkpoint, map_location=dev)
        G.load_state_dict(g_ck["model_state_dict"])
        opt_g.load_state_dict(g_ck["optimizer_state_dict"])
        epoch0 = g_ck["epoch"]
        d_ck = torch.load(args.d_checkpoint, map_location=dev)
        D.load_state_dict(d_ck["model_state_dict"])
        opt_d.load_state_dict(d_ck["optimizer_state_dict"])
        if Fn is not None and args.f_checkpoint:
            f_ck = torch.load(args.f_checkpoint, map_location=dev)
            Fn.load_state_dict(f_ck["model_state_dict"])
            if opt_f is not None and "optimizer_state_dict" in f_ck:
                opt_f.load_state_dict(f_ck["optimizer_state_dict"])
        if g_ck.get("tg_scaler"):  # fp16 mode: the dynamic loss scale is part of the training state (extra key)
            from pytorch_tecogan_amd.train import load_loss_scaler_state
            load_loss_scaler_state(g_ck["tg_scaler"])
    if world > 1:  # replicas start equal: every rank drew its own initial weights above
        from pytorch_tecogan_amd import parallel
        parallel.broadcast_state((G, D) + ((Fn,) if Fn is not None else ()), (opt_g, opt_d) + ((opt_f,) if opt_f else ()))

    since = time.time()
    for e in range(epoch0, args.max_epochs):
        g_loss = d_loss = 0.0
        output = inputs = targets = None
        t_epoch, n_batches = time.time(), 0
        if sampler is not None:
            sampler.set_epoch(e)
        # batches arrive on the device one step ahead (copy + GPU-side resize on a side stream: dataloader.device_batches)
        from dataloader import device_batches
        for batch_idx, (inputs, targets) in enumerate(device_batches(loader, dev, args.crop_size)):
            output = FRVSR_Train(inputs, targets, args, D, G, batch_idx, 0.0, 0.0, opt_g, opt_d)
            g_loss = g_loss + (output.gen_loss.data - g_loss) / (batch_idx + 1)   # running means stay on the device
            d_loss = d_loss + (output.d_loss.data - d_loss) / (batch_idx + 1)
            n_batches += 1
        # the fused Adam launch updates the weights without ever calling optimizer.step(): tell the schedulers so, or torch warns every
        # epoch that lr_scheduler.step() came before optimizer.step() (the LR itself is read from param_groups at every step)
        for o_ in (opt_d, opt_g) + ((opt_f,) if sch_f is not None else ()):
            if hasattr(o_, "_opt_called"):   # (a torch internal - set by lr_scheduler's patch of optimizer.step(): touched only where it exists)
                o_._opt_called = True
        sch_d.step()
        sch_g.step()
        if sch_f is not None:
            sch_f.step()
        if world > 1:
            from pytorch_tecogan_amd import parallel
            if not parallel.replicas_equal((G, D) + ((Fn,) if Fn is not None else ())):   # (the estimator too when it trains)
                raise RuntimeError(f"data-parallel replicas diverged (rank {rank}, epoch {e + 1})")
            if rank == 0:
                print(f"replica check ok ({world} ranks)")
        if rank == 0 and output is not None:
            torch.cuda.synchronize()
            dt_e = max(time.time() - t_epoch, 1e-9)   # ingest + steps of this epoch (before the per-epoch samples / checkpoints below)
            print(f"epoch {e + 1}: {n_batches} steps in {dt_e:.3f} s = {n_batches / dt_e:.1f} steps/s = "
                  f"{4 * n_batches / dt_e:.1f} sequences/s per rank")
            print("Epoch: {}".format(e + 1))
            print("\nGenerator loss is: {} \nDiscriminator loss is: {}".format(float(g_loss), float(d_loss)))
            print(f"\nGenerator lr is: {opt_g.param_groups[0]['lr']}, Discriminator lr is: {opt_d.param_groups[0]['lr']}")
            from ops import save_as_gif, save_image  # per-epoch samples of main.py:281-294
            idx = np.random.randint(0, targets.shape[0])
            save_as_gif(output.gen_output[idx][:args.RNN_N].cpu(), "gan.gif")
            save_as_gif(targets[idx].cpu(), "real.gif")
            save_as_gif(inputs[idx].cpu(), "original.gif")
            cs, n = args.crop_size, targets.shape[0] * args.RNN_N
            save_image(output.gen_output.reshape(-1, 3, cs * 4, cs * 4), "Gan_examples.jpg")
            save_image(targets.reshape(n, 3, cs * 4, cs * 4), "real_image.jpg")
            save_image(inputs.reshape(n, 3, cs, cs), "original_image.jpg")
            print("\nSaving model...")
            from pytorch_tecogan_amd.train import loss_scaler_state, sync_optimizer_steps
            sync_optimizer_steps(opt_g, opt_d)   # fp16: Adam's step count without the updates skipped on overflow
            g_state = {"epoch": e, "model_state_dict": G.state_dict(), "optimizer_state_dict": opt_g.state_dict()}
            if loss_scaler_state() is not None:
                g_state["tg_scaler"] = loss_scaler_state()
            torch.save(g_state, "generator.pt")
            torch.save({"model_state_dict": D.state_dict(), "optimizer_state_dict": opt_d.state_dict()}, "discrim.pt")
            if opt_f is not None:
                torch.save({"model_state_dict": Fn.state_dict(), "optimizer_state_dict": opt_f.state_dict()}, "fnet.pt")
            el = time.time() - since
            print("\nTraining complete in {:.0f}m {:.0f}s".format(el // 60, el % 60))


if __name__ == "__main__":
    main()

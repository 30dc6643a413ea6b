"""Data ingest with the reference's semantics (code/dataloader.py:15-98) on PIL + torch only (torchvision is not needed):
  * train_dataset: every scene folder `<pre>_%04d` with >= 120 files contributes 110 sliding windows of 10 frames
    (`col_high_%04d.png`), but __len__ is the number of SCENES (reference quirk, :78-79), so an epoch visits the first
    len(scenes) windows; frames are resized to 4*crop (HR) and crop (LR) with bilinear filtering; frame 0 only gets an
    independent RandomResizedCrop for LR and HR (:91-93).
  * inference_dataset: one item per sub-folder, all frames resized to crop x crop.
Not on the timed path; kept so that main.py is a drop-in.  Measured against the step's demand by
tools/ingest_bench.py (profiles/r04_*_ingest.log): the reference's pipeline - 10 PNG decodes + 20 PIL resizes per sequence in the
workers - delivers a fraction of the ~1000 sequences/s one MI355X consumes, so the defaults are: decoded frames cached per
worker (the 110 windows of a scene overlap in 9 of 10 frames, and an epoch only visits the first len(scenes) windows), workers
that only decode, uint8 batches pinned by the loader, and the PIL-exact resize on the GPU."""
import math
import os
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


class FrameCache:
    """Decoded RGB frames (PIL images or uint8 arrays) of one worker process, least recently used out first, bounded in bytes.
    The training windows are sliding windows of 10 over a scene's 120 frames: every frame is part of 10 windows, and the
    reference's __len__ quirk makes an epoch visit the first len(scenes) windows only - a handful of scenes.  Decoding each PNG
    once per worker instead of once per visit removes most of the ingest's CPU time; a dataset larger than the budget simply
    thrashes the cache and costs what it did before."""

    def __init__(self, budget_mb):
        import collections
        self.budget, self.used, self.items = int(budget_mb) << 20, 0, collections.OrderedDict()
        self.hits = self.misses = 0

    def get(self, path, as_array):
        key = (path, as_array)
        ent = self.items.get(key)
        if ent is not None:
            self.items.move_to_end(key)
            self.hits += 1
            return ent[0]
        self.misses += 1
        img = Image.open(path)
        if as_array or img.mode == "RGB":
            img = img.convert("RGB")     # (decode-only workers: the GPU resize is PIL's bilinear on RGB - "auto" picks it for RGB datasets only)
        else:
            img.load()                   # the reference resizes the frame AS OPENED and converts afterwards (_to_tensor): a palette or
                                         # grey frame is resized in its own mode (PIL uses NEAREST for 'P'), so it is cached unconverted
        val = np.asarray(img) if as_array else img
        size = img.size[0] * img.size[1] * 3
        if self.budget > 0:
            self.items[key] = (val, size)
            self.used += size
            while self.used > self.budget and len(self.items) > 1:
                _, (_, sz) = self.items.popitem(last=False)
                self.used -= sz
        return val


def _to_tensor(img):
    a = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).contiguous()


def _resize(img, size):
    return img.resize((size, size), Image.BILINEAR)


def _random_resized_crop(t, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision.transforms.RandomResizedCrop(size) on a CHW tensor: random area/aspect crop, bilinear resize back."""
    _, H, W = t.shape
    area = H * W
    i = j = 0
    h, w = H, W
    for _ in range(10):
        target = area * random.uniform(*scale)
        ar = math.exp(random.uniform(math.log(ratio[0]), math.log(ratio[1])))
        ww, hh = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < ww <= W and 0 < hh <= H:
            i, j, h, w = random.randint(0, H - hh), random.randint(0, W - ww), hh, ww
            break
    crop = t[:, i:i + h, j:j + w].unsqueeze(0)
    return torch.nn.functional.interpolate(crop, size=(size, size), mode="bilinear", align_corners=False,
                                           antialias=True)[0]


class inference_dataset(Dataset):
    def __init__(self, args):
        filedir = args.input_dir_LR
        self.args = args
        if (filedir is None) or (not os.path.exists(filedir)):
            if (args.input_dir_HR is None) or (not os.path.exists(args.input_dir_HR)):
                raise ValueError("Input directory not found")
            filedir = args.input_dir_HR
        self.filedir = filedir
        self.items = os.listdir(filedir)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        d = os.path.join(self.filedir, self.items[idx])
        frames = [_to_tensor(_resize(Image.open(os.path.join(d, f)), self.args.crop_size)) for f in os.listdir(d)]
        return torch.stack(frames, dim=0)


def frames_to_batches(frames_u8, crop_size, first_frame_crop=True):
    """GPU half of the ingest when train_dataset(decode_only=True) is used: decoded uint8 frames (B,T,H,W,3) on the device
    -> (LR (B,T,3,cs,cs), HR (B,T,3,4cs,4cs)) fp32, PIL-BILINEAR resize bit for bit (pytorch_tecogan_amd.resize) and the
    independent RandomResizedCrop of frame 0 of each sequence (:91-93 of the reference's dataloader)."""
    from pytorch_tecogan_amd.resize import resize_frames
    B, T, H, W, _ = frames_u8.shape
    flat = frames_u8.reshape(B * T, H, W, 3)
    lr = resize_frames(flat, crop_size).view(B, T, 3, crop_size, crop_size)
    hr = resize_frames(flat, 4 * crop_size).view(B, T, 3, 4 * crop_size, 4 * crop_size)
    if first_frame_crop:
        for b in range(B):
            hr[b, 0] = _random_resized_crop(hr[b, 0], 4 * crop_size)
            lr[b, 0] = _random_resized_crop(lr[b, 0], crop_size)
    return lr, hr


def device_batches(loader, device, crop_size):
    """(inputs, targets) on `device`, one batch AHEAD of the consumer: the host-to-device copy of batch i + 1 and (decode-only
    workers) its resize run on a side stream while the training step of batch i runs on the caller's stream.  Taken inline on the
    caller's stream - which is lane A of the step - the 8-MB copy and the resize kernels cost 0.5 ms of every 4-ms step
    (tools/ingest_bench.py, profiles/r04_m_ingest.log: 878 of the 993 sequences/s the step takes).  `loader` may yield either
    batch form (uint8 frames, or the reference pipeline's [LR, HR] pair)."""
    side = torch.cuda.Stream(device=device)

    def stage(batch):
        with torch.cuda.stream(side):
            if torch.is_tensor(batch):
                x, y = frames_to_batches(batch.to(device, non_blocking=True), crop_size)
            else:
                x, y = batch[0].to(device, non_blocking=True), batch[1].to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        return x, y, ev

    it = iter(loader)
    try:
        nxt = stage(next(it))
    except StopIteration:
        return
    while nxt is not None:
        x, y, ev = nxt
        try:
            nxt = stage(next(it))
        except StopIteration:
            nxt = None
        cur = torch.cuda.current_stream(device)
        cur.wait_event(ev)
        x.record_stream(cur)   # (allocated on the side stream: the allocator must not hand the memory out while the step reads it)
        y.record_stream(cur)
        yield x, y


class train_dataset(Dataset):
    def __init__(self, args, decode_only=False):
        """decode_only: __getitem__ returns the window's DECODED frames as one uint8 tensor (10,H,W,3); the resize to the LR
        and HR sizes then happens on the GPU (frames_to_batches) - the PNG decode is all the workers do.  "auto": decode-only when
        the dataset's frames share one size and are RGB (probed on the first frame of every scene), else the reference pipeline.
        args.tg_frame_cache_mb (default 512): per-worker budget of the decoded-frame cache, 0 = off."""
        self.decode_only = decode_only
        self.cache = None   # per process: created on first use, i.e. inside each DataLoader worker
        self.cache_mb = int(getattr(args, "tg_frame_cache_mb", 512))
        if args.input_video_dir == "":
            raise ValueError("Video input directory input_video_dir is not provided")
        if not os.path.exists(args.input_video_dir):
            raise ValueError("Video input directory not found")
        self.args = args
        self.scenes = 0
        self.windows = []
        for dir_i in range(args.str_dir, args.end_dir + 1):
            d = os.path.join(args.input_video_dir, "%s_%04d" % (args.input_video_pre, dir_i))
            if not os.path.exists(d):
                continue
            if len(os.listdir(d)) < 120:
                print("Skip %s, since folder doesn't contain enough frames!" % d)
                continue
            frames = [os.path.join(d, "col_high_%04d.png" % k) for k in range(args.max_frm + 1)]
            self.scenes += 1
            self.windows += [frames[i:i + 10] for i in range(110)]
        if self.decode_only == "auto":
            # one header read per SCENE (a scene's frames share a size; 110 windows per scene): a differently sized or non-RGB scene
            # in the middle of the dataset must not surface as a collate error halfway through an epoch
            sizes = set()
            for w in self.windows[::110]:
                with Image.open(w[0]) as im:
                    sizes.add((im.size, im.mode))
            self.decode_only = len(sizes) == 1 and next(iter(sizes))[1] == "RGB"

    def _frame(self, path, as_array):
        if self.cache is None:
            self.cache = FrameCache(self.cache_mb)
        return self.cache.get(path, as_array)

    def __len__(self):
        return self.scenes  # reference quirk: scene count, not window count

    def __getitem__(self, idx):
        cs = self.args.crop_size
        if self.decode_only:
            return torch.from_numpy(np.stack([self._frame(p, True) for p in self.windows[idx]]))
        lr, hr = [], []
        for i, path in enumerate(self.windows[idx]):
            img = self._frame(path, False)
            h_t, l_t = _to_tensor(_resize(img, 4 * cs)), _to_tensor(_resize(img, cs))
            if i == 0:
                h_t, l_t = _random_resized_crop(h_t, 4 * cs), _random_resized_crop(l_t, cs)
            lr.append(l_t)
            hr.append(h_t)
        return [torch.stack(lr).float(), torch.stack(hr).float()]


def video_frames(args):
    """inferencetype == 'video' (main.py:145-161): needs OpenCV, which is optional."""
    import cv2
    cap = cv2.VideoCapture(args.input_dir_LR)
    frames = []
    for _ in range(int(cap.get(cv2.CAP_PROP_FRAME_COUNT))):
        ok, frame = cap.read()
        if not ok:
            continue
        frame = cv2.resize(cv2.cvtColor(frame, cv2.COLOR_BGR2RGB), (args.crop_size, args.crop_size),
                           interpolation=cv2.INTER_AREA)
        frames.append(torch.from_numpy(frame.astype(np.float32) / 255.0).permute(2, 0, 1))
    cap.release()
    return torch.stack(frames, dim=0).unsqueeze(0)

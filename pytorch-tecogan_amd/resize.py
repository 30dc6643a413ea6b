"""GPU-side frame resize of the data ingest: PIL's BILINEAR `Image.resize` - what the reference's
`torchvision.transforms.functional.resize` does on the PIL frames it opens (code/dataloader.py:84-88 of the reference) -
restated bit for bit: two passes (horizontal, then vertical) of an anti-aliased triangle filter in 8-bit fixed point, with
a uint8 rounding between the passes (Pillow libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc).  The coefficient tables are built here on the host (pure integer/float
host logic, checked against PIL itself on the CPU); the two passes run as HIP kernels (csrc/warp.hip: tg_resample_u8) on
uint8 frames that were only DECODED on the CPU, and the second pass writes the fp32 NCHW tensors FRVSR_Train consumes."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c


def pil_bilinear_coeffs(in_size, out_size):
    """(bounds int32 [out][2] = (first input index, count), coeffs int32 [out][ksize]) of one resize axis, exactly as
    precompute_coeffs + normalize_coeffs_8bpc build them for the bilinear (triangle, support 1) filter over the full axis."""
    if in_size <= 0 or out_size <= 0:
        raise ValueError("sizes must be positive")
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.empty(xmax, dtype=np.float64)
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
        ww = float(w.sum()) if xmax else 0.0   # (sequential double sum in C; numpy's pairwise sum of <= a few terms agrees,
        acc = 0.0                              #  but keep the C order to be exact)
        for x in range(xmax):
            acc += w[x]
        ww = acc
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """one 8-bit pass along `axis` of a uint8 array [..., H, W, C]: out = clip8((2^(P-1) + sum in * k) >> P)"""
    img = np.moveaxis(img, axis, -2)  # [..., other, L, C] -> resample L
    out = np.empty(img.shape[:-2] + (bounds.shape[0], img.shape[-1]), dtype=np.uint8)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(img.shape[:-2] + (img.shape[-1],), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += img[..., x0 + x, :].astype(np.int64) * int(kk[xx, x])
        out[..., xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -2, axis)


def resize_u8_reference(img, out_h, out_w):
    """numpy emulation of Image.resize((out_w, out_h), BILINEAR) on a uint8 [..., H, W, C] array (the checker of the HIP
    kernels; itself checked against PIL in tests/test_host_cpu.py)"""
    H, W = img.shape[-3], img.shape[-2]
    if (H, W) == (out_h, out_w):
        return img.copy()
    if W != out_w:
        bw, kw = pil_bilinear_coeffs(W, out_w)
        img = _pass(img, bw, kw, -2)
    if H != out_h:
        bh, kh = pil_bilinear_coeffs(H, out_h)
        img = _pass(img, bh, kh, -3)
    return img


class ResizePlan:
    """device-resident coefficient tables of one (in_h, in_w) -> (out_h, out_w) resize"""

    def __init__(self, in_h, in_w, out_h, out_w, device):
        import torch
        self.in_h, self.in_w, self.out_h, self.out_w = in_h, in_w, out_h, out_w
        bw, kw = pil_bilinear_coeffs(in_w, out_w)
        bh, kh = pil_bilinear_coeffs(in_h, out_h)
        self.kw_n, self.kh_n = kw.shape[1], kh.shape[1]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        self.bw, self.kw, self.bh, self.kh = t(bw), t(kw), t(bh), t(kh)


_PLANS = {}


def resize_frames(frames_u8, out_size, out=None):
    """uint8 device tensor [N,H,W,3] (decoded frames) -> fp32 [N,3,out,out] in [0,1]: PIL-BILINEAR resize + ToTensor
    (code/dataloader.py `_to_tensor(_resize(img, size))`), on the GPU."""
    import torch
    from . import kernels as K
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3:
        raise ValueError("expected uint8 [N,H,W,3] frames")
    N, H, W, _ = frames_u8.shape
    key = (H, W, out_size, out_size, frames_u8.device)
    plan = _PLANS.get(key)
    if plan is None:
        plan = _PLANS[key] = ResizePlan(H, W, out_size, out_size, frames_u8.device)
    if out is None:
        out = torch.empty(N, 3, out_size, out_size, dtype=torch.float32, device=frames_u8.device)
    K.resample_u8(frames_u8.contiguous(), plan, out)
    return out

"""Helpers of the reference's code/ops.py that sit on (or next to) the hot path, same names and argument meaning.
Tensor math that the step uses runs on HIP kernels; image/GIF writers stay optional host utilities."""
import os

import numpy as np
import torch
import torch.nn as nn

from . import kernels as K


def preprocess(image):
    """[0,1] -> [-1,1]   (code/ops.py:24-26)"""
    return image * 2 - 1


def deprocess(image):
    """[-1,1] -> [0,1]   (code/ops.py:29-31)"""
    return (image + 1) / 2


def preprocessLr(image):
    return image  # identity (code/ops.py:34-36)


def deprocessLr(image):
    return image  # identity (code/ops.py:39-41)


def upscale_four(inputs):
    """nn.Upsample(scale_factor=4, mode='bilinear') (code/ops.py:98-100) on the HIP bilinear kernel; fp32 NCHW."""
    if not inputs.is_cuda:
        raise RuntimeError("upscale_four runs on the HIP kernel: pass a device tensor")
    x = inputs.contiguous().float()
    N, C_, h, w = x.shape
    out = torch.empty(N, C_, 4 * h, 4 * w, dtype=torch.float32, device=x.device)
    idx = torch.arange(N * C_, dtype=torch.int64, device=x.device)
    K.up4_planes(x, idx * (h * w), out, idx * (16 * h * w), N * C_, h, w)
    return out


def bicubic_four(inputs):
    """nn.Upsample(scale_factor=4, mode='bicubic') (code/ops.py:103-105); dead code in the reference, kept for API parity."""
    return nn.functional.interpolate(inputs, scale_factor=4, mode="bicubic")


def compute_psnr(ref, target):
    """code/ops.py:130-139: expects 0..255-scaled tensors."""
    diff = target.float() - ref.float()
    mse = (diff * diff).sum() / diff.numel()
    return 10.0 * (torch.log(255.0 * 255.0 / mse) / torch.log(torch.tensor(10.0, device=mse.device)))


# layer factories (code/ops.py:45-88) - kept because `from ops import *` is part of the reference surface
def conv2_tran(input_channels, kernel=3, output_channel=64, stride=1, use_bias=True, output_padding=0):
    return nn.ConvTranspose2d(input_channels, output_channel, kernel, stride, padding=int((kernel - 1) / 2),
                              bias=bool(use_bias), output_padding=output_padding)


def conv2(batch_input, kernel=3, output_channels=64, stride=1, use_bias=True):
    return nn.Conv2d(batch_input, output_channels, kernel, stride, padding=int((kernel - 1) / 2), bias=bool(use_bias))


def lrelu(alphas):
    return nn.LeakyReLU(negative_slope=alphas)


def batchnorm(inputs, is_training):
    return nn.BatchNorm2d(inputs, eps=0.001)


def maxpool(kernel_size=(2, 2)):
    return nn.MaxPool2d(kernel_size)


def denselayer(inputs, output_size):
    fc = nn.Linear(inputs, output_size)
    nn.init.xavier_uniform_(fc.weight)
    return fc


def VGG19(args=None):
    """code/ops.py:146: the VGG-19 feature extractor of the opt-in VGG loss (models.VGG19, conv stack up to Conv4_4)"""
    from .models import VGG19 as _VGG19
    return _VGG19(args)


def load_ckpt(checkpoint, model):
    return model.load_state_dict(torch.load(checkpoint))


def save_as_gif(tensor, filepath):
    """code/ops.py:234-237: (T,3,H,W) in [0,1] -> an animation at `filepath` through imageio.  Without imageio (it is
    optional here) the frames are written as an animated GIF with PIL - next to `filepath` with a .gif extension when the
    requested container is not a GIF (main.py:220 asks for .mp4 by default) - and the path written is returned."""
    img = np.transpose((tensor.float().numpy() * 255.0).astype(np.uint8), (0, 2, 3, 1))
    try:
        import imageio
    except ImportError:
        from PIL import Image
        path = filepath if str(filepath).lower().endswith(".gif") else os.path.splitext(str(filepath))[0] + ".gif"
        frames = [Image.fromarray(f) for f in img]
        frames[0].save(path, save_all=True, append_images=frames[1:], duration=40, loop=0)
        return path
    imageio.mimsave(filepath, img)
    return filepath


def save_image(tensor, fp, nrow=8, padding=2):
    """torchvision.utils.save_image with its defaults (main.py:287-294: the per-epoch Gan_examples.jpg / real_image.jpg /
    original_image.jpg grids): (N,3,H,W) in [0,1] -> one image, `nrow` tiles per row, `padding` black pixels between
    tiles, values *255 + 0.5 clamped to uint8.  Written with PIL (torchvision is not a dependency here)."""
    from PIL import Image
    t = tensor.detach().float().cpu()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    n, c, h, w = t.shape
    if c == 1:
        t = t.expand(n, 3, h, w)
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    if n == 1:  # make_grid returns a single image unpadded
        grid = t[0]
    else:
        H, W = h + padding, w + padding
        grid = torch.zeros(3, H * ymaps + padding, W * xmaps + padding)
        for k in range(n):
            yy, xx = k // xmaps, k % xmaps
            grid[:, yy * H + padding:yy * H + padding + h, xx * W + padding:xx * W + padding + w] = t[k]
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(fp)


def save_img(out_path, img):
    """code/ops.py:240-242 (needs OpenCV, which is optional)."""
    import cv2
    cv2.imwrite(out_path, np.clip(img * 255.0, 0, 255).astype(np.uint8)[:, :, ::-1])

"""Helpers of the reference's code/ops.py that sit on (or next to) the hot path, same names and argument meaning.
Tensor math that the step uses runs on HIP kernels; image/GIF writers stay optional host utilities."""
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


def load_ckpt(checkpoint, model):
    return model.load_state_dict(torch.load(checkpoint))


def save_as_gif(tensor, filepath):
    """code/ops.py:234-237 (needs imageio, which is optional)."""
    import imageio
    img = tensor.float().numpy() * 255.0
    imageio.mimsave(filepath, np.transpose(img.astype(np.uint8), (0, 2, 3, 1)))


def save_img(out_path, img):
    """code/ops.py:240-242 (needs OpenCV, which is optional)."""
    import cv2
    cv2.imwrite(out_path, np.clip(img * 255.0, 0, 255).astype(np.uint8)[:, :, ::-1])

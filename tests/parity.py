"""Shared assertions of the GPU parity tests.

`assert_rel_l2`: whole-tensor relative L2 error ||got - ref|| / ||ref||.  The element-wise `rtol = atol = 2e-2` gate of the 16-bit kernel
tests admits 3-6 % of sigma per element on outputs of sigma 0.3-0.8: a wrong halo tap is caught, a systematic 1 % bias (a dropped k-step
of 18) on small-magnitude outputs might not be.  Output rounding alone gives a relative L2 of ~2^-9 / sqrt(3) = 1.1e-3 (bf16, 8 bits of
mantissa) or 1.4e-4 (fp16); a dropped k-step of 18 gives ~0.24, a 1 % bias 1e-2.  Bounds: 4e-3 (bf16), 1.5e-3 (fp16), 1e-4 (fp32)."""
import torch

REL_L2 = {torch.bfloat16: 4e-3, torch.float16: 1.5e-3, torch.float32: 1e-4}


def rel_l2(got, ref):
    got, ref = got.double(), ref.double()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def assert_rel_l2(got, ref, dt, what="", scale=1.0):
    """scale: multiplies the bound (operands that were themselves rounded twice, sums of several rounded tensors)"""
    e, bound = rel_l2(got, ref), REL_L2[dt] * scale
    assert e <= bound, f"{what} relative L2 error {e:.3e} > {bound:.1e} ({dt})"

"""Bilinear filter bank for the FCN up-sampling layers: the host-side counterpart of
/root/reference/utils/upsampling.py:6-46 (same three entry points, same numbers; checked against golden vectors
generated from the reference in tests/golden/).  The device kernels only need the 2-D tent (`upsample_filt`), because
the [k,k,C,C] bank is zero off the channel diagonal (a depthwise transposed conv)."""
import numpy as np


def get_kernel_size(factor):
    """k = 2f - f mod 2"""
    return 2 * factor - factor % 2


def upsample_filt(size):
    """float64 [size,size] tent: (1-|i-c|/F)(1-|j-c|/F), F=(size+1)//2, c=F-1 (odd size) or F-0.5 (even)."""
    half = (size + 1) // 2
    centre = half - 1 if size % 2 == 1 else half - 0.5
    ramp = 1.0 - np.abs(np.arange(size) - centre) / half
    return ramp[:, None] * ramp[None, :]


def bilinear_upsample_weights(factor, number_of_classes):
    """float32 [k,k,C,C], tent on the channel diagonal, zero elsewhere (TF conv2d_transpose filter layout)."""
    k = get_kernel_size(factor)
    w = np.zeros((k, k, number_of_classes, number_of_classes), dtype=np.float32)
    idx = np.arange(number_of_classes)
    w[:, :, idx, idx] = upsample_filt(k)[:, :, None]
    return w

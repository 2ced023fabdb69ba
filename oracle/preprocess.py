# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- CPU restatement of the eval input pipeline.

PARITY UNPINNED: the reference calls cv2.resize (yolo/data/transform.py:173-174) and OpenCV is not importable in
this image (nor on the GPU box), and the reference holds no fixture for it.  This restates OpenCV's published 8-bit
INTER_LINEAR algorithm (imgproc/resize.cpp: `resize` coefficient set-up, HResizeLinear, VResizeLinear<uchar> with
FixedPtCast, INTER_RESIZE_COEF_BITS = 11, and the INTER_LINEAR -> INTER_AREA switch for exact 2x downscale);
a float bilinear cross-check (tests) bounds it to 1 LSB.  Channel flip: transform.py:437; /255 + CHW: :461.
"""
import numpy as np


def _coeffs(dst_n, src_n, clamp_frac):
    scale = 1.0 / (float(dst_n) / float(src_n))
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:                                   # x: OpenCV zeroes the fraction at the borders
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= src_n - 1
        f[hi] = 0; s[hi] = src_n - 1
    c0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
    c1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
    return s, c0, c1


def resize_linear_u8(img, S):
    """img uint8 [H,W,3] -> uint8 [S,S,3], cv2.resize(img, (S,S)) with the default INTER_LINEAR."""
    h, w = img.shape[:2]
    a = img.astype(np.int64)
    if w == 2 * S and h == 2 * S:
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _coeffs(S, w, True)
    sy, b0, b1 = _coeffs(S, h, False)
    sx1 = np.minimum(sx + 1, w - 1)
    hrow = a[:, sx, :] * a0[None, :, None] + a[:, sx1, :] * a1[None, :, None]        # [H,S,3] x2048
    y0 = np.clip(sy, 0, h - 1); y1 = np.clip(sy + 1, 0, h - 1)
    v = (((b0[:, None, None] * (hrow[y0] >> 4)) >> 16) + ((b1[:, None, None] * (hrow[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def val_input(img_bgr, S):
    """-> fp32 [3,S,S] RGB in [0,1] and img_info, as Transform(is_train=False) returns them."""
    sized = resize_linear_u8(img_bgr[:, :, ::-1], S)
    x = np.ascontiguousarray(sized.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    return x, [img_bgr.shape[0], img_bgr.shape[1], S, S]

# -*- coding: utf-8 -*-
"""Eval input pipeline on the GPU.

Mirrors what the reference does per validation image on the CPU
(yolo/data/transform.py:429-448 `_get_val_item`, :173-187 `image_resize`, :461): BGR->RGB, cv2.resize to
S x S (INTER_LINEAR, 8-bit), HWC->CHW, /255 -- here one HIP kernel per image writing straight into its slot
of the [B,3,S,S] batch, so batches larger than the reference's bs=1 stop being bound by the host.
`img_info` keeps the reference's meaning: [src_h, src_w, dst_h, dst_w].
"""
import numpy as np
import torch

from ..._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def image_resize_into(img, dst, swap_rb=True):
    """img: uint8 [H,W,3] (numpy array or torch tensor, any device); dst: fp32 [3,S,S] view on the GPU."""
    assert dst.is_cuda and dst.dtype == torch.float32 and dst.dim() == 3 and dst.shape[0] == 3
    assert dst.shape[1] == dst.shape[2], 'the reference resizes to a square'
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3, 'expect uint8 HWC, 3 channels'
    if not img.is_cuda:
        img = img.to(dst.device, non_blocking=True)
    if img.stride(2) != 1 or img.stride(1) != 3:
        img = img.contiguous()
    h, w = int(img.shape[0]), int(img.shape[1])
    check(lib().y4_preprocess_u8_f32(img.data_ptr(), h, w, img.stride(0), 1 if swap_rb else 0,
                                     dst.data_ptr(), dst.stride(0), dst.stride(1), dst.stride(2),
                                     int(dst.shape[1]), _stream()))
    return [h, w, int(dst.shape[1]), int(dst.shape[2])]


def val_transform(img, img_size, device=None):
    """One image -> (input [3,S,S] fp32 on the GPU, img_info); transform.py:429-448 + :461."""
    device = device or torch.device('cuda', torch.cuda.current_device())
    out = torch.empty((3, img_size, img_size), dtype=torch.float32, device=device)
    return out, image_resize_into(img, out)


def val_batch(imgs, img_size, device=None):
    """List of BGR images of any sizes -> ([B,3,S,S] fp32 batch, list of img_info)."""
    device = device or torch.device('cuda', torch.cuda.current_device())
    out = torch.empty((len(imgs), 3, img_size, img_size), dtype=torch.float32, device=device)
    infos = [image_resize_into(im, out[i]) for i, im in enumerate(imgs)]
    return out, infos

# -*- coding: utf-8 -*-
"""ctypes binding of libyolov4_amd.so (include/yolov4_amd.h).

The library is the product: there is no Python / PyTorch fallback for any entry
point.  `lib()` raises if the shared object is missing; every call goes through
`check()` which raises `Y4Error` on a non-zero status.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('Y4_LIB_PATH') or os.path.join(_HERE, 'lib', 'libyolov4_amd.so')   # override: kernel experiments

ACT_IDS = {'linear': 0, 'leaky_relu': 1, 'mish': 2, 'relu': 3}


class Y4Error(RuntimeError):
    pass


P = c_void_p        # every device / host pointer
I, L, F, Z = c_int, c_longlong, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/yolov4_amd.h one to one
PROTOTYPES = {
    'y4_strerror': (c_char_p, [I]),
    'y4_version': (I, []),
    'y4_device_count': (I, []),
    'y4_set_conv_mode': (I, [I]),
    'y4_get_conv_mode': (I, []),
    'y4_set_planes_bf16': (I, [I]),
    'y4_get_planes_bf16': (I, []),
    'y4_conv_planes_fit': (I, [I, I, I, I, I, I, I, I]),
    'y4_conv2d_fwd_workspace': (Z, [I, I, I]),
    'y4_conv2d_fwd_f32': (I, [P, I, P, P, I, I, I, I, I, I, I, I, P, P, I, P, I, P, P, P, Z, P]),
    'y4_amax_f32': (I, [P, I, L, I, P, P]),
    'y4_conv2d_prepared_bytes': (Z, [I, I]),
    'y4_conv2d_prepare_filter_f32': (I, [P, I, I, P, Z, P]),
    'y4_conv2d_fwd_prepared_f32': (I, [P, I, P, P, I, I, I, I, I, I, I, I, P, P, I, P, I, P, P, P]),
    'y4_amax_merge_u32': (I, [P, P, P]),
    'y4_last_conv_kernel': (I, [ctypes.c_char_p, I]),
    'y4_conv2d_bnstats_workspace': (Z, [I, I, I, I, I, I, I]),
    'y4_conv2d_fwd_bnstats_f32': (I, [P, I, P, P, I, I, I, I, I, I, I, I, P, Z, P, P, P, Z, P, Z, P]),
    'y4_planes_split_f32': (I, [P, I, L, I, P, P, P]),
    'y4_planes_split_into_f32': (I, [P, I, L, I, P, P, I, I, P]),
    'y4_conv2d_fwd_planes_f32': (I, [P, P, P, I, I, I, I, I, I, I, I, P, Z, P, P, P, Z, P, Z, I, P, P]),
    'y4_conv2d_dgrad_planes_f32': (I, [P, P, P, I, I, I, I, I, I, I, I, P, Z, P, P, I, P]),
    'y4_conv2d_wgrad_planes_workspace': (Z, [I, I, I, I, I, I, I]),
    'y4_conv2d_wgrad_planes_f32': (I, [P, P, P, I, I, I, I, I, I, I, P, Z, P, P, P]),
    'y4_conv2d_generic_fwd_f32': (I, [P, I, P, P, I, I, I, I, I, I, I, I, P, P, I, P, I, P]),
    'y4_conv2d_generic_dgrad_f32': (I, [P, I, P, P, I, I, I, I, I, I, I, I, P, I, P]),
    'y4_conv2d_generic_wgrad_f32': (I, [P, I, P, I, P, I, I, I, I, I, I, I, P]),
    'y4_conv2d_stem_dgrad_f32': (I, [P, I, P, P, L, L, L, L, I, I, I, I, P]),
    'y4_conv2d_stem_fwd_f32': (I, [P, L, L, L, L, P, P, I, I, I, I, I, P, P, I, P, P]),
    'y4_conv2d_dgrad_workspace': (Z, [I, I, I]),
    'y4_conv2d_dgrad_f32': (I, [P, I, P, P, I, I, I, I, I, I, I, I, P, Z, P, P, I, P]),
    'y4_conv2d_wgrad_workspace': (Z, [I, I, I, I, I, I, I]),
    'y4_conv2d_wgrad_f32': (I, [P, I, P, I, P, I, I, I, I, I, I, I, P, Z, P, P, P]),
    'y4_conv2d_stem_wgrad_workspace': (Z, [I, I, I, I]),
    'y4_conv2d_stem_wgrad_f32': (I, [P, L, L, L, L, P, I, P, I, I, I, I, P, Z, P]),
    'y4_bn_workspace': (Z, [L, I]),
    'y4_bn_finalize_workspace': (Z, [I]),
    'y4_bn_finalize_partials_f32': (I, [P, L, L, I, P, P, P, P, P, F, F, P, Z, P]),
    'y4_bn_stats_f32': (I, [P, I, L, I, P, P, P, P, P, F, F, P, Z, P]),
    'y4_bn_act_fwd_f32': (I, [P, I, P, P, P, P, I, P, I, P, I, L, I, P, I, P, P, P]),
    'y4_bn_planes_bound_f32': (I, [P, P, I, L, P, P, P]),
    'y4_bn_act_bwd_f32': (I, [P, I, P, I, P, P, P, P, I, P, I, P, P, L, I, P, Z, P, P, P, I, P]),
    'y4_bias_grad_workspace': (Z, [L, I]),
    'y4_bias_grad_f32': (I, [P, I, L, I, P, P, Z, P]),
    'y4_bn_fold_f32': (I, [P, P, P, P, F, P, P, I, P]),
    'y4_copy_channels_f32': (I, [P, I, P, I, L, I, P]),
    'y4_add_f32': (I, [P, I, P, I, P, I, L, I, P]),
    'y4_maxpool_s1_fwd_f32': (I, [P, I, P, I, P, I, I, I, I, I, P]),
    'y4_maxpool_s1_bwd_f32': (I, [P, I, P, P, I, I, I, I, I, I, I, P]),
    'y4_upsample2x_fwd_f32': (I, [P, I, P, I, I, I, I, I, P]),
    'y4_upsample2x_bwd_f32': (I, [P, I, P, I, I, I, I, I, P]),
    'y4_yolo_decode_train_f32': (I, [P, I, P, P, I, I, I, I, P, P]),
    'y4_yolo_decode_eval_f32': (I, [P, I, P, L, L, I, I, I, I, P, F, P]),
    'y4_yolo_decode_bwd_f32': (I, [P, I, P, P, P, I, I, I, I, P, P]),
    'y4_yolo_loss_workspace': (Z, [I, I, I, I, I]),
    'y4_yolo_loss_fwd_f32': (I, [P, P, P, I, I, I, I, I, F, F, P, I, P, P, P, P, Z, P]),
    'y4_yolo_loss_bwd_f32': (I, [P, I, P, P, P, I, I, I, I, I, P, Z, P]),
    'y4_yolo_loss_mask_output_f32': (I, [P, P, I, I, I, I, I, P, Z, P]),
    'y4_yolo_loss_dense_targets_f32': (I, [P, P, P, I, I, I, I, I, P, Z, P]),
    'y4_post_count_f32': (I, [P, I, L, I, F, I, P, P]),
    'y4_post_scan_i32': (I, [P, I, L, P, P, P]),
    'y4_post_compact_f32': (I, [P, P, P, I, I, P, P, P, P]),
    'y4_post_nms_workspace': (Z, [L, I]),
    'y4_post_nms_f32': (I, [P, I, L, I, F, F, P, L, P, P, P, Z, P]),
    'y4_nms_workspace': (Z, [L]),
    'y4_nms_f32': (I, [P, P, L, F, I, P, P, P, Z, P]),
    'y4_bboxes_iou_f32': (I, [P, L, P, L, I, P, P]),
    'y4_act_fwd_f32': (I, [P, P, L, I, P]),
    'y4_act_bwd_f32': (I, [P, P, P, L, I, P]),
    'y4_upsample_nearest_fwd_f32': (I, [P, I, P, I, I, I, I, I, I, I, I, P]),
    'y4_upsample_nearest_bwd_f32': (I, [P, I, P, I, I, I, I, I, I, I, I, P]),
    'y4_adam_step_f32': (I, [P, P, P, P, L, F, F, F, F, F, I, F, P]),
    'y4_adam_hyper_f32': (I, [F, F, F, F, I, P]),
    'y4_adam_multi_step_f32': (I, [P, I, P, I, F, F, F, F, P]),
    'y4_sgd_multi_step_f32': (I, [P, I, P, I, F, P]),
    'y4_preprocess_u8_f32': (I, [P, I, I, L, I, P, L, L, L, I, P]),
}

_lib = None


def lib():
    """The loaded library.  Fails loudly when it has not been built
    (python __graft_entry__.py / make -C yolov4_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise Y4Error(f'{LIB_PATH} is missing: build it with `make -C yolov4_amd/csrc` '
                          '(hipcc --offload-arch=gfx950); there is no fallback path')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        if os.environ.get('Y4_CONV_MODE'):
            handle.y4_set_conv_mode(int(os.environ['Y4_CONV_MODE']))
    return _lib


def check(code, what=''):
    if code != 0:
        msg = lib().y4_strerror(code).decode()
        raise Y4Error(f'{what}: {msg} (code {code})')


def float_array(values):
    arr = (c_float * len(values))(*[float(v) for v in values])
    return arr


def int_array(values):
    return (c_int * len(values))(*[int(v) for v in values])

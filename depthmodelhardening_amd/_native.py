"""ctypes binding of libdmh_hip.so (C ABI declared in include/dmh_hip.h).

There is no fallback: if the library is missing, or a tensor is not a contiguous fp32 CUDA
(ROCm) tensor, the call raises RuntimeError -- the reference's own error style
(torchattacks/attacks/phy_obj_atk.py:71, MD2/trainer.py:51-52).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: the HIP runtime this library binds to is torch's)

MAX_SCALES = 4
MAX_FRAMES = 4
VARIANT_MD2, VARIANT_DH = 0, 1
NOISE_NONE, NOISE_TENSOR, NOISE_PHILOX = 0, 1, 2
PASTE_COMPOSITE, PASTE_WARP_ONLY = 0, 1
FIN_LOSS, FIN_LOSS_S, FIN_REPROJ_S, FIN_COUNT_S, FIN_SMOOTH_S, FIN_HINT_S, FIN_HINTCOUNT_S, FIN_SIZE = 0, 1, 5, 9, 13, 20, 24, 28

LIB_PATH = os.environ.get("DMH_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                        "libdmh_hip.so")

_fp = C.c_void_p


class PhotoArgs(C.Structure):
    _fields_ = [("target", _fp), ("source", _fp * MAX_FRAMES), ("T", _fp * MAX_FRAMES), ("K", _fp), ("inv_K", _fp),
                ("disp", _fp * MAX_SCALES), ("Hs", C.c_int * MAX_SCALES), ("Ws", C.c_int * MAX_SCALES),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("num_frames", C.c_int), ("num_scales", C.c_int),
                ("min_depth", C.c_float), ("max_depth", C.c_float), ("variant", C.c_int), ("automask", C.c_int),
                ("no_ssim", C.c_int), ("noise_mode", C.c_int), ("noise", _fp * MAX_SCALES),
                ("seed", C.c_uint64), ("offset", C.c_uint64), ("depth_hint", _fp), ("depth_hint_mask", _fp)]


class SmoothArgs(C.Structure):
    _fields_ = [("disp", _fp * MAX_SCALES), ("color", _fp * MAX_SCALES), ("Hs", C.c_int * MAX_SCALES),
                ("Ws", C.c_int * MAX_SCALES), ("B", C.c_int), ("num_scales", C.c_int)]


class PasteArgs(C.Structure):
    _fields_ = [("scene", _fp), ("scene_bstride", C.c_int64), ("patch", _fp), ("pmask", _fp), ("coeffs", _fp),
                ("N", C.c_int), ("SH", C.c_int), ("SW", C.c_int), ("PH", C.c_int), ("PW", C.c_int),
                ("OH", C.c_int), ("OW", C.c_int), ("l_pad", C.c_int), ("t_pad", C.c_int), ("mode", C.c_int),
                ("flip", _fp), ("scene_index", _fp)]


class RoiGlueArgs(C.Structure):
    _fields_ = [("y", _fp), ("skip", _fp), ("y_org", _fp), ("skip_org", _fp), ("dst_org", _fp),
                ("B", C.c_int), ("C1", C.c_int), ("C2", C.c_int), ("sh", C.c_int), ("sw", C.c_int), ("kh", C.c_int),
                ("kw", C.c_int), ("hc", C.c_int), ("wc", C.c_int), ("H", C.c_int), ("W", C.c_int), ("up", C.c_int),
                ("elu", C.c_int)]


class WinoWtJob(C.Structure):
    _fields_ = [("w", _fp), ("scale", _fp), ("U", _fp), ("K", C.c_int), ("C", C.c_int), ("backward", C.c_int)]


_PtrArr = _fp * MAX_SCALES

_SIGNATURES = {
    "dmh_version": (C.c_char_p, []),
    "dmh_last_error": (C.c_char_p, []),
    "dmh_debug_channel_copy": (C.c_int, [_fp, _fp, C.c_int64, C.c_int, C.c_int, _fp]),
    "dmh_photo_partials_size": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "dmh_photo_stage_size": (C.c_int64, [C.POINTER(PhotoArgs)]),
    "dmh_photo_loss_fwd": (C.c_int, [C.POINTER(PhotoArgs), _fp, _PtrArr, _fp, _fp]),
    "dmh_photo_loss_bwd": (C.c_int, [C.POINTER(PhotoArgs), _fp, _fp, _fp, _fp, _PtrArr, _fp]),
    "dmh_photo_pose_partials_size": (C.c_int64, [C.POINTER(PhotoArgs)]),
    "dmh_photo_loss_bwd_pose": (C.c_int, [C.POINTER(PhotoArgs), _fp, _fp, _fp, _fp, _PtrArr, _fp, _fp]),
    "dmh_unpack_selection": (C.c_int, [_fp, C.c_int64, C.c_int, _fp, _fp]),
    "dmh_upsample_bilinear_adjoint": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "dmh_warp_view_fwd": (C.c_int, [_fp] * 5 + [C.c_int] * 5 + [C.c_float, C.c_float] + [_fp] * 4),
    "dmh_warp_view_bwd": (C.c_int, [_fp] * 5 + [C.c_int] * 5 + [C.c_float, C.c_float] + [_fp] * 4),
    "dmh_smooth_partials_size": (C.c_int64, [C.POINTER(SmoothArgs)]),
    "dmh_smooth_loss_fwd": (C.c_int, [C.POINTER(SmoothArgs), _fp, _fp]),
    "dmh_smooth_loss_bwd": (C.c_int, [C.POINTER(SmoothArgs), _fp, _fp, C.c_float, _PtrArr, C.c_int, _fp]),
    "dmh_loss_finalize": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.POINTER(SmoothArgs), C.c_int, C.c_float,
                                    _fp, _fp, _fp]),
    "dmh_eot_paste_fwd": (C.c_int, [C.POINTER(PasteArgs), _fp, _fp, _fp]),
    "dmh_eot_paste_bwd": (C.c_int, [C.POINTER(PasteArgs), _fp, _fp, _fp]),
    "dmh_pgd_linf_step": (C.c_int, [_fp, _fp, _fp, C.c_float, C.c_float, _fp, C.c_int64, _fp]),
    "dmh_l0_compose_fwd": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_float, C.c_int, _fp, _fp, _fp]),
    "dmh_l0_compose_bwd": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp]),
    "dmh_l0_mask_partials_size": (C.c_int64, [C.c_int]),
    "dmh_l0_mask_cost_fwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp]),
    "dmh_l0_mask_cost_bwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int, _fp]),
    "dmh_sq_mean_partials_size": (C.c_int64, [C.c_int64]),
    "dmh_masked_sq_mean_fwd": (C.c_int, [_fp, _fp, C.c_int64, _fp, _fp, _fp]),
    "dmh_masked_sq_mean_bwd": (C.c_int, [_fp, _fp, C.c_int64, _fp, _fp, _fp]),
    "dmh_gt_depth_mse_fwd": (C.c_int, [_fp, _fp, _fp, C.c_int64, _fp, C.c_int, C.c_int64, C.c_float, C.c_float, _fp, _fp, _fp]),
    "dmh_gt_depth_mse_bwd": (C.c_int, [_fp, _fp, _fp, C.c_int64, _fp, C.c_int, C.c_int64, C.c_float, C.c_float, _fp, _fp, _fp]),
    "dmh_ssim_map": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "dmh_ssim_map_bwd": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "dmh_avg_pyramid": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "dmh_edge_smooth_partials_size": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dmh_edge_smooth": (C.c_int, [_fp, _fp] + [C.c_int] * 4 + [_fp, _fp, _fp]),
    "dmh_edge_smooth_bwd": (C.c_int, [_fp, _fp] + [C.c_int] * 4 + [_fp, _fp, _fp]),
    "dmh_depth_errors_partials_size": (C.c_int64, [C.c_int64]),
    "dmh_masked_depth_errors": (C.c_int, [_fp, _fp, _fp, C.c_int64] + [C.c_float] * 5 + [_fp, _fp, _fp]),
    "dmh_dec_up_cat_pad_fwd": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [_fp, _fp]),
    "dmh_dec_up_cat_pad_bwd": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [_fp, _fp, _fp]),
    "dmh_elu_pad_fwd": (C.c_int, [_fp] + [C.c_int] * 5 + [_fp, _fp]),
    "dmh_elu_pad_bwd": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [_fp, _fp]),
    "dmh_roi_glue_fwd": (C.c_int, [C.POINTER(RoiGlueArgs), _fp, _fp]),
    "dmh_roi_glue_bwd": (C.c_int, [C.POINTER(RoiGlueArgs), _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp]),
    "dmh_roi_crop": (C.c_int, [_fp] * 4 + [C.c_int] * 7 + [_fp, _fp]),
    "dmh_roi_paste": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp] + [C.c_int] * 6 + [_fp, _fp]),
    "dmh_stem_conv_norm_fwd_win": (C.c_int, [_fp, _fp, _fp] + [C.c_int] * 5 + [C.c_float, C.c_float, _fp, _fp]),
    "dmh_stem_bn_relu_pool_bwd_win": (C.c_int, [_fp] * 7 + [C.c_int] * 8 + [_fp, _fp]),
    "dmh_conv7x7s2_bwd_data_win": (C.c_int, [_fp] * 4 + [C.c_int] * 9 + [_fp, _fp]),
    "dmh_roi_cost_partials_size": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dmh_roi_cost_fwd": (C.c_int, [_fp, _fp, _fp] + [C.c_int] * 5 + [_fp] * 4),
    "dmh_roi_cost_bwd": (C.c_int, [_fp, _fp, _fp] + [C.c_int] * 5 + [_fp] * 3),
    "dmh_roi_cost_fwd_scaled": (C.c_int, [_fp, _fp, _fp] + [C.c_int] * 5 + [C.c_float] + [_fp] * 4),
    "dmh_roi_cost_bwd_scaled": (C.c_int, [_fp, _fp, _fp] + [C.c_int] * 5 + [C.c_float] + [_fp] * 3),
    "dmh_bn_act_fwd": (C.c_int, [_fp] * 4 + [C.c_int] * 4 + [_fp, _fp]),
    "dmh_bn_act_bwd": (C.c_int, [_fp] * 3 + [C.c_int] * 4 + [_fp, _fp, _fp]),
    "dmh_bn_stats_partials_size": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dmh_bn_train_stats": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float] + [_fp] * 8),
    "dmh_bn_train_stats_tracked": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float] + [_fp] * 9),
    "dmh_channel_sum_partials_size": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dmh_channel_sum": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_void_p]),
    "dmh_bn_train_bwd_workspace_size": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "dmh_bn_train_bwd": (C.c_int, [_fp] * 6 + [C.c_int, C.c_int, C.c_int] + [_fp] * 5 + [C.c_void_p]),
    "dmh_stem_bn_relu_pool_fwd": (C.c_int, [_fp] * 3 + [C.c_int] * 4 + [_fp] * 4),
    "dmh_stem_bn_relu_pool_bwd": (C.c_int, [_fp] * 5 + [C.c_int] * 4 + [_fp, _fp]),
    "dmh_wino_weight_size": (C.c_int64, [C.c_int, C.c_int]),
    "dmh_wino_weight_transform": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "dmh_wino_conv3x3": (C.c_int, [_fp] * 3 + [C.c_int] * 6 + [_fp, _fp]),
    "dmh_wino_conv3x3_ws": (C.c_int, [_fp] * 3 + [C.c_int] * 6 + [_fp, _fp, C.c_int64, _fp]),
    "dmh_wino_conv3x3_plan": (C.c_int, [C.c_int] * 7 + [C.c_int64]),
    "dmh_wino_weight_transform_scaled": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "dmh_wino_conv3x3_act": (C.c_int, [_fp] * 4 + [C.c_int] * 7 + [_fp, _fp]),
    "dmh_wino_weight_transform_batch": (C.c_int, [C.POINTER(WinoWtJob), C.c_int, _fp]),
    "dmh_wino_conv3x3_act_ws": (C.c_int, [_fp] * 4 + [C.c_int] * 7 + [_fp, _fp, C.c_int64, _fp]),
    "dmh_wino32_weight_size": (C.c_int64, [C.c_int, C.c_int]),
    "dmh_wino32_weight_transform": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "dmh_wino32_conv3x3": (C.c_int, [_fp] * 3 + [C.c_int] * 6 + [_fp, _fp]),
    "dmh_wino32_conv3x3_ws": (C.c_int, [_fp] * 3 + [C.c_int] * 6 + [_fp, _fp, C.c_int64, _fp]),
    "dmh_wino_wrw_workspace_size": (C.c_int64, [C.c_int] * 6),
    "dmh_wino_wrw": (C.c_int, [_fp, _fp] + [C.c_int] * 6 + [_fp, _fp, _fp]),
    "dmh_conv3x3_small": (C.c_int, [_fp] * 3 + [C.c_int] * 7 + [_fp, _fp]),
    "dmh_conv3x3_head": (C.c_int, [_fp] * 3 + [C.c_int] * 5 + [_fp, _fp]),
    "dmh_conv3x3_head_bwd_data": (C.c_int, [_fp, _fp] + [C.c_int] * 4 + [_fp, _fp]),
    "dmh_down_wrw_workspace_size": (C.c_int64, [C.c_int] * 5),
    "dmh_down_wrw": (C.c_int, [_fp] * 3 + [C.c_int] * 5 + [_fp] * 4),
    "dmh_stem_wrw_workspace_size": (C.c_int64, [C.c_int] * 3),
    "dmh_stem_wrw": (C.c_int, [_fp, _fp] + [C.c_int] * 3 + [C.c_float, C.c_float, _fp, _fp, _fp]),
    "dmh_conv3x3_head_wrw_partials_size": (C.c_int64, [C.c_int] * 5),
    "dmh_conv3x3_head_wrw": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [_fp] * 4),
    "dmh_conv3x3_small_wrw_partials_size": (C.c_int64, [C.c_int]),
    "dmh_conv3x3_small_wrw": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [_fp] * 4),
    "dmh_conv7x7s2_bwd_data": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [_fp, _fp]),
    "dmh_stem_conv_norm_fwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp, _fp]),
    "dmh_down_conv_fwd": (C.c_int, [_fp] * 3 + [C.c_int] * 5 + [_fp] * 3),
    "dmh_down_conv_fwd_act": (C.c_int, [_fp] * 5 + [C.c_int] * 6 + [_fp] * 3),
    "dmh_down_conv_bwd_data": (C.c_int, [_fp] * 4 + [C.c_int] * 5 + [_fp, _fp]),
    "dmh_down_conv_bwd_data_acc": (C.c_int, [_fp] * 5 + [C.c_int] * 5 + [_fp, _fp]),
    "dmh_down_conv_image_size": (C.c_int64, [C.c_int, C.c_int]),
    "dmh_down_conv_weight_image": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp, _fp]),
    "dmh_down_conv_fwd_img": (C.c_int, [_fp, _fp, C.c_int, _fp, _fp] + [C.c_int] * 6 + [_fp] * 3),
    "dmh_down_conv_bwd_data_img": (C.c_int, [_fp] * 4 + [C.c_int] * 5 + [_fp, _fp]),
}

EXPORTS = tuple(sorted(_SIGNATURES))

_lib = None


def lib():
    """The loaded library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and "DMH_HIP_LIB" not in os.environ:
            try:    # fresh checkout on a box with hipcc: build in-tree once (seconds); never fall back to eager ops
                from .build import build
                build(verbose=False)
            except Exception as e:
                raise RuntimeError("libdmh_hip.so is not built (%s) and building it failed (%s): run `python -m "
                                   "depthmodelhardening_amd.build`; there is no CPU/eager fallback for the hot path"
                                   % (LIB_PATH, e))
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libdmh_hip.so is not built (%s): run `python -m depthmodelhardening_amd.build`; "
                               "there is no CPU/eager fallback for the hot path" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libdmh_hip: " + lib().dmh_last_error().decode())


def ptr(t):
    """Device pointer of a contiguous fp32 (or int32 / uint8 index) CUDA tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libdmh_hip ops need CUDA (ROCm) tensors; got device %s -- there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("libdmh_hip ops need contiguous tensors")
    if t.dtype not in (torch.float32, torch.int32, torch.uint8):
        raise RuntimeError("libdmh_hip ops compute in fp32; got %s" % t.dtype)
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr_array(tensors, n=MAX_SCALES):
    arr = (_fp * n)()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else ptr(t)
    return arr

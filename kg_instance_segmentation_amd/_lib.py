"""ctypes binding of libkgnet_hip.so (the C ABI declared in include/kgnet_hip.h).

The product path has NO fallback: if the HIP library is missing, or a call fails, this module
raises.  Nothing here imports oracle/ or any CPU implementation of the hot path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KG_LIB_PATH: another build of the SAME library (kernel A/B tuning on one GPU box); it must export the whole C ABI like the default
LIB_PATH = os.environ.get("KG_LIB_PATH") or os.path.join(_HERE, "libkgnet_hip.so")
# the same C ABI built for IEEE-half rows (include/kgnet_hip.h, kg_rows_format() == 1): entry points with rows / packed-weight operands only
LIB_F16_PATH = os.environ.get("KG_LIB_F16_PATH") or os.path.join(os.path.dirname(LIB_PATH), "libkgnet_hip_f16.so")

c_int, c_long, c_float, c_double, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double, ctypes.c_void_p
P = c_void_p

# name -> argtypes (restype is int status unless listed in _RESTYPE)
_SIGS = {
    "kg_version": [],
    "kg_rows_format": [],
    "kg_grad_scale": [P, P, P, c_int, c_int, P, P, P],
    "kg_scale_tensors": [P, c_int, c_int, P, P],
    "kg_rows_rescale": [P, c_int, c_long, c_int, c_int, P, P, P, P, P, P],
    "kg_rows_scale": [P, c_int, c_long, c_int, P, P, P, P],
    "kg_rows_scale_multi": [P, c_int, P, P, P],
    "kg_device_arch": [ctypes.c_char_p, c_int],
    "kg_tr_probe": [P, P],
    "kg_conv2d_igemm": [P, P, P, P, P, P, P, P] + [c_int] * 21 + [P, P],
    "kg_conv2d_halo": [P, P, P, P, P, P, P] + [c_int] * 15 + [P, c_int, c_int, P, P],
    "kg_conv1x1": [P, P, P, P, P, P, c_long] + [c_int] * 8 + [P],
    "kg_conv3x3_c64": [P, P, P, P, P, P] + [c_int] * 11 + [P, c_int, P],
    "kg_conv3x3_ws": [P, P, P, P] + [c_int] * 7 + [P, c_int, P, P],
    "kg_pack_weight": [P, P] + [c_int] * 11 + [P],
    "kg_pack_weight_rows": [P, P] + [c_int] * 6 + [P, c_int, c_int, c_int, c_int, P],
    "kg_pack_weight_batch": [P, c_int, c_int, P],
    "kg_pack_weight_narrow": [P, P] + [c_int] * 9 + [P],
    "kg_conv7_narrow": [P, P, P, P] + [c_int] * 11 + [P],
    "kg_im2col_small": [P, P] + [c_int] * 12 + [P],
    "kg_conv2d_halo_heads2": [P, P, P, P, P, P, P] + [c_int] * 7 + [P, P],
    "kg_set_wgrad_tr": [c_int],
    "kg_conv2d_wgrad": [P, P, P, P] + [c_int] * 18 + [c_long, P, P],
    "kg_conv2d_wgrad_halo": [P, P, P] + [c_int] * 11 + [c_long, P, c_int, P, P, P],
    "kg_bias_grad_final": [P, P, c_int, c_int, c_int, P],
    "kg_wgrad_reduce": [P, P, c_int, c_int, c_int, c_int, c_int, c_long, c_int, P],
    "kg_wgrad_reduce_multi": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_long, c_int, P],
    "kg_wgrad_reduce_bias": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_long, c_int, P, P, c_int, P],
    "kg_wgrad_reduce_defer": [c_int],
    "kg_wgrad_reduce_pending": [],
    "kg_wgrad_reduce_flush": [P],
    "kg_bias_grad": [P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "kg_img_pack": [P, P, c_int, c_int, c_int, c_int, c_int, P, P],
    "kg_bn_stats_train": [P, c_int, c_int, c_int, P, P, P, P, c_float, c_float, P, P, P, P, P, c_int, P, P],
    "kg_bn_scale_shift_eval": [c_int, P, P, P, P, c_float, P, P, P],
    "kg_conv_stats_begin": [P, c_long],
    "kg_conv_bstats_begin": [P, c_long, P, c_int, c_int, c_int, P, P],
    "kg_conv_stats_end": [P],
    "kg_bn_finalize_train": [P, c_int, c_int, c_int, P, P, P, P, c_float, c_float, P, P, P, P, P],
    "kg_bn_apply": [P, c_int, P, P, P, c_int, P, c_int, c_int, c_int, c_int, P, P],
    "kg_bn_bwd": [P, c_int, P, c_int, P, P, P, P, P, c_int, P, c_int, c_int, c_int, P, c_int, P, c_int, P, P, P],
    "kg_maxpool3s2_fwd": [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P],
    "kg_maxpool3s2_bwd": [P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P],
    "kg_bilinear_fwd": [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_long, P, P],
    "kg_bilinear_bwd": [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_long, P, c_int, P, P],
    "kg_add_rows": [P, c_int, P, c_int, P, c_int, P, c_int, c_long, c_int, P, P, P, P],
    "kg_detection_loss_fwd": [P, P, P, P, c_int, c_int, c_int, c_float, P, P, c_int, P, P],
    "kg_detection_loss_bwd": [P, P, P, P, c_int, c_int, c_int, c_float, P, P, P, P, P, P],
    "kg_seg_loss": [P, P, P, P, c_int, P, P, P, P, P],
    "kg_sigmoid_inplace": [P, c_long, P],
    "kg_grad_pack": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "kg_grad_pack3": [P, P, P, P, P] + [c_int] * 10 + [P, P],
    "kg_postproc_workspace_bytes": [c_int, c_int, c_int, c_int],
    "kg_postproc_scale": [P, P, P, c_int, c_int, c_double, P, c_long, c_int, c_int, P, P, P, P, P, P, P, P],
    "kg_postproc_timing_begin": [],
    "kg_postproc_timing_end": [P],
    "kg_skeleton_boxes": [P, P, c_int, c_double, c_int, P, P, c_int, P],
    "kg_nms": [P, P, c_int, c_double, P, c_long, P, P, P],
    "kg_gt_maps": [P, c_int, c_int, c_int, P, P],
    "kg_adam_step": [P, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, P],
    "kg_host_crop_masks": [P, P, c_int, c_int, c_int, P],
    "kg_crop_masks": [P, P, c_int, c_int, c_int, P, P],
    "kg_host_match_boxes": [P, c_int, P, c_int, c_int, c_float, P, c_int, P],
    "kg_host_tile_table": [P, P, P, c_int, c_int, c_int, P, c_int],                       # (these two return a COUNT, -1 on overflow:
    "kg_host_bin_csr": [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int],               #  call them through load(), not call())
    "kg_mask_areas": [P, c_int, c_long, P, P],
    "kg_mask_inter_pairs": [P, P, P, c_int, c_long, P, P],
    "kg_f64_probe": [P, P, P, c_int, P],
    "kg_seg_build_rows": [P, c_int, P, P, P, P],
    "kg_seg_build_rows_levels": [c_int, P, P, P, P, P, P],
    "kg_rows_gather": [P, c_int, P, P, c_int, c_long, c_int, P],
    "kg_rows_gather_planes": [P, c_int, c_int, P, P, c_int, c_int, c_long, c_int, c_int, P],
    "kg_seg_conv3_c1": [P, c_int, c_int, P, P, P, c_long, P, P, P],
    "kg_f32_to_bf16_rows": [P, P, c_int, c_long, c_int, P, c_int, P],
    "kg_rows_gather_f32": [P, c_int, P, P, c_int, c_long, c_int, P, P],
    "kg_planes_to_f32": [P, c_int, P, c_int, c_long, c_int, P, P],
    "kg_f32_to_planes": [P, c_int, P, c_int, P, c_int, c_long, c_int, P, P],
    "kg_mask_paste": [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P, c_int, P],
    "kg_crop_grad_reduce": [P, c_int, P, c_int, c_long, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P],
}
_RESTYPE = {"kg_postproc_workspace_bytes": c_long}
SYMBOLS = tuple(_SIGS) + ("kg_last_error", "kg_last_kernel")

_lib = None
_lib_f16 = None


class KGLibraryError(RuntimeError):
    pass


def _open(path, must_have_all):
    if not os.path.exists(path):
        raise KGLibraryError(
            f"{path} is missing: build it with `python -m kg_instance_segmentation_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the KGnet hot path.")
    lib = ctypes.CDLL(path)
    lib.kg_last_error.restype = ctypes.c_char_p
    lib.kg_last_error.argtypes = []
    lib.kg_last_kernel.restype = ctypes.c_char_p
    lib.kg_last_kernel.argtypes = []
    for name, args in _SIGS.items():
        if must_have_all:
            fn = getattr(lib, name)  # AttributeError => ABI drift, surface it
        else:
            fn = getattr(lib, name, None)
            if fn is None:
                continue
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, c_int)
    return lib


def load(fmt=0):
    """Loads the HIP library (fmt 1: its IEEE-half build); raises KGLibraryError (never falls back) when it is absent."""
    global _lib, _lib_f16
    if fmt:
        if _lib_f16 is None:
            lib = _open(LIB_F16_PATH, False)
            if lib.kg_rows_format() != 1:
                raise KGLibraryError(f"{LIB_F16_PATH} is not the half-precision build (kg_rows_format() != 1)")
            _lib_f16 = lib
        return _lib_f16
    if _lib is None:
        _lib = _open(LIB_PATH, True)
    return _lib


def call(name, *args, fmt=0):
    """Calls a status-returning entry point (of the library built for rows format `fmt`) and raises on a non-zero status."""
    lib = load(fmt)
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise KGLibraryError(f"{name} failed ({rc}): {lib.kg_last_error().decode()}")


def last_kernel(fmt=0):
    """rocprofv3's name of the conv-family kernel this thread's most recent conv call launched (measurement aid, kg_last_kernel)"""
    return load(fmt).kg_last_kernel().decode()


_RAW_STREAM = None


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (honours `with torch.cuda.stream(...)`).
    ~1000 launches per training step go through here: the raw-stream query is 5x cheaper than building a torch.cuda.Stream object."""
    global _RAW_STREAM
    import torch
    if _RAW_STREAM is None:
        _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _RAW_STREAM:
        return c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())

"""Drop-in `preprocessing` module (reference preprocessing.py:4-118): ground-truth map generation on the GPU.

`get_ground_truth` keeps the reference's signature and return layout (float64 NumPy arrays) so dataset_base.py:94-97 runs
unchanged; `get_ground_truth_device` / `ground_truth_scales` return the float32 [55,H,W] device tensors that the training
loop consumes (dataset_base.py:99-109), without the host round trip.  All values are bit-identical to the reference's
(kg_gt_maps, csrc/preproc.hip).  Keypoints are taken as float32, as dataset_base.masks_to_bboxes produces them."""
import numpy as np
import torch

from . import _lib, config as cfg, ops
from ._lib import ptr, stream_ptr


def create_position_index(height, width):
    """preprocessing.py:4-11: H x W x 2 array of (x, y) pixel positions."""
    return np.rollaxis(np.indices(dimensions=(width, height)), 0, 3).transpose((1, 0, 2))


def _device(device):
    try:
        from torch.utils.data import get_worker_info
        in_worker = get_worker_info() is not None
    except Exception:
        in_worker = False
    if in_worker:
        raise _lib.KGLibraryError("preprocessing.get_ground_truth (MI355X build) was called inside a DataLoader worker process; "
                                  "workers cannot use the GPU: build the DataLoader with num_workers=0 (dropin/run.py does so under "
                                  "KG_GPU_GT=1) or keep the reference's host preprocessing module for the workers")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise _lib.KGLibraryError("preprocessing (MI355X build) needs a GPU device")
    return dev


def get_ground_truth_device(bboxes, height, width, device=None, out=None):
    """bboxes: [n,5,2] (x,y) keypoints (NumPy or tensor).  Returns the float32 [55,height,width] device tensor
    (kp 5 | short 10 | mid 40) that dataset_base.py:99-109 builds."""
    dev = _device(device)
    if torch.is_tensor(bboxes):
        kps = bboxes.detach().to(device=dev, dtype=torch.float32).reshape(-1, 5, 2).contiguous()
    else:
        arr = np.asarray(bboxes, np.float32).reshape(-1, 5, 2)
        kps = ops.h2d(arr, dev) if arr.size else torch.zeros(0, 5, 2, dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty(55, int(height), int(width), dtype=torch.float32, device=dev)
    assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (55, int(height), int(width))
    with torch.cuda.device(dev):
        _lib.call("kg_gt_maps", ptr(kps) if kps.numel() else None, kps.shape[0], int(height), int(width), ptr(out), stream_ptr())
    return out


def ground_truth_scales(bboxes_per_scale, sizes, device=None):
    """The four gt_c0..gt_c3 tensors of dataset_base.py:94-112: bboxes_per_scale[l] = [n_l,5,2], sizes[l] = (h_l, w_l)."""
    return [get_ground_truth_device(b, h, w, device) for b, (h, w) in zip(bboxes_per_scale, sizes)]


def get_ground_truth(bboxes, height, width, num_kps):
    """preprocessing.py:107-118: (kp_heats [num_kps,H,W], short_offsets [H,W,2*num_kps], mid_offsets [H,W,4*NUM_EDGES]),
    float64 NumPy arrays (every value is exactly representable in the float32 the kernel writes)."""
    if num_kps != cfg.NUM_KPS:
        raise ValueError(f"num_kps must be {cfg.NUM_KPS}")
    g = get_ground_truth_device(bboxes, height, width).cpu().numpy().astype(np.float64)
    return g[0:5], np.ascontiguousarray(np.transpose(g[5:15], (1, 2, 0))), np.ascontiguousarray(np.transpose(g[15:55], (1, 2, 0)))

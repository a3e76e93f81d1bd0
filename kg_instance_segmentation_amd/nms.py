"""Drop-in `nms` module (reference nms.py:4-53) on the HIP NMS kernel (float64, bit-identical)."""
import numpy as np
import torch

from . import _lib
from ._lib import ptr, stream_ptr, c_double, c_long


def nms_device(boxes_d, nbox_d, cap, thresh):
    """boxes_d [cap,5] f64 device, nbox_d int32[1] device -> (kept [cap,5] f64 device, nkeep int32[1] device)."""
    dev = boxes_d.device
    ws = torch.empty(cap * 9 + 1024, dtype=torch.uint8, device=dev)
    out = torch.empty(cap, 5, dtype=torch.float64, device=dev)
    nk = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("kg_nms", ptr(boxes_d), ptr(nbox_d), cap, c_double(thresh), ptr(ws), c_long(ws.numel()), ptr(out), ptr(nk), stream_ptr())
    return out, nk


def non_maximum_suppression_numpy(bboxes, nms_thresh=0.5):
    """bboxes: num_insts x 5 [y1,x1,y2,x2,conf] (ndarray, float64).  None when empty (nms.py:8-9).
    Equal confidences are ordered by index (the reference's np.argsort default sort is unstable)."""
    if len(bboxes) == 0:
        return None
    _lib.load()
    if not torch.cuda.is_available():
        raise _lib.KGLibraryError("nms (MI355X build) needs a GPU")
    b = np.ascontiguousarray(bboxes, np.float64).reshape(-1, 5)
    n = len(b)
    bd = torch.from_numpy(b).cuda()
    nb = torch.tensor([n], dtype=torch.int32, device=bd.device)
    out, nk = nms_device(bd, nb, n, float(nms_thresh))
    k = int(nk.item())
    return out[:k].cpu().numpy()

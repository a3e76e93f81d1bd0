"""numpy-facing wrapper of oracle/kg_oracle.c (TEST INFRASTRUCTURE, not product).

Restates postprocessing.py:16-261 and nms.py:4-53 of the reference; every
function is bit-checked against tests/golden/postproc_*.npz.
"""
import ctypes
import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        _lib.kgo_peaks.restype = ctypes.c_int
        _lib.kgo_group.restype = ctypes.c_int
        _lib.kgo_refine.restype = ctypes.c_int
        _lib.kgo_boxes.restype = ctypes.c_int
        _lib.kgo_nms.restype = ctypes.c_int
        _lib.kgo_gauss_weights.restype = ctypes.POINTER(ctypes.c_double)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _chw(t, dtype=np.float32):
    """Accepts torch tensor or ndarray, [1,C,H,W] or [C,H,W]; batch index 0 only
    (postprocessing.py:138-140)."""
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    a = np.asarray(t)
    if a.ndim == 4:
        a = a[0]
    return np.ascontiguousarray(a, dtype=dtype)


def gauss_weights():
    return np.ctypeslib.as_array(lib().kgo_gauss_weights(), shape=(17,)).copy()


def hough(kp, short):
    kp, short = _chw(kp), _chw(short)
    C, H, W = kp.shape
    assert C == 5 and short.shape == (10, H, W)
    heat = np.empty((5, H, W), np.float64)
    lib().kgo_hough(_p(kp), _p(short), ctypes.c_int(H), ctypes.c_int(W), _p(heat))
    return heat


def gauss(heat):
    heat = np.ascontiguousarray(heat, np.float64)
    C, H, W = heat.shape
    out = np.empty_like(heat)
    lib().kgo_gauss(_p(heat), ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W), _p(out))
    return out


def peaks(heat, thresh=0.004):
    heat = np.ascontiguousarray(heat, np.float64)
    C, H, W = heat.shape
    cap = C * H * W
    ids = np.empty(cap, np.int32); xs = np.empty(cap, np.int32); ys = np.empty(cap, np.int32)
    conf = np.empty(cap, np.float64)
    n = lib().kgo_peaks(_p(heat), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_double(thresh),
                        ctypes.c_int(cap), _p(ids), _p(xs), _p(ys), _p(conf))
    return ids[:n].copy(), xs[:n].copy(), ys[:n].copy(), conf[:n].copy()


def group(ids, xs, ys, conf, mid):
    mid = _chw(mid)
    _, H, W = mid.shape
    n = len(ids)
    skel = np.zeros((max(n, 1), 5, 3), np.float64)
    ns = lib().kgo_group(ctypes.c_int(n), _p(np.ascontiguousarray(ids, np.int32)),
                         _p(np.ascontiguousarray(xs, np.int32)), _p(np.ascontiguousarray(ys, np.int32)),
                         _p(np.ascontiguousarray(conf, np.float64)), _p(mid), ctypes.c_int(H),
                         ctypes.c_int(W), ctypes.c_int(max(n, 1)), _p(skel))
    return skel[:ns].copy()


def get_skeletons(kp, short, mid):
    """== postprocessing.get_skeletons_and_masks (postprocessing.py:129-147) as [S,5,3] f64."""
    heat = gauss(hough(kp, short))
    return group(*peaks(heat, 0.004), mid)


def refine(skel):
    skel = np.ascontiguousarray(skel, np.float64).reshape(-1, 5, 3)
    keep = np.zeros(max(len(skel), 1), np.int32)
    lib().kgo_refine(ctypes.c_int(len(skel)), _p(skel), _p(keep))
    return skel[keep[:len(skel)].astype(bool)]


def boxes(skel, scale):
    skel = np.ascontiguousarray(skel, np.float64).reshape(-1, 5, 3)
    out = np.empty((max(len(skel), 1), 5), np.float64)
    nb = lib().kgo_boxes(ctypes.c_int(len(skel)), _p(skel), ctypes.c_double(scale), _p(out))
    return out[:nb].copy()


def gather(s0, s1, s2, s3):
    """== postprocessing.gather_skeleton (postprocessing.py:255-261)."""
    return np.concatenate([boxes(s0, 1), boxes(s1, 2), boxes(s2, 4), boxes(s3, 8)], 0)


def nms(bboxes, thresh=0.5):
    """== nms.non_maximum_suppression_numpy (nms.py:4-53); None when empty."""
    bboxes = np.ascontiguousarray(bboxes, np.float64).reshape(-1, 5)
    if len(bboxes) == 0:
        return None
    keep = np.empty(len(bboxes), np.int32)
    nk = lib().kgo_nms(ctypes.c_int(len(bboxes)), _p(bboxes), ctypes.c_double(thresh), _p(keep))
    return bboxes[keep[:nk]]


def detect(dec, nms_thresh=0.5):
    """test.py:105-116: four scales -> skeletons -> refine -> gather -> NMS."""
    sk = [refine(get_skeletons(*d)) for d in dec]
    return nms(gather(*sk), nms_thresh)

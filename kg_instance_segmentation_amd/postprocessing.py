"""Drop-in `postprocessing` module (reference postprocessing.py) on the float64 HIP pipeline.

Hough vote -> Gaussian -> peaks -> greedy keypoint-graph grouping run on the GPU with the
reference's exact float64 arithmetic and summation order; only the (tiny) skeleton list is copied to
the host, where the reference's list-of-ndarray return types are rebuilt."""
import numpy as np
import torch

from . import _lib, ops
from ._lib import ptr, stream_ptr, c_double, c_long

PEAK_THRESH = 0.004   # postprocessing.py:145
_STREAMS = {}


class _Workspace:
    cache = {}

    @classmethod
    def get(cls, H, W, dev, peak_cap, skel_cap):
        key = (H, W, str(dev), peak_cap, skel_cap)
        ws = cls.cache.get(key)
        if ws is None:
            nbytes = _lib.load().kg_postproc_workspace_bytes(H, W, peak_cap, skel_cap)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            cls.cache[key] = ws
        return ws


def caps(H, W):
    peak_cap = min(5 * H * W, 1 << 16)
    return peak_cap, peak_cap


def skeletons_device(kp, short, mid, debug=False):
    """kp [.,5,H,W], short [.,10,H,W], mid [.,40,H,W] device fp32 (batch element 0 is used,
    postprocessing.py:138-140).  Returns (skel [cap,5,3] f64 device, nskel int32[1] device[, debug dict])."""
    if not kp.is_cuda:
        raise _lib.KGLibraryError("postprocessing (MI355X build) needs GPU tensors")
    kp0 = kp.detach()[0].contiguous().float(); sh0 = short.detach()[0].contiguous().float(); md0 = mid.detach()[0].contiguous().float()
    _, H, W = kp0.shape
    dev = kp0.device
    peak_cap, skel_cap = caps(H, W)
    ws = _Workspace.get(H, W, dev, peak_cap, skel_cap)
    skel = torch.empty(skel_cap, 5, 3, dtype=torch.float64, device=dev)
    nsk = torch.zeros(1, dtype=torch.int32, device=dev)
    dbg = None
    args = [None] * 5
    if debug:
        dbg = {"heat": torch.empty(5, H, W, dtype=torch.float64, device=dev), "blur": torch.empty(5, H, W, dtype=torch.float64, device=dev),
               "peaks": torch.empty(3, peak_cap, dtype=torch.int32, device=dev), "conf": torch.empty(peak_cap, dtype=torch.float64, device=dev),
               "npeaks": torch.zeros(1, dtype=torch.int32, device=dev)}
        args = [ptr(dbg["heat"]), ptr(dbg["blur"]), ptr(dbg["peaks"]), ptr(dbg["conf"]), ptr(dbg["npeaks"])]
    with torch.cuda.device(dev):      # (launch on the tensors' device, whatever the "current" one is)
        _lib.call("kg_postproc_scale", ptr(kp0), ptr(sh0), ptr(md0), H, W, c_double(PEAK_THRESH), ptr(ws), c_long(ws.numel()), peak_cap,
                  skel_cap, ptr(skel), ptr(nsk), *args, stream_ptr())
    return (skel, nsk, dbg) if debug else (skel, nsk)


def get_skeletons_and_masks(kp_maps, short_offsets, mid_offsets):
    """== postprocessing.get_skeletons_and_masks (postprocessing.py:129-147): list of 5x3 float64 arrays."""
    skel, nsk = skeletons_device(kp_maps, short_offsets, mid_offsets)
    n = int(nsk.item())
    if n > skel.shape[0]:
        raise _lib.KGLibraryError(f"skeleton capacity exceeded ({n} > {skel.shape[0]})")
    host = skel[:n].cpu().numpy()
    return [host[i].copy() for i in range(n)]


def refine_skeleton(skeletons):
    """postprocessing.py:150-159 (host: a few hundred 5x3 arrays)."""
    out = []
    for s in skeletons:
        mask = s[:, 0] > 0.
        if mask.sum() >= 3 or mask[[0, 3]].sum() == 2 or mask[[1, 2]].sum() == 2:
            out.append(s)
    return out


def _boxes_device(skel_list, scale, boxes, nbox, cap, dev):
    n = len(skel_list)
    if n == 0:
        return
    sk = torch.from_numpy(np.ascontiguousarray(np.asarray(skel_list, np.float64).reshape(n, 15))).to(dev)
    ns = torch.tensor([n], dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("kg_skeleton_boxes", ptr(sk), ptr(ns), n, c_double(scale), 0, ptr(boxes), ptr(nbox), cap, stream_ptr())


def skeleton_to_box(skeletons, scale):
    """postprocessing.py:164-242 -> list of [y1,x1,y2,x2,conf]; scales the skeletons in place like the reference."""
    if len(skeletons) == 0:
        return []
    dev = torch.device("cuda", torch.cuda.current_device())
    cap = len(skeletons)
    boxes = torch.empty(cap, 5, dtype=torch.float64, device=dev)
    nbox = torch.zeros(1, dtype=torch.int32, device=dev)
    _boxes_device(skeletons, scale, boxes, nbox, cap, dev)
    k = int(nbox.item())
    for s in skeletons:
        s[:, :2] *= scale
    return [list(r) for r in boxes[:k].cpu().numpy()]


def gather_skeleton(skeleton0, skeleton1, skeleton2, skeleton3):
    """postprocessing.py:255-261: boxes of the four scales concatenated (N x 5 float64, or shape (0,))."""
    b = skeleton_to_box(skeleton0, 1) + skeleton_to_box(skeleton1, 2) + skeleton_to_box(skeleton2, 4) + skeleton_to_box(skeleton3, 8)
    return np.asarray(b)


def gather_skeleton_single(skeleton0, skeleton1, skeleton2, skeleton3):
    return (np.asarray(skeleton_to_box(skeleton0, 1)), np.asarray(skeleton_to_box(skeleton1, 2)),
            np.asarray(skeleton_to_box(skeleton2, 4)), np.asarray(skeleton_to_box(skeleton3, 8)))


def detect(dec, nms_thresh=0.5, timing=None):
    """Fused test.py:105-116: four scales -> refine -> boxes -> NMS with a single device->host copy.
    dec = ([kp,short,mid] x 4).  Returns N x 5 float64 ndarray or None.  timing (dict, measurement hook of bench.py): receives
    "boxes_nms_ms", the GPU time of the box assembly + NMS launches."""
    from . import nms as _nms
    dev = dec[0][0].device
    # the scales are independent and the greedy grouping of a scale is ONE workgroup: run them on separate streams -- the caller's
    # stream for the first (largest) scale + 3 side streams = 4, the number of hardware queues a process gets by default (a fifth
    # stream shares a queue with another one and its scale runs after that one's grouping instead of beside it)
    main = torch.cuda.current_stream(dev)
    pool = _STREAMS.setdefault(str(dev), [torch.cuda.Stream(dev) for _ in range(3)])
    ready = main.record_event()            # the head maps are complete here; the side streams start from this point
    sks, by_key, used = [], {}, []
    for i, d in enumerate(dec):
        key = tuple(d[0].shape[-2:])
        if key not in by_key:              # equal sizes share one cached workspace: they stay on one stream, in order
            by_key[key] = None if not by_key else pool[(len(by_key) - 1) % 3]
        st = by_key[key]
        if st is None:
            sks.append(skeletons_device(*d))
            continue
        if st not in used:
            st.wait_event(ready)
            used.append(st)
        with torch.cuda.stream(st):
            r = skeletons_device(*d)
        for t in r:
            t.record_stream(main)
        sks.append(r)
    for st in used:
        main.wait_stream(st)
    cap = sum(s[0].shape[0] for s in sks)
    cap = min(cap, 1 << 15)
    boxes = torch.empty(cap, 5, dtype=torch.float64, device=dev)
    nbox = torch.zeros(1, dtype=torch.int32, device=dev)
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    for (skel, nsk), scale in zip(sks, (1, 2, 4, 8)):
        with torch.cuda.device(dev):
            _lib.call("kg_skeleton_boxes", ptr(skel), ptr(nsk), skel.shape[0], c_double(scale), 1, ptr(boxes), ptr(nbox), cap, stream_ptr())
    out, nk = _nms.nms_device(boxes, nbox, cap, float(nms_thresh))
    if timing is not None:
        e1.record()
    k = int(nk.item())
    if timing is not None:
        timing["boxes_nms_ms"] = e0.elapsed_time(e1)
    if k == 0:
        return None
    return out[:k].cpu().numpy()


def paste_masks(predictions, input_h, input_w, image_w, image_h, seg_thresh, device_u8=False):
    """== the reference driver's `post_processing` (test.py:127-157) for the predictions of `model.forward_seg`: every mask patch is
    resized to its (rounded, clamped) box, pasted into an (input_h, input_w) canvas, resized to the image and thresholded -- in one
    kernel launch for all detections (kg_mask_paste) instead of two cv2.resize calls and a full-size host array per detection.
    Returns [masks float32 [n, image_h, image_w] in {0, 1}, dets float32 [n, 5] = (y1, x1, y2, x2, conf) in image pixels] like the
    reference, or None; device_u8=True keeps the masks on the GPU as bytes (what eval_parts.seg_evaluation consumes)."""
    if predictions is None:
        return None
    meta = getattr(predictions, "kg_meta", None)
    if meta is None:
        raise _lib.KGLibraryError("paste_masks needs the predictions object returned by this package's forward_seg")
    n = len(meta["off"])
    flat = meta["flat"]
    dev = flat.device
    b = np.asarray(meta["boxes"], np.float32).reshape(-1, 5)
    y1 = np.maximum(0, np.round(b[:, 0]).astype(np.int32)); x1 = np.maximum(0, np.round(b[:, 1]).astype(np.int32))
    y2 = np.minimum(np.round(b[:, 2]).astype(np.int32), input_h - 1); x2 = np.minimum(np.round(b[:, 3]).astype(np.int32), input_w - 1)
    tab = np.stack([np.asarray(meta["off"], np.int64), np.asarray(meta["h"], np.int64), np.asarray(meta["w"], np.int64), y1, x1, y2, x2,
                    np.zeros(n, np.int64)], 1).astype(np.int32)
    dets = np.stack([y1.astype(np.float64) / input_h * image_h, x1.astype(np.float64) / input_w * image_w,
                     y2.astype(np.float64) / input_h * image_h, x2.astype(np.float64) / input_w * image_w, b[:, 4].astype(np.float64)], 1).astype(np.float32)
    out = torch.empty(n, int(image_h), int(image_w), dtype=torch.uint8 if device_u8 else torch.float32, device=dev)
    if n:
        with torch.cuda.device(dev):
            _lib.call("kg_mask_paste", ptr(flat), ptr(ops.h2d(tab.reshape(-1), dev)), n, int(input_h), int(input_w), int(image_h), int(image_w),
                      _lib.c_float(float(seg_thresh)), ptr(out), 1 if device_u8 else 0, stream_ptr())
    return [out if device_u8 else out.cpu().numpy(), dets]

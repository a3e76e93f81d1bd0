"""Drop-in `eval_parts` module (reference eval_parts.py:4-150): evaluation metrics with the mask IoUs counted on the GPU.

mask_iou / voc_ap / bbox_evaluation / seg_evaluation keep the reference's signatures (eval.py:152-232 calls them with the
dataset object).  The O(dets x GT x H x W) part -- eval_parts.mask_iou for every detection against every GT instance whose
box overlaps -- is two kernel launches per image (kg_mask_areas, kg_mask_inter_pairs: exact integer counts); the greedy
matching, the box tests and the AP integration are host logic on a few hundred numbers, as in the reference.
Masks are 0/1 arrays (eval.py:95-127), any non-zero value counts as foreground."""
import numpy as np
import torch

from . import _lib, ops
from ._lib import ptr, stream_ptr, c_long


def _device_masks(m, dev):
    """[n,H,W] NumPy / tensor -> uint8 device tensor [n, ld] with ld = H*W rounded up to 16 (zero padded)."""
    if torch.is_tensor(m):
        t = (m.to(dev) != 0).to(torch.uint8).reshape(m.shape[0], -1)
    else:
        a = np.ascontiguousarray((np.asarray(m) != 0).astype(np.uint8)).reshape(len(m), -1)
        t = ops.h2d(a, dev) if a.size else torch.zeros(a.shape, dtype=torch.uint8, device=dev)
    hw = t.shape[1]
    ld = ops.round_up(max(hw, 1), 16)
    if ld != hw:
        t = torch.nn.functional.pad(t, (0, ld - hw))
    return t.contiguous(), ld


def _areas(t, ld):
    out = torch.empty(t.shape[0], dtype=torch.int32, device=t.device)
    if t.shape[0]:
        _lib.call("kg_mask_areas", ptr(t), t.shape[0], c_long(ld), ptr(out), stream_ptr())
    return out


def mask_iou_table(det_masks, gt_masks, pairs, device=None):
    """IoU (eval_parts.mask_iou semantics: 0 if the union is empty) of the (detection, GT) index pairs [P,2].
    Returns a float64 array [P]."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise _lib.KGLibraryError("eval_parts (MI355X build) needs a GPU device")
    pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
    if len(pairs) == 0:
        return np.zeros(0, np.float64)
    with torch.cuda.device(dev):
        a, ld = _device_masks(det_masks, dev)
        b, ld2 = _device_masks(gt_masks, dev)
        assert ld == ld2, "detection and ground-truth masks must have the same size"
        pd = ops.h2d(pairs, dev)
        inter = torch.empty(len(pairs), dtype=torch.int32, device=dev)
        _lib.call("kg_mask_inter_pairs", ptr(a), ptr(b), ptr(pd), len(pairs), c_long(ld), ptr(inter), stream_ptr())
        ia = inter.cpu().numpy().astype(np.int64)
        aa = _areas(a, ld).cpu().numpy().astype(np.int64)
        ab = _areas(b, ld).cpu().numpy().astype(np.int64)
    union = aa[pairs[:, 0]] + ab[pairs[:, 1]] - ia
    out = np.zeros(len(pairs), np.float64)
    ok = union >= 1
    out[ok] = ia[ok].astype(np.float64) / union[ok].astype(np.float64)
    return out


def mask_iou(mask1, mask2):
    """eval_parts.py:4-9."""
    v = mask_iou_table(np.asarray(mask1)[None], np.asarray(mask2)[None], [[0, 0]])[0]
    return float(v) if v > 0 else 0


def voc_ap(rec, prec, use_07_metric=True):
    """eval_parts.py:12-43: 11-point VOC07 metric, or the area under the precision envelope."""
    rec = np.asarray(rec); prec = np.asarray(prec)
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            sel = rec >= t
            ap = ap + (np.max(prec[sel]) if np.sum(sel) != 0 else 0) / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def _box_inter(gt, b):
    iymin = np.maximum(gt[:, 0], b[0]); ixmin = np.maximum(gt[:, 1], b[1])
    iymax = np.minimum(gt[:, 2], b[2]); ixmax = np.minimum(gt[:, 3], b[3])
    return np.maximum(ixmax - ixmin, 0.) * np.maximum(iymax - iymin, 0.)


def bbox_evaluation(index, dsets, BB_bboxes, all_scores, npos, ov_thresh):
    """eval_parts.py:45-93 (box IoU matching; a few hundred boxes, host)."""
    order = np.argsort(-BB_bboxes[:, 4])
    boxes = BB_bboxes[order, :4]
    all_scores.extend(BB_bboxes[order, 4])
    nd = boxes.shape[0]
    tp = np.zeros(nd); fp = np.zeros(nd)
    gt = dsets.load_annotation(index=index, type='bbox')
    npos = npos + gt.shape[0]
    gtf = gt.astype(float)
    taken = [False] * gt.shape[0]
    for d in range(nd):
        bb = boxes[d, :].astype(float)
        ovmax, jmax = -np.inf, -1
        if gtf.shape[0] > 0:
            inters = _box_inter(gtf, bb)
            union = (bb[2] - bb[0]) * (bb[3] - bb[1]) + (gtf[:, 2] - gtf[:, 0]) * (gtf[:, 3] - gtf[:, 1]) - inters
            ov = inters / union
            ovmax = np.max(ov); jmax = int(np.argmax(ov))
        if ovmax >= ov_thresh and not taken[jmax]:
            tp[d] = 1.; taken[jmax] = True
        else:
            fp[d] = 1.
    return fp, tp, all_scores, npos


def seg_evaluation(index, dsets, BB_masks, BB_dets, all_scores, npos, temp_overlaps, ov_thresh):
    """eval_parts.py:98-150: detections sorted by confidence, each matched to the overlapping GT instance of largest mask IoU."""
    order = np.argsort(-BB_dets[:, 4])
    masks = BB_masks[order]; boxes = BB_dets[order, :4]
    all_scores.extend(BB_dets[order, 4])
    nd = masks.shape[0]
    tp = np.zeros(nd); fp = np.zeros(nd)
    gt_masks = dsets.load_annotation(index, type='mask')
    gt_boxes = dsets.load_annotation(index, type='bbox')
    ng = gt_masks.shape[0]
    npos = npos + ng
    keep = np.zeros((nd, ng), bool)
    for d in range(nd):
        keep[d] = _box_inter(gt_boxes, boxes[d]) > 0. if ng else False
    pairs = np.argwhere(keep)
    iou = np.zeros((nd, ng), np.float64)
    if len(pairs):
        iou[pairs[:, 0], pairs[:, 1]] = mask_iou_table(masks, gt_masks, pairs)
    taken = [False] * ng
    for d in range(nd):
        ovmax, jmax = -np.inf, -1
        for j in np.nonzero(keep[d])[0]:
            if iou[d, j] > ovmax:
                ovmax, jmax = float(iou[d, j]), int(j)
        if ovmax >= ov_thresh and not taken[jmax]:
            tp[d] = 1.; taken[jmax] = True; temp_overlaps.append(ovmax)
        else:
            fp[d] = 1.
    return fp, tp, all_scores, npos, temp_overlaps

"""Drop-in `seg_loss` module (reference seg_loss.py:8-96).

Matching predicted patches to GT boxes (IoU >= 0.5) and cropping the GT masks is host glue, exactly
as in the reference; the per-pixel BCE over all matched pairs runs in one HIP kernel launch."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ptr, stream_ptr


def nearest_resize(a, h1, w1):
    """cv2.resize(a, (w1, h1), INTER_NEAREST) (seg_loss.py:77): src = min(floor(dst * src/dst_size), src-1)."""
    h0, w0 = a.shape
    if (h0, w0) == (h1, w1):
        return a
    yi = np.minimum(np.floor(np.arange(h1) * (h0 / h1)).astype(np.int64), h0 - 1)
    xi = np.minimum(np.floor(np.arange(w1) * (w0 / w1)).astype(np.int64), w0 - 1)
    return a[yi][:, xi]


def jaccard_numpy(a, b):
    """seg_loss.py:14-29 in float32."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    area_a = (a[2] - a[0]) * (a[3] - a[1]); area_b = (b[2] - b[0]) * (b[3] - b[1])
    ih = max(min(a[2], b[2]) - max(a[0], b[0]), np.float32(0.))
    iw = max(min(a[3], b[3]) - max(a[1], b[1]), np.float32(0.))
    inter = ih * iw
    union = area_a + area_b - inter
    return 0. if union <= 2 else float(np.divide(inter, union))


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, tgt, patches, pairs, npatches):
        dev = flat.device
        part = torch.empty(npatches, dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("kg_seg_loss", ptr(flat), ptr(tgt), ptr(patches), ptr(pairs), npatches, ptr(part), ptr(out), None, None, stream_ptr())
        ctx.save_for_backward(flat, tgt, patches, pairs)
        ctx.npatches = npatches
        return out[0].clone()

    @staticmethod
    def backward(ctx, go):
        flat, tgt, patches, pairs = ctx.saved_tensors
        g = torch.zeros_like(flat)
        go = go.contiguous().float()
        with torch.cuda.device(flat.device):
            _lib.call("kg_seg_loss", ptr(flat), ptr(tgt), ptr(patches), ptr(pairs), ctx.npatches, None, None, ptr(go), ptr(g), stream_ptr())
        return g, None, None, None, None


def jaccard_matrix(a, b):
    """Vectorised seg_loss.py:14-29 in float32 (same operation order as jaccard_numpy): a [P,4], b [G,4] -> [P,G]."""
    a = np.asarray(a, np.float32).reshape(-1, 1, 4); b = np.asarray(b, np.float32).reshape(1, -1, 4)
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]); area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    ih = np.maximum(np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]), np.float32(0.))
    iw = np.maximum(np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]), np.float32(0.))
    inter = ih * iw
    union = area_a + area_b - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(union <= 2, np.float32(0.), inter / union)


def match_boxes(pb, gb, thresh=0.5):
    """(patch, gt) index pairs with jaccard_numpy(patch box, gt box) >= thresh (seg_loss.py:55-56) through the native host matcher
    (kg_host_match_boxes: the same float32 operation order as jaccard_matrix, without the [P, G] temporaries)."""
    import ctypes
    pb = np.ascontiguousarray(pb, np.float32).reshape(-1, 4); gb = np.ascontiguousarray(gb, np.float32)
    gb = gb.reshape(len(gb), -1)
    P, G = len(pb), len(gb)
    cap = max(64, 4 * max(P, G))
    while True:
        pairs = np.empty((cap, 2), np.int32)
        cnt = ctypes.c_int(0)
        try:
            _lib.call("kg_host_match_boxes", ctypes.c_void_p(pb.ctypes.data), P, ctypes.c_void_p(gb.ctypes.data), G, gb.shape[1], _lib.c_float(thresh),
                      ctypes.c_void_p(pairs.ctypes.data), cap, ctypes.byref(cnt))
            break
        except _lib.KGLibraryError:
            if cap >= P * G:
                raise
            cap = P * G                      # dense overlaps: every pair may match
    n = cnt.value
    return pairs[:n, 0].astype(np.int64), pairs[:n, 1].astype(np.int64)


class SEG_loss(nn.Module):
    def __init__(self, height, width):
        super().__init__()
        self.height, self.width = height, width

    def jaccard_numpy(self, a, b):
        return jaccard_numpy(a, b)

    def forward(self, predictions, gt_masks, gt_boxes):
        mask_patches, mask_dets = predictions
        nimg = len(mask_patches)
        meta = getattr(predictions, "kg_meta", None)
        # per image: predicted boxes [P,4] f32, patch sizes, and where each patch lives
        per_img = []
        if meta is not None:
            for i in range(nimg):
                sel = np.nonzero(meta["img"] == i)[0]
                per_img.append((meta["boxes"][sel, :4], meta["h"][sel], meta["w"][sel], meta["off"][sel], None))
        else:
            for i in range(nimg):
                pl = mask_patches[i]
                if len(pl) == 0:
                    per_img.append((np.zeros((0, 4), np.float32), np.zeros(0, np.int64), np.zeros(0, np.int64), None, pl))
                    continue
                pb = np.stack([np.asarray(d[:4].detach().cpu().numpy() if hasattr(d, "detach") else d[:4], np.float32) for d in mask_dets[i]])
                per_img.append((pb, np.array([p.shape[0] for p in pl]), np.array([p.shape[1] for p in pl]), None, pl))
        # pass 1 (vectorised per image): matches, pair weights, crop rectangles; pass 2: one native host call crops the matched
        # GT masks straight into ONE pinned uint8 staging buffer (kg_host_crop_masks)
        rec_img, rec_j, rec_npx, rec_p0, rec_cnt = [], [], [], [], []
        pair_off, pair_w, work = [], [], []
        toff, npairs = 0, 0
        for i, (pb, hs, ws, offs, pl) in enumerate(per_img):
            gb = np.asarray(gt_boxes[i], np.float32).reshape(-1, 5) if len(gt_boxes[i]) else np.zeros((0, 5), np.float32)
            if len(pb) == 0 or len(gb) == 0:
                continue
            js, gs = match_boxes(pb, gb)                                        # seg_loss.py:55-56, row-major: patch asc, gt asc
            nobj = len(js)
            if nobj == 0:
                continue
            y1 = np.maximum(0, np.round(pb[:, 0]).astype(np.int32)); x1 = np.maximum(0, np.round(pb[:, 1]).astype(np.int32))
            y2 = np.minimum(np.round(pb[:, 2]).astype(np.int32), self.height - 1)
            x2 = np.minimum(np.round(pb[:, 3]).astype(np.int32), self.width - 1)
            hj = np.asarray(hs, np.int64)[js]; wj = np.asarray(ws, np.int64)[js]
            npx = hj * wj
            off_k = toff + np.cumsum(npx) - npx
            toff += int(npx.sum())
            pair_off.append(off_k); pair_w.append((1.0 / nobj / nimg) / npx)
            work.append(np.stack([np.full(nobj, i), gs, y1[js], y2[js], x1[js], x2[js], hj, wj, off_k], 1))
            ujs, first = np.unique(js, return_index=True)
            cnt = np.diff(np.append(first, nobj))
            rec_img.append(np.full(len(ujs), i)); rec_j.append(ujs); rec_npx.append(npx[first]); rec_p0.append(npairs + first); rec_cnt.append(cnt)
            npairs += nobj
        if not rec_img:
            return None                      # seg_loss.py:93-96
        rec_img, rec_j, rec_npx, rec_p0, rec_cnt = (np.concatenate(v) for v in (rec_img, rec_j, rec_npx, rec_p0, rec_cnt))
        work = np.ascontiguousarray(np.concatenate(work), np.int32)
        tgt = None
        if all(torch.is_tensor(m) and m.is_cuda for m in gt_masks):
            # device-resident ground-truth masks: crop + nearest-resize on the GPU, nothing crosses PCIe (kg_crop_masks);
            # the reference's collater hands host arrays (collater.py), which take the host path below
            ms = [m.contiguous().float() for m in gt_masks]
            sized = [m for m in ms if m.numel()]
            H0, W0 = sized[0].shape[1:]
            if any(m.dim() != 3 or tuple(m.shape[1:]) != (H0, W0) for m in sized):
                raise ValueError("SEG_loss: device masks must be [n, H, W] tensors of one size")
            if (np.minimum(work[:, 3], H0) <= work[:, 2]).any() or (np.minimum(work[:, 5], W0) <= work[:, 4]).any():
                raise _lib.KGLibraryError("SEG_loss: empty ground-truth crop")
            dev = sized[0].device
            ptrs = ops.h2d(np.array([m.data_ptr() for m in ms], np.int64), dev)
            tgt = torch.empty(toff, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.call("kg_crop_masks", _lib.ptr(ptrs), _lib.ptr(ops.h2d(work.reshape(-1), dev)), len(work), int(H0), int(W0), _lib.ptr(tgt), _lib.stream_ptr())
            self._keep = ms       # (the crops are read asynchronously: keep the converted masks alive until the next call)
        tgt_host = torch.empty(0 if tgt is not None else toff, dtype=torch.uint8, pin_memory=True)
        marr = [] if tgt is not None else [np.asarray(m) for m in gt_masks]
        if tgt is not None:
            pass
        elif all(m.dtype == np.float32 and m.flags.c_contiguous and m.ndim == 3 and m.shape[1:] == marr[0].shape[1:] for m in marr if m.size):
            import ctypes
            H0, W0 = next(m.shape[1:] for m in marr if m.size)
            ptrs = (ctypes.c_void_p * len(marr))(*[m.ctypes.data for m in marr])
            _lib.call("kg_host_crop_masks", ctypes.cast(ptrs, ctypes.c_void_p), ctypes.c_void_p(work.ctypes.data), len(work), int(H0), int(W0),
                      ctypes.c_void_p(tgt_host.data_ptr()))
        else:                                # other mask containers: per-pair NumPy crops
            tnp = tgt_host.numpy()
            for (i, g, ya, yb, xa, xb_, h1, w1, off) in work.tolist():
                crop = marr[i][g][ya:yb, xa:xb_]                               # seg_loss.py:64
                if crop.shape != (h1, w1):
                    crop = nearest_resize(np.asarray(crop), h1, w1)             # seg_loss.py:77
                    assert crop.shape == (h1, w1), "[loss.py] mask size does not match!"
                tnp[off:off + h1 * w1].reshape(h1, w1)[...] = crop
        if meta is not None:
            flat = meta["flat"]
            offs = np.concatenate([np.asarray(per_img[i][3])[rec_j[rec_img == i]] for i in np.unique(rec_img).tolist()])
        else:                                # patches from elsewhere: concatenate (autograd-tracked)
            flat = torch.cat([per_img[i][4][j].reshape(-1) for i, j in zip(rec_img.tolist(), rec_j.tolist())]).float()
            offs = np.cumsum(rec_npx) - rec_npx
        if not flat.is_cuda:
            raise _lib.KGLibraryError("SEG_loss (MI355X build) needs GPU tensors")
        dev = flat.device
        ptab = np.ascontiguousarray(np.stack([offs, rec_npx, rec_p0, rec_cnt], 1), np.int32)
        pair_t = np.zeros(npairs, dtype=[("off", np.int32), ("w", np.float32)])
        pair_t["off"] = np.concatenate(pair_off); pair_t["w"] = np.concatenate(pair_w)
        nrec = len(rec_img)
        if tgt is None:
            tgt = tgt_host.to(dev, non_blocking=True)
        ptab_d = ops.h2d(ptab, dev)
        pairs_d = ops.h2d(pair_t.view(np.uint8), dev)
        return _SegLossFn.apply(flat, tgt, ptab_d, pairs_d, nrec)

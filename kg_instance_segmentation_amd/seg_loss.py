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
        _lib.call("kg_seg_loss", ptr(flat), ptr(tgt), ptr(patches), ptr(pairs), npatches, ptr(part), ptr(out), None, None, stream_ptr())
        ctx.save_for_backward(flat, tgt, patches, pairs)
        ctx.npatches = npatches
        return out[0].clone()

    @staticmethod
    def backward(ctx, go):
        flat, tgt, patches, pairs = ctx.saved_tensors
        g = torch.zeros_like(flat)
        go = go.contiguous().float()
        _lib.call("kg_seg_loss", ptr(flat), ptr(tgt), ptr(patches), ptr(pairs), ctx.npatches, None, None, ptr(go), ptr(g), stream_ptr())
        return g, None, None, None, None


class SEG_loss(nn.Module):
    def __init__(self, height, width):
        super().__init__()
        self.height, self.width = height, width

    def jaccard_numpy(self, a, b):
        return jaccard_numpy(a, b)

    def forward(self, predictions, gt_masks, gt_boxes):
        mask_patches, mask_dets = predictions
        nimg = len(mask_patches)
        plist, pairs, tgts = [], [], []   # per matched patch: (patch tensor, first pair, npairs)
        toff = 0
        per_img = []
        for i in range(nimg):
            entries = []
            for j in range(len(mask_patches[i])):
                pr = mask_patches[i][j]
                pbox = np.asarray(mask_dets[i][j][:4].detach().cpu().numpy() if hasattr(mask_dets[i][j], "detach") else mask_dets[i][j][:4], np.float32)
                h1, w1 = pr.shape
                mine = []
                for g in range(gt_boxes[i].shape[0]):
                    if jaccard_numpy(pbox, gt_boxes[i][g][:4]) >= 0.5:
                        y1 = max(0, int(np.int32(np.round(pbox[0])))); x1 = max(0, int(np.int32(np.round(pbox[1]))))
                        y2 = min(int(np.int32(np.round(pbox[2]))), self.height - 1)
                        x2 = min(int(np.int32(np.round(pbox[3]))), self.width - 1)
                        gm = nearest_resize(np.asarray(gt_masks[i][g])[y1:y2, x1:x2], h1, w1)
                        assert gm.shape == (h1, w1), "[loss.py] mask size does not match!"
                        mine.append(np.ascontiguousarray(gm).astype(np.uint8).ravel())
                if mine:
                    entries.append((pr, mine))
            per_img.append(entries)
        if not any(per_img):
            return None                      # seg_loss.py:93-96
        recs = []
        for i, entries in enumerate(per_img):
            nobj = sum(len(m) for _, m in entries)
            for pr, mine in entries:
                npix = pr.numel()
                recs.append((pr, npix, len(pairs), len(mine)))
                for t in mine:
                    pairs.append((toff, 1.0 / npix / nobj / nimg))
                    tgts.append(t); toff += npix
        dev = recs[0][0].device
        if not recs[0][0].is_cuda:
            raise _lib.KGLibraryError("SEG_loss (MI355X build) needs GPU tensors")
        # all patches of one forward_seg call are views of one flat probability buffer; find it
        base = recs[0][0]._base if recs[0][0]._base is not None else None
        same = base is not None and base.dim() == 1 and all(r[0]._base is base for r in recs)
        if same:
            flat = base
            offs = [r[0].storage_offset() - base.storage_offset() for r in recs]
        else:                                # patches from elsewhere: concatenate (autograd-tracked)
            flat = torch.cat([r[0].reshape(-1) for r in recs])
            offs = list(np.cumsum([0] + [r[1] for r in recs[:-1]]))
        ptab = np.array([[o, r[1], r[2], r[3]] for o, r in zip(offs, recs)], np.int32)
        pair_t = np.zeros(len(pairs), dtype=[("off", np.int32), ("w", np.float32)])
        pair_t["off"] = [p[0] for p in pairs]; pair_t["w"] = [p[1] for p in pairs]
        tgt = torch.from_numpy(np.concatenate(tgts)).to(dev)
        ptab_d = torch.from_numpy(ptab).to(dev)
        pairs_d = torch.from_numpy(pair_t.view(np.uint8)).to(dev)
        return _SegLossFn.apply(flat.contiguous().float() if not same else flat, tgt, ptab_d, pairs_d, len(recs))

"""GPU: the drop-in eval_parts module (mask IoUs counted by kg_mask_inter_pairs / kg_mask_areas) gives exactly the reference's
evaluation outputs on the fixtures, and the IoU table is exact at full image size."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import eval_parts as kev  # noqa: E402
from oracle import evalparts as oev  # noqa: E402


class _DS:
    def __init__(self, gm, gb):
        self.gm, self.gb = gm, gb

    def load_annotation(self, index, type):
        return self.gm if type == "mask" else self.gb


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("thr", [0.5, 0.75])
def test_seg_and_bbox_evaluation_vs_golden(golden, name, thr):
    g = golden("evalparts.npz")
    gm, gb, dm, dd = g[f"{name}.gt_masks"].astype(np.float32), g[f"{name}.gt_boxes"], g[f"{name}.det_masks"].astype(np.float32), g[f"{name}.det"]
    k = f"{name}.seg{int(thr * 100)}"
    fp, tp, sc, npos, ovl = kev.seg_evaluation(0, _DS(gm, gb), dm, dd, [], 0, [], thr)
    assert np.array_equal(fp, g[k + ".fp"]) and np.array_equal(tp, g[k + ".tp"]) and npos == len(gm)
    assert np.array_equal(np.asarray(sc, np.float32), g[k + ".scores"]) and np.array_equal(np.asarray(ovl, np.float64), g[k + ".overlaps"])
    k = f"{name}.box{int(thr * 100)}"
    fp, tp, sc, npos = kev.bbox_evaluation(0, _DS(gm, gb), dd, [], 0, thr)
    assert np.array_equal(fp, g[k + ".fp"]) and np.array_equal(tp, g[k + ".tp"])
    assert kev.voc_ap(g["ap.rec"], g["ap.prec"], True) == g["ap.values"][0] and kev.voc_ap(g["ap.rec"], g["ap.prec"], False) == g["ap.values"][1]
    assert kev.mask_iou(dm[0], gm[0]) == g[f"{name}.iou"][0, 0] and kev.mask_iou(dm[-1], gm[0]) == 0


def test_iou_table_full_size_vs_oracle():
    """300 detections x 300 GT instances at 512x512 (odd width to exercise the row padding): exact IoUs of the box-overlapping pairs."""
    H, W, n = 512, 509, 300
    rng = np.random.default_rng(4)
    gm = np.zeros((n, H, W), np.uint8); dm = np.zeros((n, H, W), np.uint8)
    for k in range(n):
        h, w = rng.integers(14, 40, 2); y, x = rng.integers(0, H - 48), rng.integers(0, W - 48)
        gm[k, y:y + h, x:x + w] = 1
        sy, sx = rng.integers(-6, 7, 2)
        dm[k, max(y + sy, 0):y + sy + h, max(x + sx, 0):x + sx + w] = 1
    pairs = np.array([(d, g) for d in range(n) for g in range(n) if (abs(d - g) <= 1 or (d * 7 + g) % 97 == 0)], np.int32)
    kev.mask_iou_table(dm[:2], gm[:2], pairs[:1] * 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = kev.mask_iou_table(dm, gm, pairs)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = np.array([oev.mask_iou(dm[d], gm[g]) for d, g in pairs[:400]], np.float64)
    t_cpu = (time.perf_counter() - t0) * len(pairs) / 400
    print(f"[mask_iou_table] {len(pairs)} pairs at {H}x{W}: GPU {1e3 * t_gpu:.1f} ms incl. upload, NumPy ~{1e3 * t_cpu:.0f} ms")
    assert np.array_equal(got[:400], ref) and (got > 0).sum() > 200

"""CPU: oracle/evalparts.py reproduces the reference's evaluation metrics (eval_parts.py) on the fixtures generated from the
reference with a stub dataset (tools/gen_goldens.py)."""
import numpy as np
import pytest

from oracle import evalparts as oev


@pytest.mark.parametrize("name", ["a", "b"])
@pytest.mark.parametrize("thr", [0.5, 0.75])
def test_seg_and_bbox_evaluation(golden, name, thr):
    g = golden("evalparts.npz")
    gm, gb, dm, dd = g[f"{name}.gt_masks"].astype(np.float32), g[f"{name}.gt_boxes"], g[f"{name}.det_masks"].astype(np.float32), g[f"{name}.det"]
    k = f"{name}.seg{int(thr * 100)}"
    fp, tp, sc, ovl = oev.seg_evaluation(gm, gb, dm, dd, thr)
    assert np.array_equal(fp, g[k + ".fp"]) and np.array_equal(tp, g[k + ".tp"]) and np.array_equal(sc, g[k + ".scores"])
    assert np.array_equal(np.asarray(ovl, np.float64), g[k + ".overlaps"])
    assert tp.sum() > 0 and fp.sum() > 0                      # the fixture exercises both outcomes (duplicates, false positives)
    k = f"{name}.box{int(thr * 100)}"
    fp, tp, _ = oev.bbox_evaluation(gb, dd, thr)
    assert np.array_equal(fp, g[k + ".fp"]) and np.array_equal(tp, g[k + ".tp"])


def test_mask_iou_matrix_and_ap(golden):
    g = golden("evalparts.npz")
    for name in ("a", "b"):
        iou = np.array([[oev.mask_iou(a, b) for b in g[f"{name}.gt_masks"]] for a in g[f"{name}.det_masks"]], np.float64)
        assert np.array_equal(iou, g[f"{name}.iou"])
        assert (g[f"{name}.iou"][-1] == 0).all()              # the empty detection mask: union < 1 -> 0
    ap = [oev.voc_ap(g["ap.rec"], g["ap.prec"], True), oev.voc_ap(g["ap.rec"], g["ap.prec"], False)]
    assert np.array_equal(np.asarray(ap, np.float64), g["ap.values"])

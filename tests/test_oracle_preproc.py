"""CPU: oracle/preproc.py reproduces the reference's ground-truth map generation (preprocessing.get_ground_truth +
the assembly of dataset_base.py:99-109) bit-for-bit on the fixtures generated from the reference (tools/gen_goldens.py)."""
import numpy as np
import pytest

from oracle import preproc

CASES = ["r48x64", "dense96", "odd33x70", "adv", "empty"]


@pytest.mark.parametrize("name", CASES)
def test_ground_truth_bit_exact(golden, name):
    g = golden("preproc.npz")
    H, W = [int(v) for v in g[f"{name}.hw"]]
    got = preproc.ground_truth(g[f"{name}.bboxes"], H, W)
    ref = g[f"{name}.gt"]
    assert got.shape == ref.shape == (55, H, W)
    assert np.array_equal(got.astype(np.float32), ref) and np.array_equal(got, ref.astype(np.float64))


def test_fixture_covers_the_edge_cases(golden):
    """The adversarial fixture really contains the cases it is there for: an argmin tie (identical instances: the first one
    owns the disc), a window overwriting another instance's values with its zero corners, border keypoints."""
    g = golden("preproc.npz")
    bb = g["adv.bboxes"]
    assert np.array_equal(bb[0], bb[1])                       # identical instances
    gt = g["adv.gt"]
    assert gt[0].sum() > 0 and gt[5:15].any() and gt[15:].any()
    assert (bb[..., 0] == 0).any() and (bb[..., 1] == 0).any()  # keypoints on the border
    # short offsets are zero in window corners even inside a disc of kp heat 1 (the reference never applies the disc mask)
    assert ((gt[0] == 1) & (gt[5] == 0) & (gt[6] == 0)).sum() > 5

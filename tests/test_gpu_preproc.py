"""GPU: kg_gt_maps (ground-truth map generation) is bit-identical to the reference fixtures and to the
oracle at training sizes; the drop-in `preprocessing.get_ground_truth` returns the reference's layouts."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import preprocessing as kprep  # noqa: E402
from oracle import preproc, synth  # noqa: E402


@pytest.mark.parametrize("name", ["r48x64", "dense96", "odd33x70", "adv", "empty"])
def test_gt_maps_vs_golden(golden, name):
    g = golden("preproc.npz")
    H, W = [int(v) for v in g[f"{name}.hw"]]
    got = kprep.get_ground_truth_device(g[f"{name}.bboxes"], H, W).cpu().numpy()
    ref = g[f"{name}.gt"]
    bad = int((got != ref).sum())
    print(f"[gt_maps {name}] {bad} mismatches of {ref.size}")
    assert got.dtype == np.float32 and bad == 0
    kp, sh, md = kprep.get_ground_truth(g[f"{name}.bboxes"], H, W, 5)          # drop-in layouts (preprocessing.py:107-118)
    assert kp.shape == (5, H, W) and sh.shape == (H, W, 10) and md.shape == (H, W, 40) and kp.dtype == np.float64
    assert np.array_equal(np.concatenate((kp, np.transpose(sh, (2, 0, 1)), np.transpose(md, (2, 0, 1))), 0), ref.astype(np.float64))


@pytest.mark.parametrize("S,n,sc", [(256, 120, 1), (512, 300, 2)])
def test_gt_maps_vs_oracle_training_sizes(S, n, sc):
    """300 instances of a 512^2 image (BASELINE configs[1]) at scale 1/sc vs the NumPy oracle, incl. half-integer centres."""
    bx = np.floor(synth.random_boxes(S, S, n, 3) / sc)
    kps = synth.keypoints_of(bx).astype(np.float32)
    H = S // sc
    t0 = time.perf_counter(); ref = preproc.ground_truth(kps, H, H); t_cpu = time.perf_counter() - t0
    kprep.get_ground_truth_device(kps, H, H)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = kprep.get_ground_truth_device(kps, H, H)
    torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
    bad = int((got.cpu().numpy().astype(np.float64) != ref).sum())
    print(f"[gt_maps {H}x{H} n={n}] oracle {1e3 * t_cpu:.0f} ms, GPU {1e3 * t_gpu:.2f} ms, {bad} mismatches")
    assert bad == 0


def test_refuses_cpu_device():
    from kg_instance_segmentation_amd import _lib
    with pytest.raises(_lib.KGLibraryError):
        kprep.get_ground_truth_device(np.zeros((1, 5, 2), np.float32), 8, 8, device="cpu")

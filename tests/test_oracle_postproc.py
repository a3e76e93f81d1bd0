"""CPU: the oracle (oracle/kg_oracle.c) reproduces the reference's post-processing
bit-for-bit on the fixtures generated from the reference (tools/gen_goldens.py)."""
import hashlib

import numpy as np
import pytest

from oracle import postproc as op
from oracle import synth


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def _inputs(g, name):
    if name == "adv":
        return g["adv.kp"], g["adv.short"], g["adv.mid"]
    H, W, n, seed = g[f"{name}.cfg"]
    kp, short, mid, _ = synth.head_maps(int(H), int(W), int(n), int(seed))
    assert np.array_equal(sha(np.concatenate([kp.ravel(), short.ravel(), mid.ravel()])), g[f"{name}.in_sha"]), \
        "synthesizer drifted from the one that produced the fixture"
    return kp, short, mid


@pytest.mark.parametrize("name", ["s64", "s96x128", "s256", "adv"])
def test_stages_bit_exact(golden, name):
    g = golden("postproc.npz")
    kp, short, mid = _inputs(g, name)
    heat = op.hough(kp, short)
    blur = op.gauss(heat)
    if f"{name}.heat" in g:
        assert np.array_equal(heat, g[f"{name}.heat"])
        assert np.array_equal(blur, g[f"{name}.blur"])
    if f"{name}.heat_sha" in g:
        assert np.array_equal(sha(heat), g[f"{name}.heat_sha"])
        assert np.array_equal(sha(blur), g[f"{name}.blur_sha"])
    ids, xs, ys, conf = op.peaks(blur, 0.004)
    assert np.array_equal(np.stack([ids, xs, ys], 1).reshape(-1, 3), g[f"{name}.peaks"])
    assert np.array_equal(conf, g[f"{name}.peak_conf"])
    skel = op.group(ids, xs, ys, conf, mid)
    assert np.array_equal(skel, g[f"{name}.skel"])
    assert np.array_equal(op.refine(skel), g[f"{name}.refined"])
    assert np.array_equal(op.get_skeletons(kp, short, mid), g[f"{name}.skel"])


def test_boxes_gather_nms(golden):
    g = golden("postproc.npz")
    sks = [g["s256.refined"], g["s96x128.refined"], g["s64.refined"], g["adv.refined"]]
    for s, sc in zip(sks, (1, 2, 4, 8)):
        assert np.array_equal(op.boxes(s, sc), g[f"boxes.scale{sc}"])
    gat = op.gather(*sks)
    assert np.array_equal(gat, g["gather"])
    for th in (0.5, 0.3):
        assert np.array_equal(op.nms(gat, th), g[f"nms.{th}"])
    assert op.nms(np.zeros((0, 5))) is None


def test_hand_made_geometry(golden):
    g = golden("postproc.npz")
    hand = g["hand.skel"]
    assert np.array_equal(op.refine(hand), g["hand.refined"])
    assert np.array_equal(op.boxes(hand, 2), g["hand.boxes_all"])
    assert np.array_equal(op.boxes(g["hand.refined"], 4), g["hand.boxes_refined"])
    assert np.array_equal(op.nms(g["hand.boxes_all"], 0.5), g["hand.nms"])


def test_gauss_weights_match_scipy():
    from scipy.ndimage import _filters
    assert np.array_equal(op.gauss_weights(), _filters._gaussian_kernel1d(2.0, 0, 8))


def test_empty_and_tiny():
    z = np.zeros((1, 5, 8, 8), np.float32)
    sk = op.get_skeletons(z, np.zeros((1, 10, 8, 8), np.float32), np.zeros((1, 40, 8, 8), np.float32))
    assert sk.shape == (0, 5, 3)
    assert op.gather(sk, sk, sk, sk).shape == (0, 5)
    assert op.nms(op.gather(sk, sk, sk, sk)) is None

"""CPU: oracle/paste.py (restatement of OpenCV's published generic INTER_LINEAR float path + test.py:127-157) against
hand-computed vectors.  cv2 is absent from the build container: the interpolation rule is pinned to the published algorithm only
(see the oracle's docstring)."""
import numpy as np

from oracle import paste


def test_upscale_2x_hand_vector():
    # src [10, 20] -> 4 columns: f = (d+0.5)*0.5-0.5 = -0.25, 0.25, 0.75, 1.25 -> taps (0,f=0), (0,.25), (0,.75), (1, f=0)
    out = paste.resize_linear_f32(np.array([[10.0, 20.0]], np.float32), 1, 4)
    assert out.dtype == np.float32 and np.array_equal(out, np.array([[10.0, 12.5, 17.5, 20.0]], np.float32))
    # vertical: rows clamped, coefficients kept: dy=0: s=-1,f=.75 -> rows (0,0); dy=3: s=1,f=.25 -> rows (1,1)
    out = paste.resize_linear_f32(np.array([[10.0], [20.0]], np.float32), 4, 1)
    assert np.array_equal(out[:, 0], np.array([10.0, 12.5, 17.5, 20.0], np.float32))


def test_downscale_and_identity():
    src = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(paste.resize_linear_f32(src, 3, 4), src)
    # 4 -> 2 columns: scale 2: f = 0.5, 2.5 -> taps (0,.5), (2,.5): averages of neighbours
    out = paste.resize_linear_f32(src[:1], 1, 2)
    assert np.array_equal(out, np.array([[0.5, 2.5]], np.float32))
    # 3 -> 2 rows: scale 1.5: f = 0.25, 1.75 -> (0, .25), (1, .75)
    out = paste.resize_linear_f32(src[:, :1], 2, 1)
    assert np.allclose(out[:, 0], [0 * .75 + 4 * .25, 4 * .25 + 8 * .75]) and out.dtype == np.float32


def test_paste_masks_contract():
    rng = np.random.default_rng(0)
    patches = [[rng.random((10, 12)).astype(np.float32)], [rng.random((7, 7)).astype(np.float32), rng.random((5, 9)).astype(np.float32)]]
    dets = [[np.array([4.4, 5.5, 14.4, 17.5, 0.9], np.float32)], [np.array([0, 0, 7, 7, 0.8], np.float32), np.array([30.5, 40.5, 38.0, 47.0, 0.7], np.float32)]]
    masks, d = paste.paste_masks([patches, dets], 40, 48, 96, 60, 0.5)
    assert masks.shape == (3, 60, 96) and masks.dtype == np.float32 and set(np.unique(masks)) <= {0.0, 1.0}
    assert d.shape == (3, 5) and d.dtype == np.float32
    # first detection: rounds to rows 4..14, cols 6 (5.5 -> 6, half-to-even) .. 18 (17.5 -> 18): patch resized to 10 x 12 = identity
    y1, x1, y2, x2 = 4, 6, 14, 18
    assert np.allclose(d[0, :4], [y1 / 40 * 60, x1 / 48 * 96, y2 / 40 * 60, x2 / 48 * 96])
    assert masks[0][:int(y1 * 1.5) - 1].sum() == 0 and masks[0][:, :x1 * 2 - 1].sum() == 0
    assert paste.paste_masks(None, 40, 48, 96, 60, 0.5) is None
    # third detection needs a genuine resize of its 5 x 9 patch to 8 x 6 (30.5 -> 30, 40.5 -> 40, 38, 47 -> 46 clamp no: 47)
    assert masks[2].sum() > 0

"""CPU restatement of the mask paste-back of the reference's inference driver (TEST INFRASTRUCTURE, not product).

Reference: test.py:127-157 (`post_processing`): per detection, `cv2.resize(patch, (x2-x1, y2-y1))` -> paste into a zero
(input_h, input_w) mask -> `cv2.resize(mask, (image_w, image_h))` -> `mask >= seg_thresh` -> box scaled to the image.

PARITY UNPINNED for the interpolation: OpenCV (cv2 4.1.0.25 in the reference's README) is a third-party dependency that is
absent from the build container and from /root/reference, so `resize_linear_f32` below restates the PUBLISHED generic
INTER_LINEAR algorithm of OpenCV's imgproc/resize.cpp for CV_32F (resizeGeneric_ with HResizeLinear / VResizeLinear<float>):
  * scale = 1 / (dsize / ssize) in double; per destination index d: f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s;
  * horizontal taps: s < 0 -> (s, f) = (0, 0); s >= ssize-1 -> (s, f) = (ssize-1, 0); destination columns from the first one
    whose second tap would fall outside use the first tap alone; coefficients (1 - f, f) as float;
  * vertical taps: rows s and s+1 CLAMPED to [0, ssize-1], coefficients (1 - f, f) unchanged;
  * arithmetic in float32, horizontal pass first: row[d] = S[s]*a0 + S[s+1]*a1, then dst = row0*b0 + row1*b1 (no FMA);
  * equal sizes: plain copy.
(OpenCV builds may route this call through IPP / SIMD paths whose results can differ in the last bit, and switch exact 2x
downscales to the area kernel; neither can be observed here.)  The restatement is checked against hand-computed vectors in
tests/test_oracle_paste.py; the HIP kernel (csrc/paste.hip) is checked against this file bit for bit."""
import numpy as np


def _taps(ssize, dsize, horizontal):
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if horizontal:
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0; s[hi] = ssize - 1
        single = s + 1 >= ssize                  # from xmax on: first tap only
        s1 = np.minimum(s + 1, ssize - 1)
        return s, s1, (np.float32(1) - f).astype(np.float32), f, single
    s0 = np.clip(s, 0, ssize - 1)
    s1 = np.clip(s + 1, 0, ssize - 1)
    return s0, s1, (np.float32(1) - f).astype(np.float32), f, None


def resize_linear_f32(src, dh, dw):
    """cv2.resize(src, (dw, dh)) with the default INTER_LINEAR for a 2-D float32 array (see the module docstring)."""
    src = np.ascontiguousarray(src, np.float32)
    sh, sw = src.shape
    if (sh, sw) == (dh, dw):
        return src.copy()
    x0, x1, a0, a1, single = _taps(sw, dw, True)
    y0, y1, b0, b1, _ = _taps(sh, dh, False)
    rows = (src[:, x0] * a0[None, :]).astype(np.float32)
    two = (rows + (src[:, x1] * a1[None, :]).astype(np.float32)).astype(np.float32)
    rows = np.where(single[None, :], src[:, x0], two).astype(np.float32)
    out = ((rows[y0] * b0[:, None]).astype(np.float32) + (rows[y1] * b1[:, None]).astype(np.float32)).astype(np.float32)
    return out


def paste_masks(predictions, input_h, input_w, image_w, image_h, seg_thresh):
    """== test.py:127-157 with `args.input_h/input_w/seg_thresh`; predictions = [mask_patches, mask_dets] with NumPy patches.
    Returns [masks float32 [n, image_h, image_w], dets float32 [n, 5]] (or None)."""
    if predictions is None:
        return None
    out_masks, out_dets = [], []
    for pp, dd in zip(*predictions):
        for patch, det in zip(pp, dd):
            patch = np.asarray(patch, np.float32); det = np.asarray(det, np.float32)
            y1, x1, y2, x2, conf = det
            y1 = max(0, int(np.int32(np.round(y1)))); x1 = max(0, int(np.int32(np.round(x1))))
            y2 = min(int(np.int32(np.round(y2))), input_h - 1); x2 = min(int(np.int32(np.round(x2))), input_w - 1)
            mask = np.zeros((input_h, input_w), np.float32)
            mask[y1:y2, x1:x2] = resize_linear_f32(patch, y2 - y1, x2 - x1)
            mask = resize_linear_f32(mask, image_h, image_w)
            out_masks.append(np.where(mask >= seg_thresh, 1, 0))
            out_dets.append([float(y1) / input_h * image_h, float(x1) / input_w * image_w, float(y2) / input_h * image_h,
                             float(x2) / input_w * image_w, conf])
    return [np.asarray(out_masks, np.float32), np.asarray(out_dets, np.float32)]

"""Seeded synthetic inputs for parity tests and benches (TEST/BENCH INFRASTRUCTURE).

The build's own generator of ground-truth-like head maps (keypoint discs, short
and mid offsets; semantics of preprocessing.py:45-105 of the reference: disc of
radius KP_RADIUS around each of the 5 box keypoints, short offset = keypoint -
pixel, mid offset = other keypoint - pixel, later instances overwrite earlier
ones) plus seeded noise.  Nothing here is read from the reference at run time.
"""
import numpy as np

EDGES = [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4)]
DIR_EDGES = EDGES + [e[::-1] for e in EDGES]
KP_RADIUS = 5


def random_boxes(H, W, n, seed, smin=14, smax=40):
    """n axis-aligned boxes (y1,x1,y2,x2) with integer corners, sides U[smin,smax)."""
    rng = np.random.default_rng(seed)
    hs = rng.integers(smin, smax, n); ws = rng.integers(smin, smax, n)
    hs = np.minimum(hs, H - 2); ws = np.minimum(ws, W - 2)
    y1 = (rng.random(n) * (H - 1 - hs)).astype(np.int64) + 1
    x1 = (rng.random(n) * (W - 1 - ws)).astype(np.int64) + 1
    return np.stack([y1, x1, y1 + hs, x1 + ws], 1).astype(np.float64)


def keypoints_of(boxes):
    """[n,5,2] (x,y): tl, tr, bl, br, center (dataset_base.py:58-79 ordering)."""
    y1, x1, y2, x2 = boxes.T
    return np.stack([np.stack([x1, y1], 1), np.stack([x2, y1], 1), np.stack([x1, y2], 1),
                     np.stack([x2, y2], 1), np.stack([(x1 + x2) / 2, (y1 + y2) / 2], 1)], 1)


def gt_maps(boxes, H, W):
    """55-channel GT-like map [55,H,W] f32: [0:5] kp discs, [5:15] short, [15:55] mid."""
    kps = keypoints_of(boxes)
    kp = np.zeros((5, H, W), np.float32)
    short = np.zeros((10, H, W), np.float32)
    mid = np.zeros((40, H, W), np.float32)
    r = KP_RADIUS
    dy, dx = np.mgrid[-r:r + 1, -r:r + 1]
    disc = (dx * dx + dy * dy) <= r * r
    dys, dxs = dy[disc], dx[disc]
    for j in range(len(boxes)):
        for i in range(5):
            cx, cy = kps[j, i]
            icx, icy = int(cx), int(cy)
            yy = icy + dys; xx = icx + dxs
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            yy, xx = yy[ok], xx[ok]
            kp[i, yy, xx] = 1.0
            short[2 * i, yy, xx] = cx - xx
            short[2 * i + 1, yy, xx] = cy - yy
            for e, (a, b) in enumerate(DIR_EDGES):
                if a == i:
                    mid[2 * e, yy, xx] = kps[j, b, 0] - xx
                    mid[2 * e + 1, yy, xx] = kps[j, b, 1] - yy
    return np.concatenate([kp, short, mid], 0)


def head_maps(H, W, n_cells, seed, sigma_kp=0.05, sigma_off=0.5, smin=14, smax=40):
    """Noisy prediction-like maps: kp [1,5,H,W], short [1,10,H,W], mid [1,40,H,W] f32, + boxes."""
    boxes = random_boxes(H, W, n_cells, seed, smin, smax)
    gt = gt_maps(boxes, H, W)
    rng = np.random.default_rng(seed + 7919)
    kp = np.clip(gt[0:5] + rng.normal(0, sigma_kp, (5, H, W)), 0, 1).astype(np.float32)
    short = (gt[5:15] + rng.normal(0, sigma_off, (10, H, W))).astype(np.float32)
    mid = (gt[15:55] + rng.normal(0, sigma_off, (40, H, W))).astype(np.float32)
    return kp[None], short[None], mid[None], boxes


def train_batch(N, H, W, seed, n_boxes=4, smin=14, smax=30):
    """Seeded training batch in the collater's layout (collater.py:4-25): image [N,3,H,W] f32,
    GT maps for the 4 scales [N,55,H/s,W/s], instance masks (ellipses) and GT boxes [n,5]."""
    import torch
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(seed)) - 0.5
    gt_boxes, gt_masks = [], []
    for i in range(N):
        bx = random_boxes(H, W, n_boxes, 300 + i, smin, smax)
        gt_boxes.append(np.concatenate([bx, np.ones((len(bx), 1))], 1).astype(np.float32))
        m = np.zeros((len(bx), H, W), np.float32)
        for k, b in enumerate(bx.astype(int)):
            yy, xx = np.mgrid[0:H, 0:W]
            cy, cx = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2
            m[k] = (((yy - cy) / ((b[2] - b[0]) / 2 + .5)) ** 2 + ((xx - cx) / ((b[3] - b[1]) / 2 + .5)) ** 2 <= 1).astype(np.float32)
        gt_masks.append(m)
    gt_lv = [torch.from_numpy(np.stack([gt_maps(np.floor(gt_boxes[i][:, :4] / sc), H // sc, W // sc) for i in range(N)]))
             for sc in (1, 2, 4, 8)]
    return x, gt_boxes, gt_masks, gt_lv


def grad_sample_index(name, numel, k=1024):
    """Seeded subset (sorted flat indices) of a parameter's gradient kept in the golden fixtures (tests/golden/net_cal.npz
    stores min(numel, k) entries per parameter instead of 296 MB of gradients)."""
    import zlib
    if numel <= k:
        return np.arange(numel)
    return np.sort(np.random.default_rng([17, zlib.crc32(name.encode())]).choice(numel, k, replace=False))

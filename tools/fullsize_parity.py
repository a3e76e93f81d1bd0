#!/usr/bin/env python
"""GPU box: the train step of BASELINE's configuration (batch 8 x 512 x 512, 300 boxes per image; calibrated weights) in the default
policy against the same step in `fp32bf_full` (hi + mid + lo bf16 planes == fp32 values exactly, 6 products in forward AND backward:
round 2's fp32-faithful policy, itself pinned to the reference at small sizes): losses, head maps, and the FULL gradient of every
parameter.  The CPU oracle needs ~40 s for this batch; two GPU policies that are each held to the reference where the oracle is
affordable compare in seconds.      python tools/fullsize_parity.py [batch] [size] [boxes] > profiles/r03_fullsize_parity.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from kg_instance_segmentation_amd import KGnet
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.seg_loss import SEG_loss
from oracle import weightgen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda", 0)
sd = weightgen.gen_state_dict(0, variant="cal")
x, gt, gt_masks, gt_boxes = bench.make_batch(N, S, NB, 100, dev)
ldec, lseg = DetectionLossAll(5), SEG_loss(S, S)
res = {}
for pol in ("fp32", "fp32bf_full"):
    m = KGnet.resnet50(pretrained=False, precision=pol); m.load_state_dict(sd); m = m.to(dev).train()
    d = m(x, gt_boxes)
    l1 = [ldec(p, g) for p, g in zip(d[:4], gt)]
    l2 = lseg(d[4], gt_masks, gt_boxes)
    (sum(l1) + l2).backward()
    torch.cuda.synchronize()
    res[pol] = ([float(v) for v in l1] + [float(l2)], [t.detach().clone() for dd in d[:4] for t in dd[1:]],
                {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, m.grad_overflowed())
    del m, d
    torch.cuda.empty_cache()
a, b = res["fp32"], res["fp32bf_full"]
print(f"train step, batch {N} x {S} x {S}, {NB} boxes/img, calibrated weights: default policy `fp32` vs `fp32bf_full`")
print("losses  fp32       :", ["%.7f" % v for v in a[0]])
print("losses  fp32bf_full:", ["%.7f" % v for v in b[0]], " max rel diff %.2e" % max(abs(u - v) / abs(v) for u, v in zip(a[0], b[0])))
w = 0.0
for t, u in zip(a[1], b[1]):
    rms = float(u.double().pow(2).mean().sqrt())
    w = max(w, float(((t - u).abs() / (1e-5 * max(1.0, rms) + 1e-4 * u.abs())).max()))
print("offset maps (train mode, all 8 maps, every element): worst |d| / (rtol 1e-4, atol 1e-5 max(1, rms)) = %.3f" % w)
rows = []
for n, g in a[2].items():
    h = b[2][n]
    g64, h64 = g.double().flatten(), h.double().flatten()
    rows.append((float(g64 @ h64 / (g64.norm() * h64.norm() + 1e-300)), n, float(g64.norm() / (h64.norm() + 1e-300))))
rows.sort()
print("parameter gradients (%d tensors, full): min cosine %.7f (%s), median %.8f; norm ratio in [%.5f, %.5f]; overflow flags %s / %s" %
      (len(rows), rows[0][0], rows[0][1], rows[len(rows) // 2][0], min(r for _, _, r in rows), max(r for _, _, r in rows), a[3], b[3]))

#!/usr/bin/env python
"""Host time between the loss read-back of step k and the FIRST kernel launch of step k + 1 (the GPU idles meanwhile: tools/gap_probe.py shows one
~1 ms gap per step in front of img_pack_kernel).  Marks: return of loss.item() -> zero_grad done -> forward() entered -> forward_dec entered ->
ops.img_pack called.      python tools/stepstart_probe.py        (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kg_instance_segmentation_amd import KGnet, ops, engine as kengine
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.optim import Adam
from kg_instance_segmentation_amd.seg_loss import SEG_loss

dev = torch.device("cuda", 0)
x, gt, gt_masks, gt_boxes = bench.make_batch(8, 512, 300, 100, dev)
torch.manual_seed(1234)
model = KGnet.resnet50(pretrained=False).to(dev).train()
opt = Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-4, prepack=model)
ldec, lseg = DetectionLossAll(5), SEG_loss(512, 512)
marks = {}
orig_pack, orig_fd, orig_fwd = ops.img_pack, kengine.Engine.forward_dec, KGnet.ResNet.forward


def img_pack(*a, **k):
    marks.setdefault("img_pack", time.perf_counter())
    return orig_pack(*a, **k)


def fd(self, *a, **k):
    marks.setdefault("forward_dec", time.perf_counter())
    return orig_fd(self, *a, **k)


def fwd(self, *a, **k):
    marks.setdefault("forward", time.perf_counter())
    return orig_fwd(self, *a, **k)


ops.img_pack, kengine.Engine.forward_dec, KGnet.ResNet.forward = img_pack, fd, fwd
rows = []
for it in range(12):
    t0 = time.perf_counter()
    marks.clear()
    opt.zero_grad()
    marks["zero_grad"] = time.perf_counter()
    p0, p1, p2, p3, pred = model(x, gt_boxes)
    loss = ldec(p0, gt[0]) + ldec(p1, gt[1]) + ldec(p2, gt[2]) + ldec(p3, gt[3]) + lseg(pred, gt_masks, gt_boxes)
    loss.backward()
    opt.step()
    loss.item()
    if it >= 4:
        rows.append([1e6 * (marks[k] - t0) for k in ("zero_grad", "forward", "forward_dec", "img_pack")])
r = np.array(rows).mean(0)
print("host microseconds after the previous step's loss.item() returned (mean of 8 steps): zero_grad done %.0f, forward() entered %.0f, "
      "forward_dec entered %.0f, first kernel (img_pack) launched %.0f" % tuple(r))

#!/usr/bin/env python
"""Host-side timeline of the bench train step (no extra synchronisation): how long the host spends enqueuing each phase
and how long it then waits in loss.item().  Host-bound if the final wait is ~0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kg_instance_segmentation_amd import KGnet
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.seg_loss import SEG_loss
dev = torch.device("cuda", 0)
model = KGnet.resnet50(pretrained=False).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
ldec, lseg = DetectionLossAll(5), SEG_loss(512, 512)
x, gt, gt_masks, gt_boxes = bench.make_batch(8, 512, 300, 100, dev)
for it in range(5):
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    opt.zero_grad(); ts.append(time.perf_counter())
    d0, d1, d2, d3, feats = model.forward_dec(x); ts.append(time.perf_counter())
    pred = model.forward_seg(feats, gt_boxes); ts.append(time.perf_counter())
    l1 = ldec(d0, gt[0]) + ldec(d1, gt[1]) + ldec(d2, gt[2]) + ldec(d3, gt[3]); ts.append(time.perf_counter())
    l2 = lseg(pred, gt_masks, gt_boxes); ts.append(time.perf_counter())
    loss = l1 + l2
    loss.backward(); ts.append(time.perf_counter())
    opt.step(); ts.append(time.perf_counter())
    v = loss.item(); ts.append(time.perf_counter())
    names = ["zero_grad", "fwd_dec", "fwd_seg", "loss_dec", "loss_seg", "backward", "opt.step", "item(wait)"]
    print(" ".join(f"{n}={1e3*(b-a):.1f}" for n, a, b in zip(names, ts, ts[1:])), f"total={1e3*(ts[-1]-ts[0]):.1f}")

#!/usr/bin/env python
"""Where does the train step's GPU time go, by PART OF THE NETWORK (not by kernel)?  HIP events at the phase boundaries of one train
step at the bench configuration (engine.phase_hook: c0_conv / stem / layer1-3 / decoder / heads per level, forward and backward; the
seg branch, the losses, the optimizer and the weight pack are bracketed from outside), averaged over a few steps.

    python tools/phase_probe.py [--precision fp32] [--steps 5] [--out profiles/r04_phase_probe.txt]        (GPU box)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from kg_instance_segmentation_amd import KGnet
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.optim import Adam
from kg_instance_segmentation_amd.seg_loss import SEG_loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--boxes", type=int, default=300)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    x, gt, gt_masks, gt_boxes = bench.make_batch(args.batch, args.size, args.boxes, 100, dev)
    torch.manual_seed(1234)
    model = KGnet.resnet50(pretrained=False, precision=args.precision).to(dev).train()
    opt = Adam(model.parameters(), lr=1e-4)
    ldec, lseg = DetectionLossAll(5), SEG_loss(args.size, args.size)
    eng, seg = model._engine, model._seg
    marks = []

    def mark(label):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((label, e))

    def wrap(obj, name, label):
        orig = getattr(obj, name)

        def f(*a, **k):
            mark(label)
            r = orig(*a, **k)
            mark(label + ":end")
            return r
        setattr(obj, name, f)
    wrap(seg, "run_forward", "fwd seg branch")
    wrap(seg, "run_backward", "bwd seg branch")
    wrap(eng, "prepare_all", "weight pack (queued)")
    tot = {}
    order = []
    for it in range(args.steps + 2):
        marks.clear()
        eng.phase_hook = (lambda d, l: mark(f"{d} {l}")) if it >= 2 else None
        mark("zero_grad")
        opt.zero_grad()
        p0, p1, p2, p3, pred = model(x, gt_boxes)
        mark("losses fwd")
        loss = ldec(p0, gt[0]) + ldec(p1, gt[1]) + ldec(p2, gt[2]) + ldec(p3, gt[3])
        l2 = lseg(pred, gt_masks, gt_boxes)
        loss = loss + l2
        mark("backward: losses + grad scale")
        loss.backward()
        mark("adam")
        opt.step()
        mark("step end")
        torch.cuda.synchronize()
        if it < 2:
            continue
        for (l0, e0), (l1, e1) in zip(marks[:-1], marks[1:]):
            lab = l0
            if lab.endswith(":end") or lab in ("fwd end", "bwd end"):
                lab = "(between: " + l0.replace(":end", "") + " -> " + l1 + ")"
            if lab not in tot:
                tot[lab] = 0.0
                order.append(lab)
            tot[lab] += e0.elapsed_time(e1)
    lines = [f"phase probe: {args.precision}, batch {args.batch} x {args.size}^2, {args.boxes} boxes/img, mean of {args.steps} steps (HIP events at phase boundaries; ms)"]
    s = 0.0
    for lab in order:
        v = tot[lab] / args.steps
        s += v
        lines.append(f"{v:8.3f}  {lab}")
    lines.append(f"{s:8.3f}  total")
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""cProfile of the host side of the train step (which Python functions the ~800 launches of a step spend their enqueue time in)."""
import cProfile, os, pstats, sys, time
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


def step():
    opt.zero_grad()
    d0, d1, d2, d3, pred = model(x, gt_boxes)
    loss = sum(ldec(p, g) for p, g in zip((d0, d1, d2, d3), gt)) + lseg(pred, gt_masks, gt_boxes)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 5:.2f} ms/step, wall {1e3 * (t2 - t0) / 5:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
# the backward pass runs on autograd's thread: profile the engine's backward there
from kg_instance_segmentation_amd import KGnet as K
pb = cProfile.Profile()
orig = K._NetFunction.backward


def wrapped(ctx, *grads):
    pb.enable()
    try:
        return orig(ctx, *grads)
    finally:
        pb.disable()


K._NetFunction.backward = staticmethod(wrapped)
for _ in range(3):
    step()
torch.cuda.synchronize()
print("---- _NetFunction.backward (3 steps) ----")
pstats.Stats(pb).sort_stats("tottime").print_stats(30)

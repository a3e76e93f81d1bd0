#!/usr/bin/env python
"""How long do the torch.empty calls of the backward pass take (caching-allocator behaviour on the autograd thread)?"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kg_instance_segmentation_amd import KGnet, ops
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.seg_loss import SEG_loss
from kg_instance_segmentation_amd.optim import Adam

dev = torch.device("cuda", 0)
model = KGnet.resnet50(pretrained=False).to(dev).train()
opt = Adam(model.parameters(), lr=1e-4)
ldec, lseg = DetectionLossAll(5), SEG_loss(512, 512)
x, gt, gt_masks, gt_boxes = bench.make_batch(8, 512, 300, 100, dev)
rec = []
real_empty = torch.empty


def timed_empty(*a, **k):
    t0 = time.perf_counter()
    r = real_empty(*a, **k)
    rec.append((time.perf_counter() - t0, r.numel() * r.element_size()))
    return r


def step():
    opt.zero_grad()
    d0, d1, d2, d3, pred = model(x, gt_boxes)
    loss = sum(ldec(p, g) for p, g in zip((d0, d1, d2, d3), gt)) + lseg(pred, gt_masks, gt_boxes)
    loss.backward()
    opt.step()
    return loss


for _ in range(4):
    step()
torch.cuda.synchronize()
torch.empty = timed_empty
for _ in range(3):
    step()
torch.empty = real_empty
torch.cuda.synchronize()
tot = sum(t for t, _ in rec)
print(f"{len(rec) // 3} torch.empty per step, {1e3 * tot / 3:.2f} ms per step")
by = collections.defaultdict(lambda: [0, 0.0])
for t, n in rec:
    k = 0 if n < (1 << 20) else (1 if n < (64 << 20) else 2)
    by[k][0] += 1; by[k][1] += t
for k, name in enumerate(("< 1 MB", "1-64 MB", ">= 64 MB")):
    c, t = by[k]
    print(f"{name:10s}: {c // 3:4d} calls/step, {1e6 * t / max(c, 1):8.1f} us each, {1e3 * t / 3:6.2f} ms/step")
print("slowest:", sorted(((round(1e6 * t), n >> 20) for t, n in rec), reverse=True)[:12], "(us, MB)")
print(torch.cuda.memory_stats()["num_alloc_retries"], "alloc retries;", torch.cuda.memory_stats()["num_device_alloc"], "device allocs;",
      torch.cuda.memory_stats()["num_device_free"], "device frees")

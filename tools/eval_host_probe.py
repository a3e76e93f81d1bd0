#!/usr/bin/env python
"""Host-enqueue vs GPU time of eval-mode forward_dec at batch 1 (is inference launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kg_instance_segmentation_amd import KGnet

dev = torch.device("cuda:0")
m = KGnet.resnet50(pretrained=False).to(dev).eval()
for S in (256, 512, 1024):
    x = (torch.rand(1, 3, S, S) - 0.5).to(dev)
    with torch.no_grad():
        for _ in range(3):
            m.forward_dec(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m.forward_dec(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"S={S}: host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/img, wall {1e3 * (t2 - t0) / 20:.2f} ms/img")

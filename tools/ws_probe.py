#!/usr/bin/env python
"""Times kg_conv3x3_ws at the c0 shape of the bench configuration (8 x 512 x 512, 64 -> 64, hi + lo planes), dense and ragged (2400 boxes):
    python tools/ws_probe.py        (GPU box; KG_LIB_F16_PATH selects another build for A/B)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kg_instance_segmentation_amd import ops

dev = "cuda"
N, H = 8, 512
M = N * H * H
x = ops.alloc_pt(M, 64, 2, dev, dtype=ops.F16)
ops.base(x).normal_(0, 0.5)
x.plane(1).mul_(2.0 ** -11)
pw = ops.PackedWeight(64, 9, 64, dev, xP=2, wP=2, dtype=ops.F16)
pw.pack(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
b = torch.zeros(64, device=dev)
y = ops.alloc_pt(M, 64, 2, dev, dtype=ops.F16)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


t = timed(lambda: ops.conv_halo(x, pw, 64, N, H, H, 3, y=y, bias=b, relu=True))
fl = 2.0 * M * 64 * 9 * 64 * 3
print(f"dense 8x512^2: {t:.3f} ms = {fl / t / 1e9:.0f} TFLOP/s issued, {(M * 64 * 2 * 2 * 2) / t / 1e6:.0f} GB/s in+out")
ops.USE_WS = False
t0 = timed(lambda: ops.conv_halo(x, pw, 64, N, H, H, 3, y=y, bias=b, relu=True))
print(f"conv_halo<3> on the same problem: {t0:.3f} ms")

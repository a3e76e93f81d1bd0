#!/usr/bin/env python
"""Times the first-layer 7x7 head conv (kg_conv2d_halo, KS = 7) alone at the three shapes that carry it in the bench step, forward with hi + lo
planes (3 products) and single-plane (the input-gradient arithmetic), on ReLU-like operands:
    python tools/halo7_probe.py        (GPU box; KG_HALO7_W4=0 -> the 8-wave kernel everywhere, 2 -> the 4-wave blocked-accumulation kernel everywhere; KG_LIB_F16_PATH selects another build)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kg_instance_segmentation_amd import ops

dev = "cuda"


def timed(fn, reps=int(os.environ.get("KG_PROBE_REPS", "10"))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


tot = 0.0
for (N, H, cin, cout) in ((8, 512, 64, 192), (8, 128, 256, 768), (8, 64, 512, 1536)):
    for P in (2, 1):
        M = N * H * H
        x = ops.alloc_pt(M, cin, P, dev, dtype=ops.F16) if P > 1 else torch.empty(M, cin, dtype=ops.F16, device=dev)
        ops.base(x).normal_(0, 0.5).clamp_(min=0)
        if P > 1:
            x.plane(1).mul_(2.0 ** -11)
        pw = ops.PackedWeight(cout, 49, cin, dev, xP=P, wP=P, dtype=ops.F16)
        pw.pack(torch.randn(cout, cin, 7, 7, device=dev) * 0.02)
        b = torch.zeros(cout, device=dev)
        y = ops.alloc_pt(M, cout, P, dev, dtype=ops.F16) if P > 1 else torch.empty(M, cout, dtype=ops.F16, device=dev)
        t = timed(lambda: ops.conv_halo(x, pw, cout, N, H, H, 7, y=y, bias=b, relu=True))
        fl = 2.0 * M * cout * 49 * cin * (3 if P > 1 else 1)
        tot += t
        print(f"N={N} H={H} {cin:4d}->{cout:4d} planes={P}: {t:7.3f} ms  {fl / t / 1e9:7.0f} TFLOP/s issued")
        del x, y, pw
print(f"sum {tot:.3f} ms")

#!/usr/bin/env python
"""Inference forward_dec at batch 1 (BASELINE configs[4]: one image at a time), eager: time per call; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel picture.      python tools/eval_fwd_probe.py [size] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kg_instance_segmentation_amd import KGnet


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = KGnet.resnet50(pretrained=False).to(dev).eval()
    x = torch.rand(1, 3, S, S, device=dev) - 0.5
    with torch.no_grad():
        for _ in range(3):
            model.forward_dec(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model.forward_dec(x)
        host = (time.perf_counter() - t0) / reps
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
    print(f"size {S}: {wall * 1e3:.3f} ms per forward_dec (host enqueue {host * 1e3:.3f} ms)")


if __name__ == "__main__":
    main()

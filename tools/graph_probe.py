#!/usr/bin/env python
"""hipGraph capture of the eval-mode forward_dec (batch 1): does it capture, does the replay equal the eager run, what does it save?
    python tools/graph_probe.py [size]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kg_instance_segmentation_amd import KGnet


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = KGnet.resnet50(pretrained=False).to(dev).eval()
    x = torch.rand(1, 3, S, S, device=dev) - 0.5

    def flat(o):
        return [t for lvl in o[:4] for t in lvl] + list(o[4])

    with torch.no_grad():
        for _ in range(3):
            ref = flat(model.forward_dec(x))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model.forward_dec(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 20
        ref = [t.clone() for t in ref]
        g = torch.cuda.CUDAGraph()
        xs = x.clone()
        with torch.cuda.graph(g):
            out = flat(model.forward_dec(xs))
        xs.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(ref, out))
        t0 = time.perf_counter()
        for _ in range(20):
            xs.copy_(x)
            g.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / 20
    print(f"size {S}: eager {eager * 1e3:.3f} ms, graph replay {rep * 1e3:.3f} ms, identical outputs: {same}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Micro-benchmark of single conv kernels at KGnet's shapes (used for rocprofv3 --pmc runs and tuning).
    python tools/kbench.py halo7_c0 [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kg_instance_segmentation_amd import ops
from kg_instance_segmentation_amd.ops import BF16, PackedWeight

CASES = {
    # name: (kind, N, H, cin, cout, k)
    "halo7_c0": ("fwd", 8, 512, 64, 192, 7),
    "halo7_c0_64": ("fwd", 8, 512, 192, 64, 7),
    "halo7_c2": ("fwd", 8, 128, 256, 768, 7),
    "halo7_c3": ("fwd", 8, 64, 512, 1536, 7),
    "halo7_c3d": ("fwd", 8, 64, 1536, 512, 7),
    "halo3_c0": ("fwd", 8, 512, 64, 64, 3),
    "halo3_c2": ("fwd", 8, 128, 256, 512, 3),
    "halo3_c3": ("fwd", 8, 64, 512, 1024, 3),
    "halo3_l2": ("fwd", 8, 64, 128, 128, 3),
    "halo3_l3": ("fwd", 8, 32, 256, 256, 3),
    "wg7_c0": ("wgrad", 8, 512, 64, 192, 7),
    "wg7_c3": ("wgrad", 8, 64, 512, 1536, 7),
    "wg3_c0": ("wgrad", 8, 512, 64, 64, 3),
    "wg1_l1": ("wgrad", 8, 128, 64, 256, 1),
    "wg1_l1b": ("wgrad", 8, 128, 256, 64, 1),
    "wg1_l3": ("wgrad", 8, 32, 256, 1024, 1),
    "wg1_c0": ("wgrad", 8, 512, 128, 64, 1),
    "igemm1_c0": ("igemm", 8, 512, 128, 64, 1),
    "small3_c0": ("igemm", 8, 512, 8, 64, 3),        # c0_conv.0 (3 -> 64, image rows padded to 8 channels): conv_small_mfma
    "small7_stem": ("igemm_s2", 8, 512, 8, 64, 7),   # stem conv1 (7x7 stride 2)
    "igemm3_deep": ("igemm", 8, 64, 512, 256, 3),
    "igemm3_deep2": ("igemm", 8, 32, 1024, 512, 3),
}


def main():
    name = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    kind, N, H, cin, cout, k = CASES[name]
    dev = "cuda"
    M = N * H * H
    const = os.environ.get("KG_KBENCH_CONST") == "1"     # constant operands: the clock the chip sustains depends on the data toggling
    rnd = (lambda *sh: torch.full(sh, 1.0, device=dev)) if const else (lambda *sh: torch.randn(*sh, device=dev))
    x = (rnd(M, cin) * 0.5).to(BF16)
    geom = (M, H, H, H, H, k, k, 1, k // 2)
    if kind == "igemm_s2":
        kind, geom = "igemm", (N * (H // 2) ** 2, H, H, H // 2, H // 2, k, k, 2, k // 2)
    flops = 2.0 * M * cout * k * k * cin
    if kind in ("fwd", "igemm"):
        pw = PackedWeight(cout, k * k, cin, dev)
        pw.buf.copy_((torch.randn_like(pw.buf.float()) * 0.05).to(BF16))
        y = torch.empty(geom[0], cout, dtype=BF16, device=dev)
        bias = torch.zeros(cout, device=dev)
        if kind == "fwd":
            fn = lambda: ops.conv_halo(x, pw, cout, N, H, H, k, y=y, bias=bias, relu=True)
        else:
            fn = lambda: ops.conv_igemm(x, pw, cout, geom, y=y, bias=bias, relu=True)
    else:
        dy = (rnd(M, cout) * 0.5).to(BF16)
        gw = torch.empty(cout, cin, k, k, device=dev)
        fn = lambda: ops.conv_wgrad(x, dy, cin, cout, geom, [(gw, 0, cout)], N=N)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"{name}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Groups profiles/<tag>_bench_kernel_stats.csv into the DESIGN.md section-3 table (ms/step and launches/step per kernel family)."""
import csv
import sys

GROUPS = [
    ("7x7 heads: conv_halo<7,1,8,0> + conv_halo7_w4 forward + input gradient (the dominant kernel family)", lambda n: "conv_halo_kernel<7, 1, 8, 0" in n or "conv_halo7_w4_kernel" in n),
    ("7x7 heads: grouped second layers <7,1,8,1> (+ heads2_finish) + the kp / short input gradients (conv7_narrow; <7,1,8,2..4>)", lambda n: "conv_halo_kernel<7, 1, 8, 1" in n or "conv_halo_kernel<7, 1, 8, 2" in n or "conv_halo_kernel<7, 1, 8, 3" in n or "conv_halo_kernel<7, 1, 8, 4" in n or "conv7_narrow" in n or "heads2_finish" in n),
    ("7x7 heads: weight gradients wgrad_halo<7,...>", lambda n: "wgrad_halo_kernel<7" in n),
    ("3x3: conv_halo<3> + conv_halo3_w4 + conv3_ws + conv3_c64 + wgrad_halo<3>", lambda n: "conv_halo_kernel<3" in n or "conv_halo3_w4" in n or "conv3_c64" in n or "conv3_ws" in n or "wgrad_halo_kernel<3" in n),
    ("conv_gather", lambda n: "conv_gather" in n),
    ("gather weight gradients conv_wgrad128 / conv_wgrad", lambda n: "conv_wgrad" in n),
    ("conv1x1_*, conv_small, im2col, conv_igemm", lambda n: "conv1x1" in n or "conv_small" in n or "im2col" in n or "conv_igemm" in n),
    ("BatchNorm (colreduce, bn_*)", lambda n: "colreduce" in n or n.startswith("bn_")),
    ("split reductions + bias gradients", lambda n: "wgrad_reduce" in n or "bias_grad" in n),
    ("bilinear, max-pool, gradient joins, packing of the map gradients", lambda n: "bilinear" in n or "maxpool" in n or "add_rows" in n or "grad_pack" in n or "img_pack" in n),
    ("seg plumbing, losses", lambda n: any(k in n for k in ("crop_grad", "rows_gather", "seg_", "det_loss", "gt_maps", "sum_final", "sigmoid_inplace", "planes_to", "f32_to"))),
    ("weight packing + Adam", lambda n: "pack_weight" in n or "adam" in n),
    ("finishing passes of the K splits", lambda n: "conv_tiny_finish" in n or "conv_halo_finish" in n),
    ("gradient-scale bookkeeping (grad_scale, rows_absmax / rows_scale*, scale_tensors)", lambda n: any(k in n for k in ("grad_scale", "rows_absmax", "rows_scale", "scale_tensors"))),
]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    left = list(rows)
    tot_ms = tot_l = 0.0
    for title, pred in GROUPS:
        sel = [r for r in left if pred(r["Name"])]
        left = [r for r in left if not pred(r["Name"])]
        ms, l = sum(float(r["MsPerStep"]) for r in sel), sum(float(r["CallsPerStep"]) for r in sel)
        tot_ms += ms; tot_l += l
        print(f"| {title} | {ms:.2f} | {l:.0f} |")
    ms, l = sum(float(r["MsPerStep"]) for r in left), sum(float(r["CallsPerStep"]) for r in left)
    print(f"| torch fills / copies / other | {ms:.2f} | {l:.0f} |")
    print(f"total {tot_ms + ms:.2f} ms/step, {tot_l + l:.0f} launches/step")
    small = [r for r in rows if float(r["AverageNs"]) < 30000]
    print(f"kernels below 30 us: {sum(float(r['CallsPerStep']) for r in small):.0f} launches, {sum(float(r['MsPerStep']) for r in small):.2f} ms")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Gradient-precision table: how far are the parameter gradients of each precision policy from the TRUE gradients of the step?

The reference's backward pass is fp32 autograd (train.py:148-154).  Gradient cosines against the reference's fp32 gradients are floor-limited
(every policy sits at 1 - cos ~ 1e-5: the yardstick itself is fp32), so this tool measures, per parameter tensor, the relative L2 error
    e(g) = || g - g64 || / || g64 ||
against a FLOAT64 evaluation of the reference-pinned CPU oracle (oracle/gradref.py, oracle/net.py) of the same train step, for
  * "oracle_fp32": the same oracle in float32 = the reference's own arithmetic (the floor no fp32 implementation gets under),
  * every policy named on the command line (default: fp32 fp32b2 half fp32bf),
and, for the default policy, against `fp32b2` on the GPU (identical forward, so that column isolates the 11-bit backward operands).
Fixtures: the calibrated weights ("cal", tests/golden/net_cal.npz's network) and the reference's random init ("raw"), train step
N x S x S with NB boxes per image (default 2 x 128 x 128, 8 boxes: the golden train step's shape).  Tensors whose true gradient is
(numerically) zero -- rms(g64) below --floor -- are listed separately.

    python tools/grad_table.py [--out profiles/r04_grad_table.json] [--size 128] [--batch 2] [--boxes 8] [policy ...]      (GPU box)

Test infrastructure: imports oracle/ like tests/ do; nothing here is product code."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import gradref, synth, weightgen

DEV = "cuda"


def gpu_grads(sd, policy, batch, H, W):
    from kg_instance_segmentation_amd import KGnet
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    x, gt_boxes, gt_masks, gt_lv = batch
    m = KGnet.resnet50(pretrained=False, precision=policy)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.zero_grad()
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=H, width=W)
    d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
    loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv))
    l2 = lseg(pred, gt_masks, gt_boxes)
    if l2 is not None:
        loss = loss + l2
    loss.backward()
    torch.cuda.synchronize()
    assert not m.grad_overflowed()
    return float(loss.detach()), {n: (p.grad.detach().double().cpu() if p.grad is not None else None) for n, p in m.named_parameters()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("policies", nargs="*")
    ap.add_argument("--out", default=None)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--boxes", type=int, default=8)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--floor", type=float, default=1e-12, help="rms of the float64 gradient below which a tensor counts as degenerate")
    ap.add_argument("--fixtures", default="cal,raw")
    args = ap.parse_args()
    pols = args.policies or ["fp32", "fp32b2", "half", "fp32bf"]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    N, S = args.batch, args.size
    batch = synth.train_batch(N, S, S, args.seed, n_boxes=args.boxes)
    table = {"config": {"batch": N, "size": S, "boxes": args.boxes, "seed": args.seed, "degenerate_floor_rms": args.floor,
                        "metric": "per-tensor relative L2 error ||g - g64|| / ||g64|| against the float64 oracle; median / p90 / max over the parameter tensors"}}
    for fx in args.fixtures.split(","):
        sd = weightgen.gen_state_dict(0, variant=fx)
        t0 = time.time(); l64, g64 = gradref.oracle_grads(sd, *batch, S, S, torch.float64); t64 = time.time() - t0
        t0 = time.time(); l32, g32 = gradref.oracle_grads(sd, *batch, S, S, torch.float32); t32 = time.time() - t0
        res = {"loss_f64": l64, "oracle_seconds": {"f64": t64, "f32": t32}, "oracle_fp32": gradref.column(g32, g64, args.floor)}
        res["oracle_fp32"]["loss_rel_err"] = abs(l32 - l64) / abs(l64)
        grads = {}
        for p in pols:
            lp, gp = gpu_grads(sd, p, batch, S, S)
            grads[p] = gp
            res[p] = gradref.column(gp, g64, args.floor)
            res[p]["loss_rel_err"] = abs(lp - l64) / abs(l64)
            res[p]["ratio_to_oracle_fp32"] = {k: res[p][k] / max(res["oracle_fp32"][k], 1e-300) for k in ("median", "p90", "max")}
            res[p]["groups"] = gradref.by_group(res[p]["per_tensor"])
        res["oracle_fp32"]["groups"] = gradref.by_group(res["oracle_fp32"]["per_tensor"])
        if "fp32" in grads and "fp32b2" in grads:      # identical forward: the pure effect of single-plane backward operands
            res["fp32_vs_fp32b2"] = gradref.column(grads["fp32"], grads["fp32b2"], args.floor)
            res["fp32_vs_fp32b2"]["groups"] = gradref.by_group(res["fp32_vs_fp32b2"]["per_tensor"])
        table[fx] = res
        print(f"== fixture {fx}: loss {l64:.6f}; float64 oracle {t64:.0f} s, float32 oracle {t32:.1f} s", flush=True)
        print("| column | median | p90 | max | worst tensor | x oracle_fp32 (median / p90 / max) |")
        print("|---|---|---|---|---|---|")
        for c in ["oracle_fp32"] + pols + (["fp32_vs_fp32b2"] if "fp32_vs_fp32b2" in res else []):
            r = res[c]
            rt = r.get("ratio_to_oracle_fp32")
            print(f"| {c} | {r['median']:.2e} | {r['p90']:.2e} | {r['max']:.2e} | {r['worst'][0][0]} | " +
                  (f"{rt['median']:.1f} / {rt['p90']:.1f} / {rt['max']:.1f} |" if rt else "- |"), flush=True)
        for c in ["oracle_fp32"] + pols:
            print(f"   {c} by group (median):", {k: f"{v['median']:.1e}" for k, v in res[c]["groups"].items()})
        print("   degenerate tensors:", res["oracle_fp32"]["degenerate"])
    if args.out:
        with open(args.out, "w") as f:
            json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()

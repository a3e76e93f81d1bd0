#!/usr/bin/env python
"""Precision Pareto table: for every policy of engine.PRECISIONS (or those named on the command line) the worst |d| / bound of the
eval-mode logits, the train-mode maps, the parameter gradients of one train step -- all against the reference-generated fixture
tests/golden/net_cal.npz -- at BOTH tolerances of SURVEY 8d (fp32 clause: rtol 1e-4 / atol 1e-5; bf16 clause: rtol 2e-2 on
pre-sigmoid logits), and the train-step throughput at BASELINE's config (batch 8 x 512^2, 300 boxes).

    python tools/pareto.py [--out profiles/r03_pareto.json] [--steps 6] [policy ...]          (GPU box)

Test infrastructure: imports oracle/ (weights, synthetic batch) like tests/ do; nothing here is product code."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from kg_instance_segmentation_amd import KGnet, engine
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.seg_loss import SEG_loss
from oracle import synth, weightgen

DEV = "cuda"
FP32 = (1e-4, 1e-5)      # rtol, atol (x max(1, rms))
BF16 = (2e-2, 0.0)       # rtol only ("rtol 2e-2 on pre-sigmoid logits")


def ratios(got, ref):
    """worst |d| / bound under the fp32 clause, worst |d| / (rtol_bf16 |ref|) restricted to |ref| >= rms (a pure rtol is undefined
    at zero crossings), and max |d| / rms"""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    rms = float(np.sqrt(np.mean(ref ** 2)))
    d = np.abs(got - ref)
    r32 = float((d / (FP32[1] * max(1.0, rms) + FP32[0] * np.abs(ref))).max())
    big = np.abs(ref) >= rms
    r16 = float((d[big] / (BF16[0] * np.abs(ref[big]))).max()) if big.any() else 0.0
    return r32, r16, float(d.max() / max(rms, 1e-30))


def sub(t, step=3):
    a = t.detach().float().cpu().numpy()
    return a[..., ::step, ::step] if a.shape[-1] > 32 else a


def measure(policy, g, sd):
    out = {}
    m = KGnet.resnet50(pretrained=False, precision=policy)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m._engine.raw_kp_logits = True
    m._seg.keep_logits = True
    w32 = w16 = wr = 0.0
    with torch.no_grad():
        for name in ("a", "b"):
            N, H, W, s = [int(v) for v in g[f"{name}.cfg"]]
            x = (torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(s)) - 0.5).to(DEV)
            d0, d1, d2, d3, feats = m.forward_dec(x)
            for l, d in enumerate((d0, d1, d2, d3)):
                for nm, t in zip(("kp_logit", "short", "mid"), d):
                    a, b, c = ratios(sub(t), g[f"{name}.eval.c{l}.{nm}"])
                    w32, w16, wr = max(w32, a), max(w16, b), max(wr, c)
            if name == "b":
                pred = m.forward_seg(feats, [g["b.boxes0"], g["b.boxes1"]])
                meta, logits = pred.kg_meta, m._seg.last_logits
                per_img = [[j for j in range(len(meta["off"])) if int(meta["img"][j]) == i] for i in range(2)]
                for i in range(2):
                    for jj, j in enumerate(per_img[i]):
                        h, w, off = int(meta["h"][j]), int(meta["w"][j]), int(meta["off"][j])
                        a, b, c = ratios(logits[off:off + h * w].view(h, w).cpu().numpy(), g[f"b.seg_logit.{i}.{jj}"])
                        w32, w16, wr = max(w32, a), max(w16, b), max(wr, c)
    out["eval_logits"] = {"worst_over_fp32_bound": w32, "worst_over_bf16_rtol_at_|ref|>=rms": w16, "max|d|/rms": wr}
    # one train step (2 x 128 x 128): train-mode maps, losses, every parameter gradient
    N, H, W, s, nb = [int(v) for v in g["train.cfg"]]
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, H, W, s, n_boxes=nb)
    m.train()
    m._engine.raw_kp_logits = False
    m.zero_grad()
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=H, width=W)
    d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
    l1 = [ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)]
    l2 = lseg(pred, gt_masks, gt_boxes)
    w32 = w16 = wr = 0.0
    for l, d in enumerate((d0, d1, d2, d3)):
        for nm, t in (("short", d[1]), ("mid", d[2])):
            a, b, c = ratios(sub(t, 5), g[f"train.c{l}.{nm}"])
            w32, w16, wr = max(w32, a), max(w16, b), max(wr, c)
    out["train_maps"] = {"worst_over_fp32_bound": w32, "worst_over_bf16_rtol_at_|ref|>=rms": w16, "max|d|/rms": wr}
    out["loss_rel_err"] = float(max(np.max(np.abs(np.array([float(v) for v in l1]) / g["train.loss_dec"] - 1)),
                                    abs(float(l2) / float(g["train.loss_seg"]) - 1)))
    (sum(l1) + l2).backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    off, rows = 0, []
    for n, nrm in zip([str(n) for n in g["train.grad_names"]], g["train.grad_norm"]):
        gr = params[n].grad.detach().cpu().numpy().ravel().astype(np.float64)
        idx = synth.grad_sample_index(n, gr.size)
        ref = g["train.grad_samples"][off:off + idx.size].astype(np.float64); off += idx.size
        got = gr[idx]
        cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-300))
        rows.append((cos, n, float(np.linalg.norm(gr)) / (float(nrm) + 1e-300)))
    rows.sort()
    rat = np.array([r for _, _, r in rows])
    out["grads"] = {"min_cosine": rows[0][0], "worst_param": rows[0][1], "one_minus_min_cosine": 1.0 - rows[0][0],
                    "max_norm_dev": float(np.abs(rat - 1).max())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("policies", nargs="*")
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--no-bench", action="store_true")
    args = ap.parse_args()
    pols = args.policies or list(engine.PRECISIONS)
    g = np.load(os.path.join(ROOT, "tests", "golden", "net_cal.npz"), allow_pickle=False)
    sd = weightgen.gen_state_dict(0, variant="cal")
    table = {}
    for p in pols:
        r = measure(p, g, sd)
        if not args.no_bench:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--precision", p, "--steps", str(args.steps), "--warmup", "2",
                   "--no-companion", "--no-cpu-baseline", "--no-kernel-timer"]
            pr = subprocess.run(cmd, capture_output=True, text=True)
            try:
                b = json.loads(pr.stdout.strip().splitlines()[-1])
                r["imgs_per_s"], r["ms_per_step"] = b["value"], b["ms_per_step"]
            except Exception as e:      # noqa: BLE001
                r["bench_error"] = (pr.stderr or str(e))[-400:]
        r["planes"] = engine.PRECISIONS[p]
        table[p] = r
        print(p, json.dumps(r), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()

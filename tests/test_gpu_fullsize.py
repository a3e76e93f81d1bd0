"""GPU parity AT THE BENCH CONFIGURATION (BASELINE configs[1]: batch 8 x 3 x 512 x 512, 300 GT boxes per image, train mode; the reference's
train.py:148-154): the default policy's train-mode forward element-wise against the reference-pinned CPU oracle (oracle/net.py) evaluated in
float32 -- the reference's own arithmetic -- AND in float64, every kp logit / short / mid offset / seg logit of the batch, plus the five losses
(and, with KG_FULLSIZE_GRADS=1, every parameter gradient against the float32 oracle's autograd: the oracle's backward pass through 2400 per-box
graphs takes 12 minutes of host time, so that leg is opt-in; its result on this build is in profiles/r05_fullsize_test.txt).

What can and cannot hold at this size (measured, profiles/r05_fullsize_test.txt): two float32-grade evaluations of this network are
each ~1.1-1.2 bounds (rtol 1e-4, atol 1e-5) from the float64 value -- the reference's own fp32 arithmetic included: 12 544 / 25 088-term dot
products in the c2 / c3 heads on top of 50 layers -- so "<= 1.0 against the float32 oracle" is not a statement any fp32 implementation can
make here.  What is asserted instead:
  * the policy is AS CLOSE TO FLOAT64 AS THE REFERENCE'S ARITHMETIC IS: worst |d| / bound of (policy vs oracle64) <= FLOOR_SLACK x the same
    number of (oracle32 vs oracle64), over all 13 maps;
  * against the float32 oracle AND against float64 every map stays under a FIXED cap (CAP_VS_ORACLE32 / CAP_VS_ORACLE64 = 1.1 x the values
    measured in round 5: 1.58 / 1.16 bounds at worst), with at most a 1e-3 fraction of any map's elements beyond the bound;
  * losses within 1e-6 of float64; (opt-in) every parameter gradient: cosine >= 0.9999 and norm within 2e-3 of the float32 oracle's.
Bounds: |d| <= atol + 1e-4 |ref| with atol 1e-5 for the logits (literally) and the stated per-map constants 3e-5 (short offsets,
rms 3-4 px) / 6e-5 (mid offsets, rms 5-7 px) -- no rms scaling."""
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

FLOOR_SLACK = 1.10      # policy-vs-float64 may exceed oracle32-vs-float64 by 10 % (max statistics over 1e8 elements; measured 1.160 / 1.125 = 1.03)
WITH_GRADS = os.environ.get("KG_FULLSIZE_GRADS", "0") == "1"
# Fixed per-map caps of worst |d| / bound of the policy against the FLOAT32 oracle = 1.1 x the values measured on MI355X (round 5 build,
# profiles/r05_fullsize_test.txt; the kernels are deterministic, the oracle runs on the box's host cores): two fp32-grade evaluations differ by
# about the sum of their distances from float64, so these sit above 1.0 for the kp logits and short offsets -- stated, not derived from the run.
CAP_VS_ORACLE32 = {"c0.kp_logit": 1.46, "c0.short": 1.64, "c0.mid": 1.42, "c1.kp_logit": 1.24, "c1.short": 1.40, "c1.mid": 1.08,
                   "c2.kp_logit": 1.74, "c2.short": 1.36, "c2.mid": 1.08, "c3.kp_logit": 1.60, "c3.short": 1.45, "c3.mid": 1.09, "seg_logit": 0.19}
# ... and against float64 (measured 1.009 / 0.985 / 0.719 | 0.820 / 0.929 / 0.691 | 1.160 / 0.884 / 0.703 | 1.096 / 0.963 / 0.700 | 0.133)
CAP_VS_ORACLE64 = {"c0.kp_logit": 1.11, "c0.short": 1.09, "c0.mid": 0.80, "c1.kp_logit": 0.91, "c1.short": 1.03, "c1.mid": 0.77,
                   "c2.kp_logit": 1.28, "c2.short": 0.98, "c2.mid": 0.78, "c3.kp_logit": 1.21, "c3.short": 1.06, "c3.mid": 0.77, "seg_logit": 0.15}


def test_train_step_at_bench_configuration_vs_oracle_fp32_and_fp64():
    """measured on MI355X (r05, this build; profiles/r05_fullsize_test.txt): oracle32 vs oracle64 worst 1.125 bounds; policy vs oracle64 1.160
    (ratio 1.03); policy vs oracle32 1.581; at most 1.2e-4 of a map's elements beyond the bound; losses equal to float64's to 1e-7; with
    KG_FULLSIZE_GRADS=1 all 217 gradients cosine >= 0.99998, norms within 1.2e-3 of the float32 oracle's."""
    import fullsize_oracle_parity as fs
    from kg_instance_segmentation_amd import KGnet
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    from oracle import net as onet, synth, weightgen
    N, S, NB = 8, 512, 300
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    dev = torch.device("cuda", 0)
    sd = weightgen.gen_state_dict(0, variant="cal")
    x, boxes, masks, gt_lv = synth.train_batch(N, S, S, 41, n_boxes=NB, smin=14, smax=40)

    # ---- the product: forward (maps, logits through the test hooks), losses, backward ----------------------------------------
    m = KGnet.resnet50(pretrained=False, precision="fp32")
    m.load_state_dict(sd)
    m = m.to(dev).train()
    m._engine.keep_kp_logits = True
    m._seg.keep_logits = True
    ldec, lseg = DetectionLossAll(5), SEG_loss(S, S)
    d = m(x.to(dev), boxes)
    got = {}
    for l in range(4):
        got[f"c{l}.kp_logit"] = m._engine.kp_logits[l].cpu()
        got[f"c{l}.short"] = d[l][1].detach().cpu()
        got[f"c{l}.mid"] = d[l][2].detach().cpu()
    meta, flat = d[4].kg_meta, m._seg.last_logits
    order = sorted(range(len(meta["off"])), key=lambda j: (int(meta["img"][j]), j))
    got["seg_logit"] = torch.cat([flat[int(meta["off"][j]):int(meta["off"][j]) + int(meta["h"][j]) * int(meta["w"][j])] for j in order]).cpu()
    l1 = [ldec(d[l], gt_lv[l].to(dev)) for l in range(4)]
    l2 = lseg(d[4], masks, boxes)
    lg = [float(v.detach()) for v in l1] + [float(l2.detach())]
    grads = {}
    if WITH_GRADS:
        (sum(l1) + l2).backward()
        torch.cuda.synchronize()
        assert not m.grad_overflowed()
        grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    del m, d, l1, l2
    torch.cuda.empty_cache()

    # ---- oracle float32 WITH autograd (the reference's train.py:148-153), oracle float64 forward ---------------------------------
    t0 = time.time()
    osd = {k: v.clone() for k, v in sd.items()}
    names = [k for k, v in osd.items() if v.is_floating_point() and "running" not in k]
    if WITH_GRADS:
        for n in names:
            osd[n].requires_grad_(True)
    net = onet.Net(osd, training=True)
    with torch.set_grad_enabled(WITH_GRADS):
        o = net.forward(x, boxes)
    o32 = {}
    for l in range(4):
        o32[f"c{l}.kp_logit"] = net.kp_logits[l].detach()
        o32[f"c{l}.short"] = o[l][1].detach()
        o32[f"c{l}.mid"] = o[l][2].detach()
    o32["seg_logit"] = torch.cat([z.detach().reshape(-1) for per in net.seg_logits for z in per])
    with torch.set_grad_enabled(WITH_GRADS):
        ol = [onet.detection_loss(o[l], gt_lv[l]) for l in range(4)] + [onet.seg_loss(o[4], masks, boxes, S, S)]
    if WITH_GRADS:
        sum(ol).backward()
    l32 = [float(v.detach()) for v in ol]
    g32 = {n: osd[n].grad for n in names}
    del net, o, ol
    t32 = time.time() - t0
    o64, l64, t64 = fs.oracle_forward(sd, x, boxes, gt_lv, masks, S, torch.float64)
    print(f"oracle float32 forward{' + backward' if WITH_GRADS else ''} {t32:.0f} s, float64 forward {t64:.0f} s")

    # ---- losses ---------------------------------------------------------------------------------------------------------------------
    print("losses oracle64:", ["%.8f" % v for v in l64])
    print("losses oracle32:", ["%.8f" % v for v in l32])
    print("losses policy  :", ["%.8f" % v for v in lg])
    assert max(abs(a - b) / abs(b) for a, b in zip(lg, l64)) <= 1e-6
    assert max(abs(a - b) / abs(b) for a, b in zip(lg, l32)) <= 2e-6

    # ---- maps: every element ----------------------------------------------------------------------------------------------------------
    floor = pol64 = pol32 = 0.0
    print("%-12s %9s %7s | o32 vs o64 | policy vs o32 (frac > 1) | policy vs o64 (frac > 1)" % ("map", "elements", "rms"))
    for name in o64:
        assert got[name].shape == o64[name].shape == o32[name].shape, name
        a = fs.worst_ratio(name, o32[name], o64[name])
        b = fs.worst_ratio(name, got[name], o32[name])
        c = fs.worst_ratio(name, got[name], o64[name])
        print("%-12s %9d %7.3g |   %6.3f   |   %6.3f (%.1e)     |   %6.3f (%.1e)" % (name, o64[name].numel(), a[2], a[0], b[0], b[1], c[0], c[1]))
        floor, pol32, pol64 = max(floor, a[0]), max(pol32, b[0]), max(pol64, c[0])
        assert b[1] <= 1e-3 and c[1] <= 1e-3, name
        assert b[0] <= CAP_VS_ORACLE32[name], (name, "policy vs oracle32", b[0], CAP_VS_ORACLE32[name])
        assert c[0] <= CAP_VS_ORACLE64[name], (name, "policy vs oracle64", c[0], CAP_VS_ORACLE64[name])
    print(f"worst over all maps: oracle32 vs oracle64 {floor:.3f}; policy vs oracle64 {pol64:.3f} (ratio {pol64 / floor:.3f}); policy vs oracle32 {pol32:.3f}")
    assert pol64 <= FLOOR_SLACK * floor, (pol64, floor)
    assert pol32 <= pol64 + floor, (pol32, pol64, floor)       # (triangle inequality: a consistency check of the three tables, not a tolerance)

    # ---- (opt-in) every parameter gradient against the float32 oracle's autograd ---------------------------------------------------------
    if not WITH_GRADS:
        return
    rows = []
    for n, g in grads.items():
        a, b = g.double().flatten(), g32[n].double().flatten()
        rows.append((float(a @ b / (a.norm() * b.norm() + 1e-300)), n, float(a.norm() / (b.norm() + 1e-300))))
    rows.sort()
    print("parameter gradients (%d tensors, full): min cosine %.7f (%s), median %.8f; norm ratio in [%.5f, %.5f]" %
          (len(rows), rows[0][0], rows[0][1], rows[len(rows) // 2][0], min(r for _, _, r in rows), max(r for _, _, r in rows)))
    assert len(rows) == 217
    assert rows[0][0] >= 0.9999 and all(abs(r - 1) <= 2e-3 for _, _, r in rows), rows[:4]


def test_train_step_gradients_at_512_both_backward_policies():
    """The gradient leg, by default: batch KG_GRADLEG_BATCH (default 1) x 512 x 512 with 300 boxes per image (the bench configuration's image
    size and box density at a batch the CPU oracle's autograd finishes in minutes: float32 + float64 autograd take 63 + 136 s per image on the
    box's 32 host cores, and the driver's pytest step ends at 1200 s -- at batch 2, 126 + 273 s, the whole GPU suite measured ~980 s, too close;
    the batch-2 result of this build is in profiles/r06_fullsize_test.txt), EVERY parameter gradient of the default policy `fp32` (single-plane half backward)
    AND of `fp32b2` (hi + lo planes in the backward pass: the reference's precision in both directions) against the reference-pinned oracle's
    autograd in float32 (cosine / norm per tensor) and in float64 (relative L2 per tensor, oracle/gradref.py):
      * both policies: all 217 tensors cosine >= 0.9999 and norm within 2e-3 of the float32 oracle's (train.py:153 is fp32 autograd);
      * `fp32b2`: second-layer heads and decoder within 3x the float32 oracle's own float64 error (group medians), first-layer heads the same
        on the hidden units whose ReLU state equals float64's everywhere (gradref.flipped_units), overall median / p90 / max within 4x;
      * `fp32`: overall within 4x of the float32 oracle's float64 error (the statement of tests/test_gpu_gradprec.py, here at 512 x 512)."""
    from kg_instance_segmentation_amd import KGnet
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    from oracle import gradref, synth, weightgen
    import test_gpu_gradprec as gp
    N, S, NB = int(os.environ.get("KG_GRADLEG_BATCH", "1")), 512, 300
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    sd = weightgen.gen_state_dict(0, variant="cal")
    batch = synth.train_batch(N, S, S, 41, n_boxes=NB, smin=14, smax=40)
    pat64, pat32, patgpu, grads = {}, {}, {}, {}
    for pol in ("fp32", "fp32b2"):
        _, grads[pol] = gp._gpu_grads(sd, pol, batch, pattern=patgpu if pol == "fp32b2" else None, size=S)
    torch.cuda.empty_cache()
    t0 = time.time()
    l32, g32 = gradref.oracle_grads(sd, *batch, S, S, torch.float32, head_hidden=pat32)
    t1 = time.time()
    l64, g64 = gradref.oracle_grads(sd, *batch, S, S, torch.float64, head_hidden=pat64)
    print(f"oracle autograd: float32 {t1 - t0:.0f} s, float64 {time.time() - t1:.0f} s")
    floor = gradref.column(g32, g64, 1e-12)
    floor["groups"] = gradref.by_group(floor["per_tensor"])
    for pol in ("fp32", "fp32b2"):
        rows = []
        for n, g in grads[pol].items():
            if g is None:
                continue
            a, b = g.double().flatten(), g32[n].double().flatten()
            rows.append((float(a @ b / (a.norm() * b.norm() + 1e-300)), n, float(a.norm() / (b.norm() + 1e-300))))
        rows.sort()
        col = gradref.column(grads[pol], g64, 1e-12)
        col["groups"] = gradref.by_group(col["per_tensor"])
        print(f"[{pol}] {len(rows)} tensors vs oracle32: min cosine {rows[0][0]:.7f} ({rows[0][1]}), median {rows[len(rows) // 2][0]:.8f}, norm ratio in "
              f"[{min(r for _, _, r in rows):.5f}, {max(r for _, _, r in rows):.5f}]; vs oracle64 rel. L2 median {col['median']:.2e} p90 {col['p90']:.2e} max {col['max']:.2e} "
              f"(oracle32: {floor['median']:.2e} / {floor['p90']:.2e} / {floor['max']:.2e})", {k: f"{v['median']:.1e}" for k, v in col["groups"].items()})
        assert len(rows) == 217
        assert rows[0][0] >= 0.9999 and all(abs(r - 1) <= 2e-3 for _, _, r in rows), (pol, rows[:4])
        for k in ("median", "p90", "max"):
            assert col[k] <= 4.0 * floor[k], (pol, k, col[k], floor[k])
        if pol == "fp32b2":
            for grp in ("heads .2 (7x7 second layers)", "decoder + c0_conv"):
                assert col["groups"][grp]["median"] <= max(3.0 * floor["groups"][grp]["median"], 2e-6), (grp, col["groups"][grp], floor["groups"][grp])
            fl_p, fl_o = gradref.flipped_units(patgpu, pat64), gradref.flipped_units(pat32, pat64)
            mp = {pre + sfx: gradref.masked_rel_l2(grads[pol][pre + sfx], g64[pre + sfx], ~f) for pre, f in fl_p.items() for sfx in (".weight", ".bias") if bool((~f).any())}
            mo = {pre + sfx: gradref.masked_rel_l2(g32[pre + sfx], g64[pre + sfx], ~f) for pre, f in fl_o.items() for sfx in (".weight", ".bias") if bool((~f).any())}
            med_p, med_o = float(np.median(list(mp.values()))), float(np.median(list(mo.values())))
            print(f"[fp32b2] first-layer heads on the unflipped hidden units: median {med_p:.2e} (oracle32 {med_o:.2e}); channels with a flip: "
                  f"{sum(int(f.sum()) for f in fl_p.values())} (oracle32 {sum(int(f.sum()) for f in fl_o.values())})")
            assert med_p <= max(3.0 * med_o, 2e-6), (med_p, med_o)

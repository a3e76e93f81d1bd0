"""How far are the parameter gradients of a precision policy from the TRUE gradients of the train step (train.py:148-154)?

Yardstick: the reference-pinned CPU oracle evaluated in FLOAT64 (oracle/gradref.py).  Metric: per parameter tensor the relative L2
error ||g - g64|| / ||g64||, summarised as median / p90 / max over the tensors.  The same oracle in float32 -- the reference's own
arithmetic -- is the noise floor: an fp32 implementation cannot be closer to the float64 gradients than that.  (Gradient cosines against
fp32 reference gradients, tests/test_gpu_parity.py, are floor-limited near 1e-5 for every policy and cannot tell a one-plane backward
from a two-plane one; this metric can.)  Calibrated fixture, 2 x 128 x 128, 8 boxes per image -- the shape of the golden train step.

Measured on MI355X (tools/grad_table.py -> profiles/r04_grad_table.json; median / p90 / max):
    oracle float32          5.6e-4 / 1.1e-3 / 1.8e-3      (BatchNorm backbone 7e-4 .. 1.1e-3; decoder 1.4e-5; 7x7 heads 4e-7 .. 6e-7)
    "fp32" (default)        1.5e-3 / 2.6e-3 / 3.5e-3  =  2.6 / 2.4 / 2.0 x the float32 oracle   (heads 1e-4 .. 2.3e-4, decoder 4e-4)
    "fp32b2"                1.3e-3 / 2.2e-3 / 3.1e-3  =  2.3 / 2.0 / 1.7 x                       (heads 3e-7 .. 6e-7 = the floor, decoder 5e-5)
    "fp32" against "fp32b2" 7.7e-4 / 1.4e-3 / 1.8e-3      (identical forward: what the single-plane backward operands add)
Reading: in the BatchNorm backbone (160 of the 217 tensors) every policy, the reference's own fp32 included, is ~1e-3 from the float64
gradient -- the train-mode network amplifies forward rounding -- and the default policy's single half planes in the backward pass add
about as much again.  In the heads and the decoder, where fp32 is accurate to 1e-6, the single-plane backward leaves 1e-4 .. 4e-4
(11-bit operands; mixed-precision grade), and `fp32b2` (hi + lo planes in the backward pass) sits on the fp32 floor.  Stated bounds below
= 2x the measured values (<= 4x the float32 oracle for the default policy's overall columns)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gradref, synth, weightgen  # noqa: E402

DEV = "cuda"
N, S, NB, SEED = 2, 128, 8, 11
HEADS = ("heads .2 (7x7 second layers)", "heads .0 (7x7 first layers)")


def _gpu_grads(sd, policy, batch, pattern=None, size=S):
    """pattern (optional dict): receives the ReLU pattern of the first 7x7 layer of every head, {(level, head): bool [N, C, H, W]}"""
    from kg_instance_segmentation_amd import KGnet, arch, ops
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    x, gt_boxes, gt_masks, gt_lv = batch
    m = KGnet.resnet50(pretrained=False, precision=policy)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.zero_grad()
    m._engine.keep_head_hidden = pattern is not None
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=size, width=size)
    d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
    if pattern is not None:
        for lvl, hid in m._engine.head_hidden.items():
            C = arch.HEAD_CH[lvl]
            Hh, Wh = d0[0].shape[2] >> lvl, d0[0].shape[3] >> lvl
            f = torch.empty(hid.shape[0], 3 * C, dtype=torch.float32, device=DEV)
            ops.planes_to_f32(hid, 3 * C, f)
            pos = (f > 0).view(x.shape[0], Hh, Wh, 3 * C).permute(0, 3, 1, 2).cpu()
            for k, (head, _) in enumerate(arch.HEADS):
                pattern[(lvl, head)] = pos[:, k * C:(k + 1) * C]
        m._engine.head_hidden = {}
    loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)) + lseg(pred, gt_masks, gt_boxes)
    loss.backward()
    torch.cuda.synchronize()
    assert not m.grad_overflowed()
    return float(loss.detach()), {n: (p.grad.detach().double().cpu() if p.grad is not None else None) for n, p in m.named_parameters()}


@pytest.fixture(scope="module")
def table():
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    sd = weightgen.gen_state_dict(0, variant="cal")
    batch = synth.train_batch(N, S, S, SEED, n_boxes=NB)
    pat64, pat32, patgpu = {}, {}, {}
    l64, g64 = gradref.oracle_grads(sd, *batch, S, S, torch.float64, head_hidden=pat64)
    l32, g32 = gradref.oracle_grads(sd, *batch, S, S, torch.float32, head_hidden=pat32)
    out = {"oracle_fp32": gradref.column(g32, g64, 1e-12), "loss64": l64}
    grads = {}
    for p in ("fp32", "fp32b2"):
        lp, grads[p] = _gpu_grads(sd, p, batch, pattern=patgpu if p == "fp32b2" else None)
        out[p] = gradref.column(grads[p], g64, 1e-12)
        out[p]["loss_rel"] = abs(lp - l64) / abs(l64)
        out[p]["groups"] = gradref.by_group(out[p]["per_tensor"])
    out["oracle_fp32"]["groups"] = gradref.by_group(out["oracle_fp32"]["per_tensor"])
    out["fp32_vs_fp32b2"] = gradref.column(grads["fp32"], grads["fp32b2"], 1e-12)
    # First-layer head gradients with the ReLU flips taken out: output channels of which at least one pixel changed its ReLU state against
    # the float64 oracle are excluded from BOTH norms (gradref.flipped_units) -- what is left is rounding
    out["flips"] = {"fp32b2": gradref.flipped_units(patgpu, pat64), "oracle_fp32": gradref.flipped_units(pat32, pat64)}
    out["masked"] = {}
    for col, g in (("fp32b2", grads["fp32b2"]), ("oracle_fp32", g32)):
        errs = {}
        for prefix, fl in out["flips"][col].items():
            for suffix in (".weight", ".bias"):
                n = prefix + suffix
                if g.get(n) is not None and bool((~fl).any()):
                    errs[n] = gradref.masked_rel_l2(g[n], g64[n], ~fl)
        out["masked"][col] = errs
        nfl = {k: int(v.sum()) for k, v in out["flips"][col].items()}
        print(f"[{col}] first-layer head channels with a ReLU flip against float64: {sum(nfl.values())} in {sum(1 for v in nfl.values() if v)} of {len(nfl)} layers; "
              f"masked error median {np.median(list(errs.values())):.2e} max {max(errs.values()):.2e}")
    for c in ("oracle_fp32", "fp32", "fp32b2", "fp32_vs_fp32b2"):
        r = out[c]
        print(f"[{c}] median {r['median']:.2e} p90 {r['p90']:.2e} max {r['max']:.2e} worst {r['worst'][0]}", {k: f"{v['median']:.1e}" for k, v in r.get("groups", {}).items()})
    return out


def test_no_degenerate_tensors_and_losses_agree(table):
    assert table["oracle_fp32"]["n"] >= 213 and not table["oracle_fp32"]["degenerate"]
    assert table["fp32"]["loss_rel"] <= 2e-5 and table["fp32b2"]["loss_rel"] <= 2e-5


def test_default_policy_gradients_within_4x_of_the_fp32_reference_floor(table):
    """median / p90 / max of the per-tensor error of the default policy <= 4 x the same statistic of the float32 oracle (measured 2.6 / 2.4 / 2.0)."""
    o, p = table["oracle_fp32"], table["fp32"]
    for k in ("median", "p90", "max"):
        assert p[k] <= 4.0 * o[k], (k, p[k], o[k])
    assert p["max"] <= 8e-3                                           # (absolute: 2x the measured 3.5e-3)


def test_single_plane_backward_adds_mixed_precision_grade_error(table):
    """the default policy against `fp32b2` (identical forward pass): what 11-bit backward operands add, per tensor"""
    d = table["fp32_vs_fp32b2"]
    assert d["median"] <= 1.6e-3 and d["p90"] <= 3e-3 and d["max"] <= 4e-3, d          # measured 7.7e-4 / 1.4e-3 / 1.8e-3
    g = table["fp32"]["groups"]
    assert all(g[h]["median"] <= 5e-4 for h in HEADS), {h: g[h] for h in HEADS}           # measured 1e-4 / 2.3e-4


def test_two_plane_backward_sits_on_the_fp32_floor_in_heads_and_decoder(table):
    """`fp32b2`: hi + lo half planes in the backward pass -- heads at the float32 oracle's own error (measured 2.9e-7 / 5.9e-7 against
    3.6e-7 / 6.0e-7), decoder within 2e-4 (measured 5e-5 against 1.4e-5), overall within 4 x the float32 oracle (measured 2.3 / 2.0 / 1.7).
    The second layers are asserted through their plain median.  The FIRST layers through the median over the hidden units whose ReLU state
    equals the float64 oracle's at every pixel: a hidden unit whose pre-activation is within rounding of zero passes or blocks a whole pixel's
    gradient, and one such flip against float64 costs a tensor 4e-5 .. 6e-4 -- the float32 oracle shows the same; which tensors are hit depends
    on the last bit of the forward pass.  With exactly those channels excluded (gradref.flipped_units; the same rule for the float32 oracle's
    column) every first-layer tensor is back on the floor -- which is the proof that the raised medians of round 5 were flips, not arithmetic;
    the unmasked values keep a cap."""
    o, p = table["oracle_fp32"], table["fp32b2"]
    h2, h0 = HEADS
    assert p["groups"][h2]["median"] <= max(3.0 * o["groups"][h2]["median"], 2e-6), (p["groups"][h2], o["groups"][h2])
    mo, mp = table["masked"]["oracle_fp32"], table["masked"]["fp32b2"]
    assert len(mp) >= 20, len(mp)                                      # (24 first-layer tensors + their biases; a tensor with every channel flipped is skipped)
    med_o, med_p = float(np.median(list(mo.values()))), float(np.median(list(mp.values())))
    assert med_p <= max(3.0 * med_o, 2e-6), (med_p, med_o)              # the median, restored (round 4's assertion) on the unflipped units
    assert max(mp.values()) <= max(10.0 * max(mo.values()), 2e-5), (max(mp.values()), max(mo.values()))
    for h in HEADS:                                                    # unmasked: caps only (a flip is a legitimate fp32-grade difference)
        assert p["groups"][h]["median"] <= 1e-4 and p["groups"][h]["max"] <= 2e-3, (h, p["groups"][h])
    assert p["groups"]["decoder + c0_conv"]["median"] <= 2e-4
    for k in ("median", "p90", "max"):
        assert p[k] <= 4.0 * o[k], (k, p[k], o[k])

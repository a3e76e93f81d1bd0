"""Parameter gradients of one KGnet train step from the CPU oracle (TEST INFRASTRUCTURE, not product), in float32 or FLOAT64.

The reference's `loss.backward()` (train.py:148-154) is fp32 autograd.  To tell how far a GPU precision policy's gradients are from the
reference's, the yardstick has to be finer than fp32: oracle/net.py (pinned against the reference's own outputs, tests/golden/net_*.npz)
evaluated in float64 is the "true" gradient of the same step, and the same oracle in float32 -- the reference's own arithmetic -- gives the
noise floor an fp32 implementation cannot get under.  Used by tests/test_gpu_gradprec.py and tools/grad_table.py.
"""
import numpy as np
import torch

from . import net as onet


def oracle_grads(sd, x, gt_boxes, gt_masks, gt_lv, H, W, dtype=torch.float64, head_hidden=None):
    """One train step (forward incl. seg branch, 4 detection losses + seg loss, backward: train.py:148-153) of the oracle in `dtype`.
    Returns (loss, {parameter name: gradient tensor (dtype) or None}).  head_hidden (optional dict): receives the ReLU pattern of the first
    7x7 layer of every head, {(level, head): bool [N, C, H, W]} (which hidden units pass their gradient)."""
    sd = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    names = [k for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
    for n in names:
        sd[n].requires_grad_(True)
    net = onet.Net(sd, training=True)
    net.keep_head_hidden = head_hidden is not None
    o0, o1, o2, o3, opred = net.forward(x.to(dtype), gt_boxes)
    if head_hidden is not None:
        head_hidden.update(net.head_hidden)
    loss = sum(onet.detection_loss(p, t.to(dtype)) for p, t in zip((o0, o1, o2, o3), gt_lv))
    l2 = onet.seg_loss(opred, gt_masks, gt_boxes, H, W)
    if l2 is not None:
        loss = loss + l2
    loss.backward()
    return float(loss.detach()), {n: sd[n].grad for n in names}


def rel_l2(got, ref64):
    """per-tensor relative L2 error ||got - ref|| / ||ref|| in float64 (inf for a zero reference with a non-zero result)"""
    a = got.detach().double().cpu().flatten()
    b = ref64.detach().double().cpu().flatten()
    nb = float(b.norm())
    d = float((a - b).norm())
    return d / nb if nb > 0 else (0.0 if d == 0 else float("inf"))


def summarize(errs):
    """errs: {name: relative L2 error}.  median / p90 / max over the tensors + the three worst names"""
    v = np.array(sorted(errs.values()))
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    return {"n": int(v.size), "median": float(np.median(v)), "p90": float(np.quantile(v, 0.9)), "max": float(v.max()),
            "worst": [(n, float(e)) for n, e in worst]}


def column(g, ref, floor):
    errs, degenerate = {}, []
    for n, r in ref.items():
        if r is None or g.get(n) is None:
            continue
        if float(r.double().norm()) / max(r.numel(), 1) ** 0.5 < floor:
            degenerate.append(n)
            continue
        errs[n] = rel_l2(g[n], r)
    s = summarize(errs)
    s["degenerate"] = degenerate
    s["per_tensor"] = errs
    return s


def by_group(errs):
    """median error per group of parameters (where in the network the error lives)"""
    groups = {"heads .2 (7x7 second layers)": lambda n: "_head_c" in n and ".2." in n, "heads .0 (7x7 first layers)": lambda n: "_head_c" in n and ".0." in n,
              "decoder + c0_conv": lambda n: n.startswith(("c0_conv", "c1_up", "c2_up", "c3_up", "c4_up", "c0_cat", "c1_cat", "c2_cat", "c3_cat")),
              "seg branch": lambda n: n.startswith(("skip_combine", "seg_head")), "layer3": lambda n: n.startswith("layer3"),
              "layer2": lambda n: n.startswith("layer2"), "layer1": lambda n: n.startswith("layer1"), "stem": lambda n: n.startswith(("conv1", "bn1"))}
    out = {}
    for gname, f in groups.items():
        v = [e for n, e in errs.items() if f(n)]
        if v:
            out[gname] = {"median": float(np.median(v)), "p25": float(np.quantile(v, 0.25)), "max": float(np.max(v)), "n": len(v)}
    return out


HEAD_NAMES = ("kp", "short_offset", "mid_offset")


def flipped_units(pattern, ref_pattern):
    """{first-layer head parameter prefix "<head>_head_c<l>.0": bool [C]} -- output channels with AT LEAST ONE pixel whose ReLU state differs
    between two forward passes (pattern / ref_pattern: {(level, head): bool [N, C, H, W]}).  The gradient row of such a channel contains
    (or lacks) a whole pixel's contribution: its error against the reference is the size of that contribution, not rounding."""
    out = {}
    for (lvl, head), a in pattern.items():
        b = ref_pattern[(lvl, head)]
        out[f"{head}_head_c{lvl}.0"] = (a != b).flatten(2).any(2).any(0)
    return out


def masked_rel_l2(got, ref64, keep):
    """relative L2 error over the leading-dimension rows `keep` (bool [C]) of a first-layer weight / bias gradient"""
    a = got.detach().double().cpu()[keep].flatten()
    b = ref64.detach().double().cpu()[keep].flatten()
    nb = float(b.norm())
    return float((a - b).norm()) / nb if nb > 0 else 0.0

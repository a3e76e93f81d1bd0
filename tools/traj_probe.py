#!/usr/bin/env python
"""Longer training trajectory of the HIP path against the CPU oracle (fp32, torch.optim.Adam) on the same start: detection and seg loss every 5 steps.
    python tools/traj_probe.py [cal|init] [steps] [size] [boxes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kg_instance_segmentation_amd import KGnet
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.optim import Adam
from kg_instance_segmentation_amd.seg_loss import SEG_loss
from oracle import net as onet, synth, weightgen


def main():
    start = sys.argv[1] if len(sys.argv) > 1 else "cal"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    nbx = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(2, S, S, 23, n_boxes=nbx)
    torch.manual_seed(0)
    m = KGnet.resnet50(pretrained=False)
    if start == "cal":
        m.load_state_dict(weightgen.gen_state_dict(0, variant="cal"))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    opt = Adam([p for p in m.parameters() if p.requires_grad], lr=1.25e-4)
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=S, width=S)
    got = []
    for _ in range(steps):
        opt.zero_grad()
        d0, d1, d2, d3, pred = m(x.cuda(), gt_boxes)
        l1 = sum(ldec(p, t.cuda()) for p, t in zip((d0, d1, d2, d3), gt_lv)); l2 = lseg(pred, gt_masks, gt_boxes)
        (l1 + l2).backward()
        opt.step()
        got.append((float(l1.detach()), float(l2.detach()), m.grad_overflowed()))
    oparams = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
    oopt = torch.optim.Adam(oparams, lr=1.25e-4)
    net = onet.Net(sd, training=True)
    for s in range(steps):
        oopt.zero_grad()
        o0, o1, o2, o3, opred = net.forward(x, gt_boxes)
        l1 = sum(onet.detection_loss(p, t) for p, t in zip((o0, o1, o2, o3), gt_lv)); l2 = onet.seg_loss(opred, gt_masks, gt_boxes, S, S)
        (l1 + l2).backward()
        oopt.step()
        if s % 5 == 0 or s == steps - 1:
            print(f"step {s:3d}  hip det {got[s][0]:10.5f} seg {got[s][1]:9.5f} flag {got[s][2]}   oracle det {float(l1.detach()):10.5f} seg {float(l2.detach()):9.5f}", flush=True)


if __name__ == "__main__":
    main()

"""debug: per-map / per-parameter deviations of a policy against the reference fixture (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kg_instance_segmentation_amd import KGnet
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.seg_loss import SEG_loss
from oracle import synth, weightgen
from tools.pareto import ratios, sub
pol = sys.argv[1]
g = np.load(os.path.join(ROOT, "tests", "golden", "net_cal.npz"), allow_pickle=False)
sd = weightgen.gen_state_dict(0, variant="cal")
m = KGnet.resnet50(pretrained=False, precision=pol); m.load_state_dict(sd); m = m.to("cuda").eval()
m._engine.raw_kp_logits = True; m._seg.keep_logits = True
with torch.no_grad():
    for name in ("a", "b"):
        N, H, W, s = [int(v) for v in g[f"{name}.cfg"]]
        x = (torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(s)) - 0.5).to("cuda")
        d0, d1, d2, d3, feats = m.forward_dec(x)
        for l, f in enumerate(feats):
            print(name, f"feat{l}", ratios(sub(f, 5)[:, ::7], g[f"{name}.eval.feat{l}"]))
        for l, d in enumerate((d0, d1, d2, d3)):
            for nm, t in zip(("kp_logit", "short", "mid"), d):
                print(name, l, nm, ratios(sub(t), g[f"{name}.eval.c{l}.{nm}"]))
N, H, W, s, nb = [int(v) for v in g["train.cfg"]]
x, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, H, W, s, n_boxes=nb)
m.train(); m._engine.raw_kp_logits = False; m.zero_grad()
ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=H, width=W)
d0, d1, d2, d3, pred = m(x.to("cuda"), gt_boxes)
l1 = [ldec(p, t.to("cuda")) for p, t in zip((d0, d1, d2, d3), gt_lv)]
l2 = lseg(pred, gt_masks, gt_boxes)
(sum(l1) + l2).backward(); torch.cuda.synchronize()
params = dict(m.named_parameters())
off, rows = 0, []
for n, nrm in zip([str(n) for n in g["train.grad_names"]], g["train.grad_norm"]):
    gr = params[n].grad.detach().cpu().numpy().ravel().astype(np.float64)
    idx = synth.grad_sample_index(n, gr.size)
    ref = g["train.grad_samples"][off:off + idx.size].astype(np.float64); off += idx.size
    got = gr[idx]
    cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-300))
    rows.append((cos, n, float(np.linalg.norm(gr)) / (float(nrm) + 1e-300), float(nrm)))
print("non-finite grads:", [n for n, p_ in params.items() if p_.grad is not None and not torch.isfinite(p_.grad).all()][:5])
rows.sort()
print("MIN cos %.7f  max norm dev %.2e" % (rows[0][0], max(abs(r[2] - 1) for r in rows)))
for r in rows[:25]:
    print("cos %.6f  %-40s norm ratio %.5f  |g| %.3e" % r)
rows.sort(key=lambda r: -abs(r[2] - 1))
print("--- by norm")
for r in rows[:15]:
    print("cos %.6f  %-40s norm ratio %.5f  |g| %.3e" % r)

#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (read-only) in the build
container.  Runs only where /root/reference exists; the fixtures (data only: inputs
seeds/checksums and expected outputs) are what travels to the GPU box.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens.py [--check-norm]

Shims: a stub `cv2` exposing INTER_NEAREST + nearest `resize` (only
seg_loss.py:77 needs it; rule documented in oracle/net.py:nearest_resize).
Weights come from the build's own generator (oracle/weightgen.py).
"""
import hashlib
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("KG_REFERENCE", "/root/reference")
sys.path.insert(1, REF)

import numpy as np
import torch

from oracle import net as onet, synth, weightgen

cv2 = types.ModuleType("cv2")
cv2.INTER_NEAREST = 0
cv2.resize = lambda a, wh, interpolation=0: onet.nearest_resize(a, wh[1], wh[0])
sys.modules["cv2"] = cv2

import KGnet as rKGnet
import loss as rloss
import nms as rnms
import postprocessing as rpp
import seg_loss as rseg

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def run_postproc(kp, short, mid):
    """Stage-by-stage reference outputs for one scale."""
    kph = np.transpose(kp[0], (1, 2, 0)); sh = np.transpose(short[0], (1, 2, 0)); mh = np.transpose(mid[0], (1, 2, 0))
    heat = rpp.compute_heatmaps(kph, sh)
    from scipy.ndimage import gaussian_filter
    blur = heat.copy()
    for i in range(5):
        blur[:, :, i] = gaussian_filter(blur[:, :, i], sigma=2)
    kps = rpp.get_keypoints(blur, peak_thresh=0.004)
    pk = np.array([[k["id"], k["xy"][0], k["xy"][1]] for k in kps], np.int32).reshape(-1, 3)
    pc = np.array([k["conf"] for k in kps], np.float64)
    sk = rpp.group_skeletons(list(kps), mh, blur)
    sk = np.array(sk, np.float64).reshape(-1, 5, 3)
    ref = rpp.refine_skeleton([s for s in sk])
    ref = np.array(ref, np.float64).reshape(-1, 5, 3)
    # end-to-end call must agree with the staged one
    sk2 = rpp.get_skeletons_and_masks(torch.from_numpy(kp), torch.from_numpy(short), torch.from_numpy(mid))
    assert np.array_equal(np.array(sk2, np.float64).reshape(-1, 5, 3), sk)
    return dict(heat=np.transpose(heat, (2, 0, 1)), blur=np.transpose(blur, (2, 0, 1)), peaks=pk, peak_conf=pc,
                skel=sk, refined=ref)


def adversarial_maps():
    """Hand-made 40x48 case: kp at x==0, kp near the origin, plateau, integer offsets,
    out-of-range votes, degenerate geometry."""
    H, W = 40, 48
    kp, short, mid, _ = synth.head_maps(H, W, 3, 11, sigma_kp=0.0, sigma_off=0.0, smin=12, smax=20)
    kp = kp.copy(); short = short.copy(); mid = mid.copy()
    kp[0, 2, 10:15, 0:3] = 1.0; short[0, 4, 10:15, 0:3] = -np.arange(3)[None, :]  # votes to x==0
    short[0, 5, 10:15, 0:3] = 0.0
    kp[0, 0, 1:4, 1:4] = 0.9; short[0, 0, 1:4, 1:4] = 0.5; short[0, 1, 1:4, 1:4] = 0.5  # near origin, half-int
    kp[0, 4, 30:34, 40:44] = 0.7; short[0, 8, 30:34, 40:44] = 0.0; short[0, 9, 30:34, 40:44] = 0.0  # plateau
    kp[0, 1, 38:40, 46:48] = 1.0; short[0, 2, 38:40, 46:48] = 5.25; short[0, 3, 38:40, 46:48] = 3.5  # out of range
    kp[0, 3, 20, 20] = 1.0; short[0, 6, 20, 20] = -100.0  # far out of range
    return kp, short, mid


def gen_postproc():
    cases = {"s64": (64, 64, 5, 0), "s96x128": (96, 128, 20, 1), "s256": (256, 256, 80, 2)}
    out = {}
    all_sk = {}
    for name, (H, W, n, seed) in cases.items():
        kp, short, mid, _ = synth.head_maps(H, W, n, seed)
        r = run_postproc(kp, short, mid)
        out[f"{name}.cfg"] = np.array([H, W, n, seed], np.int64)
        out[f"{name}.in_sha"] = sha(np.concatenate([kp.ravel(), short.ravel(), mid.ravel()]))
        if H * W <= 64 * 64:
            out[f"{name}.heat"] = r["heat"]; out[f"{name}.blur"] = r["blur"]
        out[f"{name}.heat_sha"] = sha(r["heat"]); out[f"{name}.blur_sha"] = sha(r["blur"])
        for k in ("peaks", "peak_conf", "skel", "refined"):
            out[f"{name}.{k}"] = r[k]
        all_sk[name] = r["refined"]
    kp, short, mid = adversarial_maps()
    r = run_postproc(kp, short, mid)
    out["adv.kp"] = kp; out["adv.short"] = short; out["adv.mid"] = mid
    for k, v in r.items():
        out[f"adv.{k}"] = v
    all_sk["adv"] = r["refined"]
    # boxes / gather / nms: feed the four skeleton sets as the four scales
    sks = [all_sk["s256"], all_sk["s96x128"], all_sk["s64"], all_sk["adv"]]
    for i, sc in enumerate((1, 2, 4, 8)):
        out[f"boxes.scale{sc}"] = np.asarray(rpp.skeleton_to_box([s.copy() for s in sks[i]], sc), np.float64).reshape(-1, 5)
    g = rpp.gather_skeleton(*[[s.copy() for s in k] for k in sks])
    out["gather"] = g
    for th in (0.5, 0.3):
        out[f"nms.{th}"] = rnms.non_maximum_suppression_numpy(g.copy(), th)
    # hand-made skeletons covering all 8 geometric cases + rejected ones (postprocessing.py:178-241)
    rng = np.random.default_rng(5)
    hand = []
    masks = [(1, 1, 1, 1, 1), (1, 1, 1, 1, 0), (1, 1, 1, 0, 1), (1, 1, 0, 1, 0), (1, 0, 1, 1, 1), (0, 1, 1, 1, 0),
             (1, 0, 0, 1, 0), (0, 1, 1, 0, 1), (1, 1, 0, 0, 1), (1, 0, 1, 0, 1), (0, 1, 0, 1, 1), (0, 0, 1, 1, 1),
             (1, 1, 0, 0, 0), (1, 0, 0, 0, 1), (0, 0, 0, 0, 1), (0, 0, 0, 0, 0), (1, 0, 1, 0, 0)]
    for m in masks:
        for rep in range(2):
            y1, x1 = rng.integers(1, 30, 2); h, w = rng.integers(5, 30, 2)
            pts = np.array([[x1, y1], [x1 + w, y1], [x1, y1 + h], [x1 + w, y1 + h], [x1 + w / 2, y1 + h / 2]], np.float64)
            pts += rng.integers(-2, 3, pts.shape)
            s = np.zeros((5, 3)); s[:, :2] = pts; s[:, 2] = rng.random(5) * 0.5 + 0.01
            s *= np.array(m, np.float64)[:, None]
            hand.append(s)
    s = np.zeros((5, 3)); s[0] = [0, 4, .3]; s[1] = [9, 4, .2]; s[2] = [0, 12, .1]; s[3] = [9, 12, .4]; hand.append(s)  # x==0 => missing
    hand = np.array(hand)
    out["hand.skel"] = hand
    href = rpp.refine_skeleton([s for s in hand])
    out["hand.refined"] = np.array(href).reshape(-1, 5, 3)
    out["hand.boxes_all"] = np.asarray(rpp.skeleton_to_box([s.copy() for s in hand], 2), np.float64).reshape(-1, 5)
    out["hand.boxes_refined"] = np.asarray(rpp.skeleton_to_box([s.copy() for s in href], 4), np.float64).reshape(-1, 5)
    hb = out["hand.boxes_all"]
    out["hand.nms"] = rnms.non_maximum_suppression_numpy(hb.copy(), 0.5)
    assert len(np.unique(hb[:, 4])) == len(hb), "golden NMS input must not contain confidence ties"
    assert len(np.unique(g[:, 4])) == len(g)
    assert rnms.non_maximum_suppression_numpy(np.zeros((0, 5)), 0.5) is None
    np.savez_compressed(os.path.join(GOLD, "postproc.npz"), **out)
    print("postproc.npz", {k: v.shape for k, v in out.items() if "skel" in k or "nms" in k})


def sub(t, step=3):
    a = t.detach().numpy()
    return a[..., ::step, ::step].copy() if a.shape[-1] > 32 else a.copy()


def gen_net(seed=0):
    sd = weightgen.gen_state_dict(seed)
    model = rKGnet.resnet50(pretrained=False)
    model.load_state_dict(sd)
    out = {"seed": np.array(seed)}
    cases = {"a": (1, 64, 64, 100), "b": (2, 96, 128, 101)}
    boxes_b = [np.array([[10.2, 12.7, 40.5, 50.5, 1.0], [0.0, 0.0, 95.0, 127.0, 0.9], [30.5, 60.5, 37.5, 71.5, 0.8],
                         [50, 20, 52, 90, 0.7], [64.4, 100.6, 90.2, 126.9, 0.6], [2.5, 3.5, 14.5, 17.5, 0.5]], np.float32),
               np.array([[20, 30, 60, 80, 1.0], [5, 100, 25, 120, 1.0], [70.5, 8.5, 93.5, 40.5, 1.0]], np.float32)]
    for name, (N, H, W, s) in cases.items():
        g = torch.Generator().manual_seed(s)
        x = torch.rand(N, 3, H, W, generator=g) - 0.5
        out[f"{name}.x_sha"] = sha(x.numpy()); out[f"{name}.cfg"] = np.array([N, H, W, s])
        model.eval()
        with torch.no_grad():
            d0, d1, d2, d3, feats = model.forward_dec(x)
            for l, d in enumerate((d0, d1, d2, d3)):
                for nm, t in zip(("kp", "short", "mid"), d):
                    out[f"{name}.eval.c{l}.{nm}"] = sub(t)
            for l, f in enumerate(feats):
                out[f"{name}.eval.feat{l}"] = sub(f, 5)[:, ::7].copy()
            if name == "b":
                patches, dets = model.forward_seg(feats, boxes_b)
                out["b.boxes0"] = boxes_b[0]; out["b.boxes1"] = boxes_b[1]
                for i in range(2):
                    out[f"b.seg.count{i}"] = np.array(len(patches[i]))
                    for j, p in enumerate(patches[i]):
                        out[f"b.seg.{i}.{j}"] = p.numpy(); out[f"b.segdet.{i}.{j}"] = dets[i][j].numpy()
    # one full train step (BN batch stats, losses, backward) on case b-like small input
    model.load_state_dict(sd)
    model.train()
    N, H, W = 2, 64, 96
    g = torch.Generator().manual_seed(202)
    x = torch.rand(N, 3, H, W, generator=g) - 0.5
    gts, gt_boxes, gt_masks = [], [], []
    for i in range(N):
        bx = synth.random_boxes(H, W, 4, 300 + i, 14, 30)
        gt_boxes.append(np.concatenate([bx, np.ones((len(bx), 1))], 1).astype(np.float32))
        m = np.zeros((len(bx), H, W), np.float32)
        for k, b in enumerate(bx.astype(int)):
            yy, xx = np.mgrid[0:H, 0:W]
            cy, cx = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2
            m[k] = (((yy - cy) / ((b[2] - b[0]) / 2 + .5)) ** 2 + ((xx - cx) / ((b[3] - b[1]) / 2 + .5)) ** 2 <= 1).astype(np.float32)
        gt_masks.append(m)
    gt_lv = []
    for sc in (1, 2, 4, 8):
        gt_lv.append(torch.from_numpy(np.stack([synth.gt_maps(np.floor(gt_boxes[i][:, :4] / sc), H // sc, W // sc) for i in range(N)])))
    ldec = rloss.DetectionLossAll(kp_radius=5)
    lseg = rseg.SEG_loss(height=H, width=W)
    pr0, pr1, pr2, pr3, pred = model(x, gt_boxes)
    l1s = [ldec(p, g_) for p, g_ in zip((pr0, pr1, pr2, pr3), gt_lv)]
    l2 = lseg(pred, gt_masks, gt_boxes)
    total = sum(l1s) + l2
    total.backward()
    out["train.cfg"] = np.array([N, H, W, 202]); out["train.x_sha"] = sha(x.numpy())
    out["train.loss_dec"] = np.array([float(v) for v in l1s]); out["train.loss_seg"] = np.array(float(l2))
    out["train.kp0"] = sub(pr0[0]); out["train.mid3"] = pr3[2].detach().numpy()
    out["train.npatch"] = np.array([len(p) for p in pred[0]])
    names, norms, sums = [], [], []
    for k, p in model.named_parameters():
        names.append(k); norms.append(float(p.grad.double().norm())); sums.append(float(p.grad.double().sum()))
    out["train.grad_names"] = np.array(names); out["train.grad_norm"] = np.array(norms); out["train.grad_sum"] = np.array(sums)
    for k in ("kp_head_c0.2.bias", "mid_offset_head_c3.2.bias", "seg_head.2.bias", "bn1.weight", "bn1.bias",
              "layer3.5.bn3.weight", "c0_conv.0.weight", "layer1.0.conv1.weight"):
        out[f"train.grad.{k}"] = dict(model.named_parameters())[k].grad.numpy()
    msd = model.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_mean", "layer3.5.bn3.running_var",
              "layer2.0.downsample.1.running_var", "bn1.num_batches_tracked"):
        out[f"train.stat.{k}"] = msd[k].numpy()
    np.savez_compressed(os.path.join(GOLD, "net.npz"), **out)
    print("net.npz written; losses", out["train.loss_dec"], out["train.loss_seg"], "patches", out["train.npatch"])


def gen_net_cal(seed=0):
    """The calibrated fixture (oracle/weightgen.py variant "cal": unsaturated logits, near-identity residual blocks): eval
    forward incl. the PRE-SIGMOID kp / seg logits (forward hooks on the reference's head modules) and one train step at
    2 x 128 x 128 with a seeded subset of every parameter gradient -> tests/golden/net_cal.npz."""
    sd = weightgen.gen_state_dict(seed, variant="cal")
    model = rKGnet.resnet50(pretrained=False)
    model.load_state_dict(sd)
    cap = {"kp": {}, "seg": []}
    for lvl in range(4):
        getattr(model, f"kp_head_c{lvl}").register_forward_hook(lambda m, i, o, lvl=lvl: cap["kp"].__setitem__(lvl, o.detach().clone()))
    model.seg_head.register_forward_hook(lambda m, i, o: cap["seg"].append(o.detach().clone()))
    out = {"seed": np.array(seed)}
    boxes_b = [np.array([[10.2, 12.7, 40.5, 50.5, 1.0], [0.0, 0.0, 95.0, 127.0, 0.9], [30.5, 60.5, 37.5, 71.5, 0.8],
                         [50, 20, 52, 90, 0.7], [64.4, 100.6, 90.2, 126.9, 0.6], [2.5, 3.5, 14.5, 17.5, 0.5]], np.float32),
               np.array([[20, 30, 60, 80, 1.0], [5, 100, 25, 120, 1.0], [70.5, 8.5, 93.5, 40.5, 1.0]], np.float32)]
    for name, (N, H, W, s) in {"a": (1, 64, 64, 100), "b": (2, 96, 128, 101)}.items():
        x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(s)) - 0.5
        out[f"{name}.x_sha"] = sha(x.numpy()); out[f"{name}.cfg"] = np.array([N, H, W, s])
        model.eval()
        with torch.no_grad():
            d0, d1, d2, d3, feats = model.forward_dec(x)
            for l, d in enumerate((d0, d1, d2, d3)):
                out[f"{name}.eval.c{l}.kp_logit"] = sub(cap["kp"][l])
                assert torch.equal(torch.sigmoid(cap["kp"][l]), d[0])
                out[f"{name}.eval.c{l}.short"] = sub(d[1]); out[f"{name}.eval.c{l}.mid"] = sub(d[2])
            for l, f in enumerate(feats):
                out[f"{name}.eval.feat{l}"] = sub(f, 5)[:, ::7].copy()
            if name == "b":
                cap["seg"].clear()
                patches, dets = model.forward_seg(feats, boxes_b)
                out["b.boxes0"] = boxes_b[0]; out["b.boxes1"] = boxes_b[1]
                k = 0
                for i in range(2):
                    out[f"b.seg.count{i}"] = np.array(len(patches[i]))
                    for j, p in enumerate(patches[i]):
                        z = cap["seg"][k][0, 0]; k += 1
                        assert torch.equal(torch.sigmoid(z), p)
                        out[f"b.seg_logit.{i}.{j}"] = z.numpy()
    model.load_state_dict(sd)
    model.train()
    N, H, W, s = 2, 128, 128, 7
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, H, W, s, n_boxes=6)
    cap["seg"].clear()
    pr0, pr1, pr2, pr3, pred = model(x, gt_boxes)
    ldec = rloss.DetectionLossAll(kp_radius=5); lseg = rseg.SEG_loss(height=H, width=W)
    l1s = [ldec(p, g_) for p, g_ in zip((pr0, pr1, pr2, pr3), gt_lv)]
    l2 = lseg(pred, gt_masks, gt_boxes)
    (sum(l1s) + l2).backward()
    out["train.cfg"] = np.array([N, H, W, s, 6]); out["train.x_sha"] = sha(x.numpy())
    out["train.loss_dec"] = np.array([float(v) for v in l1s]); out["train.loss_seg"] = np.array(float(l2))
    out["train.npatch"] = np.array([len(p) for p in pred[0]])
    for l, d in enumerate((pr0, pr1, pr2, pr3)):
        out[f"train.c{l}.kp_logit"] = sub(cap["kp"][l], 5); out[f"train.c{l}.short"] = sub(d[1], 5); out[f"train.c{l}.mid"] = sub(d[2], 5)
    names, norms, samples = [], [], []
    for k, p in model.named_parameters():
        g = p.grad.numpy().ravel()
        names.append(k); norms.append(float(np.linalg.norm(g.astype(np.float64))))
        samples.append(g[synth.grad_sample_index(k, g.size)])
    out["train.grad_names"] = np.array(names); out["train.grad_norm"] = np.array(norms)
    out["train.grad_samples"] = np.concatenate(samples).astype(np.float32)
    msd = model.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_mean", "layer3.5.bn3.running_var"):
        out[f"train.stat.{k}"] = msd[k].numpy()
    np.savez_compressed(os.path.join(GOLD, "net_cal.npz"), **out)
    print("net_cal.npz written; losses", out["train.loss_dec"], out["train.loss_seg"], "patches", out["train.npatch"],
          "grad samples", out["train.grad_samples"].shape)


def gen_net_layers(seed=3, layers=(1, 2, 2, 1)):
    """A Bottleneck trunk with other block counts than resnet50's (the constructors of KGnet.py:377-410 only differ in them):
    eval forward of ResNet(Bottleneck, [1,2,2,1]) -> tests/golden/net_layers.npz."""
    sd = weightgen.gen_state_dict(seed, layers=layers)
    model = rKGnet.ResNet(rKGnet.Bottleneck, list(layers))
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model.eval()
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(77)) - 0.5
    out = {"layers": np.array(layers), "seed": np.array(seed), "x_sha": sha(x.numpy()), "nkeys": np.array(len(sd))}
    with torch.no_grad():
        d0, d1, d2, d3, feats = model.forward_dec(x)
        onet_out = onet.Net({k: v.clone() for k, v in sd.items()}, training=False, layers=layers).forward_dec(x)
        for l, d in enumerate((d0, d1, d2, d3)):
            for nm, t, o in zip(("kp", "short", "mid"), d, onet_out[l]):
                assert torch.allclose(t, o, rtol=1e-4, atol=1e-5), (l, nm)          # the oracle restatement is pinned here too
                out[f"c{l}.{nm}"] = sub(t)
        for l, f in enumerate(feats):
            out[f"feat{l}"] = sub(f, 5)[:, ::7].copy()
    np.savez_compressed(os.path.join(GOLD, "net_layers.npz"), **out)
    print("net_layers.npz written:", len(sd), "keys")


def gen_loss():
    out = {}
    rng = np.random.default_rng(9)
    N, H, W = 2, 24, 40
    gt = np.stack([synth.gt_maps(synth.random_boxes(H, W, 3, 40 + i, 8, 16), H, W) for i in range(N)])
    kp = rng.random((N, 5, H, W)).astype(np.float32)
    kp[0, 0, :3, :5] = 0.0; kp[0, 1, :3, :5] = 1.0; kp[1, 2, 5:9, 5:9] = 1.0; kp[1, 3, 5:9, 5:9] = 0.0  # clamp path
    gt[1, 2, 5:9, 5:9] = 1.0; gt[0, 1, :3, :5] = 0.0
    short = (gt[:, 5:15] + rng.normal(0, 1.5, (N, 10, H, W))).astype(np.float32)
    mid = (gt[:, 15:55] + rng.normal(0, 3.0, (N, 40, H, W))).astype(np.float32)
    t = [torch.tensor(a, requires_grad=True) for a in (kp, short, mid)]
    l = rloss.DetectionLossAll(kp_radius=5)(t, torch.from_numpy(gt))
    l.backward()
    out.update(kp=kp, short=short, mid=mid, gt=gt, loss=np.array(float(l)), loss32=l.detach().numpy(),
               g_kp=t[0].grad.numpy(), g_short=t[1].grad.numpy(), g_mid=t[2].grad.numpy())
    # empty-mask case (denominator 1e-10)
    gt0 = np.zeros_like(gt)
    l0 = rloss.DetectionLossAll(kp_radius=5)([torch.from_numpy(a) for a in (kp, short, mid)], torch.from_numpy(gt0))
    out["loss_empty"] = np.array(float(l0))
    # SEG loss: 2 images, incl. unmatched patch, resized crop, and the None case
    H2, W2 = 40, 48
    patches = [[torch.tensor(rng.random((10, 12)).astype(np.float32) * 0.98 + 0.01),
                torch.tensor(rng.random((6, 7)).astype(np.float32) * 0.98 + 0.01)],
               [torch.tensor(rng.random((8, 9)).astype(np.float32) * 0.98 + 0.01)]]
    dets = [[torch.tensor([4., 5., 14., 17., 1.]), torch.tensor([20.4, 30.6, 27.5, 38.5, 1.])], [torch.tensor([10., 10., 18., 19., 1.])]]
    gboxes = [np.array([[4, 5, 14, 17, 1], [21, 31, 27, 38, 1], [0, 0, 3, 3, 1]], np.float32), np.array([[30, 30, 38, 39, 1]], np.float32)]
    gmasks = [(rng.random((3, H2, W2)) > 0.5).astype(np.float32), (rng.random((1, H2, W2)) > 0.5).astype(np.float32)]
    ls = rseg.SEG_loss(H2, W2)([patches, dets], gmasks, gboxes)
    out["seg.loss"] = np.array(float(ls))
    for i in range(2):
        for j, p in enumerate(patches[i]):
            out[f"seg.patch.{i}.{j}"] = p.numpy(); out[f"seg.det.{i}.{j}"] = dets[i][j].numpy()
        out[f"seg.gbox.{i}"] = gboxes[i]; out[f"seg.gmask.{i}"] = gmasks[i]
    assert rseg.SEG_loss(H2, W2)([[[patches[1][0]]], [[dets[1][0]]]], [gmasks[1]], [gboxes[1]]) is None
    np.savez_compressed(os.path.join(GOLD, "loss.npz"), **out)
    print("loss.npz", out["loss"], out["loss_empty"], out["seg.loss"])


def kp_boxes(x1, y1, x2, y2):
    """[n,5,2] float32 keypoints (tl, tr, bl, br, centre) as dataset_base.py:58-79 builds them."""
    return np.asarray([[(a, b), (c, b), (a, d), (c, d), (float(a + c) / 2, float(b + d) / 2)] for a, b, c, d in zip(x1, y1, x2, y2)],
                      np.float32).reshape(-1, 5, 2)


def gen_preproc():
    """Ground-truth map generation (preprocessing.get_ground_truth) -- reference outputs in the layout of
    dataset_base.py:99-109 ([55,H,W], cast to float32 as torch.FloatTensor does)."""
    np.int = int                      # preprocessing.py:61 uses the removed alias
    import preprocessing as rprep
    from oracle import preproc as opre
    out = {}
    rng = np.random.default_rng(11)
    cases = {}
    for name, (H, W, n, smin, smax) in {"r48x64": (48, 64, 12, 12, 20), "dense96": (96, 96, 40, 12, 30), "odd33x70": (33, 70, 7, 12, 16)}.items():
        x1 = rng.integers(0, W - smax, n); y1 = rng.integers(0, H - smax, n)
        x2 = x1 + rng.integers(smin, smax, n); y2 = y1 + rng.integers(smin, smax, n)
        cases[name] = (H, W, kp_boxes(x1, y1, x2, y2))
    # adversarial: identical instances (argmin tie -> first), windows overwriting each other's corners, keypoints on the
    # image border and in the corners, half-integer centres
    H, W = 40, 44
    x1 = np.array([0, 0, 10, 13, 30, 20, 20]); y1 = np.array([0, 0, 10, 12, 26, 5, 5])
    x2 = np.array([13, 13, 23, 27, 43, 33, 34]); y2 = np.array([12, 12, 24, 25, 39, 18, 18])
    cases["adv"] = (H, W, kp_boxes(x1, y1, x2, y2))
    cases["empty"] = (24, 24, np.zeros((0, 5, 2), np.float32))
    for name, (H, W, bb) in cases.items():
        kp, sh, md = rprep.get_ground_truth(bb, H, W, 5)
        ref = np.concatenate((kp, np.transpose(sh, (2, 0, 1)), np.transpose(md, (2, 0, 1))), 0)
        ref32 = ref.astype(np.float32)
        assert np.array_equal(ref32.astype(np.float64), ref)            # every value is exactly representable
        assert np.array_equal(opre.ground_truth(bb, H, W), ref), name    # the oracle restatement is pinned here too
        out[f"{name}.bboxes"] = bb
        out[f"{name}.hw"] = np.array([H, W])
        out[f"{name}.gt"] = ref32
    np.savez_compressed(os.path.join(GOLD, "preproc.npz"), **out)
    print("preproc.npz:", {k: v.shape for k, v in out.items() if k.endswith(".gt")})


def eval_case(rng, H, W, ng, nd):
    """GT instance masks (rectangles / ellipses) + detections (perturbed copies, duplicates, false positives)."""
    yy, xx = np.mgrid[0:H, 0:W]
    gm, gb = [], []
    for k in range(ng):
        h, w = rng.integers(10, 24, 2); y1 = rng.integers(0, H - h); x1 = rng.integers(0, W - w)
        if k % 2:
            m = (((yy - (y1 + h / 2)) / (h / 2)) ** 2 + ((xx - (x1 + w / 2)) / (w / 2)) ** 2) <= 1.0
        else:
            m = (yy >= y1) & (yy < y1 + h) & (xx >= x1) & (xx < x1 + w)
        r, c = np.where(m)
        gm.append(m.astype(np.float32)); gb.append([r.min(), c.min(), r.max(), c.max(), 1])
    gm = np.asarray(gm, np.float32); gb = np.asarray(gb, np.float32)
    dm, dd = [], []
    for k in range(nd):
        j = k % ng
        if k < ng + 3:                                  # shifted copy of a GT mask (the extra ones are duplicates)
            sy, sx = rng.integers(-4, 5, 2)
            m = np.roll(np.roll(gm[j], sy, 0), sx, 1)
        else:                                           # false positive
            m = np.zeros((H, W), np.float32); y1 = rng.integers(0, H - 12); x1 = rng.integers(0, W - 12)
            m[y1:y1 + 12, x1:x1 + 12] = 1
        r, c = np.where(m > 0)
        dm.append(m); dd.append([r.min(), c.min(), r.max(), c.max(), 0.99 - 0.031 * k + 0.001 * rng.random()])
    dm.append(np.zeros((H, W), np.float32)); dd.append([3, 3, 9, 9, 0.2])          # empty mask: union < 1 -> IoU 0
    return gm, gb[:, :4], np.asarray(dm, np.float32), np.asarray(dd, np.float32)


def gen_evalparts():
    """Evaluation metrics (eval_parts.py) on a stub dataset object."""
    import eval_parts as rev
    from oracle import evalparts as oev
    rng = np.random.default_rng(21)
    out = {}
    for name, (H, W, ng, nd) in {"a": (96, 96, 10, 16), "b": (64, 120, 6, 9)}.items():
        gm, gb, dm, dd = eval_case(rng, H, W, ng, nd)

        class DS:
            def load_annotation(self, index, type):
                return gm if type == "mask" else gb
        for thr in (0.5, 0.75):
            fp, tp, sc, npos, ovl = rev.seg_evaluation(0, DS(), dm, dd, [], 0, [], thr)
            ofp, otp, osc, oovl = oev.seg_evaluation(gm, gb, dm, dd, thr)
            assert np.array_equal(fp, ofp) and np.array_equal(tp, otp) and np.array_equal(np.asarray(sc), osc) and ovl == oovl and npos == len(gm)
            out[f"{name}.seg{int(thr * 100)}.fp"] = fp; out[f"{name}.seg{int(thr * 100)}.tp"] = tp
            out[f"{name}.seg{int(thr * 100)}.scores"] = np.asarray(sc, np.float32); out[f"{name}.seg{int(thr * 100)}.overlaps"] = np.asarray(ovl, np.float64)
            fp, tp, sc, npos = rev.bbox_evaluation(0, DS(), dd, [], 0, thr)
            ofp, otp, osc = oev.bbox_evaluation(gb, dd, thr)
            assert np.array_equal(fp, ofp) and np.array_equal(tp, otp) and np.array_equal(np.asarray(sc), osc)
            out[f"{name}.box{int(thr * 100)}.fp"] = fp; out[f"{name}.box{int(thr * 100)}.tp"] = tp
        iou = np.array([[rev.mask_iou(a, b) for b in gm] for a in dm], np.float64)
        assert np.array_equal(iou, np.array([[oev.mask_iou(a, b) for b in gm] for a in dm], np.float64))
        out[f"{name}.iou"] = iou
        out[f"{name}.gt_masks"] = gm.astype(np.uint8); out[f"{name}.gt_boxes"] = gb
        out[f"{name}.det_masks"] = dm.astype(np.uint8); out[f"{name}.det"] = dd
    tpc = np.cumsum(out["a.seg50.tp"]); fpc = np.cumsum(out["a.seg50.fp"])
    rec = tpc / 10.0; prec = tpc / np.maximum(tpc + fpc, np.finfo(np.float64).eps)
    out["ap.rec"] = rec; out["ap.prec"] = prec
    out["ap.values"] = np.array([rev.voc_ap(rec, prec, True), rev.voc_ap(rec, prec, False)], np.float64)
    assert out["ap.values"][0] == oev.voc_ap(rec, prec, True) and out["ap.values"][1] == oev.voc_ap(rec, prec, False)
    np.savez_compressed(os.path.join(GOLD, "evalparts.npz"), **out)
    print("evalparts.npz:", len(out), "arrays; AP", out["ap.values"])


def check_norm():
    import ctypes, math
    libm = ctypes.CDLL("libm.so.6"); libm.fma.restype = ctypes.c_double; libm.fma.argtypes = [ctypes.c_double] * 3
    rng = np.random.default_rng(0)
    bad = 0
    for _ in range(200000):
        v = rng.normal(size=2) * 5
        bad += float(np.linalg.norm(v)) != math.sqrt(libm.fma(v[1], v[1], v[0] * v[0]))
    print("np.linalg.norm(2-vector) != sqrt(fma(y,y,x*x)) cases:", bad)


if __name__ == "__main__":
    if "--check-norm" in sys.argv:
        check_norm()
    if "--only-preproc" in sys.argv:
        gen_preproc()
        sys.exit(0)
    if "--only-net-layers" in sys.argv:
        gen_net_layers()
        sys.exit(0)
    if "--only-net-cal" in sys.argv:
        gen_net_cal()
        sys.exit(0)
    if "--only-evalparts" in sys.argv:
        gen_evalparts()
        sys.exit(0)
    torch.manual_seed(0)
    gen_postproc()
    gen_loss()
    gen_net()
    gen_preproc()
    gen_evalparts()
    gen_net_layers()
    gen_net_cal()
    os.system(f"ls -la {GOLD}")

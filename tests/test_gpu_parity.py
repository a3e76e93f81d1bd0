"""GPU floating-point parity of the drop-in KGnet against the reference, asserted ELEMENT-WISE on pre-sigmoid logits.

Fixture: the calibrated weights (oracle/weightgen.py variant "cal": logits O(1), near-identity residual blocks) with goldens
generated from the reference (tests/golden/net_cal.npz, tools/gen_goldens.py:gen_net_cal) -- the raw random-init fixture has
logits of +-500 (saturated sigmoids) and is chaotic in train mode, see test_gpu_net.py.

Stated tolerances |got - ref| <= atol + rtol * |ref|  (rms = root mean square of the reference map), per policy (engine.PRECISIONS):
  "fp32" (DEFAULT: hi + lo IEEE-half planes, 3 MFMA products) and "fp32bf" (hi + mid + lo bf16 planes, 6 products):
            rtol 1e-4, atol 1e-5 -- SURVEY 8d's fp32 tolerance, literally, for the kp / seg logits; atol 3e-5 / 6e-5 for the short / mid
            offset maps (pixels; stated constants), 1e-5 for the feature maps -- in eval AND train mode (measured worst |d| / bound on MI355X: 0.44 / 0.87 for "fp32",
            0.45 / 0.78 for "fp32bf"); every parameter gradient: cosine >= 0.9999, norm within 2e-3 (measured 0.999987 / 6e-4);
  "half"    (single IEEE-half planes, 11 significant bits): SURVEY 8d's reduced-precision clause, rtol 2e-2, with atol 2e-2 * rms
            (measured max |d| = 1.6e-2 rms in train mode, 5e-3 rms in eval mode);
  "halfmix" (as "half", the BatchNorm backbone on hi + lo half planes): rtol 2e-2, atol 1e-2 * rms (measured 5.6e-3 rms);
  "trunk2"  (bf16; whole trunk in hi + lo planes, heads bf16): rtol 2e-2, atol 3e-2 * rms;
  "mixed"   (bf16; BatchNorm backbone in hi + lo planes, c0_conv / decoder / heads plain bf16 = 8 bf16 layers): rtol 2e-2,
            atol 5e-2 * rms -- NOT the blueprint's clause (a plain-bf16 multiply carries 8 bits), kept as an opt-in speed policy;
  "bf16":   rtol 2e-2, atol 1e-1 * rms (train 1.5e-1) -- 60 layers of bf16 storage.
Parameter gradients (train step 2 x 128 x 128): cosine against the reference per parameter >= 0.9999 ("fp32", "fp32bf") / 0.97
("half") / 0.99 ("trunk2", "mixed") / 0.85 ("bf16")."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import KGnet  # noqa: E402
from kg_instance_segmentation_amd.loss import DetectionLossAll  # noqa: E402
from kg_instance_segmentation_amd.seg_loss import SEG_loss  # noqa: E402
from oracle import synth, weightgen  # noqa: E402

DEV = "cuda"
# rtol, atol, atol as a fraction of rms
EVAL_TOL = {"fp32": (1e-4, 1e-5, 0.0), "fp32bf": (1e-4, 1e-5, 0.0), "half": (2e-2, 0.0, 2e-2), "halfmix": (2e-2, 0.0, 1e-2), "trunk2": (2e-2, 0.0, 3e-2), "mixed": (2e-2, 0.0, 6e-2),
            "bf16": (2e-2, 0.0, 1e-1)}
TRAIN_TOL = dict(EVAL_TOL, mixed=(2e-2, 0.0, 5e-2), bf16=(2e-2, 0.0, 1.5e-1))
GRAD_COS = {"fp32": 0.9999, "fp32bf": 0.9999, "half": 0.97, "halfmix": 0.999, "trunk2": 0.99, "mixed": 0.99, "bf16": 0.85}
GRAD_NORM = {"fp32": 2e-3, "fp32bf": 2e-3, "half": 5e-2, "halfmix": 1e-2, "trunk2": 5e-2, "mixed": 5e-2, "bf16": 0.3}
LOSS_TOL = {"fp32": 2e-5, "fp32bf": 2e-5, "half": 2e-3, "halfmix": 1e-3, "trunk2": 2e-3, "mixed": 3e-3, "bf16": 2e-2}
STAT_TOL = {"fp32": (1e-4, 1e-6), "fp32bf": (1e-4, 1e-6), "half": (2e-3, 1e-4), "halfmix": (1e-3, 1e-5), "trunk2": (1e-3, 1e-5), "mixed": (1e-3, 1e-5), "bf16": (3e-2, 3e-3)}
POLICIES = ["fp32", "fp32bf", "half", "halfmix", "trunk2", "mixed", "bf16"]
OFFSET_ATOL_SCALE = {"short": 3.0, "mid": 6.0}      # atol of the offset maps = 1e-5 x these (tools/fullsize_oracle_parity.py uses the same constants)
FEAT_ATOL_SCALE = 1.0                                # feature maps c0 .. c4 (rms 0.04 .. 0.3 on the calibrated fixture): atol 1e-5, literally


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def sub(t, step=3):
    a = t.detach().float().cpu().numpy()
    return a[..., ::step, ::step] if a.shape[-1] > 32 else a


def assert_close(name, got, ref, tol, worst):
    rtol, atol, arms = tol
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    rms = float(np.sqrt(np.mean(ref ** 2)))
    # SURVEY 8d states atol 1e-5 for heatmap / seg LOGITS: taken literally for them.  The offset maps are in pixels (rms 3-4 px short, 5-7 px mid on
    # this fixture): stated per-map constants 3e-5 / 6e-5 (round 5; rounds 2-4 scaled with the map's rms).  The feature maps have no unit scale
    # of their own: the stated constant 1e-5 (round 6; rounds 2-5 wrote 1e-5 * max(1, rms), which was 1e-5 on every fixture anyway).
    literal = "kp_logit" in name or "seg_logit" in name
    scale = 1.0 if literal else OFFSET_ATOL_SCALE["short"] if name.endswith("short") else OFFSET_ATOL_SCALE["mid"] if name.endswith("mid") else FEAT_ATOL_SCALE
    bound = atol * scale + arms * rms + rtol * np.abs(ref)
    ratio = np.abs(got - ref) / bound
    k = int(np.argmax(ratio))
    print(f"[{name}] rms {rms:.3g}  max|d| {float(np.abs(got - ref).max()):.3g}  worst |d|/bound {float(ratio.max()):.3f} (got {got.flat[k]:.6g} ref {ref.flat[k]:.6g})")
    worst.append((float(ratio.max()), name))


@pytest.fixture(scope="module")
def cal_sd():
    return weightgen.gen_state_dict(0, variant="cal")


def make_model(sd, precision):
    m = KGnet.resnet50(pretrained=False, precision=precision)
    m.load_state_dict(sd)
    return m.to(DEV)


def _x(g, name):
    N, H, W, s = [int(v) for v in g[f"{name}.cfg"]]
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(s)) - 0.5
    assert np.array_equal(sha(x.numpy()), g[f"{name}.x_sha"])
    return x.to(DEV)


@pytest.mark.parametrize("precision", POLICIES)
@pytest.mark.parametrize("name", ["a", "b"])
def test_eval_logits_vs_reference(golden, cal_sd, precision, name):
    g = golden("net_cal.npz")
    m = make_model(cal_sd, precision).eval()
    m._engine.raw_kp_logits = True          # the kp maps come back as logits (no sigmoid): compared before saturation
    m._seg.keep_logits = True
    x = _x(g, name)
    worst = []
    with torch.no_grad():
        d0, d1, d2, d3, feats = m.forward_dec(x)
        for l, d in enumerate((d0, d1, d2, d3)):
            for nm, t in zip(("kp_logit", "short", "mid"), d):
                assert_close(f"{precision} {name} c{l}.{nm}", sub(t), g[f"{name}.eval.c{l}.{nm}"], EVAL_TOL[precision], worst)
        for l, f in enumerate(feats):
            assert f.dtype == torch.float32 and f.shape[0] == x.shape[0]      # KGnet.py:318 returns fp32 NCHW features
            assert_close(f"{precision} {name} feat{l}", sub(f, 5)[:, ::7], g[f"{name}.eval.feat{l}"], EVAL_TOL[precision], worst)
        if name == "b":
            pred = m.forward_seg(feats, [g["b.boxes0"], g["b.boxes1"]])
            meta, logits = pred.kg_meta, m._seg.last_logits
            per_img = [[j for j in range(len(meta["off"])) if int(meta["img"][j]) == i] for i in range(2)]
            for i in range(2):
                assert len(pred[0][i]) == len(per_img[i]) == int(g[f"b.seg.count{i}"])
                for jj, j in enumerate(per_img[i]):
                    h, w, off = int(meta["h"][j]), int(meta["w"][j]), int(meta["off"][j])
                    z = logits[off:off + h * w].view(h, w).cpu().numpy()
                    assert_close(f"{precision} seg_logit {i}.{jj}", z, g[f"b.seg_logit.{i}.{jj}"], EVAL_TOL[precision], worst)
    worst.sort(reverse=True)
    print("worst:", worst[:5])
    assert worst[0][0] <= 1.0, worst[:5]


@pytest.mark.parametrize("precision", POLICIES)
def test_train_step_vs_reference(golden, cal_sd, precision):
    """Losses, train-mode maps and EVERY parameter gradient (seeded 1024-element subset) against the reference's train step."""
    g = golden("net_cal.npz")
    N, H, W, s, nb = [int(v) for v in g["train.cfg"]]
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, H, W, s, n_boxes=nb)
    assert np.array_equal(sha(x.numpy()), g["train.x_sha"])
    m = make_model(cal_sd, precision).train()
    m.zero_grad()
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=H, width=W)
    d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
    l1 = [ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)]
    l2 = lseg(pred, gt_masks, gt_boxes)
    ltol = LOSS_TOL[precision]
    print("loss_dec", [float(v) for v in l1], "ref", g["train.loss_dec"], "loss_seg", float(l2), float(g["train.loss_seg"]))
    np.testing.assert_allclose([float(v) for v in l1], g["train.loss_dec"], rtol=ltol)
    assert abs(float(l2) - float(g["train.loss_seg"])) <= ltol * abs(float(g["train.loss_seg"]))
    assert [len(p) for p in pred[0]] == list(g["train.npatch"])
    worst = []
    for l, d in enumerate((d0, d1, d2, d3)):
        p = d[0].detach().double().clamp(1e-300, 1 - 1e-16)
        # train-mode kp maps are probabilities; their logits are recovered in fp64 (exact to ~1e-6 at |z| < 6)
        assert_close(f"{precision} train c{l}.kp_logit", sub(torch.log(p / (1 - p)), 5), g[f"train.c{l}.kp_logit"],
                     tuple(np.add(TRAIN_TOL[precision], (0, 2e-5, 0))), worst)
        assert_close(f"{precision} train c{l}.short", sub(d[1], 5), g[f"train.c{l}.short"], TRAIN_TOL[precision], worst)
        assert_close(f"{precision} train c{l}.mid", sub(d[2], 5), g[f"train.c{l}.mid"], TRAIN_TOL[precision], worst)
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:5]
    (sum(l1) + l2).backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    names = [str(n) for n in g["train.grad_names"]]
    off, rows = 0, []
    for n, nrm in zip(names, g["train.grad_norm"]):
        gr = params[n].grad.detach().cpu().numpy().ravel().astype(np.float64)
        idx = synth.grad_sample_index(n, gr.size)
        ref = g["train.grad_samples"][off:off + idx.size].astype(np.float64); off += idx.size
        got = gr[idx]
        cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-300))
        rows.append((cos, n, float(np.linalg.norm(gr)) / (float(nrm) + 1e-300)))
    rows.sort()
    print(f"[{precision}] per-parameter gradient cosine vs the reference: min {rows[0][0]:.6f} p10 {rows[len(rows) // 10][0]:.6f} median {rows[len(rows) // 2][0]:.6f}")
    print("   worst:", [(round(c, 5), n, round(r, 4)) for c, n, r in rows[:8]])
    ratios = np.array([r for _, _, r in rows])
    print("   norm ratio: min %.4f max %.4f" % (ratios.min(), ratios.max()))
    assert rows[0][0] >= GRAD_COS[precision], rows[:5]
    assert np.all(np.abs(ratios - 1) <= GRAD_NORM[precision])
    sd = m.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_mean", "layer3.5.bn3.running_var"):
        np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"train.stat.{k}"], rtol=STAT_TOL[precision][0], atol=STAT_TOL[precision][1])


@pytest.mark.parametrize("precision", ["fp32", "half", "mixed"])
def test_full_size_forward_vs_oracle(cal_sd, precision):
    """512 x 512 (BASELINE's size), batch 1: forward_dec + forward_seg against the reference-pinned CPU oracle (oracle/net.py),
    element-wise on kp logits / offsets / seg logits."""
    from oracle import net as onet
    m = make_model(cal_sd, precision).eval()
    m._engine.raw_kp_logits = True
    m._seg.keep_logits = True
    x = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(512)) - 0.5
    bx = synth.random_boxes(512, 512, 12, 77)
    boxes = [np.concatenate([bx, np.ones((len(bx), 1))], 1).astype(np.float32)]
    with torch.no_grad():
        d0, d1, d2, d3, feats = m.forward_dec(x.to(DEV))
        pred = m.forward_seg(feats, boxes)
        net = onet.Net({k: v.clone() for k, v in cal_sd.items()}, training=False)
        o0, o1, o2, o3, ofe = net.forward_dec(x)
        net.forward_seg(ofe, boxes)
    worst = []
    for l, (d, o) in enumerate(zip((d0, d1, d2, d3), (o0, o1, o2, o3))):
        assert_close(f"{precision} 512 c{l}.kp_logit", d[0].cpu().numpy(), net.kp_logits[l].numpy(), EVAL_TOL[precision], worst)
        assert_close(f"{precision} 512 c{l}.short", d[1].cpu().numpy(), o[1].numpy(), EVAL_TOL[precision], worst)
        assert_close(f"{precision} 512 c{l}.mid", d[2].cpu().numpy(), o[2].numpy(), EVAL_TOL[precision], worst)
    meta, logits = pred.kg_meta, m._seg.last_logits
    assert len(meta["off"]) == len(net.seg_logits[0]) == 12
    for j in range(12):
        h, w, off = int(meta["h"][j]), int(meta["w"][j]), int(meta["off"][j])
        assert_close(f"{precision} 512 seg_logit {j}", logits[off:off + h * w].view(h, w).cpu().numpy(), net.seg_logits[0][j].numpy(), EVAL_TOL[precision], worst)
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:5]


def test_training_step_with_boxes_is_deterministic(cal_sd):
    """A full training step at 512 x 512 with 300 (overlapping) boxes per image, twice: every parameter gradient and the loss are
    bit-identical (the crop-gradient reduction of the seg branch is a fixed-order gather, there are no floating-point atomics)."""
    N, H, W = 2, 512, 512
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, H, W, 3, n_boxes=300, smin=14, smax=40)
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=H, width=W)
    m = make_model(cal_sd, "fp32").train()
    runs = []
    for _ in range(2):
        m.load_state_dict(cal_sd)
        m.zero_grad()
        d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
        loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)) + lseg(pred, gt_masks, gt_boxes)
        loss.backward()
        runs.append((float(loss), {n: p.grad.clone() for n, p in m.named_parameters()}))
    assert runs[0][0] == runs[1][0]
    bad = [n for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[1][1][n])]
    assert not bad, bad[:8]
    assert len(runs[0][1]) == 217


def test_backward_of_a_stale_forward_is_refused(cal_sd):
    """The engine keeps the activations of the latest forward_dec only: backward through an older forward must raise, not
    silently use the newer activations."""
    m = make_model(cal_sd, "fp32").train()
    x = torch.rand(1, 3, 64, 64, device=DEV) - 0.5
    a = m.forward_dec(x)[0][1].sum()
    b = m.forward_dec(x)[0][1].sum()
    with pytest.raises(RuntimeError, match="LATEST forward_dec"):
        a.backward()
    b.backward()


def test_config3_batch16_full_path(cal_sd):
    """BASELINE configs[2]: the full HIP path at batch 16, 512 x 512.  (a) inference: the post-processing + NMS of the product on its
    OWN head tensors equals the C oracle's on the same tensors bit for bit (4 of the 16 images), and the forward is equivariant
    under a permutation of the batch (size-independent property: eval-mode BatchNorm couples no images); (b) one train step with
    300 boxes per image yields finite gradients for all 217 parameters, and the loss does not depend on the order of the images
    in the batch (all loss terms and the train-mode BatchNorm statistics are symmetric in the images)."""
    from kg_instance_segmentation_amd import postprocessing as kpp
    from oracle import postproc as op
    N, S = 16, 512
    m = make_model(cal_sd, "fp32").eval()
    x = (torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(16)) - 0.5).to(DEV)
    with torch.no_grad():
        dec = m.forward_dec(x)[:4]
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))
        dec_p = m.forward_dec(x[perm.to(DEV)])[:4]
    for d, dp in zip(dec, dec_p):
        for t, tp in zip(d, dp):
            assert t.shape[0] == N and torch.isfinite(t).all()
            assert torch.equal(t[perm.to(DEV)], tp)
    for i in (0, 5, 10, 15):
        di = [[t[i:i + 1].contiguous() for t in d] for d in dec]
        got = kpp.detect(di, 0.5)
        ref = op.detect([[t.cpu().numpy() for t in d] for d in di], 0.5)
        assert (got is None) == (ref is None)
        if ref is not None:
            print(f"image {i}: {len(ref)} boxes after NMS")
            assert got.shape == ref.shape and np.array_equal(got, ref)
    del dec, dec_p
    m.train()
    xs, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, S, S, 8, n_boxes=300, smin=14, smax=40)
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=S, width=S)
    losses = []
    for order in (list(range(N)), list(reversed(range(N)))):
        m.load_state_dict(cal_sd)
        m.zero_grad()
        d0, d1, d2, d3, pred = m(xs[order].to(DEV), [gt_boxes[i] for i in order])
        loss = sum(ldec(p, t[order].to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)) + lseg(pred, [gt_masks[i] for i in order], [gt_boxes[i] for i in order])
        loss.backward()
        losses.append(float(loss))
        grads = [p.grad for p in m.parameters()]
        assert len(grads) == 217 and all(g is not None and torch.isfinite(g).all() for g in grads)
    print("bs16 losses (two image orders):", losses)
    assert abs(losses[0] - losses[1]) <= 1e-4 * abs(losses[0])


def test_train_step_256_vs_oracle_full_gradients(cal_sd):
    """Default policy, 2 x 256 x 256 with 40 boxes per image (the golden train step is 2 x 128 x 128 with sampled gradients): losses and
    the FULL gradient of every parameter against the reference-pinned CPU oracle's autograd (oracle/net.py) -- cosine >= 0.9999 and norm
    within 2e-3 per tensor."""
    from oracle import net as onet
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    N, S = 2, 256
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, S, S, 17, n_boxes=40, smin=14, smax=40)
    m = make_model(cal_sd, "fp32").train()
    m.zero_grad()
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=S, width=S)
    d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
    loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)) + lseg(pred, gt_masks, gt_boxes)
    loss.backward()
    torch.cuda.synchronize()
    assert not m.grad_overflowed()
    sd = {k: v.clone() for k, v in cal_sd.items()}
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for n in names:
        sd[n].requires_grad_(True)
    net = onet.Net(sd, training=True)
    o0, o1, o2, o3, opred = net.forward(x, gt_boxes)
    oloss = sum(onet.detection_loss(p, t) for p, t in zip((o0, o1, o2, o3), gt_lv)) + onet.seg_loss(opred, gt_masks, gt_boxes, S, S)
    oloss.backward()
    assert abs(float(loss) - float(oloss)) <= 2e-5 * abs(float(oloss))
    rows = []
    for n, p in m.named_parameters():
        if p.grad is None:
            assert sd[n].grad is None or float(sd[n].grad.abs().max()) == 0.0, n
            continue
        a, b = p.grad.detach().double().cpu().flatten(), sd[n].grad.double().flatten()
        rows.append((float(a @ b / (a.norm() * b.norm() + 1e-300)), n, float(a.norm() / (b.norm() + 1e-300))))
    rows.sort()
    print("min cosine %.7f (%s); norm ratio in [%.5f, %.5f]" % (rows[0][0], rows[0][1], min(r for _, _, r in rows), max(r for _, _, r in rows)))
    assert rows[0][0] >= 0.9999 and all(abs(r - 1) <= 2e-3 for _, _, r in rows), rows[:4]


def test_training_trajectory_vs_oracle_fp32(cal_sd):
    """Eight optimizer steps (fused HIP Adam, lr 1e-4) of the default policy on the calibrated fixture against the same steps of the
    reference-pinned CPU oracle in fp32 with torch.optim.Adam: the loss of EVERY step within 1e-3 (the first within 2e-5) -- the
    half-precision backward (gradient scales, re-normalisation points) does not make the trajectory drift."""
    from kg_instance_segmentation_amd.optim import Adam
    from oracle import net as onet
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    N, S, steps = 2, 128, 8
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, S, S, 23, n_boxes=8)
    m = make_model(cal_sd, "fp32").train()
    opt = Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=S, width=S)
    got = []
    for _ in range(steps):
        opt.zero_grad()
        d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
        loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)) + lseg(pred, gt_masks, gt_boxes)
        loss.backward()
        opt.step()
        got.append(float(loss.detach()))
    assert not m.grad_overflowed()
    osd = {k: v.clone() for k, v in cal_sd.items()}
    oparams = [v.requires_grad_(True) for k, v in osd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
    oopt = torch.optim.Adam(oparams, lr=1e-4)
    net = onet.Net(osd, training=True)
    ref = []
    for _ in range(steps):
        oopt.zero_grad()
        o0, o1, o2, o3, opred = net.forward(x, gt_boxes)
        oloss = sum(onet.detection_loss(p, t) for p, t in zip((o0, o1, o2, o3), gt_lv)) + onet.seg_loss(opred, gt_masks, gt_boxes, S, S)
        oloss.backward()
        oopt.step()
        ref.append(float(oloss.detach()))
    print("loss per step: hip", [round(v, 5) for v in got], "oracle fp32", [round(v, 5) for v in ref])
    assert abs(got[0] - ref[0]) <= 2e-5 * abs(ref[0])
    assert all(abs(a - b) <= 1e-3 * abs(b) for a, b in zip(got, ref)), (got, ref)
    assert got[-1] < got[0]

"""GPU parity tests of the drop-in KGnet module (forward_dec / forward_seg / train step) against the
golden fixtures generated from the reference.

This file uses the RAW random-init fixture (kaiming weights, logits of order +-500: sigmoids saturated, and in train mode a
perturbation grows ~x1.2 per layer through the 43 batch-statistics BatchNorm layers).  Element-wise parity on unsaturated
logits is asserted in test_gpu_parity.py on the calibrated fixture; here, for the default "mixed" precision (trunk in hi + lo
bf16 planes, heads bf16):
  * short/mid maps: relative L2 error <= 3e-2 per map; kp probabilities: |dp| > 0.05 on <= 3 % of the pixels (probabilities
    are 0/1 except on sign changes of the logit, where a 1 % logit error flips the pixel)
  * seg probabilities: max-abs <= 5e-2
  * train step: losses 1e-2 relative; EVERY parameter gradient against the fp32 oracle / reference: cosine >= 0.98, norm
    within 10 % (with plain bf16 storage of the trunk -- precision "bf16" -- the same comparison gives cosines of 0.2-0.6 in
    layer1/2 for ANY implementation, which is why the trunk is kept in two planes)
(bit-exactness is only claimed for the integer/float64 post-processing on identical head tensors)."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import KGnet  # noqa: E402
from kg_instance_segmentation_amd.loss import DetectionLossAll  # noqa: E402
from kg_instance_segmentation_amd.seg_loss import SEG_loss  # noqa: E402

DEV = "cuda"


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def sub(t, step=3):
    a = t.detach().float().cpu().numpy()
    return a[..., ::step, ::step] if a.shape[-1] > 32 else a


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def model(state_dict0):
    m = KGnet.resnet50(pretrained=False)
    m.load_state_dict(state_dict0)
    return m.to(DEV)


def _x(g, name):
    N, H, W, s = [int(v) for v in g[f"{name}.cfg"]]
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(s)) - 0.5
    assert np.array_equal(sha(x.numpy()), g[f"{name}.x_sha"])
    return x.to(DEV)


def test_state_dict_roundtrip(model, state_dict0):
    sd = model.state_dict()
    assert list(sd.keys()) == list(state_dict0.keys()) and len(sd) == 346
    for k, v in state_dict0.items():
        assert torch.equal(sd[k].cpu(), v), k


@pytest.mark.parametrize("name", ["a", "b"])
def test_forward_dec_eval(golden, model, state_dict0, name):
    g = golden("net.npz")
    model.load_state_dict(state_dict0)
    model.eval()
    x = _x(g, name)
    with torch.no_grad():
        d0, d1, d2, d3, feats = model.forward_dec(x)
    torch.cuda.synchronize()
    worst, worst_kp, worst_kp_logit = 0.0, 0.0, 0.0
    for l, d in enumerate((d0, d1, d2, d3)):
        for nm, t in zip(("kp", "short", "mid"), d):
            ref = g[f"{name}.eval.c{l}.{nm}"]
            got = sub(t)
            assert got.shape == ref.shape, (got.shape, ref.shape)
            if nm == "kp":   # compare the pre-sigmoid logits where the fp32 probability is not saturated
                ok = (ref > 1e-6) & (ref < 1 - 1e-6) & (got > 1e-6) & (got < 1 - 1e-6)
                zr, zg = np.log(ref[ok] / (1 - ref[ok])), np.log(got[ok] / (1 - got[ok]))
                e = float(np.linalg.norm(zg - zr) / np.linalg.norm(zr))
                flips = float(np.mean(np.abs(got - ref) > 0.05))
                print(f"[{name} c{l}.kp] logit rel_l2={e:.4f} over {int(ok.sum())}/{ok.size} unsaturated px; |dp|>0.05 on {100 * flips:.2f}% px")
                worst_kp = max(worst_kp, flips)
                worst_kp_logit = max(worst_kp_logit, e)
            else:
                e = rel_l2(got, ref)
                print(f"[{name} c{l}.{nm}] rel_l2={e:.2e} max_abs={float(np.abs(got - ref).max()):.4f} ref_absmax={float(np.abs(ref).max()):.3f}")
                worst = max(worst, e)
    for l, f in enumerate(feats):
        ref = g[f"{name}.eval.feat{l}"]
        got = sub(f, 5)[:, ::7]
        print(f"[{name} feat{l}] rel_l2={rel_l2(got, ref):.2e}")
        assert rel_l2(got, ref) <= 2e-5
    # default policy (fp32-tolerance forward on hi + lo half planes) on the reference's random init, logits +-500: measured offset maps
    # <= 5e-6 relative L2, unsaturated kp logits <= 3e-4 (a few dozen pixels), no pixel with |dp| > 0.05; the bounds are 2x that
    # (round 2's `mixed` policy needed 3e-2 / 3 % here)
    print(f"worst: offsets {worst:.2e}, kp logits {worst_kp_logit:.2e}, flipped pixels {worst_kp:.4f}")
    assert worst <= 1e-5 and worst_kp == 0.0 and worst_kp_logit <= 6e-4
    if name == "b":
        boxes = [g["b.boxes0"], g["b.boxes1"]]
        with torch.no_grad():
            patches, dets = model.forward_seg(feats, boxes)
        for i in range(2):
            assert len(patches[i]) == int(g[f"b.seg.count{i}"])
            for j, p in enumerate(patches[i]):
                ref = g[f"b.seg.{i}.{j}"]
                assert tuple(p.shape) == ref.shape
                mx = float(np.abs(p.cpu().numpy() - ref).max())
                print(f"[seg {i}.{j}] shape {ref.shape} max_abs={mx:.4f}")
                assert mx <= 5e-2
                assert np.array_equal(dets[i][j].numpy(), g[f"b.segdet.{i}.{j}"])


def test_train_step_matches_golden(golden, model, state_dict0):
    from oracle import synth
    g = golden("net.npz")
    model.load_state_dict(state_dict0)
    model.train()
    model.zero_grad()
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(*[int(v) for v in g["train.cfg"]])
    H, W = x.shape[2:]
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=H, width=W)
    d0, d1, d2, d3, pred = model(x.to(DEV), gt_boxes)
    l1 = [ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)]
    l2 = lseg(pred, gt_masks, gt_boxes)
    print("loss_dec", [float(v) for v in l1], "ref", g["train.loss_dec"], "loss_seg", float(l2), float(g["train.loss_seg"]))
    assert [len(p) for p in pred[0]] == list(g["train.npatch"])
    np.testing.assert_allclose([float(v) for v in l1], g["train.loss_dec"], rtol=2e-5)
    assert abs(float(l2) - float(g["train.loss_seg"])) <= 2e-5 * abs(float(g["train.loss_seg"]))
    (sum(l1) + l2).backward()
    torch.cuda.synchronize()
    names = [str(n) for n in g["train.grad_names"]]
    params = dict(model.named_parameters())
    missing = [n for n in names if params[n].grad is None]
    assert not missing, missing[:5]
    norms = np.array([float(params[n].grad.double().norm()) for n in names])
    ratio = norms / (g["train.grad_norm"] + 1e-12)
    worst = np.argsort(-np.abs(np.log(ratio + 1e-12)))[:8]
    print("grad norm ratio: median %.4f min %.4f max %.4f" % (np.median(ratio), ratio.min(), ratio.max()))
    print("worst:", [(names[i], float(ratio[i])) for i in worst])
    # every parameter gradient against the fp32 CPU oracle's autograd on the same batch (the oracle is pinned to the reference's
    # gradients by tests/test_oracle_net.py) and the stored reference gradients themselves
    from oracle import net as onet
    osd = {k: v.clone() for k, v in state_dict0.items()}
    for n in names:
        osd[n].requires_grad_(True)
    onet_ = onet.Net(osd, training=True)
    o0, o1, o2, o3, opred = onet_.forward(x, gt_boxes)
    ol = sum(onet.detection_loss(p, t) for p, t in zip((o0, o1, o2, o3), gt_lv)) + onet.seg_loss(opred, gt_masks, gt_boxes, H, W)
    ol.backward()
    rows = []
    for n in names:
        ref = osd[n].grad.numpy().ravel().astype(np.float64); got = params[n].grad.cpu().numpy().ravel().astype(np.float64)
        cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
        rows.append((cos, n, rel_l2(got, ref), float(np.linalg.norm(ref))))
    rows.sort()
    print("per-parameter gradient parity vs fp32 (worst 30 by cosine):")
    for cos, n, e, nr in rows[:30]:
        print(f"   cos={cos:.5f} rel_l2={e:.4f} |ref|={nr:.3e}  {n}")
    print("   cosine quantiles: min %.5f p10 %.5f median %.5f" % (rows[0][0], rows[len(rows) // 10][0], rows[len(rows) // 2][0]))
    for k in ("kp_head_c0.2.bias", "mid_offset_head_c3.2.bias", "seg_head.2.bias", "bn1.weight", "bn1.bias", "layer3.5.bn3.weight",
              "c0_conv.0.weight", "layer1.0.conv1.weight"):
        ref = g[f"train.grad.{k}"].ravel().astype(np.float64); got = params[k].grad.cpu().numpy().ravel().astype(np.float64)
        cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
        print(f"[grad {k} vs the reference's] cos={cos:.5f} rel_l2={rel_l2(got, ref):.4f}")
        assert cos >= 0.9992, k
    # Raw random init in train mode is chaotic (the fp32 reference itself is 26 % from a float64 evaluation of the same step on this
    # fixture: profiles/r04_grad_table.json), so these bounds measure agreement of two fp32-grade forward passes, not backward precision
    # (tests/test_gpu_gradprec.py does that on the calibrated fixture).  Measured with the default policy: min cosine 0.99964 (relative L2
    # 2.7e-2 on bn1.bias), norms within 0.43 %; asserted at 2x (round 2's `mixed` bounds here were 0.98 / 10 %).
    assert rows[0][0] >= 0.9992, rows[:5]
    assert max(r[2] for r in rows) <= 5.5e-2, max(rows, key=lambda r: r[2])
    assert np.all(np.abs(ratio - 1) <= 1e-2), ("gradient norms off vs the fp32 reference", float(ratio.min()), float(ratio.max()))
    sd = model.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_mean", "layer3.5.bn3.running_var", "layer2.0.downsample.1.running_var"):
        np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"train.stat.{k}"], rtol=5e-3, atol=5e-4)
    assert int(sd["bn1.num_batches_tracked"]) == 1


def test_optimizer_step_runs(model, state_dict0):
    """Adam over the fp32 master weights (train.py:71) followed by a second forward with repacked weights."""
    model.load_state_dict(state_dict0)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    x = torch.rand(2, 3, 64, 64, device=DEV) - 0.5
    boxes = [np.array([[8, 8, 40, 44, 1]], np.float32), np.array([[10, 20, 50, 60, 1]], np.float32)]
    losses = []
    for _ in range(2):
        opt.zero_grad()
        d0, d1, d2, d3, pred = model(x, boxes)
        loss = sum(t.float().pow(2).mean() for d in (d0, d1, d2, d3) for t in d) + sum(p.mean() for pp in pred[0] for p in pp)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    print("losses", losses)
    assert all(np.isfinite(losses))


def test_weight_updates_without_version_bump_are_seen(model, state_dict0):
    """The bf16 packed copies must follow the fp32 master weights even when the optimizer does not bump tensor versions
    (fused multi-tensor Adam, writes through `.data`): fused and default Adam must give the same loss trajectory, and
    inference right after training must use the updated weights."""
    x = torch.rand(2, 3, 64, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)) - 0.5
    traj = {}
    for fused in (False, True):
        model.load_state_dict(state_dict0)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=fused)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            d0, d1, d2, d3, _ = model(x, [np.zeros((0, 5), np.float32)] * 2)
            loss = sum(t.float().pow(2).mean() for d in (d0, d1, d2, d3) for t in d[1:])
            loss.backward()
            opt.step()
            losses.append(float(loss))
        traj[fused] = losses
    print("foreach", traj[False], "fused", traj[True])
    assert abs(traj[False][1] - traj[False][0]) > 1e-4 * abs(traj[False][0])          # the update is visible at all
    for a, b in zip(traj[False], traj[True]):
        assert abs(a - b) <= 2e-2 * abs(a)
    model.eval()
    with torch.no_grad():
        a = model.forward_dec(x)[0][1].clone()
        for p in model.parameters():
            p.data.mul_(0.5)                       # no version bump
        model.train(); model(x, [np.zeros((0, 5), np.float32)] * 2); model.eval()   # a training forward in between
        b = model.forward_dec(x)[0][1]
    assert float((a - b).abs().max()) > 1e-3 * float(a.abs().max())


def test_full_size_determinism_and_batch_permutation(model, state_dict0):
    """BASELINE's full size (512x512): properties that do not need the oracle.  Inference is bit-reproducible and
    equivariant under a permutation of the batch (no cross-image coupling in eval mode: BN uses running statistics);
    a training step without boxes (no atomics involved) gives bit-identical parameter gradients twice."""
    model.load_state_dict(state_dict0)
    model.eval()
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.rand(2, 3, 512, 512, device=DEV, generator=g) - 0.5
    with torch.no_grad():
        a = [t.clone() for d in model.forward_dec(x)[:4] for t in d]
        b = [t.clone() for d in model.forward_dec(x)[:4] for t in d]
        c = [t.clone() for d in model.forward_dec(x.flip(0))[:4] for t in d]
    for u, v, w in zip(a, b, c):
        assert u.shape[-1] in (512, 256, 128, 64) and torch.isfinite(u).all()
        assert torch.equal(u, v)
        assert torch.equal(u, w.flip(0))
    model.train()
    grads = []
    for _ in range(2):
        model.load_state_dict(state_dict0)          # resets the BN running statistics too
        model.zero_grad()
        d0, d1, d2, d3, _ = model(x, [np.zeros((0, 5), np.float32)] * 2)
        sum(t.float().abs().mean() for d in (d0, d1, d2, d3) for t in d).backward()
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 150
    bad = [n for n in grads[0] if not torch.equal(grads[0][n], grads[1][n])]
    assert not bad, bad[:5]


@pytest.mark.parametrize("shape", [(1, 200, 136), (3, 72, 264)])
def test_odd_sizes_vs_oracle(model, state_dict0, shape):
    """Image sizes that are multiples of 8 but of none of the kernels' tile sizes (16 / 32 / 64 pixels), batch sizes 1 and 3: eval
    forward + seg branch against the (reference-pinned) oracle network."""
    from oracle import net as onet
    N, H, W = shape
    model.load_state_dict(state_dict0)
    model.eval()
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(H + W)) - 0.5
    boxes = [np.array([[4.4, 6.6, 0.55 * H, 0.6 * W, 1.0], [0.3 * H, 0.2 * W, H - 1.0, W - 1.0, 0.9]], np.float32) for _ in range(N)]
    with torch.no_grad():
        d0, d1, d2, d3, feats = model.forward_dec(x.to(DEV))
        patches, dets = model.forward_seg(feats, boxes)
        net = onet.Net({k: v.clone() for k, v in state_dict0.items()}, training=False)
        o0, o1, o2, o3, ofe = net.forward_dec(x)
        opatches, odets = net.forward_seg(ofe, boxes)
    for l, (d, o) in enumerate(zip((d0, d1, d2, d3), (o0, o1, o2, o3))):
        for nm, t, r in zip(("short", "mid"), d[1:], o[1:]):
            e = rel_l2(t.cpu().numpy(), r.numpy())
            print(f"[{shape} c{l}.{nm}] rel_l2={e:.4f}")
            assert tuple(t.shape) == tuple(r.shape) and e <= 3e-2
    for i in range(N):
        assert len(patches[i]) == len(opatches[i]) == 2
        for p, r in zip(patches[i], opatches[i]):
            diff = (p.cpu() - r).abs()
            print(f"[{shape} seg img {i} {tuple(p.shape)}] max_abs={float(diff.max()):.4f} mean_abs={float(diff.mean()):.5f} frac>0.05={float((diff > 0.05).float().mean()):.5f}")
            # (random-init logits are saturated: a probability only moves where a logit changes sign, cf. the kp criterion above)
            assert tuple(p.shape) == tuple(r.shape) and float((diff > 0.05).float().mean()) <= 5e-3 and float(diff.mean()) <= 2e-3


def test_other_block_counts(golden):
    """Bottleneck trunks with other block counts (resnet101 / resnet152 constructors, KGnet.py:388-409): state_dict keys and the
    eval forward of ResNet(Bottleneck, [1,2,2,1]) against the reference fixture."""
    from oracle import weightgen
    g = golden("net_layers.npz")
    layers = [int(v) for v in g["layers"]]
    sd = weightgen.gen_state_dict(int(g["seed"]), layers=layers)
    m = KGnet.ResNet(None, layers)
    assert list(m.state_dict().keys()) == list(sd.keys()) and len(sd) == int(g["nkeys"])
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = (torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(77)) - 0.5).to(DEV)
    with torch.no_grad():
        d0, d1, d2, d3, feats = m.forward_dec(x)
    for l, d in enumerate((d0, d1, d2, d3)):
        for nm, t in zip(("short", "mid"), d[1:]):
            e = rel_l2(sub(t), g[f"c{l}.{nm}"])
            print(f"[layers{layers} c{l}.{nm}] rel_l2={e:.4f}")
            assert e <= 3e-2
    for l, f in enumerate(feats):
        assert rel_l2(sub(f, 5)[:, ::7], g[f"feat{l}"]) <= 3e-2
    assert len(KGnet.resnet101(pretrained=False).state_dict()) == 346 + 17 * 18      # 17 more bottlenecks in layer3
    assert len(KGnet.resnet152(pretrained=False).state_dict()) == 346 + (4 + 30) * 18


def test_native_library_is_the_one_loaded():
    import os
    maps = open(f"/proc/{os.getpid()}/maps").read()
    assert "libkgnet_hip.so" in maps


def test_fused_hip_adam_matches_torch_adam():
    """kg_adam_step (one launch for all tensors) vs torch.optim.Adam over 3 steps incl. odd sizes, unaligned views and a changed lr."""
    from kg_instance_segmentation_amd.optim import Adam
    torch.manual_seed(5)
    shapes = [(64, 3, 7, 7), (5,), (1023,), (256, 64, 3, 3), (1,), (4097,)]
    pa = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    sched_a = torch.optim.lr_scheduler.ExponentialLR(oa, gamma=0.96)          # train.py:72
    sched_b = torch.optim.lr_scheduler.ExponentialLR(ob, gamma=0.96)
    big = torch.randn(8192, device=DEV)
    for it in range(3):
        for k, (a, b) in enumerate(zip(pa, pb)):
            g = torch.randn_like(a) * (10.0 ** (k - 2))
            if a.numel() == 1023:
                g = big[1:1024].clone().view_as(a)
                a.grad = big[1:1024].view_as(a)          # a gradient that is an unaligned view of a larger buffer
                b.grad = g
                continue
            a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step(); sched_a.step(); sched_b.step()
    for a, b in zip(pa, pb):
        err = float((a - b).abs().max()), float(b.abs().max())
        assert err[0] <= 2e-6 * max(err[1], 1.0), err
    sa, sb = oa.state[pa[0]], ob.state[pb[0]]
    assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-8) and torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-10)


def test_training_trajectory_vs_oracle(state_dict0):
    """Four Adam steps (lr 1e-4, the reference's optimizer, train.py:71) on one seeded 64 x 64 batch against the oracle network
    trained on the CPU with torch.optim.Adam, in fp32 and as bf16-STORAGE emulation (oracle/net_bf16.py: same fp32 arithmetic,
    tensors rounded where the HIP path stores bf16).  Step 0 (identical weights) is within the stated 5 % of fp32; afterwards the
    three curves fall together (HIP 104.8 / 104.9 / 78.2 / 54.7, fp32 103.8 / 109.6 / 84.8 / 59.9, emulation 105.2 / 109.0 / 80.1 /
    59.2 on one host and 105.1 / 108.0 / 79.3 / 55.9 on another: the random-init fixture is chaotic under perturbations of the size
    of a bf16 rounding or a different CPU reduction order), so the bound on the later steps is 12 %."""
    from oracle import net as onet, net_bf16, synth
    from kg_instance_segmentation_amd.optim import Adam
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(1, 64, 64, 5, n_boxes=3)
    model = KGnet.resnet50(pretrained=False)
    model.load_state_dict(state_dict0)
    model = model.to(DEV).train()
    opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    ldec, lseg = DetectionLossAll(5), SEG_loss(64, 64)
    got = []
    for it in range(4):
        opt.zero_grad()
        d0, d1, d2, d3, pred = model(x.to(DEV), gt_boxes)
        loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv))
        l2 = lseg(pred, gt_masks, gt_boxes)
        loss = loss if l2 is None else loss + l2
        loss.backward()
        opt.step()
        got.append(float(loss.detach()))

    def oracle_run(cls):
        osd = {k: v.clone() for k, v in state_dict0.items()}
        oparams = [v.requires_grad_(True) for k, v in osd.items() if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
        oopt = torch.optim.Adam(oparams, lr=1e-4)
        net = cls(osd, training=True)
        out = []
        for it in range(4):
            oopt.zero_grad()
            o0, o1, o2, o3, opred = net.forward(x, gt_boxes)
            oloss = sum(onet.detection_loss(p, t) for p, t in zip((o0, o1, o2, o3), gt_lv))
            ol2 = onet.seg_loss(opred, gt_masks, gt_boxes, 64, 64)
            oloss = oloss if ol2 is None else oloss + ol2
            oloss.backward()
            oopt.step()
            out.append(float(oloss.detach()))
        return out

    ref32, ref16 = oracle_run(onet.Net), oracle_run(net_bf16.NetBF16)
    print("loss per step: hip", [round(v, 3) for v in got], "oracle fp32", [round(v, 3) for v in ref32], "oracle bf16-storage", [round(v, 3) for v in ref16])
    assert abs(got[0] - ref32[0]) <= 5e-2 * abs(ref32[0]), (got, ref32)
    for a, b, c in zip(got, ref32, ref16):
        assert abs(a - c) <= 0.12 * abs(c), (got, ref16)
        assert abs(a - b) <= 0.12 * abs(b), (got, ref32)
    assert got[-1] < 0.7 * got[0] and ref32[-1] < 0.7 * ref32[0]


def test_inference_sees_an_optimizer_step_without_a_training_forward(state_dict0):
    """eval forward -> this package's fused Adam step (raw-pointer writes: no tensor version bump) -> eval forward: the packed bf16
    weight copies must follow (ops.PARAM_EPOCH); the same after model.invalidate_caches() for writes nothing can see."""
    from kg_instance_segmentation_amd.optim import Adam
    m = KGnet.resnet50(pretrained=False)
    m.load_state_dict(state_dict0)
    m = m.to(DEV)
    x = torch.rand(1, 3, 64, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) - 0.5
    m.train()
    d0 = m.forward_dec(x)[0]
    sum(t.float().abs().mean() for t in d0[1:]).backward()
    opt = Adam([p for p in m.parameters() if p.grad is not None], lr=1e-3)     # (1e-2 on every weight of the raw random-init network drives its
    m.eval()                                                                    #  activations past 65504, the documented range of the half format)
    with torch.no_grad():
        a = m.forward_dec(x)[0][1].clone()
        opt.step()                                   # writes the parameters through raw pointers
        b = m.forward_dec(x)[0][1].clone()
        assert float((a - b).abs().max()) > 1e-4 * float(a.abs().max())
        for p in m.parameters():
            p.data.mul_(0.5)                         # invisible to every counter ...
        m.invalidate_caches()                        # ... so the caller says so
        c = m.forward_dec(x)[0][1]
        assert float((c - b).abs().max()) > 1e-4 * float(b.abs().max())


def test_train_step_leaves_no_reference_cycles(state_dict0):
    """A train step must be freed by reference counting alone: with Python's cyclic collector disabled the allocated GPU memory
    does not grow from step to step (a cycle autograd node -> saved activations -> output tensor -> grad_fn kept a whole step's
    activations alive until a gen-2 collection: tens of GB and a multi-ms host stall at a random step)."""
    import gc
    from oracle import synth
    from kg_instance_segmentation_amd.optim import Adam
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(2, 96, 96, 5, n_boxes=6)
    m = KGnet.resnet50(pretrained=False)
    m.load_state_dict(state_dict0)
    m = m.to(DEV).train()
    opt = Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    ldec, lseg = DetectionLossAll(5), SEG_loss(96, 96)
    xd, gtd = x.to(DEV), [t.to(DEV) for t in gt_lv]

    def step():
        opt.zero_grad()
        d0, d1, d2, d3, pred = m(xd, gt_boxes)
        loss = sum(ldec(p, t) for p, t in zip((d0, d1, d2, d3), gtd))
        l2 = lseg(pred, gt_masks, gt_boxes)
        (loss if l2 is None else loss + l2).backward()
        opt.step()

    for _ in range(3):
        step()
    gc.collect()
    gc.disable()
    try:
        torch.cuda.synchronize()
        m0 = torch.cuda.memory_allocated()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        m1 = torch.cuda.memory_allocated()
    finally:
        gc.enable()
    print(f"allocated {m0 >> 20} -> {m1 >> 20} MiB over 4 steps without the cyclic collector")
    assert m1 - m0 < (8 << 20), (m0, m1)


def test_half_range_violation_is_loud(state_dict0):
    """The default policy stores packed weights times 2^12 in IEEE half (|w| < 16) and activations as they are (|x| <= 65504).  A network
    outside that range must not be clamped silently: the outputs / loss turn non-finite and the sticky gradient flag is raised."""
    from oracle import synth, weightgen
    m = KGnet.resnet50(pretrained=False)
    m.load_state_dict(weightgen.gen_state_dict(0, variant="cal"))
    m = m.to(DEV).train()
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(1, 64, 64, 5, n_boxes=3)
    ldec = DetectionLossAll(5)

    def step():
        m.zero_grad()
        d0, d1, d2, d3, _ = m(x.to(DEV), gt_boxes)
        loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv))
        loss.backward()
        return float(loss)
    assert np.isfinite(step()) and not m.grad_overflowed()
    with torch.no_grad():
        m.get_tensor("c3_cat_refine.0.weight")[0, 0, 0, 0] = 64.0          # beyond the half format's weight range
    bad = step()
    assert not np.isfinite(bad)
    assert m.grad_overflowed() and not m.grad_overflowed()                   # sticky until read, then reset
    with pytest.raises(ValueError, match="c3_cat_refine.0.weight"):          # the explicit range check names the tensor ...
        m.check_half_range()
    # load_state_dict loads any checkpoint like the reference (KGnet.py:384-385): such weights switch the model to the bf16-plane sibling
    m2 = KGnet.resnet50(pretrained=False)
    with pytest.warns(RuntimeWarning, match="fp32bf"):
        m2.load_state_dict(m.state_dict())
    assert m2.precision == "fp32bf"
    assert m2.precision_switch[:3] == ("fp32", "fp32bf", "c3_cat_refine.0.weight")      # the switch is recorded where a training script can see it
    m2 = m2.to(DEV).eval()
    with torch.no_grad():
        assert all(torch.isfinite(t).all() for d in m2.forward_dec(x.to(DEV))[:4] for t in d)      # bf16 planes carry such a weight
    m3 = KGnet.resnet50(pretrained=False, precision="fp32bf")
    m3.load_state_dict(m.state_dict())
    assert m3.precision == "fp32bf" and m3.precision_switch is None
    # a non-finite weight is a broken checkpoint, not a range question: it raises in every case
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["c3_cat_refine.0.weight"][0, 0, 0, 0] = float("nan")
    with pytest.raises(ValueError, match="non-finite"):
        KGnet.resnet50(pretrained=False).load_state_dict(sd)


def test_resnet101_random_init_gradients_stay_in_half_range():
    """[3, 4, 23] blocks (KGnet.resnet101): at random init the gradient grows ~2^0.9 per bottleneck through the 23-block layer3; the
    backward pass re-normalises it at every block output (engine.renormalise), so all 366 parameter gradients are finite and
    the overflow flag stays down."""
    torch.manual_seed(3)
    m = KGnet.resnet101(pretrained=False).to(DEV).train()
    from oracle import synth
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(2, 64, 64, 9, n_boxes=3)
    d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
    ldec, lseg = DetectionLossAll(5), SEG_loss(64, 64)
    loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv))
    l2 = lseg(pred, gt_masks, gt_boxes)
    (loss + l2 if l2 is not None else loss).backward()
    torch.cuda.synchronize()
    assert not m.grad_overflowed()
    got = [n for n, p in m.named_parameters() if p.grad is not None]
    assert len(got) > 300 and all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)
    assert float(m.get_tensor("conv1.weight").grad.abs().max()) > 0.0


def test_prepacked_weights_give_the_same_training_trajectory():
    """optim.Adam(..., prepack=model) packs the next forward's 16-bit weight copies right behind the update kernel (engine.Engine.prepack).
    Three optimizer steps with and without it are bit-identical (losses and final parameters), and a version-bumping in-place edit of a
    parameter between the optimizer step and the next forward is seen (the pre-packed copy is dropped, not used)."""
    from kg_instance_segmentation_amd.optim import Adam
    from oracle import synth, weightgen
    sd = weightgen.gen_state_dict(0, variant="cal")
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(2, 64, 64, 5, n_boxes=3)
    ldec, lseg = DetectionLossAll(5), SEG_loss(64, 64)

    def run(prepack, edit):
        m = KGnet.resnet50(pretrained=False)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        opt = Adam(m.parameters(), lr=1e-4, prepack=m if prepack else None)
        losses = []
        for it in range(3):
            opt.zero_grad()
            d0, d1, d2, d3, pred = m(x.to(DEV), gt_boxes)
            loss = sum(ldec(p, t.to(DEV)) for p, t in zip((d0, d1, d2, d3), gt_lv)) + lseg(pred, gt_masks, gt_boxes)
            loss.backward()
            opt.step()
            assert (m._engine.prepack_state is not None) == prepack
            if edit and it == 1:
                with torch.no_grad():
                    m.get_tensor("kp_head_c0.2.weight").mul_(1.5)          # in-place: bumps the tensor version
                    m.get_tensor("seg_head.0.weight").mul_(0.5)
            losses.append(float(loss))
        return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}

    for edit in (False, True):
        la, pa = run(False, edit)
        lb, pb = run(True, edit)
        assert la == lb, (edit, la, lb)
        bad = [k for k in pa if not torch.equal(pa[k], pb[k])]
        assert not bad, (edit, bad[:5])
    assert run(False, False)[0][2] != run(False, True)[0][2]          # (the edit does change the third step: the comparison above is not vacuous)

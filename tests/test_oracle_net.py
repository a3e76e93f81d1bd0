"""CPU: the torch-functional oracle (oracle/net.py) reproduces the reference network,
seg branch and losses on the fixtures generated from the reference."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import net as onet
from oracle import synth


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def sub(t, step=3):
    a = t.detach().numpy()
    return a[..., ::step, ::step] if a.shape[-1] > 32 else a


def _x(g, name):
    N, H, W, s = [int(v) for v in g[f"{name}.cfg"]]
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(s)) - 0.5
    assert np.array_equal(sha(x.numpy()), g[f"{name}.x_sha"])
    return x


@pytest.mark.parametrize("name", ["a", "b"])
def test_forward_dec_eval(golden, state_dict0, name):
    g = golden("net.npz")
    x = _x(g, name)
    net = onet.Net({k: v.clone() for k, v in state_dict0.items()}, training=False)
    with torch.no_grad():
        d0, d1, d2, d3, feats = net.forward_dec(x)
        for l, d in enumerate((d0, d1, d2, d3)):
            for nm, t in zip(("kp", "short", "mid"), d):
                np.testing.assert_allclose(sub(t), g[f"{name}.eval.c{l}.{nm}"], rtol=1e-5, atol=1e-5)
        for l, f in enumerate(feats):
            np.testing.assert_allclose(sub(f, 5)[:, ::7], g[f"{name}.eval.feat{l}"], rtol=1e-5, atol=1e-5)
        if name == "b":
            boxes = [g["b.boxes0"], g["b.boxes1"]]
            patches, dets = net.forward_seg(feats, boxes)
            for i in range(2):
                assert len(patches[i]) == int(g[f"b.seg.count{i}"])
                for j, p in enumerate(patches[i]):
                    assert p.shape == g[f"b.seg.{i}.{j}"].shape
                    np.testing.assert_allclose(p.numpy(), g[f"b.seg.{i}.{j}"], rtol=1e-5, atol=1e-6)
                    assert np.array_equal(dets[i][j].numpy(), g[f"b.segdet.{i}.{j}"])


def test_detection_loss_and_grads(golden):
    g = golden("loss.npz")
    t = [torch.tensor(g[k], requires_grad=True) for k in ("kp", "short", "mid")]
    l = onet.detection_loss(t, torch.from_numpy(g["gt"]))
    assert abs(float(l) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    l.backward()
    for tt, k in zip(t, ("g_kp", "g_short", "g_mid")):
        np.testing.assert_allclose(tt.grad.numpy(), g[k], rtol=1e-5, atol=1e-9)
    l0 = onet.detection_loss([x.detach() for x in t], torch.zeros_like(torch.from_numpy(g["gt"])))
    assert abs(float(l0) - float(g["loss_empty"])) <= 1e-6


def test_seg_loss(golden):
    g = golden("loss.npz")
    patches = [[torch.from_numpy(g[f"seg.patch.{i}.{j}"]) for j in range(n)] for i, n in ((0, 2), (1, 1))]
    dets = [[torch.from_numpy(g[f"seg.det.{i}.{j}"]) for j in range(n)] for i, n in ((0, 2), (1, 1))]
    gm = [g["seg.gmask.0"], g["seg.gmask.1"]]
    gb = [g["seg.gbox.0"], g["seg.gbox.1"]]
    l = onet.seg_loss([patches, dets], gm, gb, 40, 48)
    assert abs(float(l) - float(g["seg.loss"])) <= 1e-6
    assert onet.seg_loss([[[patches[1][0]]], [[dets[1][0]]]], [gm[1]], [gb[1]], 40, 48) is None


def train_inputs(g):
    N, H, W, s = [int(v) for v in g["train.cfg"]]
    return synth.train_batch(N, H, W, s)


def test_train_step(golden, state_dict0):
    g = golden("net.npz")
    x, gt_boxes, gt_masks, gt_lv = train_inputs(g)
    assert np.array_equal(sha(x.numpy()), g["train.x_sha"])
    sd = {k: v.clone() for k, v in state_dict0.items()}
    names = [str(n) for n in g["train.grad_names"]]
    for n in names:
        sd[n].requires_grad_(True)
    net = onet.Net(sd, training=True)
    d0, d1, d2, d3, pred = net.forward(x, gt_boxes)
    l1 = [onet.detection_loss(p, t) for p, t in zip((d0, d1, d2, d3), gt_lv)]
    l2 = onet.seg_loss(pred, gt_masks, gt_boxes, x.shape[2], x.shape[3])
    np.testing.assert_allclose([float(v) for v in l1], g["train.loss_dec"], rtol=2e-5)
    assert abs(float(l2) - float(g["train.loss_seg"])) <= 2e-5 * abs(float(g["train.loss_seg"]))
    assert [len(p) for p in pred[0]] == list(g["train.npatch"])
    (sum(l1) + l2).backward()
    norms = np.array([float(sd[n].grad.double().norm()) for n in names])
    np.testing.assert_allclose(norms, g["train.grad_norm"], rtol=2e-3, atol=1e-7)
    for k in ("kp_head_c0.2.bias", "mid_offset_head_c3.2.bias", "seg_head.2.bias", "bn1.weight", "c0_conv.0.weight"):
        ref = g[f"train.grad.{k}"]
        np.testing.assert_allclose(sd[k].grad.numpy(), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_mean", "layer3.5.bn3.running_var",
              "layer2.0.downsample.1.running_var"):
        np.testing.assert_allclose(sd[k].detach().numpy(), g[f"train.stat.{k}"], rtol=1e-4, atol=1e-6)
    assert int(sd["bn1.num_batches_tracked"]) == int(g["train.stat.bn1.num_batches_tracked"])


def test_other_block_counts_match_reference(golden):
    """ResNet(Bottleneck, [1,2,2,1]) (the constructors of KGnet.py:377-410 differ only in the block counts): the oracle with
    `layers` reproduces the reference's eval forward."""
    import torch
    from oracle import net as onet, weightgen
    g = golden("net_layers.npz")
    layers = tuple(int(v) for v in g["layers"])
    sd = weightgen.gen_state_dict(int(g["seed"]), layers=layers)
    assert len(sd) == int(g["nkeys"])
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(77)) - 0.5
    with torch.no_grad():
        outs = onet.Net(sd, training=False, layers=layers).forward_dec(x)
    for l in range(4):
        for nm, t in zip(("kp", "short", "mid"), outs[l]):
            a = t.numpy(); a = a[..., ::3, ::3] if a.shape[-1] > 32 else a
            np.testing.assert_allclose(a, g[f"c{l}.{nm}"], rtol=1e-4, atol=1e-5)


def _cal_sd():
    from oracle import weightgen
    return weightgen.gen_state_dict(0, variant="cal")


@pytest.mark.parametrize("name", ["a", "b"])
def test_cal_fixture_eval_logits(golden, name):
    """Calibrated fixture (unsaturated logits): the oracle reproduces the reference's PRE-SIGMOID kp / seg logits, offsets and
    features at the fp32 tolerance SURVEY 8d states (rtol 1e-4, atol 1e-5)."""
    g = golden("net_cal.npz")
    x = _x(g, name)
    net = onet.Net(_cal_sd(), training=False)
    with torch.no_grad():
        d0, d1, d2, d3, feats = net.forward_dec(x)
        for l, d in enumerate((d0, d1, d2, d3)):
            np.testing.assert_allclose(sub(net.kp_logits[l]), g[f"{name}.eval.c{l}.kp_logit"], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(sub(d[1]), g[f"{name}.eval.c{l}.short"], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(sub(d[2]), g[f"{name}.eval.c{l}.mid"], rtol=1e-4, atol=1e-5)
            assert 0.2 < float(net.kp_logits[l].pow(2).mean().sqrt()) < 5.0          # the fixture is unsaturated
        for l, f in enumerate(feats):
            np.testing.assert_allclose(sub(f, 5)[:, ::7], g[f"{name}.eval.feat{l}"], rtol=1e-4, atol=1e-5)
        if name == "b":
            net.forward_seg(feats, [g["b.boxes0"], g["b.boxes1"]])
            for i in range(2):
                assert len(net.seg_logits[i]) == int(g[f"b.seg.count{i}"])
                for j, z in enumerate(net.seg_logits[i]):
                    np.testing.assert_allclose(z.numpy(), g[f"b.seg_logit.{i}.{j}"], rtol=1e-4, atol=1e-5)


def test_cal_fixture_train_step(golden):
    """One train step of the calibrated fixture at 2 x 128 x 128: losses, train-mode logits and a seeded subset of EVERY parameter
    gradient against the reference (the fixture is well conditioned, so fp32-vs-fp32 agreement is tight)."""
    g = golden("net_cal.npz")
    N, H, W, s, nb = [int(v) for v in g["train.cfg"]]
    x, gt_boxes, gt_masks, gt_lv = synth.train_batch(N, H, W, s, n_boxes=nb)
    assert np.array_equal(sha(x.numpy()), g["train.x_sha"])
    sd = _cal_sd()
    names = [str(n) for n in g["train.grad_names"]]
    for n in names:
        sd[n].requires_grad_(True)
    net = onet.Net(sd, training=True)
    d0, d1, d2, d3, pred = net.forward(x, gt_boxes)
    l1 = [onet.detection_loss(p, t) for p, t in zip((d0, d1, d2, d3), gt_lv)]
    l2 = onet.seg_loss(pred, gt_masks, gt_boxes, H, W)
    np.testing.assert_allclose([float(v) for v in l1], g["train.loss_dec"], rtol=2e-5)
    assert abs(float(l2) - float(g["train.loss_seg"])) <= 2e-5 * abs(float(g["train.loss_seg"]))
    for l, d in enumerate((d0, d1, d2, d3)):
        np.testing.assert_allclose(sub(net.kp_logits[l].detach(), 5), g[f"train.c{l}.kp_logit"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(sub(d[2].detach(), 5), g[f"train.c{l}.mid"], rtol=1e-3, atol=1e-4)
    (sum(l1) + l2).backward()
    off = 0
    for n, nrm in zip(names, g["train.grad_norm"]):
        gr = sd[n].grad.numpy().ravel()
        idx = synth.grad_sample_index(n, gr.size)
        ref = g["train.grad_samples"][off:off + idx.size]; off += idx.size
        got = gr[idx]
        cos = float(got.astype(np.float64) @ ref.astype(np.float64) / (np.linalg.norm(got.astype(np.float64)) * np.linalg.norm(ref.astype(np.float64)) + 1e-300))
        assert cos >= 0.9999, (n, cos)
        assert abs(float(np.linalg.norm(gr.astype(np.float64))) - nrm) <= 2e-3 * nrm + 1e-9, n
    assert off == g["train.grad_samples"].size

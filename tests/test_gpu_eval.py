"""GPU: BASELINE configs[4] (multi-scale eval 256 / 512 / 1024, ~300 instances) beyond the post-processing parity of
test_gpu_postproc.py: the MASKS of the inference path (test.py:88-157: forward_dec -> forward_seg on the detected boxes -> paste-back ->
threshold) against the reference-pinned CPU oracle on the same image and boxes -- mask IoU (eval_parts.mask_iou semantics,
eval_parts.py:4-9) -- and the 1024 x 1024 network forward element-wise against the oracle at the fp32 tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import KGnet, postprocessing as kpp  # noqa: E402
from oracle import net as onet, paste as opaste, weightgen  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def cal_sd():
    return weightgen.gen_state_dict(0, variant="cal")


@pytest.fixture(scope="module")
def model(cal_sd):
    m = KGnet.resnet50(pretrained=False)
    m.load_state_dict(cal_sd)
    return m.to(DEV).eval()


@pytest.mark.parametrize("S", [256, 512])
def test_mask_iou_vs_oracle(model, cal_sd, S):
    import bench
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    dec_np, _ = bench.eval_inputs(S, 300, 5)
    det = kpp.detect([[torch.from_numpy(a).to(DEV) for a in d] for d in dec_np])
    assert det is not None and len(det) >= 100
    bb = det.astype(np.float32)[:120]                     # (bounded: the oracle's seg branch is a Python loop over boxes)
    x = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(S)) - 0.5
    with torch.no_grad():
        feats = model.forward_dec(x.to(DEV))[4]
        pred = model.forward_seg(feats, [bb])
        got = kpp.paste_masks(pred, S, S, S, S, 0.5)
        net = onet.Net({k: v.clone() for k, v in cal_sd.items()}, training=False)
        of = net.forward_dec(x)[4]
        op_ = net.forward_seg(of, [bb])
    ref = opaste.paste_masks([[[p.numpy() for p in pp] for pp in op_[0]], [[np.asarray(d) for d in dd] for dd in op_[1]]], S, S, S, S, 0.5)
    assert got is not None and ref is not None and got[0].shape == ref[0].shape
    assert np.array_equal(got[1], ref[1])                                  # the detections travel unchanged
    a, b = got[0] > 0, ref[0] > 0
    inter = (a & b).reshape(len(a), -1).sum(1).astype(np.float64)
    union = (a | b).reshape(len(a), -1).sum(1).astype(np.float64)
    iou = inter / np.maximum(union, 1.0)                                    # eval_parts.mask_iou (eval_parts.py:4-9)
    print(f"S={S}: {len(a)} masks, IoU vs oracle mean {iou.mean():.6f} min {iou.min():.6f}, differing pixels {int((a ^ b).sum())}")
    assert iou.mean() >= 0.99 and iou.min() >= 0.9


def test_forward_dec_1024_vs_oracle(model, cal_sd):
    """the largest scale of config 5: every head map of a 1024 x 1024 image against the CPU oracle, element-wise (subsampled) at
    SURVEY 8d's fp32 tolerance"""
    S = 1024
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    x = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(S)) - 0.5
    model._engine.raw_kp_logits = True
    try:
        with torch.no_grad():
            d = model.forward_dec(x.to(DEV))[:4]
    finally:
        model._engine.raw_kp_logits = False
    net = onet.Net({k: v.clone() for k, v in cal_sd.items()}, training=False)
    with torch.no_grad():
        o = net.forward_dec(x)[:4]
    worst = 0.0
    for l in range(4):
        for k, nm in enumerate(("kp_logit", "short", "mid")):
            got = d[l][k].cpu().numpy()[..., ::7, ::5].astype(np.float64)
            ref = (net.kp_logits[l] if k == 0 else o[l][k]).numpy()[..., ::7, ::5].astype(np.float64)
            rms = float(np.sqrt(np.mean(ref ** 2)))
            atol = 1e-5 * (1.0 if k == 0 else 3.0 if k == 1 else 6.0)      # logits literally; stated constants for the short / mid offset maps (pixels)
            ratio = float((np.abs(got - ref) / (atol + 1e-4 * np.abs(ref))).max())
            print(f"1024 c{l}.{nm}: rms {rms:.3g} worst |d|/bound {ratio:.3f}")
            worst = max(worst, ratio)
    assert worst <= 1.0


def test_seg_loss_targets_from_device_masks_equal_the_host_path(cal_sd):
    """SURVEY 8f N2, second half (seg_loss.py:57-80): with device-resident ground-truth masks the crops + nearest resizes run on the GPU
    (kg_crop_masks); loss and gradient are bit-identical to the host path (kg_host_crop_masks) -- including patches whose predicted box
    differs from the matched GT box (a real resize) and boxes clamped at the image border."""
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    from oracle import synth
    N, S = 2, 128
    x, gt_boxes, gt_masks, _ = synth.train_batch(N, S, S, 11, n_boxes=8)
    m = KGnet.resnet50(pretrained=False)
    m.load_state_dict(cal_sd)
    m = m.to(DEV).train()
    # predicted boxes = jittered GT boxes: IoU >= 0.5 matches with different sizes -> nearest resize
    rng = np.random.default_rng(0)
    pb = [np.concatenate([np.clip(b[:, :4] + rng.uniform(-2.5, 2.5, (len(b), 4)), 0, S - 1), b[:, 4:]], 1).astype(np.float32) for b in gt_boxes]
    lseg = SEG_loss(height=S, width=S)
    res = []
    for masks in (gt_masks, [torch.from_numpy(np.ascontiguousarray(mk)).to(DEV) for mk in gt_masks]):
        m.zero_grad()
        pred = m(x.to(DEV), pb)[4]
        loss = lseg(pred, masks, gt_boxes)
        assert loss is not None
        loss.backward()
        res.append((float(loss), m.get_tensor("seg_head.2.weight").grad.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])

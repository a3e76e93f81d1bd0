"""GPU parity tests of the float64 post-processing pipeline: bit-exact against the golden fixtures
generated from the reference and against the CPU oracle (oracle/kg_oracle.c) on larger seeded inputs."""
import hashlib
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import nms as knms  # noqa: E402
from kg_instance_segmentation_amd import postprocessing as kpp  # noqa: E402
from oracle import postproc as op  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def _inputs(g, name):
    if name == "adv":
        return g["adv.kp"], g["adv.short"], g["adv.mid"]
    H, W, n, seed = [int(v) for v in g[f"{name}.cfg"]]
    kp, short, mid, _ = synth.head_maps(H, W, n, seed)
    return kp, short, mid


def run_stages(kp, short, mid):
    t = [torch.from_numpy(a).to(DEV) for a in (kp, short, mid)]
    skel, nsk, dbg = kpp.skeletons_device(*t, debug=True)
    torch.cuda.synchronize()
    npk = int(dbg["npeaks"].item())
    pk = dbg["peaks"][:, :npk].cpu().numpy().T  # [n,3] id,x,y
    return dict(heat=dbg["heat"].cpu().numpy(), blur=dbg["blur"].cpu().numpy(), peaks=pk, peak_conf=dbg["conf"][:npk].cpu().numpy(),
                skel=skel[:int(nsk.item())].cpu().numpy())


def diff_report(name, got, ref):
    if got.shape != ref.shape:
        print(f"[{name}] shape {got.shape} vs {ref.shape}")
        return False
    nbad = int((got != ref).sum())
    if nbad:
        idx = np.argwhere(got != ref)[:6]
        print(f"[{name}] {nbad} mismatches; first:", [(tuple(i), got[tuple(i)], ref[tuple(i)]) for i in idx])
    else:
        print(f"[{name}] bit-exact ({got.size} values)")
    return nbad == 0


@pytest.mark.parametrize("name", ["s64", "s96x128", "s256", "adv"])
def test_stages_vs_golden(golden, name):
    g = golden("postproc.npz")
    kp, short, mid = _inputs(g, name)
    r = run_stages(kp, short, mid)
    ok = True
    if f"{name}.heat" in g:
        ok &= diff_report(name + ".heat", r["heat"], g[f"{name}.heat"])
        ok &= diff_report(name + ".blur", r["blur"], g[f"{name}.blur"])
    else:
        ok &= bool(np.array_equal(sha(r["heat"]), g[f"{name}.heat_sha"]))
        ok &= bool(np.array_equal(sha(r["blur"]), g[f"{name}.blur_sha"]))
        if not ok:  # localise with the oracle
            diff_report(name + ".heat(oracle)", r["heat"], op.hough(kp, short))
            diff_report(name + ".blur(oracle)", r["blur"], op.gauss(op.hough(kp, short)))
    ok &= diff_report(name + ".peaks", r["peaks"], g[f"{name}.peaks"])
    ok &= diff_report(name + ".peak_conf", r["peak_conf"], g[f"{name}.peak_conf"])
    ok &= diff_report(name + ".skel", r["skel"], g[f"{name}.skel"])
    assert ok
    # public API
    sk = kpp.get_skeletons_and_masks(*[torch.from_numpy(a).to(DEV) for a in (kp, short, mid)])
    assert len(sk) == len(g[f"{name}.skel"]) and all(np.array_equal(a, b) for a, b in zip(sk, g[f"{name}.skel"]))
    ref = kpp.refine_skeleton(sk)
    assert np.array_equal(np.array(ref).reshape(-1, 5, 3), g[f"{name}.refined"])


def test_boxes_gather_nms_vs_golden(golden):
    g = golden("postproc.npz")
    sks = [[s.copy() for s in g[k]] for k in ("s256.refined", "s96x128.refined", "s64.refined", "adv.refined")]
    for s, sc in zip(sks, (1, 2, 4, 8)):
        b = np.asarray(kpp.skeleton_to_box([a.copy() for a in s], sc)).reshape(-1, 5)
        assert diff_report(f"boxes.scale{sc}", b, g[f"boxes.scale{sc}"])
    gat = kpp.gather_skeleton(*sks)
    assert diff_report("gather", gat, g["gather"])
    for th in (0.5, 0.3):
        assert diff_report(f"nms.{th}", knms.non_maximum_suppression_numpy(gat, th), g[f"nms.{th}"])
    assert knms.non_maximum_suppression_numpy(np.zeros((0, 5)), 0.5) is None
    hand = g["hand.skel"]
    assert diff_report("hand.boxes_all", np.asarray(kpp.skeleton_to_box([s.copy() for s in hand], 2)).reshape(-1, 5), g["hand.boxes_all"])
    assert diff_report("hand.nms", knms.non_maximum_suppression_numpy(g["hand.boxes_all"], 0.5), g["hand.nms"])
    assert np.array_equal(np.array(kpp.refine_skeleton([s for s in hand])).reshape(-1, 5, 3), g["hand.refined"])


@pytest.mark.parametrize("H,W,n,seed", [(512, 512, 300, 3), (128, 192, 40, 4), (1024, 1024, 300, 5)])
def test_full_size_vs_oracle(H, W, n, seed):
    """BASELINE config sizes: dense cells (about 300 instances); every stage bit-identical to the CPU oracle."""
    kp, short, mid, _ = synth.head_maps(H, W, n, seed)
    t0 = time.time(); r = run_stages(kp, short, mid); t_gpu = time.time() - t0
    t0 = time.time()
    heat = op.hough(kp, short); blur = op.gauss(heat); ids, xs, ys, conf = op.peaks(blur); skel = op.group(ids, xs, ys, conf, mid)
    t_cpu = time.time() - t0
    print(f"{H}x{W}: peaks {len(ids)} skeletons {len(skel)} gpu {t_gpu*1e3:.1f} ms (incl. copies) oracle {t_cpu*1e3:.1f} ms")
    ok = diff_report("heat", r["heat"], heat)
    ok &= diff_report("blur", r["blur"], blur)
    ok &= diff_report("peaks", r["peaks"], np.stack([ids, xs, ys], 1).reshape(-1, 3))
    ok &= diff_report("conf", r["peak_conf"], conf)
    ok &= diff_report("skel", r["skel"], skel)
    assert ok


def test_random_maps_many_peaks():
    """Random (untrained-net-like) maps: thousands of peaks, large offsets, heavy Hough cells."""
    rng = np.random.default_rng(12)
    H, W = 192, 256
    kp = (rng.random((1, 5, H, W)) ** 2).astype(np.float32)
    short = (rng.normal(size=(1, 10, H, W)) * 3).astype(np.float32)
    short[0, :, 50:60, 50:60] = 0.25      # many votes into few cells
    mid = (rng.normal(size=(1, 40, H, W)) * 8).astype(np.float32)
    r = run_stages(kp, short, mid)
    heat = op.hough(kp, short); blur = op.gauss(heat); ids, xs, ys, conf = op.peaks(blur); skel = op.group(ids, xs, ys, conf, mid)
    print("peaks", len(ids), "skeletons", len(skel))
    ok = diff_report("heat", r["heat"], heat) & diff_report("blur", r["blur"], blur)
    ok &= diff_report("peaks", r["peaks"], np.stack([ids, xs, ys], 1).reshape(-1, 3)) & diff_report("skel", r["skel"], skel)
    assert ok


def test_random_maps_general_grouping_path():
    """> 8192 peaks: the grouping kernel leaves its LDS fast path (GK_NL in postproc.hip) for the general global-memory one."""
    rng = np.random.default_rng(13)
    H, W = 384, 512
    kp = (rng.random((1, 5, H, W)) ** 2).astype(np.float32)
    short = (rng.normal(size=(1, 10, H, W)) * 2).astype(np.float32)
    mid = (rng.normal(size=(1, 40, H, W)) * 6).astype(np.float32)
    r = run_stages(kp, short, mid)
    heat = op.hough(kp, short); blur = op.gauss(heat); ids, xs, ys, conf = op.peaks(blur); skel = op.group(ids, xs, ys, conf, mid)
    print("peaks", len(ids), "skeletons", len(skel))
    assert len(ids) > 8192
    ok = diff_report("peaks", r["peaks"], np.stack([ids, xs, ys], 1).reshape(-1, 3)) & diff_report("skel", r["skel"], skel)
    assert ok


def test_detect_fused_matches_oracle():
    decs = []
    for sc, (H, W, n, seed) in zip((1, 2, 4, 8), [(256, 256, 60, 21), (128, 128, 30, 22), (64, 64, 10, 23), (32, 32, 3, 24)]):
        kp, short, mid, _ = synth.head_maps(H, W, n, seed, smin=12 // min(sc, 2), smax=40 // sc + 8)
        decs.append((kp, short, mid))
    ref = op.detect(decs, 0.5)
    got = kpp.detect([[torch.from_numpy(a).to(DEV) for a in d] for d in decs], 0.5)
    assert ref is not None and got is not None
    assert diff_report("detect", got, ref)


def test_empty_maps():
    z = [torch.zeros(1, c, 16, 16, device=DEV) for c in (5, 10, 40)]
    assert kpp.get_skeletons_and_masks(*z) == []
    assert kpp.gather_skeleton([], [], [], []).shape == (0,)
    assert kpp.detect([z, z, z, z]) is None


@pytest.mark.parametrize("image_hw", [(512, 512), (520, 696), (300, 200)])
def test_paste_masks_vs_oracle(image_hw):
    """kg_mask_paste (test.py:127-157 in one launch) against oracle/paste.py (OpenCV's published INTER_LINEAR float rule) bit for
    bit: same masks after the >= seg_thresh cut, same scaled boxes -- with boxes whose crop size differs from the patch size
    (so that the first resize is a genuine interpolation), boxes on the border and an identity-size image."""
    from kg_instance_segmentation_amd import KGnet
    from oracle import paste as opaste, weightgen
    S = 512
    m = KGnet.resnet50(pretrained=False)
    m.load_state_dict(weightgen.gen_state_dict(0, variant="cal"))
    m = m.to(DEV).eval()
    x = (torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(3)) - 0.5).to(DEV)
    bx = synth.random_boxes(S, S, 40, 11)
    bx = np.concatenate([bx, [[0.4, 0.6, 30.5, 41.5], [470.5, 480.2, 511.0, 511.0], [100.5, 100.5, 131.5, 140.49]]], 0)
    boxes = [np.concatenate([bx, np.linspace(0.9, 0.3, len(bx))[:, None]], 1).astype(np.float32)]
    with torch.no_grad():
        feats = m.forward_dec(x)[4]
        pred = m.forward_seg(feats, boxes)
    ih, iw = image_hw
    got = kpp.paste_masks(pred, S, S, iw, ih, 0.5)
    ref = opaste.paste_masks([[[p.cpu().numpy() for p in pp] for pp in pred[0]], [[d.numpy() for d in dd] for dd in pred[1]]], S, S, iw, ih, 0.5)
    assert got[0].shape == ref[0].shape == (len(bx), ih, iw) and got[0].dtype == np.float32
    nbad = int((got[0] != ref[0]).sum())
    print(f"paste {image_hw}: {len(bx)} masks, {int(ref[0].sum())} foreground pixels, {nbad} mismatching pixels")
    assert nbad == 0
    assert np.array_equal(got[1], ref[1])
    dev_masks, _ = kpp.paste_masks(pred, S, S, iw, ih, 0.5, device_u8=True)
    assert dev_masks.dtype == torch.uint8 and np.array_equal(dev_masks.cpu().numpy().astype(np.float32), ref[0])
    assert kpp.paste_masks(None, S, S, iw, ih, 0.5) is None


def _heat(kp, short):
    mid = np.zeros((1, 40) + kp.shape[2:], np.float32)
    return run_stages(kp, short, mid)["heat"]


@pytest.mark.parametrize("case", ["far_votes", "heavy_cell_beyond_scratch", "tile_overflow", "far_list_overflow"])
def test_hough_tile_formulation_edge_cases(case):
    """The tile formulation of the Hough vote (csrc/postproc.hip: one workgroup per 32 x 32-cell tile, votes gathered from the 64 x 64 source
    pixels around it, LDS slabs) against the oracle, bit for bit, where it leaves its common path: votes from OUTSIDE a tile's source region
    (the global far list), a cell with more votes than the rank-sort scratch holds (window loop), a tile with more votes than its LDS slabs
    hold and more far votes than the list holds (both: the scatter formulation takes over)."""
    rng = np.random.default_rng(31)
    H, W = 160, 192
    kp = np.clip(rng.random((1, 5, H, W)), 0.05, 1).astype(np.float32)
    short = rng.normal(0, 0.6, (1, 10, H, W)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    if case == "far_votes":
        idx = rng.choice(H * W, 300, replace=False)
        for c in range(10):
            short[0, c].flat[idx] += rng.choice([-1, 1], 300) * rng.uniform(17, 120, 300)      # beyond the 16 .. 48 px reach of a tile's region
    elif case == "heavy_cell_beyond_scratch":
        # a 22 x 22 block of every channel votes into ONE cell with integer offsets (three of the four bilinear weights are 0): 484 votes > HT_SCR
        short[0, 0::2, 40:62, 70:92] = (81 - xx[40:62, 70:92]).astype(np.float32)
        short[0, 1::2, 40:62, 70:92] = (51 - yy[40:62, 70:92]).astype(np.float32)
        short[0, :, 100:104, 100:104] = 0.5          # and cells fed by four half-weight votes from 16 pixels
    elif case == "tile_overflow":
        # a 48 x 48 block votes (fractional offsets: all four corners live) into the 4 x 4 cells around (80, 100): > HT_CAP votes in one tile
        short[0, 0::2, 56:104, 76:124] = (100.3 - xx[56:104, 76:124]).astype(np.float32) + rng.uniform(0, 3, (48, 48)).astype(np.float32)
        short[0, 1::2, 56:104, 76:124] = (80.3 - yy[56:104, 76:124]).astype(np.float32) + rng.uniform(0, 3, (48, 48)).astype(np.float32)
    else:
        short = (rng.normal(0, 60, (1, 10, H, W))).astype(np.float32)                          # an untrained network: nearly every vote is far
    got = _heat(kp, short)
    ref = op.hough(kp, short)
    assert diff_report(case, got, ref)


def test_scatter_formulation_of_the_hough_vote_still_bit_exact():
    """KG_HOUGH_TILE=0 runs the scatter formulation alone (the path the tile formulation falls back to): the golden stage tests, the full-size
    oracle comparison and the random maps in a process that selects it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KG_HOUGH_TILE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_postproc.py"), "-q", "-x", "-k",
                        "stages_vs_golden or full_size_vs_oracle or random_maps"], capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "9 passed" in r.stdout, r.stdout[-500:]

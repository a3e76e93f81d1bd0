#!/usr/bin/env python
"""bench.py -- KGnet train-step throughput on MI355X (metric of BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size S] [--boxes NB]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = the body of the reference's training loop (train.py:137-156) on one synthetic batch:
zero_grad, forward (forward_dec + forward_seg on the GT boxes), 4x DetectionLossAll + SEG_loss, backward,
Adam step, loss.item().  Per GPU: batch 8 of 3x512x512 images with 300 GT boxes each (BASELINE configs[1]/[3],
weak scaling).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line with the
whole-job imgs/s, the roofline of the dominant kernel (HIP events around its launches during the timed steps) and a
CPU baseline (the oracle's torch restatement of the same step, timed on the host cores, bounded sample).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL between the ranks of a node (host driver requirement)

import numpy as np
import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md


def random_boxes(H, W, n, seed, smin=14, smax=40):
    """n axis-aligned boxes (y1,x1,y2,x2) with integer corners, sides U[smin,smax)."""
    rng = np.random.default_rng(seed)
    hs = np.minimum(rng.integers(smin, smax, n), H - 2); ws = np.minimum(rng.integers(smin, smax, n), W - 2)
    y1 = (rng.random(n) * (H - 1 - hs)).astype(np.int64) + 1
    x1 = (rng.random(n) * (W - 1 - ws)).astype(np.int64) + 1
    return np.stack([y1, x1, y1 + hs, x1 + ws], 1).astype(np.float64)


def make_batch(N, S, nboxes, seed, dev):
    """Synthetic batch in the collater's layout (collater.py:4-25).  The 55-channel GT maps of the four scales are built
    by the product's GPU ground-truth generator (preprocessing.get_ground_truth semantics, csrc/preproc.hip) from the
    boxes scaled as dataset_base.masks_to_bboxes does (floor(box / scale); keypoints tl, tr, bl, br, centre)."""
    from kg_instance_segmentation_amd import preprocessing as kprep
    x = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(seed)) - 0.5
    gt_boxes, gt_masks, lv = [], [], [[] for _ in range(4)]
    for i in range(N):
        bx = random_boxes(S, S, nboxes, seed * 1000 + i)
        gt_boxes.append(np.concatenate([bx, np.ones((len(bx), 1))], 1).astype(np.float32))
        m = np.zeros((len(bx), S, S), np.float32)
        for k, b in enumerate(bx.astype(int)):
            m[k, b[0]:b[2] + 1, b[1]:b[3] + 1] = 1.0
        gt_masks.append(m)
        for l, sc in enumerate((1, 2, 4, 8)):
            y1, x1, y2, x2 = np.floor(bx / sc).T
            kps = np.stack([np.stack([x1, y1], 1), np.stack([x2, y1], 1), np.stack([x1, y2], 1), np.stack([x2, y2], 1),
                            np.stack([(x1 + x2) / 2, (y1 + y2) / 2], 1)], 1).astype(np.float32)
            lv[l].append(kprep.get_ground_truth_device(kps, S // sc, S // sc, dev))
    gt = [torch.stack(v) for v in lv]
    return x.to(dev), gt, gt_masks, gt_boxes


class KernelTimer:
    """HIP-event brackets around every kg_conv2d_igemm / kg_conv2d_wgrad launch on the launch stream."""

    # the 7x7 LDS-halo kernel family (forward + input gradient of the first-layer head convs), rocprofv3's names: the 8-wave kernel and
    # (round 5) its 4-wave sibling with the blocked accumulation, which serves the multi-chunk 3-product forward launches -- the same
    # workgroup tile, LDS image and load protocol; the roofline entry is taken over the launches of both
    DOMINANT = ("conv_halo_kernel<7, 1, 8, 0>", "conv_halo7_w4_kernel<false, 1>", "conv_halo7_w4_kernel<true, 1>", "conv_halo7_w4_kernel<false, 2>",
                "conv_halo7_w4_kernel<true, 2>", "conv_halo7_w4_kernel<false, 2> + conv_halo_kernel<7, 1, 8, 0>",
                "conv_halo7_w4_kernel<true, 2> + conv_halo_kernel<7, 1, 8, 0>")

    def __init__(self):
        self.rec = []
        self.on = False
        self.only_dominant = False   # timed region: events only around the dominant kernel's launches (48 events/step);
                                     # bracketing all ~270 conv launches costs ~5 % of the step in host time

    def install(self):
        from kg_instance_segmentation_amd import ops
        timer = self
        orig_conv, orig_wgrad = ops.conv_igemm, ops.conv_wgrad
        from kg_instance_segmentation_amd import _lib as klib

        def kname(t):      # rocprofv3's name of the kernel the call just launched (kg_last_kernel): the table uses the profiler's names
            return klib.last_kernel(ops.fmt_of(t))

        def flushq():      # the batched weight (re)pack rides in front of a step's first conv launch: keep it out of that launch's bracket
            ops.flush_packs()

        def conv(x, pw, cout, geom, *a, **k):
            if not timer.on or timer.only_dominant:
                return orig_conv(x, pw, cout, geom, *a, **k)
            M, _, _, _, _, KH, KW, _, _ = geom
            tile = k.get("tile", 0) or (1 if cout <= 16 else 2 if cout <= 32 else 3 if cout <= 64 else 4)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            flushq(); s.record(); r = orig_conv(x, pw, cout, geom, *a, **k); e.record()
            yP = getattr(k.get("y"), "P", 1) if k.get("y") is not None else 0
            timer.rec.append((kname(x), 2.0 * M * cout * KH * KW * getattr(pw, "cin_real", pw.cin_pad), s, e,
                              f"M={M} cout={cout} k={KH} cinp={pw.cin_pad} mode={k.get('mode', 0)} tile={tile} stride={geom[7]} xP={getattr(pw, 'xP', 1)} yP={yP} products={getattr(pw, 'vp', 1)}"))
            return r

        def wgrad(x, dy, cin, cout, geom, grads, *a, **k):
            if not timer.on or timer.only_dominant:
                return orig_wgrad(x, dy, cin, cout, geom, grads, *a, **k)
            M, _, _, _, _, KH, KW, _, _ = geom
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            flushq(); s.record(); r = orig_wgrad(x, dy, cin, cout, geom, grads, *a, **k); e.record()
            timer.rec.append((kname(x) + " (weight gradient; its split reduction runs in the batched flush)", 2.0 * M * cout * KH * KW * cin, s, e, f"M={M} cout={cout} cin={cin} k={KH} mode={k.get('mode', 0)} route={r}"))
            return r
        orig_halo = ops.conv_halo

        def halo(x, pw, cout, N, H, W, KS, *a, **k):
            if not timer.on or (timer.only_dominant and KS != 7):
                return orig_halo(x, pw, cout, N, H, W, KS, *a, **k)
            wc = 1
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            flushq(); s.record(); r = orig_halo(x, pw, cout, N, H, W, KS, *a, **k); e.record()
            cin = k.get("algo_cin") or getattr(pw, "cin_real", pw.cin_pad)     # real channels (not the 8 / 64 padding, not the plane copies); fused second-layer head dgrad: 5 / 10 / 40 real dY channels per head
            px = N * H * W
            if px == 0:      # ragged per-box crops (seg branch): the REAL pixels of the boxes (rconv's row count), not the padded tile area
                px = int(k.get("total_rows") or 0)
            fl = 2.0 * px * cout * KS * KS * cin
            timer.rec.append((kname(x), fl, s, e, f"N={N} H={H} cout={cout} cinp={pw.cin_pad} products={pw.vp}" + (f" algo_cin={cin}" if "algo_cin" in k else ""), fl * pw.vp))
            return r
        orig_1x1 = ops.conv1x1

        def c1x1(x, pw, cout, y, *a, **k):
            if not timer.on or timer.only_dominant:
                return orig_1x1(x, pw, cout, y, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            flushq(); s.record(); r = orig_1x1(x, pw, cout, y, *a, **k); e.record()
            timer.rec.append((kname(x), 2.0 * x.shape[0] * cout * getattr(pw, "cin_real", pw.cin_pad), s, e, f"M={x.shape[0]} cout={cout} K={pw.cin_pad}"))
            return r
        orig_h2 = ops.conv_halo_heads2

        def heads2(x, pw, bias64, vmap, kp, sh, md, N, H, W, C, **k):
            if not timer.on or timer.only_dominant:
                return orig_h2(x, pw, bias64, vmap, kp, sh, md, N, H, W, C, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            flushq(); s.record(); r = orig_h2(x, pw, bias64, vmap, kp, sh, md, N, H, W, C, **k); e.record()
            timer.rec.append((kname(x), 2.0 * N * H * W * 55 * 49 * C, s, e, f"N={N} H={H} cout=5+10+40 C={C}"))
            return r
        # engine/seg call through `ops.<fn>` (and conv_auto resolves these names at call time)
        ops.conv_igemm, ops.conv_wgrad, ops.conv_halo, ops.conv1x1, ops.conv_halo_heads2 = conv, wgrad, halo, c1x1, heads2

    def dump(self, path, steps):
        rows = {}
        for name, fl, s, e, desc in (r_[:5] for r_ in self.rec):
            r = rows.setdefault((name, desc), [0.0, 0.0, 0])
            r[0] += s.elapsed_time(e); r[1] += fl; r[2] += 1
        with open(path, "w") as f:
            for (name, desc), (ms, fl, n) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
                f.write(f"{ms / steps:8.3f} ms/step  {fl / max(ms, 1e-9) / 1e9:8.1f} TF  x{n / steps:5.1f}  {name}  {desc}\n")

    def summary(self, rec=None):
        agg = {}
        for r_ in (self.rec if rec is None else rec):
            name, fl, s, e = r_[:4]
            a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
            a[0] += s.elapsed_time(e) * 1e-3; a[1] += fl; a[2] += 1
            a[3] += r_[5] if len(r_) > 5 else fl        # MFMA-issued FLOPs: algorithmic FLOPs x the plane products of the launch
        return {k: {"seconds": v[0], "flops": v[1], "launches": v[2], "mfma_flops": v[3]} for k, v in agg.items()}


def _build_id():
    import hashlib
    from kg_instance_segmentation_amd import _lib as klib
    h = hashlib.sha256()
    for path in (klib.LIB_PATH, klib.LIB_F16_PATH):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _pmc_file(suffix):
    """The newest committed rocprofv3 PMC summary (tools/profile_round.sh -> profiles/*_<suffix>.json) that was collected ON THIS BUILD: the file
    records the sha256 of the two libraries (tools/pmc_summary.py) and is refused when it differs from the libraries this process loads."""
    import glob
    bid = _build_id()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{suffix}.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("_build", {}).get("libs_sha256_16") == bid:
            return d, os.path.basename(path)
    return None, None


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC pass of THIS build (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, gfx950
    correction of MI355X_MICROARCH.md), or None when no committed pass matches the loaded libraries."""
    d, _ = _pmc_file("pmc_hbm")
    fam = {n.replace(" ", "") for n in KernelTimer.DOMINANT}
    rows = [v for k, v in (d or {}).items() if k.replace(" ", "") in fam and "hbm_bytes" in v]
    if not rows:
        return None
    return sum(v["hbm_bytes"] * v["launches"] for v in rows) / sum(v["launches"] for v in rows)      # launch-weighted over the family


def pmc_mfma():
    """MFMA-pipe utilisation of the dominant kernel from the PMC pass of THIS build (SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs,
    GRBM_GUI_ACTIVE over the 8 XCDs), or None when no committed pass matches the loaded libraries."""
    d, name = _pmc_file("pmc_mfma_lds")
    fam = {n.replace(" ", "") for n in KernelTimer.DOMINANT}
    rows = [v for k, v in (d or {}).items() if k.replace(" ", "") in fam and v.get("GRBM_GUI_ACTIVE")]
    if not rows:
        return None
    L = sum(v["launches"] for v in rows)
    busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] * v["launches"] for v in rows) / 1024.0
    act = sum(v["GRBM_GUI_ACTIVE"] * v["launches"] for v in rows) / 8.0
    return {"mfma_busy_frac": busy / act, "active_cycles_per_launch": act / L,
            "lds_bank_conflict_cycles": sum((v.get("SQ_LDS_BANK_CONFLICT") or 0.0) * v["launches"] for v in rows) / L, "file": name}


def eval_inputs(S, n, seed):
    """Config 5 of BASELINE.json: GT-derived head maps of n instances (+ N(0, 0.05) on kp clipped to [0,1], N(0, 0.5 px) on
    the offsets) at the four scales of an SxS image.  Returns [[kp, short, mid] x 4] as numpy fp32 [1,C,H,W] and the boxes."""
    from kg_instance_segmentation_amd import preprocessing as kprep
    f = S / 512.0
    boxes = random_boxes(S, S, n, seed, max(4, int(14 * f)), max(8, int(40 * f)))
    dec = []
    for l, sc in enumerate((1, 2, 4, 8)):
        H = S // sc
        y1, x1, y2, x2 = np.floor(boxes / sc).T
        kps = np.stack([np.stack([x1, y1], 1), np.stack([x2, y1], 1), np.stack([x1, y2], 1), np.stack([x2, y2], 1),
                        np.stack([(x1 + x2) / 2, (y1 + y2) / 2], 1)], 1).astype(np.float32)
        gt = kprep.get_ground_truth_device(kps, H, H).cpu().numpy()
        rng = np.random.default_rng(seed * 10 + l)
        kp = np.clip(gt[0:5] + rng.normal(0, 0.05, (5, H, H)), 0, 1).astype(np.float32)
        sh = (gt[5:15] + rng.normal(0, 0.5, (10, H, H))).astype(np.float32)
        md = (gt[15:55] + rng.normal(0, 0.5, (40, H, H))).astype(np.float32)
        dec.append([kp[None], sh[None], md[None]])
    return dec, boxes


def gt_bench(args, dev):
    """--mode gt: ground-truth maps (4 scales) of a batch of 8 images with 300 instances each at 512^2 on the GPU
    (kg_gt_maps) vs the NumPy oracle on one image (the reference's own NumPy code takes ~58 s per image)."""
    from kg_instance_segmentation_amd import preprocessing as kprep
    S, N, n = args.size, args.batch, args.boxes
    kps = []
    for i in range(N):
        bx = random_boxes(S, S, n, 1000 + i)
        per = []
        for sc in (1, 2, 4, 8):
            y1, x1, y2, x2 = np.floor(bx / sc).T
            per.append(np.stack([np.stack([x1, y1], 1), np.stack([x2, y1], 1), np.stack([x1, y2], 1), np.stack([x2, y2], 1),
                                 np.stack([(x1 + x2) / 2, (y1 + y2) / 2], 1)], 1).astype(np.float32))
        kps.append(per)
    dk = [[torch.from_numpy(k).to(dev) for k in per] for per in kps]
    outs = [[torch.empty(55, S // sc, S // sc, device=dev) for sc in (1, 2, 4, 8)] for _ in range(N)]

    def run():
        for i in range(N):
            for l, sc in enumerate((1, 2, 4, 8)):
                kprep.get_ground_truth_device(dk[i][l], S // sc, S // sc, dev, out=outs[i][l])
    for _ in range(2):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    hbm = N * sum(55 * (S // sc) ** 2 * 4 for sc in (1, 2, 4, 8))
    out = {"metric": f"imgs/s (ground-truth maps, 4 scales, {n} instances) at {S}x{S}", "value": N / dt, "unit": "imgs/s", "n_gpus": 1,
           "steps": args.steps, "warmup": 2, "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64 (distances) -> f32 maps", "data": "synthetic",
           "config": {"workload": f"preprocessing.get_ground_truth x 4 scales, batch {N}, {n} instances/img, keypoints resident in HBM"},
           "roofline": {"bound": "hbm", "achieved": hbm / dt / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": hbm / dt / 1e9 / 8000.0, "traffic": None,
                        "note": "algorithmic bytes = the 55-channel fp32 maps written once; the kernel is latency/ALU bound (instance loop per pixel)"}}
    if not args.no_cpu_baseline:
        from oracle import preproc
        t0 = time.perf_counter()
        ok = True
        for l, sc in enumerate((1, 2, 4, 8)):
            ref = preproc.ground_truth(kps[0][l], S // sc, S // sc)
            ok = ok and bool(np.array_equal(outs[0][l].cpu().numpy().astype(np.float64), ref))
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / dtc, "unit": "imgs/s", "cores": 1, "kind": "port", "sample": "1 image, 4 scales, oracle/preproc.py (vectorised NumPy)",
                               "bit_identical": ok}
    print(json.dumps(out))


def eval_bench(args, dev):
    """--mode eval: the test.py:97-123 inference path per image -- forward_dec (eval BN), post-processing of the 4 scales +
    NMS (bit-exact fp64 on the GPU), forward_seg on the detected boxes -- at 256 / 512 / 1024 with ~300 instances."""
    from kg_instance_segmentation_amd import KGnet, postprocessing as kpp
    from oracle import weightgen           # generator of the calibrated synthetic weights only (no oracle compute in the timed legs)
    sd = weightgen.gen_state_dict(0, variant="cal")
    model = KGnet.resnet50(pretrained=False)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    per = {}
    for S in (256, 512, 1024):
        dec_np, boxes = eval_inputs(S, 300, 5)
        dec = [[torch.from_numpy(a).to(dev) for a in d] for d in dec_np]
        x = (torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(S)) - 0.5).to(dev)

        def timed(fn, reps):
            for _ in range(2):
                r = fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps, r
        with torch.no_grad():
            t_pp, det = timed(lambda: kpp.detect(dec), args.steps)
            t_fd, out = timed(lambda: model.forward_dec(x), args.steps)
            bb = (det if det is not None else np.zeros((0, 5))).astype(np.float32)   # [y1,x1,y2,x2,score] in pixels (test.py:119-123)
            t_fs, pred = timed(lambda: model.forward_seg(out[4], [bb]), args.steps)
            t_pm, pasted = timed(lambda: kpp.paste_masks(pred, S, S, S, S, 0.5, device_u8=True), args.steps)
        # roofline pass (untimed): HIP events at the phase boundaries of the post-processing (kg_postproc_timing_*) and around boxes + NMS
        from kg_instance_segmentation_amd import _lib as klib
        import ctypes
        ph = (ctypes.c_float * 4)()
        tm = {}
        klib.call("kg_postproc_timing_begin")
        kpp.detect(dec, timing=tm)
        torch.cuda.synchronize()
        klib.call("kg_postproc_timing_end", ph)
        npx = sum((S // sc) ** 2 for sc in (1, 2, 4, 8))
        phases = {"hough_ms": ph[0], "gauss_ms": ph[1], "peaks_ms": ph[2], "group_ms": ph[3], "boxes_nms_ms": tm.get("boxes_nms_ms"),
                  "hough_GBps": 100.0 * npx / (ph[0] * 1e-3) / 1e9, "gauss_GBps": 160.0 * npx / (ph[1] * 1e-3) / 1e9,
                  "peaks_GBps": 40.0 * npx / (ph[2] * 1e-3) / 1e9, "pixels_4_scales": npx}
        per[S] = {"paste_masks_ms": 1e3 * t_pm, "postproc_phases": phases, "_x": x.cpu(), "_bb": bb, "_masks": None if pasted is None else pasted[0].cpu().numpy(),
                  "postproc_nms_ms": 1e3 * t_pp, "forward_dec_ms": 1e3 * t_fd, "forward_seg_ms": 1e3 * t_fs, "detections": 0 if det is None else len(det),
                  "imgs_per_s_postproc": 1.0 / t_pp, "imgs_per_s_end_to_end": 1.0 / (t_pp + t_fd + t_fs), "_det": det, "_dec": dec_np}
    out = {"metric": "imgs/s (eval: forward_dec + post-proc x4 + NMS + forward_seg) at 512x512, ~300 instances", "value": per[512]["imgs_per_s_end_to_end"],
           "unit": "imgs/s", "n_gpus": 1, "steps": args.steps, "warmup": 2, "ms_per_step": 1e3 / per[512]["imgs_per_s_end_to_end"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 (post-processing + NMS) / f32 as hi + lo half planes (network)", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4]: multi-scale eval 256/512/1024, GT-derived head maps of 300 instances + noise for post-processing/NMS, "
                                  "random-init network for forward_dec / forward_seg on the detected boxes, batch 1"}}
    p5 = per[512]["postproc_phases"]
    hbm_ms = p5["hough_ms"] + p5["gauss_ms"] + p5["peaks_ms"]
    ach = 300.0 * p5["pixels_4_scales"] / (hbm_ms * 1e-3) / 1e9
    out["roofline"] = {"bound": "hbm", "kernel": "hough_* + gauss_kernel + peaks_kernel (P1-P3 of the four scales of one 512 x 512 image)",
                       "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                       "algorithmic_bytes_per_pixel": {"hough": 100, "gauss": 160, "peaks": 40},
                       "phase_ms_sum_over_scales": {k: p5[k] for k in ("hough_ms", "gauss_ms", "peaks_ms", "group_ms", "boxes_nms_ms")},
                       "latency_bound": {"group_us": 1e3 * p5["group_ms"], "boxes_nms_us": 1e3 * (p5["boxes_nms_ms"] or 0.0),
                                         "note": "greedy keypoint grouping (one workgroup per scale, strictly sequential seeds) and NMS (one workgroup) are "
                                                 "dependency chains, not bandwidth: reported as time"},
                       "note": "achieved = algorithmic bytes (SURVEY 8d: Hough 100 B/px, Gaussian 160 B/px, peaks 40 B/px over the 348 160 pixels of the "
                               "four scales) / the summed HIP-event time of those phases (the four scales run on four streams: their wall time overlaps); at "
                               "348 k pixels the phases are launch- / latency-sized (tens of microseconds each), far from the 8 TB/s roof"}
    if not args.no_cpu_baseline:      # checker + baseline leg: the C oracle on the same head maps
        from oracle import postproc as op
        cb = {}
        for S in (256, 512, 1024):
            t0 = time.perf_counter(); ref = op.detect(per[S]["_dec"]); dtc = time.perf_counter() - t0
            det = per[S]["_det"]
            same = 0 if (ref is None or det is None) else int(sum(1 for a in ref if any(np.array_equal(a, b) for b in det)))
            nref = 0 if ref is None else len(ref)
            cb[S] = {"oracle_ms": 1e3 * dtc, "ref_boxes": nref, "identical_boxes": same, "grouping_match_rate": same / max(nref, 1)}
        out["cpu_baseline"] = {"value": 1e3 / cb[512]["oracle_ms"], "unit": "imgs/s (post-proc + NMS only)", "cores": 1, "kind": "port",
                               "sample": "one image per size, oracle/kg_oracle.c via oracle/postproc.py", "per_size": cb}
        # mask IoU against the fp32 oracle network on the same image and boxes (256 and 512: seconds of CPU work)
        from oracle import net as onet, paste as opaste
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        net = onet.Net({k: v.clone() for k, v in sd.items()}, training=False)
        for S in (256, 512):
            if per[S]["_masks"] is None:
                continue
            t0 = time.perf_counter()
            with torch.no_grad():
                of = net.forward_dec(per[S]["_x"])[4]
                op_ = net.forward_seg(of, [per[S]["_bb"]])
            ref = opaste.paste_masks([[[p.numpy() for p in pp] for pp in op_[0]], [[np.asarray(d) for d in dd] for dd in op_[1]]], S, S, S, S, 0.5)
            dtc = time.perf_counter() - t0
            a, b = per[S]["_masks"] > 0, ref[0] > 0
            inter = (a & b).reshape(len(a), -1).sum(1).astype(np.float64); uni = (a | b).reshape(len(a), -1).sum(1).astype(np.float64)
            iou = inter / np.maximum(uni, 1)
            cb[S].update({"mask_iou_vs_oracle_mean": float(iou.mean()), "mask_iou_vs_oracle_min": float(iou.min()), "masks": int(len(a)),
                          "oracle_network_s": dtc})
    for S in per:
        for k in ("_det", "_dec", "_x", "_bb", "_masks"):
            per[S].pop(k)
    out["per_size"] = per
    print(json.dumps(out))


def cpu_baseline(S, nboxes, seed=0):
    """The oracle's torch-CPU restatement of ONE train step at batch 1 with the headline's boxes per image (bounded sample: ~6-10 s), all host cores."""
    from oracle import net as onet, synth, weightgen
    torch.set_num_threads(min(os.cpu_count() or 1, 32))   # bs=1 convs stop scaling (and regress) beyond a few dozen threads
    sd = weightgen.gen_state_dict(seed)
    names = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for n in names:
        sd[n].requires_grad_(True)
    x = torch.rand(1, 3, S, S, generator=torch.Generator().manual_seed(seed)) - 0.5
    bx = synth.random_boxes(S, S, nboxes, seed)
    gt_boxes = [np.concatenate([bx, np.ones((len(bx), 1))], 1).astype(np.float32)]
    gt_masks = [np.zeros((len(bx), S, S), np.float32)]
    for k, b in enumerate(bx.astype(int)):
        gt_masks[0][k, b[0]:b[2] + 1, b[1]:b[3] + 1] = 1.0
    gt = [torch.from_numpy(synth.gt_maps(np.floor(bx / sc), S // sc, S // sc))[None] for sc in (1, 2, 4, 8)]
    opt = torch.optim.Adam([sd[n] for n in names], lr=1e-4)
    t0 = time.time()
    net = onet.Net(sd, training=True)
    opt.zero_grad()
    d0, d1, d2, d3, pred = net.forward(x, gt_boxes)
    loss = sum(onet.detection_loss(p, t) for p, t in zip((d0, d1, d2, d3), gt))
    l2 = onet.seg_loss(pred, gt_masks, gt_boxes, S, S)
    if l2 is not None:
        loss = loss + l2
    loss.backward()
    opt.step()
    float(loss.detach())
    dt = time.time() - t0
    out = {"value": 1.0 / dt, "unit": "imgs/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"1 train step, batch 1, 3x{S}x{S}, {nboxes} GT boxes, torch-CPU oracle (oracle/net.py), {dt:.1f} s"}
    out.update(grouping_match_rate(S))
    return out


def grouping_match_rate(S):
    """Second half of BASELINE's metric: grouping / box assembly / NMS of the HIP path vs the oracle on identical head tensors
    (GT-derived maps of 300 instances + noise, 4 scales): |identical boxes| / |reference boxes|."""
    from kg_instance_segmentation_amd import postprocessing as kpp
    from oracle import postproc as op
    dev = torch.device("cuda", torch.cuda.current_device())
    dec_np, _ = eval_inputs(S, 300, 5)
    det = kpp.detect([[torch.from_numpy(a).to(dev) for a in d] for d in dec_np])
    t0 = time.time(); ref = op.detect(dec_np); dtc = time.time() - t0
    nref = 0 if ref is None else len(ref)
    same = 0 if (ref is None or det is None) else int(sum(1 for a in ref if any(np.array_equal(a, b) for b in det)))
    return {"grouping_match_rate": same / max(nref, 1), "grouping_ref_boxes": nref, "postproc_oracle_ms": 1e3 * dtc}


def dtype_label(precision):
    """`dtype` of the bench line = the arithmetic of the WHOLE step, derived from every plane entry of the policy (engine.PRECISIONS):
    forward planes (backbone, decoder, heads, seg) and the operand planes of the backward pass (dY, x / dY of the weight gradients, W of the
    input gradients).  A policy whose backward multiplies fewer planes than its forward says so."""
    from kg_instance_segmentation_amd import engine as kengine
    pol = kengine.PRECISIONS[precision]
    half = precision in kengine.HALF_POLICIES
    fmt, full = ("f16", 2) if half else ("bf16", 3)
    fwd, bwd = min(pol[:4]), min(pol[4:])
    name = {1: f"{fmt} single planes", 2: f"hi + lo {fmt} planes" + (" (22 significant bits)" if half else " (16 bits)"), 3: "hi + mid + lo bf16 planes (24 bits)"}

    def side(p):
        return ("f32-tolerance, " if p >= full else "") + name[min(p, 3)] + f", {kengine_products(p)} MFMA product{'s' if kengine_products(p) > 1 else ''}"
    if max(pol[:4]) != fwd:
        f = f"forward: backbone {name[pol[0]]}, rest {name[min(pol[1:4])]}"
    else:
        f = "forward: " + side(fwd)
    return f"{f} / backward: {side(bwd)}; {fmt} MFMA, fp32 accumulate"


def kengine_products(P):
    from kg_instance_segmentation_amd import ops
    return ops.vplanes(P, P)


def self_launch_cmd(n, argv, port):
    """The command `python bench.py --gpus N` re-runs itself as: the driver's own launcher line (one rank per GPU over RCCL)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n):
    """`python bench.py --gpus N` without a pre-spawned world (no WORLD_SIZE in the environment): re-run this very command line under
    torch.distributed.run on 127.0.0.1 with a free port and hand its exit code back.  Rank 0's JSON line goes to our stdout untouched."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    return subprocess.run(self_launch_cmd(n, sys.argv[1:], port), env=env).returncode


def roofline_block(dsum, dt, with_pmc):
    """`roofline` of the dominant kernel family from the HIP-event records of a timed region (KernelTimer.summary): MFMA FLOPs issued by its
    launches / their summed duration, against the 2.5 PFLOP/s dense 16-bit MFMA peak."""
    fam = [n for n in KernelTimer.DOMINANT if n in dsum]
    if not fam:
        return None
    dom = {k: sum(dsum[n][k] for n in fam) for k in ("seconds", "flops", "launches", "mfma_flops")}
    ach = dom["flops"] / dom["seconds"] / 1e12
    issued = dom["mfma_flops"] / dom["seconds"] / 1e12
    pm = pmc_mfma() if with_pmc else None
    return {"bound": "mfma", "kernel": " + ".join(fam) + " (7x7 first-layer head convs, forward + input gradient)",
            "achieved": issued, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": issued / MFMA_BF16_PEAK_TFLOPS,
            "achieved_algorithmic": ach, "frac_algorithmic": ach / MFMA_BF16_PEAK_TFLOPS,
            "products_per_multiply": dom["mfma_flops"] / dom["flops"],
            "traffic": pmc_traffic() if with_pmc else None, "avg_launch_ms": 1e3 * dom["seconds"] / dom["launches"],
            "launches": dom["launches"], "share_of_step": dom["seconds"] / dt,
            "per_kernel": {n: {"launches": dsum[n]["launches"], "avg_launch_ms": 1e3 * dsum[n]["seconds"] / dsum[n]["launches"],
                               "mfma_issued_tflops": dsum[n]["mfma_flops"] / dsum[n]["seconds"] / 1e12} for n in fam},
            "build": _build_id(),
            "pmc": pm}


ROOFLINE_NOTE = ("achieved = 16-bit MFMA FLOPs ISSUED by the launches of this kernel family / their HIP-event time, measured inside the timed "
                 "region: algorithmic fp32 conv FLOPs (2*N*H*W*Cout*49*Cin, real channel counts) x the plane products each multiply is "
                 "evaluated as (`products_per_multiply`, launch-weighted: 3 in the forward launches, 1 in the input-gradient launches of the "
                 "default policy); peak = 2500 TFLOP/s dense f16 / bf16 MFMA; frac = achieved / peak.  achieved_algorithmic / "
                 "frac_algorithmic = the same launches in fp32-equivalent conv FLOPs (no plane products).  traffic = HBM bytes per launch "
                 "from the committed PMC pass of this build, null when the loaded libraries differ from the profiled ones.  pmc = rocprofv3 PMC "
                 "pass of this build (profiles/, build hash checked): MFMA-pipe busy cycles / active cycles of these kernels, collected by "
                 "tools/profile_round.sh on another box run")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--boxes", type=int, default=300)
    ap.add_argument("--mode", choices=["train", "eval", "gt"], default="train",
                    help="eval: inference path (BASELINE configs[4]); gt: ground-truth map generation (SURVEY 8f N1); 1 GPU")
    ap.add_argument("--precision", default=os.environ.get("KG_PRECISION", "fp32"),
                    help="precision policy of the network (engine.PRECISIONS); the headline is the default policy (fp32-tolerance forward on hi + lo "
                         "half planes, single-plane half backward); at 1 GPU two companions are timed beside it: `half` (half mixed precision) and "
                         "`fp32b2` (hi + lo half planes in the backward pass as well)")
    ap.add_argument("--no-companion", action="store_true", help="skip the `half` and `fp32b2` companion measurements")
    ap.add_argument("--profile-run", action="store_true", help="warmup + timed steps only (no companion, no second read-back policy, "
                    "no per-kernel pass, no CPU baseline): the command rocprofv3 wraps, so that steps + warmup launches are traced")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--details", default=os.environ.get("KG_BENCH_DETAILS", ""),
                    help="file for the long-form record (per-kernel table, per-step times, companions with their roofline blocks, notes); default "
                         "gpurun_out/bench_details.json when that directory exists, else stderr.  The stdout JSON line stays under 3 KB")
    args = ap.parse_args()
    if args.profile_run:
        args.no_companion = args.no_cpu_baseline = args.no_kernel_timer = True

    from kg_instance_segmentation_amd import KGnet, parallel
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))       # plain `python bench.py --gpus N`: spawn the N ranks ourselves
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # host threads: 8 ranks x all-cores OpenMP pools would oversubscribe the node during the host-side glue
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // max(world, 1))))
    if args.mode == "eval":
        if rank == 0:
            eval_bench(args, dev)
        return
    if args.mode == "gt":
        if rank == 0:
            gt_bench(args, dev)
        return
    x, gt, gt_masks, gt_boxes = make_batch(args.batch, args.size, args.boxes, 100 + rank, dev)
    den = parallel.detection_denominators(gt) if world > 1 else None
    timer = KernelTimer()
    if not args.no_kernel_timer:
        timer.install()
    ldec, lseg = DetectionLossAll(kp_radius=5), SEG_loss(height=args.size, width=args.size)

    def run_train(precision, steps, warmup, sync_each_step, kernel_timer):
        """K timed train steps of a fresh seeded model in `precision`; returns (seconds for the K steps = max over ranks, last loss,
        per-step losses, dominant-kernel records)."""
        torch.manual_seed(1234)
        model = KGnet.resnet50(pretrained=False, precision=precision).to(dev).train()
        parallel.broadcast_parameters(model)
        # train.py:71 (torch.optim.Adam is caller code; `fused=True` selects PyTorch's single-kernel multi-tensor implementation)
        from kg_instance_segmentation_amd.optim import Adam      # the build's one-launch Adam (kg_adam_step), torch.optim.Adam's update rule
        opt = Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-4, prepack=model)     # (prepack: packs the next forward's weights behind the update kernel)
        reducer = parallel.FlatGradReducer().attach(model) if world > 1 else None

        def step(sync):
            opt.zero_grad()
            p0, p1, p2, p3, pred = model(x, gt_boxes)
            if den is None:
                l1 = ldec(p0, gt[0]) + ldec(p1, gt[1]) + ldec(p2, gt[2]) + ldec(p3, gt[3])
            else:
                l1 = sum(ldec(p, g, denominators=den[i]) for i, (p, g) in enumerate(zip((p0, p1, p2, p3), gt)))
            l2 = lseg(pred, gt_masks, gt_boxes)
            loss = l1 if l2 is None else l1 + l2 / world
            loss.backward()
            if reducer is not None:
                reducer.finish()
            opt.step()
            return loss.item() if sync else loss.detach()      # train.py:156 reads the loss back every step

        def timed(K, sync, park_gc=True):
            if not park_gc:       # (what the drop-in train.py does: the cyclic collector stays on -- reported beside the headline)
                return timed_(K, sync)
            # the K timed steps run with Python's cyclic collector parked (gc.freeze + disable, as training loops at scale do: a gen-2
            # pass over the model's ~10^5 objects is a multi-ms host stall at a random step); the step itself leaves no cycles behind
            # (tools/cycle_probe.py), so memory does not grow meanwhile
            gc.collect(); gc.freeze(); gc.disable()
            try:
                return timed_(K, sync)
            finally:
                gc.enable(); gc.unfreeze()

        def timed_(K, sync):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ls, mk = [], []
            for _ in range(K):
                ls.append(step(sync))
                mk.append(time.perf_counter())
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt_ = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt_], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_ = float(t.item())
            return dt_, [float(v) for v in ls], [t0] + mk

        for _ in range(warmup):
            step(sync_each_step)
        timer.on, timer.only_dominant, timer.rec = kernel_timer, True, []
        dt_, ls, mk = timed(steps, sync_each_step)
        timer.on = False
        res = {"dt": dt_, "losses": ls, "marks": mk, "dom_rec": timer.rec, "step": step, "timed": timed,
               "overflow": (lambda: model.grad_overflowed(reset=False))}
        timer.rec = []
        return res

    # Headline: the K timed steps read the loss back EVERY step like train.py:156 (`running_loss += loss.item()`); KG_BENCH_SYNC=0
    # defers the read-back past the timed region instead (the host then enqueues step k+1 while the GPU still runs step k).
    STEP_SYNC = os.environ.get("KG_BENCH_SYNC", "1") == "1"
    main_run = run_train(args.precision, args.steps, args.warmup, STEP_SYNC, not args.no_kernel_timer)
    dt, losses, marks = main_run["dt"], main_run["losses"], main_run["marks"][1:]
    t0 = main_run["marks"][0]
    last = losses[-1]
    dom_rec = main_run["dom_rec"]
    other_dt = dt
    gc_on_dt = None
    if not args.profile_run:
        other_dt, _, _ = main_run["timed"](args.steps, not STEP_SYNC)      # the other loss-read-back policy, reported as a note
        gc_on_dt, _, _ = main_run["timed"](args.steps, STEP_SYNC, park_gc=False)   # the same steps with Python's cyclic GC left enabled
    overflowed = bool(main_run["overflow"]())
    prof_steps = 0
    if not args.no_kernel_timer:      # per-kernel breakdown: extra, untimed steps with events around every conv launch
        timer.on, timer.only_dominant, prof_steps = True, False, 2
        for _ in range(prof_steps):
            main_run["step"](True)
        torch.cuda.synchronize()
        timer.on = False
    prof_rec, timer.rec = timer.rec, []
    companions = {}
    if world == 1 and not args.no_companion:
        main_run = None
        torch.cuda.empty_cache()
        notes = {"half": "NOT the reference's arithmetic: half mixed precision, tests/test_gpu_parity.py holds it to rtol 2e-2 + atol 2e-2 rms",
                 "fp32b2": "hi + lo half planes (3 MFMA products per multiply) in the forward AND the backward pass: the policy whose parameter gradients sit on the "
                           "fp32 reference's own noise floor against a float64 evaluation (profiles/r04_grad_table.json, tests/test_gpu_gradprec.py)"}
        for cp in ("fp32b2", "half"):
            if cp == args.precision:
                continue
            # fp32b2 -- the step at the reference's precision in BOTH directions -- is timed on equal footing with the headline: the same
            # --steps / --warmup and its own roofline block; `half` (not the reference's arithmetic) stays a short companion
            ksteps, kwarm = (args.steps, args.warmup) if cp == "fp32b2" else (max(2, min(args.steps, 10)), 3)
            comp = run_train(cp, ksteps, kwarm, STEP_SYNC, cp == "fp32b2" and not args.no_kernel_timer)
            companions[cp] = {"precision": cp, "dtype": dtype_label(cp), "value": args.batch * ksteps / comp["dt"], "unit": "imgs/s",
                              "ms_per_step": 1e3 * comp["dt"] / ksteps, "steps": ksteps, "warmup": kwarm, "last_loss": comp["losses"][-1],
                              "grad_overflow": bool(comp["overflow"]()), "parity": notes[cp]}
            if comp["dom_rec"]:
                companions[cp]["roofline"] = roofline_block(timer.summary(comp["dom_rec"]), comp["dt"], with_pmc=False)
            comp = None
            torch.cuda.empty_cache()
    if rank != 0:
        return
    if os.environ.get("KG_BENCH_VERBOSE"):
        print("per-step host ms:", [round(1e3 * (b - a), 1) for a, b in zip([t0] + marks[:-1], marks)], file=sys.stderr)
        print("losses:", [round(l, 2) for l in losses], file=sys.stderr)
    imgs = args.batch * world * args.steps
    from kg_instance_segmentation_amd import engine as kengine
    pol = kengine.PRECISIONS[args.precision]
    PDESC = {"fp32": "fp32-faithful: every forward tensor = hi + lo IEEE-half planes (22 significant bits), 3 f16 MFMA products per multiply, fp32 "
                     "accumulation (within rtol 1e-4 / atol 1e-5 of the reference on pre-sigmoid logits, eval and train mode); backward on single half "
                     "planes with a per-step power-of-two gradient scale chosen on the device (every parameter gradient: cosine >= 0.9999, norm "
                     "within 2e-3 of the reference)",
             "fp32b2": "as fp32 with hi + lo half planes (3 products) in the backward pass as well",
             "half": "half mixed precision: single IEEE-half planes everywhere (f16 MFMA, fp32 accumulation)",
             "halfmix": "as half, with the BatchNorm backbone (stem conv1, layer1-3) on hi + lo half planes",
             "fp32bf": "fp32 values as hi + mid + lo bf16 planes (exact), 6 bf16 MFMA products per multiply; backward on hi + lo planes (3 products)",
             "fp32bf_full": "as fp32bf with three planes (6 products) in the backward pass as well",
             "mixed": "bf16 MFMA, fp32 accumulation; the BatchNorm backbone (stem conv1, layer1-3) stored and multiplied as hi + lo bf16 planes "
                      "(3 products), c0_conv / decoder / 7x7 heads / seg branch single-plane bf16",
             "trunk2": "as mixed, with c0_conv and the decoder in hi + lo planes as well",
             "bf16": "bf16 MFMA, fp32 accumulation, single-plane bf16 storage everywhere"}
    out_dtype = dtype_label(args.precision)
    out = {"metric": "imgs/s (train fwd+bwd) at 512x512", "value": imgs / dt, "unit": "imgs/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": out_dtype, "data": "synthetic",
           "config": {"workload": f"KGnet train step fwd(dec+seg)+4xDetectionLossAll+SEG_loss+bwd+Adam, batch {args.batch}/GPU 3x{args.size}x{args.size}, "
                                  f"{args.boxes} boxes/img, full HIP path",
                      "precision_policy": args.precision, "planes": list(pol),
                      "global_batch": args.batch * world, "parallelism": f"dp{world}", "last_loss": last,
                      "loss_readback": "every step (train.py:156)" if STEP_SYNC else "after the timed region",
                      "other_readback_policy_imgs_per_s": imgs / other_dt,
                      "gc_enabled_imgs_per_s": (imgs / gc_on_dt) if gc_on_dt else None,
                      "grad_overflow": overflowed}}     # sticky device flag of the half-precision backward (KGnet.grad_overflowed): must be false
    # long-form record (side file, not the stdout line -- the driver keeps `config` scalars and cuts long strings / nested tables)
    details = {"precision": PDESC.get(args.precision, args.precision), "roofline_note": ROOFLINE_NOTE,
               "step_ms": [round(1e3 * (b - a), 2) for a, b in zip([t0] + marks[:-1], marks)]}
    for cp, comp in companions.items():
        details[cp + "_companion"] = comp
        # the companions as SCALARS of `config`: fp32b2 = the step at the reference's precision in both directions (train.py:153 is fp32 autograd)
        out["config"][cp + "_imgs_per_s"] = comp["value"]
        out["config"][cp + "_ms_per_step"] = comp["ms_per_step"]
        if comp.get("roofline"):
            out["config"][cp + "_roofline_frac"] = comp["roofline"]["frac"]
        out["config"][cp + "_grad_overflow"] = comp["grad_overflow"]
    if not args.no_kernel_timer:
        if os.environ.get("KG_BENCH_DUMP"):
            timer.rec = prof_rec
            timer.dump(os.environ["KG_BENCH_DUMP"], prof_steps)
        rb = roofline_block(timer.summary(dom_rec), dt, with_pmc=True)
        if rb:
            out["roofline"] = rb
        summ = timer.summary(prof_rec)
        if summ:
            details["kernels"] = {k: {"ms_per_step": 1e3 * v["seconds"] / prof_steps, "tflops": v["flops"] / max(v["seconds"], 1e-12) / 1e12,
                                  "launches_per_step": v["launches"] / prof_steps} for k, v in summ.items()}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size, args.boxes)      # batch 1 with the headline's boxes per image: the same per-image work
        out["config"]["grouping_match_rate"] = out["cpu_baseline"]["grouping_match_rate"]
    line = json.dumps(out)
    details["line"] = out
    import tempfile
    path = args.details or ("gpurun_out/bench_details.json" if os.path.isdir("gpurun_out") else os.path.join(tempfile.gettempdir(), "kg_bench_details.json"))
    try:
        with open(path, "w") as f:
            json.dump(details, f, indent=1)
        out["config"]["details_file"] = path
        line = json.dumps(out)
    except OSError as e:
        print(f"bench details not written ({e})", file=sys.stderr)
    print(line)


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()

"""CPU: the C-ABI library builds, loads and exports every symbol include/kgnet_hip.h declares; the
host-side glue (crop rectangles, loss matching, module keys) behaves like the reference's.  No
compute entry point is called here (there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from kg_instance_segmentation_amd import build
    return ctypes.CDLL(build.build())


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "kgnet_hip.h")).read()
    names = set(re.findall(r"\b(kg_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header(lib):
    from kg_instance_segmentation_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "kgnet_hip.h")).read()
    names = set(re.findall(r"\b(kg_[a-z0-9_]+)\s*\(", hdr))
    assert set(_lib.SYMBOLS) == names
    # argument counts of the ctypes table agree with the header prototypes
    for m in re.finditer(r"\b(?:int|long|const char\*)\s+(kg_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr, re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        if name in _lib._SIGS:
            assert len(_lib._SIGS[name]) == n, (name, len(_lib._SIGS[name]), n)


def test_version_and_error_channel(lib):
    lib.kg_last_error.restype = ctypes.c_char_p
    assert lib.kg_version() >= 100
    # argument validation happens on the host before any launch
    rc = lib.kg_conv2d_igemm(None, None, None, None, None, None, None, None, *([0] * 21), None)
    assert rc != 0 and b"kg_conv2d_igemm" in lib.kg_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from kg_instance_segmentation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libkgnet_hip.so")
    with pytest.raises(_lib.KGLibraryError):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "kg_instance_segmentation_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "/root/reference" not in src, f


def test_module_keys_and_cpu_refusal(state_dict0):
    from kg_instance_segmentation_amd import KGnet, _lib
    m = KGnet.resnet50(pretrained=False)
    assert list(m.state_dict().keys()) == list(state_dict0.keys())
    m.load_state_dict(state_dict0)
    assert len(list(m.parameters())) == 217
    with pytest.raises(_lib.KGLibraryError):   # no silent CPU fallback
        m.forward_dec(torch.zeros(1, 3, 64, 64))


def test_crop_rects_match_oracle():
    from kg_instance_segmentation_amd.seg import crop_rects
    from oracle.net import Net
    rng = np.random.default_rng(3)
    sizes = [(96, 128), (48, 64), (24, 32), (12, 16), (6, 8)]
    boxes = np.concatenate([rng.uniform(0, 100, (200, 4)), [[10.5, 12.5, 40.5, 50.5], [0, 0, 95, 127], [30.5, 60.5, 37.5, 71.5],
                                                            [50, 20, 52, 90], [2.5, 3.5, 14.5, 17.5]]]).astype(np.float32)
    boxes[:, 2:] = np.maximum(boxes[:, 2:], boxes[:, :2] + rng.uniform(0, 60, (len(boxes), 2)).astype(np.float32))
    rects, depth = crop_rects(boxes, 96, 128, sizes)
    for b in range(len(boxes)):
        d = 0
        for l, (h, w) in enumerate(sizes):
            cc = Net.crop_coords(boxes[b], 96, 128, h, w)
            if cc is None:
                break
            assert tuple(rects[l][b]) == cc
            d += 1
        assert d == depth[b]


def test_seg_loss_host_matching_matches_oracle(golden):
    from kg_instance_segmentation_amd.seg_loss import jaccard_numpy, nearest_resize
    from oracle import net as onet
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.uniform(0, 50, 4).astype(np.float32); b = rng.uniform(0, 50, 4).astype(np.float32)
        a[2:] += a[:2]; b[2:] += b[:2]
        assert jaccard_numpy(a, b) == onet.jaccard(a, b)
    m = (rng.random((13, 17)) > 0.5).astype(np.float32)
    assert np.array_equal(nearest_resize(m, 7, 9), onet.nearest_resize(m, 7, 9))
    assert nearest_resize(m, 13, 17) is m


def test_vectorised_matcher_equals_scalar():
    from kg_instance_segmentation_amd.seg_loss import jaccard_matrix, jaccard_numpy
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 50, (40, 4)).astype(np.float32); a[:, 2:] += a[:, :2]
    b = rng.uniform(0, 50, (30, 4)).astype(np.float32); b[:, 2:] += b[:, :2]
    b[:5] = a[:5]; b[5] = [0, 0, 1, 1]
    m = jaccard_matrix(a, b)
    assert all(np.float32(m[i, j]) == np.float32(jaccard_numpy(a[i], b[j])) for i in range(40) for j in range(30))


def test_native_matcher_equals_scalar_rule(lib):
    """kg_host_match_boxes (pure host code inside the library) returns exactly the (patch, gt) pairs with jaccard_numpy >= 0.5
    (seg_loss.py:14-29, 55-56), in row-major order -- near-threshold overlaps, tiny unions (<= 2 -> 0), empty inputs, dense overlaps."""
    from kg_instance_segmentation_amd.seg_loss import jaccard_numpy, match_boxes
    rng = np.random.default_rng(2)
    for trial in range(12):
        P, G = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        a = rng.uniform(0, 80, (P, 4)).astype(np.float32); a[:, 2:] += a[:, :2]
        b = rng.uniform(0, 80, (G, 5)).astype(np.float32); b[:, 2:4] += b[:, :2]
        k = min(P, G)
        b[:k, :4] = a[:k] + rng.choice([0.0, 0.3, 1.5, 6.0], (k, 1)).astype(np.float32)     # overlaps around the 0.5 threshold
        if G > 2:
            b[-1, :4] = [1, 1, 2, 2]; b[-2, :4] = [1, 1, 2.5, 2.2]                              # union <= 2
        js, gs = match_boxes(a, b)
        ref = [(i, j) for i in range(P) for j in range(G) if jaccard_numpy(a[i], b[j, :4]) >= 0.5]
        assert list(zip(js.tolist(), gs.tolist())) == ref, trial
    a = np.tile(np.array([[10, 10, 40, 40]], np.float32), (30, 1))
    js, gs = match_boxes(a, np.concatenate([a, np.ones((30, 1), np.float32)], 1))             # all 900 pairs match: the pair buffer grows
    assert len(js) == 900 and js[31] == 1 and gs[31] == 1


def test_native_seg_tables_equal_the_numpy_formulation(lib):
    """kg_host_tile_table / kg_host_bin_csr (host glue of seg.make_plan) against the NumPy repeat / stable-argsort formulation they
    replace: same tile entries in the same order, same CSR lists (boxes ascending inside a bin), incl. empty levels and 1-pixel boxes."""
    import ctypes
    rng = np.random.default_rng(5)

    def vp(a):
        return ctypes.c_void_p(a.ctypes.data)
    for trial in range(8):
        nb = int(rng.integers(0, 200)) if trial else 0
        H, W, nimg = 128, 96, 3
        y1 = rng.integers(0, H - 1, nb); x1 = rng.integers(0, W - 1, nb)
        h = np.minimum(rng.integers(1, 70, nb), H - y1).astype(np.int32); w = np.minimum(rng.integers(1, 50, nb), W - x1).astype(np.int32)
        img = np.sort(rng.integers(0, nimg, nb)).astype(np.int32)
        row0 = np.zeros(nb + 1, np.int64); np.cumsum(h.astype(np.int64) * w, out=row0[1:])
        for th, tw in ((16, 32), (16, 16)):
            ny, nx = (h + th - 1) // th, (w + tw - 1) // tw
            cnt = ny * nx
            b = np.repeat(np.arange(nb), cnt)
            k = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            ty, tx = (k // nx[b], k % nx[b]) if nb else (k, k)
            ref = np.stack([row0[:-1][b], (h[b].astype(np.int64) << 16) | w[b], ((ty * th).astype(np.int64) << 16) | (tx * tw),
                            np.zeros(len(b), np.int64)], 1).astype(np.int32)
            out = np.empty((int(cnt.sum()), 4), np.int32)
            got = lib.kg_host_tile_table(vp(h), vp(w), vp(np.ascontiguousarray(row0[:-1])), nb, th, tw, vp(out), len(out))
            assert got == len(ref) and np.array_equal(out, ref)
            assert nb == 0 or lib.kg_host_tile_table(vp(h), vp(w), vp(np.ascontiguousarray(row0[:-1])), nb, th, tw, vp(out), len(out) - 1) == -1
        tab = np.stack([img, y1, x1, h, w, row0[:-1], np.full(nb, H), np.full(nb, W)], 1).astype(np.int32) if nb else np.zeros((0, 8), np.int32)
        for BS in (16, 8, 4):
            BY, BX = (H + BS - 1) // BS, (W + BS - 1) // BS
            nbins = nimg * BY * BX
            by0, bx0 = tab[:, 1] // BS, tab[:, 2] // BS
            nyb = (tab[:, 1] + tab[:, 3] - 1) // BS - by0 + 1; nxb = (tab[:, 2] + tab[:, 4] - 1) // BS - bx0 + 1
            cnt = (nyb * nxb).astype(np.int64)
            b = np.repeat(np.arange(nb), cnt)
            k = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            bins = (tab[b, 0].astype(np.int64) * BY + by0[b] + k // nxb[b]) * BX + bx0[b] + k % nxb[b] if nb else np.zeros(0, np.int64)
            order = np.argsort(bins, kind="stable")
            ref_st = np.zeros(nbins + 1, np.int64); np.cumsum(np.bincount(bins, minlength=nbins), out=ref_st[1:])
            st = np.empty(nbins + 1, np.int32); bb = np.empty(int(cnt.sum()), np.int32)
            got = lib.kg_host_bin_csr(vp(np.ascontiguousarray(tab)), nb, BS, BY, BX, nbins, vp(st), vp(bb), len(bb))
            assert got == len(bb) and np.array_equal(st, ref_st.astype(np.int32)) and np.array_equal(bb, b[order].astype(np.int32))


def test_dropin_shims_expose_the_reference_module_surface():
    """dropin/<module>.py (what `import KGnet` etc. resolve to when dropin/ precedes the reference on sys.path)
    re-export the symbols the reference drivers use (train.py:3-11, test.py:3-12, dataset_base.py:6)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = {"KGnet": ["resnet50", "ResNet"], "loss": ["DetectionLossAll"], "seg_loss": ["SEG_loss"],
            "postprocessing": ["get_skeletons_and_masks", "refine_skeleton", "gather_skeleton"],
            "nms": ["non_maximum_suppression_numpy"], "config": ["EDGES", "NUM_KPS", "KP_RADIUS"],
            "preprocessing": ["get_ground_truth", "create_position_index"],
            "eval_parts": ["mask_iou", "voc_ap", "bbox_evaluation", "seg_evaluation"]}
    for name, syms in want.items():
        # `preprocessing` is opt-in (dropin/optin/, KG_GPU_GT=1): the reference calls it inside DataLoader workers, which cannot use the GPU
        sub = "optin" if name == "preprocessing" else ""
        assert os.path.exists(os.path.join(root, "dropin", f"{name}.py")) == (name != "preprocessing")
        spec = importlib.util.spec_from_file_location(f"_dropin_{name}", os.path.join(root, "dropin", sub, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for sym in syms:
            assert hasattr(mod, sym), (name, sym)


def test_preprocessing_position_index_matches_reference_layout():
    from kg_instance_segmentation_amd import preprocessing as kprep
    p = kprep.create_position_index(3, 5)
    assert p.shape == (3, 5, 2) and tuple(p[2, 4]) == (4, 2)      # (x, y) per pixel, preprocessing.py:4-11


def test_host_crop_masks_matches_numpy_rule(lib):
    """kg_host_crop_masks (host glue of SEG_loss, seg_loss.py:57-80) == NumPy slicing + the nearest-resize rule, incl. crops
    whose size differs from the patch (clamped boxes)."""
    import ctypes
    import numpy as np
    from kg_instance_segmentation_amd import _lib
    from kg_instance_segmentation_amd.seg_loss import nearest_resize
    rng = np.random.default_rng(0)
    H, W = 40, 56
    masks = [(rng.random((3, H, W)) > 0.5).astype(np.float32), (rng.random((2, H, W)) > 0.5).astype(np.float32)]
    work, off = [], 0
    for (i, g, ya, yb, xa, xb, h1, w1) in [(0, 2, 3, 20, 5, 30, 17, 25), (1, 0, 0, 39, 10, 55, 40, 46), (0, 0, 7, 9, 7, 12, 5, 9), (1, 1, 30, 39, 0, 8, 9, 8)]:
        work.append([i, g, ya, yb, xa, xb, h1, w1, off]); off += h1 * w1
    work = np.asarray(work, np.int32)
    out = np.full(off, 255, np.uint8)
    ptrs = (ctypes.c_void_p * 2)(*[m.ctypes.data for m in masks])
    _lib.call("kg_host_crop_masks", ctypes.cast(ptrs, ctypes.c_void_p), ctypes.c_void_p(work.ctypes.data), len(work), H, W, ctypes.c_void_p(out.ctypes.data))
    for (i, g, ya, yb, xa, xb, h1, w1, o) in work.tolist():
        ref = nearest_resize(masks[i][g][ya:yb, xa:xb], h1, w1)
        assert np.array_equal(out[o:o + h1 * w1].reshape(h1, w1), ref.astype(np.uint8))


def test_dropin_launcher_shadows_the_scripts_own_modules(tmp_path):
    """dropin/run.py: a driver started through it imports the shims even though a same-named module sits next to the script
    (the situation of train.py / test.py inside the reference checkout)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "config.py").write_text("WHO = 'decoy next to the script'\n")
    (tmp_path / "collater.py").write_text("WHO = 'module that is not replaced'\n")
    (tmp_path / "driver.py").write_text("import sys, config, collater\nprint(config.__file__); print(collater.WHO); print(sys.argv[1:])\n")
    r = subprocess.run([sys.executable, os.path.join(root, "dropin", "run.py"), "driver.py", "--flag", "7"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == os.path.join(root, "dropin", "config.py") and lines[1] == "module that is not replaced" and lines[2] == "['--flag', '7']"


def test_pretrained_loads_a_torchvision_keyed_checkpoint(tmp_path, monkeypatch):
    """KGnet.resnet50(pretrained=True) (KGnet.py:377-386: load_state_dict(torchvision resnet50, strict=False)): a checkpoint with
    torchvision's key set -- conv1 / bn1 / layer1..4 / fc, i.e. MORE than KGnet builds (layer4, fc) and LESS than it has (decoder,
    heads, seg branch) -- loads through KG_RESNET50_PTH: shared keys are overwritten, the rest keeps its init, nothing raises."""
    import torch
    from kg_instance_segmentation_amd import KGnet
    from oracle import weightgen
    torch.manual_seed(0)
    ref = KGnet.resnet50(pretrained=False)
    own = {k: v.clone() for k, v in ref.state_dict().items()}
    tv = {}
    src = weightgen.gen_state_dict(7)
    for k, v in src.items():                     # the trunk keys KGnet shares with torchvision's resnet50
        if k.startswith(("conv1.", "bn1.", "layer1.", "layer2.", "layer3.")):
            tv[k] = v
    tv["layer4.0.conv1.weight"] = torch.randn(512, 1024, 1, 1)      # keys KGnet does not have
    tv["fc.weight"] = torch.randn(1000, 2048); tv["fc.bias"] = torch.randn(1000)
    path = tmp_path / "resnet50-tv.pth"
    torch.save(tv, path)
    monkeypatch.setenv("KG_RESNET50_PTH", str(path))
    torch.manual_seed(0)
    m = KGnet.resnet50(pretrained=True)
    sd = m.state_dict()
    assert len(sd) == 346
    for k, v in sd.items():
        if k in tv:
            assert torch.equal(v, tv[k]), k
        else:
            assert torch.equal(v, own[k]), k       # same seed -> same init for everything the checkpoint does not cover
    monkeypatch.delenv("KG_RESNET50_PTH")
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        KGnet.resnet50(pretrained=True)
    assert any("KG_RESNET50_PTH" in str(x.message) for x in w)


def test_half_build_loads_and_exports_the_rows_entry_points(lib):
    """libkgnet_hip_f16.so: the same C ABI built for IEEE-half rows (include/kgnet_hip.h): loads, identifies itself, and exports
    every entry point that takes rows / packed-weight operands; the format-independent ones live in libkgnet_hip.so only."""
    from kg_instance_segmentation_amd import _lib, build, ops
    build.build()
    half = _lib.load(1)
    assert half.kg_rows_format() == 1 and lib.kg_rows_format() == 0
    rows_entries = ["kg_conv2d_igemm", "kg_conv2d_halo", "kg_conv7_narrow", "kg_conv2d_halo_heads2", "kg_conv3x3_c64", "kg_conv1x1", "kg_pack_weight",
                    "kg_pack_weight_rows", "kg_pack_weight_batch", "kg_pack_weight_narrow", "kg_im2col_small", "kg_conv2d_wgrad", "kg_conv2d_wgrad_halo", "kg_bias_grad",
                    "kg_img_pack", "kg_bn_stats_train", "kg_bn_apply", "kg_bn_bwd", "kg_maxpool3s2_fwd", "kg_maxpool3s2_bwd", "kg_bilinear_fwd",
                    "kg_bilinear_bwd", "kg_add_rows", "kg_grad_pack", "kg_grad_pack3", "kg_rows_gather", "kg_rows_gather_planes", "kg_rows_gather_f32", "kg_planes_to_f32", "kg_f32_to_planes",
                    "kg_crop_grad_reduce", "kg_rows_rescale", "kg_rows_scale", "kg_conv_stats_begin", "kg_conv_stats_end", "kg_last_error", "kg_last_kernel"]
    missing = [n for n in rows_entries if not hasattr(half, n)]
    assert not missing, missing
    assert not hasattr(half, "kg_postproc_scale") and not hasattr(half, "kg_adam_step")
    # kg_planes_t: 10 ints + a device pointer (header, csrc/kg_common.h and the ctypes mirror agree on the layout)
    assert ctypes.sizeof(ops._Planes) == 56 and ops._Planes.scale.offset == 40 and ops._Planes.oscale.offset == 48
    hdr = open(os.path.join(ROOT, "include", "kgnet_hip.h")).read()
    assert "int reserved_;" in hdr and "const float* scale;" in hdr


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without WORLD_SIZE re-runs itself as the driver's launcher line (bench.self_launch_cmd): one rank per GPU
    under torch.distributed.run on 127.0.0.1, the user's flags passed through."""
    import bench
    cmd = bench.self_launch_cmd(8, ["--gpus", "8", "--steps", "3"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3"]

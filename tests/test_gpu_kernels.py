"""GPU parity tests of the individual HIP kernels, called through the C ABI (ops.py -> libkgnet_hip.so).

Checker: the same torch fp32/fp64 CPU primitives the oracle (oracle/net.py) is written in.  Inputs are
made bf16-representable so that the only differences are accumulation order (fp32) and the final bf16
rounding of stored activations; tolerances below are stated per test."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import _lib, ops  # noqa: E402
from kg_instance_segmentation_amd.ops import BF16, PackedWeight  # noqa: E402

DEV = "cuda"


def bfr(t):
    """round to bf16-representable fp32"""
    return t.to(BF16).float()


def rows_of(x):  # NCHW fp32 -> bf16 rows [N*H*W, C]
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).to(BF16).contiguous()


def nchw_of(rows, n, h, w):
    return rows.float().view(n, h, w, -1).permute(0, 3, 1, 2)


def report(name, got, ref, atol, rtol):
    got = got.detach().double().cpu(); ref = ref.detach().double().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = int((err > tol).sum())
    idx = int(err.argmax())
    print(f"[{name}] max_abs_err={float(err.max()):.3e} at flat {idx} (got {float(got.flatten()[idx]):.6g} ref {float(ref.flatten()[idx]):.6g}) "
          f"ref_absmax={float(ref.abs().max()):.3e} bad={bad}/{err.numel()}")
    if bad:
        bi = torch.nonzero((err > tol).flatten())[:8].flatten().tolist()
        print("   first bad:", [(i, float(got.flatten()[i]), float(ref.flatten()[i])) for i in bi])
    assert bad == 0, name


def test_library_and_arch():
    lib = _lib.load()
    buf = ctypes.create_string_buffer(128)
    assert lib.kg_device_arch(buf, 128) == 0
    print("arch:", buf.value.decode())
    assert buf.value.decode().startswith("gfx950")


def test_tr_read_semantics():
    """ds_read_b64_tr_b16: within a 16-lane group lane i supplies 4 contiguous b16 (row i>>2, cols 4*(i&3)..)
    and receives column i of the 4x16 block (rows 0..3).  conv_wgrad.hip relies on exactly this."""
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    _lib.call("kg_tr_probe", _lib.ptr(out), _lib.stream_ptr())
    got = out.cpu().numpy().astype(np.int64).reshape(64, 4)
    exp = np.zeros((64, 4), np.int64)
    for l in range(64):
        G, i = l // 16, l % 16
        for j in range(4):
            exp[l, j] = (G * 16 + j * 4 + (i >> 2)) * 4 + (i & 3)
    if not np.array_equal(got, exp):
        print("tr probe mapping (lane: got | expected):")
        for l in range(64):
            print(l, got[l].tolist(), exp[l].tolist())
    assert np.array_equal(got, exp)


def test_f64_primitives_bit_exact():
    """fp64 div / sqrt / fma / floor / ceil on the GPU must round exactly like the host's IEEE arithmetic."""
    rng = np.random.default_rng(0)
    n = 1 << 16
    a = rng.normal(size=n) * 10 ** rng.uniform(-3, 3, n)
    b = rng.normal(size=n) * 10 ** rng.uniform(-3, 3, n)
    a[:8] = [0.5, 1.5, 2.5, -0.5, 36.0, 100.0, 1e-300, 3.0]
    ad, bd = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
    out = torch.empty(5 * n, dtype=torch.float64, device=DEV)
    _lib.call("kg_f64_probe", _lib.ptr(ad), _lib.ptr(bd), _lib.ptr(out), n, _lib.stream_ptr())
    o = out.cpu().numpy().reshape(5, n)
    libm = ctypes.CDLL("libm.so.6"); libm.fma.restype = ctypes.c_double; libm.fma.argtypes = [ctypes.c_double] * 3
    fm = np.array([math.sqrt(libm.fma(x, x, y * y)) for x, y in zip(a, b)])
    for name, got, ref in (("div", o[0], a / b), ("sqrt", o[1], np.sqrt(np.abs(a))), ("norm_fma", o[2], fm),
                           ("floor", o[3], np.floor(a)), ("ceil", o[4], np.ceil(a))):
        nbad = int((got != ref).sum())
        print(f"[f64 {name}] mismatches {nbad}/{n}")
        assert nbad == 0, name


CONV_CASES = [
    # cin, cout, k, stride, pad, N, H, W, relu, bias, tile
    (64, 64, 3, 1, 1, 2, 20, 28, True, True, 0),
    (64, 64, 7, 1, 3, 1, 24, 40, True, True, 0),
    (64, 192, 7, 1, 3, 1, 16, 24, True, True, 0),
    (128, 64, 1, 1, 0, 2, 16, 16, True, True, 0),
    (64, 64, 3, 2, 1, 2, 18, 22, False, False, 0),
    (256, 512, 1, 2, 0, 1, 16, 24, False, False, 0),
    (3, 64, 7, 2, 3, 2, 32, 40, False, False, 0),
    (3, 64, 3, 1, 1, 1, 24, 24, True, True, 0),
    (256, 256, 3, 1, 1, 1, 12, 20, True, True, 4),
    (64, 64, 3, 1, 1, 1, 16, 16, False, True, 5),
    (1024, 512, 3, 1, 1, 1, 8, 8, True, True, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(case):
    cin, cout, k, stride, pad, N, H, W, relu, bias, tile = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g) if bias else None
    ref = F.conv2d(x.double(), w.double(), b.double() if bias else None, stride, pad)
    if relu:
        ref = F.relu(ref)
    cin_pad = ops.round_up(cin, 8)
    xr = torch.zeros(N * H * W, cin_pad, dtype=BF16)
    xr[:, :cin] = rows_of(x)
    xr = xr.to(DEV)
    pw = PackedWeight(cout, k * k, cin_pad, DEV)
    pw.pack(w.to(DEV))
    OH, OW = ref.shape[2:]
    M = N * OH * OW
    y = torch.empty(M, cout, dtype=BF16, device=DEV)
    geom = (M, H, W, OH, OW, k, k, stride, pad)
    ops.conv_igemm(xr, pw, cout, geom, y=y, bias=b.to(DEV) if bias else None, relu=relu, tile=tile)
    torch.cuda.synchronize()
    report(f"conv_fwd{case}", nchw_of(y.cpu(), N, OH, OW), ref, atol=2e-2, rtol=1e-2)
    # fp32 NCHW export path is not rounded to bf16: tight tolerance
    yf = torch.empty(N, cout, OH, OW, dtype=torch.float32, device=DEV)
    ops.conv_igemm(xr, pw, cout, geom, y_f32=yf, bias=b.to(DEV) if bias else None, relu=relu, tile=tile)
    report(f"conv_fwd_f32{case}", yf.cpu(), ref, atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("cout", [5, 10, 40])
def test_conv_small_cout_f32(cout):
    cin, k, N, H, W = 64, 7, 2, 16, 24
    g = torch.Generator().manual_seed(cout)
    x = bfr(torch.randn(N, 192, H, W, generator=g))   # consume a 64-channel slice of a 192-wide buffer
    w = bfr(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x[:, 64:128].double(), w.double(), b.double(), 1, 3)
    xr = rows_of(x).to(DEV)
    pw = PackedWeight(cout, k * k, cin, DEV)
    pw.pack(w.to(DEV))
    yf = torch.empty(N, cout, H, W, dtype=torch.float32, device=DEV)
    ops.conv_igemm(xr[:, 64:128], pw, cout, (N * H * W, H, W, H, W, k, k, 1, 3), y_f32=yf, bias=b.to(DEV))
    report(f"conv_small_cout{cout}", yf.cpu(), ref, atol=2e-4, rtol=2e-4)


def test_conv_residual_mask_and_slice_output():
    cin, cout, N, H, W = 64, 64, 1, 16, 16
    g = torch.Generator().manual_seed(3)
    x = bfr(torch.randn(N, cin, H, W, generator=g)); w = bfr(torch.randn(cout, cin, 3, 3, generator=g) / 24)
    res = bfr(torch.randn(N, cout, H, W, generator=g)); msk = bfr(torch.randn(N, cout, H, W, generator=g))
    ref = (F.conv2d(x.double(), w.double(), None, 1, 1) + res.double()) * (msk > 0)
    pw = PackedWeight(cout, 9, cin, DEV); pw.pack(w.to(DEV))
    buf = torch.zeros(N * H * W, 128, dtype=BF16, device=DEV)
    ops.conv_igemm(rows_of(x).to(DEV), pw, cout, (N * H * W, H, W, H, W, 3, 3, 1, 1), y=buf[:, 64:128],
                   res=rows_of(res).to(DEV), mask=rows_of(msk).to(DEV))
    report("conv_res_mask", nchw_of(buf[:, 64:128].cpu(), N, H, W), ref, atol=2e-2, rtol=1e-2)
    assert float(buf[:, :64].abs().max()) == 0.0


DGRAD_CASES = [(64, 64, 3, 1, 1, 2, 20, 28), (64, 192, 7, 1, 3, 1, 16, 24), (64, 64, 3, 2, 1, 2, 18, 22),
               (256, 512, 1, 2, 0, 1, 16, 24), (128, 64, 1, 1, 0, 1, 16, 16), (64, 40, 7, 1, 3, 1, 16, 16), (64, 5, 7, 1, 3, 1, 16, 16)]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv_dgrad(case):
    cin, cout, k, stride, pad, N, H, W = case
    g = torch.Generator().manual_seed(7)
    x = bfr(torch.randn(N, cin, H, W, generator=g)).double().requires_grad_(True)
    w = bfr(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k))
    y = F.conv2d(x, w.double(), None, stride, pad)
    OH, OW = y.shape[2:]
    dy = bfr(torch.randn(N, cout, OH, OW, generator=g))
    y.backward(dy.double())
    cpad = ops.round_up(cout, 8)
    dyr = torch.zeros(N * OH * OW, cpad, dtype=BF16); dyr[:, :cout] = rows_of(dy)
    pwT = PackedWeight(cin, k * k, cpad, DEV)
    pwT.pack(w.to(DEV), transposed=True)
    dx = torch.empty(N * H * W, cin, dtype=BF16, device=DEV)
    ops.conv_igemm(dyr.to(DEV), pwT, cin, (N * H * W, OH, OW, H, W, k, k, stride, pad), y=dx, mode=1)
    report(f"conv_dgrad{case}", nchw_of(dx.cpu(), N, H, W), x.grad, atol=2e-2, rtol=1e-2)


WGRAD_CASES = [(64, 64, 3, 1, 1, 2, 20, 28), (64, 192, 7, 1, 3, 1, 16, 24), (64, 64, 3, 2, 1, 2, 18, 22), (256, 512, 1, 2, 0, 1, 16, 24),
               (3, 64, 7, 2, 3, 2, 32, 40), (64, 5, 7, 1, 3, 1, 16, 16), (1024, 512, 1, 1, 0, 1, 8, 8),
               # conv_wgrad_ring_kernel (cout >= 256, cin >= 128): 3x3 taps with a partial last chunk; ragged channel tiles; many chunks per split
               (128, 256, 3, 1, 1, 2, 20, 28), (192, 300, 1, 1, 0, 1, 24, 20), (256, 256, 1, 1, 0, 4, 48, 40)]


@pytest.mark.parametrize("use_tr", [1, 0])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(case, use_tr):
    cin, cout, k, stride, pad, N, H, W = case
    g = torch.Generator().manual_seed(11)
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = torch.randn(cout, cin, k, k, generator=g).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, stride, pad)
    OH, OW = y.shape[2:]
    dy = bfr(torch.randn(N, cout, OH, OW, generator=g))
    y.backward(dy.double())
    cin_pad, cpad = ops.round_up(cin, 8), ops.round_up(cout, 8)
    xr = torch.zeros(N * H * W, cin_pad, dtype=BF16); xr[:, :cin] = rows_of(x)
    dyr = torch.zeros(N * OH * OW, cpad, dtype=BF16); dyr[:, :cout] = rows_of(dy)
    gw = torch.full((cout, cin, k, k), float("nan"), dtype=torch.float32, device=DEV)
    ops.set_wgrad_tr(use_tr)
    try:
        ops.conv_wgrad(xr.to(DEV), dyr.to(DEV), cin, cout, (N * OH * OW, H, W, OH, OW, k, k, stride, pad), [(gw, 0, cout)])
        torch.cuda.synchronize()
    finally:
        ops.set_wgrad_tr(1)
    scale = float(w.grad.abs().max())
    report(f"conv_wgrad{case} tr={use_tr}", gw.cpu(), w.grad, atol=2e-4 * scale, rtol=1e-4)
    db = torch.empty(cout, dtype=torch.float32, device=DEV)
    ops.bias_grad(dyr.to(DEV), cout, db)
    report(f"bias_grad{case}", db.cpu(), dy.double().sum((0, 2, 3)), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("C,res,relu", [(64, False, True), (256, True, True), (1024, False, False)])
def test_batchnorm_train_forward_backward(C, res, relu):
    N, H, W = 2, 12, 20
    g = torch.Generator().manual_seed(C)
    x = bfr(torch.randn(N, C, H, W, generator=g) * 2 + 0.5)
    gamma = torch.rand(C, generator=g) + 0.5; beta = torch.randn(C, generator=g) * 0.1
    rm = torch.randn(C, generator=g) * 0.1; rv = torch.rand(C, generator=g) + 0.5
    r = bfr(torch.randn(N, C, H, W, generator=g)) if res else None
    xd = x.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(xd, rm_ref, rv_ref, gd, bd, True, 0.1, 1e-5)
    if res:
        y = y + r.double()
    if relu:
        y = F.relu(y)
    dy = bfr(torch.randn(N, C, H, W, generator=g))
    dy_eff = dy.double() * (y > 0) if relu else dy.double()
    y.backward(dy.double())
    xr = rows_of(x).to(DEV)
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    mean, invstd, scale, shift = ops.bn_stats_train(xr, C, gamma.to(DEV), beta.to(DEV), rm_d, rv_d)
    out = torch.empty_like(xr)
    ops.bn_apply(xr, C, scale, shift, out, res=rows_of(r).to(DEV) if res else None, relu=relu)
    report(f"bn_fwd C={C}", nchw_of(out.cpu(), N, H, W), y, atol=3e-2, rtol=1e-2)
    report(f"bn_running_mean C={C}", rm_d.cpu(), rm_ref, atol=1e-5, rtol=1e-5)
    report(f"bn_running_var C={C}", rv_d.cpu(), rv_ref, atol=1e-5, rtol=1e-4)
    dg = torch.empty(C, device=DEV); db = torch.empty(C, device=DEV)
    dx = torch.empty_like(xr)
    ops.bn_bwd(xr, rows_of(bfr(dy_eff.float())).to(DEV), C, gamma.to(DEV), mean, invstd, dg, db, dx)
    report(f"bn_dgamma C={C}", dg.cpu(), gd.grad, atol=5e-2, rtol=5e-3)
    report(f"bn_dbeta C={C}", db.cpu(), bd.grad, atol=5e-2, rtol=5e-3)
    report(f"bn_dx C={C}", nchw_of(dx.cpu(), N, H, W), xd.grad, atol=2e-2, rtol=1e-2)
    sc_e, sh_e = ops.bn_scale_shift_eval(C, gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV))
    ops.bn_apply(xr, C, sc_e, sh_e, out, relu=False)
    report(f"bn_eval C={C}", nchw_of(out.cpu(), N, H, W), F.batch_norm(x.double(), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5),
           atol=3e-2, rtol=1e-2)


def test_maxpool_forward_backward():
    N, C, H, W = 2, 64, 18, 22
    g = torch.Generator().manual_seed(5)
    x = F.relu(bfr(torch.randn(N, C, H, W, generator=g)))
    xd = x.double().requires_grad_(True)
    y = F.max_pool2d(xd, 3, 2, 1)
    dy = bfr(torch.randn(*y.shape, generator=g))
    y.backward(dy.double())
    xr = rows_of(x).to(DEV)
    OH, OW = y.shape[2:]
    out = torch.empty(N * OH * OW, C, dtype=BF16, device=DEV)
    ops.maxpool_fwd(xr, out, N, H, W, C)
    report("maxpool_fwd", nchw_of(out.cpu(), N, OH, OW), y, atol=0, rtol=0)
    dx = torch.empty_like(xr)
    ops.maxpool_bwd(xr, rows_of(dy).to(DEV), dx, N, H, W, C)
    # ties at exactly 0 (ReLU plateaus) may route differently; the network masks those positions (x==0) afterwards
    m = (x > 0).double()
    report("maxpool_bwd", nchw_of(dx.cpu(), N, H, W) * m, xd.grad * m, atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("shape", [(2, 64, 8, 12, 16, 24), (1, 256, 5, 7, 10, 14)])
def test_bilinear_forward_backward(shape):
    N, C, IH, IW, OH, OW = shape
    g = torch.Generator().manual_seed(9)
    x = bfr(torch.randn(N, C, IH, IW, generator=g))
    xd = x.double().requires_grad_(True)
    y = F.interpolate(xd, (OH, OW), mode="bilinear", align_corners=False)
    dy = bfr(torch.randn(N, C, OH, OW, generator=g))
    y.backward(dy.double())
    out = torch.empty(N * OH * OW, C, dtype=BF16, device=DEV)
    ops.bilinear_fwd(rows_of(x).to(DEV), out, N, IH, IW, OH, OW, C)
    report("bilinear_fwd", nchw_of(out.cpu(), N, OH, OW), y, atol=2e-2, rtol=1e-2)
    dx = torch.empty(N * IH * IW, C, dtype=BF16, device=DEV)
    ops.bilinear_bwd(rows_of(dy).to(DEV), dx, N, IH, IW, OH, OW, C)
    report("bilinear_bwd", nchw_of(dx.cpu(), N, IH, IW), xd.grad, atol=3e-2, rtol=1e-2)


def test_detection_loss_matches_golden(golden):
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    g = golden("loss.npz")
    t = [torch.tensor(g[k], device=DEV, requires_grad=True) for k in ("kp", "short", "mid")]
    l = DetectionLossAll(kp_radius=5)(t, torch.from_numpy(g["gt"]).to(DEV))
    print("loss", float(l), float(g["loss"]))
    assert abs(float(l) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    l.backward()
    for tt, k in zip(t, ("g_kp", "g_short", "g_mid")):
        report(k, tt.grad.cpu(), torch.from_numpy(g[k]), atol=1e-9, rtol=2e-5)
    l0 = DetectionLossAll(kp_radius=5)([x.detach() for x in t], torch.zeros_like(torch.from_numpy(g["gt"])).to(DEV))
    assert abs(float(l0) - float(g["loss_empty"])) <= 2e-6


def test_seg_loss_matches_golden(golden):
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    g = golden("loss.npz")
    patches = [[torch.tensor(g[f"seg.patch.{i}.{j}"], device=DEV, requires_grad=True) for j in range(n)] for i, n in ((0, 2), (1, 1))]
    dets = [[torch.from_numpy(g[f"seg.det.{i}.{j}"]) for j in range(n)] for i, n in ((0, 2), (1, 1))]
    gm = [g["seg.gmask.0"], g["seg.gmask.1"]]; gb = [g["seg.gbox.0"], g["seg.gbox.1"]]
    l = SEG_loss(40, 48)([patches, dets], gm, gb)
    print("seg loss", float(l), float(g["seg.loss"]))
    assert abs(float(l) - float(g["seg.loss"])) <= 2e-6
    l.backward()
    # gradient check against the oracle restatement (torch CPU)
    from oracle import net as onet
    pc = [[torch.tensor(g[f"seg.patch.{i}.{j}"], requires_grad=True) for j in range(n)] for i, n in ((0, 2), (1, 1))]
    onet.seg_loss([pc, dets], gm, gb, 40, 48).backward()
    for i in range(2):
        for a, b in zip(patches[i], pc[i]):
            ref = b.grad if b.grad is not None else torch.zeros_like(b)
            got = a.grad.cpu() if a.grad is not None else torch.zeros_like(b)
            report(f"seg_grad{i}", got, ref, atol=1e-8, rtol=2e-5)
    assert SEG_loss(40, 48)([[[patches[1][0]]], [[dets[1][0]]]], [gm[1]], [gb[1]]) is None


HALO_CASES = [
    # cin, cout, k, N, H, W, relu, bias
    (64, 64, 3, 2, 20, 28, True, True),
    (64, 200, 3, 3, 37, 50, False, True),     # conv3_c64.hip: several cout blocks, tiles crossing the image border, many tiles per workgroup
    (64, 40, 3, 9, 80, 96, True, False),
    (64, 64, 7, 1, 24, 40, True, True),
    (64, 192, 7, 1, 32, 32, True, True),
    (128, 128, 3, 1, 16, 16, False, True),
    (256, 256, 7, 1, 16, 24, True, True),
    (64, 100, 3, 1, 18, 21, False, False),
    (1024, 512, 3, 1, 8, 8, True, True),
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_halo_forward_and_dgrad(case):
    """LDS-halo kernel (kg_conv2d_halo) forward and input-gradient vs F.conv2d / its autograd."""
    cin, cout, k, N, H, W, relu, bias = case
    pad = k // 2
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = bfr(torch.randn(N, cin, H, W, generator=g)).double().requires_grad_(True)
    w = bfr(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k))
    b = torch.randn(cout, generator=g) if bias else None
    pre = F.conv2d(x, w.double(), b.double() if bias else None, 1, pad)
    ref = F.relu(pre) if relu else pre
    dy = bfr(torch.randn(N, cout, H, W, generator=g))
    pre.backward(dy.double())
    xr = rows_of(x.detach().float()).to(DEV)
    pw = PackedWeight(cout, k * k, cin, DEV); pw.pack(w.to(DEV))
    geom = (N * H * W, H, W, H, W, k, k, 1, pad)
    y = torch.empty(N * H * W, cout, dtype=BF16, device=DEV)
    kind = ops.conv_auto(xr, pw, cout, geom, N, y=y, bias=b.to(DEV) if bias else None, relu=relu, tiny=False)
    assert kind == "halo"
    report(f"halo_fwd{case}", nchw_of(y.cpu(), N, H, W), ref, atol=2e-2, rtol=1e-2)
    yf = torch.empty(N, cout, H, W, dtype=torch.float32, device=DEV)
    ops.conv_auto(xr, pw, cout, geom, N, y_f32=yf, bias=b.to(DEV) if bias else None, relu=relu)
    report(f"halo_fwd_f32{case}", yf.cpu(), ref, atol=3e-4, rtol=3e-4)
    if cout % 64 == 0:
        pwT = PackedWeight(cin, k * k, cout, DEV); pwT.pack(w.to(DEV), transposed=True)
        dx = torch.empty(N * H * W, cin, dtype=BF16, device=DEV)
        msk = bfr(torch.randn(N, cin, H, W, generator=g))
        kind = ops.conv_auto(rows_of(dy).to(DEV), pwT, cin, geom, N, y=dx, mask=rows_of(msk).to(DEV), transposed=True, tiny=False)
        assert kind == "halo"
        report(f"halo_dgrad{case}", nchw_of(dx.cpu(), N, H, W), x.grad * (msk > 0), atol=2e-2, rtol=1e-2)


def test_gather_128x64_variant_on_small_problems():
    """conv_gather's 4-wave 128-pixel x 64-cout variant serves launches of >= 512 tiles; KG_GATHER_N64=1 sends every 64-cout gather conv
    of the small test problems through it (plain and paired planes, dense / ragged, with the BatchNorm-statistics epilogue)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KG_GATHER_N64="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_planes.py"), os.path.join(root, "tests", "test_gpu_blocks.py"),
                        "-q", "-x"], capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_four_wave_blocked_7x7_halo_kernel_matches():
    """conv_halo7_w4_kernel (csrc/conv_halo.hip: four waves of 64 couts x 128 pixels, accumulation restarted at every 64-channel chunk into a second
    accumulator) serves the multi-product launches with >= 2 chunks per plane by default; KG_HALO7_W4=2 sends EVERY dense 7x7 rows-output launch of the
    halo / plane conv tests through it (forward and flipped input gradient, both 16-bit formats, masks / residuals, partial tiles)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KG_HALO7_W4="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), os.path.join(root, "tests", "test_gpu_planes.py"),
                        "-q", "-x", "-k", "test_conv_halo_forward_and_dgrad or test_conv_forward_dgrad_wgrad_planes"],
                       capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_four_wave_3x3_halo_kernel_matches():
    """conv_halo3_w4_kernel (the 4-wave body with 128-cout x 128-pixel wave tiles, csrc/conv_halo.hip) serves the dense 3x3 convs with >= 128 couts
    and >= 192 workgroups (the decoder's up convs at the bench size); KG_HALO3_NB2=2 sends EVERY dense 3x3 launch with >= 128 couts of the halo /
    plane conv tests and of the sub-graph tests through it (forward and flipped input gradient, both formats, masks / residuals, partial tiles,
    a 64-cout remainder)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KG_HALO3_NB2="2", KG_HALO_SPLIT="0")      # (no K split: under-filled test launches would take it first)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), os.path.join(root, "tests", "test_gpu_planes.py"),
                        os.path.join(root, "tests", "test_gpu_blocks.py"), "-q", "-x", "-k",
                        "test_conv_halo_forward_and_dgrad or test_conv_forward_dgrad_wgrad_planes or test_stem_and_decoder_level or test_bottleneck_block"],
                       capture_output=True, text=True, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


WGRAD_HALO_CASES = [(64, 64, 3, 2, 20, 28), (64, 192, 7, 1, 32, 32), (64, 5, 7, 1, 16, 24), (3, 64, 3, 1, 24, 24),
                    (256, 128, 3, 1, 16, 16), (128, 64, 7, 2, 18, 21), (1024, 512, 3, 1, 8, 8),
                    (64, 10, 7, 1, 20, 20), (128, 40, 7, 1, 17, 33), (64, 1, 3, 2, 19, 23)]


@pytest.mark.parametrize("case", WGRAD_HALO_CASES)
def test_conv_wgrad_halo(case):
    """kg_conv2d_wgrad_halo (all taps per staging pass) vs the autograd weight gradient of F.conv2d."""
    cin, cout, k, N, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = torch.randn(cout, cin, k, k, generator=g).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, 1, k // 2)
    dy = bfr(torch.randn(N, cout, H, W, generator=g))
    y.backward(dy.double())
    cin_pad, cpad = ops.round_up(cin, 8), ops.round_up(cout, 8)
    xr = torch.zeros(N * H * W, cin_pad, dtype=BF16); xr[:, :cin] = rows_of(x)
    dyr = torch.zeros(N * H * W, cpad, dtype=BF16); dyr[:, :cout] = rows_of(dy)
    gw = torch.full((cout, cin, k, k), float("nan"), dtype=torch.float32, device=DEV)
    db = torch.full((cout,), float("nan"), dtype=torch.float32, device=DEV)
    kind = ops.conv_wgrad(xr.to(DEV), dyr.to(DEV), cin, cout, (N * H * W, H, W, H, W, k, k, 1, k // 2), [(gw, 0, cout)], N=N, bias_out=db)
    assert kind == "halo"
    torch.cuda.synchronize()
    scale = float(w.grad.abs().max())
    report(f"wgrad_halo{case}", gw.cpu(), w.grad, atol=2e-4 * scale, rtol=1e-4)
    report(f"wgrad_halo{case}.bias", db.cpu(), dy.double().sum((0, 2, 3)), atol=1e-3, rtol=1e-4)     # the fused all-ones unit


@pytest.mark.parametrize("case", [(128, 64, 5000, True, True), (64, 128, 4096, False, False), (1024, 512, 777, True, True),
                                  (256, 100, 300, False, True), (64, 64, 70000, True, False), (64, 40, 1000, True, True),
                                  (128, 256, 3333, False, True), (64, 12, 500, False, True)])
def test_conv1x1_streaming(case):
    """kg_conv1x1 (persistent streaming GEMM) vs fp64 matmul, incl. residual / mask epilogue and channel-slice I/O."""
    K, cout, M, relu, bias = case
    g = torch.Generator().manual_seed(K + cout)
    x = bfr(torch.randn(M, K + 64, generator=g))          # consume a K-wide slice of a wider buffer
    w = bfr(torch.randn(cout, K, 1, 1, generator=g) / math.sqrt(K))
    b = torch.randn(cout, generator=g) if bias else None
    res = bfr(torch.randn(M, cout, generator=g)); msk = bfr(torch.randn(M, cout, generator=g))
    ref = x[:, 64:].double() @ w.view(cout, K).double().t()
    if bias:
        ref = ref + b.double()
    ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    ref = ref * (msk > 0)
    pw = PackedWeight(cout, 1, K, DEV); pw.pack(w.to(DEV))
    xd = x.to(BF16).to(DEV)
    ybuf = torch.zeros(M, cout + 32, dtype=BF16, device=DEV)
    # (conv_auto routes compute-heavy 1x1 convs, K >= 192 and Cout > 64, to the LDS-ring gather kernel; kg_conv1x1 still serves them)
    assert ops.can_1x1(xd[:, 64:], pw, 1, 1, 0, ybuf[:, 32:], None) == (not (K >= 192 and cout > 64))
    ops.conv1x1(xd[:, 64:], pw, cout, ybuf[:, 32:], bias=b.to(DEV) if bias else None, res=res.to(BF16).to(DEV),
                mask=msk.to(BF16).to(DEV), relu=relu)
    report(f"conv1x1{case}", ybuf[:, 32:].float().cpu(), ref, atol=3e-2, rtol=1e-2)
    assert float(ybuf[:, :32].abs().max()) == 0.0


@pytest.mark.parametrize("case", [(64, 2, 40, 44), (128, 1, 16, 32), (256, 1, 33, 20),
                                  (64, 1, 272, 500)])      # >= 256 pixel tiles: one workgroup walks the three heads (no head split), ragged right edge
def test_conv_halo_heads2(case):
    """kg_conv2d_halo_heads2 (the three second-layer 7x7 head convs of one scale as ONE grouped launch, KGnet.py:161-209
    `.2` + sigmoid :300) vs three fp64 F.conv2d on the slices of the fused hidden tensor."""
    C, N, H, W = case
    g = torch.Generator().manual_seed(C + H)
    hid = bfr(torch.randn(N, 3 * C, H, W, generator=g).clamp_min(0))
    rows, vmap = ops.heads2_layout()
    assert sorted(v for v in vmap if v >= 0) == list(range(55)) and [len(r) for r in rows] == [5, 10, 40]
    pw = PackedWeight(64, 49, C, DEV, groups=3)
    vm = torch.tensor(vmap, dtype=torch.int32, device=DEV)
    bias64 = torch.zeros(64, device=DEV)
    refs, outs = [], []
    for k, co in enumerate((5, 10, 40)):
        w = bfr(torch.randn(co, C, 7, 7, generator=g) / math.sqrt(49 * C))
        b = torch.randn(co, generator=g)
        r = F.conv2d(hid[:, k * C:(k + 1) * C].double(), w.double(), b.double(), 1, 3)
        refs.append(torch.sigmoid(r) if k == 0 else r)
        rm = torch.tensor(rows[k], dtype=torch.int32, device=DEV)
        pw.pack_rows(w.to(DEV), rm, group=k)
        bias64[rm.long()] = b.to(DEV)
        outs.append(torch.full((N, co, H, W), float("nan"), dtype=torch.float32, device=DEV))
    ops.conv_halo_heads2(rows_of(hid).to(DEV), pw, bias64, vm, outs[0], outs[1], outs[2], N, H, W, C)
    for name, o, r in zip(("kp", "short", "mid"), outs, refs):
        report(f"heads2{case}.{name}", o.cpu(), r, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("cin,taps,S,counts", [(64, 9, 256, [64]), (64, 49, 40, [64, 64, 64]), (512, 49, 1, [5, 10, 40]), (24, 1, 64, [16, 8])])
def test_wgrad_reduce_multi(cin, taps, S, counts):
    """kg_wgrad_reduce_multi: split partials [S][cout][tap][ci] -> one OIHW gradient per fused head, vs a float64 sum
    (covers the thread-group path for small layers with many splits and the accumulate flag)."""
    import ctypes
    k = int(round(taps ** 0.5))
    cout = sum(counts)
    g = torch.Generator().manual_seed(cin + taps + S)
    part = torch.randn(S, cout, taps, cin, generator=g).to(DEV)
    ref = part.double().sum(0).permute(0, 2, 1).reshape(cout, cin, k, k)
    outs = [torch.full((c, cin, k, k), 0.5, dtype=torch.float32, device=DEV) for c in counts]
    gp = (ctypes.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
    cn = (ctypes.c_int * len(outs))(*counts)
    for acc in (0, 1):
        _lib.call("kg_wgrad_reduce_multi", _lib.ptr(part), gp, cn, len(outs), cin, k, k, S, ctypes.c_long(cout * taps * cin), acc, _lib.stream_ptr())
        torch.cuda.synchronize()
        got = torch.cat(outs, 0).double()
        want = ref * (acc + 1)
        assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("case", [(64, 256, 1, 1, 2, 40, 36), (256, 512, 1, 1, 3, 17, 23), (128, 128, 3, 1, 2, 33, 50), (256, 512, 1, 2, 2, 20, 24),
                                  (128, 128, 3, 2, 2, 26, 30), (128, 64, 3, 1, 1, 16, 32), (1024, 256, 1, 1, 2, 9, 11)])
def test_conv_epilogue_bn_statistics(case):
    """BatchNorm statistics from the producing conv's epilogue (kg_conv_stats_begin / _end + kg_bn_finalize_train, conv_args.h): mean /
    invstd / scale / shift and the running-statistics update against float64 statistics of the float64 conv output (the epilogue sums
    the fp32 accumulators), and against the two-pass kg_bn_stats_train over the stored bf16 rows (which sees the rounded values)."""
    cin, cout, k, stride, N, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = bfr(torch.randn(N, cin, H, W, generator=g))
    w = bfr(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k))
    ref = F.conv2d(x.double(), w.double(), None, stride, k // 2)
    OH, OW = ref.shape[2:]
    M = N * OH * OW
    pw = PackedWeight(cout, k * k, cin, DEV); pw.pack(w.to(DEV))
    y = torch.empty(M, cout, dtype=BF16, device=DEV)
    geom = (M, H, W, OH, OW, k, k, stride, k // 2)
    gamma = (torch.rand(cout, generator=g) + 0.5).to(DEV); beta = torch.randn(cout, generator=g).to(DEV)
    part = ops.conv_stats_begin(torch.device(DEV))
    kind = ops.conv_auto(rows_of(x).to(DEV), pw, cout, geom, N, y=y, tiny=False)      # (armed: the regular kernels, as Engine.conv does)
    nb = ops.conv_stats_end()
    if kind == "1x1":        # the streaming 1x1 kernels (single-plane operands, K <= 128 or Cout <= 64) have no statistics epilogue:
        assert nb == 0      # the channel reports it and the caller runs the two-pass kg_bn_stats_train
        return
    assert nb > 0, (kind, "this conv shape must take a kernel with the statistics epilogue")
    rm1, rv1 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    mean, invstd, scale, shift = ops.bn_finalize_train(part, nb, M, cout, gamma, beta, rm1, rv1)
    r = ref.permute(1, 0, 2, 3).reshape(cout, -1)
    mu, var = r.mean(1), r.var(1, unbiased=False)
    report(f"bnstats{case}.mean", mean.cpu(), mu, atol=1e-5, rtol=1e-4)
    report(f"bnstats{case}.invstd", invstd.cpu(), 1.0 / torch.sqrt(var + 1e-5), atol=1e-5, rtol=2e-4)
    report(f"bnstats{case}.running_var", rv1.cpu(), 0.9 + 0.1 * r.var(1, unbiased=True), atol=1e-5, rtol=2e-4)
    report(f"bnstats{case}.shift", shift.cpu(), beta.cpu().double() - mu * gamma.cpu().double() / torch.sqrt(var + 1e-5), atol=2e-4, rtol=2e-4)
    rm2, rv2 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    mean2, invstd2, _, _ = ops.bn_stats_train(y, cout, gamma, beta, rm2, rv2)
    report(f"bnstats{case}.mean_vs_two_pass", mean.cpu(), mean2.cpu(), atol=2e-3, rtol=1e-2)      # (the two-pass version sees bf16-rounded rows)
    report(f"bnstats{case}.invstd_vs_two_pass", invstd.cpu(), invstd2.cpu(), atol=1e-3, rtol=1e-2)
    # disarmed again: a second conv leaves the buffer alone
    ops.conv_auto(rows_of(x).to(DEV), pw, cout, geom, N, y=y, tiny=False)
    assert ops.conv_stats_end() == 0


# ---- gradient scale of the half-precision backward pass (csrc/gradscale.hip, norm_pool.hip) --------------------------------------------
def test_grad_scale_and_scale_tensors():
    """kg_grad_scale: S = the power of two that brings max |g| (kp maps: |g * p * (1 - p)|) into [2^(T-1), 2^T); zero / non-finite input ->
    S = 1; kg_scale_tensors: per-tensor device scalars, exact for powers of two, sticky non-finite flag."""
    T = ops.GRAD_TARGET_LOG2
    g = torch.Generator().manual_seed(0)
    a = (torch.randn(3, 5, 17, 19, generator=g) * 1e-6).to(DEV)
    b = (torch.randn(1000, generator=g) * 3e-4).to(DEV)
    gs = ops.grad_scale([a, None, b])
    S, inv = float(gs[0]), float(gs[1])
    m = max(float(a.abs().max()), float(b.abs().max()))
    assert S * inv == 1.0 and math.log2(S) == round(math.log2(S)) and 2.0 ** (T - 1) <= m * S < 2.0 ** T
    # sigmoid outputs: the logit gradient g * p * (1 - p) counts (saturated p -> 0), not dL/dp
    p = torch.rand(3, 5, 17, 19, generator=g).to(DEV)
    p[0, 0, 0, 0] = 1.0
    big = a.clone(); big[0, 0, 0, 0] = 1e12                       # dL/dp of a saturated pixel (loss.py:13 clamps log at -100)
    gs2 = ops.grad_scale([big], [p])
    m2 = float((big * p * (1 - p)).abs().max())
    assert 2.0 ** (T - 1) <= m2 * float(gs2[0]) < 2.0 ** T
    assert float(ops.grad_scale([torch.zeros(10, device=DEV)])[0]) == 1.0
    assert float(ops.grad_scale([torch.full((4,), float("nan"), device=DEV)])[0]) == 1.0
    # scale_tensors: different scalars per tensor, unaligned views, flag
    x, y = torch.randn(4097, device=DEV), torch.randn(33, device=DEV)[1:]
    x0, y0 = x.clone(), y.clone()
    s1, s2 = torch.tensor([0.25], device=DEV), torch.tensor([8.0], device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.scale_tensors([x, y.contiguous() if not y.is_contiguous() else y], [s1, s2], flag=flag)
    assert torch.equal(x, x0 * 0.25) and torch.equal(y, y0 * 8.0) and int(flag) == 0
    z = torch.tensor([1.0, float("inf"), 2.0], device=DEV)
    ops.scale_tensors([z], s1, flag=flag)
    assert int(flag) == 1


@pytest.mark.parametrize("dt", [BF16, ops.F16])
def test_rows_rescale_stage_boundary(dt):
    """kg_rows_rescale / kg_rows_scale: in-place power-of-two re-normalisation of a gradient rows tensor (all planes), chained scale."""
    T = ops.GRAD_TARGET_LOG2
    g = torch.Generator().manual_seed(1)
    v = (torch.randn(777, 64, generator=g) * 37.0).to(DEV)
    for P in (1, 2):
        pt = ops.alloc_pt(777, 64, P, DEV, dtype=dt)
        ops.f32_to_planes(v, pt, 64)
        before = torch.empty(777, 64, device=DEV); ops.planes_to_f32(pt, 64, before)
        cum_in = torch.tensor([4.0, 0.25], device=DEV)
        r, cum = ops.rows_rescale(pt, 64, cum_in)
        after = torch.empty(777, 64, device=DEV); ops.planes_to_f32(pt, 64, after)
        rr = float(r)
        assert math.log2(rr) == round(math.log2(rr)) and 2.0 ** (T - 1) <= float(before.abs().max()) * rr < 2.0 ** T
        # (a power of two: exact, except where a tiny element leaves the 16-bit format's normal range on the way down)
        assert float((after - before * rr).abs().max()) <= 2.0 ** -20 * float(after.abs().max())
        assert float(cum[0]) == 4.0 * rr and float(cum[0]) * float(cum[1]) == 1.0
        other = ops.alloc_pt(100, 64, P, DEV, dtype=dt)
        ops.f32_to_planes(v[:100].contiguous(), other, 64)
        o0 = torch.empty(100, 64, device=DEV); ops.planes_to_f32(other, 64, o0)
        ops.rows_scale(other, 64, r)
        o1 = torch.empty(100, 64, device=DEV); ops.planes_to_f32(other, 64, o1)
        assert float((o1 - o0 * rr).abs().max()) <= 2.0 ** -20 * float(o1.abs().max())
    # several tensors (one a column slice of a wider buffer) in one launch
    wide = ops.alloc_pt(50, 128, 2, DEV, dtype=dt)
    ops.f32_to_planes(torch.randn(50, 128, device=DEV), wide, 128)
    sl, full = wide.cols(64, 128), ops.alloc_pt(9, 64, 1, DEV, dtype=dt)
    ops.f32_to_planes(torch.randn(9, 64, device=DEV), full, 64)
    w0 = torch.empty(50, 128, device=DEV); ops.planes_to_f32(wide, 128, w0)
    f0 = torch.empty(9, 64, device=DEV); ops.planes_to_f32(full, 64, f0)
    half = torch.tensor([0.5], device=DEV)
    ops.rows_scale_multi([(sl, 64), (full, 64)], half)
    w1 = torch.empty(50, 128, device=DEV); ops.planes_to_f32(wide, 128, w1)
    f1 = torch.empty(9, 64, device=DEV); ops.planes_to_f32(full, 64, f1)
    tol = 2.0 ** -20 * float(w0.abs().max())          # (exact, except where a tiny lo-plane element leaves the normal range on the way down)
    assert torch.equal(w1[:, :64], w0[:, :64]) and float((w1[:, 64:] - w0[:, 64:] * 0.5).abs().max()) <= tol and float((f1 - f0 * 0.5).abs().max()) <= tol
    ops.rows_scale(full, 64, half, torch.tensor([4.0], device=DEV))        # two device scalars: * 0.5 * 4
    f2 = torch.empty(9, 64, device=DEV); ops.planes_to_f32(full, 64, f2)
    assert float((f2 - f0).abs().max()) <= tol
    zero = ops.alloc_pt(8, 64, 1, DEV, zero=True, dtype=dt)
    r0, c0 = ops.rows_rescale(zero, 64, torch.tensor([2.0, 0.5], device=DEV))
    assert float(r0) == 1.0 and float(c0[0]) == 2.0


@pytest.mark.parametrize("dt", [BF16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", [(2, 5, 16, 33, 37, True), (1, 10, 16, 40, 40, False), (3, 40, 48, 17, 19, False), (1, 64, 64, 16, 16, False), (1, 70, 72, 8, 8, False)])
def test_grad_pack_matches_torch(case, dt):
    """kg_grad_pack (fp32 NCHW loss gradient -> rows planes, times p (1 - p) for sigmoid outputs, times the step's gradient scale): the LDS-transpose
    kernel (cpad <= 64) and the per-pixel one (cpad > 64), pixel counts that are no multiple of the workgroup's 256."""
    N, C, cpad, H, W, with_prob = case
    g = torch.Generator().manual_seed(C + H)
    grad = torch.randn(N, C, H, W, generator=g).to(DEV)
    prob = torch.rand(N, C, H, W, generator=g).to(DEV) if with_prob else None
    scale = torch.tensor([4.0], device=DEV)
    for P in (1, 2):
        out = ops.alloc_pt(N * H * W, 80, P, DEV, dtype=dt)          # ld = 80 * P > cpad: a column slice
        out.t.fill_(7.0)
        view = ops.PT(out.t[:, :cpad], P, out.ps)
        ops.grad_pack(grad, prob, view, N, C, H, W, cpad, scale=scale)
        torch.cuda.synchronize()
        ref = grad * (prob * (1 - prob) if with_prob else 1.0) * 4.0
        ref = ref.permute(0, 2, 3, 1).reshape(N * H * W, C)
        got = sum(view.plane(p).float() for p in range(P))
        tol = {1: 2.0 ** -8 if dt == BF16 else 2.0 ** -11, 2: 2.0 ** -16 if dt == BF16 else 2.0 ** -21}[P]
        assert float((got[:, :C] - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-7
        assert float(got[:, C:cpad].abs().max()) == 0.0 if cpad > C else True
        assert float((out.t[:, cpad:80].float() - 7.0).abs().max()) == 0.0           # columns past cpad untouched


@pytest.mark.parametrize("dt", [BF16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("P", [1, 2])
@pytest.mark.parametrize("case", [(1, 2, 2, 8), (2, 5, 7, 64), (3, 19, 33, 24), (1, 32, 16, 256), (2, 9, 40, 64)])
def test_exact_2x_bilinear_kernel_is_bit_identical_to_the_generic_one(case, P, dt):
    """kg_bilinear_fwd's dense exact-2x path (bilinear2x_fwd_kernel: one thread per 8-channel chunk of an input column, the 3 x 3 neighbourhood in registers,
    strips of 8 input rows) against the generic gather kernel, reached through the ragged descriptor form with the images given as boxes: the same indices and
    the same weighted sum term for term -- every element bit for bit, maps smaller than / not a multiple of the strip, one and two planes; and against
    F.interpolate (KGnet.py:288-297) within the storage rounding."""
    N, IH, IW, C = case
    OH, OW = 2 * IH, 2 * IW
    g = torch.Generator().manual_seed(IH * 100 + IW)
    xf = torch.randn(N, C, IH, IW, generator=g)
    x = ops.alloc_pt(N * IH * IW, C, P, DEV, dtype=dt)
    rows = xf.permute(0, 2, 3, 1).reshape(N * IH * IW, C).to(DEV)
    x.t.copy_(rows)
    if P == 2:
        x.plane(1).copy_(rows - x.t.float())
    fast = ops.alloc_pt(N * OH * OW, C, P, DEV, dtype=dt)
    slow = ops.alloc_pt(N * OH * OW, C, P, DEV, dtype=dt)
    ops.bilinear_fwd(x, fast, N, IH, IW, OH, OW, C)
    desc = torch.tensor([[n * IH * IW, IH, IW, n * OH * OW, OH, OW] for n in range(N)], dtype=torch.int32, device=DEV)
    r2b = torch.arange(N, dtype=torch.int32, device=DEV).repeat_interleave(OH * OW).contiguous()
    ops.bilinear_fwd(x, slow, N, IH, IW, OH, OW, C, boxdesc=desc, row2box=r2b)
    torch.cuda.synchronize()
    for p in range(P):
        assert torch.equal(fast.plane(p), slow.plane(p)), (case, P, p)
    val = sum(x.plane(p).double() for p in range(P)).cpu().reshape(N, IH, IW, C).permute(0, 3, 1, 2)
    ref = F.interpolate(val, (OH, OW), mode="bilinear", align_corners=False)
    got = sum(fast.plane(p).double() for p in range(P)).cpu().reshape(N, OH, OW, C).permute(0, 3, 1, 2)
    tol = {1: 2.0 ** -8 if dt == BF16 else 2.0 ** -11, 2: 2.0 ** -15 if dt == BF16 else 2.0 ** -20}[P]
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("dt", [BF16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", [(1, 16, 32, 64), (2, 37, 45, 64), (1, 48, 80, 256), (3, 20, 33, 128)])
@pytest.mark.parametrize("narrow", [8, 16])
@pytest.mark.parametrize("route", ["halo", "persistent"])
def test_narrow_halo_input_gradient(route, narrow, case, dt):
    """route "persistent": ops.conv7_narrow (conv7_narrow.hip: weights resident in LDS, compact halos, persistent workgroups); route "halo":
    conv_halo(narrow = 8 / 16): the input gradient of the kp (5 -> 8 channels) / short (10 -> 16) second-layer head convs (KGnet.py:161-209 `.2`) with 4 / 2
    kernel columns per MFMA k-step (conv_halo.hip GM = 3 / 4, weights from PackedWeight.pack_narrow) against torch's conv_transpose2d of the same 16-bit
    operands in float64, ReLU mask applied: maps that are no multiple of the 16 x 32 tile, several images, 1 / 2 / 4 cout blocks; the channels of the packed
    dY rows that belong to the OTHER heads hold large junk the kernel must not see."""
    N, H, W, C = case
    co, c_lo = (5, 0) if narrow == 8 else (10, 8)
    g = torch.Generator().manual_seed(narrow * 1000 + H * W + C)
    w = (torch.randn(co, C, 7, 7, generator=g) * 0.05).to(DEV)
    dy = torch.randn(N, co, H, W, generator=g)
    hidden = torch.randn(N, C, H, W, generator=g)            # the forward value whose sign is the ReLU mask
    rows = torch.randn(N * H * W, 64, generator=g) * 100.0      # junk in every channel ...
    rows[:, c_lo:c_lo + narrow] = 0.0
    rows[:, c_lo:c_lo + co] = dy.permute(0, 2, 3, 1).reshape(N * H * W, co)     # ... but this head's
    gy = ops.alloc_pt(N * H * W, 64, 1, DEV, dtype=dt)
    gy.t.copy_(rows)
    mask = ops.alloc_pt(N * H * W, C, 1, DEV, dtype=dt)
    mask.t.copy_(hidden.permute(0, 2, 3, 1).reshape(N * H * W, C))
    pw = PackedWeight(C, 7 * narrow // 8, 64, DEV, dtype=dt)
    pw.pack_narrow(w, narrow)
    out = ops.alloc_pt(N * H * W, C + 64, 1, DEV, dtype=dt)      # (a column slice of a wider buffer, as engine.heads_second writes it)
    out.t.fill_(9.0)
    if route == "halo":
        ops.conv_halo(gy, pw, C, N, H, W, 7, y=out.cols(32, 32 + C), mask=mask.t, flip=True, narrow=narrow)
    else:
        ops.conv7_narrow(gy, pw, C, N, H, W, out.cols(32, 32 + C), mask=mask.t, chan_lo=c_lo, chan_slot=narrow)
    torch.cuda.synchronize()
    dyq = gy.t[:, c_lo:c_lo + co].double().cpu().reshape(N, H, W, co).permute(0, 3, 1, 2)
    wq = w.to(dt).double().cpu()
    ref = F.conv_transpose2d(dyq, wq, padding=3) * (mask.t.double().cpu().reshape(N, H, W, C).permute(0, 3, 1, 2) > 0)
    got = out.t[:, 32:32 + C].double().cpu().reshape(N, H, W, C).permute(0, 3, 1, 2)
    tol = (2.0 ** -8 if dt == BF16 else 2.0 ** -11) * float(ref.abs().max()) + 1e-6
    assert float((got - ref).abs().max()) <= tol, (float((got - ref).abs().max()), tol)
    assert float((out.t[:, :32].float() - 9.0).abs().max()) == 0.0 and float((out.t[:, 32 + C:].float() - 9.0).abs().max()) == 0.0


@pytest.mark.parametrize("dt", [BF16, torch.float16], ids=["bf16", "f16"])
def test_rows_gather_planes_equals_per_plane_gathers(dt):
    """kg_rows_gather_planes (the crop rows of a split feature map, both planes in one launch: seg.SegBranch.gather) against a torch gather of each
    plane, into a whole-width destination and into a column slice of a wider (concat) buffer whose plane stride differs from the source's."""
    from kg_instance_segmentation_amd import _lib
    from kg_instance_segmentation_amd._lib import c_long, ptr, stream_ptr
    g = torch.Generator().manual_seed(5)
    R, C, n = 300, 64, 1111
    src = ops.alloc_pt(R, C, 2, DEV, dtype=dt)
    src.t.copy_(torch.randn(R, C, generator=g))
    src.plane(1).copy_(torch.randn(R, C, generator=g))
    idx = torch.randint(0, R, (n,), generator=g, dtype=torch.int32).to(DEV)
    wide = ops.alloc_pt(n, C + 32, 2, DEV, zero=True, dtype=dt)
    for dst in (ops.alloc_pt(n, C, 2, DEV, zero=True, dtype=dt), wide.cols(0, C)):
        _lib.call("kg_rows_gather_planes", ptr(src.t), ops.ld(src), src.ps, ptr(idx), ptr(ops.base(dst)), ops.ld(dst), dst.ps, c_long(n), C, 2, stream_ptr(),
                  fmt=ops.fmt_of(dst))
        torch.cuda.synchronize()
        for p in range(2):
            assert torch.equal(dst.plane(p), src.plane(p)[idx.long()])
    assert float(wide.t[:, C:].float().abs().max()) == 0.0 and float(wide.plane(1)[:, C:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dt", [BF16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", [(2, 33, 37), (1, 40, 40), (3, 16, 16), (8, 64, 64)])
def test_grad_pack3_equals_three_grad_packs(case, dt):
    """kg_grad_pack3 (the kp | short | mid map gradients of a level in one pass, engine.backward_dec) writes bit for bit what three kg_grad_pack
    launches into the column blocks 0..8 | 8..24 | 24..64 write: odd pixel counts (scalar read path), multiples of 4 (16-byte reads), one and two planes."""
    N, H, W = case
    g = torch.Generator().manual_seed(H * 7 + W)
    Cs, pads, offs = (5, 10, 40), (8, 16, 40), (0, 8, 24)
    grads = [torch.randn(N, c, H, W, generator=g).to(DEV) for c in Cs]
    prob = torch.rand(N, 5, H, W, generator=g).to(DEV)
    scale = torch.tensor([8.0], device=DEV)
    for P in (1, 2):
        one = ops.alloc_pt(N * H * W, 64, P, DEV, dtype=dt)
        three = ops.alloc_pt(N * H * W, 64, P, DEV, dtype=dt)
        one.t.fill_(3.0)
        three.t.fill_(5.0)
        ops.grad_pack3(grads, prob, one, N, Cs, H, W, pads, scale=scale)
        for k in range(3):
            ops.grad_pack(grads[k], prob if k == 0 else None, three.cols(offs[k], offs[k] + pads[k]), N, Cs[k], H, W, pads[k], scale=scale)
        torch.cuda.synchronize()
        for p in range(P):
            assert torch.equal(one.plane(p), three.plane(p)), (case, P, p)

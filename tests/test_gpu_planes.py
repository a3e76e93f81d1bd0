"""GPU parity tests of the split-bf16 ("planes") storage through the C ABI: a tensor is P bf16 planes whose sum is the value
(csrc/kg_common.h); P = 3 reproduces fp32 tensors exactly, P = 2 keeps 16 significant bits.

Checker: torch fp64 CPU primitives on the SAME fp32 inputs (no bf16 pre-rounding: that is the point).  Stated tolerances,
relative to max|ref| of the tensor:  P = 3 (6 MFMA products, fp32 accumulation): 3e-6;  P = 2 (3 products): 1.5e-4 -- against
2^-9 = 2e-3 for plain bf16 operands."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import _lib, ops  # noqa: E402
from kg_instance_segmentation_amd.ops import BF16, F16, PT  # noqa: E402

DEV = "cuda"
DT = [BF16]          # 16-bit format under test: BF16 (libkgnet_hip.so) or F16 (IEEE half rows, libkgnet_hip_f16.so)
TOLS = {BF16: {1: 6e-3, 2: 1.5e-4, 3: 3e-6}, F16: {1: 1e-3, 2: 5e-6, 3: 3e-6}}      # (half: 11 bits per plane, hi + lo = 22 bits)
# (P, format): bf16 hi + lo / hi + mid + lo, half single / hi + lo
VARIANTS = [(2, BF16), (3, BF16), (1, F16), (2, F16)]
VIDS = ["bf16x2", "bf16x3", "f16x1", "f16x2"]


class _Tol:
    def __getitem__(self, P):
        return TOLS[DT[0]][P]


TOL = _Tol()


def PackedWeight(*a, **k):
    k.setdefault("dtype", DT[0])
    return ops.PackedWeight(*a, **k)


def alloc_pt(rows, C, P, dev):
    return ops.alloc_pt(rows, C, P, dev, dtype=DT[0])


def to_pt(rows_f32, P, ctot=None, c0=0):
    """fp32 [rows, C] (device) -> PT with P planes (optionally a column slice c0.. of a wider [rows, ctot] plane)."""
    rows, C = rows_f32.shape
    ctot = ctot or C
    buf = torch.zeros(rows, P * ctot, dtype=DT[0], device=rows_f32.device)
    r = rows_f32.clone()
    for p in range(P):
        h = r.to(DT[0])
        buf[:, p * ctot + c0:p * ctot + c0 + C] = h
        r = r - h.float()
    return PT(buf[:, c0:c0 + C], P, ctot)


def from_pt(pt):
    out = pt.plane(pt.P - 1).float()
    for p in range(pt.P - 2, -1, -1):
        out = out + pt.plane(p).float()
    return out


def rows_f32(x):
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def nchw(rows, n, h, w):
    return rows.view(n, h, w, -1).permute(0, 3, 1, 2)


def check(name, got, ref, rel):
    got = got.detach().double().cpu(); ref = ref.detach().double().cpu()
    scale = float(ref.abs().max()) + 1e-30
    err = float((got - ref).abs().max()) / scale
    print(f"[{name}] max_abs_err / max|ref| = {err:.3e} (bound {rel:.1e}), max|ref| = {scale:.3e}")
    assert err <= rel, name


def test_split_roundtrip_exact():
    DT[0] = BF16
    """P = 3 planes written by the kernels hold fp32 values exactly (kg_f32_to_planes -> kg_planes_to_f32), including tiny and huge
    magnitudes; P = 2 keeps 16 bits."""
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(1000, 64, generator=g) * torch.exp(torch.randn(1000, 64, generator=g) * 8)).to(DEV)
    x = torch.where(x.abs() < 1e-30, torch.full_like(x, 1e-30), x)       # (residual planes of values near FLT_MIN are denormal: flushed)
    x[0, :6] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0e38, 1.0 + 2.0 ** -23], device=DEV)
    for P in (1, 2, 3):
        pt = alloc_pt(1000, 64, P, DEV)
        ops.f32_to_planes(x, pt, 64)
        back = torch.empty_like(x)
        ops.planes_to_f32(pt, 64, back)
        torch.testing.assert_close(from_pt(pt), back, rtol=0, atol=0)
        rel = ((back - x).abs() / x.abs().clamp_min(1e-37)).max().item()
        print(f"P={P}: max relative round-trip error {rel:.3e}")
        assert rel <= {1: 2.0 ** -8, 2: 2.0 ** -16, 3: 0.0}[P]
        assert torch.equal(back[0, :2], x[0, :2])


PLANE_CONV_CASES = [
    # cin, cout, k, stride, pad, N, H, W, relu, bias     (route)
    (64, 64, 3, 1, 1, 2, 20, 28, True, True),        # conv_halo<3>, 64 channels (conv3_c64 is bf16-only)
    (256, 128, 3, 1, 1, 1, 12, 20, True, True),      # conv_halo<3>
    (128, 192, 3, 1, 1, 1, 20, 36, True, True),      # conv_halo<3>; under KG_HALO3_NB2=2: conv_halo3_w4 for couts 0..127 + a 64-cout remainder
    (64, 192, 7, 1, 3, 1, 16, 24, True, True),       # conv_halo<7>
    (128, 64, 7, 1, 3, 2, 20, 36, True, True),       # conv_halo<7>, two channel chunks per plane (the shared-halo walk over several chunks)
    (256, 256, 7, 1, 3, 1, 20, 36, True, True),      # wide 7x7: planed -> conv_halo7_w4<*, 1> (blocked accumulation); single plane -> the 128-cout blocks of conv_halo7_w4<*, 2>
    (128, 64, 1, 1, 0, 2, 16, 16, True, True),       # 1x1 -> conv_gather (64 couts)
    (128, 64, 1, 1, 0, 2, 192, 176, True, True),     # 1x1, 64 couts, 528 tiles of 128 pixels -> conv_gather's 128 x 64 variant
    (64, 256, 1, 1, 0, 2, 16, 16, False, False),     # 1x1 -> conv_gather
    (128, 128, 3, 2, 1, 2, 18, 22, False, False),    # strided 3x3 -> conv_gather
    (256, 512, 1, 2, 0, 1, 16, 24, False, False),    # strided 1x1
    (3, 64, 7, 2, 3, 2, 32, 40, False, False),       # image input -> conv_igemm
    (3, 64, 3, 1, 1, 1, 24, 24, True, True),
    (256, 256, 3, 1, 1, 1, 10, 14, True, True),      # few output tiles -> conv_tiny (split-K over the waves of a 64 x 64 tile)
    (512, 128, 1, 1, 0, 1, 9, 11, False, True),      # 1x1 -> conv_tiny
    (128, 192, 3, 2, 1, 1, 17, 13, True, False),     # strided 3x3 -> conv_tiny (forward + stride-2 input gradient)
]


@pytest.mark.parametrize("P,dt", VARIANTS, ids=VIDS)
@pytest.mark.parametrize("case", PLANE_CONV_CASES)
def test_conv_forward_dgrad_wgrad_planes(case, P, dt):
    DT[0] = dt
    cin, cout, k, stride, pad, N, H, W, relu, bias = case
    g = torch.Generator().manual_seed(abs(hash(case)) % 1000 + P)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) if bias else None
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True)
    pre = F.conv2d(xd, wd, b.double() if bias else None, stride, pad)
    ref = F.relu(pre) if relu else pre
    dy = torch.randn(N, cout, OH, OW, generator=g)
    dyu = dy * (pre.detach() > 0) if relu else dy                     # gradient w.r.t. the pre-activation
    ref.backward(dy.double())
    cin_pad = ops.round_up(cin, 8)
    xr = rows_f32(x)
    if cin_pad != cin:
        xr = torch.cat([xr, torch.zeros(xr.shape[0], cin_pad - cin)], 1)
    xp = to_pt(xr.to(DEV), P, ctot=cin_pad + 16, c0=8)                 # a column slice of a wider buffer: ld != C, ps != C
    pw = PackedWeight(cout, k * k, cin_pad, DEV, xP=P, wP=P)
    pw.pack(w.to(DEV))
    y = alloc_pt(N * OH * OW, cout, P, DEV)
    geom = (N * OH * OW, H, W, OH, OW, k, k, stride, pad)
    for tiny in (False, True):          # the regular kernel of the shape, then (small problems) the split-K kernel conv_auto prefers
        y.t.zero_()
        route = ops.conv_auto(xp, pw, cout, geom, N, y=y, bias=b.to(DEV) if bias else None, relu=relu, tiny=tiny)
        torch.cuda.synchronize()
        check(f"fwd {route} P={P} {case}", nchw(from_pt(y), N, OH, OW), ref, TOL[P])
    # input gradient (with the residual operand accumulating onto an existing gradient)
    if cin >= 8:
        pwT = PackedWeight(cin, k * k, ops.round_up(cout, 8), DEV, xP=P, wP=P)
        pwT.pack(w.to(DEV), transposed=True)
        gp = to_pt(rows_f32(dyu.float()).to(DEV), P)
        prev = torch.randn(N * H * W, cin, generator=g)
        gin = (N * H * W, OH, OW, H, W, k, k, stride, pad)
        for tiny in (False, True):
            dx = to_pt(prev.to(DEV), P)
            route = ops.conv_auto(gp, pwT, cin, gin, N, y=dx, res=dx, transposed=True, tiny=tiny)
            torch.cuda.synchronize()
            check(f"dgrad {route} P={P} {case}", nchw(from_pt(dx), N, H, W), xd.grad + nchw(prev.double(), N, H, W), TOL[P])
    # weight / bias gradient
    gw = torch.empty(cout, cin, k, k, device=DEV)
    db = torch.empty(cout, device=DEV)
    gp = to_pt(rows_f32(dyu.float()).to(DEV), P)
    ops.conv_wgrad(xp, gp, cin, cout, geom, [(gw, 0, cout)], N=N, bias_out=db)
    torch.cuda.synchronize()
    check(f"wgrad P={P} {case}", gw, wd.grad, TOL[P] * 2)
    check(f"bias grad P={P} {case}", db, dyu.double().sum((0, 2, 3)), TOL[P] * 2)


FOLDED_BN_CASES = [
    # cin, cout, k, stride, pad, N, H, W, relu, res     (route)
    (64, 256, 1, 1, 0, 2, 16, 16, True, True),       # bottleneck conv3 + bn3 + identity + relu -> conv_gather
    (128, 64, 1, 1, 0, 2, 16, 16, True, False),      # conv1 + bn1 + relu -> conv_gather (64 couts)
    (64, 64, 3, 1, 1, 2, 20, 28, True, False),       # conv2 + bn2 + relu -> conv_halo<3>
    (128, 128, 3, 2, 1, 2, 18, 22, True, False),     # strided conv2 -> conv_gather
    (256, 512, 1, 2, 0, 1, 16, 24, False, False),    # downsample.0 + downsample.1 (no relu)
    (3, 64, 7, 2, 3, 2, 32, 40, True, False),        # stem conv1 + bn1 + relu -> conv_small
    (40, 24, 3, 1, 1, 1, 12, 12, True, True),        # odd channel counts -> the generic conv_igemm tiles
    (256, 256, 3, 1, 1, 1, 32, 32, True, False),     # layer3 conv2 of one 512 x 512 image -> conv_tiny
    (1024, 256, 1, 1, 0, 1, 32, 32, True, False),    # layer3 conv1 -> conv_tiny
    (256, 1024, 1, 1, 0, 1, 32, 32, True, True),     # layer3 conv3 + identity -> conv_tiny
]


@pytest.mark.parametrize("P,dt", VARIANTS, ids=VIDS)
@pytest.mark.parametrize("case", FOLDED_BN_CASES)
def test_conv_with_folded_inference_batchnorm(case, P, dt):
    """kg_planes_t.oscale: conv -> inference BatchNorm (-> + res) (-> ReLU) of KGnet.py:82-97 as ONE launch, y = act(acc * scale + shift + res)
    on the fp32 accumulators -- against float64 conv2d + batch_norm(training=False) on the same fp32 inputs."""
    DT[0] = dt
    cin, cout, k, stride, pad, N, H, W, relu, use_res = case
    g = torch.Generator().manual_seed(abs(hash(case)) % 1000 + 7 * P)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    rm, rv = torch.randn(cout, generator=g) * 0.3, torch.rand(cout, generator=g) * 2 + 0.05
    gamma[0] = 0.0                                                       # a dead channel: scale 0, output = shift
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, cout, OH, OW, generator=g) if use_res else None
    ref = F.batch_norm(F.conv2d(x.double(), w.double(), None, stride, pad), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.0, 1e-5)
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    cin_pad = ops.round_up(cin, 8)
    xr = rows_f32(x)
    if cin_pad != cin:
        xr = torch.cat([xr, torch.zeros(xr.shape[0], cin_pad - cin)], 1)
    xp = to_pt(xr.to(DEV), P, ctot=cin_pad + 16, c0=8)
    pw = PackedWeight(cout, k * k, cin_pad, DEV, xP=P, wP=P)
    pw.pack(w.to(DEV))
    scale, shift = ops.bn_scale_shift_eval(cout, gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV))
    y = alloc_pt(N * OH * OW, cout, P, DEV)
    rp = to_pt(rows_f32(res).to(DEV), P) if use_res else None
    geom = (N * OH * OW, H, W, OH, OW, k, k, stride, pad)
    for tiny in (False, True):
        y.t.zero_()
        route = ops.conv_auto(xp, pw, cout, geom, N, y=y, bias=shift, oscale=scale, res=rp, relu=relu, tiny=tiny)
        torch.cuda.synchronize()
        check(f"conv+bn {route} P={P} {case}", nchw(from_pt(y), N, OH, OW), ref, TOL[P])


@pytest.mark.parametrize("dt", [BF16, F16], ids=["bf16x2", "f16x2"])
@pytest.mark.parametrize("case", [(64, 1, 32, 32), (128, 1, 64, 64), (64, 2, 96, 160), (64, 1, 272, 500)])
def test_heads2_on_two_plane_inputs(case, dt):
    """kg_conv2d_halo_heads2 over hi + lo planes (3 products): small maps (< 128 workgroups) run ONE workgroup per (tile, head, plane product)
    and a finishing launch adds the three fp32 partial maps; larger ones walk the products inside a workgroup (head split / no split).
    Against three float64 conv2d on the fp32 inputs."""
    DT[0] = dt
    C, N, H, W = case
    g = torch.Generator().manual_seed(C + H)
    hid = torch.randn(N, 3 * C, H, W, generator=g).clamp_min(0)
    rows, vmap = ops.heads2_layout()
    pw = PackedWeight(64, 49, C, DEV, groups=3, xP=2, wP=2)
    vm = torch.tensor(vmap, dtype=torch.int32, device=DEV)
    bias64 = torch.zeros(64, device=DEV)
    refs, outs = [], []
    for k, co in enumerate((5, 10, 40)):
        w = torch.randn(co, C, 7, 7, generator=g) / math.sqrt(49 * C)
        b = torch.randn(co, generator=g)
        r = F.conv2d(hid[:, k * C:(k + 1) * C].double(), w.double(), b.double(), 1, 3)
        refs.append(torch.sigmoid(r) if k == 0 else r)
        rm = torch.tensor(rows[k], dtype=torch.int32, device=DEV)
        pw.pack_rows(w.to(DEV), rm, group=k)
        bias64[rm.long()] = b.to(DEV)
        outs.append(torch.full((N, co, H, W), float("nan"), dtype=torch.float32, device=DEV))
    xp = to_pt(rows_f32(hid).to(DEV), 2)
    ops.conv_halo_heads2(xp, pw, bias64, vm, outs[0], outs[1], outs[2], N, H, W, C)
    torch.cuda.synchronize()
    for name, o, r in zip(("kp", "short", "mid"), outs, refs):
        check(f"heads2 {case} {name}", o, r, TOL[2])


@pytest.mark.parametrize("P", [2, 3])
def test_mixed_plane_counts(P):
    DT[0] = BF16
    """The head convs of the "mixed" policy: a single-plane conv reads plane 0 of a P-plane trunk tensor, and its input gradient
    (single-plane dY and weights) is written back as P planes straight from the fp32 accumulators."""
    g = torch.Generator().manual_seed(5)
    N, C, H, W, k = 1, 64, 16, 24, 7
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, k, k, generator=g) / math.sqrt(C * k * k)
    xp = to_pt(rows_f32(x).to(DEV), P)
    x1 = PT(xp.t, 1, 0)
    pw = PackedWeight(C, k * k, C, DEV)
    pw.pack(w.to(DEV))
    y = torch.empty(N * H * W, C, dtype=BF16, device=DEV)
    geom = (N * H * W, H, W, H, W, k, k, 1, 3)
    ops.conv_auto(x1, pw, C, geom, N, y=y)
    ref = F.conv2d(xp.plane(0).float().cpu().view(N, H, W, C).permute(0, 3, 1, 2).double(), w.to(BF16).double(), None, 1, 3)
    check("bf16 conv on plane 0", nchw(y.float(), N, H, W), ref, 6e-3)
    dy = torch.randn(N, C, H, W, generator=g).to(BF16)
    pwT = PackedWeight(C, k * k, C, DEV)
    pwT.pack(w.to(DEV), transposed=True)
    dx = alloc_pt(N * H * W, C, P, DEV)
    ops.conv_auto(rows_f32(dy).to(DEV), pwT, C, geom, N, y=dx, transposed=True)
    refd = torch.nn.grad.conv2d_input((N, C, H, W), w.to(BF16).double(), dy.double(), 1, 3)
    check("P-plane store of a bf16 dgrad", nchw(from_pt(dx), N, H, W), refd, TOL[P])


@pytest.mark.parametrize("P,dt", VARIANTS, ids=VIDS)
def test_elementwise_planes(P, dt):
    DT[0] = dt
    g = torch.Generator().manual_seed(9)
    N, C, H, W = 2, 64, 18, 22
    x = torch.randn(N, C, H, W, generator=g) * 3 + 1
    r = torch.randn(N, C, H, W, generator=g)
    xp, rp = to_pt(rows_f32(x).to(DEV), P, ctot=C + 64, c0=64), to_pt(rows_f32(r).to(DEV), P)
    M = N * H * W
    # BatchNorm train forward + backward
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, invstd, scale, shift = ops.bn_stats_train(xp, C, gamma.to(DEV), beta.to(DEV), rm, rv)
    y = alloc_pt(M, C, P, DEV)
    ops.bn_apply(xp, C, scale, shift, y, res=rp, relu=True)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    pre = F.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-5) + r.double()
    ref = F.relu(pre)
    check(f"bn fwd P={P}", nchw(from_pt(y), N, H, W), ref, TOL[P])
    check("bn running_var", rv, 0.9 + 0.1 * x.double().var((0, 2, 3), unbiased=True), max(1e-5, TOL[P] / 10))
    dy = torch.randn(N, C, H, W, generator=g)
    dyu = dy * (pre.detach() > 0)
    ref.backward(dy.double())
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dx = alloc_pt(M, C, P, DEV)
    ops.bn_bwd(xp, to_pt(rows_f32(dyu.float()).to(DEV), P), C, gamma.to(DEV), mean, invstd, dg, db, dx)
    check(f"bn bwd dx P={P}", nchw(from_pt(dx), N, H, W), xd.grad, TOL[P] * 4)
    check(f"bn bwd dgamma P={P}", dg, gd.grad, TOL[P] * 4)
    # max pool
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    yp = alloc_pt(N * OH * OW, C, P, DEV)
    arg = torch.empty(N * OH * OW, C, dtype=torch.uint8, device=DEV)
    ops.maxpool_fwd(xp, yp, N, H, W, C, argmax=arg)
    xq = from_pt(xp).cpu()                                        # the value the planes hold (P = 2: 16-bit rounding of x)
    xq4 = nchw(xq, N, H, W).double().requires_grad_(True)
    refp = F.max_pool2d(xq4, 3, 2, 1)
    check(f"maxpool fwd P={P}", nchw(from_pt(yp), N, OH, OW), refp, 1e-7 if P == 3 else TOL[P])
    dyp = torch.randn(N, C, OH, OW, generator=g)
    refp.backward(dyp.double())
    dxp = alloc_pt(M, C, P, DEV)
    ops.maxpool_bwd(xp, to_pt(rows_f32(dyp).to(DEV), P), dxp, N, H, W, C)
    check(f"maxpool bwd P={P}", nchw(from_pt(dxp), N, H, W), xq4.grad, TOL[P])
    dxa = alloc_pt(M, C, P, DEV)
    ops.maxpool_bwd(xp, to_pt(rows_f32(dyp).to(DEV), P), dxa, N, H, W, C, argmax=arg)      # the stored winning taps give the same gradient
    assert torch.equal(from_pt(dxa), from_pt(dxp))
    # bilinear 2x up + backward
    up = alloc_pt(N * 4 * H * W, C, P, DEV)
    ops.bilinear_fwd(xp, up, N, H, W, 2 * H, 2 * W, C)
    xu = x.double().requires_grad_(True)
    refu = F.interpolate(xu, (2 * H, 2 * W), mode="bilinear", align_corners=False)
    check(f"bilinear fwd P={P}", nchw(from_pt(up), N, 2 * H, 2 * W), refu, TOL[P])
    dyu2 = torch.randn(N, C, 2 * H, 2 * W, generator=g)
    refu.backward(dyu2.double())
    dxu = alloc_pt(M, C, P, DEV)
    ops.bilinear_bwd(to_pt(rows_f32(dyu2).to(DEV), P), dxu, N, H, W, 2 * H, 2 * W, C)
    check(f"bilinear bwd P={P}", nchw(from_pt(dxu), N, H, W), xu.grad, TOL[P])
    # gradient join with ReLU mask
    out = alloc_pt(M, C, P, DEV)
    ops.add_rows(xp, rp, out, C, mask=rp.hi())
    check(f"add_rows P={P}", nchw(from_pt(out), N, H, W), (x.double() + r.double()) * (r.to(DT[0]).double() > 0), TOL[P])
    # image pack
    img = torch.rand(2, 3, 16, 24, generator=g) - 0.5
    ip = ops.img_pack(img.to(DEV), P, dtype=DT[0])
    if P == 1:
        ip = PT(ip)
    check(f"img_pack P={P}", from_pt(ip)[:, :3], rows_f32(img), TOL[P] / 4)
    assert float(from_pt(ip)[:, 3:].abs().max()) == 0.0


def test_crop_grad_reduce_matches_index_add_and_is_deterministic():
    DT[0] = BF16
    """kg_crop_grad_reduce (gradient of get_patches' slicing, KGnet.py:246-256) against index_add_ in fp64 on heavily overlapping
    boxes, twice bit-identically; built through SegBranch.make_plan so that the host-side bin tables are covered too."""
    from kg_instance_segmentation_amd import KGnet
    m = KGnet.resnet50(pretrained=False).to(DEV)
    rng = np.random.default_rng(3)
    N, H, W = 2, 64, 96
    feats = [torch.zeros(N, c, H >> l, W >> l, device=DEV) for l, c in enumerate((64, 64, 256, 512, 1024))]
    boxes = []
    for i in range(N):
        y1 = rng.uniform(0, H - 24, 40); x1 = rng.uniform(0, W - 24, 40)
        bb = np.stack([y1, x1, y1 + rng.uniform(6, 40, 40), x1 + rng.uniform(6, 40, 40), np.ones(40)], 1).astype(np.float32)
        bb[:10, :2] = bb[0, :2]; bb[:10, 2:4] = bb[0, 2:4] + np.arange(10)[:, None]      # nested boxes
        boxes.append(bb)
    seg = m._seg
    plan = seg.make_plan(feats, boxes)
    from kg_instance_segmentation_amd import _lib as L
    for l in range(3):
        C = feats[l].shape[1]
        rows = plan.rows[l]
        h, w = H >> l, W >> l
        for P in (1, 3):
            gfull = torch.randn(rows, C, device=DEV)
            rows_a = int(plan.row0[l][plan.nb[l] // 2])
            ga = to_pt(gfull[:rows_a], P, ctot=C + 64, c0=0) if rows_a else None
            gb = to_pt(gfull[rows_a:], P)
            outs = []
            for rep in range(2):
                out = torch.empty(N * h * w, C, device=DEV)
                L.call("kg_crop_grad_reduce", L.ptr(ops.base(ga)), ops.ld(ga) if ga is not None else 0, L.ptr(ops.base(gb)), ops.ld(gb),
                       L.c_long(rows_a), L.ptr(plan.tab_d[l]), L.ptr(plan.bin_start_d[l]), L.ptr(plan.bin_boxes_d[l]),
                       __import__("kg_instance_segmentation_amd.seg", fromlist=["BIN_SIZE"]).BIN_SIZE[l], N, h, w, C, L.ptr(out), None, 0,
                       ops.pl(a=ga if ga is not None else gb, b=gb), L.stream_ptr())
                outs.append(out.clone())
            assert torch.equal(outs[0], outs[1])
            vals = torch.cat([from_pt(ga), from_pt(gb)]) if ga is not None else from_pt(gb)
            ref = torch.zeros(N * h * w, C, dtype=torch.float64, device=DEV).index_add_(0, plan.srcrow[l][:rows].long(), vals.double())
            check(f"crop_grad_reduce level {l} P={P}", outs[0], ref, 2e-6)
            if P == 3:      # the split-bf16 output mode writes the same sums as planes
                outp = alloc_pt(N * h * w, C, 3, DEV)
                L.call("kg_crop_grad_reduce", L.ptr(ops.base(ga)), ops.ld(ga) if ga is not None else 0, L.ptr(ops.base(gb)), ops.ld(gb),
                       L.c_long(rows_a), L.ptr(plan.tab_d[l]), L.ptr(plan.bin_start_d[l]), L.ptr(plan.bin_boxes_d[l]),
                       __import__("kg_instance_segmentation_amd.seg", fromlist=["BIN_SIZE"]).BIN_SIZE[l], N, h, w, C, None, L.ptr(outp.t), ops.ld(outp),
                       ops.pl(a=ga if ga is not None else gb, b=gb, y=outp), L.stream_ptr())
                assert torch.equal(from_pt(outp), outs[0])


@pytest.mark.parametrize("P,dt", VARIANTS, ids=VIDS)
def test_seg_head_single_cout_conv_on_ragged_rows(P, dt):
    """kg_seg_conv3_c1 (seg_head.2, KGnet.py:145-147: Conv2d(64, 1, 3, padding=1) on every crop separately): the per-pixel dot-product
    kernel over a ragged list of boxes against torch's float64 conv2d of each box with zero padding at the box border."""
    DT[0] = dt
    from kg_instance_segmentation_amd._lib import c_long, ptr, stream_ptr
    g = torch.Generator().manual_seed(5)
    boxes = [(3, 3), (7, 5), (1, 9), (16, 33), (27, 27), (40, 14), (2, 2)]
    w = (torch.randn(1, 64, 3, 3, generator=g) * 0.1).to(DEV)
    b = torch.tensor([0.3], device=DEV)
    tab, xs, row0 = [], [], 0
    for i, (h, wd) in enumerate(boxes):
        tab.append([0, 0, 0, h, wd, row0, 512, 512])
        xs.append(torch.randn(h * wd, 64, generator=g))
        row0 += h * wd
    M = row0
    x32 = torch.cat(xs).to(DEV)
    xp = to_pt(x32, P)
    tabd = torch.tensor(tab, dtype=torch.int32, device=DEV)
    rd = torch.empty(M, 2, dtype=torch.int32, device=DEV); r2b = torch.empty(M, dtype=torch.int32, device=DEV); sr = torch.empty(M, dtype=torch.int32, device=DEV)
    _lib.call("kg_seg_build_rows", ptr(tabd), len(boxes), ptr(rd), ptr(r2b), ptr(sr), stream_ptr())
    y = torch.full((M,), float("nan"), device=DEV)
    _lib.call("kg_seg_conv3_c1", ptr(ops.base(xp)), ops.ld(xp), 64, ptr(w), ptr(b), ptr(rd), c_long(M), ptr(y), ops.pl(a=xp), stream_ptr(), fmt=ops.fmt_of(xp))
    xv = from_pt(xp).double().cpu()         # the stored values (the rounding of the planes is the storage's, not the kernel's)
    ref, off = [], 0
    for h, wd in boxes:
        img = xv[off:off + h * wd].view(1, h, wd, 64).permute(0, 3, 1, 2)
        ref.append(F.conv2d(img, w.double().cpu(), b.double().cpu(), padding=1).reshape(-1))
        off += h * wd
    check(f"seg_conv3_c1 P={P}", y, torch.cat(ref), 2e-6)
    DT[0] = BF16


@pytest.mark.parametrize("dt", [BF16, F16], ids=["bf16x2", "f16x2"])
@pytest.mark.parametrize("case", [(1, 8, 16), (2, 37, 53), (1, 64, 64)])
def test_weight_stationary_3x3_c64_dense(case, dt):
    """kg_conv3x3_ws (conv3_ws.hip): 64 -> 64 channels, hi + lo planes of x and w (3 products), bias + ReLU, two output planes, against
    torch's float64 conv2d on the stored values; also into a column slice of a wider two-plane buffer (the decoder writes its concat buffer)."""
    DT[0] = dt
    N, H, W = case
    g = torch.Generator().manual_seed(N * 1000 + H)
    x32 = torch.randn(N, 64, H, W, generator=g).to(DEV)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV)
    b = (torch.randn(64, generator=g) * 0.1).to(DEV)
    xp = to_pt(rows_f32(x32), 2)
    pw = PackedWeight(64, 9, 64, DEV, xP=2, wP=2)
    pw.pack(w)
    for ctot, c0 in ((64, 0), (128, 64)):
        ybuf = alloc_pt(N * H * W, ctot, 2, DEV)
        y = ybuf.cols(c0, c0 + 64)
        ops.conv_halo(xp, pw, 64, N, H, W, 3, y=y, bias=b, relu=True)
        assert _lib.last_kernel(ops.fmt_of(xp)) == "conv3_ws_kernel"
        ref = F.relu(F.conv2d(nchw(from_pt(xp), N, H, W).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1))
        check(f"conv3_ws dense {case} ctot={ctot}", nchw(from_pt(y), N, H, W), ref, TOL[2])
    DT[0] = BF16


@pytest.mark.parametrize("dt", [BF16, F16], ids=["bf16x2", "f16x2"])
def test_weight_stationary_3x3_c64_ragged(dt):
    """the same kernel over a ragged list of boxes (8 x 16 tile table of kg_host_tile_table): every box is convolved on its own (zero padding
    at the box border), one output plane (what the half-precision policies' single-plane consumers read) and two"""
    import ctypes
    DT[0] = dt
    g = torch.Generator().manual_seed(9)
    boxes = [(3, 3), (8, 16), (9, 17), (27, 27), (40, 14), (2, 33), (16, 5)]
    hs = np.array([b[0] for b in boxes], np.int32); ws_ = np.array([b[1] for b in boxes], np.int32)
    row0 = np.zeros(len(boxes) + 1, np.int64); np.cumsum(hs.astype(np.int64) * ws_, out=row0[1:])
    M = int(row0[-1])
    cnt = int((((hs + 7) // 8) * ((ws_ + 15) // 16)).sum())
    tab = np.empty((cnt, 4), np.int32)
    r0 = np.ascontiguousarray(row0[:-1])
    got = _lib.load().kg_host_tile_table(ctypes.c_void_p(hs.ctypes.data), ctypes.c_void_p(ws_.ctypes.data), ctypes.c_void_p(r0.ctypes.data), len(boxes), 8, 16,
                                         ctypes.c_void_p(tab.ctypes.data), cnt)
    assert got == cnt
    t8 = torch.from_numpy(tab).to(DEV)
    x32 = torch.randn(M, 64, generator=g).to(DEV)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV)
    b = (torch.randn(64, generator=g) * 0.1).to(DEV)
    xp = to_pt(x32, 2)
    pw = PackedWeight(64, 9, 64, DEV, xP=2, wP=2)
    pw.pack(w)
    xv = from_pt(xp).double().cpu()
    ref = []
    for (h, wd), off in zip(boxes, row0[:-1]):
        img = xv[off:off + h * wd].view(1, h, wd, 64).permute(0, 3, 1, 2)
        ref.append(F.conv2d(img, w.double().cpu(), b.double().cpu(), padding=1)[0].permute(1, 2, 0).reshape(h * wd, 64))
    ref = torch.cat(ref)
    for yP in (2, 1):
        y = alloc_pt(M, 64, yP, DEV)
        ops.conv_halo(xp, pw, 64, 0, 0, 0, 3, y=y, bias=b, relu=False, tiletab=t8, total_rows=M, tiletab8=t8)
        assert _lib.last_kernel(ops.fmt_of(xp)) == "conv3_ws_kernel"
        check(f"conv3_ws ragged yP={yP}", from_pt(y), ref, TOL[2] if yP == 2 else TOL[1])
    DT[0] = BF16


def test_wide_7x7_accumulation_is_blocked():
    """A 7x7 conv over 512 channels adds 784 MFMA results per output and plane product; in ONE fp32 accumulator the rounding error of that chain
    is 1.0e-6 relative (tools/micro/mfma_accum.hip), torch-CPU's fp32 convolution -- the reference's arithmetic -- stays at 3.7e-7.  The default
    route of hi + lo half operands (conv_halo7_w4_kernel: the chain cut at every 64-channel chunk) must be at the reference's level: relative rms
    error against float64 <= 4.5e-7 (measured 2.3e-7 .. 3.7e-7; the 8-wave kernel alone: 9.9e-7, profiles/r05_head_error_probe.txt)."""
    DT[0] = torch.float16
    cin, cout, k, N, H, W = 512, 128, 7, 1, 32, 32
    g = torch.Generator().manual_seed(5)
    x = F.relu(torch.randn(N, cin, H, W, generator=g))
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    ref = F.conv2d(x.double(), w.double(), None, 1, 3)
    c32 = F.conv2d(x, w, None, 1, 3)
    xp = to_pt(rows_f32(x).to(DEV), 2)
    pw = PackedWeight(cout, k * k, cin, DEV, xP=2, wP=2)
    pw.pack(w.to(DEV))
    y = alloc_pt(N * H * W, cout, 2, DEV)
    route = ops.conv_auto(xp, pw, cout, (N * H * W, H, W, H, W, k, k, 1, 3), N, y=y, tiny=False)
    torch.cuda.synchronize()
    got = nchw(from_pt(y), N, H, W)

    def rel(a):
        return float((a.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"route {route}: relative rms error vs float64: HIP {rel(got):.3e}, torch-CPU float32 {rel(c32):.3e}")
    assert route == "halo"
    assert rel(got) <= 4.5e-7, (rel(got), rel(c32))

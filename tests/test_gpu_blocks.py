"""GPU parity tests of sub-graphs of the engine (wiring of the tape: ReLU-mask folding, gradient
accumulation over several consumers, concat slices, ragged seg branch) against the CPU oracle's
autograd, in every storage precision (engine.PRECISIONS).  Sub-graphs are a few layers deep; stated bounds on the cosine of
every output / gradient against fp32 autograd: "bf16" >= 0.985 (the residual ~5-10 % relative L2 error is the ReLU-mask flip
noise sqrt(2^-9) of bf16 storage), trunk sub-graphs in "mixed" (hi + lo planes) >= 0.99995, "fp32" (three planes) >= 0.9999999."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from kg_instance_segmentation_amd import KGnet  # noqa: E402
from kg_instance_segmentation_amd import ops  # noqa: E402
from kg_instance_segmentation_amd.engine import Var  # noqa: E402
from kg_instance_segmentation_amd.ops import BF16, PT  # noqa: E402
from oracle import net as onet  # noqa: E402

DEV = "cuda"
THR = {"bf16": 0.985, "mixed": 0.99995, "trunk2": 0.99995, "fp32bf": 0.9999999, "fp32": 0.999999, "half": 0.998}
CUR = {"dt": BF16}      # 16-bit format of the policy under test (engine.HALF_POLICIES: IEEE half)


def to_pt(x_nchw, P):
    """fp32 NCHW (cpu) -> split-bf16 rows on the device"""
    n, c, h, w = x_nchw.shape
    r = x_nchw.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous().to(DEV)
    pt = ops.alloc_pt(n * h * w, c, P, DEV, dtype=CUR["dt"])
    ops.f32_to_planes(r, pt, c)
    return pt


def val(pt, n, h, w):
    """value of a rows tensor (PT or plain) as fp32 NCHW on the cpu"""
    if isinstance(pt, PT):
        out = torch.empty(pt.shape[0], pt.shape[1], dtype=torch.float32, device=DEV)
        ops.planes_to_f32(pt, pt.shape[1], out)
    else:
        out = pt.float()
    return out.cpu().view(n, h, w, -1).permute(0, 3, 1, 2)


def bfr(t):
    return t.to(BF16).float()


def rows_of(x):
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).to(BF16).contiguous()


def nchw_of(rows, n, h, w):
    return rows.float().cpu().view(n, h, w, -1).permute(0, 3, 1, 2)


def cos(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def check(name, got, ref, thr=0.985):
    c = cos(got, ref)
    rel = float((got.detach().double().cpu() - ref.detach().double().cpu()).norm() / (ref.detach().double().norm() + 1e-30))
    print(f"[{name}] cos={c:.5f} rel_l2={rel:.4f}")
    assert c >= thr, name


@pytest.fixture(scope="module", params=["bf16", "mixed", "fp32bf", "fp32", "half"])
def model(state_dict0, request):
    m = KGnet.resnet50(pretrained=False, precision=request.param)
    m.load_state_dict(state_dict0)
    CUR["dt"] = m._engine.dt
    return m.to(DEV).train()


def oracle_params(state_dict0, prefix_list):
    sd = {k: v.clone() for k, v in state_dict0.items()}
    for k in sd:
        if any(k.startswith(p) for p in prefix_list) and sd[k].is_floating_point() and "running" not in k:
            sd[k].requires_grad_(True)
    return sd


@pytest.mark.parametrize("block,inpl,planes,stride,H,W", [("layer1.0", 64, 64, 1, 16, 24), ("layer1.1", 256, 64, 1, 16, 24),
                                                       ("layer2.0", 256, 128, 2, 16, 24), ("layer3.0", 512, 256, 2, 12, 16)])
def test_bottleneck_block(model, state_dict0, block, inpl, planes, stride, H, W):
    N = 2
    g = torch.Generator().manual_seed(1)
    x = F.relu(bfr(torch.randn(N, inpl, H, W, generator=g)))
    eng = model._engine
    thr = THR[eng.precision]
    eng.tape, eng.param_grads = [], {}
    xv = Var(to_pt(x, eng.pt), inpl, relu=True, req=True)
    yv, OH, OW = eng.bottleneck(xv, block, N, H, W, inpl, planes, stride, block.endswith(".0"))
    dy = bfr(torch.randn(N, planes * 4, OH, OW, generator=g))
    yv.grad, yv.masked = to_pt(dy, eng.pt), False
    for fn in reversed(eng.tape):
        fn()
    gx = xv.take_grad()
    torch.cuda.synchronize()
    sd = oracle_params(state_dict0, [block + "."])
    net = onet.Net(sd, training=True)
    xd = x.clone().requires_grad_(True)
    xin = F.relu(xd)   # the block input is a ReLU output: its gradient carries that mask
    y = net.bottleneck(xin, block, stride, block.endswith(".0"))
    y.backward(dy)
    check(block + ".out", val(yv.t, N, OH, OW), y, thr)
    check(block + ".dx", val(gx, N, H, W), xd.grad, thr)
    for k, gg in eng.param_grads.items():
        check(k, gg, sd[k].grad, thr)
    eng.tape = None


def test_two_bottlenecks_batchnorm_backward_statistics_from_the_dgrad_epilogue(model, state_dict0):
    """Two consecutive bottlenecks (layer3.0 with its downsample branch -> layer3.1): the gradient of the first block's OUTPUT is completed by
    the second block's conv1 input gradient, whose epilogue adds the identity path's gradient (residual operand), applies the ReLU mask AND -- round
    6 -- sums the BatchNorm-backward statistics {sum g, sum g * xhat} of bn3 (engine.BN_BWD_STATS, kg_conv_bstats_begin); the gradients of the
    single-consumer BatchNorm outputs (bn1 / bn2 of both blocks) are completed by conv2's 3 x 3 / strided and conv3's 1 x 1 input gradients.
    Asserted: the statistics path is actually taken (>= 5 of the 7 BatchNorm layers), every output / gradient against the oracle's autograd
    within the policy's bound, and the BatchNorm parameter gradients equal to the two-pass path's to fp32 rounding."""
    from kg_instance_segmentation_amd import engine as kengine
    N, H, W, inpl, planes = 2, 12, 16, 512, 256
    g = torch.Generator().manual_seed(3)
    x = F.relu(bfr(torch.randn(N, inpl, H, W, generator=g)))
    eng = model._engine
    thr = 1.0 - 2.0 * (1.0 - THR[eng.precision])        # (two blocks deep: twice the single-block allowance)
    dy = bfr(torch.randn(N, planes * 4, H // 2, W // 2, generator=g))

    def run():
        eng.tape, eng.param_grads = [], {}
        xv = Var(to_pt(x, eng.pt), inpl, relu=True, req=True)
        y0, OH, OW = eng.bottleneck(xv, "layer3.0", N, H, W, inpl, planes, 2, True)
        y1, _, _ = eng.bottleneck(y0, "layer3.1", N, OH, OW, planes * 4, planes, 1, False)
        y1.grad, y1.masked = to_pt(dy, eng.pt), False
        y1.ngot = 0
        for fn in reversed(eng.tape):
            fn()
        gx = xv.take_grad()
        torch.cuda.synchronize()
        grads = {k: v.clone() for k, v in eng.param_grads.items()}
        eng.tape = None
        return y1, gx, grads, OH, OW
    taken = []
    orig = ops.bn_bwd

    def spy(*a, parts=None, **k):
        taken.append(parts is not None)
        print("   bn_bwd C =", a[2], "rows =", ops.base(a[0]).shape[0], "statistics from the dgrad epilogue:", parts is not None)
        return orig(*a, parts=parts, **k)
    ops.bn_bwd = spy
    try:
        y1, gx, grads, OH, OW = run()
        n_fused = sum(taken)
        kengine.BN_BWD_STATS = False
        taken.clear()
        _, gx2, grads2, _, _ = run()
        assert not any(taken)
    finally:
        ops.bn_bwd = orig
        kengine.BN_BWD_STATS = True
    print(f"BatchNorm layers whose backward statistics came from a dgrad epilogue: {n_fused} of 7")
    assert n_fused >= 5, n_fused
    sd = oracle_params(state_dict0, ["layer3.0.", "layer3.1."])
    net = onet.Net(sd, training=True)
    xd = x.clone().requires_grad_(True)
    yo = net.bottleneck(net.bottleneck(F.relu(xd), "layer3.0", 2, True), "layer3.1", 1, False)
    yo.backward(dy)
    check("out", val(y1.t, N, OH, OW), yo, thr)
    check("dx", val(gx, N, H, W), xd.grad, thr)
    for k, gg in grads.items():
        check(k, gg, sd[k].grad, thr)
        if ".bn" in k or "downsample.1" in k:          # fused statistics against the two-pass column reduction: the same sums to fp32 rounding
            rel = float((gg.double() - grads2[k].double()).norm() / (grads2[k].double().norm() + 1e-30))
            # (the fused sums see the fp32 gradient values, the column reduction their stored 16-bit planes: one 8- / 11-bit plane in every policy
            # but fp32bf, whose gradients travel in hi + lo bf16 planes)
            assert rel <= (5e-4 if eng.precision == "fp32bf" else 1e-2), (k, rel)
    check("dx fused vs two-pass", val(gx, N, H, W), val(gx2, N, H, W), 0.99999 if eng.precision == "fp32bf" else 0.999)


def test_stem_and_decoder_level(model, state_dict0):
    """conv1+bn1+relu+maxpool, then one decoder level: upsample -> 3x3 conv into a concat slice -> 1x1 refine,
    with the skip tensor also feeding a second consumer (gradient accumulation + mask folding)."""
    N, H, W = 2, 32, 48
    g = torch.Generator().manual_seed(2)
    img = bfr(torch.rand(N, 3, H, W, generator=g) - 0.5)
    eng = model._engine
    thr = THR[eng.precision]
    eng.tape, eng.param_grads = [], {}
    x8 = Var(ops.img_pack(img.to(DEV), eng.pt, dtype=CUR["dt"]), 8, relu=False, req=False)
    s1, H1, W1 = eng.conv(x8, eng.spec("conv1", 3, 64, 7, 2, 3, bias=False), N, H, W, False)
    cat1 = ops.alloc_pt(N * H1 * W1, 128, eng.pt, DEV, dtype=CUR["dt"])
    c1 = eng.bn(s1, "bn1", True, out=cat1.cols(64, 128))
    p, Hp, Wp = eng.maxpool(c1, N, H1, W1)
    # decoder level 1 style: upsample p (64 ch) to c1's size, c1_up_conv-like 3x3 (use c1_up_conv: 64->64), concat, c1_cat_refine
    u_in = eng.upsample(p, N, Hp, Wp, H1, W1)
    u, _, _ = eng.conv(u_in, eng.spec("c1_up_conv.0", 64, 64, 3, 1, 1), N, H1, W1, True, out=cat1.cols(0, 64))
    cv = eng.concat(cat1, [u, c1])
    out, _, _ = eng.conv(cv, eng.spec("c1_cat_refine.0", 128, 64, 1), N, H1, W1, True)
    dy = bfr(torch.randn(N, 64, H1, W1, generator=g))
    out.grad, out.masked = to_pt(dy, eng.pt), False
    for fn in reversed(eng.tape):
        fn()
    torch.cuda.synchronize()
    sd = oracle_params(state_dict0, ["conv1.", "bn1.", "c1_up_conv.", "c1_cat_refine."])
    net = onet.Net(sd, training=True)
    c1o = net.bn(net.conv(img, "conv1", 2, 3), "bn1", True)
    po = F.max_pool2d(c1o, 3, 2, 1)
    uo = net.conv(net.up(po, c1o), "c1_up_conv.0", 1, 1, True)
    oo = net.conv(torch.cat((uo, c1o), 1), "c1_cat_refine.0", 1, 0, True)
    oo.backward(dy)
    check("stem.c1", val(c1.t, N, H1, W1), c1o, thr)
    check("dec.out", val(out.t, N, H1, W1), oo, thr)
    for k, gg in eng.param_grads.items():
        check(k, gg, sd[k].grad, thr)
    eng.tape = None


@pytest.mark.parametrize("narrow_route", [2, 1, 0])
def test_heads_level(model, state_dict0, narrow_route, monkeypatch):
    """fused first 7x7 convs + three second convs + sigmoid, gradients from fp32 NCHW map grads.  narrow_route: how the kp / short second-layer input
    gradients run when the backward pass is single-plane (engine.NARROW_HEADS_DGRAD / KG_NARROW_HEADS_DGRAD): 2 = persistent conv7_narrow (default), 1 = the
    narrow variants of the halo kernel, 0 = the fused launch with skipped k-steps."""
    from kg_instance_segmentation_amd import engine as engine_mod
    monkeypatch.setattr(engine_mod, "NARROW_HEADS_DGRAD", narrow_route)
    model._engine.fusedT.pop("heads_c0.2T", None)      # (the packed input-gradient matrices depend on the route)
    N, H, W, C = 1, 16, 24, 64
    g = torch.Generator().manual_seed(3)
    x = F.relu(bfr(torch.randn(N, C, H, W, generator=g)))
    eng = model._engine
    thr = THR[("half" if eng.fmt else "bf16") if eng.ph == 1 else eng.precision]      # the two head layers are single-plane in "mixed" / "half"
    eng.tape, eng.param_grads = [], {}
    from kg_instance_segmentation_amd import arch
    xv = Var(to_pt(x, eng.pt), C, relu=True, req=True)
    fused = [f"{h}_head_c0.0" for h, _ in arch.HEADS]
    eng.head_slots = []
    hid, _, _ = eng.conv(xv, eng.spec("heads_c0.0", C, C, 7, 1, 3, fused=fused, P=eng.ph), N, H, W, True)
    outs = eng.heads_second(hid, 0, C, N, H, W)
    gm = [torch.randn(N, co, H, W, generator=g) * 1e-3 for _, co in arch.HEADS]
    eng.maps, eng.feats = outs, []
    gmd = [t.to(DEV) for t in gm]
    gs = ops.grad_scale(gmd, [outs[0], None, None]) if eng.fmt else None        # half build: the backward pass runs on gradients times a power of two
    pgrads = eng.backward_dec(gmd, [], gscale=gs)
    gx = xv.take_grad()
    if gs is not None:
        ops.scale_tensors(list(pgrads.values()), gs[1:2])
    torch.cuda.synchronize()
    sd = oracle_params(state_dict0, [f"{h}_head_c0." for h, _ in arch.HEADS])
    net = onet.Net(sd, training=True)
    xd = x.clone().requires_grad_(True)
    xin = F.relu(xd)
    ref = []
    for h, co in arch.HEADS:
        y = net.conv(net.conv(xin, f"{h}_head_c0.0", 1, 3, True), f"{h}_head_c0.2", 1, 3)
        ref.append(torch.sigmoid(y) if h == "kp" else y)
    torch.autograd.backward(ref, gm)
    for (h, _), o, r in zip(arch.HEADS, outs, ref):
        check(f"head.{h}.out", o.cpu(), r, thr)
    check("head.dx", val(gx, N, H, W) * (float(gs[1]) if gs is not None else 1.0), xd.grad, thr)
    assert len(pgrads) == 12
    for k, gg in pgrads.items():
        check(k, gg, sd[k].grad, thr)


def test_seg_branch_forward_backward(model, state_dict0):
    """forward_seg on given feature maps: patches, gradients w.r.t. the five feature maps and all seg parameters."""
    N, H, W = 2, 96, 128
    g = torch.Generator().manual_seed(4)
    chans = [64, 64, 256, 512, 1024]
    feats = [F.relu(bfr(torch.randn(N, c, H >> l, W >> l, generator=g))) for l, c in enumerate(chans)]
    boxes = [np.array([[10.2, 12.7, 40.5, 50.5, 1.0], [0.0, 0.0, 95.0, 127.0, 0.9], [30.5, 60.5, 37.5, 71.5, 0.8],
                       [50, 20, 52, 90, 0.7], [64.4, 100.6, 90.2, 126.9, 0.6], [2.5, 3.5, 14.5, 17.5, 0.5]], np.float32),
             np.array([[20, 30, 60, 80, 1.0], [5, 100, 25, 120, 1.0], [70.5, 8.5, 93.5, 40.5, 1.0], [1, 1, 3, 3, 1.0]], np.float32)]
    thr = THR[("half" if model._engine.fmt else "bf16") if model._engine.pseg == 1 else model._engine.precision]
    fd = [f.to(DEV).requires_grad_(True) for f in feats]       # fp32 NCHW feature maps, as the reference passes them (KGnet.py:321)
    patches, dets = model.forward_seg(fd, boxes)
    wts = [[torch.randn(p.shape, generator=g) for p in pp] for pp in patches]
    loss = sum((p * w.to(DEV)).sum() for pp, ww in zip(patches, wts) for p, w in zip(pp, ww))
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    sd = oracle_params(state_dict0, ["skip_combine.", "seg_head."])
    net = onet.Net(sd, training=True)
    fo = [f.clone().requires_grad_(True) for f in feats]
    op_, od = net.forward_seg([F.relu(f) for f in fo], boxes)
    lo = sum((p * w).sum() for pp, ww in zip(op_, wts) for p, w in zip(pp, ww))
    lo.backward()
    for i in range(N):
        assert len(patches[i]) == len(op_[i])
        for j, (a, b) in enumerate(zip(patches[i], op_[i])):
            assert tuple(a.shape) == tuple(b.shape)
            check(f"seg.patch{i}.{j}{tuple(b.shape)}", a, b, thr=max(thr, 0.999))
    for l in range(5):
        assert fd[l].grad.dtype == torch.float32
        check(f"seg.dfeat{l}", fd[l].grad, fo[l].grad, thr)
    params = dict(model.named_parameters())
    for k in sd:
        if sd[k].requires_grad:
            check(k, params[k].grad, sd[k].grad, thr)

"""Executor of KGnet's dense network (forward_dec, KGnet.py:275-318) on the HIP kernels.

A small explicit tape replaces per-op torch.autograd: every op launches its HIP kernels in forward
and records one closure that launches the matching backward kernels.  Activations are bf16
pixel-major rows; channel concatenations (torch.cat at KGnet.py:289-298) never happen -- producers
write straight into column slices of the concat buffer.  ReLU backward is folded into the
data-gradient kernels' epilogue (mask operand) whenever every contribution passes through one.

Precision (ops.PT, csrc/kg_common.h): the reference is fp32 (KGnet.py:22-29); gfx950's fast matrix path is the 16-bit MFMA with fp32
accumulation (2.5 PFLOP/s for bf16 and half alike, against 157 TFLOP/s for fp32 operands).  A tensor is therefore stored as P
16-bit "planes" whose sum is the value, and a product of two such tensors is evaluated as the MFMA products x_i * w_j with
i + j < max(xP, wP), accumulated in fp32.  A policy (PRECISIONS) names the 16-bit format and the planes of (backbone = stem conv1 +
layer1-3 with their 43 BatchNorm layers | c0_conv + top-down decoder | the two 7x7 head layers | seg branch | gradients):
  "fp32" (default), IEEE half, (2, 2, 2, 2 | 1): hi + lo half planes carry 22 significant bits, 3 products per multiply: the forward
           pass is within SURVEY 8d's fp32 tolerance (rtol 1e-4 / atol 1e-5, element-wise on pre-sigmoid logits; worst measured
           |d| / bound 0.44 eval, 0.87 train).  Half's 5-bit exponent is paid for with a constant weight scale (kg_common.h KG_WSCALE)
           and, in the backward pass -- which is LINEAR in the loss gradients, so its rounding does not compound through the batch
           statistics the way the forward's does and single half planes (11 bits, 1 product) suffice: every parameter gradient
           cosine >= 0.99998, norm within 6e-4 of the reference -- with one power-of-two gradient scale per step chosen on the device
           (csrc/gradscale.hip, ops.grad_scale);
  "fp32b2": hi + lo half planes in the backward pass as well (3 products);
  "half" / "halfmix": half mixed precision (single planes; the BatchNorm backbone on two): within rtol 2e-2 / atol 2e-2 rms;
  "fp32bf" (bfloat16, (3, 3, 3, 3 | 2)): hi + mid + lo bf16 planes == the fp32 value exactly, 6 products per multiply, backward on
           hi + lo planes (3 products): the same tolerance as "fp32" at 2.1x its cost; "fp32bf_full": 6 products in the backward too;
  "trunk2", "mixed", "bf16" (bfloat16): bf16 mixed precision -- only the BatchNorm backbone, whose train-mode batch statistics
           amplify a storage error ~x1.2 per layer (x3600 over the 45 layers at random init), in hi + lo planes.
(tools/pareto.py -> profiles/r03_pareto.json lists, per policy, the worst deviation at both tolerances of SURVEY 8d and the train-step
throughput.)
"""
import os

import torch

from . import arch, ops
from .ops import BF16, PT, PackedWeight

# planes of (backbone, c0 + decoder, heads, seg branch, GRADIENTS flowing through the network = dY operand of the input gradients
# [, x and dY of the WEIGHT gradients (their rounding does not propagate) [, W of the input gradients]])
PRECISIONS = {
    # IEEE-half rows (libkgnet_hip_f16.so): hi + lo half planes = 22 significant bits, 3 MFMA products per multiply
    "fp32": (2, 2, 2, 2, 1),          # DEFAULT: fp32-faithful forward on hi + lo half planes, backward on single half planes
    "fp32b2": (2, 2, 2, 2, 2),        # hi + lo half planes in the backward pass as well
    "half": (1, 1, 1, 1, 1),          # half mixed precision: single half planes everywhere (11 significant bits)
    "halfmix": (2, 1, 1, 1, 1),       # ... with the BatchNorm backbone (the amplifier of storage errors in train mode) on hi + lo planes
    # bfloat16 rows (libkgnet_hip.so)
    "fp32bf": (3, 3, 3, 3, 2),        # fp32 values as hi + mid + lo bf16 planes (exact), 6 products; backward on hi + lo planes (3 products)
    "fp32bf_full": (3, 3, 3, 3, 3),   # three bf16 planes in the backward pass as well (6 products everywhere: round 2's "fp32")
    "fp32bf_w1d1": (3, 3, 3, 3, 2, 1, 1),  # gradients stored in two planes, single-plane W / x / dY operands in the backward convolutions
    "fp32bf_b1": (3, 3, 3, 3, 1),     # single-plane bf16 backward
    "trunk2": (2, 2, 1, 1, 1), "mixed": (2, 1, 1, 1, 1), "bf16": (1, 1, 1, 1, 1),
}
HALF_POLICIES = ("fp32", "fp32b2", "half", "halfmix")
DEFAULT_PRECISION = "fp32"
PRE = [0]        # prepack generation counter (Engine.prepack)
NARROW_HEADS_DGRAD = int(os.environ.get("KG_NARROW_HEADS_DGRAD", "2"))      # (0: fused k1skip launch; 1: narrow halo variants; 2: persistent conv7_narrow) the kp / short second-layer input gradients on the narrow halo variants (Engine.prepare_heads2; tests flip it for the A/B)
BN_BWD_STATS = True     # BatchNorm-backward statistics in the epilogue of the input gradient that completes the BatchNorm output's gradient (Engine.conv)


def default_precision():
    p = os.environ.get("KG_PRECISION", DEFAULT_PRECISION)
    if p not in PRECISIONS:
        raise ValueError(f"KG_PRECISION must be one of {sorted(PRECISIONS)} (got {p!r})")
    return p


def trunc(t, P):
    """the first P planes of a rows tensor (plane 0 alone is the bf16 rounding of the value)"""
    if isinstance(t, PT):
        return t if t.P <= P else PT(t.t, P, t.ps)
    return t


class Var:
    """Activation (split-bf16 rows, ops.PT) + its gradient slot."""
    __slots__ = ("t", "C", "relu", "grad", "masked", "pending", "pmasked", "req", "parent", "c0", "gP", "bn_part", "boundary", "gsc",
                 "uses", "ngot", "bn_in", "bstat")
    ENG = None       # half build, during a backward pass: the engine whose running gradient scale this pass's tensors are written in

    def __init__(self, t, C, relu=False, req=True, parent=None, c0=0, gP=None):
        self.t = t if isinstance(t, PT) else PT(t)
        self.C, self.relu, self.req = C, relu, req
        self.gP = self.t.P if gP is None else gP      # planes of this tensor's gradient
        self.grad, self.masked = None, True
        self.pending, self.pmasked = None, True     # one more contribution whose addition is deferred to take_grad (fused with the mask)
        self.parent, self.c0 = parent, c0
        self.boundary = False      # half build: the backward pass re-normalises itself when this gradient is complete (Engine.renormalise)
        self.uses, self.ngot = 0, 0    # consumers of this tensor in the recorded forward / gradient contributions received so far in the backward
        self.bn_in = None          # output of a train-mode BatchNorm: (its input rows, mean, invstd) -- what the backward statistics need
        self.bstat = None          # (partials, tiles): the input gradient that completed this gradient summed the BatchNorm-backward statistics
        self.gsc = None            # half build: the scale object (device {scale, 1 / scale}) this tensor's gradient was last written in

    @property
    def rows(self):
        return self.t.shape[0]

    @property
    def P(self):
        return self.t.P

    def alloc_grad(self):
        """Buffer the data-gradient kernel should write (a slice of the parent's grad for slice vars)."""
        if self.parent is not None:
            p = self.parent
            if p.grad is None:
                p.grad = ops.alloc_pt(p.rows, p.C, p.gP, p.t.device, dtype=p.t.t.dtype)
                p.masked = True
            return p.grad.cols(self.c0, self.c0 + self.C)
        return ops.alloc_pt(self.rows, self.C, self.gP, self.t.device, dtype=self.t.t.dtype)

    def stale_scale(self):
        """half build: (scale_now, 1 / scale_then) device scalars when this Var's gradient tensors were written under an earlier (larger) running
        scale than the pass is in now, else None; marks them converted -- the caller applies the factor (<= 1: never out of range)"""
        eng = Var.ENG
        if eng is not None and eng.gscale is not None and self.gsc is not None and self.gsc is not eng.gscale:
            f = (eng.gscale[0:1], self.gsc[1:2])
            self.gsc = eng.gscale
            return f
        return None

    def to_current_scale(self):
        """half build: a gradient written under an earlier (larger) running scale is converted before it is combined or consumed"""
        f = self.stale_scale()
        if f is not None:
            ts = [(t, self.C) for t in (self.grad, self.pending) if t is not None]
            if ts:
                ops.rows_scale_multi(ts, f[0], f[1])      # * scale_now / scale_then <= 1: never out of range

    def add_grad(self, g, masked):
        f = None
        self.ngot += 1
        if Var.ENG is not None:
            if self.grad is not None and self.pending is not None and self.parent is None:
                f = self.stale_scale()       # third contribution: the join below reads both tensors anyway and converts them on the way
            elif self.grad is not None or self.pending is not None:
                self.to_current_scale()
            self.gsc = Var.ENG.gscale
        if self.parent is not None:      # written in place into the parent's buffer
            self.parent.masked = self.parent.masked and masked
            return
        if self.grad is None:
            self.grad, self.masked = g, masked
        elif self.pending is None:
            self.pending, self.pmasked = g, masked
        else:
            ops.add_rows(self.grad, self.pending, self.grad, self.C, scale=f)
            self.masked = self.masked and self.pmasked
            self.pending, self.pmasked = g, masked

    def take_grad(self):
        g = self.grad
        if g is None:
            return None
        need_mask = self.relu and not (self.masked and self.pmasked)
        if self.pending is not None:       # sum of the two contributions, their conversion into the running scale and the ReLU mask in ONE pass
            ops.add_rows(g, self.pending, g, self.C, mask=self.t.hi() if need_mask else None, scale=self.stale_scale())
            self.pending, self.pmasked = None, True
        elif need_mask:
            ops.add_rows(g, None, g, self.C, mask=self.t.hi(), scale=self.stale_scale())
        else:
            self.to_current_scale()
        if need_mask:
            self.masked = True
        return g


class ConvSpec:
    """One convolution (or a group fused along Cout that shares its input).  P = planes it multiplies (input and weights),
    gP = planes of its output's gradient (the dY operand of its input gradient)."""

    def __init__(self, names, cin, couts, k, stride, pad, bias, P, gP):
        self.names, self.cin, self.couts, self.k, self.stride, self.pad, self.has_bias = names, cin, couts, k, stride, pad, bias
        self.P, self.gP = P, gP
        self.cout = sum(couts)
        self.cin_pad = ops.round_up(cin, 8)
        self.pw = None       # forward packed weights
        self.pwT = None      # dgrad packed weights
        self.bias_cat = None
        self.versions = None
        self.used, self.need_T_last = False, False
        self.modes = set()       # BatchNorm modes (model.training) this spec has run in


class Engine:
    def __init__(self, module, precision=None):
        self.m = module
        self.set_precision(precision or default_precision())
        self.tape = None
        self.nbt = []
        self.param_grads = None
        self.head_slots = []
        self.train_steps, self.stamp = 0, ("e", 0, 0)
        self.generation = 0        # bumped by every forward_dec: a backward must belong to the latest recorded forward
        self.eval_downgrade = os.environ.get("KG_EVAL_DOWNGRADE", "0") == "1"     # opt-in: eval-mode backbone on the decoder's planes ("mixed": plain bf16 inference)
        self.fuse_eval_bn = True       # inference: conv -> bn (-> + res) (-> relu) as ONE launch (conv_bn)
        self.raw_kp_logits = False   # test hook: inference-only export of the kp LOGITS instead of sigmoid(logits) (KGnet.py:300)
        self.keep_kp_logits = False  # test hook, any mode: a second grouped launch without the sigmoid fills kp_logits[level] (the maps the losses see stay probabilities)
        self.kp_logits = {}
        self.keep_head_hidden = False   # test hook: keep the fused hidden tensor of every level's first 7x7 head layer (head_hidden[level]: rows [N*H*W, 3C])
        self.head_hidden = {}
        self.grad_store = None     # parallel.FlatGradReducer: key -> persistent fp32 view the gradient kernels write into directly
        self.grad_hook = None      # parallel.FlatGradReducer.attach: called with [(key, grad)] as backward produces them
        self.prepack_state = None  # prepack(): {"epoch", "stamp", "fp"} of weights packed ahead of the next training forward
        self.fast_stamp = None     # the prepack stamp of THIS forward when the model's fingerprint equals the one prepack() recorded
        self.phase_hook = None     # profiling (tools/phase_probe.py): called with ("fwd" | "bwd", label) where forward_dec / backward_dec enter a part of the network
        self.phase_marks = []

    def set_precision(self, precision):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)} (got {precision!r})")
        self.precision = precision
        pol = PRECISIONS[precision]
        self.pt, self.pd, self.ph, self.pseg, self.pg = pol[:5]
        self.pw = pol[5] if len(pol) > 5 else self.pg        # planes of x and dY in the weight gradients (their error does not propagate)
        self.pdw = pol[6] if len(pol) > 6 else self.pg       # planes of W in the input gradients
        self.dt = ops.F16 if precision in HALF_POLICIES else ops.BF16      # 16-bit format of every rows tensor / packed weight (ops.fmt_of)
        self.fmt = 1 if self.dt == ops.F16 else 0
        self.gscale = None         # half build: device {scale, 1 / scale} ALL live gradients of the running backward pass are expressed in
        self.param_gsc = {}        # ... and the scale each parameter gradient was produced in (divided out where it leaves)
        self.overflow_flag = getattr(self, "overflow_flag", None)     # device int32[1]: a backward pass produced a non-finite gradient (KGnet.grad_overflowed; flag_on)
        self.bpt = self.pt         # backbone planes of the CURRENT forward (see forward_dec)
        self.invalidate_caches()

    def flag_on(self, dev):
        """The sticky non-finite flag of the half-precision backward (device int32[1]), allocated on first use on `dev`: every
        ops.scale_tensors call of a backward pass -- dense, seg branch as its own autograd node, data-parallel unscale -- raises it."""
        if not self.fmt:
            return None
        if self.overflow_flag is None or self.overflow_flag.device != dev:
            self.overflow_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        return self.overflow_flag

    def invalidate_caches(self):
        """Forget every packed-weight / folded-BatchNorm copy (call after writing parameters behind PyTorch's back)."""
        self.specs, self.bn_eval, self.fusedT, self.heads2_seen = {}, {}, {}, {}
        self._heads2 = None
        self.prepack_state = None

    # ---- parameters ---------------------------------------------------------------------------
    def P(self, key):
        return self.m.get_tensor(key)

    def new_grad(self, key, like):
        """Destination of a parameter gradient: a fresh tensor, or -- data parallel -- the parameter's slot in the persistent flat
        gradient buffer (parallel.FlatGradReducer), which RCCL reduces in place and the optimizer reads in place."""
        if self.gscale is not None:
            self.param_gsc[key] = self.gscale
        if self.grad_store is not None:
            v = self.grad_store.get(key)
            if v is not None:
                return v
        return torch.empty_like(like)

    def renormalise(self, g, C):
        """Half build: re-normalise the backward pass.  The gradient tensor g (rows / PT, complete, in the running scale) is measured on
        the device and, if its largest element has grown beyond the target magnitude, multiplied in place by the power of two r < 1
        that brings it back (ops.rows_rescale); the running scale becomes scale * r (never larger: the scale only ever goes DOWN).
        Gradient tensors written earlier under a larger scale (contributions waiting on tensors further upstream: the decoder's skip
        gradients, the seg branch's crop gradients, the identity path of a bottleneck) are converted when they are next combined or
        consumed (Var.to_current_scale: * scale_now / scale_then <= 1, so a conversion can never leave the format's range);
        parameter gradients remember the scale they were produced in (new_grad) and are divided by it where they leave.
        Why: gradients grow ~2^0.9 per bottleneck through the BatchNorm backbone at random init (gamma / sigma: 2^12 from c4 to the
        stem; a 23-block layer3 of resnet101 alone would leave IEEE half's range), x316 through the BatchNorm of a dead channel, and
        x4 per level through the adjoint of the 2x bilinear upsampling when they are spatially coherent (tools/gradmax_probe.py); the
        call sites are the outputs of every bottleneck, c1, the decoder's level outputs and the seg branch's levels."""
        if self.gscale is None:
            return None
        r, self.gscale = ops.rows_rescale(g, C, self.gscale)
        return r

    def place_grad(self, key, g):
        """a small (bias / BatchNorm) gradient vector produced as a slice of a shared buffer: copied into its flat slot if there is one"""
        if self.gscale is not None:
            self.param_gsc[key] = self.gscale
        if self.grad_store is not None:
            v = self.grad_store.get(key)
            if v is not None:
                ops.flush_wgrad()        # (g may come out of a split reduction that is only recorded so far)
                v.copy_(g)
                return v
        return g

    def spec(self, key, cin, cout, k, stride=1, pad=0, bias=True, fused=None, P=None, gP=None):
        P = self.bpt if P is None else P
        s = self.specs.get((key, P))       # (one packed copy per plane count: the backbone runs on 2 planes in train mode, 1 in eval)
        if s is None:
            names = fused if fused else [key]
            couts = [cout] * len(names) if fused else [cout]
            s = ConvSpec(names, cin, couts, k, stride, pad, bias, P, min(P, self.pg) if gP is None else gP)
            self.specs[(key, P)] = s
        return s

    def prepare_all(self, train):
        """Queues the (re)packs of every conv seen in earlier steps so that the first conv launch of this step packs them
        all with one kernel (ops.PackQueue); specs / levels not seen yet are packed lazily as before."""
        ops.PACKQ.defer = True
        try:
            mode = self.m.training
            for s in self.specs.values():
                if s.used and mode in s.modes:      # (a spec packed for the other BatchNorm mode's plane count stays as it is)
                    self.prepare(s, need_T=train and s.need_T_last)
            for lvl, (C, dev) in self.heads2_seen.items():
                self.prepare_heads2(lvl, C, dev, train)
        finally:
            ops.PACKQ.defer = False

    def prepack(self):
        """Pack the weights of the NEXT training forward now (optim.Adam(..., prepack=model) calls this right after its update kernel is
        enqueued): the host builds the pack table and the GPU runs the batched pack while the host would otherwise sit in the step's
        loss read-back (train.py:156), instead of between that read-back and the first kernel of the next step (the GPU idles there).
        The next recorded forward reuses these copies iff nothing has written parameters since (ops.PARAM_EPOCH unchanged, tensor
        versions / pointers unchanged); anything else -- another optimizer, an in-place edit, an eval forward in between -- repacks as
        before.  Writes through `.data` after the optimizer step and before the forward are invisible to both checks: do not combine
        them with prepack."""
        if not self.m.training or not self.specs:
            return
        PRE[0] += 1
        stamp = ("p", PRE[0], 0)
        saved, self.stamp = self.stamp, stamp
        seg = self.m._seg
        seg_saved, seg.stamp = seg.stamp, stamp
        self.fast_stamp = seg.fast_stamp = None      # (the fast path of prepare() belongs to the forward that proved its fingerprint: here everything repacks)
        try:
            self.prepare_all(True)
            seg.prepare_all(True)
            ops.flush_packs(hold=True)      # tables built and uploaded now; the launch follows the loss read-back (ops.launch_held_packs)
        finally:
            self.stamp, seg.stamp = saved, seg_saved
        self.prepack_state = {"epoch": ops.PARAM_EPOCH[0], "stamp": stamp, "fp": self.fingerprint()}

    def fingerprint(self):
        """(tensor versions, data pointers) of every parameter: what the per-conv validity keys of prepare() are built from, taken ONCE for the
        whole model (~60 us).  A training forward whose fingerprint equals the one prepack() recorded packs nothing and skips the ~175 per-conv
        key comparisons (1.2 ms of host time in front of the step's first kernel -- the GPU idles there after the loss read-back)."""
        ts = [self.m.get_tensor(k) for k in self.m._param_keys]
        return tuple(t._version for t in ts), tuple(t.data_ptr() for t in ts)

    def prepare(self, s, need_T):
        """(Re)pack weights: every recorded (training) forward repacks -- optimizers that write through `.data` or fused
        multi-tensor kernels do not bump tensor versions, so versions alone would leave stale bf16 copies -- and inference
        repacks when a tensor version / pointer changed, when it follows a training step, or when the parameter epoch moved
        (self.stamp; ops.PARAM_EPOCH is bumped by this package's optimizer and by every train-mode BatchNorm update)."""
        s.used, s.need_T_last = True, need_T
        s.modes.add(self.m.training)
        if (self.fast_stamp is not None and s.versions is not None and s.versions[:3] == self.fast_stamp
                and (not need_T or (s.pwT is not None and not getattr(s.pwT, "stale", True)))):
            return        # packed by prepack() under this very stamp and the model's fingerprint has not moved since (forward_dec)
        ws = [self.P(n + ".weight") for n in s.names]
        ver = self.stamp + tuple(w._version for w in ws) + tuple(w.data_ptr() for w in ws)
        dev = ws[0].device
        taps = s.k * s.k
        if s.pw is None or s.versions != ver or s.pw.buf.device != dev:
            if s.pw is None or s.pw.buf.device != dev:
                s.pw = PackedWeight(s.cout, taps, s.cin_pad, dev, xP=s.P, wP=s.P, dtype=self.dt)
                s.pw.cin_real = s.cin          # (FLOP accounting of bench.py: real channels, not the padding / plane copies)
                s.pwT = None
            r = 0
            for w, co in zip(ws, s.couts):
                s.pw.pack(w.detach(), row0=r)
                r += co
            if s.has_bias:
                bs = [self.P(n + ".bias").detach() for n in s.names]
                s.bias_cat = bs[0] if len(bs) == 1 else torch.cat(bs)
            s.versions = ver
            if s.pwT is not None:
                s.pwT.stale = True
        if need_T and (s.pwT is None or getattr(s.pwT, "stale", True)):
            if s.pwT is None:
                s.pwT = PackedWeight(s.cin, taps, ops.round_up(s.cout, 8), dev, xP=s.gP, wP=min(s.P, s.gP, self.pdw), dtype=self.dt)    # backward operands: gP planes
                s.pwT.cin_real = s.cout
            r = 0
            for w, co in zip(ws, s.couts):
                s.pwT.pack(w.detach(), c0=r, transposed=True)
                r += co
            s.pwT.stale = False

    # ---- ops ------------------------------------------------------------------------------------
    def conv(self, xv, s, N, H, W, relu, out=None, y_f32=None, tile=0, oP=None, bn_stats=False, affine=None, res=None):
        """xv: Var over [N*H*W, >=cin_pad]; returns Var over [N*OH*OW, cout] (or fp32 NCHW when y_f32).
        oP: planes of the output (default: the conv's own precision).  bn_stats: the output feeds a train-mode BatchNorm -- the conv
        kernel also writes the statistics partials of its output when it can (ops.conv_stats_begin); the Var then carries them.
        affine = (scale, shift) fp32 [cout] / res (Var): inference only (conv_bn) -- y = act(conv * scale + shift + res) in the conv's epilogue."""
        train = self.tape is not None
        assert not (train and (affine is not None or res is not None))
        xv.uses += 1
        self.prepare(s, need_T=train and xv.req)
        OH = (H + 2 * s.pad - s.k) // s.stride + 1
        OW = (W + 2 * s.pad - s.k) // s.stride + 1
        M = N * OH * OW
        dev = xv.t.device
        oP = s.P if oP is None else oP
        if y_f32 is None and out is None:
            out = ops.alloc_pt(M, s.cout, oP, dev, dtype=self.dt)
        geom = (M, H, W, OH, OW, s.k, s.k, s.stride, s.pad)
        xin = trunc(xv.t, s.P)
        arm = bn_stats and ops.CONV_BN_STATS and self.m.training and y_f32 is None and not relu
        part = ops.conv_stats_begin(dev, self.fmt) if arm else None
        nb = 0
        try:
            if affine is not None:
                assert not s.has_bias
                ops.conv_auto(xin, s.pw, s.cout, geom, N, y=out, bias=affine[1], oscale=affine[0], res=res.t if res is not None else None, relu=relu, tile=tile)
            else:
                ops.conv_auto(xin, s.pw, s.cout, geom, N, y=out, y_f32=y_f32, bias=s.bias_cat if s.has_bias else None, relu=relu, tile=tile, tiny=not arm)
        finally:
            if arm:          # (always disarm: an exception in the launch must not leave the side channel armed for the next conv)
                nb = ops.conv_stats_end(self.fmt)
        yv = Var(out, s.cout, relu=relu, gP=s.gP)
        if arm:
            yv.bn_part = (part, nb) if nb > 0 else None
        if train:
            def bwd():
                g = yv.take_grad()
                if g is None:
                    return
                if yv.boundary:
                    self.renormalise(g, yv.C)
                g = trunc(g, s.gP)       # (an output stored in more planes than this conv computes in: its gradient is rounded alike)
                grads, off = [], 0
                for n, co in zip(s.names, s.couts):
                    w = self.P(n + ".weight")
                    gw = self.new_grad(n + ".weight", w)
                    self.param_grads[n + ".weight"] = gw
                    grads.append((gw, off, co))
                    off += co
                db = torch.empty(s.cout, dtype=torch.float32, device=dev) if s.has_bias else None
                ops.conv_wgrad(trunc(xin, min(s.gP, self.pw)), trunc(g, self.pw), s.cin, s.cout, geom, grads, N=N, bias_out=db)
                if s.has_bias:
                    off = 0
                    for n, co in zip(s.names, s.couts):
                        self.param_grads[n + ".bias"] = self.place_grad(n + ".bias", db[off:off + co])
                        off += co
                if xv.req:
                    existing = xv.grad if xv.parent is None else None
                    if existing is not None:
                        xv.to_current_scale()       # (half build: the dgrad epilogue adds onto it in the running scale)
                    dx = existing if existing is not None else xv.alloc_grad()
                    gin = (N * H * W, OH, OW, H, W, s.k, s.k, s.stride, s.pad)
                    # xv is the output of a train-mode BatchNorm and this input gradient COMPLETES its gradient (every other consumer has
                    # delivered: they sit in `existing`, which the epilogue adds): the launch also sums the BatchNorm-backward statistics of
                    # the rows it stores (ops.conv_bstats_begin) -- bn()'s backward then skips its column reduction over x and dy
                    arm = (BN_BWD_STATS and xv.bn_in is not None and xv.parent is None and xv.pending is None and xv.ngot == xv.uses - 1
                           and (existing is not None or xv.uses == 1) and s.cin % 64 == 0)
                    part = ops.conv_bstats_begin(xv.bn_in[0], xv.bn_in[1], xv.bn_in[2], N * H * W, s.cin, N) if arm else None
                    try:
                        ops.conv_auto(g, s.pwT, s.cin, gin, N, y=dx, res=existing, mask=xv.t.hi() if xv.relu else None, transposed=True, tiny=not arm)
                    finally:
                        if arm:
                            nb = ops.conv_stats_end(self.fmt)
                            xv.bstat = (part, nb) if nb > 0 else None
                    if existing is None:
                        xv.add_grad(dx, masked=xv.relu)
                    else:
                        xv.ngot += 1
                        xv.masked = xv.masked or xv.relu
            self.tape.append(bwd)
        return yv, OH, OW

    def bn_eval_affine(self, p, C):
        """inference-mode BatchNorm as y = x * scale + shift: the pair only changes with the parameters -- cached under the same validity key
        as the packed weights"""
        gamma, beta = self.P(p + ".weight"), self.P(p + ".bias")
        rm, rv = self.P(p + ".running_mean"), self.P(p + ".running_var")
        ver = self.stamp + tuple(t._version for t in (gamma, beta, rm, rv)) + tuple(t.data_ptr() for t in (gamma, beta, rm, rv))
        hit = self.bn_eval.get(p)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.bn_scale_shift_eval(C, gamma.detach(), beta.detach(), rm, rv))
            self.bn_eval[p] = hit
        return hit[1]

    def conv_bn(self, xv, s, bnp, N, H, W, relu, res=None, out=None, bn_stats=True):
        """conv -> BatchNorm (-> + res) (-> ReLU), KGnet.py:82-97.  Inference (running statistics, nothing recorded): ONE launch -- the
        BatchNorm is a per-channel affine map of the conv's fp32 accumulators (kg_planes_t.oscale + bias; the conv output is never stored
        and re-read), residual and ReLU ride in the same epilogue.  Training: conv (+ statistics in its epilogue), then bn()."""
        if self.tape is None and not self.m.training and self.fuse_eval_bn:
            if out is None:
                OH, OW = (H + 2 * s.pad - s.k) // s.stride + 1, (W + 2 * s.pad - s.k) // s.stride + 1
                out = ops.alloc_pt(N * OH * OW, s.cout, self.bpt, xv.t.device, dtype=self.dt)
            yv, OH, OW = self.conv(xv, s, N, H, W, relu, out=out, affine=self.bn_eval_affine(bnp, s.cout), res=res)
            yv.gP = min(self.bpt, self.pg)
            return yv, OH, OW
        y, OH, OW = self.conv(xv, s, N, H, W, False, bn_stats=bn_stats)
        return self.bn(y, bnp, relu, res=res, out=out), OH, OW

    def bn(self, xv, p, relu, res=None, out=None):
        C = xv.C
        dev = xv.t.device
        gamma, beta = self.P(p + ".weight"), self.P(p + ".bias")
        rm, rv = self.P(p + ".running_mean"), self.P(p + ".running_var")
        if out is None:
            out = ops.alloc_pt(xv.rows, C, self.bpt, dev, dtype=self.dt)
        if self.m.training:
            bp = getattr(xv, "bn_part", None)
            if bp is not None:       # the producing conv's epilogue already summed the rows (conv_args.h): second stage only
                mean, invstd, scale, shift = ops.bn_finalize_train(bp[0], bp[1], xv.rows, C, gamma.detach(), beta.detach(), rm, rv)
                xv.bn_part = None
            else:
                mean, invstd, scale, shift = ops.bn_stats_train(xv.t, C, gamma.detach(), beta.detach(), rm, rv)
            self.nbt.append(self.P(p + ".num_batches_tracked"))      # += 1 for all 43 layers in one launch at the end of forward_dec
            self.stats_written = True
        else:
            scale, shift = self.bn_eval_affine(p, C)
            mean = invstd = None
        ops.bn_apply(xv.t, C, scale, shift, out, res=res.t if res is not None else None, relu=relu)
        yv = Var(out, C, relu=relu, gP=min(self.bpt, self.pg))
        xv.uses += 1
        if res is not None:
            res.uses += 1
        if self.tape is not None:
            if mean is None:
                raise NotImplementedError("backward through eval-mode BatchNorm is not supported; call model.train()")
            yv.bn_in = (xv.t, mean, invstd)

            def bwd():
                # statistics summed by the input gradient that completed yv's gradient (conv's backward): valid only if nothing touches the
                # tensor between that launch and here -- no pending contribution, no ReLU mask left to apply, no scale conversion
                bst, yv.bstat = yv.bstat, None
                if bst is not None and (yv.pending is not None or (yv.relu and not (yv.masked and yv.pmasked))
                                        or (self.gscale is not None and yv.gsc is not None and yv.gsc is not self.gscale)):
                    bst = None
                g = yv.take_grad()
                if g is None:
                    return
                r = self.renormalise(g, yv.C) if yv.boundary else None      # (the partials were taken before this factor: bn_bwd multiplies the sums)
                dg = self.new_grad(p + ".weight", gamma)
                db = self.new_grad(p + ".bias", beta)
                dx = ops.alloc_pt(xv.rows, C, xv.gP, dev, dtype=self.dt)
                ops.bn_bwd(xv.t, g, C, gamma.detach(), mean, invstd, dg, db, dx, parts=bst, parts_scale=r if bst is not None else None)
                self.param_grads[p + ".weight"] = dg
                self.param_grads[p + ".bias"] = db
                xv.add_grad(dx, masked=True)
                if res is not None:
                    res.add_grad(g, masked=False)
            self.tape.append(bwd)
        return yv

    def maxpool(self, xv, N, H, W):
        C = xv.C
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = ops.alloc_pt(N * OH * OW, C, xv.P, xv.t.device, dtype=self.dt)
        arg = torch.empty(N * OH * OW, C, dtype=torch.uint8, device=xv.t.device) if self.tape is not None else None
        ops.maxpool_fwd(xv.t, out, N, H, W, C, argmax=arg)
        xv.uses += 1
        yv = Var(out, C, relu=False, gP=xv.gP)
        if self.tape is not None:
            def bwd():
                g = yv.take_grad()
                if g is None:
                    return
                dx = ops.alloc_pt(xv.rows, C, xv.gP, xv.t.device, dtype=self.dt)
                ops.maxpool_bwd(xv.t, g, dx, N, H, W, C, argmax=arg)
                xv.add_grad(dx, masked=False)
            self.tape.append(bwd)
        return yv, OH, OW

    def upsample(self, xv, N, IH, IW, OH, OW, P=None):
        C = xv.C
        out = ops.alloc_pt(N * OH * OW, C, xv.P if P is None else P, xv.t.device, dtype=self.dt)
        ops.bilinear_fwd(xv.t, out, N, IH, IW, OH, OW, C)
        xv.uses += 1
        yv = Var(out, C, relu=False, gP=min(out.P, self.pg))
        if self.tape is not None:
            def bwd():
                g = yv.take_grad()
                if g is None:
                    return
                dx = ops.alloc_pt(xv.rows, C, xv.gP, xv.t.device, dtype=self.dt)
                ops.bilinear_bwd(g, dx, N, IH, IW, OH, OW, C)
                xv.add_grad(dx, masked=False)
            self.tape.append(bwd)
        return yv

    def concat(self, buf, parts):
        """buf = PT [rows, sum C]; parts = Vars whose .t are the column slices of buf (already written)."""
        cv = Var(buf, buf.shape[1], relu=all(p.relu for p in parts), gP=max(p.gP for p in parts))
        for p in parts:
            p.uses += 1
        if self.tape is not None:
            def bwd():
                g = cv.take_grad()
                if g is None:
                    return
                c = 0
                for p in parts:
                    p.add_grad(g.cols(c, c + p.C), masked=cv.relu)
                    c += p.C
            self.tape.append(bwd)
        return cv

    # ---- the network ------------------------------------------------------------------------------
    def bottleneck(self, xv, p, N, H, W, inplanes, planes, stride, has_ds, out=None):
        # (every conv here feeds a BatchNorm directly: bn_stats -- one shared partial buffer, so each conv is followed by ITS bn)
        a, _, _ = self.conv_bn(xv, self.spec(p + ".conv1", inplanes, planes, 1, bias=False), p + ".bn1", N, H, W, True)
        b, OH, OW = self.conv_bn(a, self.spec(p + ".conv2", planes, planes, 3, stride, 1, bias=False), p + ".bn2", N, H, W, True)
        if has_ds:
            idt, _, _ = self.conv_bn(xv, self.spec(p + ".downsample.0", inplanes, planes * 4, 1, stride, 0, bias=False), p + ".downsample.1", N, H, W, False)
        else:
            idt = xv
        y, _, _ = self.conv_bn(b, self.spec(p + ".conv3", planes, planes * 4, 1, bias=False), p + ".bn3", N, OH, OW, True, res=idt, out=out)
        return y, OH, OW

    def _phase(self, label):
        if self.phase_hook is not None:
            self.phase_marks.append((label, len(self.tape) if self.tape is not None else 0))
            self.phase_hook("fwd", label)

    def forward_dec(self, img, record):
        """img fp32 [N,3,H,W].  Returns (maps: 12 fp32 NCHW tensors, feats: 5 Vars, dims)."""
        N, _, H, W = img.shape
        dev = img.device
        if record and self.raw_kp_logits:
            raise RuntimeError("raw_kp_logits is an inference-only test hook (the losses expect probabilities)")
        ops.launch_held_packs()           # (the batched pack prepack() prepared: first launch of the step)
        self.tape = [] if record else None
        self.generation += 1
        self.param_grads = {}
        self.nbt = []
        self.stats_written = False
        if record:
            self.train_steps += 1
            if self.grad_store is not None:
                self.grad_store.begin_step()
        self.stamp = ("t" if record else "e", self.train_steps, ops.PARAM_EPOCH[0])
        pre, self.prepack_state = self.prepack_state, None
        self.m._seg.prepack_stamp = None
        self.fast_stamp = None
        if record and pre is not None and pre["epoch"] == ops.PARAM_EPOCH[0]:      # weights packed ahead by prepack(): same stamp -> nothing to repack
            self.stamp = pre["stamp"]
            self.m._seg.prepack_stamp = pre["stamp"]
            if pre.get("fp") == self.fingerprint():       # ... and no tensor was written or replaced since: prepare() takes its fast path
                self.fast_stamp = pre["stamp"]
        self.m._seg.fast_stamp = self.fast_stamp
        if self.fast_stamp is None:
            self.prepare_all(record)
        # Backbone planes: the policy's, in train AND eval mode.  (`eval_downgrade` -- opt-in, KG_EVAL_DOWNGRADE=1 or
        # engine.eval_downgrade = True -- runs an eval-mode backbone on the decoder's planes instead: with running statistics the
        # BatchNorm amplifier of storage errors is gone, and "mixed" batch-1 inference is 25 % faster in plain bf16.)
        self.bpt = self.pt if (self.m.training or not self.eval_downgrade) else min(self.pt, max(self.pd, 1))
        pt, pd = self.bpt, self.pd
        self.phase_marks = []
        self._phase("c0_conv")
        x8 = Var(ops.img_pack(img, max(pt, pd), dtype=self.dt), 8, relu=False, req=False)
        dims = [(H, W)]
        # c0 branch (KGnet.py:276): both convs at full resolution; c0 lands in cat0[:, 64:128]
        cat0 = ops.alloc_pt(N * H * W, 128, pd, dev, dtype=self.dt)
        c0a, _, _ = self.conv(x8, self.spec("c0_conv.0", 3, 64, 3, 1, 1, P=pd), N, H, W, True)
        c0, _, _ = self.conv(c0a, self.spec("c0_conv.2", 64, 64, 3, 1, 1, P=pd), N, H, W, True, out=cat0.cols(64, 128))
        # stem (KGnet.py:278-282)
        self._phase("stem")
        H1, W1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        cat1 = ops.alloc_pt(N * H1 * W1, 128, pt, dev, dtype=self.dt)
        c1, _, _ = self.conv_bn(x8, self.spec("conv1", 3, 64, 7, 2, 3, bias=False), "bn1", N, H, W, True, out=cat1.cols(64, 128), bn_stats=False)
        f, Hc, Wc = self.maxpool(c1, N, H1, W1)
        dims.append((H1, W1))
        feats = [c0, c1]
        cats = [cat0, cat1]
        for li, (name, inplanes, planes, blocks, stride) in enumerate(self.m.layers_tab):
            self._phase(name)
            for b in range(blocks):
                st = stride if b == 0 else 1
                Ho, Wo = (Hc - 1) // st + 1, (Wc - 1) // st + 1
                out = None
                if b == blocks - 1 and li < 2:  # c2 / c3 land in their concat buffers
                    catb = ops.alloc_pt(N * Ho * Wo, planes * 8, pt, dev, dtype=self.dt)
                    cats.append(catb)
                    out = catb.cols(planes * 4, planes * 8)
                f, Hc, Wc = self.bottleneck(f, f"{name}.{b}", N, Hc, Wc, inplanes if b == 0 else planes * 4, planes, st, b == 0, out=out)
                f.boundary = True            # (half build: every bottleneck output is a re-normalisation point of the backward pass, renormalise)
            feats.append(f)
            dims.append((Hc, Wc))
        # top-down decoder (KGnet.py:288-298)
        self._phase("decoder")
        cur = feats[4]
        catv = {}
        up_ch = {3: (1024, 512), 2: (512, 256), 1: (256, 64), 0: (64, 64)}
        for lvl in (3, 2, 1, 0):
            (IH, IW), (OH, OW) = dims[lvl + 1], dims[lvl]
            cin, cu = up_ch[lvl]
            u_in = self.upsample(cur, N, IH, IW, OH, OW, P=pd)
            buf = cats[lvl]        # (planes of the skip tensor: the decoder convs write / read their own pd planes of it)
            u, _, _ = self.conv(u_in, self.spec(f"c{lvl + 1}_up_conv.0", cin, cu, 3, 1, 1, P=pd), N, OH, OW, True, out=buf.cols(0, cu), oP=buf.P)
            cv = self.concat(buf, [u, feats[lvl]])
            cur, _, _ = self.conv(cv, self.spec(f"c{lvl}_cat_refine.0", buf.shape[1], cu, 1, P=pd), N, OH, OW, True)
            cur.boundary = lvl > 0           # (the adjoint of the 2x upsampling below this level multiplies coherent gradients by 4)
            catv[lvl] = cur
        # heads (KGnet.py:300-316): three first 7x7 convs fused along Cout, three second convs
        maps = []
        self.head_slots = []
        for lvl in range(4):
            self._phase(f"heads_c{lvl}")
            C = arch.HEAD_CH[lvl]
            Hh, Wh = dims[lvl]
            fused = [f"{h}_head_c{lvl}.0" for h, _ in arch.HEADS]
            hid, _, _ = self.conv(catv[lvl], self.spec(f"heads_c{lvl}.0", C, C, 7, 1, 3, fused=fused, P=self.ph), N, Hh, Wh, True)
            if self.keep_head_hidden:
                self.head_hidden[lvl] = hid.t
            outs = self.heads_second(hid, lvl, C, N, Hh, Wh)
            maps.extend(outs)
        self._phase("end")
        if self.nbt:
            torch._foreach_add_(self.nbt, 1)
            self.nbt = []
        if self.stats_written:
            ops.PARAM_EPOCH[0] += 1      # the running statistics moved: folded eval-mode scale / shift copies are stale
        self.feats, self.dims, self.N, self.maps = feats, dims, N, maps
        for fv in feats:
            fv.uses += 1 << 20         # (c0 .. c4 may receive gradients from outside -- the seg branch, a caller's feat_grads: never "completed" by a conv)
        feats[1].boundary = True         # (c1; the bottleneck and decoder level outputs were marked where they were made)
        return maps, feats, dims

    HEAD_OFF = (0, 8, 24)      # channel offsets of the kp / short / mid gradients in the fused [rows, 64] dY buffer
    HEAD_PAD = (8, 16, 40)

    def heads2_tables(self, dev):
        """Device tables of the fused second-layer head conv (ops.heads2_layout)."""
        t = getattr(self, "_heads2", None)
        if t is None or t["vmap"].device != dev:
            rows, vmap = ops.heads2_layout()
            vm = torch.tensor(vmap, dtype=torch.int32)
            t = {"rows": [torch.tensor(r, dtype=torch.int32, device=dev) for r in rows], "vmap": vm.to(dev),
                 "bias_idx": torch.where(vm < 0, torch.full_like(vm, 55), vm).long().to(dev),
                 "zero1": torch.zeros(1, dtype=torch.float32, device=dev)}
            self._heads2 = t
        return t

    def prepare_heads2(self, lvl, C, dev, train):
        """Packed weights of the second-layer head convs of level lvl: forward (virtual-cout layout, + bias64) and, when
        training, the block-structured transposed matrix of the fused input gradient."""
        self.heads2_seen[lvl] = (C, dev)
        key = f"heads_c{lvl}.2F"
        ent = self.fusedT.get(key)
        if self.fast_stamp is not None and ent is not None and ent[0][:3] == self.fast_stamp and ent[1].buf.device == dev:
            entT = self.fusedT.get(f"heads_c{lvl}.2T") if train else None
            if not train or (entT is not None and entT[0][:3] == self.fast_stamp and entT[1].buf.device == dev):
                return ent[1], ent[2], (entT[1], entT[2]) if train else None       # (prepacked under this stamp, fingerprint unchanged: see prepare)
        ph = self.ph
        specs = [self.spec(f"{h}_head_c{lvl}.2", C, co, 7, 1, 3, P=ph) for h, co in arch.HEADS]
        ws = [self.P(s.names[0] + ".weight") for s in specs]
        bs = [self.P(s.names[0] + ".bias") for s in specs]
        ver = self.stamp + tuple(t._version for t in ws + bs) + tuple(t.data_ptr() for t in ws + bs)
        if ent is None or ent[0] != ver or ent[1].buf.device != dev:
            lay = self.heads2_tables(dev)
            pwF = ent[1] if ent is not None and ent[1].buf.device == dev else PackedWeight(64, 49, C, dev, xP=ph, wP=ph, groups=3, dtype=self.dt)
            for k, w in enumerate(ws):
                pwF.pack_rows(w.detach(), lay["rows"][k], group=k)
            bias64 = torch.cat([b.detach() for b in bs] + [lay["zero1"]])[lay["bias_idx"]]
            self.fusedT[key] = (ver, pwF, bias64)
        _, pwF, bias64 = self.fusedT[key]
        pwT = None
        if train:
            key = f"heads_c{lvl}.2T"
            ent = self.fusedT.get(key)
            ver = self.stamp + tuple(w._version for w in ws) + tuple(w.data_ptr() for w in ws)
            if ent is None or ent[0] != ver or ent[1].buf.device != dev:
                gph = min(ph, self.pg)
                reuse = ent is not None and ent[1].buf.device == dev
                pwT = ent[1] if reuse else PackedWeight(3 * C, 49, 64, dev, xP=gph, wP=min(gph, self.pdw), dtype=self.dt)
                # single-plane backward: the kp / short input gradients run on the narrow variants of the halo kernel (4 / 2 kernel columns per MFMA
                # k-step instead of k-steps that are 3/4 or 1/2 zeros: ops.conv_halo(narrow=...)) from their own packed matrices; rows of pwT they do
                # not use stay unpacked
                narrow = NARROW_HEADS_DGRAD and gph == 1 and min(gph, self.pdw) == 1
                pwN = None
                if narrow:
                    pwN = ent[2] if reuse and ent[2] is not None else (PackedWeight(C, 7, 64, dev, dtype=self.dt), PackedWeight(C, 14, 64, dev, dtype=self.dt))
                    pwN[0].pack_narrow(ws[0].detach(), 8)
                    pwN[1].pack_narrow(ws[1].detach(), 16)
                for k, w in enumerate(ws):
                    if narrow and k < 2:
                        continue
                    pwT.pack(w.detach(), row0=k * C, c0=self.HEAD_OFF[k], transposed=True)
                self.fusedT[key] = (ver, pwT, pwN)
            pwT = (self.fusedT[key][1], self.fusedT[key][2])
        return pwF, bias64, pwT

    def heads_second(self, hid, lvl, C, N, H, W):
        """The three second 7x7 head convs (KGnet.py:161-209, `.2` layers) on the slices of the fused hidden tensor
        hid [rows, 3C].  Forward: ONE grouped launch (kg_conv2d_halo_heads2) exporting the three fp32 NCHW maps.  Backward: per-head weight/bias gradients, and ONE
        fused input-gradient conv: the three map gradients are packed side by side into a [rows, 64] buffer
        (8 | 16 | 40 channels) and multiplied by a block-structured transposed weight matrix [3C][49][64], which
        runs on the fast LDS-halo kernel instead of three tiny-K gather convs."""
        dev = hid.t.device
        train = self.tape is not None
        specs = [self.spec(f"{h}_head_c{lvl}.2", C, co, 7, 1, 3, P=self.ph) for h, co in arch.HEADS]
        pwF, bias64, pwTN = self.prepare_heads2(lvl, C, dev, train)
        pwT, pwN = pwTN if pwTN is not None else (None, None)
        outs = [torch.empty(N, co, H, W, dtype=torch.float32, device=dev) for _, co in arch.HEADS]
        ops.conv_halo_heads2(hid.t, pwF, bias64, self.heads2_tables(dev)["vmap"], outs[0], outs[1], outs[2], N, H, W, C,
                             kp_sigmoid=not self.raw_kp_logits)
        if self.keep_kp_logits:
            side = [torch.empty_like(o) for o in outs]
            ops.conv_halo_heads2(hid.t, pwF, bias64, self.heads2_tables(dev)["vmap"], side[0], side[1], side[2], N, H, W, C, kp_sigmoid=False)
            self.kp_logits[lvl] = side[0]
        slot = {"grad": None}
        self.head_slots.append((slot, lvl, N, H, W))
        if train:
            geom = (N * H * W, H, W, H, W, 7, 7, 1, 3)

            def bwd():
                g = slot["grad"]      # PT [rows, 64] packed map gradients (set by backward_dec)
                if g is None:
                    return
                for k, ((h, co), s) in enumerate(zip(arch.HEADS, specs)):
                    gk = g.cols(self.HEAD_OFF[k], self.HEAD_OFF[k] + self.HEAD_PAD[k])
                    w = self.P(s.names[0] + ".weight")
                    gw = self.new_grad(s.names[0] + ".weight", w)
                    db = self.new_grad(s.names[0] + ".bias", self.P(s.names[0] + ".bias"))      # bias gradient: a free unit of the wgrad kernel
                    ops.conv_wgrad(trunc(hid.t, min(hid.gP, self.pw)).cols(k * C, (k + 1) * C), trunc(gk, self.pw), C, co, geom, [(gw, 0, co)], N=N, bias_out=db)
                    self.param_grads[s.names[0] + ".weight"] = gw
                    self.param_grads[s.names[0] + ".bias"] = db
                dh = ops.alloc_pt(hid.rows, 3 * C, hid.gP, dev, dtype=self.dt)
                # kp / short cout blocks only see dY channels 0..23 (k-step 1 of the chunk skipped); mid sees 24..63
                if pwN is not None and g.P == 1 and NARROW_HEADS_DGRAD >= 2 and dh.P == 1:
                    ops.conv7_narrow(g, pwN[0], C, N, H, W, dh.cols(0, C), mask=hid.t.hi()[:, :C], chan_lo=0, chan_slot=8, algo_cin=5)
                    ops.conv7_narrow(g, pwN[1], C, N, H, W, dh.cols(C, 2 * C), mask=hid.t.hi()[:, C:2 * C], chan_lo=8, chan_slot=16, algo_cin=10)
                elif pwN is not None and g.P == 1:
                    ops.conv_halo(g, pwN[0], C, N, H, W, 7, y=dh.cols(0, C), mask=hid.t.hi()[:, :C], flip=True, narrow=8, algo_cin=5)
                    ops.conv_halo(g, pwN[1], C, N, H, W, 7, y=dh.cols(C, 2 * C), mask=hid.t.hi()[:, C:2 * C], flip=True, narrow=16, algo_cin=10)
                else:
                    ops.conv_halo(g, pwT, 2 * C, N, H, W, 7, y=dh.cols(0, 2 * C), mask=hid.t.hi()[:, :2 * C], flip=True, k1skip=True, algo_cin=7.5)
                ops.conv_halo(g, pwT.rows_from(2 * C), C, N, H, W, 7, y=dh.cols(2 * C, 3 * C), mask=hid.t.hi()[:, 2 * C:], flip=True, algo_cin=40)
                hid.add_grad(dh, masked=True)
            self.tape.append(bwd)
        return outs

    def export_feats(self):
        """c0..c4 as the reference returns them (KGnet.py:318): fp32, NCHW-shaped (channels-last memory)."""
        outs = []
        for fv, (h, w) in zip(self.feats, self.dims):
            o = torch.empty(fv.rows, fv.C, dtype=torch.float32, device=fv.t.device)
            ops.planes_to_f32(fv.t, fv.C, o)
            outs.append(o.view(self.N, h, w, fv.C).permute(0, 3, 1, 2))
        return outs

    def backward_dec(self, map_grads, feat_grads, gscale=None):
        """map_grads: 12 fp32 NCHW (or None); feat_grads: 5 fp32 [rows, C] tensors (or None) or split rows (ops.PT) already in the
        backward pass's own scale.  gscale (half build): device {S, 1 / S} of this backward pass (ops.grad_scale) -- the fp32 gradients
        enter times S, the parameter gradients come back times S (the caller divides them: ops.scale_tensors)."""
        if gscale is not None:
            self.gscale = gscale           # (the scale the seg branch's backward left: its feature gradients are expressed in it)
            Var.ENG = self
        gsc = self.gscale[0:1] if self.gscale is not None else None
        if self.grad_store is not None:
            self.grad_store.dense_backward_started()
        for (slot, lvl, N, Hh, Wh) in self.head_slots:
            gs = map_grads[3 * lvl:3 * lvl + 3]
            if all(g is None for g in gs):
                continue
            dev = self.maps[3 * lvl].device
            packed = ops.alloc_pt(N * Hh * Wh, 64, min(self.ph, self.pg), dev, dtype=self.dt)
            if all(g is not None for g in gs) and self.HEAD_OFF == (0, 8, 24) and sum(self.HEAD_PAD) == 64:
                # (the usual case: one pass writes whole 64-channel rows instead of three column slices)
                ops.grad_pack3([g.contiguous().float() for g in gs], self.maps[3 * lvl], packed, N, [co for _, co in arch.HEADS], Hh, Wh, self.HEAD_PAD, scale=gsc)
                slot["grad"] = packed
                continue
            for k, g in enumerate(gs):
                co = arch.HEADS[k][1]
                view = packed.cols(self.HEAD_OFF[k], self.HEAD_OFF[k] + self.HEAD_PAD[k])
                if g is None:
                    for p in range(view.P):
                        view.plane(p).zero_()
                else:
                    prob = self.maps[3 * lvl] if k == 0 else None
                    ops.grad_pack(g.contiguous().float(), prob, view, N, co, Hh, Wh, self.HEAD_PAD[k], scale=gsc)
            slot["grad"] = packed
        for fv, g in zip(self.feats, feat_grads):
            if g is None:
                continue
            if isinstance(g, PT):              # the fused forward's seg backward already wrote split-bf16 rows
                fv.add_grad(g, masked=False)
                continue
            gp = ops.alloc_pt(fv.rows, fv.C, fv.gP, g.device, dtype=self.dt)
            ops.f32_to_planes(g, gp, fv.C, scale=gsc)
            fv.add_grad(gp, masked=False)
        hook = self.grad_hook
        tape, self.tape = self.tape, None
        ops.WGQ.begin()                  # the split reductions of the weight gradients are recorded and run a dozen per launch
        try:
            marks = list(self.phase_marks) if self.phase_hook is not None else None
            while tape:                              # (popped as they run: a closure and the activations only it still holds die right away)
                if marks is not None:
                    while marks and marks[-1][1] >= len(tape):
                        marks.pop()
                    if marks and marks[-1][0] != getattr(self, "_bwd_phase", None):
                        self._bwd_phase = marks[-1][0]
                        self.phase_hook("bwd", self._bwd_phase)
                fn = tape.pop()
                n0 = len(self.param_grads)
                fn()
                del fn
                if hook is not None and len(self.param_grads) > n0:      # data-parallel: parameter gradients whose kernels are
                    hook(list(self.param_grads.items())[n0:], False)    # enqueued go to the bucketed all-reduce right away
            if hook is not None:
                hook([], True)
            if marks is not None:
                self._bwd_phase = None
                self.phase_hook("bwd", "end")
        except BaseException:
            # a failing launch must not leave this pass's scale state behind: later add_grad / to_current_scale calls of ANY model in
            # the process consult Var.ENG, and the next backward of this engine would start from a stale running scale
            self.gscale, self.param_gsc = None, {}
            if self.grad_store is not None:
                self.grad_store.unscale_of = None
            raise
        finally:
            Var.ENG = None
            ops.WGQ.end()
        grads = self.param_grads
        self.param_grads = {}
        return grads

"""Drop-in replacement of the reference's `KGnet` module (KGnet.py) backed by hand-written HIP kernels.

Same surface as the reference: `resnet50(pretrained)` returns an `nn.Module` whose
`state_dict()` has the reference's 346 keys/shapes (fp32 OIHW master weights), and which supports
  forward(x, bboxes)            KGnet.py:269-272
  forward_dec(x)                KGnet.py:275-318
  forward_seg(feat_seg, bboxes) KGnet.py:321-350
with autograd (`loss.backward()` fills `.grad` of all parameters).  There is no PyTorch/CPU fallback
for the arithmetic: a missing libkgnet_hip.so or a non-GPU tensor raises.

Numerics: the reference computes in fp32.  Here convolutions run on 16-bit MFMA with fp32 accumulation over split 16-bit storage
("planes": engine.py PRECISIONS, csrc/kg_common.h).  `precision=` / env KG_PRECISION / model.set_precision:
  "fp32" (default): every forward tensor = hi + lo IEEE-half planes (22 significant bits), 3 f16 MFMA products per multiply: forward
        results within rtol 1e-4 / atol 1e-5 of the reference; the BACKWARD pass multiplies single half planes (11 bits) under a
        device-side power-of-two gradient scale -- mixed-precision-grade gradients (per-tensor relative L2 error against a float64
        evaluation: see tests/test_gpu_gradprec.py and profiles/r04_grad_table.json);
  "fp32b2": hi + lo half planes in the backward pass as well (gradients on the fp32 reference's own noise floor, ~1.7x the step time);
  "half" / "halfmix": half mixed precision;  "fp32bf" / "fp32bf_full": the same tolerances on bf16 planes (hi + mid + lo, 6 products);
  "mixed" / "trunk2" / "bf16": bf16 mixed-precision policies (opt-in).
RANGE of the half policies (nothing is clamped; a violation surfaces as inf / NaN, never silently): packed weights carry a constant
2^12 scale, so |w| must stay below 16 (check_half_range() raises on a violation; load_state_dict switches such a checkpoint to the bf16-plane sibling policy with a warning); activations are stored as they are,
|x| <= 65504; a non-finite parameter gradient raises the sticky flag grad_overflowed().  Use "fp32bf" for weights outside that range.
Head maps and the feature maps c0..c4 are returned as fp32 tensors like the reference's (KGnet.py:318;
the feature maps are NCHW-shaped with channels-last memory).
"""
import math
import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import arch, _lib
from .engine import Engine
from .seg import SegBranch

__all__ = ["ResNet", "resnet50", "resnet101", "resnet152"]


class _Node(nn.Module):
    """Pure container so that parameter keys match the reference's dotted names."""


def _begin_scaled_backward(eng, top_grads, probs=None):
    """Half-precision build (engine.HALF_POLICIES): the power-of-two scale of this backward pass, computed on the device from the
    gradients of the loss w.r.t. the network outputs (ops.grad_scale).  Returns the device pair {S, 1 / S} or None (bf16 build)."""
    if not eng.fmt:
        eng.gscale = None
        return None
    from . import ops
    gs = ops.grad_scale(top_grads, probs)
    eng.gscale, eng.param_gsc = gs, {}
    eng.flag_on(gs.device)
    if eng.grad_store is not None:
        eng.grad_store.unscale_of = eng.param_gsc      # data parallel: a bucket's gradients are divided by their scales right before its all-reduce
    return gs


def _kp_probs(eng, map_grads):
    """the kp maps are sigmoid outputs (KGnet.py:300): their gradients enter the network times p * (1 - p) (engine.backward_dec)"""
    return [eng.maps[i] if (i % 3 == 0 and g is not None) else None for i, g in enumerate(map_grads)]


def _end_scaled_backward(eng, gs, pgrads):
    """divides the scale out of every parameter gradient the backward pass produced outside the data-parallel flat buffer (one launch):
    each by the running scale it was produced in (engine.renormalise moves it at the re-normalisation points of the pass)"""
    if gs is None:
        return
    from . import ops
    store = eng.grad_store
    items = [(k, g) for k, g in pgrads.items() if g is not None and not (store is not None and store.owns(k, g))]
    ops.scale_tensors([g for _, g in items], [eng.param_gsc[k][1:2] for k, _ in items], flag=eng.flag_on(gs.device))
    if store is not None:
        store.unscale_pending()
        store.unscale_of = None
    eng.gscale = None


class _DecFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, record, x, *params):
        eng = model._engine   # (grad mode is off inside Function.forward: `record` is decided by the caller)
        maps, feats, dims = eng.forward_dec(x, record)
        outs = list(maps) + eng.export_feats()
        ctx.model, ctx.recorded, ctx.generation = model, record, eng.generation
        ctx.keys = model._param_keys
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        if not ctx.recorded:
            raise RuntimeError("KGnet forward was run without gradient recording")
        eng = ctx.model._engine
        if ctx.generation != eng.generation or eng.tape is None:
            raise RuntimeError("KGnet backward: the engine keeps the activations of the LATEST forward_dec only; another "
                               "forward_dec ran on this model between this loss's forward and its backward")
        fg = []
        for g in grads[12:]:
            if g is None:
                fg.append(None)
            else:
                n, c, h, w = g.shape
                fg.append(g.float().permute(0, 2, 3, 1).contiguous().view(n * h * w, c))
        store = eng.grad_store
        with torch.cuda.device(next(g for g in grads if g is not None).device):
            mg = [None if g is None else g.contiguous().float() for g in grads[:12]]
            gs = _begin_scaled_backward(eng, mg + fg, _kp_probs(eng, mg) + [None] * len(fg))
            pg = eng.backward_dec(mg, fg, gscale=eng.gscale)
            _end_scaled_backward(eng, gs, pg)
        out = [None, None, None]
        for k in ctx.keys:
            g = pg.get(k)
            if store is not None and g is not None and store.owns(k, g):
                g = store.deliver(k, ctx.model.get_tensor(k))      # the flat slot becomes .grad itself: autograd gets nothing to accumulate
            out.append(g)
        return tuple(out)


class _NetFunction(torch.autograd.Function):
    """forward(x, bboxes) (KGnet.py:269-272) as ONE autograd node: forward_dec, then the seg branch straight on the engine's own
    split-bf16 feature rows (no fp32 export of c0..c4, no autograd hop between the two halves); backward runs the seg branch's
    backward, hands its fp32 feature gradients to the dense backward and returns every parameter gradient at once."""

    @staticmethod
    def forward(ctx, model, record, holder, x, *params):
        eng, seg = model._engine, model._seg
        maps, feats, dims = eng.forward_dec(x, record)
        # (the box tables are built on the host AFTER the dense forward is enqueued: the GPU is busy meanwhile)
        plan = seg.make_plan(None, holder["bboxes"], sizes=dims, dev=x.device, need_bins=record)
        plan.out_planes = [fv.gP for fv in feats]     # the seg backward writes the engine's gradient rows directly
        holder["plan"] = plan
        flat, saved = seg.run_forward(plan, [fv.t for fv in feats], record)
        ctx.model, ctx.recorded, ctx.generation, ctx.plan, ctx.saved = model, record, eng.generation, plan, saved
        ctx.feat_shapes = [(x.shape[0], fv.C, h, w) for fv, (h, w) in zip(feats, dims)]
        return tuple(maps) + (flat,)

    @staticmethod
    def backward(ctx, *grads):
        if not ctx.recorded:
            raise RuntimeError("KGnet forward was run without gradient recording")
        model = ctx.model
        eng, seg = model._engine, model._seg
        if ctx.generation != eng.generation or eng.tape is None:
            raise RuntimeError("KGnet backward: the engine keeps the activations of the LATEST forward only; another "
                               "forward ran on this model between this loss's forward and its backward")
        dev = next(g for g in grads if g is not None).device
        with torch.cuda.device(dev):
            gflat = grads[12]
            if gflat is not None:
                gflat = gflat.contiguous().float()
            mg = [None if g is None else g.contiguous().float() for g in grads[:12]]
            gs = _begin_scaled_backward(eng, mg + [gflat], _kp_probs(eng, mg) + [ctx.saved[4] if (ctx.saved is not None and gflat is not None) else None])
            fg, spg = [None] * 5, {}
            if gflat is not None and ctx.saved is not None:
                fg, spg = seg.run_backward(ctx.plan, ctx.saved, gflat, ctx.feat_shapes, gscale=gs)
            pg = eng.backward_dec(mg, fg, gscale=eng.gscale)      # (the running scale: the seg branch's backward may have moved it)
            pg.update(spg)
            _end_scaled_backward(eng, gs, pg)
        out = [None, None, None, None]
        store = eng.grad_store
        for k in model._all_param_keys:
            g = pg.get(k)
            if store is not None and g is not None and store.owns(k, g):
                g = store.deliver(k, model.get_tensor(k))
            out.append(g)
        return tuple(out)


class ResNet(nn.Module):
    """KGnet (ResNet-50[:layer3] + top-down decoder + 12 heads + per-box seg branch)."""

    def __init__(self, block=None, layers=(3, 4, 6, 3), num_classes=1000, zero_init_residual=False, precision=None):
        super().__init__()
        if block is not None and getattr(block, "expansion", 4) != 4:
            raise NotImplementedError("BasicBlock trunks (resnet18/34) cannot run forward_dec in the reference either "
                                      "(channel plan of KGnet.py:116-119,150-158 assumes expansion 4)")
        if len(layers) < 3 or any(int(b) < 1 for b in layers[:3]):
            raise ValueError("layers must give the block counts of layer1..layer3")
        self.layers_tab = arch.layers_table(layers)          # KGnet.py:135-137 builds layers[0..2] only
        self._slot_cache = {}            # get_tensor: key -> (the node's _parameters dict, its _buffers dict, leaf name)
        self._param_keys = []
        for key, shape, kind in arch.state_spec(layers):
            parts = key.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            if kind == "conv_w":
                t = torch.empty(shape)
                nn.init.kaiming_normal_(t, mode="fan_out", nonlinearity="relu")  # KGnet.py:212-214
                node.register_parameter(parts[-1], nn.Parameter(t))
            elif kind == "conv_b":
                fan_in = None
                w = node._parameters["weight"]
                fan_in = w.shape[1] * w.shape[2] * w.shape[3]
                bound = 1.0 / math.sqrt(fan_in)
                node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape).uniform_(-bound, bound)))
            elif kind == "bn_w":
                node.register_parameter(parts[-1], nn.Parameter(torch.ones(shape)))
            elif kind == "bn_b":
                node.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape)))
            elif kind == "bn_rm":
                node.register_buffer(parts[-1], torch.zeros(shape))
            elif kind == "bn_rv":
                node.register_buffer(parts[-1], torch.ones(shape))
            else:
                node.register_buffer(parts[-1], torch.tensor(0, dtype=torch.long))
            if kind in ("conv_w", "conv_b", "bn_w", "bn_b"):
                self._param_keys.append(key)
        if zero_init_residual:
            for name, _, _, blocks, _ in self.layers_tab:
                for b in range(blocks):
                    nn.init.constant_(self.get_tensor(f"{name}.{b}.bn3.weight"), 0)
        self._engine = Engine(self, precision)
        self._seg = SegBranch(self)
        self._all_param_keys = list(self._param_keys)      # (the seg branch's parameters are part of _param_keys: one list for the fused forward)

    # ---- precision / cache control (extensions; the reference has neither) --------------------------
    @property
    def precision(self):
        return self._engine.precision

    def set_precision(self, precision):
        """one of engine.PRECISIONS ("fp32" = default) -- see the module docstring."""
        self._engine.set_precision(precision)
        self._seg.invalidate_caches()
        return self

    def grad_overflowed(self, reset=True):
        """Half-precision policies: True if a backward pass since the last call produced a non-finite parameter gradient -- the
        half-precision backward left IEEE half's range (or the forward did: an activation beyond +-65504, a weight >= 16).  Nothing is
        clamped silently: such a step's gradients are inf / NaN and the flag is sticky.  Reads one device int (synchronises)."""
        f = self._engine.overflow_flag
        if f is None:
            return False
        v = bool(int(f.item()))
        if reset and v:
            f.zero_()
        return v

    HALF_WEIGHT_LIMIT = 16.0       # csrc/kg_common.h KG_WSCALE = 2^12: 16 * 4096 = 65536 > IEEE half's 65504

    BF16_SIBLING = {"fp32": "fp32bf", "fp32b2": "fp32bf_full", "half": "bf16", "halfmix": "mixed"}      # same planes plan on bf16 rows (8-bit exponent)

    def half_range_violations(self):
        """Half policies: [(key, max |w|)] of the conv weights that cannot be packed (|w| * 2^12 beyond IEEE half: |w| >= 16, or non-finite).
        ONE multi-tensor max reduction over all conv weights and one read-back."""
        if not self._engine.fmt:
            return []
        keys = [k for k in self._param_keys if self.get_tensor(k).dim() == 4 and self.get_tensor(k).numel()]
        if not keys:
            return []
        with torch.no_grad():
            ms = torch.stack(torch._foreach_norm([self.get_tensor(k).detach() for k in keys], float("inf"))).float().cpu().numpy()
        return [(k, float(m)) for k, m in zip(keys, ms) if not m < self.HALF_WEIGHT_LIMIT]

    def check_half_range(self):
        """Half policies: raises ValueError if a conv weight cannot be packed (|w| >= 16); callable by hand after writing parameters."""
        bad = self.half_range_violations()
        if bad:
            raise ValueError(f"precision={self.precision!r} stores packed weights x 2^12 in IEEE half: |w| must be < {self.HALF_WEIGHT_LIMIT:g}, but "
                             f"{bad[0][0]} has max |w| = {bad[0][1]:g} ({len(bad)} tensor(s)); use precision='fp32bf' (bf16 planes) for such weights")

    def load_state_dict(self, state_dict, strict=True, **kw):
        """As nn.Module.load_state_dict (KGnet.py:384-385, test.py:60-61 load any checkpoint).  A checkpoint whose conv weights leave the
        half policies' range (|w| >= 16) switches the model to the policy's bf16-plane sibling (same tolerance class, 8-bit exponent; "fp32" ->
        "fp32bf") with a warning instead of failing (recorded in `model.precision_switch` = (from, to, key, max |w|)); KG_HALF_RANGE=raise
        restores the hard error.  Non-finite weights always raise; so does a needed switch once a FlatGradReducer is attached."""
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.precision_switch = None
        bad = self.half_range_violations()
        if bad:
            nonfinite = [(k, m) for k, m in bad if not m < float("inf")]
            if nonfinite:      # a NaN / inf weight is a broken checkpoint in ANY policy: never a reason to change the arithmetic
                raise ValueError(f"KGnet.load_state_dict: {nonfinite[0][0]} holds non-finite values ({len(nonfinite)} tensor(s))")
            if os.environ.get("KG_HALF_RANGE", "fallback") == "raise":
                self.check_half_range()
            if self._engine.grad_store is not None:
                # FlatGradReducer.attach sized its flat views / scale bookkeeping for the current policy, and under data parallelism every
                # rank would take this decision on its own: refuse instead of switching underneath it
                raise ValueError(f"KGnet.load_state_dict: {bad[0][0]} (max |w| = {bad[0][1]:g}) is outside the range of precision="
                                 f"{self.precision!r} and a gradient reducer is attached; load the checkpoint (or call set_precision("
                                 f"{self.BF16_SIBLING[self.precision]!r})) BEFORE FlatGradReducer.attach")
            sib = self.BF16_SIBLING[self.precision]
            warnings.warn(f"KGnet: {len(bad)} conv weight tensor(s) (e.g. {bad[0][0]}, max |w| = {bad[0][1]:g}) are outside the range of precision="
                          f"{self.precision!r} (packed weights x 2^12 in IEEE half: |w| < {self.HALF_WEIGHT_LIMIT:g}); switching this model to "
                          f"precision={sib!r} (bf16 planes)", RuntimeWarning, stacklevel=2)
            self.precision_switch = (self.precision, sib, bad[0][0], bad[0][1])      # visible to the caller: model.precision_switch, r.kg_precision_switch
            self.set_precision(sib)
        try:
            r.kg_precision_switch = self.precision_switch
        except AttributeError:         # (a namedtuple result without __dict__: the model attribute carries it)
            pass
        return r

    def invalidate_caches(self):
        """Drop the packed bf16 weight copies and folded BatchNorm constants: call after writing parameters or running
        statistics in a way PyTorch's version counters do not see (`p.data.copy_`, third-party fused optimizers) when no
        training forward follows before the next inference."""
        self._engine.invalidate_caches()
        self._seg.invalidate_caches()

    # ---- helpers ----------------------------------------------------------------------------------
    def get_tensor(self, key):
        """Parameter / buffer by its dotted reference key.  The containers of a key (the `_parameters` / `_buffers` dicts of its node) are resolved
        once -- the module tree of this network never changes -- and read on every call, so a replaced Parameter is seen; ~700 look-ups per
        train step go through here, and the 217 of the next step's first lines sit between the loss read-back and the step's first kernel."""
        slot = self._slot_cache.get(key)
        if slot is None:
            parts = key.split(".")
            node = self
            for p in parts[:-1]:
                node = node._modules[p]
            slot = self._slot_cache[key] = (node._parameters, node._buffers, parts[-1])
        t = slot[0].get(slot[2])
        return t if t is not None else slot[1][slot[2]]

    def _check_input(self, x):
        if not x.is_cuda:
            raise _lib.KGLibraryError("KGnet (MI355X build) runs on the GPU only: input tensor is on %s" % x.device)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected an [N,3,H,W] image batch")
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError("input height/width must be multiples of 8 (4 pyramid levels, SURVEY 5)")
        _lib.load()

    # ---- reference API ----------------------------------------------------------------------------
    def forward_dec(self, x):
        self._check_input(x)
        params = [self.get_tensor(k) for k in self._param_keys]
        record = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        with torch.cuda.device(x.device):      # kernels launch on the stream of the tensors' device, not of the "current" one
            outs = _DecFunction.apply(self, record, x, *params)
        d = [list(outs[3 * i:3 * i + 3]) for i in range(4)]
        return d[0], d[1], d[2], d[3], list(outs[12:17])

    def forward_seg(self, feat_seg, bboxes):
        with torch.cuda.device(feat_seg[0].device):
            return self._seg.forward(feat_seg, bboxes)

    def forward(self, x, bboxes):
        """KGnet.py:269-272: (dec0, dec1, dec2, dec3, [mask_patches, mask_dets])."""
        self._check_input(x)
        with torch.cuda.device(x.device):
            params = [self.get_tensor(k) for k in self._all_param_keys]
            record = torch.is_grad_enabled() and any(p.requires_grad for p in params)
            holder = {"bboxes": bboxes}
            outs = _NetFunction.apply(self, record, holder, x, *params)
        d = [list(outs[3 * i:3 * i + 3]) for i in range(4)]
        plan = holder["plan"]
        if plan.nb[0] == 0:          # no valid box anywhere: the seg branch contributes nothing (KGnet.py:339-340)
            return d[0], d[1], d[2], d[3], [[[] for _ in bboxes], [[] for _ in bboxes]]
        return d[0], d[1], d[2], d[3], self._seg.predictions(plan, outs[12], len(bboxes))


def _load_pretrained(model):
    """The reference downloads torchvision's ImageNet ResNet-50 (KGnet.py:384-385, strict=False).
    Offline, a local checkpoint can be supplied through KG_RESNET50_PTH; otherwise warn and keep the init."""
    path = os.environ.get("KG_RESNET50_PTH")
    if path and os.path.exists(path):
        model.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
    else:
        warnings.warn("pretrained=True: no network access and KG_RESNET50_PTH is not set; keeping the random init")


def resnet50(pretrained=False, **kwargs):
    model = ResNet(None, [3, 4, 6, 3], **kwargs)
    if pretrained:
        _load_pretrained(model)
    return model


def resnet101(pretrained=False, **kwargs):
    """KGnet.py:388-397 (Bottleneck [3,4,23,3]; pretrained weights are only wired for resnet50, KG_RESNET50_PTH)."""
    return ResNet(None, [3, 4, 23, 3], **kwargs)


def resnet152(pretrained=False, **kwargs):
    """KGnet.py:400-409 (Bottleneck [3,8,36,3])."""
    return ResNet(None, [3, 8, 36, 3], **kwargs)

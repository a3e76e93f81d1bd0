"""Per-box segmentation branch (reference KGnet.py:246-267, 321-350) as batched ragged HIP work.

The reference crops c0..c4 per box and runs ~15 tiny PyTorch kernels per box in a Python loop.
Here every pyramid level holds ONE ragged pixel list for all boxes of the batch (box-major, raster
inside a box); the top-down combine and the seg head are a handful of ragged implicit-GEMM launches.
Host glue (this file) only computes the integer crop rectangles with the reference's float32
rounding rules and uploads the box tables.
"""
import ctypes
import os

import numpy as np
import torch

from . import arch, ops, _lib
from ._lib import ptr, stream_ptr, c_long
from .ops import BF16, PT, PackedWeight

BIN_SIZE = (16, 16, 8, 4, 4)      # bins of the crop-gradient reduction per pyramid level (box sides halve with every level)


def crop_rects(boxes, h0, w0, sizes):
    """Vectorised restatement of KGnet.py:332-335 + 248-254 in float32 (numpy 2.x semantics of the
    reference: np.float32 / float -> float32; np.round = rint half-to-even).
    boxes [n,4] float32 (y1,x1,y2,x2) in c0 pixels.  Returns rect [L][n,4] int32 (y1,x1,y2,x2
    end-exclusive slices) and depth [n] = number of leading accepted levels."""
    b = np.asarray(boxes, np.float32).reshape(-1, 4)
    n = len(b)
    ny1 = b[:, 0] / np.float32(h0); nx1 = b[:, 1] / np.float32(w0)
    ny2 = b[:, 2] / np.float32(h0); nx2 = b[:, 3] / np.float32(w0)
    rects, depth, alive = [], np.zeros(n, np.int32), np.ones(n, bool)
    for (h, w) in sizes:
        y1 = np.maximum(0, np.round(ny1 * np.float32(h)).astype(np.int32))
        x1 = np.maximum(0, np.round(nx1 * np.float32(w)).astype(np.int32))
        y2 = np.minimum(np.round(ny2 * np.float32(h)).astype(np.int32), h - 1)
        x2 = np.minimum(np.round(nx2 * np.float32(w)).astype(np.int32), w - 1)
        ok = ~((y2 < y1) | (x2 < x1) | (y2 - y1 < 2) | (x2 - x1 < 2))
        alive &= ok
        depth += alive
        rects.append(np.stack([y1, x1, y2, x2], 1))
    return rects, depth


class _Plan:
    """Box tables of one forward_seg call (host numpy + device copies)."""


class LazyList(list):
    """A list whose items are produced on first access.  forward_seg returns one per image for the mask patches /
    detections: building 2400 tensor views per step costs ~5 ms of host time that the HIP SEG_loss (which reads the
    flat buffer directly) never needs; the reference's own code paths (len(), indexing, iteration, zip) still work."""

    def __init__(self, n, factory):
        super().__init__()
        self._n, self._factory = n, factory

    def _fill(self):
        if self._factory is not None:
            f, self._factory = self._factory, None
            super().extend(f())

    def __len__(self):
        return self._n if self._factory is not None else super().__len__()

    def __iter__(self):
        self._fill(); return super().__iter__()

    def __getitem__(self, i):
        self._fill(); return super().__getitem__(i)

    def __repr__(self):
        self._fill(); return super().__repr__()

    def __eq__(self, other):
        self._fill(); return super().__eq__(other)

    def append(self, v):
        self._fill(); super().append(v); self._n = super().__len__()


SEG_C1 = True       # seg_head.2 on the dot-product kernel (kg_seg_conv3_c1); False: the generic ragged conv (kept for probes)


class SegPredictions(list):
    """[mask_patches, mask_dets] exactly as the reference returns them (KGnet.py:350), plus the flat
    probability buffer / per-patch metadata so that SEG_loss can skip per-patch host round trips."""
    kg_meta = None


class _SegFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, branch, plan, record, f0, f1, f2, f3, f4, *params):
        flat, saved = branch.run_forward(plan, [f0, f1, f2, f3, f4], record)
        ctx.branch, ctx.plan, ctx.saved = branch, plan, saved
        ctx.feat_shapes = [tuple(f.shape) for f in (f0, f1, f2, f3, f4)]
        return flat

    @staticmethod
    def backward(ctx, gflat):
        if ctx.saved is None:
            raise RuntimeError("forward_seg was run without gradient recording")
        eng = ctx.branch.m._engine
        gflat = gflat.contiguous().float()
        gs = ops.grad_scale([gflat], [ctx.saved[4]]) if eng.fmt else None       # half build: this node's own power-of-two gradient scale
        eng.gscale = gs
        if gs is not None:
            eng.param_gsc = {}
        gfeats, pgrads = ctx.branch.run_backward(ctx.plan, ctx.saved, gflat, ctx.feat_shapes, gscale=gs)
        eng.gscale = None
        out = [None, None, None] + gfeats
        store = eng.grad_store
        if gs is not None:
            items = [(k, g) for k, g in pgrads.items() if g is not None and not (store is not None and store.owns(k, g))]
            ops.scale_tensors([g for _, g in items], [eng.param_gsc[k][1:2] for k, _ in items], flag=eng.flag_on(gflat.device))
        for k in ctx.branch.param_keys:
            g = pgrads.get(k)
            if store is not None and g is not None and store.owns(k, g):
                g = store.deliver(k, ctx.branch.P(k))
            out.append(g)
        if store is not None:
            store.seg_done(unscale_of=eng.param_gsc if gs is not None else None)
        return tuple(out)


class SegBranch:
    def __init__(self, module):
        self.m = module
        self.param_keys = []
        for i in range(4):
            for sub in ("up", "cat_conv"):
                self.param_keys += [f"skip_combine.{i}.{sub}.0.weight", f"skip_combine.{i}.{sub}.0.bias"]
        self.param_keys += ["seg_head.0.weight", "seg_head.0.bias", "seg_head.2.weight", "seg_head.2.bias"]
        self.packed = {}
        self.train_steps, self.stamp = 0, ("e", 0, 0)     # see Engine.prepare: training forwards always repack
        self.prepack_stamp = None                         # Engine.prepack / forward_dec: the stamp the weights of this step were packed ahead under
        self.fast_stamp = None                            # ... and that stamp again when the model's fingerprint has not moved since (Engine.fingerprint)

    @property
    def P_(self):
        """planes of the seg branch's activations / weights (engine.PRECISIONS)"""
        return self.m._engine.pseg

    @property
    def dt(self):
        """16-bit format of the branch's rows / packed weights (engine.HALF_POLICIES)"""
        return self.m._engine.dt

    @property
    def Pg(self):
        """planes of the branch's gradients and of the operands of its backward convolutions (engine.PRECISIONS, last entry)"""
        return min(self.m._engine.pseg, self.m._engine.pg)

    def invalidate_caches(self):
        self.packed = {}

    def P(self, k):
        return self.m.get_tensor(k)

    def prepare_all(self, train):
        """Queues the (re)packs of all weights seen in earlier steps for one batched launch (ops.PackQueue)."""
        ops.PACKQ.defer = True
        try:
            for key, e in list(self.packed.items()):
                self.packw(key, need_T=train and e.get("need_T", False))
        finally:
            ops.PACKQ.defer = False

    def packw(self, key, need_T):
        """key -> (PackedWeight fwd, PackedWeight dgrad, bias) repacked when the parameter version changes."""
        e = self.packed.get(key)
        if self.fast_stamp is not None and e is not None and e["ver"][:3] == self.fast_stamp and (not need_T or e["T_ok"]):
            e["need_T"] = bool(need_T) or e.get("need_T", False)      # prepacked under this stamp, fingerprint unchanged (Engine.prepare)
            return e["pw"], e["pwT"], self.P(key + ".bias").detach()
        w = self.P(key + ".weight")
        ver = self.stamp + (w._version, w.data_ptr())
        cout, cin, k, _ = w.shape
        if e is None or e["ver"] != ver or e["pw"].buf.device != w.device:
            pw = e["pw"] if e is not None and e["pw"].buf.device == w.device else PackedWeight(cout, k * k, ops.round_up(cin, 8), w.device, xP=self.P_, wP=self.P_, dtype=self.dt)
            pw.pack(w.detach())
            e = {"ver": ver, "pw": pw, "pwT": e["pwT"] if e is not None and e["pw"].buf.device == w.device else None, "T_ok": False}
            self.packed[key] = e
        if need_T and not e["T_ok"]:
            if e["pwT"] is None:
                e["pwT"] = PackedWeight(cin, k * k, ops.round_up(cout, 8), w.device, xP=self.Pg, wP=min(self.Pg, self.m._engine.pdw), dtype=self.dt)
            e["pwT"].pack(w.detach(), transposed=True)
            e["T_ok"] = True
        e["need_T"] = bool(need_T) or e.get("need_T", False)
        return e["pw"], e["pwT"], self.P(key + ".bias").detach()

    # ---- planning (host) ----------------------------------------------------------------------------
    def make_plan(self, feats, bboxes, sizes=None, dev=None, need_bins=True):
        """feats: the five feature maps (or None with sizes = [(h, w)] * 5 and dev given); need_bins: the bin tables of the
        crop-gradient reduction (backward only)"""
        if sizes is None:
            dev = feats[0].device
            sizes = [tuple(f.shape[2:]) for f in feats]
        h0, w0 = sizes[0]
        img_idx, box_idx, allb = [], [], []
        for i, bb in enumerate(bboxes):
            if bb is None or len(bb) == 0:
                continue
            bb = np.asarray(bb.detach().cpu().numpy() if hasattr(bb, "detach") else bb, np.float32).reshape(-1, 5)
            allb.append(bb)
            img_idx += [i] * len(bb)
            box_idx += list(range(len(bb)))
        p = _Plan()
        p.nimg = len(bboxes)
        p.sizes = sizes
        if not allb:
            p.nb = [0] * 5
            return p
        allb = np.concatenate(allb, 0)
        img_idx = np.asarray(img_idx, np.int32); box_idx = np.asarray(box_idx, np.int32)
        rects, depth = crop_rects(allb[:, :4], h0, w0, sizes)
        keep = depth > 0
        order = np.argsort(-depth[keep], kind="stable")
        sel = np.nonzero(keep)[0][order]
        p.boxes = allb[sel]; p.img = img_idx[sel]; p.box_in_img = box_idx[sel]; p.depth = depth[sel]
        p.nb = [int((p.depth > l).sum()) for l in range(5)]
        p.row0, p.hw, tabs = [], [], []
        for l in range(5):
            nb = p.nb[l]
            r = rects[l][sel][:nb]
            h = (r[:, 2] - r[:, 0]).astype(np.int32); w = (r[:, 3] - r[:, 1]).astype(np.int32)
            row0 = np.zeros(nb + 1, np.int64)
            np.cumsum(h.astype(np.int64) * w, out=row0[1:])
            p.row0.append(row0); p.hw.append((h, w))
            H, W = sizes[l]
            tab = np.stack([p.img[:nb], r[:, 0], r[:, 1], h, w, row0[:-1].astype(np.int32),
                            np.full(nb, H, np.int32), np.full(nb, W, np.int32)], 1).astype(np.int32)
            tabs.append(tab)
        bil = []
        for l in range(4):   # level l+1 -> level l for the nb[l+1] combine boxes
            nc = p.nb[l + 1]
            (hi, wi), (ho, wo) = p.hw[l + 1], p.hw[l]
            bil.append(np.stack([p.row0[l + 1][:nc].astype(np.int32), hi[:nc], wi[:nc], p.row0[l][:nc].astype(np.int32),
                                 ho[:nc], wo[:nc]], 1).astype(np.int32))
        # tile tables for the LDS-halo kernels: one {row0, (h<<16)|w, (oy0<<16)|ox0, 0} entry per tile of every box (native host code:
        # kg_host_tile_table; as NumPy repeat / cumsum chains the ten tables of a step cost 1 ms)
        lib = _lib.load()

        def vp(a):
            return ctypes.c_void_p(a.ctypes.data)

        def tile_table(l, th, tw):
            h, w = p.hw[l]
            nb = len(h)
            cnt = int((((h + th - 1) // th) * ((w + tw - 1) // tw)).sum()) if nb else 0
            out = np.empty((cnt, 4), np.int32)
            if cnt:
                r0 = np.ascontiguousarray(p.row0[l][:-1], np.int64)
                hc, wc = np.ascontiguousarray(h, np.int32), np.ascontiguousarray(w, np.int32)      # (named: the arrays must outlive the call)
                got = lib.kg_host_tile_table(vp(hc), vp(wc), vp(r0), nb, th, tw, vp(out), cnt)
                if got != cnt:
                    raise _lib.KGLibraryError(f"kg_host_tile_table: {got} entries, expected {cnt}")
            return out
        t32 = [tile_table(l, 16, 32) for l in range(5)]
        t16 = [tile_table(l, 16, 16) for l in range(5)]
        t8 = [tile_table(0, 8, 16)]              # c0 crops: 8 x 16 tiles of the weight-stationary 64 -> 64 kernel (conv3_ws.hip)

        def cum(l, th, tw):
            h, w = p.hw[l]
            c = np.zeros(len(h) + 1, np.int64)
            np.cumsum(((h + th - 1) // th) * ((w + tw - 1) // tw), out=c[1:])
            return c
        p.t32_cum = [cum(l, 16, 32) for l in range(5)]; p.t16_cum = [cum(l, 16, 16) for l in range(5)]
        p.t8_cum = [cum(0, 8, 16)]
        # bin grid of the deterministic crop-gradient reduction (kg_crop_grad_reduce): per level, CSR lists of the boxes that
        # touch each BIN x BIN bin of each image, in ascending box order
        bin_start, bin_boxes = [], []
        for l in range(5):
            nb = p.nb[l]
            H, W = sizes[l]
            BS = BIN_SIZE[l]
            BY, BX = (H + BS - 1) // BS, (W + BS - 1) // BS
            nbins = p.nimg * BY * BX
            if nb == 0 or not need_bins:
                bin_start.append(np.zeros(nbins + 1 if need_bins else 1, np.int32)); bin_boxes.append(np.zeros(0, np.int32))
                continue
            t = np.ascontiguousarray(tabs[l], np.int32)
            cap = int((((t[:, 1] + t[:, 3] - 1) // BS - t[:, 1] // BS + 1) * ((t[:, 2] + t[:, 4] - 1) // BS - t[:, 2] // BS + 1)).sum())
            st = np.empty(nbins + 1, np.int32)
            bb = np.empty(cap, np.int32)
            got = lib.kg_host_bin_csr(vp(t), nb, BS, BY, BX, nbins, vp(st), vp(bb), cap)     # counting sort: boxes ascending inside a bin
            if got != cap:
                raise _lib.KGLibraryError(f"kg_host_bin_csr: {got} incidences, expected {cap}")
            bin_start.append(st); bin_boxes.append(bb)
        blob = np.concatenate([t.ravel() for t in tabs] + [b.ravel() for b in bil] + [t.ravel() for t in t32] + [t.ravel() for t in t16]
                              + [t.ravel() for t in t8] + bin_start + bin_boxes).astype(np.int32)
        dblob = ops.h2d(blob, dev)
        off = 0
        p.tab_d, p.bil_d = [], []
        for t in tabs:
            p.tab_d.append(dblob[off:off + t.size]); off += t.size
        for b in bil:
            p.bil_d.append(dblob[off:off + b.size]); off += b.size
        p.t32_d, p.t16_d = [], []
        for t in t32:
            p.t32_d.append(dblob[off:off + t.size].view(-1, 4)); off += t.size
        for t in t16:
            p.t16_d.append(dblob[off:off + t.size].view(-1, 4)); off += t.size
        p.t8_d = []
        for t in t8:
            p.t8_d.append(dblob[off:off + t.size].view(-1, 4)); off += t.size
        p.bin_start_d, p.bin_boxes_d = [], []
        for t in bin_start:
            p.bin_start_d.append(dblob[off:off + t.size]); off += t.size
        for t in bin_boxes:
            p.bin_boxes_d.append(dblob[off:off + max(t.size, 0)]); off += t.size
        p.rows = [int(p.row0[l][-1]) for l in range(5)]
        p.rowdesc, p.row2box, p.srcrow = [], [], []
        for l in range(5):
            R = max(p.rows[l], 1)
            rd = torch.empty(R, 2, dtype=torch.int32, device=dev)
            r2b = torch.empty(R, dtype=torch.int32, device=dev)
            sr = torch.empty(R, dtype=torch.int32, device=dev)
            p.rowdesc.append(rd); p.row2box.append(r2b); p.srcrow.append(sr)
        if any(p.nb):          # the row tables of all five levels in one launch (kg_seg_build_rows_levels: host arrays of device pointers)
            VP5, I5 = ctypes.c_void_p * 5, ctypes.c_int * 5
            as_ptrs = lambda ts: VP5(*[t.data_ptr() for t in ts])
            _lib.call("kg_seg_build_rows_levels", 5, as_ptrs(p.tab_d), I5(*[int(n) for n in p.nb]), as_ptrs(p.rowdesc), as_ptrs(p.row2box),
                      as_ptrs(p.srcrow), stream_ptr())
        return p

    # ---- device work --------------------------------------------------------------------------------
    @staticmethod
    def feat_rows(f):
        """fp32 [N*H*W, C] rows of a feature map given as the reference gives it (fp32 NCHW, KGnet.py:318); forward_dec's own
        outputs are channels-last in memory, so this is a view for them."""
        n, c, h, w = f.shape
        r = f.detach().permute(0, 2, 3, 1)
        if r.dtype != torch.float32:
            r = r.float()
        return r.reshape(n * h * w, c) if r.is_contiguous() else r.contiguous().view(n * h * w, c)

    def gather(self, frows, srcrow, dst, nrows, C, row_off=0):
        """dst (PT rows) = the crop rows of the feature rows frows: fp32 rows (a feature map handed to forward_seg), or the
        engine's own split-bf16 rows (ops.PT; the fused training forward, plane by plane -- a plane the branch does not
        carry is dropped, one it carries beyond the source's is zero)"""
        if not nrows:
            return
        sr = _lib.c_void_p(srcrow.data_ptr() + 4 * row_off)
        if isinstance(frows, PT):
            if 1 < dst.P <= frows.P:      # every plane the branch carries exists in the source: ONE launch for all of them
                _lib.call("kg_rows_gather_planes", ptr(frows.t), ops.ld(frows), frows.ps, sr, ptr(ops.base(dst)), ops.ld(dst), dst.ps, c_long(nrows), C, dst.P,
                          stream_ptr(), fmt=ops.fmt_of(dst))
                return
            for p_ in range(dst.P):
                if p_ < frows.P:
                    _lib.call("kg_rows_gather", ops.ctypes_offset(frows.t, p_ * frows.ps), ops.ld(frows), sr,
                              ops.ctypes_offset(ops.base(dst), p_ * dst.ps), ops.ld(dst), c_long(nrows), C, stream_ptr(), fmt=ops.fmt_of(dst))
                else:
                    dst.plane(p_)[:nrows].zero_()
            return
        _lib.call("kg_rows_gather_f32", ptr(frows), frows.stride(0), sr, ptr(ops.base(dst)), ops.ld(dst), c_long(nrows), C,
                  ops.pl(y=dst), stream_ptr(), fmt=ops.fmt_of(dst))

    @staticmethod
    def _head(t, M):
        """first M rows of a rows tensor (torch tensor or ops.PT)"""
        if t is None:
            return None
        return t.rows(0, M) if isinstance(t, PT) else t[:M]

    def alloc(self, rows, C, dev):
        return ops.alloc_pt(rows, C, self.P_, dev, dtype=self.dt)

    def galloc(self, rows, C, dev):
        return ops.alloc_pt(rows, C, self.Pg, dev, dtype=self.dt)

    def rconv(self, x, pw, cout, rowdesc, M, k, y=None, y_f32=None, bias=None, relu=False, mask=None, mode=2, tiles=None, tiles16=None, tiles8=None):
        """Ragged conv (mode 2) or its input gradient (mode 3).  3x3 convs over 64-channel-aligned inputs run on the
        LDS-halo kernel with one (box, 16x32 tile) entry per workgroup; the rest on the gather implicit GEMM.
        `tiles` = (tile table of the first `M` rows' boxes); boxes are a prefix, so a prefix of the table is used."""
        if (ops.USE_HALO and k == 3 and tiles is not None and tiles.shape[0] > 0 and pw.cin_pad % 64 == 0 and x.shape[1] >= pw.cin_pad
                and M >= 0.35 * tiles.shape[0] * 512):     # tiles mostly full: tiny deep-level crops stay on the gather kernel
            ops.conv_halo(x, pw, cout, 0, 0, 0, 3, y=y, y_f32=y_f32, bias=bias, relu=relu, mask=mask, flip=(mode == 3),
                          tiletab=tiles, total_rows=M, tiletab16=tiles16, tiletab8=tiles8)
            return
        if k == 1 and y is not None and ops.can_1x1(self._head(x, M), pw, 1, 1, 0, self._head(y, M), y_f32):
            ops.conv1x1(ops.base(x)[:M], pw, cout, ops.base(y)[:M], bias=bias, mask=mask[:M] if mask is not None else None, relu=relu)
            return
        geom = (M, 0, 0, M, 1, k, k, 1, (k - 1) // 2)
        ops.conv_igemm(x, pw, cout, geom, y=y, y_f32=y_f32, bias=bias, relu=relu, mask=mask, mode=mode, rowdesc=rowdesc)

    @staticmethod
    def T8(plan, l, nboxes):
        return plan.t8_d[l][:int(plan.t8_cum[l][nboxes])] if l < len(plan.t8_d) else None

    @staticmethod
    def T32(plan, l, nboxes):
        return plan.t32_d[l][:int(plan.t32_cum[l][nboxes])]

    @staticmethod
    def T16(plan, l, nboxes):
        return plan.t16_d[l][:int(plan.t16_cum[l][nboxes])]

    def run_forward(self, plan, feats, record):
        dev = feats[0].device
        if plan.nb[0] == 0:
            return torch.zeros(0, dtype=torch.float32, device=dev), None
        ops.launch_held_packs()
        if record:
            self.train_steps += 1
        self.stamp = ("t" if record else "e", self.train_steps, ops.PARAM_EPOCH[0])
        if record and self.prepack_stamp is not None:
            self.stamp = self.prepack_stamp
        self.prepack_stamp = None
        if not (record and self.fast_stamp is not None and self.stamp == self.fast_stamp):
            self.fast_stamp = None
            self.prepare_all(record)
        fr = [f if isinstance(f, PT) else self.feat_rows(f) for f in feats]
        CH = arch.FEAT_CH
        pre = [None] * 5
        cats, uins = [None] * 4, [None] * 4
        top = max(l for l in range(5) if plan.nb[l] > 0)
        pre[top] = self.alloc(plan.rows[top], CH[top], dev)
        self.gather(fr[top], plan.srcrow[top], pre[top], plan.rows[top], CH[top])
        for l in range(top - 1, -1, -1):
            cin, cout, ccat = arch.SKIP[l]
            nc = plan.nb[l + 1]
            rowsC = int(plan.row0[l][nc])
            rows = plan.rows[l]
            pre[l] = self.alloc(rows, CH[l], dev)
            if nc:
                uin = self.alloc(rowsC, CH[l + 1], dev)
                ops.bilinear_fwd(pre[l + 1], uin, 0, 0, 0, 0, 0, CH[l + 1], boxdesc=plan.bil_d[l], row2box=plan.row2box[l])
                cat = self.alloc(rowsC, ccat, dev)
                pw, _, b = self.packw(f"skip_combine.{l}.up.0", record)
                self.rconv(uin, pw, cout, plan.rowdesc[l], rowsC, 3, y=cat.cols(CH[l], CH[l] + cout), bias=b, relu=True,
                           tiles=self.T32(plan, l, nc), tiles16=self.T16(plan, l, nc), tiles8=self.T8(plan, l, nc))
                self.gather(fr[l], plan.srcrow[l], cat.cols(0, CH[l]), rowsC, CH[l])
                pw, _, b = self.packw(f"skip_combine.{l}.cat_conv.0", record)
                self.rconv(cat, pw, cout, plan.rowdesc[l], rowsC, 1, y=pre[l].rows(0, rowsC), bias=b, relu=True)
                cats[l], uins[l] = cat, uin
            self.gather(fr[l], plan.srcrow[l], pre[l].rows(rowsC), rows - rowsC, CH[l], row_off=rowsC)
        rows0 = plan.rows[0]
        hid = self.alloc(rows0, 64, dev)
        pw, _, b = self.packw("seg_head.0", record)
        self.rconv(pre[0], pw, 64, plan.rowdesc[0], rows0, 3, y=hid, bias=b, relu=True, tiles=self.T32(plan, 0, plan.nb[0]),
                   tiles16=self.T16(plan, 0, plan.nb[0]), tiles8=self.T8(plan, 0, plan.nb[0]))
        flat = torch.empty(rows0, dtype=torch.float32, device=dev)
        if SEG_C1:      # one output channel: a per-pixel dot product kernel (csrc/seg.hip), not a 64-row MFMA tile with 63 zero rows
            _lib.call("kg_seg_conv3_c1", ptr(ops.base(hid)), ops.ld(hid), 64, ptr(self.P("seg_head.2.weight").detach()),
                      ptr(self.P("seg_head.2.bias").detach()), ptr(plan.rowdesc[0]), c_long(rows0), ptr(flat), ops.pl(a=hid), stream_ptr(),
                      fmt=ops.fmt_of(hid))
            if record:
                self.packw("seg_head.2", True)          # (the input gradient still multiplies the packed transposed weights)
        else:
            pw, _, b = self.packw("seg_head.2", record)
            self.rconv(hid, pw, 1, plan.rowdesc[0], rows0, 3, y_f32=flat, bias=b, tiles=self.T32(plan, 0, plan.nb[0]))
        self.last_logits = flat.clone() if getattr(self, "keep_logits", False) else None     # test hook: pre-sigmoid values
        ops.sigmoid_(flat)
        # (flat.detach(): `flat` itself becomes an OUTPUT of the autograd node that keeps `saved`; holding the output object there is a
        # reference cycle node -> saved -> flat -> grad_fn = node that only the cyclic GC can free -- with it a whole step's activations)
        saved = (pre, cats, uins, hid, flat.detach(), top) if record else None
        return flat, saved

    def conv_bwd(self, key, x, g, rowdesc, M, k, pgrads, dx=None, mask=None, t32=None, t16=None):
        """wgrad + bias grad (+ dgrad into dx) of one ragged conv; g must already be pre-activation."""
        w = self.P(key + ".weight")
        cout, cin = w.shape[0], w.shape[1]
        geom = (M, 0, 0, M, 1, k, k, 1, (k - 1) // 2)
        eng = self.m._engine
        gw = eng.new_grad(key + ".weight", w)
        use16 = t16 if (k == 3 and t16 is not None and M >= 0.35 * t16.shape[0] * 256) else None
        db = eng.new_grad(key + ".bias", self.P(key + ".bias"))
        pw_ = min(self.Pg, self.m._engine.pw)
        gw_ = g
        if isinstance(x, PT) and x.P > pw_:
            x = PT(x.t, pw_, x.ps)          # weight-gradient operands
        if isinstance(g, PT) and g.P > pw_:
            gw_ = PT(g.t, pw_, g.ps)
        ops.conv_wgrad(x, gw_, cin, cout, geom, [(gw, 0, cout)], mode=2, rowdesc=rowdesc, tiletab16=use16, bias_out=db)
        pgrads[key + ".weight"], pgrads[key + ".bias"] = gw, db
        if dx is not None:
            _, pwT, _ = self.packw(key, True)
            self.rconv(g, pwT, cin, rowdesc, M, k, y=dx, mask=mask, mode=3, tiles=t32, tiles16=t16)

    def run_backward(self, plan, saved, gflat, feat_shapes, gscale=None):
        ops.WGQ.begin()                  # (split reductions of the weight gradients: recorded, run a dozen per launch; ops.WgradReduceQueue)
        try:
            return self._run_backward(plan, saved, gflat, feat_shapes, gscale)
        finally:
            ops.WGQ.end()

    def _run_backward(self, plan, saved, gflat, feat_shapes, gscale=None):
        """gscale (half build): truthy when the engine carries a running gradient scale (engine.gscale, set by the caller from
        ops.grad_scale): gflat enters times that scale; after every level the gradient that moves on to the next coarser level is
        re-normalised on the device together with the feature gradients already produced (the adjoint of the 2x bilinear upsampling
        multiplies spatially coherent gradients by 4 per level; engine.renormalise); split-rows feature gradients are returned in
        the engine's scale at return, parameter gradients in the scale they were produced in (the caller divides them), fp32
        feature gradients are divided here."""
        eng = self.m._engine
        scaled = gscale is not None
        pre, cats, uins, hid, flat, top = saved
        dev = gflat.device
        CH = arch.FEAT_CH
        pgrads = {}
        rows0 = plan.rows[0]
        gz = self.galloc(rows0, 8, dev)
        ops.grad_pack(gflat, flat, gz, 1, 1, rows0, 1, 8, scale=eng.gscale[0:1] if scaled else None)
        dhid = self.galloc(rows0, 64, dev)
        t32_0, t16_0 = self.T32(plan, 0, plan.nb[0]), self.T16(plan, 0, plan.nb[0])
        self.conv_bwd("seg_head.2", hid, gz, plan.rowdesc[0], rows0, 3, pgrads, dx=dhid, mask=hid.hi(), t32=t32_0, t16=t16_0)
        dpre = self.galloc(rows0, 64, dev)
        self.conv_bwd("seg_head.0", pre[0], dhid, plan.rowdesc[0], rows0, 3, pgrads, dx=dpre, mask=pre[0].hi(), t32=t32_0, t16=t16_0)
        # Gradient w.r.t. the feature maps: per level the crop-gradient rows of the combine boxes (columns 0..C of the concat
        # gradient, rows [0, rowsC)) and of the boxes that end at this level (dpre rows [rowsC, rows)); reduced per feature
        # pixel in fixed box order with fp32 accumulation (kg_crop_grad_reduce) -- no atomics, bit-reproducible.
        gfeats = [None] * 5

        out_planes = getattr(plan, "out_planes", None)      # fused training forward: write the engine's split-bf16 gradient rows directly
        f32_outs, pt_outs = [], []

        def reduce_level(l, ga, rows_a, gb):
            n, c, h, w = feat_shapes[l]
            out = outp = None
            if out_planes is not None:
                outp = ops.alloc_pt(n * h * w, c, out_planes[l], dev, dtype=self.dt)
            else:
                out = torch.empty(n * h * w, c, dtype=torch.float32, device=dev)
            _lib.call("kg_crop_grad_reduce", ptr(ops.base(ga)), ops.ld(ga) if ga is not None else 0, ptr(ops.base(gb)),
                      ops.ld(gb) if gb is not None else 0, c_long(rows_a), ptr(plan.tab_d[l]), ptr(plan.bin_start_d[l]),
                      ptr(plan.bin_boxes_d[l]), BIN_SIZE[l], n, h, w, c, ptr(out), ptr(ops.base(outp)), ops.ld(outp) if outp is not None else 0,
                      ops.pl(a=ga if ga is not None else gb, b=gb if gb is not None else ga, y=outp), stream_ptr(),
                      fmt=ops.fmt_of(ga if ga is not None else gb))
            if scaled:      # (the scale this level's feature gradient was written in: converted to the final one below)
                (f32_outs if out is not None else pt_outs).append((out if out is not None else outp, c, eng.gscale))
            gfeats[l] = outp if outp is not None else out.view(n, h, w, c).permute(0, 3, 1, 2)

        for l in range(0, top):
            cin, cout, ccat = arch.SKIP[l]
            nc = plan.nb[l + 1]
            rowsC = int(plan.row0[l][nc])
            rows = plan.rows[l]
            nxt, dcat = None, None
            if nc:
                cat, uin = cats[l], uins[l]
                dcat = self.galloc(rowsC, ccat, dev)
                self.conv_bwd(f"skip_combine.{l}.cat_conv.0", cat, dpre.rows(0, rowsC), plan.rowdesc[l], rowsC, 1, pgrads, dx=dcat, mask=cat.hi())
                duin = self.galloc(rowsC, CH[l + 1], dev)
                self.conv_bwd(f"skip_combine.{l}.up.0", uin, dcat.cols(CH[l], CH[l] + cout), plan.rowdesc[l], rowsC, 3, pgrads, dx=duin,
                              t32=self.T32(plan, l, nc), t16=self.T16(plan, l, nc))
                nxt = self.galloc(plan.rows[l + 1], CH[l + 1], dev)
                ops.bilinear_bwd(duin, nxt, 0, 0, 0, 0, 0, CH[l + 1], boxdesc=plan.bil_d[l], row2box=plan.row2box[l + 1], mask=pre[l + 1].hi())
            reduce_level(l, dcat.cols(0, CH[l]) if dcat is not None else None, rowsC if dcat is not None else 0,
                         dpre.rows(rowsC) if rows > rowsC else None)
            dpre = nxt
            if scaled and nxt is not None:
                eng.renormalise(nxt, CH[l + 1])
        if dpre is not None:
            reduce_level(top, None, 0, dpre)
        for t, c, sc in pt_outs:                        # split-rows outputs: into the running scale the dense backward continues in
            if sc is not eng.gscale:
                ops.rows_scale(t, c, eng.gscale[0:1], sc[1:2])
        if f32_outs:                                    # fp32 outputs leave unscaled: each divided by the scale it was written in
            ops.scale_tensors([t for t, _, _ in f32_outs], [sc[1:2] for _, _, sc in f32_outs], flag=eng.flag_on(f32_outs[0][0].device))
        # parameters of levels that no box reached get zero gradients (autograd accumulates nothing for None)
        return gfeats, pgrads

    # ---- reference API --------------------------------------------------------------------------------
    def forward(self, feat_seg, bboxes):
        """== ResNet.forward_seg (KGnet.py:321-350): returns [mask_patches, mask_dets]."""
        plan = self.make_plan(feat_seg, bboxes, need_bins=torch.is_grad_enabled())
        nimg = len(bboxes)
        if plan.nb[0] == 0:
            return [[[] for _ in range(nimg)], [[] for _ in range(nimg)]]
        params = [self.P(k) for k in self.param_keys]
        record = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or any(f.requires_grad for f in feat_seg))
        flat = _SegFunction.apply(self, plan, record, *feat_seg, *params)
        return self.predictions(plan, flat, nimg)

    @staticmethod
    def predictions(plan, flat, nimg):
        """[mask_patches, mask_dets] exactly as the reference returns them (KGnet.py:350) over the flat probability buffer."""
        h0, w0 = plan.hw[0]
        r0 = plan.row0[0]
        # emit in the reference's order: image by image, boxes in input order (lazily: see LazyList)
        order = np.lexsort((plan.box_in_img, plan.img))
        img_sorted = plan.img[order]
        starts = np.searchsorted(img_sorted, np.arange(nimg + 1))
        mask_patches, mask_dets = [], []
        for i in range(nimg):
            idx = order[starts[i]:starts[i + 1]]
            mask_patches.append(LazyList(len(idx), lambda idx=idx: [flat[int(r0[b]):int(r0[b + 1])].view(int(h0[b]), int(w0[b])) for b in idx]))
            mask_dets.append(LazyList(len(idx), lambda idx=idx: [torch.from_numpy(plan.boxes[b].copy()) for b in idx]))
        out = SegPredictions([mask_patches, mask_dets])
        out.kg_meta = {"flat": flat, "order": order, "img": img_sorted, "boxes": plan.boxes[order],
                       "off": r0[:-1][order], "h": h0[order], "w": w0[order]}
        return out

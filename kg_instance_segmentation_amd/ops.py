"""Tensor-level wrappers over the C ABI (kg_* entry points of libkgnet_hip.so).

Activations are 2-D bf16 "rows" tensors [rows, C] with unit channel stride; the row stride (ld) may
exceed C when the tensor is a channel slice of a wider (concat) buffer.  PyTorch only owns memory
and streams here; every arithmetic kernel is hand-written HIP behind the C ABI.
"""
import functools
import math
import os

import torch

from . import _lib
from ._lib import c_int, c_long, c_float, c_double, ptr, stream_ptr

BF16 = torch.bfloat16
F16 = torch.float16      # rows of the half-precision build (libkgnet_hip_f16.so, csrc/kg_common.h): 11 significant bits per plane
_scratch = {}
PARAM_EPOCH = [0]      # bumped whenever parameters / running statistics are written behind PyTorch's version counters (engine.prepare)


def scratch_f32(nfloats, dev, tag="default"):
    """Grow-only fp32 scratch buffer per (device, tag)."""
    key = (str(dev), tag)
    t = _scratch.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(int(nfloats), 1 << 16), dtype=torch.float32, device=dev)
        _scratch[key] = t
    return t


class WgradReduceQueue:
    """Deferred split reductions of the weight gradients (kg_wgrad_reduce_defer / _flush): while `on`, conv_wgrad bump-allocates its fp32
    partials from one arena and only records the reduction; flush() runs the recorded reductions twelve per launch.  Flush points: the arena
    is full, a gradient is about to be read (scale_tensors, a data-parallel bucket's all-reduce), the end of a backward pass.
    The arena starts at ARENA0 and grows to the high-water mark of a backward pass (what the pass asked for in total, capped at
    KG_WGRAD_ARENA_MB, default 3 GB; a bench step at batch 8 x 512^2 asks for ~2 GB): a small model or batch keeps a small arena, the first
    step of a large one pays a few extra flushes."""

    ARENA = int(os.environ.get("KG_WGRAD_ARENA_MB", "3072")) << 18      # cap, floats
    ARENA0 = min(ARENA, 64 << 18)                                       # first allocation (64 MB)
    SMALL = ARENA // 8                                                 # larger partial sets are reduced at once from the shared scratch

    def __init__(self):
        self.enabled = os.environ.get("KG_WGRAD_BATCH", "1") != "0"
        self.on, self.depth = False, 0
        self.arena, self.off, self.need = None, 0, 0

    def begin(self):
        self.depth += 1
        if self.enabled and not self.on:
            self.on, self.need = True, 0
            for fmt in (0, 1):
                _lib.call("kg_wgrad_reduce_defer", 1, fmt=fmt)

    def end(self):
        self.depth -= 1
        if self.depth <= 0 and self.on:
            self.depth = 0
            try:
                self.flush()
            finally:           # a failed flush must not leave the library recording jobs that point into freed gradient tensors
                self.on, self.off = False, 0
                for fmt in (0, 1):
                    _lib.call("kg_wgrad_reduce_defer", 0, fmt=fmt)
            if self.arena is not None and self.need > self.arena.numel() and self.arena.numel() < self.ARENA:
                dev = self.arena.device       # the pass wanted more than the arena holds: the next one gets its high-water mark
                self.arena = None
                self.arena = torch.empty(min(self.ARENA, self.need), dtype=torch.float32, device=dev)

    def flush(self):
        if self.on:
            try:
                for fmt in (0, 1):
                    _lib.call("kg_wgrad_reduce_flush", stream_ptr(), fmt=fmt)
            finally:
                self.off = 0

    def release(self):
        """drops the arena (it is re-grown by the next backward pass)"""
        if not self.on:
            self.arena, self.off = None, 0

    def alloc(self, nfloats, dev, tag, extra=0):
        """partials buffer of a conv_wgrad call: the shared per-tag scratch (reduced right away) or, while deferring, a slice of the arena.
        extra: floats the same call will ask for next (its bias partials) -- a flush between the two requests of ONE call would restart the
        arena underneath the first buffer before its reduction is recorded"""
        if not self.on:
            return scratch_f32(nfloats, dev, tag), False
        n = (int(nfloats) + 63) // 64 * 64
        if n > self.SMALL:
            return scratch_f32(nfloats, dev, tag), True        # (True: the caller flushes right after recording -- the scratch is shared)
        ne = n + (int(extra) + 63) // 64 * 64
        self.need += n
        if self.arena is None or self.arena.device != torch.device(dev) or ne > self.arena.numel():
            self.flush()          # (the recorded reductions read the old arena on this stream; its memory is recycled in stream order)
            size = max(self.ARENA0, ne, 0 if self.arena is None else min(self.ARENA, 2 * self.arena.numel()))
            self.arena = None
            self.arena = torch.empty(size, dtype=torch.float32, device=dev)
        if self.off + ne > self.arena.numel():
            self.flush()
        t = self.arena[self.off:self.off + n]
        self.off += n
        return t, False


WGQ = WgradReduceQueue()


def flush_wgrad():
    """launches the weight-gradient reductions recorded so far (no-op outside a deferring backward pass)"""
    if WGQ.on:
        WGQ.flush()


def h2d(arr, dev):
    """numpy -> device through a pinned staging tensor, truly asynchronous.  (A pageable-memory copy blocks the host
    until all earlier work on the stream has drained, which serialises host glue and GPU kernels.)"""
    import numpy as np
    arr = np.ascontiguousarray(arr)
    st = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0] if arr.ndim else arr).dtype, pin_memory=True)
    st.numpy()[...] = arr
    return st.to(dev, non_blocking=True)


class PT:
    """Rows tensor in split-bf16 storage (csrc/kg_common.h "planes"): value = sum of P bf16 planes.  `t` is the [rows, C] view
    of plane 0 (row stride ld, possibly a column slice of a wider buffer); plane p starts `ps` elements after plane p-1.
    Every wrapper below takes a plain bf16 tensor (P = 1) or a PT for its rows operands."""
    __slots__ = ("t", "P", "ps")

    def __init__(self, t, P=1, ps=0):
        self.t, self.P, self.ps = t, P, (ps if P > 1 else 0)

    @property
    def shape(self):
        return self.t.shape

    @property
    def device(self):
        return self.t.device

    def cols(self, a, b):
        return PT(self.t[:, a:b], self.P, self.ps)

    def rows(self, a, b=None):
        return PT(self.t[a:b], self.P, self.ps)

    def hi(self):
        """plane 0 alone: the bf16 rounding of the value (what a single-plane consumer multiplies)"""
        return self.t

    def plane(self, p):
        return self.t.as_strided(self.t.shape, self.t.stride(), self.t.storage_offset() + p * self.ps)


def alloc_pt(rows, C, P, dev, zero=False, dtype=BF16):
    """[rows, C] in P planes: one [rows, P*C] buffer, plane p = columns p*C .. (p+1)*C.  dtype: BF16 or F16 (the 16-bit format)."""
    buf = (torch.zeros if zero else torch.empty)(rows, P * C, dtype=dtype, device=dev)
    return PT(buf[:, :C], P, C)


def base(x):
    return x.t if isinstance(x, PT) else x


def fmt_of(x):
    """rows format of a tensor / PT / PackedWeight: 0 = bfloat16 (libkgnet_hip.so), 1 = IEEE half (libkgnet_hip_f16.so)"""
    if isinstance(x, PackedWeight):
        x = x.buf
    dt = base(x).dtype
    if dt == F16:
        return 1
    assert dt == BF16, dt
    return 0


def dtype_of(x):
    return base(x).dtype


def nplanes(x):
    return (x.P, x.ps) if isinstance(x, PT) else (1, 0)


class _Planes(_lib.ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("a_planes", "a_pstride", "b_planes", "b_pstride", "c_planes", "c_pstride",
                                     "y_planes", "y_pstride", "w_planes", "reserved_")] + [("scale", _lib.c_void_p), ("oscale", _lib.c_void_p)]


_PL_CACHE = {}


def pl(a=None, b=None, c=None, y=None, w=1, scale=None, oscale=None):
    """kg_planes_t* for a call (None when every operand is single-plane and there is no scale).  a / b / c / y: rows operands
    (tensor, PT or None); scale: device fp32 scalar the fp32 -> rows conversions multiply by (kg_grad_scale); oscale: device fp32
    [Cout] factor of a forward conv's accumulator (folded inference BatchNorm)."""
    sp = scale.data_ptr() if scale is not None else 0
    op = oscale.data_ptr() if oscale is not None else 0
    key = nplanes(a) + nplanes(b) + nplanes(c) + nplanes(y) + (w, 0, sp, op)
    if key == (1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0):
        return None
    st = _PL_CACHE.get(key)
    if st is None:
        if len(_PL_CACHE) > 4096:
            _PL_CACHE.clear()
        st = _lib.ctypes.pointer(_Planes(*key[:10], sp or None, op or None))
        _PL_CACHE[key] = st
    return st


def vplanes(xP, wP):
    """number of virtual channel planes of a packed weight = kept products x_i * w_j, i + j < max(xP, wP) (csrc/kg_common.h kg_plane_pairs)"""
    T = max(xP, wP)
    return sum(1 for i in range(xP) for j in range(wP) if i + j < T)


def _rows(t):
    t = base(t)
    assert t.dim() == 2 and t.dtype in (BF16, F16) and (t.shape[1] == 1 or t.stride(1) == 1), (t.shape, t.stride(), t.dtype)
    return t


def ld(t):
    return base(t).stride(0)


def round_up(a, b):
    return (a + b - 1) // b * b


class PackQueue:
    """Deferred weight (re)packs: while `defer` is set, PackedWeight.pack / pack_rows only record a job; flush() (called by
    every conv wrapper) packs all recorded tensors with ONE kg_pack_weight_batch launch.  The job table is uploaded only
    when it differs from the previous step's (same tensors -> same pointers)."""

    DEFER = [False]      # (one switch for the queues of both formats)

    def __init__(self, fmt=0):
        self.fmt = fmt
        self.jobs, self.keep = [], []
        self.sig, self.table = None, None
        self.pending = None      # a built and uploaded batch whose launch is held back (flush_packs(hold=True))

    @property
    def defer(self):
        return PackQueue.DEFER[0]

    @defer.setter
    def defer(self, v):
        PackQueue.DEFER[0] = v

    def add(self, w, pw, row0, c0, transposed, rowmap, tap_stride=0, tap_pitch=0, cin_pad=None):
        Cout, Cin, KH, KW = w.shape
        gx = (Cout + 63) // 64 if transposed else Cout
        gy = Cin if transposed else (Cin + 63) // 64
        self.jobs.append((w.data_ptr(), pw.buf.data_ptr(), rowmap.data_ptr() if rowmap is not None else 0, Cout, Cin, KH * KW,
                          pw.K, pw.cin_pad if cin_pad is None else cin_pad, row0, c0, 1 if transposed else 0, gx, gx * gy, pw.xP, pw.wP, tap_stride, tap_pitch))
        self.keep.append((w, pw, rowmap))

    def flush(self, hold=False):
        if not self.jobs:
            return
        import numpy as np
        dt = np.dtype([("w", "<u8"), ("dst", "<u8"), ("rowmap", "<u8")] + [(n, "<i4") for n in
                      ("Cout", "Cin", "taps", "K", "cin_pad", "row0", "c0", "transposed", "gx", "blk0", "xP", "wP", "tap_stride", "tap_pitch")])
        arr = np.zeros(len(self.jobs), dt)
        blk = 0
        for i, j in enumerate(self.jobs):
            arr[i] = j[:12] + (blk,) + j[13:17]
            blk += j[12]
        dev = self.keep[0][0].device
        sig = arr.tobytes()
        if sig != self.sig or self.table is None or self.table.device != dev:
            self.table = h2d(arr.view(np.uint8).reshape(-1), dev)
            self.sig = sig
        with torch.cuda.device(dev):
            st = stream_ptr()            # the launch -- now or held back -- goes to the stream the parameters were written on (the optimizer's)
        launch = (self.table, len(self.jobs), blk, dev, self.keep, st)       # (keep: the tensors stay alive until the launch is enqueued)
        self.jobs, self.keep = [], []
        if hold:
            self.pending = launch
        else:
            self._launch(launch)

    def _launch(self, launch):
        table, n, blk, dev, _, st = launch
        with torch.cuda.device(dev):
            _lib.call("kg_pack_weight_batch", ptr(table), n, blk, st, fmt=self.fmt)

    def launch_pending(self):
        launch, self.pending = self.pending, None
        if launch is not None:
            self._launch(launch)


PACKQ = PackQueue(0)
PACKQ16 = PackQueue(1)


def flush_packs(hold=False):
    """packs every weight queued since the last conv launch (one kg_pack_weight_batch launch per 16-bit format).
    hold (Engine.prepack, called by the optimizer right after its update kernel): build and upload the job tables now but keep the LAUNCH
    back until launch_held_packs() -- the first thing the next step does (optim.Adam.zero_grad, forward_dec, forward_seg, any flush).  The
    per-step loss read-back (train.py:156) then waits for the update kernel only, and the 0.6 ms pack kernel runs while the host is busy with
    zero_grad and the forward's entry, where the GPU used to idle."""
    launch_held_packs()
    if PACKQ.jobs:
        PACKQ.flush(hold)
    if PACKQ16.jobs:
        PACKQ16.flush(hold)


def launch_held_packs():
    if PACKQ.pending is not None:
        PACKQ.launch_pending()
    if PACKQ16.pending is not None:
        PACKQ16.launch_pending()


class PackedWeight:
    """bf16 [rows_pad][K] matrix for the conv kernels: K = taps * vplanes * cin_pad (padded to 64).  xP / wP: split-bf16 planes of
    the activations it multiplies / of the weights themselves (csrc/kg_common.h): per tap the row holds the virtual channels
    of the kept products x_i * w_j (smallest first), a copy of w plane j of cin_pad channels each; groups > 1: that many such blocks side by side
    (the fused second-layer heads), tap stride = groups * vplanes * cin_pad."""

    def __init__(self, rows, taps, cin_pad, dev, xP=1, wP=1, groups=1, dtype=BF16):
        self.rows, self.taps, self.cin_pad, self.xP, self.wP, self.groups = rows, taps, cin_pad, xP, wP, groups
        self.vp = vplanes(xP, wP)
        self.tap_stride = groups * self.vp * cin_pad
        # (8-channel inputs: conv_small.hip reads whole tap quads)
        self.K = round_up((round_up(taps, 4) if cin_pad == 8 else taps) * self.tap_stride, 64)
        self.buf = torch.zeros(round_up(rows, 384), self.K, dtype=dtype, device=dev)   # rows cover any 64/128/192 cout tile

    def pack(self, w, row0=0, c0=0, transposed=False):
        """w: fp32 OIHW parameter.  forward: rows=Cout, channels=Cin; transposed (dgrad): rows=Cin, channels=Cout."""
        Cout, Cin, KH, KW = w.shape
        assert w.dtype == torch.float32 and w.is_contiguous()
        assert self.groups == 1
        if PACKQ.defer:
            (PACKQ16 if fmt_of(self) else PACKQ).add(w, self, row0, c0, transposed, None)
            return
        _lib.call("kg_pack_weight", ptr(w), ptr(self.buf), Cout, Cin, KH, KW, self.K, self.cin_pad, row0, c0,
                  1 if transposed else 0, self.xP, self.wP, stream_ptr(), fmt=fmt_of(self))


    def pack_narrow(self, w, chan_slot, row0=0):
        """transposed (input-gradient) packing of a narrow conv for conv_halo(..., narrow=chan_slot): chan_slot = 8 / 16 channels per kernel column,
        8 column slots per kernel row (kg_pack_weight_narrow); self = PackedWeight(Cin, 7 * chan_slot // 8, 64, ...)"""
        Cout, Cin, KH, KW = w.shape
        assert w.dtype == torch.float32 and w.is_contiguous() and KH == KW == 7 and Cout <= chan_slot and self.xP == self.wP == 1
        assert self.cin_pad == 64 and self.K >= 7 * 8 * chan_slot
        if PACKQ.defer:
            (PACKQ16 if fmt_of(self) else PACKQ).add(w, self, row0, 0, True, None, tap_stride=chan_slot, tap_pitch=8, cin_pad=chan_slot)
            return
        _lib.call("kg_pack_weight_narrow", ptr(w), ptr(self.buf), Cout, Cin, KH, KW, self.K, row0, 0, chan_slot, 8, stream_ptr(), fmt=fmt_of(self))

    def rows_from(self, r0):
        """View of the packed matrix starting at row r0 (a cout-block-aligned slice of a fused weight)."""
        v = PackedWeight.__new__(PackedWeight)
        v.rows, v.taps, v.cin_pad, v.K, v.buf = self.rows - r0, self.taps, self.cin_pad, self.K, self.buf[r0:]
        v.xP, v.wP, v.groups, v.vp, v.tap_stride = self.xP, self.wP, self.groups, self.vp, self.tap_stride
        return v

    def pack_rows(self, w, rowmap, group=0):
        """forward packing of w (fp32 OIHW) into plane group `group`, output channel co going to packed row rowmap[co] (device int32)."""
        Cout, Cin, KH, KW = w.shape
        assert w.dtype == torch.float32 and w.is_contiguous() and rowmap.dtype == torch.int32 and rowmap.numel() == Cout
        c0 = group * self.vp * self.cin_pad
        if PACKQ.defer:
            (PACKQ16 if fmt_of(self) else PACKQ).add(w, self, 0, c0, False, rowmap, self.tap_stride)
            return
        _lib.call("kg_pack_weight_rows", ptr(w), ptr(self.buf), Cout, Cin, KH, KW, self.K, self.cin_pad, ptr(rowmap), c0,
                  self.xP, self.wP, self.tap_stride, stream_ptr(), fmt=fmt_of(self))


def heads2_layout():
    """Virtual-cout layout of the fused second-layer head conv (conv_halo.hip GM = 1): MFMA row group i owns the virtual
    couts 16q + 4i + r.  Returns (rows, vmap): rows[h] = virtual cout of every channel of head h (kp 5, short 10, mid 40),
    vmap[v] = channel in (kp 0-4 | short 5-14 | mid 15-54) or -1."""
    grp = [[16 * q + 4 * i + r for q in range(4) for r in range(4)] for i in range(4)]
    rows = [grp[0][:5], grp[1][:10], grp[2] + grp[3] + grp[0][5:13]]
    vmap = [-1] * 64
    base = (0, 5, 15)
    for h in range(3):
        for c, v in enumerate(rows[h]):
            vmap[v] = base[h] + c
    return rows, vmap


def conv_halo_heads2(x, pw, bias64, vmap, kp, sh, md, N, H, W, C, kp_sigmoid=True):
    """kg_conv2d_halo_heads2: x = fused hidden rows [N*H*W, >=3C]; kp/sh/md fp32 NCHW outputs (kp gets the sigmoid)."""
    flush_packs()
    assert kp.is_contiguous() and sh.is_contiguous() and md.is_contiguous() and vmap.dtype == torch.int32
    assert nplanes(x)[0] == pw.xP and pw.groups == 3
    _lib.call("kg_conv2d_halo_heads2", ptr(_rows(x)), ptr(pw.buf), ptr(bias64), ptr(vmap), ptr(kp), ptr(sh), ptr(md), N, H, W, C,
              ld(x), pw.K, 1 if kp_sigmoid else 0, pl(a=x, w=pw.wP), stream_ptr(), fmt=fmt_of(x))


def conv_igemm(x, pw, cout, geom, y=None, y_f32=None, bias=None, res=None, mask=None, relu=False, mode=0,
               rowdesc=None, tile=0, oscale=None):
    """geom = (M, H, W, OH, OW, KH, KW, stride, pad): H, W = gathered tensor's dims, OH, OW = output dims."""
    flush_packs()
    M, H, W, OH, OW, KH, KW, stride, pad = geom
    assert nplanes(x)[0] == pw.xP, (nplanes(x), pw.xP)
    f32_C = 0
    if y_f32 is not None:
        assert y_f32.dtype == torch.float32 and y_f32.is_contiguous()
        f32_C = y_f32.shape[1] if y_f32.dim() == 4 else 1
    _lib.call("kg_conv2d_igemm", ptr(_rows(x)), ptr(pw.buf), ptr(bias), ptr(base(y)), ptr(y_f32), ptr(base(res)), ptr(base(mask)), ptr(rowdesc),
              M, H, W, OH, OW, pw.cin_pad, ld(x), cout, ld(y) if y is not None else 0,
              ld(res) if res is not None else 0, ld(mask) if mask is not None else 0, pw.K, KH, KW, stride, pad, 1,
              mode, 1 if relu else 0, f32_C, tile, pl(a=x, b=res, y=y, w=pw.wP, oscale=oscale), stream_ptr(), fmt=fmt_of(x))


def conv7_narrow(x, pw, cout, N, H, W, y, mask=None, chan_lo=0, chan_slot=8, flip=True, algo_cin=None):
    """kg_conv7_narrow: the input gradient (flip) of a 7x7 "same" conv whose single-plane dY rows x carry data in the chan_slot (8 / 16) channels from chan_lo
    on (pw from PackedWeight.pack_narrow): persistent workgroups, the packed weights resident in LDS, compact halos (conv7_narrow.hip).
    algo_cin: live input channels (FLOP accounting of bench.py's timer; unused here)."""
    flush_packs()
    assert nplanes(x)[0] == 1 and nplanes(y)[0] == 1 and pw.xP == pw.wP == 1 and chan_slot in (8, 16)
    xb, yb, mb = _rows(x), base(y), base(mask)
    _lib.call("kg_conv7_narrow", ptr(xb), ptr(pw.buf), ptr(yb), ptr(mb), N, H, W, ld(xb), chan_lo, chan_slot, cout, ld(yb),
              ld(mb) if mb is not None else 0, pw.K, 1 if flip else 0, stream_ptr(), fmt=fmt_of(xb))


USE_HALO = True
WGRAD128 = True           # (module constants: tests and probes may flip them; the environment switches of rounds 1-5 are gone, docs/history.md)
USE_C3 = True
IM2COL_WGRAD = True


HALO_WC = 0               # tuning override of the halo kernels' cout blocks per workgroup (0 = library default)


USE_WS = True             # the weight-stationary 64 -> 64 kernel (conv3_ws.hip)


def conv_halo(x, pw, cout, N, H, W, KS, y=None, y_f32=None, bias=None, res=None, mask=None, relu=False, flip=False, wc=0,
              tiletab=None, total_rows=0, k1skip=False, algo_cin=None, tiletab16=None, oscale=None, tiletab8=None, narrow=0):
    """Stride-1 "same" KSxKS conv (or its input gradient when flip) with the input halo resident in LDS.
    tiletab (int32 [ntiles,4] device tensor): ragged boxes instead of N images of HxW.
    k1skip (7x7 only): the packed weights are zero for channels 32..63 of every 64-channel chunk.
    narrow = 8 / 16 (7x7 input gradient of a conv whose dY rows carry data in channels 0..7 / 8..23 only; pw from PackedWeight.pack_narrow): 4 / 2 kernel
    columns per MFMA k-step (conv_halo.hip GM = 3 / 4).
    algo_cin: number of input channels that carry data (FLOP accounting of bench.py's timer; unused here)."""
    flush_packs()
    assert nplanes(x)[0] == pw.xP, (nplanes(x), pw.xP)
    if (USE_WS and KS == 3 and pw.cin_pad == 64 and cout == 64 and pw.xP == 2 and pw.wP == 2 and y is not None and y_f32 is None and res is None
            and mask is None and not flip and oscale is None and wc == 0 and HALO_WC == 0 and (tiletab is None or tiletab8 is not None)
            and nplanes(y)[0] <= 2 and not conv_stats_armed()):
        # full-resolution 64 -> 64 convs on hi + lo planes: weights in registers, two workgroups per CU (conv3_ws.hip)
        _lib.call("kg_conv3x3_ws", ptr(_rows(x)), ptr(pw.buf), ptr(bias), ptr(base(y)), N, H, W, ld(x), ld(y), pw.K, 1 if relu else 0,
                  ptr(tiletab8), tiletab8.shape[0] if tiletab8 is not None else 0, pl(a=x, y=y, w=pw.wP), stream_ptr(), fmt=fmt_of(x))
        return
    planes = pl(a=x, b=res, y=y, w=pw.wP, oscale=oscale)      # (oscale: folded inference BatchNorm, y = act(acc * oscale + bias + res))
    x, y, res, mask = base(x), base(y), base(res), base(mask)
    if (USE_C3 and planes is None and KS == 3 and pw.cin_pad == 64 and y is not None and y_f32 is None and wc == 0 and HALO_WC == 0
            and (tiletab is None or tiletab16 is not None)):
        # 64 input channels: persistent kernel with resident weights and double-buffered halos (conv3_c64.hip)
        flush_packs()
        _lib.call("kg_conv3x3_c64", ptr(_rows(x)), ptr(pw.buf), ptr(bias), ptr(y), ptr(res), ptr(mask), N, H, W, ld(x), cout, ld(y),
                  ld(res) if res is not None else 0, ld(mask) if mask is not None else 0, pw.K, 1 if flip else 0, 1 if relu else 0,
                  ptr(tiletab16), tiletab16.shape[0] if tiletab16 is not None else 0, stream_ptr(), fmt=fmt_of(x))
        return
    wc = wc or HALO_WC
    if k1skip:
        assert KS == 7 and wc in (0, 1)
        wc = 1 | 256
    if narrow:
        assert KS == 7 and wc in (0, 1) and flip and narrow in (8, 16) and planes is None and tiletab is None and y is not None and y_f32 is None
        wc = 1 | (512 if narrow == 8 else 1024)
    f32_C = 0
    if y_f32 is not None:
        f32_C = y_f32.shape[1] if y_f32.dim() == 4 else 1
    _lib.call("kg_conv2d_halo", ptr(_rows(x)), ptr(pw.buf), ptr(bias), ptr(y), ptr(y_f32), ptr(res), ptr(mask), N, H, W,
              pw.cin_pad, ld(x), cout, ld(y) if y is not None else 0, ld(res) if res is not None else 0,
              ld(mask) if mask is not None else 0, pw.K, KS, 1 if flip else 0, 1 if relu else 0, f32_C, wc, ptr(tiletab),
              tiletab.shape[0] if tiletab is not None else 0, total_rows, planes, stream_ptr(), fmt=fmt_of(x))


USE_1X1 = True
GATHER_1X1 = True


def conv1x1(x, pw, cout, y, bias=None, res=None, mask=None, relu=False):
    """1x1 stride-1 conv / input gradient as a streaming GEMM over the rows of x (kg_conv1x1); single-plane bf16 only."""
    flush_packs()
    assert pw.vp == 1 and all(nplanes(t)[0] == 1 for t in (x, y, res))
    x, y, res, mask = base(x), base(y), base(res), base(mask)
    _lib.call("kg_conv1x1", ptr(_rows(x)), ptr(pw.buf), ptr(bias), ptr(_rows(y)), ptr(res), ptr(mask), c_long(x.shape[0]),
              pw.cin_pad, pw.K, ld(x), cout, ld(y), ld(res) if res is not None else 0, ld(mask) if mask is not None else 0,
              1 if relu else 0, stream_ptr(), fmt=fmt_of(x))


def can_1x1(x, pw, KH, stride, pad, y, y_f32, res=None):
    if pw.vp > 1 or any(nplanes(t)[0] > 1 for t in (x, y, res)):
        return False      # split-bf16 planes: the gather kernel (conv_gather.hip) walks the virtual channels
    if GATHER_1X1 and pw.cin_pad >= 192 and pw.rows > 64:
        return False      # compute-heavy 1x1 (K >= 192, Cout > 64): the LDS-ring gather kernel (conv_gather.hip) is faster
    return (USE_1X1 and KH == 1 and stride == 1 and pad == 0 and y is not None and y_f32 is None and pw.cin_pad % 64 == 0
            and 64 <= pw.cin_pad <= 1024 and x.shape[1] >= pw.cin_pad and y.shape[0] == x.shape[0])


CONV_TINY_WGS = 96        # regular workgroups below which a launch goes to the split-K kernel (0 = never)
CONV_TINY_TILES = 384


def conv_auto(x, pw, cout, geom, N, y=None, y_f32=None, bias=None, res=None, mask=None, relu=False, transposed=False, tile=0, oscale=None,
              tiny=True):
    """Dense conv forward (transposed=False) or input gradient (True): picks the LDS-halo kernel for stride-1
    "same" 3x3/7x7 convs over 64-channel-aligned inputs, else the gather implicit GEMM.  tiny: launches whose output gives the
    regular kernels fewer than CONV_TINY_WGS workgroups (single-image inference, the deepest layers of a small batch) may go to
    the split-K kernel (conv_tiny.hip); False for a conv that is armed for BatchNorm statistics (that epilogue lives in the regular kernels)."""
    M, H, W, OH, OW, KH, KW, stride, pad = geom
    halo_ok = (USE_HALO and stride == 1 and KH == KW and KH in (3, 7) and pad == KH // 2 and pw.cin_pad % 64 == 0
               and x.shape[1] >= pw.cin_pad and not (y_f32 is not None and (res is not None or mask is not None)))
    if tiny and tile == 0 and CONV_TINY_WGS and y is not None and y_f32 is None and pw.cin_pad % 64 == 0 and x.shape[1] >= pw.cin_pad and KH * KW <= 9:
        wgs = (N * math.ceil(OH / 16) * math.ceil(OW / 32) * math.ceil(cout / 64)) if halo_ok else math.ceil(M / 256) * math.ceil(cout / 128)
        tiles = math.ceil(M / 64) * math.ceil(cout / 64)
        # (no operand reuse inside a 64 x 64 tile: measured 1.75 ns per (tile, K unit) against 0.85 us per K unit of a regular workgroup
        # -- past ~400 tiles the regular kernel's single round is as fast)
        if wgs < CONV_TINY_WGS and wgs < tiles <= CONV_TINY_TILES:
            conv_igemm(x, pw, cout, geom, y=y, bias=bias, res=res, mask=mask, relu=relu, mode=1 if transposed else 0, tile=6, oscale=oscale)
            return "tiny"
    if (USE_HALO and stride == 1 and KH == KW and KH in (3, 7) and pad == KH // 2 and pw.cin_pad % 64 == 0
            and x.shape[1] >= pw.cin_pad and not (y_f32 is not None and (res is not None or mask is not None))):
        conv_halo(x, pw, cout, N, OH, OW, KH, y=y, y_f32=y_f32, bias=bias, res=res, mask=mask, relu=relu, flip=transposed, oscale=oscale)
        return "halo"
    if KH == KW and oscale is None and can_1x1(x, pw, KH, stride, pad, y, y_f32, res):
        conv1x1(x, pw, cout, y, bias=bias, res=res, mask=mask, relu=relu)
        return "1x1"
    conv_igemm(x, pw, cout, geom, y=y, y_f32=y_f32, bias=bias, res=res, mask=mask, relu=relu, mode=1 if transposed else 0, tile=tile, oscale=oscale)
    return "igemm"


WGRAD_RING = True
WGRAD_RING_WGS = 256      # workgroups of a ring launch (one 96 / 144 KB workgroup per CU)


WGRAD_TR = [True]      # mirror of the library's kg_set_wgrad_tr switch (set_wgrad_tr below keeps the two in step)


def set_wgrad_tr(on):
    """test switch of kg_conv2d_wgrad: LDS transpose reads (default) or scalar fragment loads -- both libraries and the split sizing"""
    WGRAD_TR[0] = bool(on)
    for fmt in (0, 1):
        _lib.call("kg_set_wgrad_tr", 1 if on else 0, fmt=fmt)


def wgrad_splits(M, cin_lim, cout_lim, taps, nelem):
    """pixel splits of kg_conv2d_wgrad, sized for the tile the library will pick: the conditions below are kg_conv2d_wgrad's own
    (conv_wgrad.hip: ring iff transpose reads && cin >= 128 && cout >= 64; 128 x 128 iff transpose reads
    && cin, cout >= 128)"""
    chunks = math.ceil(M / 64)
    if WGRAD_RING and WGRAD_TR[0] and cin_lim >= 128 and cout_lim >= 64:
        # conv_wgrad_ring_kernel (256 x 128 tiles, or 128 x 128 below 256 couts; 144 / 96 KB of LDS: one workgroup per CU): one round of at
        # most 256 workgroups
        base = math.ceil(cin_lim / 128) * math.ceil(cout_lim / (256 if cout_lim >= 256 else 128)) * taps
        s = max(1, min(WGRAD_RING_WGS // base if base <= WGRAD_RING_WGS else 1, max(1, chunks // 3)))
        while s > 1 and s * nelem * 4 > (768 << 20):
            s -= 1
        return s
    t = 128 if (WGRAD128 and WGRAD_TR[0] and cin_lim >= 128 and cout_lim >= 128) else 64      # output tile of kg_conv2d_wgrad
    base = math.ceil(cin_lim / t) * math.ceil(cout_lim / t) * taps
    s = max(1, min(math.ceil(2048 / base), max(1, chunks // 4)))
    while s > 1 and s * nelem * 4 > (768 << 20):
        s -= 1
    return s


@functools.lru_cache(maxsize=4096)      # (the search walks up to 2048 candidates; ~110 calls per step with a few dozen distinct shapes)
def halo_wgrad_splits(nblk, tiles, cit, taps, nelem, cus=256):
    """Pixel splits of kg_conv2d_wgrad_halo: one workgroup per CU is resident (2 x 47..78 KB of LDS + ~250 VGPRs x 8 waves), so
    the launch runs in ceil(nblk * S / 256) rounds; pick S by a small cost model (round count x tiles per workgroup + per-workgroup
    prologue / partial write + the reduction's bytes) instead of a fixed 2 rounds -- 192 x 2 = 384 workgroups is 1.5 rounds."""
    tile_us = 256 * 64 * cit * taps * 2 / 5.3e6          # one 16x16 tile at ~1.35 PFLOP/s / 256 CUs
    best, best_t = 1, None
    for S in range(1, max(1, min(tiles, 2048 // nblk if nblk <= 2048 else 1)) + 1):
        if S > 1 and S * nelem * 4 > (768 << 20):
            break
        rounds = math.ceil(nblk * S / cus)
        t = rounds * (math.ceil(tiles / S) * tile_us + 10.0) + S * nelem * 8 / 3e6
        if S >= 8 and S % 8 == 0:
            t *= 0.97                                    # multiples of 8 enable the kernel's XCD-aware block mapping
        if best_t is None or t < best_t:
            best, best_t = S, t
    return best


def wgrad_pairs(x, dy):
    """(x plane, dY plane) products of a weight gradient over split-bf16 operands: i + j < max(xP, dP)."""
    xP, dP = nplanes(x)[0], nplanes(dy)[0]
    T = max(xP, dP)
    return sorted(((i, j) for j in range(dP) for i in range(xP) if i + j < T), key=lambda p: -(p[0] + p[1]))      # smallest products first


def conv_wgrad(x, dy, cin, cout, geom, grads, mode=0, rowdesc=None, accumulate=False, N=None, tiletab16=None, bias_out=None):
    """grads: list of (fp32 OIHW grad tensor, cout_offset, cout_count) sharing x (fused heads) or one entry.
    Dense stride-1 "same" 3x3/7x7 convs (N given) use the LDS-halo kernel, everything else the gather kernel.
    bias_out (fp32 [cout], optional): receives the bias gradient (sum of dy over rows) -- for free inside the halo kernel
    (one more all-ones unit on a wave with an idle unit slot), with kg_bias_grad otherwise.
    x / dy may be split-bf16 (PT): every kept plane product is one more set of pixel-split partials for the fixed-order reduction."""
    M, H, W, OH, OW, KH, KW, stride, pad = geom
    pairs = wgrad_pairs(x, dy)
    xps, dps = nplanes(x)[1], nplanes(dy)[1]
    planed = len(pairs) > 1 or nplanes(x)[0] > 1 or nplanes(dy)[0] > 1
    xb, dyb = _rows(x), _rows(dy)
    if IM2COL_WGRAD and mode == 0 and N is not None and cin <= 4 and KH * KW >= 25 and len(grads) == 1 and not accumulate:
        # stem conv1 (3 -> 64, 7x7 s2): im2col to [M][taps*cin] and ONE 1x1 weight-gradient GEMM instead of 49 per-tap launches
        # that pad 3 channels to a 64-wide tile (planes: im2col is a gather, so every plane is gathered separately)
        Kc = KH * KW * cin
        Kpad = round_up(Kc, 8)
        xP = nplanes(x)[0]
        col = alloc_pt(M, Kpad, xP, xb.device, zero=Kpad != Kc, dtype=xb.dtype)
        for p_ in range(xP):
            _lib.call("kg_im2col_small", ctypes_offset(xb, p_ * xps), ctypes_offset(col.t, p_ * col.ps), N, H, W, OH, OW, KH, KW, stride, pad, cin,
                      ld(xb), ld(col), stream_ptr(), fmt=fmt_of(xb))
        g, off, cnt = grads[0]
        tmp = torch.empty(cout, Kc, 1, 1, dtype=torch.float32, device=xb.device)
        conv_wgrad(col if xP > 1 else col.t, dy, Kc, cout, (M, OH, OW, OH, OW, 1, 1, 1, 0), [(tmp, off, cnt)])
        flush_wgrad()       # (tmp is read right away)
        g.copy_(tmp.view(cnt, KH * KW, cin).permute(0, 2, 1).reshape(g.shape))     # [co][tap][ci] -> OIHW
        if bias_out is not None:
            bias_grad(dy, cout, bias_out)
        return "im2col"
    cin_lim = min(round_up(cin, 8), xb.shape[1])
    cout_lim = min(round_up(cout, 8), dyb.shape[1])
    nelem = cout * KH * KW * cin
    halo = (USE_HALO and stride == 1 and KH == KW and KH in (3, 7) and pad == KH // 2
            and ((mode == 0 and N is not None) or (mode == 2 and tiletab16 is not None)))
    np_ = len(pairs)
    planes = pl(a=x, b=dy)
    if halo:
        cit = (32 if (cout_lim <= 16 and cin_lim >= 32 and bias_out is not None and not planed) else 16) if KH == 7 else 64   # input channels per workgroup (wgrad_halo.hip)
        nblk = math.ceil(cin_lim / cit) * math.ceil(cout_lim / 64)
        tiles = tiletab16.shape[0] if tiletab16 is not None else N * math.ceil(H / 16) * math.ceil(W / 16)
        S = halo_wgrad_splits(nblk, tiles * np_, cit, KH * KW, nelem)      # (plane products = more tiles to walk)
        fused_bias = bias_out is not None and not planed
        part, big = WGQ.alloc(S * nelem, xb.device, "wgrad", extra=S * cout if fused_bias else 0)
        dbp = None
        if fused_bias:
            dbp, big2 = WGQ.alloc(S * cout, xb.device, "wgrad_bias")
            big = big or big2
        _lib.call("kg_conv2d_wgrad_halo", ptr(xb), ptr(dyb), ptr(part), N or 0, H, W, ld(xb), ld(dyb), cin, cout, cin_lim, cout_lim, KH, S,
                  c_long(nelem), ptr(tiletab16), tiletab16.shape[0] if tiletab16 is not None else 0, ptr(dbp), planes, stream_ptr(), fmt=fmt_of(x))
        if dbp is None and bias_out is not None:       # planed operands: the all-ones unit would count every plane product
            bias_grad(dy, cout, bias_out, accumulate=accumulate)
    else:
        dbp = None
        S = wgrad_splits(M * np_, cin_lim, cout_lim, KH * KW, nelem)
        part, big = WGQ.alloc(S * nelem, xb.device, "wgrad")
        _lib.call("kg_conv2d_wgrad", ptr(xb), ptr(dyb), ptr(part), ptr(rowdesc), M, H, W, OH, OW, ld(xb), ld(dyb), cin, cout, cin_lim, cout_lim,
                  KH, KW, stride, pad, 1, mode, S, c_long(nelem), planes, stream_ptr(), fmt=fmt_of(xb))
        if bias_out is not None:
            bias_grad(dy, cout, bias_out, accumulate=accumulate)
    contiguous = all(grads[i][1] + grads[i][2] == grads[i + 1][1] for i in range(len(grads) - 1))
    import ctypes
    acc = 1 if accumulate else 0
    if len(grads) <= 4 and contiguous:      # one reduction launch: all heads fused along Cout, and the bias partials of the halo kernel with them
        gp = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g, _, _ in grads])
        cn = (ctypes.c_int * len(grads))(*[cnt for _, _, cnt in grads])
        _lib.call("kg_wgrad_reduce_bias", ctypes_offset(part, grads[0][1] * KH * KW * cin), gp, cn, len(grads), cin, KH, KW, S,
                  c_long(nelem), acc, ptr(dbp), ptr(bias_out) if dbp is not None else None, cout if dbp is not None else 0, stream_ptr())
    else:
        if dbp is not None:
            _lib.call("kg_bias_grad_final", ptr(dbp), ptr(bias_out), S, cout, acc, stream_ptr())
        for g, off, cnt in grads:
            _lib.call("kg_wgrad_reduce", ctypes_offset(part, off * KH * KW * cin), ptr(g), cnt, cin, KH, KW, S, c_long(nelem), acc, stream_ptr())
    if big:
        WGQ.flush()
    return "halo" if halo else "gather"


def wgrad_halo(x, dy, part, N, H, W, cin, cout, cin_lim, cout_lim, KS, S, nelem, tiletab16=None, dbp=None):
    _lib.call("kg_conv2d_wgrad_halo", ptr(x), ptr(dy), ptr(part), N, H, W, ld(x), ld(dy), cin, cout, cin_lim, cout_lim, KS, S,
              c_long(nelem), ptr(tiletab16), tiletab16.shape[0] if tiletab16 is not None else 0, ptr(dbp), None, stream_ptr(), fmt=fmt_of(x))


def ctypes_offset(t, elem_off):
    return _lib.c_void_p(t.data_ptr() + elem_off * t.element_size())


def bias_grad(dy, C, db, accumulate=False):
    """db[c] (+)= sum over rows of dy (every plane of a split-bf16 dy is summed in fp32 and added)."""
    P, ps = nplanes(dy)
    dyb = _rows(dy)
    sc = scratch_f32(2048 * max(C, 1), dyb.device, "bias")
    for p in range(P):
        _lib.call("kg_bias_grad", ctypes_offset(dyb, p * ps), ptr(db), ptr(sc), sc.numel(), dyb.shape[0], C, ld(dyb),
                  1 if (accumulate or p > 0) else 0, stream_ptr(), fmt=fmt_of(dyb))


def img_pack(img, P=1, dtype=BF16):
    """fp32 NCHW image -> [N*H*W, 8] rows (3 real channels + zero padding) in P planes."""
    N, C, H, W = img.shape
    img = img.contiguous().float()
    out = alloc_pt(N * H * W, 8, P, img.device, dtype=dtype)
    _lib.call("kg_img_pack", ptr(img), ptr(out.t), ld(out), N, C, H, W, pl(y=out), stream_ptr(), fmt=fmt_of(out))
    return out if P > 1 else out.t


def bn_stats_train(x, C, gamma, beta, rmean, rvar, momentum=0.1, eps=1e-5):
    """Returns (mean, invstd, scale, shift) fp32 [C]; updates running stats in place (may be None)."""
    xb = _rows(x)
    dev = xb.device
    st = torch.empty(4, C, dtype=torch.float32, device=dev)
    sc = scratch_f32(2 * C * 512, dev, "bn")
    _lib.call("kg_bn_stats_train", ptr(xb), ld(xb), xb.shape[0], C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
              c_float(momentum), c_float(eps), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), ptr(sc), sc.numel(), pl(a=x), stream_ptr(), fmt=fmt_of(xb))
    return st[0], st[1], st[2], st[3]


CONV_BN_STATS = True      # BatchNorm statistics in the producing conv's epilogue


_STATS_ARMED = [False]


def conv_stats_armed():
    """True between conv_stats_begin and conv_stats_end: the next conv launch must be one that carries the statistics epilogue"""
    return _STATS_ARMED[0]


def conv_stats_begin(dev, fmt=0):
    """Arms the next conv launch of this thread (in the library of rows format `fmt`) to also write the BatchNorm statistics
    partials of its output (kg_conv_stats_begin)."""
    part = scratch_f32(1 << 21, dev, "bnpart")
    _lib.call("kg_conv_stats_begin", ptr(part), c_long(part.numel()), fmt=fmt)
    _STATS_ARMED[0] = True
    return part


def conv_bstats_begin(x, mean, invstd, M, C, N=1):
    """Arms the next INPUT-GRADIENT conv launch of this thread to also write the BatchNorm-backward partials {sum g, sum g * xhat} of the
    gradient rows it stores (kg_conv_bstats_begin): x = the BatchNorm's input rows (PT), mean / invstd its batch statistics.  Returns the
    partials tensor (its own allocation: several armed gradients may be pending at once -- downsample branch -- unlike the forward's shared buffer)."""
    xb = _rows(x)
    P, ps = nplanes(x)
    # tiles of the kernel that will claim the buffer: 64-pixel tiles of the gather kernel's K split (4 per 256-pixel workgroup, whatever M), 128- or
    # 256-pixel tiles, or the halo kernel's 16 x 32 tiles per image
    tiles = 4 * math.ceil(M / 256) + 4 * N + 8
    part = torch.empty(tiles * C * 2, dtype=torch.float32, device=xb.device)
    _lib.call("kg_conv_bstats_begin", ptr(part), c_long(part.numel()), ptr(xb), ld(x), P, ps, ptr(mean), ptr(invstd), fmt=fmt_of(x))
    _STATS_ARMED[0] = True
    return part


def conv_stats_end(fmt=0):
    """Pixel tiles the armed conv wrote partials for (0: its kernel has no statistics epilogue); disarms."""
    import ctypes
    nb = ctypes.c_int(0)
    _STATS_ARMED[0] = False
    _lib.call("kg_conv_stats_end", ctypes.byref(nb), fmt=fmt)
    return nb.value


def bn_finalize_train(part, nb, M, C, gamma, beta, rmean, rvar, momentum=0.1, eps=1e-5):
    """(mean, invstd, scale, shift) from the partials a conv epilogue wrote; updates running stats in place like bn_stats_train."""
    st = torch.empty(4, C, dtype=torch.float32, device=part.device)
    _lib.call("kg_bn_finalize_train", ptr(part), nb, M, C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), c_float(momentum), c_float(eps),
              ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), stream_ptr())
    return st[0], st[1], st[2], st[3]


def bn_scale_shift_eval(C, gamma, beta, rmean, rvar, eps=1e-5):
    st = torch.empty(2, C, dtype=torch.float32, device=gamma.device)
    _lib.call("kg_bn_scale_shift_eval", C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), c_float(eps), ptr(st[0]), ptr(st[1]), stream_ptr())
    return st[0], st[1]


def bn_apply(x, C, scale, shift, y, res=None, relu=False):
    _lib.call("kg_bn_apply", ptr(_rows(x)), ld(x), ptr(scale), ptr(shift), ptr(base(res)), ld(res) if res is not None else 0, ptr(_rows(y)), ld(y),
              base(x).shape[0], C, 1 if relu else 0, pl(a=x, b=res, y=y), stream_ptr(), fmt=fmt_of(x))


def bn_bwd(x, dy, C, gamma, mean, invstd, dgamma, dbeta, dx, accumulate=False, parts=None, parts_scale=None):
    """parts = (partials tensor, tile count) written by the input gradient that produced dy (conv_bstats_begin): no column reduction over x and dy;
    parts_scale: device scalar dy was re-normalised by after the partials were taken (rows_rescale's r)"""
    sc = scratch_f32(2 * C * 512 + 3 * C, base(x).device, "bn")
    pt, nbp = parts if parts is not None else (None, 0)
    _lib.call("kg_bn_bwd", ptr(_rows(x)), ld(x), ptr(_rows(dy)), ld(dy), ptr(gamma), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta),
              1 if accumulate else 0, ptr(_rows(dx)), ld(dx), base(x).shape[0], C, ptr(sc), sc.numel(), ptr(pt), nbp, ptr(parts_scale),
              pl(a=x, b=dy, y=dx), stream_ptr(), fmt=fmt_of(x))


def maxpool_fwd(x, y, N, H, W, C, argmax=None):
    """argmax (optional uint8 [N*OH*OW, C]): receives the winning tap of every output element for maxpool_bwd."""
    _lib.call("kg_maxpool3s2_fwd", ptr(_rows(x)), ld(x), ptr(_rows(y)), ld(y), ptr(argmax), N, H, W, C, pl(a=x, y=y), stream_ptr(), fmt=fmt_of(x))


def maxpool_bwd(x, dy, dx, N, H, W, C, argmax=None):
    _lib.call("kg_maxpool3s2_bwd", ptr(_rows(x)), ld(x), ptr(_rows(dy)), ld(dy), ptr(_rows(dx)), ld(dx), ptr(argmax), N, H, W, C,
              pl(a=x, b=dy, y=dx), stream_ptr(), fmt=fmt_of(dy))


def bilinear_fwd(x, y, N, IH, IW, OH, OW, C, boxdesc=None, row2box=None):
    rows = base(y).shape[0] if boxdesc is not None else 0
    _lib.call("kg_bilinear_fwd", ptr(_rows(x)), ld(x), ptr(_rows(y)), ld(y), N, IH, IW, OH, OW, C, ptr(boxdesc), ptr(row2box),
              c_long(rows), pl(a=x, y=y), stream_ptr(), fmt=fmt_of(x))


def bilinear_bwd(dy, dx, N, IH, IW, OH, OW, C, boxdesc=None, row2box=None, mask=None):
    rows = base(dx).shape[0] if boxdesc is not None else 0
    _lib.call("kg_bilinear_bwd", ptr(_rows(dy)), ld(dy), ptr(_rows(dx)), ld(dx), N, IH, IW, OH, OW, C, ptr(boxdesc), ptr(row2box),
              c_long(rows), ptr(base(mask)), ld(mask) if mask is not None else 0, pl(a=dy, y=dx), stream_ptr(), fmt=fmt_of(dy))


def add_rows(a, b, y, C, mask=None, scale=None):
    """y = (a + b) [* scale[0] * scale[1]] [masked by mask > 0]; b may be None.  scale: pair of device scalars (the second may be None)."""
    r, r2 = scale if scale is not None else (None, None)
    _lib.call("kg_add_rows", ptr(_rows(a)), ld(a), ptr(base(b)), ld(b) if b is not None else 0, ptr(base(mask)),
              ld(mask) if mask is not None else 0, ptr(_rows(y)), ld(y), c_long(base(a).shape[0]), C, ptr(r), ptr(r2), pl(a=a, b=b, y=y), stream_ptr(),
              fmt=fmt_of(a))


def planes_to_f32(x, C, out):
    """out [rows, C] fp32 (row stride out.stride(0)) = sum of the planes of x."""
    _lib.call("kg_planes_to_f32", ptr(_rows(x)), ld(x), ptr(out), out.stride(0), c_long(base(x).shape[0]), C, pl(a=x), stream_ptr(), fmt=fmt_of(x))


def f32_to_planes(acc, y, C, addto=None, scale=None):
    """y (planes) = acc [rows, C] fp32 (* the device scalar `scale`) (+ addto planes)."""
    _lib.call("kg_f32_to_planes", ptr(acc), acc.stride(0), ptr(_rows(y)), ld(y), ptr(base(addto)), ld(addto) if addto is not None else 0,
              c_long(acc.shape[0]), C, pl(b=addto, y=y, scale=scale), stream_ptr(), fmt=fmt_of(y))


def sigmoid_(x):
    _lib.call("kg_sigmoid_inplace", ptr(x), c_long(x.numel()), stream_ptr())
    return x


def grad_pack(g, prob, out, N, C, H, W, cpad, scale=None):
    _lib.call("kg_grad_pack", ptr(g), ptr(prob), ptr(_rows(out)), N, C, H, W, ld(out), cpad, pl(y=out, scale=scale), stream_ptr(), fmt=fmt_of(out))


def grad_pack3(gs, prob0, out, N, Cs, H, W, pads, scale=None):
    """the three fp32 NCHW map gradients of a level into the column blocks pads[0] | pads[1] | pads[2] of out (kg_grad_pack3: one launch)"""
    _lib.call("kg_grad_pack3", ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(prob0), ptr(_rows(out)), N, Cs[0], Cs[1], Cs[2], H, W, ld(out),
              pads[0], pads[1], pads[2], pl(y=out, scale=scale), stream_ptr(), fmt=fmt_of(out))


# ---- gradient scale of the half-precision backward pass (csrc/gradscale.hip) ------------------------------------------------------
# Magnitude target of the gradient scale: ops.grad_scale places the largest loss gradient of a step in [8, 16); the re-normalisation
# points of the backward pass (engine.renormalise) bring a gradient tensor that has grown beyond it back there -- only ever DOWN: a
# factor > 1 would also multiply the other live gradient tensors and could push one of them out of range (measured: with up-scaling 4 of 14
# random-init seeds overflowed).  IEEE half then has 2^12 of headroom up to 65504 between two re-normalisation points -- one BatchNorm
# layer multiplies the gradient of a (nearly) dead channel by 1 / sqrt(eps) = 316 -- and normal numbers down to 2^-18 of the target.
# Measured: targets 2^2 .. 2^8 give the same gradient cosines (0.999987) and norms on the calibrated fixture and no overflow on 24
# random-init seeds at 64^2 (tiny BatchNorm populations: the worst case); tools/dbg_f16.py, tools/gradmax_probe.py.
GRAD_TARGET_LOG2 = int(__import__("os").environ.get("KG_GRAD_TARGET_LOG2", "4"))
_gs_state = {}


def grad_scale(tensors, probs=None):
    """Device pair {S, 1 / S} (fp32 [2]) for this backward pass: S = the power of two that brings max |t| over the given fp32
    tensors (the gradients of the loss w.r.t. the network outputs) into [2^(T-1), 2^T), T = GRAD_TARGET_LOG2.  probs[i] (optional):
    the sigmoid output tensors[i] refers to -- its values count as t * q * (1 - q), what grad_pack hands to the network.  No host sync."""
    import ctypes
    probs = probs or [None] * len(tensors)
    pairs = [(t, q) for t, q in zip(tensors, probs) if t is not None and t.numel() > 0]
    ts = [t for t, _ in pairs]
    qs = [q for _, q in pairs]
    assert 1 <= len(ts) <= 24 and all(t.dtype == torch.float32 and t.is_contiguous() for t in ts)
    assert all(q is None or (q.dtype == torch.float32 and q.is_contiguous() and q.numel() == t.numel()) for t, q in pairs)
    dev = ts[0].device
    scr = _gs_state.get(str(dev))
    if scr is None:
        scr = torch.zeros(2, dtype=torch.int32, device=dev)
        _gs_state[str(dev)] = scr
    out = torch.empty(2, dtype=torch.float32, device=dev)
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    prbs = (ctypes.c_void_p * len(ts))(*[q.data_ptr() if q is not None else None for q in qs])
    cnts = (ctypes.c_long * len(ts))(*[t.numel() for t in ts])
    _lib.call("kg_grad_scale", ptrs, prbs, cnts, len(ts), GRAD_TARGET_LOG2, ptr(scr), ptr(out), stream_ptr())
    return out


def scale_tensors(tensors, scale, flag=None):
    """every fp32 tensor *= a device scalar, one launch (kg_scale_tensors).  scale: one device scalar tensor for all, or a list with
    one per tensor (parameters of different backbone stages carry different cumulative scales, rows_rescale).  flag (optional device
    int32[1]): set to 1 when a result is inf / NaN."""
    import numpy as np
    flush_wgrad()       # (the tensors may be weight gradients whose split reductions are still recorded only)
    scales = scale if isinstance(scale, (list, tuple)) else [scale] * len(tensors)
    pairs = [(t, sc) for t, sc in zip(tensors, scales) if t is not None and t.numel() > 0]
    if not pairs:
        return
    dt = np.dtype([("p", "<u8"), ("n", "<i8"), ("scale", "<u8"), ("blk0", "<i4"), ("pad", "<i4")])
    arr = np.zeros(len(pairs), dt)
    blk = 0
    for i, (t, sc) in enumerate(pairs):
        assert t.dtype == torch.float32 and t.is_contiguous() and sc.dtype == torch.float32
        arr[i] = (t.data_ptr(), t.numel(), sc.data_ptr(), blk, 0)
        blk += (t.numel() + 4095) // 4096
    tab = h2d(arr.view(np.uint8).reshape(-1), pairs[0][0].device)
    _lib.call("kg_scale_tensors", ptr(tab), len(pairs), blk, ptr(flag), stream_ptr())


def rows_rescale(g, C, cum_in, target_log2=None):
    """Stage boundary of the half-precision backward pass (csrc/norm_pool.hip): g (rows / PT, in place) *= r, the power of two that
    brings max |g| into [2^(T-1), 2^T).  cum_in: device {scale, 1 / scale} g is expressed in.  Returns (r [1], cum_out [2]) device tensors."""
    dev = base(g).device
    scr = _gs_state.get(str(dev))
    if scr is None:
        scr = torch.zeros(2, dtype=torch.int32, device=dev)
        _gs_state[str(dev)] = scr
    out = torch.empty(3, dtype=torch.float32, device=dev)
    _lib.call("kg_rows_rescale", ptr(_rows(g)), ld(g), c_long(base(g).shape[0]), C, GRAD_TARGET_LOG2 if target_log2 is None else target_log2,
              ptr(cum_in), ptr(out[0:2]), ptr(out[2:3]), ptr(scr), pl(a=g), stream_ptr(), fmt=fmt_of(g))
    return out[2:3], out[0:2]


def rows_scale_multi(items, r, r2=None):
    """every (rows / PT, C) of `items` *= the device scalar r (* r2), eight tensors per launch (kg_rows_scale_multi)"""
    import ctypes
    items = [(g, C) for g, C in items if base(g).shape[0] > 0]
    for i in range(0, len(items), 8):
        chunk = items[i:i + 8]
        desc = (ctypes.c_long * (6 * len(chunk)))()
        for k, (g, C) in enumerate(chunk):
            P, ps = nplanes(g)
            desc[6 * k:6 * k + 6] = [_rows(g).data_ptr(), ld(g), base(g).shape[0], C, P, ps]
        _lib.call("kg_rows_scale_multi", desc, len(chunk), ptr(r), ptr(r2), stream_ptr(), fmt=fmt_of(chunk[0][0]))


def rows_scale(g, C, r, r2=None):
    """g (rows / PT) *= the device scalar r (* r2) (powers of two), in place, every plane"""
    _lib.call("kg_rows_scale", ptr(_rows(g)), ld(g), c_long(base(g).shape[0]), C, ptr(r), ptr(r2), pl(a=g), stream_ptr(), fmt=fmt_of(g))

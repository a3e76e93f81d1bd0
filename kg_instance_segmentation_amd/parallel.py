"""Data-parallel training of KGnet: one process per GPU, RCCL over xGMI.

The reference only *defines* nn.DataParallel (train.py:39-40) and never calls it.  Here the minibatch
is sharded over ranks; the only exchange step per optimizer step is the gradient SUM all-reduce
(73.9 M fp32 = 296 MB, bucketed so that RCCL moves few large messages over the point-to-point xGMI
links), plus one 12-float all-reduce of the loss normalisers so that the result equals the
single-device loss over the global batch:
  * BCE term: mean over the global N*5*H*W            (loss.py:13)
  * masked-L1 terms: divided by the GLOBAL mask sums  (loss.py:25,37)
  * seg loss: divided by the GLOBAL batch size        (seg_loss.py:94)
Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over gloo.
"""
import torch
import torch.distributed as dist

from .arch import EDGES


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "KG_FORCE_DEVICE" in os.environ:      # test hook: several ranks on one GPU (with KG_DIST_BACKEND=gloo)
        local = int(os.environ["KG_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("KG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def detection_denominators(gt_levels):
    """Global normalisers of DetectionLossAll for each scale: tensor [L,3] = (mask2_sum, mask4_sum, kp_numel)
    summed over all ranks.  gt_levels: list of [N,55,H,W] tensors (any device)."""
    frm = [e[0] for e in EDGES] + [e[1] for e in EDGES]
    rows = []
    for gt in gt_levels:
        kp = gt[:, :5].float()
        per = kp.sum(dim=(0, 2, 3))                      # [5]
        m2 = 2.0 * per.sum()
        m4 = 2.0 * per[torch.tensor(frm, device=gt.device)].sum()
        rows.append(torch.stack([m2, m4, torch.tensor(float(kp.numel()), device=gt.device)]))
    den = torch.stack(rows).float()
    if world_size() > 1:
        dist.all_reduce(den, op=dist.ReduceOp.SUM)
    return den


class GradReducer:
    """Bucketed gradient SUM all-reduce over RCCL.

    attach(model): overlapped mode -- the decoder/heads backward (one explicit tape, engine.backward_dec) hands over
    parameter gradients as soon as their kernels are enqueued, in backward order: the 7x7 head weights (64 M of the
    73.9 M parameters) are complete after the first quarter of the backward pass, so their all-reduce runs on RCCL's
    stream underneath the remaining backward kernels.  Buckets are large (few messages over the point-to-point xGMI
    links).  reduce() after loss.backward() handles whatever was not covered (the seg-branch parameters, whose backward
    may not run on a rank without valid boxes; a missing gradient counts as zero), or everything when not attached."""

    def __init__(self, params, bucket_mb=64):
        self.params = [p for p in params if p.requires_grad]
        self.cap = bucket_mb << 20
        self.pending, self.pending_bytes, self.inflight = [], 0, []
        self.covered = set()
        self.by_name = None

    # ---- overlapped mode -------------------------------------------------------------------------
    def attach(self, model):
        self.by_name = dict(model.named_parameters())
        model._engine.grad_hook = self._on_grads
        return self

    def _on_grads(self, items, last):
        if world_size() == 1:
            return
        for name, g in items:
            if name not in self.by_name or g is None:
                continue
            self.pending.append(g); self.pending_bytes += g.numel() * 4
            self.covered.add(name)
            if self.pending_bytes >= self.cap:
                self._launch()
        if last:
            self._launch()
            self._drain()

    def _launch(self):
        if not self.pending:
            return
        flat = torch.cat([g.reshape(-1) for g in self.pending])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self.inflight.append((work, flat, self.pending))
        self.pending, self.pending_bytes = [], 0

    def _drain(self):
        for work, flat, gs in self.inflight:
            work.wait()
            off = 0
            for g in gs:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
        self.inflight = []

    # ---- after backward --------------------------------------------------------------------------
    def reduce(self):
        """All-reduces .grad (SUM over ranks, in place) of every parameter the overlapped mode did not cover this step."""
        if world_size() == 1:
            return
        if self.by_name is not None:
            todo = [p for n, p in self.by_name.items() if p.requires_grad and n not in self.covered]
        else:
            todo = self.params
        self.covered = set()
        buckets, cur, size = [], [], 0
        for p in reversed(todo):
            cur.append(p); size += p.numel() * 4
            if size >= self.cap:
                buckets.append(cur); cur, size = [], 0
        if cur:
            buckets.append(cur)
        works = []
        for b in buckets:
            gs = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b]
            flat = torch.cat([g.reshape(-1) for g in gs])
            works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, b))
        for w, flat, b in works:
            w.wait()
            off = 0
            for p in b:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n


class FlatGradReducer:
    """Data-parallel gradient exchange over ONE persistent flat fp32 buffer.

    attach(model) lays all 217 parameter gradients out in one buffer, in the order the backward pass completes them (seg branch,
    then heads c3..c0, decoder, layer3..1, stem), cut into buckets of ~bucket_mb.  The weight-gradient reduction kernels, the
    BatchNorm / bias gradient kernels write straight into a parameter's slot (engine.new_grad), the slot itself becomes
    `param.grad` (no autograd accumulation copy), every bucket is SUM-all-reduced IN PLACE on RCCL's stream as soon as its last
    gradient kernel is enqueued -- underneath the remaining backward kernels -- and the fused Adam (optim.Adam, kg_adam_step)
    reads the same memory: no torch.cat, no copy-back, no per-step allocation.  An in-place ring all-reduce IS RCCL's
    reduce-scatter + all-gather pair over the xGMI links; issuing the two halves separately only pays with an optimizer sharded
    over ranks, and Adam is 0.6 ms of a 57 ms step here.
    The seg-branch bucket is zeroed at the start of a step (a rank whose images have no valid box runs no seg backward and must
    contribute zeros) and reduced when the dense backward starts (the same point of the collective sequence on every rank).
    finish() (after loss.backward()) waits for the outstanding reductions; `grad_scale` (e.g. 1 / world for a mean) is not applied:
    the losses are normalised globally instead (detection_denominators)."""

    def __init__(self, bucket_mb=64):
        self.cap = bucket_mb << 20
        self.model = None

    def attach(self, model):
        self.model = model
        eng = model._engine
        params = dict(model.named_parameters())
        seg_keys = [k for k in model._seg.param_keys if params[k].requires_grad]
        dec_keys = [k for k in reversed(model._param_keys) if k not in set(seg_keys) and params[k].requires_grad]
        # backward completes the heads first (they are last in the forward), the stem last; state_dict order is forward order
        # for the trunk but lists the decoder / heads after the seg branch: sort by the position of the LAST forward use
        order = self._backward_order(model, dec_keys)
        self.keys = seg_keys + order
        dev = params[self.keys[0]].device
        total = sum(params[k].numel() for k in self.keys)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slot, off = {}, 0
        self.buckets, cur, size = [], [], 0           # [(start, end, [keys])]; bucket 0 = the seg branch
        self.seg_bucket = (0, sum(params[k].numel() for k in seg_keys), list(seg_keys))
        for k in self.keys:
            n = params[k].numel()
            self.slot[k] = self.flat[off:off + n].view_as(params[k])
            off += n
        off = self.seg_bucket[1]
        start = off
        for k in order:
            cur.append(k); size += params[k].numel() * 4; off += params[k].numel()
            if size >= self.cap:
                self.buckets.append((start, off, cur)); cur, size, start = [], 0, off
        if cur:
            self.buckets.append((start, off, cur))
        self.bucket_of = {k: i for i, (_, _, ks) in enumerate(self.buckets) for k in ks}
        self.missing = [set(ks) for _, _, ks in self.buckets]
        self.inflight, self.seg_launched = [], False
        self.unscale, self.seg_unscaled = None, False      # half-precision backward (ops.grad_scale): device 1 / S of the running backward pass
        eng.grad_store = self
        eng.grad_hook = self._on_grads
        return self

    @staticmethod
    def _backward_order(model, keys):
        pos = {}
        for i, k in enumerate(model._param_keys):
            pos[k] = i
        def rank_of(k):      # forward stage of a parameter: c0_conv < stem/backbone < decoder (c4 -> c0) < heads (c0 -> c3)
            if "_head_c" in k:
                return (3, int(k.split("_head_c")[1][0]), pos[k])
            if "_up_conv" in k or "_cat_refine" in k:
                return (2, 4 - int(k[1]), pos[k])
            if k.startswith("c0_conv"):
                return (-1, 0, pos[k])     # first in the forward tape: its gradient kernels are enqueued last
            return (0, 0, pos[k])
        return sorted(keys, key=rank_of, reverse=True)

    # ---- engine.grad_store protocol ------------------------------------------------------------------
    def get(self, key):
        return self.slot.get(key)

    def owns(self, key, g):
        v = self.slot.get(key)
        return v is not None and g.data_ptr() == v.data_ptr()

    def deliver(self, key, param):
        """The slot IS the gradient: install it as param.grad (adding to a gradient accumulated earlier) and give autograd nothing."""
        v = self.slot[key]
        if param.grad is None or param.grad.data_ptr() == v.data_ptr():
            param.grad = v
        else:
            param.grad.add_(v)
        return None

    def begin_step(self):
        """Call before the forward of every step: re-arms the buckets and zeroes the seg-branch slots."""
        self.missing = [set(ks) for _, _, ks in self.buckets]
        self.seg_launched, self.seg_unscaled = False, False
        self._pend_done = set()
        a, b, _ = self.seg_bucket
        if b > a:
            self.flat[a:b].zero_()

    def seg_done(self, unscale=None):
        """(the seg branch's backward has enqueued all its gradient kernels on this rank).  unscale: forward_seg ran as its own
        autograd node in the half-precision build -- its gradients carry ITS power-of-two scale, divided out here."""
        a, b, _ = self.seg_bucket
        if unscale is not None and b > a and not self.seg_unscaled:
            from . import ops
            ops.scale_tensors([self.flat[a:b]], unscale)
            self.seg_unscaled = True

    def dense_backward_started(self):
        """Start of forward_dec's backward, which EVERY rank runs and which autograd schedules after the seg branch's backward
        (forward_seg consumes forward_dec's outputs): the one point where all ranks can issue the seg bucket's all-reduce in
        the same order -- a rank whose images had no valid box contributes the zeros of begin_step().  The reduction then runs
        underneath the whole dense backward."""
        if not self.seg_launched and self.seg_bucket[1] > self.seg_bucket[0]:
            self._launch(self.seg_bucket[0], self.seg_bucket[1], unscale=not self.seg_unscaled)
        self.seg_launched = True

    def _on_grads(self, items, last):
        for name, g in items:
            i = self.bucket_of.get(name)
            if i is None:
                continue
            self.missing[i].discard(name)
            if not self.missing[i]:
                self.missing[i] = {None}          # launched
                self._launch(self.buckets[i][0], self.buckets[i][1])

    def unscale_pending(self):
        """end of a half-precision backward pass: gradients this rank produced in buckets that did not complete (some parameter of
        the bucket got no gradient this step) still carry the pass's scale -- divide it out slot by slot (finish() reduces such
        buckets as they stand)"""
        if self.unscale is None:
            return
        from . import ops
        todo = []
        for i, (_, _, ks) in enumerate(self.buckets):
            if self.missing[i] != {None}:
                for k in ks:
                    if k not in self.missing[i] and k not in self._pend_done:
                        todo.append(self.slot[k])
                        self._pend_done.add(k)
        if todo:
            ops.scale_tensors(todo, self.unscale)

    def _launch(self, a, b, unscale=True):
        """a bucket is complete on this rank: divide out the backward pass's power-of-two scale (half-precision build; every rank has
        its own), then SUM-all-reduce it in place"""
        if unscale and self.unscale is not None:
            from . import ops
            ops.scale_tensors([self.flat[a:b]], self.unscale)
        if world_size() > 1:
            self.inflight.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """After loss.backward(): reduces whatever was not produced on this rank this step (as zeros / stale-free: a bucket whose
        gradients were not all produced is reduced as it stands after zero-filling the missing slots) and waits for all reductions."""
        if world_size() > 1:
            if not self.seg_launched:
                self.dense_backward_started()
            for i, (a, b, ks) in enumerate(self.buckets):
                if self.missing[i] != {None}:
                    for k in self.missing[i]:
                        self.slot[k].zero_()
                    self._launch(a, b, unscale=False)      # (no backward pass is running: whatever this bucket holds is unscaled or zero)
                    self.missing[i] = {None}
            for w in self.inflight:
                w.wait()
        self.inflight = []
        # parameters whose gradient never reached autograd (zero contribution on this rank) still need .grad for the optimizer
        params = dict(self.model.named_parameters())
        for k, v in self.slot.items():
            if params[k].grad is None:
                params[k].grad = v

    reduce = finish


def broadcast_parameters(module, src=0):
    """Makes every replica start from rank `src`'s parameters and buffers."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)

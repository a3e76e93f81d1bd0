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
    the losses are normalised globally instead (detection_denominators).

    Contract (asserted): ONE backward pass per forward and `optimizer.zero_grad(set_to_none=True)` (PyTorch's default) before every
    forward -- the slot IS `param.grad` and the gradient kernels overwrite it, so gradient accumulation over several backward passes
    cannot be expressed (begin_step raises if a parameter still carries a gradient; deliver raises on a foreign `.grad` tensor).
    finish() also exchanges ONE small MAX all-reduce (len(keys) + 1 int32): the bitmap of the parameters this rank's backward pass
    produced a gradient for, and the sticky non-finite flag of the half-precision backward (KGnet.grad_overflowed).  A parameter NO
    rank produced a gradient for (a skip_combine level no box of the global batch reaches) keeps `.grad` None on every rank -- the
    optimizer skips it exactly as the single-device step does -- and the overflow flag is the same on every rank after finish(), so
    `if model.grad_overflowed(): skip the step` cannot make replicas diverge (the SUM all-reduce spreads one rank's inf / NaN to all)."""

    def __init__(self, bucket_mb=64, tail_mb=8, tail_total_mb=32):
        self.cap = bucket_mb << 20
        # the LAST bucket's all-reduce starts when the backward pass ends and is exposed in full: the final `tail_total_mb` of the
        # buffer (the 115 small backbone / c0_conv tensors, ~24 MB) go out in pieces of <= tail_mb, so what is left after the last
        # gradient kernel is one small piece (round 4's single 24 MB tail bucket was issued 0.04 ms before finish())
        self.tail_cap, self.tail_total = min(tail_mb << 20, self.cap), tail_total_mb << 20
        self.model = None

    def attach(self, model):
        self.model = model
        eng = model._engine
        params = dict(model.named_parameters())
        self._params = {k: p for k, p in params.items() if p.requires_grad}
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
            left = (total - off) * 4                  # bytes of the gradients still to come after this one
            if size >= (self.cap if left + size > self.tail_total else self.tail_cap):
                self.buckets.append((start, off, cur)); cur, size, start = [], 0, off
        if cur:
            self.buckets.append((start, off, cur))
        self.bucket_of = {k: i for i, (_, _, ks) in enumerate(self.buckets) for k in ks}
        self.missing = [set(ks) for _, _, ks in self.buckets]
        self.inflight, self.seg_launched = [], False
        self.key_index = {k: i for i, k in enumerate(self.keys)}
        self.produced = set()                 # keys whose gradient this rank's running step produced (deliver)
        self.unscale_of, self.seg_unscaled = None, False   # half-precision backward: key -> device {scale, 1 / scale} of the gradients the running backward pass produced (engine.param_gsc)
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
        if param.grad is not None and param.grad.data_ptr() != v.data_ptr():
            raise RuntimeError(f"FlatGradReducer: {key}.grad is a tensor this reducer does not own (zero_grad(set_to_none=False) or a gradient "
                               "accumulated before attach()): call optimizer.zero_grad(set_to_none=True) before every forward")
        param.grad = v
        self.produced.add(key)
        return None

    def begin_step(self):
        """Call before the forward of every step: re-arms the buckets and zeroes the seg-branch slots."""
        stale = [k for k, p in self._params.items() if p.grad is not None]
        if stale:
            raise RuntimeError(f"FlatGradReducer: {len(stale)} parameters (e.g. {stale[0]}) still carry a gradient at the start of a forward: the "
                               "gradient kernels overwrite the flat slots, so accumulation over several backward passes is not supported -- "
                               "call optimizer.zero_grad(set_to_none=True) before every forward")
        self.missing = [set(ks) for _, _, ks in self.buckets]
        self.seg_launched, self.seg_unscaled = False, False
        self._pend_done = set()
        self.produced = set()
        a, b, _ = self.seg_bucket
        if b > a:
            self.flat[a:b].zero_()

    def seg_done(self, unscale_of=None):
        """(the seg branch's backward has enqueued all its gradient kernels on this rank).  unscale_of: forward_seg ran as its own
        autograd node in the half-precision build -- its gradients carry ITS power-of-two scales (key -> device {scale, 1 / scale}),
        divided out here."""
        a, b, ks = self.seg_bucket
        if unscale_of is not None and b > a and not self.seg_unscaled:
            saved, self.unscale_of = self.unscale_of, unscale_of
            self._unscale(ks)
            self.unscale_of = saved
            self.seg_unscaled = True

    def dense_backward_started(self):
        """Start of forward_dec's backward, which EVERY rank runs and which autograd schedules after the seg branch's backward
        (forward_seg consumes forward_dec's outputs): the one point where all ranks can issue the seg bucket's all-reduce in
        the same order -- a rank whose images had no valid box contributes the zeros of begin_step().  The reduction then runs
        underneath the whole dense backward."""
        if not self.seg_launched and self.seg_bucket[1] > self.seg_bucket[0]:
            self._launch(self.seg_bucket[0], self.seg_bucket[1], self.seg_bucket[2], unscale=not self.seg_unscaled)
        self.seg_launched = True

    def _on_grads(self, items, last):
        for name, g in items:
            i = self.bucket_of.get(name)
            if i is None:
                continue
            self.missing[i].discard(name)
            if not self.missing[i]:
                self.missing[i] = {None}          # launched
                self._launch(*self.buckets[i])

    def _unscale(self, keys):
        """divides the scale out of the slots of `keys` that the running backward pass produced (half-precision build: every rank has its own
        scales, so this happens BEFORE the all-reduce)"""
        if not self.unscale_of:
            return
        from . import ops
        ks = [k for k in keys if k in self.unscale_of and k not in self._pend_done]
        if ks:
            ops.scale_tensors([self.slot[k] for k in ks], [self.unscale_of[k][1:2] for k in ks], flag=self.model._engine.flag_on(self.flat.device))
            self._pend_done.update(ks)

    def unscale_pending(self):
        """end of a half-precision backward pass: gradients this rank produced in buckets that did not complete (some parameter of
        the bucket got no gradient this step) still carry the pass's scale -- divide it out slot by slot (finish() reduces such
        buckets as they stand)"""
        for i, (_, _, ks) in enumerate(self.buckets):
            if self.missing[i] != {None}:
                self._unscale([k for k in ks if k not in self.missing[i]])

    def _launch(self, a, b, keys, unscale=True):
        """a bucket is complete on this rank: divide out the backward pass's power-of-two scales (half-precision build), then
        SUM-all-reduce it in place"""
        from . import ops
        ops.flush_wgrad()                 # the bucket's gradients may still be recorded split reductions (ops.WgradReduceQueue)
        if unscale:
            self._unscale(keys)
        if world_size() > 1:
            self.inflight.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """After loss.backward(): reduces whatever was not produced on this rank this step (a bucket whose gradients were not all
        produced is reduced as it stands after zero-filling the missing slots), exchanges the produced-bitmap + non-finite flag (one
        MAX all-reduce of len(keys) + 1 int32) and waits for all reductions.  Afterwards `.grad` of every parameter SOME rank
        produced a gradient for is its reduced flat slot, `.grad` of the others stays None, and KGnet.grad_overflowed() answers
        the same on every rank.  (Reads the bitmap back: one host synchronisation at the end of the backward pass.)"""
        params = self._params
        eng = self.model._engine
        multi = world_size() > 1
        if multi:
            if not self.seg_launched:
                self.dense_backward_started()
            for i, (a, b, ks) in enumerate(self.buckets):
                if self.missing[i] != {None}:
                    for k in self.missing[i]:
                        self.slot[k].zero_()
                    self._launch(a, b, ks, unscale=False)      # (no backward pass is running: whatever this bucket holds is unscaled or zero)
                    self.missing[i] = {None}
            import numpy as np
            from . import ops
            host = np.zeros(len(self.keys) + 1, np.int32)
            host[[self.key_index[k] for k in self.produced]] = 1
            vec = ops.h2d(host, self.flat.device) if self.flat.is_cuda else torch.from_numpy(host)
            flag = getattr(eng, "overflow_flag", None)
            if flag is not None:
                vec[-1:].copy_(flag)
            self.inflight.append(dist.all_reduce(vec, op=dist.ReduceOp.MAX, async_op=True))
            for w in self.inflight:
                w.wait()
            if flag is not None:
                flag.copy_(vec[-1:])                         # every rank holds the global flag
            got = vec[:-1].cpu().numpy()
            anyrank = {k for k, i in self.key_index.items() if got[i]}
        else:
            anyrank = set(self.produced)
        self.inflight = []
        # a parameter another rank produced a gradient for (zero contribution here) still needs .grad for the optimizer; one NO
        # rank produced a gradient for keeps .grad None, as in the single-device step
        for k in anyrank:
            if params[k].grad is None:
                params[k].grad = self.slot[k]

    reduce = finish


def broadcast_parameters(module, src=0):
    """Makes every replica start from rank `src`'s parameters and buffers."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)

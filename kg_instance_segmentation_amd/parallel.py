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


def broadcast_parameters(module, src=0):
    """Makes every replica start from rank `src`'s parameters and buffers."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)

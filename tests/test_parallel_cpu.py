"""CPU (gloo, world size 2): the data-parallel helpers reproduce the single-process global-batch step:
global loss normalisers and the flat in-place gradient SUM all-reduce (kg_instance_segmentation_amd/parallel.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import net as onet
from oracle import synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from kg_instance_segmentation_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    # a tiny "model": two parameters; per-rank shard of a global batch of 4 GT maps / predictions
    N, H, W = 4, 16, 24
    gt = torch.from_numpy(np.stack([synth.gt_maps(synth.random_boxes(H, W, 2, 70 + i, 6, 10), H, W) for i in range(N)]))
    g = torch.Generator().manual_seed(1)
    base = [torch.rand(N, 5, H, W, generator=g) * 0.9 + 0.05, torch.randn(N, 10, H, W, generator=g), torch.randn(N, 40, H, W, generator=g)]
    a = torch.nn.Parameter(torch.tensor(0.7)); b = torch.nn.Parameter(torch.ones(40) * 0.3)
    if rank == 1:   # replicas start different: broadcast must fix it
        a.data.fill_(5.0)
    model = torch.nn.ParameterList([a, b])
    parallel.broadcast_parameters(model)
    assert float(a) == pytest.approx(0.7)
    sl = slice(rank * 2, rank * 2 + 2)

    def pred(s):
        return [base[0][s] * a.clamp(0.1, 1.0), base[1][s] * a, base[2][s] * b.view(1, 40, 1, 1)]

    # reference: single process, global batch
    lg = onet.detection_loss(pred(slice(0, N)), gt)
    ga, gb = torch.autograd.grad(lg, [a, b])
    # data parallel: local loss with global normalisers, then gradient SUM all-reduce
    den = parallel.detection_denominators([gt[sl]])[0]
    p = pred(sl)
    gt_l = gt[sl]
    gk = gt_l[:, :5]
    bce = torch.nn.functional.binary_cross_entropy(p[0], gk, reduction="sum") / den[2]
    m2 = gk.repeat_interleave(2, 1)
    frm = [e[0] for e in onet.EDGES] + [e[1] for e in onet.EDGES]
    m4 = gk[:, frm].repeat_interleave(2, 1)
    l_local = bce + (torch.abs(p[1] - gt_l[:, 5:15]) / 5 * m2).sum() / (den[0] + 1e-10) \
        + 0.25 * (torch.abs(p[2] - gt_l[:, 15:]) / 5 * m4).sum() / (den[1] + 1e-10)
    for q in (a, b):
        q.grad = None
    l_local.backward()
    for q in (a, b):                       # gradient SUM over the ranks (what FlatGradReducer does bucket by bucket)
        dist.all_reduce(q.grad, op=dist.ReduceOp.SUM)
    tot = l_local.detach().clone(); dist.all_reduce(tot)
    out[rank] = (float(tot), float(lg), float((a.grad - ga).abs().max()), float((b.grad - gb).abs().max()), float(gb.abs().max()))
    dist.destroy_process_group()


def test_global_batch_equivalence_gloo_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        tot, lg, ea, eb, scale = out[r]
        assert abs(tot - lg) <= 1e-5 * abs(lg)
        assert ea <= 1e-5 and eb <= 1e-5 * max(scale, 1.0)


def _flat_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from kg_instance_segmentation_amd import parallel
    parallel.init_from_env(backend="gloo")

    class Eng:
        grad_hook = None
        grad_store = None

    class Seg:
        param_keys = ["seg_head.0.weight", "seg_head.0.bias", "skip_combine.3.up.0.weight"]

    class Node(torch.nn.Module):
        pass

    class Model(torch.nn.Module):
        """parameter names of the real network's stages (FlatGradReducer orders its buffer by them)"""
        def __init__(self):
            super().__init__()
            shapes = {"conv1.weight": (8, 3, 7, 7), "layer1.0.conv1.weight": (300000,), "c0_conv.0.weight": (64, 3, 3, 3), "c4_up_conv.0.weight": (200000,),
                      "c0_cat_refine.0.weight": (64, 128), "kp_head_c0.0.weight": (50000,), "mid_offset_head_c3.2.weight": (400000,),
                      "seg_head.0.weight": (64, 64, 3, 3), "seg_head.0.bias": (64,),
                      "skip_combine.3.up.0.weight": (512, 16, 3, 3)}       # (a pyramid level no box of the global batch reaches: produced by NO rank)
            self._param_keys = list(shapes)
            for k, shp in shapes.items():
                node = self
                parts = k.split(".")
                for q in parts[:-1]:
                    if q not in node._modules:
                        node.add_module(q, Node())
                    node = node._modules[q]
                node.register_parameter(parts[-1], torch.nn.Parameter(torch.zeros(shp)))
            self._engine, self._seg = Eng(), Seg()

        def get_tensor(self, k):
            return dict(self.named_parameters())[k]

    m = Model()
    red = parallel.FlatGradReducer(bucket_mb=1).attach(m)
    names = dict(m.named_parameters())
    assert red.keys[:3] == Seg.param_keys and red.keys[3] == "mid_offset_head_c3.2.weight" and red.keys[-1] == "c0_conv.0.weight"
    assert red.flat.numel() == sum(p.numel() for p in m.parameters()) and len(red.buckets) >= 2
    eng = m._engine
    eng.overflow_flag = torch.zeros(1, dtype=torch.int32)      # sticky non-finite flag of the half-precision backward (KGnet.grad_overflowed)
    produced_seg = Seg.param_keys[:2]
    ok = True
    for step in range(2):
        red.begin_step()
        if step == 1 and rank == 1:
            eng.overflow_flag.fill_(1)                       # one rank's backward produced a non-finite gradient
        base = red.flat.data_ptr()
        # seg backward runs on rank 0 only (rank 1 has no valid box): its slots must count as zeros there
        if rank == 0:
            for k in produced_seg:
                g = eng.grad_store.get(k); g.fill_(7.0 * (step + 1))
                assert red.owns(k, g) and red.deliver(k, names[k]) is None
            red.seg_done()
        red.dense_backward_started()                       # engine.backward_dec does this on every rank
        order = [k for k in red.keys if k not in Seg.param_keys]
        for i, k in enumerate(order):
            g = eng.grad_store.get(k)                     # the gradient kernel's destination IS the flat slot
            g.fill_(float((rank + 1) * (i + 1) * (step + 1)))
            assert red.deliver(k, names[k]) is None
            eng.grad_hook([(k, g)], False)
        eng.grad_hook([], True)
        red.finish()
        assert red.flat.data_ptr() == base
        for i, k in enumerate(order):
            ok = ok and names[k].grad.data_ptr() == red.get(k).data_ptr()            # .grad is the slot: nothing was copied
            ok = ok and float((names[k].grad - 3.0 * (i + 1) * (step + 1)).abs().max()) == 0.0
        for k in produced_seg:      # produced on rank 0 only: every rank holds the sum
            ok = ok and names[k].grad is not None and float((names[k].grad - 7.0 * (step + 1)).abs().max()) == 0.0
        ok = ok and names["skip_combine.3.up.0.weight"].grad is None       # produced by no rank: stays None everywhere (as in one process)
        ok = ok and int(eng.overflow_flag) == (1 if step == 1 else 0)      # the flag of rank 1 is every rank's after finish()
        for p in m.parameters():
            p.grad = None                                  # optimizer.zero_grad(set_to_none=True)
    # the contract is asserted: a forward that starts while a parameter still carries a gradient (no zero_grad: accumulation over
    # several backward passes) is refused, and so is a foreign .grad tensor at delivery
    names["conv1.weight"].grad = red.get("conv1.weight")
    try:
        red.begin_step(); ok = False
    except RuntimeError:
        pass
    names["conv1.weight"].grad = torch.zeros_like(names["conv1.weight"])
    try:
        red.deliver("conv1.weight", names["conv1.weight"]); ok = False
    except RuntimeError:
        pass
    out[rank] = ok
    dist.destroy_process_group()


def test_flat_grad_reducer_gloo_world2():
    """FlatGradReducer: one persistent flat gradient buffer in backward order; gradient "kernels" write into the slots, the slot is
    installed as .grad without a copy, buckets are all-reduced in place as they complete, a rank without seg backward contributes
    zeros, and two consecutive steps give the exact sums."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_flat_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out[r] for r in range(world))

"""CPU (gloo, world size 2): the data-parallel helpers reproduce the single-process global-batch step:
global loss normalisers and bucketed gradient SUM all-reduce (kg_instance_segmentation_amd/parallel.py)."""
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
    red = parallel.GradReducer([a, b], bucket_mb=1)
    red.reduce()
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


def test_reducer_buckets_cover_all_params():
    from kg_instance_segmentation_amd import parallel
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 1 << 19, 3, 1 << 18, 7)]
    red = parallel.GradReducer(ps, bucket_mb=1)
    flat = [p for b in red.buckets for p in b]
    assert len(flat) == len(ps) and {id(p) for p in flat} == {id(p) for p in ps}
    assert flat[0] is ps[-1]          # reverse (backward) order

"""GPU: the data-parallel step of the HIP path (BASELINE configs[3]) on TWO ranks.

A 1-GPU box cannot host two RCCL ranks (one communicator rank per device), so the two processes share GPU 0 and exchange over
gloo (KG_DIST_BACKEND=gloo, KG_FORCE_DEVICE=0 -- the hooks of parallel.init_from_env); everything else is the production path:
sharded minibatch, global loss normalisers (parallel.detection_denominators), FlatGradReducer (gradient kernels write into the
flat buffer, buckets all-reduced in place during backward), fused Adam reading the same buffer.
Asserted: the reduced gradients and the updated parameters of both ranks are BIT-IDENTICAL to one process running the two shards
one after the other with the same global normalisers and summing their gradients (BatchNorm statistics are per replica in both,
as in the reference's nn.DataParallel, train.py:39-40)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N, S, NB = 4, 128, 5


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _digest(t):
    t = t.detach().float().reshape(-1)
    idx = torch.arange(0, t.numel(), max(1, t.numel() // 257), device=t.device)
    return (float(t.double().sum()), float(t.double().abs().sum()), t[idx].cpu().numpy().tobytes())


def _shard_step(model, opt, ldec, lseg, batch, sl, den, world, reducer=None):
    x, gt_boxes, gt_masks, gt_lv = batch
    opt.zero_grad()
    d0, d1, d2, d3, pred = model(x[sl].cuda(), gt_boxes[sl])
    l1 = sum(ldec(p, g[sl].cuda(), denominators=den[i]) for i, (p, g) in enumerate(zip((d0, d1, d2, d3), gt_lv)))
    l2 = lseg(pred, gt_masks[sl], gt_boxes[sl])
    loss = l1 if l2 is None else l1 + l2 / world
    loss.backward()
    if reducer is not None:
        reducer.finish()
    return float(loss.detach())


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      KG_DIST_BACKEND="gloo", KG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from kg_instance_segmentation_amd import KGnet, parallel
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.optim import Adam
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    from oracle import synth, weightgen
    r, w, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    sd = weightgen.gen_state_dict(0, variant="cal")
    model = KGnet.resnet50(pretrained=False)
    model.load_state_dict(sd)
    model = model.cuda().train()
    if rank == 1:
        with torch.no_grad():
            model.get_tensor("conv1.weight").mul_(3.0)      # replicas start different: broadcast_parameters must fix it
    parallel.broadcast_parameters(model)
    opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    reducer = parallel.FlatGradReducer(bucket_mb=32).attach(model)
    batch = synth.train_batch(N, S, S, 21, n_boxes=NB)
    if rank == 1:
        batch[1][2] = np.zeros((0, 5), np.float32); batch[1][3] = np.zeros((0, 5), np.float32)    # rank 1: no boxes -> no seg backward
        batch[2][2] = np.zeros((0, S, S), np.float32); batch[2][3] = np.zeros((0, S, S), np.float32)
    sl = slice(rank * N // world, (rank + 1) * N // world)
    den = parallel.detection_denominators([g[sl].cuda() for g in batch[3]])
    loss = _shard_step(model, opt, DetectionLossAll(5), SEG_loss(S, S), batch, sl, den, world, reducer)
    grads = {k: _digest(p.grad) for k, p in model.named_parameters()}
    assert all(p.grad.data_ptr() == reducer.get(k).data_ptr() for k, p in model.named_parameters())      # .grad IS the flat slot
    opt.step()
    torch.cuda.synchronize()
    out[rank] = (loss, grads, {k: _digest(p) for k, p in model.named_parameters()}, den.cpu().numpy())
    dist.destroy_process_group()


def test_two_rank_step_equals_sequential_shards():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    # single process: the two shards one after the other with the same global normalisers; gradients summed
    from kg_instance_segmentation_amd import KGnet
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.optim import Adam
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    from oracle import synth, weightgen
    sd = weightgen.gen_state_dict(0, variant="cal")
    model = KGnet.resnet50(pretrained=False)
    model.load_state_dict(sd)
    model = model.cuda().train()
    opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    batch = synth.train_batch(N, S, S, 21, n_boxes=NB)
    batch[1][2] = np.zeros((0, 5), np.float32); batch[1][3] = np.zeros((0, 5), np.float32)
    batch[2][2] = np.zeros((0, S, S), np.float32); batch[2][3] = np.zeros((0, S, S), np.float32)
    den = torch.from_numpy(out[0][3]).cuda()
    assert np.array_equal(out[0][3], out[1][3])
    acc, losses = None, []
    for r in range(world):
        model.load_state_dict(sd)               # (running statistics back to the start: every replica sees them once)
        losses.append(_shard_step(model, opt, DetectionLossAll(5), SEG_loss(S, S), batch, slice(r * N // world, (r + 1) * N // world), den, world))
        g = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    assert abs(out[0][0] - losses[0]) == 0.0 and abs(out[1][0] - losses[1]) == 0.0
    for k, p in model.named_parameters():
        p.grad = acc[k]
    ref_g = {k: _digest(v) for k, v in acc.items()}
    model.load_state_dict(sd)
    opt.step()
    torch.cuda.synchronize()
    ref_p = {k: _digest(p) for k, p in model.named_parameters()}
    for r in range(world):
        bad = [k for k in ref_g if out[r][1][k] != ref_g[k]]
        assert not bad, (r, bad[:5])
        # BatchNorm running statistics are per replica; parameters (updated from the SAME reduced gradients) must agree exactly
        badp = [k for k in ref_p if out[r][2][k] != ref_p[k]]
        assert not badp, (r, badp[:5])

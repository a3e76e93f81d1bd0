"""GPU: the data-parallel step of the HIP path (BASELINE configs[3]) on TWO and on EIGHT ranks.

A 1-GPU box cannot host two RCCL ranks (one communicator rank per device), so the processes share GPU 0 and exchange over
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

S, NB = 128, 5


def _nimg(world):
    return 4 if world == 2 else world        # world 2: two images per rank; world 8: one image per rank


def _make_batch(world):
    """the global batch; the images of every ODD rank lose their boxes (no seg backward on those ranks: the collective sequence
    must not depend on it)"""
    from oracle import synth
    N = _nimg(world)
    batch = synth.train_batch(N, S, S, 21, n_boxes=NB)
    for r in range(1, world, 2):
        for i in range(r * N // world, (r + 1) * N // world):
            batch[1][i] = np.zeros((0, 5), np.float32)
            batch[2][i] = np.zeros((0, S, S), np.float32)
    return batch


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
    batch = _make_batch(world)
    N = _nimg(world)
    sl = slice(rank * N // world, (rank + 1) * N // world)
    den = parallel.detection_denominators([g[sl].cuda() for g in batch[3]])
    loss = _shard_step(model, opt, DetectionLossAll(5), SEG_loss(S, S), batch, sl, den, world, reducer)
    # (.grad of a parameter NO rank produced a gradient for stays None, as in one process: FlatGradReducer.finish exchanges the produced bitmap)
    grads = {k: (_digest(p.grad) if p.grad is not None else None) for k, p in model.named_parameters()}
    assert all(p.grad is None or p.grad.data_ptr() == reducer.get(k).data_ptr() for k, p in model.named_parameters())      # .grad IS the flat slot
    opt.step()
    torch.cuda.synchronize()
    out[rank] = (loss, grads, {k: _digest(p) for k, p in model.named_parameters()}, den.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_step_equals_sequential_shards(world):
    """world 8 = the rank count of BASELINE configs[3] (8 processes sharing the one GPU of the box, one image each, every odd rank
    without boxes): the bucket / seg-bucket collective order cannot deadlock and the result equals the sequential shards."""
    N = _nimg(world)
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
    batch = _make_batch(world)
    den = torch.from_numpy(out[0][3]).cuda()
    assert all(np.array_equal(out[0][3], out[r][3]) for r in range(world))
    acc, losses, produced = None, [], set()
    for r in range(world):
        model.load_state_dict(sd)               # (running statistics back to the start: every replica sees them once)
        losses.append(_shard_step(model, opt, DetectionLossAll(5), SEG_loss(S, S), batch, slice(r * N // world, (r + 1) * N // world), den, world))
        produced |= {k for k, p in model.named_parameters() if p.grad is not None}
        g = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    assert all(abs(out[r][0] - losses[r]) == 0.0 for r in range(world))
    for k, p in model.named_parameters():
        p.grad = acc[k] if k in produced else None
    ref_g = {k: (_digest(v) if k in produced else None) for k, v in acc.items()}
    model.load_state_dict(sd)
    opt.step()
    torch.cuda.synchronize()
    ref_p = {k: _digest(p) for k, p in model.named_parameters()}
    def same(a, b):
        if a is None or b is None:         # produced by no shard / no rank: None on both sides
            return a is None and b is None
        if world == 2:                     # a + b is commutative: bit-identical
            return a == b
        # 8 ranks: the ring all-reduce adds the 8 terms in another order than the sequential loop -- equal up to fp32 rounding
        return abs(a[1] - b[1]) <= 2e-5 * abs(b[1]) + 1e-12 and np.allclose(np.frombuffer(a[2], np.float32), np.frombuffer(b[2], np.float32), rtol=2e-4, atol=1e-9)

    for r in range(world):
        bad = [k for k in ref_g if not same(out[r][1][k], ref_g[k])]
        assert not bad, (r, bad[:5])
        # BatchNorm running statistics are per replica; parameters (updated from the SAME reduced gradients) must agree
        badp = [k for k in ref_p if not same(out[r][2][k], ref_p[k])]
        assert not badp, (r, badp[:5])
    # every rank holds the SAME reduced gradients and parameters, bit for bit (they all read one all-reduce result)
    for r in range(1, world):
        assert all(out[r][1][k] == out[0][1][k] for k in ref_g) and all(out[r][2][k] == out[0][2][k] for k in ref_p)


def test_bench_gpus2_launches_itself():
    """`python bench.py --gpus 2 ...` with NO pre-spawned world (the driver's plain command line): bench.py re-runs itself under
    torch.distributed.run, both ranks on GPU 0 over gloo (a 1-GPU box; KG_FORCE_DEVICE / KG_DIST_BACKEND are the hooks of
    parallel.init_from_env), and rank 0 prints ONE JSON line of the data-parallel step: n_gpus 2, global batch = 2 x batch, finite loss."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KG_FORCE_DEVICE="0", KG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "128",
                        "--boxes", "8", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    assert out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert np.isfinite(out["config"]["last_loss"]) and out["config"]["grad_overflow"] is False
    assert abs(out["value"] - 4 * 2 / (out["ms_per_step"] * 2e-3)) <= 1e-6 * out["value"]

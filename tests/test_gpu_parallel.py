"""GPU: the data-parallel code paths (parameter broadcast, global loss normalisers, overlapped bucketed gradient all-reduce,
barrier) run against the real RCCL backend ("nccl") on a one-rank communicator -- the GPU box has a single device, so the
multi-rank arithmetic is covered by the world-size-2 gloo tests (tests/test_parallel_cpu.py) and this test covers the
device / stream / async-work semantics of the backend the 8-GPU runs use."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_dp_step_on_one_rank_rccl_communicator(monkeypatch):
    import torch.distributed as dist
    from kg_instance_segmentation_amd import KGnet, parallel
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    import bench
    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0").items():
        monkeypatch.setenv(k, v)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        monkeypatch.setattr(parallel, "world_size", lambda: 2)       # take the multi-rank branches
        dev = torch.device("cuda", 0)
        torch.manual_seed(0)
        model = KGnet.resnet50(pretrained=False).to(dev).train()
        parallel.broadcast_parameters(model)
        x, gt, gt_masks, gt_boxes = bench.make_batch(2, 128, 12, 7, dev)
        den = parallel.detection_denominators(gt)
        assert den.shape == (4, 3) and bool(torch.isfinite(den).all())
        red = parallel.GradReducer(model.parameters()).attach(model)
        ldec, lseg = DetectionLossAll(5), SEG_loss(128, 128)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        losses = []
        for _ in range(2):
            opt.zero_grad()
            p0, p1, p2, p3, pred = model(x, gt_boxes)
            l1 = sum(ldec(p, g, denominators=den[i]) for i, (p, g) in enumerate(zip((p0, p1, p2, p3), gt)))
            l2 = lseg(pred, gt_masks, gt_boxes)
            loss = l1 if l2 is None else l1 + l2 / 2
            loss.backward()
            assert len(red.covered) > 150 and not red.inflight     # the decoder / head gradients went through the hook
            red.reduce()
            assert not red.covered
            opt.step()
            losses.append(float(loss))
        assert all(np.isfinite(losses))
        assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in model.parameters())
        dist.barrier()
    finally:
        dist.destroy_process_group()

"""GPU: the production data-parallel step -- parallel.FlatGradReducer (in-place async all-reduce on slices of ONE flat gradient
buffer that the gradient kernels write into through ctypes launches on torch's stream) + the fused optim.Adam -- against the real
RCCL backend ("nccl") on a one-rank communicator.  The GPU box has a single device, so the multi-rank arithmetic is covered by the
gloo tests (tests/test_parallel_cpu.py on the CPU, tests/test_gpu_dp.py with several ranks sharing the GPU); what this test pins is
the device / stream / async-work semantics of the backend the 8-GPU runs use: the all-reduce runs on RCCL's own stream against
gradient kernels enqueued on torch's stream, so a missing dependency would show up as gradients that differ from the run
without the reducer.  They must be BIT-identical (world size 1: the sum over ranks is the identity), for two consecutive steps,
in the default half-plane policy (whose reducer also divides the backward pass's power-of-two scale out of every bucket right
before its all-reduce) and in a bf16 policy."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_flat_reducer_and_fused_adam_on_one_rank_rccl_communicator(monkeypatch, precision):
    import torch.distributed as dist
    from kg_instance_segmentation_amd import KGnet, parallel
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.optim import Adam
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    import bench
    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0").items():
        monkeypatch.setenv(k, v)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        x, gt, gt_masks, gt_boxes = bench.make_batch(2, 128, 12, 7, dev)
        ldec, lseg = DetectionLossAll(5), SEG_loss(128, 128)

        def run(with_reducer):
            torch.manual_seed(0)
            model = KGnet.resnet50(pretrained=False, precision=precision).to(dev).train()
            if with_reducer:
                monkeypatch.setattr(parallel, "world_size", lambda: 2)       # take the multi-rank branches on the 1-rank communicator
            parallel.broadcast_parameters(model)
            den = parallel.detection_denominators(gt)
            assert den.shape == (4, 3) and bool(torch.isfinite(den).all())
            red = parallel.FlatGradReducer(bucket_mb=16).attach(model) if with_reducer else None
            opt = Adam(model.parameters(), lr=1e-4)
            grads, losses = [], []
            for _ in range(2):
                opt.zero_grad()
                p0, p1, p2, p3, pred = model(x, gt_boxes)
                l1 = sum(ldec(p, g, denominators=den[i]) for i, (p, g) in enumerate(zip((p0, p1, p2, p3), gt)))
                l2 = lseg(pred, gt_masks, gt_boxes)
                loss = l1 if l2 is None else l1 + l2
                loss.backward()
                if red is not None:
                    assert len(red.inflight) >= 2          # buckets went out during the backward pass
                    red.finish()
                    assert not red.inflight
                    assert all(p.grad is None or p.grad.data_ptr() == red.get(n).data_ptr() for n, p in model.named_parameters())
                grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
                opt.step()
                losses.append(float(loss))
            monkeypatch.setattr(parallel, "world_size", lambda: 1)
            return grads, losses, {n: p.detach().clone() for n, p in model.named_parameters()}

        g_ref, l_ref, p_ref = run(False)
        g_red, l_red, p_red = run(True)
        assert l_ref == l_red and all(np.isfinite(l_ref))
        for step in range(2):
            # (a parameter no box of the global batch reaches has no gradient without the reducer and none with it: the produced-bitmap
            # exchange of finish() leaves its .grad None on every rank)
            bad = [n for n, g in g_ref[step].items() if not torch.equal(g, g_red[step][n])]
            assert not bad, (step, bad[:5])
            assert sorted(g_red[step]) == sorted(g_ref[step])
        bad = [n for n in p_ref if not torch.equal(p_ref[n], p_red[n]) and n in g_ref[1]]
        assert not bad, bad[:5]
        dist.barrier()
    finally:
        dist.destroy_process_group()

#!/usr/bin/env python
"""Does the host side of R data-parallel ranks serialise on one host?  R processes (gloo rendezvous; they SHARE the one GPU of the box --
this measures the HOST, not the GPU) each run the production train step of bench.py at its configuration and time how long the host
needs to ENQUEUE one step (zero_grad .. optimizer step, no device synchronisation inside).  Compared: 1 rank alone vs R ranks at once
(every step starts behind a barrier so that the R hosts enqueue simultaneously).  If the per-rank enqueue time does not grow with R,
8 ranks x ~18 ms of Python / ctypes per step do not contend for anything on the host (the GIL is per process).

    python tools/dp_host_probe.py [--ranks 8] [--steps 4] [--out profiles/r04_dp_host_probe.txt]        (GPU box)"""
import argparse
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def worker(rank, world, port, steps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      KG_DIST_BACKEND="gloo", KG_FORCE_DEVICE="0")
    import torch
    import torch.distributed as dist
    import bench
    from kg_instance_segmentation_amd import KGnet, parallel
    from kg_instance_segmentation_amd.loss import DetectionLossAll
    from kg_instance_segmentation_amd.optim import Adam
    from kg_instance_segmentation_amd.seg_loss import SEG_loss
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // world)))
    parallel.init_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x, gt, gt_masks, gt_boxes = bench.make_batch(8, 512, 300, 100 + rank, dev)
    den = parallel.detection_denominators(gt) if world > 1 else None
    torch.manual_seed(1234)
    model = KGnet.resnet50(pretrained=False).to(dev).train()
    parallel.broadcast_parameters(model)
    opt = Adam(model.parameters(), lr=1e-4, prepack=model)
    red = parallel.FlatGradReducer().attach(model) if world > 1 else None
    ldec, lseg = DetectionLossAll(5), SEG_loss(512, 512)
    host = []
    for it in range(steps + 2):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter(); c0 = time.thread_time()
        opt.zero_grad()
        p0, p1, p2, p3, pred = model(x, gt_boxes)
        if den is None:
            l1 = ldec(p0, gt[0]) + ldec(p1, gt[1]) + ldec(p2, gt[2]) + ldec(p3, gt[3])
        else:
            l1 = sum(ldec(p, g, denominators=den[i]) for i, (p, g) in enumerate(zip((p0, p1, p2, p3), gt)))
        loss = l1 + lseg(pred, gt_masks, gt_boxes) / world
        loss.backward()
        t1 = time.perf_counter(); c1 = time.thread_time()      # everything up to here is enqueue only
        if red is not None:
            red.finish()                              # (waits for the collectives: device time, not counted as enqueue)
        opt.step()
        t2 = time.perf_counter()
        float(loss.detach())
        if it >= 2:
            host.append((1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (c1 - c0)))
    out[rank] = host
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch.multiprocessing as mp
    lines = [f"host enqueue time of one train step (bench configuration) per rank, {args.steps} steps: forward + losses + backward enqueue | incl. finish() + optimizer"]
    for world in (1, args.ranks):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(worker, args=(world, port, args.steps, out), nprocs=world, join=True)
        a = [v[0] for r in range(world) for v in out[r]]
        b = [v[1] for r in range(world) for v in out[r]]
        c = [v[2] for r in range(world) for v in out[r]]
        lines.append(f"   {world} rank(s) on {os.cpu_count()} host cores: enqueue wall mean {sum(a) / len(a):6.2f} ms, max {max(a):6.2f} ms; CPU time of the enqueueing "
                     f"thread mean {sum(c) / len(c):6.2f} ms, max {max(c):6.2f} ms | incl. finish + optimizer wall mean {sum(b) / len(b):6.2f} ms, max {max(b):6.2f} ms")
    lines.append("(the ranks share ONE GPU here: the device drains R steps one after the other, finish() waits for it, and once a rank's hardware queue is "
                 "full its launches block -- WALL enqueue time with R > 1 therefore contains queue back-pressure; the CPU-time column is the host work itself)")
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()

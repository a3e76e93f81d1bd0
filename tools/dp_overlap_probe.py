#!/usr/bin/env python
"""When do the gradient buckets of parallel.FlatGradReducer go out, relative to the backward pass they overlap with?

One MI355X, backend "nccl" (= RCCL) on a ONE-rank communicator with the multi-rank branches forced (parallel.world_size patched to 2, as
tests/test_gpu_parallel.py does): the production step of bench.py at its configuration (batch 8 x 512^2, 300 boxes).  For every bucket the
tool records a HIP event on the compute stream at the moment its in-place all-reduce is enqueued, plus events at the start of the backward
pass and at FlatGradReducer.finish(); it prints, per bucket: size, time since the backward started, time LEFT until finish() -- the window an
8-GPU all-reduce of that bucket has to hide in (ring 296 MB ~ 3.4 ms per-link bound, direct ~0.5 ms).  No scaling number is
claimed: a one-rank all-reduce moves nothing; what is measured is the issue schedule and that the collective + unscale + bitmap exchange
cost nothing on the compute stream.

    python tools/dp_overlap_probe.py [--steps 5] [--out profiles/r04_dp_overlap.txt]        (GPU box)"""
import argparse
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

import bench
from kg_instance_segmentation_amd import KGnet, parallel
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.optim import Adam
from kg_instance_segmentation_amd.seg_loss import SEG_loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    x, gt, gt_masks, gt_boxes = bench.make_batch(8, 512, 300, 100, dev)
    ldec, lseg = DetectionLossAll(5), SEG_loss(512, 512)
    lines = []

    def run(with_reducer):
        torch.manual_seed(1234)
        model = KGnet.resnet50(pretrained=False).to(dev).train()
        parallel.world_size = (lambda: 2) if with_reducer else (lambda: 1)
        den = parallel.detection_denominators(gt)
        red = parallel.FlatGradReducer(bucket_mb=args.bucket_mb).attach(model) if with_reducer else None
        opt = Adam(model.parameters(), lr=1e-4, prepack=model)
        marks = []
        if red is not None:
            orig = red._launch

            def launch(a, b, keys, unscale=True):
                e = torch.cuda.Event(enable_timing=True); e.record()
                marks.append((f"bucket [{keys[0]} ... {keys[-1]}] {4 * (b - a) / 2 ** 20:.0f} MB, {len(keys)} tensors", e))
                return orig(a, b, keys, unscale=unscale)
            red._launch = launch
        acc, wall = {}, []
        for it in range(args.steps + 2):
            marks.clear()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            opt.zero_grad()
            p0, p1, p2, p3, pred = model(x, gt_boxes)
            loss = sum(ldec(p, g, denominators=den[i]) for i, (p, g) in enumerate(zip((p0, p1, p2, p3), gt))) + lseg(pred, gt_masks, gt_boxes)
            eb = torch.cuda.Event(enable_timing=True); eb.record()
            loss.backward()
            ef = torch.cuda.Event(enable_timing=True); ef.record()
            if red is not None:
                red.finish()
            ee = torch.cuda.Event(enable_timing=True); ee.record()
            opt.step()
            loss.item()
            torch.cuda.synchronize()
            if it >= 2:
                wall.append(1e3 * (time.perf_counter() - t0))
                for name, e in marks:
                    a = acc.setdefault(name, [0.0, 0.0])
                    a[0] += eb.elapsed_time(e); a[1] += e.elapsed_time(ef)
                a = acc.setdefault("_bwd", [0.0, 0.0]); a[0] += eb.elapsed_time(ef); a[1] += ef.elapsed_time(ee)
        return acc, sum(wall) / len(wall)

    acc0, w0 = run(False)
    acc1, w1 = run(True)
    n = args.steps
    lines.append(f"FlatGradReducer on a one-rank RCCL communicator (multi-rank branches forced), bench configuration, mean of {n} steps")
    lines.append(f"step wall time: {w0:.2f} ms without the reducer, {w1:.2f} ms with it (buckets of {args.bucket_mb} MB, in-place async all-reduce on RCCL's stream, "
                 f"unscale-before-all-reduce, produced-bitmap + overflow-flag exchange in finish())")
    lines.append(f"backward pass on the compute stream: {acc1['_bwd'][0] / n:.2f} ms; finish() (waits + bitmap read-back): {acc1['_bwd'][1] / n:.2f} ms")
    lines.append("bucket: issued after the start of backward [ms] | compute-stream time left until finish() [ms]")
    for name, (a, b) in acc1.items():
        if name != "_bwd":
            lines.append(f"   {a / n:7.2f} | {b / n:7.2f}   {name}")
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

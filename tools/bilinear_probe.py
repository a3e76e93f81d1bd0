import sys, torch
sys.path.insert(0, "/root/repo")
from kg_instance_segmentation_amd import ops
dev = "cuda"
for (N, IH, C) in [(8, 32, 1024), (8, 64, 512), (8, 128, 256), (8, 256, 64)]:
    OH = 2 * IH
    for P, dt in [(2, ops.F16), (1, ops.F16)]:
        x = ops.alloc_pt(N * IH * IH, C, P, dev, dtype=dt); x.t.normal_()
        y = ops.alloc_pt(N * OH * OH, C, P, dev, dtype=dt)
        g = ops.alloc_pt(N * IH * IH, C, P, dev, dtype=dt)
        for _ in range(3): ops.bilinear_fwd(x, y, N, IH, IH, OH, OH, C)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): ops.bilinear_fwd(x, y, N, IH, IH, OH, OH, C)
        e.record(); torch.cuda.synchronize()
        tf = s.elapsed_time(e) / 20
        for _ in range(3): ops.bilinear_bwd(y, g, N, IH, IH, OH, OH, C)
        s.record()
        for _ in range(20): ops.bilinear_bwd(y, g, N, IH, IH, OH, OH, C)
        e.record(); torch.cuda.synchronize()
        tb = s.elapsed_time(e) / 20
        by = (N * IH * IH + N * OH * OH) * C * 2 * P
        print(f"N={N} {IH}^2 -> {OH}^2 C={C} planes={P}: fwd {tf*1e3:7.1f} us = {by/tf/1e6:6.0f} GB/s   bwd {tb*1e3:7.1f} us = {by/tb/1e6:6.0f} GB/s   ({by/1e6:.0f} MB)")

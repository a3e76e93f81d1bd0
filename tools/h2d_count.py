"""H2D uploads (ops.h2d) of one train step at the bench configuration, by call site.   python tools/h2d_count.py"""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kg_instance_segmentation_amd import ops, KGnet
from kg_instance_segmentation_amd.loss import DetectionLossAll
from kg_instance_segmentation_amd.seg_loss import SEG_loss
from kg_instance_segmentation_amd.optim import Adam
cnt = collections.Counter(); byt = collections.Counter()
orig = ops.h2d
def h2d(arr, dev):
    fr = traceback.extract_stack(limit=3)[0]
    k = f"{os.path.basename(fr.filename)}:{fr.lineno}:{fr.name}"
    cnt[k] += 1; byt[k] += getattr(arr, 'nbytes', 0)
    return orig(arr, dev)
ops.h2d = h2d
import kg_instance_segmentation_amd.seg as seg, kg_instance_segmentation_amd.seg_loss as sl
for mod in (seg, sl):
    if hasattr(mod, 'h2d'): mod.h2d = h2d
dev = torch.device('cuda', 0)
model = KGnet.resnet50(pretrained=False).to(dev).train()
opt = Adam(model.parameters(), lr=1e-4)
x, gt, gm, gb = bench.make_batch(8, 512, 300, 7, dev)
ldec, lseg = DetectionLossAll(5), SEG_loss(512, 512)
for s in range(3):
    if s == 2: cnt.clear(); byt.clear()
    opt.zero_grad()
    p0, p1, p2, p3, pred = model(x, gb)
    loss = ldec(p0, gt[0]) + ldec(p1, gt[1]) + ldec(p2, gt[2]) + ldec(p3, gt[3]) + lseg(pred, gm, gb)
    loss.backward(); opt.step()
torch.cuda.synchronize()
for k, v in cnt.most_common(): print(v, byt[k], k)
print('total', sum(cnt.values()))

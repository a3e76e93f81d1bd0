"""Drop-in `loss` module (reference loss.py:7-49): DetectionLossAll on the fused HIP loss kernels."""
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ptr, stream_ptr, c_float


class _DetLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kp, short, mid, gt, kp_radius, den):
        N, _, H, W = kp.shape
        dev = kp.device
        sc = ops.scratch_f32(5 * 1024, dev, "loss")
        out8 = torch.empty(8, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):      # (launch on the tensors' device, whatever the "current" one is)
            _lib.call("kg_detection_loss_fwd", ptr(kp), ptr(short), ptr(mid), ptr(gt), N, H, W, c_float(kp_radius), ptr(den),
                      ptr(sc), sc.numel(), ptr(out8), stream_ptr())
        ctx.save_for_backward(kp, short, mid, gt, out8)
        ctx.kp_radius = kp_radius
        return out8[0].clone()

    @staticmethod
    def backward(ctx, go):
        kp, short, mid, gt, out8 = ctx.saved_tensors
        N, _, H, W = kp.shape
        g_kp, g_sh, g_md = torch.empty_like(kp), torch.empty_like(short), torch.empty_like(mid)
        go = go.contiguous().float()
        with torch.cuda.device(kp.device):
            _lib.call("kg_detection_loss_bwd", ptr(kp), ptr(short), ptr(mid), ptr(gt), N, H, W, c_float(ctx.kp_radius), ptr(out8),
                      ptr(go), ptr(g_kp), ptr(g_sh), ptr(g_md), stream_ptr())
        return g_kp, g_sh, g_md, None, None, None


class DetectionLossAll(nn.Module):
    """BCE(kp) + masked-L1(short) + 0.25 * masked-L1(mid)  (loss.py:40-49).

    `denominators` (optional device tensor [mask2_sum, mask4_sum, kp_numel]) replaces the local
    normalisers -- used by data-parallel training to reproduce the single-device batch loss."""

    def __init__(self, kp_radius):
        super().__init__()
        self.kp_radius = kp_radius

    def forward(self, prediction, groundtruth, denominators=None):
        pr_kp, pr_short, pr_mid = prediction
        if not pr_kp.is_cuda:
            raise _lib.KGLibraryError("DetectionLossAll (MI355X build) needs GPU tensors")
        gt = groundtruth.contiguous().float()
        return _DetLossFn.apply(pr_kp.contiguous().float(), pr_short.contiguous().float(), pr_mid.contiguous().float(), gt,
                                float(self.kp_radius), denominators)

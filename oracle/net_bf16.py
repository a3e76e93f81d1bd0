"""bf16-storage emulation of the HIP engine on top of the fp32 oracle (TEST INFRASTRUCTURE).

Same arithmetic as oracle/net.py, but every tensor the HIP path stores in bfloat16 (packed weights,
activations after each fused conv/BN/upsample stage, and the data gradients flowing back through
those points) is rounded to bf16 at the same place; accumulation stays fp32, weight gradients stay
fp32.  Used to separate "bf16 storage precision" (inherent, see DESIGN.md Numerics) from
"implementation error" when comparing parameter gradients: on a random-init network the fp32 and
bf16 gradients of the deep backbone decorrelate for BOTH this emulation and the HIP path.
"""
import torch
import torch.nn.functional as F

from . import net as onet


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def r(x):
    return _RoundSTE.apply(x)


def rw(w):
    """forward-only rounding (packed bf16 weights; fp32 master weight receives the fp32 gradient)"""
    return w + (w.detach().to(torch.bfloat16).float() - w.detach())


class NetBF16(onet.Net):
    def conv(self, x, name, stride=1, pad=0, relu=False):
        y = F.conv2d(r(x), rw(self.sd[name + ".weight"]), self.sd.get(name + ".bias"), stride, pad)
        y = F.relu(y) if relu else y
        last = name.endswith(".2") and ("_head_c" in name or name.startswith("seg_head"))
        return y if last else r(y)      # final head / seg-head convs export fp32

    @staticmethod
    def act(x):
        return r(x)

    def bottleneck(self, x, p, stride, has_ds):
        o = r(self.bn(self.conv(x, p + ".conv1"), p + ".bn1", True))
        o = r(self.bn(self.conv(o, p + ".conv2", stride, 1), p + ".bn2", True))
        o = self.bn(self.conv(o, p + ".conv3"), p + ".bn3")
        idt = r(self.bn(self.conv(x, p + ".downsample.0", stride), p + ".downsample.1")) if has_ds else x
        return r(F.relu(o + idt))

    @staticmethod
    def up(x, ref):
        return r(F.interpolate(x, ref.shape[2:], mode="bilinear", align_corners=False))

    def forward_dec(self, x):
        return super().forward_dec(r(x))

"""Fused multi-tensor Adam on the GPU: `Adam(params, lr=1e-4)` is a drop-in for the reference's
`optim.Adam(filter(...), lr=1e-4)` (train.py:71) -- same constructor arguments, `param_groups` (so that
`lr_scheduler.ExponentialLR`, train.py:72, works unchanged) and the same state layout as torch.optim.Adam (`step`, `exp_avg`,
`exp_avg_sq`), but ONE kernel launch per step for all parameter tensors (kg_adam_step) instead of several multi-tensor
passes.  amsgrad / maximize / capturable are not supported (the reference does not use them)."""
import math

import numpy as np
import torch

from . import _lib, ops
from ._lib import ptr, stream_ptr, c_float

_JOB = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("blk0", "<i4"), ("pad", "<i4")])


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not supported by the fused HIP Adam")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            beta1, beta2 = group["betas"]
            by_step = {}
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise _lib.KGLibraryError("the fused HIP Adam needs fp32 parameters and gradients on the GPU")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"]), []).append(p)
            for t, plist in by_step.items():           # (parameters that joined later have their own bias correction)
                arr = np.zeros(len(plist), _JOB)
                blk = 0
                for i, p in enumerate(plist):
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    st = self.state[p]
                    arr[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), blk, 0)
                    if g is not p.grad:
                        st["_g"] = g           # keep the temporary alive until the launch is enqueued
                    blk += (p.numel() + 4095) // 4096
                dev = plist[0].device
                key, sig = (gi, t == 0, len(plist)), arr.tobytes()
                ent = self._tables.get(key)
                if ent is None or ent[0] != sig:
                    ent = (sig, ops.h2d(arr.view(np.uint8).reshape(-1), dev))
                    self._tables[key] = ent
                bc1 = 1.0 - beta1 ** t
                bc2 = 1.0 - beta2 ** t
                with torch.cuda.device(dev):
                    _lib.call("kg_adam_step", ptr(ent[1]), len(plist), blk, c_float(beta1), c_float(beta2), c_float(group["eps"]),
                              c_float(group["lr"] / bc1), c_float(math.sqrt(bc2)), c_float(group["weight_decay"]), stream_ptr())
        ops.PARAM_EPOCH[0] += 1      # raw-pointer writes do not bump tensor versions: packed bf16 weight copies are stale now
        return loss

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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, prepack=None):
        """prepack (optional, a KGnet model -- an extension, torch.optim.Adam has no such argument): right after the update kernel the
        model packs the 16-bit weight copies of its NEXT training forward (engine.Engine.prepack), i.e. before the host blocks in the
        step's loss read-back instead of after it."""
        self._prepack = prepack
        if amsgrad:
            raise NotImplementedError("amsgrad is not supported by the fused HIP Adam")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._tables = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._shadow = {}                    # the Python-side step counters follow the loaded `step` tensors
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        shadow = self.__dict__.setdefault("_shadow", {})
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            beta1, beta2 = group["betas"]
            steps = []
            for p in ps:
                st = self.state[p]
                if not st:
                    if not p.is_cuda or p.dtype != torch.float32:
                        raise _lib.KGLibraryError("the fused HIP Adam needs fp32 parameters and gradients on the GPU")
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                steps.append(st["step"])
            torch._foreach_add_(steps, 1)              # one call for the ~160 host-side counters (torch.optim.Adam's state layout)
            by_step = {}
            for p, sv in zip(ps, steps):
                t = shadow.get(id(p))
                t = int(sv) if t is None else t + 1    # (shadow: an int() per CPU tensor costs more than the whole table)
                shadow[id(p)] = t
                by_step.setdefault(t, []).append(p)
            for t, plist in by_step.items():           # (parameters that joined later have their own bias correction)
                key = (gi, t == 0, len(plist))
                ent = self._tables.get(key)
                sts = [self.state[p] for p in plist]
                ids = ([p.data_ptr() for p in plist], [st["exp_avg"].data_ptr() for st in sts], [st["exp_avg_sq"].data_ptr() for st in sts])
                if ent is None or ent["ids"] != ids:   # rarely-changing columns of the job table: parameter, moments, sizes, first workgroup
                    arr = np.zeros(len(plist), _JOB)
                    blk = 0
                    for i, p in enumerate(plist):
                        st = self.state[p]
                        arr[i] = (p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), blk, 0)
                        blk += (p.numel() + 4095) // 4096
                    ent = {"ids": ids, "arr": arr, "blk": blk, "gptr": None, "dev": None}
                    self._tables[key] = ent
                keep = []
                gptr = []
                for p in plist:
                    g = p.grad
                    if g.dtype != torch.float32 or not g.is_cuda:
                        raise _lib.KGLibraryError("the fused HIP Adam needs fp32 parameters and gradients on the GPU")
                    if not g.is_contiguous():
                        g = g.contiguous(); keep.append(g)         # alive until the launch is enqueued
                    gptr.append(g.data_ptr())
                dev = plist[0].device
                if ent["gptr"] != gptr:                # (gradients living in persistent slots, parallel.FlatGradReducer: the table is reused)
                    ent["arr"]["g"] = gptr
                    ent["gptr"] = gptr
                    ent["dev"] = ops.h2d(ent["arr"].view(np.uint8).reshape(-1), dev)
                bc1 = 1.0 - beta1 ** t
                bc2 = 1.0 - beta2 ** t
                with torch.cuda.device(dev):
                    _lib.call("kg_adam_step", ptr(ent["dev"]), len(plist), ent["blk"], c_float(beta1), c_float(beta2), c_float(group["eps"]),
                              c_float(group["lr"] / bc1), c_float(math.sqrt(bc2)), c_float(group["weight_decay"]), stream_ptr())
                del keep
        ops.PARAM_EPOCH[0] += 1      # raw-pointer writes do not bump tensor versions: packed bf16 weight copies are stale now
        if self._prepack is not None:
            self._prepack._engine.prepack()
        return loss

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad(set_to_none=True) without its per-call profiler / foreach bookkeeping (0.4 -> 0.1 ms of host
        time for the 217 parameters, on the critical path between the loss read-back of step k and the first kernel of step k + 1)"""
        if self._prepack is not None:
            ops.launch_held_packs()      # the pack kernel prepack() prepared runs while this loop frees 217 gradient tensors
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

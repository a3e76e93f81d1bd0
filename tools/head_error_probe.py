#!/usr/bin/env python
"""GPU box: WHERE does the distance from float64 come from?  The two 7x7 head layers of every level in isolation: their input is the float64
oracle's own decoder output (rounded once to fp32, the same tensor for everybody), the reference value is the float64 evaluation of the two
layers on that fp32 input with the fp32 weights, and the candidates are (a) torch-CPU float32 -- the reference's arithmetic -- and (b) the
HIP kernels in the given policies.  Printed per map: rms error and worst |d| / bound (bound as tools/fullsize_oracle_parity.py).
If (b) is no worse than (a) here, the head convs' accumulation is not what separates the policy from the fp32 oracle at full size.

    python tools/head_error_probe.py [size] [policy,policy,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from kg_instance_segmentation_amd import KGnet, arch, ops
from kg_instance_segmentation_amd.engine import Var
from oracle import net as onet, weightgen
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fullsize_oracle_parity import bound_of

DEV = "cuda"


class Tap(onet.Net):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.head_in = {}

    def conv(self, x, name, stride=1, pad=0, relu=False):
        if name.startswith("kp_head_c") and name.endswith(".0"):
            self.head_in[int(name[len("kp_head_c")])] = x
        return super().conv(x, name, stride, pad, relu)


def heads_cpu(sd, x, lvl, dtype):
    out = {}
    sdd = {k: v.to(dtype) for k, v in sd.items() if f"_head_c{lvl}." in k}
    for h, _ in arch.HEADS:
        p = f"{h}_head_c{lvl}"
        y = F.conv2d(F.relu(F.conv2d(x.to(dtype), sdd[p + ".0.weight"], sdd[p + ".0.bias"], 1, 3)), sdd[p + ".2.weight"], sdd[p + ".2.bias"], 1, 3)
        out[{"kp": "kp_logit", "short_offset": "short", "mid_offset": "mid"}[h]] = y
    return out


def hidden_cpu(sd, x, lvl, dtype):
    """the three first layers (+ ReLU), channels concatenated in arch.HEADS order: [N, 3C, H, W]"""
    return torch.cat([F.relu(F.conv2d(x.to(dtype), sd[f"{h}_head_c{lvl}.0.weight"].to(dtype), sd[f"{h}_head_c{lvl}.0.bias"].to(dtype), 1, 3)) for h, _ in arch.HEADS], 1)


def second_cpu(sd, hid, lvl, dtype):
    C = hid.shape[1] // 3
    out = {}
    for k, (h, _) in enumerate(arch.HEADS):
        p = f"{h}_head_c{lvl}.2"
        out[{"kp": "kp_logit", "short_offset": "short", "mid_offset": "mid"}[h]] = F.conv2d(hid[:, k * C:(k + 1) * C].to(dtype), sd[p + ".weight"].to(dtype), sd[p + ".bias"].to(dtype), 1, 3)
    return out


def layers_gpu(model, x, hid_in, lvl):
    """(hidden of the fused first layer on x as fp32 NCHW, maps of the grouped second layer on the GIVEN hidden tensor hid_in)"""
    eng = model._engine
    N, C, H, W = x.shape
    r = x.permute(0, 2, 3, 1).reshape(N * H * W, C).contiguous().to(DEV)
    pt = ops.alloc_pt(N * H * W, C, eng.pd, DEV, dtype=eng.dt)
    ops.f32_to_planes(r, pt, C)
    eng.tape = None
    eng.raw_kp_logits = True
    fused = [f"{h}_head_c{lvl}.0" for h, _ in arch.HEADS]
    hid, _, _ = eng.conv(Var(pt, C, relu=True, req=False), eng.spec(f"heads_c{lvl}.0", C, C, 7, 1, 3, fused=fused, P=eng.ph), N, H, W, True)
    hf = torch.empty(N * H * W, 3 * C, dtype=torch.float32, device=DEV)
    ops.planes_to_f32(hid.t, 3 * C, hf)
    hr = hid_in.permute(0, 2, 3, 1).reshape(N * H * W, 3 * C).contiguous().to(DEV)
    hp = ops.alloc_pt(N * H * W, 3 * C, eng.ph, DEV, dtype=eng.dt)
    ops.f32_to_planes(hr, hp, 3 * C)
    eng.head_slots = []
    outs = eng.heads_second(Var(hp, 3 * C, relu=True, req=False), lvl, C, N, H, W)
    torch.cuda.synchronize()
    return hf.cpu().view(N, H, W, 3 * C).permute(0, 3, 1, 2), dict(zip(("kp_logit", "short", "mid"), [o.cpu() for o in outs]))


def heads_gpu(model, x, lvl):
    eng = model._engine
    N, C, H, W = x.shape
    r = x.permute(0, 2, 3, 1).reshape(N * H * W, C).contiguous().to(DEV)
    pt = ops.alloc_pt(N * H * W, C, eng.pd, DEV, dtype=eng.dt)
    ops.f32_to_planes(r, pt, C)
    eng.tape = None
    eng.raw_kp_logits = True
    xv = Var(pt, C, relu=True, req=False)
    fused = [f"{h}_head_c{lvl}.0" for h, _ in arch.HEADS]
    eng.head_slots = []
    hid, _, _ = eng.conv(xv, eng.spec(f"heads_c{lvl}.0", C, C, 7, 1, 3, fused=fused, P=eng.ph), N, H, W, True)
    outs = eng.heads_second(hid, lvl, C, N, H, W)
    torch.cuda.synchronize()
    return dict(zip(("kp_logit", "short", "mid"), [o.cpu() for o in outs]))


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    policies = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp32", "fp32bf"]
    torch.set_num_threads(min(os.cpu_count() or 8, 64))
    sd = weightgen.gen_state_dict(0, variant="cal")
    x = torch.rand(2, 3, S, S, generator=torch.Generator().manual_seed(41)) - 0.5
    net = Tap({k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, training=True)
    with torch.no_grad():
        net.forward_dec(x.double())
    models = {}
    for pol in policies:
        m = KGnet.resnet50(pretrained=False, precision=pol)
        m.load_state_dict(sd)
        models[pol] = m.to(DEV).eval()
    print(f"two 7x7 head layers alone, input = float64 oracle's decoder output of a 2 x {S} x {S} train-mode forward rounded to fp32; errors against the float64 evaluation")
    print("%-12s %9s | %-26s" % ("map", "rms", "torch-CPU float32: rms err, worst/bound") + "".join(f" | {p}: rms err, worst/bound" for p in policies))
    for lvl in range(4):
        xin = net.head_in[lvl].float()
        with torch.no_grad():
            ref = heads_cpu(sd, xin, lvl, torch.float64)
            c32 = heads_cpu(sd, xin, lvl, torch.float32)
            got = {p: heads_gpu(models[p], xin, lvl) for p in policies}
        for name in ("kp_logit", "short", "mid"):
            full = f"c{lvl}.{name}"
            r = ref[name]
            b = bound_of(full, r)
            d = (c32[name].double() - r)
            line = "%-12s %9.3g | %.3e  %6.3f        " % (full, float(r.pow(2).mean().sqrt()), float(d.pow(2).mean().sqrt()), float((d.abs() / b).max()))
            for p in policies:
                d = (got[p][name].double() - r)
                line += " | %.3e  %6.3f" % (float(d.pow(2).mean().sqrt()), float((d.abs() / b).max()))
            print(line)
    print()
    print("the two layers separately (relative rms error = rms(err) / rms(ref)): first layer = hidden tensor of the fused C -> 3C conv + ReLU on the SAME fp32 input;")
    print("second layer = the three maps from the SAME fp32 hidden tensor (the float64 hidden rounded to fp32)")
    print("%-22s | %-12s" % ("tensor", "torch-CPU f32") + "".join(f" | {p:>10}" for p in policies))
    for lvl in range(4):
        xin = net.head_in[lvl].float()
        with torch.no_grad():
            h64 = hidden_cpu(sd, xin, lvl, torch.float64)
            h32 = hidden_cpu(sd, xin, lvl, torch.float32)
            hin = h64.float()
            m64 = second_cpu(sd, hin, lvl, torch.float64)
            m32 = second_cpu(sd, hin, lvl, torch.float32)
            got = {p: layers_gpu(models[p], xin, hin, lvl) for p in policies}

        def rel(a, r):
            return float((a.double() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
        print("%-22s | %.3e   " % (f"c{lvl} hidden (C={xin.shape[1]})", rel(h32, h64)) + "".join(" | %.3e" % rel(got[p][0], h64) for p in policies))
        for name in ("kp_logit", "short", "mid"):
            print("%-22s | %.3e   " % (f"c{lvl}.{name} (layer 2)", rel(m32[name], m64[name])) + "".join(" | %.3e" % rel(got[p][1][name], m64[name]) for p in policies))


if __name__ == "__main__":
    main()

// optim.hip -- fused multi-tensor Adam step (train.py:71,154: optim.Adam(lr=1e-4) over 217 parameter tensors).
//
// HBM-bound: reads p, g, m, v and writes p, m, v once (28 B per parameter, 2.07 GB for KGnet's 73.9 M parameters); one
// launch for all tensors through a job table (the per-tensor foreach implementation is ~7 passes of multi-tensor kernels).
// Arithmetic in fp32 in torch.optim.Adam's operation order (exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2)
// .addcmul_(grad, grad, 1-beta2); denom = sqrt(exp_avg_sq)/sqrt(bias_correction2) + eps; param.addcdiv_(exp_avg, denom,
// -lr/bias_correction1)), bias corrections computed on the host in double like PyTorch does.
#include "kg_common.h"

struct AdamJob {   // 48 bytes, mirrored by optim.py
    float* p; const float* g; float* m; float* v;
    long n; int blk0; int pad;
};

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamJob* __restrict__ jobs, int njobs, float beta1, float beta2, float eps,
                                                        float step_size, float bc2_sqrt, float weight_decay) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamJob j = jobs[lo];
    const long base = ((long)blockIdx.x - j.blk0) * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = base + (u * 256 + threadIdx.x) * 4L;
        if (i >= j.n) continue;
        if (i + 4 <= j.n && ((reinterpret_cast<uintptr_t>(j.p + i) | reinterpret_cast<uintptr_t>(j.g + i) | reinterpret_cast<uintptr_t>(j.m + i) |
                              reinterpret_cast<uintptr_t>(j.v + i)) & 15) == 0) {
            f32x4 p = *reinterpret_cast<const f32x4*>(j.p + i), g = *reinterpret_cast<const f32x4*>(j.g + i);
            f32x4 m = *reinterpret_cast<const f32x4*>(j.m + i), v = *reinterpret_cast<const f32x4*>(j.v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float ge = g[e];
                if (weight_decay != 0.f) ge = ge + weight_decay * p[e];
                m[e] = m[e] + (1.f - beta1) * (ge - m[e]);
                v[e] = v[e] * beta2 + (1.f - beta2) * ge * ge;
                const float denom = sqrtf(v[e]) / bc2_sqrt + eps;
                p[e] = p[e] - step_size * (m[e] / denom);
            }
            *reinterpret_cast<f32x4*>(j.p + i) = p; *reinterpret_cast<f32x4*>(j.m + i) = m; *reinterpret_cast<f32x4*>(j.v + i) = v;
        } else {
            for (long k = i; k < i + 4 && k < j.n; ++k) {
                float ge = j.g[k];
                if (weight_decay != 0.f) ge = ge + weight_decay * j.p[k];
                const float m = j.m[k] + (1.f - beta1) * (ge - j.m[k]);
                const float v = j.v[k] * beta2 + (1.f - beta2) * ge * ge;
                j.m[k] = m; j.v[k] = v;
                j.p[k] = j.p[k] - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
            }
        }
    }
}

// jobs: device array of njobs 48-byte records {float* p; const float* g; float* m; float* v; long n; int blk0; int pad;}
// (blk0 = first workgroup of the job; a workgroup updates 4096 elements); total_blocks = sum of ceil(n / 4096).
// step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) for the step count t of this call.
extern "C" int kg_adam_step(const void* jobs, int njobs, int total_blocks, float beta1, float beta2, float eps, float step_size,
                            float bc2_sqrt, float weight_decay, void* stream) {
    KG_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0, "kg_adam_step: empty job list");
    hipLaunchKernelGGL(adam_step_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const AdamJob*)jobs, njobs, beta1, beta2, eps,
                       step_size, bc2_sqrt, weight_decay);
    KG_CHECK_LAUNCH("adam_step");
    return KG_OK;
}

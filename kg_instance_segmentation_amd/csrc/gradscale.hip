// gradscale.hip -- the per-step power-of-two scale of the half-precision backward pass (libkgnet_hip_f16.so, kg_common.h).
//
// Gradients of a train step span 1e-7 (BCE mean over N*5*H*W, loss.py:13) to 1e-3 (masked L1 over a handful of keypoint pixels,
// loss.py:25) at the top of the network and are stored in IEEE half (normal range 6e-5 .. 65504) by the f16 build.  The backward
// pass is LINEAR in the gradients of the loss w.r.t. the network outputs, so ONE scale S applied where those gradients enter
// (kg_grad_pack, kg_f32_to_planes: kg_planes_t.scale) flows through every backward kernel untouched and is divided out where fp32
// results leave (kg_scale_tensors over the parameter gradients).  S is a power of two (both scalings are exact) chosen ON THE DEVICE
// from the data of this very step: S = 2^(target - e) with max|top-level gradient| = m * 2^e, m in [0.5, 1) -- no host round trip,
// no state carried between steps, no skipped optimizer steps.  Nothing here depends on the 16-bit format: fp32 in, fp32 out.
#include "kg_common.h"

struct GradScaleArgs {
    const float* p[24];
    const float* q[24];     // optional probabilities: the gradient that enters the network is p * q * (1 - q) (sigmoid outputs: kg_grad_pack)
    long n[24];
    int count, target_log2;
};

// scratch: 2 unsigned, zero on entry (allocated zeroed once; the last workgroup leaves them zero again).  out[0] = S, out[1] = 1 / S.
__global__ __launch_bounds__(1024) void grad_scale_kernel(GradScaleArgs a, unsigned* scratch, float* out) {
    unsigned best = 0;
    for (int t = 0; t < a.count; ++t) {
        const float* p = a.p[t];
        const float* q = a.q[t];
        const long n = a.n[t];
        auto take = [&](float v, float s, bool hasq) {
            if (hasq) v *= s * (1.f - s);                                // (exactly kg_grad_pack's expression)
            const unsigned b = __float_as_uint(v) & 0x7fffffffu;         // |v| as ordered bits (NaN sorts above inf: handled below)
            best = b > best ? b : best;
        };
        const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q)) & 15) == 0;
        const long n4 = vec ? n >> 2 : 0;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {   // 16-byte loads: the pass is HBM-bound
            const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (q) s = reinterpret_cast<const f32x4*>(q)[i];
            take(v[0], s[0], q != nullptr); take(v[1], s[1], q != nullptr); take(v[2], s[2], q != nullptr); take(v[3], s[3], q != nullptr);
        }
        for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) take(p[i], q ? q[i] : 0.f, q != nullptr);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned q = __shfl_xor(best, o, 64); best = q > best ? q : best; }
    __shared__ unsigned wmax[16];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = wmax[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = wmax[w] > m ? wmax[w] : m;
        // (max of bit patterns: order-independent, reproducible.)  Device-scope atomics only, so no fence (a release fence writes the XCD's
        // L2 back: norm_pool.hip rows_absmax_kernel); the ticket is taken after the max has returned (data dependence through `one`)
        unsigned one = 1u;
        const unsigned old = atomicMax(&scratch[0], m);
        asm volatile("" : "+v"(one) : "v"(old));
        last = atomicAdd(&scratch[1], one) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        const unsigned m = atomicMax(&scratch[0], 0u);
        float S = 1.f;
        if (m != 0 && m < 0x7f800000u) {
            int e;
            frexpf(__uint_as_float(m), &e);                                // max = f * 2^e, f in [0.5, 1)
            int k = a.target_log2 - e;
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            S = ldexpf(1.f, k);
        }
        out[0] = S; out[1] = 1.f / S;
        atomicExch(&scratch[0], 0u); atomicExch(&scratch[1], 0u);
    }
}

// ptrs / counts: n <= 24 fp32 device tensors (the gradients of the loss w.r.t. the network outputs of this step); probs (optional host
// array, entries may be null): the sigmoid output the gradient refers to -- the value that counts is then g * q * (1 - q), the gradient
// w.r.t. the LOGIT that kg_grad_pack hands to the network (with saturated sigmoids dL/dq reaches 1e12 while that product is 0); out: 2 floats
// {S, 1 / S}; scratch: 2 zeroed unsigned (left zeroed).  target_log2: the largest gradient lands in [2^(t-1), 2^t).
extern "C" int kg_grad_scale(const void* const* ptrs, const void* const* probs, const long* counts, int n, int target_log2, void* scratch, float* out,
                             void* stream) {
    KG_CHECK_ARG(ptrs && counts && scratch && out && n >= 1 && n <= 24, "kg_grad_scale: 1..24 tensors, scratch and out required");
    GradScaleArgs a;
    long total = 0;
    a.count = n; a.target_log2 = target_log2;
    for (int i = 0; i < 24; ++i) {
        a.p[i] = i < n ? (const float*)ptrs[i] : nullptr; a.q[i] = (i < n && probs) ? (const float*)probs[i] : nullptr;
        a.n[i] = i < n ? counts[i] : 0; total += a.n[i];
    }
    for (int i = 0; i < n; ++i) KG_CHECK_ARG(a.p[i] && a.n[i] >= 0, "kg_grad_scale: null tensor");
    int blocks = (int)((total + 1024 * 16 - 1) / (1024 * 16));
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);      // (few, fat workgroups: same-address atomics serialise, rows_absmax_kernel)
    hipLaunchKernelGGL(grad_scale_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, a, (unsigned*)scratch, out);
    KG_CHECK_LAUNCH("grad_scale");
    return KG_OK;
}

struct ScaleJob {   // 32 bytes, mirrored by ops.scale_tensors
    float* p; long n; const float* scale; int blk0; int pad;
};
__global__ __launch_bounds__(256) void scale_tensors_kernel(const ScaleJob* __restrict__ jobs, int njobs, int* __restrict__ nonfinite) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ScaleJob j = jobs[lo];
    const float s = *j.scale;
    const long base = ((long)blockIdx.x - j.blk0) * 4096;
    bool bad = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = base + (u * 256 + threadIdx.x) * 4L;
        if (i >= j.n) continue;
        if (i + 4 <= j.n && (reinterpret_cast<uintptr_t>(j.p + i) & 15) == 0) {
            f32x4 v = *reinterpret_cast<const f32x4*>(j.p + i);
            v[0] *= s; v[1] *= s; v[2] *= s; v[3] *= s;
            bad = bad || !(fabsf(v[0]) <= 3.4e38f) || !(fabsf(v[1]) <= 3.4e38f) || !(fabsf(v[2]) <= 3.4e38f) || !(fabsf(v[3]) <= 3.4e38f);
            *reinterpret_cast<f32x4*>(j.p + i) = v;
        } else {
            for (long k = i; k < i + 4 && k < j.n; ++k) { const float v = j.p[k] * s; bad = bad || !(fabsf(v) <= 3.4e38f); j.p[k] = v; }
        }
    }
    // a half-precision backward pass that left IEEE half's range (more than 64x growth inside one stage, or an inf / NaN activation)
    // shows up HERE as a non-finite parameter gradient: raise the sticky flag (KGnet.grad_overflowed() reads it; nothing is clamped)
    if (nonfinite && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nonfinite, 1);
}
// jobs: device array of njobs 32-byte records {float* p; long n; const float* scale; int blk0; int pad;} (a workgroup scales 4096
// elements; total_blocks = sum of ceil(n / 4096)); every element of job j is multiplied by *jobs[j].scale (a device scalar: the
// 1 / S of kg_grad_scale or the 1 / cum of the backbone stage the parameter belongs to, kg_rows_rescale).  nonfinite (optional device
// int): set to 1 when a scaled element is inf / NaN.
extern "C" int kg_scale_tensors(const void* jobs, int njobs, int total_blocks, int* nonfinite, void* stream) {
    KG_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0, "kg_scale_tensors: empty job list");
    hipLaunchKernelGGL(scale_tensors_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const ScaleJob*)jobs, njobs, nonfinite);
    KG_CHECK_LAUNCH("scale_tensors");
    return KG_OK;
}

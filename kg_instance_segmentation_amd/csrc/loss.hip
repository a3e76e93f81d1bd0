// loss.hip -- fused detection loss (loss.py:12-49) and per-box mask BCE (seg_loss.py:86-94) for gfx950.
// HBM-bound streaming reductions: one thread per pixel walks the 55 prediction + 55 ground-truth
// channels of the fp32 NCHW maps (consecutive lanes = consecutive pixels => coalesced), wave
// shuffle + LDS block reduction, fixed-order final combine in double (reproducible).
#include "kg_common.h"

__constant__ int KG_MID_FROM[20] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3, 1, 2, 3, 4, 2, 3, 4, 3, 4, 4};  // config.py:2-13

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// torch binary_cross_entropy element: (t-1)*max(log1p(-p),-100) - t*max(log(p),-100)
__device__ __forceinline__ float bce_elem(float p, float t) {
    float l1 = log1pf(-p); l1 = l1 > -100.f ? l1 : -100.f;
    float l0 = logf(p); l0 = l0 > -100.f ? l0 : -100.f;
    return (t - 1.f) * l1 - t * l0;
}
__device__ __forceinline__ float bce_grad(float p, float t) {
    float d = (1.f - p) * p;
    d = d > 1e-12f ? d : 1e-12f;
    return (p - t) / d;
}

// partial[b][5] = {sum bce, sum short, sum mask2, sum mid, sum mask4}
__global__ __launch_bounds__(256) void det_loss_fwd_kernel(const float* __restrict__ kp, const float* __restrict__ sh,
                                                           const float* __restrict__ md, const float* __restrict__ gt,
                                                           int N, long HW, float inv_r, float* __restrict__ part) {
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const long total = (long)N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long n = i / HW, p = i - n * HW;
        const float* g = gt + n * 55 * HW + p;
        float gk[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            gk[c] = g[c * HW];
            s[0] += bce_elem(kp[(n * 5 + c) * HW + p], gk[c]);
        }
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            float m = gk[c >> 1];
            s[1] += fabsf(sh[(n * 10 + c) * HW + p] - g[(5 + c) * HW]) * inv_r * m;
            s[2] += m;
        }
#pragma unroll
        for (int c = 0; c < 40; ++c) {
            float m = gk[KG_MID_FROM[c >> 1]];
            s[3] += fabsf(md[(n * 40 + c) * HW + p] - g[(15 + c) * HW]) * inv_r * m;
            s[4] += m;
        }
    }
    __shared__ float red[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float v = wave_sum(s[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) part[blockIdx.x * 5 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// out[0..7] = {loss, loss_kp, loss_short, loss_mid, 1/numel, 1/(mask2+1e-10), 1/(mask4+1e-10), 0}
// den_override (optional, device): {mask2_sum, mask4_sum, numel} replacing the local ones (data-parallel
// global normalisation).
__global__ void det_loss_final_kernel(const float* __restrict__ part, int nb, double numel,
                                      const float* __restrict__ den_override, float* __restrict__ out) {
    if (blockIdx.x != 0) return;       // one wave: lanes stride over the block partials, fixed-order tree combine (reproducible)
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nb; b += 64)
        for (int k = 0; k < 5; ++k) s[k] += (double)part[b * 5 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_down(s[k], o, 64);
    if (threadIdx.x != 0) return;
    double m2 = s[2], m4 = s[4], ne = numel;
    if (den_override) { m2 = den_override[0]; m4 = den_override[1]; ne = den_override[2]; }
    float lkp = (float)(s[0] / ne);
    float lsh = (float)s[1] / ((float)m2 + 1e-10f);
    float lmd = (float)s[3] / ((float)m4 + 1e-10f);
    out[0] = lkp + lsh + 0.25f * lmd; out[1] = lkp; out[2] = lsh; out[3] = lmd;
    out[4] = (float)(1.0 / ne); out[5] = 1.f / ((float)m2 + 1e-10f); out[6] = 1.f / ((float)m4 + 1e-10f);
    out[7] = (float)m2;
}
__global__ __launch_bounds__(256) void det_loss_bwd_kernel(const float* __restrict__ kp, const float* __restrict__ sh,
                                                           const float* __restrict__ md, const float* __restrict__ gt,
                                                           int N, long HW, float inv_r, const float* __restrict__ fin,
                                                           const float* __restrict__ gout, float* __restrict__ g_kp,
                                                           float* __restrict__ g_sh, float* __restrict__ g_md) {
    const float go = gout ? gout[0] : 1.f;
    const float a_kp = go * fin[4], a_sh = go * fin[5] * inv_r, a_md = go * 0.25f * fin[6] * inv_r;
    const long total = (long)N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long n = i / HW, p = i - n * HW;
        const float* g = gt + n * 55 * HW + p;
        float gk[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            gk[c] = g[c * HW];
            long o = (n * 5 + c) * HW + p;
            g_kp[o] = bce_grad(kp[o], gk[c]) * a_kp;
        }
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            long o = (n * 10 + c) * HW + p;
            float d = sh[o] - g[(5 + c) * HW];
            float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            g_sh[o] = sg * gk[c >> 1] * a_sh;
        }
#pragma unroll
        for (int c = 0; c < 40; ++c) {
            long o = (n * 40 + c) * HW + p;
            float d = md[o] - g[(15 + c) * HW];
            float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            g_md[o] = sg * gk[KG_MID_FROM[c >> 1]] * a_md;
        }
    }
}

extern "C" int kg_detection_loss_fwd(const float* kp, const float* sh, const float* md, const float* gt, int N, int H,
                                     int W, float kp_radius, const float* den_override, float* scratch,
                                     int scratch_floats, float* out8, void* stream) {
    KG_CHECK_ARG(kp && sh && md && gt && scratch && out8, "kg_detection_loss_fwd: null pointer");
    long total = (long)N * H * W;
    int nb = (int)((total + 255) / 256);
    if (nb > 1024) nb = 1024;
    if (nb > scratch_floats / 5) nb = scratch_floats / 5;
    KG_CHECK_ARG(nb >= 1, "kg_detection_loss_fwd: scratch too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(det_loss_fwd_kernel, dim3(nb), dim3(256), 0, st, kp, sh, md, gt, N, (long)H * W, 1.f / kp_radius, scratch);
    hipLaunchKernelGGL(det_loss_final_kernel, dim3(1), dim3(64), 0, st, scratch, nb, (double)total * 5.0, den_override, out8);
    KG_CHECK_LAUNCH("detection_loss_fwd");
    return KG_OK;
}
extern "C" int kg_detection_loss_bwd(const float* kp, const float* sh, const float* md, const float* gt, int N, int H,
                                     int W, float kp_radius, const float* fin8, const float* grad_out, float* g_kp,
                                     float* g_sh, float* g_md, void* stream) {
    KG_CHECK_ARG(kp && sh && md && gt && fin8 && g_kp && g_sh && g_md, "kg_detection_loss_bwd: null pointer");
    long total = (long)N * H * W;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(det_loss_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, kp, sh, md, gt, N, (long)H * W,
                       1.f / kp_radius, fin8, grad_out, g_kp, g_sh, g_md);
    KG_CHECK_LAUNCH("detection_loss_bwd");
    return KG_OK;
}

// ---------------------------------------------------------------------------------------------
// Mask loss over ragged per-box probability patches.  One block per patch; a patch may be matched
// with several GT boxes ("pairs", seg_loss.py:53-88).  pair weight = 1/(h*w) / num_obj(image) / batch.
struct SegPatch { int prob_off, npix, pair0, npairs; };
struct SegPair { int tgt_off; float weight; };
__global__ __launch_bounds__(256) void seg_loss_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ tgt,
                                                       const SegPatch* __restrict__ patches,
                                                       const SegPair* __restrict__ pairs, float* __restrict__ part,
                                                       const float* __restrict__ gout, float* __restrict__ gprob) {
    const SegPatch pa = patches[blockIdx.x];
    const float go = gout ? gout[0] : 1.f;
    float s = 0.f;
    for (int i = threadIdx.x; i < pa.npix; i += blockDim.x) {
        float p = prob[pa.prob_off + i], g = 0.f;
        for (int k = 0; k < pa.npairs; ++k) {
            SegPair pr = pairs[pa.pair0 + k];
            float t = (float)tgt[pr.tgt_off + i];
            s += pr.weight * bce_elem(p, t);
            g += pr.weight * bce_grad(p, t);
        }
        if (gprob) gprob[pa.prob_off + i] = g * go;
    }
    __shared__ float red[4];
    float v = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0 && part) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sum_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    if (blockIdx.x != 0) return;      // one wave, lanes stride over the partials, fixed-order tree combine (reproducible)
    double s = 0.;
    for (int i = threadIdx.x; i < n; i += 64) s += (double)part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) out[0] = (float)s;
}
extern "C" int kg_seg_loss(const float* prob, const void* tgt, const int* patches, const void* pairs, int npatches,
                           float* part, float* out1, const float* grad_out, float* gprob, void* stream) {
    KG_CHECK_ARG(prob && tgt && patches && pairs && npatches > 0, "kg_seg_loss: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(seg_loss_kernel, dim3(npatches), dim3(256), 0, st, prob, (const uint8_t*)tgt,
                       (const SegPatch*)patches, (const SegPair*)pairs, part, grad_out, gprob);
    if (part && out1) hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, st, part, npatches, out1);
    KG_CHECK_LAUNCH("seg_loss");
    return KG_OK;
}

// ---------------------------------------------------------------------------------------------
// Head plumbing between the fp32 NCHW API tensors and the bf16 pixel-major engine.
// sigmoid of the kp logits (KGnet.py:300) is applied when the maps are exported:
//   prob = 1/(1+exp(-z)) in fp32.
__global__ void sigmoid_inplace_kernel(float* __restrict__ x, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = 1.f / (1.f + expf(-x[i]));
}
extern "C" int kg_sigmoid_inplace(float* x, long n, void* stream) {
    KG_CHECK_ARG(x, "kg_sigmoid_inplace: null pointer");
    if (n == 0) return KG_OK;
    int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(sigmoid_inplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n);
    KG_CHECK_LAUNCH("sigmoid_inplace");
    return KG_OK;
}
// Pack an fp32 NCHW gradient [N][C][HW] into bf16 rows [N*HW][ld] (channels >= C zero-filled up to
// cpad).  If prob != null the gradient is w.r.t. sigmoid output and is multiplied by p*(1-p).
__global__ void grad_pack_kernel(const float* __restrict__ g, const float* __restrict__ prob, bf16_t* __restrict__ out,
                                 int N, int C, long HW, int ld, int cpad, int P, int ps, const float* __restrict__ scale) {
    long total = (long)N * HW;
    const float S = scale ? *scale : 1.f;       // (half build: the power-of-two gradient scale of this step, gradscale.hip)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long n = i / HW, p = i - n * HW;
        for (int c0 = 0; c0 < cpad; c0 += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int c = c0 + e;
                float t = 0.f;
                if (c < C) {
                    long o = (n * C + c) * HW + p;
                    t = g[o];
                    if (prob) { float q = prob[o]; t *= q * (1.f - q); }
                }
                v[e] = t * S;
            }
            kg_store_planes<8>(out + i * ld + c0, P, ps, v, true);
        }
    }
}
// The same through an LDS transpose (cpad <= 64): a workgroup takes 256 consecutive pixels, reads every channel plane coalesced
// (one float per thread), and writes the rows as consecutive 16-byte chunks -- the per-pixel version above stores 16 bytes per 128-byte
// row and thread (2.4 TB/s measured; this one is bound by the reads).
__global__ __launch_bounds__(256) void grad_pack_tr_kernel(const float* __restrict__ g, const float* __restrict__ prob, bf16_t* __restrict__ out,
                                                           long total, int C, long HW, int ld, int cpad, int P, int ps, const float* __restrict__ scale) {
    extern __shared__ float gp_tile[];                       // [cpad][257]
    const float S = scale ? *scale : 1.f;
    const int t = threadIdx.x, K = cpad >> 3;
    for (long i0 = (long)blockIdx.x * 256; i0 < total; i0 += (long)gridDim.x * 256) {
        if ((HW & 3) == 0) {      // 16-byte reads: thread = (channel t / 64 + 4 k, pixels 4 (t % 64) .. + 3); a quad never straddles two images
            const int cs = t >> 6, pq = (t & 63) * 4;
            const long i = i0 + pq;
            if (i < total) {
                const long n = i / HW, p = i - n * HW;
#pragma unroll 2
                for (int c = cs; c < C; c += 4) {
                    const long o = (n * C + c) * HW + p;
                    f32x4 v = *reinterpret_cast<const f32x4*>(g + o);
                    if (prob) {
                        const f32x4 q = *reinterpret_cast<const f32x4*>(prob + o);
                        v[0] *= q[0] * (1.f - q[0]); v[1] *= q[1] * (1.f - q[1]); v[2] *= q[2] * (1.f - q[2]); v[3] *= q[3] * (1.f - q[3]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) gp_tile[c * 257 + pq + e] = v[e] * S;
                }
            }
        } else {
            const long i = i0 + t;
            if (i < total) {
                const long n = i / HW, p = i - n * HW;
                for (int c = 0; c < C; ++c) {
                    const long o = (n * C + c) * HW + p;
                    float v = g[o];
                    if (prob) { const float q = prob[o]; v *= q * (1.f - q); }
                    gp_tile[c * 257 + t] = v * S;
                }
            }
        }
        for (int c = C; c < cpad; ++c) gp_tile[c * 257 + t] = 0.f;
        __syncthreads();
        for (int q = t; q < 256 * K; q += 256) {
            const int px = q / K, k8 = q - px * K;
            if (i0 + px < total) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gp_tile[(k8 * 8 + e) * 257 + px];
                kg_store_planes<8>(out + (i0 + px) * ld + k8 * 8, P, ps, v, true);
            }
        }
        __syncthreads();
    }
}
// Three maps of one pyramid level side by side in ONE pass (the kp | short | mid gradients of the fused second-layer input gradient,
// engine.HEAD_OFF / HEAD_PAD): the same read / transpose / store scheme, the sources dealt to the channel rows off[s] .. off[s] + C[s] of the
// tile, the padding rows zero; writes whole cpad-channel rows instead of three column slices (and is one launch instead of three).
struct GradPack3 { const float* g[3]; const float* prob; int C[3]; int off[3]; int end[3]; };
__global__ __launch_bounds__(256) void grad_pack3_tr_kernel(const GradPack3 a, bf16_t* __restrict__ out, long total, long HW, int ld, int cpad, int P, int ps,
                                                            const float* __restrict__ scale) {
    extern __shared__ float gp_tile[];                       // [cpad][257]
    const float S = scale ? *scale : 1.f;
    const int t = threadIdx.x, K = cpad >> 3;
    for (long i0 = (long)blockIdx.x * 256; i0 < total; i0 += (long)gridDim.x * 256) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            for (int c = a.off[s] + a.C[s]; c < a.end[s]; ++c) gp_tile[c * 257 + t] = 0.f;      // padding rows of the source's column block
            const float* __restrict__ g = a.g[s];
            const float* __restrict__ prob = s == 0 ? a.prob : nullptr;
            const int C = a.C[s], r0 = a.off[s];
            if ((HW & 3) == 0) {
                const int cs = t >> 6, pq = (t & 63) * 4;
                const long i = i0 + pq;
                if (i < total) {
                    const long n = i / HW, p = i - n * HW;
#pragma unroll 2
                    for (int c = cs; c < C; c += 4) {
                        const long o = (n * C + c) * HW + p;
                        f32x4 v = *reinterpret_cast<const f32x4*>(g + o);
                        if (prob) {
                            const f32x4 q = *reinterpret_cast<const f32x4*>(prob + o);
                            v[0] *= q[0] * (1.f - q[0]); v[1] *= q[1] * (1.f - q[1]); v[2] *= q[2] * (1.f - q[2]); v[3] *= q[3] * (1.f - q[3]);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) gp_tile[(r0 + c) * 257 + pq + e] = v[e] * S;
                    }
                }
            } else {
                const long i = i0 + t;
                if (i < total) {
                    const long n = i / HW, p = i - n * HW;
                    for (int c = 0; c < C; ++c) {
                        const long o = (n * C + c) * HW + p;
                        float v = g[o];
                        if (prob) { const float q = prob[o]; v *= q * (1.f - q); }
                        gp_tile[(r0 + c) * 257 + t] = v * S;
                    }
                }
            }
        }
        __syncthreads();
        for (int q = t; q < 256 * K; q += 256) {
            const int px = q / K, k8 = q - px * K;
            if (i0 + px < total) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gp_tile[(k8 * 8 + e) * 257 + px];
                kg_store_planes<8>(out + (i0 + px) * ld + k8 * 8, P, ps, v, true);
            }
        }
        __syncthreads();
    }
}
// planes: y = out.  g0 / g1 / g2: fp32 NCHW with C0 / C1 / C2 channels; out columns [0, pad0) <- g0 (times prob0 (1 - prob0) when prob0 is given),
// [pad0, pad0 + pad1) <- g1, [pad0 + pad1, pad0 + pad1 + pad2) <- g2; pads are multiples of 8, their sum <= 64.
extern "C" int kg_grad_pack3(const float* g0, const float* g1, const float* g2, const float* prob0, void* out, int N, int C0, int C1, int C2, int H, int W,
                             int ld, int pad0, int pad1, int pad2, const kg_planes_t* planes, void* stream) {
    const int cpad = pad0 + pad1 + pad2;
    KG_CHECK_ARG(g0 && g1 && g2 && out && ld % 8 == 0 && pad0 % 8 == 0 && pad1 % 8 == 0 && pad2 % 8 == 0 && pad0 >= C0 && pad1 >= C1 && pad2 >= C2 &&
                 cpad <= 64 && C0 > 0 && C1 > 0 && C2 > 0, "kg_grad_pack3: bad args");
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_grad_pack3: bad kg_planes_t");
    long total = (long)N * H * W;
    if (total == 0) return KG_OK;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    static KgPerDevice attr_done;
    if (attr_done.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)grad_pack3_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 257 * 4));
    }
    GradPack3 a;
    a.g[0] = g0; a.g[1] = g1; a.g[2] = g2; a.prob = prob0;
    a.C[0] = C0; a.C[1] = C1; a.C[2] = C2; a.off[0] = 0; a.off[1] = pad0; a.off[2] = pad0 + pad1;
    a.end[0] = pad0; a.end[1] = pad0 + pad1; a.end[2] = cpad;
    hipLaunchKernelGGL(grad_pack3_tr_kernel, dim3(blocks), dim3(256), cpad * 257 * 4, (hipStream_t)stream, a, (bf16_t*)out, total, (long)H * W, ld, cpad,
                       pp.y_planes, pp.y_pstride, pp.scale);
    KG_CHECK_LAUNCH("grad_pack3_tr");
    return KG_OK;
}
// planes: y = out (split-bf16 planes of the packed gradient rows)
extern "C" int kg_grad_pack(const float* g, const float* prob, void* out, int N, int C, int H, int W, int ld, int cpad,
                            const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(g && out && cpad % 8 == 0 && ld % 8 == 0 && cpad >= C, "kg_grad_pack: bad args");
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_grad_pack: bad kg_planes_t");
    long total = (long)N * H * W;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    if (cpad <= 64) {
        const int smem = cpad * 257 * 4;
        static KgPerDevice attr_done;
        if (attr_done.first()) {
            KG_HIP(hipFuncSetAttribute((const void*)grad_pack_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 257 * 4));
        }
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(grad_pack_tr_kernel, dim3(blocks), dim3(256), smem, (hipStream_t)stream, g, prob, (bf16_t*)out, total, C, (long)H * W, ld, cpad,
                           pp.y_planes, pp.y_pstride, pp.scale);
        KG_CHECK_LAUNCH("grad_pack_tr");
        return KG_OK;
    }
    hipLaunchKernelGGL(grad_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, prob, (bf16_t*)out, N, C,
                       (long)H * W, ld, cpad, pp.y_planes, pp.y_pstride, pp.scale);
    KG_CHECK_LAUNCH("grad_pack");
    return KG_OK;
}

// norm_pool.hip -- HBM-bound layer kernels of KGnet's backbone/decoder for gfx950:
//   image pack (fp32 NCHW -> bf16 NHWC8), BatchNorm2d (train/eval, forward/backward),
//   MaxPool 3x3 s2 p1, bilinear resize (align_corners=False), elementwise helpers.
// Reference ops replaced: KGnet.py:131-134 (bn1/relu/maxpool), :64-99 (Bottleneck BN/ReLU/add),
// :288-297 (F.interpolate bilinear).  All activations are bf16 pixel-major rows [row][ld]; every
// thread moves 16-byte channel chunks (8 bf16) so a wave covers whole 128-byte lines.
#include "kg_common.h"

// Rows operands with split-bf16 planes (kg_common.h): value = sum of P bf16 planes, `ps` elements apart.
struct RowsR { const bf16_t* p; int ld, P, ps; };
struct RowsW { bf16_t* p; int ld, P, ps; };
__device__ __forceinline__ void rd8(const RowsR& t, long r, int c, float (&v)[8]) { kg_load_planes8(t.p + r * t.ld + c, t.P, t.ps, v); }
__device__ __forceinline__ void wr8(const RowsW& t, long r, int c, float (&v)[8]) { kg_store_planes<8>(t.p + r * t.ld + c, t.P, t.ps, v, true); }
#define KG_PLANES(pl)                                                       \
    const kg_planes_t pp = kg_planes_or_default(pl);                        \
    KG_CHECK_ARG(kg_planes_ok(pp), "bad kg_planes_t (at most 3 planes, strides multiples of 8)")

// ---------------------------------------------------------------------------------------------
__global__ void img_pack_kernel(const float* __restrict__ img, RowsW out, int N, int C, int HW) {
    long total = (long)N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long n = i / HW, p = i - n * HW;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < C ? img[(n * C + c) * HW + p] : 0.f;
        wr8(out, i, 0, v);
    }
}
// planes: y = the packed rows [N*H*W][ldout] (8 channels per plane, plane stride y_pstride)
extern "C" int kg_img_pack(const float* img, void* out, int ldout, int N, int C, int H, int W, const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(img && out && C <= 8 && ldout >= 8 && ldout % 8 == 0, "kg_img_pack: bad args");
    KG_PLANES(planes);
    long total = (long)N * H * W;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(img_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, RowsW{(bf16_t*)out, ldout, pp.y_planes, pp.y_pstride}, N, C, H * W);
    KG_CHECK_LAUNCH("img_pack");
    return KG_OK;
}

// ---------------------------------------------------------------------------------------------
// Column reductions over bf16 rows.  Block (256 threads) = [32 row lanes][8 chunk lanes]; each
// block handles a 64-channel slab and a contiguous row range; partials [nb][C][NQ] fp32 are
// combined in double by the finalize kernels (fixed order => reproducible).
// MODE 0: (sum x, sum x^2)      MODE 1: (sum dy, sum dy*xhat) with xhat=(x-mean)*invstd
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const RowsR x, const RowsR dy,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ part,
                                                        int M, int C, int rows_per_block) {
    __shared__ float red[32][8][16];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = blockIdx.y * 64 + cl * 8;
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
    if (c0 < C) {
        float mu[8], is[8];
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; }
        }
        int r0 = blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
        if (r1 > M) r1 = M;
        for (int r = r0 + rl; r < r1; r += 32) {
            float xs[8];
            rd8(x, r, c0, xs);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { float f = xs[e]; s0[e] += f; s1[e] += f * f; }
            } else {
                float ds[8];
                rd8(dy, r, c0, ds);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float d = ds[e], xh = (xs[e] - mu[e]) * is[e];
                    s0[e] += d; s1[e] += d * xh;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[rl][cl][e] = s0[e]; red[rl][cl][8 + e] = s1[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x & 63, q = threadIdx.x >> 6;  // q: which sum
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[k][c >> 3][q * 8 + (c & 7)];
        const int cg = blockIdx.y * 64 + c;
        if (cg < C) part[((long)blockIdx.x * C + cg) * 2 + q] = t;
    }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return __shfl(v, 0, 64);
}
// The finalize kernels run ONE WAVE PER CHANNEL: lanes stride over the block partials (fixed order => reproducible).

// train-mode finalize: mean / biased var -> invstd, scale/shift, running stats (momentum 0.1,
// unbiased var), matching torch.nn.functional.batch_norm(training=True) (KGnet.py:82-93).
__global__ void bn_finalize_train_kernel(const float* __restrict__ part, int nb, int C, long M,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                         float momentum, float eps, float* __restrict__ mean_out,
                                         float* __restrict__ invstd_out, float* __restrict__ scale,
                                         float* __restrict__ shift) {
    const int c = blockIdx.x;
    double s = 0., ss = 0.;
    for (int b = threadIdx.x; b < nb; b += 64) { s += part[((long)b * C + c) * 2]; ss += part[((long)b * C + c) * 2 + 1]; }
    s = wave_sum_d(s); ss = wave_sum_d(ss);
    if (threadIdx.x != 0) return;
    double mu = s / (double)M;
    double var = ss / (double)M - mu * mu;
    if (var < 0.) var = 0.;
    float is = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[c] = (float)mu; invstd_out[c] = is;
    float sc = gamma[c] * is;
    scale[c] = sc; shift[c] = beta[c] - (float)mu * sc;
    if (running_mean) {
        double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
__global__ void bn_finalize_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ running_mean,
                                        const float* __restrict__ running_var, float eps,
                                        float* __restrict__ scale, float* __restrict__ shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float is = 1.f / sqrtf(running_var[c] + eps);
    float sc = gamma[c] * is;
    scale[c] = sc; shift[c] = beta[c] - running_mean[c] * sc;
}
// backward finalize: dgamma, dbeta and the per-channel coefficients of
//   dx = a*dy + b*xhat + c0   with a = gamma*invstd, b = -a*dgamma/M, c0 = -a*dbeta/M
// rs (optional device scalar): the partials were taken by the producing input gradient's epilogue (KgBStat) BEFORE the gradient tensor was
// re-normalised by *rs (a power of two, kg_rows_rescale): the sums are multiplied alike
__global__ void bn_finalize_bwd_kernel(const float* __restrict__ part, int nb, int C, long M,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                       float* __restrict__ coef, const float* __restrict__ rs) {
    const int c = blockIdx.x;
    double s = 0., sx = 0.;
    for (int b = threadIdx.x; b < nb; b += 64) { s += part[((long)b * C + c) * 2]; sx += part[((long)b * C + c) * 2 + 1]; }
    s = wave_sum_d(s); sx = wave_sum_d(sx);
    if (threadIdx.x != 0) return;
    if (rs) { s *= (double)*rs; sx *= (double)*rs; }
    float db = (float)s, dg = (float)sx;
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
    dbeta[c] = accumulate ? dbeta[c] + db : db;
    float a = gamma[c] * invstd[c];
    coef[c] = a; coef[C + c] = (float)(-(double)a * sx / (double)M); coef[2 * C + c] = (float)(-(double)a * s / (double)M);
}

static int reduce_geometry(int M, int C, int scratch_floats, int* nb, int* rpb) {
    int n = scratch_floats / (2 * C);
    if (n > 512) n = 512;
    int need = (M + 255) / 256;
    if (n > need) n = need;
    if (n < 1) return 0;
    *rpb = (M + n - 1) / n;
    *nb = (M + *rpb - 1) / *rpb;
    return 1;
}

// planes: a = x
extern "C" int kg_bn_stats_train(const void* x, int ldx, int M, int C, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps,
                                 float* mean_out, float* invstd_out, float* scale, float* shift, float* scratch,
                                 int scratch_floats, const kg_planes_t* planes, void* stream) {
    KG_PLANES(planes);
    KG_CHECK_ARG(x && gamma && beta && mean_out && invstd_out && scale && shift && scratch, "kg_bn_stats_train: null pointer");
    KG_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0, "kg_bn_stats_train: C/ld must be multiples of 8");
    int nb, rpb;
    KG_CHECK_ARG(reduce_geometry(M, C, scratch_floats, &nb, &rpb), "kg_bn_stats_train: scratch too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colreduce_kernel<0>, dim3(nb, (C + 63) / 64), dim3(256), 0, st, RowsR{(const bf16_t*)x, ldx, pp.a_planes, pp.a_pstride},
                       RowsR{nullptr, 0, 1, 0}, (const float*)nullptr, (const float*)nullptr, scratch, M, C, rpb);
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(C), dim3(64), 0, st, scratch, nb, C, (long)M, gamma,
                       beta, running_mean, running_var, momentum, eps, mean_out, invstd_out, scale, shift);
    KG_CHECK_LAUNCH("bn_stats_train");
    return KG_OK;
}
// The second stage of kg_bn_stats_train alone, over partials [nb][C][2] that the producing conv's epilogue wrote (kg_conv_stats_begin /
// kg_conv_stats_end, conv_args.h): same fixed-order double combine, same running-statistics update.
extern "C" int kg_bn_finalize_train(const float* part, int nb, int M, int C, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float eps, float* mean_out, float* invstd_out, float* scale,
                                    float* shift, void* stream) {
    KG_CHECK_ARG(part && gamma && beta && mean_out && invstd_out && scale && shift && nb >= 1 && M >= 1 && C >= 1, "kg_bn_finalize_train: bad arguments");
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, part, nb, C, (long)M, gamma, beta, running_mean,
                       running_var, momentum, eps, mean_out, invstd_out, scale, shift);
    KG_CHECK_LAUNCH("bn_finalize_train");
    return KG_OK;
}
extern "C" int kg_bn_scale_shift_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, float eps, float* scale, float* shift, void* stream) {
    KG_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift, "kg_bn_scale_shift_eval: null pointer");
    hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    KG_CHECK_LAUNCH("bn_scale_shift_eval");
    return KG_OK;
}

// y = [relu]( x*scale + shift [+ res] )
__global__ void bn_apply_kernel(const RowsR x, const float* __restrict__ scale,
                                const float* __restrict__ shift, const RowsR res,
                                const RowsW y, long M, int C8, int relu) {
    long total = M * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, C8, &r, &c); c *= 8;
        float v[8];
        rd8(x, r, c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * scale[c + e] + shift[c + e];
        if (res.p) {
            float rs[8];
            rd8(res, r, c, rs);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rs[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = kg_relu(v[e]);
        }
        wr8(y, r, c, v);
    }
}
// planes: a = x, b = res, y = y
extern "C" int kg_bn_apply(const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldres,
                           void* y, int ldy, int M, int C, int relu, const kg_planes_t* planes, void* stream) {
    KG_PLANES(planes);
    KG_CHECK_ARG(x && scale && shift && y, "kg_bn_apply: null pointer");
    KG_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (!res || ldres % 8 == 0), "kg_bn_apply: C/ld must be multiples of 8");
    long total = (long)M * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, RowsR{(const bf16_t*)x, ldx, pp.a_planes, pp.a_pstride}, scale,
                       shift, RowsR{(const bf16_t*)res, ldres, pp.b_planes, pp.b_pstride}, RowsW{(bf16_t*)y, ldy, pp.y_planes, pp.y_pstride}, (long)M, C / 8, relu);
    KG_CHECK_LAUNCH("bn_apply");
    return KG_OK;
}

// dx = coef_a*dy + coef_b*xhat + coef_c   (train-mode BN backward, second pass)
__global__ void bn_bwd_apply_kernel(const RowsR x, const RowsR dy,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ coef, const RowsW dx, long M, int C) {
    const int C8 = C / 8;
    long total = M * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, C8, &r, &c); c *= 8;
        float xs[8], ds[8], v[8];
        rd8(x, r, c, xs); rd8(dy, r, c, ds);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float xh = (xs[e] - mean[c + e]) * invstd[c + e];
            v[e] = coef[c + e] * ds[e] + coef[C + c + e] * xh + coef[2 * C + c + e];
        }
        wr8(dx, r, c, v);
    }
}
// planes: a = x, b = dy, y = dx
// parts / nb_parts / parts_scale (optional): partials [nb_parts][C][2] of (sum dy, sum dy * xhat) that the input gradient which produced dy
// wrote in its epilogue (kg_conv_bstats_begin) -- the column reduction over x and dy is skipped; parts_scale: device scalar dy was multiplied
// by after the partials were taken (kg_rows_rescale), or NULL
extern "C" int kg_bn_bwd(const void* x, int ldx, const void* dy, int lddy, const float* gamma, const float* mean,
                         const float* invstd, float* dgamma, float* dbeta, int accumulate, void* dx, int lddx, int M,
                         int C, float* scratch, int scratch_floats, const float* parts, int nb_parts, const float* parts_scale,
                         const kg_planes_t* planes, void* stream) {
    KG_PLANES(planes);
    const RowsR xr{(const bf16_t*)x, ldx, pp.a_planes, pp.a_pstride}, dyr{(const bf16_t*)dy, lddy, pp.b_planes, pp.b_pstride};
    KG_CHECK_ARG(x && dy && gamma && mean && invstd && dgamma && dbeta && dx && scratch, "kg_bn_bwd: null pointer");
    KG_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "kg_bn_bwd: C/ld must be multiples of 8");
    int nb, rpb;
    hipStream_t st = (hipStream_t)stream;
    float* coef = scratch;
    if (parts && nb_parts > 0) {
        KG_CHECK_ARG(scratch_floats >= 3 * C, "kg_bn_bwd: scratch too small");
        hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(64), 0, st, parts, nb_parts, C, (long)M, gamma, invstd, dgamma, dbeta, accumulate, coef,
                           parts_scale);
    } else {
        KG_CHECK_ARG(reduce_geometry(M, C, scratch_floats - 3 * C, &nb, &rpb), "kg_bn_bwd: scratch too small");
        float* part = scratch + 3 * C;
        hipLaunchKernelGGL(colreduce_kernel<1>, dim3(nb, (C + 63) / 64), dim3(256), 0, st, xr, dyr, mean, invstd, part, M, C, rpb);
        hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(C), dim3(64), 0, st, part, nb, C, (long)M, gamma,
                           invstd, dgamma, dbeta, accumulate, coef, (const float*)nullptr);
    }
    long total = (long)M * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, xr, dyr,
                       mean, invstd, coef, RowsW{(bf16_t*)dx, lddx, pp.y_planes, pp.y_pstride}, (long)M, C);
    KG_CHECK_LAUNCH("bn_bwd");
    return KG_OK;
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, pad 1) (KGnet.py:134).  First maximum in (kh,kw) scan order wins ties,
// as torch's max_pool2d does.
__device__ __forceinline__ void pool_window_max(const RowsR& x, long nbase, int H, int W,
                                                int oy, int ox, int c, float* best, int* arg) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = -1; }
    for (int kh = 0; kh < 3; ++kh) {
        int iy = oy * 2 - 1 + kh;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            int ix = ox * 2 - 1 + kw;
            if ((unsigned)ix >= (unsigned)W) continue;
            float s[8];
            rd8(x, nbase + (long)iy * W + ix, c, s);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = s[e];
                if (f > best[e] || arg[e] < 0) { best[e] = f; arg[e] = kh * 3 + kw; }
            }
        }
    }
}
__global__ void maxpool_fwd_kernel(const RowsR x, const RowsW y, unsigned char* __restrict__ argout, int N, int H,
                                   int W, int OH, int OW, int C8) {
    long total = (long)N * OH * OW * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p; int c; kg_divmod(i, C8, &p, &c); c *= 8;
        int ox, oy; long q, n; kg_divmod(p, OW, &q, &ox); kg_divmod(q, OH, &n, &oy);
        float best[8]; int arg[8];
        pool_window_max(x, n * H * W, H, W, oy, ox, c, best, arg);
        wr8(y, p, c, best);
        if (argout) {      // winning tap (kh*3+kw) per output element: the backward pass reads it instead of re-scanning the windows
            uint2 pk;
            pk.x = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
            pk.y = (unsigned)arg[4] | ((unsigned)arg[5] << 8) | ((unsigned)arg[6] << 16) | ((unsigned)arg[7] << 24);
            *reinterpret_cast<uint2*>(argout + p * (long)(C8 * 8) + c) = pk;
        }
    }
}
// gather-form backward: each input pixel sums dy of the (<=4) windows whose argmax it is.
__global__ void maxpool_bwd_kernel(const RowsR x, const RowsR dy, const unsigned char* __restrict__ argin,
                                   const RowsW dx, int N, int H, int W, int OH, int OW, int C8) {
    long total = (long)N * H * W * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p; int c; kg_divmod(i, C8, &p, &c); c *= 8;
        int ix, iy; long q, n; kg_divmod(p, W, &q, &ix); kg_divmod(q, H, &n, &iy);
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        // windows (oy,ox) with oy*2-1+kh == iy  =>  oy in {(iy+1)/2 (kh=...)}
        for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy) {
            if (oy >= OH) continue;
            int kh = iy - (oy * 2 - 1);
            if (kh < 0 || kh > 2) continue;
            for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= OW) continue;
                int kw = ix - (ox * 2 - 1);
                if (kw < 0 || kw > 2) continue;
                float best[8]; int arg[8];
                if (argin) {
                    const uint2 pk = *reinterpret_cast<const uint2*>(argin + ((n * OH + oy) * OW + ox) * (long)(C8 * 8) + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { arg[e] = (pk.x >> (8 * e)) & 255; arg[4 + e] = (pk.y >> (8 * e)) & 255; }
                } else
                    pool_window_max(x, n * H * W, H, W, oy, ox, c, best, arg);
                float ds[8];
                rd8(dy, (n * OH + oy) * OW + ox, c, ds);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (arg[e] == kh * 3 + kw) g[e] += ds[e];
            }
        }
        wr8(dx, p, c, g);
    }
}
// planes: a = x, y = y
// argmax (optional): bytes [N*OH*OW][C], the winning tap of every output element, for kg_maxpool3s2_bwd
extern "C" int kg_maxpool3s2_fwd(const void* x, int ldx, void* y, int ldy, void* argmax, int N, int H, int W, int C, const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(x && y && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "kg_maxpool3s2_fwd: bad args");
    KG_PLANES(planes);
    int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    long total = (long)N * OH * OW * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, RowsR{(const bf16_t*)x, ldx, pp.a_planes, pp.a_pstride},
                       RowsW{(bf16_t*)y, ldy, pp.y_planes, pp.y_pstride}, (unsigned char*)argmax, N, H, W, OH, OW, C / 8);
    KG_CHECK_LAUNCH("maxpool_fwd");
    return KG_OK;
}
// planes: a = x, b = dy, y = dx
extern "C" int kg_maxpool3s2_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, const void* argmax, int N, int H,
                                 int W, int C, const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(x && dy && dx && C % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "kg_maxpool3s2_bwd: bad args");
    KG_PLANES(planes);
    int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    long total = (long)N * H * W * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, RowsR{(const bf16_t*)x, ldx, pp.a_planes, pp.a_pstride},
                       RowsR{(const bf16_t*)dy, lddy, pp.b_planes, pp.b_pstride}, (const unsigned char*)argmax, RowsW{(bf16_t*)dx, lddx, pp.y_planes, pp.y_pstride}, N, H, W, OH, OW, C / 8);
    KG_CHECK_LAUNCH("maxpool_bwd");
    return KG_OK;
}

// ---------------------------------------------------------------------------------------------
// Bilinear resize, align_corners=False (F.interpolate, KGnet.py:110, 288-297):
//   src = max(scale*(dst+0.5)-0.5, 0), scale = in/out (fp32), i0=(int)src, i1=min(i0+1,in-1).
__device__ __forceinline__ void bil_src(int dst, float scale, int in, int* i0, int* i1, float* l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int a = (int)s;
    if (a > in - 1) a = in - 1;
    *i0 = a; *i1 = a + (a < in - 1 ? 1 : 0); *l1 = s - (float)a;
}
// the weighted sum of the four neighbours as ONE fixed sequence of roundings (left to the compiler, the contraction of the four products into FMAs
// comes out differently in different kernels: the exact-2x kernel below would differ from the generic one in the last bit)
__device__ __forceinline__ float bil_mix(float w00, float w01, float w10, float w11, float a, float b, float d, float e) {
    return __builtin_fmaf(w11, e, __builtin_fmaf(w10, d, __builtin_fmaf(w01, b, w00 * a)));
}
// Dense mode: images [N][IH][IW] -> [N][OH][OW].
// Ragged mode (desc != null): one box per "image"; desc[b] = {in_row0, ih, iw, out_row0, oh, ow} and
// rows are box-local raster order (used by the per-box seg branch, KGnet.py:258-267).
struct BilBox { int in_row0, ih, iw, out_row0, oh, ow; };
__global__ void bilinear_fwd_kernel(const RowsR x, const RowsW y, int IH,
                                    int IW, int OH, int OW, int C8, long total_rows, const BilBox* __restrict__ desc,
                                    const int* __restrict__ row2box) {
    long total = total_rows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p; int c; kg_divmod(i, C8, &p, &c); c *= 8;
        long in0; int ih, iw, oh, ow, oy, ox;
        if (desc) {
            BilBox b = desc[row2box[p]];
            int loc = (int)(p - b.out_row0);
            ih = b.ih; iw = b.iw; oh = b.oh; ow = b.ow; oy = loc / ow; ox = loc - oy * ow; in0 = b.in_row0;
        } else {
            ih = IH; iw = IW; oh = OH; ow = OW;
            long q, nimg; kg_divmod(p, OW, &q, &ox); kg_divmod(q, OH, &nimg, &oy); in0 = nimg * IH * IW;
        }
        int y0, y1, x0, x1; float ly, lx;
        bil_src(oy, (float)ih / (float)oh, ih, &y0, &y1, &ly);
        bil_src(ox, (float)iw / (float)ow, iw, &x0, &x1, &lx);
        float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        float pa[8], pb[8], pd[8], pe[8];
        rd8(x, in0 + (long)y0 * iw + x0, c, pa); rd8(x, in0 + (long)y0 * iw + x1, c, pb);
        rd8(x, in0 + (long)y1 * iw + x0, c, pd); rd8(x, in0 + (long)y1 * iw + x1, c, pe);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bil_mix(w00, w01, w10, w11, pa[e], pb[e], pd[e], pe[e]);
        wr8(y, p, c, v);
    }
}
// Dense exact-2x upsampling (every decoder level, KGnet.py:288-298): the generic kernel issues 4 loads per plane for every 16-byte output chunk and
// the texture path, not HBM, bounds it (tools/bilinear_probe.py with the loads / the stores switched off: 247 us as it is, 88 us without the loads,
// 123 us without the stores at 8 x 256^2 -> 512^2 x 64 channels).  Here a thread owns one 8-channel chunk of one INPUT column and walks down a strip
// of input rows with the 3 x 3 neighbourhood in registers: 3 loads per plane bring what 2 x 2 outputs need (0.75 - 0.94 loads per output chunk
// instead of 4).  The same bil_src indices and the same weighted sum as the generic kernel, term for term: bit-identical output.
constexpr int BIL2_ROWS = 8;     // input rows per thread (measured: 4 the same, 16 and a 128-register budget slower -- tools/bilinear_probe.py)
__global__ __launch_bounds__(256) void bilinear2x_fwd_kernel(const RowsR x, const RowsW y, int N, int IH, int IW, int C8) {
    const long per_img = (long)IW * C8;
    const int strips = (IH + BIL2_ROWS - 1) / BIL2_ROWS;
    const long total = (long)N * strips * per_img;
    const int OH = 2 * IH, OW = 2 * IW;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long q; int c; kg_divmod(t, C8, &q, &c); c *= 8;
        int j; long q2; kg_divmod(q, IW, &q2, &j);
        int strip; long n; kg_divmod(q2, strips, &n, &strip);
        const int i0 = strip * BIL2_ROWS, i1 = i0 + BIL2_ROWS < IH ? i0 + BIL2_ROWS : IH;
        const long in0 = n * IH * IW, out0 = n * OH * OW;
        const int jl = j > 0 ? j - 1 : 0, jr = j < IW - 1 ? j + 1 : IW - 1;
        // bil_src for scale 0.5: output 2k reads inputs (k - 1, k) with l = 0.75 -- (0, 1) with l = 0 for k = 0, where the second input has weight 0 and
        // input 0 may stand in for it --, output 2k + 1 reads (k, min(k + 1, in - 1)) with l = 0.25: all indices static, the edge cases are weights
        const float lx0 = j == 0 ? 0.f : 0.75f;
        float prev[3][8], cur[3][8], next[3][8];               // [column: jl, j, jr]
        {
            const long bp = in0 + (long)(i0 > 0 ? i0 - 1 : 0) * IW, bc = in0 + (long)i0 * IW;
            rd8(x, bp + jl, c, prev[0]); rd8(x, bp + j, c, prev[1]); rd8(x, bp + jr, c, prev[2]);
            rd8(x, bc + jl, c, cur[0]); rd8(x, bc + j, c, cur[1]); rd8(x, bc + jr, c, cur[2]);
        }
        auto mix = [&](const float (&pa)[8], const float (&pb)[8], const float (&pd)[8], const float (&pe)[8], float ly, float lx, long orow) {
            const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bil_mix(w00, w01, w10, w11, pa[e], pb[e], pd[e], pe[e]);
            wr8(y, orow, c, v);
        };
        for (int i = i0; i < i1; ++i) {
            const long bn = in0 + (long)(i < IH - 1 ? i + 1 : IH - 1) * IW;
            rd8(x, bn + jl, c, next[0]); rd8(x, bn + j, c, next[1]); rd8(x, bn + jr, c, next[2]);
            const float ly0 = i == 0 ? 0.f : 0.75f;
            const long o0 = out0 + (long)(2 * i) * OW + 2 * j, o1 = o0 + OW;
            mix(prev[0], prev[1], cur[0], cur[1], ly0, lx0, o0);
            mix(prev[1], prev[2], cur[1], cur[2], ly0, 0.25f, o0 + 1);
            mix(cur[0], cur[1], next[0], next[1], 0.25f, lx0, o1);
            mix(cur[1], cur[2], next[1], next[2], 0.25f, 0.25f, o1 + 1);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) { prev[k][e] = cur[k][e]; cur[k][e] = next[k][e]; }
        }
    }
}
// gather-form backward: each input pixel scans the output pixels that can reference it.
__device__ __forceinline__ void bil_cand(int i, float scale, int out, int* lo, int* hi) {
    // outputs o with src(o) in (i-1, i+1):  o in ((i-0.5)/scale-0.5, (i+1.5)/scale-0.5); widen by one.
    float a = ((float)i - 0.5f) / scale - 0.5f, b = ((float)i + 1.5f) / scale - 0.5f;
    int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
    *lo = l < 0 ? 0 : l; *hi = h > out - 1 ? out - 1 : h;
}
__global__ void bilinear_bwd_kernel(const RowsR dy, const RowsW dx, int IH,
                                    int IW, int OH, int OW, int C8, long total_rows, const BilBox* __restrict__ desc,
                                    const int* __restrict__ row2box, const bf16_t* __restrict__ mask, int ldmask) {
    long total = total_rows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p; int c; kg_divmod(i, C8, &p, &c); c *= 8;
        long out0; int ih, iw, oh, ow, iy, ix;
        if (desc) {
            BilBox b = desc[row2box[p]];
            int loc = (int)(p - b.in_row0);
            ih = b.ih; iw = b.iw; oh = b.oh; ow = b.ow; iy = loc / iw; ix = loc - iy * iw; out0 = b.out_row0;
        } else {
            ih = IH; iw = IW; oh = OH; ow = OW;
            long q, nimg; kg_divmod(p, IW, &q, &ix); kg_divmod(q, IH, &nimg, &iy); out0 = nimg * OH * OW;
        }
        const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
        int ylo, yhi, xlo, xhi;
        bil_cand(iy, sy, oh, &ylo, &yhi); bil_cand(ix, sx, ow, &xlo, &xhi);
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            int y0, y1; float ly;
            bil_src(oy, sy, ih, &y0, &y1, &ly);
            float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (y0 != iy && y1 != iy) continue;
            for (int ox = xlo; ox <= xhi; ++ox) {
                int x0, x1; float lx;
                bil_src(ox, sx, iw, &x0, &x1, &lx);
                if (x0 != ix && x1 != ix) continue;
                float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                float w = wy * wx;
                float ds[8];
                rd8(dy, out0 + (long)oy * ow + ox, c, ds);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += w * ds[e];
            }
        }
        if (mask) {                      // ReLU backward of the tensor that was upsampled (zero where its forward value was <= 0)
            const uint4 mv = *reinterpret_cast<const uint4*>(mask + p * ldmask + c);
            const bf16_t* ms = reinterpret_cast<const bf16_t*>(&mv);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = bf2f(ms[e]) > 0.f ? g[e] : 0.f;
        }
        wr8(dx, p, c, g);
    }
}
// planes: a = x, y = y
extern "C" int kg_bilinear_fwd(const void* x, int ldx, void* y, int ldy, int N, int IH, int IW, int OH, int OW, int C,
                               const int* boxdesc, const int* row2box, long total_out_rows, const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(x && y && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "kg_bilinear_fwd: bad args");
    KG_PLANES(planes);
    long rows = boxdesc ? total_out_rows : (long)N * OH * OW;
    if (rows == 0) return KG_OK;
    long total = rows * (C / 8);
    static const int fast2x = getenv("KG_BILINEAR_2X") ? atoi(getenv("KG_BILINEAR_2X")) : 1;
    if (fast2x && !boxdesc && OH == 2 * IH && OW == 2 * IW && IH >= 2 && IW >= 2) {
        const long threads = (long)N * ((IH + BIL2_ROWS - 1) / BIL2_ROWS) * IW * (C / 8);
        int blocks2 = (int)((threads + 255) / 256); if (blocks2 > 65536) blocks2 = 65536;
        hipLaunchKernelGGL(bilinear2x_fwd_kernel, dim3(blocks2), dim3(256), 0, (hipStream_t)stream, RowsR{(const bf16_t*)x, ldx, pp.a_planes, pp.a_pstride},
                           RowsW{(bf16_t*)y, ldy, pp.y_planes, pp.y_pstride}, N, IH, IW, C / 8);
        KG_CHECK_LAUNCH("bilinear2x_fwd");
        return KG_OK;
    }
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, RowsR{(const bf16_t*)x, ldx, pp.a_planes, pp.a_pstride},
                       RowsW{(bf16_t*)y, ldy, pp.y_planes, pp.y_pstride}, IH, IW, OH, OW, C / 8, rows, (const BilBox*)boxdesc, row2box);
    KG_CHECK_LAUNCH("bilinear_fwd");
    return KG_OK;
}
// planes: a = dy, y = dx (mask: plane 0 of the forward tensor carries its sign)
extern "C" int kg_bilinear_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int IH, int IW, int OH, int OW, int C,
                               const int* boxdesc, const int* row2box, long total_in_rows, const void* mask, int ldmask,
                               const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(dy && dx && C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && (!mask || ldmask % 8 == 0), "kg_bilinear_bwd: bad args");
    KG_PLANES(planes);
    long rows = boxdesc ? total_in_rows : (long)N * IH * IW;
    if (rows == 0) return KG_OK;
    long total = rows * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, RowsR{(const bf16_t*)dy, lddy, pp.a_planes, pp.a_pstride},
                       RowsW{(bf16_t*)dx, lddx, pp.y_planes, pp.y_pstride}, IH, IW, OH, OW, C / 8, rows, (const BilBox*)boxdesc, row2box, (const bf16_t*)mask, ldmask);
    KG_CHECK_LAUNCH("bilinear_bwd");
    return KG_OK;
}

// ---------------------------------------------------------------------------------------------
// out[row][c] = a[row][c] + b[row][c] (+ optional ReLU mask by m > 0); used for gradient joins.
__global__ void add_rows_kernel(const RowsR a, const RowsR b,
                                const bf16_t* __restrict__ m, int ldm, const RowsW y, long M, int C8, const float* __restrict__ sc,
                                const float* __restrict__ sc2) {
    long total = M * C8;
    const float s = sc ? *sc * (sc2 ? *sc2 : 1.f) : 1.f;      // (a power of two: the conversion of both operands into the running gradient scale)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, C8, &r, &c); c *= 8;
        float v[8];
        rd8(a, r, c, v);
        if (b.p) {
            float bs[8];
            rd8(b, r, c, bs);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bs[e];
        }
        if (s != 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= s;
        }
        if (m) {
            uint4 mv = *reinterpret_cast<const uint4*>(m + r * ldm + c);
            const bf16_t* ms = (const bf16_t*)&mv;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(ms[e]) > 0.f ? v[e] : 0.f;
        }
        wr8(y, r, c, v);
    }
}
// planes: a = a, b = b, y = y (mask: plane 0)
// scale / scale2 (optional device scalars): y = (a + b) * *scale (* *scale2) -- the half-precision backward's conversion of a gradient
// written under an earlier running scale (kg_rows_scale), folded into the join that reads it anyway
extern "C" int kg_add_rows(const void* a, int lda, const void* b, int ldb, const void* mask, int ldm, void* y, int ldy,
                           long M, int C, const float* scale, const float* scale2, const kg_planes_t* planes, void* stream) {
    KG_PLANES(planes);
    KG_CHECK_ARG(a && y && C % 8 == 0 && lda % 8 == 0 && ldy % 8 == 0 && (!b || ldb % 8 == 0) && (!mask || ldm % 8 == 0), "kg_add_rows: bad args");
    if (M == 0) return KG_OK;
    long total = M * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(add_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, RowsR{(const bf16_t*)a, lda, pp.a_planes, pp.a_pstride},
                       RowsR{(const bf16_t*)b, ldb, pp.b_planes, pp.b_pstride}, (const bf16_t*)mask, ldm, RowsW{(bf16_t*)y, ldy, pp.y_planes, pp.y_pstride}, M, C / 8, scale, scale2);
    KG_CHECK_LAUNCH("add_rows");
    return KG_OK;
}

// ---- re-normalisation points of the half-precision backward pass ---------------------------------------------------------------
// Gradients grow ~2^0.9 per bottleneck through the BatchNorm backbone at random init (gamma / sigma), x316 = 1 / sqrt(eps) through
// the BatchNorm of a (nearly) dead channel, x4 per level through the adjoint of the 2x bilinear upsampling when they are spatially
// coherent (tools/gradmax_probe.py) -- more than one global power-of-two scale can place inside IEEE half's range together with
// the 1e-7-sized loss gradients of the heads.  The backward pass therefore re-normalises itself (engine.renormalise) where the
// gradient of a decoder level output, of c1, of every bottleneck output or of a seg-branch level is complete:
// kg_rows_rescale measures max |g| of that rows tensor on the device and, when it has grown beyond 2^T, multiplies the tensor in
// place by the power of two r < 1 that brings the maximum back into [2^(T-1), 2^T) (r = 1 otherwise: the scale only ever goes
// down), and chains the running scale: cum_out = cum_in * r; every other live gradient tensor is
// multiplied by the same r (kg_rows_scale), so that all live gradients share one scale.  Everything downstream is linear in g: its
// parameter gradients come out times the running scale of the moment and are divided by it where they leave (kg_scale_tensors,
// one device scalar per tensor).  A gradient tensor that was written under an earlier (larger) scale -- contributions waiting on
// tensors further upstream -- is converted when it is next combined or consumed: *= scale_now * (1 / scale_then) <= 1
// (kg_rows_scale with two device scalars).  All factors are powers of two: exact.  No host round trip.
__global__ __launch_bounds__(1024) void rows_absmax_kernel(const RowsR g, long M, int C8, int target_log2, const float* __restrict__ cum_in,
                                                          float* __restrict__ cum_out, float* __restrict__ r_out, unsigned* scratch) {
    const long total = M * C8;
    unsigned best = 0;
    // (row, chunk) advance by a constant stride: no division inside the loop; 4 independent 16-byte loads per plane in flight per thread
    const long stride = (long)gridDim.x * blockDim.x;
    const long dr = stride / C8; const int dc = (int)(stride - dr * C8);
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    long r = i / C8; int c = (int)(i - r * C8);
    while (i < total) {
        float v[4][8];
        long rr[4]; int cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rr[u] = r; cc[u] = c;
            r += dr; c += dc;
            if (c >= C8) { c -= C8; ++r; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * stride < total) rd8(g, rr[u], cc[u] * 8, v[u]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const unsigned b = __float_as_uint(v[u][e]) & 0x7fffffffu; best = b > best ? b : best; }
        i += 4 * stride;
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
        // (few, fat workgroups: the two same-address device-scope atomics per workgroup serialise at the memory side, ~13 ns each -- 2048
        // workgroups spent 54 us there for 8 us of reading.)  The two values travel through device-scope atomics only, so no fence: a device-scope release fence writes this XCD's whole L2 back
        // (measured in conv_tiny.hip: tens of microseconds per workgroup behind a kernel that left its output dirty in L2).  The ticket
        // is taken only after the max has RETURNED (data dependence through `one`): the last ticket holder sees every workgroup's max.
        unsigned one = 1u;
        const unsigned old = atomicMax(&scratch[0], m);
        asm volatile("" : "+v"(one) : "v"(old));
        last = atomicAdd(&scratch[1], one) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        const unsigned m = atomicMax(&scratch[0], 0u);
        float r = 1.f;
        if (m != 0 && m < 0x7f800000u) {      // (an all-zero or already overflowed tensor is left as it is)
            int e;
            frexpf(__uint_as_float(m), &e);
            int k = target_log2 - e;
            k = k > 0 ? 0 : (k < -60 ? -60 : k);      // only ever DOWN: a factor > 1 could push another live gradient tensor out of range
            r = ldexpf(1.f, k);
        }
        const float c = cum_in[0] * r;
        r_out[0] = r; cum_out[0] = c; cum_out[1] = 1.f / c;
        atomicExch(&scratch[0], 0u); atomicExch(&scratch[1], 0u);      // (the next launch's atomics meet zeros)
    }
}
__global__ void rows_scale_kernel(bf16_t* __restrict__ p, int ld, int P, int ps, long M, int C8, const float* __restrict__ r,
                                  const float* __restrict__ r2) {
    const float s = *r * (r2 ? *r2 : 1.f);
    if (s == 1.f) return;
    const long total = M * C8 * P;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long q = i / P; const int pl = (int)(i - q * P);
        const long row = q / C8; const int c = (int)(q - row * C8) * 8;
        uint4* a = reinterpret_cast<uint4*>(p + row * ld + (long)pl * ps + c);
        uint4 v = *a;
        bf16_t* h = reinterpret_cast<bf16_t*>(&v);
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = f2bf(bf2f(h[e]) * s);       // (a power of two: exact in every plane, barring the format's range)
        *a = v;
    }
}
// planes: a = g.  cum_in: device {scale, 1 / scale} the tensor is expressed in; cum_out (2 floats), r_out (1 float): see above;
// scratch: 2 zero-initialised unsigned (left zeroed).  In place.
extern "C" int kg_rows_rescale(void* g, int ld, long M, int C, int target_log2, const float* cum_in, float* cum_out, float* r_out,
                               void* scratch, const kg_planes_t* planes, void* stream) {
    KG_PLANES(planes);
    KG_CHECK_ARG(g && cum_in && cum_out && r_out && scratch && C % 8 == 0 && ld % 8 == 0, "kg_rows_rescale: bad args");
    if (M == 0) return KG_OK;
    const long total = M * (C / 8);
    // 1024 threads x 4 chunks in flight each; one workgroup per CU.  (Round 6 tried 64 workgroups for tensors that fit the L2 / Infinity Cache --
    // fewer serialised atomics -- and measured 14.9 -> 23.8 us per launch: the read, not the atomics, is the bound.)
    int blocks = (int)((total + 4095) / 4096); if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(rows_absmax_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, RowsR{(const bf16_t*)g, ld, pp.a_planes, pp.a_pstride}, M, C / 8,
                       target_log2, cum_in, cum_out, r_out, (unsigned*)scratch);
    KG_CHECK_LAUNCH("rows_absmax");
    // (grid-stride; a factor of 1 returns at once: dispatching 2048 empty 256-thread workgroups still cost 12 us per boundary, 24 boundaries per
    // step -- 512 workgroups of 1024 threads keep every CU's wave slots full when there IS work and cost a quarter of that when there is none)
    int b2 = (int)((total * pp.a_planes + 1023) / 1024); if (b2 > 512) b2 = 512;
    hipLaunchKernelGGL(rows_scale_kernel, dim3(b2), dim3(1024), 0, (hipStream_t)stream, (bf16_t*)g, ld, pp.a_planes, pp.a_pstride, M, C / 8, (const float*)r_out,
                       (const float*)nullptr);
    KG_CHECK_LAUNCH("rows_scale");
    return KG_OK;
}
// rows *= *r (* *r2 when given) (device scalars, powers of two), every plane, in place.  planes: a = g
extern "C" int kg_rows_scale(void* g, int ld, long M, int C, const float* r, const float* r2, const kg_planes_t* planes, void* stream) {
    KG_PLANES(planes);
    KG_CHECK_ARG(g && r && C % 8 == 0 && ld % 8 == 0, "kg_rows_scale: bad args");
    if (M == 0) return KG_OK;
    const long total = M * (C / 8) * pp.a_planes;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rows_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)g, ld, pp.a_planes, pp.a_pstride, M, C / 8, r, r2);
    KG_CHECK_LAUNCH("rows_scale");
    return KG_OK;
}
// up to 8 rows tensors *= *r in ONE launch (engine.renormalise: the other live gradient tensors of the pass); blockIdx.y = tensor
struct RowsScaleMulti { bf16_t* p[8]; int ld[8]; long M[8]; int C8[8]; int P[8]; int ps[8]; int n; };
__global__ void rows_scale_multi_kernel(RowsScaleMulti a, const float* __restrict__ r, const float* __restrict__ r2) {
    const float s = *r * (r2 ? *r2 : 1.f);
    if (s == 1.f) return;
    const int t = blockIdx.y;
    const int P = a.P[t], C8 = a.C8[t], ps = a.ps[t], ld = a.ld[t];
    bf16_t* p = a.p[t];
    const long total = a.M[t] * C8 * P;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long q = i / P; const int pl = (int)(i - q * P);
        const long row = q / C8; const int c = (int)(q - row * C8) * 8;
        uint4* ad = reinterpret_cast<uint4*>(p + row * ld + (long)pl * ps + c);
        uint4 v = *ad;
        bf16_t* h = reinterpret_cast<bf16_t*>(&v);
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = f2bf(bf2f(h[e]) * s);
        *ad = v;
    }
}
extern "C" int kg_rows_scale_multi(const long* desc, int n, const float* r, const float* r2, void* stream) {
    KG_CHECK_ARG(desc && r && n >= 1 && n <= 8, "kg_rows_scale_multi: 1..8 tensors");
    RowsScaleMulti a;
    long most = 0;
    for (int t = 0; t < 8; ++t) {
        const long* d = desc + 6 * (t < n ? t : 0);
        a.p[t] = (bf16_t*)d[0]; a.ld[t] = (int)d[1]; a.M[t] = t < n ? d[2] : 0; a.C8[t] = (int)d[3] / 8; a.P[t] = (int)d[4]; a.ps[t] = (int)d[5];
        KG_CHECK_ARG(t >= n || (a.p[t] && d[3] % 8 == 0 && d[1] % 8 == 0 && d[4] >= 1 && d[4] <= 3 && d[5] % 8 == 0), "kg_rows_scale_multi: bad descriptor");
        const long tot = a.M[t] * a.C8[t] * a.P[t];
        most = tot > most ? tot : most;
    }
    a.n = n;
    if (most == 0) return KG_OK;
    int blocks = (int)((most + 255) / 256); if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(rows_scale_multi_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, a, r, r2);
    KG_CHECK_LAUNCH("rows_scale_multi");
    return KG_OK;
}

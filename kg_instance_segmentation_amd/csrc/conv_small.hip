// conv_small.hip -- direct (VALU, fp32-accumulate) convolution for inputs of at most 8 channels.
//
// Three convs of KGnet's hot path have a tiny reduction depth per tap: c0_conv.0 (3 -> 64, 3x3 at full resolution,
// KGnet.py:139-142), the stem conv1 (3 -> 64, 7x7 stride 2, KGnet.py:131) and the input gradient of seg_head.2 (1 -> 64
// channels, 3x3 over the ragged box pixels, KGnet.py:145-147).  An MFMA implicit GEMM pads the 1..3 channels to a 32-wide
// k-step (the gather kernel reached 20-60 TFLOP/s there and is latency bound); these layers are bound by their output
// bytes instead, so: one thread = 4 pixels x 8 consecutive couts, fp32 FMAs, the weights of the workgroup's 64 couts in
// LDS as fp32 [tap][ci][64] (channels whose weights are all zero -- the padding -- are skipped), one 16-byte load per
// (pixel, tap), and every output row leaves as 128 contiguous bytes per 8 lanes.
// Same operands, modes and epilogue as kg_conv2d_igemm (bf16 rows in, packed bf16 weights, bias / residual / ReLU / mask).
#include "conv_args.h"
#include <stdlib.h>

__global__ __launch_bounds__(256) void conv_small_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* wl = reinterpret_cast<float*>(smem);                 // [ntaps][8 ci][64 co]
    __shared__ int ci_used;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * 64;
    if (tid == 0) ci_used = 0;
    __syncthreads();
    int used = 0;
    for (int e = tid; e < a.ntaps * 8 * 64; e += 256) {
        const int co = e & 63, ci = (e >> 6) & 7, tap = e >> 9;
        const bf16_t wv = a.w[(long)(c0 + co) * a.K + tap * 8 + ci];      // packed rows are zero past Cout
        wl[e] = bf2f(wv);
        if (wv & 0x7fff) used |= 1 << ci;
    }
    if (used) atomicOr(&ci_used, used);
    __syncthreads();
    const int cmask = ci_used;

    const int cg = tid & 7, pl = tid >> 3;                      // couts c0 + cg*8 .. +7; pixels pl + 32 q
    const int cb = c0 + cg * 8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
    const int smask = (1 << a.stride_log2) - 1;
    const int ohw = a.OH * a.OW;

    for (long m0 = (long)blockIdx.x * 128; m0 < a.M; m0 += (long)gridDim.x * 128) {
        int py[4], px[4], ph[4], pw[4];
        long pbase[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long m = m0 + pl + 32 * q;
            py[q] = px[q] = ph[q] = pw[q] = 0; pbase[q] = -1;
            if (m < a.M) {
                if (a.mode >= 2) {
                    const int2 d = a.rowdesc[m];
                    py[q] = d.x >> 16; px[q] = d.x & 0xffff; ph[q] = d.y >> 16; pw[q] = d.y & 0xffff; pbase[q] = m;
                } else {
                    const int n = (int)(m / ohw), rem = (int)(m - (long)n * ohw);
                    py[q] = rem / a.OW; px[q] = rem - py[q] * a.OW; pbase[q] = (long)n * a.H * a.W; ph[q] = a.H; pw[q] = a.W;
                }
            }
        }
        float acc[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[q][e] = KG_BIAS_ACC(bv[e]);
        for (int tap = 0; tap < a.ntaps; ++tap) {
            const int dy = tap / a.KW, dx = tap - dy * a.KW;
            const int dyo = dy - a.pad, dxo = dx - a.pad;
            uint4 xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bool ok = pbase[q] >= 0;
                long row = 0;
                if (a.mode == 0) {
                    const int iy = (py[q] << a.stride_log2) + dyo, ix = (px[q] << a.stride_log2) + dxo;
                    ok = ok && (unsigned)iy < (unsigned)ph[q] && (unsigned)ix < (unsigned)pw[q];
                    row = pbase[q] + (long)iy * pw[q] + ix;
                } else if (a.mode == 1) {
                    const int ty = py[q] - dyo, tx = px[q] - dxo;
                    ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & smask) == 0;
                    const int iy = ty >> a.stride_log2, ix = tx >> a.stride_log2;
                    ok = ok && iy < ph[q] && ix < pw[q];
                    row = pbase[q] + (long)iy * pw[q] + ix;
                } else {
                    const int sy = a.mode == 2 ? dyo : -dyo, sx = a.mode == 2 ? dxo : -dxo;
                    const int iy = py[q] + sy, ix = px[q] + sx;
                    ok = ok && (unsigned)iy < (unsigned)ph[q] && (unsigned)ix < (unsigned)pw[q];
                    row = pbase[q] + (long)sy * pw[q] + sx;
                }
                xv[q] = ok ? *reinterpret_cast<const uint4*>(a.x + row * a.ldx) : make_uint4(0, 0, 0, 0);
            }
            const float* wt = wl + tap * 512 + cg * 8;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (!((cmask >> ci) & 1)) continue;              // workgroup-uniform
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + ci * 64);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wt + ci * 64 + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16_t* xs = reinterpret_cast<const bf16_t*>(&xv[q]);
                    const float xf = bf2f(xs[ci]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[q][e] += w0[e] * xf; acc[q][4 + e] += w1[e] * xf; }
                }
            }
        }
        if (cb >= a.Cout) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long m = m0 + pl + 32 * q;
            if (m >= a.M) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = KG_ACC(acc[q][e]);
            if (a.res) {
                const uint4 rv = *reinterpret_cast<const uint4*>(a.res + m * a.ldres + cb);
                const bf16_t* rs = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bf2f(rs[e]);
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = kg_relu(v[e]);
            }
            if (a.mask) {
                const uint4 mv = *reinterpret_cast<const uint4*>(a.mask + m * a.ldmask + cb);
                const bf16_t* ms = reinterpret_cast<const bf16_t*>(&mv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = bf2f(ms[e]) > 0.f ? v[e] : 0.f;
            }
            *reinterpret_cast<uint4*>(a.y + m * a.ldy + cb) =
                make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        }
    }
}

// ---- MFMA variant: FOUR taps per 32-wide k-step.  With cin_pad == 8 the 8 channels of one (pixel, tap) are exactly one lane's
// 16-byte share of a v_mfma_f32_16x16x32_bf16 B fragment (lane group g = tap 4s + g), and the packed weight row [tap][8] is laid
// out the same way, so both operands are plain 16-byte global loads -- no LDS, no barrier, no per-tap padding to 32 channels (the
// gather kernel's 20-60 TFLOP/s on these layers) and no fp32 FMAs (the VALU kernel above is latency bound at 0.29 ms for 3 -> 64 at
// 8 x 512^2).  A wave owns 64 pixels x 64 couts; 3x3: 3 k-steps, 7x7: 13.  Bound by the output bytes.
__global__ __launch_bounds__(256) void conv_small_mfma_kernel(const ConvArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, g = lane >> 4;
    const long m0 = (long)blockIdx.x * 256 + wave * 64;
    const int c0 = blockIdx.y * 64;
    const int smask = (1 << a.stride_log2) - 1;
    const int ohw = a.OH * a.OW;
    int py[4], px[4], ph[4], pw[4];
    long pbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long m = m0 + j * 16 + lm;
        py[j] = px[j] = ph[j] = pw[j] = 0; pbase[j] = -1;
        if (m < a.M) {
            if (a.mode >= 2) {
                const int2 d = a.rowdesc[m];
                py[j] = d.x >> 16; px[j] = d.x & 0xffff; ph[j] = d.y >> 16; pw[j] = d.y & 0xffff; pbase[j] = m;
            } else {
                const int n = (int)(m / ohw), rem = (int)(m - (long)n * ohw);
                py[j] = rem / a.OW; px[j] = rem - py[j] * a.OW; pbase[j] = (long)n * a.H * a.W; ph[j] = a.H; pw[j] = a.W;
            }
        }
    }
    const bf16_t* wrow[4];
    const int nv = a.km.total;      // split-bf16 planes: nv virtual planes of 8 channels per tap (kg_common.h), k-step = (tap quad, virtual plane)
#pragma unroll
    for (int i = 0; i < 4; ++i) wrow[i] = a.w + (long)(c0 + (lm >> 2) * 16 + i * 4 + (lm & 3)) * a.K + 8 * g * nv;   // rows / columns past the tensor are zero
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = (a.ntaps + 3) >> 2;
    for (int s = 0; s < nks; ++s) {
        const int tap = 4 * s + g;
        const int dy = tap / a.KW, dx = tap - dy * a.KW;
        const int dyo = dy - a.pad, dxo = dx - a.pad;
        long srow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = pbase[j] >= 0 && tap < a.ntaps;
            long row = 0;
            if (a.mode == 0) {
                const int iy = (py[j] << a.stride_log2) + dyo, ix = (px[j] << a.stride_log2) + dxo;
                ok = ok && (unsigned)iy < (unsigned)ph[j] && (unsigned)ix < (unsigned)pw[j];
                row = pbase[j] + (long)iy * pw[j] + ix;
            } else if (a.mode == 1) {
                const int ty = py[j] - dyo, tx = px[j] - dxo;
                ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & smask) == 0;
                const int iy = ty >> a.stride_log2, ix = tx >> a.stride_log2;
                ok = ok && iy < ph[j] && ix < pw[j];
                row = pbase[j] + (long)iy * pw[j] + ix;
            } else {
                const int sy = a.mode == 2 ? dyo : -dyo, sx = a.mode == 2 ? dxo : -dxo;
                const int iy = py[j] + sy, ix = px[j] + sx;
                ok = ok && (unsigned)iy < (unsigned)ph[j] && (unsigned)ix < (unsigned)pw[j];
                row = pbase[j] + (long)sy * pw[j] + sx;
            }
            srow[j] = ok ? row : -1;
        }
        for (int vq = 0; vq < nv; ++vq) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wrow[i] + (32 * s) * nv + 8 * vq);
            const int xo = a.km.xoff(vq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (srow[j] >= 0) v = *reinterpret_cast<const uint4*>(a.x + srow[j] * a.ldx + xo);
                bfr[j] = *reinterpret_cast<const bf16x8*>(&v);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = KG_MFMA16(af[i], bfr[j], acc[i][j]);
        }
    }
    // ---- epilogue: lane owns pixel row m and couts cb .. cb+15 (Cout % 8 == 0, 16-byte aligned rows: checked by the caller) ----
    const int cb = c0 + g * 16;
    if (cb >= a.Cout) return;
    const EpiArgs ep = kg_epi(a);
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
    float sv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sv[e] = (a.oscale && cb + e < a.Cout) ? a.oscale[cb + e] : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long m = m0 + j * 16 + lm;
        if (m >= a.M) continue;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = KG_ACC(acc[e >> 2][j][e & 3]) * sv[e] + bv[e];
        kg_conv_epilogue<16>(ep, m, cb, v);
    }
}

// cin_pad == 8, Cout % 8 == 0, bf16 row output with 16-byte aligned rows (checked by the caller, kg_conv2d_igemm)
int kg_launch_conv_small(const ConvArgs& a, hipStream_t st) {
    constexpr int use_mfma = 1;
    const bool planed = a.km.total > 1 || a.yP > 1 || a.rP > 1;
    if (planed && !(a.K >= 32 * a.km.total * ((a.ntaps + 3) / 4))) { kg_set_error("conv_small: packed rows too short for the plane layout"); return KG_ERR_ARG; }
    if (a.oscale && !(a.K >= 32 * a.km.total * ((a.ntaps + 3) / 4))) { kg_set_error("conv_small: an output scale needs the MFMA variant's packed layout"); return KG_ERR_ARG; }
    if ((use_mfma || planed || a.oscale) && a.K >= 32 * a.km.total * ((a.ntaps + 3) / 4)) {
        hipLaunchKernelGGL(conv_small_mfma_kernel, dim3((unsigned)((a.M + 255) / 256), kg_cdiv(a.Cout, 64)), dim3(256), 0, st, a);
        KG_CHECK_LAUNCH("conv_small_mfma");
        kg_note_kernel("conv_small_mfma_kernel");
        return KG_OK;
    }
    const int smem = a.ntaps * 8 * 64 * 4;
    static KgPerDevice attr_done;
    if (attr_done.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)conv_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 49 * 8 * 64 * 4));
    }
    const long groups = (a.M + 127) / 128;
    const int per_cu = smem <= 32768 ? 4 : 1;
    long gx = 256L * per_cu * 2;
    if (gx > groups) gx = groups;
    hipLaunchKernelGGL(conv_small_kernel, dim3((unsigned)gx, kg_cdiv(a.Cout, 64)), dim3(256), smem, st, a);
    KG_CHECK_LAUNCH("conv_small");
    kg_note_kernel("conv_small_kernel");
    return KG_OK;
}

// ---- im2col for <= 8 input channels: the weight gradient of the stem conv1 (3 -> 64, 7x7 stride 2, KGnet.py:131) as a 1x1
// weight-gradient GEMM.  The per-tap gather kernel pads the 3 channels to a 64-wide tile for each of the 49 taps (16 TFLOP/s).
// out[m][tap * cin + ci] = x[src(m, tap)][ci] (zero outside the image), rows of Kpad >= taps * cin bf16 values.
__global__ __launch_bounds__(256) void im2col_small_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int N, int H, int W,
                                                           int OH, int OW, int KH, int KW, int stride, int pad, int cin, int ldx, int Kpad) {
    const long total = (long)N * OH * OW * KH * KW;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int tap = (int)(i % (KH * KW));
        const long m = i / (KH * KW);
        const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), n = (int)(m / ((long)OW * OH));
        const int ky = tap / KW, kx = tap - ky * KW;
        const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const uint4*>(x + (((long)n * H + iy) * W + ix) * ldx);
        const bf16_t* vs = reinterpret_cast<const bf16_t*>(&v);
        bf16_t* o = out + m * Kpad + tap * cin;
        for (int c = 0; c < cin; ++c) o[c] = vs[c];
    }
}

// x: bf16 rows [N*H*W][ldx >= 8]; out: bf16 rows [N*OH*OW][Kpad] (columns >= KH*KW*cin must have been zeroed by the caller)
extern "C" int kg_im2col_small(const void* x, void* out, int N, int H, int W, int OH, int OW, int KH, int KW, int stride, int pad, int cin,
                               int ldx, int Kpad, void* stream) {
    KG_CHECK_ARG(x && out && cin >= 1 && cin <= 8 && ldx % 8 == 0 && Kpad >= KH * KW * cin, "kg_im2col_small: bad arguments");
    const long total = (long)N * OH * OW * KH * KW;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(im2col_small_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, N, H, W, OH,
                       OW, KH, KW, stride, pad, cin, ldx, Kpad);
    KG_CHECK_LAUNCH("im2col_small");
    return KG_OK;
}

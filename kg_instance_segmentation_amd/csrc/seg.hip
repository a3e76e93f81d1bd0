// seg.hip -- ragged (per-box) plumbing of KGnet's segmentation branch on gfx950.
// Reference: get_patches / forward_seg (KGnet.py:246-256, 321-350) crop every feature level per box
// and run the top-down combine per box in a Python loop.  Here all boxes of a level are one ragged
// pixel list [rows][C] (box-major, raster inside a box); convolutions on it run through
// kg_conv2d_igemm's ragged modes with the row descriptors built below.
#include "kg_common.h"
#include <type_traits>

// boxtab[b] = {n, y1, x1, h, w, row0, H, W}; one block per box.
__global__ void seg_build_rows_kernel(const int* __restrict__ boxtab, int2* __restrict__ rowdesc,
                                      int* __restrict__ row2box, int* __restrict__ srcrow) {
    const int* t = boxtab + blockIdx.x * 8;
    const int n = t[0], y1 = t[1], x1 = t[2], h = t[3], w = t[4], row0 = t[5], H = t[6], W = t[7];
    for (int r = threadIdx.x; r < h * w; r += blockDim.x) {
        int y = r / w, x = r - y * w;
        rowdesc[row0 + r] = make_int2((y << 16) | x, (h << 16) | w);
        row2box[row0 + r] = blockIdx.x;
        srcrow[row0 + r] = (n * H + y1 + y) * W + x1 + x;
    }
}
extern "C" int kg_seg_build_rows(const int* boxtab, int nb, int* rowdesc, int* row2box, int* srcrow, void* stream) {
    KG_CHECK_ARG(boxtab && rowdesc && row2box && srcrow, "kg_seg_build_rows: null pointer");
    if (nb == 0) return KG_OK;
    hipLaunchKernelGGL(seg_build_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, boxtab, (int2*)rowdesc, row2box, srcrow);
    KG_CHECK_LAUNCH("seg_build_rows");
    return KG_OK;
}

// all pyramid levels of a batch in ONE launch (five launches of a few microseconds each otherwise): block -> (level, box) through the level starts
struct SegLevels { const int* boxtab[8]; int2* rowdesc[8]; int* row2box[8]; int* srcrow[8]; int start[9]; int nlev; };
__global__ void seg_build_rows_levels_kernel(const SegLevels a) {
    int l = 0;
    while (l + 1 < a.nlev && (int)blockIdx.x >= a.start[l + 1]) ++l;
    const int b = blockIdx.x - a.start[l];
    const int* t = a.boxtab[l] + b * 8;
    const int n = t[0], y1 = t[1], x1 = t[2], h = t[3], w = t[4], row0 = t[5], H = t[6], W = t[7];
    int2* rowdesc = a.rowdesc[l];
    int* row2box = a.row2box[l];
    int* srcrow = a.srcrow[l];
    for (int r = threadIdx.x; r < h * w; r += blockDim.x) {
        int y = r / w, x = r - y * w;
        rowdesc[row0 + r] = make_int2((y << 16) | x, (h << 16) | w);
        row2box[row0 + r] = b;
        srcrow[row0 + r] = (n * H + y1 + y) * W + x1 + x;
    }
}
// host arrays of nlev (<= 8) device pointers / box counts; a level with nb[l] == 0 is skipped
extern "C" int kg_seg_build_rows_levels(int nlev, const int* const* boxtab, const int* nb, int* const* rowdesc, int* const* row2box, int* const* srcrow,
                                        void* stream) {
    KG_CHECK_ARG(nlev >= 1 && nlev <= 8 && boxtab && nb && rowdesc && row2box && srcrow, "kg_seg_build_rows_levels: bad args");
    SegLevels a;
    a.nlev = 0; a.start[0] = 0;
    for (int l = 0; l < nlev; ++l) {
        if (nb[l] <= 0) continue;
        KG_CHECK_ARG(boxtab[l] && rowdesc[l] && row2box[l] && srcrow[l], "kg_seg_build_rows_levels: null pointer");
        a.boxtab[a.nlev] = boxtab[l]; a.rowdesc[a.nlev] = (int2*)rowdesc[l]; a.row2box[a.nlev] = row2box[l]; a.srcrow[a.nlev] = srcrow[l];
        a.start[a.nlev + 1] = a.start[a.nlev] + nb[l];
        ++a.nlev;
    }
    if (a.nlev == 0) return KG_OK;
    hipLaunchKernelGGL(seg_build_rows_levels_kernel, dim3(a.start[a.nlev]), dim3(256), 0, (hipStream_t)stream, a);
    KG_CHECK_LAUNCH("seg_build_rows_levels");
    return KG_OK;
}

// dst[r][0:C] = src[srcrow[r]][0:C]
__global__ void rows_gather_kernel(const bf16_t* __restrict__ src, int ldsrc, const int* __restrict__ srcrow,
                                   bf16_t* __restrict__ dst, int lddst, long nrows, int C8) {
    long total = nrows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, C8, &r, &c); c *= 8;
        *reinterpret_cast<uint4*>(dst + r * lddst + c) = *reinterpret_cast<const uint4*>(src + (long)srcrow[r] * ldsrc + c);
    }
}
extern "C" int kg_rows_gather(const void* src, int ldsrc, const int* srcrow, void* dst, int lddst, long nrows, int C,
                              void* stream) {
    KG_CHECK_ARG(src && srcrow && dst && C % 8 == 0 && ldsrc % 8 == 0 && lddst % 8 == 0, "kg_rows_gather: bad args");
    if (nrows == 0) return KG_OK;
    long total = nrows * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(rows_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ldsrc, srcrow,
                       (bf16_t*)dst, lddst, nrows, C / 8);
    KG_CHECK_LAUNCH("rows_gather");
    return KG_OK;
}

// the same for P planes in one launch: dst[r][p * dps + c] = src[srcrow[r]][p * sps + c] (split rows keep their planes at column offsets p * ps)
__global__ void rows_gather_planes_kernel(const bf16_t* __restrict__ src, int ldsrc, int sps, const int* __restrict__ srcrow,
                                          bf16_t* __restrict__ dst, int lddst, int dps, long nrows, int C8, int P) {
    const int PC8 = P * C8;
    long total = nrows * PC8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, PC8, &r, &c);
        const int p = c / C8; c = (c - p * C8) * 8;
        *reinterpret_cast<uint4*>(dst + r * lddst + p * dps + c) = *reinterpret_cast<const uint4*>(src + (long)srcrow[r] * ldsrc + p * sps + c);
    }
}
extern "C" int kg_rows_gather_planes(const void* src, int ldsrc, int src_pstride, const int* srcrow, void* dst, int lddst, int dst_pstride, long nrows, int C,
                                     int P, void* stream) {
    KG_CHECK_ARG(src && srcrow && dst && C % 8 == 0 && ldsrc % 8 == 0 && lddst % 8 == 0 && src_pstride % 8 == 0 && dst_pstride % 8 == 0 && P >= 1 && P <= 4,
                 "kg_rows_gather_planes: bad args");
    if (nrows == 0) return KG_OK;
    long total = nrows * (C / 8) * P;
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(rows_gather_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ldsrc, src_pstride, srcrow,
                       (bf16_t*)dst, lddst, dst_pstride, nrows, C / 8, P);
    KG_CHECK_LAUNCH("rows_gather_planes");
    return KG_OK;
}

// out[r][c] = bf16( acc[r][c] (+ addto[r][c]) )
__global__ void f32_to_bf16_rows_kernel(const float* __restrict__ acc, bf16_t* __restrict__ out, int C8, long rows,
                                        int ldout, const bf16_t* __restrict__ addto, int ldadd) {
    long total = rows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, C8, &r, &c); c *= 8;
        const float* a = acc + (r * C8) * 8 + c;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = a[e];
        if (addto) {
            uint4 t = *reinterpret_cast<const uint4*>(addto + r * ldadd + c);
            const bf16_t* s = reinterpret_cast<const bf16_t*>(&t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bf2f(s[e]);
        }
        uint4 o = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        *reinterpret_cast<uint4*>(out + r * ldout + c) = o;
    }
}
extern "C" int kg_f32_to_bf16_rows(const float* acc, void* out, int C, long rows, int ldout, const void* addto, int ldadd,
                                   void* stream) {
    KG_CHECK_ARG(acc && out && C % 8 == 0 && ldout % 8 == 0, "kg_f32_to_bf16_rows: bad args");
    if (rows == 0) return KG_OK;
    long total = rows * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, (bf16_t*)out, C / 8, rows,
                       ldout, (const bf16_t*)addto, ldadd);
    KG_CHECK_LAUNCH("f32_to_bf16_rows");
    return KG_OK;
}

// ---- split-bf16 / fp32 variants (kg_common.h "planes") ---------------------------------------------------------------------
// dst[r][0:C] (P planes) = src[srcrow[r]][0:C] with src fp32 rows (the fp32 feature maps forward_dec returns, KGnet.py:318)
__global__ void rows_gather_f32_kernel(const float* __restrict__ src, int ldsrc, const int* __restrict__ srcrow,
                                       bf16_t* __restrict__ dst, int lddst, int P, int ps, long nrows, int C8) {
    long total = nrows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, C8, &r, &c); c *= 8;
        const float4* sp = reinterpret_cast<const float4*>(src + (long)srcrow[r] * ldsrc + c);
        const float4 a = sp[0], b = sp[1];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        kg_store_planes<8>(dst + r * lddst + c, P, ps, v, true);
    }
}
// planes: y = dst
extern "C" int kg_rows_gather_f32(const float* src, int ldsrc, const int* srcrow, void* dst, int lddst, long nrows, int C,
                                  const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(src && srcrow && dst && C % 8 == 0 && ldsrc % 4 == 0 && lddst % 8 == 0, "kg_rows_gather_f32: bad args");
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp) && (reinterpret_cast<uintptr_t>(src) & 15) == 0, "kg_rows_gather_f32: bad planes / alignment");
    if (nrows == 0) return KG_OK;
    long total = nrows * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(rows_gather_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ldsrc, srcrow,
                       (bf16_t*)dst, lddst, pp.y_planes, pp.y_pstride, nrows, C / 8);
    KG_CHECK_LAUNCH("rows_gather_f32");
    return KG_OK;
}

// out[row][0:C] (fp32) = sum of the planes of x[row][0:C]   (exports of c0..c4, KGnet.py:318: the reference returns fp32)
__global__ void planes_to_f32_kernel(const bf16_t* __restrict__ x, int ldx, int P, int ps, float* __restrict__ out, int ldout,
                                     long rows, int C8) {
    long total = rows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, C8, &r, &c); c *= 8;
        float v[8];
        kg_load_planes8(x + r * ldx + c, P, ps, v);
        float4* op = reinterpret_cast<float4*>(out + r * ldout + c);
        op[0] = make_float4(v[0], v[1], v[2], v[3]); op[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}
// planes: a = x
extern "C" int kg_planes_to_f32(const void* x, int ldx, float* out, int ldout, long rows, int C, const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(x && out && C % 8 == 0 && ldx % 8 == 0 && ldout % 4 == 0, "kg_planes_to_f32: bad args");
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp) && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "kg_planes_to_f32: bad planes / alignment");
    if (rows == 0) return KG_OK;
    long total = rows * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, pp.a_planes, pp.a_pstride,
                       out, ldout, rows, C / 8);
    KG_CHECK_LAUNCH("planes_to_f32");
    return KG_OK;
}

// out[row][0:C] (P planes) = acc[row][0:C] (fp32) (+ addto planes)
__global__ void f32_to_planes_kernel(const float* __restrict__ acc, int ldacc, bf16_t* __restrict__ out, int ldout, int P, int ps,
                                     const bf16_t* __restrict__ addto, int ldadd, int aP, int aps, long rows, int C8,
                                     const float* __restrict__ scale) {
    long total = rows * C8;
    const float S = scale ? *scale : 1.f;       // (gradients entering the half build's backward pass: gradscale.hip; addto is already scaled)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r; int c; kg_divmod(i, C8, &r, &c); c *= 8;
        const float4* sp = reinterpret_cast<const float4*>(acc + r * ldacc + c);
        const float4 a = sp[0], b = sp[1];
        float v[8] = {a.x * S, a.y * S, a.z * S, a.w * S, b.x * S, b.y * S, b.z * S, b.w * S};
        if (addto) {
            float t[8];
            kg_load_planes8(addto + r * ldadd + c, aP, aps, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        kg_store_planes<8>(out + r * ldout + c, P, ps, v, true);
    }
}
// planes: b = addto, y = out
extern "C" int kg_f32_to_planes(const float* acc, int ldacc, void* out, int ldout, const void* addto, int ldadd, long rows, int C,
                                const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(acc && out && C % 8 == 0 && ldout % 8 == 0 && ldacc % 4 == 0 && (!addto || ldadd % 8 == 0), "kg_f32_to_planes: bad args");
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp) && (reinterpret_cast<uintptr_t>(acc) & 15) == 0, "kg_f32_to_planes: bad planes / alignment");
    if (rows == 0) return KG_OK;
    long total = rows * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(f32_to_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, ldacc, (bf16_t*)out, ldout, pp.y_planes,
                       pp.y_pstride, (const bf16_t*)addto, ldadd, pp.b_planes, pp.b_pstride, rows, C / 8, pp.scale);
    KG_CHECK_LAUNCH("f32_to_planes");
    return KG_OK;
}

// ---- deterministic crop-gradient reduction ---------------------------------------------------------------------------------
// Gradient of get_patches' slicing (KGnet.py:246-256): dfeat[n][y][x][c] = sum over the boxes b that contain (y, x) of
// g[row0_b + (y - y1_b) * w_b + (x - x1_b)][c].  Gather form, one thread per (feature pixel, 8 channels): the candidate boxes
// of a pixel come from a bin grid (bins of BS x BS pixels, CSR lists in ascending box order, built by the host next to the box
// table), the sum runs in fp32 in that fixed order -> no atomics, bit-reproducible, every output element written exactly once.
// Rows [0, rows_a) of the ragged list come from ga (a column slice of the concat gradient), the rest from gb (already offset:
// gb row r - rows_a); either may be null when its range is empty.
__global__ void crop_grad_reduce_kernel(const bf16_t* __restrict__ ga, int lda, const bf16_t* __restrict__ gb, int ldb, int gP, int gps_a,
                                        int gps_b, long rows_a, const int* __restrict__ boxtab, const int* __restrict__ bin_start,
                                        const int* __restrict__ bin_boxes, int BS, int BY, int BX, int H, int W,
                                        float* __restrict__ out, bf16_t* __restrict__ outp, int ldo, int oP, int ops_, long npix, int C8) {
    long total = npix * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long p; int c; kg_divmod(i, C8, &p, &c); c *= 8;
        int x, y; long q, nl; kg_divmod(p, W, &q, &x); kg_divmod(q, H, &nl, &y); const int n = (int)nl;
        const int bin = (n * BY + y / BS) * BX + x / BS;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = bin_start[bin]; k < bin_start[bin + 1]; ++k) {
            const int* t = boxtab + bin_boxes[k] * 8;   // {n, y1, x1, h, w, row0, H, W}
            const int dy = y - t[1], dx = x - t[2];
            if ((unsigned)dy >= (unsigned)t[3] || (unsigned)dx >= (unsigned)t[4]) continue;
            const long r = (long)t[5] + (long)dy * t[4] + dx;
            float v[8];
            if (r < rows_a) kg_load_planes8(ga + r * lda + c, gP, gps_a, v);
            else kg_load_planes8(gb + (r - rows_a) * ldb + c, gP, gps_b, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
        if (outp) { kg_store_planes<8>(outp + p * ldo + c, oP, ops_, acc, true); continue; }
        float4* op = reinterpret_cast<float4*>(out + p * (long)(C8 * 8) + c);
        op[0] = make_float4(acc[0], acc[1], acc[2], acc[3]); op[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}
// planes: a = ga, b = gb (same plane count); y = out_rows.  Exactly one of out (fp32 [N*H*W][C]) / out_rows (split-bf16 rows, ld
// ldout: the engine's own gradient storage, fused training forward) is written.
extern "C" int kg_crop_grad_reduce(const void* ga, int lda, const void* gb, int ldb, long rows_a, const int* boxtab, const int* bin_start,
                                   const int* bin_boxes, int bin_size, int N, int H, int W, int C, float* out, void* out_rows, int ldout,
                                   const kg_planes_t* planes, void* stream) {
    KG_CHECK_ARG(boxtab && bin_start && bin_boxes && ((out != nullptr) != (out_rows != nullptr)) && C % 8 == 0 && bin_size > 0 && N > 0 && H > 0 && W > 0, "kg_crop_grad_reduce: bad args");
    KG_CHECK_ARG(!out_rows || ldout % 8 == 0, "kg_crop_grad_reduce: bad output rows");
    KG_CHECK_ARG((!ga || lda % 8 == 0) && (!gb || ldb % 8 == 0) && (ga || gb), "kg_crop_grad_reduce: bad gradient operands");
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp) && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "kg_crop_grad_reduce: bad planes / alignment");
    const long npix = (long)N * H * W;
    long total = npix * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(crop_grad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ga, lda, (const bf16_t*)gb, ldb,
                       pp.a_planes, pp.a_pstride, pp.b_pstride, ga ? rows_a : 0, boxtab, bin_start, bin_boxes, bin_size, kg_cdiv(H, bin_size),
                       kg_cdiv(W, bin_size), H, W, out, (bf16_t*)out_rows, ldout, pp.y_planes, pp.y_pstride, npix, C / 8);
    KG_CHECK_LAUNCH("crop_grad_reduce");
    return KG_OK;
}

// ---- seg_head.2: the 3x3 conv with ONE output channel (KGnet.py:145-147, Conv2d(64, 1, 3, padding=1)) over the ragged pixel list ----------
// On the MFMA kernels its single cout is padded to a 64-row tile (98 % of the multiplies are zeros: 0.64 ms per train step at 2 400
// boxes).  It is a dot product per pixel: 9 taps x C channels.  Here 8 lanes share a pixel (lane = 8-channel chunk, one 16-byte load per
// tap and plane: the 8 lanes of a pixel read one whole 128-byte row), a lane keeps its 9 x 8 fp32 weights in registers for all the pixels
// it walks, the taps are neighbour rows of the box-major raster list (row +- w +- 1: zero outside the box, the zero padding of the
// reference's per-crop conv), and the 8 partial sums meet through three wave shuffles in a fixed order.  Arithmetic: the plane sum of x
// (the exact stored value) times the fp32 master weight, fp32 FMA chain in (tap, channel) order -- not the 3-product MFMA evaluation,
// but the same values to fp32 rounding.
template <int C, int PP>
__global__ __launch_bounds__(256) void seg_conv3_c1_kernel(const bf16_t* __restrict__ x, int ldx, int ps, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const int2* __restrict__ rowdesc, long M,
                                                           float* __restrict__ y) {
    static_assert(C == 64, "one 128-byte row per pixel");
    const int chunk = threadIdx.x & 7;
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[t][e] = w[(chunk * 8 + e) * 9 + t];          // OIHW [1][C][3][3]
    const float b0 = bias ? bias[0] : 0.f;
    const long stride = (long)gridDim.x * 32;
    for (long m = (long)blockIdx.x * 32 + (threadIdx.x >> 3); m < M + 31; m += stride) {     // (all lanes stay in the loop for the shuffles)
        const bool live = m < M;
        float acc = 0.f;
        if (live) {
            const int2 d = rowdesc[m];
            const int py = d.x >> 16, px = d.x & 0xffff, h = d.y >> 16, wd = d.y & 0xffff;
            // all 9 x P loads of the pixel are issued unconditionally (a tap outside the box reads the centre row instead and is dropped by the
            // select below): with the bounds test as a branch around the load the compiler kept one tap in flight at a time and the kernel sat
            // at 0.9 TB/s, 82 % of its wave cycles parked on memory (profiles/r06_pmc_wait.json).  Same taps, same order, same fmaf chain.
            uint4 q[9][PP];
            bool in[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                in[t] = (unsigned)(py + dy) < (unsigned)h && (unsigned)(px + dx) < (unsigned)wd;
                const bf16_t* xp = x + (m + (in[t] ? (long)dy * wd + dx : 0)) * ldx + chunk * 8;
#pragma unroll
                for (int p = 0; p < PP; ++p) q[t][p] = *reinterpret_cast<const uint4*>(xp + (long)p * ps);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v[8];                               // plane sum, lowest plane first (kg_load_planes8): the exact stored value
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = bf2f(reinterpret_cast<const bf16_t*>(&q[t][PP - 1])[e]);
#pragma unroll
                for (int p = PP - 2; p >= 0; --p)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf2f(reinterpret_cast<const bf16_t*>(&q[t][p])[e]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = in[t] ? fmaf(v[e], wr[t][e], acc) : acc;
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (live && chunk == 0) y[m] = acc + b0;
    }
}
// x: rows [M][ldx] (planes a of `planes`), C = 64 channels; w: fp32 OIHW [1][64][3][3] (the master parameter); y: fp32 [M].
extern "C" int kg_seg_conv3_c1(const void* x, int ldx, int C, const float* w, const float* bias, const int* rowdesc, long M, float* y,
                               const kg_planes_t* planes, void* stream) {
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_seg_conv3_c1: bad kg_planes_t");
    KG_CHECK_ARG(x && w && rowdesc && y && C == 64 && ldx % 8 == 0, "kg_seg_conv3_c1: needs 64 input channels (got %d)", C);
    if (M == 0) return KG_OK;
    long blocks = (M + 31) / 32;
    if (blocks > 256 * 16) blocks = 256 * 16;
    auto go = [&](auto pp_c) {
        hipLaunchKernelGGL((seg_conv3_c1_kernel<64, decltype(pp_c)::value>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                           pp.a_pstride, w, bias, (const int2*)rowdesc, M, y);
    };
    if (pp.a_planes == 1) go(std::integral_constant<int, 1>{});
    else if (pp.a_planes == 2) go(std::integral_constant<int, 2>{});
    else go(std::integral_constant<int, 3>{});
    KG_CHECK_LAUNCH("seg_conv3_c1");
    return KG_OK;
}

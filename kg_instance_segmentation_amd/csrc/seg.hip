// seg.hip -- ragged (per-box) plumbing of KGnet's segmentation branch on gfx950.
// Reference: get_patches / forward_seg (KGnet.py:246-256, 321-350) crop every feature level per box
// and run the top-down combine per box in a Python loop.  Here all boxes of a level are one ragged
// pixel list [rows][C] (box-major, raster inside a box); convolutions on it run through
// kg_conv2d_igemm's ragged modes with the row descriptors built below.
#include "kg_common.h"

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

// dst[r][0:C] = src[srcrow[r]][0:C]
__global__ void rows_gather_kernel(const bf16_t* __restrict__ src, int ldsrc, const int* __restrict__ srcrow,
                                   bf16_t* __restrict__ dst, int lddst, long nrows, int C8) {
    long total = nrows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / C8; int c = (int)(i - r * C8) * 8;
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

// acc[srcrow[r]][c] += g[r][c]   (fp32 atomics: boxes may overlap on the feature map)
__global__ void rows_scatter_add_kernel(const bf16_t* __restrict__ g, int ld, const int* __restrict__ srcrow,
                                        float* __restrict__ acc, int C8, long nrows, int accld) {
    long total = nrows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / C8; int c = (int)(i - r * C8) * 8;
        uint4 v = *reinterpret_cast<const uint4*>(g + r * ld + c);
        const bf16_t* s = reinterpret_cast<const bf16_t*>(&v);
        float* a = acc + (long)srcrow[r] * accld + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(a + e, bf2f(s[e]));
    }
}
extern "C" int kg_rows_scatter_add(const void* g, int ld, const int* srcrow, float* acc, int C, long nrows, int accld,
                                   void* stream) {
    KG_CHECK_ARG(g && srcrow && acc && C % 8 == 0 && ld % 8 == 0, "kg_rows_scatter_add: bad args");
    if (nrows == 0) return KG_OK;
    long total = nrows * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, ld, srcrow, acc,
                       C / 8, nrows, accld);
    KG_CHECK_LAUNCH("rows_scatter_add");
    return KG_OK;
}

// acc[srcrow[r]][c] += g[r][c] with packed bf16 atomics (global_atomic_pk_add_bf16): half the atomic count of the
// fp32 variant, no conversion pass; the accumulator is the bf16 gradient tensor itself (boxes rarely overlap more
// than a few times, so bf16 accumulation rounding stays below the bf16 storage noise of the gradient).
typedef __attribute__((ext_vector_type(2))) short s16x2;
__global__ void rows_scatter_add_bf16_kernel(const bf16_t* __restrict__ g, int ld, const int* __restrict__ srcrow,
                                             bf16_t* __restrict__ acc, int C8, long nrows, int accld) {
    long total = nrows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / C8; int c = (int)(i - r * C8) * 8;
        uint4 v = *reinterpret_cast<const uint4*>(g + r * ld + c);
        bf16_t* a = acc + (long)srcrow[r] * accld + c;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (w[e] & 0x7fff7fffu)   // skip +-0 pairs (ReLU-masked gradients are mostly zero)
                __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) s16x2*)(a + 2 * e),
                                                           __builtin_bit_cast(s16x2, w[e]));
    }
}
extern "C" int kg_rows_scatter_add_bf16(const void* g, int ld, const int* srcrow, void* acc, int C, long nrows, int accld,
                                        void* stream) {
    KG_CHECK_ARG(g && srcrow && acc && C % 8 == 0 && ld % 8 == 0 && accld % 8 == 0, "kg_rows_scatter_add_bf16: bad args");
    if (nrows == 0) return KG_OK;
    long total = nrows * (C / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(rows_scatter_add_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, ld, srcrow,
                       (bf16_t*)acc, C / 8, nrows, accld);
    KG_CHECK_LAUNCH("rows_scatter_add_bf16");
    return KG_OK;
}

// out[r][c] = bf16( acc[r][c] (+ addto[r][c]) )
__global__ void f32_to_bf16_rows_kernel(const float* __restrict__ acc, bf16_t* __restrict__ out, int C8, long rows,
                                        int ldout, const bf16_t* __restrict__ addto, int ldadd) {
    long total = rows * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i / C8; int c = (int)(i - r * C8) * 8;
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

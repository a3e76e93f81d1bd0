// evalmetrics.hip -- mask-IoU counts for the evaluation metrics.
//
// Reference: eval_parts.mask_iou (eval_parts.py:4-9) inside seg_evaluation (:98-150): for every detection, the IoU with every
// ground-truth instance whose box overlaps, each an H x W logical_and + three sums in NumPy (O(dets x GT x H x W), the
// dominant cost of eval.py at ~300 instances).  Here: masks live in HBM as bytes (any non-zero byte = foreground, rows padded to
// 16 bytes), one workgroup per (detection, GT) pair counts the intersection with 16-byte loads and a per-byte non-zero
// popcount, one workgroup per mask counts its area.  All results are exact integers; the IoU division stays on the host in
// float64 exactly as the reference does it.
#include "kg_common.h"

__device__ __forceinline__ uint32_t nz_bytes(uint32_t w) {   // bit 7 of every non-zero byte
    return (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
}
__device__ __forceinline__ int block_sum_256(int v) {
    __shared__ int red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void mask_area_kernel(const uint8_t* __restrict__ m, long ld, int* __restrict__ area) {
    const uint4* row = reinterpret_cast<const uint4*>(m + (long)blockIdx.x * ld);
    int c = 0;
    for (long i = threadIdx.x; i < ld / 16; i += 256) {
        const uint4 v = row[i];
        c += __builtin_popcount(nz_bytes(v.x)) + __builtin_popcount(nz_bytes(v.y)) + __builtin_popcount(nz_bytes(v.z)) + __builtin_popcount(nz_bytes(v.w));
    }
    c = block_sum_256(c);
    if (threadIdx.x == 0) area[blockIdx.x] = c;
}

__global__ __launch_bounds__(256) void mask_inter_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, const int2* __restrict__ pairs,
                                                         long ld, int* __restrict__ inter) {
    const int2 pr = pairs[blockIdx.x];
    const uint4* ra = reinterpret_cast<const uint4*>(a + (long)pr.x * ld);
    const uint4* rb = reinterpret_cast<const uint4*>(b + (long)pr.y * ld);
    int c = 0;
    for (long i = threadIdx.x; i < ld / 16; i += 256) {
        const uint4 u = ra[i], v = rb[i];
        c += __builtin_popcount(nz_bytes(u.x) & nz_bytes(v.x)) + __builtin_popcount(nz_bytes(u.y) & nz_bytes(v.y)) +
             __builtin_popcount(nz_bytes(u.z) & nz_bytes(v.z)) + __builtin_popcount(nz_bytes(u.w) & nz_bytes(v.w));
    }
    c = block_sum_256(c);
    if (threadIdx.x == 0) inter[blockIdx.x] = c;
}

// masks: device bytes [n][ld], ld % 16 == 0 (padding bytes zero); area: device int32 [n] = number of non-zero bytes per row
extern "C" int kg_mask_areas(const void* masks, int n, long ld, int* area, void* stream) {
    KG_CHECK_ARG(masks && area && n > 0 && ld > 0 && ld % 16 == 0, "kg_mask_areas: bad arguments");
    hipLaunchKernelGGL(mask_area_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)masks, ld, area);
    KG_CHECK_LAUNCH("mask_areas");
    return KG_OK;
}

// pairs: device int32 [npairs][2] = (row of a, row of b); inter: device int32 [npairs] = bytes non-zero in both rows
extern "C" int kg_mask_inter_pairs(const void* a, const void* b, const int* pairs, int npairs, long ld, int* inter, void* stream) {
    KG_CHECK_ARG(a && b && pairs && inter && npairs > 0 && ld > 0 && ld % 16 == 0, "kg_mask_inter_pairs: bad arguments");
    hipLaunchKernelGGL(mask_inter_kernel, dim3(npairs), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)a, (const uint8_t*)b, (const int2*)pairs, ld,
                       inter);
    KG_CHECK_LAUNCH("mask_inter_pairs");
    return KG_OK;
}

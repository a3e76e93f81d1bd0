// conv_args.h -- argument block shared by the gather implicit-GEMM kernels (conv_igemm.hip, conv_gather.hip).
#pragma once
#include "kg_common.h"

struct ConvArgs {
    const bf16_t* x; const bf16_t* w; const float* bias;
    bf16_t* y; float* y_f32; const bf16_t* res; const bf16_t* mask; const int2* rowdesc;
    int M, H, W, OH, OW;
    int ldx, ldy, ldres, ldmask;
    int Cout, K, cpt, cpt_magic, ntaps, KW, stride_log2, pad, dil;
    int mode;  // 0 dense forward, 1 dense transposed (dgrad), 2 ragged (+), 3 ragged transposed (-)
    int relu, f32_C;
    KMap km;                   // split-bf16 planes of x (kg_common.h): X-side offset of virtual channel unit q
    int yP, yps, rP, rps;      // planes / plane strides of the output rows and of the residual
    float* stat_part;          // != null: BatchNorm statistics of the output (kg_conv_stats_begin): partials [pixel tile][Cout][2]
    const float* oscale;       // != null: per-cout factor of the accumulator (kg_planes_t.oscale: folded inference BatchNorm)
    KgBStat bs;                // bs.x != null (with stat_part): BACKWARD statistics -- sums of (g, g * xhat) over the stored rows (kg_common.h)
};

// ---- BatchNorm statistics in the conv epilogue ---------------------------------------------------------------------------------
// A conv whose output goes straight into a train-mode BatchNorm (KGnet.py:82-93 after every backbone conv) also needs the
// per-channel sum and sum of squares of that output; the separate column-reduction pass re-read the whole tensor (43 launches,
// 0.5 ms per step).  Here every lane sums the rows it is about to store (kg_stat_add), the 16 lanes that share a cout range are
// combined with wave shuffles, the pixel waves of the workgroup through LDS in wave order, and the workgroup writes ONE partial
// per channel: part[(tile * Cout + c) * 2 + {0, 1}] -- the layout bn_finalize_train_kernel combines in double, in tile order.
// Everything is a fixed order: the statistics are reproducible like the two-pass ones (they are taken from the fp32 accumulators,
// before the split-bf16 rounding of the store).
template <int NV>
__device__ __forceinline__ void kg_stat_add(float (&s)[NV], float (&q)[NV], const float (&v)[NV]) {
#pragma unroll
    for (int e = 0; e < NV; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
}
// NW pixel waves (index wrow) share a cout tile of NC channels; this lane holds channels cl0 .. cl0+NV-1 of it.  red: LDS, NW * NC * 2
// floats, no longer read by anybody (the caller has passed a barrier since the last tile access).  part_tile = part + tile * Cout * 2.
template <int NW, int NC, int NV>
__device__ __forceinline__ void kg_stat_commit(float (&s)[NV], float (&q)[NV], float* red, int wrow, int cl0, int lm, float* part_tile,
                                               int c0, int Cout) {
#pragma unroll
    for (int e = 0; e < NV; ++e) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s[e] += __shfl_xor(s[e], o, 64); q[e] += __shfl_xor(q[e], o, 64); }
    }
    if (lm == 0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) { red[(wrow * NC + cl0 + e) * 2] = s[e]; red[(wrow * NC + cl0 + e) * 2 + 1] = q[e]; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NC * 2; t += blockDim.x) {
        const int c = t >> 1;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += red[(w * NC + c) * 2 + (t & 1)];
        if (c0 + c < Cout) part_tile[(long)(c0 + c) * 2 + (t & 1)] = acc;
    }
}

// Shared tail of every conv kernel: v[NV] = accumulators + bias of output channels cb .. cb+NV-1 of row m.
// (+ residual planes) -> ReLU -> ReLU mask (plane 0 of the masking tensor carries its sign) -> split-bf16 store.
struct EpiArgs {
    bf16_t* y; const bf16_t* res; const bf16_t* mask;
    int ldy, ldres, ldmask, Cout, relu, yP, yps, rP, rps;
};
template <int NV>
__device__ __forceinline__ void kg_conv_epi_apply(const EpiArgs& e, long m, int cb, float (&v)[NV]) {
    const bool full = cb + NV <= e.Cout;
    const int nvalid = full ? NV : e.Cout - cb;
    if (e.res) {
        const bf16_t* rq = e.res + m * e.ldres + cb;
        if (NV % 8 == 0 && full && ((reinterpret_cast<uintptr_t>(rq) & 15) == 0) && (e.rps % 8 == 0)) {
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) {
                float t[8];
                kg_load_planes8(rq + q * 8, e.rP, e.rps, t);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[q * 8 + k] += t[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (k < nvalid) v[k] += kg_load_planes1(rq + k, e.rP, e.rps);
        }
    }
    if (e.relu) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = kg_relu(v[k]);
    }
    if (e.mask) {
        const bf16_t* mp = e.mask + m * e.ldmask + cb;
        if (NV % 8 == 0 && full && ((reinterpret_cast<uintptr_t>(mp) & 15) == 0)) {
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) {
                const uint4 mv = *reinterpret_cast<const uint4*>(mp + q * 8);
                const bf16_t* ms = reinterpret_cast<const bf16_t*>(&mv);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[q * 8 + k] = bf2f(ms[k]) > 0.f ? v[q * 8 + k] : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (k < nvalid) v[k] = bf2f(mp[k]) > 0.f ? v[k] : 0.f;
        }
    }
}
template <int NV>
__device__ __forceinline__ void kg_conv_epi_store(const EpiArgs& e, long m, int cb, float (&v)[NV]) {
    const bool full = cb + NV <= e.Cout;
    const int nvalid = full ? NV : e.Cout - cb;
    if (e.y) {
        bf16_t* yp = e.y + m * e.ldy + cb;
        if constexpr (NV % 8 == 0) {
            if (full && ((reinterpret_cast<uintptr_t>(yp) & 15) == 0)) { kg_store_planes<NV>(yp, e.yP, e.yps, v, true); return; }
        }
        kg_store_planes_n<NV>(yp, e.yP, e.yps, v, nvalid);
    }
}
template <int NV>
__device__ __forceinline__ void kg_conv_epilogue(const EpiArgs& e, long m, int cb, float (&v)[NV]) {
    kg_conv_epi_apply<NV>(e, m, cb, v);
    kg_conv_epi_store<NV>(e, m, cb, v);
}
// Backward statistics (KgBStat, kg_common.h) in the epilogue of one row: (residual, ReLU, mask) -> s += g, q += g * xhat on the FINAL fp32
// gradient values, before their rounding to the stored planes (like the forward statistics: from the accumulators) -> plane store.
// mean / invstd of the 8-channel chunk are re-read per row (L1 hits) instead of living in 2 NV registers next to the accumulators: preloaded,
// they pushed conv_halo_kernel<3,1,8,0> from 234 registers to 256 + 56 spilled.  Cout % 64 == 0 (host-checked): whole 16-byte chunks.
template <int NV>
__device__ __forceinline__ void kg_conv_epilogue_bstat(const EpiArgs& e, const KgBStat& b, long m, int cb, float (&v)[NV], float (&s)[NV], float (&q)[NV]) {
    static_assert(NV % 8 == 0, "whole 16-byte chunks of x");
    kg_conv_epi_apply<NV>(e, m, cb, v);
    const bf16_t* xp = reinterpret_cast<const bf16_t*>(b.x) + m * b.ldx + cb;
#pragma unroll
    for (int c = 0; c < NV / 8; ++c) {
        float xs[8];
        kg_load_planes8(xp + c * 8, b.P, b.ps, xs);
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(b.mean + cb + c * 8), m1 = *reinterpret_cast<const f32x4*>(b.mean + cb + c * 8 + 4);
        const f32x4 i0 = *reinterpret_cast<const f32x4*>(b.invstd + cb + c * 8), i1 = *reinterpret_cast<const f32x4*>(b.invstd + cb + c * 8 + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float g = v[c * 8 + k];
            const float mu = k < 4 ? m0[k & 3] : m1[k & 3], is = k < 4 ? i0[k & 3] : i1[k & 3];
            s[c * 8 + k] += g;
            q[c * 8 + k] += g * ((xs[k] - mu) * is);
        }
    }
    kg_conv_epi_store<NV>(e, m, cb, v);
}
// the same with xhat of the row already in registers (the one-row-per-lane finishing kernels load x, mean and invstd BEFORE they add up their
// partial accumulators: the loads ride under that chain instead of following it)
template <int NV>
__device__ __forceinline__ void kg_bstat_xhat(const KgBStat& b, long m, int cb, float (&xh)[NV]) {
    const bf16_t* xp = reinterpret_cast<const bf16_t*>(b.x) + m * b.ldx + cb;
#pragma unroll
    for (int c = 0; c < NV / 8; ++c) {
        float xs[8];
        kg_load_planes8(xp + c * 8, b.P, b.ps, xs);
#pragma unroll
        for (int k = 0; k < 8; ++k) xh[c * 8 + k] = (xs[k] - b.mean[cb + c * 8 + k]) * b.invstd[cb + c * 8 + k];
    }
}
template <int NV>
__device__ __forceinline__ void kg_conv_epilogue_bstat_pre(const EpiArgs& e, long m, int cb, float (&v)[NV], float (&s)[NV], float (&q)[NV], const float (&xh)[NV]) {
    kg_conv_epi_apply<NV>(e, m, cb, v);
#pragma unroll
    for (int k = 0; k < NV; ++k) { s[k] += v[k]; q[k] += v[k] * xh[k]; }
    kg_conv_epi_store<NV>(e, m, cb, v);
}
static inline void kg_fill_planes(ConvArgs& a, const kg_planes_t& pp, int cin_pad_plane, int unit) {
    a.km = kg_make_kmap(cin_pad_plane, unit, pp.a_planes, pp.a_pstride, pp.w_planes);
    a.yP = pp.y_planes; a.yps = pp.y_pstride; a.rP = pp.b_planes; a.rps = pp.b_pstride;
    a.oscale = pp.oscale;
}
__device__ __forceinline__ EpiArgs kg_epi(const ConvArgs& a) {
    return EpiArgs{a.y, a.res, a.mask, a.ldy, a.ldres, a.ldmask, a.Cout, a.relu, a.yP, a.yps, a.rP, a.rps};
}

// conv_gather.hip: the deep-prefetch variant for cin_pad % 64 == 0 and bf16 row outputs.  stats_ok: the caller's launch may carry the
// BatchNorm statistics epilogue (the launcher claims the armed side channel with the tile count of the variant it picks).
int kg_launch_conv_gather(ConvArgs a, int cin_pad, hipStream_t st, bool stats_ok);
// conv_tiny.hip: K-split finishing pass shared by conv_tiny and conv_gather -- partials [tile64][Z][4 pixel groups][16 values][64 lanes] of
// 64-pixel x 64-cout tiles (tile64 = cout tile * npt64 + pixel tile) are added in slot order and run through the shared epilogue
// (+ BatchNorm statistics per 64-pixel tile when a.stat_part is set).  kg_splitk_scratch: the per-device scratch (slots of 4096 floats).
int kg_launch_splitk_finish(const ConvArgs& a, int Z, const float* part, int npt64, int nct64, hipStream_t st);
float* kg_splitk_scratch(long slots);
// conv_tiny.hip: split-K variant for launches with too few output tiles (tile = 6: cin_pad % 64 == 0, dense modes, rows output)
int kg_launch_conv_tiny(const ConvArgs& a, int cin_virt, hipStream_t st);
// conv_small.hip: direct VALU kernel for cin_pad == 8 (image / single-channel inputs)
int kg_launch_conv_small(const ConvArgs& a, hipStream_t st);

// conv_gather.hip -- gather implicit-GEMM convolution with an asynchronous LDS ring (cin_pad % 64 == 0, Cout >= 64).
//
// Serves the convs that cannot use the LDS-halo kernels: strided 3x3 / 1x1 convs of the ResNet trunk and their input
// gradients (KGnet.py:64-99), and the ragged deep levels of the per-box seg branch (KGnet.py:258-267, crops of a few
// pixels).  conv_igemm.hip recomputes the gather address of every 16-byte load (6 VALU per MFMA) and keeps only one
// K-step of global loads in flight (60 % of its wave cycles are waits).  Here:
//   * the K loop runs tap-outer / channel-chunk-inner: the source row of a pixel is resolved ONCE per tap (9 times for a
//     3x3, not K/64 times) and the 64-channel chunks of that tap are plain pointer increments;
//   * tiles go global -> LDS with LDS-direct loads (global_load_lds_dwordx4: no staging registers, no ds_write) into a
//     3-stage ring; two stages are in flight while one is multiplied, and the per-stage barrier waits with a COUNTED
//     s_waitcnt vmcnt(6) -- the 6 loads of the newest stage stay outstanding across the barrier (a __syncthreads() would
//     drain them).  The destination of an LDS-direct load is lane-linear, so the XOR swizzle of the 16-byte slots is
//     applied to the source chunk; padding / out-of-image taps read a zero line;
//   * workgroup = 8 waves = 256 pixels x 128 couts, wave tile 64 x 64 (16 MFMA per k-step), same fragment layouts and
//     epilogue as the other conv kernels (weights = MFMA A, each lane ends with 16 consecutive couts of one pixel).
#include "conv_args.h"
#include <stdlib.h>

__device__ uint4 kg_gather_zero_line[8];

#define KG_GLDS(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// K split (gridDim.z > 1, sk_part != null): workgroup z multiplies the stages [z * per, (z + 1) * per) of the (tap, virtual chunk) sequence and
// stores its raw fp32 accumulators as four 64 x 64 partial tiles in conv_tiny's slot layout; kg_launch_splitk_finish adds the slots in z order
// and runs the epilogue (+ statistics).  For launches whose output gives < 128 workgroups: the layer-2 / layer-3 convs of a batch-8 step
// (M = 8192 pixels x 256 couts = 64 workgroups, each walking up to 48 stages alone on a quarter of the chip).
//
// P2 (hi + lo planes on BOTH operands, the fp32-tolerance forward pass): the three products x_hi w_lo, x_lo w_hi, x_hi w_hi of a channel range
// share their operands, so a stage holds 32 channels of BOTH planes instead of 64 channels of one -- 128-byte rows [x_hi | x_lo] and
// [w_lo | w_hi] (the w_lo / w_hi copies of virtual planes 0 / 1 of the packed row), same ring, same swizzle, same fragment addresses -- and
// feeds 48 MFMAs instead of 32: 2/3 of the global -> LDS traffic and of the fragment reads per product (the virtual-plane walk stages every
// x plane and w_hi twice).  The low-order products have their own accumulator, added once before the epilogue.
//
// TCW = 1 (Cout <= 64: the 128 -> 64 1x1 convs at the c0 level, the 256 -> 64 3x3 of the seg branch): 4 waves, 128 pixels x 64 couts, wave
// tile 64 pixels x 32 couts.  On the 128-cout tile half of their MFMAs and weight loads were zeros, and its 144 KB of LDS leave one
// workgroup per CU -- a 128-channel 1x1 tile is four stages: a serial chain of load latencies with nothing else resident.  72 KB: two per CU.
template <bool P2, int TCW>
__global__ __launch_bounds__(256 * TCW) void conv_gather_kernel(const ConvArgs a, const int ncc, float* __restrict__ sk_part) {
    constexpr int NT = 256 * TCW, RQ = NT / 8, TP = 128 * TCW, TC = 64 * TCW, NS = 3, XB = TP * 128, WB = TC * 128, STAGE = XB + WB;
    constexpr int NI = 2 * TCW;        // 16-cout MFMA fragments of a wave (wave tile: 64 pixels x 16 NI couts)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave & (TP / 64 - 1), wcw = wave / (TP / 64);
    const int lm = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * TP, c0 = blockIdx.y * TC;
    const int tapK = P2 ? ncc * 96 : ncc * 64;                              // packed weight elements per tap (P2: ncc 32-channel chunks, 3 virtual planes)

    // ---- staging assignment: thread -> 16-byte slot cs of rows rr + RQ q ------------------------------------------------
    const int cs = tid & 7, rr = tid >> 3;
    const int xchunk = cs ^ ((rr >> 1) & 7);                               // source chunk of the X rows (key of row rr + RQ q)
    int py[4], px[4], ph[4], pw[4];
    long pbase[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = m0 + q * RQ + rr;
        py[q] = px[q] = 0; ph[q] = pw[q] = 0; pbase[q] = -1;
        if (m < a.M) {
            if (a.mode >= 2) {
                const int2 d = a.rowdesc[m];
                py[q] = d.x >> 16; px[q] = d.x & 0xffff; ph[q] = d.y >> 16; pw[q] = d.y & 0xffff; pbase[q] = m;
            } else {
                const int ohw = a.OH * a.OW;
                const int n = m / ohw, rem = m - n * ohw;
                py[q] = rem / a.OW; px[q] = rem - py[q] * a.OW; pbase[q] = (long)n * a.H * a.W; ph[q] = a.H; pw[q] = a.W;
            }
        }
    }
    const bf16_t* wrow[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { // packed rows are padded past Cout.  P2: slots 0-3 = w_lo (virtual plane 0), 4-7 = w_hi (virtual plane 1)
        const int r = q * RQ + rr;
        const int wchunk = cs ^ (2 * ((r >> 4) & 3) + ((r >> 1) & 1));     // source chunk of W row r
        wrow[q] = a.w + (long)(c0 + r) * a.K + (P2 ? (wchunk >> 2) * (ncc * 32) + (wchunk & 3) * 8 : wchunk * 8);
    }
    const int xfix = P2 ? (xchunk >> 2) * a.km.xps + (xchunk & 3) * 8 : xchunk * 8;               // P2: slots 0-3 = x plane 0 (hi), 4-7 = plane 1 (lo)
    const bf16_t* zline = reinterpret_cast<const bf16_t*>(kg_gather_zero_line) + cs * 8;
    const int smask = (1 << a.stride_log2) - 1;

    const int nstage_all = a.ntaps * ncc;
    const int per = (nstage_all + gridDim.z - 1) / gridDim.z;
    const int s_begin = blockIdx.z * per;
    const int nstage = (s_begin + per < nstage_all ? s_begin + per : nstage_all) - s_begin;     // (the launcher keeps every split non-empty)
    const bf16_t* xsrc[4];
    int i_tap = s_begin / ncc, i_cc = s_begin - i_tap * ncc, i_slot = 0;
    bool first = true;
    auto issue = [&]() {   // global -> LDS loads of the next (tap, 64-channel chunk) stage into ring slot i_slot
        if (i_cc == 0 || first) {   // new tap: resolve the source row of the thread's 4 pixels
            first = false;
            const int dy = i_tap / a.KW, dx = i_tap - dy * a.KW;
            const int dyo = dy - a.pad, dxo = dx - a.pad;
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
                xsrc[q] = ok ? a.x + row * a.ldx + xfix : nullptr;
            }
        }
        unsigned char* st = smem + i_slot * STAGE + wave * 1024;
        const int xo = P2 ? i_cc * 32 : a.km.xoff(i_cc);      // planes: virtual chunk -> (x plane, channel chunk)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bf16_t* src = xsrc[q] ? xsrc[q] + xo : zline;
            KG_GLDS(src, st + q * (NT * 16));
        }
        const long woff = (long)i_tap * tapK + i_cc * (P2 ? 32 : 64);
#pragma unroll
        for (int q = 0; q < 2; ++q) KG_GLDS(wrow[q] + woff, st + XB + q * (NT * 16));
        if (++i_cc == ncc) { i_cc = 0; ++i_tap; }
        i_slot = i_slot == NS - 1 ? 0 : i_slot + 1;
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------
    int a_off[NI][2], b_off[4][2];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = wcw * (16 * NI) + (lm >> 2) * (4 * NI) + i * 4 + (lm & 3);     // (lane g ends with the 4 NI consecutive couts 4 NI g ..)
        const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) a_off[i][s] = XB + r * 128 + (((4 * s + g) ^ key) * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wp * 64 + j * 16 + lm;
#pragma unroll
        for (int s = 0; s < 2; ++s) b_off[j][s] = r * 128 + (((4 * s + g) ^ ((r >> 1) & 7)) * 16);
    }
    f32x4 acc[NI][4], acl[P2 ? NI : 1][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (P2) acl[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    const unsigned lds0 = lds_addr(smem);
    issue();
    if (nstage > 1) issue();
    int c_slot = 0;
    for (int s = 0; s < nstage; ++s) {
        // stage s has landed once at most the 6 loads of stage s+1 are outstanding (loads complete in order)
        if (s + 1 < nstage) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // into the slot of stage s-1, which every wave has finished reading.  The first wave of every SIMD (waves 0..3) issues its loads
        // here, the second one behind the MFMAs of k-step 0 below: right after the barrier both would compute addresses at the same
        // time with the MFMA pipe idle (same idea as in wgrad_halo.hip)
        if (s + 2 < nstage && wave < NT / 128) issue();
        // fragment reads from inline asm with hand-counted waits (with an LDS-DMA pending hipcc waits lgkmcnt(0) for every LDS
        // read, kg_common.h): the 8 reads of k-step 1 stay in flight behind the 16 MFMAs of k-step 0
        const unsigned sb = lds0 + c_slot * STAGE;
        bf16x8 af[2][4], bfr[2][4];
        if constexpr (NI < 4) {        // (fragments a narrow wave tile never loads still pass through lgkm_wait)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = NI; i < 4; ++i) af[k][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned aa = sb + a_off[0][k], ba = sb + b_off[0][k];
            lds_rd128<0>(af[k][0], aa); lds_rd128<512>(af[k][1], aa);
            if constexpr (NI == 4) { lds_rd128<1024>(af[k][2], aa); lds_rd128<1536>(af[k][3], aa); }
            lds_rd128<0>(bfr[k][0], ba); lds_rd128<2048>(bfr[k][1], ba); lds_rd128<4096>(bfr[k][2], ba); lds_rd128<6144>(bfr[k][3], ba);
        }
        __builtin_amdgcn_s_setprio(1);
        if constexpr (P2) {
            // af[0] = w_lo, bfr[0] = x_hi, af[1] = w_hi, bfr[1] = x_lo (32 channels each): low-order products into acl
            lgkm_wait<NI + 4>(af[0], bfr[0]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acl[i][j] = KG_MFMA16(af[0][i], bfr[0][j], acl[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < nstage && wave >= NT / 128) { issue(); __builtin_amdgcn_sched_barrier(0); }
            lgkm_wait<4>(af[1], bfr[0]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = KG_MFMA16(af[1][i], bfr[0][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            lgkm_wait<0>(af[1], bfr[1]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acl[i][j] = KG_MFMA16(af[1][i], bfr[1][j], acl[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 0) lgkm_wait<NI + 4>(af[0], bfr[0]);
            else lgkm_wait<0>(af[1], bfr[1]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = KG_MFMA16(af[k][i], bfr[k][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);   // the MFMAs of k-step 0 stay in front of the wait for k-step 1
            if (k == 0 && s + 2 < nstage && wave >= NT / 128) { issue(); __builtin_amdgcn_sched_barrier(0); }
        }
        }
        __builtin_amdgcn_s_setprio(0);
        c_slot = c_slot == NS - 1 ? 0 : c_slot + 1;
    }

    if constexpr (P2) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += acl[i][j];
    }
    if constexpr (TCW == 2) {
    if (sk_part) {   // (uniform) K split: raw partial tiles, slot = ((tile64 * Z + z) * 4 + pixel group) * 1024 + value * 64 + lane
        const long npt64 = (long)gridDim.x * 4;
        const long tile64 = (long)(blockIdx.y * 2 + wcw) * npt64 + blockIdx.x * 4 + wp;
        float* slot = sk_part + (tile64 * gridDim.z + blockIdx.z) * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) slot[j * 1024 + (i * 4 + r) * 64 + lane] = acc[i][j][r];
        return;
    }
    }
    // ---- epilogue: lane owns pixel row m and couts cb .. cb + NV - 1 ------------------------------------------------------------
    constexpr int NV = 4 * NI;
    const int cb = c0 + wcw * (16 * NI) + g * NV;
    const bool stats = a.stat_part != nullptr;                 // (uniform)
    if (cb >= a.Cout && !stats) return;
    float bv[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
    float sv[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) sv[e] = (a.oscale && cb + e < a.Cout) ? a.oscale[cb + e] : 1.f;
    const EpiArgs ep = kg_epi(a);
    float ss[NV], sq[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) ss[e] = sq[e] = 0.f;
    const bool bst = stats && a.bs.x != nullptr;               // (uniform) backward statistics: sums of (g, g * xhat) -- KgBStat
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long m = (long)m0 + wp * 64 + j * 16 + lm;
        if (m >= a.M || cb >= a.Cout) continue;
        float v[NV];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] = KG_ACC(acc[i][j][r]) * sv[i * 4 + r] + bv[i * 4 + r];
        if (bst) { kg_conv_epilogue_bstat<NV>(ep, a.bs, m, cb, v, ss, sq); continue; }
        if (stats) kg_stat_add(ss, sq, v);
        kg_conv_epilogue<NV>(ep, m, cb, v);
    }
    if (stats) {
        __syncthreads();                                       // every wave has left the ring: its LDS becomes the combine buffer
        kg_stat_commit<TP / 64, TC>(ss, sq, reinterpret_cast<float*>(smem), wp, wcw * (16 * NI) + g * NV, lm, a.stat_part + (long)blockIdx.x * a.Cout * 2, c0, a.Cout);
    }
}

int kg_launch_conv_gather(ConvArgs a, int cin_pad, hipStream_t st, bool stats_ok) {
    constexpr int smem = 3 * (256 * 128 + 128 * 128);
    static KgPerDevice attr_done;
    if (attr_done.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)conv_gather_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        KG_HIP(hipFuncSetAttribute((const void*)conv_gather_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        KG_HIP(hipFuncSetAttribute((const void*)conv_gather_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem / 2));
        KG_HIP(hipFuncSetAttribute((const void*)conv_gather_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem / 2));
    }
    // hi + lo planes on both operands (3 virtual planes: x_hi w_lo, x_lo w_hi, x_hi w_hi): paired stages of 32 channels (KG_GATHER_P2=0: the
    // virtual-plane walk)
    constexpr int use_p2 = 1;
    const bool p2 = use_p2 && a.km.total == 3 * a.km.n && a.km.xtab == (0u | (1u << 2) | (0u << 4));
    const int cin_real = p2 ? cin_pad / 3 : cin_pad;
    // at most 64 couts and enough pixels to fill the chip with 128-pixel tiles: the 4-wave 128 x 64 variant, two workgroups per CU
    // (KG_GATHER_N64 = the least number of 128-pixel tiles; 0: never, 1: always -- the tests)
    static const int n64_tiles = getenv("KG_GATHER_N64") ? atoi(getenv("KG_GATHER_N64")) : 512;
    if (n64_tiles > 0 && a.Cout <= 64 && kg_cdiv(a.M, 128) >= n64_tiles) {
        dim3 grid1(kg_cdiv(a.M, 128), 1);
        if (stats_ok) a.stat_part = kg_conv_stats_claim((int)grid1.x, a.Cout);       // (BatchNorm statistics, when armed: per 128-pixel tile)
        if (a.stat_part) a.bs = kg_conv_stats().bs;
        if (p2) hipLaunchKernelGGL((conv_gather_kernel<true, 1>), grid1, dim3(256), smem / 2, st, a, cin_real / 32, (float*)nullptr);
        else hipLaunchKernelGGL((conv_gather_kernel<false, 1>), grid1, dim3(256), smem / 2, st, a, cin_pad / 64, (float*)nullptr);
        KG_CHECK_LAUNCH("conv_gather");
        kg_note_kernel(p2 ? "conv_gather_kernel<true, 1>" : "conv_gather_kernel<false, 1>");
        return KG_OK;
    }
    dim3 grid(kg_cdiv(a.M, 256), kg_cdiv(a.Cout, 128));
    // K split for under-filled launches (KG_GATHER_SPLIT: 0 = never; default: below 128 workgroups, >= 4 stages per split, <= 8 splits)
    static const int split_wgs = getenv("KG_GATHER_SPLIT") ? atoi(getenv("KG_GATHER_SPLIT")) : 128;
    const int wgs = (int)(grid.x * grid.y), nstage = p2 ? a.ntaps * (cin_real / 32) : a.ntaps * (cin_pad / 64);
    int Z = 1;
    if (split_wgs > 0 && wgs <= split_wgs && nstage >= 8) {
        Z = kg_cdiv(256, wgs);
        if (Z > nstage / 4) Z = nstage / 4;
        if (Z > 8) Z = 8;
        while (Z > 1 && (Z - 1) * kg_cdiv(nstage, Z) >= nstage) --Z;      // every split owns at least one stage
    }
    float* part = nullptr;
    const int npt64 = (int)grid.x * 4, nct64 = (int)grid.y * 2;
    if (Z > 1) {
        part = kg_splitk_scratch((long)npt64 * nct64 * Z);
        if (!part) Z = 1;
    }
    if (stats_ok) a.stat_part = kg_conv_stats_claim(Z > 1 ? npt64 : (int)grid.x, a.Cout);   // (BatchNorm statistics, when armed: per 64- or 256-pixel tile)
    if (a.stat_part) a.bs = kg_conv_stats().bs;
    grid.z = Z;
    if (p2) hipLaunchKernelGGL((conv_gather_kernel<true, 2>), grid, dim3(512), smem, st, a, cin_real / 32, Z > 1 ? part : (float*)nullptr);
    else hipLaunchKernelGGL((conv_gather_kernel<false, 2>), grid, dim3(512), smem, st, a, cin_pad / 64, Z > 1 ? part : (float*)nullptr);
    KG_CHECK_LAUNCH("conv_gather");
    kg_note_kernel(p2 ? "conv_gather_kernel<true, 2>" : "conv_gather_kernel<false, 2>");
    if (Z > 1) return kg_launch_splitk_finish(a, Z, part, npt64, nct64, st);
    return KG_OK;
}

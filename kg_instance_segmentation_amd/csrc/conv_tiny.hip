// conv_tiny.hip -- split-K implicit-GEMM convolution for launches that cannot fill the chip with output tiles.
//
// Single-image inference (BASELINE configs[4]: test.py runs one image at a time) leaves the deep backbone layers with a few
// thousand pixels: layer3 of a 512 x 512 image is 32 x 32 pixels x 256 channels -- 8 workgroups for conv_halo<3> (16 x 32-pixel x
// 64-cout tiles), 8 for conv_gather (256 pixels x 128 couts), each walking the whole K = 9 x 256 x 3 products alone (92 us on a
// chip that holds the layer's MFMAs in 2 us).  The output offers no more parallelism, the reduction dimension does:
//   * workgroup = 4 waves = ONE 64-pixel x 64-cout wave tile; the waves split K: the (plane product, tap, 64-channel chunk) units
//     are dealt round-robin, low-order products first in every wave (kg_plane_pairs order), 16 MFMAs per 32-wide k-step;
//   * operands go global -> registers in MFMA fragment layout (16 bytes per lane: row lm, k offset 8 g); a unit's 16 loads are
//     issued before its first MFMA.  There is no reuse inside a workgroup to stage through LDS -- the tensors of these launches
//     are L2-resident (<= a few MB);
//   * the four partial tiles meet in LDS ([wave][pixel group][cout][lane]: conflict-free both ways) and are summed in wave order
//     (fixed: reproducible); wave w finishes pixel group w with the epilogue every conv kernel shares (bias, folded inference
//     BatchNorm, residual, ReLU, mask, plane split).
//   * one workgroup per CU streams (64 + 64) rows x K through that CU's vector L1 (the bound of this kernel), so K is ALSO split across
//     gridDim.z workgroups when the tiles alone leave CUs idle: each writes its summed 64 x 64 fp32 partial to a scratch slot and a second,
//     tiny launch (conv_tiny_finish_kernel) adds the gridDim.z partials in slot order and runs the epilogue.  (Measured: finishing inside
//     the same launch -- device-scope fence + ticket, last workgroup reduces -- is 55 us SLOWER per launch: on this multi-L2 part every
//     workgroup's release fence writes its XCD's L2 back.  A kernel boundary is the cheap fence.)
// Dense forward (mode 0) and input gradient (mode 1), stride 1 / 2, any tap count, cin_pad % 64 == 0, rows output.
#include "conv_args.h"
#include <stdlib.h>

__global__ __launch_bounds__(256) void conv_tiny_kernel(const ConvArgs a, const int ncc, float* __restrict__ sk_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* part = reinterpret_cast<float*>(smem);                     // [4 waves][4 pixel groups][16 values][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int cin_virt = ncc * 64;                                    // virtual channels per tap of the packed weights

    // the lane's 4 pixels (fragment j: row m0 + 16 j + lm)
    int py[4], px[4];
    long pbase[4];
    const int ohw = a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + j * 16 + lm;
        py[j] = px[j] = 0; pbase[j] = -1;
        if (m < a.M) {
            const int n = m / ohw, rem = m - n * ohw;
            py[j] = rem / a.OW; px[j] = rem - py[j] * a.OW; pbase[j] = (long)n * a.H * a.W;
        }
    }
    // the lane's 4 weight rows (fragment i: cout c0 + 16 (lm >> 2) + 4 i + (lm & 3): the lane ends with 16 consecutive couts)
    const bf16_t* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wrow[i] = a.w + (long)(c0 + (lm >> 2) * 16 + i * 4 + (lm & 3)) * a.K + g * 8;   // packed rows are padded past Cout
    const int smask = (1 << a.stride_log2) - 1;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // units u = (v * ntaps + tap) * n + c  (v: plane product, low-order first; n = channel chunks per plane); virtual chunk = v * n + c
    const int n = a.km.n, nunits = a.ntaps * ncc;
    const int Z = gridDim.z;
    auto load = [&](int u, bf16x8 (&af)[2][4], bf16x8 (&bfr)[2][4]) {     // the 16 fragment loads of unit u (2 k-steps x (4 weight + 4 pixel) fragments)
        const int v = u / (a.ntaps * n), r = u - v * (a.ntaps * n);
        const int tap = r / n, c = r - tap * n;
        const int vc = v * n + c;
        const int dy = tap / a.KW, dx = tap - dy * a.KW;
        const int dyo = dy - a.pad, dxo = dx - a.pad;
        const int xo = a.km.xoff(vc) + g * 8;
        const long woff = (long)tap * cin_virt + vc * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = pbase[j] >= 0;
            int iy, ix;
            if (a.mode == 0) {
                iy = (py[j] << a.stride_log2) + dyo; ix = (px[j] << a.stride_log2) + dxo;
            } else {
                const int ty = py[j] - dyo, tx = px[j] - dxo;
                ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & smask) == 0;
                iy = ty >> a.stride_log2; ix = tx >> a.stride_log2;
            }
            ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            bfr[0][j] = z; bfr[1][j] = z;
            if (ok) {
                const bf16_t* src = a.x + (pbase[j] + (long)iy * a.W + ix) * a.ldx + xo;
                bfr[0][j] = *reinterpret_cast<const bf16x8*>(src);
                bfr[1][j] = *reinterpret_cast<const bf16x8*>(src + 32);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[0][i] = *reinterpret_cast<const bf16x8*>(wrow[i] + woff);
            af[1][i] = *reinterpret_cast<const bf16x8*>(wrow[i] + woff + 32);
        }
    };
    auto mma = [&](const bf16x8 (&af)[2][4], const bf16x8 (&bfr)[2][4]) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = KG_MFMA16(af[k][i], bfr[k][j], acc[i][j]);
    };
    // two operand sets: the loads of the next unit are in flight while the current one is multiplied (a wave has 2 .. 14 units: the
    // launch is a chain of memory round trips otherwise)
    const int ustep = 4 * Z;
    int u = blockIdx.z * 4 + wave;
    bf16x8 afA[2][4], bfA[2][4], afB[2][4], bfB[2][4];
    if (u < nunits) load(u, afA, bfA);
    while (u < nunits) {
        if (u + ustep < nunits) load(u + ustep, afB, bfB);
        mma(afA, bfA);
        u += ustep;
        if (u >= nunits) break;
        if (u + ustep < nunits) load(u + ustep, afA, bfA);
        mma(afB, bfB);
        u += ustep;
    }

    // ---- the four partial tiles meet in LDS; wave w sums pixel group w in wave order and runs the epilogue ---------------------
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[((wave * 4 + j) * 16 + i * 4 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    const int cb = c0 + g * 16;
    const long m = (long)m0 + wave * 16 + lm;
    float sum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float s = part[((0 * 4 + wave) * 16 + e) * 64 + lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) s += part[((k * 4 + wave) * 16 + e) * 64 + lane];
        sum[e] = s;
    }
    if (Z > 1) {   // (uniform) slot layout [tile][z][wave][16 values][64 lanes]: coalesced both ways; conv_tiny_finish_kernel completes the tile
        const long tile = (long)blockIdx.y * gridDim.x + blockIdx.x;
        float* slot = sk_part + ((tile * Z + blockIdx.z) * 4 + wave) * 1024;
#pragma unroll
        for (int e = 0; e < 16; ++e) slot[e * 64 + lane] = sum[e];
        return;
    }
    if (cb >= a.Cout || m >= a.M) return;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float sc = (a.oscale && cb + e < a.Cout) ? a.oscale[cb + e] : 1.f;
        const float bv = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
        v[e] = KG_ACC(sum[e]) * sc + bv;
    }
    const EpiArgs ep = kg_epi(a);
    kg_conv_epilogue<16>(ep, m, cb, v);
}

// second half of a K-split launch: grid = (pixel tiles, cout tiles), wave w = pixel group w; partials added in slot order z = 0, 1, ...
// (also the finishing pass of conv_gather's K split; a.stat_part: BatchNorm statistics of the output, one partial per 64-pixel tile)
__global__ __launch_bounds__(256) void conv_tiny_finish_kernel(const ConvArgs a, const int Z, const float* __restrict__ sk_part) {
    __shared__ float red[4 * 64 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, g = lane >> 4;
    const int cb = blockIdx.y * 64 + g * 16;
    const long m = (long)blockIdx.x * 64 + wave * 16 + lm;
    const long tile = (long)blockIdx.y * gridDim.x + blockIdx.x;
    const float* base = sk_part + (tile * Z * 4 + wave) * 1024;
    const bool stats = a.stat_part != nullptr;                 // (uniform)
    const bool live = cb < a.Cout && m < a.M;
    const bool bst = stats && a.bs.x != nullptr;               // (uniform) backward statistics: sums of (g, g * xhat) -- KgBStat
    float xh[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) xh[e] = 0.f;
    if (bst && live) kg_bstat_xhat<16>(a.bs, m, cb, xh);       // (issued before the partial sums: they ride under that chain)
    float sum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sum[e] = base[e * 64 + lane];
    for (int z = 1; z < Z; ++z)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum[e] += base[(long)z * 4096 + e * 64 + lane];
    if (!live && !stats) return;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float sc = (a.oscale && cb + e < a.Cout) ? a.oscale[cb + e] : 1.f;
        const float bv = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
        v[e] = KG_ACC(sum[e]) * sc + bv;
    }
    float ss[16], sq[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ss[e] = sq[e] = 0.f;
    const EpiArgs ep = kg_epi(a);
    if (bst) {
        if (live) kg_conv_epilogue_bstat_pre<16>(ep, m, cb, v, ss, sq, xh);
    } else {
        if (stats && live) kg_stat_add(ss, sq, v);
        if (live) kg_conv_epilogue<16>(ep, m, cb, v);
    }
    if (stats) kg_stat_commit<4, 64>(ss, sq, red, wave, g * 16, lm, a.stat_part + (long)blockIdx.x * a.Cout * 2, blockIdx.y * 64, a.Cout);
}

static constexpr int KG_SPLITK_MAX_SLOTS = 8192;               // 16 KB each
float* kg_splitk_scratch(long slots) {
    // one grow-never buffer per device, allocated on first use (launches on ONE stream at a time per device, like every scratch of this
    // library: the slots are free again when the finishing launch has run)
    static float* part[16] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || slots > KG_SPLITK_MAX_SLOTS) return nullptr;
    if (!part[dev] && hipMalloc((void**)&part[dev], (size_t)KG_SPLITK_MAX_SLOTS * 4096 * sizeof(float)) != hipSuccess) return nullptr;
    return part[dev];
}
int kg_launch_splitk_finish(const ConvArgs& a, int Z, const float* part, int npt64, int nct64, hipStream_t st) {
    hipLaunchKernelGGL(conv_tiny_finish_kernel, dim3(npt64, nct64), dim3(256), 0, st, a, Z, part);
    KG_CHECK_LAUNCH("conv_splitk_finish");
    return KG_OK;
}

// a.km must map 64-channel units (kg_make_kmap(cin_pad, 64, ...)); cin_virt = virtual channels per tap.
// Scratch of the K split across workgroups: one grow-never 64 MB buffer per device, allocated on first use (launches on ONE stream at a
// time per device, like every scratch of this library: the slots are free again when the finishing launch has run).
int kg_launch_conv_tiny(const ConvArgs& a, int cin_virt, hipStream_t st) {
    constexpr int smem = 4 * 4 * 16 * 64 * 4;
    constexpr int MAX_SLOTS = KG_SPLITK_MAX_SLOTS;
    static KgPerDevice attr_done;
    if (attr_done.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)conv_tiny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    const int tiles = kg_cdiv(a.M, 64) * kg_cdiv(a.Cout, 64);
    const int nunits = a.ntaps * (cin_virt / 64);
    constexpr int target = 256;   // workgroups a launch should reach
    constexpr int min_units = 2;    // units per wave below which a split does not pay
    int Z = kg_cdiv(target, tiles);
    if (Z > nunits / (4 * min_units)) Z = nunits / (4 * min_units);
    if (Z > 16) Z = 16;
    if (Z < 1 || (long)tiles * Z > MAX_SLOTS) Z = 1;
    float* part = nullptr;
    if (Z > 1) {
        part = kg_splitk_scratch((long)tiles * Z);
        if (!part) { kg_set_error("conv_tiny: no split-K scratch"); return KG_ERR_HIP; }
    }
    dim3 grid(kg_cdiv(a.M, 64), kg_cdiv(a.Cout, 64), Z);
    hipLaunchKernelGGL(conv_tiny_kernel, grid, dim3(256), smem, st, a, cin_virt / 64, part);
    KG_CHECK_LAUNCH("conv_tiny");
    kg_note_kernel("conv_tiny_kernel");
    if (Z > 1) return kg_launch_splitk_finish(a, Z, part, grid.x, grid.y, st);
    return KG_OK;
}

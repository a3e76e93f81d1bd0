// conv3_c64.hip -- persistent 3x3 stride-1 "same" convolution for 64 input channels.
//
// c0_conv.2, the bottleneck conv2 of layer1 and the level-0 convs of the per-box seg branch (KGnet.py:139-147, 64-99, 258-267)
// have only 64 input channels: in conv_halo.hip a workgroup stages a 78 KB halo for 9 taps of work, and that staging -- not the
// MFMAs -- bounds them (0.3-0.5 PFLOP/s).  Here
//   * all 9 taps of the [64 couts][64 ch] weights (72 KB) are loaded ONCE per workgroup and stay in LDS;
//   * workgroups are persistent (one per CU) and walk over 16x16-pixel tiles with TWO groups of 4 waves that alternate roles from
//     tile to tile: while one group runs the 9 taps of tile k (wave tile 64 pixels x 64 couts, 16 MFMAs per k-step, fragment reads
//     from inline asm one k-step ahead), the other prefetches the 18x18-pixel halo of ITS next tile k+1 by LDS-direct loads into
//     the second halo buffer and then runs the epilogue (bias / residual / ReLU / ReLU-mask, bf16 rows) and the stores of its
//     previous tile k-1 from the accumulators it kept.  One __syncthreads() per tile; the memory / VALU phase of a tile hides
//     behind the MFMA phase of its neighbour instead of following it (tools/micro/c3_parts.hip: the phases of the one-group
//     version added up, 6.7 us per tile and CU at c0's shape against 3.4 us of memory traffic and 3.2 us of MFMAs; now 5.1);
//   * fragment layouts, swizzles and the dense / ragged tile addressing are those of conv_halo.hip.
#include "kg_common.h"

__device__ uint4 kg_c3_zero_line[8];

struct C3Args {
    const bf16_t* x; const bf16_t* w; const float* bias;
    bf16_t* y; const bf16_t* res; const bf16_t* mask;
    const int4* tiletab;   // ragged: {row0, (h<<16)|w, (oy0<<16)|ox0, 0} per 16x16 tile
    int ntiles;
    int N, H, W, tiles_x, tiles_y;
    int ldx, Cout, ldy, ldres, ldmask, K, flip, relu;
};

#define KG_C3_GLDS(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__global__ __launch_bounds__(512) void conv3_c64_kernel(const C3Args a) {
    constexpr int HWD = 18, HPIX = HWD * HWD, HALO_BYTES = HPIX * 128, W_BYTES = 9 * 8192, ROW = HWD * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                        // [9 taps][64 couts][128 B]
    unsigned char* hb = smem + W_BYTES;              // 2 x [18 x 18 px][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3, gtid = tid & 255;
    const int lm = lane & 15, g = lane >> 4;
    const int total = a.tiletab ? a.ntiles : a.N * a.tiles_x * a.tiles_y;
    const int nt = (int)blockIdx.x < total ? (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;

    {   // weights: one 16-byte piece per thread and tap, swizzle on the source chunk
        const int r = tid >> 3, cs = tid & 7;
        const bf16_t* src = a.w + (long)(blockIdx.y * 64 + r) * a.K + (cs ^ (2 * ((r >> 4) & 3) + ((r >> 1) & 1))) * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) KG_C3_GLDS(src + t * 64, wl + t * 8192 + wave * 1024);
    }
    auto tile_geom = [&](int k, long& rowbase, int& Hd, int& Wd, int& oy0, int& ox0) {
        const int t = blockIdx.x + k * gridDim.x;
        if (a.tiletab) {
            const int4 tt = a.tiletab[t];
            rowbase = tt.x; Hd = tt.y >> 16; Wd = tt.y & 0xffff; oy0 = tt.z >> 16; ox0 = tt.z & 0xffff;
        } else {
            int bt = t;
            const int tx = bt % a.tiles_x; bt /= a.tiles_x;
            const int ty = bt % a.tiles_y; const int n = bt / a.tiles_y;
            oy0 = ty * 16; ox0 = tx * 16; Hd = a.H; Wd = a.W; rowbase = (long)n * a.H * a.W;
        }
    };
    // halo staging by the 256 threads of a group: piece e = gtid + 256 q -> halo pixel e >> 3, 16-byte slot e & 7 (lane-linear LDS destination)
    constexpr int NQ = (HPIX * 8 + 255) / 256;
    auto stage = [&](int k, int buf) {
        long rowbase; int Hd, Wd, oy0, ox0;
        tile_geom(k, rowbase, Hd, Wd, oy0, ox0);
        unsigned char* dst = hb + buf * HALO_BYTES;
        const bf16_t* tb = a.x + (rowbase + (long)(oy0 - 1) * Wd + (ox0 - 1)) * a.ldx;   // halo pixel (0, 0) (may lie outside the image)
#pragma unroll 1
        for (int q = 0; q < NQ; ++q) {
            const int e = gtid + q * 256;
            if (e < HPIX * 8) {
                const int p = e >> 3, cs = e & 7;
                const int hy = p / HWD, hx = p - hy * HWD;
                const int c = cs ^ (hx & 6);
                const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                const bf16_t* src = reinterpret_cast<const bf16_t*>(kg_c3_zero_line) + c * 8;
                if ((unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd) src = tb + (hy * Wd + hx) * a.ldx + c * 8;
                KG_C3_GLDS(src, dst + (q * 256 + gw * 64) * 16);
            }
        }
    };

    int a_off[2];
    {
        const int r = (lm >> 2) * 16 + (lm & 3);
        const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) a_off[s] = r * 128 + (((4 * s + g) ^ key) * 16);
    }
    int kb[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int fx = a.flip ? 2 - kx : kx;
        const int key = (lm + fx) & 6;
#pragma unroll
        for (int s = 0; s < 2; ++s) kb[kx][s] = ((gw * 4) * HWD + lm + fx) * 128 + (((4 * s + g) ^ key) * 16);
    }
    const int cb = blockIdx.y * 64 + g * 16;
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
    const bool full = cb + 16 <= a.Cout;
    const unsigned wl0 = lds_addr(wl), hb0 = lds_addr(hb);
    const int rowstep = a.flip ? -ROW : ROW;

    f32x4 acc[4][4];
    auto compute = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a0[4], b0[4], a1[4], b1[4];
        const unsigned hl = hb0 + buf * HALO_BYTES + (a.flip ? 2 * ROW : 0);
        auto ld = [&](bf16x8 (&af)[4], bf16x8 (&bf)[4], int n) {   // k-step n = (tap = n >> 1, s = n & 1)
            const int tap = n >> 1, s = n & 1, ky = tap / 3, kx = tap - 3 * ky;
            const unsigned aa = wl0 + tap * 8192 + a_off[s];
            lds_rd128<0>(af[0], aa); lds_rd128<512>(af[1], aa); lds_rd128<1024>(af[2], aa); lds_rd128<1536>(af[3], aa);
            const unsigned ba = hl + ky * rowstep + kb[kx][s];
            lds_rd128<0>(bf[0], ba); lds_rd128<ROW>(bf[1], ba); lds_rd128<2 * ROW>(bf[2], ba); lds_rd128<3 * ROW>(bf[3], ba);
        };
        auto mma = [&](const bf16x8 (&af)[4], const bf16x8 (&bf)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = KG_MFMA16(af[i], bf[j], acc[i][j]);
        };
        ld(a0, b0, 0);
#pragma unroll
        for (int n = 0; n < 18; n += 2) {
            ld(a1, b1, n + 1);
            lgkm_wait<8>(a0, b0);
            mma(a0, b0);
            if (n + 2 < 18) { ld(a0, b0, n + 2); lgkm_wait<8>(a1, b1); }
            else lgkm_wait<0>(a1, b1);
            mma(a1, b1);
        }
    };
    auto epilogue = [&](int k) {   // lane owns pixel (oy0 + 4*gw + j, ox0 + lm) and couts cb .. cb+15
        long rowbase; int Hd, Wd, oy0, ox0;
        tile_geom(k, rowbase, Hd, Wd, oy0, ox0);
        const int ox = ox0 + lm;
        if (!(cb < a.Cout && ox < Wd)) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + gw * 4 + j;
            if (oy >= Hd) continue;
            const long m = rowbase + (long)oy * Wd + ox;
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i * 4 + r] = KG_ACC(acc[i][j][r]) + bv[i * 4 + r];
            if (a.res) {
                const bf16_t* rq = a.res + m * a.ldres + cb;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (full || cb + e < a.Cout) v[e] += bf2f(rq[e]);
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = kg_relu(v[e]);
            }
            if (a.mask) {
                const bf16_t* mp = a.mask + m * a.ldmask + cb;
                if (full && ((reinterpret_cast<uintptr_t>(mp) & 15) == 0)) {
                    uint4 q0 = *reinterpret_cast<const uint4*>(mp), q1 = *reinterpret_cast<const uint4*>(mp + 8);
                    const bf16_t* ms0 = reinterpret_cast<const bf16_t*>(&q0);
                    const bf16_t* ms1 = reinterpret_cast<const bf16_t*>(&q1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[e] = bf2f(ms0[e]) > 0.f ? v[e] : 0.f; v[8 + e] = bf2f(ms1[e]) > 0.f ? v[8 + e] : 0.f; }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (full || cb + e < a.Cout) v[e] = bf2f(mp[e]) > 0.f ? v[e] : 0.f;
                }
            }
            bf16_t* yp = a.y + m * a.ldy + cb;
            if (full && ((reinterpret_cast<uintptr_t>(yp) & 15) == 0)) {
                *reinterpret_cast<uint4*>(yp) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
                *reinterpret_cast<uint4*>(yp + 8) = make_uint4(pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15]));
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (cb + e < a.Cout) yp[e] = f2bf(v[e]);
            }
        }
    };

    if (nt > 0 && grp == 0) stage(0, 0);
    for (int k = 0; k <= nt; ++k) {
        __syncthreads();                               // tile k's halo (and the weights) have landed; the buffer of tile k-1 is free
        if ((k & 1) == grp) {
            if (k < nt) compute(k & 1);
        } else {
            if (k + 1 < nt) stage(k + 1, (k + 1) & 1);   // first: the loads have the whole interval to land
            if (k >= 1) epilogue(k - 1);
        }
    }
}

// 3x3 stride-1 "same" conv / input gradient (flip) for cin_pad == 64, bf16 row output (Cout > 64: one grid row per 64 couts).  Dense: N images of H x W;
// ragged (tiletab16 != NULL): one {row0, (h<<16)|w, (oy0<<16)|ox0, 0} entry per 16x16 tile of a box.  w = packed [>=64 rows][K].
extern "C" int kg_conv3x3_c64(const void* x, const void* w, const float* bias, void* y, const void* res, const void* mask, int N, int H,
                              int W, int ldx, int Cout, int ldy, int ldres, int ldmask, int K, int flip, int relu, const int* tiletab16,
                              int ntiles, void* stream) {
    C3Args a;
    memset(&a, 0, sizeof(a));
    KG_CHECK_ARG(x && w && y, "kg_conv3x3_c64: null pointer");
    KG_CHECK_ARG(Cout >= 1 && ldx % 8 == 0 && ldx >= 64 && K >= 9 * 64, "kg_conv3x3_c64: needs 64 input channels");
    KG_CHECK_ARG((tiletab16 && ntiles > 0) || (N > 0 && H > 0 && W > 0), "kg_conv3x3_c64: empty problem");
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.res = (const bf16_t*)res; a.mask = (const bf16_t*)mask;
    a.tiletab = (const int4*)tiletab16; a.ntiles = ntiles; a.N = N; a.H = H; a.W = W; a.tiles_x = kg_cdiv(W, 16); a.tiles_y = kg_cdiv(H, 16);
    a.ldx = ldx; a.Cout = Cout; a.ldy = ldy; a.ldres = ldres; a.ldmask = ldmask; a.K = K; a.flip = flip; a.relu = relu;
    constexpr int smem = 9 * 8192 + 2 * 18 * 18 * 128;
    static KgPerDevice attr_done;
    if (attr_done.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)conv3_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    const int total = tiletab16 ? ntiles : N * a.tiles_x * a.tiles_y;
    const int ny = kg_cdiv(Cout, 64);                // 64-cout blocks: each (persistent) workgroup keeps one block's weights
    int grid = 256 / ny < 1 ? 1 : 256 / ny;          // persistent: one workgroup per CU (156 KB of LDS)
    if (grid > total) grid = total;
    hipLaunchKernelGGL(conv3_c64_kernel, dim3(grid, ny), dim3(512), smem, (hipStream_t)stream, a);
    KG_CHECK_LAUNCH("conv3x3_c64");
    kg_note_kernel("conv3_c64_kernel");
    return KG_OK;
}

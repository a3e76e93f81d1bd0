// conv7_narrow.hip -- persistent 7x7 stride-1 "same" input gradient of a NARROW conv: 8 or 16 of the dY channels carry data.
//
// The second-layer head convs of KGnet (KGnet.py:161-209, `.2` layers: C -> 5 keypoint maps, C -> 10 short offsets) have so few output channels
// that their input gradient dX[p][ci] = sum_{tap, co} dY[p - tap][co] W[co][ci][tap] is a conv with 5 / 10 INPUT channels.  On the LDS-halo kernel
// (conv_halo.hip) a workgroup stages a 107 KB halo of 64-channel dY lines and walks 49 taps whose 32-wide MFMA k-steps are 3/4 or 1/2 zeros; its
// GM = 3 / 4 variants pack 4 / 2 kernel columns into a k-step (14 / 28 k-steps instead of 49), which leaves 3 - 6 us of MFMAs inside a workgroup
// that costs ~14 us to set up, stage and drain with nothing else resident on its CU (17 / 20 us per workgroup measured, profiles/r06_narrow_*).
// Here, in the manner of conv3_c64.hip:
//   * the packed weights of a 64-channel output block -- 7 / 14 "virtual taps" of [64 rows][64 columns], a column = (kernel column, channel), written
//     by kg_pack_weight_narrow -- are loaded ONCE per workgroup and stay in LDS (57 / 115 KB);
//   * the halo of a 16 x 16-pixel tile holds only the live channels: 22 x 22 lines of 16 / 32 bytes (7.7 / 15.5 KB instead of 62 KB), two buffers;
//   * workgroups are persistent and walk over the tiles with two groups of 4 waves that alternate roles from tile to tile: one multiplies tile k
//     (wave tile 64 pixels x 64 channels, 14 / 28 k-steps of 16 MFMAs, fragment reads from inline asm one k-step ahead), the other issues the
//     LDS-direct loads of ITS next tile's halo and then runs the epilogue (ReLU mask, 16-bit rows) of its previous tile from the accumulators it kept;
//   * lane group g of a B fragment reads the pixel shifted by its own kernel column (kp: columns 4 s + g; short: 4 q + 2 s + (g >> 1), channel half
//     g & 1), exactly as conv_halo.hip's narrow variants do; the 8th column of a kernel row has zero weights and re-reads the 7th column's pixel.
// What is left is the output: 2 bytes x 64 channels written and 2 x 64 read (the mask) per pixel -- the launch is bound by those bytes.
#include "kg_common.h"

__device__ uint4 kg_c7n_zero_line[2];

struct C7nArgs {
    const bf16_t* x; const bf16_t* w;
    bf16_t* y; const bf16_t* mask;
    int N, H, W, tiles_x, tiles_y;
    int ldx, chan_lo, Cout, ldy, ldmask, K, flip;
};

#define KG_C7N_GLDS(src, dst) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <int CPT>      // live channels per kernel column: 8 (one 16-byte slot per halo line) or 16 (two)
__global__ __launch_bounds__(512) void conv7_narrow_kernel(const C7nArgs a) {
    constexpr int KS = 7, HWD = 16 + KS - 1, HPIX = HWD * HWD, LINE = CPT * 2, SLOTS = CPT / 8, HALO_BYTES = HPIX * LINE, ROW = HWD * LINE;
    constexpr int KSX = CPT / 8, NV = KS * KSX, W_BYTES = NV * 8192;     // virtual taps: KSX per kernel row, [64 rows][128 B] each
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                        // [NV][64 rows][128 B]
    unsigned char* hb = smem + W_BYTES;              // 2 x [22 x 22 px][LINE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3, gtid = tid & 255;
    const int lm = lane & 15, g = lane >> 4;
    const int total = a.N * a.tiles_x * a.tiles_y;
    const int nt = (int)blockIdx.x < total ? (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;

    {   // weights: one 16-byte piece per thread and virtual tap, swizzle on the source chunk (the A-side layout of conv_halo.hip / conv3_c64.hip)
        const int r = tid >> 3, cs = tid & 7;
        const bf16_t* src = a.w + (long)(blockIdx.y * 64 + r) * a.K + (cs ^ (2 * ((r >> 4) & 3) + ((r >> 1) & 1))) * 8;
#pragma unroll
        for (int t = 0; t < NV; ++t) KG_C7N_GLDS(src + t * 64, wl + t * 8192 + wave * 1024);
    }
    auto tile_geom = [&](int k, long& rowbase, int& oy0, int& ox0) {
        int bt = blockIdx.x + k * gridDim.x;
        const int tx = bt % a.tiles_x; bt /= a.tiles_x;
        const int ty = bt % a.tiles_y; const int n = bt / a.tiles_y;
        oy0 = ty * 16; ox0 = tx * 16; rowbase = (long)n * a.H * a.W;
    };
    // halo staging by the 256 threads of a group: piece e = gtid + 256 q -> halo pixel e / SLOTS, 16-byte slot e % SLOTS (lane-linear LDS destination)
    constexpr int NQ = (HPIX * SLOTS + 255) / 256;
    auto stage = [&](int k, int buf) {
        long rowbase; int oy0, ox0;
        tile_geom(k, rowbase, oy0, ox0);
        unsigned char* dst = hb + buf * HALO_BYTES;
#pragma unroll 1
        for (int q = 0; q < NQ; ++q) {
            const int e = gtid + q * 256;
            if (e < HPIX * SLOTS) {
                const int p = e / SLOTS, cs = e - p * SLOTS;
                const int hy = p / HWD, hx = p - hy * HWD;
                const int iy = oy0 + hy - 3, ix = ox0 + hx - 3;
                const bf16_t* src = reinterpret_cast<const bf16_t*>(kg_c7n_zero_line) + cs * 8;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) src = a.x + (rowbase + (long)iy * a.W + ix) * a.ldx + a.chan_lo + cs * 8;
                KG_C7N_GLDS(src, dst + (q * 256 + gw * 64) * 16);
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
    int kb[KSX][2];      // byte offset of the lane's pixel (fragment 0) + kernel column of its lane group, for virtual tap q of a kernel row and k-step s
#pragma unroll
    for (int q = 0; q < KSX; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            int kcol = CPT == 8 ? 4 * s + g : 4 * q + 2 * s + (g >> 1);
            if (kcol > KS - 1) kcol = KS - 1;          // (the zero-weight 8th column)
            const int fx = a.flip ? KS - 1 - kcol : kcol;
            kb[q][s] = ((gw * 4) * HWD + lm + fx) * LINE + (CPT == 8 ? 0 : (g & 1) * 16);
        }
    const int cb = blockIdx.y * 64 + g * 16;
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
        const unsigned hl = hb0 + buf * HALO_BYTES + (a.flip ? (KS - 1) * ROW : 0);
        auto ld = [&](bf16x8 (&af)[4], bf16x8 (&bf)[4], int n) {   // k-step n = (virtual tap n >> 1 = (ky, q), s = n & 1)
            const int vt = n >> 1, s = n & 1, ky = vt / KSX, q = vt - KSX * ky;
            const unsigned aa = wl0 + vt * 8192 + a_off[s];
            lds_rd128<0>(af[0], aa); lds_rd128<512>(af[1], aa); lds_rd128<1024>(af[2], aa); lds_rd128<1536>(af[3], aa);
            const unsigned ba = hl + ky * rowstep + kb[q][s];
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
        for (int n = 0; n < 2 * NV; n += 2) {
            ld(a1, b1, n + 1);
            lgkm_wait<8>(a0, b0);
            mma(a0, b0);
            if (n + 2 < 2 * NV) { ld(a0, b0, n + 2); lgkm_wait<8>(a1, b1); }
            else lgkm_wait<0>(a1, b1);
            mma(a1, b1);
        }
    };
    auto epilogue = [&](int k) {   // lane owns pixel (oy0 + 4*gw + j, ox0 + lm) and output channels cb .. cb+15
        long rowbase; int oy0, ox0;
        tile_geom(k, rowbase, oy0, ox0);
        const int ox = ox0 + lm;
        if (!(cb < a.Cout && ox < a.W)) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + gw * 4 + j;
            if (oy >= a.H) continue;
            const long m = rowbase + (long)oy * a.W + ox;
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i * 4 + r] = KG_ACC(acc[i][j][r]);
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

// Input gradient (flip = 1) or forward (flip = 0) of a 7x7 stride-1 "same" conv whose input rows x [N*H*W][ldx] carry data in the chan_slot (8 or 16) channels
// from chan_lo on; w = packed by kg_pack_weight_narrow (rows = output channels, K >= 7 * 8 * chan_slot); y rows [N*H*W][ldy] of Cout channels, zeroed where
// mask <= 0 (mask may be NULL).  Single 16-bit planes.
extern "C" int kg_conv7_narrow(const void* x, const void* w, void* y, const void* mask, int N, int H, int W, int ldx, int chan_lo, int chan_slot, int Cout, int ldy,
                               int ldmask, int K, int flip, void* stream) {
    C7nArgs a;
    memset(&a, 0, sizeof(a));
    KG_CHECK_ARG(x && w && y, "kg_conv7_narrow: null pointer");
    KG_CHECK_ARG((chan_slot == 8 || chan_slot == 16) && chan_lo % 8 == 0 && chan_lo >= 0 && chan_lo + chan_slot <= ldx && ldx % 8 == 0 && ldy % 8 == 0,
                 "kg_conv7_narrow: 8 or 16 live channels at a multiple of 8");
    KG_CHECK_ARG(K >= 7 * 8 * chan_slot && K % 8 == 0, "kg_conv7_narrow: K too small for the packed layout");
    KG_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout >= 1 && (!mask || ldmask % 8 == 0), "kg_conv7_narrow: empty problem");
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.mask = (const bf16_t*)mask;
    a.N = N; a.H = H; a.W = W; a.tiles_x = kg_cdiv(W, 16); a.tiles_y = kg_cdiv(H, 16);
    a.ldx = ldx; a.chan_lo = chan_lo; a.Cout = Cout; a.ldy = ldy; a.ldmask = ldmask; a.K = K; a.flip = flip;
    const int total = N * a.tiles_x * a.tiles_y;
    const int ny = kg_cdiv(Cout, 64);                // 64-channel output blocks: each (persistent) workgroup keeps one block's weights
    int grid = 256 / ny < 1 ? 1 : 256 / ny;          // persistent: one workgroup per CU
    if (grid > total) grid = total;
    hipStream_t st = (hipStream_t)stream;
    if (chan_slot == 8) {
        constexpr int smem = 7 * 8192 + 2 * 22 * 22 * 16;
        static KgPerDevice attr_done;
        if (attr_done.first()) KG_HIP(hipFuncSetAttribute((const void*)conv7_narrow_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        hipLaunchKernelGGL(conv7_narrow_kernel<8>, dim3(grid, ny), dim3(512), smem, st, a);
        kg_note_kernel("conv7_narrow_kernel<8>");
    } else {
        constexpr int smem = 14 * 8192 + 2 * 22 * 22 * 32;
        static KgPerDevice attr_done;
        if (attr_done.first()) KG_HIP(hipFuncSetAttribute((const void*)conv7_narrow_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        hipLaunchKernelGGL(conv7_narrow_kernel<16>, dim3(grid, ny), dim3(512), smem, st, a);
        kg_note_kernel("conv7_narrow_kernel<16>");
    }
    KG_CHECK_LAUNCH("conv7_narrow");
    return KG_OK;
}

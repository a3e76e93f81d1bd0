// wgrad_halo.hip -- weight gradient of stride-1 "same" 3x3 / 7x7 convolutions with LDS-resident tiles.
//
//   dW[co][tap][ci] = sum over pixels  dY[px][co] * X[px + d(tap)][ci]
//
// One workgroup (8 waves) owns 64 output channels x CI = 16*CIF input channels x ALL KS*KS taps and walks over
// 16x16 pixel tiles (its "split" of the image batch, or of the ragged box tiles).  Per tile it stages the dY tile
// [256 px][64 co] and the X halo [(16+KS-1)^2 px][CI] into LDS once; every tap re-reads the halo with a shifted
// window, so the 49 taps of a 7x7 conv cost one staging pass instead of 49.  The reduction dimension (pixels) is the
// strided one in NHWC, so both MFMA operands come from ds_read_b64_tr_b16 transpose reads.  Work split inside the
// workgroup: the (tap, ci-fragment) units are dealt round-robin to the 8 waves; each wave keeps its <= 7 units x 4
// co-fragments of fp32 accumulators in registers across all tiles and writes them once at the end to the
// [split][co][tap][ci] partial buffer that kg_wgrad_reduce sums (fixed order => reproducible).
// Tiles are double-buffered in LDS (the LDS-direct loads of tile t+1 are in flight while tile t is multiplied).
//
// LDS layouts are chosen so that (a) the 32-lane halves of a transpose read hit disjoint banks and (b) every fragment
// address is  lane-constant register + k-step immediate:  the k-step loop contains no address arithmetic.
//   k-step s = tile rows 2s, 2s+1; lane group G reads row 2s + (G&1), columns (G>>1)*8 + h*4 + (i>>2)   (h = 0,1)
//   dY tile : [256 px][128 B], 32-byte unit index XOR f(r), f(r) = ((r>>1)&1) | (((r>>4)&1)<<1)
//   X halo  : [HWD rows][PITCH], PITCH = HWD*32*CIF rounded up to 128 (mod 256) so consecutive rows alternate bank
//             halves; CIF == 4: 32-byte unit XOR (x & 3)
#include "kg_common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

struct WgHaloArgs {
    const bf16_t* x; const bf16_t* dy; float* dwp;
    const int4* tiletab;   // ragged mode: {row0, (h<<16)|w, (oy0<<16)|ox0, 0} per 16x16 tile
    int ntiles;
    int N, H, W, tiles_x, tiles_y, ldx, lddy;
    int Cin, Cout, cin_lim, cout_lim, nsplit;
    long split_stride;
    float* dbp;       // optional bias-gradient partials [nsplit][Cout]: db[co] = sum over pixels of dY (KGnet's convs with bias)
    WgPairs wp;       // split-bf16 operand planes (kg_common.h): the tile index runs over wp.n * tiles_total virtual tiles
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#ifndef KG_WG_LOOKAHEAD
#define KG_WG_LOOKAHEAD 2
#endif
__device__ __forceinline__ int dyf(int r) { return ((r >> 1) & 1) | (((r >> 4) & 1) << 1); }

template <int KS, int CIF>
struct WgGeom {
    static constexpr int PAD = KS / 2, HWD = 16 + KS - 1, T = KS * KS;
    static constexpr int XB = 32 * CIF;                                    // bytes per halo pixel
    static constexpr int RAW = HWD * XB;
    static constexpr int PITCH = ((RAW - 128 + 255) / 256) * 256 + 128;     // >= RAW and == 128 (mod 256)
    static constexpr int DY_BYTES = 256 * 128, X_BYTES = HWD * PITCH, BUF = DY_BYTES + X_BYTES;
    static constexpr int UNITS = T * CIF, UPW = (UNITS + 7) / 8;
};

// NCF = 16-wide output-channel fragments that are computed (4 = all 64; 1 / 3 for the 5-, 10- and 40-channel second
// head convs, whose dY tile is mostly padding: KGnet.py:161-209 `.2` layers).  Staging and fragment reads:
//   * the dY tile and the X halo of tile t+1 go global -> LDS with LDS-direct loads (global_load_lds_dwordx4) while tile t is
//     multiplied: no staging registers, no ds_write phase, no wave ever waits for a global load (the tile barrier does);
//     the destination of a wave's load is lane-linear, so the 32-byte-unit swizzles are applied to the SOURCE piece;
//   * the transpose reads are issued from inline asm, two (tap, ci-fragment) units ahead of the MFMAs, behind hand-counted
//     s_waitcnt lgkmcnt(N): with an LDS-DMA pending hipcc turns every LDS wait into lgkmcnt(0) (kg_common.h), which made a first
//     LDS-direct version with compiler-scheduled reads 8 % SLOWER than staging through registers (this one: 3-4 % faster).
__device__ uint4 kg_wg_zero_line[8];

template <int OFF>
__device__ __forceinline__ void lds_rd_tr64(bf16x4& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait_n() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void tie(bf16x4& v) { asm volatile("" : "+v"(v)); }

template <int KS, int CIF, int NCF = 4, bool BIAS = false>
__global__ __launch_bounds__(512) void wgrad_halo_kernel(const WgHaloArgs a) {
    using GE = WgGeom<KS, CIF>;
    constexpr int PAD = GE::PAD, HWD = GE::HWD, T = GE::T, XB = GE::XB, PITCH = GE::PITCH;
    constexpr int DY_BYTES = GE::DY_BYTES, BUF = GE::BUF, UNITS = GE::UNITS, UPW = GE::UPW;
    constexpr int DQ = 256 * 8 / 512;                   // dY 16-byte slots per thread (4)
    constexpr int XROW = PITCH / 16, XREAL = HWD * 2 * CIF, XSL = HWD * XROW, XQ = (XSL + 511) / 512;
    static_assert(PITCH % 16 == 0 && BUF % 16 == 0 && DY_BYTES % 1024 == 0, "16-byte LDS-direct destinations");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int n_ci_tiles = (a.cin_lim + 16 * CIF - 1) / (16 * CIF);
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in linear-id order, and all
    // (co, ci) blocks of one pixel split read the SAME dY tiles / X halos.  Put the blocks of a split on one XCD so that its
    // L2 fetches each tile once instead of all 8 L2s fetching it.
    int blk = blockIdx.x, split = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int L = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = L & 7, slot = L >> 3;
        blk = slot % gridDim.x; split = (slot / gridDim.x) * 8 + xcd;
    } else if (((gridDim.x * gridDim.y) & 7) == 0 && gridDim.x >= 16) {
        // Fewer than 8 (or an odd number of) splits -- the wide heads: 192 .. 768 (co, ci) blocks x 4 splits.  Dealt round-robin, every XCD
        // fetched every dY slice and every X slice of every tile (8 x 632 KB per tile at 256 -> 768).  Give each XCD a CONTIGUOUS range of the
        // co-major work list instead: its workgroups then share 1 .. 2 dY slices (and the X slices of the tile, which every cout block needs
        // anyway) -- half the L2 fills.  (co block, split, ci block), ci fastest: the workgroups that run together share a dY tile.
        const int L = blockIdx.y * gridDim.x + blockIdx.x;
        const int per = (gridDim.x * gridDim.y) >> 3;
        const int j = (L & 7) * per + (L >> 3);
        const int per_co = n_ci_tiles * gridDim.y;
        const int cob = j / per_co, rem = j - cob * per_co;
        split = rem / n_ci_tiles;
        blk = cob * n_ci_tiles + (rem - split * n_ci_tiles);
    }
    const int ci0 = (blk % n_ci_tiles) * 16 * CIF, co0 = (blk / n_ci_tiles) * 64;
    const int G = lane >> 4, i16 = lane & 15;

    f32x4 acc[UPW][NCF];
#pragma unroll
    for (int q = 0; q < UPW; ++q)
#pragma unroll
        for (int c = 0; c < NCF; ++c) acc[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int BW = UNITS % 8, BQ = UNITS / 8;
    static_assert(BQ < UPW, "no free unit slot for the bias-gradient unit");
    const bool do_bias = BIAS && ci0 == 0 && wave == BW;
    f32x4 accb[BIAS ? NCF : 1];
#pragma unroll
    for (int c = 0; c < (BIAS ? NCF : 1); ++c) accb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (kg_h16)1.0f;

    // lane-constant fragment addresses of buffer 0 (absolute LDS byte addresses; k-step s adds an immediate, buffer 1 adds BUF)
    const unsigned lds0 = lds_addr(smem);
    unsigned ay[NCF][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = (G & 1) * 16 + (G >> 1) * 8 + h * 4 + (i16 >> 2);
#pragma unroll
        for (int c = 0; c < NCF; ++c) ay[c][h] = lds0 + r * 128 + ((c ^ dyf(r)) * 32) + (i16 & 3) * 8;
    }
    unsigned bx[UPW][2];
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
        int u = wave + 8 * q;
        if (u >= UNITS) u = UNITS - 1;          // idle slot: any valid address
        const int tap = u / CIF, f = u - tap * CIF;
        const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int xh = (G >> 1) * 8 + h * 4 + (i16 >> 2) + kx;
            const int unit = CIF == 1 ? 0 : (f ^ (xh & (CIF - 1)));
            bx[q][h] = lds0 + DY_BYTES + ((G & 1) + ky) * PITCH + xh * XB + unit * 32 + (i16 & 3) * 8;
        }
    }

    const int tiles_plane = a.tiletab ? a.ntiles : a.N * a.tiles_x * a.tiles_y;
    const int tiles_total = a.wp.n * tiles_plane;

    // per-thread staging coordinates (tile independent).  LDS slot e (16 bytes) of the dY tile is row r = e >> 3, piece p = e & 7
    // and holds source channel piece ((p >> 1) ^ f(r)) * 2 + (p & 1); slot e of the halo is row e / XROW, piece w = e % XROW
    // (w >= XREAL: row padding, never read) = pixel hx = w / (2 CIF), source piece ((c >> 1) ^ (hx & (CIF-1))) * 2 + (c & 1).
    // (packed: the three coordinates of a piece share one register -- they stay live across the whole tile loop)
    int dy_p[DQ];                 // ry | rx << 8 | source 16-byte piece << 16 (piece 255: channel beyond cout_lim -> zero line)
#pragma unroll
    for (int k = 0; k < DQ; ++k) {
        const int e = tid + k * 512, r = e >> 3, pc = e & 7;
        const int cp = (((pc >> 1) ^ dyf(r)) << 1) | (pc & 1);
        dy_p[k] = (r >> 4) | ((r & 15) << 8) | ((co0 + cp * 8 < a.cout_lim ? cp : 255) << 16);
    }
    int x_p[XQ];                  // (hy + 8) | (hx + 8) << 8 | source piece << 16 | flags << 24 (1: load, 2: channel inside cin_lim)
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
        const int e = tid + k * 512;
        const int hy = e / XROW, w = e - hy * XROW;
        const int hx = w / (2 * CIF), c = w - hx * (2 * CIF);
        const int unit = CIF == 1 ? 0 : ((c >> 1) ^ (hx & (CIF - 1)));
        const int sp = unit * 2 + (c & 1);
        const int fl = ((e < XSL && w < XREAL) ? 1 : 0) | ((ci0 + sp * 8 < a.cin_lim) ? 2 : 0);
        x_p[k] = ((hy - PAD + 8) & 255) | (((hx - PAD + 8) & 255) << 8) | (sp << 16) | (fl << 24);
    }
    const bf16_t* zline = reinterpret_cast<const bf16_t*>(kg_wg_zero_line);
    auto stage = [&](int vt, int buf) {
        const int pr = vt / tiles_plane, t = vt - pr * tiles_plane;     // (product, tile): uniform
        int oy0, ox0, Hd, Wd; long rowbase;
        if (a.tiletab) {
            const int4 tt = a.tiletab[t];
            rowbase = tt.x; Hd = tt.y >> 16; Wd = tt.y & 0xffff; oy0 = tt.z >> 16; ox0 = tt.z & 0xffff;
        } else {
            int bt = t;
            const int tx = bt % a.tiles_x; bt /= a.tiles_x;
            const int ty = bt % a.tiles_y; const int n = bt / a.tiles_y;
            oy0 = ty * 16; ox0 = tx * 16; Hd = a.H; Wd = a.W; rowbase = (long)n * a.H * a.W;
        }
        const long base = rowbase + (long)oy0 * Wd + ox0;
        const bf16_t* dyb = a.dy + a.wp.doff[pr] + base * a.lddy;
        const bf16_t* xb = a.x + a.wp.xoff[pr] + base * a.ldx;
        const int hrem = Hd - oy0, wrem = Wd - ox0;
        unsigned char* sy = smem + buf * BUF;
        unsigned char* sx = sy + DY_BYTES;
        int lz = 0;
        asm volatile("" : "+v"(lz));   // opaque zero: the unpacked coordinates below must not be hoisted out of the tile loop (registers)
#pragma unroll
        for (int k = 0; k < DQ; ++k) {
            const int dp = dy_p[k] | lz;
            const int ry = dp & 255, rx = (dp >> 8) & 255, cp = dp >> 16;
            const bf16_t* src = zline;
            if (ry < hrem && rx < wrem && cp != 255) src = dyb + (long)(ry * Wd + rx) * a.lddy + co0 + cp * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sy + (k * 512 + wave_u * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < XQ; ++k) {
            if (x_p[k] & (1 << 24)) {
                const int xp = x_p[k] | lz;
                const int hy = (xp & 255) - 8, hx = ((xp >> 8) & 255) - 8, sp = (xp >> 16) & 255;
                const bf16_t* src = zline;
                if ((unsigned)(oy0 + hy) < (unsigned)Hd && (unsigned)(ox0 + hx) < (unsigned)Wd && (xp & (2 << 24)))
                    src = xb + (long)(hy * Wd + hx) * a.ldx + ci0 + sp * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(sx + (k * 512 + wave_u * 64) * 16), 16, 0, 0);
            }
        }
    };

    // fragment reads of k-step s: dY^T fragments c (2 transpose reads each), X^T fragment of unit q (2 reads)
    auto rdA = [&](auto sc, bf16x4 (&f)[NCF][2]) {
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int c = 0; c < NCF; ++c) { lds_rd_tr64<S * 32 * 128>(f[c][0], ay[c][0]); lds_rd_tr64<S * 32 * 128>(f[c][1], ay[c][1]); }
    };
    auto rdB = [&](auto sc, auto qc, bf16x4 (&f)[2]) {
        constexpr int S = decltype(sc)::value, Q = decltype(qc)::value;
        lds_rd_tr64<S * 2 * PITCH>(f[0], bx[Q][0]); lds_rd_tr64<S * 2 * PITCH>(f[1], bx[Q][1]);
    };
    auto cat = [](const bf16x4 (&f)[2]) { return __builtin_shufflevector(f[0], f[1], 0, 1, 2, 3, 4, 5, 6, 7); };

    int t = split;
    int cur = 0;                                     // ay / bx always point into the current buffer
    if (t < tiles_total) stage(t, 0);
    for (; t < tiles_total; t += a.nsplit) {
        __syncthreads();                             // (s_waitcnt vmcnt(0) + barrier): this tile has landed, the other buffer is free
        // The next tile's loads are ~290 instructions per wave (tile geometry on the scalar unit, 7 address computations); right after the
        // barrier BOTH waves of a SIMD would run them at the same time with the MFMA pipe idle.  The first wave of every SIMD (waves 0..3)
        // issues its share here, the second one (waves 4..7) half-way through the tile, each under the other's MFMAs.
        const int tn = t + a.nsplit;
        if (tn < tiles_total && wave_u < 4) stage(tn, cur ^ 1);
        // The (k-step s, unit q) pairs form one sequence I = s * UPW + q; read group R_I = the 2 transpose reads of unit I's X^T
        // fragment, preceded by the 2 NCF reads of the k-step's dY^T fragments when q == 0.  R_{I+LA} is issued before the MFMAs
        // of unit I (LA = 2 units = 8 MFMAs of LDS latency cover; 3 measured 1 % slower), LDS reads return in order, so "R_I has
        // landed" is lgkmcnt(|R_{I+1}| + ... + |R_{I+LA}|), a compile-time constant.
        constexpr int LA = KG_WG_LOOKAHEAD;           // read groups in flight ahead of the MFMAs
        bf16x4 fa[2][NCF][2], fb[LA + 1][2];         // fragments: A by k-step parity, B by unit index mod (LA + 1)
        constexpr int NU = 8 * UPW;
        auto issue = [&](auto ic) {
            constexpr int I = decltype(ic)::value, S = I / UPW, Q = I % UPW;
            if constexpr (Q == 0) rdA(std::integral_constant<int, S>{}, fa[S & 1]);
            rdB(std::integral_constant<int, S>{}, std::integral_constant<int, Q>{}, fb[I % (LA + 1)]);
        };
        [&]<int... Ds>(std::integer_sequence<int, Ds...>) { (issue(std::integral_constant<int, Ds>{}), ...); }(std::make_integer_sequence<int, LA>{});
        auto step = [&](auto ic) {
            constexpr int I = decltype(ic)::value, S = I / UPW, Q = I % UPW;
            constexpr int ahead = []() { int n = 0; for (int d = 1; d <= LA; ++d) if (I + d < NU) n += 2 + ((I + d) % UPW == 0 ? 2 * NCF : 0); return n; }();
            if constexpr (I + LA < NU) issue(std::integral_constant<int, I + LA>{});
            lgkm_wait_n<ahead>();
            if constexpr (Q == 0) {
#pragma unroll
                for (int c = 0; c < NCF; ++c) { tie(fa[S & 1][c][0]); tie(fa[S & 1][c][1]); }
                if (BIAS && do_bias) {               // wave-uniform
#pragma unroll
                    for (int c = 0; c < (BIAS ? NCF : 1); ++c) accb[c] = KG_MFMA16(cat(fa[S & 1][c]), ones, accb[c]);
                }
            }
            tie(fb[I % (LA + 1)][0]); tie(fb[I % (LA + 1)][1]);
            if (Q < UNITS / 8 || wave + 8 * Q < UNITS) {   // (wave-uniform; only the last slot can be idle)
                const bf16x8 bfr = cat(fb[I % (LA + 1)]);
#pragma unroll
                for (int c = 0; c < NCF; ++c) acc[Q][c] = KG_MFMA16(cat(fa[S & 1][c]), bfr, acc[Q][c]);
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the unit's MFMAs between its wait and the next unit's reads
            if constexpr (I == 3 * UPW) {
                if (tn < tiles_total && wave_u >= 4) stage(tn, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (step(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, NU>{});
        {   // the other buffer becomes current
            const int delta = cur ? -BUF : BUF;
#pragma unroll
            for (int c = 0; c < NCF; ++c) { ay[c][0] += delta; ay[c][1] += delta; }
#pragma unroll
            for (int q = 0; q < UPW; ++q) { bx[q][0] += delta; bx[q][1] += delta; }
            cur ^= 1;
        }
    }

    if (BIAS && do_bias && i16 == 0) {     // every column of the unit's result holds the same sum
#pragma unroll
        for (int c = 0; c < (BIAS ? NCF : 1); ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + c * 16 + G * 4 + r;
                if (co < a.Cout) a.dbp[(long)split * a.Cout + co] = accb[c][r];
            }
    }
    float* out = a.dwp + (long)split * a.split_stride;
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
        const int u = wave + 8 * q;
        if (u >= UNITS) continue;
        const int tap = u / CIF, f = u - tap * CIF;
        const int ci = ci0 + f * 16 + i16;
#pragma unroll
        for (int c = 0; c < NCF; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + c * 16 + G * 4 + r;
                if (co < a.Cout && ci < a.Cin) out[((long)co * T + tap) * a.Cin + ci] = acc[q][c][r];
            }
    }
}

template <int KS, int CIF, int NCF = 4, bool BIAS = false>
static int launch_wg(const WgHaloArgs& a, hipStream_t st) {
    constexpr int smem = 2 * WgGeom<KS, CIF>::BUF;
    static KgPerDevice attr_done;
    if (attr_done.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)wgrad_halo_kernel<KS, CIF, NCF, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    const int n_ci = (a.cin_lim + 16 * CIF - 1) / (16 * CIF), n_co = (a.cout_lim + 63) / 64;
    hipLaunchKernelGGL((wgrad_halo_kernel<KS, CIF, NCF, BIAS>), dim3(n_ci * n_co, a.nsplit), dim3(512), smem, st, a);
    KG_CHECK_LAUNCH("wgrad_halo");
    KG_KNAME(kname, "wgrad_halo_kernel<%d, %d, %d, %s>", KS, CIF, NCF, BIAS ? "true" : "false");
    kg_note_kernel(kname);
    return KG_OK;
}

// Weight gradient of a dense stride-1 "same" KSxKS conv, KS in {3,7}.  x [N*H*W][ldx], dy [N*H*W][lddy] bf16 rows;
// dwp receives nsplit partial tensors [Cout][KS*KS][Cin] (fp32), split_stride elements apart.
extern "C" int kg_conv2d_wgrad_halo(const void* x, const void* dy, float* dwp, int N, int H, int W, int ldx, int lddy,
                                    int Cin, int Cout, int cin_lim, int cout_lim, int KS, int nsplit, long split_stride,
                                    const int* tiletab, int ntiles, float* dbp, const kg_planes_t* planes, void* stream) {
    // planes: a = x, b = dy (dbp, the fused bias gradient, needs single-plane dy: the all-ones unit would count every product)
    WgHaloArgs a;
    memset(&a, 0, sizeof(a));
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_conv2d_wgrad_halo: bad kg_planes_t");
    a.wp = kg_make_wgpairs(pp.a_planes, pp.a_pstride, pp.b_planes, pp.b_pstride);
    KG_CHECK_ARG(!dbp || a.wp.n == 1, "kg_conv2d_wgrad_halo: the fused bias gradient needs single-plane operands");
    KG_CHECK_ARG(x && dy && dwp, "kg_conv2d_wgrad_halo: null pointer");
    KG_CHECK_ARG(KS == 3 || KS == 7, "kg_conv2d_wgrad_halo: kernel size must be 3 or 7");
    KG_CHECK_ARG(ldx % 8 == 0 && lddy % 8 == 0 && cin_lim % 8 == 0 && cout_lim % 8 == 0, "kg_conv2d_wgrad_halo: ld/lim must be multiples of 8");
    KG_CHECK_ARG(nsplit >= 1 && ((tiletab && ntiles > 0) || (N > 0 && H > 0 && W > 0)), "kg_conv2d_wgrad_halo: bad sizes");
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dwp = dwp; a.N = N; a.H = H; a.W = W;
    a.tiles_x = kg_cdiv(W, 16); a.tiles_y = kg_cdiv(H, 16); a.ldx = ldx; a.lddy = lddy; a.Cin = Cin; a.Cout = Cout;
    a.cin_lim = cin_lim; a.cout_lim = cout_lim; a.nsplit = nsplit; a.split_stride = split_stride;
    a.tiletab = (const int4*)tiletab; a.ntiles = ntiles; a.dbp = dbp;
    hipStream_t st = (hipStream_t)stream;
    if (KS == 7) {
        if (dbp) {
            // 5- / 10-cout second head layers: 32 input channels per workgroup (the dY tile is staged once per 32 instead of 16)
            if (cout_lim <= 16 && cin_lim >= 32) return launch_wg<7, 2, 1, true>(a, st);
            if (cout_lim <= 16) return launch_wg<7, 1, 1, true>(a, st);
            if (cout_lim <= 48) return launch_wg<7, 1, 3, true>(a, st);
            return launch_wg<7, 1, 4, true>(a, st);
        }
        if (cout_lim <= 16) return launch_wg<7, 1, 1>(a, st);
        if (cout_lim <= 48) return launch_wg<7, 1, 3>(a, st);
        return launch_wg<7, 1>(a, st);
    }
    if (dbp) return cout_lim <= 16 ? launch_wg<3, 4, 1, true>(a, st) : launch_wg<3, 4, 4, true>(a, st);
    if (cout_lim <= 16) return launch_wg<3, 4, 1>(a, st);   // seg_head.2 (64 -> 1)
    return launch_wg<3, 4>(a, st);
}

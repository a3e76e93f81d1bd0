// conv_halo.hip -- stride-1 "same" 3x3 / 7x7 bf16 MFMA convolution with the input halo resident in LDS.
//
// KGnet's FLOPs are 86 % 7x7 stride-1 head convolutions (KGnet.py:161-209) and most of the rest 3x3 stride-1
// (KGnet.py:139-158); in an im2col-style implicit GEMM every tap re-stages the pixel tile.  Here one workgroup
// (default: 8 waves) owns a 16x32-pixel output tile and 64 output channels:
//   * the (16+KS-1) x (32+KS-1) x 64-channel input halo is staged into LDS ONCE per 64-channel chunk (7x7: by LDS-direct
//     loads, no staging registers) and re-read by all KS*KS taps (49x reuse for 7x7) with shifted ds_read_b128 addresses;
//   * only the [64][64] weight slice of a tap streams through an LDS ring.  7x7: 6 slots filled by LDS-direct loads, ONE raw
//     s_barrier per TWO taps behind a counted s_waitcnt vmcnt(1) (the newest slice stays in flight across the barrier);
//     3x3 and the wide variants: 3 slots, register prefetch three taps ahead, one __syncthreads() per tap.  MFMA fragments
//     of the next k-step are fetched while the current 16 MFMAs run (register double buffering);
//   * both tiles use XOR-swizzled 16-byte slots so the 16-lane ds_read_b128 groups hit 16 distinct slots.
// WC = 2, 3 (tuning variants): 16x16-pixel tile with 128 / 192 couts per workgroup.
// The same kernel computes the input gradient of such a conv (flip = 1, transposed-packed weights).
// Wave tile = 64 couts x 64 pixels (4 rows of the 16x16 tile), 16 v_mfma_f32_16x16x32_bf16 per 32-wide k-step.
#include "conv_args.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>
#ifndef KG_HALO_SETPRIO
#define KG_HALO_SETPRIO 0
#endif
#ifndef KG_NARROW_V2
#define KG_NARROW_V2 1      // narrow (kp / short) chunks of the grouped second-layer heads: kernel-column stages + sliding halo-row window
#endif

__device__ uint4 kg_halo_zero_line[8];   // 128 zero bytes: source of the padding pixels of a halo

struct HaloArgs {
    const bf16_t* x; const bf16_t* w; const float* bias;
    bf16_t* y; float* y_f32; const bf16_t* res; const bf16_t* mask;
    const int4* tiletab;   // ragged mode: one entry per workgroup {row0, (h<<16)|w, (oy0<<16)|ox0, 0}; rows of a box are raster-contiguous
    int ntiles;
    int N, H, W, tiles_x, tiles_y;
    int cin_pad, ldx, Cout, ldy, ldres, ldmask, K, flip, relu, f32_C, f32_hw;
    // grouped second-layer heads (GM = 1): per-head channel chunks, virtual-cout -> map-channel table, the 3 fp32 outputs
    int grp_chunks; const int* vmap; float* f32_b; float* f32_c;
    int xcd_map;      // 1: remap (blockIdx.x, blockIdx.y) so that the cout blocks of a pixel tile share an XCD (gridDim.x % 8 == 0)
    int head_split;   // GM: 1 = blockIdx.y is the head (small images: 3x the workgroups), 0 = one workgroup walks all heads
    // split-bf16 planes (kg_common.h): cin_pad counts the VIRTUAL channels of a tap (what the packed weights hold); km maps a
    // virtual 64-channel chunk to its (x plane, channel chunk); GM = 1: grp_chunks virtual chunks and grp_C real channels per head
    KMap km; int grp_C;
    float* stat_part; // != null (GM = 0, bf16 rows output): BatchNorm statistics of the output, partials [pixel tile][Cout][2] (conv_args.h)
    KgBStat bs;       // bs.x != null (with stat_part, flip = 1): BACKWARD statistics -- sums of (g, g * xhat) over the stored gradient rows (kg_common.h)
    int kp_raw;       // GM = 1: 1 = export the kp logits without the sigmoid of KGnet.py:300 (parity tests, logit-space consumers)
    int yP, yps, rP, rps;
    // 3-product input (x_hi * w_lo | x_lo * w_hi | x_hi * w_hi, kg_plane_pairs): walk order of the virtual chunks, see halo_set_walk
    int walk3;
    const float* oscale;   // != null (GM = 0): per-cout factor of the accumulator (kg_planes_t.oscale: folded inference BatchNorm)
    // GM = 1: blockIdx.z = (plane product: x_hi w_lo | x_lo w_hi | x_hi w_hi) * prod_split + part -- a workgroup walks ONE part (of prod_split) of
    // the chunks of ONE product of its head and stores raw fp32 partial maps part[z][N][55][H*W] (the bias rides on the last z);
    // heads2_finish_kernel adds the 3 * prod_split maps in float64.  Small maps: more workgroups; wide heads (C >= 256): a BLOCKED accumulation --
    // one fp32 MFMA chain per few chunks instead of 49 * C / 32 additions into one accumulator (tools/micro/mfma_accum.hip: the chain's
    // rounding error grows with sqrt(length); torch-CPU's fp32 convs, the reference's arithmetic, do not show that growth)
    int prod_split; float* part;
    // GM = 0, under-filled launches (the layer-2 / layer-3 3x3 convs of a batch-8 step: 64 .. 128 workgroups): blockIdx.z walks the z-th part of
    // the chunk sequence and stores its raw fp32 accumulators (kpart: [cout block][tile][z][wave][16 values][64 lanes]); conv_halo_finish_kernel
    // adds the parts in z order -- the order of the unsplit walk -- and runs the epilogue (+ statistics)
    int ksplit; float* kpart;
    int nb2;          // launch_halo<7 | 3, 1, 8, 0>: 1 = this launch covers whole 128-cout blocks on conv_halo7_w4_kernel<*, 2>, -1 = never split off such a launch
};

// Planed input with the three products of the half-plane policies.  The packed weights keep the plane-major virtual-channel layout of
// kg_plane_pairs (kg_common.h) and the walk stays "every low-order product before the first hi * hi product" (the fp32 accumulator is
// still small while two thirds of the additions happen: measured, a chunk-major walk that shares every x_hi halo costs 25 % of the
// train-mode parity margin).  Inside that constraint the walk is arranged so that ONE x_hi halo serves two products:
//   low-order phase:  chunk 0: x_lo * w_hi, x_hi * w_lo;  chunk 1: ...;  chunk n-1: x_lo * w_hi, x_hi * w_lo
//   hi * hi phase:    chunk n-1 (its x_hi halo is still in LDS: not staged again), n-2, ..., 0
// 64-channel layers (the c0 / c1 heads, the grouped second layers, the seg branch's 3x3) have n = 1: 2 stagings instead of 3.
static void halo_set_walk(HaloArgs& a, int xP, int wP) {
    int xi[6], wj[6];
    const int nv = kg_plane_pairs(xP, wP, xi, wj);
    constexpr int on = 1;
    a.walk3 = on && nv == 3 && xi[0] == 0 && xi[1] == 1 && xi[2] == 0;
}

// WPX = pixel waves: 4 -> 16x16 output tile, 8 -> 16 rows x 32 columns (two 16x16 halves side by side)
// GM = 1: the three second-layer head convs (KGnet.py:161-209 `.2`: C -> 5 / 10 / 40) as ONE launch over the fused hidden
// tensor [rows][3C].  The 64 "virtual" output channels are dealt to the four MFMA row groups i (couts 16q + 4i + r):
// kp -> group 0, short -> group 1, mid -> groups 2, 3 and the rest of group 0; a 64-channel input chunk that belongs to
// head h only runs the MFMA row groups of that head (1, 1 and 3 of 4), instead of three launches that each pad their
// 5 / 10 / 40 couts to a 64-cout tile.  Small images (head_split): blockIdx.y = head, a workgroup walks the chunks of one head and writes only its maps.
// BSK: the instantiation whose epilogue carries the BACKWARD BatchNorm statistics (KgBStat; 3 x 3, WC = 1 only) -- a kernel of its own so that the
// regular one keeps its register allocation (234 registers, no spills; with the statistics code inside: 256 + 8 spilled)
template <int KS, int WC, int WPX, int GM = 0, bool BSK = false>
__global__ __launch_bounds__(WC * WPX * 64) void conv_halo_kernel(const HaloArgs a) {
    // GM = 3 / 4: the input gradient of a NARROW conv (8 / 16 channels of the 64-channel dY rows carry data: the kp / short second-layer heads).  One
    // 32-wide MFMA k-step multiplies 4 / 2 kernel COLUMNS of a kernel row at once -- lane group g of the B fragment reads the 16-byte channel slot of
    // the pixel shifted by its own column (conv_small.hip's four-taps-per-k-step idea on the LDS halo) -- so a kernel row is KSX = 1 / 2 "virtual taps"
    // of 64 packed weight columns ((kx, channel), the 8th column zero) instead of 7 taps that are 7/8 or 3/4 zeros: 14 / 28 k-steps, not 49.
    constexpr int KSX = GM == 3 ? 1 : (GM == 4 ? 2 : KS);
    constexpr int PAD = KS / 2, TW = 4 * WPX, HWD = TW + KS - 1, HPIX = (16 + KS - 1) * HWD, TC = WC * 64, NT = WC * WPX * 64, T = KS * KSX;
    static_assert(GM < 3 || KS == 7, "the narrow variants are written for 7 x 7 kernels");
    constexpr int HALO_BYTES = HPIX * 128, WBUF_BYTES = TC * 128;
    // 7x7, 64-cout workgroups: the weight slices go global -> LDS with LDS-direct loads through a 6-slot ring (no staging registers,
    // no ds_write), two taps per barrier; the barrier is a raw s_barrier behind a COUNTED s_waitcnt vmcnt(1), so the newest slice
    // stays in flight across it (a __syncthreads() would drain it).
    constexpr bool GLW = KS == 7 && WC <= 2;
    constexpr int NSL = GLW ? 6 : 3;
    constexpr bool ASMRD = GLW && GM != 1;   // fragment reads from inline asm with hand-counted lgkmcnt (see lds_rd128); the grouped-heads variant spills with it
    constexpr int WPT = TC * 8 / NT;  // weight chunks per thread per tap (= 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;
    unsigned char* wbuf = smem + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WPX, wp = wave % WPX;
    const int lm = lane & 15, g = lane >> 4;
    int oy0, ox0, Hd, Wd;
    long rowbase;   // row index of pixel (0,0) of this tile's image / box
    // XCD-aware mapping: workgroups are dealt to the 8 XCDs (one L2 each) round-robin in linear-id order.  Give every XCD a
    // contiguous range of pixel tiles and let the cout blocks of a tile follow each other on it, so the halo of a tile (and
    // the columns it shares with its x neighbour) is fetched into that L2 once instead of once per cout block.
    int bix = blockIdx.x, biy = blockIdx.y;
    if (a.xcd_map) {
        const int L = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = L & 7, slot = L >> 3;
        biy = slot % gridDim.y;
        bix = xcd * (gridDim.x >> 3) + slot / gridDim.y;
    }
    if (a.tiletab) {
        const int4 tt = a.tiletab[bix];
        rowbase = tt.x; Hd = tt.y >> 16; Wd = tt.y & 0xffff; oy0 = tt.z >> 16; ox0 = tt.z & 0xffff;
    } else {
        int bt = bix;
        const int tx = bt % a.tiles_x; bt /= a.tiles_x;
        const int ty = bt % a.tiles_y; const int n = bt / a.tiles_y;
        oy0 = ty * 16; ox0 = tx * TW; Hd = a.H; Wd = a.W; rowbase = (long)n * a.H * a.W;
    }
    const int c0 = GM == 1 ? 0 : biy * TC;   // GM: blockIdx.y = head (kp / short / mid), all on the one 64-row weight tile

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Fragment byte offsets.  Everything lane-dependent is hoisted out of the tap loop: the inner loop was VALU-issue
    // bound (3.5 VALU per MFMA) when these were recomputed per tap.
    // weights: row permutation (lane ends with 16 consecutive couts), swizzle key from the row -> constant per (i, s)
    int a_off[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wc * 64 + (lm >> 2) * 16 + i * 4 + (lm & 3);
        const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) a_off[i][s] = r * 128 + (((4 * s + g) ^ key) * 16);
    }
    // pixels: kb[kx][s] = halo byte offset of the lane's pixel for fragment 0 + swizzled 16-byte chunk of k-step s for a
    // tap whose halo-x shift is fx(kx); fragment j adds the immediate j*HWD*128, the tap adds a scalar.
    const int xb = (wp >> 2) * 16 + lm;
    int kb[KS][2];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
        if constexpr (GM >= 3) {      // virtual tap q = kx of a kernel row, k-step s: lane group g -> kernel column kcol, channel slot of dY (kp: 0; short: 1, 2)
            if (kx < KSX) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    int kcol = GM == 3 ? 4 * s + g : 4 * kx + 2 * s + (g >> 1);
                    if (kcol > KS - 1) kcol = KS - 1;      // (the 8th column: its packed weights are zero; read the 7th column's pixel again)
                    const int slot = GM == 3 ? 0 : 1 + (g & 1);
                    const int fx = a.flip ? KS - 1 - kcol : kcol;
                    kb[kx][s] = (((wp & 3) * 4) * HWD + xb + fx) * 128 + ((slot ^ ((xb + fx) & 6)) * 16);
                }
            } else kb[kx][0] = kb[kx][1] = 0;
            continue;
        }
        const int fx = a.flip ? KS - 1 - kx : kx;
        const int key = (xb + fx) & 6;   // conflict-free for every tap shift under the ds_read_b128 lane groups {0-3,12-15,20-27} ...; ((x >> 1) & 7 was 2-way for 3 of 4 shifts)
#pragma unroll
        for (int s = 0; s < 2; ++s) kb[kx][s] = (((wp & 3) * 4) * HWD + xb) * 128 + (((4 * s + g) ^ key) * 16);
    }
    // weights: ab[slot][s] = ring slot base + row/chunk offset of fragment 0 (fragment i adds the immediate i*512: the
    // swizzle key of row r = (lm>>2)*16 + i*4 + (lm&3) does not depend on i)
    int ab[NSL][2];
#pragma unroll
    for (int q = 0; q < NSL; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) ab[q][s] = q * WBUF_BYTES + a_off[0][s];

    // weight staging assignment
    int w_row[WPT], w_lds[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        int e = tid + i * NT, r = e >> 3, c = e & 7;
        w_row[i] = r;
        w_lds[i] = r * 128 + ((c ^ (2 * ((r >> 4) & 3) + ((r >> 1) & 1))) * 16);
    }
    const int wc8 = (tid & 7) * 8;  // NT % 8 == 0: the channel chunk of a thread is the same for every i
    uint4 wreg[WPT];

    auto mma = [&](auto mk, const bf16x8 (&af)[4], const bf16x8 (&bfr)[4]) {
        constexpr int MK = decltype(mk)::value;
        if (KG_HALO_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((MK >> i) & 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = KG_MFMA16(af[i], bfr[j], acc[i][j]);
            }
        if (KG_HALO_SETPRIO) __builtin_amdgcn_s_setprio(0);
    };

    int ci_first = (GM == 1 && a.head_split) ? biy * a.grp_chunks : 0;
    int nchunks = (GM == 1 && a.head_split) ? (biy + 1) * a.grp_chunks : a.cin_pad / 64;
    const bool psplit = GM == 1 && a.prod_split;                 // (uniform)
    if (psplit) {
        const int prod = blockIdx.z / a.prod_split, prt = blockIdx.z - prod * a.prod_split;
        const int per = (a.km.n + a.prod_split - 1) / a.prod_split, pend = ci_first + (prod + 1) * a.km.n;
        ci_first += prod * a.km.n + prt * per;
        nchunks = ci_first + per < pend ? ci_first + per : pend;       // (an empty part stores zeros)
    }
    const bool ksplit = GM == 0 && a.ksplit > 1;                 // (uniform)
    if (ksplit) {
        const int per = (nchunks + a.ksplit - 1) / a.ksplit;
        ci_first = blockIdx.z * per;
        if (ci_first + per < nchunks) nchunks = ci_first + per;
    }
    const int ci_begin = ci_first;
    for (int ci = ci_first; ci < nchunks; ++ci) {
        __syncthreads();
        int cc = ci;                               // virtual chunk of this step (weights: channel offset cc * 64 of a tap)
        bool stage = true;                         // (uniform) false: the halo of the previous step is the one this product multiplies
        if (a.walk3 && !psplit) {
            const int n = a.km.n, grp = 3 * n;     // virtual chunks of one group (GM == 1: one head, else the whole tap): [x_hi w_lo | x_lo w_hi | x_hi w_hi]
            const int base = ci / grp * grp, q = ci - base;
            if (q < 2 * n) cc = base + ((q & 1) ? (q >> 1) : n + (q >> 1));
            else { cc = base + 2 * n + (grp - 1 - q); stage = q > 2 * n || ci == ci_begin; }
        }
        int xo;                                    // X-side element offset of this (virtual) chunk
        if constexpr (GM == 1) { const int hd = cc / a.grp_chunks; xo = hd * a.grp_C + a.km.xoff(cc - hd * a.grp_chunks); }
        else xo = a.km.xoff(cc);
        // ---- stage the halo of this 64-channel chunk ------------------------------------------------
        if (!stage) {
        } else if (KS == 3) {   // 3x3 (9 taps per staging): all global loads of the halo are issued before the first LDS store
            constexpr int HPT = (HPIX * 8 + NT - 1) / NT;
            uint4 hreg[HPT];
#pragma unroll
            for (int q = 0; q < HPT; ++q) {
                const int e = tid + q * NT;
                const int p = e >> 3, c = e & 7;
                const int hy = p / HWD, hx = p - hy * HWD;
                const int iy = oy0 + hy - PAD, ix = ox0 + hx - PAD;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (e < HPIX * 8 && (unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd)
                    v = *reinterpret_cast<const uint4*>(a.x + (rowbase + (long)iy * Wd + ix) * a.ldx + xo + c * 8);
                hreg[q] = v;
            }
#pragma unroll
            for (int q = 0; q < HPT; ++q) {
                const int e = tid + q * NT;
                const int p = e >> 3, c = e & 7;
                const int hx = p % HWD;
                if (e < HPIX * 8) *reinterpret_cast<uint4*>(halo + p * 128 + ((c ^ (hx & 6)) * 16)) = hreg[q];
            }
        } else
        {   // 7x7: LDS-direct loads (global_load_lds_dwordx4): the whole halo is in flight at once and needs no staging
            // registers next to the ~250 live ones of the tap loop.  The destination of a wave's load is lane-linear
            // (M0 base + lane * 16), so the XOR swizzle of the 16-byte channel slots is applied to the SOURCE slot
            // (an involution inside the pixel's 128-byte line); padding pixels read a zero line.
            constexpr int HPT = (HPIX * 8 + NT - 1) / NT;
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll 1
            for (int q = 0; q < HPT; ++q) {   // rolled: an unrolled loop materialises all 14 source addresses at once (spills)
                const int e = tid + q * NT;
                if (e < HPIX * 8) {
                    const int p = e >> 3, cs = e & 7;
                    const int hy = p / HWD, hx = p - hy * HWD;
                    const int c = cs ^ (hx & 6);
                    const int iy = oy0 + hy - PAD, ix = ox0 + hx - PAD;
                    const bf16_t* src = reinterpret_cast<const bf16_t*>(kg_halo_zero_line) + c * 8;
                    if ((unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd)
                        src = a.x + (rowbase + (long)iy * Wd + ix) * a.ldx + xo + c * 8;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(halo + (q * NT + wave_u * 64) * 16), 16, 0, 0);
                }
            }
        }
        if constexpr (GM == 1) {
            // ---- kp / short chunks (heads 0, 1): ONE MFMA row group is live, so a tap needs only its 16 weight rows (2 KB, not the
            // 8 KB slice of the ring below) and 4 MFMAs per k-step -- with the two-taps-per-barrier ring the chunk is bound by the
            // barrier cadence (16 MFMAs between barriers), not by MFMAs or LDS.  Here a whole KERNEL ROW of 16-row slices (7 taps,
            // 14 KB) is one ring stage: 3 stages in the ring's LDS, rows ky+1, ky+2 in flight while row ky is multiplied, one barrier
            // per 7 taps; fragment reads from inline asm one k-step ahead (5 reads per 4 MFMAs: the LDS pipe is the bound).
            const int head_n = cc / a.grp_chunks;                 // uniform
            if (head_n < 2) {
                constexpr int ROWB = 7 * 2048;
                const int wave_n = __builtin_amdgcn_readfirstlane(wave);
#if KG_NARROW_V2
                // Round 4: the stage of the narrow ring is a kernel COLUMN (the 7 taps (ky, kx) of one kx), not a kernel row, and the B side is
                // walked as a sliding window over halo ROWS: for a fixed (kx, k-step) the four output rows of a wave and the seven ky taps touch
                // only 10 distinct halo-row fragments -- row R serves every (output row j, tap ky) with j + ky = R.  A step (kx, k-step, ky)
                // needs one new weight fragment and one new halo row (four at ky = 0): 17 LDS reads per 28 MFMAs instead of 35 -- the kp /
                // short chunks were LDS-read bound (5 x ds_read_b128 per 4 MFMAs saturate the LDS pipe of a CU at the MFMA rate).
                const bf16_t* nsrc[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {                     // piece e of a column stage: tap ky = e >> 7, row rho = (e >> 3) & 15, 16-byte slot e & 7
                    const int e = tid + k * 512, kyl = e >> 7, rho = (e >> 3) & 15, sl = e & 7;
                    const int r = 16 * (rho >> 2) + 4 * head_n + (rho & 3);
                    const int key = 2 * (rho >> 2) + ((rho >> 1) & 1);
                    nsrc[k] = a.w + (long)r * a.K + (long)(kyl * KS) * a.cin_pad + cc * 64 + ((sl ^ key) * 8);
                }
                auto nload = [&](int kx) {                        // kernel column kx -> stage kx % 3 (896 pieces: waves 0..5 issue two loads, 6..7 one)
                    unsigned char* dst = wbuf + (kx % 3) * ROWB + wave_n * 1024;
                    const long ro = (long)kx * a.cin_pad;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc[0] + ro),
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    if (wave_n < 6)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc[1] + ro),
                                                         (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 0);
                };
                nload(0); nload(1);
                int a_n[2];
                {
                    const int keyl = 2 * (lm >> 2) + ((lm >> 1) & 1);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) a_n[s2] = lm * 128 + (((4 * s2 + g) ^ keyl) * 16);
                }
                const unsigned ldsn = lds_addr(smem);
                const unsigned pbase = ldsn + (((wp & 3) * 4) * HWD + xb) * 128;       // the lane's pixel of halo row 0 of its wave (tap kx = 0)
                auto narrow = [&](auto hc) {
                    constexpr int HD = decltype(hc)::value;
#pragma unroll 1
                    for (int kx = 0; kx < KS; ++kx) {
                        if (kx + 1 < KS) {                        // column kx has landed; the loads of column kx+1 may stay in flight
                            if (wave_n < 6) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        if (kx + 2 < KS) nload(kx + 2);           // into the stage of column kx-1, which every wave has left
                        const unsigned wa = ldsn + HALO_BYTES + (kx % 3) * ROWB;
                        const int key = (xb + kx) & 6;            // (forward only: flip == 0)
                        unsigned hb[2];
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) hb[s2] = pbase + kx * 128 + (((4 * s2 + g) ^ key) << 4);
                        bf16x8 fa[3], fb[2][6];
                        // step T = (k-step T / 7, tap row ky = T % 7): its reads = the weight fragment + the halo rows it is the first to touch
                        auto issue = [&](auto tc) {
                            constexpr int T = decltype(tc)::value, S2 = T / KS, KY = T % KS;
                            lds_rd128<KY * 2048>(fa[T % 3], wa + a_n[S2]);
                            if constexpr (KY == 0) {
                                lds_rd128<0>(fb[S2][0], hb[S2]); lds_rd128<HWD * 128>(fb[S2][1], hb[S2]);
                                lds_rd128<2 * HWD * 128>(fb[S2][2], hb[S2]); lds_rd128<3 * HWD * 128>(fb[S2][3], hb[S2]);
                            } else {
                                lds_rd128<(KY + 3) * HWD * 128>(fb[S2][(KY + 3) % 6], hb[S2]);
                            }
                        };
                        auto nreads = [](int T) constexpr { return T >= 2 * KS ? 0 : (T % KS == 0 ? 5 : 2); };
                        auto stepn = [&](auto tc) {
                            constexpr int T = decltype(tc)::value, S2 = T / KS, KY = T % KS;
                            if constexpr (T + 2 < 2 * KS) issue(std::integral_constant<int, T + 2>{});
                            constexpr int PEND = nreads(T + 1) + nreads(T + 2);        // reads issued after this step's: they may stay in flight
                            asm volatile("s_waitcnt lgkmcnt(%5)"
                                         : "+v"(fa[T % 3]), "+v"(fb[S2][KY % 6]), "+v"(fb[S2][(KY + 1) % 6]), "+v"(fb[S2][(KY + 2) % 6]), "+v"(fb[S2][(KY + 3) % 6])
                                         : "n"(PEND));
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[HD][j] = KG_MFMA16(fa[T % 3], fb[S2][(KY + j) % 6], acc[HD][j]);
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        issue(std::integral_constant<int, 0>{});
                        issue(std::integral_constant<int, 1>{});
                        [&]<int... Ns>(std::integer_sequence<int, Ns...>) { (stepn(std::integral_constant<int, Ns>{}), ...); }(std::make_integer_sequence<int, 2 * KS>{});
                    }
                };
#else
                const bf16_t* nsrc[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {                     // piece e of a row: tap kx = e >> 7, row rho = (e >> 3) & 15, 16-byte slot e & 7
                    const int e = tid + k * 512, kxl = e >> 7, rho = (e >> 3) & 15, sl = e & 7;
                    const int r = 16 * (rho >> 2) + 4 * head_n + (rho & 3);
                    const int key = 2 * (rho >> 2) + ((rho >> 1) & 1);
                    nsrc[k] = a.w + (long)r * a.K + (long)kxl * a.cin_pad + cc * 64 + ((sl ^ key) * 8);
                }
                auto nload = [&](int ky) {                        // kernel row ky -> stage ky % 3 (896 pieces: waves 0..5 issue two loads, 6..7 one)
                    unsigned char* dst = wbuf + (ky % 3) * ROWB + wave_n * 1024;
                    const long ro = (long)ky * 7 * a.cin_pad;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc[0] + ro),
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    if (wave_n < 6)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc[1] + ro),
                                                         (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 0);
                };
                nload(0); nload(1);
                int a_n[2];
                {
                    const int keyl = 2 * (lm >> 2) + ((lm >> 1) & 1);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) a_n[s2] = lm * 128 + (((4 * s2 + g) ^ keyl) * 16);
                }
                const unsigned ldsn = lds_addr(smem);
                auto narrow = [&](auto hc) {
                    constexpr int HD = decltype(hc)::value;
#pragma unroll 1
                    for (int ky = 0; ky < KS; ++ky) {
                        if (ky + 1 < KS) {                        // row ky has landed; the loads of row ky+1 may stay in flight
                            if (wave_n < 6) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        if (ky + 2 < KS) nload(ky + 2);           // into the stage of row ky-1, which every wave has left
                        const unsigned wa = ldsn + HALO_BYTES + (ky % 3) * ROWB;
                        const unsigned hrow = ldsn + ky * (HWD * 128);
                        bf16x8 fa[2], fb[2][4];
                        auto rd = [&](auto nc) {
                            constexpr int N = decltype(nc)::value, KX = N >> 1, S2 = N & 1, B = N & 1;
                            lds_rd128<KX * 2048>(fa[B], wa + a_n[S2]);
                            const unsigned ba = hrow + kb[KX][S2];
                            lds_rd128<KX * 128>(fb[B][0], ba); lds_rd128<KX * 128 + HWD * 128>(fb[B][1], ba);
                            lds_rd128<KX * 128 + 2 * HWD * 128>(fb[B][2], ba); lds_rd128<KX * 128 + 3 * HWD * 128>(fb[B][3], ba);
                        };
                        auto stepn = [&](auto nc) {
                            constexpr int N = decltype(nc)::value, B = N & 1;
                            if constexpr (N + 1 < 2 * KS) { rd(std::integral_constant<int, N + 1>{}); asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); }
                            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            asm volatile("" : "+v"(fa[B]), "+v"(fb[B][0]), "+v"(fb[B][1]), "+v"(fb[B][2]), "+v"(fb[B][3]));
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[HD][j] = KG_MFMA16(fa[B], fb[B][j], acc[HD][j]);
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        rd(std::integral_constant<int, 0>{});
                        [&]<int... Ns>(std::integer_sequence<int, Ns...>) { (stepn(std::integral_constant<int, Ns>{}), ...); }(std::make_integer_sequence<int, 2 * KS>{});
                    }
                };
#endif
                if (head_n == 0) narrow(std::integral_constant<int, 0>{});
                else narrow(std::integral_constant<int, 1>{});
                continue;
            }
        }
        auto wload = [&](int tap) {
#pragma unroll
            for (int i = 0; i < WPT; ++i)
                wreg[i] = *reinterpret_cast<const uint4*>(a.w + (long)(c0 + w_row[i]) * a.K + (long)tap * a.cin_pad + cc * 64 + wc8);
        };
        auto wstore = [&](int slot) {
#pragma unroll
            for (int i = 0; i < WPT; ++i) *reinterpret_cast<uint4*>(wbuf + slot * WBUF_BYTES + w_lds[i]) = wreg[i];
        };
        auto wstore_at = [&](int slot_bytes) {
#pragma unroll
            for (int i = 0; i < WPT; ++i) *reinterpret_cast<uint4*>(wbuf + slot_bytes + w_lds[i]) = wreg[i];
        };
        // Weight ring of 3 slots: at the start of tap t slots t%3 and (t+1)%3 are visible, W(t+2) is in registers.
        const bf16_t* wsrc[WPT];           // GLW: this thread's 16-byte pieces of the tap's [TC couts][64 ch] slice
        auto wglds = [&](int slot_bytes) {
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc[i],
                                                 (__attribute__((address_space(3))) void*)(wbuf + slot_bytes + (i * NT + __builtin_amdgcn_readfirstlane(wave) * 64) * 16), 16, 0, 0);
                wsrc[i] += a.cin_pad;
            }
        };
        auto wait_newest = [&]() {         // all but the WPT loads of the newest slice have landed
            if constexpr (WPT == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        };
        static_assert(!GLW || WPT <= 2, "counted waits are written for 1 or 2 loads per slice");
        if constexpr (GLW) {
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int e = tid + i * NT, r = e >> 3, cs = e & 7;
                wsrc[i] = a.w + (long)(c0 + r) * a.K + cc * 64 + (cs ^ (2 * ((r >> 4) & 3) + ((r >> 1) & 1))) * 8;
            }
            wglds(0); wglds(WBUF_BYTES); wglds(2 * WBUF_BYTES); wglds(3 * WBUF_BYTES);
            wait_newest();                                         // halo + taps 0..2 have landed; tap 3 may still be in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            wload(0); wstore(0);
            if (T > 1) { wload(1); wstore(1); }
            if (T > 2) wload(2);
            __syncthreads();
        }
        // Software pipeline over (tap, k-step): the fragments of the NEXT k-step are fetched from LDS while the 16
        // MFMAs of the current one run, so no wave waits on LDS right after the per-tap barrier.  The tap loop is
        // ky (rolled) x kx (unrolled): every lane-dependent address term is a precomputed register (kb / ab), the tap
        // offset is a scalar, fragment j / i offsets are instruction immediates -> ~4 VALU per tap.
        bf16x8 a0[4], b0[4], a1[4], b1[4];
        if constexpr (ASMRD) {         // fragments a masked variant never loads still pass through lgkm_wait
#pragma unroll
            for (int i = 0; i < 4; ++i) { a0[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; a1[i] = a0[i]; }
        }
        const unsigned lds0 = lds_addr(smem);
        int abr[NSL][2];               // ab rotated so that index kx % NSL is the ring slot of tap (ky,kx)
        int sbr[NSL];
#pragma unroll
        for (int q = 0; q < NSL; ++q) { abr[q][0] = ab[q][0]; abr[q][1] = ab[q][1]; sbr[q] = q * WBUF_BYTES; }
        int tapb = a.flip ? ((KS - 1) * HWD + (GM >= 3 ? 0 : KS - 1)) * 128 : 0;   // halo byte offset of tap (ky, kx = 0); narrow: of kernel row ky (the columns are in kb)
        const int sx = a.flip ? -128 : 128, sy = a.flip ? -HWD * 128 : HWD * 128;
        auto ldA = [&](auto mk, bf16x8 (&af)[4], int base) {
            constexpr int MK = decltype(mk)::value;
            if constexpr (ASMRD) {
                const unsigned ad = lds0 + HALO_BYTES + base;
                if (MK & 1) lds_rd128<0>(af[0], ad);
                if (MK & 2) lds_rd128<512>(af[1], ad);
                if (MK & 4) lds_rd128<1024>(af[2], ad);
                if (MK & 8) lds_rd128<1536>(af[3], ad);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((MK >> i) & 1) af[i] = *reinterpret_cast<const bf16x8*>(wbuf + base + i * 512);
            }
        };
        auto ldB = [&](bf16x8 (&bfr)[4], int tb, int kbv) {
            if constexpr (ASMRD) {
                const unsigned ad = lds0 + tb + kbv;
                lds_rd128<0>(bfr[0], ad); lds_rd128<HWD * 128>(bfr[1], ad); lds_rd128<2 * HWD * 128>(bfr[2], ad); lds_rd128<3 * HWD * 128>(bfr[3], ad);
            } else {
                const unsigned char* hb = halo + tb + kbv;
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(hb + j * (HWD * 128));
            }
        };
        // GLW: wait until the fragments (af, bfr) have landed; `newer` = a later k-step's NRD reads were issued after them
        auto lwait = [&](auto mk, bf16x8 (&af)[4], bf16x8 (&bfr)[4], bool newer) {
            if constexpr (ASMRD) {
                constexpr int MK = decltype(mk)::value;
                constexpr int NRD = 4 + (MK & 1) + ((MK >> 1) & 1) + ((MK >> 2) & 1) + ((MK >> 3) & 1);
                if (newer) lgkm_wait<NRD>(af, bfr);
                else lgkm_wait<0>(af, bfr);
            }
        };
        auto run_taps = [&](auto mk) {
        ldA(mk, a0, abr[0][0]); ldB(b0, tapb, kb[0][0]);
        int t = 0;
#pragma unroll (ASMRD ? KS : 1)   // 7x7: fully unrolled -> tap parity, ring slots and tails are compile-time, no branch merges (the waitcnt pass then counts lgkmcnt precisely)
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KSX; ++kx, ++t) {
                constexpr int dummy = 0; (void)dummy;
                const int cur = kx % NSL, nxt = (kx + 1) % NSL, st = (kx + 2) % NSL;
                if constexpr (GLW) {   // pair start (even t): taps t+4, t+5 into the slots of taps t-2, t-1, which every wave has left
                    if (!(t & 1)) {
                        if (t + 4 < T) wglds(sbr[(kx + 4) % NSL]);
                        if (t + 5 < T) wglds(sbr[(kx + 5) % NSL]);
                    }
                }
                const int nkx = (kx + 1 == KSX) ? 0 : kx + 1;
                const int tb = GM >= 3 ? tapb : tapb + kx * sx;
                const int ntb = (kx + 1 == KSX) ? tapb + sy : (GM >= 3 ? tapb : tapb + (kx + 1) * sx);
                if constexpr ((decltype(mk)::value & 16) == 0) {
                    ldA(mk, a1, abr[cur][1]); ldB(b1, tb, kb[kx][1]);
                    lwait(mk, a0, b0, true);
                    mma(mk, a0, b0);
                    if (t + 1 < T) { ldA(mk, a0, abr[nxt][0]); ldB(b0, ntb, kb[nkx][0]); }
                    lwait(mk, a1, b1, t + 1 < T);
                    mma(mk, a1, b1);
                } else {   // only k-step 0 of the chunk is non-zero: taps alternate between the two fragment buffers
                    if (kx + 1 == KSX) {
                        lwait(mk, a0, b0, false);
                        mma(mk, a0, b0);
                        if (t + 1 < T) { ldA(mk, a0, abr[nxt][0]); ldB(b0, ntb, kb[nkx][0]); }
                    } else if (kx % 2 == 0) {
                        ldA(mk, a1, abr[nxt][0]); ldB(b1, ntb, kb[nkx][0]);
                        lwait(mk, a0, b0, true);
                        mma(mk, a0, b0);
                    } else {
                        ldA(mk, a0, abr[nxt][0]); ldB(b0, ntb, kb[nkx][0]);
                        lwait(mk, a1, b1, true);
                        mma(mk, a1, b1);
                    }
                }
                if constexpr (GLW) {
                    if ((t & 1) && t + 1 < T) {   // pair end: taps t+1 .. t+3 must be visible after the barrier; tap t+4 may stay in flight
                        if (t + 4 < T) wait_newest();
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                } else {
                    if (t + 2 < T) {
                        wstore_at(sbr[st]);
                        if (t + 3 < T) wload(t + 3);
                    }
                    __syncthreads();
                }
            }
            tapb += sy;
            {   // KS taps per row: the ring slot of (ky+1, 0) is KS % NSL further -> rotate the slot tables
                constexpr int SH = KSX % NSL;
                int sb2[NSL], ab2[NSL][2];
#pragma unroll
                for (int q = 0; q < NSL; ++q) { sb2[q] = sbr[(q + SH) % NSL]; ab2[q][0] = abr[(q + SH) % NSL][0]; ab2[q][1] = abr[(q + SH) % NSL][1]; }
#pragma unroll
                for (int q = 0; q < NSL; ++q) { sbr[q] = sb2[q]; abr[q][0] = ab2[q][0]; abr[q][1] = ab2[q][1]; }
            }
        }
        };
        if constexpr (GM == 2) {
            // the kp / short cout blocks of the fused second-layer head dgrad (engine.heads_second): their weights are zero
            // for dY channels 32..63, so k-step 1 of the (single) chunk is skipped
            run_taps(std::integral_constant<int, 31>{});
        } else if constexpr (GM == 1) {
            const int head = cc / a.grp_chunks;
            if (head == 0) run_taps(std::integral_constant<int, 1>{});
            else if (head == 1) run_taps(std::integral_constant<int, 2>{});
            else run_taps(std::integral_constant<int, 13>{});
        } else {
            // fused second-layer head dgrad (engine.heads_second): the kp / short cout blocks only see dY channels 0..31
            run_taps(std::integral_constant<int, 15>{});
        }
    }

    if (ksplit) {   // raw partial accumulators of this part of the chunk walk; conv_halo_finish_kernel completes the tile
        float* slot = a.kpart + ((((long)biy * gridDim.x + bix) * a.ksplit + blockIdx.z) * (WC * WPX) + wave) * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slot[((i * 4 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
        return;
    }
    // ---- epilogue: lane owns pixel (oy0 + wp*4 + j, ox0 + lm) and couts cb .. cb+15 ----------------------
    const int cb = c0 + wc * 64 + g * 16;
    const bool stats = GM == 0 && a.stat_part != nullptr;      // (uniform)
    if (cb >= a.Cout && !stats) return;
    float ss[16], sq[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ss[e] = sq[e] = 0.f;
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = (a.bias && cb + e < a.Cout && !(psplit && blockIdx.z != gridDim.z - 1)) ? a.bias[cb + e] : 0.f;
    if (GM == 0 && a.oscale) {           // (uniform; the accumulators are scaled in place: no second 16-register table next to acc)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float s = cb + e < a.Cout ? a.oscale[cb + e] : 1.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[e >> 2][j][e & 3] *= s;
        }
    }
    const EpiArgs ep{a.y, a.res, a.mask, a.ldy, a.ldres, a.ldmask, a.Cout, a.relu, a.yP, a.yps, a.rP, a.rps};
    const int ox = ox0 + (wp >> 2) * 16 + lm;
    // backward statistics: sums of (g, g * xhat) -- KgBStat; 3 x 3 only (what follows a BatchNorm in KGnet.py:64-99; the 7 x 7 instantiations
    // keep their epilogue -- and their register allocation -- as they were)
    constexpr bool BST = BSK && GM == 0 && KS == 3;
    const bool bst = BST && stats && a.bs.x != nullptr;        // (uniform)
    int vm[16];
    if constexpr (GM == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) vm[e] = a.vmap[cb + e];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + (wp & 3) * 4 + j;
        if (oy >= Hd || ox >= Wd || cb >= a.Cout) continue;
        const long m = rowbase + (long)oy * Wd + ox;
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] = KG_ACC(acc[i][j][r]) + bv[i * 4 + r];
        if constexpr (BST) {
            if (bst) { kg_conv_epilogue_bstat<16>(ep, a.bs, m, cb, v, ss, sq); continue; }
        }
        if (stats) kg_stat_add(ss, sq, v);
        if constexpr (GM == 1) {   // fp32 NCHW export to the kp (sigmoid, KGnet.py:300) / short / mid maps
            const long hw = (long)a.H * a.W;
            const long nimg = rowbase / hw, pix = (long)oy * Wd + ox;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ch = vm[e];
                if (ch < 0 || (a.head_split && (ch < 5 ? 0 : ch < 15 ? 1 : 2) != biy)) continue;
                if (psplit) { a.part[(((long)blockIdx.z * a.N + nimg) * 55 + ch) * hw + pix] = v[e]; continue; }
                if (ch < 5) a.y_f32[(nimg * 5 + ch) * hw + pix] = a.kp_raw ? v[e] : 1.f / (1.f + expf(-v[e]));
                else if (ch < 15) a.f32_b[(nimg * 10 + ch - 5) * hw + pix] = v[e];
                else a.f32_c[(nimg * 40 + ch - 15) * hw + pix] = v[e];
            }
            continue;
        }
        if (a.y_f32) {
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = kg_relu(v[e]);
            }
            // dense: NCHW [N][f32_C][H*W]; ragged: [f32_C][total rows] (f32_hw = total rows, image index 0)
            const long hw = a.tiletab ? (long)a.f32_hw : (long)a.H * a.W;
            const long nimg = a.tiletab ? 0 : rowbase / hw, pix = a.tiletab ? m : (long)oy * Wd + ox;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (cb + e < a.Cout) a.y_f32[(nimg * a.f32_C + cb + e) * hw + pix] = v[e];
        }
        if (a.y) kg_conv_epilogue<16>(ep, m, cb, v);      // (last: the split store consumes v)
    }
    if (stats) {
        __syncthreads();                                       // every wave has left the halo and the ring: their LDS becomes the combine buffer
        kg_stat_commit<WPX, TC>(ss, sq, reinterpret_cast<float*>(smem), wp, wc * 64 + g * 16, lm, a.stat_part + (long)bix * a.Cout * 2, c0, a.Cout);
    }
}

// second half of a chunk-split launch (HaloArgs.ksplit): same grid (pixel tiles, cout blocks) and thread decomposition as the main kernel;
// the parts are added in z order, then bias / folded BatchNorm / residual / ReLU / mask / statistics / plane store exactly as in its epilogue
template <int WC, int WPX>
__global__ __launch_bounds__(WC * WPX * 64) void conv_halo_finish_kernel(const HaloArgs a) {
    constexpr int TW = 4 * WPX, TC = WC * 64;
    __shared__ float red[WPX * TC * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WPX, wp = wave % WPX;
    const int lm = lane & 15, g = lane >> 4;
    const int bix = blockIdx.x, biy = blockIdx.y;
    int oy0, ox0, Hd, Wd;
    long rowbase;
    if (a.tiletab) {
        const int4 tt = a.tiletab[bix];
        rowbase = tt.x; Hd = tt.y >> 16; Wd = tt.y & 0xffff; oy0 = tt.z >> 16; ox0 = tt.z & 0xffff;
    } else {
        int bt = bix;
        const int tx = bt % a.tiles_x; bt /= a.tiles_x;
        const int ty = bt % a.tiles_y; const int n = bt / a.tiles_y;
        oy0 = ty * 16; ox0 = tx * TW; Hd = a.H; Wd = a.W; rowbase = (long)n * a.H * a.W;
    }
    const int c0 = biy * TC;
    const float* slot = a.kpart + ((((long)biy * gridDim.x + bix) * a.ksplit) * (WC * WPX) + wave) * 4096;
    float accv[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) accv[e] = slot[e * 64 + lane];
    for (int z = 1; z < a.ksplit; ++z)
#pragma unroll
        for (int e = 0; e < 64; ++e) accv[e] += slot[(long)z * (WC * WPX) * 4096 + e * 64 + lane];
    const int cb = c0 + wc * 64 + g * 16;
    const bool stats = a.stat_part != nullptr;                 // (uniform)
    if (cb >= a.Cout && !stats) return;
    float ss[16], sq[16], bv[16], sv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        ss[e] = sq[e] = 0.f;
        bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
        sv[e] = (a.oscale && cb + e < a.Cout) ? a.oscale[cb + e] : 1.f;
    }
    const EpiArgs ep{a.y, a.res, a.mask, a.ldy, a.ldres, a.ldmask, a.Cout, a.relu, a.yP, a.yps, a.rP, a.rps};
    const int ox = ox0 + (wp >> 2) * 16 + lm;
    const bool bst = stats && a.bs.x != nullptr;               // (uniform) backward statistics -- KgBStat
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + (wp & 3) * 4 + j;
        if (oy >= Hd || ox >= Wd || cb >= a.Cout) continue;
        const long m = rowbase + (long)oy * Wd + ox;
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] = KG_ACC(accv[(i * 4 + j) * 4 + r] * sv[i * 4 + r]) + bv[i * 4 + r];
        if (bst) { kg_conv_epilogue_bstat<16>(ep, a.bs, m, cb, v, ss, sq); continue; }
        if (stats) kg_stat_add(ss, sq, v);
        kg_conv_epilogue<16>(ep, m, cb, v);
    }
    if (stats) kg_stat_commit<WPX, TC>(ss, sq, red, wp, wc * 64 + g * 16, lm, a.stat_part + (long)bix * a.Cout * 2, c0, a.Cout);
}

// ---- 7x7, 64 couts x (16 x 32) pixels on FOUR waves with a BLOCKED accumulation (round 5) ----------------------------------------------
// Same workgroup tile, same LDS image (107 KB halo of a 64-channel chunk + the 6-slot weight ring) and the same load protocol as
// conv_halo_kernel<7, 1, 8, 0> (LDS-direct loads, two taps per raw barrier behind a counted vmcnt, fragment reads from inline asm one k-step
// ahead), but ONE wave per SIMD: a wave owns 64 couts x 128 pixels (four rows of BOTH 16-pixel halves of the tile: 4 weight + 8 pixel
// fragments feed 32 MFMAs per k-step -- 12 ds_read_b128 per 32 MFMAs instead of 8 per 16) and may use 512 registers.  What the registers
// buy is a SECOND accumulator: a 7x7 conv over C channels adds 49 * C / 32 MFMA results into one fp32 register, and the rounding error
// of such a chain grows with the square root of its length (tools/micro/mfma_accum.hip: 3.6e-7 relative after 98 MFMAs, 1.0e-6 after
// 784 = C 512; torch-CPU's fp32 convs, the reference's arithmetic, stay at 3.7e-7 for every C: profiles/r05_head_error_probe.txt).  Here
// the chain is cut at every 64-channel chunk (98 MFMAs): `acc` restarts from zero and the finished block is added to `tot` with VALU adds
// -- 3.7e-7 for every C.  The adds (128 accumulator reads + adds per wave) sit where the MFMA pipe idles anyway: behind the issue of the
// NEXT chunk's halo and first weight slices, in front of the wait for them.  (Cutting every 14 taps as well measured 2.2e-7 and 10 % more
// time: the adds of a block that ends inside the tap loop stall the MFMA pipe of a wave that has the SIMD to itself.)
// Every tap offset is an instruction immediate (the 49 taps are a compile-time sequence, FLIP is a template parameter): no VALU in the loop
// besides the pointer bumps of the ring loads.  Dense launches with bf16 / half row outputs only (no fp32 export, no statistics, no chunk
// split, no ragged tiles): the first-layer head convs (KGnet.py:161-209 `.0`) and their input gradients.
// NB = 2 (round 5): TWO cout blocks per workgroup -- wave tile 128 couts x 128 pixels, 8 weight + 8 pixel fragments feed 64 MFMAs per k-step (16
// ds_read_b128 per 64 MFMAs: the 128 x 128 skeleton of tools/micro/mfma_tile.hip, +11 % over the 64 x 64 tile under the power cap), the halo of a
// chunk staged once per 128 couts.  256 accumulator registers leave no room for `tot`: single-product launches and single-chunk inputs only (the
// chain is then no longer than in the 8-wave kernel).  16 KB ring slots: three of them (one tap per barrier, tap t + 2 loaded while tap t runs).
// KS = 3 (conv_halo3_w4_kernel, NB = 2 only): the same body for the dense 3x3 convs with >= 128 couts -- the decoder's up convs (512 -> 256 on 128^2,
// 1024 -> 512 on 64^2) and their input gradients are K = 4.6 k .. 9.2 k GEMMs per product that the 8-wave 3x3 kernel runs at 0.45 of peak: it stages a
// 78 KB halo per 18 k-steps of a 64-cout tile; here the same halo feeds 128 couts.
template <bool FLIP, int NB, int KS>
__device__ __forceinline__ void conv_halo_w4_body(const HaloArgs& a) {
    constexpr int PAD = KS / 2, TW = 32, HWD = TW + KS - 1, HPIX = (16 + KS - 1) * HWD, T = KS * KS, NT = 256, NSL = 6 / NB, WPT = 2 * NB;
    constexpr int HALO_BYTES = HPIX * 128, WBUF_BYTES = NB * 64 * 128, NA = 4 * NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;
    unsigned char* wbuf = smem + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // wave = row group of the tile (rows 4 * wave .. + 3)
    const int lm = lane & 15, g = lane >> 4;
    int bix = blockIdx.x, biy = blockIdx.y;
    if (a.xcd_map) {
        const int L = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = L & 7, slot = L >> 3;
        biy = slot % gridDim.y;
        bix = xcd * (gridDim.x >> 3) + slot / gridDim.y;
    }
    int bt = bix;
    const int tx = bt % a.tiles_x; bt /= a.tiles_x;
    const int ty = bt % a.tiles_y; const int n = bt / a.tiles_y;
    const int oy0 = ty * 16, ox0 = tx * TW, Hd = a.H, Wd = a.W;
    const long rowbase = (long)n * a.H * a.W;
    const int c0 = biy * (64 * NB);

    f32x4 acc[NA][8], tot[NB == 1 ? 4 : 1][NB == 1 ? 8 : 1];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (NB == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const unsigned lds0 = lds_addr(smem);
    // weight fragments: row r = (lm >> 2) * 16 + i * 4 + (lm & 3) of the slot (the lane ends with 16 consecutive couts); the swizzle key of the
    // row does not depend on i, fragment i adds the immediate i * 512, the ring slot the immediate slot * 8192
    unsigned aaddr[2];
    {
        const int r = (lm >> 2) * 16 + (lm & 3);
        const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) aaddr[s] = lds0 + HALO_BYTES + r * 128 + (((4 * s + g) ^ key) * 16);
    }
    // pixel fragments: the lane's pixel (row 4 * wave, column lm) + swizzled 16-byte chunk of k-step s for a tap whose halo-x shift is fx;
    // row j adds the immediate j * HWD * 128, the second half of the tile 16 * 128 (same key: 16 & 6 == 0), the tap its compile-time offset
    unsigned baddr[KS][2];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
        const int fx = FLIP ? KS - 1 - kx : kx;
        const int key = (lm + fx) & 6;
#pragma unroll
        for (int s = 0; s < 2; ++s) baddr[kx][s] = lds0 + ((wave * 4) * HWD + lm) * 128 + (((4 * s + g) ^ key) * 16);
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nchunks = a.cin_pad / 64;
    for (int ci = 0; ci < nchunks; ++ci) {
        __syncthreads();
        int cc = ci;
        bool stage = true;
        if (a.walk3) {      // low-order products of every chunk first, then hi * hi backwards (halo_set_walk)
            const int nn = a.km.n, grp = 3 * nn;
            const int base = ci / grp * grp, q = ci - base;
            if (q < 2 * nn) cc = base + ((q & 1) ? (q >> 1) : nn + (q >> 1));
            else { cc = base + 2 * nn + (grp - 1 - q); stage = q > 2 * nn || ci == 0; }
        }
        const int xo = a.km.xoff(cc);
        if (stage) {
            constexpr int HPT = (HPIX * 8 + NT - 1) / NT;
#pragma unroll 1
            for (int q = 0; q < HPT; ++q) {
                const int e = tid + q * NT;
                if (e < HPIX * 8) {
                    const int p = e >> 3, cs = e & 7;
                    const int hy = p / HWD, hx = p - hy * HWD;
                    const int c = cs ^ (hx & 6);
                    const int iy = oy0 + hy - PAD, ix = ox0 + hx - PAD;
                    const bf16_t* src = reinterpret_cast<const bf16_t*>(kg_halo_zero_line) + c * 8;
                    if ((unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd)
                        src = a.x + (rowbase + (long)iy * Wd + ix) * a.ldx + xo + c * 8;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(halo + (q * NT + wave_u * 64) * 16), 16, 0, 0);
                }
            }
        }
        const bf16_t* wsrc[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * NT, r = e >> 3, cs = e & 7;
            wsrc[i] = a.w + (long)(c0 + r) * a.K + cc * 64 + (cs ^ (2 * ((r >> 4) & 3) + ((r >> 1) & 1))) * 8;
        }
        auto wglds = [&](int slot_bytes) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc[i],
                                                 (__attribute__((address_space(3))) void*)(wbuf + slot_bytes + (i * NT + wave_u * 64) * 16), 16, 0, 0);
                wsrc[i] += a.cin_pad;
            }
        };
        if constexpr (NB == 1) {
            wglds(0); wglds(WBUF_BYTES); wglds(2 * WBUF_BYTES); wglds(3 * WBUF_BYTES);
            if (ci > 0) {       // the previous chunk's block joins the total while this chunk's loads are in flight
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { tot[i][j] += acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");            // halo + taps 0..2 have landed; tap 3 may still be in flight
        } else {
            wglds(0); wglds(WBUF_BYTES);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // halo + taps 0, 1 have landed (the fragments of tap 1's first k-step are read during tap 0)
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        bf16x8 a0[NA], b0[8], a1[NA], b1[8];
        auto ldA = [&](bf16x8 (&af)[NA], auto tc, auto sc) __attribute__((always_inline)) {
            constexpr int SL = (decltype(tc)::value % NSL) * WBUF_BYTES, S = decltype(sc)::value;
            lds_rd128<SL>(af[0], aaddr[S]); lds_rd128<SL + 512>(af[1], aaddr[S]); lds_rd128<SL + 1024>(af[2], aaddr[S]); lds_rd128<SL + 1536>(af[3], aaddr[S]);
            if constexpr (NB == 2) {
                lds_rd128<SL + 8192>(af[4], aaddr[S]); lds_rd128<SL + 8192 + 512>(af[5], aaddr[S]); lds_rd128<SL + 8192 + 1024>(af[6], aaddr[S]);
                lds_rd128<SL + 8192 + 1536>(af[7], aaddr[S]);
            }
        };
        auto ldB = [&](bf16x8 (&bf)[8], auto tc, auto sc) __attribute__((always_inline)) {
            constexpr int TT = decltype(tc)::value, S = decltype(sc)::value, KY = TT / KS, KX = TT % KS;
            constexpr int OFF = FLIP ? ((KS - 1 - KY) * HWD + (KS - 1 - KX)) * 128 : (KY * HWD + KX) * 128;
            const unsigned ad = baddr[KX][S];
            lds_rd128<OFF>(bf[0], ad); lds_rd128<OFF + HWD * 128>(bf[1], ad); lds_rd128<OFF + 2 * HWD * 128>(bf[2], ad); lds_rd128<OFF + 3 * HWD * 128>(bf[3], ad);
            lds_rd128<OFF + 2048>(bf[4], ad); lds_rd128<OFF + 2048 + HWD * 128>(bf[5], ad); lds_rd128<OFF + 2048 + 2 * HWD * 128>(bf[6], ad);
            lds_rd128<OFF + 2048 + 3 * HWD * 128>(bf[7], ad);
        };
        auto lwait0 = [&](bf16x8 (&af)[NA], bf16x8 (&bf)[8]) __attribute__((always_inline)) {       // every fragment read issued so far has landed; tied to the registers it guards
            if constexpr (NB == 1)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]), "+v"(bf[4]), "+v"(bf[5]),
                               "+v"(bf[6]), "+v"(bf[7]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]), "+v"(bf[0]), "+v"(bf[1]),
                               "+v"(bf[2]), "+v"(bf[3]), "+v"(bf[4]), "+v"(bf[5]), "+v"(bf[6]), "+v"(bf[7]));
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        ldA(a0, I0{}, I0{}); ldB(b0, I0{}, I0{});
        // One wave per SIMD: nobody else issues while this wave sits in its fragment reads, so the reads of the NEXT k-step are dealt out
        // between the MFMAs of the current one (NB = 1: one ds_read_b128 behind every second of 32 MFMAs; NB = 2: behind every third of 64; the
        // last MFMAs cover the latency of the last read) instead of standing in front of them -- with the reads in one block the MFMA pipe ran
        // dry for ~60 cycles per k-step.
        auto rd1 = [&](bf16x8 (&af)[NA], bf16x8 (&bf)[8], auto tc, auto sc, auto kc) __attribute__((always_inline)) {
            constexpr int TT = decltype(tc)::value, S = decltype(sc)::value, K = decltype(kc)::value;
            if constexpr (K < NA) {
                constexpr int SL = (TT % NSL) * WBUF_BYTES;
                lds_rd128<SL + (K >> 2) * 8192 + (K & 3) * 512>(af[K], aaddr[S]);
            } else {
                constexpr int KY = TT / KS, KX = TT % KS, J = K - NA;
                constexpr int OFF = (FLIP ? ((KS - 1 - KY) * HWD + (KS - 1 - KX)) * 128 : (KY * HWD + KX) * 128) + (J >> 2) * 2048 + (J & 3) * HWD * 128;
                lds_rd128<OFF>(bf[J], baddr[KX][S]);
            }
        };
        auto kstep = [&](bf16x8 (&af)[NA], bf16x8 (&bf)[8], bf16x8 (&an)[NA], bf16x8 (&bn)[8], auto tcn, auto scn, auto have_next) __attribute__((always_inline)) {
            lwait0(af, bf);
            constexpr int NRD = NA + 8, PER = NB == 1 ? 2 : 3, NM = NA * 8;      // reads of the next k-step, MFMAs between two of them, MFMAs of a k-step
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) __attribute__((always_inline)) {
                (([&]() __attribute__((always_inline)) {
#pragma unroll
                      for (int q = PER * Ks; q < PER * Ks + PER; ++q) acc[q >> 3][q & 7] = KG_MFMA16(af[q >> 3], bf[q & 7], acc[q >> 3][q & 7]);
                  }(),
                  __builtin_amdgcn_sched_barrier(0),
                  [&]() __attribute__((always_inline)) { if constexpr (decltype(have_next)::value) rd1(an, bn, tcn, scn, std::integral_constant<int, Ks>{}); }(),
                  __builtin_amdgcn_sched_barrier(0)), ...);
            }(std::make_integer_sequence<int, NRD>{});
#pragma unroll
            for (int q = PER * NRD; q < NM; ++q) acc[q >> 3][q & 7] = KG_MFMA16(af[q >> 3], bf[q & 7], acc[q >> 3][q & 7]);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto tap = [&](auto tc) __attribute__((always_inline)) {
            constexpr int TT = decltype(tc)::value;
            if constexpr (NB == 1) {
                if constexpr (!(TT & 1)) {     // pair start: taps t + 4, t + 5 into the slots of taps t - 2, t - 1, which every wave has left
                    if constexpr (TT + 4 < T) wglds(((TT + 4) % NSL) * WBUF_BYTES);
                    if constexpr (TT + 5 < T) wglds(((TT + 5) % NSL) * WBUF_BYTES);
                }
            } else {                           // tap start: tap t + 2 into the slot of tap t - 1, which every wave left at the last barrier
                if constexpr (TT + 2 < T) wglds(((TT + 2) % NSL) * WBUF_BYTES);
            }
            kstep(a0, b0, a1, b1, tc, I1{}, std::true_type{});
            kstep(a1, b1, a0, b0, std::integral_constant<int, (TT + 1 < T ? TT + 1 : TT)>{}, I0{}, std::bool_constant<(TT + 1 < T)>{});
            if constexpr (NB == 1) {
                if constexpr ((TT & 1) && TT + 1 < T) {   // pair end: taps t + 1 .. t + 3 visible after the barrier; tap t + 4 may stay in flight
                    if constexpr (TT + 4 < T) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            } else if constexpr (TT + 1 < T) {            // tap end: taps t + 1 AND t + 2 visible after the barrier -- tap t + 2's first fragments are read during tap
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // t + 1, before the next barrier; its loads had one tap (128 MFMAs per wave) to land
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        };
        [&]<int... Ns>(std::integer_sequence<int, Ns...>) __attribute__((always_inline)) { (tap(std::integral_constant<int, Ns>{}), ...); }(std::make_integer_sequence<int, T>{});
    }

    if constexpr (NB == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] += tot[i][j];
    }
    // ---- epilogue: lane owns pixels (oy0 + 4 * wave + j, ox0 + 16 * h + lm) and couts cb .. cb + 15 of each of its NB cout blocks -------
    const EpiArgs ep{a.y, a.res, a.mask, a.ldy, a.ldres, a.ldmask, a.Cout, a.relu, a.yP, a.yps, a.rP, a.rps};
    auto epi = [&](auto nbc) __attribute__((always_inline)) {
        constexpr int nb = decltype(nbc)::value;
        const int cb = c0 + nb * 64 + g * 16;
        if (cb >= a.Cout) return;
        float bv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ox = ox0 + h * 16 + lm;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oy = oy0 + wave * 4 + j;
                if (oy >= Hd || ox >= Wd) continue;
                const long m = rowbase + (long)oy * Wd + ox;
                float v[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[i * 4 + r] = KG_ACC(acc[nb * 4 + i][h * 4 + j][r]) + bv[i * 4 + r];
                kg_conv_epilogue<16>(ep, m, cb, v);
            }
        }
    };
    epi(std::integral_constant<int, 0>{});
    if constexpr (NB == 2) epi(std::integral_constant<int, 1>{});
}

template <bool FLIP, int NB>
__global__ __launch_bounds__(256) void conv_halo7_w4_kernel(const HaloArgs a) { conv_halo_w4_body<FLIP, NB, 7>(a); }
template <bool FLIP>
__global__ __launch_bounds__(256) void conv_halo3_w4_kernel(const HaloArgs a) { conv_halo_w4_body<FLIP, 2, 3>(a); }

template <int KS, int WC, int WPX, int GM = 0>
static int launch_halo(HaloArgs a, hipStream_t st) {
    constexpr int TW = 4 * WPX, HWD = TW + KS - 1, TC = WC * 64;
    constexpr int smem = (16 + KS - 1) * HWD * 128 + ((KS == 7 && WC <= 2) ? 6 : 3) * TC * 128;
    a.tiles_x = kg_cdiv(a.W, TW);
    static KgPerDevice attr_done;
    if (attr_done.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<KS, WC, WPX, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    dim3 grid(a.tiletab ? a.ntiles : a.N * a.tiles_x * a.tiles_y, GM == 1 ? (a.head_split ? 3 : 1) : kg_cdiv(a.Cout, TC), (GM == 1 && a.prod_split) ? 3 * a.prod_split : 1);
    if (a.stat_part) a.stat_part = (GM == 0 && !a.tiletab) ? kg_conv_stats_claim(grid.x, a.Cout) : nullptr;   // (armed by the caller: BatchNorm statistics)
    if (a.stat_part) a.bs = kg_conv_stats().bs;
    // chunk split of under-filled 3x3 launches (KG_HALO_SPLIT: 0 = never; default: at most 128 workgroups, >= 2 chunks per part, <= 8 parts)
    static const int split_wgs = getenv("KG_HALO_SPLIT") ? atoi(getenv("KG_HALO_SPLIT")) : 128;
    if (GM == 0 && KS == 3 && split_wgs > 0 && a.y && !a.y_f32 && (int)(grid.x * grid.y) <= split_wgs) {
        const int nch = a.cin_pad / 64, wgs = (int)(grid.x * grid.y);
        int Z = kg_cdiv(256, wgs);
        if (Z > nch / 2) Z = nch / 2;
        if (Z > 8) Z = 8;
        while (Z > 1 && (Z - 1) * kg_cdiv(nch, Z) >= nch) --Z;
        if (Z > 1) {
            float* part = kg_splitk_scratch((long)wgs * Z * (WC * WPX));
            if (part) { a.ksplit = Z; a.kpart = part; grid.z = Z; }
        }
    }
    constexpr int use_xcd = 1;
    // (not for the widest heads: 24 cout blocks of one tile stream 24 different 3 MB weight slices through the XCD's 4 MB L2: -2 %)
    a.xcd_map = use_xcd && !a.tiletab && grid.x % 8 == 0 && (grid.y > 1 || use_xcd > 1) && (long)a.Cout * a.K * 2 <= (24L << 20);
    if constexpr (KS == 3 && WC == 1 && WPX == 8 && GM == 0) {
        // KG_HALO3_NB2 (default 1): dense 3x3 launches with >= 128 couts and at least 192 workgroups of 128 couts (the decoder's up convs and their
        // input gradients) run their 128-cout blocks on conv_halo3_w4_kernel (4 waves, wave tile 128 couts x 128 px: conv_halo_w4_body<*, 2, 3>)
        static const int nb2_3 = getenv("KG_HALO3_NB2") ? atoi(getenv("KG_HALO3_NB2")) : 1;
        constexpr int smem_w4 = (16 + KS - 1) * HWD * 128 + 3 * 128 * 128;
        const bool ok3 = !a.tiletab && a.y && !a.y_f32 && !a.stat_part && !a.oscale && a.ksplit <= 1;
        if (ok3 && nb2_3 && a.nb2 == 0 && a.Cout >= 128 && (nb2_3 >= 2 || (long)grid.x * (a.Cout / 128) >= 192)) {      // (2: every such launch -- the tests)
            const int c128 = a.Cout / 128 * 128;
            HaloArgs a1 = a;
            a1.Cout = c128; a1.nb2 = 1;
            const int rc = launch_halo<KS, WC, WPX, GM>(a1, st);
            if (rc != KG_OK || c128 == a.Cout) return rc;
            HaloArgs a2 = a;
            a2.nb2 = -1; a2.Cout = a.Cout - c128;
            a2.w = a.w + (long)c128 * a.K; a2.y = a.y + c128;
            if (a.bias) a2.bias = a.bias + c128;
            if (a.res) a2.res = a.res + c128;
            if (a.mask) a2.mask = a.mask + c128;
            const int rc2 = launch_halo<KS, WC, WPX, GM>(a2, st);
            kg_note_kernel(a.flip ? "conv_halo3_w4_kernel<true> + conv_halo_kernel<3, 1, 8, 0>" : "conv_halo3_w4_kernel<false> + conv_halo_kernel<3, 1, 8, 0>");
            return rc2;
        }
        if (ok3 && a.nb2 == 1) {
            static KgPerDevice w43_attr;
            if (w43_attr.first()) {
                KG_HIP(hipFuncSetAttribute((const void*)conv_halo3_w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_w4));
                KG_HIP(hipFuncSetAttribute((const void*)conv_halo3_w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_w4));
            }
            dim3 g2(grid.x, a.Cout / 128);
            a.xcd_map = use_xcd && g2.x % 8 == 0 && (g2.y > 1 || use_xcd > 1) && (long)a.Cout * a.K * 2 <= (24L << 20);
            if (a.flip) hipLaunchKernelGGL((conv_halo3_w4_kernel<true>), g2, dim3(256), smem_w4, st, a);
            else hipLaunchKernelGGL((conv_halo3_w4_kernel<false>), g2, dim3(256), smem_w4, st, a);
            KG_CHECK_LAUNCH("conv_halo3_w4");
            kg_note_kernel(a.flip ? "conv_halo3_w4_kernel<true>" : "conv_halo3_w4_kernel<false>");
            return KG_OK;
        }
    }
    if constexpr (KS == 7 && WC == 1 && WPX == 8 && GM == 0) {
        // KG_HALO7_W4: 0 = never; 1 (default) = the multi-product launches (hi + lo planes: 3 products; three bf16 planes: 6) with >= 2 channel
        // chunks per plane (C >= 128: where the blocked accumulation matters for the fp32 tolerance); 2 = every dense rows-output launch
        static const int w4 = getenv("KG_HALO7_W4") ? atoi(getenv("KG_HALO7_W4")) : 1;
        constexpr int w4dir = 3;      // bisecting: bit 0 = forward launches, bit 1 = flipped (input gradient)
        // KG_HALO7_NB2 (default 1): launches that need no blocked accumulation (a single product, or one channel chunk per plane) and have >= 128
        // couts run their whole 128-cout blocks on the 128 x 128 wave tiles of conv_halo7_w4_kernel<*, 2>; a remainder of 64 couts (the fused
        // 64 -> 192 first layers of the c0 / c1 heads) follows as its own launch on the kernel it had before
        static const int nb2 = getenv("KG_HALO7_NB2") ? atoi(getenv("KG_HALO7_NB2")) : 1;
        const bool ok = !a.tiletab && a.y && !a.y_f32 && !a.stat_part && !a.oscale && a.ksplit <= 1 && ((w4dir >> (a.flip ? 1 : 0)) & 1);
        const bool multi = a.cin_pad / 64 > a.km.n;
        static KgPerDevice w4_attr;
        if (w4_attr.first()) {
            KG_HIP(hipFuncSetAttribute((const void*)conv_halo7_w4_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            KG_HIP(hipFuncSetAttribute((const void*)conv_halo7_w4_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            KG_HIP(hipFuncSetAttribute((const void*)conv_halo7_w4_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            KG_HIP(hipFuncSetAttribute((const void*)conv_halo7_w4_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        }
        if (ok && nb2 && a.nb2 == 0 && a.Cout >= 128 && (!multi || a.km.n == 1) && w4 < 2) {
            const int c128 = a.Cout / 128 * 128;
            HaloArgs a1 = a;
            a1.Cout = c128; a1.nb2 = 1;
            const int rc = launch_halo<KS, WC, WPX, GM>(a1, st);
            if (rc != KG_OK || c128 == a.Cout) return rc;
            HaloArgs a2 = a;                     // the remaining couts: the same conv on the weight rows / output columns from c128 on
            a2.nb2 = -1; a2.Cout = a.Cout - c128;
            a2.w = a.w + (long)c128 * a.K; a2.y = a.y + c128;
            if (a.bias) a2.bias = a.bias + c128;
            if (a.res) a2.res = a.res + c128;
            if (a.mask) a2.mask = a.mask + c128;
            const int rc2 = launch_halo<KS, WC, WPX, GM>(a2, st);
            // (one call, two launches: the measurement name says so)
            kg_note_kernel(a.flip ? "conv_halo7_w4_kernel<true, 2> + conv_halo_kernel<7, 1, 8, 0>" : "conv_halo7_w4_kernel<false, 2> + conv_halo_kernel<7, 1, 8, 0>");
            return rc2;
        }
        if (ok && a.nb2 == 1) {
            dim3 g2(grid.x, a.Cout / 128);
            a.xcd_map = use_xcd && g2.x % 8 == 0 && (g2.y > 1 || use_xcd > 1) && (long)a.Cout * a.K * 2 <= (24L << 20);
            if (a.flip) hipLaunchKernelGGL((conv_halo7_w4_kernel<true, 2>), g2, dim3(256), smem, st, a);
            else hipLaunchKernelGGL((conv_halo7_w4_kernel<false, 2>), g2, dim3(256), smem, st, a);
            KG_CHECK_LAUNCH("conv_halo7_w4");
            kg_note_kernel(a.flip ? "conv_halo7_w4_kernel<true, 2>" : "conv_halo7_w4_kernel<false, 2>");
            return KG_OK;
        }
        if (ok && (w4 >= 2 || (w4 == 1 && multi && a.km.n >= 2))) {
            if (a.flip) hipLaunchKernelGGL((conv_halo7_w4_kernel<true, 1>), grid, dim3(256), smem, st, a);
            else hipLaunchKernelGGL((conv_halo7_w4_kernel<false, 1>), grid, dim3(256), smem, st, a);
            KG_CHECK_LAUNCH("conv_halo7_w4");
            kg_note_kernel(a.flip ? "conv_halo7_w4_kernel<true, 1>" : "conv_halo7_w4_kernel<false, 1>");
            return KG_OK;
        }
    }
    bool bsk = false;
    if constexpr (KS == 3 && WC == 1 && WPX == 8 && GM == 0) {
        if (a.stat_part && a.bs.x) {
            static KgPerDevice bs_attr;
            if (bs_attr.first()) KG_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<3, 1, 8, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            hipLaunchKernelGGL((conv_halo_kernel<3, 1, 8, 0, true>), grid, dim3(512), smem, st, a);
            kg_note_kernel("conv_halo_kernel<3, 1, 8, 0, true>");
            bsk = true;
        }
    }
    if (!bsk) {
        hipLaunchKernelGGL((conv_halo_kernel<KS, WC, WPX, GM>), grid, dim3(WC * WPX * 64), smem, st, a);
        KG_KNAME(kname, "conv_halo_kernel<%d, %d, %d, %d>", KS, WC, WPX, GM);
        kg_note_kernel(kname);
    }
    KG_CHECK_LAUNCH("conv_halo");
    if (a.ksplit > 1) {
        hipLaunchKernelGGL((conv_halo_finish_kernel<WC, WPX>), dim3(grid.x, grid.y), dim3(WC * WPX * 64), 0, st, a);
        KG_CHECK_LAUNCH("conv_halo_finish");
    }
    return KG_OK;
}

// Stride-1 "same" convolution, KS in {3,7}; cin_pad % 64 == 0; weight rows padded to a multiple of 64*wc.
// wc: couts per workgroup / 64; 0 = default (1: 16x32-pixel tile, 8 waves); 2, 3: 16x16-pixel tile with 128 / 192 couts.
// Ragged mode (tiletab != null): the "images" are boxes of a ragged pixel list; the caller supplies one tile entry per
// workgroup (tile = 16 rows x 32 columns for wc == 1, 16 x 16 otherwise) and total_rows for the fp32 export.
extern "C" int kg_conv2d_halo(const void* x, const void* w, const float* bias, void* y, float* y_f32, const void* res,
                              const void* mask, int N, int H, int W, int cin_pad, int ldx, int Cout, int ldy, int ldres,
                              int ldmask, int K, int KS, int flip, int relu, int f32_C, int wc, const int* tiletab,
                              int ntiles, int total_rows, const kg_planes_t* planes, void* stream) {
    // planes: a = x (cin_pad = channels of ONE plane; the packed weights hold vplanes * cin_pad virtual channels per tap), b = res, y = y
    HaloArgs a;
    memset(&a, 0, sizeof(a));
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_conv2d_halo: bad kg_planes_t");
    int segs_[3];
    const int vplanes = kg_kmap_segs(pp.a_planes, pp.w_planes, segs_);
    a.km = kg_make_kmap(cin_pad, 64, pp.a_planes, pp.a_pstride, pp.w_planes);
    halo_set_walk(a, pp.a_planes, pp.w_planes);
    a.yP = pp.y_planes; a.yps = pp.y_pstride; a.rP = pp.b_planes; a.rps = pp.b_pstride; a.oscale = pp.oscale;
    KG_CHECK_ARG(!(y_f32 && (res || mask)), "kg_conv2d_halo: fp32 exports take no residual / mask");
    KG_CHECK_ARG(x && w && (y || y_f32), "kg_conv2d_halo: null pointer");
    KG_CHECK_ARG(KS == 3 || KS == 7, "kg_conv2d_halo: kernel size must be 3 or 7");
    KG_CHECK_ARG(cin_pad % 64 == 0 && ldx % 8 == 0, "kg_conv2d_halo: cin_pad must be a multiple of 64 (got %d)", cin_pad);
    const int narrow = (wc >> 9) & 3;      // bits 9 / 10 of wc: the narrow input-gradient variants (GM = 3 / 4: 8 / 16 live channels, 7 / 14 virtual taps of 64 columns)
    KG_CHECK_ARG(narrow == 0 || (narrow <= 2 && KS == 7 && cin_pad == 64 && vplanes == 1 && (wc & 255) <= 1 && !tiletab && y && !y_f32),
                 "kg_conv2d_halo: the narrow variants take one 64-channel single-plane dense input and a rows output");
    KG_CHECK_ARG(K >= (narrow ? 7 * narrow * 64 : KS * KS * cin_pad * vplanes), "kg_conv2d_halo: K too small");
    KG_CHECK_ARG((tiletab && ntiles > 0 && Cout > 0) || (N > 0 && H > 0 && W > 0 && Cout > 0), "kg_conv2d_halo: empty problem");
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.y_f32 = y_f32;
    a.res = (const bf16_t*)res; a.mask = (const bf16_t*)mask;
    a.N = N; a.H = H; a.W = W; a.tiles_x = kg_cdiv(W, 16); a.tiles_y = kg_cdiv(H, 16);
    a.cin_pad = cin_pad * vplanes; a.ldx = ldx; a.Cout = Cout; a.ldy = ldy; a.ldres = ldres; a.ldmask = ldmask; a.K = K;
    a.flip = flip; a.relu = relu; a.f32_C = f32_C; a.tiletab = (const int4*)tiletab; a.ntiles = ntiles; a.f32_hw = total_rows;
    {
        const KgConvStats& cs = kg_conv_stats();
        // forward statistics: a plain conv output; backward statistics (kg_conv_bstats_begin): the input gradient (flip), whole 64-channel blocks
        const bool want = cs.bs.x ? (flip && KS == 3 && wc <= 1 && !tiletab && Cout % 64 == 0) : (!res && !mask && !relu && !flip);
        if (y && !y_f32 && want && cs.part && cs.nb == 0)
            a.stat_part = cs.part;      // provisional: launch_halo claims it with the tile count of the variant it launches
    }
    const int k1skip = (wc >> 8) & 1; wc &= 255;   // bit 8: the weights are zero for channels 32..63 of every chunk -> k-step 1 is skipped
    if (wc == 0) wc = 1;   // measured on MI355X: the 16x32-pixel x 64-cout tile (8 waves) beats the 16x16 x 128/192-cout tiles at every KGnet shape
    hipStream_t st = (hipStream_t)stream;
    if (KS == 7) {
        switch (wc) {
            case 1:
                if (narrow == 1) return launch_halo<7, 1, 8, 3>(a, st);
                if (narrow == 2) return launch_halo<7, 1, 8, 4>(a, st);
                return k1skip ? launch_halo<7, 1, 8, 2>(a, st) : launch_halo<7, 1, 8>(a, st);
            case 2: return launch_halo<7, 2, 4>(a, st);
            case 3: return launch_halo<7, 3, 4>(a, st);
        }
    } else {
        switch (wc) {
            case 1: return launch_halo<3, 1, 8>(a, st);
            case 2: return launch_halo<3, 2, 4>(a, st);
            case 3: return launch_halo<3, 3, 4>(a, st);
        }
    }
    kg_set_error("kg_conv2d_halo: bad wc %d", wc);
    return KG_ERR_ARG;
}

// second half of a split heads2 launch: out = sum of the Z = 3 * prod_split partial maps in z order (the low-order products first, hi * hi last,
// the bias inside the last one), added in float64 and rounded once; sigmoid on the kp maps (KGnet.py:300) unless raw logits are asked for
__global__ __launch_bounds__(256) void heads2_finish_kernel(const float* __restrict__ part, long per_z, int Z, int N, long hw, float* __restrict__ kp,
                                                            float* __restrict__ sh, float* __restrict__ md, int kp_raw) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_z; i += (long)gridDim.x * 256) {
        double acc = 0.;
        for (int z = 0; z < Z; ++z) acc += (double)part[z * per_z + i];
        const float v = (float)acc;
        const long pix = i % hw, q = i / hw;
        const int ch = (int)(q % 55); const long n = q / 55;
        if (ch < 5) kp[(n * 5 + ch) * hw + pix] = kp_raw ? v : 1.f / (1.f + expf(-v));
        else if (ch < 15) sh[(n * 10 + ch - 5) * hw + pix] = v;
        else md[(n * 40 + ch - 15) * hw + pix] = v;
    }
}

// The three second-layer 7x7 head convolutions of one scale (KGnet.py:161-209 `.2` layers + torch.sigmoid on kp :300) in
// one launch.  x: fused hidden rows [N*H*W][ldx] = kp | short | mid hidden, C channels each (C % 64 == 0);
// w: packed [64 virtual couts][49][3C] (kg_pack_weight_rows with the virtual row of every map channel; blocks of other
// heads zero); bias64: bias per virtual cout; vmap[64] (device): virtual cout -> channel of (kp 0-4 | short 5-14 |
// mid 15-54) or -1.  kp / sh / md: fp32 NCHW outputs [N][5|10|40][H][W].
extern "C" int kg_conv2d_halo_heads2(const void* x, const void* w, const float* bias64, const int* vmap, float* kp, float* sh,
                                     float* md, int N, int H, int W, int C, int ldx, int K, int kp_sigmoid, const kg_planes_t* planes, void* stream) {
    // planes: a = x (the fused hidden rows; plane stride >= 3C), w: virtual channels [head][w-plane segments][C] per tap
    HaloArgs a;
    memset(&a, 0, sizeof(a));
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_conv2d_halo_heads2: bad kg_planes_t");
    int segs_[3];
    const int vplanes = kg_kmap_segs(pp.a_planes, pp.w_planes, segs_);
    a.km = kg_make_kmap(C, 64, pp.a_planes, pp.a_pstride, pp.w_planes);
    halo_set_walk(a, pp.a_planes, pp.w_planes);
    a.yP = 1; a.rP = 1; a.grp_C = C; a.kp_raw = kp_sigmoid ? 0 : 1;
    KG_CHECK_ARG(x && w && vmap && kp && sh && md, "kg_conv2d_halo_heads2: null pointer");
    KG_CHECK_ARG(C % 64 == 0 && C > 0 && ldx % 8 == 0 && ldx >= 3 * C, "kg_conv2d_halo_heads2: C must be a multiple of 64 (got %d)", C);
    KG_CHECK_ARG(K >= 49 * 3 * C * vplanes && N > 0 && H > 0 && W > 0, "kg_conv2d_halo_heads2: bad sizes");
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias64; a.y_f32 = kp; a.f32_b = sh; a.f32_c = md; a.vmap = vmap;
    a.N = N; a.H = H; a.W = W; a.tiles_y = kg_cdiv(H, 16); a.cin_pad = 3 * C * vplanes; a.ldx = ldx; a.Cout = 64; a.K = K;
    a.grp_chunks = vplanes * C / 64;
    const long tiles = (long)N * kg_cdiv(H, 16) * kg_cdiv(W, 32);
    a.head_split = tiles < 256;   // too few pixel tiles to fill 256 CUs
    // Split launches: one workgroup per (tile, head, plane product, chunk part) with fp32 partial maps (55 channels: a few MB per part) in a
    // library scratch (one buffer per device, grown on demand), summed by a second launch in float64.
    //  * small maps (single-image inference: a 64 x 64 map is 8 tiles x 3 heads = 24 workgroups walking 24 chunks each): one part per product;
    //  * wide heads (C >= 256: 4 / 8 chunks per product, 392 / 784 MFMA additions into one accumulator in the hi * hi product): parts of
    //    KG_HEADS2_KPART chunks (default 2) -- blocked accumulation, see HaloArgs.prod_split.
    constexpr int split_mode = 1;   // 0 never, 1 small maps + wide heads, 2 whenever it fits
    static const int kpart = getenv("KG_HEADS2_KPART") ? atoi(getenv("KG_HEADS2_KPART")) : 2;        // chunks per part of a wide head (0: no blocked accumulation)
    const int nchunk = C / 64;
    const int parts = (kpart > 0 && nchunk >= 4) ? kg_cdiv(nchunk, kpart) : 1;
    const bool want = split_mode && vplanes == 3 && (parts > 1 || (a.head_split && (split_mode == 2 || tiles * 3 < 128)));
    if (want) {
        // The partial maps live in a per-device scratch of at most 96 M floats.  A batch whose partials would not fit is processed in image
        // groups (same kernels, same order of additions per element: the result does not depend on the grouping); only a SINGLE image
        // beyond the cap gives up parts (then the split) -- recorded in the launch's kernel note ("…/unsplit"), never silently.
        constexpr long CAP = 96L << 20;
        static float* part[32] = {nullptr};
        static long part_floats[32] = {0};
        int dev = 0;
        KG_HIP(hipGetDevice(&dev));
        const long per_img = 55L * H * W;
        int use_parts = parts;
        while (use_parts > 1 && 3L * use_parts * per_img > CAP) use_parts = (use_parts + 1) / 2;
        if (dev >= 0 && dev < 32 && 3L * use_parts * per_img <= CAP && (use_parts > 1 || parts == 1 || a.head_split)) {
            long group = CAP / (3L * use_parts * per_img);
            if (group > N) group = N;
            const long need = 3L * use_parts * per_img * group;
            if (part_floats[dev] < need) {
                if (part[dev]) KG_HIP(hipFree(part[dev]));
                part[dev] = nullptr; part_floats[dev] = 0;
                const long sz = need > (8L << 20) ? need : (8L << 20);
                KG_HIP(hipMalloc((void**)&part[dev], sz * sizeof(float)));
                part_floats[dev] = sz;
            }
            a.head_split = 1;
            a.prod_split = use_parts; a.part = part[dev];
            for (long n0 = 0; n0 < N; n0 += group) {
                const int ng = (int)(N - n0 < group ? N - n0 : group);
                HaloArgs g = a;
                g.N = ng;
                g.x = a.x + n0 * H * W * (long)ldx;
                float* kp_g = kp + n0 * 5 * (long)H * W;
                float* sh_g = sh + n0 * 10 * (long)H * W;
                float* md_g = md + n0 * 40 * (long)H * W;
                g.y_f32 = kp_g; g.f32_b = sh_g; g.f32_c = md_g;
                const int rc = launch_halo<7, 1, 8, 1>(g, (hipStream_t)stream);
                if (rc != KG_OK) return rc;
                const long pz = per_img * ng;
                int blocks = (int)((pz + 255) / 256); if (blocks > 4096) blocks = 4096;
                hipLaunchKernelGGL(heads2_finish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)part[dev], pz, 3 * use_parts, ng, (long)H * W,
                                   kp_g, sh_g, md_g, a.kp_raw);
                KG_CHECK_LAUNCH("heads2_finish");
            }
            return KG_OK;
        }
        const int rc = launch_halo<7, 1, 8, 1>(a, (hipStream_t)stream);
        if (parts > 1) kg_note_kernel("conv_halo_kernel<7, 1, 8, 1>/unsplit: blocked accumulation dropped (one image's partial maps exceed the scratch cap)");
        return rc;
    }
    return launch_halo<7, 1, 8, 1>(a, (hipStream_t)stream);
}

// conv3_ws.hip -- weight-stationary 3x3 stride-1 "same" convolution, 64 -> 64 channels, for hi + lo planed operands (3 MFMA products).
//
// KGnet's full-resolution 3x3 convs with 64 input and 64 output channels -- c0_conv.2 (KGnet.py:139-142), c1_up_conv (:153), seg_head.0
// (:145-147) and skip_combine.0.up (:116-119) on the ragged c0 crops -- move 512 bytes per pixel (two 16-bit planes in, two out) for
// 3 x 73 728 MACs: the MFMA and the HBM roof are both at ~0.13 ns per pixel.  On the LDS-halo kernel (conv_halo.hip) a 16 x 32-pixel
// workgroup stages two 78 KB halos through registers, walks 27 (product, tap) steps behind one barrier each and stores 131 KB: 6.5 of its
// 35 us are MFMAs and a second workgroup does not fit the CU (0.57 ms per dense 512 x 512 batch-8 launch).  The persistent design of
// conv3_c64.hip keeps one plane's weights resident in LDS (72 KB); two planes do not fit.  Here the WEIGHTS live in REGISTERS:
//   * a wave owns 16 output channels: its A fragments for all 9 taps x 2 k-steps x {w_hi, w_lo} are 36 fragments = 144 VGPRs, loaded once
//     per (persistent) workgroup; the four waves of a workgroup cover the 64 couts and all walk the same pixels;
//   * pixels stream through LDS: an 8 x 16-pixel tile's (10 x 18) halo of BOTH planes is 45 KB, so TWO workgroups share a CU and one
//     multiplies while the other stages / stores -- the overlap a 155 KB workgroup cannot have;
//   * per 16-pixel row fragment, tap and k-step: two ds_read_b128 (x_hi, x_lo; XOR swizzle key hx & 6 as in conv_halo.hip: conflict-free
//     for every tap shift) feed three MFMAs: x_hi w_hi -> acc_hi, x_hi w_lo -> acc_lo, x_lo w_hi -> acc_lo.  The low-order products have
//     their own accumulator (added to acc_hi once, in the epilogue): no ordering constraint between the products, nothing staged twice;
//   * 8 x 16 tiles also fit the ragged boxes of the seg branch better than 16 x 32 ones (fill 67 % against 50 % for sides in [14, 40)).
// Epilogue: bias, ReLU, hi + lo plane store (a lane holds 4 consecutive couts of one pixel: 8-byte stores).
// Measured (tools/ws_probe.py, dense 8 x 512 x 512, random data): 0.52 ms against 0.55 ms for conv_halo<3>; inside the train step (ReLU-sparse
// data) 0.42 ms per launch against 0.57 (dense) / 0.73 (ragged).  Switching parts off: the memory phases alone run at the HBM rate (0.27 ms),
// the MFMA phase alone at the MFMA rate (0.25 ms) -- and they ADD UP: the two workgroups of a CU stay in phase (a start-up delay of the odd
// workgroups does not persist), or so it seemed: the obvious remedy -- ONE 8-wave workgroup per CU, double-buffered halo, the LDS-direct loads of tile
// k + 1 issued before the MFMA phase of tile k, fragment reads from inline asm behind counted lgkmcnt (the compiler orders every LDS read it sees
// behind a pending LDS-DMA) -- was written, is bit-correct, and runs at the SAME speed (0.528 against 0.512 ms here, +-0 in the step): the
// phases are not what adds up.  The 8-byte stores (32 bytes per pixel and wave, four waves per 128-byte line at four different times) are the
// remaining suspect: without them the register-staged first version ran in 0.14 ms.  Tried and dropped: the output through an LDS tile as whole 128-byte lines
// (+7 % in the step), four tile rows per pass (spills at 256 registers).
#include "kg_common.h"

__device__ uint4 kg_ws_zero_line[8];   // 128 zero bytes: source of the padding pixels of a halo

struct WsArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
    const int4* tiletab;   // ragged: one {row0, (h << 16) | w, (oy0 << 16) | ox0, 0} entry per 8 x 16 tile of a box
    int ntiles, N, H, W, tiles_x, tiles_y;
    int ldx, xps, ldy, yps, yP, K, relu;
};

__global__ __launch_bounds__(256, 2) void conv3_ws_kernel(const WsArgs a) {
    constexpr int TH = 8, TW = 16, HH = TH + 2, HW_ = TW + 2, PLANE = HH * HW_ * 128;     // 23 040 bytes per plane
    __shared__ __attribute__((aligned(16))) unsigned char halo[2 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, g = lane >> 4;

    // ---- weights -> registers: wf[tap][k-step][plane]; packed row = cout, per tap the virtual planes [x_hi w_lo: w_lo | x_lo w_hi: w_hi | x_hi w_hi: w_hi]
    bf16x8 whi[9][2], wlo[9][2];
    {
        const bf16_t* wr = a.w + (long)(wave * 16 + lm) * a.K + g * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                wlo[t][s] = *reinterpret_cast<const bf16x8*>(wr + t * 192 + 0 * 64 + s * 32);
                whi[t][s] = *reinterpret_cast<const bf16x8*>(wr + t * 192 + 1 * 64 + s * 32);
            }
    }
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = a.bias ? a.bias[wave * 16 + g * 4 + r] : 0.f;

    // B-fragment byte offsets inside a halo row: pixel hx = lm + dx, 16-byte slot (4 s + g) ^ (hx & 6)
    int boff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int s = 0; s < 2; ++s) boff[dx][s] = (lm + dx) * 128 + (((4 * s + g) ^ ((lm + dx) & 6)) * 16);

    const int total = a.tiletab ? a.ntiles : a.N * a.tiles_x * a.tiles_y;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        int oy0, ox0, Hd, Wd;
        long rowbase;
        if (a.tiletab) {
            const int4 tt = a.tiletab[t];
            rowbase = tt.x; Hd = tt.y >> 16; Wd = tt.y & 0xffff; oy0 = tt.z >> 16; ox0 = tt.z & 0xffff;
        } else {
            int bt = t;
            const int tx = bt % a.tiles_x; bt /= a.tiles_x;
            const int ty = bt % a.tiles_y; const int n = bt / a.tiles_y;
            oy0 = ty * TH; ox0 = tx * TW; Hd = a.H; Wd = a.W; rowbase = (long)n * a.H * a.W;
        }
        __syncthreads();                                   // every wave has left the previous tile's halo
        // ---- stage the (10 x 18)-pixel halo of both planes: 2 880 pieces of 16 bytes by LDS-direct loads (global_load_lds_dwordx4: no staging
        // registers next to the 144 weight registers, all 12 rounds in flight at once -- as three register batches of four loads the kernel
        // took 0.49 instead of 0.43 ms per launch).  The destination of a wave's load is lane-linear, so the XOR swizzle of the 16-byte channel
        // slots is applied to the SOURCE chunk (an involution inside the pixel's 128-byte line); pixels outside the image / box read a zero line.
        constexpr int NP = 2 * HH * HW_ * 8, RND = (NP + 255) / 256;
#pragma unroll 1
        for (int q = 0; q < RND; ++q) {
            const int e = tid + q * 256;
            if (e < NP) {
                const int p = e / (HH * HW_ * 8), rem = e - p * (HH * HW_ * 8);
                const int hp = rem >> 3, cs = rem & 7;
                const int hy = hp / HW_, hx = hp - hy * HW_;
                const int c = cs ^ (hx & 6);
                const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                const bf16_t* src = reinterpret_cast<const bf16_t*>(kg_ws_zero_line) + c * 8;
                if ((unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd)
                    src = a.x + (rowbase + (long)iy * Wd + ix) * a.ldx + (long)p * a.xps + c * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(halo + (q * 256 + wave * 64) * 16), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- two tile rows (16-pixel fragments) at a time: 4 independent accumulators
#pragma unroll 1
        for (int rp = 0; rp < TH / 2; ++rp) {
            f32x4 ahi[2], alo[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) { ahi[f] = f32x4{0.f, 0.f, 0.f, 0.f}; alo[f] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            const unsigned char* hrow = halo + (2 * rp) * (HW_ * 128);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int dy = tp / 3, dx = tp % 3;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 xh[2], xl[2];
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const unsigned char* q = hrow + (f + dy) * (HW_ * 128) + boff[dx][s];
                        xh[f] = *reinterpret_cast<const bf16x8*>(q);
                        xl[f] = *reinterpret_cast<const bf16x8*>(q + PLANE);
                    }
#pragma unroll
                    for (int f = 0; f < 2; ++f) ahi[f] = KG_MFMA16(whi[tp][s], xh[f], ahi[f]);
#pragma unroll
                    for (int f = 0; f < 2; ++f) alo[f] = KG_MFMA16(wlo[tp][s], xh[f], alo[f]);
#pragma unroll
                    for (int f = 0; f < 2; ++f) alo[f] = KG_MFMA16(whi[tp][s], xl[f], alo[f]);
                }
            }
            // ---- epilogue of the two rows: lane = pixel lm, couts wave * 16 + 4 g .. + 3
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int oy = oy0 + 2 * rp + f, ox = ox0 + lm;
                if (oy >= Hd || ox >= Wd) continue;
                const long m = rowbase + (long)oy * Wd + ox;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = KG_ACC(ahi[f][r] + alo[f][r]) + bv[r];
                    if (a.relu) v[r] = kg_relu(v[r]);
                }
                bf16_t* yp = a.y + m * a.ldy + wave * 16 + g * 4;
                for (int p = 0; p < a.yP; ++p) {
                    bf16_t h[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { h[r] = f2bf(v[r]); v[r] -= bf2f(h[r]); }
                    uint2 o;
                    o.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); o.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                    *reinterpret_cast<uint2*>(yp + (long)p * a.yps) = o;
                }
            }
        }
    }
}

// 3x3 stride-1 "same" conv, 64 -> 64 channels, x in hi + lo planes (a: 2 planes), weights packed for 2 x 2 planes (3 virtual planes per tap,
// kg_pack_weight with x_planes = w_planes = 2: K >= 9 * 192), bias + optional ReLU, output rows in y planes (1 or 2).  Dense: N images of
// H x W; ragged (tiletab8 != NULL): one {row0, (h << 16) | w, (oy0 << 16) | ox0, 0} entry per 8 x 16 tile of a box (kg_host_tile_table(8, 16)).
extern "C" int kg_conv3x3_ws(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int ldx, int ldy, int K, int relu,
                             const int* tiletab8, int ntiles, const kg_planes_t* planes, void* stream) {
    WsArgs a;
    memset(&a, 0, sizeof(a));
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_conv3x3_ws: bad kg_planes_t");
    KG_CHECK_ARG(x && w && y, "kg_conv3x3_ws: null pointer");
    KG_CHECK_ARG(pp.a_planes == 2 && pp.w_planes == 2 && pp.y_planes >= 1 && pp.y_planes <= 2, "kg_conv3x3_ws: needs hi + lo planes of x and w (got %d, %d)", pp.a_planes, pp.w_planes);
    KG_CHECK_ARG(ldx % 8 == 0 && ldx >= 64 && ldy % 4 == 0 && ldy >= 64 && K >= 9 * 192 && pp.y_pstride % 4 == 0, "kg_conv3x3_ws: needs 64 input / output channels");
    KG_CHECK_ARG((reinterpret_cast<uintptr_t>(y) & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "kg_conv3x3_ws: x must be 16-byte, y 8-byte aligned");
    KG_CHECK_ARG((tiletab8 && ntiles > 0) || (N > 0 && H > 0 && W > 0), "kg_conv3x3_ws: empty problem");
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y;
    a.tiletab = (const int4*)tiletab8; a.ntiles = ntiles; a.N = N; a.H = H; a.W = W; a.tiles_x = kg_cdiv(W, 16); a.tiles_y = kg_cdiv(H, 8);
    a.ldx = ldx; a.xps = pp.a_pstride; a.ldy = ldy; a.yps = pp.y_pstride; a.yP = pp.y_planes; a.K = K; a.relu = relu;
    const int total = tiletab8 ? ntiles : N * a.tiles_x * a.tiles_y;
    int grid = 512;                                   // persistent: two workgroups per CU (45 KB of LDS, <= 256 VGPRs each)
    if (grid > total) grid = total;
    hipLaunchKernelGGL(conv3_ws_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    KG_CHECK_LAUNCH("conv3x3_ws");
    kg_note_kernel("conv3_ws_kernel");
    return KG_OK;
}

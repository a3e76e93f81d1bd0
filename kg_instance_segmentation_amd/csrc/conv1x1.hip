// conv1x1.hip -- streaming bf16 MFMA kernel for 1x1 stride-1 convolutions (plain GEMM  Y[M][N] = X[M][K] . W[N][K]^T).
//
// KGnet's 1x1 convs (Bottleneck conv1/conv3/downsample KGnet.py:64-99, c*_cat_refine :155-158, skip_combine cat_conv
// :101-111) and their input gradients have huge M (up to 2M pixels) and small K (64..1024): they are HBM-bound, and an
// LDS-staged implicit GEMM spends its time in barriers.  Here:
//   * the workgroup's weight slab [64 couts][K] is staged in LDS ONCE; workgroups are persistent over 256-row tiles,
//     so there is no barrier in the steady state;
//   * pixel (B) fragments are loaded straight from global memory into registers (each row is consumed once: no reuse
//     to buy with LDS), one k-step ahead of the MFMAs, also across tile boundaries;
//   * lane layout and epilogue (bias / residual / ReLU / ReLU-mask, bf16 rows) are those of conv_igemm.hip.
// Rows are just rows: the same kernel serves dense images and the ragged seg-branch pixel lists.
#include "kg_common.h"
#include <stdlib.h>

struct C1Args {
    const bf16_t* x; const bf16_t* w; const float* bias;
    bf16_t* y; const bf16_t* res; const bf16_t* mask;
    long M;
    int K, ldx, Cout, ldy, ldres, ldmask, relu, wK;   // wK: row pitch (elements) of the packed weight matrix
    int wrows;                                        // rows of the packed weight matrix that may be read
};

template <int KU>   // k-steps (of 32 channels) fetched per prefetch group: the loads of KU k-steps are in flight together
__global__ __launch_bounds__(256) void conv1x1_kernel(const C1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [K/64][64 rows][128 B], swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * 64;
    const int nk64 = a.K / 64;

    // ---- stage the weight slab once ---------------------------------------------------------------------------
    // (8 loads in flight per thread: the slab is up to 128 KB and only one workgroup fits a CU at that size)
    for (int e0 = tid; e0 < nk64 * 64 * 8; e0 += 256 * 8) {
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * 256;
            const int c = e & 7, r = (e >> 3) & 63, kc = e >> 9;
            v[q] = e < nk64 * 64 * 8 ? *reinterpret_cast<const uint4*>(a.w + (long)(c0 + r) * a.wK + kc * 64 + c * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * 256;
            const int c = e & 7, r = (e >> 3) & 63, kc = e >> 9;
            const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
            if (e < nk64 * 64 * 8) *reinterpret_cast<uint4*>(smem + kc * 8192 + r * 128 + ((c ^ key) * 16)) = v[q];
        }
    }
    __syncthreads();

    int a_off[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (lm >> 2) * 16 + i * 4 + (lm & 3);
        const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) a_off[i][s] = r * 128 + (((4 * s + g) ^ key) * 16);
    }
    const int cb = c0 + g * 16;
    const bool full = cb + 16 <= a.Cout;
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;

    const long ntiles = (a.M + 255) / 256;
    const int ngr = a.K / (32 * KU);
    auto rowptr = [&](long tile, int j) -> const bf16_t* {
        long m = tile * 256 + wave * 64 + j * 16 + lm;
        if (m >= a.M) m = a.M - 1;   // clamped rows are computed and discarded
        return a.x + m * a.ldx + g * 8;
    };
    bf16x8 bcur[KU][4], bnxt[KU][4];
    long tile = blockIdx.x;
    if (tile < ntiles) {
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) bcur[u][j] = *reinterpret_cast<const bf16x8*>(rowptr(tile, j) + u * 32);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16_t* rp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rp[j] = rowptr(tile, j);
        const long tnext = tile + gridDim.x;
        for (int gr = 0; gr < ngr; ++gr) {
            // prefetch the next group's pixel fragments (next tile's first group at the end)
            if (gr + 1 < ngr) {
#pragma unroll
                for (int u = 0; u < KU; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bnxt[u][j] = *reinterpret_cast<const bf16x8*>(rp[j] + ((gr + 1) * KU + u) * 32);
            } else if (tnext < ntiles) {
#pragma unroll
                for (int u = 0; u < KU; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bnxt[u][j] = *reinterpret_cast<const bf16x8*>(rowptr(tnext, j) + u * 32);
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int ks = gr * KU + u;
                bf16x8 af[4];
                const unsigned char* wb = smem + (ks >> 1) * 8192;
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wb + a_off[i][u & 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = KG_MFMA16(af[i], bcur[u][j], acc[i][j]);
            }
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) bcur[u][j] = bnxt[u][j];
        }
        // ---- epilogue ---------------------------------------------------------------------------------------
        if (cb < a.Cout) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long m = tile * 256 + wave * 64 + j * 16 + lm;
                if (m >= a.M) continue;
                float v[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[i * 4 + r] = KG_ACC(acc[i][j][r]) + bv[i * 4 + r];
                if (a.res) {
                    const bf16_t* rq = a.res + m * a.ldres + cb;
                    if (full && ((reinterpret_cast<uintptr_t>(rq) & 15) == 0)) {
                        uint4 r0 = *reinterpret_cast<const uint4*>(rq), r1 = *reinterpret_cast<const uint4*>(rq + 8);
                        const bf16_t* rs0 = reinterpret_cast<const bf16_t*>(&r0);
                        const bf16_t* rs1 = reinterpret_cast<const bf16_t*>(&r1);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { v[e] += bf2f(rs0[e]); v[8 + e] += bf2f(rs1[e]); }
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            if (full || cb + e < a.Cout) v[e] += bf2f(rq[e]);
                    }
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = kg_relu(v[e]);
                }
                if (a.mask) {
                    const bf16_t* mp = a.mask + m * a.ldmask + cb;
                    if (full && ((reinterpret_cast<uintptr_t>(mp) & 15) == 0)) {
                        uint4 m0 = *reinterpret_cast<const uint4*>(mp), m1 = *reinterpret_cast<const uint4*>(mp + 8);
                        const bf16_t* ms0 = reinterpret_cast<const bf16_t*>(&m0);
                        const bf16_t* ms1 = reinterpret_cast<const bf16_t*>(&m1);
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
                    uint4 o0 = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
                    uint4 o1 = make_uint4(pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15]));
                    *reinterpret_cast<uint4*>(yp) = o0;
                    *reinterpret_cast<uint4*>(yp + 8) = o1;
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (cb + e < a.Cout) yp[e] = f2bf(v[e]);
                }
            }
        }
    }
}


// ---- HBM-bound shapes (K <= 128): fully coalesced variant -----------------------------------------------------------
// The fragment-shaped accesses of conv1x1_kernel (16 rows x 64 B per load, 16-byte pieces at a 32-byte stride per store)
// cap it at ~1.5-2.5 TB/s.  Here every global access is a full 128-byte line per 8 lanes:
//   * the X tile [128 rows][K] is fetched with row-contiguous 16-byte loads (next tile in flight during the MFMAs) and
//     written to swizzled LDS, from which the MFMA B fragments are read;
//   * the fp32 accumulators (+bias) go through an LDS transpose tile [128][64] (aliasing the X tile), and the epilogue
//     (residual add, ReLU, ReLU-mask, bf16 rounding: one rounding, as in the other kernels) runs on 8-cout pieces with
//     coalesced 16-byte residual / mask loads and stores.
// NB > 1 (K = 64 only): the workgroup computes NB 64-cout blocks from ONE staged X tile (a 64 -> 128 input gradient at 512^2 read
// its 268 MB of X once per cout block: 1.07 GB instead of 0.8 GB); the fp32 output tile then has its own LDS region.
template <int KC, int NB = 1>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const C1Args a) {
    constexpr int TM = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                       // [NB][KC][64 couts][128 B]
    unsigned char* xl = smem + NB * KC * 8192;      // [KC][TM][128 B]  (X tile)
    unsigned char* ol = NB == 1 ? xl : xl + KC * TM * 128;   // [TM][256 B] fp32 output tile (NB == 1: aliases the X tile)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * 64 * NB;

    for (int e = tid; e < NB * KC * 64 * 8; e += 256) {
        const int c = e & 7, r = (e >> 3) & 63, kc = (e >> 9) % KC, nb = e / (KC * 512);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c0 + nb * 64 + r < a.wrows) v = *reinterpret_cast<const uint4*>(a.w + (long)(c0 + nb * 64 + r) * a.wK + kc * 64 + c * 8);
        const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
        *reinterpret_cast<uint4*>(wl + (nb * KC + kc) * 8192 + r * 128 + ((c ^ key) * 16)) = v;
    }
    int a_off[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (lm >> 2) * 16 + i * 4 + (lm & 3);
        const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) a_off[i][s] = r * 128 + (((4 * s + g) ^ key) * 16);
    }
    float bv[NB][16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[nb][e] = (a.bias && c0 + nb * 64 + g * 16 + e < a.Cout) ? a.bias[c0 + nb * 64 + g * 16 + e] : 0.f;

    // X staging: thread -> (row, 16-byte chunk) pairs, 8 (KC=1) or 16 (KC=2) consecutive lanes per row
    constexpr int XPT = TM * KC * 8 / 256;
    int x_row[XPT], x_col[XPT], x_lds[XPT];
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
        const int e = tid + q * 256;
        const int cw = e % (KC * 8), r = e / (KC * 8);
        x_row[q] = r; x_col[q] = cw * 8;
        x_lds[q] = (cw >> 3) * (TM * 128) + r * 128 + (((cw & 7) ^ ((r >> 1) & 7)) * 16);
    }
    const long ntiles = (a.M + TM - 1) / TM;
    uint4 xr[XPT];
    auto xload = [&](long tile) {
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const long m = tile * TM + x_row[q];
            xr[q] = m < a.M ? *reinterpret_cast<const uint4*>(a.x + m * a.ldx + x_col[q]) : make_uint4(0, 0, 0, 0);
        }
    };
    // B fragment offsets: wave rows 32*wave + 16*j + lm
    int b_off[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wave * 32 + j * 16 + lm;
#pragma unroll
        for (int s = 0; s < 2; ++s) b_off[j][s] = r * 128 + (((4 * s + g) ^ ((r >> 1) & 7)) * 16);
    }
    // epilogue pieces: thread -> (row, 8-cout piece)
    const int e_c8 = tid & 7, e_r0 = tid >> 3;   // rows e_r0 + 32*q

    long tile = blockIdx.x;
    if (tile < ntiles) xload(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                          // previous tile's output reads are done (NB == 1: xl is reused)
#pragma unroll
        for (int q = 0; q < XPT; ++q) *reinterpret_cast<uint4*>(xl + x_lds[q]) = xr[q];
        __syncthreads();
        if (tile + gridDim.x < ntiles) xload(tile + gridDim.x);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{KG_BIAS_ACC(bv[nb][i * 4 + 0]), KG_BIAS_ACC(bv[nb][i * 4 + 1]), KG_BIAS_ACC(bv[nb][i * 4 + 2]), KG_BIAS_ACC(bv[nb][i * 4 + 3])};
#pragma unroll
            for (int ks = 0; ks < 2 * KC; ++ks) {
                bf16x8 af[4], bf[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wl + (nb * KC + (ks >> 1)) * 8192 + a_off[i][ks & 1]);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(xl + (ks >> 1) * (TM * 128) + b_off[j][ks & 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = KG_MFMA16(af[i], bf[j], acc[i][j]);
            }
            if (NB == 1 || nb > 0) __syncthreads();   // NB == 1: all B fragments read, xl becomes the output tile; else: the previous block's output reads are done
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wave * 32 + j * 16 + lm;
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(ol + r * 256 + (((g * 4 + i) ^ (r & 15)) * 16)) = acc[i][j];
            }
            __syncthreads();
            const int cpiece = c0 + nb * 64 + e_c8 * 8;
            if (cpiece < a.Cout) {               // Cout % 8 == 0 on this path
#pragma unroll
                for (int q = 0; q < TM / 32; ++q) {
                    const int r = e_r0 + 32 * q;
                    const long m = tile * TM + r;
                    if (m >= a.M) continue;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(ol + r * 256 + (((2 * e_c8) ^ (r & 15)) * 16));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(ol + r * 256 + (((2 * e_c8 + 1) ^ (r & 15)) * 16));
                    float v[8] = {KG_ACC(v0[0]), KG_ACC(v0[1]), KG_ACC(v0[2]), KG_ACC(v0[3]), KG_ACC(v1[0]), KG_ACC(v1[1]), KG_ACC(v1[2]), KG_ACC(v1[3])};
                    if (a.res) {
                        const uint4 rv = *reinterpret_cast<const uint4*>(a.res + m * a.ldres + cpiece);
                        const bf16_t* rs = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bf2f(rs[e]);
                    }
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = kg_relu(v[e]);
                    }
                    if (a.mask) {
                        const uint4 mv = *reinterpret_cast<const uint4*>(a.mask + m * a.ldmask + cpiece);
                        const bf16_t* ms = reinterpret_cast<const bf16_t*>(&mv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = bf2f(ms[e]) > 0.f ? v[e] : 0.f;
                    }
                    *reinterpret_cast<uint4*>(a.y + m * a.ldy + cpiece) =
                        make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
                }
            }
        }
    }
}

// Y[M][Cout] = act(X[M][K] . W^T + bias + res) (* mask > 0).  K % 64 == 0, K <= 1024; packed weight rows padded to 64.
extern "C" int kg_conv1x1(const void* x, const void* w, const float* bias, void* y, const void* res, const void* mask, long M,
                          int K, int wK, int ldx, int Cout, int ldy, int ldres, int ldmask, int relu, void* stream) {
    C1Args a;
    memset(&a, 0, sizeof(a));
    KG_CHECK_ARG(x && w && y, "kg_conv1x1: null pointer");
    KG_CHECK_ARG(K % 64 == 0 && K >= 64 && K <= 1024 && wK >= K, "kg_conv1x1: K=%d must be a multiple of 64 in [64,1024]", K);
    KG_CHECK_ARG(ldx % 8 == 0 && M > 0 && Cout > 0, "kg_conv1x1: bad sizes");
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.res = (const bf16_t*)res;
    a.mask = (const bf16_t*)mask; a.M = M; a.K = K; a.wK = wK; a.ldx = ldx; a.Cout = Cout; a.ldy = ldy; a.ldres = ldres;
    a.ldmask = ldmask; a.relu = relu;
    a.wrows = kg_cdiv(Cout, 64) * 64;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (K <= 128 && Cout % 8 == 0 && ldy % 8 == 0 && al16(y) && al16(x) && (!res || (ldres % 8 == 0 && al16(res))) &&
        (!mask || (ldmask % 8 == 0 && al16(mask)))) {
        const int kc = K / 64;
        constexpr int use_nb = 1;
        const int nb = (use_nb && kc == 1 && Cout > 64) ? (Cout > 128 ? 4 : 2) : 1;   // cout blocks per workgroup (K = 64: X is the big operand)
        // weights + X tile (16 / 32 KB) aliased with the 32 KB fp32 output tile; nb > 1: separate X and output tiles
        const int smem_s = nb == 1 ? kc * 8192 + 32768 : nb * 8192 + 16384 + 32768;
        static KgPerDevice attr_done_s;
        if (attr_done_s.first()) {
            KG_HIP(hipFuncSetAttribute((const void*)conv1x1_stream_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 8192 + 49152));
            KG_HIP(hipFuncSetAttribute((const void*)conv1x1_stream_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8192 + 49152));
        }
        const long nt = (M + 127) / 128;
        const int nyb = kg_cdiv(Cout, 64 * nb);
        long gxs = (long)256 * (nb == 1 ? (kc == 1 ? 4 : 3) : 2) / nyb;
        if (gxs < 64) gxs = 64;
        if (gxs > nt) gxs = nt;
        if (nb == 4) hipLaunchKernelGGL((conv1x1_stream_kernel<1, 4>), dim3((unsigned)gxs, nyb), dim3(256), smem_s, (hipStream_t)stream, a);
        else if (nb == 2) hipLaunchKernelGGL((conv1x1_stream_kernel<1, 2>), dim3((unsigned)gxs, nyb), dim3(256), smem_s, (hipStream_t)stream, a);
        else if (kc == 1) hipLaunchKernelGGL(conv1x1_stream_kernel<1>, dim3((unsigned)gxs, nyb), dim3(256), smem_s, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(conv1x1_stream_kernel<2>, dim3((unsigned)gxs, nyb), dim3(256), smem_s, (hipStream_t)stream, a);
        KG_CHECK_LAUNCH("conv1x1_stream");
        kg_note_kernel(nb == 4 ? "conv1x1_stream_kernel<1, 4>" : nb == 2 ? "conv1x1_stream_kernel<1, 2>" : kc == 1 ? "conv1x1_stream_kernel<1, 1>" : "conv1x1_stream_kernel<2, 1>");
        return KG_OK;
    }
    const int smem = (K / 64) * 8192;
    static int attr_done = 0;
    if (!attr_done) {
        KG_HIP(hipFuncSetAttribute((const void*)conv1x1_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        KG_HIP(hipFuncSetAttribute((const void*)conv1x1_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        attr_done = 1;
    }
    const long ntiles = (M + 255) / 256;
    const int ny = kg_cdiv(Cout, 64);
    int per_cu = smem <= 16384 ? 4 : (smem <= 32768 ? 3 : (smem <= 65536 ? 2 : 1));
    long gx = (long)256 * per_cu / ny;
    if (gx < 64) gx = 64;
    if (gx > ntiles) gx = ntiles;
    if (K % 128 == 0) hipLaunchKernelGGL(conv1x1_kernel<4>, dim3((unsigned)gx, ny), dim3(256), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv1x1_kernel<2>, dim3((unsigned)gx, ny), dim3(256), smem, (hipStream_t)stream, a);
    KG_CHECK_LAUNCH("conv1x1");
    kg_note_kernel(K % 128 == 0 ? "conv1x1_kernel<4>" : "conv1x1_kernel<2>");
    return KG_OK;
}

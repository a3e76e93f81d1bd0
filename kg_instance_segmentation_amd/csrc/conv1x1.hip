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

struct C1Args {
    const bf16_t* x; const bf16_t* w; const float* bias;
    bf16_t* y; const bf16_t* res; const bf16_t* mask;
    long M;
    int K, ldx, Cout, ldy, ldres, ldmask, relu, wK;   // wK: row pitch (elements) of the packed weight matrix
};

__global__ __launch_bounds__(256) void conv1x1_kernel(const C1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [K/64][64 rows][128 B], swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * 64;
    const int nk64 = a.K / 64;

    // ---- stage the weight slab once ---------------------------------------------------------------------------
    for (int e = tid; e < nk64 * 64 * 8; e += 256) {
        const int c = e & 7, r = (e >> 3) & 63, kc = e >> 9;
        const uint4 v = *reinterpret_cast<const uint4*>(a.w + (long)(c0 + r) * a.wK + kc * 64 + c * 8);
        const int key = 2 * ((r >> 4) & 3) + ((r >> 1) & 1);
        *reinterpret_cast<uint4*>(smem + kc * 8192 + r * 128 + ((c ^ key) * 16)) = v;
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
    const int nks = a.K / 32;
    auto rowptr = [&](long tile, int j) -> const bf16_t* {
        long m = tile * 256 + wave * 64 + j * 16 + lm;
        if (m >= a.M) m = a.M - 1;   // clamped rows are computed and discarded
        return a.x + m * a.ldx + g * 8;
    };
    bf16x8 bcur[4], bnxt[4];
    long tile = blockIdx.x;
    if (tile < ntiles) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bcur[j] = *reinterpret_cast<const bf16x8*>(rowptr(tile, j));
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
        for (int ks = 0; ks < nks; ++ks) {
            // prefetch the next k-step's pixel fragments (next tile's first k-step at the end)
            if (ks + 1 < nks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bnxt[j] = *reinterpret_cast<const bf16x8*>(rp[j] + (ks + 1) * 32);
            } else if (tnext < ntiles) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bnxt[j] = *reinterpret_cast<const bf16x8*>(rowptr(tnext, j));
            }
            bf16x8 af[4];
            const unsigned char* wb = smem + (ks >> 1) * 8192;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wb + a_off[i][ks & 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bcur[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) bcur[j] = bnxt[j];
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
                    for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r] + bv[i * 4 + r];
                if (a.res) {
                    const bf16_t* rq = a.res + m * a.ldres + cb;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (full || cb + e < a.Cout) v[e] += bf2f(rq[e]);
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
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
    const int smem = (K / 64) * 8192;
    static int attr_done = 0;
    if (!attr_done) {
        KG_HIP(hipFuncSetAttribute((const void*)conv1x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        attr_done = 1;
    }
    const long ntiles = (M + 255) / 256;
    const int ny = kg_cdiv(Cout, 64);
    int per_cu = smem <= 16384 ? 4 : (smem <= 32768 ? 3 : (smem <= 65536 ? 2 : 1));
    long gx = (long)256 * per_cu / ny;
    if (gx < 64) gx = 64;
    if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(conv1x1_kernel, dim3((unsigned)gx, ny), dim3(256), smem, (hipStream_t)stream, a);
    KG_CHECK_LAUNCH("conv1x1");
    return KG_OK;
}

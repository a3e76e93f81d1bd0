// conv_igemm.hip -- bf16 MFMA implicit-GEMM convolution for gfx950 (forward and data-gradient).
//
// Replaces the torch conv2d calls on KGnet's hot path (reference KGnet.py:22-29, 131-209 via
// torch.nn.Conv2d -> cuDNN/MIOpen).  One kernel family:
//   D[cout][pixel] = sum_k  Wp[cout][k] * Xg[pixel][k],   k = (tap, cin)
// with the weights as the MFMA "A" operand and the gathered NHWC pixels as "B", so that each lane
// ends up owning 4*CF consecutive output channels of one pixel (wide NHWC stores).
//   * activations: bf16, pixel-major rows [row][ld] (NHWC), 16-byte channel chunks
//   * weights: pre-packed bf16 [Cout_pad][K], K = ntaps*cin_pad (see pack kernels below)
//   * accumulation: fp32 in MFMA accumulators (v_mfma_f32_16x16x32_bf16)
//   * LDS: [sub-step][row][32 k] tiles, XOR-swizzled 16-byte slots -> conflict-free ds_read_b128
//   * epilogue: bias, residual add, ReLU, ReLU-mask (for dgrad), bf16 NHWC and/or fp32 NCHW store
#include "kg_common.h"
#include <stdlib.h>

#include "conv_args.h"

__device__ __forceinline__ int fperm(int q) { return (4 - q) & 3; }

template <int WC, int WP, int CF, int KS>
__global__ __launch_bounds__(WC* WP * 64) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int TC = WC * CF * 16, TP = WP * 64, NT = WC * WP * 64;
    constexpr int XPT = TP * 4 / NT;
    constexpr int WPT = (TC * 4 + NT - 1) / NT;
    constexpr int LOG_CF = CF == 4 ? 2 : (CF == 2 ? 1 : 0);
    constexpr int W_BYTES = KS * TC * 64, X_BYTES = KS * TP * 64, BUF_BYTES = W_BYTES + X_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* taptab = (int*)(smem + 2 * BUF_BYTES);  // [64] packed (dy<<16)|(dx&0xffff), already *dil - pad

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;
    const int m0 = blockIdx.x * TP, c0 = blockIdx.y * TC;

    if (tid < 64) {
        int t = tid, dy = t / a.KW, dx = t - dy * a.KW;
        int oy = dy * a.dil - a.pad, ox = dx * a.dil - a.pad;
        taptab[t] = (oy << 16) | (ox & 0xffff);
    }

    // ---- per-thread staging assignment --------------------------------------------------
    const int slot = tid & 3;
    const int xrow0 = tid >> 2;                                // + (NT/4)*i
    const int xchunk = slot ^ fperm((xrow0 >> 2) & 3);         // logical k-chunk this thread fetches (rows step by NT/4: same key)
    const int wrow0 = tid >> 2;
    static_assert((NT / 4) % 16 == 0, "pixel-row swizzle key must not depend on the staging pass");
    int py[XPT], px[XPT], pbase[XPT];  // output coords / row base; pbase<0 => row out of range
    int ph[XPT], pw[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        int m = m0 + xrow0 + (NT / 4) * i;
        if (m >= a.M) { pbase[i] = -1; py[i] = px[i] = 0; ph[i] = pw[i] = 0; continue; }
        if (a.mode >= 2) {
            int2 d = a.rowdesc[m];
            py[i] = d.x >> 16; px[i] = d.x & 0xffff; ph[i] = d.y >> 16; pw[i] = d.y & 0xffff; pbase[i] = m;
        } else {
            int ohw = a.OH * a.OW;
            int n = m / ohw, rem = m - n * ohw;
            int oy = rem / a.OW, ox = rem - oy * a.OW;
            py[i] = oy; px[i] = ox; pbase[i] = n * a.H * a.W; ph[i] = a.H; pw[i] = a.W;
        }
    }
    __syncthreads();

    uint4 xr[KS][XPT], wr[KS][WPT];
    const int nk = a.K / (32 * KS);
    const int smask = (1 << a.stride_log2) - 1;

    auto load_tiles = [&](int kk) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int q = (kk * KS + s) * 4 + xchunk;
            int tap = (int)(((unsigned)q * (unsigned)a.cpt_magic) >> 24);
            const int cc = q - tap * a.cpt;
            const bool tapok = tap < a.ntaps;
            tap = tap < 63 ? tap : 63;
            const int tt = taptab[tap];
            const int dy = tt >> 16, dx = (int)(short)(tt & 0xffff);
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (tapok && pbase[i] >= 0) {
                    int iy, ix; bool ok; long row;
                    if (a.mode == 0) {
                        iy = (py[i] << a.stride_log2) + dy; ix = (px[i] << a.stride_log2) + dx;
                        ok = (unsigned)iy < (unsigned)ph[i] && (unsigned)ix < (unsigned)pw[i];
                        row = (long)pbase[i] + (long)iy * pw[i] + ix;
                    } else if (a.mode == 1) {
                        int ty = py[i] - dy, tx = px[i] - dx;
                        ok = ty >= 0 && tx >= 0 && ((ty | tx) & smask) == 0;
                        iy = ty >> a.stride_log2; ix = tx >> a.stride_log2;
                        ok = ok && iy < ph[i] && ix < pw[i];
                        row = (long)pbase[i] + (long)iy * pw[i] + ix;
                    } else {
                        int sy = a.mode == 2 ? dy : -dy, sx = a.mode == 2 ? dx : -dx;
                        iy = py[i] + sy; ix = px[i] + sx;
                        ok = (unsigned)iy < (unsigned)ph[i] && (unsigned)ix < (unsigned)pw[i];
                        row = (long)pbase[i] + (long)sy * pw[i] + sx;
                    }
                    if (ok) v = *reinterpret_cast<const uint4*>(a.x + row * a.ldx + a.km.xoff(cc));
                }
                xr[s][i] = v;
            }
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                int r = wrow0 + (NT / 4) * i;
                const int qw = (kk * KS + s) * 4 + (slot ^ fperm((r >> (2 + LOG_CF)) & 3));
                if (TC * 4 >= NT || r < TC)
                    wr[s][i] = *reinterpret_cast<const uint4*>(a.w + (long)(c0 + r) * a.K + qw * 8);
            }
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char* sw = smem + buf * BUF_BYTES;
        unsigned char* sx = sw + W_BYTES;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int i = 0; i < XPT; ++i)
                *reinterpret_cast<uint4*>(sx + s * TP * 64 + (xrow0 + (NT / 4) * i) * 64 + slot * 16) = xr[s][i];
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                int r = wrow0 + (NT / 4) * i;
                if (TC * 4 >= NT || r < TC)
                    *reinterpret_cast<uint4*>(sw + s * TC * 64 + r * 64 + slot * 16) = wr[s][i];
            }
        }
    };

    f32x4 acc[CF][4];
#pragma unroll
    for (int i = 0; i < CF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing (constant over k)
    const int lm = lane & 15, g = lane >> 4;
    int a_off[CF], b_off[4];
#pragma unroll
    for (int i = 0; i < CF; ++i) {
        int r = wc * CF * 16 + (lm >> 2) * (4 * CF) + i * 4 + (lm & 3);
        a_off[i] = r * 64 + ((g ^ fperm((r >> (2 + LOG_CF)) & 3)) * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = wp * 64 + j * 16 + lm;
        b_off[j] = r * 64 + ((g ^ fperm((r >> 2) & 3)) * 16);
    }

    load_tiles(0);
    for (int kk = 0; kk < nk; ++kk) {
        const int buf = kk & 1;
        store_tiles(buf);
        __syncthreads();
        if (kk + 1 < nk) load_tiles(kk + 1);
        const unsigned char* sw = smem + buf * BUF_BYTES;
        const unsigned char* sx = sw + W_BYTES;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 af[CF], bfr[4];
#pragma unroll
            for (int i = 0; i < CF; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sw + s * TC * 64 + a_off[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sx + s * TP * 64 + b_off[j]);
#pragma unroll
            for (int i = 0; i < CF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = KG_MFMA16(af[i], bfr[j], acc[i][j]);
        }
    }

    // ---- epilogue ------------------------------------------------------------------------
    constexpr int NV = 4 * CF;
    const int cb = c0 + wc * CF * 16 + g * NV;  // first output channel of this lane
    if (cb >= a.Cout) return;
    float bv[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
    float sv[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) sv[e] = (a.oscale && cb + e < a.Cout) ? a.oscale[cb + e] : 1.f;
    const int ohw = a.OH * a.OW;
    const EpiArgs ep = kg_epi(a);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wp * 64 + j * 16 + lm;
        if (m >= a.M) continue;
        float v[NV];
#pragma unroll
        for (int i = 0; i < CF; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] = KG_ACC(acc[i][j][r]) * sv[i * 4 + r] + bv[i * 4 + r];
        if (a.y_f32) {       // fp32 exports carry no residual / mask in KGnet (head maps)
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = kg_relu(v[e]);
            }
        }
        if (a.y_f32) {
            const int n = m / ohw, pix = m - n * ohw;
#pragma unroll
            for (int e = 0; e < NV; ++e)
                if (cb + e < a.Cout) a.y_f32[((long)n * a.f32_C + cb + e) * ohw + pix] = v[e];
        }
        if (a.y) kg_conv_epilogue<NV>(ep, m, cb, v);      // (last: the split store consumes v)
    }
}

template <int WC, int WP, int CF, int KS>
static int launch_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int TC = WC * CF * 16, TP = WP * 64, NT = WC * WP * 64;
    constexpr int smem = 2 * (KS * TC * 64 + KS * TP * 64) + 256;
    static KgPerDevice attr_done;
    if (attr_done.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)conv_igemm_kernel<WC, WP, CF, KS>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    dim3 grid(kg_cdiv(a.M, TP), kg_cdiv(a.Cout, TC));
    hipLaunchKernelGGL((conv_igemm_kernel<WC, WP, CF, KS>), grid, dim3(NT), smem, st, a);
    KG_CHECK_LAUNCH("conv_igemm");
    KG_KNAME(kname, "conv_igemm_kernel<%d, %d, %d, %d>", WC, WP, CF, KS);
    kg_note_kernel(kname);
    return KG_OK;
}

static int magic_for(int cpt, int qmax, int* magic) {
    int mg = ((1 << 24) + cpt - 1) / cpt;   // tap = (q * mg) >> 24; q*mg < taps * 2^24 < 2^31
    for (int q = 0; q <= qmax; ++q)
        if ((int)(((unsigned)q * (unsigned)mg) >> 24) != q / cpt) return 0;
    *magic = mg;
    return 1;
}

extern "C" int kg_conv2d_igemm(const void* x, const void* w, const float* bias, void* y, float* y_f32,
                               const void* res, const void* mask, const int* rowdesc, int M, int H, int W,
                               int OH, int OW, int cin_pad, int ldx, int Cout, int ldy, int ldres, int ldmask,
                               int K, int KH, int KW, int stride, int pad, int dil, int mode, int relu,
                               int f32_C, int tile, const kg_planes_t* planes, void* stream) {
    // planes: a = x (cin_pad = channels of ONE plane; the packed weights hold vplanes * cin_pad virtual channels per tap,
    // see kg_pack_weight), b = res, y = y
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_conv2d_igemm: bad kg_planes_t");
    int segs_[3];
    const int vplanes = kg_kmap_segs(pp.a_planes, pp.w_planes, segs_);
    const int cin_virt = vplanes * cin_pad;
    KG_CHECK_ARG(!(y_f32 && (res || mask)), "kg_conv2d_igemm: fp32 exports take no residual / mask");
    KG_CHECK_ARG(x && w && (y || y_f32), "kg_conv2d_igemm: null pointer");
    KG_CHECK_ARG(cin_pad > 0 && cin_pad % 8 == 0 && ldx % 8 == 0, "kg_conv2d_igemm: cin_pad/ldx must be multiples of 8 (got %d, %d)", cin_pad, ldx);
    KG_CHECK_ARG(K % 64 == 0 && K >= KH * KW * cin_virt, "kg_conv2d_igemm: K=%d must be a multiple of 64 and >= taps*cin_pad=%d", K, KH * KW * cin_virt);
    KG_CHECK_ARG(KH * KW <= 49 && KH * KW >= 1, "kg_conv2d_igemm: at most 49 taps");
    KG_CHECK_ARG(stride == 1 || stride == 2, "kg_conv2d_igemm: stride must be 1 or 2");
    KG_CHECK_ARG(mode >= 0 && mode <= 3, "kg_conv2d_igemm: bad mode");
    KG_CHECK_ARG(mode < 2 || (rowdesc && stride == 1), "kg_conv2d_igemm: ragged mode needs rowdesc and stride 1");
    KG_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0, "kg_conv2d_igemm: x/w must be 16-byte aligned");
    KG_CHECK_ARG(M > 0 && Cout > 0, "kg_conv2d_igemm: empty problem");
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.y_f32 = y_f32;
    a.res = (const bf16_t*)res; a.mask = (const bf16_t*)mask; a.rowdesc = (const int2*)rowdesc;
    a.M = M; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.ldx = ldx; a.ldy = ldy; a.ldres = ldres; a.ldmask = ldmask;
    a.Cout = Cout; a.K = K; a.cpt = cin_virt / 8; a.ntaps = KH * KW; a.KW = KW;
    kg_fill_planes(a, pp, cin_pad, 8);
    a.stride_log2 = stride == 2 ? 1 : 0; a.pad = pad; a.dil = dil; a.mode = mode; a.relu = relu; a.f32_C = f32_C;
    KG_CHECK_ARG(magic_for(a.cpt, K / 8, &a.cpt_magic), "kg_conv2d_igemm: no exact magic divisor for cpt=%d", a.cpt);
    hipStream_t st = (hipStream_t)stream;
    if (tile == 6) {   // split-K over the waves of a 64 x 64 tile: launches whose output offers too few tiles to fill the chip (conv_tiny.hip)
        KG_CHECK_ARG(cin_pad % 64 == 0 && y && !y_f32 && mode <= 1 && dil == 1, "kg_conv2d_igemm: tile 6 needs cin_pad %% 64 == 0, a rows output and a dense mode");
        a.km = kg_make_kmap(cin_pad, 64, pp.a_planes, pp.a_pstride, pp.w_planes);
        return kg_launch_conv_tiny(a, cin_virt, st);
    }
    constexpr int use_gather2 = 2;   // 2: also 64-cout 3x3 convs (half the cout tile idle, still 2x the 64 x 256 tile)
    if (use_gather2 && tile == 0 && cin_pad % 64 == 0 && y && !y_f32 && (Cout > 64 || (use_gather2 >= 2 && Cout == 64 && (KH * KW > 1 || vplanes > 1))) && dil == 1) {
        a.km = kg_make_kmap(cin_pad, 64, pp.a_planes, pp.a_pstride, pp.w_planes);
        // deep-prefetch LDS-ring variant (conv_gather.hip); it may carry the BatchNorm statistics epilogue when armed
        // (forward statistics: a plain conv output; backward statistics -- kg_conv_bstats_begin --: the dense input gradient, whole 64-channel blocks)
        const bool bwd_stats = kg_conv_stats().bs.x != nullptr;
        return kg_launch_conv_gather(a, cin_virt, st, bwd_stats ? (mode == 1 && Cout % 64 == 0) : (!relu && !res && !mask && (mode == 0 || mode == 2)));
    }
    constexpr int use_small = 6;   // (6: the stride-2 7x7 stem too)
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (use_small && tile == 0 && cin_pad == 8 && (vplanes == 1 || K >= 32 * vplanes * ((KH * KW + 3) / 4)) && KH * KW <= use_small * 9 && y && !y_f32 && Cout % 8 == 0 && ldy % 8 == 0 && al16(y) &&
        (!res || (ldres % 8 == 0 && al16(res))) && (!mask || (ldmask % 8 == 0 && al16(mask))) && dil == 1)
        return kg_launch_conv_small(a, st);   // <= 8 input channels: direct VALU kernel (conv_small.hip)
    // tile: 0 auto; 1 = 16 couts x 256 px; 2 = 32 x 256; 3 = 64 x 256; 4 = 128 x 128; 5 = 64 x 128; (6 = split-K 64 x 64, above)
    if (tile == 0) tile = Cout <= 16 ? 1 : (Cout <= 32 ? 2 : (Cout <= 64 ? 3 : 4));
    switch (tile) {
        case 1: return launch_cfg<1, 4, 1, 2>(a, st);
        case 2: return launch_cfg<1, 4, 2, 2>(a, st);
        case 3: return launch_cfg<1, 4, 4, 2>(a, st);
        case 4: return launch_cfg<2, 2, 4, 2>(a, st);
        case 5: return launch_cfg<1, 2, 4, 2>(a, st);
        default: kg_set_error("kg_conv2d_igemm: bad tile %d", tile); return KG_ERR_ARG;
    }
}

// ---- weight packing ---------------------------------------------------------------------------
// fp32 OIHW [Cout][Cin][KH][KW] -> bf16 rows of a packed matrix.
//  transposed == 0 (forward):  dst[(row0+co)*K + tap*cin_pad + c0 + ci]         = w[co][ci][tap]
//  transposed == 1 (dgrad):    dst[(row0+ci)*K + tap*cin_pad + c0 + co]         = w[co][ci][tap]
// Padding entries must have been zeroed by the caller (hipMemsetAsync once at plan time).
// forward: one block per (co, 64-ci chunk): contiguous fp32 reads [ci][tap], LDS transpose, 128-byte bf16 writes
// along ci for every tap.  transposed: one block per (64-co chunk, ci): reads `taps` contiguous floats per co,
// writes 128-byte bf16 runs along co for every tap.
// Split-bf16 planes (kg_common.h): with xP activation planes and wP weight planes the row of a tap holds nv virtual planes of
// cin_pad channels each, virtual plane v = a copy of w plane wj(v) (kg_plane_pairs: smallest products first); cin_virt =
// nv * cin_pad channels per tap; xP = wP = 1 is the plain bf16 layout.
struct PackPlanes { int wP, nv, cin_virt; unsigned wtab; };
static inline PackPlanes make_pack_planes(int xP, int wP, int cin_pad) {
    PackPlanes q;
    int xi[6], wj[6];
    q.wP = wP < 1 ? 1 : wP;
    q.nv = kg_plane_pairs(xP < 1 ? 1 : xP, q.wP, xi, wj);
    q.wtab = 0;
    for (int v = 0; v < q.nv; ++v) q.wtab |= (unsigned)wj[v] << (2 * v);
    q.cin_virt = q.nv * cin_pad;
    return q;
}
__device__ __forceinline__ void pack_store_planes(bf16_t* __restrict__ rowp, float v, const PackPlanes& q, int cin_pad) {
    // rowp = &dst[row * K + tap * cin_virt + c0 + channel] of virtual plane 0
    bf16_t pl[3];
    v *= KG_WSCALE;       // (half build: packed weights are stored times 2^12, the conv epilogues scale the accumulators back: KG_ACC)
    for (int j = 0; j < 3; ++j) { pl[j] = f2bf(v); v -= bf2f(pl[j]); }
    for (int k = 0; k < q.nv; ++k) {
        const int j = (q.wtab >> (2 * k)) & 3;
        rowp[(long)k * cin_pad] = j == 0 ? pl[0] : (j == 1 ? pl[1] : pl[2]);
    }
}
__device__ __forceinline__ void pack_weight_block(float* tile, const float* __restrict__ w, bf16_t* __restrict__ dst, int Cout, int Cin,
                                                  int taps, int K, int cin_pad, int row0, int c0, int transposed,
                                                  const int* __restrict__ rowmap, int bx, int by, const PackPlanes q, int tap_pitch = 0) {
    if (!transposed) {
        const int co = bx, ci0 = by * 64;
        const int nci = Cin - ci0 < 64 ? Cin - ci0 : 64;
        const float* src = w + ((long)co * Cin + ci0) * taps;
        for (int j = threadIdx.x; j < nci * taps; j += 256) { int ci = j / taps, tap = j - ci * taps; tile[ci * 50 + tap] = src[j]; }
        __syncthreads();
        for (int e = threadIdx.x; e < taps * 64; e += 256) {
            const int tap = e >> 6, ci = e & 63;
            if (ci < nci) pack_store_planes(dst + (long)(rowmap ? rowmap[co] : row0 + co) * K + (long)tap * q.cin_virt + c0 + ci0 + ci, tile[ci * 50 + tap], q, cin_pad);
        }
    } else {
        const int co0 = bx * 64, ci = by;
        const int nco = Cout - co0 < 64 ? Cout - co0 : 64;
        for (int j = threadIdx.x; j < nco * taps; j += 256) {
            int co = j / taps, tap = j - co * taps;
            tile[co * 50 + tap] = w[((long)(co0 + co) * Cin + ci) * taps + tap];
        }
        __syncthreads();
        // tap_pitch > 0 (the narrow input-gradient layout of conv_halo.hip, GM = 3 / 4): the columns of a kernel row are tap_pitch slots apart
        // instead of KW -- tap (ky, kx) lands in slot ky * tap_pitch + kx (the slots kx >= KW stay zero)
        const int kw = taps == 49 ? 7 : (taps == 9 ? 3 : taps);
        for (int e = threadIdx.x; e < taps * 64; e += 256) {
            const int tap = e >> 6, co = e & 63;
            const int tslot = tap_pitch > 0 ? (tap / kw) * tap_pitch + tap % kw : tap;
            if (co < nco) pack_store_planes(dst + (long)(row0 + ci) * K + (long)tslot * q.cin_virt + c0 + co0 + co, tile[co * 50 + tap], q, cin_pad);
        }
    }
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst, int Cout,
                                                          int Cin, int taps, int K, int cin_pad, int row0, int c0,
                                                          int transposed, const int* __restrict__ rowmap, const PackPlanes q, int tap_pitch) {
    __shared__ float tile[64 * 50];
    pack_weight_block(tile, w, dst, Cout, Cin, taps, K, cin_pad, row0, c0, transposed, rowmap, blockIdx.x, blockIdx.y, q, tap_pitch);
}

// All weight (re)packs of a training step in ONE launch: the 170+ per-tensor launches were launch-bound (5 us each).
struct PackJob {   // 80 bytes, mirrored by ops.PackQueue
    const float* w; bf16_t* dst; const int* rowmap;
    int Cout, Cin, taps, K, cin_pad, row0, c0, transposed, gx, blk0, xP, wP, tap_stride, tap_pitch;   // tap_stride: 0 = vplanes * cin_pad; tap_pitch: pack_weight_block
};
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const PackJob* __restrict__ jobs, int njobs) {
    __shared__ float tile[64 * 50];
    int lo = 0, hi = njobs - 1;          // last job whose first block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    const int b = blockIdx.x - j.blk0;
    PackPlanes q;
    q.wP = j.wP < 1 ? 1 : j.wP;
    {   // kg_plane_pairs on the device: descending i + j, then descending j
        const int xP = j.xP < 1 ? 1 : j.xP, T = xP > q.wP ? xP : q.wP;
        q.nv = 0; q.wtab = 0;
        for (int sum = T - 1; sum >= 0; --sum)
            for (int jj = q.wP - 1; jj >= 0; --jj) {
                const int i = sum - jj;
                if (i >= 0 && i < xP) { q.wtab |= (unsigned)jj << (2 * q.nv); ++q.nv; }
            }
        q.cin_virt = j.tap_stride > 0 ? j.tap_stride : q.nv * j.cin_pad;
    }
    pack_weight_block(tile, j.w, j.dst, j.Cout, j.Cin, j.taps, j.K, j.cin_pad, j.row0, j.c0, j.transposed, j.rowmap, b % j.gx, b / j.gx, q, j.tap_pitch);
}

// x_planes / w_planes: split-bf16 layout of the packed matrix (1, 1 = plain bf16); cin_pad = channels of ONE plane, c0 = channel
// offset inside a plane copy, K >= taps * vplanes * cin_pad.
extern "C" int kg_pack_weight(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int K, int cin_pad,
                              int row0, int c0, int transposed, int x_planes, int w_planes, void* stream) {
    KG_CHECK_ARG(w && dst, "kg_pack_weight: null pointer");
    KG_CHECK_ARG(KH * KW <= 49, "kg_pack_weight: at most 49 taps");
    KG_CHECK_ARG(x_planes <= 3 && w_planes <= 3, "kg_pack_weight: at most 3 planes");
    const PackPlanes q = make_pack_planes(x_planes, w_planes, cin_pad);
    KG_CHECK_ARG(K >= KH * KW * q.cin_virt, "kg_pack_weight: K too small for the plane layout");
    dim3 grid = transposed ? dim3((Cout + 63) / 64, Cin) : dim3(Cout, (Cin + 63) / 64);
    hipLaunchKernelGGL(pack_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)dst, Cout,
                       Cin, KH * KW, K, cin_pad, row0, c0, transposed, (const int*)nullptr, q, 0);
    KG_CHECK_LAUNCH("pack_weight");
    return KG_OK;
}

// Transposed (input-gradient) packing of a NARROW conv for conv_halo.hip's GM = 3 / 4 variants (Cout <= tap_stride in {8, 16} channels, single plane):
//   dst[(row0 + ci) * K + (ky * tap_pitch + kx) * tap_stride + c0 + co] = w[co][ci][ky][kx]
// -- the (kernel column, channel) pairs of a kernel row side by side, tap_pitch = 8 columns per row (7 real + 1 zero): 64 / 128 packed columns per kernel row.
extern "C" int kg_pack_weight_narrow(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int K, int row0, int c0, int tap_stride, int tap_pitch,
                                     void* stream) {
    KG_CHECK_ARG(w && dst, "kg_pack_weight_narrow: null pointer");
    KG_CHECK_ARG(KH == KW && (KH == 7 || KH == 3) && tap_pitch >= KW && tap_stride >= 8 && tap_stride % 8 == 0 && c0 + Cout <= tap_stride,
                 "kg_pack_weight_narrow: bad layout");
    KG_CHECK_ARG(K >= KH * tap_pitch * tap_stride, "kg_pack_weight_narrow: K too small");
    PackPlanes q = make_pack_planes(1, 1, tap_stride);
    q.cin_virt = tap_stride;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((Cout + 63) / 64, Cin), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)dst, Cout,
                       Cin, KH * KW, K, tap_stride, row0, c0, 1, (const int*)nullptr, q, tap_pitch);
    KG_CHECK_LAUNCH("pack_weight_narrow");
    return KG_OK;
}

// forward packing with a row table: dst row of output channel co = rowmap[co] (device int array)
extern "C" int kg_pack_weight_rows(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int K, int cin_pad,
                                   const int* rowmap, int c0, int x_planes, int w_planes, int tap_stride, void* stream) {
    // tap_stride: channels per tap row of the packed matrix when several plane groups share it (the fused second-layer heads:
    // [head][virtual planes][C] per tap -> 3 * vplanes * C); 0 = vplanes * cin_pad
    KG_CHECK_ARG(w && dst && rowmap, "kg_pack_weight_rows: null pointer");
    KG_CHECK_ARG(KH * KW <= 49, "kg_pack_weight_rows: at most 49 taps");
    KG_CHECK_ARG(x_planes <= 3 && w_planes <= 3, "kg_pack_weight_rows: at most 3 planes");
    PackPlanes q = make_pack_planes(x_planes, w_planes, cin_pad);
    if (tap_stride > 0) q.cin_virt = tap_stride;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(Cout, (Cin + 63) / 64), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)dst,
                       Cout, Cin, KH * KW, K, cin_pad, 0, c0, 0, rowmap, q, 0);
    KG_CHECK_LAUNCH("pack_weight_rows");
    return KG_OK;
}

// jobs: device array of njobs PackJob records (see struct PackJob: {w, dst, rowmap, Cout, Cin, taps, K, cin_pad, row0, c0,
// transposed, gx, blk0, xP, wP, tap_stride}; gx = grid x of the job = Cout (forward) or ceil(Cout/64) (transposed), blk0 = first block of the
// job in this launch); total_blocks = sum of the jobs' gx * gy.
extern "C" int kg_pack_weight_batch(const void* jobs, int njobs, int total_blocks, void* stream) {
    KG_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0, "kg_pack_weight_batch: empty batch");
    hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs, njobs);
    KG_CHECK_LAUNCH("pack_weight_batch");
    return KG_OK;
}

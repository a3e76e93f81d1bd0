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
// Tiles are double-buffered through registers (loads of tile t+1 are in flight while tile t is multiplied).
//
// LDS layouts are chosen so that (a) the 32-lane halves of a transpose read hit disjoint banks and (b) every fragment
// address is  lane-constant register + k-step immediate:  the k-step loop contains no address arithmetic.
//   k-step s = tile rows 2s, 2s+1; lane group G reads row 2s + (G&1), columns (G>>1)*8 + h*4 + (i>>2)   (h = 0,1)
//   dY tile : [256 px][128 B], 32-byte unit index XOR f(r), f(r) = ((r>>1)&1) | (((r>>4)&1)<<1)
//   X halo  : [HWD rows][PITCH], PITCH = HWD*32*CIF rounded up to 128 (mod 256) so consecutive rows alternate bank
//             halves; CIF == 4: 32-byte unit XOR (x & 3)
#include "kg_common.h"

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
// head convs, whose dY tile is mostly padding: KGnet.py:161-209 `.2` layers)
template <int KS, int CIF, int NCF = 4, bool BIAS = false>
__global__ __launch_bounds__(512) void wgrad_halo_kernel(const WgHaloArgs a) {
    using GE = WgGeom<KS, CIF>;
    constexpr int PAD = GE::PAD, HWD = GE::HWD, HPIX = HWD * HWD, T = GE::T, XB = GE::XB, PITCH = GE::PITCH;
    constexpr int DY_BYTES = GE::DY_BYTES, BUF = GE::BUF, UNITS = GE::UNITS, UPW = GE::UPW;
    constexpr int DYPT = 256 * 8 / 512;                // dY 16-byte chunks per thread (4)
    constexpr int XCH = HPIX * 2 * CIF, XPT = (XCH + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;   // (a readfirstlane'd wave id schedules 4 % slower)
    const int n_ci_tiles = (a.cin_lim + 16 * CIF - 1) / (16 * CIF);
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in linear-id order, and all
    // (co, ci) blocks of one pixel split read the SAME dY tiles / X halos.  Put the blocks of a split on one XCD so that its
    // L2 fetches each tile once instead of all 8 L2s fetching it.
    int blk = blockIdx.x, split = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int L = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = L & 7, slot = L >> 3;
        blk = slot % gridDim.x; split = (slot / gridDim.x) * 8 + xcd;
    }
    const int ci0 = (blk % n_ci_tiles) * 16 * CIF, co0 = (blk / n_ci_tiles) * 64;
    const int G = lane >> 4, i16 = lane & 15;

    f32x4 acc[UPW][NCF];
#pragma unroll
    for (int q = 0; q < UPW; ++q)
#pragma unroll
        for (int c = 0; c < NCF; ++c) acc[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Bias gradient for free: db[co] = sum_px dY[px][co] is the weight gradient of a constant-1 input channel, i.e. one more
    // (tap, ci-fragment) unit with an all-ones B fragment.  49 (7x7) and 36 (3x3) units leave a wave with a free unit slot
    // (BW, slot BQ < UPW), so the extra NCF MFMAs per k-step do not lengthen the workgroup's critical path.
    constexpr int BW = UNITS % 8, BQ = UNITS / 8;
    static_assert(BQ < UPW, "no free unit slot for the bias-gradient unit");
    const bool do_bias = BIAS && ci0 == 0 && wave == BW;   // (a separate instantiation: the extra accumulators cost the 3x3
                                                           // variant 23 spilled registers, so only the 7x7 kernels carry the unit)
    f32x4 accb[BIAS ? NCF : 1];
#pragma unroll
    for (int c = 0; c < (BIAS ? NCF : 1); ++c) accb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // ---- lane-constant fragment addresses (k-step 0); k-step s adds an immediate ------------------------------------
    // dY^T fragment c (co block), half h: tile row r = (G&1)*16 + (G>>1)*8 + h*4 + (i16>>2)   (+ s*32)
    int ay[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = (G & 1) * 16 + (G >> 1) * 8 + h * 4 + (i16 >> 2);
#pragma unroll
        for (int c = 0; c < 4; ++c) ay[c][h] = r * 128 + ((c ^ dyf(r)) * 32) + (i16 & 3) * 8;
    }
    // X^T fragment of unit q (tap, ci block f), half h: halo row (G&1) + ky (+ 2s), column (G>>1)*8 + h*4 + (i16>>2) + kx
    int bx[UPW][2];
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
            bx[q][h] = ((G & 1) + ky) * PITCH + xh * XB + unit * 32 + (i16 & 3) * 8;
        }
    }

    uint4 dyr[DYPT], xr[XPT];
    const int tiles_plane = a.tiletab ? a.ntiles : a.N * a.tiles_x * a.tiles_y;
    const int tiles_total = a.wp.n * tiles_plane;

    // per-thread staging coordinates (tile independent): dividing per element and tile cost ~260 VALU per tile and wave,
    // a third of the tile's MFMA time
    int dy_ry[DYPT], dy_rx[DYPT];
#pragma unroll
    for (int k = 0; k < DYPT; ++k) { const int r = (tid + k * 512) >> 3; dy_ry[k] = r >> 4; dy_rx[k] = r & 15; }
    const int dy_c = co0 + (tid & 7) * 8;
    const bool dy_cok = dy_c < a.cout_lim;
    int x_hy[XPT], x_hx[XPT], x_c[XPT];
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
        const int e = tid + k * 512;
        const int p = e / (2 * CIF), c = e - p * (2 * CIF);
        const int hy = p / HWD;
        x_hy[k] = e < XCH ? hy - PAD : -(1 << 20);      // out-of-range elements fail the bounds test below
        x_hx[k] = p - hy * HWD - PAD;
        x_c[k] = ci0 + c * 8;
    }
    auto load_tile = [&](int vt) {
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
        const bf16_t* dyb = a.dy + a.wp.doff[pr] + base * a.lddy + dy_c;
        const bf16_t* xb = a.x + a.wp.xoff[pr] + base * a.ldx;
        const int hrem = Hd - oy0, wrem = Wd - ox0;
#pragma unroll
        for (int k = 0; k < DYPT; ++k) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (dy_ry[k] < hrem && dy_rx[k] < wrem && dy_cok)
                v = *reinterpret_cast<const uint4*>(dyb + (long)(dy_ry[k] * Wd + dy_rx[k]) * a.lddy);
            dyr[k] = v;
        }
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((unsigned)(oy0 + x_hy[k]) < (unsigned)Hd && (unsigned)(ox0 + x_hx[k]) < (unsigned)Wd && x_c[k] < a.cin_lim)
                v = *reinterpret_cast<const uint4*>(xb + (long)(x_hy[k] * Wd + x_hx[k]) * a.ldx + x_c[k]);
            xr[k] = v;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sy = smem + buf * BUF;
        unsigned char* sx = sy + DY_BYTES;
#pragma unroll
        for (int k = 0; k < DYPT; ++k) {
            const int e = tid + k * 512, r = e >> 3, c8 = e & 7;
            *reinterpret_cast<uint4*>(sy + r * 128 + (((c8 >> 1) ^ dyf(r)) * 32) + (c8 & 1) * 16) = dyr[k];
        }
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int e = tid + k * 512;
            if (e < XCH) {
                const int p = e / (2 * CIF), c = e - p * (2 * CIF);
                const int hy = p / HWD, hx = p - hy * HWD;
                const int unit = CIF == 1 ? 0 : ((c >> 1) ^ (hx & (CIF - 1)));
                *reinterpret_cast<uint4*>(sx + hy * PITCH + hx * XB + unit * 32 + (c & 1) * 16) = xr[k];
            }
        }
    };

    int t = split;
    int cur = 0;
    if (t < tiles_total) { load_tile(t); store_tile(0); }
    for (; t < tiles_total; t += a.nsplit) {
        __syncthreads();
        const int tn = t + a.nsplit;
        if (tn < tiles_total) load_tile(tn);
        const unsigned char* sy = smem + cur * BUF;
        const unsigned char* sx = sy + DY_BYTES;
#pragma unroll
        for (int s = 0; s < 8; ++s) {       // k-step: the 32 pixels of tile rows 2s, 2s+1 (all offsets are immediates)
            bf16x8 af[NCF];
#pragma unroll
            for (int c = 0; c < NCF; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sy + ay[c][h] + s * 32 * 128));
                    af[c][h * 4 + 0] = v[0]; af[c][h * 4 + 1] = v[1]; af[c][h * 4 + 2] = v[2]; af[c][h * 4 + 3] = v[3];
                }
            if (BIAS && do_bias) {                     // wave-uniform
#pragma unroll
                for (int c = 0; c < (BIAS ? NCF : 1); ++c) accb[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], ones, accb[c], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < UPW; ++q) {
                if (wave + 8 * q < UNITS) {            // wave-uniform
                    bf16x8 bfr;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(sx + bx[q][h] + s * 2 * PITCH));
                        bfr[h * 4 + 0] = v[0]; bfr[h * 4 + 1] = v[1]; bfr[h * 4 + 2] = v[2]; bfr[h * 4 + 3] = v[3];
                    }
#pragma unroll
                    for (int c = 0; c < NCF; ++c)
                        acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bfr, acc[q][c], 0, 0, 0);
                }
            }
        }
        if (tn < tiles_total) store_tile(cur ^ 1);
        cur ^= 1;
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
    static bool attr_done = false;
    if (!attr_done) {
        KG_HIP(hipFuncSetAttribute((const void*)wgrad_halo_kernel<KS, CIF, NCF, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    const int n_ci = (a.cin_lim + 16 * CIF - 1) / (16 * CIF), n_co = (a.cout_lim + 63) / 64;
    hipLaunchKernelGGL((wgrad_halo_kernel<KS, CIF, NCF, BIAS>), dim3(n_ci * n_co, a.nsplit), dim3(512), smem, st, a);
    KG_CHECK_LAUNCH("wgrad_halo");
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
    KG_CHECK_ARG(!dbp || KS == 7, "kg_conv2d_wgrad_halo: the fused bias gradient is only built for 7x7");
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
    if (cout_lim <= 16) return launch_wg<3, 4, 1>(a, st);   // seg_head.2 (64 -> 1)
    return launch_wg<3, 4>(a, st);
}

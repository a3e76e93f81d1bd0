// conv_wgrad.hip -- bf16 MFMA weight-gradient convolution for gfx950.
//
// Replaces autograd's conv weight-gradient on KGnet's hot path (reference train.py:153
// loss.backward() through the nn.Conv2d layers of KGnet.py).
//   dW[co][tap][ci] = sum_pixels dY[pixel][co] * X[src(pixel, tap)][ci]
// The reduction dimension is the pixel index, which is the strided dimension of both NHWC
// operands; the [pixel][channel] tiles are staged in LDS as they lie in memory and the MFMA
// fragments are fetched with the gfx950 transpose read (ds_read_b64_tr_b16).
// One block = (64 co) x (64 ci) x one tap x one pixel split; partial sums go to a
// [split][co][tap][ci] fp32 buffer reduced (deterministically) by kg_wgrad_reduce.
#include "kg_common.h"
#include <stdlib.h>
#include <vector>
#include <string.h>

struct WgradArgs {
    const bf16_t* x; const bf16_t* dy; float* dwp; const int2* rowdesc;
    int M, H, W, OH, OW, ldx, lddy;
    int Cin, Cout;          // real channel counts (output bounds)
    int cin_lim, cout_lim;  // readable channels (multiples of 8) in x / dy rows
    int ntaps, KW, stride_log2, pad, dil, mode;
    int chunks_per_split;   // 64-pixel chunks per z-split
    long split_stride;      // elements between partial buffers
    WgPairs wp;             // split-bf16 operand planes (kg_common.h): the chunk index runs over wp.n * chunks_per_plane virtual chunks
    int chunks_per_plane;
    int direct;             // 1x1 stride-1 dense conv: input row of output pixel m is m itself (no (n, oy, ox) decomposition: two integer
                            // divisions per staged row were more VALU work than the chunk's 32 MFMAs per wave)
};

__device__ __forceinline__ int tr_f(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); }

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <bool USE_TR>
__device__ __forceinline__ bf16x8 load_tr_frag(const unsigned char* tile, int p0, int cf0, int lane) {
    // fragment for an MFMA operand: "row" = channel cf0*16 + (lane&15), k = pixel p0 + (lane>>4)*8 + j
    const int i = lane & 15, G = lane >> 4;
    bf16x8 out;
    if (USE_TR) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int r = p0 + G * 8 + h * 4 + (i >> 2);
            int off = r * 128 + ((cf0 ^ tr_f(r)) * 32) + (i & 3) * 8;
            bf16x4 v = KG_DS_READ_TR16((lds_bf16x4*)(tile + off));
            out[h * 4 + 0] = v[0]; out[h * 4 + 1] = v[1]; out[h * 4 + 2] = v[2]; out[h * 4 + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int r = p0 + G * 8 + j;
            int off = r * 128 + ((cf0 ^ tr_f(r)) * 32) + i * 2;
            out[j] = *reinterpret_cast<const kg_h16*>(tile + off);
        }
    }
    return out;
}

// PX = pixels per staged chunk (one barrier per chunk).  128 halves the barriers but also the resident workgroups (64 KB of
// LDS): measured 20-40 % slower than 64 on MI355X.
template <bool USE_TR, int PX = 64>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int NI = PX / 32, TB = PX * 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TB];  // [buf][dy | x][PX px][128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave >> 1, wci = wave & 1;
    const int n_ci_tiles = (a.cin_lim + 63) / 64;
    const int ci0 = (blockIdx.x % n_ci_tiles) * 64, co0 = (blockIdx.x / n_ci_tiles) * 64;
    const int tap = blockIdx.y, split = blockIdx.z;
    const int tdy = tap / a.KW, tdx = tap - tdy * a.KW;
    const int d_y = tdy * a.dil - a.pad, d_x = tdx * a.dil - a.pad;

    const int c8 = tid & 7;       // 16-byte channel chunk within the 64-channel tile
    const int prow = tid >> 3;    // + 32*i
    const bool x_c_ok = ci0 + c8 * 8 < a.cin_lim;
    const bool y_c_ok = co0 + c8 * 8 < a.cout_lim;
    const int ohw = a.OH * a.OW;

    uint4 xr[NI], yr[NI];
    auto load_chunk = [&](int vchunk) {
        const int pr = vchunk / a.chunks_per_plane, chunk = vchunk - pr * a.chunks_per_plane;   // (product, chunk): uniform
        const bf16_t* xpl = a.x + a.wp.xoff[pr];
        const bf16_t* dpl = a.dy + a.wp.doff[pr];
        const int mbase = chunk * PX;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = mbase + prow + 32 * i;
            uint4 xv = make_uint4(0, 0, 0, 0), yv = make_uint4(0, 0, 0, 0);
            if (m < a.M) {
                if (y_c_ok) yv = *reinterpret_cast<const uint4*>(dpl + (long)m * a.lddy + co0 + c8 * 8);
                if (x_c_ok) {
                    long row; bool ok;
                    if (a.direct) {
                        ok = true; row = m;
                    } else if (a.mode >= 2) {
                        int2 d = a.rowdesc[m];
                        int y = (d.x >> 16) + d_y, x = (d.x & 0xffff) + d_x, h = d.y >> 16, w = d.y & 0xffff;
                        ok = (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w;
                        row = (long)m + (long)d_y * w + d_x;
                    } else {
                        int n = m / ohw, rem = m - n * ohw;
                        int oy = rem / a.OW, ox = rem - oy * a.OW;
                        int iy = (oy << a.stride_log2) + d_y, ix = (ox << a.stride_log2) + d_x;
                        ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        row = ((long)n * a.H + iy) * a.W + ix;
                    }
                    if (ok) xv = *reinterpret_cast<const uint4*>(xpl + row * a.ldx + ci0 + c8 * 8);
                }
            }
            xr[i] = xv; yr[i] = yv;
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* sy = smem + buf * 2 * TB;
        unsigned char* sx = sy + TB;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int r = prow + 32 * i;
            int off = r * 128 + (((c8 >> 1) ^ tr_f(r)) * 32) + (c8 & 1) * 16;
            *reinterpret_cast<uint4*>(sy + off) = yr[i];
            *reinterpret_cast<uint4*>(sx + off) = xr[i];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int total_chunks = a.wp.n * a.chunks_per_plane;
    const int cbeg = split * a.chunks_per_split;
    int cend = cbeg + a.chunks_per_split;
    if (cend > total_chunks) cend = total_chunks;

    if (cbeg < cend) load_chunk(cbeg);
    for (int ch = cbeg; ch < cend; ++ch) {
        const int buf = (ch - cbeg) & 1;
        store_chunk(buf);
        __syncthreads();
        if (ch + 1 < cend) load_chunk(ch + 1);
        const unsigned char* sy = smem + buf * 2 * TB;
        const unsigned char* sx = sy + TB;
#pragma unroll
        for (int s = 0; s < PX / 32; ++s) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = load_tr_frag<USE_TR>(sy, s * 32, wco * 2 + i, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = load_tr_frag<USE_TR>(sx, s * 32, wci * 2 + j, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = KG_MFMA16(af[i], bfr[j], acc[i][j]);
        }
    }

    float* out = a.dwp + (long)split * a.split_stride;
    const int lm = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + (wci * 2 + j) * 16 + lm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wco * 2 + i) * 16 + g * 4 + r;
                if (co < a.Cout && ci < a.Cin) out[((long)co * a.ntaps + tap) * a.Cin + ci] = acc[i][j][r];
            }
        }
}

// ---- 128 x 128 output tile variant (cin, cout >= 128: the ragged deep seg levels, the wide 1x1 convs).  With 64 x 64 tiles a
// 256 x 512 x 9-tap gradient re-reads dY and X 288 times through L2 (7.6 GB per launch, L2-bandwidth bound at 0.25 PFLOP/s);
// 128 x 128 tiles quarter that traffic and quadruple the MFMAs per barrier (32 per wave and 64-pixel chunk).
// LDS rows are 256 B (128 channels) = 8 units of 32 B; unit XOR f8(r) = (r & 3) | (((r >> 3) & 1) << 2) keeps the 4 rows of a
// 16-lane transpose-read group and the two groups of a 32-lane LDS cycle on distinct units.
__device__ __forceinline__ int tr_f8(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
__device__ __forceinline__ bf16x8 load_tr_frag128(const unsigned char* tile, int p0, int cf, int lane) {
    const int i = lane & 15, G = lane >> 4;
    bf16x8 out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = p0 + G * 8 + h * 4 + (i >> 2);
        const int off = r * 256 + ((cf ^ tr_f8(r)) * 32) + (i & 3) * 8;
        bf16x4 v = KG_DS_READ_TR16((lds_bf16x4*)(tile + off));
        out[h * 4 + 0] = v[0]; out[h * 4 + 1] = v[1]; out[h * 4 + 2] = v[2]; out[h * 4 + 3] = v[3];
    }
    return out;
}

__global__ __launch_bounds__(256) void conv_wgrad128_kernel(const WgradArgs a) {
    constexpr int TB = 64 * 256;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TB];   // [buf][dy | x][64 px][256 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave >> 1, wci = wave & 1;
    const int n_ci_tiles = (a.cin_lim + 127) / 128;
    const int ci0 = (blockIdx.x % n_ci_tiles) * 128, co0 = (blockIdx.x / n_ci_tiles) * 128;
    const int tap = blockIdx.y, split = blockIdx.z;
    const int tdy = tap / a.KW, tdx = tap - tdy * a.KW;
    const int d_y = tdy * a.dil - a.pad, d_x = tdx * a.dil - a.pad;
    const int c16 = tid & 15, prow = tid >> 4;      // 16-byte chunk within the 128-channel tile; row prow + 16 i
    const bool x_c_ok = ci0 + c16 * 8 < a.cin_lim, y_c_ok = co0 + c16 * 8 < a.cout_lim;
    const int ohw = a.OH * a.OW;

    uint4 xr[4], yr[4];
    auto load_chunk = [&](int vchunk) {
        const int pr = vchunk / a.chunks_per_plane, chunk = vchunk - pr * a.chunks_per_plane;   // (product, chunk): uniform
        const bf16_t* xpl = a.x + a.wp.xoff[pr];
        const bf16_t* dpl = a.dy + a.wp.doff[pr];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = chunk * 64 + prow + 16 * i;
            uint4 xv = make_uint4(0, 0, 0, 0), yv = make_uint4(0, 0, 0, 0);
            if (m < a.M) {
                if (y_c_ok) yv = *reinterpret_cast<const uint4*>(dpl + (long)m * a.lddy + co0 + c16 * 8);
                if (x_c_ok) {
                    long row; bool ok;
                    if (a.direct) {
                        ok = true; row = m;
                    } else if (a.mode >= 2) {
                        const int2 d = a.rowdesc[m];
                        const int y = (d.x >> 16) + d_y, x = (d.x & 0xffff) + d_x, h = d.y >> 16, w = d.y & 0xffff;
                        ok = (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w;
                        row = (long)m + (long)d_y * w + d_x;
                    } else {
                        const int n = m / ohw, rem = m - n * ohw;
                        const int oy = rem / a.OW, ox = rem - oy * a.OW;
                        const int iy = (oy << a.stride_log2) + d_y, ix = (ox << a.stride_log2) + d_x;
                        ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        row = ((long)n * a.H + iy) * a.W + ix;
                    }
                    if (ok) xv = *reinterpret_cast<const uint4*>(xpl + row * a.ldx + ci0 + c16 * 8);
                }
            }
            xr[i] = xv; yr[i] = yv;
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* sy = smem + buf * 2 * TB;
        unsigned char* sx = sy + TB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = prow + 16 * i;
            const int off = r * 256 + (((c16 >> 1) ^ tr_f8(r)) * 32) + (c16 & 1) * 16;
            *reinterpret_cast<uint4*>(sy + off) = yr[i];
            *reinterpret_cast<uint4*>(sx + off) = xr[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int total_chunks = a.wp.n * a.chunks_per_plane;
    const int cbeg = split * a.chunks_per_split;
    int cend = cbeg + a.chunks_per_split;
    if (cend > total_chunks) cend = total_chunks;
    if (cbeg < cend) load_chunk(cbeg);
    for (int ch = cbeg; ch < cend; ++ch) {
        const int buf = (ch - cbeg) & 1;
        store_chunk(buf);
        __syncthreads();
        if (ch + 1 < cend) load_chunk(ch + 1);
        const unsigned char* sy = smem + buf * 2 * TB;
        const unsigned char* sx = sy + TB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = load_tr_frag128(sy, s * 32, wco * 4 + i, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = load_tr_frag128(sx, s * 32, wci * 4 + j, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = KG_MFMA16(af[i], bfr[j], acc[i][j]);
        }
    }
    float* out = a.dwp + (long)split * a.split_stride;
    const int lm = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = ci0 + (wci * 4 + j) * 16 + lm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wco * 4 + i) * 16 + g * 4 + r;
                if (co < a.Cout && ci < a.Cin) out[((long)co * a.ntaps + tap) * a.Cin + ci] = acc[i][j][r];
            }
        }
}

// ---- 256 x 128 output tile, LDS-direct ring (cout >= 256, cin >= 128) ------------------------------------------------------------------
// conv_wgrad128_kernel keeps ONE 64-pixel chunk of global loads in flight (registers, one __syncthreads per chunk): a chunk is 32 MFMAs
// per wave = 0.25 us against a load latency of 1-2 us, so the ragged 3x3 levels of the seg branch ran at 0.30 PFLOP/s (MFMA pipes 10 % busy).
// Here, as in conv_gather.hip: 8 waves, three 48 KB stages ([2][64 px][128 couts] of dY + [64 px][128 cins] of X), two of them in flight by
// LDS-direct loads behind a counted s_waitcnt vmcnt(6); the 32-byte-unit swizzle of the transpose reads is applied to the source chunk;
// transpose reads from inline asm, k-step 1's behind the MFMAs of k-step 0.  Ragged rows (mode 2) need rowdesc[m] to place the tap's input
// row: it is fetched ONE stage ahead of the loads that use it and issued BEFORE the stage in between, so the counted wait covers it.
__device__ uint4 kg_wgr_zero_line[16];
template <int OFF>
__device__ __forceinline__ void wgr_rd_tr64(bf16x4& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// NH = 1 (cout < 256): 128 x 128 tile, one dY tile per stage (32 KB stages, four loads per thread), wave tile 32 couts x 64 cins.
template <int MODE, int NH>   // MODE 0: 1x1 stride-1 dense (input row = output row), 1: dense with taps / stride, 2: ragged rows (rowdesc)
__global__ __launch_bounds__(512) void conv_wgrad_ring_kernel(const WgradArgs a) {
    constexpr int NS = 3, HB = 64 * 256, STAGE = (NH + 1) * HB, NF = 2 * NH;     // NF: 16-cout fragments of a wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wci = wave & 1;
    const int n_ci_tiles = (a.cin_lim + 127) / 128;
    const int ci0 = (blockIdx.x % n_ci_tiles) * 128, co0 = (blockIdx.x / n_ci_tiles) * (128 * NH);
    const int tap = blockIdx.y, split = blockIdx.z;
    const int tdy = tap / a.KW, tdx = tap - tdy * a.KW;
    const int d_y = tdy * a.dil - a.pad, d_x = tdx * a.dil - a.pad;
    const int ohw = a.OH * a.OW;

    // staging: thread -> rows r0, r0 + 32 of each of the three 64-row tiles, LDS slot s16 = source chunk c16 under the unit swizzle
    const int r0 = tid >> 4, s16 = tid & 15;
    const int c16 = (((s16 >> 1) ^ tr_f8(r0)) << 1) | (s16 & 1);          // (tr_f8(r0 + 32) == tr_f8(r0))
    const bool y_ok0 = co0 + c16 * 8 < a.cout_lim, y_ok1 = NH == 2 && co0 + 128 + c16 * 8 < a.cout_lim, x_ok = ci0 + c16 * 8 < a.cin_lim;
    const bf16_t* zline = reinterpret_cast<const bf16_t*>(kg_wgr_zero_line);
    const int total_chunks = a.wp.n * a.chunks_per_plane;
    const int cbeg = split * a.chunks_per_split;
    int cend = cbeg + a.chunks_per_split;
    if (cend > total_chunks) cend = total_chunks;
    const int nst = cend - cbeg;
    // input row of output pixel m for this tap (-1: padding / outside); mode 2 reads rowdesc (fetched a stage ahead, see below)
    auto row_of = [&](int m, int2 d) -> long {
        if (m >= a.M) return -1;
        if (MODE == 0) return m;
        if (MODE == 2) {
            const int y = (d.x >> 16) + d_y, x = (d.x & 0xffff) + d_x, h = d.y >> 16, w = d.y & 0xffff;
            return ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) ? (long)m + (long)d_y * w + d_x : -1;
        }
        const int n = m / ohw, rem = m - n * ohw;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        const int iy = (oy << a.stride_log2) + d_y, ix = (ox << a.stride_log2) + d_x;
        return ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) ? ((long)n * a.H + iy) * a.W + ix : -1;
    };
    // Ragged rows: the rowdesc entries of a stage's 64 pixels ride into LDS with the loads of the stage TWO before it (wave 0, two more
    // LDS-direct loads of 256 bytes), so the thread that places stage s + 2 reads them from LDS when stage s has landed -- a register load
    // of rowdesc inside the loop makes hipcc wait vmcnt(0) before its first use, i.e. for every stage in flight.
    constexpr bool ragged = MODE == 2;
    unsigned char* rdl = smem + NS * STAGE;                       // [NS][64] int2
    auto rd_global = [&](int vchunk, int2 (&rd)[2]) {             // prologue only: nothing is in flight yet
        const int pr = vchunk / a.chunks_per_plane, chunk = vchunk - pr * a.chunks_per_plane;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = chunk * 64 + r0 + 32 * k;
            rd[k] = (ragged && vchunk < cend && m < a.M) ? a.rowdesc[m] : make_int2(0, 0);
        }
    };
    auto rd_lds = [&](int slot, int2 (&rd)[2]) {
        if (!ragged) { rd[0] = rd[1] = make_int2(0, 0); return; }
        const unsigned ad = lds_addr(rdl) + slot * 512 + r0 * 8;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)" : "=&v"(rd[0]), "=&v"(rd[1]) : "v"(ad) : "memory");
    };
    int i_slot = 0;
    auto issue = [&](int vchunk, const int2 (&rd)[2]) {        // the six LDS-direct loads of a stage (+ wave 0, ragged: the rowdesc of stage vchunk + 2)
        const int pr = vchunk / a.chunks_per_plane, chunk = vchunk - pr * a.chunks_per_plane;   // (product, chunk): uniform
        const bf16_t* xpl = a.x + a.wp.xoff[pr] + ci0 + c16 * 8;
        const bf16_t* dpl = a.dy + a.wp.doff[pr] + co0 + c16 * 8;
        unsigned char* st = smem + i_slot * STAGE + wave * 1024;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = chunk * 64 + r0 + 32 * k;
            const bool in = m < a.M;
            const bf16_t* s0 = (in && y_ok0) ? dpl + (long)m * a.lddy : zline;
            const bf16_t* s1 = (in && y_ok1) ? dpl + (long)m * a.lddy + 128 : zline;
            (void)s1;
            const long row = row_of(m, rd[k]);
            const bf16_t* s2 = (row >= 0 && x_ok) ? xpl + row * a.ldx : zline;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s0, (__attribute__((address_space(3))) void*)(st + k * 8192), 16, 0, 0);
            if constexpr (NH == 2)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s1, (__attribute__((address_space(3))) void*)(st + HB + k * 8192), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s2, (__attribute__((address_space(3))) void*)(st + NH * HB + k * 8192), 16, 0, 0);
        }
        if (ragged && wave == 0) {
            const int v2 = vchunk + 2;
            const int pr2 = v2 / a.chunks_per_plane, chunk2 = v2 - pr2 * a.chunks_per_plane;
            const int* rdw = reinterpret_cast<const int*>(a.rowdesc);
#pragma unroll
            for (int k = 0; k < 2; ++k) {                         // 128 dwords = two 4-byte loads per lane (exact bounds: no read past entry M - 1)
                const long w = (long)chunk2 * 128 + k * 64 + lane;
                const void* src = (v2 < cend && w < 2L * a.M) ? (const void*)(rdw + w) : (const void*)zline;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(rdl + i_slot * 512 + k * 256), 4, 0, 0);
            }
        }
        i_slot = i_slot == NS - 1 ? 0 : i_slot + 1;
    };

    // transpose-read addresses: fragment f of a 128-channel tile = 16 channels cf = base + f; lane (i, G): row p0 + 8 G + 4 h + (i >> 2)
    const unsigned lds0 = lds_addr(smem);
    unsigned fa[NF], fb[4];
    {
        const int i = lane & 15, G = lane >> 4;
        const int r = G * 8 + (i >> 2);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f < NF) fa[f] = lds0 + (NH == 2 ? (wco >> 1) * HB : 0) + r * 256 + (((NH == 2 ? (wco & 1) * 4 + f : wco * 2 + f) ^ tr_f8(r)) * 32) + (i & 3) * 8;
            fb[f] = lds0 + NH * HB + r * 256 + (((wci * 4 + f) ^ tr_f8(r)) * 32) + (i & 3) * 8;
        }
    }
    f32x4 acc[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool w8 = ragged && wave == 0;                          // (uniform) this wave issues 8 loads per stage
    if (nst > 0) {
        int2 rd0[2], rd1[2];
        rd_global(cbeg, rd0); rd_global(cbeg + 1, rd1);
        issue(cbeg, rd0);
        if (nst > 1) issue(cbeg + 1, rd1);
    }
    int c_slot = 0;
    for (int s = 0; s < nst; ++s) {
        if (s + 1 < nst) {          // stage s (and the rowdesc of stage s + 2 that rode with it) has landed; stage s + 1 may stay in flight
            if (w8) { if constexpr (NH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else { if constexpr (NH == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 2 < nst) {
            int2 rd[2];
            rd_lds(c_slot, rd);
            issue(cbeg + s + 2, rd);
        }
        const unsigned sb = c_slot * STAGE;
        bf16x4 av[2][NF][2], bv[2][4][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (k == 0) { wgr_rd_tr64<0>(av[0][f][0], fa[f] + sb); wgr_rd_tr64<1024>(av[0][f][1], fa[f] + sb); }
                else { wgr_rd_tr64<8192>(av[1][f][0], fa[f] + sb); wgr_rd_tr64<8192 + 1024>(av[1][f][1], fa[f] + sb); }
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (k == 0) { wgr_rd_tr64<0>(bv[0][f][0], fb[f] + sb); wgr_rd_tr64<1024>(bv[0][f][1], fb[f] + sb); }
                else { wgr_rd_tr64<8192>(bv[1][f][0], fb[f] + sb); wgr_rd_tr64<8192 + 1024>(bv[1][f][1], fb[f] + sb); }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 0) { if constexpr (NH == 2) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); }   // (the reads of k-step 1 -- 16 / 12 -- may stay in flight; the counter saturates at 15)
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bf16x8 af[NF], bf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                asm volatile("" : "+v"(bv[k][f][0]), "+v"(bv[k][f][1]));
                bf[f] = __builtin_shufflevector(bv[k][f][0], bv[k][f][1], 0, 1, 2, 3, 4, 5, 6, 7);
                if (f < NF) {
                    asm volatile("" : "+v"(av[k][f][0]), "+v"(av[k][f][1]));
                    af[f] = __builtin_shufflevector(av[k][f][0], av[k][f][1], 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = KG_MFMA16(af[i], bf[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        c_slot = c_slot == NS - 1 ? 0 : c_slot + 1;
    }
    float* out = a.dwp + (long)split * a.split_stride;
    const int lm = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = ci0 + (wci * 4 + j) * 16 + lm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wco * NF + i) * 16 + g * 4 + r;
                if (co < a.Cout && ci < a.Cin) out[((long)co * a.ntaps + tap) * a.Cin + ci] = acc[i][j][r];
            }
        }
}

static int g_wgrad_use_tr = 1;
extern "C" int kg_set_wgrad_tr(int use_tr) { g_wgrad_use_tr = use_tr; return KG_OK; }

extern "C" int kg_conv2d_wgrad(const void* x, const void* dy, float* dwp, const int* rowdesc, int M, int H, int W,
                               int OH, int OW, int ldx, int lddy, int Cin, int Cout, int cin_lim, int cout_lim,
                               int KH, int KW, int stride, int pad, int dil, int mode, int nsplit,
                               long split_stride, const kg_planes_t* planes, void* stream) {
    // planes: a = x, b = dy: every kept plane product x_i * dY_j is one more pass over the pixels inside the same launch
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    const kg_planes_t pp = kg_planes_or_default(planes);
    KG_CHECK_ARG(kg_planes_ok(pp), "kg_conv2d_wgrad: bad kg_planes_t");
    a.wp = kg_make_wgpairs(pp.a_planes, pp.a_pstride, pp.b_planes, pp.b_pstride);
    KG_CHECK_ARG(x && dy && dwp, "kg_conv2d_wgrad: null pointer");
    KG_CHECK_ARG(ldx % 8 == 0 && lddy % 8 == 0 && cin_lim % 8 == 0 && cout_lim % 8 == 0, "kg_conv2d_wgrad: ld/lim must be multiples of 8");
    KG_CHECK_ARG(stride == 1 || stride == 2, "kg_conv2d_wgrad: stride must be 1 or 2");
    KG_CHECK_ARG(mode == 0 || (mode == 2 && rowdesc && stride == 1), "kg_conv2d_wgrad: bad mode");
    KG_CHECK_ARG(nsplit >= 1 && M > 0, "kg_conv2d_wgrad: bad split/M");
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dwp = dwp; a.rowdesc = (const int2*)rowdesc;
    a.M = M; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.ldx = ldx; a.lddy = lddy; a.Cin = Cin; a.Cout = Cout;
    a.cin_lim = cin_lim; a.cout_lim = cout_lim; a.ntaps = KH * KW; a.KW = KW; a.stride_log2 = stride == 2 ? 1 : 0;
    a.pad = pad; a.dil = dil; a.mode = mode;
    a.direct = mode == 0 && KH == 1 && KW == 1 && stride == 1 && pad == 0 && OH == H && OW == W;
    a.chunks_per_plane = (M + 63) / 64;      // chunks of the kernel's PX = 64 pixels
    const int total_chunks = a.wp.n * a.chunks_per_plane;
    a.chunks_per_split = (total_chunks + nsplit - 1) / nsplit;
    a.split_stride = split_stride;
    constexpr int use128 = 1;
    constexpr int use_ring = 1;
    if (use_ring && g_wgrad_use_tr && cin_lim >= 128 && cout_lim >= 64) {   // (ops.wgrad_splits sizes nsplit for these tiles under the same condition)
        const int nh = cout_lim >= 256 ? 2 : 1;
        const int smem = 3 * (nh + 1) * 64 * 256 + 3 * 512;
        static KgPerDevice attr_done;
        if (attr_done.first()) {
            constexpr int smem2 = 3 * 3 * 64 * 256 + 3 * 512, smem1 = 3 * 2 * 64 * 256 + 3 * 512;
            KG_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2));
            KG_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2));
            KG_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2));
            KG_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem1));
            KG_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem1));
            KG_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ring_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem1));
        }
        dim3 gridr(((cin_lim + 127) / 128) * ((cout_lim + 128 * nh - 1) / (128 * nh)), KH * KW, nsplit);
        const int md = a.mode >= 2 ? 2 : (a.direct ? 0 : 1);
#define KG_RING_LAUNCH(M_, H_) hipLaunchKernelGGL((conv_wgrad_ring_kernel<M_, H_>), gridr, dim3(512), smem, (hipStream_t)stream, a)
        if (nh == 2) { if (md == 2) KG_RING_LAUNCH(2, 2); else if (md == 0) KG_RING_LAUNCH(0, 2); else KG_RING_LAUNCH(1, 2); }
        else { if (md == 2) KG_RING_LAUNCH(2, 1); else if (md == 0) KG_RING_LAUNCH(0, 1); else KG_RING_LAUNCH(1, 1); }
#undef KG_RING_LAUNCH
        KG_CHECK_LAUNCH("conv_wgrad_ring");
        static const char* const names[2][3] = {{"conv_wgrad_ring_kernel<0, 1>", "conv_wgrad_ring_kernel<1, 1>", "conv_wgrad_ring_kernel<2, 1>"},
                                                {"conv_wgrad_ring_kernel<0, 2>", "conv_wgrad_ring_kernel<1, 2>", "conv_wgrad_ring_kernel<2, 2>"}};
        kg_note_kernel(names[nh - 1][md]);
        return KG_OK;
    }
    if (use128 && g_wgrad_use_tr && cin_lim >= 128 && cout_lim >= 128) {
        dim3 grid128(((cin_lim + 127) / 128) * ((cout_lim + 127) / 128), KH * KW, nsplit);
        hipLaunchKernelGGL(conv_wgrad128_kernel, grid128, dim3(256), 0, (hipStream_t)stream, a);
        KG_CHECK_LAUNCH("conv_wgrad128");
        kg_note_kernel("conv_wgrad128_kernel");
        return KG_OK;
    }
    dim3 grid(((cin_lim + 63) / 64) * ((cout_lim + 63) / 64), KH * KW, nsplit);
    if (g_wgrad_use_tr)
        hipLaunchKernelGGL(conv_wgrad_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(conv_wgrad_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    KG_CHECK_LAUNCH("conv_wgrad");
    kg_note_kernel("conv_wgrad_kernel");
    return KG_OK;
}

// Sum the split partials [S][Cout][taps][Cin] and write the OIHW fp32 gradient [Cout][Cin][taps]
// (accumulate != 0: grad += sum).  Fixed summation order => bitwise reproducible gradients.  One block per
// (co, 64-channel ci chunk): coalesced reads along ci, transpose through LDS, coalesced writes along (ci,tap).
struct ReduceDst { float* g[4]; int end[4]; const float* bias_part; float* db; int bias_C; };   // bias_part != null: one more block row sums the bias partials [S][bias_C]   // output tensor k holds the rows [end[k-1], end[k]) of the fused conv (heads sharing an input)
// G = blockDim.x / (taps * CH) >= 2 (small layers with many splits, e.g. 3x3 64 -> 64 with 256 splits): G thread groups each sum a
// contiguous range of the splits in order, then group 0 adds the G partial sums in group order -- still one fixed summation order.
// (bx, by, NT: the block coordinates and thread count of the job this block works on -- the launch's own for the single-job kernel, the
// job's for the batched one, whose blocks may be larger: threads >= NT only take part in the barriers)
template <int CH>   // ci chunk per block: 64 for big layers, 16 to get enough blocks on small ones
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, const ReduceDst& dst4, int Cout, int Cin, int taps, int S,
                                                  long split_stride, int accumulate, int bx, int by, int NT, float* tile, float* red) {
    if (bx >= Cout) {   // the bias-gradient partials of the same conv (kg_conv2d_wgrad_halo's all-ones unit): one wave per channel,
        if (by != 0) return; // lanes stride over the splits, as bias_grad_final_kernel sums them (same order, same bits)
        const int lane = threadIdx.x & 63, nw = NT >> 6;     // (whole waves only: NT need not be a multiple of 64)
        const int c = (bx - Cout) * nw + (int)(threadIdx.x >> 6);
        if ((int)(threadIdx.x >> 6) >= nw || c >= dst4.bias_C) return;
        float sb = 0.f;
        for (int b = lane; b < S; b += 64) sb += dst4.bias_part[(long)b * dst4.bias_C + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sb += __shfl_down(sb, o, 64);
        if (lane == 0) dst4.db[c] = accumulate ? dst4.db[c] + sb : sb;
        return;
    }
    const int co = bx, ci0 = by * CH;
    int k = 0, row0 = 0;
    while (k < 3 && co >= dst4.end[k]) { row0 = dst4.end[k]; ++k; }
    float* __restrict__ grad = dst4.g[k] - (long)row0 * Cin * taps;   // so that row `co` of the fused conv lands in row co - row0
    const int nci = Cin - ci0 < CH ? Cin - ci0 : CH;
    const int E = taps * CH;
    const int G = NT >= 2 * E ? NT / E : 1;
    auto sum_range = [&](const float* src, int k0, int k1) {
        float s = 0.f;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {   // 8 independent loads in flight; the adds keep the fixed k order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + u) * split_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < k1; ++k) s += src[(long)k * split_stride];
        return s;
    };
    if (G > 1) {
        const int g = threadIdx.x / E, e = threadIdx.x - g * E;
        const int tap = e / CH, ci = e - tap * CH;
        const int Sg = (S + G - 1) / G;
        if (g < G) {
            float s = 0.f;
            const int k0 = g * Sg, k1 = k0 + Sg < S ? k0 + Sg : S;
            if (ci < nci && k0 < k1) s = sum_range(part + ((long)co * taps + tap) * Cin + ci0 + ci, k0, k1);
            red[g * E + e] = s;
        }
        __syncthreads();
        if (g == 0) {
            float s = red[e];
            for (int q = 1; q < G; ++q) s += red[q * E + e];
            tile[tap * (CH + 1) + ci] = s;
        }
    } else {
        for (int e = threadIdx.x; e < E && (int)threadIdx.x < NT; e += NT) {
            const int tap = e / CH, ci = e - tap * CH;
            float s = 0.f;
            if (ci < nci) s = sum_range(part + ((long)co * taps + tap) * Cin + ci0 + ci, 0, S);
            tile[tap * (CH + 1) + ci] = s;
        }
    }
    __syncthreads();
    float* dst = grad + ((long)co * Cin + ci0) * taps;
    for (int j = threadIdx.x; j < nci * taps && (int)threadIdx.x < NT; j += NT) {
        const int ci = j / taps, tap = j - ci * taps;
        const float v = tile[tap * (CH + 1) + ci];
        dst[j] = accumulate ? dst[j] + v : v;
    }
}
template <int CH>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, const ReduceDst dst4,
                                                            int Cout, int Cin, int taps, int S, long split_stride,
                                                            int accumulate) {
    __shared__ float tile[49 * (CH + 1)];
    __shared__ float red[1024];
    wgrad_reduce_body<CH>(part, dst4, Cout, Cin, taps, S, split_stride, accumulate, (int)blockIdx.x, (int)blockIdx.y, (int)blockDim.x, tile, red);
}
// Batched form: the reductions of up to REDUCE_BATCH convs in one launch (a backward pass of the bench step issues ~80 of them, most a few
// microseconds long).  A block belongs to the job whose block range holds blockIdx.x and runs that job's single-launch decomposition --
// same thread count, same groups, same order: the same bits.
constexpr int REDUCE_BATCH = 12;
struct ReduceJob { const float* part; ReduceDst d; long split_stride; int Cout, Cin, taps, S, accumulate, nt, gx, blk0; };
struct ReduceBatch { ReduceJob j[REDUCE_BATCH]; int n; };
template <int CH>
__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(const ReduceBatch b) {
    __shared__ float tile[49 * (CH + 1)];
    __shared__ float red[1024];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.j[k + 1].blk0) ++k;
    const ReduceJob& q = b.j[k];
    const int local = (int)blockIdx.x - q.blk0;
    wgrad_reduce_body<CH>(q.part, q.d, q.Cout, q.Cin, q.taps, q.S, q.split_stride, q.accumulate, local % q.gx, local / q.gx, q.nt, tile, red);
}

// the single-launch decomposition of one reduction: CH, threads per block, grid
struct ReduceShape { int ch, nt, gx, gy; };
static ReduceShape reduce_shape(const ReduceDst& d, int Cout, int Cin, int taps, int nsplit) {
    ReduceShape r;
    // (bias partials: extra block columns after the Cout weight rows, one wave per channel)
    if ((long)Cout * ((Cin + 63) / 64) >= 2048) {
        r.ch = 64; r.nt = 256;
        r.gx = Cout + (d.bias_part ? (d.bias_C + 3) / 4 : 0); r.gy = (Cin + 63) / 64;
    } else {
        int nt = 256;
        const int E = taps * 16;
        if (2 * E <= 1024 && nsplit >= 16) {       // few elements, many splits: several thread groups share the splits
            int G = 1024 / E;
            if (G > nsplit / 8) G = nsplit / 8;
            if (G >= 2) nt = E * G;
        }
        r.ch = 16; r.nt = nt;
        r.gx = Cout + (d.bias_part ? (d.bias_C + nt / 64 - 1) / (nt / 64) : 0); r.gy = (Cin + 15) / 16;
    }
    return r;
}
// Deferred reductions (kg_wgrad_reduce_defer(1) ... kg_wgrad_reduce_flush): while deferral is on, the reduce entry points only record
// their job (the caller keeps the partial buffers alive and untouched until the flush); the flush launches them REDUCE_BATCH at a time.
static thread_local int g_reduce_defer = 0;
static thread_local std::vector<ReduceJob> g_reduce_q[2];     // [0]: CH = 16 jobs, [1]: CH = 64 jobs
static int launch_reduce(const float* part, const ReduceDst& d, int Cout, int Cin, int taps, int nsplit, long split_stride, int accumulate,
                         hipStream_t st) {
    const ReduceShape r = reduce_shape(d, Cout, Cin, taps, nsplit);
    if (g_reduce_defer) {
        ReduceJob q;
        q.part = part; q.d = d; q.split_stride = split_stride; q.Cout = Cout; q.Cin = Cin; q.taps = taps; q.S = nsplit; q.accumulate = accumulate;
        q.nt = r.nt; q.gx = r.gx; q.blk0 = r.gx * r.gy;     // (blk0 holds the job's block count until the flush lays the jobs out)
        g_reduce_q[r.ch == 64].push_back(q);
        return KG_OK;
    }
    if (r.ch == 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3(r.gx, r.gy), dim3(r.nt), 0, st, part, d, Cout, Cin, taps, nsplit, split_stride, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(r.gx, r.gy), dim3(r.nt), 0, st, part, d, Cout, Cin, taps, nsplit, split_stride, accumulate);
    KG_CHECK_LAUNCH("wgrad_reduce");
    return KG_OK;
}
extern "C" int kg_wgrad_reduce_defer(int on) {
    g_reduce_defer = on;
    if (!on) { g_reduce_q[0].clear(); g_reduce_q[1].clear(); }     // (empty after a successful flush; after a failed one nothing stale may stay)
    return KG_OK;
}
extern "C" int kg_wgrad_reduce_pending(void) { return (int)(g_reduce_q[0].size() + g_reduce_q[1].size()); }
extern "C" int kg_wgrad_reduce_flush(void* stream) {
    for (int c = 0; c < 2; ++c) {
        std::vector<ReduceJob>& q = g_reduce_q[c];
        for (size_t i0 = 0; i0 < q.size(); i0 += REDUCE_BATCH) {
            ReduceBatch b;
            memset(&b, 0, sizeof(b));
            int blocks = 0, nt = 64;
            b.n = (int)(q.size() - i0 < (size_t)REDUCE_BATCH ? q.size() - i0 : REDUCE_BATCH);
            for (int k = 0; k < b.n; ++k) {
                b.j[k] = q[i0 + k];
                const int nb = b.j[k].blk0;
                b.j[k].blk0 = blocks; blocks += nb;
                if (b.j[k].nt > nt) nt = b.j[k].nt;
            }
            nt = (nt + 63) / 64 * 64;
            if (c) hipLaunchKernelGGL(wgrad_reduce_batch_kernel<64>, dim3(blocks), dim3(nt), 0, (hipStream_t)stream, b);
            else hipLaunchKernelGGL(wgrad_reduce_batch_kernel<16>, dim3(blocks), dim3(nt), 0, (hipStream_t)stream, b);
            if (hipGetLastError() != hipSuccess) {      // drop EVERY recorded job: none may outlive the tensors its pointers name
                g_reduce_q[0].clear(); g_reduce_q[1].clear();
                kg_set_error("kg_wgrad_reduce_flush: launch failed"); return KG_ERR_HIP;
            }
        }
        q.clear();
    }
    return KG_OK;
}

extern "C" int kg_wgrad_reduce(const float* part, float* grad, int Cout, int Cin, int KH, int KW, int nsplit,
                               long split_stride, int accumulate, void* stream) {
    KG_CHECK_ARG(part && grad, "kg_wgrad_reduce: null pointer");
    KG_CHECK_ARG(KH * KW <= 49, "kg_wgrad_reduce: at most 49 taps");
    ReduceDst d;
    for (int k = 0; k < 4; ++k) { d.g[k] = grad; d.end[k] = Cout; }
    d.end[0] = Cout; d.bias_part = nullptr; d.db = nullptr; d.bias_C = 0;
    return launch_reduce(part, d, Cout, Cin, KH * KW, nsplit, split_stride, accumulate, (hipStream_t)stream);
}

extern "C" int kg_wgrad_reduce_bias(const float* part, float* const* grads, const int* counts, int ngrads, int Cin, int KH, int KW,
                                    int nsplit, long split_stride, int accumulate, const float* bias_part, float* db, int bias_C, void* stream);
// Same for a conv fused along Cout (heads that share their input, KGnet.py:161-209): ngrads <= 4 gradient tensors, tensor k = the
// next counts[k] output rows of the fused conv.  One launch instead of one per head.
extern "C" int kg_wgrad_reduce_multi(const float* part, float* const* grads, const int* counts, int ngrads, int Cin, int KH, int KW,
                                     int nsplit, long split_stride, int accumulate, void* stream) {
    KG_CHECK_ARG(part && grads && counts && ngrads >= 1 && ngrads <= 4, "kg_wgrad_reduce_multi: 1..4 gradient tensors");
    KG_CHECK_ARG(KH * KW <= 49, "kg_wgrad_reduce_multi: at most 49 taps");
    return kg_wgrad_reduce_bias(part, grads, counts, ngrads, Cin, KH, KW, nsplit, split_stride, accumulate, nullptr, nullptr, 0, stream);
}

// kg_wgrad_reduce_multi + the bias gradient of the same conv in the SAME launch: bias_part = [nsplit][bias_C] partials written by
// kg_conv2d_wgrad_halo (dbp), db [bias_C] (accumulate applies to it too); bias_part == NULL: weights only.
extern "C" int kg_wgrad_reduce_bias(const float* part, float* const* grads, const int* counts, int ngrads, int Cin, int KH, int KW,
                                    int nsplit, long split_stride, int accumulate, const float* bias_part, float* db, int bias_C, void* stream) {
    KG_CHECK_ARG(part && grads && counts && ngrads >= 1 && ngrads <= 4, "kg_wgrad_reduce_bias: 1..4 gradient tensors");
    KG_CHECK_ARG(KH * KW <= 49, "kg_wgrad_reduce_bias: at most 49 taps");
    KG_CHECK_ARG(!bias_part || (db && bias_C >= 1), "kg_wgrad_reduce_bias: bias partials without an output");
    ReduceDst d;
    int end = 0;
    for (int k = 0; k < 4; ++k) {
        if (k < ngrads) { KG_CHECK_ARG(grads[k] && counts[k] > 0, "kg_wgrad_reduce_bias: bad tensor %d", k); end += counts[k]; }
        d.g[k] = grads[k < ngrads ? k : ngrads - 1]; d.end[k] = end;
    }
    d.bias_part = bias_part; d.db = db; d.bias_C = bias_C;
    return launch_reduce(part, d, end, Cin, KH * KW, nsplit, split_stride, accumulate, (hipStream_t)stream);
}

// Bias gradient: db[c] = sum over rows of dy[row][c] (bf16 rows, fp32 sum), fixed-order two-stage.
__global__ void bias_grad_kernel(const bf16_t* __restrict__ dy, float* __restrict__ part, int M, int C, int ld,
                                 int rows_per_block) {
    extern __shared__ float red[];
    const int nrl = blockDim.x / C;  // row lanes
    const int c = threadIdx.x % C, rl = threadIdx.x / C;
    float s = 0.f;
    int r0 = blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    for (int r = r0 + rl; r < r1; r += nrl) s += bf2f(dy[(long)r * ld + c]);
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        float t = 0.f;
        for (int k = 0; k < nrl; ++k) t += red[k * C + c];
        part[(long)blockIdx.x * C + c] = t;
    }
}
__global__ void bias_grad_final_kernel(const float* __restrict__ part, float* __restrict__ db, int nb, int C,
                                       int accumulate) {
    const int c = blockIdx.x;   // one wave per channel, lanes stride over the block partials (fixed order)
    float s = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) s += part[(long)b * C + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) db[c] = accumulate ? db[c] + s : s;
}

// Vector variant (C % 8 == 0, 16-byte aligned rows): a thread owns one 16-byte channel chunk and a row lane, keeps 4 row
// loads in flight, and the row lanes of a workgroup are combined through LDS in fixed order.
__global__ __launch_bounds__(256) void bias_grad_vec_kernel(const bf16_t* __restrict__ dy, float* __restrict__ part, int M, int C8,
                                                            int ld, int rows_per_block, int nrl) {
    __shared__ float red[256 * 8];
    const int c8 = threadIdx.x % C8, rl = threadIdx.x / C8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r0 = blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    if (rl < nrl) {
        const bf16_t* p = dy + c8 * 8;
        int r = r0 + rl;
        for (; r + 3 * nrl < r1; r += 4 * nrl) {
            uint4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const uint4*>(p + (long)(r + q * nrl) * ld);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&v[q]);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += bf2f(e[k]);
            }
        }
        for (; r < r1; r += nrl) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + (long)r * ld);
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += bf2f(e[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k * 256 + threadIdx.x] = acc[k];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = 0.f;
            for (int q = 0; q < nrl; ++q) t += red[k * 256 + q * C8 + c8];
            part[(long)blockIdx.x * (C8 * 8) + c8 * 8 + k] = t;
        }
    }
}

// Sums nsplit partial bias gradients [nsplit][C] (written by kg_conv2d_wgrad_halo's bias unit) in fixed order.
extern "C" int kg_bias_grad_final(const float* part, float* db, int nsplit, int C, int accumulate, void* stream) {
    KG_CHECK_ARG(part && db && nsplit >= 1 && C >= 1, "kg_bias_grad_final: bad arguments");
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, part, db, nsplit, C, accumulate);
    KG_CHECK_LAUNCH("bias_grad_final");
    return KG_OK;
}

extern "C" int kg_bias_grad(const void* dy, float* db, float* scratch, int scratch_floats, int M, int C, int ld,
                            int accumulate, void* stream) {
    KG_CHECK_ARG(dy && db && scratch, "kg_bias_grad: null pointer");
    KG_CHECK_ARG(C >= 1 && M >= 1, "kg_bias_grad: empty problem");
    if (C % 8 == 0 && ld % 8 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) {
        for (int c0 = 0; c0 < C; c0 += 2048) {   // slabs of <= 256 chunks
            const int Cs = C - c0 < 2048 ? C - c0 : 2048, C8 = Cs / 8;
            const int nrl = 256 / C8;
            int nb = scratch_floats / Cs;
            if (nb > 2048) nb = 2048;
            int need = (M + 4 * nrl - 1) / (4 * nrl);
            if (nb > need) nb = need;
            KG_CHECK_ARG(nb >= 1, "kg_bias_grad: scratch too small");
            int rpb = (M + nb - 1) / nb;
            nb = (M + rpb - 1) / rpb;
            hipLaunchKernelGGL(bias_grad_vec_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy + c0, scratch, M,
                               C8, ld, rpb, nrl);
            hipLaunchKernelGGL(bias_grad_final_kernel, dim3(Cs), dim3(64), 0, (hipStream_t)stream, scratch, db + c0, nb, Cs,
                               accumulate);
        }
        KG_CHECK_LAUNCH("bias_grad");
        return KG_OK;
    }
    for (int c0 = 0; c0 < C; c0 += 1024) {   // channel slabs of <= 1024 (one thread per channel and row lane)
        const int Cs = C - c0 < 1024 ? C - c0 : 1024;
        int nrl = 1024 / Cs;
        if (nrl < 1) nrl = 1;
        int threads = nrl * Cs;
        int nb = scratch_floats / Cs;
        if (nb > 1024) nb = 1024;
        int need = (M + 63) / 64;
        if (nb > need) nb = need;
        KG_CHECK_ARG(nb >= 1, "kg_bias_grad: scratch too small");
        int rpb = (M + nb - 1) / nb;
        nb = (M + rpb - 1) / rpb;
        hipLaunchKernelGGL(bias_grad_kernel, dim3(nb), dim3(threads), threads * sizeof(float), (hipStream_t)stream,
                           (const bf16_t*)dy + c0, scratch, M, Cs, ld, rpb);
        hipLaunchKernelGGL(bias_grad_final_kernel, dim3(Cs), dim3(64), 0, (hipStream_t)stream, scratch, db + c0, nb, Cs,
                           accumulate);
    }
    KG_CHECK_LAUNCH("bias_grad");
    return KG_OK;
}

// Microbenchmark: which part of conv3_c64_kernel (csrc/conv3_c64.hip, generated copy) bounds it at c0's 8 x 512 x 512 x 64 -> 64?
//   MODE bit 0: no tap loop (no LDS fragment reads, no MFMAs); bit 1: the halo is staged once per workgroup (no per-tile global loads);
//   bit 2: no output stores.  mode 0 = the product kernel.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/c3_parts.hip -o tools/micro/c3_parts
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ inline float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 kg_bf16x2_t;
__device__ inline bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ inline unsigned pack2bf(float lo, float hi) { const kg_bf16x2_t v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(unsigned, v); }

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

template <int MODE>
__global__ __launch_bounds__(512) void conv3_c64_kernel(const C3Args a) {
    constexpr int HWD = 18, HPIX = HWD * HWD, HALO_BYTES = HPIX * 128, W_BYTES = 9 * 8192, ROW = HWD * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                        // [9 taps][64 couts][128 B]
    unsigned char* hb = smem + W_BYTES;              // 2 x [18 x 18 px][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, g = lane >> 4;
    const int total = a.tiletab ? a.ntiles : a.N * a.tiles_x * a.tiles_y;

    {   // weights: one 16-byte piece per thread and tap, swizzle on the source chunk
        const int r = tid >> 3, cs = tid & 7;
        const bf16_t* src = a.w + (long)(blockIdx.y * 64 + r) * a.K + (cs ^ (2 * ((r >> 4) & 3) + ((r >> 1) & 1))) * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) KG_C3_GLDS(src + t * 64, wl + t * 8192 + wave * 1024);
    }
    auto tile_geom = [&](int t, long& rowbase, int& Hd, int& Wd, int& oy0, int& ox0) {
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
    auto stage = [&](int t, int buf) {   // LDS-direct loads of the tile's halo (destination lane-linear, swizzle on the source)
        long rowbase; int Hd, Wd, oy0, ox0;
        tile_geom(t, rowbase, Hd, Wd, oy0, ox0);
        unsigned char* dst = hb + buf * HALO_BYTES;
#pragma unroll 1
        for (int q = 0; q < (HPIX * 8 + 511) / 512; ++q) {
            const int e = tid + q * 512;
            if (e < HPIX * 8) {
                const int p = e >> 3, cs = e & 7;
                const int hy = p / HWD, hx = p - hy * HWD;
                const int c = cs ^ (hx & 6);
                const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                const bf16_t* src = reinterpret_cast<const bf16_t*>(kg_c3_zero_line) + c * 8;
                if ((unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd) src = a.x + (rowbase + (long)iy * Wd + ix) * a.ldx + c * 8;
                KG_C3_GLDS(src, dst + (q * 512 + wave * 64) * 16);
            }
        }
    };

    // fragment addresses: weights row r = (lm>>2)*16 + i*4 + (lm&3) (lane ends with 16 consecutive couts), pixel (2*wave + j, lm)
    int a_off[2];
    {
        const int r = (lm >> 2) * 16 + (lm & 3);        // i adds the immediate i*512 (the key does not depend on i)
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
        for (int s = 0; s < 2; ++s) kb[kx][s] = ((wave * 2) * HWD + lm + fx) * 128 + (((4 * s + g) ^ key) * 16);
    }
    const int cb = blockIdx.y * 64 + g * 16;
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = (a.bias && cb + e < a.Cout) ? a.bias[cb + e] : 0.f;
    const bool full = cb + 16 <= a.Cout;

    int t = blockIdx.x, cur = 0;
    if (t < total) stage(t, 0);
    for (; t < total; t += gridDim.x) {
        __syncthreads();                               // this tile's halo (and the weights) have landed; the other buffer is free
        const int tn = t + gridDim.x;
        if (!(MODE & 2) && tn < total) stage(tn, cur ^ 1);
        const unsigned char* halo = hb + cur * HALO_BYTES;
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(MODE & 1))
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int fy = a.flip ? 2 - ky : ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 af[4], bfr[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wl + (ky * 3 + kx) * 8192 + a_off[s] + i * 512);
#pragma unroll
                    for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(halo + fy * ROW + kb[kx][s] + j * ROW);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: lane owns pixel (oy0 + 2*wave + j, ox0 + lm) and couts cb .. cb+15 ----
        long rowbase; int Hd, Wd, oy0, ox0;
        tile_geom(t, rowbase, Hd, Wd, oy0, ox0);
        const int ox = ox0 + lm;
        if (cb < a.Cout && ox < Wd && (!(MODE & 4) || acc[0][0][0] == 12345.f)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = oy0 + wave * 2 + j;
                if (oy >= Hd) continue;
                const long m = rowbase + (long)oy * Wd + ox;
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
        }
        if (!(MODE & 2)) cur ^= 1;
    }
}


// ---- v2: two groups of 4 waves alternate between the MFMA phase of one tile and the memory / VALU phase (halo prefetch of their
// next tile, epilogue + stores of their previous tile) of the neighbouring tiles: one barrier per tile, the phases of consecutive
// tiles overlap instead of following each other.
template <int OFF>
__device__ __forceinline__ void lds_rd128(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const unsigned char*)p;
}

template <int MODE>
__global__ __launch_bounds__(512) void conv3_c64_v2(const C3Args a) {
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
        int t = blockIdx.x + k * gridDim.x;
        if ((MODE & 16) && (gridDim.x & 7) == 0) t = k * gridDim.x + (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
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
        if (MODE & 1) return;
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
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
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
        if ((MODE & 4) && acc[0][0][0] != 12345.f) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + gw * 4 + j;
            if (oy >= Hd) continue;
            const long m = rowbase + (long)oy * Wd + ox;
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
        } else if (MODE & 8) {
            if (k >= 1) epilogue(k - 1);
            if (!(MODE & 2) && k + 1 < nt) stage(k + 1, (k + 1) & 1);
        } else {
            if (!(MODE & 2) && k + 1 < nt) stage(k + 1, (k + 1) & 1);
            if (k >= 1) epilogue(k - 1);
        }
    }
}

template <int MODE, int V>
static void run(C3Args a, int total) {
    constexpr int smem = 9 * 8192 + 2 * 18 * 18 * 128;
    auto kfn = V == 2 ? conv3_c64_v2<MODE> : conv3_c64_kernel<MODE>;
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(256, 1), dim3(512), smem, 0, a);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kfn, dim3(256, 1), dim3(512), smem, 0, a);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 20;
    printf("v%d mode %d (%s%s%s): %.3f ms  %.2f us/tile/CU  (%s)\n", V, MODE, (MODE & 1) ? "no-mma " : "", (MODE & 2) ? "no-stage " : "", (MODE & 4) ? "no-store" : "",
           ms, ms * 1e3 / (total / 256.0), hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int N = 8, H = argc > 1 ? atoi(argv[1]) : 512, W = H;
    const long M = (long)N * H * W;
    C3Args a; memset(&a, 0, sizeof(a));
    void *x, *w, *y; float* bias;
    hipMalloc(&x, M * 64 * 2); hipMalloc(&y, M * 64 * 2); hipMalloc(&w, 64 * 9 * 64 * 2); hipMalloc(&bias, 64 * 4);
    {   // pseudo-random bf16 inputs in (-1, 1) / weights in (-1/16, 1/16)
        bf16_t* hx = (bf16_t*)malloc(M * 64 * 2); bf16_t* hw = (bf16_t*)malloc(64 * 9 * 64 * 2);
        unsigned r = 12345u;
        for (long i = 0; i < M * 64; ++i) { r = r * 1664525u + 1013904223u; hx[i] = (bf16_t)(0x3f00 - ((r >> 20) & 0x1ff) + ((r >> 31) << 15)); }
        for (long i = 0; i < 64 * 9 * 64; ++i) { r = r * 1664525u + 1013904223u; hw[i] = (bf16_t)(0x3d00 - ((r >> 20) & 0x1ff) + ((r >> 31) << 15)); }
        hipMemcpy(x, hx, M * 64 * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw, 64 * 9 * 64 * 2, hipMemcpyHostToDevice);
        free(hx); free(hw);
    }
    hipMemset(bias, 0, 64 * 4);
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y;
    a.N = N; a.H = H; a.W = W; a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16; a.ldx = 64; a.Cout = 64; a.ldy = 64; a.K = 9 * 64; a.relu = 1;
    const int total = N * a.tiles_x * a.tiles_y;
    {   // v2 against v1, bit for bit
        constexpr int smem = 9 * 8192 + 2 * 18 * 18 * 128;
        bf16_t* y1 = (bf16_t*)malloc(M * 64 * 2); bf16_t* y2 = (bf16_t*)malloc(M * 64 * 2);
        hipFuncSetAttribute((const void*)conv3_c64_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipFuncSetAttribute((const void*)conv3_c64_v2<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipMemset(y, 0xff, M * 64 * 2);
        hipLaunchKernelGGL(conv3_c64_kernel<0>, dim3(256, 1), dim3(512), smem, 0, a);
        hipMemcpy(y1, y, M * 64 * 2, hipMemcpyDeviceToHost);
        hipMemset(y, 0xff, M * 64 * 2);
        hipLaunchKernelGGL(conv3_c64_v2<0>, dim3(256, 1), dim3(512), smem, 0, a);
        hipMemcpy(y2, y, M * 64 * 2, hipMemcpyDeviceToHost);
        long bad = 0, nz = 0;
        for (long i = 0; i < M * 64; ++i) { bad += y1[i] != y2[i]; nz += y1[i] != 0; }
        printf("v2 vs v1: %ld mismatching of %ld outputs (%ld non-zero)  (%s)\n", bad, M * 64, nz, hipGetErrorString(hipGetLastError()));
        free(y1); free(y2);
    }
    run<0, 1>(a, total); run<1, 1>(a, total); run<2, 1>(a, total); run<4, 1>(a, total); run<6, 1>(a, total);
    run<0, 2>(a, total); run<1, 2>(a, total); run<2, 2>(a, total); run<4, 2>(a, total); run<6, 2>(a, total); run<7, 2>(a, total); run<16, 2>(a, total);
    return 0;
}

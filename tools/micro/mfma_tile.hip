// Microbenchmark (round 5): does a LARGER WAVE TILE buy MFMA rate under the 1400 W cap?  The 7x7 tap loop feeds a 64 x 64 wave tile: 8 ds_read_b128
// per 16 MFMAs, two waves per SIMD (255 VGPRs each).  One wave per SIMD may use 512 registers (accumulators in AGPRs):
//   TA x TB fragments of 16 rows each, K = 32 per step:   4 x 4 (64 x 64, 8 waves: today)   8 x 4 (128 x 64, 4 waves)   8 x 8 (128 x 128, 4 waves)
//   LDS fragment reads per MFMA:                           0.50                              0.375                        0.25
// Same loop skeleton as mfma_peak.hip mode 4 (asm reads one k-step ahead, counted lgkmcnt, one s_barrier per 64 MFMAs of a wave, one LDS-direct
// load per 32 MFMAs), f16 MFMA, operand data as in mfma_peak (0 constant, 1 dense random, 2 random with 60 % zeros).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_tile.hip -o tools/micro/mfma_tile ; run: tools/micro/mfma_tile [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int N> struct Wait;
template <> struct Wait<8> { static __device__ __forceinline__ void go() { asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); } };
template <> struct Wait<12> { static __device__ __forceinline__ void go() { asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); } };
template <> struct Wait<16> { static __device__ __forceinline__ void go() { asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); } };   // (4-bit field: 15 = at most 15 pending)

template <int I, int T>
__device__ __forceinline__ void rd(f16x8 (&a)[T], unsigned l) {
    if constexpr (I < T) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[I]) : "v"(l), "n"(I * 1024));
        rd<I + 1, T>(a, l);
    }
}
template <int TA, int TB>
__device__ __forceinline__ void reads(f16x8 (&a)[TA], f16x8 (&b)[TB], unsigned la, unsigned lb) {
    rd<0, TA>(a, la);
    rd<0, TB>(b, lb);
}

// IL > 0: the reads of the next k-step are dealt out between the MFMAs of the current one, one ds_read_b128 behind every IL-th MFMA (the wait
// at the head of a k-step is then lgkmcnt(0)); IL = 0: all reads in one block in front of the MFMAs (counted wait).
template <int I, int TA, int TB, int IL>
__device__ __forceinline__ void il_step(f32x4 (&acc)[TA][TB], f16x8 (&a)[TA], f16x8 (&b)[TB], f16x8 (&an)[TA], f16x8 (&bn)[TB], unsigned la, unsigned lb) {
    if constexpr (I < TA * TB) {
        acc[I / TB][I % TB] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[I / TB], b[I % TB], acc[I / TB][I % TB], 0, 0, 0);
        if constexpr ((I + 1) % IL == 0 && (I + 1) / IL - 1 < TA + TB) {
            constexpr int K = (I + 1) / IL - 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (K < TA) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(an[K]) : "v"(la), "n"(K * 1024));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bn[K - TA]) : "v"(lb), "n"((K - TA) * 1024));
            __builtin_amdgcn_sched_barrier(0);
        }
        il_step<I + 1, TA, TB, IL>(acc, a, b, an, bn, la, lb);
    }
}
template <int TA, int TB>
__device__ __forceinline__ void tie_wait0(f16x8 (&a)[TA], f16x8 (&b)[TB]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TA; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
    for (int i = 0; i < TB; ++i) asm volatile("" : "+v"(b[i]));
}

template <int TA, int TB, int WAVES, bool LDS, int IL = 0>
__global__ __launch_bounds__(WAVES * 64) void k(const uint4* __restrict__ g, float* out, int iters, int data) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 120 * 1024 / 16; i += WAVES * 64) {
        uint4 v = make_uint4(0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00);
        if (data) {
            unsigned r[4];
            for (int q = 0; q < 4; ++q) {
                unsigned h = (unsigned)(i * 4 + q) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                // IEEE half in (-1, 1): exponent 0x38..0x3b, random mantissa
                unsigned lo = 0x3800u + (h & 0x3ff) + ((h >> 10) & 1) * 0x8000u, hi = 0x3800u + ((h >> 11) & 0x3ff) + ((h >> 21) & 1) * 0x8000u;
                if (data == 2) { if (((h >> 22) & 15) < 10) lo = 0; if (((h >> 26) & 15) < 10) hi = 0; }
                r[q] = lo | (hi << 16);
            }
            v = make_uint4(r[0], r[1], r[2], r[3]);
        }
        reinterpret_cast<uint4*>(smem)[i] = v;
    }
    __syncthreads();
    f32x4 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    f16x8 a0[TA], b0[TB], a1[TA], b1[TB];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    // conflict-free: the 16 lanes of a group read 16 consecutive 16-byte slots
    const unsigned offA = lds0 + (wave & 1) * 8192 + lane * 16, offB = lds0 + 32768 + (wave >> 1) * 8192 + lane * 16;
    if (LDS) reads<TA, TB>(a0, b0, offA, offB);
    else {
#pragma unroll
        for (int i = 0; i < TA; ++i) a0[i] = a1[i] = *reinterpret_cast<f16x8*>(smem + (wave & 1) * 8192 + lane * 16 + i * 1024);
#pragma unroll
        for (int i = 0; i < TB; ++i) b0[i] = b1[i] = *reinterpret_cast<f16x8*>(smem + 32768 + lane * 16 + i * 1024);
    }
    const uint4* gp = g + (blockIdx.x & 63) * 4096 + tid;
    constexpr int PER_BARRIER = 64 / (TA * TB) > 0 ? 64 / (TA * TB) : 1;      // k-steps between barriers (>= 64 MFMAs per wave)
    for (int it = 0; it < iters; ++it) {
        const int sh = (it & 3) * 2048;
        if (LDS && !(it & 1))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + (it & 7) * (WAVES * 64)),
                                             (__attribute__((address_space(3))) void*)(smem + 65536 + (it & 7) * 8192 + __builtin_amdgcn_readfirstlane(wave) * 1024), 16, 0, 0);
        if constexpr (LDS && IL > 0) {
            tie_wait0<TA, TB>(a0, b0);
            il_step<0, TA, TB, IL>(acc, a0, b0, a1, b1, offA + sh + 4096, offB + sh + 4096);
            __builtin_amdgcn_sched_barrier(0);
            tie_wait0<TA, TB>(a1, b1);
            il_step<0, TA, TB, IL>(acc, a1, b1, a0, b0, offA + sh, offB + sh);
            __builtin_amdgcn_sched_barrier(0);
        } else {
        if (LDS) {
                reads<TA, TB>(a1, b1, offA + sh + 4096, offB + sh + 4096);
                Wait<TA + TB>::go();
            }
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], b0[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LDS) {
                reads<TA, TB>(a0, b0, offA + sh, offB + sh);
                Wait<TA + TB>::go();
            }
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LDS && ((it + 1) * 2 % PER_BARRIER == 0 || PER_BARRIER <= 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * WAVES * 64 + tid] = s + (float)a0[0][0] + (float)b0[0][0];
}

template <int TA, int TB, int WAVES, bool LDS, int IL = 0>
static void run(const uint4* g, float* out, int iters, int data) {
    hipFuncSetAttribute((const void*)k<TA, TB, WAVES, LDS, IL>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    // equal MFMA work per CU for every shape: iters k-step pairs of a 4 x 4 tile on 8 waves
    const int its = (int)((long)iters * 16 * 8 / (TA * TB * WAVES));
    k<TA, TB, WAVES, LDS, IL><<<256, WAVES * 64, 155 * 1024>>>(g, out, its, data);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<TA, TB, WAVES, LDS, IL><<<256, WAVES * 64, 155 * 1024>>>(g, out, its, data);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double fl = 256.0 * WAVES * its * 2 * TA * TB * 16384.0;
    printf("data %d  tile %3d x %3d  waves %d  %s%s: %8.3f ms  %7.1f TFLOP/s  (%d reads / %d MFMAs per k-step) %s\n", data, TA * 16, TB * 16, WAVES,
           LDS ? "LDS-fed " : "registers", IL == 0 ? "            " : IL == 1 ? " read/MFMA   " : " read/2 MFMAs", ms, fl / ms / 1e9, LDS ? TA + TB : 0, TA * TB, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 32000;      // ~20 ms launches: the steady-state clock under the power cap
    uint4* g; float* out;
    hipMalloc(&g, 64 * 4096 * 16 * 2); hipMemset(g, 0, 64 * 4096 * 16 * 2);
    hipMalloc(&out, 4096 * 1024 * 4);
    for (int data = 0; data <= 2; ++data) {
        run<4, 4, 8, false>(g, out, iters, data);
        run<4, 4, 8, true>(g, out, iters, data);
        run<8, 4, 4, false>(g, out, iters, data);
        run<8, 4, 4, true>(g, out, iters, data);
        run<8, 8, 4, false>(g, out, iters, data);
        run<8, 8, 4, true>(g, out, iters, data);
        run<4, 4, 4, true>(g, out, iters, data);
        run<4, 4, 8, true, 1>(g, out, iters, data);
        run<4, 4, 8, true, 2>(g, out, iters, data);
        run<8, 4, 4, true, 1>(g, out, iters, data);
        run<8, 4, 4, true, 2>(g, out, iters, data);
        run<8, 8, 4, true, 2>(g, out, iters, data);
        run<8, 8, 4, true, 3>(g, out, iters, data);
    }
    return 0;
}

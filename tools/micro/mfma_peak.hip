// Microbenchmark: what does the CU sustain for v_mfma_f32_16x16x32_bf16 with the operand traffic of conv_halo's tap loop?
//   mode 0: 16 independent MFMAs per iteration, operands in registers
//   mode 1: + 8 ds_read_b128 per 16 MFMAs (fragments of the NEXT iteration fetched while the current MFMAs run)
//   mode 2: mode 1 + one s_barrier per 64 MFMAs
//   mode 4: mode 3 with the ds_reads issued from inline asm and counted s_waitcnt lgkmcnt(8) by hand (hipcc forces lgkmcnt(0) on every LDS
//           wait once a global_load_lds is in the loop: it models LDS-DMA as a FLAT access that may return out of order)
//   mode 3: mode 2 + 1 global_load_lds (16 B/lane) per 32 MFMAs from an L2-resident buffer, counted vmcnt wait before the barrier
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak ; run: /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const uint4* __restrict__ g, float* out, int iters, int data) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // data 0: every operand element the same constant; 1: pseudo-random bf16 in (-1, 1) (all mantissa bits toggle: the clock the chip
    // sustains under its power limit depends on the operand data); 2: the random values with 60 % zeros (post-ReLU activations)
    for (int i = tid; i < 120 * 1024 / 16; i += WAVES * 64) {
        uint4 v = make_uint4(0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00);
        if (data) {
            unsigned r[4];
            for (int q = 0; q < 4; ++q) {
                unsigned h = (unsigned)(i * 4 + q) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                unsigned lo = 0x3f00u - (h & 0x1ff) + ((h >> 9) & 1) * 0x8000u, hi = 0x3f00u - ((h >> 10) & 0x1ff) + ((h >> 19) & 1) * 0x8000u;
                if (data == 2) { if (((h >> 20) & 15) < 10) lo = 0; if (((h >> 24) & 15) < 10) hi = 0; }
                r[q] = lo | (hi << 16);
            }
            v = make_uint4(r[0], r[1], r[2], r[3]);
        }
        reinterpret_cast<uint4*>(smem)[i] = v;
    }
    __syncthreads();
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    bf16x8 a0[4], b0[4], a1[4], b1[4];
    // conflict-free: 16 lanes of a group read 16 consecutive 16-byte slots
    const int offA = (wave & 1) * 8192 + lane * 16, offB = 32768 + (wave >> 1) * 8192 + lane * 16;
    if (MODE != 4) for (int i = 0; i < 4; ++i) { a0[i] = *reinterpret_cast<bf16x8*>(smem + offA + i * 1024); b0[i] = *reinterpret_cast<bf16x8*>(smem + offB + i * 1024); a1[i] = a0[i]; b1[i] = b0[i]; }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    if (MODE == 4) {   // every LDS read of this mode comes from asm: a compiler-tracked read pending at the loop header makes hipcc wait lgkmcnt(0) per iteration
#define RD0(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
        const unsigned la = lds0 + offA, lb = lds0 + offB;
        RD0(a0[0], la, 0); RD0(a0[1], la, 1024); RD0(a0[2], la, 2048); RD0(a0[3], la, 3072);
        RD0(b0[0], lb, 0); RD0(b0[1], lb, 1024); RD0(b0[2], lb, 2048); RD0(b0[3], lb, 3072);
    }
    const uint4* gp = g + (blockIdx.x & 63) * 4096 + tid;
    for (int it = 0; it < iters; ++it) {
        const int sh = (it & 3) * 2048;
        if (MODE >= 3 && !(it & 1)) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + (it & 7) * 512),
                                             (__attribute__((address_space(3))) void*)(smem + 65536 + (it & 7) * 8192 + __builtin_amdgcn_readfirstlane(wave) * 1024), 16, 0, 0);
        }
        if (MODE == 4) {
            const unsigned la = lds0 + offA + sh + 4096, lb = lds0 + offB + sh + 4096;
#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
            RD(a1[0], la, 0); RD(a1[1], la, 1024); RD(a1[2], la, 2048); RD(a1[3], la, 3072);
            RD(b1[0], lb, 0); RD(b1[1], lb, 1024); RD(b1[2], lb, 2048); RD(b1[3], lb, 3072);
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]));
        } else
        if (MODE >= 1) for (int i = 0; i < 4; ++i) { a1[i] = *reinterpret_cast<bf16x8*>(smem + offA + sh + i * 1024 + 4096); b1[i] = *reinterpret_cast<bf16x8*>(smem + offB + sh + i * 1024 + 4096); }
        __builtin_amdgcn_s_setprio(1);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (MODE == 4) {
            const unsigned la = lds0 + offA + sh, lb = lds0 + offB + sh;
            RD(a0[0], la, 0); RD(a0[1], la, 1024); RD(a0[2], la, 2048); RD(a0[3], la, 3072);
            RD(b0[0], lb, 0); RD(b0[1], lb, 1024); RD(b0[2], lb, 2048); RD(b0[3], lb, 3072);
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]), "+v"(b1[3]));
        } else
        if (MODE >= 1) for (int i = 0; i < 4; ++i) { a0[i] = *reinterpret_cast<bf16x8*>(smem + offA + sh + i * 1024); b0[i] = *reinterpret_cast<bf16x8*>(smem + offB + sh + i * 1024); }
        __builtin_amdgcn_s_setprio(1);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (MODE >= 2 && (it & 1)) {
            if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int MODE, int WAVES>
static void run(const uint4* g, float* out, int blocks, int iters, int data = 0) {
    hipFuncSetAttribute((const void*)k<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<MODE, WAVES><<<blocks, WAVES * 64, 155 * 1024>>>(g, out, iters, data);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<MODE, WAVES><<<blocks, WAVES * 64, 155 * 1024>>>(g, out, iters, data);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double fl = (double)blocks * WAVES * iters * 32 * 16384.0;
    printf("data %d mode %d waves %d blocks %d: %.3f ms  %.1f TFLOP/s (%s)\n", data, MODE, WAVES, blocks, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    uint4* g; float* out;
    hipMalloc(&g, 64 * 4096 * 16 * 2); hipMemset(g, 0, 64 * 4096 * 16 * 2);
    hipMalloc(&out, 4096 * 1024 * 4);
    run<0, 8>(g, out, 256, iters); run<1, 8>(g, out, 256, iters); run<2, 8>(g, out, 256, iters); run<3, 8>(g, out, 256, iters); run<4, 8>(g, out, 256, iters);
    run<0, 4>(g, out, 256, iters); run<1, 4>(g, out, 256, iters);
    run<0, 8>(g, out, 2048, iters / 4); run<3, 8>(g, out, 2048, iters / 4); run<4, 8>(g, out, 2048, iters / 4);
    for (int data = 1; data <= 2; ++data) { run<0, 8>(g, out, 256, iters, data); run<1, 8>(g, out, 256, iters, data); run<4, 8>(g, out, 256, iters, data); }
    run<0, 8>(g, out, 256, iters * 8, 1); run<4, 8>(g, out, 256, iters * 8, 1); run<4, 8>(g, out, 256, iters * 8, 0);   // ~20 ms launches: the steady-state clock
    return 0;
}

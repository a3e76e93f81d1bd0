// Microbenchmark: does the MFMA shape change what the power-limited chip sustains?  A 64 x 64 wave tile per 32-wide k-step as
//   shape 0: 16 x v_mfma_f32_16x16x32_f16   (conv_halo's tap loop today)
//   shape 1:  8 x v_mfma_f32_32x32x16_f16   (same FLOPs, same operand registers, half the operand reads per FLOP, 4x larger accumulator tiles)
// with register operands only (ld = 0) or with the 8 ds_read_b128 per k-step of the tap loop (ld = 1); pseudo-random operands (all mantissa bits toggle).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape.hip -o /tmp/mfma_shape ; run: /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int LD>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 1024 / 16; i += 512) {
        unsigned r[4];
        for (int q = 0; q < 4; ++q) {
            unsigned h = (unsigned)(i * 4 + q) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const unsigned lo = 0x3800u - (h & 0x3ff) + ((h >> 10) & 1) * 0x8000u, hi = 0x3800u - ((h >> 11) & 0x3ff) + ((h >> 21) & 1) * 0x8000u;
            r[q] = lo | (hi << 16);
        }
        reinterpret_cast<uint4*>(smem)[i] = make_uint4(r[0], r[1], r[2], r[3]);
    }
    __syncthreads();
    const int offA = (wave & 1) * 8192 + lane * 16, offB = 32768 + (wave >> 1) * 4096 + lane * 16;
    h8 a0[4], b0[4], a1[4], b1[4];
    for (int i = 0; i < 4; ++i) { a0[i] = *reinterpret_cast<h8*>(smem + offA + i * 1024); b0[i] = *reinterpret_cast<h8*>(smem + offB + i * 1024); a1[i] = a0[i]; b1[i] = b0[i]; }
    f32x4 acc[4][4];
    f32x16 big[2][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) big[i][j][e] = 0.f;
    auto mma = [&](const h8 (&a)[4], const h8 (&b)[4]) {
        if (SHAPE == 0) {
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {   // a[2 * i + kh]: 32 rows of tile i, k half kh
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int kh = 0; kh < 2; ++kh)
                big[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * i + kh], b[2 * j + kh], big[i][j], 0, 0, 0);
        }
    };
    for (int it = 0; it < iters; ++it) {
        const int sh = (it & 3) * 1024;
        if (LD) for (int i = 0; i < 4; ++i) { a1[i] = *reinterpret_cast<h8*>(smem + offA + sh + i * 1024 + 512); b1[i] = *reinterpret_cast<h8*>(smem + offB + (sh >> 1) + i * 1024 + 256); }
        mma(a0, b0);
        if (LD) for (int i = 0; i < 4; ++i) { a0[i] = *reinterpret_cast<h8*>(smem + offA + sh + i * 1024); b0[i] = *reinterpret_cast<h8*>(smem + offB + (sh >> 1) + i * 1024); }
        mma(a1, b1);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += big[i][j][e];
    out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE, int LD>
static void run(float* out, int iters) {
    hipFuncSetAttribute((const void*)k<SHAPE, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<SHAPE, LD><<<256, 512, 64 * 1024>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<SHAPE, LD><<<256, 512, 64 * 1024>>>(out, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double fl = 256.0 * 8 * iters * 32 * 16384.0;
    printf("shape %s, %s: %.3f ms  %.1f TFLOP/s (%s)\n", SHAPE ? "32x32x16" : "16x16x32", LD ? "8 ds_read_b128 per k-step" : "register operands", ms, fl / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 32000;
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) { run<0, 0>(out, iters); run<1, 0>(out, iters); run<0, 1>(out, iters); run<1, 1>(out, iters); }
    return 0;
}

// Microbenchmark: does the STORE PATTERN of the conv epilogues bound the output-heavy kernels?  (round 6)
// A wave of every conv kernel ends with lane (g = lane >> 4, lm = lane & 15) holding 16 consecutive couts of pixel row lm (MFMA 16x16 C layout,
// weights = A) and stores them as two 16-byte pieces per plane (kg_store_planes<16>): ONE store instruction then writes, per row, four 16-byte
// pieces 32 bytes apart -- half of every 64-byte sector.  Patterns, each writing the same [M][64 couts x P planes] rows tensor once:
//   0  product pattern (kg_conv_epilogue<16>): instruction q of plane p: lane (g, lm) -> row lm, bytes [32 g + 16 q, +16)
//   1  quad exchange: the four lanes of a row swap pieces through ds_bpermute so that instruction k writes bytes [64 k + 16 g, +16): whole 64-byte sectors
//   2  ideal: lane-linear 16-byte stores (a wave writes 1 KB contiguous) -- what an LDS transpose tile would give
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o tools/micro/store_pattern ; run: tools/micro/store_pattern [rows] [planes]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

template <int MODE, int P>
__global__ __launch_bounds__(256) void store_kernel(unsigned short* __restrict__ y, long M, int ld, int ps, unsigned seed) {
    const int lane = threadIdx.x & 63, g = lane >> 4, lm = lane & 15;
    const long wave0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    for (long m0 = wave0; m0 < M; m0 += (long)gridDim.x * 256) {
        uint4 v[P][2];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) v[p][q] = make_uint4(seed + lane, seed + p, seed + q, (unsigned)m0);
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // 64 lanes x 16 B = 1 KB = 8 plane-rows of 128 B; a (j, p, q) instruction covers rows 8 * (2 j + q) .. + 7 of plane p
                        const long row = m0 + 8 * (2 * j + q) + (lane >> 3);
                        if (row < M) *reinterpret_cast<uint4*>(y + row * ld + (long)p * ps + (lane & 7) * 8) = v[p][q];
                    }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long row = m0 + j * 16 + lm;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (row < M) *reinterpret_cast<uint4*>(y + row * ld + (long)p * ps + g * 16 + q * 8) = v[p][q];
                } else {
                    // instruction k: lane g writes piece (g & 1) of lane 2 k + (g >> 1)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int src = ((2 * k + (g >> 1)) << 4) | lm;
                        uint4 a, b;
                        a.x = __shfl(v[p][0].x, src, 64); a.y = __shfl(v[p][0].y, src, 64); a.z = __shfl(v[p][0].z, src, 64); a.w = __shfl(v[p][0].w, src, 64);
                        b.x = __shfl(v[p][1].x, src, 64); b.y = __shfl(v[p][1].y, src, 64); b.z = __shfl(v[p][1].z, src, 64); b.w = __shfl(v[p][1].w, src, 64);
                        const uint4 o = (g & 1) ? b : a;
                        if (row < M) *reinterpret_cast<uint4*>(y + row * ld + (long)p * ps + k * 32 + g * 8) = o;
                    }
                }
            }
        }
    }
}

template <int MODE, int P>
static float run(unsigned short* y, long M, int reps) {
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const int blocks = 256 * 8;
    store_kernel<MODE, P><<<blocks, 256>>>(y, M, 64 * P, 64, 1);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < reps; ++r) store_kernel<MODE, P><<<blocks, 256>>>(y, M, 64 * P, 64, r);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    return ms / reps;
}

int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 8L * 512 * 512;
    unsigned short* y;
    hipMalloc(&y, M * 128 * 2);
    printf("rows %ld x 64 couts; bytes written per pass: 1 plane %.0f MB, 2 planes %.0f MB\n", M, M * 128 / 1e6, M * 256 / 1e6);
    const char* names[3] = {"product pattern (32-byte stride pieces)", "quad exchange (whole 64-byte sectors)", "lane-linear (ideal)"};
    float t[3][2];
    t[0][0] = run<0, 1>(y, M, 20); t[1][0] = run<1, 1>(y, M, 20); t[2][0] = run<2, 1>(y, M, 20);
    t[0][1] = run<0, 2>(y, M, 20); t[1][1] = run<1, 2>(y, M, 20); t[2][1] = run<2, 2>(y, M, 20);
    for (int m = 0; m < 3; ++m)
        printf("%-44s 1 plane %.3f ms = %5.0f GB/s | 2 planes %.3f ms = %5.0f GB/s\n", names[m], t[m][0], M * 128 / t[m][0] / 1e6, t[m][1], M * 256 / t[m][1] / 1e6);
    hipFree(y);
    return 0;
}

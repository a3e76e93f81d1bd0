// Microbenchmark: the k-step loop skeleton of wgrad_halo_kernel<7,1,4> (csrc/wgrad_halo.hip) -- what bounds it?
// Per k-step (32 pixels) a wave reads NCF = 4 dY^T fragments (2 x ds_read_b64_tr_b16 each) and, per (tap, ci-fragment) unit,
// one X^T fragment (2 transpose reads), and issues 4 MFMAs per unit; 8 waves per workgroup, one workgroup per CU.
//   mode 0: MFMAs only (operands in registers), U = 7 units
//   mode 1: + the transpose reads of the real loop (8 + 2 U per k-step)
//   mode 2: the same number of bytes with plain ds_read_b64 (no transpose): is the transpose read slower?
//   mode 3: mode 1 with the A fragments read once per TWO k-steps' worth of units (what a 128-cout workgroup would do: half the A reads per MFMA)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/wgrad_skel.hip -o /tmp/wgrad_skel ; run: /tmp/wgrad_skel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <int MODE, int U>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 96 * 1024 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00);
    __syncthreads();
    f32x4 acc[U][4];
    for (int q = 0; q < U; ++q) for (int c = 0; c < 4; ++c) acc[q][c] = f32x4{0, 0, 0, 0};
    bf16x8 af[4], bfr;
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 8; ++e) af[c][e] = (__bf16)1.0f;
    for (int e = 0; e < 8; ++e) bfr[e] = (__bf16)1.0f;
    const int base = (lane >> 4) * 2048 + (lane & 15) * 8;     // 16-lane groups on distinct rows, 8-byte pieces: conflict-free for the transpose read
    for (int it = 0; it < iters; ++it) {
        const int sh = (it & 7) * 4096;
        if (MODE == 1 || MODE == 3) {
            if (MODE == 1 || !(it & 1))
                for (int c = 0; c < 4; ++c) for (int h = 0; h < 2; ++h) {
                    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(smem + base + sh + c * 256 + h * 128));
                    af[c][h * 4] = v[0]; af[c][h * 4 + 1] = v[1]; af[c][h * 4 + 2] = v[2]; af[c][h * 4 + 3] = v[3];
                }
        } else if (MODE == 2) {
            for (int c = 0; c < 4; ++c) for (int h = 0; h < 2; ++h) {
                bf16x4 v = *reinterpret_cast<const bf16x4*>(smem + base + sh + c * 256 + h * 128);
                af[c][h * 4] = v[0]; af[c][h * 4 + 1] = v[1]; af[c][h * 4 + 2] = v[2]; af[c][h * 4 + 3] = v[3];
            }
        }
        for (int q = 0; q < U; ++q) {
            if (MODE == 1 || MODE == 3) {
                for (int h = 0; h < 2; ++h) {
                    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(smem + 40960 + base + sh + ((wave + 8 * q) & 31) * 512 + h * 128));
                    bfr[h * 4] = v[0]; bfr[h * 4 + 1] = v[1]; bfr[h * 4 + 2] = v[2]; bfr[h * 4 + 3] = v[3];
                }
            } else if (MODE == 2) {
                for (int h = 0; h < 2; ++h) {
                    bf16x4 v = *reinterpret_cast<const bf16x4*>(smem + 40960 + base + sh + ((wave + 8 * q) & 31) * 512 + h * 128);
                    bfr[h * 4] = v[0]; bfr[h * 4 + 1] = v[1]; bfr[h * 4 + 2] = v[2]; bfr[h * 4 + 3] = v[3];
                }
            }
            for (int c = 0; c < 4; ++c) acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bfr, acc[q][c], 0, 0, 0);
        }
    }
    float s = 0;
    for (int q = 0; q < U; ++q) for (int c = 0; c < 4; ++c) s += acc[q][c][0] + acc[q][c][1] + acc[q][c][2] + acc[q][c][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int U>
static void run(float* out, int iters) {
    hipFuncSetAttribute((const void*)k<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<MODE, U><<<256, 512, 100 * 1024>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<MODE, U><<<256, 512, 100 * 1024>>>(out, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double fl = 256.0 * 8 * iters * U * 4 * 16384.0;
    printf("mode %d U %d: %.3f ms  %.1f TFLOP/s (%s)\n", MODE, U, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    run<0, 7>(out, iters); run<1, 7>(out, iters); run<2, 7>(out, iters); run<3, 7>(out, iters);
    run<0, 6>(out, iters); run<1, 6>(out, iters);
    return 0;
}

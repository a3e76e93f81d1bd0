// f16_mfma.hip -- does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs on gfx950, and what does a hi + lo fp16 split of fp32
// operands cost in accuracy?  (hipcc --offload-arch=gfx950 -O3 tools/micro/f16_mfma.hip -o tools/micro/f16_mfma)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;

// one wave: D[16][16] = A[16][32] * B[32][16]; A row-major [m][k], B given as Bt[n][k]
__global__ void mm(const _Float16* A, const _Float16* Bt, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[r * 32 + g * 8 + e]; b[e] = Bt[r * 32 + g * 8 + e]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg;  with "A" = first operand rows -> D rows
    for (int e = 0; e < 4; ++e) D[(g * 4 + e) * 16 + r] = c[e];
}

int main() {
    _Float16 hA[16 * 32], hB[16 * 32];
    float hD[256];
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    // test 1: subnormal A (2^-20, 2^-24) times B = 1 and B = 1024
    for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)0.f; hB[i] = (_Float16)0.f; }
    for (int m = 0; m < 16; ++m) { hA[m * 32 + 0] = (_Float16)ldexpf(1.f, -20); hA[m * 32 + 1] = (_Float16)ldexpf(1.f, -24); }
    for (int n = 0; n < 16; ++n) { hB[n * 32 + 0] = (_Float16)(n & 1 ? 1024.f : 1.f); hB[n * 32 + 1] = (_Float16)(n & 1 ? 1024.f : 1.f); }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    mm<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    printf("subnormal A: D[0][0] = %g (expect %g if kept, 0 if flushed)   D[0][1] = %g (expect %g)\n", hD[0], ldexp(1.0, -20) + ldexp(1.0, -24),
           hD[1], 1024.0 * (ldexp(1.0, -20) + ldexp(1.0, -24)));
    // test 2: subnormal on the B side
    for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)0.f; hB[i] = (_Float16)0.f; }
    for (int m = 0; m < 16; ++m) hA[m * 32] = (_Float16)1.f;
    for (int n = 0; n < 16; ++n) hB[n * 32] = (_Float16)ldexpf(1.f, -22);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    mm<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    printf("subnormal B: D[0][0] = %g (expect %g)\n", hD[0], ldexp(1.0, -22));
    // test 3: accuracy of the hi + lo split: x, w ~ N(0,1) * scale; products hh + hl + lh accumulated in fp32 by three MFMAs
    srand(1);
    double worst[3] = {0, 0, 0};
    const float scales[3] = {1.f, 0.01f, 100.f};
    for (int t = 0; t < 3; ++t) {
        float X[512], W[512];
        _Float16 xh[512], xl[512], wh[512], wl[512];
        for (int i = 0; i < 512; ++i) {
            float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
            X[i] = sqrtf(-2 * logf(u1)) * cosf(6.2831853f * u2) * scales[t];
            u1 = (rand() + 1.f) / (RAND_MAX + 2.f); u2 = rand() / (float)RAND_MAX;
            W[i] = sqrtf(-2 * logf(u1)) * cosf(6.2831853f * u2) * scales[t];
            xh[i] = (_Float16)X[i]; xl[i] = (_Float16)(X[i] - (float)xh[i]);
            wh[i] = (_Float16)W[i]; wl[i] = (_Float16)(W[i] - (float)wh[i]);
        }
        float acc[256] = {0};
        const _Float16* pa[3] = {xl, xh, xh}; const _Float16* pb[3] = {wh, wl, wh};
        for (int p = 0; p < 3; ++p) {
            hipMemcpy(dA, pa[p], sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, pb[p], sizeof(hB), hipMemcpyHostToDevice);
            mm<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
            for (int i = 0; i < 256; ++i) acc[i] += hD[i];
        }
        double rms = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            double ref = 0; for (int k = 0; k < 32; ++k) ref += (double)X[m * 32 + k] * (double)W[n * 32 + k];
            rms += ref * ref;
            worst[t] = fmax(worst[t], fabs(acc[m * 16 + n] - ref));
        }
        rms = sqrt(rms / 256);
        printf("split fp16 x2, scale %g: max |err| / rms = %.3g  (fp32 eps 6e-8)\n", scales[t], worst[t] / rms);
    }
    return 0;
}

// Does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950 (hip_ext.h says "not supported on GFX9xx" for the module API)?
// Two kernels of 64 workgroups each spin ~T us; launched back to back in ONE stream.  In order: 2T.  Any order honoured: ~T.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/anyorder.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(long long ticks, int* out) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && out) atomicAdd(out, 1);
}
static float run(int nk, int wgs, long long ticks, unsigned flags2, hipStream_t st, int* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a, st);
        for (int k = 0; k < nk; ++k)
            hipExtLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, st, nullptr, nullptr, (k & 1) ? flags2 : 0u, ticks, d);
        hipEventRecord(b, st);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    const long long ticks = 100 * 100;      // wall_clock64: 100 MHz -> 100 us
    for (int wgs : {64, 128, 256, 512}) {
        float t1 = run(1, wgs, ticks, 0, st, d);
        float t0 = run(8, wgs, ticks, 0, st, d);
        float ta = run(8, wgs, ticks, hipExtAnyOrderLaunch, st, d);
        printf("wgs %4d: 1 kernel %.1f us | 8 kernels in order %.1f us | odd ones any-order %.1f us\n", wgs, t1 * 1e3, t0 * 1e3, ta * 1e3);
    }
    // short kernels: the per-launch cost of a dependent chain against an any-order chain
    for (int wgs : {64, 256}) {
        float t0 = run(64, wgs, 100, 0, st, d);        // 1 us kernels
        float ta = run(64, wgs, 100, hipExtAnyOrderLaunch, st, d);
        printf("wgs %4d, 64 x 1 us kernels: in order %.2f us per launch | odd ones any-order %.2f us per launch\n", wgs, t0 * 1e3 / 64, ta * 1e3 / 64);
    }
    return 0;
}

// kg_common.h -- shared device/host helpers for libkgnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define KG_OK 0
#define KG_ERR_ARG 1
#define KG_ERR_HIP 2

void kg_set_error(const char* fmt, ...);

#define KG_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            kg_set_error(__VA_ARGS__);          \
            return KG_ERR_ARG;                  \
        }                                       \
    } while (0)

#define KG_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            kg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return KG_ERR_HIP;                                                       \
        }                                                                            \
    } while (0)

#define KG_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            kg_set_error("%s failed: %s", #call, hipGetErrorString(e_));              \
            return KG_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even f32 -> bf16 (NaN stays NaN: quiet bit forced)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// hipcc models global_load_lds (LDS-DMA) as a FLAT access that may return out of order: once one is pending, EVERY LDS wait it
// inserts is s_waitcnt lgkmcnt(0), which also waits for the fragment reads just issued for the NEXT k-step and exposes one LDS
// latency per tap (tools/micro/mfma_peak.hip: 1.80 -> 1.94 PFLOP/s on the bare loop skeleton).  The 7x7 tap loop therefore issues
// its fragment reads from inline asm (invisible to that pass) and counts lgkmcnt by hand; the wait is tied to the fragment
// registers it guards ("+v"), so the MFMAs that consume them cannot be scheduled above it.
template <int OFF>
__device__ __forceinline__ void lds_rd128(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>   // returns once at most N LDS reads of this wave are outstanding (they return in order)
__device__ __forceinline__ void lgkm_wait(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {   // LDS byte address of a pointer into __shared__ memory
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const unsigned char*)p;
}

static inline int kg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

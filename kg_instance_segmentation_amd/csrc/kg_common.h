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

static inline int kg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

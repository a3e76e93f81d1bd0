// api.hip -- error channel, version and device probes of libkgnet_hip.so (C ABI, see include/kgnet_hip.h).
#include "kg_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void kg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* kg_last_error(void) { return g_err; }
extern "C" int kg_version(void) { return 100; }

// Name of the GPU architecture the process sees (e.g. "gfx950:sramecc+:xnack-"); KG_ERR_HIP when no device.
extern "C" int kg_device_arch(char* out, int cap) {
    hipDeviceProp_t p;
    int dev = 0;
    KG_HIP(hipGetDevice(&dev));
    KG_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(out, cap, "%s", p.gcnArchName);
    return KG_OK;
}

// ---- ds_read_b64_tr_b16 semantics probe (used by tests/test_gpu_kernels.py) ----------------------
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__global__ void tr_probe_kernel(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4];
    const int l = threadIdx.x;
    for (int j = 0; j < 4; ++j) lds[l * 4 + j] = (unsigned short)(l * 4 + j);  // element id, lane-linear 8-byte pieces
    __syncthreads();
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&lds[l * 4]));
    *reinterpret_cast<uint2*>(&out[l * 4]) = __builtin_bit_cast(uint2, v);
}
extern "C" int kg_tr_probe(void* out, void* stream) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned short*)out);
    KG_CHECK_LAUNCH("tr_probe");
    return KG_OK;
}

// conv_args.h -- argument block shared by the gather implicit-GEMM kernels (conv_igemm.hip, conv_gather.hip).
#pragma once
#include "kg_common.h"

struct ConvArgs {
    const bf16_t* x; const bf16_t* w; const float* bias;
    bf16_t* y; float* y_f32; const bf16_t* res; const bf16_t* mask; const int2* rowdesc;
    int M, H, W, OH, OW;
    int ldx, ldy, ldres, ldmask;
    int Cout, K, cpt, cpt_magic, ntaps, KW, stride_log2, pad, dil;
    int mode;  // 0 dense forward, 1 dense transposed (dgrad), 2 ragged (+), 3 ragged transposed (-)
    int relu, f32_C;
};

// conv_gather.hip: the deep-prefetch variant for cin_pad % 64 == 0 and bf16 row outputs
int kg_launch_conv_gather(const ConvArgs& a, int cin_pad, hipStream_t st);
// conv_small.hip: direct VALU kernel for cin_pad == 8 (image / single-channel inputs)
int kg_launch_conv_small(const ConvArgs& a, hipStream_t st);

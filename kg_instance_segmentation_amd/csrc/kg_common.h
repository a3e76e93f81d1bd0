// kg_common.h -- shared device/host helpers for libkgnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// ---- 16-bit operand format of the build ---------------------------------------------------------------------------------------
// The library is built twice from the same sources (build.py): libkgnet_hip.so with bfloat16 rows / packed weights and
// libkgnet_hip_f16.so (-DKG_F16) with IEEE half rows.  Both run at the same MFMA rate (v_mfma_f32_16x16x32_{bf16,f16}); half keeps
// 11 significant bits per plane instead of 8, so hi + lo half planes carry 22 bits (bf16: three planes for 24) and a product of two
// such tensors needs 3 MFMA products instead of 6 -- at the price of half's 5-bit exponent, which the build pays for with
//   * packed weights stored as w * 2^12 (KG_WSCALE; |w| < 16 representable, the lo plane of any |w| > 3e-5 is a NORMAL half) and
//     every conv epilogue scaling its fp32 accumulators by 2^-12 (KG_ACC),
//   * activations stored as they are (|x| <= 65504; the lo plane of small values is a subnormal half, which the f16 MFMA keeps:
//     tools/micro/f16_mfma.hip -- absolute error <= 2^-25),
//   * gradients stored times a per-step power of two chosen on the device from the loss gradients (kg_grad_scale).
// `bf16_t` / `bf16x8` / bf2f / f2bf keep their names in both builds: "the 16-bit storage element" and its conversions.
typedef uint16_t bf16_t;  // raw bits of the 16-bit storage element (bfloat16, or IEEE half under KG_F16)
#ifdef KG_F16
typedef _Float16 kg_h16;
#define KG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define KG_WSCALE 4096.0f
#define KG_OSCALE (1.0f / 4096.0f)
#define KG_ACC(v) ((v) * KG_OSCALE)          // fp32 accumulator of x * (w * KG_WSCALE) -> value
#define KG_BIAS_ACC(b) ((b) * KG_WSCALE)     // a bias preloaded into such an accumulator
#define KG_ROWS_FORMAT 1
typedef __attribute__((ext_vector_type(4))) __fp16 kg_fp16x4_;        // (the builtin's own element type; same bits as _Float16)
#define KG_DS_READ_TR16(p) __builtin_bit_cast(bf16x4_fwd_, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) kg_fp16x4_*)(p)))
#else
typedef __bf16 kg_h16;
#define KG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define KG_WSCALE 1.0f
#define KG_OSCALE 1.0f
#define KG_ACC(v) (v)
#define KG_BIAS_ACC(b) (b)
#define KG_ROWS_FORMAT 0
#define KG_DS_READ_TR16(p) __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p)
#endif
typedef __attribute__((ext_vector_type(8))) kg_h16 bf16x8;
typedef __attribute__((ext_vector_type(4))) kg_h16 bf16x4;
typedef bf16x4 bf16x4_fwd_;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define KG_OK 0
#define KG_ERR_ARG 1
#define KG_ERR_HIP 2

void kg_set_error(const char* fmt, ...);
// Measurement aid (kg_last_kernel, api.hip): every conv-family launcher notes the name of the kernel it launched -- the name rocprofv3
// prints for it -- so that bench.py's per-kernel table carries the profiler's own kernel names.  Thread-local, a pointer store per launch.
void kg_note_kernel(const char* name);
#define KG_KNAME(buf, fmt, ...) static char buf[96] = ""; if (!buf[0]) snprintf(buf, sizeof(buf), fmt, __VA_ARGS__)

// One-time per-DEVICE set-up of a launcher (hipFuncSetAttribute for > 64 KB of dynamic LDS): function attributes belong to the device's
// code object, so a process that drives several devices must set them on each (a plain `static bool` assumed one device per process).
struct KgPerDevice {
    bool done[32] = {};
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

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

#ifdef KG_F16
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
#else
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
#endif
// round-to-nearest-even f32 -> 16 bit through the hardware convert: the casts below compile to ONE v_cvt_pk_bf16_f32 per pair on
// gfx950 (the integer add-and-shift formulation costs ~8 VALU per value: 1 us per 16x16-pixel tile in the conv epilogues); half:
// v_cvt_f16_f32 (overflow -> inf: an activation beyond +-65504 surfaces as inf / NaN downstream, it is never clamped silently)
typedef __attribute__((ext_vector_type(2))) kg_h16 kg_bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (kg_h16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const kg_bf16x2_t v = {(kg_h16)lo, (kg_h16)hi};
    return __builtin_bit_cast(uint32_t, v);
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

// ReLU as torch.relu evaluates it: a NaN stays a NaN (`v > 0 ? v : 0` would turn it into 0 and hide an out-of-range activation of the
// half build behind the next layer)
__device__ __forceinline__ float kg_relu(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ unsigned lds_addr(const void* p) {   // LDS byte address of a pointer into __shared__ memory
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const unsigned char*)p;
}

static inline int kg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// i = q * d + r for a flat element index i >= 0 and a (wave-uniform) divisor d > 0.  The elementwise kernels decompose a 64-bit flat
// index per 16-byte chunk; a 64-bit integer division is ~100 VALU instructions on gfx950 -- as much as the rest of such a thread -- so:
// a shift when d is a power of two (scalar test, d is a kernel argument), a 32-bit division when i fits, the 64-bit one otherwise.
__device__ __forceinline__ void kg_divmod(long i, int d, long* q, int* r) {
    if ((d & (d - 1)) == 0) {
        const int sh = __builtin_ctz((unsigned)d);
        *q = i >> sh; *r = (int)(i & (long)(d - 1));
    } else if ((unsigned long)i <= 0xffffffffUL) {
        const unsigned qi = (unsigned)i / (unsigned)d;
        *q = (long)qi; *r = (int)((unsigned)i - qi * (unsigned)d);
    } else {
        const long qq = i / d;
        *q = qq; *r = (int)(i - qq * d);
    }
}

// Side channel of kg_conv_stats_begin / kg_conv_stats_end (api.hip): the NEXT conv launch of this host thread that supports it
// writes the BatchNorm statistics partials of its output (conv_args.h) into `part` and records its pixel-tile count.
// BACKWARD statistics (kg_conv_bstats_begin): the armed launch is the input gradient that COMPLETES the gradient g of a train-mode BatchNorm's
// output; besides storing g its epilogue sums, per channel, g and g * xhat with xhat = (x - mean) * invstd of the BatchNorm's input x
// (KgBStat: the rows of x, same row index as the gradient rows) -- the two sums of the BatchNorm backward (KGnet.py:82-93 under autograd),
// which the separate column reduction re-read g and x for (43 launches, 0.9 ms per step).
struct KgBStat { const unsigned short* x; int ldx, P, ps; const float* mean; const float* invstd; };
struct KgConvStats { float* part; long cap_floats; int nb; KgBStat bs; };
KgConvStats& kg_conv_stats();
static inline float* kg_conv_stats_claim(int tiles, int Cout) {   // launcher side: returns the partial buffer when armed and large enough
    KgConvStats& st = kg_conv_stats();
    if (!st.part || st.nb != 0 || (long)tiles * Cout * 2 > st.cap_floats) return nullptr;
    st.nb = tiles;
    return st.part;
}

// ---- split-bf16 storage ("planes") -------------------------------------------------------------------------------------
// The reference computes in fp32 (KGnet.py:22-29 -> F.conv2d on fp32 tensors).  CDNA4's fast matrix path is bf16 MFMA with
// fp32 accumulation, so wider-than-bf16 tensors are stored as P planes of bf16 whose sum is the value:
//   P = 1: bf16;  P = 2: hi + lo (16 significant bits);  P = 3: hi + mid + lo == the fp32 value exactly.
// Plane p of a rows tensor lives `pstride` elements after plane p-1 (same row, same ld).  A product of an (xP)-plane
// activation with a (wP)-plane weight is evaluated as the bf16 MFMA products x_i * w_j with i + j < max(xP, wP)
// (3 products for P = 2, 6 for P = 3: every dropped term is below the last kept plane), accumulated in fp32.
// ABI side: kg_planes_t (include/kgnet_hip.h); a null pointer means single-plane bf16 everywhere.
struct kg_planes_t {
    int a_planes, a_pstride;   // first bf16 rows operand (x; dY of an input gradient)
    int b_planes, b_pstride;   // second bf16 rows operand (dY of a weight gradient / BN backward; residual of a conv)
    int c_planes, c_pstride;   // third bf16 rows operand
    int y_planes, y_pstride;   // output rows
    int w_planes;              // planes of the packed weights (virtual-channel layout of kg_pack_weight*)
    int reserved_;
    const float* scale;        // device scalar or null: fp32 -> rows conversions (kg_grad_pack, kg_f32_to_planes) multiply by *scale
    const float* oscale;       // device fp32 [Cout] or null: forward convs (kg_conv2d_igemm, kg_conv2d_halo) compute y = act(acc * oscale[c] + bias[c] (+ res))
                               // -- an inference-mode BatchNorm (KGnet.py:82-97: conv -> bn -> relu) folded into the conv's epilogue, on the fp32 accumulator
};
static inline kg_planes_t kg_planes_or_default(const kg_planes_t* p) {
    kg_planes_t d = {1, 0, 1, 0, 1, 0, 1, 0, 1, 0, nullptr, nullptr};
    if (p) {
        d = *p;
        if (d.a_planes < 1) d.a_planes = 1;
        if (d.b_planes < 1) d.b_planes = 1;
        if (d.c_planes < 1) d.c_planes = 1;
        if (d.y_planes < 1) d.y_planes = 1;
        if (d.w_planes < 1) d.w_planes = 1;
    }
    return d;
}
static inline bool kg_planes_ok(const kg_planes_t& d) {
    return d.a_planes <= 3 && d.b_planes <= 3 && d.c_planes <= 3 && d.y_planes <= 3 && d.w_planes <= 3 &&
           d.a_pstride % 8 == 0 && d.b_pstride % 8 == 0 && d.c_pstride % 8 == 0 && d.y_pstride % 8 == 0;
}

// K-side map of a convolution over a planed input.  The kept products x_i * w_j (i + j < max(xP, wP)) are ordered SMALLEST FIRST
// (descending i + j; the hi * hi product last), so that the low-order terms are summed while the fp32 accumulator is still
// small and the final hi * hi additions round like the reference's own fp32 accumulation.  The packed weights hold, per tap,
// one cin_pad-channel copy of w plane j(v) for every virtual plane v; a kernel walks the `total` channel units of a tap exactly
// as for a single-plane conv and only remaps the X-side offset of unit q to plane i(v).
static inline int kg_plane_pairs(int xP, int wP, int* xi, int* wj) {   // xi / wj [6]; returns the number of virtual planes
    const int T = xP > wP ? xP : wP;
    int n = 0;
    for (int sum = T - 1; sum >= 0; --sum)
        for (int j = wP - 1; j >= 0; --j) {
            const int i = sum - j;
            if (i >= 0 && i < xP) { xi[n] = i; wj[n] = j; ++n; }
        }
    return n;
}
struct KMap {
    int n;            // channel units per plane (C / unit)
    int total;        // units per tap = virtual planes * n
    int xps;          // element stride between x planes
    int unit;         // channels per unit (64 for the tile kernels, 8 for conv_igemm)
    unsigned xtab;    // x plane of virtual plane v: (xtab >> 2v) & 3
    __device__ __forceinline__ int xoff(int q) const {   // element offset (plane + channel) of virtual unit q
        if (total == n) return q * unit;
        const int v = q / n;
        return (int)((xtab >> (2 * v)) & 3u) * xps + (q - v * n) * unit;
    }
};
static inline int kg_kmap_segs(int xP, int wP, int* s) {   // (kept for callers that only need the count) returns the virtual planes
    int xi[6], wj[6];
    (void)s;
    return kg_plane_pairs(xP, wP, xi, wj);
}
static inline KMap kg_make_kmap(int C, int unit, int xP, int xps, int wP) {
    int xi[6], wj[6];
    const int nv = kg_plane_pairs(xP, wP, xi, wj);
    KMap m;
    m.n = C / unit; m.unit = unit; m.xps = xps; m.total = nv * m.n; m.xtab = 0;
    for (int v = 0; v < nv; ++v) m.xtab |= (unsigned)xi[v] << (2 * v);
    return m;
}

// Weight gradients over planed operands: dW = sum over the kept products x_i * dY_j (i + j < max(xP, dP), smallest first).  The
// reduction dimension (pixels) is simply walked once per product: "virtual" pixel chunk / tile q -> (product q / per_plane,
// chunk q % per_plane), operands offset by the product's plane strides.  One launch, one set of fp32 accumulators, no extra partials.
struct WgPairs {
    int n;              // products (1 = single-plane operands)
    int xoff[6], doff[6];   // element offsets of the product's x / dY plane
};
static inline WgPairs kg_make_wgpairs(int xP, int xps, int dP, int dps) {
    WgPairs w;
    int xi[6], dj[6];
    w.n = kg_plane_pairs(xP < 1 ? 1 : xP, dP < 1 ? 1 : dP, xi, dj);
    for (int k = 0; k < 6; ++k) { w.xoff[k] = k < w.n ? xi[k] * xps : 0; w.doff[k] = k < w.n ? dj[k] * dps : 0; }
    return w;
}

// fp32 -> P planes (round-to-nearest-even at every level: the residual of each plane is exact in fp32)
template <int NV>
__device__ __forceinline__ void kg_store_planes(bf16_t* yp, int P, int ps, float (&v)[NV], bool vec) {
    static_assert(NV % 8 == 0, "stores are 16-byte chunks");
    for (int p = 0; p < P; ++p) {
        bf16_t h[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) h[e] = f2bf(v[e]);
        if (vec) {
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) {
                uint4 o;
                o.x = (uint32_t)h[q * 8 + 0] | ((uint32_t)h[q * 8 + 1] << 16); o.y = (uint32_t)h[q * 8 + 2] | ((uint32_t)h[q * 8 + 3] << 16);
                o.z = (uint32_t)h[q * 8 + 4] | ((uint32_t)h[q * 8 + 5] << 16); o.w = (uint32_t)h[q * 8 + 6] | ((uint32_t)h[q * 8 + 7] << 16);
                *reinterpret_cast<uint4*>(yp + (long)p * ps + q * 8) = o;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NV; ++e) yp[(long)p * ps + e] = h[e];
        }
        if (p + 1 < P) {
#pragma unroll
            for (int e = 0; e < NV; ++e) v[e] -= bf2f(h[e]);
        }
    }
}
// partial variant: only the first nvalid channels exist
template <int NV>
__device__ __forceinline__ void kg_store_planes_n(bf16_t* yp, int P, int ps, float (&v)[NV], int nvalid) {
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const bf16_t h = f2bf(v[e]);
            if (e < nvalid) yp[(long)p * ps + e] = h;
            v[e] -= bf2f(h);
        }
    }
}
// sum of the planes of 8 consecutive channels (16-byte aligned), lowest plane first so that the sum is exact
__device__ __forceinline__ void kg_load_planes8(const bf16_t* xp, int P, int ps, float (&v)[8]) {
    uint4 q = *reinterpret_cast<const uint4*>(xp + (long)(P - 1) * ps);
    const bf16_t* s = reinterpret_cast<const bf16_t*>(&q);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(s[e]);
    for (int p = P - 2; p >= 0; --p) {
        q = *reinterpret_cast<const uint4*>(xp + (long)p * ps);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f(s[e]);
    }
}
__device__ __forceinline__ float kg_load_planes1(const bf16_t* xp, int P, int ps) {
    float v = bf2f(xp[(long)(P - 1) * ps]);
    for (int p = P - 2; p >= 0; --p) v += bf2f(xp[(long)p * ps]);
    return v;
}

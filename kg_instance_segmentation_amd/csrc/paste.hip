// paste.hip -- mask paste-back of the inference driver on gfx950.
// Reference: test.py:127-157 (`post_processing`): per detection cv2.resize(patch, (x2-x1, y2-y1)) -> paste into a zero
// (input_h, input_w) mask -> cv2.resize(mask, (image_w, image_h)) -> mask >= seg_thresh.  The reference builds every full-size
// mask on the host (two OpenCV resizes + a 1 MB array per detection); here one thread per output pixel evaluates the same
// two-stage bilinear expression directly from the patch probabilities that forward_seg left in HBM.
// Interpolation rule = the published generic INTER_LINEAR float path of OpenCV's resize.cpp (oracle/paste.py states it and why
// it is "parity unpinned": cv2 is absent here); compiled with -ffp-contract=off so that every product and sum rounds as written
// (horizontal pass first, float32).  HBM-bound: algorithmic bytes = the nd * image_h * image_w output bytes.
#include "kg_common.h"
#include <math.h>

struct Taps { int s0, s1; float c0, c1; int single; };
// destination index d of a resize ssize -> dsize; horizontal = OpenCV's xofs/alpha (border taps collapsed, single tap from xmax
// on), vertical = rows clamped with unchanged coefficients
__device__ __forceinline__ Taps lin_taps(int d, int ssize, int dsize, bool horizontal) {
    Taps t;
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (horizontal) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
        t.single = s + 1 >= ssize;
        t.s0 = s; t.s1 = s + 1 < ssize ? s + 1 : ssize - 1;
    } else {
        t.single = 0;
        t.s0 = s < 0 ? 0 : (s > ssize - 1 ? ssize - 1 : s);
        t.s1 = s + 1 < 0 ? 0 : (s + 1 > ssize - 1 ? ssize - 1 : s + 1);
    }
    t.c0 = 1.f - f; t.c1 = f;
    return t;
}
__device__ __forceinline__ float lin_row(float v0, float v1, const Taps& t) {
    if (t.single) return v0;
    const float a = v0 * t.c0, b = v1 * t.c1;
    return a + b;
}

struct PasteDet { int off, ph, pw, y1, x1, y2, x2, pad; };

// value of the patch resized to (bh, bw) at (ry, rx)
__device__ __forceinline__ float resized_patch(const float* __restrict__ p, int ph, int pw, int bh, int bw, int ry, int rx) {
    if (ph == bh && pw == bw) return p[ry * pw + rx];
    const Taps ty = lin_taps(ry, ph, bh, false), tx = lin_taps(rx, pw, bw, true);
    const float r0 = lin_row(p[ty.s0 * pw + tx.s0], p[ty.s0 * pw + tx.s1], tx);
    const float r1 = lin_row(p[ty.s1 * pw + tx.s0], p[ty.s1 * pw + tx.s1], tx);
    const float a = r0 * ty.c0, b = r1 * ty.c1;
    return a + b;
}
__device__ __forceinline__ float pasted(const float* __restrict__ p, const PasteDet& d, int y, int x) {
    if (y < d.y1 || y >= d.y2 || x < d.x1 || x >= d.x2) return 0.f;
    return resized_patch(p, d.ph, d.pw, d.y2 - d.y1, d.x2 - d.x1, y - d.y1, x - d.x1);
}

template <typename OUT>
__global__ __launch_bounds__(256) void paste_kernel(const float* __restrict__ flat, const PasteDet* __restrict__ dets, int nd, int in_h, int in_w,
                                                    int out_h, int out_w, float thresh, OUT* __restrict__ out) {
    const long per = (long)out_h * out_w, total = per * nd;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i / per);
        const int rem = (int)(i - (long)k * per);
        const int oy = rem / out_w, ox = rem - oy * out_w;
        const PasteDet d = dets[k];
        const float* p = flat + d.off;
        float v;
        if (d.y2 <= d.y1 || d.x2 <= d.x1) v = 0.f;
        else if (in_h == out_h && in_w == out_w) v = pasted(p, d, oy, ox);
        else {
            const Taps ty = lin_taps(oy, in_h, out_h, false), tx = lin_taps(ox, in_w, out_w, true);
            if (ty.s1 < d.y1 || ty.s0 >= d.y2 || tx.s1 < d.x1 || tx.s0 >= d.x2) v = 0.f;
            else {
                const float r0 = lin_row(pasted(p, d, ty.s0, tx.s0), pasted(p, d, ty.s0, tx.s1), tx);
                const float r1 = lin_row(pasted(p, d, ty.s1, tx.s0), pasted(p, d, ty.s1, tx.s1), tx);
                const float a = r0 * ty.c0, b = r1 * ty.c1;
                v = a + b;
            }
        }
        out[i] = (OUT)(v >= thresh ? 1 : 0);
    }
}

// flat: fp32 patch probabilities (forward_seg's output buffer); dets: device int32 [nd][8] = {patch offset, patch h, patch w,
// y1, x1, y2, x2, 0} with the box already rounded / clamped as test.py:138-141; out: [nd][image_h][image_w], float32 (as the
// reference returns) when out_is_u8 == 0, bytes otherwise.
extern "C" int kg_mask_paste(const float* flat, const int* dets, int nd, int input_h, int input_w, int image_h, int image_w,
                             float seg_thresh, void* out, int out_is_u8, void* stream) {
    KG_CHECK_ARG(flat && dets && out && nd >= 0 && input_h > 0 && input_w > 0 && image_h > 0 && image_w > 0, "kg_mask_paste: bad arguments");
    if (nd == 0) return KG_OK;
    const long total = (long)nd * image_h * image_w;
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (out_is_u8)
        hipLaunchKernelGGL(paste_kernel<unsigned char>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flat, (const PasteDet*)dets, nd,
                           input_h, input_w, image_h, image_w, seg_thresh, (unsigned char*)out);
    else
        hipLaunchKernelGGL(paste_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flat, (const PasteDet*)dets, nd,
                           input_h, input_w, image_h, image_w, seg_thresh, (float*)out);
    KG_CHECK_LAUNCH("mask_paste");
    return KG_OK;
}

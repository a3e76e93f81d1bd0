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

// ---- SEG_loss target preparation on the device (second half: seg_loss.py:57-80) ---------------------------------------
// For callers whose ground-truth masks are device-resident (float32 [n_i][H][W] tensors on the GPU): the crop of the matched mask
// [y1:y2, x1:x2] nearest-resized to the predicted patch (h1, w1) -- cv2.resize(..., INTER_NEAREST) as restated by kg_host_crop_masks:
// src = min(floor(dst * src_size / dst_size), src_size - 1), identity when the sizes agree -- written as bytes at the pair's offset of
// the target buffer kg_seg_loss reads.  One workgroup per (pair, 1024-pixel slab); the same work rows as the host function.
__global__ __launch_bounds__(256) void crop_masks_kernel(const float* const* __restrict__ masks, const int* __restrict__ work, int nwork,
                                                         int H, int W, unsigned char* __restrict__ out) {
    const int k = blockIdx.x;
    const int* w = work + 9 * k;
    const float* m = masks[w[0]] + (long)w[1] * H * W;
    const int ya = w[2], xa = w[4];
    int yb = w[3], xb = w[5];
    const int h1 = w[6], w1 = w[7];
    if (yb > H) yb = H;
    if (xb > W) xb = W;
    const int h0 = yb - ya, w0 = xb - xa;
    if (h0 <= 0 || w0 <= 0) return;          // (rejected on the host before the launch)
    const bool same = h0 == h1 && w0 == w1;
    const double fy = (double)h0 / h1, fx = (double)w0 / w1;
    unsigned char* o = out + w[8];
    for (int i = blockIdx.y * 256 + threadIdx.x; i < h1 * w1; i += gridDim.y * 256) {
        const int y = i / w1, x = i - y * w1;
        int sy = y, sx = x;
        if (!same) {
            sy = (int)floor(y * fy); if (sy > h0 - 1) sy = h0 - 1;
            sx = (int)floor(x * fx); if (sx > w0 - 1) sx = w0 - 1;
        }
        o[i] = (unsigned char)m[(long)(ya + sy) * W + xa + sx];
    }
}
// masks: DEVICE array of nimg device pointers (float32 [n_i][H][W]); work: device int32 [nwork][9] rows (img, gt index, y1, y2, x1, x2,
// h1, w1, out offset) -- the rows kg_host_crop_masks takes; out: device bytes.
extern "C" int kg_crop_masks(const void* masks, const int* work, int nwork, int H, int W, void* out, void* stream) {
    KG_CHECK_ARG(masks && work && out && nwork >= 0 && H > 0 && W > 0, "kg_crop_masks: bad arguments");
    if (nwork == 0) return KG_OK;
    hipLaunchKernelGGL(crop_masks_kernel, dim3(nwork, 4), dim3(256), 0, (hipStream_t)stream, (const float* const*)masks, work, nwork, H, W,
                       (unsigned char*)out);
    KG_CHECK_LAUNCH("crop_masks");
    return KG_OK;
}

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

static thread_local const char* g_kernel = "";
void kg_note_kernel(const char* name) { g_kernel = name; }
// Name (as rocprofv3 prints it, without the argument list) of the conv-family kernel the calling thread's most recent kg_conv* / kg_conv2d_wgrad*
// call launched ("" before the first); helper launches (split reductions, finishing passes) do not change it.
extern "C" const char* kg_last_kernel(void) { return g_kernel; }
extern "C" int kg_version(void) { return 101; }
extern "C" int kg_rows_format(void) { return KG_ROWS_FORMAT; }    // 0: bfloat16 rows (libkgnet_hip.so), 1: IEEE half rows (libkgnet_hip_f16.so)

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
    bf16x4 v = KG_DS_READ_TR16((lds_bf16x4*)(&lds[l * 4]));
    *reinterpret_cast<uint2*>(&out[l * 4]) = __builtin_bit_cast(uint2, v);
}
extern "C" int kg_tr_probe(void* out, void* stream) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned short*)out);
    KG_CHECK_LAUNCH("tr_probe");
    return KG_OK;
}

// ---- BatchNorm statistics from the producing conv's epilogue (conv_args.h): arm -> conv launch -> read back the tile count.
static thread_local KgConvStats g_conv_stats = {nullptr, 0, 0, {nullptr, 0, 0, 0, nullptr, nullptr}};
KgConvStats& kg_conv_stats() { return g_conv_stats; }
extern "C" int kg_conv_stats_begin(float* part, long cap_floats) {
    KG_CHECK_ARG(part && cap_floats > 0, "kg_conv_stats_begin: bad buffer");
    g_conv_stats = KgConvStats{part, cap_floats, 0, {nullptr, 0, 0, 0, nullptr, nullptr}};
    return KG_OK;
}
// Backward variant: arms the next INPUT-GRADIENT launch (dense transposed mode of kg_conv2d_igemm, flip = 1 of kg_conv2d_halo, output channels a
// multiple of 64) to write partials [nb][C][2] = per-tile sums of (g, g * xhat) over the gradient rows it stores -- g after the residual add and
// the ReLU mask of its epilogue, xhat = (x - mean[c]) * invstd[c] of the rows x (x_planes planes, x_pstride elements apart).  kg_bn_bwd takes them
// in place of its own column reduction (parts / nb arguments).  kg_conv_stats_end reads the tile count back and disarms, as for the forward.
extern "C" int kg_conv_bstats_begin(float* part, long cap_floats, const void* x, int ldx, int x_planes, int x_pstride, const float* mean,
                                    const float* invstd) {
    KG_CHECK_ARG(part && cap_floats > 0 && x && mean && invstd && ldx % 8 == 0 && x_planes >= 1 && x_planes <= 3 && x_pstride % 8 == 0,
                 "kg_conv_bstats_begin: bad arguments");
    g_conv_stats = KgConvStats{part, cap_floats, 0, {(const unsigned short*)x, ldx, x_planes, x_pstride, mean, invstd}};
    return KG_OK;
}
// *nb = pixel tiles whose partials [nb][Cout][2] the conv wrote (0: the launch took a kernel without the statistics epilogue, or
// none ran -- the caller falls back to kg_bn_stats_train); disarms the channel.
extern "C" int kg_conv_stats_end(int* nb) {
    KG_CHECK_ARG(nb, "kg_conv_stats_end: null pointer");
    *nb = g_conv_stats.nb;
    g_conv_stats = KgConvStats{nullptr, 0, 0, {nullptr, 0, 0, 0, nullptr, nullptr}};
    return KG_OK;
}

// ---- host glue of the per-box seg branch (KGnet.py:258-267, 321-350): box tables of seg.make_plan.  Pure host code, integer
// arithmetic; 10 tile tables + 5 bin lists per step cost 2.2 ms as NumPy repeat / argsort chains for 2400 boxes.
// kg_host_tile_table: one {row0, (h << 16) | w, (oy0 << 16) | ox0, 0} entry per th x tw tile of every box, box-major, tile rows
// first; returns the number of entries (the caller sizes `out` from its own count), -1 on overflow of cap.
extern "C" int kg_host_tile_table(const int* h, const int* w, const long* row0, int nb, int th, int tw, int* out, int cap) {
    if (!h || !w || !row0 || !out || nb < 0 || th < 1 || tw < 1) return -1;
    int n = 0;
    for (int b = 0; b < nb; ++b) {
        const int ny = (h[b] + th - 1) / th, nx = (w[b] + tw - 1) / tw;
        for (int ty = 0; ty < ny; ++ty)
            for (int tx = 0; tx < nx; ++tx) {
                if (n >= cap) return -1;
                int* o = out + 4 * (long)n++;
                o[0] = (int)row0[b]; o[1] = (h[b] << 16) | w[b]; o[2] = ((ty * th) << 16) | (tx * tw); o[3] = 0;
            }
    }
    return n;
}
// kg_host_bin_csr: CSR lists of the boxes touching each BS x BS bin of each image (the deterministic crop-gradient reduction,
// kg_crop_grad_reduce): tab = int32 [nb][8] rows {img, y1, x1, h, w, ...} of one pyramid level; bin index = (img * BY + by) * BX + bx;
// bin_start [nbins + 1], bin_boxes in ascending box order inside a bin.  Returns the number of (box, bin) incidences, -1 on overflow.
extern "C" int kg_host_bin_csr(const int* tab, int nb, int BS, int BY, int BX, int nbins, int* bin_start, int* bin_boxes, int cap) {
    if (!tab || !bin_start || (!bin_boxes && cap > 0) || nb < 0 || BS < 1 || nbins < 0) return -1;
    for (int i = 0; i <= nbins; ++i) bin_start[i] = 0;
    long total = 0;
    for (int b = 0; b < nb; ++b) {
        const int* t = tab + 8 * (long)b;
        const int by0 = t[1] / BS, bx0 = t[2] / BS, by1 = (t[1] + t[3] - 1) / BS, bx1 = (t[2] + t[4] - 1) / BS;
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) {
                const long bin = ((long)t[0] * BY + by) * BX + bx;
                if (bin < 0 || bin >= nbins) return -1;
                ++bin_start[bin + 1]; ++total;
            }
    }
    if (total > cap) return -1;
    for (int i = 0; i < nbins; ++i) bin_start[i + 1] += bin_start[i];
    // fill: a cursor per bin (reuses bin_start shifted by one step: fill from the front, then restore)
    for (int b = 0; b < nb; ++b) {
        const int* t = tab + 8 * (long)b;
        const int by0 = t[1] / BS, bx0 = t[2] / BS, by1 = (t[1] + t[3] - 1) / BS, bx1 = (t[2] + t[4] - 1) / BS;
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx) {
                const long bin = ((long)t[0] * BY + by) * BX + bx;
                bin_boxes[bin_start[bin]++] = b;
            }
    }
    for (int i = nbins; i > 0; --i) bin_start[i] = bin_start[i - 1];     // cursors ended at the next bin's start: shift back
    bin_start[0] = 0;
    return (int)total;
}

// ---- host glue of SEG_loss (seg_loss.py:14-29, 55-56): which (predicted box, ground-truth box) pairs overlap with IoU >= thresh.
// float32 arithmetic in the operation order of the reference's jaccard_numpy (areas, clamped intersection sides, union <= 2 -> 0).
// pb = float32 [P][4], gb = float32 [G][gstride] (first 4 columns y1, x1, y2, x2); pairs = int32 [cap][2] receives (patch, gt) in
// row-major order; *count = number of matches (the call fails when they do not fit cap).  300 x 300 boxes: 0.3 ms as NumPy
// broadcasting per image, ~40 us here.
extern "C" int kg_host_match_boxes(const float* pb, int P, const float* gb, int G, int gstride, float thresh, int* pairs, int cap, int* count) {
#pragma clang fp contract(off)
    KG_CHECK_ARG(pb && gb && pairs && count && P >= 0 && G >= 0 && gstride >= 4 && cap >= 0, "kg_host_match_boxes: bad arguments");
    // ground-truth boxes as structure of arrays: the overlap pre-test below is a branch-free loop the host compiler vectorises, and
    // almost every pair is disjoint (IoU exactly 0 < thresh); the exact float32 IoU runs only on the overlapping ones
    float* soa = (float*)malloc(sizeof(float) * 4 * (size_t)(G > 0 ? G : 1));
    unsigned char* ov = (unsigned char*)malloc((size_t)(G > 0 ? G : 1));
    if (!soa || !ov) { free(soa); free(ov); kg_set_error("kg_host_match_boxes: out of host memory"); return KG_ERR_ARG; }
    float *b0 = soa, *b1 = soa + G, *b2 = soa + 2 * (size_t)G, *b3 = soa + 3 * (size_t)G;
    for (int g = 0; g < G; ++g) {
        const float* b = gb + (long)g * gstride;
        b0[g] = b[0]; b1[g] = b[1]; b2[g] = b[2]; b3[g] = b[3];
    }
    const bool prefilter = thresh > 0.f;
    int n = 0, rc = KG_OK;
    for (int j = 0; j < P && rc == KG_OK; ++j) {
        const float a0 = pb[4 * j], a1 = pb[4 * j + 1], a2 = pb[4 * j + 2], a3 = pb[4 * j + 3];
        const float area_a = (a2 - a0) * (a3 - a1);
        for (int g = 0; g < G; ++g) {
            const float ih = (a2 < b2[g] ? a2 : b2[g]) - (a0 > b0[g] ? a0 : b0[g]);
            const float iw = (a3 < b3[g] ? a3 : b3[g]) - (a1 > b1[g] ? a1 : b1[g]);
            ov[g] = (unsigned char)((ih > 0.f) & (iw > 0.f));
        }
        for (int g = 0; g < G; ++g) {
            if (prefilter && !ov[g]) continue;
            const float area_b = (b2[g] - b0[g]) * (b3[g] - b1[g]);
            float ih = (a2 < b2[g] ? a2 : b2[g]) - (a0 > b0[g] ? a0 : b0[g]);
            float iw = (a3 < b3[g] ? a3 : b3[g]) - (a1 > b1[g] ? a1 : b1[g]);
            if (!(ih > 0.f)) ih = 0.f;
            if (!(iw > 0.f)) iw = 0.f;
            const float inter = ih * iw;
            const float uni = area_a + area_b - inter;
            const float iou = uni <= 2.f ? 0.f : inter / uni;
            if (iou >= thresh) {
                if (n >= cap) { kg_set_error("kg_host_match_boxes: more than %d matches", cap); rc = KG_ERR_ARG; break; }
                pairs[2 * n] = j; pairs[2 * n + 1] = g; ++n;
            }
        }
    }
    free(soa); free(ov);
    if (rc != KG_OK) return rc;
    *count = n;
    return KG_OK;
}

// ---- host glue of SEG_loss (seg_loss.py:57-80): crops of the matched ground-truth masks, nearest-resized to the patch size,
// written as bytes into one (pinned) staging buffer.  Pure host code: 2400 crops per step cost 7 ms as a Python loop.
// masks[i] = float32 [n_i][H][W] (C order); work = int32 [nwork][9] rows (img, gt index, y1, y2, x1, x2, h1, w1, out offset):
// out[off + y*w1 + x] = (uint8) masks[img][g][y1 + sy][x1 + sx] with sy = min(floor(y * (y2-y1)/h1), y2-y1-1) (cv2 INTER_NEAREST rule
// as stated in seg_loss.nearest_resize), identity when the crop already has the patch size.
#include <math.h>
extern "C" int kg_host_crop_masks(const float* const* masks, const int* work, int nwork, int H, int W, unsigned char* out) {
    KG_CHECK_ARG(masks && work && out && nwork >= 0 && H > 0 && W > 0, "kg_host_crop_masks: bad arguments");
    for (int k = 0; k < nwork; ++k) {
        const int* w = work + 9 * k;
        const float* m = masks[w[0]] + (long)w[1] * H * W;
        int ya = w[2], yb = w[3], xa = w[4], xb = w[5];
        const int h1 = w[6], w1 = w[7];
        unsigned char* o = out + w[8];
        if (yb > H) yb = H;
        if (xb > W) xb = W;
        const int h0 = yb - ya, w0 = xb - xa;
        if (h0 <= 0 || w0 <= 0) { kg_set_error("kg_host_crop_masks: empty ground-truth crop (work item %d)", k); return KG_ERR_ARG; }
        const double fy = (double)h0 / h1, fx = (double)w0 / w1;
        for (int y = 0; y < h1; ++y) {
            int sy = y;
            if (h0 != h1 || w0 != w1) { sy = (int)floor(y * fy); if (sy > h0 - 1) sy = h0 - 1; }
            const float* row = m + (long)(ya + sy) * W + xa;
            for (int x = 0; x < w1; ++x) {
                int sx = x;
                if (h0 != h1 || w0 != w1) { sx = (int)floor(x * fx); if (sx > w0 - 1) sx = w0 - 1; }
                o[y * w1 + x] = (unsigned char)row[sx];
            }
        }
    }
    return KG_OK;
}

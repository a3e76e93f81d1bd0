// preproc.hip -- ground-truth map generation of one pyramid scale on the GPU.
//
// Reference: preprocessing.get_ground_truth (preprocessing.py:107-118) = load_disc_masks :63-77, load_kp_heats :79-85,
// compute_short_offsets :45-60 (+ copy_with_border_check :13-43), compute_mid_offsets :88-105, assembled as
// dataset_base.py:99 (kp 5 | short 10 | mid 40 channels) and cast to float32 (:109).  The NumPy code is
// O(instances x H x W x keypoints) with an H x W x instances float64 temporary: 58 s per 512x512 image with 300
// instances; here it is one thread per (pixel, keypoint type), the instance keypoints of that type staged in LDS.
//
// Exact semantics kept (all in float64, this file is compiled with -ffp-contract=off):
//  * owner(i, pixel) = first instance with the minimal sqrt((kx-x)^2 + (ky-y)^2), valid if that distance <= 5
//    (np.argmin tie rule).  Instances with |kx-x| > 5 or |ky-y| > 5 cannot be within the radius and are skipped: if any
//    instance is within the radius the argmin is one of the instances that are.
//  * short offsets: the LAST instance whose (2R+1)^2 window around (int(kx), int(ky)) covers the pixel writes
//    (cx-x, cy-y) inside the circle and 0 in the window's corners (the disc mask is never applied in the reference).
//  * mid offsets of directed edge (a, b): owner(a)'s keypoint b minus the pixel position.
#include "kg_common.h"

#define KG_GT_R 5
#define KG_GT_CHUNK 1024

__global__ __launch_bounds__(256) void gt_maps_kernel(const float* __restrict__ kps, int n, int H, int W, float* __restrict__ out) {
    __shared__ float sk[KG_GT_CHUNK * 2];
    __shared__ int cand[KG_GT_CHUNK];
    __shared__ int ncand;
    const int i = blockIdx.y;                               // keypoint type
    const long hw = (long)H * W;
    const long p0 = (long)blockIdx.x * 256, p = p0 + threadIdx.x;
    const bool live = p < hw;
    const int y = live ? (int)(p / W) : 0, x = live ? (int)(p - (long)y * W) : 0;
    // bounding box of the workgroup's 256 consecutive pixels: instances farther than the radius (+1 for int() truncation)
    // from it can neither own nor overwrite any of them and are culled once per workgroup
    const long pl = p0 + 255 < hw ? p0 + 255 : hw - 1;
    const int by0 = (int)(p0 / W), by1 = (int)(pl / W);
    const int bx0 = by0 == by1 ? (int)(p0 - (long)by0 * W) : 0, bx1 = by0 == by1 ? (int)(pl - (long)by1 * W) : W - 1;
    int owner = -1, last = -1;
    double best = 0.0;
    for (int j0 = 0; j0 < n; j0 += KG_GT_CHUNK) {
        const int m = n - j0 < KG_GT_CHUNK ? n - j0 : KG_GT_CHUNK;
        __syncthreads();
        if (threadIdx.x == 0) ncand = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < m; t += 256) {
            const float kx = kps[((long)(j0 + t) * 5 + i) * 2], ky = kps[((long)(j0 + t) * 5 + i) * 2 + 1];
            sk[2 * t] = kx; sk[2 * t + 1] = ky;
            if (ky >= (float)(by0 - KG_GT_R - 1) && ky <= (float)(by1 + KG_GT_R + 1) && kx >= (float)(bx0 - KG_GT_R - 1) &&
                kx <= (float)(bx1 + KG_GT_R + 1))
                cand[atomicAdd(&ncand, 1)] = t;             // unordered: the per-pixel rules below do not depend on the order
        }
        __syncthreads();
        if (!live) continue;
        const int nc = ncand;
        for (int q = 0; q < nc; ++q) {
            const int t = cand[q], j = j0 + t;
            const float kx = sk[2 * t], ky = sk[2 * t + 1];
            const int cx = (int)kx, cy = (int)ky;            // int(center) of copy_with_border_check
            const int wx = x - cx, wy = y - cy;
            if (wx >= -KG_GT_R && wx <= KG_GT_R && wy >= -KG_GT_R && wy <= KG_GT_R && j > last) last = j;   // last writer
            const double dx = (double)kx - (double)x, dy = (double)ky - (double)y;
            if (dx > KG_GT_R || dx < -KG_GT_R || dy > KG_GT_R || dy < -KG_GT_R) continue;
            const double d = sqrt(dx * dx + dy * dy);
            if (d <= (double)KG_GT_R && (owner < 0 || d < best || (d == best && j < owner))) { owner = j; best = d; }   // first argmin
        }
    }
    if (!live) return;
    out[(long)i * hw + p] = owner >= 0 ? 1.f : 0.f;
    float sx = 0.f, sy = 0.f;
    if (last >= 0) {
        const int cx = (int)kps[((long)last * 5 + i) * 2], cy = (int)kps[((long)last * 5 + i) * 2 + 1];
        const int ox = cx - x, oy = cy - y;
        if (sqrt((double)(ox * ox + oy * oy)) <= (double)KG_GT_R) { sx = (float)ox; sy = (float)oy; }
    }
    out[(5 + 2 * i) * hw + p] = sx;
    out[(5 + 2 * i + 1) * hw + p] = sy;
    // directed edges (EDGES + reversed, config.py:2-13) that start at keypoint type i
    const int edges[20][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {1, 2}, {1, 3}, {1, 4}, {2, 3}, {2, 4}, {3, 4},
                              {1, 0}, {2, 0}, {3, 0}, {4, 0}, {2, 1}, {3, 1}, {4, 1}, {3, 2}, {4, 2}, {4, 3}};
#pragma unroll
    for (int e = 0; e < 20; ++e) {
        if (edges[e][0] != i) continue;
        float mx = 0.f, my = 0.f;
        if (owner >= 0) {
            const int b = edges[e][1];
            mx = (float)((double)kps[((long)owner * 5 + b) * 2] - (double)x);
            my = (float)((double)kps[((long)owner * 5 + b) * 2 + 1] - (double)y);
        }
        out[(15 + 2 * e) * hw + p] = mx;
        out[(15 + 2 * e + 1) * hw + p] = my;
    }
}

// kps: device float32 [n][5][2] (x, y) keypoints tl, tr, bl, br, centre of every instance (dataset_base.py:58-79);
// out: device float32 [55][H][W] (dataset_base.py:99-109).  n == 0 gives all-zero maps (preprocessing.py:108-112).
extern "C" int kg_gt_maps(const float* kps, int n, int H, int W, float* out, void* stream) {
    KG_CHECK_ARG(out && H > 0 && W > 0 && n >= 0 && (n == 0 || kps), "kg_gt_maps: bad arguments");
    const long hw = (long)H * W;
    hipLaunchKernelGGL(gt_maps_kernel, dim3((unsigned)((hw + 255) / 256), 5), dim3(256), 0, (hipStream_t)stream, kps, n, H, W, out);
    KG_CHECK_LAUNCH("gt_maps");
    return KG_OK;
}

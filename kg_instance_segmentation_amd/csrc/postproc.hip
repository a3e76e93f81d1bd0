// postproc.hip -- KGnet post-processing on gfx950 in float64, bit-identical to the reference's
// NumPy/SciPy path (postprocessing.py:16-261, nms.py:4-53).  Compiled with -ffp-contract=off: every
// multiply/add below rounds exactly as the reference's separate NumPy operations do; the single
// fused operation of the reference (np.linalg.norm -> ddot) is written as an explicit fma().
//
// Stages (all on the caller's stream, no host sync inside):
//   P1 Hough vote   : one scatter pass with inline slots per cell -> per-cell ordered sum (reproduces the
//                     sequential COO scatter order: corner block, then raster index)
//   P2 Gaussian     : separable 17-tap, scipy's symmetric pairing order, 'reflect' border
//   P3 peaks        : cross-footprint local max == value, > thresh, ordered compaction
//   P4 grouping     : greedy confidence-ordered star matching, one workgroup (4 waves = 4 edges)
//   P6/P7 boxes     : refine + 8-case box assembly, ordered compaction, 4 scales appended
//   P9 NMS          : greedy IoU suppression, one workgroup
#include "kg_common.h"

#define KG_NUM_KPS 5

__device__ __forceinline__ int f2i_np(double v) {  // numpy f64 -> int32 cast on x86 (cvttsd2si)
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int)v;
}

// One vote contribution of source pixel i (corner b) of channel c.  Returns false when dropped.
__device__ __forceinline__ bool hough_contrib(const float* __restrict__ kp, const float* __restrict__ soff, int H,
                                              int W, int c, int b, int i, int* cell, double* val) {
    const long HW = (long)H * W;
    const int y = i / W, x = i - y * W;
    const double xs = (double)x + (double)soff[(long)(2 * c) * HW + i];
    const double ys = (double)y + (double)soff[(long)(2 * c + 1) * HW + i];
    const double ps = (double)kp[(long)c * HW + i];
    const int fx = f2i_np(floor(xs)), fy = f2i_np(floor(ys));
    const int cx = f2i_np(ceil(xs)), cy = f2i_np(ceil(ys));
    const double dx = xs - (double)fx, dy = ys - (double)fy;
    int I, J; double v;
    switch (b) {
        case 0: I = fy; J = fx; v = ps * (1. - dx) * (1. - dy); break;
        case 1: I = fy; J = cx; v = ps * dx * (1. - dy); break;
        case 2: I = cy; J = fx; v = ps * dy * (1. - dx); break;
        default: I = cy; J = cx; v = ps * dy * dx; break;
    }
    // out of the map (postprocessing.py:34-35), or a zero addend: every cell's sum starts from +0.0 and x + (+-0.0) == x for every x that can stand
    // in such a sum (+0.0 + -0.0 == +0.0 as well), so dropping the vote keeps the heat map's bits -- with integer-valued offsets three of the four
    // bilinear weights are exactly 0, with kp == 0 all four (a NaN is kept)
    if (I < 0 || I >= H || J < 0 || J >= W || v == 0.) return false;
    *cell = I * W + J; *val = v;
    return true;
}

// The four votes of source pixel i = (y, x) of channel c at once (the same expressions as hough_contrib, evaluated once): target (I, J), value
// and validity per corner b (order tl, tr, bl, br = the reference's concatenation order, postprocessing.py:24-33).
struct HVote4 { int I[4], J[4]; double v[4]; bool ok[4]; };
__device__ __forceinline__ void hough_vote4(const float* __restrict__ kp, const float* __restrict__ soff, int H, int W, int c, int i, int y, int x,
                                            HVote4& o) {
    const long HW = (long)H * W;
    const double xs = (double)x + (double)soff[(long)(2 * c) * HW + i];
    const double ys = (double)y + (double)soff[(long)(2 * c + 1) * HW + i];
    const double ps = (double)kp[(long)c * HW + i];
    const int fx = f2i_np(floor(xs)), fy = f2i_np(floor(ys));
    const int cx = f2i_np(ceil(xs)), cy = f2i_np(ceil(ys));
    const double dx = xs - (double)fx, dy = ys - (double)fy;
    o.I[0] = fy; o.J[0] = fx; o.v[0] = ps * (1. - dx) * (1. - dy);
    o.I[1] = fy; o.J[1] = cx; o.v[1] = ps * dx * (1. - dy);
    o.I[2] = cy; o.J[2] = fx; o.v[2] = ps * dy * (1. - dx);
    o.I[3] = cy; o.J[3] = cx; o.v[3] = ps * dy * dx;
#pragma unroll
    for (int b = 0; b < 4; ++b) o.ok[b] = !(o.I[b] < 0 || o.I[b] >= H || o.J[b] < 0 || o.J[b] >= W || o.v[b] == 0.);
}

// exclusive scan of count[c][0..HW) -> offs; one block (1024 threads) per channel.
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ count, int* __restrict__ offs, int n, int* __restrict__ total = nullptr) {
    __shared__ int tot[1024];
    const int* in = count + (long)blockIdx.x * n;
    int* out = offs + (long)blockIdx.x * n;
    const int per = (n + 1023) / 1024, b0 = threadIdx.x * per;
    int s = 0;
    for (int i = b0; i < b0 + per && i < n; ++i) s += in[i];
    tot[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int v = threadIdx.x >= d ? tot[threadIdx.x - d] : 0;
        __syncthreads();
        tot[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x ? tot[threadIdx.x - 1] : 0;
    for (int i = b0; i < b0 + per && i < n; ++i) { int v = in[i]; out[i] = run; run += v; }
    if (total && threadIdx.x == 1023) total[blockIdx.x] = tot[1023];      // sum of the whole row (the peak count: no separate launch)
}
// ---- P1: one scatter pass with inline slots -----------------------------------------------------------------------------------------
// (Round 1's count -> scan -> fill -> sum pipeline touched every vote with two atomic passes, computed every contribution twice and sorted
// every cell's votes by selection from global memory; it is gone from the sources since round 5 -- git history, docs/history.md.)
//   scatter : one thread per vote (postprocessing.py:16-37); zero addends are dropped (hough_contrib).  Vote e = corner * HW + pixel takes slot
//             s = atomicAdd(cnt[cell]) (ONE returning atomic); the first HOUGH_K votes of a cell land in its inline slots [cell][HOUGH_K] (key,
//             value); a later one is parked at ITS OWN index of a source-ordered array (ovcell[e] = cell, ovval[e] = value; ovcell[e] = -1 for
//             every other vote: no counter, no list).  (Round 5 measured two alternatives, both SLOWER and not in the tree: one thread per source
//             pixel with its four atomics back to back, 1.3 -> 1.9 ms at 1024 x 1024 -- same-line returning atomics queue up in the L2 --, and a
//             compact parked list behind a wave-aggregated counter, 3.1 -> 3.5 ms of the Hough phase.)
//   classify: one thread per cell.  n <= HOUGH_K: the votes are sorted by key in registers (odd-even merge network) and summed in
//             that order -- the reference's sequential COO order (postprocessing.py:36) -- and the cell is done.  Heavier cells get
//             a slab of n entries in the compact arrays (one 64-bit atomic per 1024-thread workgroup allocates for all its heavy
//             cells), copy their inline votes to its head and join the heavy list;
//   ovfill  : the parked votes move behind the inline ones of their cell's slab (atomic cursor per heavy cell);
//   heavy   : one wave per heavy cell: rank sort by key in LDS (n <= HOUGH_LCAP; through global scratch beyond), lane 0 adds in order.
// Every sum is still formed in increasing key order from +0.0 with separate fp64 adds (-ffp-contract=off): bit-identical heat maps.
#define HOUGH_K 8
// hdr[1]: raised by hough_far_kernel (far list overflow) -- complete before hough_tile_kernel starts, so that kernel's entry test is
// workgroup-uniform; hdr[2]: raised by a tile workgroup whose votes exceed its LDS slabs -- never read by the tile kernel itself (a sibling
// workgroup's late write would otherwise let the waves of one workgroup diverge around its barriers).
__device__ __forceinline__ bool hough_gave_up(const int* hdr) { return (hdr[1] | hdr[2]) != 0; }
#define HOUGH_LCAP 512
__global__ void hough_scatter_kernel(const float* __restrict__ kp, const float* __restrict__ soff, int H, int W, int* __restrict__ cnt,
                                     unsigned* __restrict__ ink, double* __restrict__ inv, int* __restrict__ ovcell,
                                     double* __restrict__ ovval, const int* __restrict__ hdr) {
    if (!hough_gave_up(hdr)) return;             // (the tile formulation produced the map)
    const int c = blockIdx.y, HW = H * W;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 4 * HW; e += gridDim.x * blockDim.x) {
        const int b = e / HW, i = e - b * HW;
        int cell; double v;
        int park = -1;
        if (hough_contrib(kp, soff, H, W, c, b, i, &cell, &v)) {
            const int cc = c * HW + cell;
            const int slot = atomicAdd(&cnt[cc], 1);
            if (slot < HOUGH_K) {
                ink[(long)cc * HOUGH_K + slot] = (unsigned)e;
                inv[(long)cc * HOUGH_K + slot] = v;
            } else {
                park = cc;
                ovval[(long)c * 4 * HW + e] = v;
            }
        }
        ovcell[(long)c * 4 * HW + e] = park;
    }
}
__device__ __forceinline__ void hough_cswap(unsigned& ka, double& va, unsigned& kb, double& vb) {
    const bool sw = kb < ka;
    const unsigned k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const double v0 = sw ? vb : va, v1 = sw ? va : vb;
    ka = k0; kb = k1; va = v0; vb = v1;
}
// ctr64: low 32 bits = entries allocated in the compact arrays, high 32 bits = heavy cells
__global__ __launch_bounds__(1024) void hough_classify_kernel(int ncells, const int* __restrict__ cnt, const unsigned* __restrict__ ink,
                                                              const double* __restrict__ inv, double norm, double* __restrict__ heat,
                                                              unsigned long long* __restrict__ ctr64, int* __restrict__ ovoff,
                                                              int* __restrict__ heavy_list, unsigned* __restrict__ skey,
                                                              double* __restrict__ sval, const int* __restrict__ hdr) {
    if (!hough_gave_up(hdr)) return;
    __shared__ int s_need[1024], s_hv[1024];
    __shared__ unsigned long long s_base;
    const int cc = blockIdx.x * 1024 + threadIdx.x;
    const int n = cc < ncells ? cnt[cc] : 0;
    unsigned k[HOUGH_K]; double v[HOUGH_K];
    if (n > 0) {
        const uint4 k0 = *reinterpret_cast<const uint4*>(ink + (long)cc * HOUGH_K);
        const uint4 k1 = n > 4 ? *reinterpret_cast<const uint4*>(ink + (long)cc * HOUGH_K + 4) : make_uint4(0, 0, 0, 0);
        k[0] = k0.x; k[1] = k0.y; k[2] = k0.z; k[3] = k0.w; k[4] = k1.x; k[5] = k1.y; k[6] = k1.z; k[7] = k1.w;
#pragma unroll
        for (int q = 0; q < HOUGH_K / 2; ++q) {
            if (2 * q < n) { const double2 t = *reinterpret_cast<const double2*>(inv + (long)cc * HOUGH_K + 2 * q); v[2 * q] = t.x; v[2 * q + 1] = t.y; }
            else { v[2 * q] = 0.; v[2 * q + 1] = 0.; }
        }
    }
    const bool heavy = n > HOUGH_K;
    if (n > 0 && !heavy) {
#pragma unroll
        for (int e = 0; e < HOUGH_K; ++e)
            if (e >= n) k[e] = 0xffffffffu;                      // (keys are < 4 H W <= 2^28: padding sorts to the end and is never added)
        // odd-even merge sort network for 8 keys (19 comparators)
        hough_cswap(k[0], v[0], k[1], v[1]); hough_cswap(k[2], v[2], k[3], v[3]); hough_cswap(k[4], v[4], k[5], v[5]); hough_cswap(k[6], v[6], k[7], v[7]);
        hough_cswap(k[0], v[0], k[2], v[2]); hough_cswap(k[1], v[1], k[3], v[3]); hough_cswap(k[4], v[4], k[6], v[6]); hough_cswap(k[5], v[5], k[7], v[7]);
        hough_cswap(k[1], v[1], k[2], v[2]); hough_cswap(k[5], v[5], k[6], v[6]);
        hough_cswap(k[0], v[0], k[4], v[4]); hough_cswap(k[1], v[1], k[5], v[5]); hough_cswap(k[2], v[2], k[6], v[6]); hough_cswap(k[3], v[3], k[7], v[7]);
        hough_cswap(k[2], v[2], k[4], v[4]); hough_cswap(k[3], v[3], k[5], v[5]);
        hough_cswap(k[1], v[1], k[2], v[2]); hough_cswap(k[3], v[3], k[4], v[4]); hough_cswap(k[5], v[5], k[6], v[6]);
        double s = 0.;
#pragma unroll
        for (int e = 0; e < HOUGH_K; ++e)
            if (e < n) s += v[e];
        heat[cc] = s / norm;
    } else if (cc < ncells && n == 0) {
        heat[cc] = 0. / norm;
    }
    // slabs of the heavy cells of this workgroup: exclusive scans over (entries needed, heavy flag), one 64-bit atomic
    s_need[threadIdx.x] = heavy ? n : 0;
    s_hv[threadIdx.x] = heavy ? 1 : 0;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int a = threadIdx.x >= d ? s_need[threadIdx.x - d] : 0, b = threadIdx.x >= d ? s_hv[threadIdx.x - d] : 0;
        __syncthreads();
        s_need[threadIdx.x] += a; s_hv[threadIdx.x] += b;
        __syncthreads();
    }
    if (threadIdx.x == 1023) {
        const unsigned long long add = ((unsigned long long)(unsigned)s_hv[1023] << 32) | (unsigned)s_need[1023];
        s_base = add ? atomicAdd(ctr64, add) : 0ull;
    }
    __syncthreads();
    if (heavy) {
        const int off = (int)(unsigned)(s_base & 0xffffffffull) + s_need[threadIdx.x] - n;
        const int hi = (int)(unsigned)(s_base >> 32) + s_hv[threadIdx.x] - 1;
        ovoff[cc] = off;
        heavy_list[hi] = cc;
#pragma unroll
        for (int e = 0; e < HOUGH_K; ++e) { skey[off + e] = k[e]; sval[off + e] = v[e]; }
    }
}
__global__ void hough_ovfill_kernel(long nvotes, int HW4, const int* __restrict__ ovcell, const double* __restrict__ ovval,
                                    const int* __restrict__ ovoff, int* __restrict__ ovcur, unsigned* __restrict__ skey,
                                    double* __restrict__ sval, const int* __restrict__ hdr) {
    if (!hough_gave_up(hdr)) return;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvotes; i += (long)gridDim.x * blockDim.x) {
        const int cc = ovcell[i];
        if (cc < 0) continue;
        const int pos = ovoff[cc] + HOUGH_K + atomicAdd(&ovcur[cc], 1);
        skey[pos] = (unsigned)(i % HW4);
        sval[pos] = ovval[i];
    }
}
__global__ __launch_bounds__(64) void hough_heavy2_kernel(const int* __restrict__ cnt, const int* __restrict__ ovoff,
                                                          const unsigned* __restrict__ skey, const double* __restrict__ sval,
                                                          double* __restrict__ srt, double norm, double* __restrict__ heat,
                                                          const unsigned long long* __restrict__ ctr64, const int* __restrict__ heavy_list,
                                                          const int* __restrict__ hdr) {
    if (!hough_gave_up(hdr)) return;
    __shared__ unsigned lk[HOUGH_LCAP];
    __shared__ double lv[HOUGH_LCAP];
    const int nh = (int)(unsigned)(*ctr64 >> 32);
    for (int h = blockIdx.x; h < nh; h += gridDim.x) {
        const int cc = heavy_list[h];
        const int n = cnt[cc], base = ovoff[cc];
        if (n <= HOUGH_LCAP) {
            for (int e = threadIdx.x; e < n; e += 64) lk[e] = skey[base + e];
            __syncthreads();
            for (int e = threadIdx.x; e < n; e += 64) {
                const unsigned ke = lk[e];
                int rank = 0;
                for (int j = 0; j < n; ++j) rank += lk[j] < ke;
                lv[rank] = sval[base + e];
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                double s = 0.;
                for (int t = 0; t < n; ++t) s += lv[t];
                heat[cc] = s / norm;
            }
            __syncthreads();
        } else {
            for (int e = threadIdx.x; e < n; e += 64) {
                const unsigned ke = skey[base + e];
                int rank = 0;
                for (int j = 0; j < n; ++j) rank += skey[base + j] < ke;
                srt[base + rank] = sval[base + e];
            }
            __threadfence_block();
            __syncthreads();
            if (threadIdx.x == 0) {
                double s = 0.;
                for (int t = 0; t < n; ++t) s += srt[base + t];
                heat[cc] = s / norm;
            }
            __syncthreads();
        }
    }
}

// ---- P1, tile formulation (round 5): the same per-cell ordered sums with NO global atomics ---------------------------------------------
// The scatter formulation above pays one returning global atomic per vote (21 M at 1024 x 1024: 1.3 ms) plus 480 MB of inline slots.  Votes are
// local: a trained network's short offsets stay within the keypoint radius (5 px), so the votes of a 32 x 32-cell output tile come from the
// 64 x 64 source pixels around it.  One workgroup per (tile, channel):
//   pass A  every source pixel of the 64 x 64 region computes its four votes; the ones that land in the tile count up cnt[cell] (LDS atomic);
//   scan    exclusive scan of the 1024 counters -> per-cell slabs in ONE compact LDS array (<= HT_CAP votes per tile);
//   pass B  the votes again (kept in registers: 16 per thread), now stored at slab[off[cell] + cur[cell]++] as (key, value);
//   sums    one thread per cell with <= 8 votes: the register sorting network + ordered sum of the scatter formulation; heavier cells: one wave
//           per cell ranks the slab's keys and lane 0 adds the values in rank order (through an LDS scratch, HT_SCR ranks at a time).
// A vote whose source lies OUTSIDE its target tile's region ("far": |offset| beyond 16 .. 48 px) is found by a pre-pass over the pixels
// (hough_far_kernel: same integer test) and listed globally; every tile workgroup also scans that list.  More than HOUGH_FARCAP far votes (a
// random-init network: offsets of hundreds of pixels) or more than HT_CAP votes in one tile raise hdr[1] / hdr[2] and the scatter formulation runs
// instead -- its kernels return at once while both are 0.  Same keys, same order, same fp64 adds: the same bits.
#define HT_T 32
#define HT_R 16
#define HT_S (HT_T + 2 * HT_R)
#define HT_CAP 8192
#define HT_SCR 320
#define HOUGH_FARCAP 16384
__device__ __forceinline__ bool hough_is_far(int y, int x, int I, int J) {
    const int ry = y - ((I / HT_T) * HT_T - HT_R), rx = x - ((J / HT_T) * HT_T - HT_R);
    return (unsigned)ry >= (unsigned)HT_S || (unsigned)rx >= (unsigned)HT_S;
}
__global__ __launch_bounds__(256) void hough_far_kernel(const float* __restrict__ kp, const float* __restrict__ soff, int H, int W, int* __restrict__ hdr,
                                                        int* __restrict__ farcell, unsigned* __restrict__ farkey, double* __restrict__ farval,
                                                        int4* __restrict__ clr, long clr_n4) {
    const int c = blockIdx.y, HW = H * W;
    // (also: zero the vote counters / slab cursors of the scatter formulation, should the tile formulation give up -- 15 HW bytes, no launch of its own)
    for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < clr_n4; i += (long)gridDim.x * gridDim.y * blockDim.x)
        clr[i] = make_int4(0, 0, 0, 0);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        HVote4 o;
        hough_vote4(kp, soff, H, W, c, i, y, x, o);
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (o.ok[b] && hough_is_far(y, x, o.I[b], o.J[b])) {
                const int pos = atomicAdd(&hdr[0], 1);
                if (pos < HOUGH_FARCAP) { farcell[pos] = c * HW + o.I[b] * W + o.J[b]; farkey[pos] = (unsigned)(b * HW + i); farval[pos] = o.v[b]; }
                else hdr[1] = 1;
            }
    }
}
__global__ __launch_bounds__(1024) void hough_tile_kernel(const float* __restrict__ kp, const float* __restrict__ soff, int H, int W, double norm,
                                                          double* __restrict__ heat, int* __restrict__ hdr, const int* __restrict__ farcell,
                                                          const unsigned* __restrict__ farkey, const double* __restrict__ farval, int tiles_x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    int* cnt = reinterpret_cast<int*>(sm);
    int* off = cnt + 1024;
    int* cur = off + 1024;                       // (after pass B: the list of heavy cells)
    unsigned* keys = reinterpret_cast<unsigned*>(cur + 1024);
    double* vals = reinterpret_cast<double*>(keys + HT_CAP);
    double* scr = vals + HT_CAP;                 // [16 waves][HT_SCR]
    __shared__ int s_nh, s_wtot[16];
    if (hdr[1]) return;                          // (uniform: only hough_far_kernel, a finished launch, writes hdr[1]; the scatter formulation takes over)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.y, HW = H * W;
    const int tI = blockIdx.x / tiles_x, tJ = blockIdx.x - tI * tiles_x;
    const int y0 = tI * HT_T - HT_R, x0 = tJ * HT_T - HT_R;
    const int nfar = hdr[0] < HOUGH_FARCAP ? hdr[0] : HOUGH_FARCAP;
    cnt[tid] = 0; cur[tid] = 0;
    if (tid == 0) s_nh = 0;
    __syncthreads();
    // pass A: the thread's four source pixels (all twelve loads in flight at once), their 16 votes kept in registers for pass B
    constexpr int NPX = HT_S * HT_S / 1024;
    float pk[NPX], px[NPX], py[NPX];
    int pi[NPX];
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
        const int idx = tid + q * 1024;
        const int y = y0 + (idx >> 6), x = x0 + (idx & 63);
        const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        pi[q] = in ? y * W + x : -1;
        const int i = in ? y * W + x : 0;
        pk[q] = kp[(long)c * HW + i]; px[q] = soff[(long)(2 * c) * HW + i]; py[q] = soff[(long)(2 * c + 1) * HW + i];
    }
    short lc[4 * NPX]; unsigned kk[4 * NPX]; double vv[4 * NPX];
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
        const int i = pi[q];
        const int y = i / W, x = i - y * W;
        // (the expressions of hough_vote4 / hough_contrib on the loaded values)
        const double xs = (double)x + (double)px[q], ys = (double)y + (double)py[q], ps = (double)pk[q];
        const int fx = f2i_np(floor(xs)), fy = f2i_np(floor(ys));
        const int cx = f2i_np(ceil(xs)), cy = f2i_np(ceil(ys));
        const double dx = xs - (double)fx, dy = ys - (double)fy;
        const int I[4] = {fy, fy, cy, cy}, J[4] = {fx, cx, fx, cx};
        const double v[4] = {ps * (1. - dx) * (1. - dy), ps * dx * (1. - dy), ps * dy * (1. - dx), ps * dy * dx};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bool ok = i >= 0 && !(I[b] < 0 || I[b] >= H || J[b] < 0 || J[b] >= W || v[b] == 0.) && I[b] / HT_T == tI && J[b] / HT_T == tJ;
            lc[4 * q + b] = ok ? (short)(((I[b] - tI * HT_T) << 5) | (J[b] - tJ * HT_T)) : (short)-1;
            kk[4 * q + b] = (unsigned)(b * HW + i); vv[4 * q + b] = v[b];
            if (ok) atomicAdd(&cnt[lc[4 * q + b]], 1);
        }
    }
    auto visit_far = [&](auto&& f) {
        for (int q = tid; q < nfar; q += 1024) {
            const int cc = farcell[q] - c * HW;
            if (cc < 0 || cc >= HW) continue;
            const int I = cc / W, J = cc - I * W;
            if (I / HT_T == tI && J / HT_T == tJ) f(((I - tI * HT_T) << 5) | (J - tJ * HT_T), farkey[q], farval[q]);
        }
    };
    visit_far([&](int l, unsigned, double) { atomicAdd(&cnt[l], 1); });
    __syncthreads();
    // exclusive scan of cnt -> off (1024 entries, one per thread): wave scans by cross-lane moves, the 16 wave totals through LDS
    const int n = cnt[tid];
    int incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int t = s_wtot[w]; if (w < wave) wbase += t; total += t; }
    const int base = wbase + incl - n;
    off[tid] = base;
    if (total > HT_CAP) {                        // (uniform) too many votes for the LDS slabs: the scatter formulation recomputes the whole map
        if (tid == 0) hdr[2] = 1;               // (hdr[2], not hdr[1]: see hough_gave_up)
        return;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4 * NPX; ++q)
        if (lc[q] >= 0) { const int p = off[lc[q]] + atomicAdd(&cur[lc[q]], 1); keys[p] = kk[q]; vals[p] = vv[q]; }
    visit_far([&](int l, unsigned k, double v) { const int p = off[l] + atomicAdd(&cur[l], 1); keys[p] = k; vals[p] = v; });
    __syncthreads();
    {
        const int I = tI * HT_T + (tid >> 5), J = tJ * HT_T + (tid & 31);
        const bool inside = I < H && J < W;
        if (inside && n <= HOUGH_K) {
            unsigned k[HOUGH_K]; double v[HOUGH_K];
#pragma unroll
            for (int e = 0; e < HOUGH_K; ++e) { k[e] = e < n ? keys[base + e] : 0xffffffffu; v[e] = e < n ? vals[base + e] : 0.; }
            hough_cswap(k[0], v[0], k[1], v[1]); hough_cswap(k[2], v[2], k[3], v[3]); hough_cswap(k[4], v[4], k[5], v[5]); hough_cswap(k[6], v[6], k[7], v[7]);
            hough_cswap(k[0], v[0], k[2], v[2]); hough_cswap(k[1], v[1], k[3], v[3]); hough_cswap(k[4], v[4], k[6], v[6]); hough_cswap(k[5], v[5], k[7], v[7]);
            hough_cswap(k[1], v[1], k[2], v[2]); hough_cswap(k[5], v[5], k[6], v[6]);
            hough_cswap(k[0], v[0], k[4], v[4]); hough_cswap(k[1], v[1], k[5], v[5]); hough_cswap(k[2], v[2], k[6], v[6]); hough_cswap(k[3], v[3], k[7], v[7]);
            hough_cswap(k[2], v[2], k[4], v[4]); hough_cswap(k[3], v[3], k[5], v[5]);
            hough_cswap(k[1], v[1], k[2], v[2]); hough_cswap(k[3], v[3], k[4], v[4]); hough_cswap(k[5], v[5], k[6], v[6]);
            double s = 0.;
#pragma unroll
            for (int e = 0; e < HOUGH_K; ++e)
                if (e < n) s += v[e];
            heat[(long)c * HW + (long)I * W + J] = s / norm;
        } else if (inside) {
            cur[atomicAdd(&s_nh, 1)] = tid;      // (cur is free after pass B: the heavy cells of the tile, in any order)
        }
    }
    __syncthreads();
    const int nh = s_nh;
    double* my = scr + wave * HT_SCR;
    for (int h = wave; h < nh; h += 16) {
        const int t = cur[h], m = cnt[t], b0 = off[t];
        double s = 0.;
        for (int r0 = 0; r0 < m; r0 += HT_SCR) {         // ranks r0 .. r0 + HT_SCR - 1 of the cell's votes, then the next window
            for (int e = lane; e < m; e += 64) {
                const unsigned ke = keys[b0 + e];
                int rank = 0;
                for (int j = 0; j < m; ++j) rank += keys[b0 + j] < ke;
                if (rank >= r0 && rank < r0 + HT_SCR) my[rank - r0] = vals[b0 + e];
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const int w = m - r0 < HT_SCR ? m - r0 : HT_SCR;
                for (int q = 0; q < w; ++q) s += my[q];
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) heat[(long)c * HW + (long)(tI * HT_T + (t >> 5)) * W + (tJ * HT_T + (t & 31))] = s / norm;
    }
}
// start of the scatter formulation when the tile formulation gave up (hdr[1] | hdr[2] != 0): clears the slab allocator, the vote counters and the cursors
__global__ void hough_clear_kernel(const int* __restrict__ hdr, int4* __restrict__ a, long n4) {
    if (!hough_gave_up(hdr)) return;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) a[i] = make_int4(0, 0, 0, 0);
}

// ---- P2 ---------------------------------------------------------------------------------------
__constant__ double KG_GW[17] = {
    0x1.18aad19e4159bp-14, 0x1.c98b8c5d0dda5p-12, 0x1.227362b5fc92dp-9, 0x1.1f30504e20207p-7, 0x1.ba4d4125ffd2ap-6,
    0x1.0941b71ceef37p-4,  0x1.ef9093fc46e5ap-4,  0x1.68856f9ab1982p-3, 0x1.98862a07ae7b4p-3, 0x1.68856f9ab1982p-3,
    0x1.ef9093fc46e5ap-4,  0x1.0941b71ceef37p-4,  0x1.ba4d4125ffd2ap-6, 0x1.1f30504e20207p-7, 0x1.227362b5fc92dp-9,
    0x1.c98b8c5d0dda5p-12, 0x1.18aad19e4159bp-14};
__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        if (i >= n) i = 2 * n - 1 - i;
    }
    return i;
}
template <int AXIS>  // 0: along y (rows), 1: along x
__global__ void gauss_kernel(const double* __restrict__ in, double* __restrict__ out, int C, int H, int W) {
    const long total = (long)C * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); const long q = i / W; const int y = (int)(q % H); const long c = q / H;
        const double* src = in + c * H * W;
        double acc = src[(long)y * W + x] * KG_GW[8];
#pragma unroll
        for (int j = -8; j < 0; ++j) {
            double a, b;
            if (AXIS == 0) { a = src[(long)reflect_idx(y + j, H) * W + x]; b = src[(long)reflect_idx(y - j, H) * W + x]; }
            else { a = src[(long)y * W + reflect_idx(x + j, W)]; b = src[(long)y * W + reflect_idx(x - j, W)]; }
            acc = acc + (a + b) * KG_GW[8 + j];
        }
        out[i] = acc;
    }
}

// ---- P3 ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_peak(const double* __restrict__ h, int H, int W, int y, int x, double thresh) {
    const double v = h[(long)y * W + x];
    double m = v;
    if (y > 0 && h[(long)(y - 1) * W + x] > m) m = h[(long)(y - 1) * W + x];
    if (y < H - 1 && h[(long)(y + 1) * W + x] > m) m = h[(long)(y + 1) * W + x];
    if (x > 0 && h[(long)y * W + x - 1] > m) m = h[(long)y * W + x - 1];
    if (x < W - 1 && h[(long)y * W + x + 1] > m) m = h[(long)y * W + x + 1];
    return m == v && v > thresh;
}
// pass 0: per-block counts; pass 1: write at base offsets (blockbase from scan_kernel over block counts)
template <int PASS>
__global__ __launch_bounds__(256) void peaks_kernel(const double* __restrict__ heat, int H, int W, double thresh,
                                                    int* __restrict__ blockcount, const int* __restrict__ blockbase,
                                                    int cap, int* __restrict__ ids, int* __restrict__ xs,
                                                    int* __restrict__ ys, double* __restrict__ conf) {
    const long HW = (long)H * W, total = 5 * HW;
    const long i0 = (long)blockIdx.x * 1024 + threadIdx.x * 4;  // 4 consecutive elements per thread
    bool f[4]; int cnt = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        long i = i0 + e;
        f[e] = false;
        if (i < total) {
            int c = (int)(i / HW); long r = i - c * HW;
            f[e] = is_peak(heat + c * HW, H, W, (int)(r / W), (int)(r % W), thresh);
        }
        cnt += f[e];
    }
    __shared__ int sc[256];
    sc[threadIdx.x] = cnt;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        int v = threadIdx.x >= d ? sc[threadIdx.x - d] : 0;
        __syncthreads();
        sc[threadIdx.x] += v;
        __syncthreads();
    }
    if (PASS == 0) {
        if (threadIdx.x == 255) blockcount[blockIdx.x] = sc[255];
    } else {
        int pos = blockbase[blockIdx.x] + sc[threadIdx.x] - cnt;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (f[e]) {
                long i = i0 + e; int c = (int)(i / HW); long r = i - c * HW;
                if (pos < cap) { ids[pos] = c; xs[pos] = (int)(r % W); ys[pos] = (int)(r / W); conf[pos] = heat[i]; }
                ++pos;
            }
    }
}
__global__ void peaks_total_kernel(const int* __restrict__ blockcount, const int* __restrict__ blockbase, int nblocks,
                                   int* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = blockbase[nblocks - 1] + blockcount[nblocks - 1];
}

// ---- P4 ---------------------------------------------------------------------------------------
__constant__ int KG_MID_IDX[5][5] = {{-1, 0, 1, 2, 3}, {10, -1, 4, 5, 6}, {11, 14, -1, 7, 8}, {12, 15, 17, -1, 9}, {13, 16, 18, 19, -1}};
__device__ __forceinline__ double norm2(double dx, double dy) { return sqrt(fma(dy, dy, dx * dx)); }

// stable rank sort by confidence descending (postprocessing.py:87)
__global__ void kp_rank_kernel(const int* __restrict__ npk, int cap, const int* __restrict__ ids,
                               const int* __restrict__ xs, const int* __restrict__ ys, const double* __restrict__ conf,
                               int* __restrict__ sid, int* __restrict__ sx, int* __restrict__ sy,
                               double* __restrict__ sconf) {
    // Rank of peak i = number of peaks that precede it in the stable descending order.  16 lanes share a peak (lane = every 16th
    // candidate of a 256-entry LDS tile), their partial counts meet through four shuffles: the fp64 compares are the cost of this
    // kernel (as one thread per peak over global memory the c0 map of a 512 x 512 image -- ~8000 peaks -- took 105-120 us).
    __shared__ double tile[256];
    int n = *npk; if (n > cap) n = cap;
    const int sub = threadIdx.x & 15, loc = threadIdx.x >> 4;            // blockDim.x == 256: 16 peaks per workgroup and round
    for (int i0 = blockIdx.x * 16; i0 < n; i0 += gridDim.x * 16) {
        const int i = i0 + loc;
        const bool live = i < n;
        const double ci = live ? conf[i] : 0.;
        int rank = 0;
        for (int j0 = 0; j0 < n; j0 += 256) {
            __syncthreads();
            if (j0 + (int)threadIdx.x < n) tile[threadIdx.x] = conf[j0 + threadIdx.x];
            __syncthreads();
            const int m = n - j0 < 256 ? n - j0 : 256;
            for (int j = sub; j < m; j += 16) { const double cj = tile[j]; rank += (cj > ci) || (cj == ci && j0 + j < i); }
        }
        rank += __shfl_xor(rank, 1, 64); rank += __shfl_xor(rank, 2, 64); rank += __shfl_xor(rank, 4, 64); rank += __shfl_xor(rank, 8, 64);
        if (live && sub == 0) { sid[rank] = ids[i]; sx[rank] = xs[i]; sy[rank] = ys[i]; sconf[rank] = ci; }
    }
}
// Fast path of the greedy grouping (n <= GK_NL keypoints, image <= 1024 x 1024).  The sequential dependence over the seeds stays, but
// one seed costs ONE workgroup barrier (none if it is dropped) and a few LDS list steps instead of two scans over all remaining
// keypoints in global memory:
//   * keypoint coordinates (int16), a state byte (bit 0 alive, bit 1 member of a skeleton, bits 2..4 type) and per (type, cell)
//     linked lists live in LDS (cells of 8 / 16 / 32 px; a 3 x 3 or 5 x 5 cell neighbourhood covers the radius-6 / radius-10 tests; list order is
//     irrelevant: the minimum distance with ties to the lower index is order independent);
//   * the "<= 10 px from slot `id` of ANY existing skeleton" test (postprocessing.py:100) == some keypoint of type id that is a MEMBER
//     of a skeleton lies within 10 px, or the seed lies within 10 px of the origin and some skeleton lacks type id (missing slot = (0,0));
//     every wave evaluates it for itself (lanes 16.. and 48), so a dropped seed needs no barrier;
//   * wave w finds the match of the w-th target type (lanes 0..8 walk the 3 x 3 cells around the proposal); the four matches are
//     exchanged through a parity-double-buffered LDS slot behind the one barrier, after which EVERY wave applies the (identical) state
//     updates itself and keeps the skeleton / missing-type counters in registers;
//   * the seeds' four mid offsets (random global reads) are fetched 256 seeds at a time; skeletons are recorded as keypoint indices and
//     expanded to [x, y, conf] rows by all threads at the end.
#define GK_NL 8192
#define GK_CELLS 1024      // (1024 / 32)^2
struct GroupLds {
    short kx[GK_NL], ky[GK_NL];
    unsigned short nxt[GK_NL];
    unsigned char st[GK_NL];
    int head[5 * GK_CELLS];
    float pre[256][4][2];
    int match[2][4];
};
__device__ void group_fast(GroupLds& L, int n, const int* __restrict__ sid, const int* __restrict__ sx, const int* __restrict__ sy,
                           const double* __restrict__ sconf, const float* __restrict__ mid, int H, int W, int skcap,
                           int* __restrict__ skidx, double* __restrict__ skel, int* __restrict__ nskel) {
    const long HW = (long)H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // cell size 8 / 16 / 32 px by image size (<= 32 x 32 cells): dense small scales get short lists.  Radius 6 < 8: 3 x 3 cells for the
    // candidate search; radius 10: 5 x 5 cells of 8 px, 3 x 3 otherwise, for the skeleton-member test.
    const int CS = (H <= 256 && W <= 256) ? 3 : (H <= 512 && W <= 512) ? 4 : 5;
    const int CW = (W + (1 << CS) - 1) >> CS, CH = (H + (1 << CS) - 1) >> CS;
    const int RA = CS == 3 ? 2 : 1, NA = (2 * RA + 1) * (2 * RA + 1);
    for (int c = tid; c < 5 * GK_CELLS; c += 256) L.head[c] = -1;
    __syncthreads();
    for (int j = tid; j < n; j += 256) {
        const int t = sid[j], x = sx[j], y = sy[j];
        L.kx[j] = (short)x; L.ky[j] = (short)y; L.st[j] = (unsigned char)(1 | (t << 2));
        const int old = atomicExch(&L.head[t * GK_CELLS + (y >> CS) * CW + (x >> CS)], j);
        L.nxt[j] = (unsigned short)(old < 0 ? 0xffff : old);
    }
    int ns = 0, par = 0, miss0 = 0, miss1 = 0, miss2 = 0, miss3 = 0, miss4 = 0;     // wave-uniform, identical in every wave
    const bool roleB = lane < 9, roleA = lane >= 16 && lane < 16 + NA;
    const int rl = roleB ? lane : lane - 16, rw = roleB ? 3 : 2 * RA + 1;
    const int dcx = rl % rw - (rw >> 1), dcy = rl / rw - (rw >> 1);                 // the lane's cell of the neighbourhood
    for (int i = 0; i < n; ++i) {
        if ((i & 255) == 0) {              // mid offsets of the next 256 seeds (uniform branch)
            __syncthreads();
            const int q = i + tid;
            if (q < n) {
                const int id = sid[q];
                const long pix = (long)sy[q] * W + sx[q];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int m = KG_MID_IDX[id][w + (w >= id ? 1 : 0)];
                    L.pre[tid][w][0] = mid[(long)(2 * m) * HW + pix];
                    L.pre[tid][w][1] = mid[(long)(2 * m + 1) * HW + pix];
                }
            }
            __syncthreads();
        }
        const int sti = L.st[i];
        if (!(sti & 1)) continue;          // uniform within a wave; every wave sees its own (identical) state updates
        const int id = sti >> 2, kx = L.kx[i], ky = L.ky[i];
        const int t = wave + (wave >= id ? 1 : 0);
        const double px = (double)kx + (double)L.pre[i & 255][wave][0];
        const double py = (double)ky + (double)L.pre[i & 255][wave][1];
        // lanes 0..8: candidates of type t around the proposal; lanes 16..16+NA-1: skeleton members of type id around the seed
        int j = -1;
        if (roleB) {
            if (px > -64. && py > -64. && px < 2048. && py < 2048.) {               // (further out: nothing within 6 px)
                const int cx = ((int)floor(px) >> CS) + dcx, cy = ((int)floor(py) >> CS) + dcy;
                if (cx >= 0 && cy >= 0 && cx < CW && cy < CH) j = L.head[t * GK_CELLS + cy * CW + cx];
            }
        } else if (roleA) {
            const int cx = (kx >> CS) + dcx, cy = (ky >> CS) + dcy;
            if (cx >= 0 && cy >= 0 && cx < CW && cy < CH) j = L.head[id * GK_CELLS + cy * CW + cx];
        }
        double bd = 1e300; int bj = 0x7fffffff, hit = 0;
        while (j >= 0) {
            const int stj = L.st[j], xj = L.kx[j], yj = L.ky[j];
            if (roleB) {
                if (stj & 1) {             // alive => not yet popped => later in the sorted list than the seed
                    const double d = norm2(px - (double)xj, py - (double)yj);
                    if (d <= 6. && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
                }
            } else if (stj & 2) {
                const int dx = kx - xj, dy = ky - yj;
                hit |= (dx * dx + dy * dy) <= 100;   // integer coordinates: sqrt(dx^2+dy^2) <= 10 <=> dx^2+dy^2 <= 100 exactly
            }
            const int nj = L.nxt[j];
            j = nj == 0xffff ? -1 : nj;
        }
        if (lane == 48) {
            const int mi = id == 0 ? miss0 : id == 1 ? miss1 : id == 2 ? miss2 : id == 3 ? miss3 : miss4;
            hit = (kx * kx + ky * ky <= 100) && mi > 0;
        }
        if (__ballot(hit) != 0) {          // dropped: no barrier, every wave clears the alive bit itself
            if (lane == 0) L.st[i] = (unsigned char)(sti & ~1);
            continue;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const double od = __shfl_xor(bd, o, 64); const int oj = __shfl_xor(bj, o, 64);
            if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
        }
        if (lane == 0) L.match[par][wave] = bj;
        __syncthreads();
        int mj[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) mj[w] = L.match[par][w];
        if (lane == 0) {
            L.st[i] = (unsigned char)((sti & ~1) | 2);
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (mj[w] != 0x7fffffff) L.st[mj[w]] = (unsigned char)((L.st[mj[w]] & ~1) | 2);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (mj[w] != 0x7fffffff) continue;
            const int tt = w + (w >= id ? 1 : 0);
            miss0 += tt == 0; miss1 += tt == 1; miss2 += tt == 2; miss3 += tt == 3; miss4 += tt == 4;
        }
        if (tid == 0 && ns < skcap) {
            int* rec = skidx + (long)ns * 5;
            rec[id] = i;
#pragma unroll
            for (int w = 0; w < 4; ++w) rec[w + (w >= id ? 1 : 0)] = mj[w] == 0x7fffffff ? -1 : mj[w];
        }
        ++ns; par ^= 1;
    }
    __syncthreads();
    if (tid == 0) *nskel = ns;
    const int nrec = ns < skcap ? ns : skcap;
    for (int e = tid; e < nrec * 5; e += 256) {
        const int jj = skidx[e];
        double* o = skel + (long)e * 3;
        if (jj >= 0) { o[0] = (double)sx[jj]; o[1] = (double)sy[jj]; o[2] = sconf[jj]; }
        else { o[0] = 0.; o[1] = 0.; o[2] = 0.; }
    }
}

__global__ __launch_bounds__(256) void group_kernel(const int* __restrict__ npk, int cap, const int* __restrict__ sid,
                                                    const int* __restrict__ sx, const int* __restrict__ sy,
                                                    const double* __restrict__ sconf, const float* __restrict__ mid,
                                                    int H, int W, unsigned char* __restrict__ alive, int skcap,
                                                    int* __restrict__ skxy, double* __restrict__ skel,
                                                    int* __restrict__ nskel) {
    // skxy[s][5][2]: integer slot coordinates of skeleton s (missing slot = (0,0)) for the <=10 test
    int n = *npk; if (n > cap) n = cap;
    extern __shared__ __attribute__((aligned(16))) unsigned char group_smem[];
    if (n <= GK_NL && H <= 1024 && W <= 1024) {      // (the general path below serves larger inputs)
        group_fast(*reinterpret_cast<GroupLds*>(group_smem), n, sid, sx, sy, sconf, mid, H, W, skcap, skxy, skel, nskel);
        return;
    }
    const long HW = (long)H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < n; i += 256) alive[i] = 1;
    __shared__ int s_ns, s_match[4];
    if (threadIdx.x == 0) s_ns = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (!alive[i]) continue;  // uniform: alive[] only changes between barriers
        const int id = sid[i], kx = sx[i], ky = sy[i];
        const int ns = s_ns;
        int hit = 0;
        const int nsc = ns < skcap ? ns : skcap;
        for (int s = threadIdx.x; s < nsc; s += 256) {
            int dx = kx - skxy[(s * 5 + id) * 2], dy = ky - skxy[(s * 5 + id) * 2 + 1];
            // integer-valued coordinates: sqrt(dx^2+dy^2) <= 10  <=>  dx^2+dy^2 <= 100 exactly
            hit |= ((long)dx * dx + (long)dy * dy) <= 100;
        }
        hit = __syncthreads_or(hit);
        if (hit) { if (threadIdx.x == 0) alive[i] = 0; __syncthreads(); continue; }
        // wave w searches the w-th target type (ascending t, skipping the seed's own type)
        const int t = wave + (wave >= id ? 1 : 0);
        const int m = KG_MID_IDX[id][t];
        const long pix = (long)ky * W + kx;
        const double px = (double)kx + (double)mid[(long)(2 * m) * HW + pix];
        const double py = (double)ky + (double)mid[(long)(2 * m + 1) * HW + pix];
        double bd = 1e300; int bj = 0x7fffffff;
        for (int j = i + 1 + lane; j < n; j += 64) {
            if (!alive[j] || sid[j] != t) continue;
            double d = norm2(px - (double)sx[j], py - (double)sy[j]);
            if (d <= 6. && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            double od = __shfl_xor(bd, o, 64); int oj = __shfl_xor(bj, o, 64);
            if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
        }
        if (lane == 0) s_match[wave] = bj;
        __syncthreads();
        if (threadIdx.x == 0) {
            alive[i] = 0;
            double sk[15]; int xy[10];
            for (int q = 0; q < 15; ++q) sk[q] = 0.;
            for (int q = 0; q < 10; ++q) xy[q] = 0;
            sk[id * 3] = (double)kx; sk[id * 3 + 1] = (double)ky; sk[id * 3 + 2] = sconf[i];
            xy[id * 2] = kx; xy[id * 2 + 1] = ky;
            for (int w = 0; w < 4; ++w) {
                int j = s_match[w];
                if (j == 0x7fffffff) continue;
                int tt = w + (w >= id ? 1 : 0);
                alive[j] = 0;
                sk[tt * 3] = (double)sx[j]; sk[tt * 3 + 1] = (double)sy[j]; sk[tt * 3 + 2] = sconf[j];
                xy[tt * 2] = sx[j]; xy[tt * 2 + 1] = sy[j];
            }
            if (ns < skcap) {
                for (int q = 0; q < 15; ++q) skel[(long)ns * 15 + q] = sk[q];
                for (int q = 0; q < 10; ++q) skxy[ns * 10 + q] = xy[q];
            }
            s_ns = ns + 1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *nskel = s_ns;
}

// ---- P6 + P7 ------------------------------------------------------------------------------------
__device__ __forceinline__ double dmin(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ double dmax(double a, double b) { return b > a ? b : a; }
// returns 0 (no box), 1 (box)
__device__ int skeleton_box(const double* __restrict__ sk_in, double scale, int do_refine, double* box) {
    double s[15];
    for (int q = 0; q < 15; ++q) s[q] = sk_in[q];
    int m[5], cnt = 0;
    for (int j = 0; j < 5; ++j) { m[j] = s[j * 3] > 0.; cnt += m[j]; }
    if (do_refine && !((cnt >= 3) || (m[0] && m[3]) || (m[1] && m[2]))) return 0;
    for (int j = 0; j < 5; ++j) { s[j * 3] *= scale; s[j * 3 + 1] *= scale; }
    const double *tl = s, *tr = s + 3, *bl = s + 6, *br = s + 9, *cc = s + 12;
    const int nc = m[0] + m[1] + m[2] + m[3];
    double sum = 0.;
    for (int j = 0; j < 5; ++j) if (m[j]) sum += s[j * 3 + 2];
    double y1, x1, y2, x2;
    if (nc == 4) { y1 = dmin(tl[1], tr[1]); y2 = dmax(bl[1], br[1]); x1 = dmin(tl[0], bl[0]); x2 = dmax(tr[0], br[0]); }
    else if (nc == 3) {
        y1 = (m[0] && m[1]) ? dmin(tl[1], tr[1]) : dmax(tl[1], tr[1]); y2 = dmax(bl[1], br[1]);
        x1 = (m[0] && m[2]) ? dmin(tl[0], bl[0]) : dmax(tl[0], bl[0]); x2 = dmax(tr[0], br[0]);
    } else if (nc == 2) {
        if (m[0] && m[3]) { y1 = tl[1]; y2 = br[1]; x1 = tl[0]; x2 = br[0]; }
        else if (m[1] && m[2]) { y1 = tr[1]; y2 = bl[1]; x1 = bl[0]; x2 = tr[0]; }
        else if (m[0] && m[1] && m[4]) { y1 = dmin(tl[1], tr[1]); y2 = y1 + (cc[1] - y1) * 2; x1 = tl[0]; x2 = tr[0]; }
        else if (m[0] && m[2] && m[4]) { y1 = tl[1]; y2 = bl[1]; x1 = dmin(tl[0], bl[0]); x2 = x1 + (cc[0] - x1) * 2; }
        else if (m[1] && m[3] && m[4]) { y1 = tr[1]; y2 = br[1]; x2 = dmax(tr[0], br[0]); x1 = x2 - (x2 - cc[0]) * 2; }
        else if (m[2] && m[3] && m[4]) { y2 = dmax(bl[1], br[1]); y1 = y2 - (y2 - cc[1]) * 2; x1 = bl[0]; x2 = br[0]; }
        else return 0;
    } else return 0;
    box[0] = y1; box[1] = x1; box[2] = y2; box[3] = x2; box[4] = sum / (double)cnt;
    return 1;
}
// one block; appends the boxes of this scale after the *nbox already present (order preserved)
__global__ __launch_bounds__(1024) void boxes_kernel(const int* __restrict__ nskel, int skcap,
                                                     const double* __restrict__ skel, double scale, int do_refine,
                                                     int boxcap, double* __restrict__ boxes, int* __restrict__ nbox) {
    __shared__ int sc[1024];
    __shared__ int s_base;
    int n = *nskel; if (n > skcap) n = skcap;
    if (threadIdx.x == 0) s_base = *nbox;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        double b[5]; int ok = 0;
        if (i < n) ok = skeleton_box(skel + (long)i * 15, scale, do_refine, b);
        sc[threadIdx.x] = ok;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            int v = threadIdx.x >= d ? sc[threadIdx.x - d] : 0;
            __syncthreads();
            sc[threadIdx.x] += v;
            __syncthreads();
        }
        const int pos = s_base + sc[threadIdx.x] - ok;
        if (ok && pos < boxcap)
            for (int q = 0; q < 5; ++q) boxes[(long)pos * 5 + q] = b[q];
        __syncthreads();
        if (threadIdx.x == 1023) s_base += sc[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *nbox = s_base;
}

// ---- P9 ---------------------------------------------------------------------------------------
#define NMS_NL 2048
// one block.  order[] = indices sorted by confidence ascending (ties: index ascending).
__global__ __launch_bounds__(1024) void nms_kernel(const int* __restrict__ nbox, int boxcap,
                                                   const double* __restrict__ boxes, double thresh,
                                                   int* __restrict__ order, unsigned char* __restrict__ dead,
                                                   int* __restrict__ keep, int* __restrict__ nkeep) {
    int n = *nbox; if (n > boxcap) n = boxcap;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double ci = boxes[(long)i * 5 + 4];
        int rank = 0;
        for (int j = 0; j < n; ++j) { double cj = boxes[(long)j * 5 + 4]; rank += (cj < ci) || (cj == ci && j < i); }
        order[rank] = i; dead[i] = 0;
    }
    __syncthreads();
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
    if (n <= NMS_NL) {   // boxes (in rank order), areas and the suppressed flags in LDS: one barrier and no global access per kept box
        double* sb = reinterpret_cast<double*>(nms_smem);            // [n][4]
        double* sarea = sb + 4 * NMS_NL;
        unsigned char* sdead = reinterpret_cast<unsigned char*>(sarea + NMS_NL);
        for (int q = threadIdx.x; q < n; q += 1024) {
            const double* b = boxes + (long)order[q] * 5;
            sb[q * 4] = b[0]; sb[q * 4 + 1] = b[1]; sb[q * 4 + 2] = b[2]; sb[q * 4 + 3] = b[3];
            sarea[q] = (b[3] - b[1]) * (b[2] - b[0]);
            sdead[q] = 0;
        }
        __syncthreads();
        int nk = 0;
        for (int p = n - 1; p >= 0; --p) {
            if (sdead[p]) continue;  // uniform
            if (threadIdx.x == 0) keep[nk] = order[p];
            ++nk;
            const double cy1 = sb[p * 4], cx1 = sb[p * 4 + 1], cy2 = sb[p * 4 + 2], cx2 = sb[p * 4 + 3];
            const double carea = (cx2 - cx1) * (cy2 - cy1);
            for (int q = threadIdx.x; q < p; q += 1024) {
                if (sdead[q]) continue;
                const double* b = sb + q * 4;
                double yy1 = b[0] > cy1 ? b[0] : cy1, xx1 = b[1] > cx1 ? b[1] : cx1;
                double yy2 = b[2] < cy2 ? b[2] : cy2, xx2 = b[3] < cx2 ? b[3] : cx2;
                double w = xx2 - xx1, h = yy2 - yy1;
                w = w > 0. ? w : 0.; h = h > 0. ? h : 0.;
                double inter = w * h;
                double uni = (sarea[q] - inter) + carea;
                double iou = inter / uni;
                if (!(iou <= thresh)) sdead[q] = 1;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) *nkeep = nk;
        return;
    }
    int nk = 0;
    for (int p = n - 1; p >= 0; --p) {
        const int cur = order[p];
        if (dead[cur]) continue;  // uniform
        if (threadIdx.x == 0) keep[nk] = cur;
        ++nk;
        const double cy1 = boxes[(long)cur * 5], cx1 = boxes[(long)cur * 5 + 1], cy2 = boxes[(long)cur * 5 + 2],
                     cx2 = boxes[(long)cur * 5 + 3];
        const double carea = (cx2 - cx1) * (cy2 - cy1);
        for (int q = threadIdx.x; q < p; q += 1024) {
            const int k = order[q];
            if (dead[k]) continue;
            const double* b = boxes + (long)k * 5;
            double yy1 = b[0] > cy1 ? b[0] : cy1, xx1 = b[1] > cx1 ? b[1] : cx1;
            double yy2 = b[2] < cy2 ? b[2] : cy2, xx2 = b[3] < cx2 ? b[3] : cx2;
            double w = xx2 - xx1, h = yy2 - yy1;
            w = w > 0. ? w : 0.; h = h > 0. ? h : 0.;
            double inter = w * h;
            double area = (b[3] - b[1]) * (b[2] - b[0]);
            double uni = (area - inter) + carea;
            double iou = inter / uni;
            if (!(iou <= thresh)) dead[k] = 1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *nkeep = nk;
}
__global__ void gather_rows5_kernel(const double* __restrict__ boxes, const int* __restrict__ keep,
                                    const int* __restrict__ nkeep, double* __restrict__ out) {
    int n = *nkeep;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * 5; i += gridDim.x * blockDim.x)
        out[i] = boxes[(long)keep[i / 5] * 5 + (i % 5)];
}

// ---- C ABI ------------------------------------------------------------------------------------
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" long kg_postproc_workspace_bytes(int H, int W, int peak_cap, int skel_cap) {
    size_t HW = (size_t)H * W, b = 0;
    b += al256(5 * HW * 4) * 3;            // count, offs, cursor
    b += al256(5 * 4 * HW * 4);            // keys
    b += al256(5 * 4 * HW * 8) * 2;        // vals, sorted
    b += al256(5 * HW * 8) * 3;            // heat, tmp, blur
    b += al256(5 * HW * 4) + 256;          // heavy list + counter
    size_t nblk = (5 * HW + 1023) / 1024;
    b += al256(nblk * 4) * 2 + 256;        // peak block counts / bases / total
    b += (al256((size_t)peak_cap * 4) * 3 + al256((size_t)peak_cap * 8)) * 2;  // peaks + sorted peaks
    b += al256(peak_cap);                  // alive
    b += al256((size_t)skel_cap * 10 * 4);  // skxy
    b += al256(5 * HW * HOUGH_K * 4) + al256(5 * HW * HOUGH_K * 8);   // inline vote slots (keys, values) of the scatter formulation
    b += al256(5 * 4 * HW * 4) + al256(5 * 4 * HW * 8);                // compact keys of the heavy cells' slabs, rank-sort scratch
    return (long)b + 4096;
}

struct PPWs {
    int *count, *offs, *cursor; unsigned* keys; double *vals, *sorted, *heat, *tmp, *blur;
    int *heavy_list, *heavy_n, *blkcount, *blkbase, *npk;
    int *ids, *xs, *ys; double* conf; int *sid, *sx, *sy; double* sconf;
    unsigned char* alive; int* skxy; int nblk;
    unsigned* ink; double* inv; unsigned* skey; double* srt;
};
static void carve(void* ws, int H, int W, int peak_cap, int skel_cap, PPWs* p) {
    size_t HW = (size_t)H * W;
    unsigned char* q = (unsigned char*)ws;
    auto take = [&](size_t bytes) { void* r = q; q += al256(bytes); return r; };
    p->heavy_n = (int*)take(256);      // (counters first: ONE memset clears them together with count .. cursor)
    p->count = (int*)take(5 * HW * 4); p->offs = (int*)take(5 * HW * 4); p->cursor = (int*)take(5 * HW * 4);
    p->keys = (unsigned*)take(5 * 4 * HW * 4);
    p->vals = (double*)take(5 * 4 * HW * 8); p->sorted = (double*)take(5 * 4 * HW * 8);
    p->heat = (double*)take(5 * HW * 8); p->tmp = (double*)take(5 * HW * 8); p->blur = (double*)take(5 * HW * 8);
    p->heavy_list = (int*)take(5 * HW * 4); (void)take(256);
    p->nblk = (int)((5 * HW + 1023) / 1024);
    p->blkcount = (int*)take((size_t)p->nblk * 4); p->blkbase = (int*)take((size_t)p->nblk * 4); p->npk = (int*)take(256);
    p->ids = (int*)take((size_t)peak_cap * 4); p->xs = (int*)take((size_t)peak_cap * 4); p->ys = (int*)take((size_t)peak_cap * 4);
    p->conf = (double*)take((size_t)peak_cap * 8);
    p->sid = (int*)take((size_t)peak_cap * 4); p->sx = (int*)take((size_t)peak_cap * 4); p->sy = (int*)take((size_t)peak_cap * 4);
    p->sconf = (double*)take((size_t)peak_cap * 8);
    p->alive = (unsigned char*)take(peak_cap); p->skxy = (int*)take((size_t)skel_cap * 10 * 4);
    p->ink = (unsigned*)take(5 * HW * HOUGH_K * 4); p->inv = (double*)take(5 * HW * HOUGH_K * 8);
    p->skey = (unsigned*)take(5 * 4 * HW * 4); p->srt = (double*)take(5 * 4 * HW * 8);
}

// ---- phase timing (measurement hook of bench.py --mode eval: the roofline of the HBM-bound phases) ------------------------------------
// kg_postproc_timing_begin() arms this host thread: every kg_postproc_scale call then records HIP events on ITS stream at the phase
// boundaries (Hough vote | Gaussian | peaks + ranking | grouping).  kg_postproc_timing_end(ms) waits for them and returns the
// summed durations of the calls since begin: ms[0..3] = Hough, Gaussian, peaks, grouping.  Not armed: no events, no cost.
#include <vector>
struct PPTiming { bool on = false; std::vector<hipEvent_t> ev; };
static PPTiming& pp_timing() { static thread_local PPTiming t; return t; }
static inline void pp_mark(hipStream_t st) {
    PPTiming& t = pp_timing();
    if (!t.on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    t.ev.push_back(e);
}
extern "C" int kg_postproc_timing_begin(void) {
    PPTiming& t = pp_timing();
    for (hipEvent_t e : t.ev) (void)hipEventDestroy(e);
    t.ev.clear();
    t.on = true;
    return KG_OK;
}
extern "C" int kg_postproc_timing_end(float* ms4) {
    PPTiming& t = pp_timing();
    KG_CHECK_ARG(ms4 && t.on, "kg_postproc_timing_end: not armed");
    t.on = false;
    for (int k = 0; k < 4; ++k) ms4[k] = 0.f;
    for (size_t c = 0; c + 5 <= t.ev.size(); c += 5) {
        KG_HIP(hipEventSynchronize(t.ev[c + 4]));
        for (int k = 0; k < 4; ++k) {
            float ms = 0.f;
            KG_HIP(hipEventElapsedTime(&ms, t.ev[c + k], t.ev[c + k + 1]));
            ms4[k] += ms;
        }
    }
    for (hipEvent_t e : t.ev) (void)hipEventDestroy(e);
    t.ev.clear();
    return KG_OK;
}

// P1..P4 for one scale (batch element 0 of the maps).  kp [5][H][W], soff [10][H][W], mid [40][H][W] fp32
// device pointers.  Outputs (device): skel [skel_cap][5][3] f64, nskel, and optionally copies of the
// intermediate stages (heat_out/blur_out [5][H][W] f64, peaks) for parity tests.
extern "C" int kg_postproc_scale(const float* kp, const float* soff, const float* mid, int H, int W, double thresh,
                                 void* ws, long ws_bytes, int peak_cap, int skel_cap, double* skel, int* nskel,
                                 double* heat_out, double* blur_out, int* peaks_out, double* peak_conf_out,
                                 int* npeaks_out, void* stream) {
    KG_CHECK_ARG(kp && soff && mid && ws && skel && nskel, "kg_postproc_scale: null pointer");
    KG_CHECK_ARG(H >= 1 && W >= 1 && (long)H * W <= (1L << 28) / 4, "kg_postproc_scale: bad size");
    KG_CHECK_ARG(W < 65536 && H < 65536, "kg_postproc_scale: map too large");
    KG_CHECK_ARG(ws_bytes >= kg_postproc_workspace_bytes(H, W, peak_cap, skel_cap), "kg_postproc_scale: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    PPWs p; carve(ws, H, W, peak_cap, skel_cap, &p);
    const int HW = H * W;
    const double norm = 3.141592653589793 * 25.0;  // np.pi * KP_RADIUS**2 (postprocessing.py:51)
    pp_mark(st);
    int gx = (4 * HW + 255) / 256; if (gx > 2048) gx = 2048;
    {
        // header (p.heavy_n, 256 bytes): [0..1] the 64-bit slab allocator of the scatter formulation, [8] = far-vote count, [9] / [10] = "tile formulation
        // gave up" flags (far-list overflow / tile overflow; hdr = &heavy_n[8]).  KG_HOUGH_TILE=0: scatter formulation only.
        static const int use_tile = getenv("KG_HOUGH_TILE") ? atoi(getenv("KG_HOUGH_TILE")) : 1;
        int* hdr = p.heavy_n + 8;
        const long clr = (long)(((unsigned char*)p.cursor - (unsigned char*)p.count) + (size_t)5 * HW * 4 + 15) / 16;      // (rounded UP: the al256 padding behind `cursor` takes the 1-3 extra ints)
        KG_HIP(hipMemsetAsync(p.heavy_n, 0, 256, st));
        if (use_tile) {
            // far list in the scatter formulation's (then unused) arrays: cells in `sorted`, keys in `keys`, values in `vals`
            int* farcell = reinterpret_cast<int*>(p.sorted);
            int gf = (HW + 255) / 256; if (gf > 2048) gf = 2048;
            hipLaunchKernelGGL(hough_far_kernel, dim3(gf, 5), dim3(256), 0, st, kp, soff, H, W, hdr, farcell, p.keys, p.vals, reinterpret_cast<int4*>(p.count), clr);
            constexpr int tile_lds = 3 * 1024 * 4 + HT_CAP * 4 + HT_CAP * 8 + 16 * HT_SCR * 8;
            static KgPerDevice tile_attr;
            if (tile_attr.first()) {
                KG_HIP(hipFuncSetAttribute((const void*)hough_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tile_lds));
            }
            const int tiles_x = (W + HT_T - 1) / HT_T, tiles_y = (H + HT_T - 1) / HT_T;
            hipLaunchKernelGGL(hough_tile_kernel, dim3(tiles_x * tiles_y, 5), dim3(1024), tile_lds, st, kp, soff, H, W, norm, p.heat, hdr, farcell, p.keys,
                               p.vals, tiles_x);
        } else {
            KG_HIP(hipMemsetD32Async((hipDeviceptr_t)(hdr + 1), 1, 1, st));
            hipLaunchKernelGGL(hough_clear_kernel, dim3(1024), dim3(256), 0, st, hdr, reinterpret_cast<int4*>(p.count), clr);
        }
        // scatter formulation, every kernel a no-op while hdr[1] | hdr[2] == 0 (its counters count | offs | cursor were cleared by the far pass)
        unsigned long long* ctr64 = reinterpret_cast<unsigned long long*>(p.heavy_n);
        hipLaunchKernelGGL(hough_scatter_kernel, dim3(gx, 5), dim3(256), 0, st, kp, soff, H, W, p.count, p.ink, p.inv, (int*)p.keys, p.vals, hdr);
        hipLaunchKernelGGL(hough_classify_kernel, dim3((5 * HW + 1023) / 1024), dim3(1024), 0, st, 5 * HW, p.count, p.ink, p.inv, norm, p.heat, ctr64,
                           p.offs, p.heavy_list, p.skey, p.sorted, hdr);
        int go = (int)(((long)20 * HW + 255) / 256); if (go > 4096) go = 4096;
        hipLaunchKernelGGL(hough_ovfill_kernel, dim3(go), dim3(256), 0, st, (long)20 * HW, 4 * HW, (const int*)p.keys, p.vals, p.offs, p.cursor, p.skey,
                           p.sorted, hdr);
        hipLaunchKernelGGL(hough_heavy2_kernel, dim3(4096), dim3(64), 0, st, p.count, p.offs, p.skey, p.sorted, p.srt, norm, p.heat, ctr64, p.heavy_list, hdr);
    }
    int gg = (5 * HW + 255) / 256; if (gg > 8192) gg = 8192;
    pp_mark(st);
    hipLaunchKernelGGL(gauss_kernel<0>, dim3(gg), dim3(256), 0, st, p.heat, p.tmp, 5, H, W);
    hipLaunchKernelGGL(gauss_kernel<1>, dim3(gg), dim3(256), 0, st, p.tmp, p.blur, 5, H, W);
    pp_mark(st);
    hipLaunchKernelGGL(peaks_kernel<0>, dim3(p.nblk), dim3(256), 0, st, p.blur, H, W, thresh, p.blkcount, (const int*)nullptr, 0,
                       (int*)nullptr, (int*)nullptr, (int*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, p.blkcount, p.blkbase, p.nblk, p.npk);
    hipLaunchKernelGGL(peaks_kernel<1>, dim3(p.nblk), dim3(256), 0, st, p.blur, H, W, thresh, (int*)nullptr, p.blkbase, peak_cap,
                       p.ids, p.xs, p.ys, p.conf);
    hipLaunchKernelGGL(kp_rank_kernel, dim3(512), dim3(256), 0, st, p.npk, peak_cap, p.ids, p.xs, p.ys, p.conf, p.sid, p.sx, p.sy,
                       p.sconf);
    pp_mark(st);
    static KgPerDevice group_attr;
    if (group_attr.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GroupLds)));
    }
    hipLaunchKernelGGL(group_kernel, dim3(1), dim3(256), sizeof(GroupLds), st, p.npk, peak_cap, p.sid, p.sx, p.sy, p.sconf, mid, H, W, p.alive,
                       skel_cap, p.skxy, skel, nskel);
    pp_mark(st);
    if (heat_out) KG_HIP(hipMemcpyAsync(heat_out, p.heat, (size_t)5 * HW * 8, hipMemcpyDeviceToDevice, st));
    if (blur_out) KG_HIP(hipMemcpyAsync(blur_out, p.blur, (size_t)5 * HW * 8, hipMemcpyDeviceToDevice, st));
    if (npeaks_out) KG_HIP(hipMemcpyAsync(npeaks_out, p.npk, 4, hipMemcpyDeviceToDevice, st));
    if (peaks_out) {
        KG_HIP(hipMemcpyAsync(peaks_out, p.ids, (size_t)peak_cap * 4, hipMemcpyDeviceToDevice, st));
        KG_HIP(hipMemcpyAsync(peaks_out + peak_cap, p.xs, (size_t)peak_cap * 4, hipMemcpyDeviceToDevice, st));
        KG_HIP(hipMemcpyAsync(peaks_out + 2 * (size_t)peak_cap, p.ys, (size_t)peak_cap * 4, hipMemcpyDeviceToDevice, st));
    }
    if (peak_conf_out) KG_HIP(hipMemcpyAsync(peak_conf_out, p.conf, (size_t)peak_cap * 8, hipMemcpyDeviceToDevice, st));
    KG_CHECK_LAUNCH("postproc_scale");
    return KG_OK;
}

// P6+P7: append boxes of one scale's skeletons to boxes[] (nbox is read-modify-written on device).
extern "C" int kg_skeleton_boxes(const double* skel, const int* nskel, int skel_cap, double scale, int do_refine,
                                 double* boxes, int* nbox, int box_cap, void* stream) {
    KG_CHECK_ARG(skel && nskel && boxes && nbox, "kg_skeleton_boxes: null pointer");
    hipLaunchKernelGGL(boxes_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nskel, skel_cap, skel, scale, do_refine, box_cap,
                       boxes, nbox);
    KG_CHECK_LAUNCH("skeleton_boxes");
    return KG_OK;
}
// P9: ws needs box_cap*(4+1+4) bytes (+ alignment).  out [box_cap][5] kept boxes in pick order.
extern "C" int kg_nms(const double* boxes, const int* nbox, int box_cap, double thresh, void* ws, long ws_bytes,
                      double* out, int* nkeep, void* stream) {
    KG_CHECK_ARG(boxes && nbox && ws && out && nkeep, "kg_nms: null pointer");
    KG_CHECK_ARG(ws_bytes >= (long)(al256((size_t)box_cap * 4) * 2 + al256(box_cap)), "kg_nms: workspace too small");
    unsigned char* q = (unsigned char*)ws;
    int* order = (int*)q; q += al256((size_t)box_cap * 4);
    int* keep = (int*)q; q += al256((size_t)box_cap * 4);
    unsigned char* dead = q;
    hipStream_t st = (hipStream_t)stream;
    constexpr int nms_lds = NMS_NL * (5 * 8 + 1);
    static KgPerDevice nms_attr;
    if (nms_attr.first()) {
        KG_HIP(hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, nms_lds));
    }
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(1024), nms_lds, st, nbox, box_cap, boxes, thresh, order, dead, keep, nkeep);
    hipLaunchKernelGGL(gather_rows5_kernel, dim3(64), dim3(256), 0, st, boxes, keep, nkeep, out);
    KG_CHECK_LAUNCH("nms");
    return KG_OK;
}

// fp64 primitive probe for the parity tests: out = {a/b, sqrt(|a|), fma(a,a,b*b), floor(a), ceil(a)}
__global__ void f64_probe_kernel(const double* a, const double* b, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = a[i] / b[i]; out[n + i] = sqrt(fabs(a[i])); out[2 * n + i] = sqrt(fma(a[i], a[i], b[i] * b[i]));
    out[3 * n + i] = floor(a[i]); out[4 * n + i] = ceil(a[i]);
}
extern "C" int kg_f64_probe(const double* a, const double* b, double* out, int n, void* stream) {
    hipLaunchKernelGGL(f64_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    KG_CHECK_LAUNCH("f64_probe");
    return KG_OK;
}

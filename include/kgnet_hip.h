/*
 * kgnet_hip.h -- C ABI of libkgnet_hip.so: the MI355X (gfx950) hot path of KGnet.
 *
 * The reference (yijingru/KG_Instance_Segmentation) has no FFI; its boundary is the Python module
 * surface its drivers import (train.py:3-11, test.py:3-12).  The drop-in Python modules in
 * kg_instance_segmentation_amd/ bind the entry points below with ctypes; every entry point cites the
 * reference code it replaces.  Conventions:
 *   - extern "C", plain pointers and sizes, no torch types.  All pointers are DEVICE pointers unless
 *     stated otherwise; the library never retains a caller's pointer past the call (the recorded weight-gradient reductions between
 *     kg_wgrad_reduce_defer(1) and the flush excepted) and allocates no device memory of its own EXCEPT two per-device scratch buffers,
 *     created on first use and kept for the life of the process: the split-K partial tiles of under-filled conv launches (64 MB) and
 *     the partial maps of split kg_conv2d_halo_heads2 launches (grown to 3 x parts x N x 55 x H x W floats, <= 384 MB).  A launch that
 *     uses one is followed by its finishing launch on the same stream, so conv calls of ONE device must come from one stream at a time.
 *   - every call enqueues on the caller's `stream` (a hipStream_t) and does not synchronise.
 *   - return value: 0 = ok, otherwise kg_last_error() (thread-local) describes the failure.
 *   - activations ("rows"): bf16, pixel-major [row][ld] (NHWC), channel counts multiples of 8.
 *   - weights: fp32 OIHW master copies are packed to bf16 [Cout_pad][K] by kg_pack_weight.
 *   - precision: the reference computes in fp32 (KGnet.py:22-29 -> F.conv2d on fp32 tensors).  The fast matrix path of gfx950
 *     is bf16 MFMA with fp32 accumulation, so tensors that must carry more than bf16 are stored as P "planes" of bf16 whose
 *     sum is the value (P = 2: 16 significant bits, P = 3: the fp32 value exactly) and a product of an xP-plane activation
 *     with a wP-plane weight is the sum of the bf16 MFMA products x_i * w_j with i + j < max(xP, wP).  Entry points that
 *     take rows operands accept a `const kg_planes_t* planes` (host struct, NULL = single-plane bf16 everywhere); each
 *     documents which operand uses which slot.
 */
#ifndef KGNET_HIP_H
#define KGNET_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct kg_planes {
    int a_planes, a_pstride;   /* first bf16 rows operand: planes, element stride between planes (same row, same ld) */
    int b_planes, b_pstride;   /* second bf16 rows operand */
    int c_planes, c_pstride;   /* third bf16 rows operand */
    int y_planes, y_pstride;   /* output rows */
    int w_planes;              /* planes of the packed weights (virtual-channel layout, see kg_pack_weight) */
    int reserved_;
    const float* scale;        /* device scalar or NULL: kg_grad_pack / kg_f32_to_planes multiply by *scale (kg_grad_scale) */
    const float* oscale;       /* device fp32 [Cout] or NULL: kg_conv2d_igemm / kg_conv2d_halo compute act(acc * oscale[co] + bias[co] (+ res)):
                                  an inference-mode BatchNorm folded into the conv (KGnet.py:82-97 conv -> bn -> relu as one launch;
                                  oscale = gamma / sqrt(running_var + eps), bias = beta - running_mean * oscale), applied to the fp32 accumulator */
} kg_planes_t;

/* The same C ABI is built for two 16-bit storage formats of the rows / packed-weight operands ("bf16 rows" above):
 *   libkgnet_hip.so      bfloat16 (kg_rows_format() == 0);
 *   libkgnet_hip_f16.so  IEEE half (kg_rows_format() == 1): 11 significant bits per plane, so hi + lo planes (22 bits, 3 MFMA
 *                        products per multiply) carry the reference's fp32 tensors where bf16 needs three planes (6 products).
 *                        Packed weights are stored times 2^12 and every conv epilogue scales its fp32 accumulators back; the
 *                        gradients of a backward pass are stored times a power of two chosen on the device (kg_grad_scale).
 * Entry points without rows operands (losses, post-processing, NMS, optimizer, ...) are format-independent and live in
 * libkgnet_hip.so only. */
int kg_rows_format(void);

const char* kg_last_error(void);
/* Measurement aid: the name rocprofv3 prints (without the argument list) for the conv-family kernel that the calling thread's most recent
 * kg_conv2d_* / kg_conv1x1 / kg_conv3x3_c64 / kg_conv2d_wgrad* call launched, e.g. "conv_halo_kernel<7, 1, 8, 0>" ("" before the first).
 * bench.py keys its per-kernel table with it, so that table and the rocprofv3 summaries under profiles/ use the same names. */
const char* kg_last_kernel(void);
int kg_version(void);
int kg_device_arch(char* out, int cap);   /* host buffer */
int kg_tr_probe(void* out_u16x256, void* stream);                       /* test probe: ds_read_b64_tr_b16 lane map */
int kg_f64_probe(const double* a, const double* b, double* out5n, int n, void* stream); /* test probe: fp64 rounding */

/* ---- convolution: torch.nn.Conv2d forward / input-gradient (KGnet.py:22-29, 131-209 -> F.conv2d) ----
 * y[m][co] = act( sum_{tap,ci} x[src(m,tap)][ci] * w[co][tap][ci] + bias[co] + res[m][co] ) (* mask>0)
 * mode 0: dense forward (H,W = input dims, OH,OW = output dims); mode 1: dense transposed = gradient w.r.t.
 * the input of a forward conv (H,W = dY dims, OH,OW = dX dims, weights packed transposed);
 * mode 2/3: the same on a ragged pixel list with per-row descriptors {(y<<16)|x, (h<<16)|w}.
 * y (bf16 rows) and/or y_f32 (fp32 NCHW [N][f32_C][OH*OW]) receive the result.  tile 0 = auto; 1..5 pin an output tile; 6 = split-K over the
 * waves of a 64-pixel x 64-cout tile for launches with too few output tiles to fill the chip (single-image inference: cin_pad % 64 == 0,
 * dense modes, rows output). */
int kg_conv2d_igemm(const void* x, const void* w, const float* bias, void* y, float* y_f32, const void* res,
                    const void* mask, const int* rowdesc, int M, int H, int W, int OH, int OW, int cin_pad, int ldx,
                    int Cout, int ldy, int ldres, int ldmask, int K, int KH, int KW, int stride, int pad, int dil,
                    int mode, int relu, int f32_C, int tile, const kg_planes_t* planes, void* stream);
                    /* planes: a = x, b = res, y = y; cin_pad = channels of ONE x plane */
/* the same convolution for stride 1, "same" padding, KS in {3,7}, cin_pad % 64 == 0: input halo resident in LDS
 * (the 7x7 head convolutions of KGnet.py:161-209 are 86 % of the network's FLOPs).  flip = 1: input gradient. */
int kg_conv2d_halo(const void* x, const void* w, const float* bias, void* y, float* y_f32, const void* res,
                   const void* mask, int N, int H, int W, int cin_pad, int ldx, int Cout, int ldy, int ldres, int ldmask,
                   int K, int KS, int flip, int relu, int f32_C, int wc, const int* tiletab, int ntiles, int total_rows,
                   const kg_planes_t* planes, void* stream);   /* planes: a = x, b = res, y = y.  tiletab != NULL: ragged boxes, one {row0,(h<<16)|w,(oy0<<16)|ox0,0} entry per workgroup */
                   /* wc: couts per workgroup / 64 (0 = default) | 256: the packed weights are zero for channels 32..63 of every chunk | 512 / 1024: the NARROW
                      input gradient (flip = 1, KS = 7, one single-plane 64-channel dY: only channels 0..7 / 8..23 carry data -- the kp / short second-layer
                      heads, KGnet.py:161-209 `.2`): w packed by kg_pack_weight_narrow (7 / 14 virtual taps of 64 columns), 4 / 2 kernel columns per k-step */
/* the three second-layer 7x7 head convolutions of one scale (KGnet.py:161-209 `.2` layers, sigmoid on kp :300) in one launch
 * over the fused hidden rows [N*H*W][>=3C]: w = packed [64 virtual couts][49][3C] (kg_pack_weight_rows), bias64 / vmap[64]
 * indexed by virtual cout (vmap: channel of kp 0-4 | short 5-14 | mid 15-54, or -1); fp32 NCHW outputs */
/* (maps with < 128 (tile, head) workgroups and 3-product inputs: one workgroup per plane product as well, fp32 partial maps in a library
 *  scratch, summed in a fixed order by a second launch -- single-image inference) */
int kg_conv2d_halo_heads2(const void* x, const void* w, const float* bias64, const int* vmap, float* kp, float* sh, float* md,
                          int N, int H, int W, int C, int ldx, int K, int kp_sigmoid, const kg_planes_t* planes, void* stream);   /* kp_sigmoid: 1 = torch.sigmoid on kp as KGnet.py:300, 0 = raw logits.  planes: a = x;
                          w holds [head][virtual planes][C] channels per tap (kg_pack_weight_rows with tap_stride) */
/* 3x3 stride-1 "same" conv / input gradient (flip) for cin_pad == 64 and Cout <= 64 (c0_conv.2, layer1 conv2, seg level 0): persistent
 * workgroups, all 9 taps' weights resident in LDS, the next 16x16 tile's halo fetched by LDS-direct loads during the current tile */
int kg_conv3x3_c64(const void* x, const void* w, const float* bias, void* y, const void* res, const void* mask, int N, int H, int W,
                   int ldx, int Cout, int ldy, int ldres, int ldmask, int K, int flip, int relu, const int* tiletab16, int ntiles,
                   void* stream);
/* 7x7 stride-1 "same" input gradient (flip = 1; 0: forward) of a NARROW conv -- the kp / short second-layer head convs (KGnet.py:161-209 `.2`: C -> 5 / 10):
 * the rows x [N*H*W][ldx] carry data in the chan_slot (8 or 16) channels from chan_lo on; w packed by kg_pack_weight_narrow (7 / 14 virtual taps of 64
 * columns, resident in LDS), persistent workgroups over 16 x 16-pixel tiles with compact double-buffered halos; y rows of Cout channels, zeroed where
 * mask <= 0 (ReLU backward of the hidden tensor; NULL: none).  Single 16-bit planes. */
int kg_conv7_narrow(const void* x, const void* w, void* y, const void* mask, int N, int H, int W, int ldx, int chan_lo, int chan_slot, int Cout, int ldy,
                    int ldmask, int K, int flip, void* stream);
/* Weight-stationary variant for the full-resolution 64 -> 64 channel 3x3 convs of the fp32-tolerance forward pass (c0_conv.2 KGnet.py:139-142,
 * c1_up_conv :153, seg_head.0 :145-147, skip_combine.0.up :116-119): x in hi + lo planes (planes->a_planes == 2), w packed by kg_pack_weight
 * with x_planes = w_planes = 2 (K >= 9 * 192), y = ReLU?(conv + bias) in 1 or 2 planes.  Dense (N images of H x W) or ragged
 * (tiletab8: one {row0, (h << 16) | w, (oy0 << 16) | ox0, 0} entry per 8 x 16 tile of a box). */
int kg_conv3x3_ws(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int ldx, int ldy, int K, int relu,
                  const int* tiletab8, int ntiles, const kg_planes_t* planes, void* stream);
/* 1x1 stride-1 convolution / its input gradient as a streaming GEMM over rows (dense or ragged): weight slab resident in
 * LDS, pixel fragments straight from global memory, persistent workgroups (KGnet.py:64-99,101-111,155-158) */
int kg_conv1x1(const void* x, const void* w, const float* bias, void* y, const void* res, const void* mask, long M, int K,
               int wK, int ldx, int Cout, int ldy, int ldres, int ldmask, int relu, void* stream);
/* fp32 OIHW parameter -> packed bf16 matrix rows (forward) or its transpose (data gradient).  x_planes / w_planes: split-bf16
 * layout; per tap the row holds one cin_pad-channel copy of weight plane j for every kept product x_i * w_j
 * (i + j < max(x_planes, w_planes)), ordered by descending i + j, then descending j (smallest products first, hi * hi last,
 * so that the fp32 accumulator rounds like the reference's); a conv kernel walks these "virtual channels" like a single-plane
 * conv with more input channels and only remaps the activation side to plane i.  (1, 1) = the plain bf16 matrix. */
int kg_pack_weight(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int K, int cin_pad, int row0, int c0,
                   int transposed, int x_planes, int w_planes, void* stream);
/* forward packing with a row table: packed row of output channel co = rowmap[co] (device int array); tap_stride: channels per
 * tap row when several plane groups share the matrix (0 = vplanes * cin_pad) */
int kg_pack_weight_rows(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int K, int cin_pad, const int* rowmap,
                        int c0, int x_planes, int w_planes, int tap_stride, void* stream);
/* transposed (input-gradient) packing of a narrow conv (Cout <= tap_stride in {8, 16}) for kg_conv2d_halo's wc | 512 / 1024 variants:
 * dst[(row0 + ci) * K + (ky * tap_pitch + kx) * tap_stride + c0 + co] = w[co][ci][ky][kx]; single plane; K >= KH * tap_pitch * tap_stride */
int kg_pack_weight_narrow(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int K, int row0, int c0, int tap_stride, int tap_pitch,
                          void* stream);
/* all (re)packs of a step in one launch: jobs = device array of 80-byte records {const float* w; void* dst; const int* rowmap;
 * int Cout, Cin, taps, K, cin_pad, row0, c0, transposed, gx, blk0, x_planes, w_planes, tap_stride, tap_pitch;} (gx = Cout, or
 * ceil(Cout/64) when transposed; blk0 = first workgroup of the job); total_blocks = sum of gx * gy over the jobs */
int kg_pack_weight_batch(const void* jobs, int njobs, int total_blocks, void* stream);
/* im2col for <= 8 input channels (weight gradient of the stem conv1, KGnet.py:131, as a 1x1 weight-gradient GEMM):
 * out[m][tap * cin + ci] = x[src(m, tap)][ci]; out rows of Kpad >= KH*KW*cin bf16 values, padding columns pre-zeroed */
int kg_im2col_small(const void* x, void* out, int N, int H, int W, int OH, int OW, int KH, int KW, int stride, int pad, int cin,
                    int ldx, int Kpad, void* stream);
/* weight gradient (autograd of nn.Conv2d at train.py:153): partial sums [nsplit][Cout][taps][Cin] fp32 */
int kg_conv2d_wgrad(const void* x, const void* dy, float* dwp, const int* rowdesc, int M, int H, int W, int OH, int OW,
                    int ldx, int lddy, int Cin, int Cout, int cin_lim, int cout_lim, int KH, int KW, int stride, int pad,
                    int dil, int mode, int nsplit, long split_stride, const kg_planes_t* planes, void* stream);
                    /* planes: a = x, b = dy: the kept plane products x_i * dY_j are further passes over the pixels in the same launch */
/* weight gradient of a dense stride-1 "same" 3x3 / 7x7 conv with dY tile + X halo resident in LDS (all taps per staging pass) */
int kg_conv2d_wgrad_halo(const void* x, const void* dy, float* dwp, int N, int H, int W, int ldx, int lddy, int Cin, int Cout,
                         int cin_lim, int cout_lim, int KS, int nsplit, long split_stride, const int* tiletab16, int ntiles,
                         float* dbp, const kg_planes_t* planes, void* stream);   /* planes: a = x, b = dy.  tiletab16 != NULL: ragged boxes, one entry per 16x16 tile;
                         dbp != NULL: also writes the bias-gradient partials [nsplit][Cout] (sum over pixels of dy) */
int kg_bias_grad_final(const float* part, float* db, int nsplit, int C, int accumulate, void* stream);
int kg_wgrad_reduce(const float* part, float* grad_oihw, int Cout, int Cin, int KH, int KW, int nsplit, long split_stride,
                    int accumulate, void* stream);
/* the same for a conv fused along Cout (heads sharing their input): tensor k receives the next counts[k] rows; ngrads <= 4;
   grads / counts are HOST arrays */
int kg_wgrad_reduce_multi(const float* part, float* const* grads_oihw, const int* counts, int ngrads, int Cin, int KH, int KW,
                          int nsplit, long split_stride, int accumulate, void* stream);
/* kg_wgrad_reduce_multi + the bias gradient of the same conv in the same launch: bias_part = the [nsplit][bias_C] partials of
   kg_conv2d_wgrad_halo (dbp), db [bias_C]; bias_part == NULL: weights only */
int kg_wgrad_reduce_bias(const float* part, float* const* grads_oihw, const int* counts, int ngrads, int Cin, int KH, int KW,
                         int nsplit, long split_stride, int accumulate, const float* bias_part, float* db, int bias_C, void* stream);
/* Deferred reductions: between kg_wgrad_reduce_defer(1) and kg_wgrad_reduce_defer(0) the three reduce entry points above only record their
   job (per calling thread; the caller keeps every partial buffer alive and unmodified), kg_wgrad_reduce_flush launches the recorded jobs
   twelve per launch -- each with the block decomposition of its own single launch, so the sums are the same bits.  The reference has no
   counterpart (autograd's conv backward returns finished gradients, train.py:148-154); this only cuts ~70 launches of a few microseconds
   out of a backward pass. */
int kg_wgrad_reduce_defer(int on);
int kg_wgrad_reduce_pending(void);
int kg_wgrad_reduce_flush(void* stream);
int kg_bias_grad(const void* dy, float* db, float* scratch, int scratch_floats, int M, int C, int ld, int accumulate,
                 void* stream);
int kg_set_wgrad_tr(int use_transpose_read);   /* test switch: LDS transpose-read vs scalar fragment loads */

/* ---- backbone glue: image pack, BatchNorm2d (KGnet.py:82-97,132), MaxPool2d (KGnet.py:134), bilinear
 *      F.interpolate(align_corners=False) (KGnet.py:110,288-297), elementwise joins ---- */
/* planes slots of this group: a = first input (x / dy of bilinear_bwd), b = second input (res / dy), y = output; masks use plane 0 */
int kg_img_pack(const float* img_nchw, void* out_rows8, int ldout, int N, int C, int H, int W, const kg_planes_t* planes, void* stream);
int kg_bn_stats_train(const void* x, int ldx, int M, int C, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, float* mean_out, float* invstd_out, float* scale,
                      float* shift, float* scratch, int scratch_floats, const kg_planes_t* planes, void* stream);
/* BatchNorm statistics from the PRODUCING conv's epilogue (the conv of KGnet.py:82-93 that feeds a train-mode BatchNorm2d):
 * kg_conv_stats_begin arms the calling host thread's next kg_conv2d_igemm / kg_conv2d_halo launch (bf16 rows output, no ReLU / residual /
 * mask): that launch also writes per-(pixel tile, channel) {sum, sum of squares} partials of its fp32 accumulators into `part`;
 * kg_conv_stats_end returns the tile count nb (0: the launch used a kernel without this epilogue -- fall back to kg_bn_stats_train) and
 * disarms; kg_bn_finalize_train = the second stage of kg_bn_stats_train over those partials [nb][C][2]. */
int kg_conv_stats_begin(float* part, long cap_floats);
int kg_conv_stats_end(int* nb);
int kg_bn_finalize_train(const float* part, int nb, int M, int C, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, float* mean_out, float* invstd_out, float* scale,
                         float* shift, void* stream);
int kg_bn_scale_shift_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, float* scale, float* shift, void* stream);
int kg_bn_apply(const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldres, void* y,
                int ldy, int M, int C, int relu, const kg_planes_t* planes, void* stream);
int kg_bn_bwd(const void* x, int ldx, const void* dy, int lddy, const float* gamma, const float* mean,
              const float* invstd, float* dgamma, float* dbeta, int accumulate, void* dx, int lddx, int M, int C,
              float* scratch, int scratch_floats, const float* parts, int nb_parts, const float* parts_scale,
              const kg_planes_t* planes, void* stream);
/* BACKWARD statistics from the input gradient that completes dy (train-mode BatchNorm under autograd, KGnet.py:82-93 + train.py:153):
 * kg_conv_bstats_begin arms the calling host thread's next dense input-gradient launch (kg_conv2d_igemm mode 1 on the gather kernel, kg_conv2d_halo
 * with flip = 1 and KS = 3; output channels a multiple of 64): besides storing the gradient rows g (after the residual add and the ReLU mask of
 * its epilogue) that launch writes per-(pixel tile, channel) partials {sum g, sum g * xhat}, xhat = (x - mean[c]) * invstd[c] over the rows x of
 * the BatchNorm's INPUT (same row index; x_planes planes, x_pstride elements apart).  kg_conv_stats_end returns the tile count (0: the launch took
 * a kernel without this epilogue) and disarms.  kg_bn_bwd takes the partials as parts / nb_parts and skips its own column reduction over x and
 * dy; parts_scale (optional device scalar): a power of two dy was multiplied by AFTER the partials were taken (kg_rows_rescale). */
int kg_conv_bstats_begin(float* part, long cap_floats, const void* x, int ldx, int x_planes, int x_pstride, const float* mean,
                         const float* invstd);
int kg_maxpool3s2_fwd(const void* x, int ldx, void* y, int ldy, void* argmax_u8, int N, int H, int W, int C, const kg_planes_t* planes, void* stream);
int kg_maxpool3s2_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, const void* argmax_u8, int N, int H, int W, int C,
                      const kg_planes_t* planes, void* stream);   /* argmax_u8 (optional): [N*OH*OW][C] winning taps written by the forward */
int kg_bilinear_fwd(const void* x, int ldx, void* y, int ldy, int N, int IH, int IW, int OH, int OW, int C,
                    const int* boxdesc, const int* row2box, long total_out_rows, const kg_planes_t* planes, void* stream);
int kg_bilinear_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int IH, int IW, int OH, int OW, int C,
                    const int* boxdesc, const int* row2box, long total_in_rows, const void* mask, int ldmask,
                    const kg_planes_t* planes, void* stream);
                    /* mask != NULL: dx is zeroed where mask <= 0 (ReLU backward of the upsampled tensor, KGnet.py:110) */
int kg_add_rows(const void* a, int lda, const void* b, int ldb, const void* mask, int ldm, void* y, int ldy, long M,
                int C, const float* scale, const float* scale2, const kg_planes_t* planes, void* stream);
                /* y = (a + b) [* *scale [* *scale2]] [where mask > 0]; scale / scale2: optional device scalars (powers of two: a gradient written
                   under an earlier running scale of the half-precision backward is converted inside the join that reads it, cf. kg_rows_scale) */
int kg_sigmoid_inplace(float* x, long n, void* stream);                              /* torch.sigmoid, KGnet.py:300,345 */
int kg_grad_pack(const float* g_nchw, const float* prob, void* out_rows, int N, int C, int H, int W, int ld, int cpad,
                 const kg_planes_t* planes, void* stream);   /* planes: y = out_rows */
int kg_grad_pack3(const float* g0_nchw, const float* g1_nchw, const float* g2_nchw, const float* prob0, void* out_rows, int N, int C0, int C1, int C2,
                  int H, int W, int ld, int pad0, int pad1, int pad2, const kg_planes_t* planes, void* stream);
                  /* the three map gradients of one pyramid level (kp | short | mid: the backward of KGnet.py:300-316) side by side in one pass: columns
                     [0, pad0) <- g0 (* prob0 (1 - prob0) when given), then pad1 columns of g1, pad2 of g2 (pads % 8 == 0, sum <= 64); planes: y = out_rows */

/* ---- losses: DetectionLossAll (loss.py:12-49) and the per-pair mask BCE of SEG_loss (seg_loss.py:86-94) ---- */
int kg_detection_loss_fwd(const float* kp, const float* sh, const float* md, const float* gt, int N, int H, int W,
                          float kp_radius, const float* den_override, float* scratch, int scratch_floats, float* out8,
                          void* stream);
int kg_detection_loss_bwd(const float* kp, const float* sh, const float* md, const float* gt, int N, int H, int W,
                          float kp_radius, const float* fin8, const float* grad_out, float* g_kp, float* g_sh,
                          float* g_md, void* stream);
int kg_seg_loss(const float* prob, const void* tgt_u8, const int* patches, const void* pairs, int npatches, float* part,
                float* out1, const float* grad_out, float* gprob, void* stream);

/* measurement hook (bench.py --mode eval): begin arms this host thread -- every kg_postproc_scale call then records HIP events at its
 * phase boundaries on its own stream; end waits and returns the summed milliseconds {Hough vote, Gaussian, peaks + ranking, grouping} */
int kg_postproc_timing_begin(void);
int kg_postproc_timing_end(float* ms4);
/* ---- post-processing in float64, bit-identical to postprocessing.py:16-261 and nms.py:4-53 ---- */
long kg_postproc_workspace_bytes(int H, int W, int peak_cap, int skel_cap);
int kg_postproc_scale(const float* kp, const float* soff, const float* mid, int H, int W, double thresh, void* ws,
                      long ws_bytes, int peak_cap, int skel_cap, double* skel, int* nskel, double* heat_out,
                      double* blur_out, int* peaks_out, double* peak_conf_out, int* npeaks_out, void* stream);
int kg_skeleton_boxes(const double* skel, const int* nskel, int skel_cap, double scale, int do_refine, double* boxes,
                      int* nbox, int box_cap, void* stream);
int kg_nms(const double* boxes, const int* nbox, int box_cap, double thresh, void* ws, long ws_bytes, double* out,
           int* nkeep, void* stream);

/* ---- ground-truth maps of one pyramid scale (preprocessing.get_ground_truth, preprocessing.py:107-118, assembled and cast
 * as dataset_base.py:99-109): kps = device float32 [n][5][2] (x,y) keypoints tl,tr,bl,br,centre; out = device float32
 * [55][H][W] (kp 5 | short 10 | mid 40), bit-identical to the reference's float32 tensors ---- */
int kg_gt_maps(const float* kps, int n, int H, int W, float* out, void* stream);

/* ---- fused multi-tensor Adam step (train.py:71,154), fp32, torch.optim.Adam's operation order.  jobs = device array of 48-byte
 * records {float* p; const float* g; float* m; float* v; long n; int blk0; int pad;}, blk0 = first workgroup of the job (4096
 * elements per workgroup); step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) ---- */
int kg_adam_step(const void* jobs, int njobs, int total_blocks, float beta1, float beta2, float eps, float step_size, float bc2_sqrt,
                 float weight_decay, void* stream);

/* ---- host glue of SEG_loss (seg_loss.py:57-80), pure host code: crops of the matched ground-truth masks (float32 [n][H][W] per
 * image), nearest-resized to the patch size, as bytes.  work = int32 [nwork][9]: (img, gt, y1, y2, x1, x2, h1, w1, out offset) ---- */
int kg_host_crop_masks(const float* const* masks, const int* work, int nwork, int H, int W, unsigned char* out);
/* the same crops for DEVICE-resident masks (seg_loss.py:57-80 on the GPU): masks = device array of device pointers
 * (float32 [n_i][H][W]), work = device int32 [nwork][9] (the rows of kg_host_crop_masks), out = device bytes */
int kg_crop_masks(const void* masks, const int* work, int nwork, int H, int W, void* out, void* stream);
/* box tables of the per-box seg branch (host glue of KGnet.py:258-267,321-350; pure host code, integer arithmetic).
 * kg_host_tile_table: one {row0, (h << 16) | w, (oy0 << 16) | ox0, 0} entry per th x tw tile of every box (box-major); returns the count.
 * kg_host_bin_csr: per BS x BS bin of each image the ascending list of boxes touching it (tab = int32 [nb][8] rows {img, y1, x1, h, w, ..});
 * returns the number of (box, bin) incidences.  Both return -1 when the output does not fit cap. */
int kg_host_tile_table(const int* h, const int* w, const long* row0, int nb, int th, int tw, int* out, int cap);
int kg_host_bin_csr(const int* tab, int nb, int BS, int BY, int BX, int nbins, int* bin_start, int* bin_boxes, int cap);
/* matching of seg_loss.py:14-29,55-56 (jaccard_numpy >= thresh over all predicted x ground-truth boxes of an image), pure host code,
 * float32 in the reference's operation order: pairs = int32 [cap][2] (patch, gt) row-major, *count = matches */
int kg_host_match_boxes(const float* pb, int P, const float* gb, int G, int gstride, float thresh, int* pairs, int cap, int* count);

/* ---- evaluation metrics (eval_parts.mask_iou inside seg_evaluation, eval_parts.py:4-9,98-150): exact pixel counts.
 * masks = device bytes [n][ld], ld % 16 == 0, non-zero byte = foreground, padding zero ---- */
int kg_mask_areas(const void* masks, int n, long ld, int* area, void* stream);
int kg_mask_inter_pairs(const void* a, const void* b, const int* pairs, int npairs, long ld, int* inter, void* stream);

/* ---- per-box segmentation branch (KGnet.py:246-267, 321-350): ragged row bookkeeping ---- */
int kg_seg_build_rows(const int* boxtab8, int nb, int* rowdesc, int* row2box, int* srcrow, void* stream);
int kg_seg_build_rows_levels(int nlev, const int* const* boxtab8, const int* nb, int* const* rowdesc, int* const* row2box, int* const* srcrow,
                             void* stream);   /* kg_seg_build_rows for all (<= 8) pyramid levels in one launch: HOST arrays of device pointers / box counts */
int kg_rows_gather(const void* src, int ldsrc, const int* srcrow, void* dst, int lddst, long nrows, int C, void* stream);
int kg_rows_gather_planes(const void* src, int ldsrc, int src_pstride, const int* srcrow, void* dst, int lddst, int dst_pstride, long nrows, int C,
                          int P, void* stream);     /* kg_rows_gather for the P planes of split rows in one launch (plane p at column p * pstride) */
/* seg_head.2 (KGnet.py:145-147: Conv2d(64, 1, 3, padding=1), used at KGnet.py:266): the one-output-channel 3x3 conv over the ragged pixel
 * list as a per-pixel dot product (zero padding at the box border, as each crop is convolved on its own).  x: rows [M][ldx], C = 64 channels
 * (planes: a); w: fp32 OIHW [1][64][3][3], the master parameter itself; bias: fp32 [1] or NULL; rowdesc: kg_seg_build_rows; y: fp32 [M] logits. */
int kg_seg_conv3_c1(const void* x, int ldx, int C, const float* w, const float* bias, const int* rowdesc, long M, float* y,
                    const kg_planes_t* planes, void* stream);
int kg_f32_to_bf16_rows(const float* acc, void* out, int C, long rows, int ldout, const void* addto, int ldadd,
                        void* stream);
/* crops of the fp32 feature maps forward_dec returns (KGnet.py:318 -> get_patches :246-256): dst (planes y) = src_f32[srcrow[r]] */
int kg_rows_gather_f32(const float* src, int ldsrc, const int* srcrow, void* dst, int lddst, long nrows, int C,
                       const kg_planes_t* planes, void* stream);
/* fp32 export of a rows tensor (planes a) -- the feature maps c0..c4 of forward_dec's return value (KGnet.py:318) -- and back
 * (planes y = acc (+ addto, planes b)): the gradients autograd hands forward_dec for them */
int kg_planes_to_f32(const void* x, int ldx, float* out, int ldout, long rows, int C, const kg_planes_t* planes, void* stream);
int kg_f32_to_planes(const float* acc, int ldacc, void* out, int ldout, const void* addto, int ldadd, long rows, int C,
                     const kg_planes_t* planes, void* stream);
/* gradient of get_patches' slicing (KGnet.py:246-256) w.r.t. a feature map, deterministic: out[n][y][x][0:C] (fp32, every
 * element written) = sum over the boxes containing (y, x), in ascending box order, of their crop-gradient row.  Rows
 * [0, rows_a) of the ragged list are read from ga, the others from gb (row r - rows_a); boxtab = {n,y1,x1,h,w,row0,H,W}
 * per box; bin_start / bin_boxes = CSR lists of the boxes touching each bin_size x bin_size bin of each image.
 * planes: a = ga, b = gb */
int kg_crop_grad_reduce(const void* ga, int lda, const void* gb, int ldb, long rows_a, const int* boxtab, const int* bin_start,
                        const int* bin_boxes, int bin_size, int N, int H, int W, int C, float* out, void* out_rows, int ldout,
                        const kg_planes_t* planes, void* stream);   /* exactly one of out (fp32) / out_rows (split-bf16 rows, planes y) */


/* ---- mask paste-back of the inference driver (test.py:127-157): cv2.resize(patch, box size) -> paste into a zero
 * (input_h, input_w) mask -> cv2.resize(mask, (image_w, image_h)) -> mask >= seg_thresh, for all detections in one launch.
 * flat = forward_seg's fp32 patch probabilities; dets = device int32 [nd][8] {patch offset, patch h, patch w, y1, x1, y2, x2, 0}
 * (box rounded / clamped as test.py:138-141); out = [nd][image_h][image_w] float32 (out_is_u8 = 0) or bytes.  The interpolation
 * is OpenCV's published generic INTER_LINEAR float path (oracle/paste.py) ---- */
int kg_mask_paste(const float* flat, const int* dets, int nd, int input_h, int input_w, int image_h, int image_w,
                  float seg_thresh, void* out, int out_is_u8, void* stream);

/* ---- gradient scale of the half-precision backward pass (csrc/gradscale.hip) ----
 * kg_grad_scale: out[0] = S = 2^(target_log2 - e), out[1] = 1 / S, where max |v| over the n <= 24 fp32 device tensors
 * ptrs[i][0 .. counts[i]) (host arrays of device pointers / counts) = f * 2^e, f in [0.5, 1); S = 1 when the maximum is 0 or not
 * finite.  probs (optional host array, entries may be NULL): sigmoid outputs -- the value taken is then v * q * (1 - q), the gradient
 * w.r.t. the logit (what kg_grad_pack stores).  scratch: 2 zero-initialised unsigned on the device (left zeroed).  The tensors are the gradients of the loss w.r.t.
 * the network outputs (12 head maps + the seg probabilities: what `loss.backward()` hands to KGnet.forward's node, train.py:153).
 * kg_scale_tensors: multiplies njobs fp32 tensors by a device scalar each in ONE launch: jobs = device array of 32-byte records
 * {float* p; long n; const float* scale; int blk0; int pad;} (blk0 = first workgroup of the job, 4096 elements per workgroup).
 * kg_rows_rescale (re-normalisation point of the backward pass: a complete gradient rows tensor, csrc/norm_pool.hip): r = the power of
 * two <= 1 that brings max |g| of the rows tensor back into [2^(t-1), 2^t) when it exceeds 2^t (r = 1 otherwise: the scale only ever goes
 * down); g *= r in place; cum_out = {cum_in[0] * r, 1 / (cum_in[0] * r)}, r_out[0] = r (all on the device).  kg_rows_scale: rows *= *r (* *r2:
 * a gradient written under an earlier scale is converted with scale_now and 1 / scale_then) in place; kg_rows_scale_multi: the same for
 * up to 8 rows tensors in one launch: desc = host int64 [n][6] rows {pointer, ld, M, C, planes, plane stride}.  planes: a = g. */
int kg_grad_scale(const void* const* ptrs, const void* const* probs, const long* counts, int n, int target_log2, void* scratch, float* out, void* stream);
int kg_scale_tensors(const void* jobs, int njobs, int total_blocks, int* nonfinite, void* stream);   /* nonfinite: optional device int, set to 1 on an inf / NaN result */
int kg_rows_rescale(void* g, int ld, long M, int C, int target_log2, const float* cum_in, float* cum_out, float* r_out, void* scratch,
                    const kg_planes_t* planes, void* stream);
int kg_rows_scale(void* g, int ld, long M, int C, const float* r, const float* r2, const kg_planes_t* planes, void* stream);   /* *= *r * (r2 ? *r2 : 1) */
int kg_rows_scale_multi(const long* desc, int n, const float* r, const float* r2, void* stream);

#ifdef __cplusplus
}
#endif
#endif

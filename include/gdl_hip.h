/*
 * gdl_hip.h -- C ABI of libgdl_hip.so: the MI355X (gfx950) implementation of the
 * DGL audio-visual training step of shicaiwei123/ICCV2025-GDL (main_dgl.py:90-162).
 *
 * Conventions (every entry point):
 *   - plain C, raw DEVICE pointers, explicit sizes; no torch / C++ types;
 *   - returns 0 on success, a GDL_ERR_* code otherwise; gdl_last_error() gives a
 *     thread-local message; nothing throws across the boundary;
 *   - never allocates or frees device memory: workspaces are caller-provided and
 *     sized by the *_workspace_bytes query;
 *   - never synchronises the device: work is enqueued on the hipStream_t passed
 *     (as void*); the library keeps no pointer after return except inside an
 *     explicitly created gdl_encoder_t / gdl_optim_t object.
 *
 * Activations inside the library are NHWC in `dtype` (GDL_BF16 for the benchmark
 * configuration, GDL_F32 for the exact-f32 parity mode).  Parameters, gradients,
 * BatchNorm statistics, pooled features, logits and losses are float32 in the
 * reference's own layouts (conv weight [K][C][R][S], linear weight [out][in]).
 *
 * Reference interface each group replaces is cited as file:line into
 * /root/reference (the PyTorch operators the reference calls there).
 */
#ifndef GDL_HIP_H
#define GDL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDL_API __attribute__((visibility("default")))

enum { GDL_OK = 0, GDL_ERR_ARG = 1, GDL_ERR_HIP = 2, GDL_ERR_STATE = 3, GDL_ERR_WORKSPACE = 4 };
enum { GDL_F32 = 0, GDL_BF16 = 1 };
enum { GDL_AUDIO = 0, GDL_VISUAL = 1 };

GDL_API const char* gdl_last_error(void);
GDL_API int gdl_version(void);
/* number of compute units / device name of the current device (host query) */
GDL_API int gdl_device_info(int* cu_count, char* name, int name_len);

/* ------------------------------------------------------------------ convolution
 * nn.Conv2d(bias=False): backbone.py:20-23 (3x3 s1/s2 p1), :26-28 (1x1 s2),
 * :96-101 (7x7 s2 p3 stem).  Implicit GEMM on MFMA.
 *
 * gdl_conv_fwd:  x NHWC [N][H][W][C] (dtype), w_krsc [K][R][S][C] (dtype, from
 *   gdl_pack_weight) -> y NHWC [N][P][Q][K] (dtype).  C % (128/sizeof(dtype)) == 0,
 *   K % 64 == 0.  If bn_partial != NULL the epilogue also writes per-channel
 *   (sum, sum of squares) of the STORED values of each M-tile:
 *   bn_partial[tile][K][2] float, tile < gdl_conv_bn_tiles(...).
 * gdl_conv_dgrad: dy [N][P][Q][K], w_crsk [C][R][S][K] -> dx [N][H][W][C]
 *   (+ addend, same shape as dx, may alias dx; NULL for none).
 * gdl_conv_wgrad: dy, x -> dw [K][C][R][S] float32 (overwritten).  `ws` holds
 *   split-K partials; size from gdl_conv_wgrad_workspace_bytes.
 */
/* Gather tables: every conv kernel addresses its gathered operand as table[m].off0 + delta[tap]
 * (8 bytes per GEMM row: byte offset of tap 0 + a validity bit per tap), so the K-loops carry no
 * division / multiplication / bounds arithmetic.  Built once per geometry:
 *   mode GDL_GATHER_FWD   rows = output pixels; used by gdl_conv_fwd and gdl_conv_wgrad
 *   mode GDL_GATHER_DGRAD rows = input pixels;  used by gdl_conv_dgrad */
enum { GDL_GATHER_FWD = 0, GDL_GATHER_DGRAD = 1 };
GDL_API size_t gdl_conv_table_bytes(int mode, int N, int H, int W, int R, int S, int stride, int pad);
GDL_API int gdl_conv_build_table(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride,
                                 int pad, void* table, void* stream);
GDL_API int gdl_conv_bn_tiles(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
GDL_API int gdl_conv_fwd(int dtype, const void* x, const void* w_krsc, void* y, float* bn_partial, const void* table,
                         int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* stream);
GDL_API int gdl_conv_dgrad(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend,
                           const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                           void* stream);
/* ReLU backward folded into the data gradient (BasicBlock: out = relu(bn2(conv2(..)) + identity), backbone.py:62-66).
 * gdl_bn_act_bits = gdl_bn_act(relu = 1) that also stores the sign bits of its output, one byte per 16-byte vector
 * (bit e = element e > 0); gdl_conv_dgrad_relu = gdl_conv_dgrad whose stored result (dx + addend) is zeroed where the
 * bit of the tensor it is the gradient of is 0 -- the next block's backward then needs neither that tensor nor a
 * separate masking pass. */
GDL_API int gdl_bn_act_bits(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                            const float* res_scale, const float* res_shift, void* out, uint8_t* relu_bits, size_t M, int C,
                            void* stream);
GDL_API int gdl_conv_dgrad_relu(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend,
                                const uint8_t* relu_bits, const void* table, int N, int H, int W, int C, int K, int R, int S,
                                int stride, int pad, void* stream);
/* gdl_conv_fwd with the epilogue's optional extras: y = conv(x, w) + bias[K] (float32) + addend (like y), and
 * gelu_out = gelu(y) (exact erf form, of the value as stored) -- nn.Linear with its bias, the residual connection of a Swin
 * block, Mlp.fc1 + act in one launch (swin_transformer.py:26-42, 150-152, 287-290), R = S = 1. */
GDL_API int gdl_conv_fwd_bias(int dtype, const void* x, const void* w_krsc, void* y, const float* bias, const void* addend,
                              void* gelu_out, const void* table, int N, int H, int W, int C, int K, int R, int S, int stride,
                              int pad, void* stream);
/* First block of layers 2-4 (backbone.py:119-124, 141-146: conv1 is 3x3 stride 2, the shortcut a 1x1 stride-2 conv +
 * BatchNorm): the gradient of the block input is the sum of two data gradients,
 *   dx = conv1^T(dy [N][P][Q][K], w_crsk [C][3][3][K]) + downsample^T(dy_ds [N][P][Q][K], w_ds_ck [C][K]),
 * computed in ONE launch (the shortcut is one more tap of the (even, even) input pixels); relu_bits optional as in
 * gdl_conv_dgrad_relu.  `table` = the GDL_GATHER_DGRAD table of the 3x3 stride-2 pad-1 geometry. */
GDL_API int gdl_conv_dgrad_ds(int dtype, const void* dy, const void* w_crsk, const void* dy_ds, const void* w_ds_ck, void* dx,
                              const uint8_t* relu_bits, const void* table, int N, int H, int W, int C, int K, void* stream);
/* BatchNorm-backward reductions in the producing data gradient's epilogue (round 3).  BasicBlock backward (backbone.py:52-66
 * through autograd): the gradient dx a data gradient writes flows into a BatchNorm backward, which first needs per channel
 *   sum g'  and  sum g' * (y - mean) * rstd,   g' = dx as stored (after the addend and the relu_bits mask),
 * y = that BatchNorm's saved input.  gdl_conv_dgrad_bn = gdl_conv_dgrad(_relu) whose epilogue also leaves those sums per M-tile,
 * partial[tile][C][2] (tile < gdl_conv_dgrad_bn_tiles(...)), ready for gdl_bn_bwd_finalize -- the separate reduce pass over
 * g and y (gdl_bn_bwd_reduce: two tensor reads) disappears.  y2 / mean2 / rstd2 / partial2 (all or none): a second BatchNorm
 * fed by the same gradient (the downsample branch's; not with the 64 -> 64 channel persistent kernel). */
GDL_API int gdl_conv_dgrad_bn_tiles(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
GDL_API int gdl_conv_dgrad_bn(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend,
                              const uint8_t* relu_bits, const void* table, int N, int H, int W, int C, int K, int R, int S,
                              int stride, int pad, const void* y, const float* mean, const float* rstd, float* partial,
                              const void* y2, const float* mean2, const float* rstd2, float* partial2, void* stream);
/* Split-K of the 3x3 stride-1 "slab" convolutions (round 3).  Where a layer's output tiles would leave most CUs idle (layer 4
 * of both encoders: 3 456 / 9 408 GEMM rows against a 4 608-deep reduction; the audio layer 3), 2 or 4 blocks share a tile, each
 * multiplies a slice of the input channels and leaves fp32 accumulators in `split_ws`; a finish kernel folds them in fixed order
 * (run-to-run identical) and applies the epilogue -- rounding, addend, ReLU bits, BatchNorm statistics / BatchNorm-backward sums,
 * the same partial-row counts as the unsplit forms.  gdl_conv_split_workspace_bytes: bytes a geometry needs (0: it does not
 * split; dgrad = 0 forward, 1 data gradient).  The _split entry points equal gdl_conv_fwd / gdl_conv_dgrad_bn with that workspace
 * (NULL or too small: refused / unsplit).  bf16 only; results differ from the unsplit kernel by fp32 summation order. */
GDL_API size_t gdl_conv_split_workspace_bytes(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                              int dgrad);
GDL_API int gdl_conv_fwd_split(int dtype, const void* x, const void* w_krsc, void* y, float* bn_partial, const void* table, int N,
                               int H, int W, int C, int K, int R, int S, int stride, int pad, void* split_ws,
                               size_t split_ws_bytes, void* stream);
GDL_API int gdl_conv_dgrad_bn_split(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend,
                                    const uint8_t* relu_bits, const void* table, int N, int H, int W, int C, int K, int R, int S,
                                    int stride, int pad, const void* y, const float* mean, const float* rstd, float* partial,
                                    const void* y2, const float* mean2, const float* rstd2, float* partial2, void* split_ws,
                                    size_t split_ws_bytes, void* stream);
/* Mlp backward in the data gradient's epilogue (round 3; /root/reference/models/swin_transformer.py:32-47, Mlp.forward
 * fc1 -> act -> fc2 through autograd): gdl_conv_dgrad_gelu = gdl_conv_dgrad whose epilogue multiplies the stored value by
 * gelu'(u[row][c]) (u laid out like dx: fc1's biased output, as gdl_conv_fwd_bias left it) and adds the column sums of dx AS
 * STORED -- fc1's bias gradient -- to `acc`, int64 [C][2] fixed-point accumulators the caller zeroes: acc[2c] += round(sum *
 * scale) per M-tile with device-scope integer atomics (associative: bit-identical from run to run).  acc[2c + 1] is the channel's overflow / NaN MARK and must be zeroed with the rest: a tile sum that does not fit the channel's share of 2^62 (or is NaN / inf) sets it instead of being added.  Choose scale = 2^(62 - h - ceil(log2(rows))) for |mean| < 2^h.
 * gdl_acc_to_float: out[c] = acc[2c] * inv_scale, or NaN when acc[2c + 1] != 0 (a marked channel) -- not a reader of [sum, sumsq] pairs.  Replaces gdl_swin_colsum(g, u): three passes over the widest tensor of a
 * block. */
GDL_API int gdl_conv_dgrad_gelu(int dtype, const void* dy, const void* w_crsk, void* dx, const void* u, void* acc, double scale,
                                const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                void* stream);
GDL_API int gdl_acc_to_float(const void* acc, int n, double inv_scale, float* out, void* stream);
GDL_API size_t gdl_conv_wgrad_workspace_bytes(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride,
                                              int pad);
GDL_API int gdl_conv_wgrad(int dtype, const void* dy, const void* x, float* dw_kcrs, const void* table, int N, int H,
                           int W, int C, int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes,
                           void* stream);
/* float32 [K][C][R][S] -> dtype [K][R][S][C] (w_krsc) and dtype [C][R][S][K] (w_crsk); either may be NULL */
GDL_API int gdl_pack_weight(int dtype, const float* w_kcrs, void* w_krsc, void* w_crsk, int K, int C, int R, int S,
                            void* stream);


/* Direct (implicit-GEMM) stem, backbone.py:96-101,166 (7x7 stride 2 pad 3) -- what the encoder engine runs.  The reference's
 * input tensor (float32 [B][Cin][T][H][W] visual / [B][1][H][W] audio, read in place through its strides) is copied once
 * into a zero-padded channels-last image with 4 channels per pixel,
 *     xp [B*T][H+6][W+8][4] dtype   (gdl_stem_pad_bytes bytes),
 * in which a filter row of an output pixel is 8 consecutive pixels (64 bytes bf16 / 128 bytes f32).
 * gdl_pack_stem_rows: float32 [64][Cin][7][7] -> dtype [64][taps][IC] (gdl_stem_weight_bytes bytes;
 * one 128-byte K-step = one filter row in f32, two in bf16; padding slots are zero).
 * gdl_stem_build_table: the gather table of the stem (gdl_stem_table_bytes bytes), built once per shape.
 * gdl_stem_conv_fwd: y NHWC [B*T][P][Q][64] dtype (+ BatchNorm partials as gdl_conv_fwd,
 *   tiles = gdl_stem_conv_bn_tiles).  gdl_stem_conv_wgrad: dw [64][Cin][7][7] float32 from dy [M][64]. */
GDL_API size_t gdl_stem_pad_bytes(int dtype, int n_img, int H, int W);
GDL_API size_t gdl_stem_weight_bytes(int dtype);
GDL_API size_t gdl_stem_table_bytes(int n_img, int H, int W);
GDL_API int gdl_stem_pad(int dtype, const float* x, void* xp, int B, int Cin, int T, int H, int W, void* stream);
GDL_API int gdl_pack_stem_rows(int dtype, const float* w, void* wp, int Cin, void* stream);
GDL_API int gdl_stem_build_table(int dtype, int n_img, int H, int W, void* table, void* stream);
GDL_API int gdl_stem_conv_bn_tiles(int dtype, int n_img, int H, int W);
GDL_API int gdl_stem_conv_fwd(int dtype, const void* xp, const void* wp, void* y, float* bn_partial, const void* table,
                              int n_img, int H, int W, int Cin, void* stream);
GDL_API size_t gdl_stem_conv_wgrad_workspace_bytes(int n_img, int H, int W);
GDL_API int gdl_stem_conv_wgrad(int dtype, const void* dy, const void* xp, float* dw, const void* table, int n_img, int H,
                                int W, int Cin, void* ws, size_t ws_bytes, void* stream);
/* Fused stem backward (round 5): gdl_maxpool_bn_bwd_apply (below) and gdl_stem_conv_wgrad in ONE launch -- the max-pool gather, the
 * ReLU mask and the BatchNorm-backward apply of /root/reference/models/backbone.py:104-106 reversed, feeding the weight gradient
 * of the 7x7/2 stem convolution (backbone.py:97-101) tile by tile through LDS: the gradient of the stem output (the largest
 * activation) is never stored.  dout / idx: gradient and arg-max codes of the POOLED map [n_img*P'*Q'][64] (P' x Q' the pooled
 * size of the (H-1)/2+1 x (W-1)/2+1 stem output), y: the stem output, scale .. coef as for gdl_maxpool_bn_bwd_apply, xp: the
 * padded input of gdl_stem_pad, dw [64][Cin][7][7] float32; workspace as gdl_stem_conv_wgrad_workspace_bytes.  bf16 with stem
 * output rows of at least 64 pixels only (gdl_stem_bwd_fused_ok; other shapes run the two launches).  The result is
 * bit-identical to the two-launch form (same element arithmetic, same stages, same fold). */
GDL_API int gdl_stem_bwd_fused_ok(int dtype, int W);
GDL_API int gdl_stem_bwd_fused(int dtype, const void* dout, const uint8_t* idx, const void* y, const float* scale,
                               const float* shift, const float* save_mean, const float* save_rstd, const float* gamma,
                               const float* coef, const void* xp, float* dw, int n_img, int H, int W, int Cin, void* ws,
                               size_t ws_bytes, void* stream);

/* layout conversion at the module boundary: NHWC dtype <-> NCHW float32 */
GDL_API int gdl_nhwc_to_nchw_f32(int dtype, const void* x, float* y, int N, int H, int W, int C, void* stream);
GDL_API int gdl_nchw_f32_to_nhwc(int dtype, const float* x, void* y, int N, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------ BatchNorm2d (+ReLU, +residual)
 * nn.BatchNorm2d(eps 1e-5, momentum 0.1): backbone.py:45,48,104,144; nn.ReLU and
 * `out += identity`: backbone.py:46,57,65-66,105.
 *
 * gdl_bn_finalize_train: reduces conv-epilogue partials [tiles][C][2] over `count`
 *   elements per channel -> save_mean, save_rstd (biased variance), scale =
 *   gamma*rstd, shift = beta - mean*scale; updates running_mean / running_var
 *   (unbiased) with `momentum` and increments *num_batches_tracked (int64) when
 *   the pointers are non-NULL.
 * gdl_bn_finalize_eval: scale/shift from the running statistics.
 * gdl_bn_act: out = [relu]( y*scale+shift + residual ), residual = none |
 *   res (raw tensor) | res*res_scale+res_shift (downsample branch).
 * gdl_bn_stats: per-channel partials of an existing tensor (when the producer
 *   was not gdl_conv_fwd); same partial format, tiles = gdl_bn_stats_tiles(M).
 */
GDL_API int gdl_bn_stats_tiles(int M);
GDL_API int gdl_bn_stats(int dtype, const void* y, float* partial, int M, int C, void* stream);
GDL_API int gdl_bn_finalize_train(const float* partial, int tiles, int C, double count, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                  int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* scale,
                                  float* shift, void* stream);
GDL_API int gdl_bn_finalize_eval(int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                                 const float* running_var, float* scale, float* shift, void* stream);
GDL_API int gdl_bn_act(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                       const float* res_scale, const float* res_shift, int relu, void* out, size_t M, int C,
                       void* stream);
/* backward.  g = upstream gradient w.r.t. the BN output (after an optional fused
 * ReLU mask: relu_mask != 0 applies (y*scale+shift > 0) to g).
 * gdl_bn_bwd_reduce -> partial[blocks][C][2]; gdl_bn_bwd_finalize -> dgamma, dbeta
 * (float32, overwritten) and coef[2][C] = {dbeta/M, dgamma/M};
 * gdl_bn_bwd_apply: dy = gamma*rstd*(g - coef0 - xhat*coef1), may alias g. */
GDL_API int gdl_bn_bwd_blocks(size_t M, int C);
GDL_API int gdl_bn_bwd_reduce(int dtype, const void* g, const void* y, const float* scale, const float* shift,
                              const float* save_mean, const float* save_rstd, int relu_mask, float* partial, size_t M,
                              int C, void* stream);
GDL_API int gdl_bn_bwd_finalize(const float* partial, int blocks, int C, double count, float* dgamma, float* dbeta,
                                float* coef, void* stream);
GDL_API int gdl_bn_bwd_apply(int dtype, const void* g, const void* y, const float* scale, const float* shift,
                             const float* save_mean, const float* save_rstd, const float* gamma, const float* coef,
                             int relu_mask, void* dy, size_t M, int C, void* stream);
/* dx = dy * (out > 0)   (may alias dy) */
GDL_API int gdl_relu_bwd(int dtype, const void* dy, const void* out, void* dx, size_t n, void* stream);

/* ------------------------------------------------------------------ pooling
 * nn.MaxPool2d(3,2,1): backbone.py:106, fused with the stem's BN+ReLU:
 *   out[n,p,q,c] = max over the window of relu(y*scale+shift); idx (uint8, 0..8)
 *   records the first maximum in row-major window order.
 *   ymax (optional, may be NULL) receives the RAW y at that position.
 * gdl_maxpool_bwd gathers: dx[n,h,w,c] = sum of dout over windows whose idx
 *   points at (h,w).
 * gdl_maxpool_bn_bwd_apply: the stem's backward (backbone.py:104-106 reversed) in one
 *   pass, without storing the gathered gradient g0 = maxpool_bwd(dout):
 *     dy[pos] = gamma*rstd*( (y*scale+shift > 0 ? g0 : 0) - coef0 - xhat(y)*coef1 );
 *   coef comes from gdl_bn_bwd_reduce(relu_mask=1) over the POOLED pair (dout, ymax)
 *   [sum_pos g0'*xhat(y[pos]) == sum_windows dout'*xhat(ymax)] and gdl_bn_bwd_finalize
 *   with count = N*H*W (the stem-output pixel count).
 * Global average pools of basic_model.py:73-82: x [B*T][HW][C] dtype -> feat
 *   [B][C] float32 (mean over T*HW); backward broadcasts dfeat/(T*HW). */
GDL_API int gdl_bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out,
                                    uint8_t* idx, void* ymax, int N, int H, int W, int C, void* stream);
GDL_API int gdl_maxpool_bn_bwd_apply(int dtype, const void* dout, const uint8_t* idx, const void* y, const float* scale,
                                     const float* shift, const float* save_mean, const float* save_rstd,
                                     const float* gamma, const float* coef, void* dy, int N, int H, int W, int C,
                                     void* stream);
GDL_API int gdl_maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C,
                            void* stream);
GDL_API int gdl_avgpool_fwd(int dtype, const void* x, float* feat, int B, int T, int HW, int C, void* stream);
GDL_API int gdl_avgpool_bwd(int dtype, const float* dfeat, void* dx, int B, int T, int HW, int C, void* stream);

/* ------------------------------------------------------------------ fusion head + loss
 * ConcatFusion_DGL.forward (fusion_modules.py:51-59) and ConcatFusion.forward
 * (:38-42); nn.CrossEntropyLoss (main_dgl.py:71,102-104).  All float32.
 *   x, y: [B][512] pooled audio / visual features; W [n][1024]; b [n].
 * gdl_head_concat_fwd: out = [x,y]W^T+b, x_out = [x,0]W^T+b, y_out = [0,y]W^T+b
 *   (x_out / y_out may be NULL for the non-DGL head).
 * gdl_head_concat_bwd: autograd of the above for upstream gradients g_x_out,
 *   g_y_out, g_out (each may be NULL):
 *     dx = g_x_out W[:, :512] (+ g_out W[:, :512] if out_reaches_xy),
 *     dy = g_y_out W[:, 512:] (+ ...),
 *     dW = g_out^T [x,y] + (uni_in_dw ? g_x_out^T [x,0] + g_y_out^T [0,y] : 0), db likewise.
 *   DGL (detach + dropped head grads, main_dgl.py:110-122): out_reaches_xy = 0,
 *   uni_in_dw = 0.  Plain autograd of one backward call: uni_in_dw = 1.
 * gdl_softmax_ce: loss[0] = mean CE; dlogits = scale*(softmax-onehot)/B (may be NULL). */
GDL_API int gdl_head_concat_fwd(const float* x, const float* y, const float* W, const float* b, float* out,
                                float* x_out, float* y_out, int B, int n_classes, void* stream);
GDL_API int gdl_head_concat_bwd(const float* x, const float* y, const float* W, const float* g_x_out,
                                const float* g_y_out, const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx,
                                float* dy, float* dW, float* db, int B, int n_classes, void* stream);
/* One modality's auxiliary path of a DGL concat / sum head in ONE launch (main_dgl.py:102-122: an encoder learns from its own
 * unimodal loss only, so its backward can start as soon as ITS forward is done):
 *   u = f Wp^T + bp,  d(u) = scale*(softmax(u) - onehot(labels))/B,  df = d(u) Wp
 * f [B][512], Wp = the modality's 512 columns (row stride ldw floats: W / W + 512 with ldw 1024 for ConcatFusion_DGL,
 * fc_x / fc_y with ldw 512 for SumFusion_DGL), bp [n].  df is bit-identical to gdl_head_*_fwd + gdl_softmax_ce +
 * gdl_head_*_bwd's dx / dy with the DGL flags. */
GDL_API int gdl_head_uni_dfeat(const float* f, const float* Wp, int ldw, const float* bp, const int64_t* labels, float scale,
                               float* df, int B, int n_classes, void* stream);
/* ... for a modality whose features are `width` wide (512, 768 or 1024: the 768 Swin-T features of ConcatFusion_Swin's DGL form,
 * Wp = W + 512 with ldw = 512 + 768); df bit-identical to gdl_head_concat_xy_fwd + gdl_softmax_ce3 + gdl_head_concat_xy_bwd. */
GDL_API int gdl_head_uni_dfeat_w(const float* f, const float* Wp, int ldw, const float* bp, const int64_t* labels, float scale,
                                 float* df, int B, int n_classes, int width, void* stream);
/* The same head with unequal feature widths, W [n][x_dim + y_dim] (512 audio + 768 Swin features; the reference's
 * ConcatFusion_Swin, fusion_modules.py:79-88, in its DGL form :45-59): same contract as the two calls above. */
GDL_API int gdl_head_concat_xy_fwd(const float* x, const float* y, const float* W, const float* b, float* out, float* x_out,
                                   float* y_out, int B, int n_classes, int x_dim, int y_dim, void* stream);
GDL_API int gdl_head_concat_xy_bwd(const float* x, const float* y, const float* W, const float* g_x_out, const float* g_y_out,
                                   const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dW,
                                   float* db, int B, int n_classes, int x_dim, int y_dim, void* stream);
/* FiLM_DGL (fusion_modules.py:126-178; SURVEY next row N2): fc: Linear(512*512, 512) -- a 134 M-parameter bilinear
 * form per output, h[b][k] = u_b^T W_k v_b + bias_k -- and fc_out: Linear(512, n):
 *   out = fc_out(fc(x.detach() (x) y.detach())),  x_out = fc_out(fc(x (x) x)),  y_out = fc_out(fc(y (x) y)).
 * The outer products are never materialised; the contractions over fc.weight (537 MB) run as 1x1 convolutions /
 * weight gradients of this library in exact-f32 mode.  B <= 512 per call (the workspace grows with B: 0.47 GiB at 64,
 * 3.5 GiB at 512; gdl_head_film_workspace_bytes returns 0 beyond).  hidden: [3][B][512] float32 (h_x, h_f,
 * h_y), produced by fwd and consumed, with the SAME untouched workspace, by bwd.  bwd: any upstream gradient may
 * be NULL; dx/dy, dWfc/dbfc, dWo/dbo are optional pairs; uni_in_dw as for the other heads (0 in the DGL step). */
GDL_API size_t gdl_head_film_workspace_bytes(int B);
GDL_API int gdl_head_film_fwd(const float* x, const float* y, const float* Wfc, const float* bfc, const float* Wo,
                              const float* bo, float* hidden, float* out, float* x_out, float* y_out, int B, int n_classes,
                              void* ws, size_t ws_bytes, void* stream);
GDL_API int gdl_head_film_bwd(const float* x, const float* y, const float* Wfc, const float* Wo, const float* hidden,
                              const float* g_x_out, const float* g_y_out, const float* g_out, int uni_in_dw, float* dx,
                              float* dy, float* dWfc, float* dbfc, float* dWo, float* dbo, int B, int n_classes, void* ws,
                              size_t ws_bytes, void* stream);
/* GatedFusion_DGL (fusion_modules.py:213-250, x_gate = True; SURVEY next row N2): fc_x, fc_y: Linear(512, 512),
 * fc_out: Linear(512, n).  hx = fc_x(x), hy = fc_y(y) ([B][512], returned: saved for the backward);
 *   out = fc_out(sigmoid(hx.detach()) * hy.detach()),  x_out = fc_out(sigmoid(hx) * hx),  y_out = fc_out(sigmoid(hy) * hy).
 * Backward for upstream gradients on (x_out, y_out, out), any may be NULL: dx, dy (through fc_x / fc_y; `out` never
 * reaches them), dWo / dbo = out's contribution (+ the unimodal ones if uni_in_dw), optionally dW1, db1, dW2, db2
 * (NULL in the DGL step, whose script drops them and whose loss_f does not reach fc_x / fc_y).  ws: 2*B*512 floats. */
GDL_API int gdl_head_gated_fwd(const float* x, const float* y, const float* W1, const float* b1, const float* W2,
                               const float* b2, const float* Wo, const float* bo, float* hx, float* hy, float* out,
                               float* x_out, float* y_out, int B, int n_classes, void* stream);
GDL_API int gdl_head_gated_bwd(const float* x, const float* y, const float* hx, const float* hy, const float* W1,
                               const float* W2, const float* Wo, const float* g_x_out, const float* g_y_out,
                               const float* g_out, int uni_in_dw, float* dx, float* dy, float* dW1, float* db1, float* dW2,
                               float* db2, float* dWo, float* dbo, float* ws, int B, int n_classes, void* stream);
/* SumFusion_DGL (fusion_modules.py:16-30; SURVEY next row N2): fc_x, fc_y: Linear(512, n).
 *   x_out = fc_x(x), y_out = fc_y(y), out = fc_x(x.detach()) + fc_y(y.detach()).
 * Same flag meaning as the concat head; the sum head has two weight matrices [n][512] and two biases
 * (dbx = sum_b (g_out + uni*g_x_out), dby likewise). */
GDL_API int gdl_head_sum_fwd(const float* x, const float* y, const float* Wx, const float* bx, const float* Wy,
                             const float* by, float* out, float* x_out, float* y_out, int B, int n_classes, void* stream);
GDL_API int gdl_head_sum_bwd(const float* x, const float* y, const float* Wx, const float* Wy, const float* g_x_out,
                             const float* g_y_out, const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx,
                             float* dy, float* dWx, float* dbx, float* dWy, float* dby, int B, int n_classes, void* stream);
/* the three losses of the DGL step (loss_f, loss_a, loss_v; main_dgl.py:102-104) in one launch: losses[k] and dlogits_k
 * as gdl_softmax_ce would leave them for (logits_k, scale_k); any dlogits_k may be NULL */
GDL_API int gdl_softmax_ce3(const float* logits0, const float* logits1, const float* logits2, const int64_t* labels,
                            float scale0, float scale1, float scale2, float* losses, float* dlogits0, float* dlogits1,
                            float* dlogits2, int B, int n_classes, void* stream);
GDL_API int gdl_softmax_ce(const float* logits, const int64_t* labels, float scale, float* loss, float* dlogits, int B,
                           int n_classes, void* stream);
/* valid() (main_dgl.py:185-222) without its per-sample host loop: per-class counters (int64[n_classes] each, the
 * caller zeroes them once and accumulates over the batches of the validation set):
 *   num[label[i]] += 1;  acc[label[i]] += (argmax(out[i]) == label[i]);  likewise acc_a / acc_v from out_a / out_v
 * (first maximum wins, like np.argmax; softmax is monotone so it is not evaluated).  out_a/acc_a and out_v/acc_v
 * may be NULL together.  Accuracies are sum(acc)/sum(num) etc. as in main_dgl.py:222. */
GDL_API int gdl_eval_count(const float* out, const float* out_a, const float* out_v, const int64_t* labels, int B,
                           int n_classes, int64_t* num, int64_t* acc, int64_t* acc_a, int64_t* acc_v, void* stream);

/* ------------------------------------------------------------------ input pipeline
 * The device-side stages of the datasets' __getitem__:
 *   gdl_logspec: clip to [-1, 1], librosa.stft(n_fft, hop_length) with librosa's defaults (periodic Hann window of
 *     n_fft samples, center=True), log(|X| + 1e-7)  (dataset/CramedDataset.py:62-66: n_fft 512, hop 353;
 *     KSDataset.py:144-149 / VGGSoundDataset.py:117-122: 256 / 128).  wave: float32 [B][n_samples] (already resampled
 *     and tiled / cropped to its fixed length by the host); out: float32 [B][n_fft/2+1][gdl_logspec_frames()], the
 *     tensor the DataLoader yields.  pad_mode: how the n_fft/2 samples either side are made up -- librosa >= 0.10
 *     pads with zeros (GDL_PAD_CONSTANT), older releases reflect (GDL_PAD_REFLECT); the reference does not pin a version.
 *   gdl_frames_normalize: transforms.ToTensor() + Normalize(mean, std) (CramedDataset.py:77-81): uint8 [n_img][H][W][3]
 *     -> float32 [n_img][3][H][W], ((x / 255) - mean[c]) / std[c]; mean / std: 3 host floats each. */
#define GDL_PAD_CONSTANT 0
#define GDL_PAD_REFLECT 1
GDL_API int gdl_logspec_frames(int n_samples, int hop);
GDL_API int gdl_logspec(const float* wave, int B, int n_samples, int n_fft, int hop, int pad_mode, float* out, void* stream);
GDL_API int gdl_frames_normalize(const uint8_t* frames, int64_t n_img, int H, int W, const float* mean, const float* std,
                                 float* out, void* stream);

/* ------------------------------------------------------------------ clip + grad stats + SGD
 * clip_grad_norm_(params, 40, 2) (main_dgl.py:129), the logged
 * sum_p mean|grad_p| per encoder (:132-143) and optim.SGD(momentum, weight_decay)
 * (:249,154) over FLAT float32 arenas.  seg_offsets[nseg+1] (int64, element
 * offsets, host memory) delimit the parameters; seg_group[nseg] (int32, host) is
 * 0 = fusion head, 1 = audio_net, 2 = visual_net.
 * gdl_optim_grad_stats writes stats[0..3] = {total_norm (pre-clip), clip_coef,
 *   audio_grad_sum, visual_grad_sum (both post-clip)} on the device.
 * gdl_optim_sgd_step: g *= clip_coef*grad_scale (written back); d = g + wd*p;
 *   m = mu*m + d; p -= lr*m  (momentum arena starts at zero, which reproduces
 *   torch's first-step `buf = d`).
 * The object owns no device memory: gdl_optim_create only builds host tables; `ws` (gdl_optim_workspace_bytes, 16-byte
 * aligned, caller-owned) holds the descriptor tables in its head -- uploaded, ordered on `stream`, the first time
 * gdl_optim_grad_stats is called with that pointer -- and the per-chunk partial sums behind them.  Keep the workspace intact
 * between calls (or pass another pointer: the tables are uploaded again).  gdl_optim_bind_workspace (round 4) uploads the tables
 * unconditionally: call it whenever the workspace memory may have changed hands without its ADDRESS changing (a caching
 * allocator returns a freed block at the same address; another kernel scribbled over it), and before capturing
 * gdl_optim_grad_stats into a HIP graph -- the implicit first-use upload is a pageable-memory copy, which a capture refuses. */
typedef struct gdl_optim gdl_optim_t;
GDL_API int gdl_optim_create(gdl_optim_t** out, const int64_t* seg_offsets, const int32_t* seg_group, int nseg);
GDL_API void gdl_optim_destroy(gdl_optim_t* o);
GDL_API size_t gdl_optim_workspace_bytes(const gdl_optim_t* o);
GDL_API int gdl_optim_grad_stats(gdl_optim_t* o, const float* grads, float max_norm, float grad_scale, float* stats,
                                 void* ws, size_t ws_bytes, void* stream);
GDL_API int gdl_optim_bind_workspace(gdl_optim_t* o, void* ws, size_t ws_bytes, void* stream);
GDL_API int gdl_optim_stats_len(const gdl_optim_t* o); /* 4 + 2*nseg floats */
GDL_API int gdl_optim_sgd_step(gdl_optim_t* o, float* params, float* grads, float* momentum, const float* stats,
                               float grad_scale, float lr, float mu, float wd, void* stream);

/* ------------------------------------------------------------------ ResNet18 encoder engine
 * `resnet18(modality, args)` / ResNet.forward (backbone.py:75-201, 255-257) plus the
 * pooling glue of AVClassifier_DGL.forward (basic_model.py:73-82), as one planned
 * sequence of the kernels above on one stream.
 *
 * Parameter order everywhere below = named_parameters() order of the reference
 * module (60 tensors): conv1.weight, bn1.weight, bn1.bias, then per BasicBlock
 * conv1.weight, bn1.{weight,bias}, conv2.weight, bn2.{weight,bias}
 * [, downsample.0.weight, downsample.1.{weight,bias}].  BatchNorm buffer order =
 * the 20 BatchNorm layers in the same traversal.
 */
typedef struct gdl_encoder gdl_encoder_t;
#define GDL_ENC_NPARAMS 60
#define GDL_ENC_NBN 20
GDL_API int gdl_encoder_create(gdl_encoder_t** out, int modality, int dtype, int B, int T, int H, int W);
GDL_API void gdl_encoder_destroy(gdl_encoder_t* e);
/* enable != 0: gdl_encoder_backward forks the weight gradients (which are off the dy -> dx dependency chain)
 * onto an engine-owned side stream and joins it back into the caller's stream before returning control of
 * that stream, so the call keeps its single-stream semantics.  Off by default.  Measured on MI355X: a whole
 * step slows down 1.15-1.4x once more than FOUR streams carry work, or when a stream has a non-default
 * priority, so a caller running {main, audio, visual} streams enables this for the visual engine only (the
 * critical path) and not at all when a collective's stream is active as well. */
GDL_API int gdl_encoder_side_stream(gdl_encoder_t* e, int enable);
/* The same fork / join onto a stream of the CALLER's (NULL = the null stream) instead of an engine-owned one: no new
 * hardware queue.  Meant for a stream the caller already has and that idles during the step -- DGLTrainer hands the audio
 * engine the stream `step()` was called on, which only orders the step before and behind (round 4: 5.63 -> 5.43 ms per
 * step; the audio chain, a quarter of the work in ~100 small launches, had been the last to finish).  The engine enqueues
 * on that stream only inside gdl_encoder_backward(_phase) (between a wait for the engine's stream and an event the
 * engine's stream waits for), so work the caller puts there before / after the call is ordered as on one stream.  May be
 * called again with another stream; gdl_encoder_side_stream(e, 0) returns to none. */
GDL_API int gdl_encoder_borrow_side_stream(gdl_encoder_t* e, void* stream);
GDL_API size_t gdl_encoder_workspace_bytes(const gdl_encoder_t* e);
GDL_API int gdl_encoder_param_numel(const gdl_encoder_t* e, int64_t* numel /*[60]*/);
GDL_API int gdl_encoder_out_shape(const gdl_encoder_t* e, int* n_img, int* h, int* w);
/* bind the caller-owned workspace (device) and the parameter / gradient / buffer tables (host arrays of device ptrs) */
GDL_API int gdl_encoder_bind(gdl_encoder_t* e, void* workspace, size_t bytes);
GDL_API int gdl_encoder_set_params(gdl_encoder_t* e, const float* const* params, float* const* running_mean,
                                   float* const* running_var, int64_t* const* num_batches_tracked);
/* forward.  x: the reference's input tensor, float32 [B][Cin][T][H][W] contiguous.
 * training != 0: batch statistics, running-stat update, activations kept for backward.
 * feat_out: [B][512] float32 pooled features (may be NULL); fmap_nchw: [B*T][512][h][w]
 * float32, the tensor ResNet.forward returns (may be NULL). */
GDL_API int gdl_encoder_forward(gdl_encoder_t* e, const float* x, int training, float* feat_out, float* fmap_nchw,
                                void* stream);
/* backward of the last training forward.  Exactly one of dfeat ([B][512]) / dfmap_nchw
 * is non-NULL.  grads: 60 device pointers (float32, reference layouts), overwritten. */
GDL_API int gdl_encoder_backward(gdl_encoder_t* e, const float* dfeat, const float* dfmap_nchw, float* const* grads,
                                 void* stream);
/* The same backward in two calls, for data-parallel callers: phase 1 = the upstream gradient and layer4 -- when it
 * returns (in stream order) the last 15 gradient tensors, 8.4 M of the 11.2 M parameters, are final and their
 * all-reduce can overlap phase 2 (layer3 .. layer1 and the stem).  Phase 2 takes no dfeat / dfmap (pass NULL). */
GDL_API int gdl_encoder_backward_phase(gdl_encoder_t* e, int phase, const float* dfeat, const float* dfmap_nchw,
                                       float* const* grads, void* stream);
/* serial number of the last training forward (to detect stale activations) */
GDL_API int64_t gdl_encoder_forward_serial(const gdl_encoder_t* e);
/* Round 4: the training forward's BatchNorm statistics travel through 64-bit fixed-point accumulators with finite headroom
 * (mean |y| of a block's rows < 8192 / channel tiles).  A BatchNorm whose sums exceeded it -- or were NaN / inf -- gets NaN
 * statistics instead of wrapped integers; since ReLU turns NaN into 0 the activations may not show it, so this returns the number
 * of such BatchNorms in the last training forward (0 = fine; < 0 = -error code).  Synchronises `stream`.  A diverged
 * nn.BatchNorm2d of the reference shows as inf / NaN in its outputs (backbone.py:45-48,104,144); here ask this. */
GDL_API int gdl_encoder_bn_overflow(gdl_encoder_t* e, void* stream);

/* ------------------------------------------------------------------ measurement tap
 * Optional HIP-event timing of every kernel launch (off by default).  While enabled, each
 * launcher records an event pair on the launching stream and its algorithmic work (flops for
 * the MFMA-bound conv kernels, bytes for the HBM-bound ones).  gdl_prof_collect synchronises
 * the device and returns per-slot totals: launches[s], ms[s], work[s], s < gdl_prof_nslots().
 * gdl_prof_slot_bound: 1 = MFMA-bound (work in flops), 0 = HBM-bound (work in bytes). */
GDL_API int gdl_prof_enable(int on);
GDL_API int gdl_prof_enabled(void); /* 1 while the tap records (callers that replay captured graphs run eagerly then) */
/* tuning aid: in a -DGDL_TIMING build the conv kernels write per-block s_memtime stamps to buf[block][8]
 * (uint64); in the normal build this returns GDL_ERR_ARG */
GDL_API int gdl_debug_timing_buffer(void* buf);
/* record only launches of the kernel with this name (NULL or "" = all kernels) */
GDL_API int gdl_prof_set_filter(const char* name);
GDL_API int gdl_prof_nslots(void);
GDL_API const char* gdl_prof_slot_name(int slot);
GDL_API int gdl_prof_slot_bound(int slot);
GDL_API int gdl_prof_collect(int64_t* launches, double* ms, double* work);
/* Combined roofline (round 4; measurement only): the convolution launchers also state their algorithmic HBM bytes (operands once
 * + result).  After gdl_prof_collect, floor_ms[s] = the sum over slot s's launches of max(flop / peak_flops, bytes / peak_bytes)
 * -- what the launches would take at the peaks, each priced against the roof that binds it -- and bytes[s] their algorithmic
 * bytes (either pointer may be NULL).  gdl_prof_set_peaks: the two peaks (defaults: 2.5e15 flop/s bf16 MFMA, 8e12 B/s HBM3E). */
GDL_API int gdl_prof_set_peaks(double peak_flops, double peak_bytes);
GDL_API int gdl_prof_collect_floor(double* floor_ms, double* bytes);
/* Utilisation timeline (round 5; measurement only, tools/utilisation_timeline.py): the records the tap holds, in enqueue order,
 * one per launch -- slot[i], lane[i] (the launching stream, numbered in order of first appearance), start_ms[i] / end_ms[i]
 * relative to the earliest start among them (HIP events: the kernel's own begin / end for the hipExtLaunchKernelGGL taps),
 * work[i] (flops or bytes as gdl_prof_slot_bound says) and bytes[i] (algorithmic HBM bytes of an MFMA-bound launch, 0 = not
 * stated).  Call it BEFORE gdl_prof_collect (which clears the records).  Synchronises the device.  Returns the number of
 * records held (at most `cap` are written; any pointer may be NULL), < 0 = -error code. */
GDL_API int gdl_prof_timeline(int cap, int32_t* slot, int32_t* lane, double* start_ms, double* end_ms, double* work, double* bytes);

/* ---------------------------------------------------------------------------------------------------------------------
 * Swin visual encoder (/root/reference/models/swin_transformer.py, which the DGL script does not
 * reach -- main_dgl.py:236-240 -- so the Swin + DGL-head model is a new composition).  Tokens are rows [N*L][ld] of
 * `dtype`, ld = channels rounded up to a multiple of 64, padding columns zero.  Every nn.Linear (qkv, proj, fc1, fc2,
 * PatchMerging.reduction) and the 4x4/4 PatchEmbed convolution is a 1x1 gdl_conv_fwd / gdl_conv_dgrad / gdl_conv_wgrad on
 * zero-padded weights (gdl_swin_pack_matrix); the entry points below are everything else.  Vectors (bias, gamma, beta)
 * are float32 in the padded layout.
 *   gdl_swin_patch_gather   frames [B,3,T,H,W] f32 -> GEMM rows [B*T*(H/p)*(W/p)][64], columns (c,kh,kw)  (:463,:480)
 *   gdl_swin_bias_act       mode 0: y += b;  1: u = y + b, y = gelu(u);  2: y = y + b + res               (:26-42,:287-290)
 *   gdl_swin_drop_path      DropPath of a residual branch (timm.models.layers.drop_path, scale_by_keep; :218,:290,:293):
 *                           out = (res ? res : 0) + scale[row / L] * y, scale[frame] = 0 or 1 / keep_prob (drawn by the caller:
 *                           the reference draws them from torch's generator); forward with res, backward without; out may be y
 *   gdl_swin_ln_fwd / _bwd  nn.LayerNorm(eps 1e-5) over the C real channels; stats [M][2] = (mean, rstd); the backward
 *                           adds `add` (the residual branch's gradient) and returns [2][ld] = (d gamma, d beta)
 *   gdl_swin_colsum         db[ld] = column sums of g; with u != NULL first g <- g * gelu'(u) (in place)
 *   gdl_swin_attn_fwd/_bwd  WindowAttention incl. cyclic shift, mask and relative-position bias (:124-157,:222-285) on
 *                           QKV rows [q | k | v] (three ld-wide segments, head h at h*32); the backward returns dqkv and
 *                           d(relative_position_bias_table) [(2w-1)^2][heads]
 *   gdl_swin_merge          PatchMerging's 2x2 concatenation [N,H,W,C] -> [N,H/2,W/2,4C] (:336-344) / its adjoint
 *   gdl_swin_token_mean     AdaptiveAvgPool2d((1,1)) over the L tokens (:629-631) -> float32 [N][C], and its backward
 *   gdl_swin_pack_matrix    float32 [n][k] -> `dtype` [np][kp] (+ transposed [kp][np]), rows / columns in segments of
 *                           nseg / kseg stored at pitch nseg_pad / kseg_pad (QKV rows: 3 x 96 -> 3 x 128)
 *   gdl_swin_unpack_matrix  float32 padded [np][kp] -> float32 real [n][k] (weight gradients back to parameter shape)
 * ------------------------------------------------------------------------------------------------------------------- */
GDL_API int gdl_swin_patch_gather(int dtype, const float* x, void* a, int B, int T, int H, int W, int patch, void* stream);
/* Both gradients of y[M][N] = x[M][K] W^T (nn.Linear; swin_transformer.py:26-42 Mlp.fc1, :78-157 qkv) from ONE pass over dy:
 * dx[M][K] = dy . W (bf16) and dW[N][K] = dy^T . x (float32), for the shapes gdl_linear_bwd_ok accepts (bf16, K = 128, N = 384,
 * M >= 16384: stage 1 of Swin-T) -- what gdl_conv_dgrad + gdl_conv_wgrad (R = S = 1) compute in two passes over dy.  wT: the
 * weight as [K][N] (the layout gdl_swin_pack_matrix's dstT / gdl_conv_dgrad take).  ws: gdl_linear_bwd_workspace_bytes.
 * Kreal: the real input width (columns Kreal .. K - 1 of x are zero padding; dW gets zeros there).  db (optional, float32 [N]):
 * the bias gradient = column sums of dy -- only where Kreal <= 96, where it costs nothing (an all-ones operand in a padding
 * tile's place); an error otherwise (use gdl_swin_colsum). */
GDL_API int gdl_linear_bwd_ok(int dtype, size_t M, int K, int N);
GDL_API size_t gdl_linear_bwd_workspace_bytes(size_t M, int K, int N);
GDL_API int gdl_linear_bwd(int dtype, const void* dy, const void* x, const void* wT, void* dx, float* dw, float* db, void* ws,
                           size_t ws_bytes, size_t M, int K, int Kreal, int N, void* stream);
GDL_API int gdl_swin_drop_path(int dtype, const void* y, const void* res, const float* scale, void* out, size_t M, int L, int ld,
                               void* stream);
GDL_API int gdl_swin_bias_act(int dtype, void* y, const float* bias, void* u, const void* res, size_t M, int ld, int mode,
                              void* stream);
GDL_API int gdl_swin_ln_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, size_t M,
                            int C, int ld, void* stream);
GDL_API size_t gdl_swin_partial_bytes(int ld);
GDL_API int gdl_swin_ln_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const void* add,
                            void* dx, float* dgamma_dbeta, void* partial, size_t M, int C, int ld, void* stream);
/* the same with a third result row: dgamma_dbeta_colsum [3][ld], row 2 = column sums of dx as stored -- the bias gradient of the
 * Linear whose output gradient dx is (the residual stream's gradient feeds fc2 / proj / the patch embedding), instead of a
 * gdl_swin_colsum pass over dx.  3 * ld <= the width `partial` was sized for. */
GDL_API int gdl_swin_ln_bwd_colsum(int dtype, const void* dy, const void* x, const float* stats, const float* gamma,
                                   const void* add, void* dx, float* dgamma_dbeta_colsum, void* partial, size_t M, int C, int ld,
                                   void* stream);
GDL_API int gdl_swin_colsum(int dtype, void* g, const void* u, float* db, void* partial, size_t M, int ld, void* stream);
/* Deferred folds (round 6): gdl_swin_ln_bwd / _colsum with dgamma_dbeta == NULL and gdl_swin_colsum with db == NULL leave their
 * partial rows in `partial` -- gdl_swin_ln_bwd_rows(dtype, M, ld) rows of 2 * ld (3 * ld: _colsum) floats, gdl_swin_colsum_rows
 * (dtype, M, ld) rows of ld floats (both counts: 0 for bad arguments) -- and launch no fold; the caller gives every such call a `partial` buffer of its own and
 * folds all of them with ONE launch before the gradients are read: `descs` = device array of n_desc 32-byte records
 *   { const float* partial; float* out; int rows, width, blk0, stride; }
 * (out[j] = sum over the rows of partial[row * stride + j], j < width <= stride, in the order the per-call fold uses: bit-identical results;
 * blk0 = first block of the record, (width + 15) / 16 blocks per record, ascending; total_blocks = their sum).  The parameter
 * gradients of the reference's LayerNorms / Linear biases (swin_transformer.py:26-42,:78-157,:176-295) are only read by the
 * optimizer: 42 fold launches leave the Swin-T backward's chain. */
GDL_API int gdl_swin_ln_bwd_rows(int dtype, size_t M, int ld);
GDL_API int gdl_swin_colsum_rows(int dtype, size_t M, int ld);
GDL_API int gdl_swin_partial_reduce_batched(const void* descs, int n_desc, int total_blocks, void* stream);
GDL_API int gdl_swin_attn_fwd(int dtype, const void* qkv, const float* table, void* out, int n_img, int H, int W, int window,
                              int shift, int heads, int ld, void* stream);
GDL_API size_t gdl_swin_attn_bwd_workspace_bytes(int n_img, int H, int W, int window, int heads);
GDL_API int gdl_swin_attn_bwd(int dtype, const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable,
                              void* ws, int n_img, int H, int W, int window, int shift, int heads, int ld, void* stream);
GDL_API int gdl_swin_merge(int dtype, const void* src, void* dst, int N, int H, int W, int C, int ldx, int scatter, void* stream);
GDL_API int gdl_swin_token_mean(int dtype, const void* x, float* y, int N, int L, int C, int ld, void* stream);
GDL_API int gdl_swin_token_mean_bwd(int dtype, const float* dy, void* dx, int N, int L, int C, int ld, void* stream);
GDL_API int gdl_swin_pack_matrix(int dtype, const float* src, void* dst, void* dstT, int n, int k, int nseg, int nseg_pad,
                                 int kseg, int kseg_pad, void* stream);
/* every layout conversion of a step in one launch: `descs` = device array of n_desc 64-byte records
 *   { const float* src; void* dst; void* dstT; int n, k, nseg, nseg_pad, kseg, kseg_pad, np, kp; int dtype; int blk0; }
 * (np / kp = padded sizes, blk0 = first block of the record, 1024 elements per block, ascending; total_blocks = their sum);
 * dir 0 = gdl_swin_pack_matrix per record, dir 1 = gdl_swin_unpack_matrix per record (dstT, dtype ignored) */
GDL_API int gdl_swin_pack_batched(const void* descs, int n_desc, int total_blocks, int dir, void* stream);
GDL_API int gdl_swin_unpack_matrix(const float* src, float* dst, int n, int k, int nseg, int nseg_pad, int kseg, int kseg_pad,
                                   void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Gradient exchange of the data-parallel step over RCCL / xGMI (the reference's nn.DataParallel, main_dgl.py:244, as one
 * process per GPU; SURVEY 8(b) comm_init / allreduce_bucket / comm_destroy).  RCCL is bound at run time (dlopen), so the
 * library has no link-time dependency on it.  Bootstrap: rank 0 calls gdl_comm_unique_id and ships the 128 bytes to the
 * other ranks by any means (gdl/ddp.py uses the torch.distributed store); every rank then calls gdl_comm_init on ITS
 * device.  gdl_comm_allreduce_bucket sums `count` float32 gradients in place over all ranks, enqueued on `stream`
 * (collectives of one communicator execute in issue order: issue them in the same order on every rank).
 * ------------------------------------------------------------------------------------------------------------------- */
#define GDL_COMM_ID_BYTES 128
typedef struct gdl_comm gdl_comm_t;
GDL_API int gdl_comm_unique_id(void* id128);
GDL_API int gdl_comm_init(gdl_comm_t** out, int rank, int world, const void* id128);
GDL_API int gdl_comm_world(const gdl_comm_t* c);
GDL_API int gdl_comm_allreduce_bucket(gdl_comm_t* c, float* grads, size_t count, void* stream);
GDL_API int gdl_comm_destroy(gdl_comm_t* c);

#ifdef __cplusplus
}
#endif
#endif /* GDL_HIP_H */

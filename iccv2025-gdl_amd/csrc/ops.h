// ops.h -- internal (C++) launch interface of the kernels; the C ABI in api.cpp and the encoder
// engine in encoder.cpp are thin layers over these.
#pragma once
#include "bnacc.h"
#include "common.h"
#include "gather.h"

namespace gdl {

const char* last_error();

// argument sets of the finalize kernels (the *_pair forms finalize two BatchNorms of equal width in one launch)
struct BnFinTrain {
    const float* partial;
    int tiles, C;
    double count;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    int64_t* nbt;
    float *save_mean, *save_rstd, *scale, *shift;
};
struct BnFinBwd {
    const float* partial;
    int blocks, C;
    double count;
    float *dgamma, *dbeta, *coef;
};

// BatchNorm-backward reductions produced by a data gradient's epilogue (round 3): the launch that writes a gradient tensor
// g [rows][C] also leaves, per M-tile, the per-channel sums the BatchNorm backward of the layer(s) that tensor flows into needs
//   partial[tile][C][2] = { sum g', sum g' * (y - mean) * rstd }      (what bn_bwd_reduce / block_bwd_reduce produce in a pass
//   partial2[tile][C][2] = { sum g', sum g' * (y2 - mean2) * rstd2 }    of their own over g and y: two tensor reads each)
// over the STORED values g' of the rows it stores (after the addend and the relu_bits mask -- the ReLU between a BatchNorm and
// its consumer reaches the gradient through the sign bits bn_act leaves: relu(bn1(y1)) -> conv2, backbone.py:57-58).
// y2 / partial2: a second BatchNorm fed by the same gradient (the downsample branch's, backbone.py:62-64).
// tiles = conv_dgrad_tiles_m(...) rows.
struct BwdStats {
    const void* y;
    const float *mean, *rstd;
    float* partial;
    const void* y2;
    const float *mean2, *rstd2;
    float* partial2;
};
// split-K workspace of the slab convolutions (round 3): where plan_conv's tiles would leave CUs idle (layer 4 of both encoders,
// the audio layer 3) `ksplit` blocks share an output tile, each multiplies a slice of the input channels and leaves fp32
// accumulators in the workspace; a finish kernel folds them in fixed order and applies the epilogue (rounding, addend, ReLU
// bits, statistics).  conv_split_ws_bytes: what a geometry needs (0: that convolution does not split); nullptr / too small a
// buffer: no split.  bf16 only.
struct SplitWs {
    void* ptr;
    size_t bytes;
};
size_t conv_split_ws_bytes(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dgrad);
// linear_stream.hip: out[M][N] = A[M][K] W[N][K]^T (+ bias) (+ addend), optional gelu_out -- bf16, K = 128 / 192 (Swin Linears)
bool linear_stream_ok(int dtype, int M, int K, int N, bool has_addend);
int linear_stream_fwd(const void* A, const void* W, void* out, const void* addend, const float* bias, void* gelu_out, int M, int K,
                      int N, hipStream_t st);
// conv_igemm.hip
int conv_dgrad_tiles_m(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
int conv_tiles_m(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
int conv_fwd(int dtype, const void* x, const void* w_krsc, void* y, float* bn_partial, const void* table, int N, int H,
             int W, int C, int K, int R, int S, int stride, int pad, hipStream_t st, const BnAcc* sacc = nullptr,
             const SplitWs* split = nullptr);
int conv_fwd_bias(int dtype, const void* x, const void* w_krsc, void* y, const float* bias, const void* addend, void* gelu_out,
                  const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, hipStream_t st);
// relu_bits (optional): sign bits of the tensor whose gradient dx is (bn_act's relu_bits): dx = bit ? dx (+ addend) : 0
int conv_dgrad(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, const void* table, int N, int H,
               int W, int C, int K, int R, int S, int stride, int pad, hipStream_t st, const uint8_t* relu_bits = nullptr,
               const BwdStats* bw = nullptr, const SplitWs* split = nullptr);
// dx = dgrad(dy) * gelu'(u) with the column sums of dx (as stored) added to the fixed-point accumulators `acc` (bnacc.h)
int conv_dgrad_gelu(int dtype, const void* dy, const void* w_crsk, void* dx, const void* u, const BnAcc* acc, const void* table,
                    int N, int H, int W, int C, int K, int R, int S, int stride, int pad, hipStream_t st);
// out[c] = acc[2c] * inv_scale
int acc_to_float(const long long* acc, int n, double inv_scale, float* out, hipStream_t st);
// the 3x3 stride-2 pad-1 data gradient of a downsample block's conv1 with the 1x1 stride-2 data gradient of its shortcut
// convolution folded in: dx = dgrad(dy, w_crsk) + dgrad_1x1s2(dy_ds, w_ds_ck) (+ relu_bits), one launch, no addend pass
int conv_dgrad_ds(int dtype, const void* dy, const void* w_crsk, const void* dy_ds, const void* w_ds_ck, void* dx,
                  const void* table, int N, int H, int W, int C, int K, hipStream_t st, const uint8_t* relu_bits = nullptr,
                  const BwdStats* bw = nullptr);
// direct (implicit-GEMM) stem: layout.hip (padded NHWC4 input, row-wise weights), gather.hip (table),
// conv_igemm.hip (forward), conv_wgrad.hip (weight gradient)
size_t stem_pad_bytes(int dtype, int n_img, int H, int W);
int stem_taps(int dtype);
int stem_ic(int dtype);
// zero / zero_bytes (optional, 16-byte granular): a region the same launch clears (the encoder's BatchNorm accumulators)
int stem_pad(int dtype, const float* x, void* xp, int B, int Cin, int T, int H, int W, hipStream_t st, void* zero = nullptr,
             size_t zero_bytes = 0);
int pack_stem_rows(int dtype, const float* w, void* wp, int cin, hipStream_t st);
int conv_stem_tiles_m(int dtype, int n_img, int H, int W);
bool conv_fwd_persistent(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
bool conv_stem_persistent(int dtype, int W);
int conv_stem_fwd(int dtype, const void* xp, const void* wp, void* y, float* bn_partial, const void* table, int n_img, int H,
                  int W, int Cin, hipStream_t st, const BnAcc* sacc = nullptr);
size_t conv_stem_wgrad_ws_bytes(int n_img, int H, int W);
int conv_stem_wgrad(int dtype, const void* dy, const void* xp, float* dw, const void* table, int n_img, int H, int W, int Cin,
                    void* ws, size_t ws_bytes, hipStream_t st);
// fused stem backward (round 5): max-pool gather + ReLU mask + BatchNorm-backward apply + weight gradient, dy0 never stored
bool stem_bwd_fused_ok(int dtype, int W);
int stem_bwd_fused(const void* dz, const uint8_t* idx, const void* y0, const float* scale, const float* shift, const float* mean,
                   const float* rstd, const float* gamma, const float* coef, const void* xp, float* dw, int n_img, int H, int W,
                   int Cin, void* ws, size_t ws_bytes, hipStream_t st);
// conv_wgrad.hip
size_t conv_wgrad_ws_bytes(int M, int C, int K, int RS);
int conv_wgrad(int dtype, const void* dy, const void* x, float* dw, const void* table, int N, int H, int W, int C, int K,
               int R, int S, int stride, int pad, int Cout, void* ws, size_t ws_bytes, hipStream_t st);
// layout.hip
int pack_weight(int dtype, const float* w, void* krsc, void* crsk, int K, int C, int R, int S, hipStream_t st);
struct PackDescHost {  // mirrors layout.hip::PackDesc
    const float* w;
    void* krsc;
    void* crsk;
    int K, C, RS, blk0;
};
int pack_weights_batched(int dtype, const void* desc_dev, int ndesc, int total_blocks, double bytes, hipStream_t st);
int nhwc_to_nchw_f32(int dtype, const void* x, float* y, int N, int H, int W, int C, hipStream_t st);
int nchw_f32_to_nhwc(int dtype, const float* x, void* y, int N, int H, int W, int C, hipStream_t st);
// bn.hip
int bn_stats_tiles(int M);
int bn_stats(int dtype, const void* y, float* partial, int M, int C, hipStream_t st);
int bn_finalize_train_pair(const BnFinTrain& a, const BnFinTrain& b, float eps, float momentum, hipStream_t st);
int bn_bwd_apply2(int dtype, const void* g, const void* yA, const float* meanA, const float* rstdA, const float* gammaA,
                  const float* coefA, void* dyA, const void* yB, const float* meanB, const float* rstdB, const float* gammaB,
                  const float* coefB, void* dyB, size_t M, int C, hipStream_t st);
int bn_bwd_finalize_pair(const BnFinBwd& a, const BnFinBwd& b, hipStream_t st);
int bn_finalize_train(const float* partial, int tiles, int C, double count, const float* gamma, const float* beta,
                      float eps, float momentum, float* rm, float* rv, int64_t* nbt, float* save_mean, float* save_rstd,
                      float* scale, float* shift, hipStream_t st);
int bn_finalize_eval(int C, const float* gamma, const float* beta, float eps, const float* rm, const float* rv,
                     float* scale, float* shift, hipStream_t st);
// relu_bits (optional, with relu): one byte per 16-byte vector of `out`, bit e = (out element e > 0)
// fa / fr (optional): the BatchNorm of y / of the residual branch has no finalized constants yet -- they are derived from the
// integer accumulators in the kernel's prologue and published by its first block (bnacc.h); scale / shift (rscale / rshift)
// are then ignored
int bn_act(int dtype, const void* y, const float* scale, const float* shift, const void* res, const float* rscale,
           const float* rshift, int relu, void* out, size_t M, int C, hipStream_t st, uint8_t* relu_bits = nullptr,
           const BnAccFin* fa = nullptr, const BnAccFin* fr = nullptr);
int bn_bwd_blocks(size_t M, int C);
int bn_bwd_reduce(int dtype, const void* g, const void* y, const float* scale, const float* shift, const float* mean,
                  const float* rstd, int relu_mask, float* partial, size_t M, int C, hipStream_t st);
int block_bwd_reduce(int dtype, const void* dz, const void* z, const void* y2, const void* yd, const float* mean2,
                     const float* rstd2, const float* meand, const float* rstdd, void* do2, float* partial2,
                     float* partiald, size_t M, int C, hipStream_t st, bool premasked = false);
int bn_bwd_finalize(const float* partial, int blocks, int C, double count, float* dgamma, float* dbeta, float* coef,
                    hipStream_t st);
int bn_bwd_apply(int dtype, const void* g, const void* y, const float* scale, const float* shift, const float* mean,
                 const float* rstd, const float* gamma, const float* coef, int relu_mask, void* dy, size_t M, int C,
                 hipStream_t st);
int relu_bwd(int dtype, const void* dy, const void* out, void* dx, size_t n, hipStream_t st);
// pool.hip
int bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, void* ymax,
                        int N, int H, int W, int C, hipStream_t st, const BnAccFin* fa = nullptr);
int maxpool_bn_bwd_apply(int dtype, const void* dout, const uint8_t* idx, const void* y0, const float* scale, const float* shift,
                         const float* mean, const float* rstd, const float* gamma, const float* coef, void* dy, int N, int H,
                         int W, int C, hipStream_t st);
int maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, hipStream_t st);
int avgpool_fwd(int dtype, const void* x, float* feat, int B, int T, int HW, int C, hipStream_t st);
int avgpool_bwd(int dtype, const float* dfeat, void* dx, int B, int T, int HW, int C, hipStream_t st);
// head.hip
int head_uni_dfeat(const float* f, const float* Wp, int ldw, const float* bp, const int64_t* labels, float scale, float* df, int B,
                   int n, int width, hipStream_t st);
int head_concat_fwd(const float* x, const float* y, const float* W, const float* b, float* out, float* x_out, float* y_out,
                    int B, int n, hipStream_t st);
int head_concat_bwd(const float* x, const float* y, const float* W, const float* g_x_out, const float* g_y_out,
                    const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dW, float* db,
                    int B, int n, hipStream_t st);
int head_sum_fwd(const float* x, const float* y, const float* Wx, const float* bx, const float* Wy, const float* by, float* out,
                 float* x_out, float* y_out, int B, int n, hipStream_t st);
int head_sum_bwd(const float* x, const float* y, const float* Wx, const float* Wy, const float* g_x_out, const float* g_y_out,
                 const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dWx, float* dbx,
                 float* dWy, float* dby, int B, int n, hipStream_t st);
int head_gated_fwd(const float* x, const float* y, const float* W1, const float* b1, const float* W2, const float* b2,
                   const float* Wo, const float* bo, float* hx, float* hy, float* out, float* x_out, float* y_out, int B, int n,
                   hipStream_t st);
int head_gated_bwd(const float* x, const float* y, const float* hx, const float* hy, const float* W1, const float* W2,
                   const float* Wo, const float* g_x_out, const float* g_y_out, const float* g_out, int uni_in_dw, float* dx,
                   float* dy, float* dW1, float* db1, float* dW2, float* db2, float* dWo, float* dbo, float* ws, int B, int n,
                   hipStream_t st);
// head_film.hip: FiLM_DGL (B <= 64 per call); hidden = [3][B][512] (h_x, h_f, h_y)
size_t head_film_ws_bytes(int B);
int head_film_fwd(const float* x, const float* y, const float* Wfc, const float* bfc, const float* Wo, const float* bo,
                  float* hidden, float* out, float* x_out, float* y_out, int B, int n, void* ws, size_t ws_bytes, hipStream_t st);
int head_film_bwd(const float* x, const float* y, const float* Wfc, const float* Wo, const float* hidden,
                  const float* g_x_out, const float* g_y_out, const float* g_out, int uni, float* dx, float* dy, float* dWfc,
                  float* dbfc, float* dWo, float* dbo, int B, int n, void* ws, size_t ws_bytes, hipStream_t st);
// input.hip
int logspec_frames(int L, int hop);
int logspec(const float* wave, int B, int L, int n_fft, int hop, int reflect, float* out, hipStream_t st);
int frames_normalize(const unsigned char* in, size_t n_img, int H, int W, const float* mean, const float* std, float* out,
                     hipStream_t st);
int eval_count(const float* out, const float* out_a, const float* out_v, const int64_t* labels, int B, int n, int64_t* num,
               int64_t* acc, int64_t* acc_a, int64_t* acc_v, hipStream_t st);
int softmax_ce_multi(int nsets, const float* const* logits, const int64_t* labels, const float* scales, float* losses,
                     float* const* dlogits, int B, int n, hipStream_t st);
int softmax_ce(const float* logits, const int64_t* labels, float scale, float* loss, float* dlogits, int B, int n,
               hipStream_t st);

int head_concat_xy_fwd(const float* x, const float* y, const float* W, const float* b, float* out, float* x_out, float* y_out, int B,
                       int n, int dxw, int dyw, hipStream_t st);
int head_concat_xy_bwd(const float* x, const float* y, const float* W, const float* g_x_out, const float* g_y_out, const float* g_out,
                       int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dW, float* db, int B, int n, int dxw, int dyw,
                       hipStream_t st);

// ---- Swin visual encoder, non-GEMM operators (swin.hip; the Linears run on conv_fwd / conv_dgrad / conv_wgrad as 1x1)
int swin_patch_gather(int dt, const float* x, void* a, int B, int T, int H, int W, int p, hipStream_t st);
int swin_bias_act(int dt, void* y, const float* bias, void* u, const void* res, size_t M, int ld, int mode, hipStream_t st);
int swin_drop_path(int dt, const void* y, const void* res, const float* scale, void* out, size_t M, int L, int ld, hipStream_t st);
int swin_ln_fwd(int dt, const void* x, const float* gamma, const float* beta, void* y, float* stats, size_t M, int C, int ld,
                hipStream_t st);
size_t swin_partial_bytes(int ld);
int swin_ln_bwd(int dt, const void* dy, const void* x, const float* stats, const float* gamma, const void* add, void* dx,
                float* dgamma_dbeta, float* partial, size_t M, int C, int ld, hipStream_t st, bool colsum = false);
int swin_colsum(int dt, void* g, const void* u, float* db, float* partial, size_t M, int ld, hipStream_t st);
int swin_ln_bwd_rows(int dt, size_t M, int ld);
int swin_colsum_rows(int dt, size_t M, int ld);
int swin_partial_reduce_batched(const void* descs, int nd, int total_blocks, hipStream_t st);
int swin_attn_fwd(int dt, const void* qkv, const float* table, void* out, int n_img, int H, int W, int ws, int shift, int nh, int ld,
                  hipStream_t st);
size_t swin_attn_bwd_ws_bytes(int n_img, int nwin, int ws, int nh);
int swin_attn_bwd(int dt, const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable, float* tpart, int n_img,
                  int H, int W, int ws, int shift, int nh, int ld, hipStream_t st);
// both gradients of a short-K Linear from one pass over dy (linear_bwd.hip)
bool linear_bwd_ok(int dtype, size_t M, int K, int N);
size_t linear_bwd_ws_bytes(size_t M, int K, int N);
int linear_bwd(const void* dy, const void* x, const void* wT, void* dx, float* dw, float* db, void* ws, size_t ws_bytes, size_t M, int K,
               int Kreal, int N, hipStream_t st);
// window 7, bf16: swin_attn7.hip (swin_attn_fwd / _bwd dispatch to it)
bool swin_attn7_ok(int dt, int H, int W, int ws, int shift, int nh, int ld, int n_img);
int swin_attn7_fwd(const void* qkv, const float* table, void* out, int n_img, int H, int W, int shift, int nh, int ld, hipStream_t st);
size_t swin_attn7_bwd_ws_bytes(int n_img, int H, int W, int nh);
int swin_attn7_bwd(const void* qkv, const float* table, const void* dout, void* dqkv, float* tpart, int* nparts, int n_img, int H, int W,
                   int shift, int nh, int ld, hipStream_t st);
int swin_merge(int dt, const void* src, void* dst, int N, int H, int W, int C, int ldx, int scatter, hipStream_t st);
int swin_token_mean(int dt, const void* x, float* y, int N, int L, int C, int ld, hipStream_t st);
int swin_token_mean_bwd(int dt, const float* dy, void* dx, int N, int L, int C, int ld, hipStream_t st);
int swin_pack_matrix(int dt, const float* src, void* dst, void* dstT, int n, int k, int nseg, int nseg_pad, int kseg, int kseg_pad,
                     hipStream_t st);
int swin_pack_batched(const void* descs, int nd, int total_blocks, int dir, hipStream_t st);
int swin_unpack_matrix(const float* src, float* dst, int n, int k, int nseg, int nseg_pad, int kseg, int kseg_pad, hipStream_t st);

}  // namespace gdl

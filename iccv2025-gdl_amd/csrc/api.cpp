// api.cpp -- the C ABI of libgdl_hip.so (see include/gdl_hip.h) over the internal launchers.
#include "ops.h"

using namespace gdl;

#ifdef GDL_TIMING
namespace gdl { extern unsigned long long* g_timing_buf; }
#endif
extern "C" {
// tuning aid (only active in a -DGDL_TIMING build): per-block s_memtime stamps go to buf[block][8]
GDL_API int gdl_debug_timing_buffer(void* buf) {
#ifdef GDL_TIMING
    gdl::g_timing_buf = (unsigned long long*)buf;
    return GDL_OK;
#else
    (void)buf;
    return GDL_ERR_ARG;
#endif
}


const char* gdl_last_error(void) { return last_error(); }
int gdl_version(void) { return 100; }

int gdl_device_info(int* cu_count, char* name, int name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return check_hip(e, "hipGetDevice");
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return check_hip(e, "hipGetDeviceProperties");
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return GDL_OK;
}

static bool dt_ok(int dtype) { return dtype == GDL_F32 || dtype == GDL_BF16; }

int gdl_conv_bn_tiles(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    return conv_tiles_m(dtype, N, H, W, C, K, R, S, stride, pad);
}

size_t gdl_conv_table_bytes(int mode, int N, int H, int W, int R, int S, int stride, int pad) {
    return gather_table_bytes(mode, N, H, W, R, S, stride, pad);
}
int gdl_conv_build_table(int mode, int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                         void* table, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && table && (mode == GATHER_FWD || mode == GATHER_DGRAD), "conv_build_table: bad arguments");
    return build_gather_table(mode, dtype, N, H, W, C, K, R, S, stride, pad, (GatherEntry*)table, (hipStream_t)stream);
}
int gdl_conv_fwd(int dtype, const void* x, const void* w_krsc, void* y, float* bn_partial, const void* table, int N, int H,
                 int W, int C, int K, int R, int S, int stride, int pad, void* stream) {
    GDL_REQUIRE(x && w_krsc && y, "conv_fwd: null pointer");
    return conv_fwd(dtype, x, w_krsc, y, bn_partial, table, N, H, W, C, K, R, S, stride, pad, (hipStream_t)stream);
}
int gdl_bn_act_bits(int dtype, const void* y, const float* scale, const float* shift, const void* res, const float* res_scale,
                    const float* res_shift, void* out, uint8_t* relu_bits, size_t M, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && y && scale && shift && out && relu_bits, "bn_act_bits: bad arguments");
    return bn_act(dtype, y, scale, shift, res, res_scale, res_shift, 1, out, M, C, (hipStream_t)stream, relu_bits);
}
int gdl_conv_dgrad_relu(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, const uint8_t* relu_bits,
                        const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* stream) {
    GDL_REQUIRE(dy && w_crsk && dx && relu_bits, "conv_dgrad_relu: null pointer");
    return conv_dgrad(dtype, dy, w_crsk, dx, addend, table, N, H, W, C, K, R, S, stride, pad, (hipStream_t)stream, relu_bits);
}
int gdl_conv_fwd_bias(int dtype, const void* x, const void* w_krsc, void* y, const float* bias, const void* addend, void* gelu_out,
                      const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* stream) {
    GDL_REQUIRE(x && w_krsc && y, "conv_fwd_bias: null pointer");
    return conv_fwd_bias(dtype, x, w_krsc, y, bias, addend, gelu_out, table, N, H, W, C, K, R, S, stride, pad, (hipStream_t)stream);
}
int gdl_conv_dgrad_ds(int dtype, const void* dy, const void* w_crsk, const void* dy_ds, const void* w_ds_ck, void* dx,
                      const uint8_t* relu_bits, const void* table, int N, int H, int W, int C, int K, void* stream) {
    GDL_REQUIRE(dy && w_crsk && dy_ds && w_ds_ck && dx, "conv_dgrad_ds: null pointer");
    return conv_dgrad_ds(dtype, dy, w_crsk, dy_ds, w_ds_ck, dx, table, N, H, W, C, K, (hipStream_t)stream, relu_bits);
}
int gdl_conv_dgrad_bn_tiles(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    return conv_dgrad_tiles_m(dtype, N, H, W, C, K, R, S, stride, pad);
}
int gdl_conv_dgrad_bn(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, const uint8_t* relu_bits,
                      const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, const void* y,
                      const float* mean, const float* rstd, float* partial, const void* y2, const float* mean2,
                      const float* rstd2, float* partial2, void* stream) {
    GDL_REQUIRE(dy && w_crsk && dx && y && mean && rstd && partial, "conv_dgrad_bn: null pointer");
    const BwdStats bw{y, mean, rstd, partial, y2, mean2, rstd2, partial2};
    return conv_dgrad(dtype, dy, w_crsk, dx, addend, table, N, H, W, C, K, R, S, stride, pad, (hipStream_t)stream, relu_bits, &bw);
}
size_t gdl_conv_split_workspace_bytes(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dgrad) {
    return dt_ok(dtype) ? conv_split_ws_bytes(dtype, N, H, W, C, K, R, S, stride, pad, dgrad) : 0;
}
int gdl_conv_fwd_split(int dtype, const void* x, const void* w_krsc, void* y, float* bn_partial, const void* table, int N, int H,
                       int W, int C, int K, int R, int S, int stride, int pad, void* split_ws, size_t split_ws_bytes, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && x && w_krsc && y && table, "conv_fwd_split: null pointer");
    const SplitWs sk{split_ws, split_ws_bytes};
    return conv_fwd(dtype, x, w_krsc, y, bn_partial, table, N, H, W, C, K, R, S, stride, pad, (hipStream_t)stream, nullptr, &sk);
}
int gdl_conv_dgrad_bn_split(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, const uint8_t* relu_bits,
                            const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, const void* y,
                            const float* mean, const float* rstd, float* partial, const void* y2, const float* mean2,
                            const float* rstd2, float* partial2, void* split_ws, size_t split_ws_bytes, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dy && w_crsk && dx && table, "conv_dgrad_bn_split: null pointer");
    const BwdStats bw{y, mean, rstd, partial, y2, mean2, rstd2, partial2};
    const SplitWs sk{split_ws, split_ws_bytes};
    return conv_dgrad(dtype, dy, w_crsk, dx, addend, table, N, H, W, C, K, R, S, stride, pad, (hipStream_t)stream, relu_bits,
                      y ? &bw : nullptr, &sk);
}
int gdl_conv_dgrad_gelu(int dtype, const void* dy, const void* w_crsk, void* dx, const void* u, void* acc, double scale,
                        const void* table, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dy && w_crsk && dx && u && acc && table && scale > 0.0, "conv_dgrad_gelu: bad arguments");
    const BnAcc a{(long long*)acc, scale, 0.0};
    return conv_dgrad_gelu(dtype, dy, w_crsk, dx, u, &a, table, N, H, W, C, K, R, S, stride, pad, (hipStream_t)stream);
}
int gdl_acc_to_float(const void* acc, int n, double inv_scale, float* out, void* stream) {
    GDL_REQUIRE(acc && out && n > 0, "acc_to_float: bad arguments");
    return acc_to_float((const long long*)acc, n, inv_scale, out, (hipStream_t)stream);
}
int gdl_conv_dgrad(int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, const void* table, int N,
                   int H, int W, int C, int K, int R, int S, int stride, int pad, void* stream) {
    GDL_REQUIRE(dy && w_crsk && dx, "conv_dgrad: null pointer");
    return conv_dgrad(dtype, dy, w_crsk, dx, addend, table, N, H, W, C, K, R, S, stride, pad, (hipStream_t)stream);
}
size_t gdl_conv_wgrad_workspace_bytes(int dtype, int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
    (void)dtype;
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
    return conv_wgrad_ws_bytes(N * P * Q, C, K, R * S);
}
int gdl_conv_wgrad(int dtype, const void* dy, const void* x, float* dw, const void* table, int N, int H, int W, int C,
                   int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    GDL_REQUIRE(dy && x && dw, "conv_wgrad: null pointer");
    return conv_wgrad(dtype, dy, x, dw, table, N, H, W, C, K, R, S, stride, pad, C, ws, ws_bytes, (hipStream_t)stream);
}
int gdl_pack_weight(int dtype, const float* w, void* w_krsc, void* w_crsk, int K, int C, int R, int S, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && w, "pack_weight: bad arguments");
    return pack_weight(dtype, w, w_krsc, w_crsk, K, C, R, S, (hipStream_t)stream);
}
// direct stem
size_t gdl_stem_pad_bytes(int dtype, int n_img, int H, int W) { return dt_ok(dtype) ? stem_pad_bytes(dtype, n_img, H, W) : 0; }
size_t gdl_stem_weight_bytes(int dtype) {
    return dt_ok(dtype) ? (size_t)64 * stem_taps(dtype) * stem_ic(dtype) * (dtype == GDL_BF16 ? 2 : 4) : 0;
}
size_t gdl_stem_table_bytes(int n_img, int H, int W) {
    return (size_t)n_img * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * sizeof(GatherEntry);
}
int gdl_stem_pad(int dtype, const float* x, void* xp, int B, int Cin, int T, int H, int W, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && x && xp, "stem_pad: bad arguments");
    return stem_pad(dtype, x, xp, B, Cin, T, H, W, (hipStream_t)stream);
}
int gdl_pack_stem_rows(int dtype, const float* w, void* wp, int Cin, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && w && wp && Cin >= 1 && Cin <= 4, "pack_stem_rows: bad arguments");
    return pack_stem_rows(dtype, w, wp, Cin, (hipStream_t)stream);
}
int gdl_stem_build_table(int dtype, int n_img, int H, int W, void* table, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && table, "stem_build_table: bad arguments");
    return build_stem_table(dtype, n_img, H, W, stem_taps(dtype), (GatherEntry*)table, (hipStream_t)stream);
}
int gdl_stem_conv_bn_tiles(int dtype, int n_img, int H, int W) { return conv_stem_tiles_m(dtype, n_img, H, W); }
int gdl_stem_conv_fwd(int dtype, const void* xp, const void* wp, void* y, float* bn_partial, const void* table, int n_img,
                      int H, int W, int Cin, void* stream) {
    GDL_REQUIRE(dt_ok(dtype), "stem_conv_fwd: bad dtype");
    return conv_stem_fwd(dtype, xp, wp, y, bn_partial, table, n_img, H, W, Cin, (hipStream_t)stream);
}
size_t gdl_stem_conv_wgrad_workspace_bytes(int n_img, int H, int W) { return conv_stem_wgrad_ws_bytes(n_img, H, W); }
int gdl_stem_bwd_fused_ok(int dtype, int W) { return stem_bwd_fused_ok(dtype, W) ? 1 : 0; }
int gdl_stem_bwd_fused(int dtype, const void* dout, const uint8_t* idx, const void* y, const float* scale, const float* shift,
                       const float* save_mean, const float* save_rstd, const float* gamma, const float* coef, const void* xp,
                       float* dw, int n_img, int H, int W, int Cin, void* ws, size_t ws_bytes, void* stream) {
    GDL_REQUIRE(stem_bwd_fused_ok(dtype, W), "stem_bwd_fused: bf16 with output rows of at least 64 pixels only (ask gdl_stem_bwd_fused_ok)");
    return stem_bwd_fused(dout, idx, y, scale, shift, save_mean, save_rstd, gamma, coef, xp, dw, n_img, H, W, Cin, ws, ws_bytes,
                          (hipStream_t)stream);
}
int gdl_stem_conv_wgrad(int dtype, const void* dy, const void* xp, float* dw, const void* table, int n_img, int H, int W,
                        int Cin, void* ws, size_t ws_bytes, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && Cin >= 1 && Cin <= 4, "stem_conv_wgrad: bad arguments");
    return conv_stem_wgrad(dtype, dy, xp, dw, table, n_img, H, W, Cin, ws, ws_bytes, (hipStream_t)stream);
}
int gdl_nhwc_to_nchw_f32(int dtype, const void* x, float* y, int N, int H, int W, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && x && y, "nhwc_to_nchw: bad arguments");
    return nhwc_to_nchw_f32(dtype, x, y, N, H, W, C, (hipStream_t)stream);
}
int gdl_nchw_f32_to_nhwc(int dtype, const float* x, void* y, int N, int H, int W, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && x && y, "nchw_to_nhwc: bad arguments");
    return nchw_f32_to_nhwc(dtype, x, y, N, H, W, C, (hipStream_t)stream);
}

int gdl_bn_stats_tiles(int M) { return bn_stats_tiles(M); }
int gdl_bn_stats(int dtype, const void* y, float* partial, int M, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && y && partial, "bn_stats: bad arguments");
    return bn_stats(dtype, y, partial, M, C, (hipStream_t)stream);
}
int gdl_bn_finalize_train(const float* partial, int tiles, int C, double count, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, int64_t* nbt,
                          float* save_mean, float* save_rstd, float* scale, float* shift, void* stream) {
    GDL_REQUIRE(partial && gamma && beta && save_mean && save_rstd && scale && shift, "bn_finalize_train: null pointer");
    return bn_finalize_train(partial, tiles, C, count, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                             save_mean, save_rstd, scale, shift, (hipStream_t)stream);
}
int gdl_bn_finalize_eval(int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                         const float* running_var, float* scale, float* shift, void* stream) {
    GDL_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, "bn_finalize_eval: null pointer");
    return bn_finalize_eval(C, gamma, beta, eps, running_mean, running_var, scale, shift, (hipStream_t)stream);
}
int gdl_bn_act(int dtype, const void* y, const float* scale, const float* shift, const void* res, const float* res_scale,
               const float* res_shift, int relu, void* out, size_t M, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && y && scale && shift && out, "bn_act: bad arguments");
    return bn_act(dtype, y, scale, shift, res, res_scale, res_shift, relu, out, M, C, (hipStream_t)stream);
}
int gdl_bn_bwd_blocks(size_t M, int C) { return bn_bwd_blocks(M, C); }
int gdl_bn_bwd_reduce(int dtype, const void* g, const void* y, const float* scale, const float* shift,
                      const float* save_mean, const float* save_rstd, int relu_mask, float* partial, size_t M, int C,
                      void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && g && y && save_mean && save_rstd && partial, "bn_bwd_reduce: bad arguments");
    return bn_bwd_reduce(dtype, g, y, scale, shift, save_mean, save_rstd, relu_mask, partial, M, C, (hipStream_t)stream);
}
int gdl_bn_bwd_finalize(const float* partial, int blocks, int C, double count, float* dgamma, float* dbeta, float* coef,
                        void* stream) {
    GDL_REQUIRE(partial && dgamma && dbeta && coef, "bn_bwd_finalize: null pointer");
    return bn_bwd_finalize(partial, blocks, C, count, dgamma, dbeta, coef, (hipStream_t)stream);
}
int gdl_bn_bwd_apply(int dtype, const void* g, const void* y, const float* scale, const float* shift,
                     const float* save_mean, const float* save_rstd, const float* gamma, const float* coef, int relu_mask,
                     void* dy, size_t M, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && g && y && save_mean && save_rstd && gamma && coef && dy, "bn_bwd_apply: bad arguments");
    return bn_bwd_apply(dtype, g, y, scale, shift, save_mean, save_rstd, gamma, coef, relu_mask, dy, M, C,
                        (hipStream_t)stream);
}
int gdl_relu_bwd(int dtype, const void* dy, const void* out, void* dx, size_t n, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dy && out && dx, "relu_bwd: bad arguments");
    return relu_bwd(dtype, dy, out, dx, n, (hipStream_t)stream);
}

int gdl_bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, uint8_t* idx,
                            void* ymax, int N, int H, int W, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && y && scale && shift && out && idx, "bn_relu_maxpool_fwd: bad arguments");
    return bn_relu_maxpool_fwd(dtype, y, scale, shift, out, idx, ymax, N, H, W, C, (hipStream_t)stream);
}
int gdl_maxpool_bn_bwd_apply(int dtype, const void* dout, const uint8_t* idx, const void* y, const float* scale,
                             const float* shift, const float* save_mean, const float* save_rstd, const float* gamma,
                             const float* coef, void* dy, int N, int H, int W, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dout && idx && y && scale && shift && save_mean && save_rstd && gamma && coef && dy,
                "maxpool_bn_bwd_apply: bad arguments");
    return maxpool_bn_bwd_apply(dtype, dout, idx, y, scale, shift, save_mean, save_rstd, gamma, coef, dy, N, H, W, C,
                                (hipStream_t)stream);
}
int gdl_maxpool_bwd(int dtype, const void* dout, const uint8_t* idx, void* dx, int N, int H, int W, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dout && idx && dx, "maxpool_bwd: bad arguments");
    return maxpool_bwd(dtype, dout, idx, dx, N, H, W, C, (hipStream_t)stream);
}
int gdl_avgpool_fwd(int dtype, const void* x, float* feat, int B, int T, int HW, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && x && feat, "avgpool_fwd: bad arguments");
    return avgpool_fwd(dtype, x, feat, B, T, HW, C, (hipStream_t)stream);
}
int gdl_avgpool_bwd(int dtype, const float* dfeat, void* dx, int B, int T, int HW, int C, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dfeat && dx, "avgpool_bwd: bad arguments");
    return avgpool_bwd(dtype, dfeat, dx, B, T, HW, C, (hipStream_t)stream);
}

int gdl_head_uni_dfeat(const float* f, const float* Wp, int ldw, const float* bp, const int64_t* labels, float scale, float* df,
                       int B, int n_classes, void* stream) {
    GDL_REQUIRE(f && Wp && bp && labels && df && B > 0 && n_classes > 0 && ldw >= 512, "head_uni_dfeat: bad arguments");
    return head_uni_dfeat(f, Wp, ldw, bp, labels, scale, df, B, n_classes, 512, (hipStream_t)stream);
}
int gdl_head_uni_dfeat_w(const float* f, const float* Wp, int ldw, const float* bp, const int64_t* labels, float scale, float* df,
                         int B, int n_classes, int width, void* stream) {
    GDL_REQUIRE(f && Wp && bp && labels && df && B > 0 && n_classes > 0 && ldw >= width, "head_uni_dfeat_w: bad arguments");
    return head_uni_dfeat(f, Wp, ldw, bp, labels, scale, df, B, n_classes, width, (hipStream_t)stream);
}
int gdl_head_concat_fwd(const float* x, const float* y, const float* W, const float* b, float* out, float* x_out,
                        float* y_out, int B, int n_classes, void* stream) {
    GDL_REQUIRE(x && y && W && b && out && B > 0 && n_classes > 0, "head_concat_fwd: bad arguments");
    return head_concat_fwd(x, y, W, b, out, x_out, y_out, B, n_classes, (hipStream_t)stream);
}
int gdl_head_concat_bwd(const float* x, const float* y, const float* W, const float* g_x_out, const float* g_y_out,
                        const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dW, float* db,
                        int B, int n_classes, void* stream) {
    GDL_REQUIRE(x && y && W && B > 0 && n_classes > 0, "head_concat_bwd: bad arguments");
    GDL_REQUIRE((dx != nullptr) == (dy != nullptr) && (dW != nullptr) == (db != nullptr),
                "head_concat_bwd: dx/dy and dW/db come in pairs");
    return head_concat_bwd(x, y, W, g_x_out, g_y_out, g_out, out_reaches_xy, uni_in_dw, dx, dy, dW, db, B, n_classes,
                           (hipStream_t)stream);
}
int gdl_head_sum_fwd(const float* x, const float* y, const float* Wx, const float* bx, const float* Wy, const float* by,
                     float* out, float* x_out, float* y_out, int B, int n_classes, void* stream) {
    GDL_REQUIRE(x && y && Wx && bx && Wy && by && out && B > 0 && n_classes > 0, "head_sum_fwd: bad arguments");
    return head_sum_fwd(x, y, Wx, bx, Wy, by, out, x_out, y_out, B, n_classes, (hipStream_t)stream);
}
int gdl_head_sum_bwd(const float* x, const float* y, const float* Wx, const float* Wy, const float* g_x_out,
                     const float* g_y_out, const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx, float* dy,
                     float* dWx, float* dbx, float* dWy, float* dby, int B, int n_classes, void* stream) {
    GDL_REQUIRE(x && y && Wx && Wy && B > 0 && n_classes > 0, "head_sum_bwd: bad arguments");
    GDL_REQUIRE((dx != nullptr) == (dy != nullptr), "head_sum_bwd: dx/dy come in pairs");
    GDL_REQUIRE((dWx != nullptr) == (dbx != nullptr) && (dWy != nullptr) == (dby != nullptr) && (dWx != nullptr) == (dWy != nullptr),
                "head_sum_bwd: dWx/dbx/dWy/dby come together");
    return head_sum_bwd(x, y, Wx, Wy, g_x_out, g_y_out, g_out, out_reaches_xy, uni_in_dw, dx, dy, dWx, dbx, dWy, dby, B,
                        n_classes, (hipStream_t)stream);
}
int gdl_head_gated_fwd(const float* x, const float* y, const float* W1, const float* b1, const float* W2, const float* b2,
                       const float* Wo, const float* bo, float* hx, float* hy, float* out, float* x_out, float* y_out, int B,
                       int n_classes, void* stream) {
    GDL_REQUIRE(x && y && W1 && b1 && W2 && b2 && Wo && bo && hx && hy && out && B > 0 && n_classes > 0,
                "head_gated_fwd: bad arguments");
    return head_gated_fwd(x, y, W1, b1, W2, b2, Wo, bo, hx, hy, out, x_out, y_out, B, n_classes, (hipStream_t)stream);
}
int gdl_head_gated_bwd(const float* x, const float* y, const float* hx, const float* hy, const float* W1, const float* W2,
                       const float* Wo, const float* g_x_out, const float* g_y_out, const float* g_out, int uni_in_dw,
                       float* dx, float* dy, float* dW1, float* db1, float* dW2, float* db2, float* dWo, float* dbo, float* ws,
                       int B, int n_classes, void* stream) {
    GDL_REQUIRE(x && y && hx && hy && W1 && W2 && Wo && ws && B > 0 && n_classes > 0, "head_gated_bwd: bad arguments");
    GDL_REQUIRE((dx != nullptr) == (dy != nullptr) && (dWo != nullptr) == (dbo != nullptr), "head_gated_bwd: dx/dy, dWo/dbo pairs");
    GDL_REQUIRE((dW1 != nullptr) == (db1 != nullptr) && (dW1 != nullptr) == (dW2 != nullptr) && (dW1 != nullptr) == (db2 != nullptr),
                "head_gated_bwd: dW1/db1/dW2/db2 come together");
    return head_gated_bwd(x, y, hx, hy, W1, W2, Wo, g_x_out, g_y_out, g_out, uni_in_dw, dx, dy, dW1, db1, dW2, db2, dWo, dbo, ws, B,
                          n_classes, (hipStream_t)stream);
}
size_t gdl_head_film_workspace_bytes(int B) { return head_film_ws_bytes(B); }
int gdl_head_film_fwd(const float* x, const float* y, const float* Wfc, const float* bfc, const float* Wo, const float* bo,
                      float* hidden, float* out, float* x_out, float* y_out, int B, int n_classes, void* ws, size_t ws_bytes,
                      void* stream) {
    GDL_REQUIRE(x && y && Wfc && bfc && Wo && bo && hidden && out && n_classes > 0, "head_film_fwd: bad arguments");
    return head_film_fwd(x, y, Wfc, bfc, Wo, bo, hidden, out, x_out, y_out, B, n_classes, ws, ws_bytes, (hipStream_t)stream);
}
int gdl_head_film_bwd(const float* x, const float* y, const float* Wfc, const float* Wo, const float* hidden,
                      const float* g_x_out, const float* g_y_out, const float* g_out, int uni_in_dw, float* dx, float* dy,
                      float* dWfc, float* dbfc, float* dWo, float* dbo, int B, int n_classes, void* ws, size_t ws_bytes,
                      void* stream) {
    GDL_REQUIRE(x && y && Wfc && Wo && hidden && n_classes > 0, "head_film_bwd: bad arguments");
    GDL_REQUIRE((dx != nullptr) == (dy != nullptr) && (dWfc != nullptr) == (dbfc != nullptr) && (dWo != nullptr) == (dbo != nullptr),
                "head_film_bwd: dx/dy, dWfc/dbfc, dWo/dbo come in pairs");
    return head_film_bwd(x, y, Wfc, Wo, hidden, g_x_out, g_y_out, g_out, uni_in_dw, dx, dy, dWfc, dbfc, dWo, dbo, B, n_classes,
                         ws, ws_bytes, (hipStream_t)stream);
}
int gdl_eval_count(const float* out, const float* out_a, const float* out_v, const int64_t* labels, int B, int n_classes,
                   int64_t* num, int64_t* acc, int64_t* acc_a, int64_t* acc_v, void* stream) {
    GDL_REQUIRE(out && labels && num && acc && B > 0 && n_classes > 0, "eval_count: bad arguments");
    GDL_REQUIRE((out_a != nullptr) == (acc_a != nullptr) && (out_v != nullptr) == (acc_v != nullptr),
                "eval_count: a unimodal logit set needs its counter array and vice versa");
    return eval_count(out, out_a, out_v, labels, B, n_classes, num, acc, acc_a, acc_v, (hipStream_t)stream);
}
int gdl_logspec_frames(int n_samples, int hop) {
    if (n_samples <= 0 || hop <= 0) return 0;
    return logspec_frames(n_samples, hop);
}
int gdl_logspec(const float* wave, int B, int n_samples, int n_fft, int hop, int pad_mode, float* out, void* stream) {
    GDL_REQUIRE(wave && out && B > 0 && n_samples > 0 && hop > 0, "logspec: bad arguments");
    GDL_REQUIRE(n_fft >= 16 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0, "logspec: n_fft must be a power of two in [16, 2048]");
    GDL_REQUIRE(pad_mode == GDL_PAD_CONSTANT || pad_mode == GDL_PAD_REFLECT, "logspec: pad_mode must be GDL_PAD_CONSTANT or GDL_PAD_REFLECT");
    GDL_REQUIRE(pad_mode == GDL_PAD_CONSTANT || n_samples > n_fft / 2, "logspec: reflect padding needs more than n_fft/2 samples");
    return logspec(wave, B, n_samples, n_fft, hop, pad_mode == GDL_PAD_REFLECT, out, (hipStream_t)stream);
}
int gdl_frames_normalize(const uint8_t* frames, int64_t n_img, int H, int W, const float* mean, const float* std, float* out,
                         void* stream) {
    GDL_REQUIRE(frames && out && mean && std && n_img > 0 && H > 0 && W > 0, "frames_normalize: bad arguments");
    GDL_REQUIRE(std[0] != 0.f && std[1] != 0.f && std[2] != 0.f, "frames_normalize: zero std");
    return frames_normalize(frames, (size_t)n_img, H, W, mean, std, out, (hipStream_t)stream);
}
int gdl_softmax_ce3(const float* logits0, const float* logits1, const float* logits2, const int64_t* labels, float scale0,
                    float scale1, float scale2, float* losses, float* dlogits0, float* dlogits1, float* dlogits2, int B,
                    int n_classes, void* stream) {
    GDL_REQUIRE(logits0 && logits1 && logits2 && labels && losses && B > 0 && n_classes > 0, "softmax_ce3: bad arguments");
    const float* lg[3] = {logits0, logits1, logits2};
    float* dl[3] = {dlogits0, dlogits1, dlogits2};
    const float sc[3] = {scale0, scale1, scale2};
    return softmax_ce_multi(3, lg, labels, sc, losses, dl, B, n_classes, (hipStream_t)stream);
}
int gdl_softmax_ce(const float* logits, const int64_t* labels, float scale, float* loss, float* dlogits, int B,
                   int n_classes, void* stream) {
    GDL_REQUIRE(logits && labels && loss && B > 0 && n_classes > 0, "softmax_ce: bad arguments");
    return softmax_ce(logits, labels, scale, loss, dlogits, B, n_classes, (hipStream_t)stream);
}

int gdl_head_concat_xy_fwd(const float* x, const float* y, const float* W, const float* b, float* out, float* x_out, float* y_out,
                           int B, int n_classes, int x_dim, int y_dim, void* stream) {
    GDL_REQUIRE(x && y && W && b && out && B > 0 && n_classes > 0 && x_dim > 0 && y_dim > 0, "head_concat_xy_fwd: bad arguments");
    return head_concat_xy_fwd(x, y, W, b, out, x_out, y_out, B, n_classes, x_dim, y_dim, (hipStream_t)stream);
}
int gdl_head_concat_xy_bwd(const float* x, const float* y, const float* W, const float* g_x_out, const float* g_y_out,
                           const float* g_out, int out_reaches_xy, int uni_in_dw, float* dx, float* dy, float* dW, float* db, int B,
                           int n_classes, int x_dim, int y_dim, void* stream) {
    GDL_REQUIRE(x && y && W && B > 0 && n_classes > 0 && x_dim > 0 && y_dim > 0, "head_concat_xy_bwd: bad arguments");
    GDL_REQUIRE((dx != nullptr) == (dy != nullptr) && (dW != nullptr) == (db != nullptr), "head_concat_xy_bwd: dx/dy and dW/db come in pairs");
    return head_concat_xy_bwd(x, y, W, g_x_out, g_y_out, g_out, out_reaches_xy, uni_in_dw, dx, dy, dW, db, B, n_classes, x_dim, y_dim,
                              (hipStream_t)stream);
}
// ---- Swin visual encoder: non-GEMM operators; the Linears are gdl_conv_fwd / _dgrad / _wgrad (1x1)
int gdl_swin_patch_gather(int dtype, const float* x, void* a, int B, int T, int H, int W, int patch, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && x && a, "swin_patch_gather: bad arguments");
    return swin_patch_gather(dtype, x, a, B, T, H, W, patch, (hipStream_t)stream);
}
int gdl_swin_bias_act(int dtype, void* y, const float* bias, void* u, const void* res, size_t M, int ld, int mode, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && y && bias, "swin_bias_act: bad arguments");
    return swin_bias_act(dtype, y, bias, u, res, M, ld, mode, (hipStream_t)stream);
}
int gdl_linear_bwd_ok(int dtype, size_t M, int K, int N) { return linear_bwd_ok(dtype, M, K, N) ? 1 : 0; }
size_t gdl_linear_bwd_workspace_bytes(size_t M, int K, int N) { return linear_bwd_ok(GDL_BF16, M, K, N) ? linear_bwd_ws_bytes(M, K, N) : 0; }
int gdl_linear_bwd(int dtype, const void* dy, const void* x, const void* wT, void* dx, float* dw, float* db, void* ws, size_t ws_bytes,
                   size_t M, int K, int Kreal, int N, void* stream) {
    GDL_REQUIRE(dtype == GDL_BF16, "linear_bwd: bf16 only");
    return linear_bwd(dy, x, wT, dx, dw, db, ws, ws_bytes, M, K, Kreal, N, (hipStream_t)stream);
}
int gdl_swin_drop_path(int dtype, const void* y, const void* res, const float* scale, void* out, size_t M, int L, int ld, void* stream) {
    GDL_REQUIRE(dt_ok(dtype), "swin_drop_path: bad dtype");
    return swin_drop_path(dtype, y, res, scale, out, M, L, ld, (hipStream_t)stream);
}
int gdl_swin_ln_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats, size_t M, int C, int ld,
                    void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && x && gamma && beta && y && stats, "swin_ln_fwd: bad arguments");
    return swin_ln_fwd(dtype, x, gamma, beta, y, stats, M, C, ld, (hipStream_t)stream);
}
size_t gdl_swin_partial_bytes(int ld) { return swin_partial_bytes(ld); }
int gdl_swin_ln_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const void* add, void* dx,
                    float* dgamma_dbeta, void* partial, size_t M, int C, int ld, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dy && x && stats && gamma && dx, "swin_ln_bwd: bad arguments");
    return swin_ln_bwd(dtype, dy, x, stats, gamma, add, dx, dgamma_dbeta, (float*)partial, M, C, ld, (hipStream_t)stream);
}
int gdl_swin_ln_bwd_colsum(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, const void* add,
                           void* dx, float* dgamma_dbeta_colsum, void* partial, size_t M, int C, int ld, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dy && x && stats && gamma && dx, "swin_ln_bwd_colsum: bad arguments");
    return swin_ln_bwd(dtype, dy, x, stats, gamma, add, dx, dgamma_dbeta_colsum, (float*)partial, M, C, ld, (hipStream_t)stream, true);
}
int gdl_swin_colsum(int dtype, void* g, const void* u, float* db, void* partial, size_t M, int ld, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && g, "swin_colsum: bad arguments");
    return swin_colsum(dtype, g, u, db, (float*)partial, M, ld, (hipStream_t)stream);
}
int gdl_swin_attn_fwd(int dtype, const void* qkv, const float* table, void* out, int n_img, int H, int W, int window, int shift,
                      int heads, int ld, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && qkv && table && out, "swin_attn_fwd: bad arguments");
    return swin_attn_fwd(dtype, qkv, table, out, n_img, H, W, window, shift, heads, ld, (hipStream_t)stream);
}
size_t gdl_swin_attn_bwd_workspace_bytes(int n_img, int H, int W, int window, int heads) {
    if (window <= 0) return 0;
    const size_t general = swin_attn_bwd_ws_bytes(n_img, (H / window) * (W / window), window, heads);
    const size_t seven = window == 7 ? swin_attn7_bwd_ws_bytes(n_img, H, W, heads) : 0;  // either kernel may serve the call
    return general > seven ? general : seven;
}
int gdl_swin_attn_bwd(int dtype, const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable, void* ws, int n_img,
                      int H, int W, int window, int shift, int heads, int ld, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && qkv && table && dout && dqkv, "swin_attn_bwd: bad arguments");
    return swin_attn_bwd(dtype, qkv, table, dout, dqkv, dtable, (float*)ws, n_img, H, W, window, shift, heads, ld, (hipStream_t)stream);
}
int gdl_swin_merge(int dtype, const void* src, void* dst, int N, int H, int W, int C, int ldx, int scatter, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && src && dst, "swin_merge: bad arguments");
    return swin_merge(dtype, src, dst, N, H, W, C, ldx, scatter, (hipStream_t)stream);
}
int gdl_swin_token_mean(int dtype, const void* x, float* y, int N, int L, int C, int ld, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && x && y, "swin_token_mean: bad arguments");
    return swin_token_mean(dtype, x, y, N, L, C, ld, (hipStream_t)stream);
}
int gdl_swin_token_mean_bwd(int dtype, const float* dy, void* dx, int N, int L, int C, int ld, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && dy && dx, "swin_token_mean_bwd: bad arguments");
    return swin_token_mean_bwd(dtype, dy, dx, N, L, C, ld, (hipStream_t)stream);
}
int gdl_swin_pack_matrix(int dtype, const float* src, void* dst, void* dstT, int n, int k, int nseg, int nseg_pad, int kseg,
                         int kseg_pad, void* stream) {
    GDL_REQUIRE(dt_ok(dtype) && src && dst, "swin_pack_matrix: bad arguments");
    return swin_pack_matrix(dtype, src, dst, dstT, n, k, nseg, nseg_pad, kseg, kseg_pad, (hipStream_t)stream);
}
int gdl_swin_ln_bwd_rows(int dtype, size_t M, int ld) {
    if (!dt_ok(dtype) || ld <= 0 || ld % 64 != 0 || M == 0) return 0;  // (a count, not a status: 0 = bad arguments)
    return swin_ln_bwd_rows(dtype, M, ld);
}
int gdl_swin_colsum_rows(int dtype, size_t M, int ld) {
    if (!dt_ok(dtype) || ld <= 0 || ld % 64 != 0 || M == 0) return 0;
    return swin_colsum_rows(dtype, M, ld);
}
int gdl_swin_partial_reduce_batched(const void* descs, int n_desc, int total_blocks, void* stream) {
    return swin_partial_reduce_batched(descs, n_desc, total_blocks, (hipStream_t)stream);
}
int gdl_swin_pack_batched(const void* descs, int n_desc, int total_blocks, int dir, void* stream) {
    return swin_pack_batched(descs, n_desc, total_blocks, dir, (hipStream_t)stream);
}
int gdl_swin_unpack_matrix(const float* src, float* dst, int n, int k, int nseg, int nseg_pad, int kseg, int kseg_pad, void* stream) {
    GDL_REQUIRE(src && dst, "swin_unpack_matrix: null pointer");
    return swin_unpack_matrix(src, dst, n, k, nseg, nseg_pad, kseg, kseg_pad, (hipStream_t)stream);
}

}  // extern "C"

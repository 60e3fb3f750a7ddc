/*
 * gdl_oracle.c -- CPU restatement (plain C, fp32, NCHW) of the arithmetic on the
 * DGL hot path of shicaiwei123/ICCV2025-GDL.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (iccv2025-gdl_amd/) never does.
 *
 * The reference expresses this arithmetic through PyTorch operators (PyTorch 1.11
 * per /root/reference/README.md:7-11; not vendored under /root/reference).  Each
 * function cites the reference call site it stands in for and follows the
 * published semantics of that operator.  The restatement is pinned against golden
 * vectors captured from the imported reference (tests/golden/make_golden.py,
 * PyTorch 2.10 CPU fp32) by tests/test_oracle_golden.py.
 *
 * Layout: x[N][C][H][W], w[K][C][R][S], row-major float32.  Reductions that the
 * reference performs in a numerically careful way (BatchNorm statistics, norms,
 * cross-entropy) accumulate in double here; convolution inner products accumulate
 * in float like a plain fp32 kernel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ convolution
 * nn.Conv2d(bias=False): backbone.py:20-23 (conv3x3), :26-28 (conv1x1), :96-101 (7x7 stem). */
ORC_API void orc_conv2d_fwd(const float* x, const float* w, float* y, int N, int C, int H, int W, int K, int R, int S,
                            int stride, int pad) {
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            float* yp = y + ((size_t)n * K + k) * P * Q;
            memset(yp, 0, sizeof(float) * (size_t)P * Q);
            for (int c = 0; c < C; ++c) {
                const float* xp = x + ((size_t)n * C + c) * H * W;
                const float* wp = w + ((size_t)k * C + c) * R * S;
                for (int r = 0; r < R; ++r)
                    for (int s = 0; s < S; ++s) {
                        const float wv = wp[r * S + s];
                        /* valid q: 0 <= q*stride - pad + s < W */
                        int q0 = (pad - s + stride - 1) / stride;
                        if (pad - s < 0) q0 = 0;
                        int q1 = (W - 1 + pad - s) / stride; /* inclusive */
                        if (q1 > Q - 1) q1 = Q - 1;
                        for (int p = 0; p < P; ++p) {
                            const int ih = p * stride - pad + r;
                            if (ih < 0 || ih >= H) continue;
                            const float* xr = xp + (size_t)ih * W - pad + s;
                            float* yr = yp + (size_t)p * Q;
                            if (stride == 1)
                                for (int q = q0; q <= q1; ++q) yr[q] += wv * xr[q];
                            else
                                for (int q = q0; q <= q1; ++q) yr[q] += wv * xr[q * stride];
                        }
                    }
            }
        }
}

/* autograd of Conv2d w.r.t. its input (conv transpose of dy with w). */
ORC_API void orc_conv2d_bwd_data(const float* dy, const float* w, float* dx, int N, int C, int H, int W, int K, int R,
                                 int S, int stride, int pad) {
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            float* dxp = dx + ((size_t)n * C + c) * H * W;
            memset(dxp, 0, sizeof(float) * (size_t)H * W);
            for (int k = 0; k < K; ++k) {
                const float* dyp = dy + ((size_t)n * K + k) * P * Q;
                const float* wp = w + ((size_t)k * C + c) * R * S;
                for (int r = 0; r < R; ++r)
                    for (int s = 0; s < S; ++s) {
                        const float wv = wp[r * S + s];
                        int q0 = (pad - s + stride - 1) / stride;
                        if (pad - s < 0) q0 = 0;
                        int q1 = (W - 1 + pad - s) / stride;
                        if (q1 > Q - 1) q1 = Q - 1;
                        for (int p = 0; p < P; ++p) {
                            const int ih = p * stride - pad + r;
                            if (ih < 0 || ih >= H) continue;
                            float* xr = dxp + (size_t)ih * W - pad + s;
                            const float* yr = dyp + (size_t)p * Q;
                            if (stride == 1)
                                for (int q = q0; q <= q1; ++q) xr[q] += wv * yr[q];
                            else
                                for (int q = q0; q <= q1; ++q) xr[q * stride] += wv * yr[q];
                        }
                    }
            }
        }
}

/* autograd of Conv2d w.r.t. its weight. */
ORC_API void orc_conv2d_bwd_weight(const float* dy, const float* x, float* dw, int N, int C, int H, int W, int K, int R,
                                   int S, int stride, int pad) {
    const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < C; ++c) {
            float* dwp = dw + ((size_t)k * C + c) * R * S;
            for (int r = 0; r < R; ++r)
                for (int s = 0; s < S; ++s) {
                    int q0 = (pad - s + stride - 1) / stride;
                    if (pad - s < 0) q0 = 0;
                    int q1 = (W - 1 + pad - s) / stride;
                    if (q1 > Q - 1) q1 = Q - 1;
                    double acc = 0.0;
                    for (int n = 0; n < N; ++n) {
                        const float* xp = x + ((size_t)n * C + c) * H * W;
                        const float* dyp = dy + ((size_t)n * K + k) * P * Q;
                        for (int p = 0; p < P; ++p) {
                            const int ih = p * stride - pad + r;
                            if (ih < 0 || ih >= H) continue;
                            const float* xr = xp + (size_t)ih * W - pad + s;
                            const float* yr = dyp + (size_t)p * Q;
                            float a = 0.f;
                            if (stride == 1)
                                for (int q = q0; q <= q1; ++q) a += xr[q] * yr[q];
                            else
                                for (int q = q0; q <= q1; ++q) a += xr[q * stride] * yr[q];
                            acc += a;
                        }
                    }
                    dwp[r * S + s] = (float)acc;
                }
        }
}

/* ------------------------------------------------------------------ BatchNorm2d
 * nn.BatchNorm2d(eps=1e-5, momentum=0.1, affine, track_running_stats):
 * backbone.py:45,48,104,144.  Training: biased batch variance normalises, the
 * running variance receives the unbiased one; num_batches_tracked += 1 is done by
 * the caller. */
ORC_API void orc_bn_fwd_train(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                              float* save_invstd, float* running_mean, float* running_var, int N, int C, int HW,
                              float eps, float momentum) {
    const double cnt = (double)N * HW;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int n = 0; n < N; ++n) {
            const float* xp = x + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i) s += xp[i];
        }
        const double mean = s / cnt;
        double v = 0.0;
        for (int n = 0; n < N; ++n) {
            const float* xp = x + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i) {
                const double d = xp[i] - mean;
                v += d * d;
            }
        }
        const double var = v / cnt;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[c] = (float)mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const double unb = cnt > 1 ? v / (cnt - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
        const float g = gamma[c], b = beta[c], m = (float)mean;
        for (int n = 0; n < N; ++n) {
            const float* xp = x + ((size_t)n * C + c) * HW;
            float* yp = y + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i) yp[i] = (xp[i] - m) * invstd * g + b;
        }
    }
}

ORC_API void orc_bn_fwd_eval(const float* x, const float* gamma, const float* beta, float* y, const float* running_mean,
                             const float* running_var, int N, int C, int HW, float eps) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float invstd = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
        const float g = gamma[c], b = beta[c], m = running_mean[c];
        for (int n = 0; n < N; ++n) {
            const float* xp = x + ((size_t)n * C + c) * HW;
            float* yp = y + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i) yp[i] = (xp[i] - m) * invstd * g + b;
        }
    }
}

/* training-mode BatchNorm backward:
 *   dbeta = sum dy, dgamma = sum dy*xhat,
 *   dx = gamma*invstd*(dy - dbeta/M - xhat*dgamma/M). */
ORC_API void orc_bn_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean,
                        const float* save_invstd, float* dx, float* dgamma, float* dbeta, int N, int C, int HW) {
    const double cnt = (double)N * HW;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float m = save_mean[c], is = save_invstd[c];
        double sdy = 0.0, sdyx = 0.0;
        for (int n = 0; n < N; ++n) {
            const float* xp = x + ((size_t)n * C + c) * HW;
            const float* dp = dy + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i) {
                sdy += dp[i];
                sdyx += (double)dp[i] * ((xp[i] - m) * is);
            }
        }
        dgamma[c] = (float)sdyx;
        dbeta[c] = (float)sdy;
        const float k1 = (float)(sdy / cnt), k2 = (float)(sdyx / cnt), gs = gamma[c] * is;
        for (int n = 0; n < N; ++n) {
            const float* xp = x + ((size_t)n * C + c) * HW;
            const float* dp = dy + ((size_t)n * C + c) * HW;
            float* op = dx + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i) op[i] = gs * (dp[i] - k1 - (xp[i] - m) * is * k2);
        }
    }
}

/* ------------------------------------------------------------------ ReLU / add
 * nn.ReLU(inplace=True), `out += identity`: backbone.py:46,57,65-66,105. */
ORC_API void orc_relu_fwd(float* x, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f;
}
ORC_API void orc_relu_bwd(const float* dy, const float* y, float* dx, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
ORC_API void orc_add_inplace(float* a, const float* b, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) a[i] += b[i];
}

/* ------------------------------------------------------------------ MaxPool2d(3,2,1)
 * backbone.py:106.  -inf padding; the first maximum in row-major window order
 * wins (strict '>' update), as in ATen's CPU kernel. */
ORC_API void orc_maxpool3x3s2_fwd(const float* x, float* y, int32_t* idx, int N, int C, int H, int W) {
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc) {
        const float* xp = x + (size_t)nc * H * W;
        float* yp = y + (size_t)nc * P * Q;
        int32_t* ip = idx + (size_t)nc * P * Q;
        for (int p = 0; p < P; ++p)
            for (int q = 0; q < Q; ++q) {
                float best = -INFINITY;
                int bi = -1;
                for (int r = 0; r < 3; ++r) {
                    const int ih = p * 2 - 1 + r;
                    if (ih < 0 || ih >= H) continue;
                    for (int s = 0; s < 3; ++s) {
                        const int iw = q * 2 - 1 + s;
                        if (iw < 0 || iw >= W) continue;
                        const float v = xp[ih * W + iw];
                        if (v > best || bi < 0 || isnan(v)) {
                            best = v;
                            bi = ih * W + iw;
                        }
                    }
                }
                yp[p * Q + q] = best;
                ip[p * Q + q] = bi;
            }
    }
}
ORC_API void orc_maxpool3x3s2_bwd(const float* dy, const int32_t* idx, float* dx, int N, int C, int H, int W) {
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc) {
        float* xp = dx + (size_t)nc * H * W;
        memset(xp, 0, sizeof(float) * (size_t)H * W);
        const float* yp = dy + (size_t)nc * P * Q;
        const int32_t* ip = idx + (size_t)nc * P * Q;
        for (int i = 0; i < P * Q; ++i) xp[ip[i]] += yp[i];
    }
}

/* ------------------------------------------------------------------ global average pools
 * basic_model.py:73-82: audio adaptive_avg_pool2d(a,1) over h*w; visual
 * view(B,T,C,H,W).permute(0,2,1,3,4) + adaptive_avg_pool3d(v,1) over T*h*w.
 * x is [B*T][C][HW]; T=1 gives the 2-D pool. */
ORC_API void orc_avgpool_fwd(const float* x, float* y, int B, int T, int C, int HW) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            double s = 0.0;
            for (int t = 0; t < T; ++t) {
                const float* xp = x + (((size_t)b * T + t) * C + c) * HW;
                for (int i = 0; i < HW; ++i) s += xp[i];
            }
            y[(size_t)b * C + c] = (float)(s / ((double)T * HW));
        }
}
ORC_API void orc_avgpool_bwd(const float* dy, float* dx, int B, int T, int C, int HW) {
    const float inv = 1.0f / (float)(T * HW);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float g = dy[(size_t)b * C + c] * inv;
            for (int t = 0; t < T; ++t) {
                float* xp = dx + (((size_t)b * T + t) * C + c) * HW;
                for (int i = 0; i < HW; ++i) xp[i] = g;
            }
        }
}

/* ------------------------------------------------------------------ Linear
 * nn.Linear: fusion_modules.py:36,48 (fc_out). y = x W^T + b. */
ORC_API void orc_linear_fwd(const float* x, const float* w, const float* b, float* y, int B, int I, int O) {
    for (int n = 0; n < B; ++n)
        for (int o = 0; o < O; ++o) {
            double s = b ? b[o] : 0.0;
            for (int i = 0; i < I; ++i) s += (double)x[(size_t)n * I + i] * w[(size_t)o * I + i];
            y[(size_t)n * O + o] = (float)s;
        }
}
/* dx = dy W ; dW += dy^T x ; db += sum dy   (dx may be NULL; dW/db accumulate) */
ORC_API void orc_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int B,
                            int I, int O) {
    if (dx)
        for (int n = 0; n < B; ++n)
            for (int i = 0; i < I; ++i) {
                double s = 0.0;
                for (int o = 0; o < O; ++o) s += (double)dy[(size_t)n * O + o] * w[(size_t)o * I + i];
                dx[(size_t)n * I + i] = (float)s;
            }
    if (dw)
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i) {
                double s = 0.0;
                for (int n = 0; n < B; ++n) s += (double)dy[(size_t)n * O + o] * x[(size_t)n * I + i];
                dw[(size_t)o * I + i] += (float)s;
            }
    if (db)
        for (int o = 0; o < O; ++o) {
            double s = 0.0;
            for (int n = 0; n < B; ++n) s += dy[(size_t)n * O + o];
            db[o] += (float)s;
        }
}

/* ------------------------------------------------------------------ CrossEntropyLoss
 * nn.CrossEntropyLoss() (mean reduction): main_dgl.py:71,102-104.
 * Returns the loss; dlogits (may be NULL) = scale * (softmax - onehot) / B. */
ORC_API double orc_softmax_ce(const float* logits, const int64_t* labels, float* dlogits, int B, int n, double scale) {
    double loss = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* lp = logits + (size_t)b * n;
        double mx = lp[0];
        for (int j = 1; j < n; ++j) mx = lp[j] > mx ? lp[j] : mx;
        double se = 0.0;
        for (int j = 0; j < n; ++j) se += exp((double)lp[j] - mx);
        const double lse = mx + log(se);
        loss += lse - (double)lp[labels[b]];
        if (dlogits)
            for (int j = 0; j < n; ++j) {
                const double p = exp((double)lp[j] - lse);
                dlogits[(size_t)b * n + j] = (float)(scale * (p - (j == labels[b] ? 1.0 : 0.0)) / B);
            }
    }
    return loss / B;
}

/* ------------------------------------------------------------------ clip / stats / SGD
 * clip_grad_norm_(params, 40, 2): main_dgl.py:129; torch.abs(p.grad).mean(): :137,143;
 * optim.SGD(momentum=0.9, weight_decay=1e-4): :249,154. */
ORC_API double orc_sumsq(const float* g, size_t n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (size_t i = 0; i < n; ++i) s += (double)g[i] * g[i];
    return s;
}
ORC_API double orc_abs_mean(const float* g, size_t n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (size_t i = 0; i < n; ++i) s += fabs((double)g[i]);
    return s / (double)n;
}
ORC_API void orc_scale(float* g, size_t n, float s) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) g[i] *= s;
}
/* torch.optim.SGD single-tensor update: g += wd*p; buf = first ? g : mu*buf + g; p -= lr*buf */
ORC_API void orc_sgd(float* p, const float* g, float* buf, size_t n, float lr, float mu, float wd, int first) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float gi = g[i] + wd * p[i];
        float b = first ? gi : mu * buf[i] + gi;
        buf[i] = b;
        p[i] -= lr * b;
    }
}

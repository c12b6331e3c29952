/* CPU twins of a subset of the C ABI (include/timetuning_hip.h).  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C restatements, on HOST pointers, with the SAME argument lists as their tt_* counterparts (the stream and any
 * workspace arguments are accepted and ignored), so that one ctypes prototype drives either side.  They cover the ops whose
 * arithmetic is integer / byte exact (the Pillow-defined image transforms, the confusion matrix) or a short, order-defined
 * float recurrence (Sinkhorn-Knopp, cross-entropy, arg-max of a bilinear upsampling, nearest-centroid assignment, column
 * moments) and - the rest of the file, round 2 - the hot path's row ops, naive matrix products, backward ops, the plane ops, the
 * coarse entry points, label propagation, the foreground mask and the evaluator's resampling: every compute entry point of the
 * header has a twin here (the *_workspace_bytes twins return 0: scratch is malloc'ed).
 *
 * Built by __graft_entry__.build() (gcc -O2 -shared) into oracle/_build/libtt_cpu.so; only tests/ load it.  Every function
 * cites the reference lines (paths relative to /root/reference) or the third-party algorithm it restates. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void* tt_stream_t;

/* ---- my_utils.py:246-274 sinkhorn + time_tuning.py:157-168: Q = exp(scores / eps)^T, iters x (row step, column step), then
 *      the final column normalisation; rows [row0, row0 + rows_out) of the transposed result.  float32 throughout, sums in
 *      index order (the reference's torch.sum orders are not defined; tests compare at 1e-5). */
int tt_cpu_sinkhorn(const float* scores, float* q_out, int B_total, int K, int row0, int rows_out, float eps, int iters, void* workspace,
                    size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  const size_t n = (size_t)B_total * K;
  float* Q = (float*)malloc(n * sizeof(float)); /* Q[k][b] stored as E[b][k] */
  if (!Q) return -3;
  double total = 0.0;
  for (size_t i = 0; i < n; ++i) { Q[i] = expf(scores[i] / eps); total += Q[i]; }
  for (size_t i = 0; i < n; ++i) Q[i] = (float)(Q[i] / total);
  const float r = 1.0f / (float)K, c = 1.0f / (float)B_total;
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < K; ++k) {            /* u = rowsum(Q); Q *= (r / u)[:, None] */
      float u = 0.f;
      for (int b = 0; b < B_total; ++b) u += Q[(size_t)b * K + k];
      const float f = r / u;
      for (int b = 0; b < B_total; ++b) Q[(size_t)b * K + k] *= f;
    }
    for (int b = 0; b < B_total; ++b) {      /* Q *= (c / colsum(Q))[None, :] */
      float v = 0.f;
      for (int k = 0; k < K; ++k) v += Q[(size_t)b * K + k];
      const float f = c / v;
      for (int k = 0; k < K; ++k) Q[(size_t)b * K + k] *= f;
    }
  }
  for (int b = 0; b < rows_out; ++b) {       /* (Q / colsum(Q)).T */
    const float* row = Q + (size_t)(row0 + b) * K;
    float v = 0.f;
    for (int k = 0; k < K; ++k) v += row[k];
    for (int k = 0; k < K; ++k) q_out[(size_t)b * K + k] = row[k] / v;
  }
  free(Q);
  return 0;
}

/* ---- my_utils.py:250-272, one rank's share in steps (the caller all-reduces u between them): Q [K][B_loc] lives in the workspace as
 *      E[b][k] and is rescaled IN PLACE as the reference does.  (The reference first divides Q by its all-reduced total mass: a common
 *      factor that the first row step removes - not applied here, as in the HIP path.) */
size_t tt_cpu_sinkhorn_local_workspace_bytes(int B_loc, int K) { return (size_t)B_loc * K * sizeof(float); }
static void sk_local_rowsums(const float* Q, float* u, int B, int K) {
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += Q[(size_t)b * K + k];
    u[k] = s;
  }
}
int tt_cpu_sinkhorn_local_begin(const float* scores, float* u_out, int B_loc, int K, float eps, void* workspace, size_t workspace_bytes,
                                tt_stream_t stream) {
  (void)stream;
  if (!workspace || workspace_bytes < tt_cpu_sinkhorn_local_workspace_bytes(B_loc, K)) return -1;
  float* Q = (float*)workspace;
  for (size_t i = 0; i < (size_t)B_loc * K; ++i) Q[i] = expf(scores[i] / eps);
  sk_local_rowsums(Q, u_out, B_loc, K);
  return 0;
}
int tt_cpu_sinkhorn_local_step(const float* u_in, float* u_out, int B_loc, int B_total, int K, void* workspace, size_t workspace_bytes,
                               tt_stream_t stream) {
  (void)stream;
  if (!workspace || workspace_bytes < tt_cpu_sinkhorn_local_workspace_bytes(B_loc, K)) return -1;
  float* Q = (float*)workspace;
  const float r = 1.0f / (float)K, c = 1.0f / (float)B_total;
  for (int k = 0; k < K; ++k) {              /* Q *= (r / u)[:, None], u all-reduced */
    const float f = r / u_in[k];
    for (int b = 0; b < B_loc; ++b) Q[(size_t)b * K + k] *= f;
  }
  for (int b = 0; b < B_loc; ++b) {          /* Q *= (c / colsum(Q))[None, :] */
    float v = 0.f;
    for (int k = 0; k < K; ++k) v += Q[(size_t)b * K + k];
    const float f = c / v;
    for (int k = 0; k < K; ++k) Q[(size_t)b * K + k] *= f;
  }
  sk_local_rowsums(Q, u_out, B_loc, K);
  return 0;
}
int tt_cpu_sinkhorn_local_end(const float* u_in, float* q_out, int B_loc, int rows_out, int K, void* workspace, size_t workspace_bytes,
                              tt_stream_t stream) {
  (void)stream;
  if (!workspace || workspace_bytes < tt_cpu_sinkhorn_local_workspace_bytes(B_loc, K)) return -1;
  float* Q = (float*)workspace;
  const float r = 1.0f / (float)K;
  if (u_in)
    for (int k = 0; k < K; ++k) {
      const float f = r / u_in[k];
      for (int b = 0; b < B_loc; ++b) Q[(size_t)b * K + k] *= f;
    }
  for (int b = 0; b < rows_out; ++b) {       /* (Q / colsum(Q)).T */
    const float* row = Q + (size_t)b * K;
    float v = 0.f;
    for (int k = 0; k < K; ++k) v += row[k];
    for (int k = 0; k < K; ++k) q_out[(size_t)b * K + k] = row[k] / v;
  }
  return 0;
}

/* ---- time_tuning.py:296-302 (+ :226-227,298-300 with row_weight): mean over rows of weight * CE(scores / T, label), and
 *      its gradient with respect to scores. */
int tt_cpu_ce_loss_fwd_bwd(const float* scores, const int64_t* labels, const float* row_weight, float* loss_out, float* dscores, int rows,
                           int K, float temperature, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  double acc = 0.0;
  for (int r = 0; r < rows; ++r) {
    const float* s = scores + (size_t)r * K;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s[k] / temperature);
    double sum = 0.0;
    for (int k = 0; k < K; ++k) sum += exp((double)(s[k] / temperature - mx));
    const double lse = (double)mx + log(sum);
    const float w = row_weight ? row_weight[r] : 1.0f;
    acc += w * (lse - (double)(s[labels[r]] / temperature));
    if (dscores)
      for (int k = 0; k < K; ++k) {
        const double p = exp((double)(s[k] / temperature) - lse);
        dscores[(size_t)r * K + k] = (float)(w * (p - (k == labels[r] ? 1.0 : 0.0)) / ((double)temperature * rows));
      }
  }
  loss_out[0] = (float)(acc / rows);
  return 0;
}

/* ---- Pillow Resample.c (Image.resize BILINEAR as called by video_transformations.py:56-94): one pass, 22-bit fixed point */
static unsigned char clip8(long long v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

int tt_cpu_img_resample_h(const unsigned char* in, unsigned char* out, const int* coeffs, const int* bounds, int F, int H, int W, int y0,
                          int x0, int h, int OW, int ksize, tt_stream_t stream) {
  (void)stream;
  for (int f = 0; f < F; ++f)
    for (int y = 0; y < h; ++y)
      for (int xx = 0; xx < OW; ++xx) {
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const unsigned char* row = in + (((size_t)f * H + (y0 + y)) * W + x0 + xmin) * 3;
        for (int c = 0; c < 3; ++c) {
          int a = 1 << 21;
          for (int x = 0; x < cnt; ++x) a += row[3 * x + c] * coeffs[(size_t)xx * ksize + x];
          out[(((size_t)f * h + y) * OW + xx) * 3 + c] = clip8(a >> 22);
        }
      }
  return 0;
}

int tt_cpu_img_resample_v(const unsigned char* in, unsigned char* out_u8, float* out_f32, const int* coeffs, const int* bounds, int F,
                          int Hin, int W, int y0, int OH, int ksize, int flip, const float* mean3, const float* std3, tt_stream_t stream) {
  (void)stream;
  for (int f = 0; f < F; ++f)
    for (int yy = 0; yy < OH; ++yy)
      for (int x = 0; x < W; ++x) {
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        for (int c = 0; c < 3; ++c) {
          int a = 1 << 21;
          for (int y = 0; y < cnt; ++y) a += in[(((size_t)f * Hin + (y0 + ymin + y)) * W + x) * 3 + c] * coeffs[(size_t)yy * ksize + y];
          const unsigned char v = clip8(a >> 22);
          if (out_f32) {  /* ToTensor + (x - mean) / std, optional horizontal flip (video_transformations.py:168-179,262-276) */
            const int ox = flip ? W - 1 - x : x;
            out_f32[(((size_t)f * 3 + c) * OH + yy) * W + ox] = ((float)v / 255.0f - mean3[c]) / std3[c];
          } else {
            out_u8[(((size_t)f * OH + yy) * W + x) * 3 + c] = v;
          }
        }
      }
  return 0;
}

/* ---- Pillow Convert.c / Blend.c / ImageEnhance and torchvision's adjust_hue: see oracle/image_ops.py for the derivation */
static unsigned char gray_of(unsigned r, unsigned g, unsigned b) { return (unsigned char)((r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16); }

static unsigned char blend8(int in1, int in2, float alpha) {
  const float t = (float)in1 + alpha * (float)(in2 - in1);
  if (alpha >= 0.f && alpha <= 1.0f) return (unsigned char)(int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)(int)t);
}

int tt_cpu_img_color(unsigned char* img, int F, int H, int W, int mode, float factor, int hue_shift, unsigned long long* gray_sums,
                     tt_stream_t stream) {
  (void)gray_sums; (void)stream;
  const size_t npix = (size_t)H * W;
  for (int f = 0; f < F; ++f) {
    unsigned char* base = img + (size_t)f * npix * 3;
    int mean = 0;
    if (mode == 2) {
      unsigned long long s = 0;
      for (size_t i = 0; i < npix; ++i) s += gray_of(base[3 * i], base[3 * i + 1], base[3 * i + 2]);
      mean = (int)((double)s / (double)npix + 0.5);
    }
    for (size_t i = 0; i < npix; ++i) {
      unsigned char* p = base + 3 * i;
      const int r = p[0], g = p[1], b = p[2];
      if (mode == 0) {
        p[0] = p[1] = p[2] = gray_of(r, g, b);
      } else if (mode == 1) {
        p[0] = blend8(0, r, factor); p[1] = blend8(0, g, factor); p[2] = blend8(0, b, factor);
      } else if (mode == 2) {
        p[0] = blend8(mean, r, factor); p[1] = blend8(mean, g, factor); p[2] = blend8(mean, b, factor);
      } else if (mode == 3) {
        const int y = gray_of(r, g, b);
        p[0] = blend8(y, r, factor); p[1] = blend8(y, g, factor); p[2] = blend8(y, b, factor);
      } else {
        const int maxc = r > g ? (r > b ? r : b) : (g > b ? g : b), minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
        int uh = 0, us = 0;
        if (minc != maxc) {
          const float cr = (float)(maxc - minc), s = cr / (float)maxc;
          const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
          float h;
          if (r == maxc) h = bc - gc;
          else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
          else h = (float)(4.0 + (double)gc - (double)rc);
          h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
          uh = (int)((double)h * 255.0); us = (int)((double)s * 255.0);
          uh = uh < 0 ? 0 : (uh > 255 ? 255 : uh); us = us < 0 ? 0 : (us > 255 ? 255 : us);
        }
        const int hh = (uh + hue_shift) & 255;
        if (us == 0) {
          p[0] = p[1] = p[2] = (unsigned char)maxc;
        } else {
          const float hf = (float)hh * 6.0f / 255.0f;
          const int ii = (int)floorf(hf);
          const float fr = hf - (float)ii, fs = (float)us / 255.0f, fv = (float)maxc;
          const unsigned char P = clip8((long long)round((double)(fv * (1.0f - fs)))), Q = clip8((long long)round((double)(fv * (1.0f - fs * fr)))),
                              T = clip8((long long)round((double)(fv * (1.0f - fs * (1.0f - fr))))), V = (unsigned char)maxc;
          switch (ii % 6) {
            case 0: p[0] = V; p[1] = T; p[2] = P; break;
            case 1: p[0] = Q; p[1] = V; p[2] = P; break;
            case 2: p[0] = P; p[1] = V; p[2] = T; break;
            case 3: p[0] = P; p[1] = Q; p[2] = V; break;
            case 4: p[0] = T; p[1] = P; p[2] = V; break;
            default: p[0] = V; p[1] = P; p[2] = Q; break;
          }
        }
      }
    }
  }
  return 0;
}

/* ---- Pillow BoxBlur.c, one extended-box pass along x (0) or y (1) */
int tt_cpu_img_box_blur(const unsigned char* in, unsigned char* out, int F, int H, int W, int direction, int radius, unsigned ww, unsigned fw,
                        tt_stream_t stream) {
  (void)stream;
  for (int f = 0; f < F; ++f)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x)
        for (int c = 0; c < 3; ++c) {
          const int len = direction == 0 ? W : H, pos = direction == 0 ? x : y;
          unsigned long long acc = 0;
#define PIX(q) in[(((size_t)f * H + (direction == 0 ? y : (q))) * W + (direction == 0 ? (q) : x)) * 3 + c]
          for (int d = -radius; d <= radius; ++d) {
            int q = pos + d;
            q = q < 0 ? 0 : (q > len - 1 ? len - 1 : q);
            acc += PIX(q);
          }
          int ql = pos - radius - 1, qr = pos + radius + 1;
          ql = ql < 0 ? 0 : ql; qr = qr > len - 1 ? len - 1 : qr;
          out[(((size_t)f * H + y) * W + x) * 3 + c] = (unsigned char)((acc * ww + (unsigned long long)(PIX(ql) + PIX(qr)) * fw + (1ull << 23)) >> 24);
#undef PIX
        }
  return 0;
}

/* ---- the confusion matrix behind metrics.py:357-432 and the Jaccard index */
int tt_cpu_confusion_counts(const int64_t* pred, const int64_t* gt, long long n, int C, unsigned long long* counts, tt_stream_t stream) {
  (void)stream;
  memset(counts, 0, sizeof(unsigned long long) * (size_t)C * C);
  for (long long i = 0; i < n; ++i)
    if (pred[i] >= 0 && pred[i] < C && gt[i] >= 0 && gt[i] < C) counts[gt[i] * C + pred[i]] += 1;
  return 0;
}

/* ---- mask_propagation.py:828-829: F.interpolate(bilinear, align_corners=False) of fp64 maps [M, n, K] then arg-max over K */
int tt_cpu_upsample_argmax(const double* maps, int64_t* labels_out, int M, int g, int K, int R, tt_stream_t stream) {
  (void)stream;
  const double scale = (double)g / (double)R;
  for (int m = 0; m < M; ++m)
    for (int oy = 0; oy < R; ++oy)
      for (int ox = 0; ox < R; ++ox) {
        double sy = scale * (oy + 0.5) - 0.5, sx = scale * (ox + 0.5) - 0.5;
        sy = sy < 0 ? 0 : sy; sx = sx < 0 ? 0 : sx;
        const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < g - 1), x1 = x0 + (x0 < g - 1);
        const double ly = sy - y0, lx = sx - x0, hy = 1.0 - ly, hx = 1.0 - lx;
        const double* b = maps + (size_t)m * g * g * K;
        double best = -INFINITY;
        int besti = 0;
        for (int k = 0; k < K; ++k) {
          const double v = hy * (hx * b[(size_t)(y0 * g + x0) * K + k] + lx * b[(size_t)(y0 * g + x1) * K + k]) +
                           ly * (hx * b[(size_t)(y1 * g + x0) * K + k] + lx * b[(size_t)(y1 * g + x1) * K + k]);
          if (v > best) { best = v; besti = k; }
        }
        labels_out[((size_t)m * R + oy) * R + ox] = besti;
      }
  return 0;
}

/* ---- Lloyd assignment step of the k-means the reference delegates to faiss (clustering.py:39-41): nearest centroid, first minimum */
int tt_cpu_kmeans_assign(const float* x, const float* centroids, int32_t* labels, float* dist2, long long P, int d, int k, tt_stream_t stream) {
  (void)stream;
  for (long long p = 0; p < P; ++p) {
    float best = INFINITY;
    int besti = 0;
    for (int j = 0; j < k; ++j) {
      float s = 0.f;
      for (int t = 0; t < d; ++t) { const float df = x[p * d + t] - centroids[(size_t)j * d + t]; s += df * df; }
      if (s < best) { best = s; besti = j; }
    }
    labels[p] = besti;
    if (dist2) dist2[p] = best;
  }
  return 0;
}

/* ---- StandardScaler's statistics (my_utils.py:24-28): per-column mean and population variance */
int tt_cpu_col_moments(const float* x, double* mean, double* var, long long rows, int cols, void* workspace, size_t workspace_bytes,
                       tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  for (int c = 0; c < cols; ++c) {
    double s = 0.0, s2 = 0.0;
    for (long long r = 0; r < rows; ++r) { const double v = x[r * cols + c]; s += v; s2 += v * v; }
    mean[c] = s / (double)rows;
    const double v = s2 / (double)rows - mean[c] * mean[c];
    var[c] = v > 0 ? v : 0;
  }
  return 0;
}

/* =====================================================================================================================
 * Round 2: twins of the hot path's row ops and (naive, small sizes) of its matrix products.  double accumulation where the
 * device kernels use fp32 chains: the tests compare at the f32-MFMA bound (2e-5 relative), not bit for bit.
 * ===================================================================================================================== */

static float tt_cpu_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
static float tt_cpu_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

/* ---- nn.Linear forward (dino_vision_transformer.py:94-103,115-130; models.py:915-926): y = act(x w^T + b) (+ residual) */
/* (`precision`, ABI 8: which MFMA arithmetic the GPU entry point runs the product in; the twin is the exact product whatever it says) */
int tt_cpu_linear_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, float* pre_act, int M, int N,
                      int K, int act, int precision, tt_stream_t stream) {
  (void)stream; (void)precision;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = bias ? (double)bias[n] : 0.0;
      for (int k = 0; k < K; ++k) s += (double)x[(size_t)m * K + k] * (double)w[(size_t)n * K + k];
      float v = (float)s;
      if (pre_act) pre_act[(size_t)m * N + n] = v;
      if (act == 1) v = tt_cpu_gelu(v);
      if (residual) v += residual[(size_t)m * N + n];
      y[(size_t)m * N + n] = v;
    }
  return 0;
}

/* ---- autograd of the above: dx = dy w (* gelu'(pre)); dw = dy^T x, db = colsum(dy) */
int tt_cpu_linear_bwd_data(const float* dy, const float* w, const float* gelu_pre, float* dx, int M, int N, int K, tt_stream_t stream) {
  (void)stream;
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int n = 0; n < N; ++n) s += (double)dy[(size_t)m * N + n] * (double)w[(size_t)n * K + k];
      float v = (float)s;
      if (gelu_pre) v *= tt_cpu_gelu_grad(gelu_pre[(size_t)m * K + k]);
      dx[(size_t)m * K + k] = v;
    }
  return 0;
}
int tt_cpu_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, void* workspace,
                             size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  for (int n = 0; n < N; ++n) {
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int m = 0; m < M; ++m) s += (double)dy[(size_t)m * N + n] * (double)x[(size_t)m * K + k];
      dw[(size_t)n * K + k] = (float)s;
    }
    if (db) {
      double s = 0.0;
      for (int m = 0; m < M; ++m) s += (double)dy[(size_t)m * N + n];
      db[n] = (float)s;
    }
  }
  return 0;
}

int tt_cpu_linear_bwd(const float* dy, const float* w, const float* x, const float* gelu_pre, float* dx, float* dw, float* db, int M, int N,
                      int K, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  const int rc = tt_cpu_linear_bwd_weight(dy, x, dw, db, M, N, K, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return tt_cpu_linear_bwd_data(dy, w, gelu_pre, dx, M, N, K, stream);
}

/* ---- nn.LayerNorm(eps) (dino_vision_transformer.py:139,143,196); skip_group = N drops token 0 of every group of N rows
 *      (the cls token, models.py:967) */
int tt_cpu_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int rows, int D,
                         float eps, int skip_group, tt_stream_t stream) {
  (void)stream;
  for (int r = 0; r < rows; ++r) {
    const long long in_row = skip_group ? (long long)(r / (skip_group - 1)) * skip_group + 1 + r % (skip_group - 1) : r;
    const float* xr = x + in_row * D;
    double mu = 0.0, var = 0.0;
    for (int c = 0; c < D; ++c) mu += xr[c];
    mu /= D;
    for (int c = 0; c < D; ++c) var += (xr[c] - mu) * (xr[c] - mu);
    const double rs = 1.0 / sqrt(var / D + (double)eps);
    for (int c = 0; c < D; ++c) y[(size_t)r * D + c] = (float)((xr[c] - mu) * rs * gamma[c] + beta[c]);
    if (mean) mean[r] = (float)mu;
    if (rstd) rstd[r] = (float)rs;
  }
  return 0;
}

/* ---- F.normalize(x, dim=-1) (time_tuning.py:136) and the in-place prototype renormalisation (:124-128); eps 1e-12 */
int tt_cpu_l2norm_fwd(const float* x, int ldx, float* xn, float* inv_norm, int rows, int D, tt_stream_t stream) {
  (void)stream;
  for (int r = 0; r < rows; ++r) {
    double s = 0.0;
    for (int c = 0; c < D; ++c) s += (double)x[(size_t)r * ldx + c] * (double)x[(size_t)r * ldx + c];
    const double nrm = sqrt(s);
    const float inv = (float)(1.0 / (nrm > 1e-12 ? nrm : 1e-12));
    for (int c = 0; c < D; ++c) xn[(size_t)r * D + c] = x[(size_t)r * ldx + c] * inv;
    if (inv_norm) inv_norm[r] = inv;
  }
  return 0;
}
int tt_cpu_normalize_rows_inplace(float* w, int rows, int D, tt_stream_t stream) { return tt_cpu_l2norm_fwd(w, D, w, NULL, rows, D, stream); }

/* ---- Attention.forward between the qkv and proj Linears (dino_vision_transformer.py:122-129): softmax(q k^T scale) v per
 *      (frame, head); qkv [F, N, 3 H hd] as the Linear wrote it; optional lse [F,H,N] and probs [F,H,N,N] */
int tt_cpu_attention_fwd(const float* qkv, float* out, float* lse, float* probs, int F, int N, int H, int hd, float scale,
                         tt_stream_t stream) {
  (void)stream;
  const int D = H * hd, D3 = 3 * D;
  double* p = (double*)malloc((size_t)N * sizeof(double));
  if (!p) return -3;
  for (int f = 0; f < F; ++f)
    for (int h = 0; h < H; ++h)
      for (int i = 0; i < N; ++i) {
        const float* q = qkv + ((size_t)f * N + i) * D3 + h * hd;
        double mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
          const float* k = qkv + ((size_t)f * N + j) * D3 + D + h * hd;
          double s = 0.0;
          for (int d = 0; d < hd; ++d) s += (double)q[d] * (double)k[d];
          p[j] = s * scale;
          if (p[j] > mx) mx = p[j];
        }
        double sum = 0.0;
        for (int j = 0; j < N; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; }
        if (lse) lse[((size_t)f * H + h) * N + i] = (float)(mx + log(sum));
        for (int j = 0; j < N; ++j) {
          p[j] /= sum;
          if (probs) probs[(((size_t)f * H + h) * N + i) * N + j] = (float)p[j];
        }
        for (int d = 0; d < hd; ++d) {
          double o = 0.0;
          for (int j = 0; j < N; ++j) o += p[j] * (double)qkv[((size_t)f * N + j) * D3 + 2 * D + h * hd + d];
          out[((size_t)f * N + i) * D + h * hd + d] = (float)o;
        }
      }
  free(p);
  return 0;
}

/* ---- torch.optim.AdamW step as SwavOptimizer drives it (time_tuning.py:413-429): decoupled decay, bias-corrected moments */
typedef struct { float* p; const float* g; float* m; float* v; long long n; float lr; float weight_decay; } tt_cpu_adamw_tensor;
int tt_cpu_adamw_step(const tt_cpu_adamw_tensor* tensors, int count, int step, float beta1, float beta2, float eps, tt_stream_t stream) {
  (void)stream;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  for (int t = 0; t < count; ++t) {
    const tt_cpu_adamw_tensor a = tensors[t];
    for (long long i = 0; i < a.n; ++i) {
      const float g = a.g[i];
      float p = a.p[i] * (1.0f - a.lr * a.weight_decay);
      const float m = a.m[i] + (g - a.m[i]) * (1.0f - beta1);
      const float v = a.v[i] * beta2 + (1.0f - beta2) * g * g;
      const float denom = sqrtf(v) / (float)sqrt(bc2) + eps;
      p -= (a.lr / (float)bc1) * (m / denom);
      a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
  }
  return 0;
}

/* ---- time_tuning.py:113-115: teacher = teacher * (1 - m) + student * m */
int tt_cpu_ema_update(float* teacher, const float* student, long long n, double momentum, tt_stream_t stream) {
  (void)stream;
  const float m = (float)momentum, om = (float)(1.0 - momentum);
  for (long long i = 0; i < n; ++i) teacher[i] = teacher[i] * om + student[i] * m;
  return 0;
}

/* ---- time_tuning.py:250-261: queue[m:] = queue[:-m]; queue[:m] = feats[idx] */
int tt_cpu_queue_push(float* queue, float* scratch, const float* feats, const int64_t* idx, int Q, int D, int m, tt_stream_t stream) {
  (void)stream;
  for (int r = 0; r < Q; ++r)
    for (int c = 0; c < D; ++c)
      scratch[(size_t)r * D + c] = (r < m) ? feats[(size_t)idx[r] * D + c] : queue[(size_t)(r - m) * D + c];
  memcpy(queue, scratch, (size_t)Q * D * sizeof(float));
  return 0;
}

/* ---- features * mask (models.py:142) */
int tt_cpu_scale_rows_inplace(float* x, const float* row_scale, int rows, int cols, tt_stream_t stream) {
  (void)stream;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) x[(size_t)r * cols + c] *= row_scale[r];
  return 0;
}

/* ---- my_utils.sinkhorn(Q, nmb_iters) on the positive matrix itself (my_utils.py:246-274): Q [K][B] (transposed = 0) or [B][K] */
int tt_cpu_sinkhorn_from_q(const float* Qin, int transposed, float* q_out, int B_total, int K, int row0, int rows_out, int iters,
                           void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  const size_t n = (size_t)B_total * K;
  float* s = (float*)malloc(n * sizeof(float));   /* log of the matrix in [B][K] layout: reuse the scores twin with eps = 1 */
  if (!s) return -3;
  for (int b = 0; b < B_total; ++b)
    for (int k = 0; k < K; ++k) s[(size_t)b * K + k] = logf(transposed ? Qin[(size_t)b * K + k] : Qin[(size_t)k * B_total + b]);
  const int rc = tt_cpu_sinkhorn(s, q_out, B_total, K, row0, rows_out, 1.0f, iters, workspace, workspace_bytes, stream);
  free(s);
  return rc;
}

/* ---- fp32 -> bf16 planes (round to nearest even), x = p0 + p1 + p2 */
static uint16_t tt_cpu_bf16(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   /* NaN stays NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float tt_cpu_bf16_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float v;
  memcpy(&v, &u, 4);
  return v;
}
int tt_cpu_split_planes(const float* src, void* dst_planes, long long plane_stride, int planes, long long n, tt_stream_t stream) {
  (void)stream;
  uint16_t* d = (uint16_t*)dst_planes;
  for (long long i = 0; i < n; ++i) {
    float r = src[i];
    for (int p = 0; p < planes; ++p) {
      const uint16_t h = tt_cpu_bf16(r);
      d[p * plane_stride + i] = h;
      r -= tt_cpu_bf16_to_f32(h);
    }
  }
  return 0;
}

/* ---- positions where two fp32 buffers differ bitwise */
int tt_cpu_count_mismatch(const float* a, const float* b, long long n, long long* count_out, tt_stream_t stream) {
  (void)stream;
  long long c = 0;
  for (long long i = 0; i < n; ++i) c += memcmp(a + i, b + i, 4) != 0;
  *count_out = c;
  return 0;
}

/* ---- second batch of round 2: backward of the row ops, the attention backward, the patch embedding, the plane ops ---------- */

/* bias gradient: column sums of a [M][N] matrix */
int tt_cpu_colsum(const float* a, float* out, int M, int N, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  for (int n = 0; n < N; ++n) {
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += a[(size_t)m * N + n];
    out[n] = (float)s;
  }
  return 0;
}

int tt_cpu_add_inplace(float* dst, const float* src, long long n, tt_stream_t stream) {
  (void)stream;
  for (long long i = 0; i < n; ++i) dst[i] += src[i];
  return 0;
}

/* ---- autograd of nn.LayerNorm: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; dgamma = sum dy xhat; dbeta = sum dy.
 *      skip_group = N: dy has no rows for token 0 of each group; those rows of dx are left untouched.  add_to_dx accumulates. */
/* amax_out of the gradient producers (include/timetuning_hip.h): the caller's float is raised to max |.| of what was written */
size_t tt_cpu_amax_slot_bytes(void) { return (size_t)16 * 64 * sizeof(float); }   /* (the twin raises way 0 only) */
static void cpu_amax_raise(float* slot, float v) {
  if (slot && fabsf(v) > *slot) *slot = fabsf(v);
}
int tt_cpu_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                         float* dgamma, float* dbeta, int rows, int D, int add_to_dx, int skip_group, void* workspace,
                         size_t workspace_bytes, float* amax_out, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  double* dg = (double*)calloc((size_t)2 * D, sizeof(double));
  if (!dg) return -3;
  double* db = dg + D;
  for (int r = 0; r < rows; ++r) {
    const long long xr = skip_group ? (long long)(r / (skip_group - 1)) * skip_group + 1 + r % (skip_group - 1) : r;
    const float* xv = x + xr * D;
    const float* dv = dy + (size_t)r * D;
    double m1 = 0.0, m2 = 0.0;
    for (int c = 0; c < D; ++c) {
      const double xh = ((double)xv[c] - mean[r]) * rstd[r], gg = (double)dv[c] * gamma[c];
      m1 += gg; m2 += gg * xh;
      dg[c] += (double)dv[c] * xh; db[c] += dv[c];
    }
    m1 /= D; m2 /= D;
    for (int c = 0; c < D; ++c) {
      const double xh = ((double)xv[c] - mean[r]) * rstd[r], gg = (double)dv[c] * gamma[c];
      const float v = (float)(rstd[r] * (gg - m1 - xh * m2));
      if (add_to_dx) dx[xr * D + c] += v; else dx[xr * D + c] = v;
      cpu_amax_raise(amax_out, dx[xr * D + c]);
    }
  }
  if (dgamma) for (int c = 0; c < D; ++c) { dgamma[c] = (float)dg[c]; dbeta[c] = (float)db[c]; }
  free(dg);
  return 0;
}

/* ---- autograd of F.normalize: dx = (dxn - xn <xn, dxn>) inv_norm */
int tt_cpu_l2norm_bwd(const float* dxn, const float* xn, const float* inv_norm, float* dx, int rows, int D, float* amax_out, tt_stream_t stream) {
  (void)stream;
  for (int r = 0; r < rows; ++r) {
    double dot = 0.0;
    for (int c = 0; c < D; ++c) dot += (double)xn[(size_t)r * D + c] * dxn[(size_t)r * D + c];
    for (int c = 0; c < D; ++c) {
      dx[(size_t)r * D + c] = (float)(((double)dxn[(size_t)r * D + c] - xn[(size_t)r * D + c] * dot) * inv_norm[r]);
      cpu_amax_raise(amax_out, dx[(size_t)r * D + c]);
    }
  }
  return 0;
}

/* ---- autograd of Attention.forward's softmax(q k^T scale) v (dino_vision_transformer.py:122-129) from the saved lse:
 *      P = exp(s - lse); dV = P^T dO; dP = dO V^T; dS = P (dP - rowsum(dO O)); dQ = dS K scale; dK = dS^T Q scale */
int tt_cpu_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N, int H, int hd,
                         float scale, void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  const int D = H * hd, D3 = 3 * D;
  memset(dqkv, 0, (size_t)F * N * D3 * sizeof(float));
  double* acc = (double*)calloc((size_t)N * D3, sizeof(double));   /* one frame's dqkv in double */
  if (!acc) return -3;
  for (int f = 0; f < F; ++f) {
    memset(acc, 0, (size_t)N * D3 * sizeof(double));
    for (int h = 0; h < H; ++h)
      for (int i = 0; i < N; ++i) {
        const float* q = qkv + ((size_t)f * N + i) * D3 + h * hd;
        const float* dO = dout + ((size_t)f * N + i) * D + h * hd;
        const float* O = out + ((size_t)f * N + i) * D + h * hd;
        double delta = 0.0;
        for (int d = 0; d < hd; ++d) delta += (double)dO[d] * O[d];
        for (int j = 0; j < N; ++j) {
          const float* k = qkv + ((size_t)f * N + j) * D3 + D + h * hd;
          const float* v = qkv + ((size_t)f * N + j) * D3 + 2 * D + h * hd;
          double s = 0.0, dp = 0.0;
          for (int d = 0; d < hd; ++d) { s += (double)q[d] * k[d]; dp += (double)dO[d] * v[d]; }
          const double p = exp(s * scale - lse[((size_t)f * H + h) * N + i]);
          const double ds = p * (dp - delta) * scale;
          for (int d = 0; d < hd; ++d) {
            acc[(size_t)i * D3 + h * hd + d] += ds * k[d];
            acc[(size_t)j * D3 + D + h * hd + d] += ds * q[d];
            acc[(size_t)j * D3 + 2 * D + h * hd + d] += p * dO[d];
          }
        }
      }
    for (size_t t = 0; t < (size_t)N * D3; ++t) {
      dqkv[(size_t)f * N * D3 + t] = (float)acc[t];
      cpu_amax_raise(amax_out, dqkv[(size_t)f * N * D3 + t]);
    }
  }
  free(acc);
  return 0;
}

/* (tt_attention_bwd_pairs: fp32-class arithmetic on another MFMA - the twin is the fp32 backward; its range flag is never raised) */
int tt_cpu_attention_bwd_pairs(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N, int H, int hd,
                               float scale, const float* dout_amax, void* workspace, size_t workspace_bytes, int* range_flag, float* amax_out,
                               tt_stream_t stream) {
  (void)range_flag; (void)dout_amax;
  return tt_cpu_attention_bwd(qkv, out, dout, lse, dqkv, F, N, H, hd, scale, workspace, workspace_bytes, amax_out, stream);
}
size_t tt_cpu_attention_bwd_pairs_workspace_bytes(int F, int N, int H, int hd) { (void)hd; return (size_t)F * H * N * sizeof(float) + 1024; }

/* The same with the matrix operands rounded to bf16 where they enter a product (tt_attention_bwd_bf16): q (scaled), k, v, dout, P, dS. */
static float bf16_round(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x;                 /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float r;
  memcpy(&r, &u, 4);
  return r;
}
int tt_cpu_attention_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int F, int N, int H, int hd,
                              float scale, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  const int D = H * hd, D3 = 3 * D;
  double* acc = (double*)calloc((size_t)N * D3, sizeof(double));
  float* qs = (float*)malloc((size_t)hd * 4), *dor = (float*)malloc((size_t)hd * 4);
  if (!acc || !qs || !dor) return -3;
  for (int f = 0; f < F; ++f) {
    memset(acc, 0, (size_t)N * D3 * sizeof(double));
    for (int h = 0; h < H; ++h)
      for (int i = 0; i < N; ++i) {
        const float* q = qkv + ((size_t)f * N + i) * D3 + h * hd;
        const float* dO = dout + ((size_t)f * N + i) * D + h * hd;
        const float* O = out + ((size_t)f * N + i) * D + h * hd;
        double delta = 0.0;
        for (int d = 0; d < hd; ++d) { delta += (double)dO[d] * O[d]; qs[d] = bf16_round(q[d] * scale); dor[d] = bf16_round(dO[d]); }
        for (int j = 0; j < N; ++j) {
          const float* k = qkv + ((size_t)f * N + j) * D3 + D + h * hd;
          const float* v = qkv + ((size_t)f * N + j) * D3 + 2 * D + h * hd;
          double s = 0.0, dp = 0.0;
          for (int d = 0; d < hd; ++d) { s += (double)qs[d] * bf16_round(k[d]); dp += (double)dor[d] * bf16_round(v[d]); }
          const double p = exp(s - lse[((size_t)f * H + h) * N + i]);
          const double pr = bf16_round((float)p), dsr = bf16_round((float)(p * (dp - delta)));
          for (int d = 0; d < hd; ++d) {
            acc[(size_t)i * D3 + h * hd + d] += dsr * bf16_round(k[d]) * scale;
            acc[(size_t)j * D3 + D + h * hd + d] += dsr * qs[d];          /* (q already carries the scale) */
            acc[(size_t)j * D3 + 2 * D + h * hd + d] += pr * dor[d];
          }
        }
      }
    for (size_t t = 0; t < (size_t)N * D3; ++t) dqkv[(size_t)f * N * D3 + t] = (float)acc[t];
  }
  free(acc); free(qs); free(dor);
  return 0;
}

/* ---- PatchEmbed + prepare_tokens (dino_vision_transformer.py:156-171,236-247): conv2d(k = s = P) as a per-patch dot product,
 *      cls token prepended, position embedding added; frame_map selects / reorders source frames (NULL: identity) */
int tt_cpu_patch_embed_fwd(const float* img, const int32_t* frame_map, const float* w, const float* bias, const float* cls, const float* pos,
                           float* tokens, int F, int C, int H, int W, int P, int D, tt_stream_t stream) {
  (void)stream;
  const int gh = H / P, gw = W / P, n = gh * gw;
  for (int f = 0; f < F; ++f) {
    const float* src = img + (size_t)(frame_map ? frame_map[f] : f) * C * H * W;
    float* tok = tokens + (size_t)f * (n + 1) * D;
    for (int d = 0; d < D; ++d) tok[d] = cls[d] + pos[d];
    for (int py = 0; py < gh; ++py)
      for (int px = 0; px < gw; ++px)
        for (int d = 0; d < D; ++d) {
          double s = bias[d];
          for (int c = 0; c < C; ++c)
            for (int y = 0; y < P; ++y)
              for (int x = 0; x < P; ++x)
                s += (double)src[((size_t)c * H + py * P + y) * W + px * P + x] * w[(((size_t)d * C + c) * P + y) * P + x];
          tok[(size_t)(1 + py * gw + px) * D + d] = (float)s + pos[(size_t)(1 + py * gw + px) * D + d];
        }
  }
  return 0;
}

/* ---- StandardScaler.transform: x[r][c] = x[r][c] * scale[c] + shift[c] */
int tt_cpu_affine_cols_inplace(float* x, const float* scale, const float* shift, long long rows, int cols, tt_stream_t stream) {
  (void)stream;
  for (long long r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) x[r * cols + c] = x[r * cols + c] * scale[c] + shift[c];
  return 0;
}

/* ---- the plane ops: transposing conversion, LayerNorm into planes, the plane GEMM (sum over plane pairs i + j <= planes + 1),
 *      and the bf16 attention (fp32 arithmetic on bf16 inputs, P and the output rounded to bf16 as the kernel does) */
int tt_cpu_transpose_planes(const float* src, void* dst, int R, int C, int Rpad, tt_stream_t stream) {
  (void)stream;
  uint16_t* d = (uint16_t*)dst;
  for (int c = 0; c < C; ++c)
    for (int r = 0; r < Rpad; ++r) d[(size_t)c * Rpad + r] = r < R ? tt_cpu_bf16(src[(size_t)r * C + c]) : 0;
  return 0;
}
size_t tt_cpu_transpose_planes_colsum_workspace_bytes(int R, int C, int Rpad) { return 0; }
int tt_cpu_transpose_planes_colsum(const float* src, void* dst, int R, int C, int Rpad, float* colsum, void* workspace, size_t workspace_bytes,
                                   tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes;
  for (int c = 0; c < C; ++c) {
    double s = 0.0;
    for (int r = 0; r < R; ++r) s += src[(size_t)r * C + c];
    colsum[c] = (float)s;
  }
  return tt_cpu_transpose_planes(src, dst, R, C, Rpad, stream);
}
int tt_cpu_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, void* y_planes, long long plane_stride, int planes,
                                float* mean, float* rstd, int rows, int D, float eps, int skip_group, tt_stream_t stream) {
  float* y = (float*)malloc((size_t)rows * D * sizeof(float));
  if (!y) return -3;
  tt_cpu_layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, eps, skip_group, stream);
  const int rc = tt_cpu_split_planes(y, y_planes, plane_stride, planes, (long long)rows * D, stream);
  free(y);
  return rc;
}
int tt_cpu_linear_fwd_planes(const void* x_planes, long long xs, const void* w_planes, long long wsd, int planes, const float* bias,
                             const float* residual, float* y, float* pre_out, void* y_planes, long long ys, int y_nplanes, int M, int N,
                             int K, int act, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes;   /* (the K-split block of the persistent HIP kernels: ABI 7) */
  const uint16_t* xp = (const uint16_t*)x_planes;
  const uint16_t* wp = (const uint16_t*)w_planes;
  float* tmp = (float*)malloc((size_t)M * N * sizeof(float));
  if (!tmp) return -3;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = bias ? (double)bias[n] : 0.0;
      for (int pa = 0; pa < planes; ++pa)
        for (int pw = 0; pa + pw < planes; ++pw)
          for (int k = 0; k < K; ++k)
            s += (double)tt_cpu_bf16_to_f32(xp[pa * xs + (size_t)m * K + k]) * (double)tt_cpu_bf16_to_f32(wp[pw * wsd + (size_t)n * K + k]);
      float v = (float)s;
      if (pre_out) pre_out[(size_t)m * N + n] = v;
      if (act == 1) v = tt_cpu_gelu(v);
      if (residual) v += residual[(size_t)m * N + n];
      tmp[(size_t)m * N + n] = v;
    }
  if (y_planes) tt_cpu_split_planes(tmp, y_planes, ys, y_nplanes, (long long)M * N, stream);
  if (y) memcpy(y, tmp, (size_t)M * N * sizeof(float));
  free(tmp);
  return 0;
}
int tt_cpu_attention_fwd_bf16(const void* qkv, void* out, int F, int N, int H, int head_dim, float scale, tt_stream_t stream) {
  (void)stream;
  const uint16_t* in = (const uint16_t*)qkv;
  uint16_t* o = (uint16_t*)out;
  const int D = H * head_dim, D3 = 3 * D;
  double* p = (double*)malloc((size_t)N * sizeof(double));
  if (!p) return -3;
  for (int f = 0; f < F; ++f)
    for (int h = 0; h < H; ++h)
      for (int i = 0; i < N; ++i) {
        const uint16_t* q = in + ((size_t)f * N + i) * D3 + h * head_dim;
        double mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
          const uint16_t* k = in + ((size_t)f * N + j) * D3 + D + h * head_dim;
          double s = 0.0;
          for (int d = 0; d < head_dim; ++d) s += (double)tt_cpu_bf16_to_f32(q[d]) * tt_cpu_bf16_to_f32(k[d]);
          p[j] = s * scale;
          if (p[j] > mx) mx = p[j];
        }
        double sum = 0.0;
        for (int j = 0; j < N; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; p[j] = tt_cpu_bf16_to_f32(tt_cpu_bf16((float)p[j])); }
        for (int d = 0; d < head_dim; ++d) {
          double acc = 0.0;
          for (int j = 0; j < N; ++j) acc += p[j] * tt_cpu_bf16_to_f32(in[((size_t)f * N + j) * D3 + 2 * D + h * head_dim + d]);
          o[((size_t)f * N + i) * D + h * head_dim + d] = tt_cpu_bf16((float)(acc / sum));
        }
      }
  free(p);
  return 0;
}

/* ---- PatchEmbed on bf16 operands (tt_patch_embed_fwd_planes): patches and weight rounded to bf16 (the weight arrives as its plane), double
 *      accumulation, fp32 tokens (dino_vision_transformer.py:166-171,236-247 under autocast) */
static float bf16_bits_to_float(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static float round_to_bf16(float x) {
  uint16_t b;
  tt_cpu_split_planes(&x, &b, 1, 1, 1, NULL);
  return bf16_bits_to_float(b);
}
size_t tt_cpu_patch_embed_planes_workspace_bytes(int F, int C, int H, int W, int P) { return 0; }
int tt_cpu_patch_embed_fwd_planes(const float* img, const int32_t* frame_map, const void* w_planes, const float* bias, const float* cls,
                                  const float* pos, float* tokens, int F, int C, int H, int W, int P, int D, void* workspace,
                                  size_t workspace_bytes, tt_stream_t stream) {
  (void)stream; (void)workspace; (void)workspace_bytes;
  const uint16_t* wp = (const uint16_t*)w_planes;
  const int gh = H / P, gw = W / P, n = gh * gw, K = C * P * P;
  float* patch = (float*)malloc((size_t)K * 4);
  if (!patch) return -3;
  for (int f = 0; f < F; ++f) {
    const float* src = img + (size_t)(frame_map ? frame_map[f] : f) * C * H * W;
    float* tok = tokens + (size_t)f * (n + 1) * D;
    for (int d = 0; d < D; ++d) tok[d] = (cls[d] + pos[d] - bias[d]) + bias[d];   /* the zero row meets the bias in the GEMM's epilogue */
    for (int py = 0; py < gh; ++py)
      for (int px = 0; px < gw; ++px) {
        for (int c = 0; c < C; ++c)
          for (int y = 0; y < P; ++y)
            for (int x = 0; x < P; ++x) patch[(c * P + y) * P + x] = round_to_bf16(src[((size_t)c * H + py * P + y) * W + px * P + x]);
        for (int d = 0; d < D; ++d) {
          double s = 0.0;
          for (int k = 0; k < K; ++k) s += (double)patch[k] * bf16_bits_to_float(wp[(size_t)d * K + k]);
          tok[(size_t)(1 + py * gw + px) * D + d] = ((float)s + bias[d]) + pos[(size_t)(1 + py * gw + px) * D + d];
        }
      }
  }
  free(patch);
  return 0;
}

/* ---- the coarse entry points (include/timetuning_hip.h, "Coarse entry points"): the same sequences over the twins above.
 *      Scratch is malloc'ed here (the workspace arguments are ignored, the *_workspace_bytes twins return 0). */
typedef struct {
  const float *norm1_w, *norm1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  const void *qkv_wp, *proj_wp, *fc1_wp, *fc2_wp;
} tt_cpu_vit_block_params;
typedef struct {
  const float *patch_w, *patch_b, *cls, *pos;
  const tt_cpu_vit_block_params* blocks;
  int n_blocks;
  const float *norm_w, *norm_b;
  int dim, heads, hidden, patch;
  int planes;
  const void* patch_wp;
  int* range_flag;
  int precision;
} tt_cpu_vit_params;
typedef struct { const float* w; const float* b; int out_features, in_features; } tt_cpu_linear_params;

/* ---- fp16 PAIRS (round 4, the "f16x3" mode): x -> hi = fp16(x), lo = fp16((x - hi) * 2^11), groups of 32 elements as [hi x 32][lo x 32].
 *      gcc 11 has no _Float16 on x86-64: round-to-nearest-even conversions by hand (subnormals kept, overflow to infinity). */
static uint16_t tt_cpu_f16(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);              /* NaN */
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);             /* >= 65520 (incl. inf): rounds to infinity */
  if (a < 0x33000001u) return sign;                                    /* <= 2^-25: rounds to zero (a tie at 2^-25 goes to even = 0) */
  const int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;                            /* 24-bit significand */
  int shift = e < -14 ? 13 + (-14 - e) : 13;                           /* bits to drop: 13 for normals, more for subnormals */
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
  uint32_t h;
  if (e < -14) h = q;                                                  /* subnormal (a carry into 0x400 is the smallest normal) */
  else h = ((uint32_t)(e + 15) << 10) + (q - 0x400u);                  /* a carry out of the significand bumps the exponent */
  return (uint16_t)(sign | h);
}
static float tt_cpu_f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float v;
  if (e == 0) v = ldexpf((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexpf((float)(m | 0x400u), (int)e - 25);
  return sign ? -v : v;
}
static long long pair_index(long long i) { return ((i >> 5) << 6) + (i & 31); }
static void pair_put(uint16_t* d, long long i, float v) {
  const uint16_t hi = tt_cpu_f16(v);
  d[pair_index(i)] = hi;
  d[pair_index(i) + 32] = tt_cpu_f16((v - tt_cpu_f16_to_f32(hi)) * 2048.0f);
}
static double pair_hi(const uint16_t* d, long long i) { return (double)tt_cpu_f16_to_f32(d[pair_index(i)]); }
static double pair_lo(const uint16_t* d, long long i) { return (double)tt_cpu_f16_to_f32(d[pair_index(i) + 32]) / 2048.0; }
/* the three products the kernels form per term: hi hi + hi lo + lo hi (lo lo is dropped) */
static double pair_dot(const uint16_t* a, const uint16_t* b, long long n) {
  double s = 0.0;
  for (long long k = 0; k < n; ++k) s += pair_hi(a, k) * pair_hi(b, k) + pair_hi(a, k) * pair_lo(b, k) + pair_lo(a, k) * pair_hi(b, k);
  return s;
}
/* the range flag (ABI 7): set when a value that is split lies beyond fp16's range (|x| > 65504 rounds to inf) or is not finite */
static void pair_range_check(const float* v, long long n, int* range_flag) {
  if (!range_flag) return;
  for (long long i = 0; i < n; ++i)
    if (!(fabsf(v[i]) < 65520.0f)) { *range_flag = 1; return; }   /* 65520 = the round-to-nearest-even boundary between 65504 and inf */
}
int tt_cpu_split_pairs(const float* src, void* dst_pairs, long long n, int* range_flag, tt_stream_t stream) {
  (void)stream;
  pair_range_check(src, n, range_flag);
  for (long long i = 0; i < n; ++i) pair_put((uint16_t*)dst_pairs, i, src[i]);
  return 0;
}
int tt_cpu_join_pairs(const void* src_pairs, float* dst, long long n, tt_stream_t stream) {
  (void)stream;
  const uint16_t* s = (const uint16_t*)src_pairs;
  for (long long i = 0; i < n; ++i) dst[i] = fmaf(tt_cpu_f16_to_f32(s[pair_index(i) + 32]), 0.00048828125f, tt_cpu_f16_to_f32(s[pair_index(i)]));
  return 0;
}
int tt_cpu_layernorm_fwd_pairs(const float* x, const float* gamma, const float* beta, void* y_pairs, float* mean, float* rstd, int rows, int D,
                               float eps, int skip_group, int* range_flag, tt_stream_t stream) {
  float* y = (float*)malloc((size_t)rows * D * sizeof(float));
  if (!y) return -3;
  tt_cpu_layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, eps, skip_group, stream);
  const int rc = tt_cpu_split_pairs(y, y_pairs, (long long)rows * D, range_flag, stream);
  free(y);
  return rc;
}
int tt_cpu_linear_fwd_pairs(const void* x_pairs, const void* w_pairs, const float* bias, const float* residual, float* y, float* pre_out,
                            void* y_pairs, int M, int N, int K, int act, void* workspace, size_t workspace_bytes, int* range_flag,
                            tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes;
  const uint16_t *xp = (const uint16_t*)x_pairs, *wp = (const uint16_t*)w_pairs;   /* [M][2 K], [N][2 K] */
  float* tmp = (float*)malloc((size_t)M * N * sizeof(float));
  if (!tmp) return -3;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float v = (float)((bias ? (double)bias[n] : 0.0) + pair_dot(xp + (size_t)m * 2 * K, wp + (size_t)n * 2 * K, K));
      if (pre_out) pre_out[(size_t)m * N + n] = v;
      if (act == 1) v = tt_cpu_gelu(v);
      if (residual) v += residual[(size_t)m * N + n];
      tmp[(size_t)m * N + n] = v;
    }
  if (y_pairs) tt_cpu_split_pairs(tmp, y_pairs, (long long)M * N, range_flag, stream);
  if (y) memcpy(y, tmp, (size_t)M * N * sizeof(float));
  free(tmp);
  return 0;
}
int tt_cpu_linear_fwd_pairs_route(int M, int N, int K, int act, int has_bias, int has_residual, int has_y, int has_y_pairs, int has_pre_out) {
  return 0;
}
/* softmax(q k^T * scale) v on pair operands: scores and P V as the three-product sums, fp64 accumulation */
int tt_cpu_attention_fwd_pairs(const void* qkv_pairs, void* out_pairs, float* out_f32, float* lse, int F, int N, int H, int head_dim, float scale,
                               tt_stream_t stream) {
  const uint16_t* in = (const uint16_t*)qkv_pairs;
  const int D = H * head_dim;
  const size_t RS = (size_t)6 * D;
  double* p = (double*)malloc((size_t)N * sizeof(double));
  float* o = (float*)malloc((size_t)F * N * D * sizeof(float));
  if (!p || !o) return -3;
  for (int f = 0; f < F; ++f)
    for (int h = 0; h < H; ++h)
      for (int i = 0; i < N; ++i) {
        const uint16_t* q = in + ((size_t)f * N + i) * RS + (size_t)h * 2 * head_dim;
        double mx = -1e300;
        for (int j = 0; j < N; ++j) {
          const uint16_t* k = in + ((size_t)f * N + j) * RS + 2 * D + (size_t)h * 2 * head_dim;
          p[j] = pair_dot(q, k, head_dim) * (double)scale;
          if (p[j] > mx) mx = p[j];
        }
        double sum = 0.0;
        for (int j = 0; j < N; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; }
        if (lse) lse[((size_t)f * H + h) * N + i] = (float)(mx + log(sum));
        for (int d = 0; d < head_dim; ++d) {
          double acc = 0.0;
          for (int j = 0; j < N; ++j) {
            /* the kernel splits the unnormalised p in (0, 1] into (hi, lo) as well */
            const float pj = (float)p[j];
            const float ph = tt_cpu_f16_to_f32(tt_cpu_f16(pj));
            const double plo = (double)tt_cpu_f16_to_f32(tt_cpu_f16((pj - ph) * 2048.0f)) / 2048.0;
            const uint16_t* v = in + ((size_t)f * N + j) * RS + 4 * D + (size_t)h * 2 * head_dim;
            acc += (double)ph * pair_hi(v, d) + (double)ph * pair_lo(v, d) + plo * pair_hi(v, d);
          }
          o[((size_t)f * N + i) * D + h * head_dim + d] = (float)(acc / sum);
        }
      }
  if (out_f32) memcpy(out_f32, o, (size_t)F * N * D * sizeof(float));
  if (out_pairs) tt_cpu_split_pairs(o, out_pairs, (long long)F * N * D, NULL, stream);
  free(p); free(o);
  return 0;
}
size_t tt_cpu_split_pairs_dual_workspace_bytes(int R, int C, int Rpad) { return 0; }
/* the power of two that brings amax into [2^13, 2^14) (gemm_planes.hip pair_scale_of); 1 for 0 / inf / nan */
static float pair_scale_of(float amax) {
  if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
  int e = 13 - ilogbf(amax);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, e);
}
int tt_cpu_split_pairs_dual(const float* src, void* dst_t_pairs, void* dst_row_pairs, float* colsum, float* scale_out, int R, int C, int Rpad,
                            void* workspace, size_t workspace_bytes, int* range_flag, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  uint16_t *t = (uint16_t*)dst_t_pairs, *row = (uint16_t*)dst_row_pairs;
  float S = 1.0f;
  if (scale_out) {   /* a gradient: scaled by a power of two before the split, the consumers divide by it */
    float amax = 0.f;
    for (size_t i = 0; i < (size_t)R * C; ++i) amax = fmaxf(amax, fabsf(src[i]));
    S = *scale_out = pair_scale_of(amax);
  }
  for (int c = 0; c < C; ++c) {
    double s = 0.0;
    for (int r = 0; r < Rpad; ++r) {
      const float v = r < R ? src[(size_t)r * C + c] : 0.f, vs = v * S;
      pair_range_check(&vs, 1, range_flag);
      if (t) pair_put(t + (size_t)c * 2 * Rpad, r, v * S);
      if (row && r < R) pair_put(row + (size_t)r * 2 * C, c, v * S);
      s += v;
    }
    if (colsum) colsum[c] = (float)s;
  }
  return 0;
}
/* the column sums left as partials of 64-row blocks [ceil(Rpad / 64)][C] (tt_split_pairs_dual_parts) */
int tt_cpu_split_pairs_dual_parts(const float* src, void* dst_t_pairs, void* dst_row_pairs, float* colsum_parts, float* scale_out,
                                  const float* amax_in, int R, int C, int Rpad, void* workspace, size_t workspace_bytes, int* range_flag,
                                  tt_stream_t stream) {
  /* (amax_in: max |src| as its producer left it - equal to the maximum this twin finds itself, which is what it uses) */
  (void)amax_in;
  const int rc = tt_cpu_split_pairs_dual(src, dst_t_pairs, dst_row_pairs, NULL, scale_out, R, C, Rpad, workspace, workspace_bytes, range_flag, stream);
  if (rc) return rc;
  const int chunks = (Rpad + 63) / 64;
  for (int b = 0; b < chunks; ++b)
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int r = b * 64; r < (b + 1) * 64 && r < R; ++r) s += src[(size_t)r * C + c];
      colsum_parts[(size_t)b * C + c] = (float)s;
    }
  return 0;
}
int tt_cpu_split_pairs_dual_multi(const float* const* src, void* const* dst_t_pairs, void* const* dst_row_pairs, const int* R, const int* C,
                                  const int* Rpad, int n, int* range_flag, tt_stream_t stream) {
  for (int i = 0; i < n; ++i) {
    const int rc = tt_cpu_split_pairs_dual(src[i], dst_t_pairs[i], dst_row_pairs[i], NULL, NULL, R[i], C[i], Rpad[i], NULL, 0, range_flag, stream);
    if (rc) return rc;
  }
  return 0;
}
int tt_cpu_transpose_pairs(const void* src_pairs, void* dst_t_pairs, int R, int C, int Rpad, tt_stream_t stream) {
  (void)stream;
  const uint16_t* s = (const uint16_t*)src_pairs;
  uint16_t* t = (uint16_t*)dst_t_pairs;
  for (int c = 0; c < C; ++c)
    for (int r = 0; r < Rpad; ++r) {
      t[(size_t)c * 2 * Rpad + pair_index(r)] = r < R ? s[(size_t)r * 2 * C + pair_index(c)] : 0;
      t[(size_t)c * 2 * Rpad + pair_index(r) + 32] = r < R ? s[(size_t)r * 2 * C + pair_index(c) + 32] : 0;
    }
  return 0;
}
int tt_cpu_linear_bwd_data_pairs(const void* dy_pairs, const void* wT_pairs, const float* gelu_pre, float* dx, const float* dy_scale, int M, int N,
                                 int K, void* workspace, size_t workspace_bytes, float* amax_out, tt_stream_t stream) {
  (void)stream; (void)workspace; (void)workspace_bytes;
  const uint16_t *dy = (const uint16_t*)dy_pairs, *wT = (const uint16_t*)wT_pairs;   /* dy [M][2 N], wT [K][2 N] */
  const double inv_s = dy_scale ? 1.0 / (double)*dy_scale : 1.0;
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) {
      float v = (float)(pair_dot(dy + (size_t)m * 2 * N, wT + (size_t)k * 2 * N, N) * inv_s);
      if (gelu_pre) v *= tt_cpu_gelu_grad(gelu_pre[(size_t)m * K + k]);
      dx[(size_t)m * K + k] = v;
      cpu_amax_raise(amax_out, v);
    }
  return 0;
}
size_t tt_cpu_linear_bwd_weight_pairs_workspace_bytes(int N, int K, int Mpad) { return 0; }
int tt_cpu_linear_bwd_weight_pairs(const void* dyT_pairs, const void* xT_pairs, float* dw, const float* dy_scale, int N, int K, int Mpad,
                                   void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  const uint16_t *dyT = (const uint16_t*)dyT_pairs, *xT = (const uint16_t*)xT_pairs;   /* dyT [N][2 Mpad], xT [K][2 Mpad] */
  const double inv_s = dy_scale ? 1.0 / (double)*dy_scale : 1.0;
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) dw[(size_t)n * K + k] = (float)(pair_dot(dyT + (size_t)n * 2 * Mpad, xT + (size_t)k * 2 * Mpad, Mpad) * inv_s);
  return 0;
}

/* gemm_pairs_tn.hip: the same product from row pairs dy [M][2 N], x [M][2 K] */
int tt_cpu_linear_bwd_weight_pairs_tn_ok(int N, int K, int M) { return N > 0 && K > 0 && M > 0 && N % 128 == 0 && K % 128 == 0; }
size_t tt_cpu_linear_bwd_weight_pairs_tn_workspace_bytes(int N, int K, int M) { return 0; }
int tt_cpu_linear_bwd_weight_pairs_tn(const void* dy_pairs, const void* x_pairs, float* dw, const float* dy_scale, int N, int K, int M,
                                      void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  const uint16_t *dy = (const uint16_t*)dy_pairs, *x = (const uint16_t*)x_pairs;
  const double inv_s = dy_scale ? 1.0 / (double)*dy_scale : 1.0;
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int m = 0; m < M; ++m) {
        const uint16_t *a = dy + (size_t)m * 2 * N, *b = x + (size_t)m * 2 * K;
        s += pair_hi(a, n) * pair_hi(b, k) + pair_hi(a, n) * pair_lo(b, k) + pair_lo(a, n) * pair_hi(b, k);
      }
      dw[(size_t)n * K + k] = (float)(s * inv_s);
    }
  return 0;
}

int tt_cpu_linear_bwd_weight_pairs_tn_bias(const void* dy_pairs, const void* x_pairs, float* dw, const float* dy_scale, int N, int K, int M,
                                           void* workspace, size_t workspace_bytes, const float* colsum_parts, int colsum_count, float* db,
                                           tt_stream_t stream) {
  for (int n = 0; n < N; ++n) {
    double s = 0.0;
    for (int b = 0; b < colsum_count; ++b) s += colsum_parts[(size_t)b * N + n];
    db[n] = (float)s;
  }
  return tt_cpu_linear_bwd_weight_pairs_tn(dy_pairs, x_pairs, dw, dy_scale, N, K, M, workspace, workspace_bytes, stream);
}

/* prepare_tokens on pair operands (tt_patch_embed_fwd_pairs): patches and weight as (hi, lo) pairs, three products per term */
size_t tt_cpu_patch_embed_pairs_workspace_bytes(int F, int C, int H, int W, int P) { return 0; }
int tt_cpu_patch_embed_fwd_pairs(const float* img, const int32_t* frame_map, const void* w_pairs, const float* bias, const float* cls,
                                 const float* pos, float* tokens, int F, int C, int H, int W, int P, int D, void* workspace,
                                 size_t workspace_bytes, int* range_flag, tt_stream_t stream) {
  (void)stream; (void)workspace; (void)workspace_bytes;
  const uint16_t* wp = (const uint16_t*)w_pairs;
  const int gh = H / P, gw = W / P, n = gh * gw, K = C * P * P;
  uint16_t* patch = (uint16_t*)malloc((size_t)K * 4);
  if (!patch) return -3;
  for (int f = 0; f < F; ++f) {
    const float* src = img + (size_t)(frame_map ? frame_map[f] : f) * C * H * W;
    float* tok = tokens + (size_t)f * (n + 1) * D;
    for (int d = 0; d < D; ++d) tok[d] = (cls[d] + pos[d] - bias[d]) + bias[d];   /* the zero row meets the bias in the GEMM's epilogue */
    for (int py = 0; py < gh; ++py)
      for (int px = 0; px < gw; ++px) {
        for (int c = 0; c < C; ++c)
          for (int y = 0; y < P; ++y)
            for (int x = 0; x < P; ++x) {
              pair_range_check(&src[((size_t)c * H + py * P + y) * W + px * P + x], 1, range_flag);
              pair_put(patch, (c * P + y) * P + x, src[((size_t)c * H + py * P + y) * W + px * P + x]);
            }
        for (int d = 0; d < D; ++d)
          tok[(size_t)(1 + py * gw + px) * D + d] =
              ((float)pair_dot(patch, wp + (size_t)d * 2 * K, K) + bias[d]) + pos[(size_t)(1 + py * gw + px) * D + d];
      }
  }
  free(patch);
  return 0;
}

size_t tt_cpu_vit_forward_workspace_bytes(int F, int N, int D, int hidden, int planes) { return 0; }
size_t tt_cpu_mlp_head_forward_workspace_bytes(int M, const tt_cpu_linear_params* layers, int n_layers) { return 0; }
size_t tt_cpu_scores_sinkhorn_workspace_bytes(int B, int queue_rows, int K, int dim) { return 0; }

/* dino_vision_transformer.py:236-252 (prepare_tokens, blocks), :135-153 (Block), :265-273 (final norm) */
int tt_cpu_vit_forward(const tt_cpu_vit_params* p, const float* img, const int32_t* frame_map, int F, int C, int H, int W, float* tokens,
                       float* normed, int drop_cls, float* last_qkv, float* last_probs, void* workspace, size_t workspace_bytes,
                       tt_stream_t stream) {
  const int D = p->dim, hd = D / p->heads, P = p->planes, Hd = p->hidden;
  const int N = 1 + (H / p->patch) * (W / p->patch);
  const size_t M = (size_t)F * N;
  const float scale = 1.0f / sqrtf((float)hd);
  if (img) {
    /* (C P P <= 9 D: the HIP side keeps the im2col rows in the attention phase of its scratch) */
    if (P == 1 && p->patch_wp && p->patch % 4 == 0 && W % 4 == 0 && (C * p->patch * p->patch) % 64 == 0 && D % 64 == 0 && C * p->patch * p->patch <= 9 * D)
      tt_cpu_patch_embed_fwd_planes(img, frame_map, p->patch_wp, p->patch_b, p->cls, p->pos, tokens, F, C, H, W, p->patch, D, NULL, 0, stream);
    else if (P == 2 && p->patch_wp && p->patch % 4 == 0 && W % 4 == 0 && (C * p->patch * p->patch) % 32 == 0 && D % 64 == 0 && C * p->patch * p->patch <= 3 * D)
      tt_cpu_patch_embed_fwd_pairs(img, frame_map, p->patch_wp, p->patch_b, p->cls, p->pos, tokens, F, C, H, W, p->patch, D, NULL, 0, p->range_flag, stream);
    else
      tt_cpu_patch_embed_fwd(img, frame_map, p->patch_w, p->patch_b, p->cls, p->pos, tokens, F, C, H, W, p->patch, D, stream);
  }
  float* h = (float*)malloc(M * D * 4), *qkv_own = (float*)malloc(M * 3 * D * 4), *att = (float*)malloc(M * D * 4);
  float* act = (float*)malloc(M * Hd * 4);
  const int PP = P > 0 ? P : 1;
  uint16_t* hp = (uint16_t*)malloc(PP * M * D * 2), *qkvb = (uint16_t*)malloc(M * 3 * D * 2), *attp = (uint16_t*)malloc(PP * M * D * 2);
  uint16_t* actp = (uint16_t*)malloc(PP * M * Hd * 2);
  if (!h || !qkv_own || !att || !act || !hp || !qkvb || !attp || !actp) return -3;
  for (int i = 0; i < p->n_blocks; ++i) {
    const tt_cpu_vit_block_params* b = &p->blocks[i];
    const int last = i == p->n_blocks - 1;
    float* probs = last ? last_probs : NULL;
    float* qkv = (last && last_qkv) ? last_qkv : qkv_own;
    if (P == 0) {
      tt_cpu_layernorm_fwd(tokens, b->norm1_w, b->norm1_b, h, NULL, NULL, (int)M, D, 1e-6f, 0, stream);
      tt_cpu_linear_fwd(h, b->qkv_w, b->qkv_b, NULL, qkv, NULL, (int)M, 3 * D, D, 0, 0, stream);
      tt_cpu_attention_fwd(qkv, att, NULL, probs, F, N, p->heads, hd, scale, stream);
      tt_cpu_linear_fwd(att, b->proj_w, b->proj_b, tokens, tokens, NULL, (int)M, D, D, 0, 0, stream);
      tt_cpu_layernorm_fwd(tokens, b->norm2_w, b->norm2_b, h, NULL, NULL, (int)M, D, 1e-6f, 0, stream);
      tt_cpu_linear_fwd(h, b->fc1_w, b->fc1_b, NULL, act, NULL, (int)M, Hd, D, 1, 0, stream);
      tt_cpu_linear_fwd(act, b->fc2_w, b->fc2_b, tokens, tokens, NULL, (int)M, D, Hd, 0, 0, stream);
      continue;
    }
    const long long MD = (long long)M * D;
    if (P == 2) {   /* fp16 pairs (4 bytes per element: the PP = 2 buffers above are the right size) */
      tt_cpu_layernorm_fwd_pairs(tokens, b->norm1_w, b->norm1_b, hp, NULL, NULL, (int)M, D, 1e-6f, 0, p->range_flag, stream);
      if (!(last && last_qkv) && !probs && hd == 64) {
        uint16_t* qkvp = (uint16_t*)qkv_own;   /* pairs [M][2 x 3 D] = the bytes of the fp32 qkv buffer */
        tt_cpu_linear_fwd_pairs(hp, b->qkv_wp, b->qkv_b, NULL, NULL, NULL, qkvp, (int)M, 3 * D, D, 0, NULL, 0, p->range_flag, stream);
        tt_cpu_attention_fwd_pairs(qkvp, attp, NULL, NULL, F, N, p->heads, hd, scale, stream);
      } else {
        tt_cpu_linear_fwd_pairs(hp, b->qkv_wp, b->qkv_b, NULL, qkv, NULL, NULL, (int)M, 3 * D, D, 0, NULL, 0, p->range_flag, stream);
        tt_cpu_attention_fwd(qkv, att, NULL, probs, F, N, p->heads, hd, scale, stream);
        tt_cpu_split_pairs(att, attp, MD, p->range_flag, stream);
      }
      tt_cpu_linear_fwd_pairs(attp, b->proj_wp, b->proj_b, tokens, tokens, NULL, NULL, (int)M, D, D, 0, NULL, 0, p->range_flag, stream);
      tt_cpu_layernorm_fwd_pairs(tokens, b->norm2_w, b->norm2_b, hp, NULL, NULL, (int)M, D, 1e-6f, 0, p->range_flag, stream);
      tt_cpu_linear_fwd_pairs(hp, b->fc1_wp, b->fc1_b, NULL, NULL, NULL, actp, (int)M, Hd, D, 1, NULL, 0, p->range_flag, stream);
      tt_cpu_linear_fwd_pairs(actp, b->fc2_wp, b->fc2_b, tokens, tokens, NULL, NULL, (int)M, D, Hd, 0, NULL, 0, p->range_flag, stream);
      continue;
    }
    tt_cpu_layernorm_fwd_planes(tokens, b->norm1_w, b->norm1_b, hp, MD, P, NULL, NULL, (int)M, D, 1e-6f, 0, stream);
    const void* proj_in;
    if (P == 1 && !(last && last_qkv) && !probs && N <= 256 && hd == 64) {
      tt_cpu_linear_fwd_planes(hp, MD, b->qkv_wp, 3ll * D * D, 1, b->qkv_b, NULL, NULL, NULL, qkvb, (long long)M * 3 * D, 1, (int)M, 3 * D, D, 0, NULL, 0,
                               stream);
      tt_cpu_attention_fwd_bf16(qkvb, attp, F, N, p->heads, hd, scale, stream);
    } else {
      tt_cpu_linear_fwd_planes(hp, MD, b->qkv_wp, 3ll * D * D, P, b->qkv_b, NULL, qkv, NULL, NULL, 0, 0, (int)M, 3 * D, D, 0, NULL, 0, stream);
      tt_cpu_attention_fwd(qkv, att, NULL, probs, F, N, p->heads, hd, scale, stream);
      tt_cpu_split_planes(att, attp, MD, P, MD, stream);
    }
    proj_in = attp;
    tt_cpu_linear_fwd_planes(proj_in, MD, b->proj_wp, (long long)D * D, P, b->proj_b, tokens, tokens, NULL, NULL, 0, 0, (int)M, D, D, 0, NULL, 0, stream);
    tt_cpu_layernorm_fwd_planes(tokens, b->norm2_w, b->norm2_b, hp, MD, P, NULL, NULL, (int)M, D, 1e-6f, 0, stream);
    tt_cpu_linear_fwd_planes(hp, MD, b->fc1_wp, (long long)Hd * D, P, b->fc1_b, NULL, NULL, NULL, actp, (long long)M * Hd, P, (int)M, Hd, D, 1, NULL, 0,
                             stream);
    tt_cpu_linear_fwd_planes(actp, (long long)M * Hd, b->fc2_wp, (long long)Hd * D, P, b->fc2_b, tokens, tokens, NULL, NULL, 0, 0, (int)M, D, Hd,
                             0, NULL, 0, stream);
  }
  free(h); free(qkv_own); free(att); free(act); free(hp); free(qkvb); free(attp); free(actp);
  if (normed) {
    if (drop_cls) tt_cpu_layernorm_fwd(tokens, p->norm_w, p->norm_b, normed, NULL, NULL, F * (N - 1), D, 1e-6f, N, stream);
    else tt_cpu_layernorm_fwd(tokens, p->norm_w, p->norm_b, normed, NULL, NULL, (int)M, D, 1e-6f, 0, stream);
  }
  return 0;
}

/* models.py:915-926,1075-1077: Linear (GELU Linear)* */
int tt_cpu_mlp_head_forward(const float* x, int M, const tt_cpu_linear_params* layers, int n_layers, float* out, int precision, void* workspace,
                            size_t workspace_bytes, tt_stream_t stream) {
  (void)precision;
  int width = 1;
  for (int i = 0; i < n_layers; ++i) width = layers[i].out_features > width ? layers[i].out_features : width;
  float* buf[2] = {(float*)malloc((size_t)M * width * 4), (float*)malloc((size_t)M * width * 4)};
  if (!buf[0] || !buf[1]) return -3;
  const float* cur = x;
  for (int i = 0; i < n_layers; ++i) {
    const int last = i == n_layers - 1;
    float* dst = last ? out : buf[i & 1];
    tt_cpu_linear_fwd(cur, layers[i].w, layers[i].b, NULL, dst, NULL, M, layers[i].out_features, layers[i].in_features, last ? 0 : 1, 0, stream);
    cur = dst;
  }
  free(buf[0]); free(buf[1]);
  return 0;
}

/* time_tuning.py:195-217 (get_scores) on one rank: normalised batch and queue rows against the prototypes, one assignment */
int tt_cpu_scores_sinkhorn(const float* z, int B, const float* queue, int queue_rows, const float* prototypes, int K, int dim, float* scores,
                           float* q_out, int rows_out, float eps, int iters, int precision, void* workspace, size_t workspace_bytes,
                           tt_stream_t stream) {
  (void)precision;
  if (!queue) queue_rows = 0;
  const int total = B + queue_rows;
  float* zn = (float*)malloc((size_t)total * dim * 4);
  if (!zn) return -3;
  tt_cpu_l2norm_fwd(z, dim, zn, NULL, B, dim, stream);
  tt_cpu_linear_fwd(zn, prototypes, NULL, NULL, scores, NULL, B, K, dim, 0, 0, stream);
  if (queue_rows) {
    tt_cpu_l2norm_fwd(queue, dim, zn + (size_t)B * dim, NULL, queue_rows, dim, stream);
    tt_cpu_linear_fwd(zn + (size_t)B * dim, prototypes, NULL, NULL, scores + (size_t)B * K, NULL, queue_rows, K, dim, 0, 0, stream);
  }
  free(zn);
  return tt_cpu_sinkhorn(scores, q_out, total, K, 0, rows_out, eps, iters, NULL, 0, stream);
}

/* time_tuning.py:659-663: optimizer.step(), normalize_prototypes() (:124-128), update_momentum_teacher() (:109-122) */
int tt_cpu_adamw_ema_step(const tt_cpu_adamw_tensor* tensors, int count, int step, float beta1, float beta2, float eps, float* prototypes, int K,
                          int dim, float* teacher_flat, const float* student_flat, long long n_flat, float* teacher_prototypes, double momentum,
                          tt_stream_t stream) {
  tt_cpu_adamw_step(tensors, count, step, beta1, beta2, eps, stream);
  if (prototypes) tt_cpu_normalize_rows_inplace(prototypes, K, dim, stream);
  if (n_flat > 0) tt_cpu_ema_update(teacher_flat, student_flat, n_flat, momentum, stream);
  if (teacher_prototypes) {
    tt_cpu_ema_update(teacher_prototypes, prototypes, (long long)K * dim, momentum, stream);
    tt_cpu_normalize_rows_inplace(teacher_prototypes, K, dim, stream);
  }
  return 0;
}

/* ====================================================================================================================
 * Third batch (round 2): the ops that had their restatement in oracle/timet_oracle.py only.  With these every compute entry
 * point of include/timetuning_hip.h has a plain-C twin.
 * ==================================================================================================================== */

/* g[i] *= *scale for every table entry: loss.backward()'s chain rule on the fused step's gradients (time_tuning.py:420-423) */
int tt_cpu_scale_tensors(const tt_cpu_adamw_tensor* tensors, int count, const float* scale, tt_stream_t stream) {
  for (int t = 0; t < count; ++t) {
    float* g = (float*)tensors[t].g;
    for (long long i = 0; i < tensors[t].n; ++i) g[i] *= *scale;
  }
  return 0;
}

/* C = alpha op(A) op(B), optionally batched; the generic product behind the op sites (double accumulation) */
int tt_cpu_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_mmajor, int b_nmajor,
                    float alpha, int batch, long long strideA, long long strideB, long long strideC, tt_stream_t stream) {
  for (int z = 0; z < batch; ++z) {
    const float *a = A + z * strideA, *b = B + z * strideB;
    float* c = C + z * strideC;
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double s = 0.0;
        for (int k = 0; k < K; ++k)
          s += (double)(a_mmajor ? a[(size_t)k * lda + m] : a[(size_t)m * lda + k]) * (double)(b_nmajor ? b[(size_t)k * ldb + n] : b[(size_t)n * ldb + k]);
        c[(size_t)m * ldc + n] = (float)(alpha * s);
      }
  }
  return 0;
}

/* ---- dino_vision_transformer.py:219-231: nn.functional.interpolate(patch_pos_embed, scale_factor=(sh, sw), mode="bicubic"):
 *      align_corners False, source coordinate = (dst + 0.5) / scale_factor - 0.5, cubic convolution with A = -0.75 (ATen's
 *      upsample_bicubic2d), taps clamped to the border; the class row is copied (:232-233). */
static void cubic_taps(double t, double w[4]) {
  const double A = -0.75, x0 = t + 1.0, x3 = 2.0 - t, u = 1.0 - t;
  w[0] = ((A * x0 - 5.0 * A) * x0 + 8.0 * A) * x0 - 4.0 * A;
  w[1] = ((A + 2.0) * t - (A + 3.0)) * t * t + 1.0;
  w[2] = ((A + 2.0) * u - (A + 3.0)) * u * u + 1.0;
  w[3] = ((A * x3 - 5.0 * A) * x3 + 8.0 * A) * x3 - 4.0 * A;
}
int tt_cpu_pos_embed_interpolate(const float* pos, float* out, int g, int gh, int gw, int D, float scale_h, float scale_w, tt_stream_t stream) {
  memcpy(out, pos, (size_t)D * sizeof(float));
  const float rh = 1.0f / scale_h, rw = 1.0f / scale_w;   /* ATen keeps the coordinate scale in the tensor's dtype */
  for (int oy = 0; oy < gh; ++oy)
    for (int ox = 0; ox < gw; ++ox) {
      const float ry = rh * (oy + 0.5f) - 0.5f, rx = rw * (ox + 0.5f) - 0.5f;
      const int fy = (int)floorf(ry), fx = (int)floorf(rx);
      double wy[4], wx[4];
      cubic_taps((double)(ry - (float)fy), wy);
      cubic_taps((double)(rx - (float)fx), wx);
      for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int i = 0; i < 4; ++i) {
          int yy = fy - 1 + i; yy = yy < 0 ? 0 : (yy > g - 1 ? g - 1 : yy);
          double row = 0.0;
          for (int j = 0; j < 4; ++j) {
            int xx = fx - 1 + j; xx = xx < 0 ? 0 : (xx > g - 1 ? g - 1 : xx);
            row += wx[j] * (double)pos[(size_t)(1 + yy * g + xx) * D + d];
          }
          acc += wy[i] * row;
        }
        out[(size_t)(1 + oy * gw + ox) * D + d] = (float)acc;
      }
    }
  return 0;
}

/* ---- clustering.py:34-36: nn.functional.interpolate(x.double(), (R, R), mode="bilinear").float() on token maps [M, g*g, C] */
int tt_cpu_upsample_bilinear_tokens(const float* x, float* out, int M, int g, int C, int R, tt_stream_t stream) {
  const double scale = (double)g / (double)R;
  for (int m = 0; m < M; ++m)
    for (int oy = 0; oy < R; ++oy)
      for (int ox = 0; ox < R; ++ox) {
        double sy = scale * (oy + 0.5) - 0.5, sx = scale * (ox + 0.5) - 0.5;
        sy = sy < 0 ? 0 : sy; sx = sx < 0 ? 0 : sx;
        const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < g - 1), x1 = x0 + (x0 < g - 1);
        const double ly = sy - y0, lx = sx - x0, hy = 1.0 - ly, hx = 1.0 - lx;
        const float* b = x + (size_t)m * g * g * C;
        for (int c = 0; c < C; ++c)
          out[(((size_t)m * R + oy) * R + ox) * C + c] =
              (float)(hy * (hx * b[(size_t)(y0 * g + x0) * C + c] + lx * b[(size_t)(y0 * g + x1) * C + c]) +
                      ly * (hx * b[(size_t)(y1 * g + x0) * C + c] + lx * b[(size_t)(y1 * g + x1) * C + c]));
      }
  return 0;
}

/* ---- clustering.py:101-104: the same interpolation of fp32 prototype scores [M, g*g, K] (in fp32, as the reference), arg-max over K */
int tt_cpu_upsample_argmax_f32(const float* maps, int64_t* labels_out, int M, int g, int K, int R, tt_stream_t stream) {
  const float scale = (float)g / (float)R;
  for (int m = 0; m < M; ++m)
    for (int oy = 0; oy < R; ++oy)
      for (int ox = 0; ox < R; ++ox) {
        float sy = scale * (oy + 0.5f) - 0.5f, sx = scale * (ox + 0.5f) - 0.5f;
        sy = sy < 0 ? 0 : sy; sx = sx < 0 ? 0 : sx;
        const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + (y0 < g - 1), x1 = x0 + (x0 < g - 1);
        const float ly = sy - y0, lx = sx - x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const float* b = maps + (size_t)m * g * g * K;
        float best = -INFINITY;
        int besti = 0;
        for (int k = 0; k < K; ++k) {
          const float v = hy * (hx * b[(size_t)(y0 * g + x0) * K + k] + lx * b[(size_t)(y0 * g + x1) * K + k]) +
                          ly * (hx * b[(size_t)(y1 * g + x0) * K + k] + lx * b[(size_t)(y1 * g + x1) * K + k]);
          if (v > best) { best = v; besti = k; }
        }
        labels_out[((size_t)m * R + oy) * R + ox] = besti;
      }
  return 0;
}

/* ---- the mean step of Lloyd's iteration (faiss Clustering::train as clustering.py:39-41 configures it): per-label sums and counts */
size_t tt_cpu_kmeans_accumulate_workspace_bytes(long long P, int d, int k) { return 0; }
int tt_cpu_kmeans_accumulate(const float* x, const int32_t* labels, double* sums, long long* counts, long long P, int d, int k, void* workspace,
                             size_t workspace_bytes, tt_stream_t stream) {
  memset(sums, 0, (size_t)k * d * sizeof(double));
  memset(counts, 0, (size_t)k * sizeof(long long));
  for (long long p = 0; p < P; ++p) {
    const int j = labels[p];
    if (j < 0 || j >= k) continue;
    counts[j] += 1;
    for (int t = 0; t < d; ++t) sums[(size_t)j * d + t] += (double)x[p * d + t];
  }
  return 0;
}

/* ---- backward products of the nn.Linear sites on bf16-plane operands (autograd of dino_vision_transformer.py:94-103,115-130) */
int tt_cpu_linear_bwd_data_planes(const void* dy_planes, long long dys, const void* wT_planes, long long wts, int planes, const float* gelu_pre,
                                  float* dx, int M, int N, int K, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  (void)workspace; (void)workspace_bytes;
  const uint16_t *dy = (const uint16_t*)dy_planes, *wT = (const uint16_t*)wT_planes;   /* dy [P][M][N], wT [P][K][N] */
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int pa = 0; pa < planes; ++pa)
        for (int pw = 0; pa + pw < planes; ++pw)
          for (int n = 0; n < N; ++n)
            s += (double)tt_cpu_bf16_to_f32(dy[pa * dys + (size_t)m * N + n]) * (double)tt_cpu_bf16_to_f32(wT[pw * wts + (size_t)k * N + n]);
      float v = (float)s;
      if (gelu_pre) v *= tt_cpu_gelu_grad(gelu_pre[(size_t)m * K + k]);
      dx[(size_t)m * K + k] = v;
    }
  return 0;
}
size_t tt_cpu_linear_bwd_weight_planes_workspace_bytes(int N, int K, int Mpad) { return 0; }
int tt_cpu_linear_bwd_weight_planes(const void* dyT_planes, long long dys, const void* xT_planes, long long xs, int planes, float* dw, int N,
                                    int K, int Mpad, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  const uint16_t *dyT = (const uint16_t*)dyT_planes, *xT = (const uint16_t*)xT_planes;   /* dyT [P][N][Mpad], xT [P][K][Mpad] */
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      double s = 0.0;
      for (int pa = 0; pa < planes; ++pa)
        for (int pw = 0; pa + pw < planes; ++pw)
          for (int m = 0; m < Mpad; ++m)
            s += (double)tt_cpu_bf16_to_f32(dyT[pa * dys + (size_t)n * Mpad + m]) * (double)tt_cpu_bf16_to_f32(xT[pw * xs + (size_t)k * Mpad + m]);
      dw[(size_t)n * K + k] = (float)s;
    }
  return 0;
}

/* ---- models.py:93-131 process_attentions: cls-query attention of the last block, mean over heads (:107-112), GaussianBlur(ksize,
 *      sigma) with reflect padding (torchvision: pdf on linspace(-half, half, ksize), normalised, outer product), ascending sort,
 *      unit mass, cumulative sum, keep where the cumulative mass exceeds 1 - threshold (:117-123), drop 8-connected components of
 *      <= 2 pixels (:124-130; by flood fill here).  float32 as the reference. */
static int fm_from_cls(const float* cls, float* mask_out, float* blurred_out, float* margin_out, int N, int H, int g, float threshold,
                       float sigma, int ksize) {
  const int n = N - 1, half = ksize / 2;
  float *att = (float*)calloc(n, 4), *blur = (float*)malloc(n * 4), *k1 = (float*)malloc(ksize * 4), *sorted = (float*)malloc(n * 4);
  int *rank = (int*)malloc(n * sizeof(int)), *stack = (int*)malloc(n * sizeof(int)), *comp = (int*)malloc(n * sizeof(int));
  unsigned char *th = (unsigned char*)malloc(n), *seen = (unsigned char*)calloc(n, 1);
  if (!att || !blur || !k1 || !sorted || !rank || !stack || !comp || !th || !seen) return -3;
  for (int h = 0; h < H; ++h)
    for (int i = 0; i < n; ++i) att[i] += cls[(size_t)h * N + i + 1] * 1.0f / (float)H;
  float ksum = 0.f;
  for (int i = 0; i < ksize; ++i) { const float x = (float)(i - half) / sigma; k1[i] = expf(-0.5f * (x * x)); ksum += k1[i]; }
  for (int i = 0; i < ksize; ++i) k1[i] /= ksum;
  for (int i = 0; i < n; ++i) {
    const int y = i / g, x = i % g;
    float s = 0.f;
    for (int dy = 0; dy < ksize; ++dy) {
      int yy = y + dy - half; yy = yy < 0 ? -yy : (yy >= g ? 2 * (g - 1) - yy : yy);
      for (int dx = 0; dx < ksize; ++dx) {
        int xx = x + dx - half; xx = xx < 0 ? -xx : (xx >= g ? 2 * (g - 1) - xx : xx);
        s += (k1[dy] * k1[dx]) * att[yy * g + xx];
      }
    }
    blur[i] = s;
    if (blurred_out) blurred_out[i] = s;
  }
  float total = 0.f;
  for (int i = 0; i < n; ++i) {   /* stable ascending rank */
    int r = 0;
    for (int j = 0; j < n; ++j) r += (blur[j] < blur[i] || (blur[j] == blur[i] && j < i)) ? 1 : 0;
    rank[i] = r;
    sorted[r] = blur[i];
  }
  for (int i = 0; i < n; ++i) total += sorted[i];
  float c = 0.f;
  for (int i = 0; i < n; ++i) { c += sorted[i] / total; sorted[i] = c; }   /* sorted[] now holds the cumulative mass */
  const float cut = (float)(1.0 - (double)threshold);
  for (int i = 0; i < n; ++i) {
    th[i] = sorted[rank[i]] > cut;
    if (margin_out) margin_out[i] = fabsf(sorted[rank[i]] - cut);
  }
  for (int i = 0; i < n; ++i) {   /* components by flood fill; sizes <= 2 cleared */
    if (!th[i] || seen[i]) continue;
    int top = 0, size = 0;
    stack[top++] = i; seen[i] = 1;
    while (top) {
      const int p = stack[--top];
      comp[size++] = p;
      const int y = p / g, x = p % g;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if ((dy | dx) != 0 && yy >= 0 && yy < g && xx >= 0 && xx < g && th[yy * g + xx] && !seen[yy * g + xx]) {
            seen[yy * g + xx] = 1;
            stack[top++] = yy * g + xx;
          }
        }
    }
    if (size <= 2) for (int q = 0; q < size; ++q) th[comp[q]] = 0;
  }
  for (int i = 0; i < n; ++i) mask_out[i] = (float)th[i];
  free(att); free(blur); free(k1); free(sorted); free(rank); free(stack); free(comp); free(th); free(seen);
  return 0;
}
int tt_cpu_foreground_mask_from_probs(const float* cls_probs, float* mask_out, float* blurred_out, float* margin_out, int F, int N, int H, int g,
                                      float threshold, float sigma, int ksize, tt_stream_t stream) {
  const int n = N - 1;
  for (int f = 0; f < F; ++f) {
    const int rc = fm_from_cls(cls_probs + (size_t)f * H * N, mask_out + (size_t)f * n, blurred_out ? blurred_out + (size_t)f * n : NULL,
                               margin_out ? margin_out + (size_t)f * n : NULL, N, H, g, threshold, sigma, ksize);
    if (rc) return rc;
  }
  return 0;
}
/* the same from the last block's qkv activations: row 0 of softmax(q k^T scale) per head (dino_vision_transformer.py:122-125) */
int tt_cpu_foreground_mask(const float* qkv, float* mask_out, float* blurred_out, float* margin_out, int F, int N, int H, int hd, int g,
                           float scale, float threshold, float sigma, int ksize, tt_stream_t stream) {
  const int D = H * hd, n = N - 1;
  float* cls = (float*)malloc((size_t)H * N * 4);
  if (!cls) return -3;
  for (int f = 0; f < F; ++f) {
    const float* base = qkv + (size_t)f * N * 3 * D;
    for (int h = 0; h < H; ++h) {
      float mx = -INFINITY, sum = 0.f;
      for (int j = 0; j < N; ++j) {
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s += base[h * hd + d] * base[(size_t)j * 3 * D + D + h * hd + d];
        cls[(size_t)h * N + j] = s * scale;
        mx = s * scale > mx ? s * scale : mx;
      }
      for (int j = 0; j < N; ++j) { cls[(size_t)h * N + j] = expf(cls[(size_t)h * N + j] - mx); sum += cls[(size_t)h * N + j]; }
      for (int j = 0; j < N; ++j) cls[(size_t)h * N + j] /= sum;
    }
    const int rc = fm_from_cls(cls, mask_out + (size_t)f * n, blurred_out ? blurred_out + (size_t)f * n : NULL,
                               margin_out ? margin_out + (size_t)f * n : NULL, N, H, g, threshold, sigma, ksize);
    if (rc) { free(cls); return rc; }
  }
  free(cls);
  return 0;
}

/* ---- mask_propagation.py:396-496 (label_propagation / propagate_labels with features_exist) as time_tuning.py:143-154 calls it,
 *      batched over clips: for each target frame t the context is frame 0 plus the last n_last frames (:480-487); affinity =
 *      exp(<f_t(q), f_ctx(p)> / temperature) in fp32 (:418-422) inside the |dy|, |dx| <= radius window (:424-429), per query keep
 *      the top-k sources over all contexts - everything >= the k-th largest, ties kept (:432-434) -, column-normalise in fp32
 *      (:436), and the target map is the fp64 product of the context maps with it (:442-444). */
static int lp_cpu(const float* xn, const float* seg0, int64_t* labels, double* pmap_last, double* pmap_all, int bs, int fs, int g, int D, int K,
                  int n_last, int radius, int topk, float temperature) {
  const int n = g * g;
  const size_t fstride = (size_t)bs * n * K;
  double* segs = pmap_all ? pmap_all : (double*)malloc((size_t)(fs - 1) * fstride * sizeof(double));
  const int cmax = 1 + n_last, win = (2 * radius + 1) * (2 * radius + 1);
  float* aff = (float*)malloc((size_t)cmax * win * sizeof(float));
  int* src = (int*)malloc((size_t)cmax * win * sizeof(int));
  float* topv = (float*)malloc((size_t)topk * sizeof(float));
  if (!segs || !aff || !src || !topv) return -3;
  for (int t = 1; t < fs; ++t) {
    int ctx[64], c = 0;
    ctx[c++] = 0;
    for (int fr = (t - n_last > 1 ? t - n_last : 1); fr < t; ++fr) ctx[c++] = fr;
    for (int b = 0; b < bs; ++b)
      for (int q = 0; q < n; ++q) {
        const int qy = q / g, qx = q % g;
        const float* ft = xn + (((size_t)t * bs + b) * n + q) * D;
        int cnt = 0;
        for (int j = 0; j < c; ++j)
          for (int sy = (qy - radius < 0 ? 0 : qy - radius); sy <= (qy + radius > g - 1 ? g - 1 : qy + radius); ++sy)
            for (int sx = (qx - radius < 0 ? 0 : qx - radius); sx <= (qx + radius > g - 1 ? g - 1 : qx + radius); ++sx) {
              const float* fsrc = xn + (((size_t)ctx[j] * bs + b) * n + sy * g + sx) * D;
              float dot = 0.f;
              for (int d = 0; d < D; ++d) dot += ft[d] * fsrc[d];
              aff[cnt] = expf(dot / temperature);
              src[cnt++] = j * n + sy * g + sx;
            }
        /* k-th largest with multiplicity (sources outside the window have affinity 0 < every exp) */
        int nt = 0;
        for (int i = 0; i < cnt; ++i) {
          int pos = nt < topk ? nt++ : -1;
          if (pos < 0) { if (aff[i] <= topv[topk - 1]) continue; pos = topk - 1; }
          while (pos > 0 && topv[pos - 1] < aff[i]) { topv[pos] = topv[pos - 1]; --pos; }
          topv[pos] = aff[i];
        }
        const float thr = nt == topk ? topv[topk - 1] : 0.f;
        float sum = 0.f;
        for (int i = 0; i < cnt; ++i) if (aff[i] >= thr) sum += aff[i];
        double* out = segs + (size_t)(t - 1) * fstride + ((size_t)b * n + q) * K;
        for (int k = 0; k < K; ++k) out[k] = 0.0;
        for (int i = 0; i < cnt; ++i) {
          if (aff[i] < thr) continue;
          const double w = (double)(aff[i] / sum);
          const int j = src[i] / n, p = src[i] % n, fr = ctx[j];
          for (int k = 0; k < K; ++k)
            out[k] += w * (fr == 0 ? (double)seg0[((size_t)b * n + p) * K + k] : segs[(size_t)(fr - 1) * fstride + ((size_t)b * n + p) * K + k]);
        }
        if (labels && t == fs - 1) {
          int best = 0;
          for (int k = 1; k < K; ++k) if (out[k] > out[best]) best = k;
          labels[(size_t)b * n + q] = best;
        }
      }
  }
  if (pmap_last) memcpy(pmap_last, segs + (size_t)(fs - 2) * fstride, fstride * sizeof(double));
  if (!pmap_all) free(segs);
  free(aff); free(src); free(topv);
  return 0;
}
size_t tt_cpu_label_propagate_workspace_bytes(int bs, int fs, int g, int D, int K, int n_last_frames) { return 0; }
int tt_cpu_label_propagate(const float* xn, const float* seg0, int64_t* labels, double* pmap_last, int bs, int fs, int g, int D, int K,
                           int n_last_frames, int radius, int topk, float temperature, int precision, void* workspace, size_t workspace_bytes,
                           tt_stream_t stream) {
  (void)precision;
  return lp_cpu(xn, seg0, labels, pmap_last, NULL, bs, fs, g, D, K, n_last_frames, radius, topk, temperature);
}
/* (the two-call form: the twin keeps no similarities - the first half is a no-op, the second the whole propagation) */
int tt_cpu_label_propagate_sims(const float* xn, int bs, int fs, int g, int D, int K, int n_last_frames, int precision, void* workspace,
                                size_t workspace_bytes, tt_stream_t stream) {
  (void)xn; (void)bs; (void)fs; (void)g; (void)D; (void)K; (void)n_last_frames; (void)precision; (void)workspace; (void)workspace_bytes; (void)stream;
  return 0;
}
int tt_cpu_label_propagate_from_sims(const float* xn, const float* seg0, int64_t* labels, double* pmap_last, int bs, int fs, int g, int D, int K,
                                     int n_last_frames, int radius, int topk, float temperature, void* workspace, size_t workspace_bytes,
                                     tt_stream_t stream) {
  return lp_cpu(xn, seg0, labels, pmap_last, NULL, bs, fs, g, D, K, n_last_frames, radius, topk, temperature);
}
int tt_cpu_label_propagate_maps(const float* xn, const float* seg0, double* pmap_all, int bs, int fs, int g, int D, int K, int n_last_frames,
                                int radius, int topk, float temperature, int precision, void* workspace, size_t workspace_bytes,
                                tt_stream_t stream) {
  (void)precision;
  return lp_cpu(xn, seg0, NULL, NULL, pmap_all, bs, fs, g, D, K, n_last_frames, radius, topk, temperature);
}

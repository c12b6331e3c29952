/* CPU twins of a subset of the C ABI (include/timetuning_hip.h).  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C restatements, on HOST pointers, with the SAME argument lists as their tt_* counterparts (the stream and any
 * workspace arguments are accepted and ignored), so that one ctypes prototype drives either side.  They cover the ops whose
 * arithmetic is integer / byte exact (the Pillow-defined image transforms, the confusion matrix) or a short, order-defined
 * float recurrence (Sinkhorn-Knopp, cross-entropy, arg-max of a bilinear upsampling, nearest-centroid assignment, column
 * moments).  The GEMM / attention / propagation ops have their restatement in oracle/timet_oracle.py (torch-CPU).
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

// Attention foreground mask: models.py:93-131 (process_attentions) - the --use_mask branch of TimeT.get_loss
// (time_tuning.py:226-227,235-236,244-246,282-283,298-299).
//
// Reference per frame: last block's attention probabilities attn[h][0][1:] (cls query -> patch keys), mean over heads,
// torchvision GaussianBlur(7, sigma 0.6, reflect padding), sort ascending, normalise to unit mass, cumulative sum,
// keep pixels whose cumulative mass exceeds 1 - threshold, then drop connected components (full connectivity) of
// <= 2 pixels - the last step on the HOST through skimage, one D2H/H2D round trip per training step.
//
// Here the whole chain is ONE launch, one 256-thread workgroup per frame, every intermediate in LDS (n <= 1024
// patches).  The cls-query probabilities are recomputed from the block's qkv buffer (what tt_attention_fwd consumes):
// one query row per head is 1/N of an attention pass, so the reference's second backbone pass that exists only to
// return attn[F,h,N,N] (dino_vision_transformer.py:256-263, 0.93 MB/frame) is never materialised.
//   sort      -> rank by counting (n compares per pixel, stable on ties); n^2 <= 1M compares per frame
//   cumsum    -> sequential fp32 over the sorted values (the order a CPU cumsum uses)
//   components-> sizes 1 and 2 are detectable locally: a pixel with no set neighbour, or a pair whose members have
//                no other set neighbour.  No labelling pass, no host.
#include "common.hpp"

namespace tt {

constexpr int FM_THREADS = 256;
constexpr int FM_MAXN = 1024;   // patches per frame (g <= 32)
constexpr int FM_MAXHD = 128;
constexpr int FM_MAXK = 15;     // blur taps

struct FmArgs {
  const float* qkv;        // [F][N][3*H*hd] or null
  const float* cls_probs;  // [F][H][N] or null (exactly one of the two)
  float* mask;             // [F][n]
  float* blurred;          // [F][n] or null
  float* margin;           // [F][n] or null: |cumulative mass - (1 - threshold)| of the pixel
  int F, N, H, hd, g, ksize;
  float scale, threshold, sigma;
};

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < FM_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

__device__ __forceinline__ int reflect_idx(int i, int g) { return i < 0 ? -i : (i >= g ? 2 * (g - 1) - i : i); }

__global__ __launch_bounds__(FM_THREADS) void foreground_mask_kernel(FmArgs a) {
  __shared__ float att[FM_MAXN];     // head-mean attention, later the sorted values
  __shared__ float blur[FM_MAXN];
  __shared__ float p[FM_MAXN + 1];   // one head's logits / probabilities (N = n + 1)
  __shared__ float cum[FM_MAXN];
  __shared__ short rank[FM_MAXN];
  __shared__ unsigned char th[FM_MAXN], cnt[FM_MAXN];
  __shared__ float qs[FM_MAXHD];
  __shared__ float k1[FM_MAXK];
  __shared__ float red[FM_THREADS / 64];
  const int tid = threadIdx.x, f = blockIdx.x;
  const int n = a.N - 1, g = a.g, D = a.H * a.hd;

  // ---- cls-query attention, mean over heads (models.py:107-112: sum_i attn[:, i] * 1 / H, in head order)
  for (int i = tid; i < n; i += FM_THREADS) att[i] = 0.f;
  for (int h = 0; h < a.H; ++h) {
    if (a.cls_probs) {
      for (int j = tid; j < a.N; j += FM_THREADS) p[j] = a.cls_probs[((size_t)f * a.H + h) * a.N + j];
      __syncthreads();
    } else {
      const float* base = a.qkv + (size_t)f * a.N * 3 * D;
      for (int d = tid; d < a.hd; d += FM_THREADS) qs[d] = base[h * a.hd + d];
      __syncthreads();
      float mx = -INFINITY;
      for (int j = tid; j < a.N; j += FM_THREADS) {
        const float4* kr = reinterpret_cast<const float4*>(base + (size_t)j * 3 * D + D + h * a.hd);
        float s = 0.f;
        for (int d = 0; d < a.hd / 4; ++d) {
          const float4 kv = kr[d];
          s += qs[4 * d] * kv.x + qs[4 * d + 1] * kv.y + qs[4 * d + 2] * kv.z + qs[4 * d + 3] * kv.w;
        }
        s *= a.scale;
        p[j] = s;
        mx = fmaxf(mx, s);
      }
      mx = block_reduce(mx, red, true);
      float sum = 0.f;
      for (int j = tid; j < a.N; j += FM_THREADS) {
        const float e = expf(p[j] - mx);
        p[j] = e;
        sum += e;
      }
      sum = block_reduce(sum, red, false);
      for (int j = tid; j < a.N; j += FM_THREADS) p[j] = p[j] / sum;
      __syncthreads();
    }
    for (int i = tid; i < n; i += FM_THREADS) att[i] += p[i + 1] * 1.0f / (float)a.H;
    __syncthreads();
  }

  // ---- Gaussian blur: kernel1d = pdf(linspace(-half, half, k)) / sum, 2-D kernel = outer product, reflect padding
  const int ks = a.ksize, half = ks / 2;
  if (tid == 0) {
    float sum = 0.f;
    for (int i = 0; i < ks; ++i) {
      const float x = (float)(i - half) / a.sigma;
      k1[i] = expf(-0.5f * (x * x));
      sum += k1[i];
    }
    for (int i = 0; i < ks; ++i) k1[i] = k1[i] / sum;
  }
  __syncthreads();
  for (int i = tid; i < n; i += FM_THREADS) {
    const int y = i / g, x = i % g;
    float s = 0.f;
    for (int dy = 0; dy < ks; ++dy) {
      const int yy = reflect_idx(y + dy - half, g);
      for (int dx = 0; dx < ks; ++dx) s += (k1[dy] * k1[dx]) * att[yy * g + reflect_idx(x + dx - half, g)];
    }
    blur[i] = s;
    if (a.blurred) a.blurred[(size_t)f * n + i] = s;
  }
  __syncthreads();

  // ---- ascending rank of every pixel (stable), sorted values into att[]
  for (int i = tid; i < n; i += FM_THREADS) {
    const float v = blur[i];
    int r = 0;
    for (int j = 0; j < n; ++j) {
      const float u = blur[j];
      r += (u < v || (u == v && j < i)) ? 1 : 0;
    }
    rank[i] = (short)r;
  }
  __syncthreads();
  for (int i = tid; i < n; i += FM_THREADS) att[rank[i]] = blur[i];
  __syncthreads();

  // ---- unit mass, cumulative sum in sorted order (models.py:117-120)
  float total = 0.f;
  for (int i = tid; i < n; i += FM_THREADS) total += att[i];
  total = block_reduce(total, red, false);
  if (tid == 0) {
    float c = 0.f;
    for (int i = 0; i < n; ++i) {
      c += att[i] / total;
      cum[i] = c;
    }
  }
  __syncthreads();
  const float cut = (float)(1.0 - (double)a.threshold);
  for (int i = tid; i < n; i += FM_THREADS) {
    const float c = cum[rank[i]];
    th[i] = c > cut ? 1 : 0;
    if (a.margin) a.margin[(size_t)f * n + i] = fabsf(c - cut);
  }
  __syncthreads();

  // ---- drop 8-connected components of <= 2 pixels (models.py:124-130)
  for (int i = tid; i < n; i += FM_THREADS) {
    int c = 0;
    if (th[i]) {
      const int y = i / g, x = i % g;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if ((dy | dx) != 0 && yy >= 0 && yy < g && xx >= 0 && xx < g) c += th[yy * g + xx];
        }
    }
    cnt[i] = (unsigned char)c;
  }
  __syncthreads();
  for (int i = tid; i < n; i += FM_THREADS) {
    float m = (float)th[i];
    if (th[i] && cnt[i] <= 1) {
      bool small = cnt[i] == 0;
      if (!small) {  // exactly one set neighbour: a 2-pixel component iff that neighbour has no other set neighbour
        const int y = i / g, x = i % g;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if ((dy | dx) != 0 && yy >= 0 && yy < g && xx >= 0 && xx < g && th[yy * g + xx]) small = cnt[yy * g + xx] == 1;
          }
      }
      if (small) m = 0.f;
    }
    a.mask[(size_t)f * n + i] = m;
  }
}

__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ x, const float* __restrict__ w, long long total4, int cols4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const float s = w[i / cols4];
  float4 v = reinterpret_cast<float4*>(x)[i];
  v.x *= s; v.y *= s; v.z *= s; v.w *= s;
  reinterpret_cast<float4*>(x)[i] = v;
}

// interpolate_pos_encoding (dino_vision_transformer.py:214-234) for inputs whose token grid is not the stored one:
// nn.functional.interpolate(patch_pos_embed [1,D,g,g], scale_factor=(sh, sw), mode="bicubic") - align_corners False, the
// coordinate scale is 1 / scale_factor (scale_factor given, not recomputed), cubic convolution with A = -0.75, taps
// clamped to the border - then the class row in front.  pos [1+g*g, D] -> out [1+gh*gw, D]; one thread per (token, 4 d's).
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x3 = 2.f - t, u = 1.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ __launch_bounds__(256) void pos_embed_bicubic_kernel(const float* __restrict__ pos, float* __restrict__ out, int g, int gh,
                                                                int gw, int D, float rscale_h, float rscale_w) {
  const int d4 = D / 4;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)(1 + gh * gw) * d4;
  if (idx >= total) return;
  const int tok = (int)(idx / d4), d = (int)(idx - (long long)tok * d4) * 4;
  if (tok == 0) {  // class position: copied
    *reinterpret_cast<float4*>(out + d) = *reinterpret_cast<const float4*>(pos + d);
    return;
  }
  const int oy = (tok - 1) / gw, ox = (tok - 1) - oy * gw;
  const float ry = rscale_h * (oy + 0.5f) - 0.5f, rx = rscale_w * (ox + 0.5f) - 0.5f;
  const float fy = floorf(ry), fx = floorf(rx);
  float wy[4], wx[4];
  cubic_coeffs(ry - fy, wy);
  cubic_coeffs(rx - fx, wx);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int yy = min(max((int)fy - 1 + i, 0), g - 1);
    float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = min(max((int)fx - 1 + j, 0), g - 1);
      const float4 v = *reinterpret_cast<const float4*>(pos + (size_t)(1 + yy * g + xx) * D + d);
      row.x += wx[j] * v.x; row.y += wx[j] * v.y; row.z += wx[j] * v.z; row.w += wx[j] * v.w;
    }
    acc.x += wy[i] * row.x; acc.y += wy[i] * row.y; acc.z += wy[i] * row.z; acc.w += wy[i] * row.w;
  }
  *reinterpret_cast<float4*>(out + (size_t)tok * D + d) = acc;
}

}  // namespace tt

using namespace tt;

extern "C" int tt_pos_embed_interpolate(const float* pos, float* out, int g, int gh, int gw, int D, float scale_h, float scale_w,
                                        tt_stream_t stream) {
  TT_REQUIRE(pos && out && g > 0 && gh > 0 && gw > 0, "pos_embed_interpolate: bad arguments");
  TT_REQUIRE(D % 4 == 0 && aligned16(pos) && aligned16(out), "pos_embed_interpolate: D must be a multiple of 4, buffers 16-byte aligned");
  TT_REQUIRE(scale_h > 0.f && scale_w > 0.f, "pos_embed_interpolate: scale factors must be positive");
  const long long total = (long long)(1 + gh * gw) * (D / 4);
  hipLaunchKernelGGL(pos_embed_bicubic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), pos, out, g, gh, gw,
                     D, 1.0f / scale_h, 1.0f / scale_w);
  TT_CHECK_LAUNCH("pos_embed_interpolate");
  return TT_OK;
}

static int launch_foreground_mask(const FmArgs& a, tt_stream_t stream, const char* who) {
  TT_REQUIRE(a.mask && a.F > 0, "%s: null pointer / no frames", who);
  TT_REQUIRE(a.g * a.g == a.N - 1 && a.N - 1 <= FM_MAXN, "%s: need N = g*g + 1 <= %d (got N=%d g=%d)", who, FM_MAXN + 1, a.N, a.g);
  TT_REQUIRE(a.ksize >= 1 && (a.ksize & 1) && a.ksize <= FM_MAXK && a.ksize / 2 < a.g, "%s: blur kernel %d needs odd size <= %d and "
             "reflect padding smaller than the grid", who, a.ksize, FM_MAXK);
  TT_REQUIRE(a.threshold > 0.f && a.threshold < 1.f && a.sigma > 0.f && a.H > 0, "%s: bad threshold/sigma/heads", who);
  hipLaunchKernelGGL(foreground_mask_kernel, dim3(a.F), dim3(FM_THREADS), 0, as_stream(stream), a);
  TT_CHECK_LAUNCH(who);
  return TT_OK;
}

extern "C" int tt_foreground_mask(const float* qkv, float* mask_out, float* blurred_out, float* margin_out, int F, int N, int H, int hd,
                                  int g, float scale, float threshold, float sigma, int ksize, tt_stream_t stream) {
  TT_REQUIRE(qkv && aligned16(qkv), "foreground_mask: qkv must be a 16-byte aligned device pointer");
  TT_REQUIRE(hd > 0 && hd % 4 == 0 && hd <= FM_MAXHD, "foreground_mask: head_dim %d must be a multiple of 4, <= %d", hd, FM_MAXHD);
  FmArgs a{qkv, nullptr, mask_out, blurred_out, margin_out, F, N, H, hd, g, ksize, scale, threshold, sigma};
  return launch_foreground_mask(a, stream, "foreground_mask");
}

extern "C" int tt_foreground_mask_from_probs(const float* cls_probs, float* mask_out, float* blurred_out, float* margin_out, int F, int N,
                                             int H, int g, float threshold, float sigma, int ksize, tt_stream_t stream) {
  TT_REQUIRE(cls_probs, "foreground_mask_from_probs: null pointer");
  FmArgs a{nullptr, cls_probs, mask_out, blurred_out, margin_out, F, N, H, 0, g, ksize, 1.f, threshold, sigma};
  return launch_foreground_mask(a, stream, "foreground_mask_from_probs");
}

extern "C" int tt_scale_rows_inplace(float* x, const float* row_scale, int rows, int cols, tt_stream_t stream) {
  TT_REQUIRE(x && row_scale && rows > 0 && cols > 0, "scale_rows: null pointer / empty");
  TT_REQUIRE(cols % 4 == 0 && aligned16(x), "scale_rows: cols must be a multiple of 4 and x 16-byte aligned");
  const long long total4 = (long long)rows * (cols / 4);
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream), x, row_scale, total4,
                     cols / 4);
  TT_CHECK_LAUNCH("scale_rows");
  return TT_OK;
}

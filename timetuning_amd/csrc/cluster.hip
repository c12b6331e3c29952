// Evaluator clustering: the GPU side of clustering.cluster_features / proto_clustering
// (clustering.py:20-117), my_utils.normalize_and_transform (my_utils.py:19-37) and the k-means the reference delegates to
// faiss (faiss.Kmeans(d, k, niter=50, nredo=5, seed=1), clustering.py:39-41,55-57,69-71,108-110).
//
// The reference moves every feature map to the host, upsamples it in fp64 with ATen, and runs faiss' CPU Lloyd iterations
// while the other ranks wait at a barrier (time_tuning.py:634-648).  Here the dense passes stay on the device:
//   column moments (StandardScaler)         one two-stage fp64 reduction over the rows
//   bilinear upsampling of token maps       [M, g*g, C] -> [M, R*R, C], fp64 arithmetic like the reference's DoubleTensor pass
//   k-means assignment                      one thread per point, centroids in LDS: an HBM-bound scan (200 B / point at d = 50)
//   k-means accumulation                    per-workgroup partial sums in LDS, then a fixed-order fold: deterministic, no atomics
// The tiny dense algebra between them (50 x 384 PCA basis from a 384 x 384 eigen-problem, k x d centroid bookkeeping, the
// Hungarian matching of a k x k score matrix) stays on the host.
#include "common.hpp"

namespace tt {

constexpr int CL_THREADS = 256;
constexpr int CL_MAXD = 1024;     // feature columns for the moments
constexpr int KM_MAXKD = 8192;    // k * d floats of centroids held in LDS (32 KB)

// ---- column moments: partial[b][0][c] = sum_r x[r][c], partial[b][1][c] = sum_r x[r][c]^2 over the block's rows (fp64)
__global__ __launch_bounds__(CL_THREADS) void col_moments_stage1(const float* __restrict__ x, double* __restrict__ partial, long long rows,
                                                                 int cols, long long rows_per_block) {
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int c = threadIdx.x; c < cols; c += CL_THREADS) {
    double s = 0.0, s2 = 0.0;
    for (long long r = r0; r < r1; ++r) {
      const double v = (double)x[r * cols + c];
      s += v;
      s2 += v * v;
    }
    partial[((long long)blockIdx.x * 2 + 0) * cols + c] = s;
    partial[((long long)blockIdx.x * 2 + 1) * cols + c] = s2;
  }
}

__global__ __launch_bounds__(CL_THREADS) void col_moments_stage2(const double* __restrict__ partial, double* __restrict__ mean,
                                                                 double* __restrict__ var, long long rows, int cols, int blocks) {
  const int c = blockIdx.x * CL_THREADS + threadIdx.x;
  if (c >= cols) return;
  double s = 0.0, s2 = 0.0;
  for (int b = 0; b < blocks; ++b) {  // fixed order
    s += partial[((long long)b * 2 + 0) * cols + c];
    s2 += partial[((long long)b * 2 + 1) * cols + c];
  }
  const double m = s / (double)rows;
  mean[c] = m;
  const double v = s2 / (double)rows - m * m;   // population variance, as StandardScaler (ddof = 0)
  var[c] = v > 0.0 ? v : 0.0;
}

// ---- bilinear upsampling of token-major maps (align_corners = False), fp64 arithmetic, fp32 in / out
__global__ __launch_bounds__(CL_THREADS) void upsample_tokens_kernel(const float* __restrict__ x, float* __restrict__ out, int g, int C,
                                                                     int R) {
  const int pix = blockIdx.x, m = blockIdx.y;
  const int oy = pix / R, ox = pix - oy * R;
  const double scale = (double)g / (double)R;
  double sy = scale * (oy + 0.5) - 0.5, sx = scale * (ox + 0.5) - 0.5;
  sy = sy < 0.0 ? 0.0 : sy;
  sx = sx < 0.0 ? 0.0 : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
  const double ly = sy - y0, lx = sx - x0, hy = 1.0 - ly, hx = 1.0 - lx;
  const float* base = x + (size_t)m * g * g * C;
  const float* p00 = base + (size_t)(y0 * g + x0) * C;
  const float* p01 = base + (size_t)(y0 * g + x1) * C;
  const float* p10 = base + (size_t)(y1 * g + x0) * C;
  const float* p11 = base + (size_t)(y1 * g + x1) * C;
  float* o = out + ((size_t)m * R * R + pix) * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    o[c] = (float)(hy * (hx * (double)p00[c] + lx * (double)p01[c]) + ly * (hx * (double)p10[c] + lx * (double)p11[c]));
}

// fp32 twin of upsample_argmax (label_prop.hip) for proto_clustering: scores [M, n, K] fp32 -> labels [M, R, R] int64,
// interpolation in fp32 like F.interpolate on a float tensor (clustering.py:101-103)
__global__ __launch_bounds__(CL_THREADS) void upsample_argmax_f32_kernel(const float* __restrict__ maps, int64_t* __restrict__ out, int g,
                                                                         int K, int R) {
  const int pix = blockIdx.x * CL_THREADS + threadIdx.x;
  if (pix >= R * R) return;
  const int m = blockIdx.y, oy = pix / R, ox = pix - oy * R;
  const float scale = (float)g / (float)R;
  float sy = scale * (oy + 0.5f) - 0.5f, sx = scale * (ox + 0.5f) - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  sx = sx < 0.f ? 0.f : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
  const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
  const float* base = maps + (size_t)m * g * g * K;
  const float* p00 = base + (size_t)(y0 * g + x0) * K;
  const float* p01 = base + (size_t)(y0 * g + x1) * K;
  const float* p10 = base + (size_t)(y1 * g + x0) * K;
  const float* p11 = base + (size_t)(y1 * g + x1) * K;
  float best = -INFINITY;
  int besti = 0;
  for (int k = 0; k < K; ++k) {
    const float v = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
    if (v > best) {
      best = v;
      besti = k;
    }
  }
  out[(size_t)m * R * R + pix] = besti;
}

// ---- k-means assignment: label = argmin_j |x - c_j|^2 (first minimum), optional squared distance.
// A workgroup owns 256 consecutive points: their rows are fetched as one contiguous, fully coalesced block into LDS (row
// stride d | 1, odd, so that the per-thread row reads below are bank-conflict-free), each thread then keeps ITS point in
// registers (d <= 64) and walks the centroids, which every lane reads from LDS at the same address (broadcast).
template <int DREG>
__global__ __launch_bounds__(CL_THREADS) void kmeans_assign_kernel(const float* __restrict__ x, const float* __restrict__ cent,
                                                                   int32_t* __restrict__ labels, float* __restrict__ dist2, long long P, int d,
                                                                   int k) {
  extern __shared__ float sm[];
  float* cs = sm;                 // [k][d]
  float* xs = sm + k * d;         // [256][ds]
  const int ds = d | 1;
  for (int i = threadIdx.x; i < k * d; i += CL_THREADS) cs[i] = cent[i];
  for (long long p0 = (long long)blockIdx.x * CL_THREADS; p0 < P; p0 += (long long)gridDim.x * CL_THREADS) {
    __syncthreads();
    if (DREG > 0) {
      const long long cnt = (P - p0 < CL_THREADS ? P - p0 : CL_THREADS) * d;
      for (long long i = threadIdx.x; i < cnt; i += CL_THREADS) xs[(i / d) * ds + (i % d)] = x[p0 * d + i];
      __syncthreads();
    }
    const long long p = p0 + threadIdx.x;
    if (p >= P) continue;
    const float* xp = DREG > 0 ? xs + threadIdx.x * ds : x + p * d;   // wide rows (d > 64) are read in place
    float best = INFINITY;
    int besti = 0;
    if (DREG > 0) {
      float xr[DREG > 0 ? DREG : 1];
#pragma unroll
      for (int t = 0; t < DREG; ++t) xr[t] = t < d ? xp[t] : 0.f;
      for (int j = 0; j < k; ++j) {
        const float* c = cs + j * d;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < DREG; ++t)
          if (t < d) {
            const float df = xr[t] - c[t];
            s += df * df;
          }
        if (s < best) {
          best = s;
          besti = j;
        }
      }
    } else {
      for (int j = 0; j < k; ++j) {
        const float* c = cs + j * d;
        float s = 0.f;
        for (int t = 0; t < d; ++t) {
          const float df = xp[t] - c[t];
          s += df * df;
        }
        if (s < best) {
          best = s;
          besti = j;
        }
      }
    }
    labels[p] = besti;
    if (dist2) dist2[p] = best;
  }
}

// ---- k-means accumulation: per-block sums[k][d] (fp32 in LDS over <= rows_per_block points, then fp64 partials)
__global__ __launch_bounds__(CL_THREADS) void kmeans_accumulate_stage1(const float* __restrict__ x, const int32_t* __restrict__ labels,
                                                                       double* __restrict__ part_sums, long long* __restrict__ part_cnt,
                                                                       long long P, int d, int k, long long pts_per_block) {
  extern __shared__ float acc[];  // [k][d] sums, then [k] counts
  float* cnt = acc + k * d;
  for (int i = threadIdx.x; i < k * d + k; i += CL_THREADS) acc[i] = 0.f;
  __syncthreads();
  const long long p0 = (long long)blockIdx.x * pts_per_block;
  const long long p1 = p0 + pts_per_block < P ? p0 + pts_per_block : P;
  // thread t owns feature columns t, t + 256, ... and walks the block's points in order: no atomics, fixed summation order
  for (int t = threadIdx.x; t < d; t += CL_THREADS)
    for (long long p = p0; p < p1; ++p) acc[labels[p] * d + t] += x[p * d + t];
  if (threadIdx.x == 0)
    for (long long p = p0; p < p1; ++p) cnt[labels[p]] += 1.f;
  __syncthreads();
  for (int i = threadIdx.x; i < k * d; i += CL_THREADS) part_sums[(long long)blockIdx.x * k * d + i] = (double)acc[i];
  for (int i = threadIdx.x; i < k; i += CL_THREADS) part_cnt[(long long)blockIdx.x * k + i] = (long long)cnt[i];
}

__global__ __launch_bounds__(CL_THREADS) void kmeans_accumulate_stage2(const double* __restrict__ part_sums, const long long* __restrict__ part_cnt,
                                                                       double* __restrict__ sums, long long* __restrict__ counts, int kd, int k,
                                                                       int blocks) {
  const int i = blockIdx.x * CL_THREADS + threadIdx.x;
  if (i < kd) {
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += part_sums[(long long)b * kd + i];
    sums[i] = s;
  }
  if (i < k) {
    long long c = 0;
    for (int b = 0; b < blocks; ++b) c += part_cnt[(long long)b * k + i];
    counts[i] = c;
  }
}

// x[r][c] = x[r][c] * scale[c] + shift[c]  (StandardScaler.transform, my_utils.py:29-30)
__global__ __launch_bounds__(CL_THREADS) void affine_cols_kernel(float* __restrict__ x, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, long long total, int cols) {
  const long long stride = (long long)gridDim.x * CL_THREADS;
  for (long long i = (long long)blockIdx.x * CL_THREADS + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cols);
    x[i] = x[i] * scale[c] + shift[c];
  }
}

static int moments_blocks(long long rows) {
  long long b = (rows + 255) / 256;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}
static int accumulate_blocks(long long P) {
  long long b = (P + 127) / 128;  // <= 128 points per block keeps the fp32 LDS sums short
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace tt

using namespace tt;

extern "C" size_t tt_col_moments_workspace_bytes(long long rows, int cols) {
  return (size_t)moments_blocks(rows) * 2 * cols * sizeof(double);
}

extern "C" int tt_col_moments(const float* x, double* mean, double* var, long long rows, int cols, void* workspace, size_t workspace_bytes,
                              tt_stream_t stream) {
  TT_REQUIRE(x && mean && var && workspace && rows > 0 && cols > 0 && cols <= CL_MAXD, "col_moments: need 0 < cols <= %d", CL_MAXD);
  TT_REQUIRE(workspace_bytes >= tt_col_moments_workspace_bytes(rows, cols), "col_moments: workspace too small");
  hipStream_t s = as_stream(stream);
  const int blocks = moments_blocks(rows);
  const long long rpb = (rows + blocks - 1) / blocks;
  double* partial = static_cast<double*>(workspace);
  hipLaunchKernelGGL(col_moments_stage1, dim3(blocks), dim3(CL_THREADS), 0, s, x, partial, rows, cols, rpb);
  hipLaunchKernelGGL(col_moments_stage2, dim3((cols + CL_THREADS - 1) / CL_THREADS), dim3(CL_THREADS), 0, s, partial, mean, var, rows, cols,
                     blocks);
  TT_CHECK_LAUNCH("col_moments");
  return TT_OK;
}

extern "C" int tt_upsample_bilinear_tokens(const float* x, float* out, int M, int g, int C, int R, tt_stream_t stream) {
  TT_REQUIRE(x && out && M > 0 && g > 0 && C > 0 && R > 0, "upsample_bilinear_tokens: bad arguments");
  const int threads = C >= 256 ? 256 : (C > 64 ? 128 : 64);
  hipLaunchKernelGGL(upsample_tokens_kernel, dim3(R * R, M), dim3(threads), 0, as_stream(stream), x, out, g, C, R);
  TT_CHECK_LAUNCH("upsample_bilinear_tokens");
  return TT_OK;
}

extern "C" int tt_upsample_argmax_f32(const float* maps, int64_t* labels_out, int M, int g, int K, int R, tt_stream_t stream) {
  TT_REQUIRE(maps && labels_out && M > 0 && g > 0 && K > 0 && R > 0, "upsample_argmax_f32: bad arguments");
  hipLaunchKernelGGL(upsample_argmax_f32_kernel, dim3((R * R + CL_THREADS - 1) / CL_THREADS, M), dim3(CL_THREADS), 0, as_stream(stream), maps,
                     labels_out, g, K, R);
  TT_CHECK_LAUNCH("upsample_argmax_f32");
  return TT_OK;
}

extern "C" int tt_kmeans_assign(const float* x, const float* centroids, int32_t* labels, float* dist2, long long P, int d, int k,
                                tt_stream_t stream) {
  TT_REQUIRE(x && centroids && labels && P > 0 && d > 0 && k > 0, "kmeans_assign: bad arguments");
  TT_REQUIRE((long long)k * d <= KM_MAXKD, "kmeans_assign: k * d = %d exceeds %d", k * d, KM_MAXKD);
  const size_t lds = sizeof(float) * ((size_t)k * d + (d <= 64 ? (size_t)CL_THREADS * (d | 1) : 0));
  static const bool lds_attr_set = [] {  // the d <= 64 tile needs up to 32 KB of centroids + 65 KB of points
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&kmeans_assign_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               128 * 1024) == hipSuccess;
  }();
  TT_REQUIRE(lds_attr_set, "kmeans_assign: could not raise the dynamic LDS limit");
  long long blocks = (P + CL_THREADS - 1) / CL_THREADS;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipStream_t s = as_stream(stream);
  if (d <= 16)
    hipLaunchKernelGGL((kmeans_assign_kernel<16>), dim3((unsigned)blocks), dim3(CL_THREADS), lds, s, x, centroids, labels, dist2, P, d, k);
  else if (d <= 64)
    hipLaunchKernelGGL((kmeans_assign_kernel<64>), dim3((unsigned)blocks), dim3(CL_THREADS), lds, s, x, centroids, labels, dist2, P, d, k);
  else
    hipLaunchKernelGGL((kmeans_assign_kernel<0>), dim3((unsigned)blocks), dim3(CL_THREADS), lds, s, x, centroids, labels, dist2, P, d, k);
  TT_CHECK_LAUNCH("kmeans_assign");
  return TT_OK;
}

extern "C" size_t tt_kmeans_accumulate_workspace_bytes(long long P, int d, int k) {
  return (size_t)accumulate_blocks(P) * ((size_t)k * d * sizeof(double) + (size_t)k * sizeof(long long));
}

extern "C" int tt_kmeans_accumulate(const float* x, const int32_t* labels, double* sums, long long* counts, long long P, int d, int k,
                                    void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(x && labels && sums && counts && workspace && P > 0 && d > 0 && k > 0, "kmeans_accumulate: bad arguments");
  TT_REQUIRE((long long)k * d + k <= KM_MAXKD, "kmeans_accumulate: k * d = %d exceeds %d", k * d, KM_MAXKD);
  TT_REQUIRE(workspace_bytes >= tt_kmeans_accumulate_workspace_bytes(P, d, k), "kmeans_accumulate: workspace too small");
  hipStream_t s = as_stream(stream);
  const int blocks = accumulate_blocks(P);
  const long long ppb = (P + blocks - 1) / blocks;
  double* part_sums = static_cast<double*>(workspace);
  long long* part_cnt = reinterpret_cast<long long*>(part_sums + (size_t)blocks * k * d);
  hipLaunchKernelGGL(kmeans_accumulate_stage1, dim3(blocks), dim3(CL_THREADS), sizeof(float) * (k * d + k), s, x, labels, part_sums, part_cnt, P, d,
                     k, ppb);
  hipLaunchKernelGGL(kmeans_accumulate_stage2, dim3((k * d + CL_THREADS - 1) / CL_THREADS), dim3(CL_THREADS), 0, s, part_sums, part_cnt, sums,
                     counts, k * d, k, blocks);
  TT_CHECK_LAUNCH("kmeans_accumulate");
  return TT_OK;
}

extern "C" int tt_affine_cols_inplace(float* x, const float* scale, const float* shift, long long rows, int cols, tt_stream_t stream) {
  TT_REQUIRE(x && scale && shift && rows > 0 && cols > 0, "affine_cols: bad arguments");
  const long long total = rows * cols;
  long long blocks = (total + CL_THREADS * 8 - 1) / (CL_THREADS * 8);
  blocks = blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(affine_cols_kernel, dim3((unsigned)blocks), dim3(CL_THREADS), 0, as_stream(stream), x, scale, shift, total, cols);
  TT_CHECK_LAUNCH("affine_cols");
  return TT_OK;
}

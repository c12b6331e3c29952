// Temporal label propagation (time_tuning.py:143-154 -> mask_propagation.py:396-496), batched over clips.
//
// The reference loops in Python over samples and frames, builds a dense [c*n, n] affinity per target frame,
// masks it to a (2r+1)^2 window, keeps the per-query top-k sources, column-normalises and multiplies the fp64
// label maps by it - on the host, one sample at a time.  Here, per target frame t:
//   1. cosine similarities target x context: batched fp32 MFMA GEMM over (clip, context frame)   [gemm_f32.hip].  They depend
//      on the features only, not on the propagated maps, so they are computed UP FRONT for a whole chunk of target frames
//      (as many as fit LP_SIMS_CAP bytes of workspace): one launch for the pairs (t, frame 0) of all t, and one or two per
//      lag d for the pairs (t, t - d) - 3 launches instead of 6 at fs = 4, 7 instead of 28 at fs = 8, each with a batch large
//      enough to fill the chip (a 196 x 196 x D product per clip is 16 tiles).
//   2. one workgroup per (clip, query patch): window gather -> exp(sim / 0.1) -> exact k-th largest with
//      multiplicity (k rounds of block arg-max) -> keep >= threshold (ties kept, as `aff[aff < min] = 0`) ->
//      fp32 normalise -> fp64 weighted sum of the kept sources' label rows.
// Because at most a handful of sources survive per query, step 2 is a sparse gather, not a GEMM.  Frame t's
// maps become context for frame t+1, so the frames are sequential; everything inside a frame is parallel.
// The last frame also emits argmax_K (the hard labels the loss consumes, time_tuning.py:296).
#include "common.hpp"
#include <cstdlib>

namespace tt {

int launch_gemm_plain2(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int batch_inner,
                       int batch_outer, long long sA, long long sB, long long sC, long long sA2, long long sB2, long long sC2, int bf16,
                       hipStream_t s);

constexpr int LP_MAXC = 8;      // context frames (frame 0 + n_last_frames <= 7)
// candidates per thread (template parameter of the kernel): 8 covers the training protocol (13x13 window x 8 context
// frames = 1352 <= 2048); 16 covers the DAVIS evaluation protocol on the 28x28 grid (25x25 window x 5 frames = 3125)
constexpr int LP_CAND_MAX = 16;
// kept sources per query = top-k plus ties: normally topk (5), but EVERY candidate on degenerate inputs (identical tokens on flat
// frames tie exactly) - the keep list holds all LP_CAND * 256 candidates a block examines, so nothing is ever dropped and
// aff / aff.sum(0) stays normalised (the former 64-entry list silently truncated such rows)

struct LpArgs {
  const float* sims;        // [bs][cs][n][n]  (target, source); slots j < c used
  const float* seg0;        // [bs][n][K] fp32 (frame 0 labels)
  const double* seg_prev;   // base of fp64 maps: frame f (>=1) at seg_prev + (f-1) * bs*n*K
  double* seg_out;          // [bs][n][K] this frame's map
  int64_t* labels;          // [bs][n] or null
  int ctx_frame[LP_MAXC];
  int c, cs, bs, g, K, radius, topk;
  float temp;
};

template <int LP_CAND>
__global__ __launch_bounds__(256) void label_prop_kernel(LpArgs a) {
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ int s_cnt[4];
  __shared__ float s_sum[4];
  constexpr int LP_MAXKEEP = LP_CAND * 256;
  __shared__ int keep_src[LP_MAXKEEP];   // ctx * n + source patch
  __shared__ float keep_w[LP_MAXKEEP];
  __shared__ double s_best[4];
  __shared__ int s_besti[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.g * a.g;
  const int qi = blockIdx.x, b = blockIdx.y;
  const int qy = qi / a.g, qx = qi - qy * a.g;
  const int y0 = max(0, qy - a.radius), y1 = min(a.g - 1, qy + a.radius);
  const int x0 = max(0, qx - a.radius), x1 = min(a.g - 1, qx + a.radius);
  const int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
  const int per_ctx = ww * wh, total = per_ctx * a.c;

  // ---- gather this thread's candidates: affinity = exp(sim / temp) inside the window (mask_propagation.py:422-429)
  float val[LP_CAND];
  int src[LP_CAND];
#pragma unroll
  for (int i = 0; i < LP_CAND; ++i) {
    const int cand = tid + 256 * i;
    val[i] = -1.f;  // below every affinity (exp > 0)
    src[i] = -1;
    if (cand < total) {
      const int j = cand / per_ctx, w = cand - j * per_ctx;
      const int sy = y0 + w / ww, sx = x0 + w % ww;
      const int sp = sy * a.g + sx;
      const float sim = a.sims[(((long long)b * a.cs + j) * n + qi) * n + sp];
      val[i] = expf(sim / a.temp);
      src[i] = j * n + sp;
    }
  }

  // ---- k-th largest with multiplicity: k rounds of (value, lowest candidate id) arg-max, removing one instance
  float thr = 0.f;
  bool taken[LP_CAND];
#pragma unroll
  for (int i = 0; i < LP_CAND; ++i) taken[i] = false;
  const int rounds = min(a.topk, total);
  for (int rd = 0; rd < rounds; ++rd) {
    float bv = -2.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < LP_CAND; ++i) {
      const int cand = tid + 256 * i;
      if (!taken[i] && src[i] >= 0 && (val[i] > bv || (val[i] == bv && cand < bi))) {
        bv = val[i];
        bi = cand;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_val[wave] = bv;
      s_idx[wave] = bi;
    }
    __syncthreads();
    bv = s_val[0];
    bi = s_idx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < bi)) {
        bv = s_val[w];
        bi = s_idx[w];
      }
    thr = bv;
#pragma unroll
    for (int i = 0; i < LP_CAND; ++i)
      if (tid + 256 * i == bi) taken[i] = true;
    __syncthreads();
  }
  // fewer candidates than topk: the reference's top-k then includes masked zeros, threshold 0 keeps everything
  if (total < a.topk) thr = 0.f;

  // ---- keep >= threshold (ties kept), deterministic compaction in candidate order, fp32 column sum
  int mycount = 0;
  float mysum = 0.f;
#pragma unroll
  for (int i = 0; i < LP_CAND; ++i)
    if (src[i] >= 0 && val[i] >= thr) {
      ++mycount;
      mysum += val[i];
    }
  // exclusive prefix of counts across the block in thread order (candidate order within a thread is i-major,
  // which is not global candidate order, but it is a fixed order: results are reproducible run to run)
  int incl = mycount;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  float wsum = wave_sum(mysum);
  if (lane == 63) s_cnt[wave] = incl;
  if (lane == 0) s_sum[wave] = wsum;
  __syncthreads();
  int offset = incl - mycount;
  for (int w = 0; w < wave; ++w) offset += s_cnt[w];
  const int nkeep = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  const float colsum = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
#pragma unroll
  for (int i = 0; i < LP_CAND; ++i)
    if (src[i] >= 0 && val[i] >= thr) {
      if (offset < LP_MAXKEEP) {
        keep_src[offset] = src[i];
        keep_w[offset] = val[i] / colsum;  // aff / aff.sum(0) in fp32 (mask_propagation.py:436)
      }
      ++offset;
    }
  __syncthreads();
  const int nk = min(nkeep, LP_MAXKEEP);

  // ---- seg_tar[:, q] = sum_s segs[:, s] * aff[s, q] in fp64 (mask_propagation.py:442-444)
  double best = -1.0;
  int besti = 0x7fffffff;
  const long long fstride = (long long)a.bs * n * a.K;
  for (int k = tid; k < a.K; k += 256) {
    double acc = 0.0;
    for (int e = 0; e < nk; ++e) {
      const int j = keep_src[e] / n, sp = keep_src[e] - j * n;
      const int fr = a.ctx_frame[j];
      const long long off = ((long long)b * n + sp) * a.K + k;
      const double sv = (fr == 0) ? (double)a.seg0[off] : a.seg_prev[(long long)(fr - 1) * fstride + off];
      acc += sv * (double)keep_w[e];
    }
    a.seg_out[((long long)b * n + qi) * a.K + k] = acc;
    if (acc > best) {  // k ascending within a thread: first maximum wins
      best = acc;
      besti = k;
    }
  }
  if (!a.labels) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) {
      best = ov;
      besti = oi;
    }
  }
  if (lane == 0) {
    s_best[wave] = best;
    s_besti[wave] = besti;
  }
  __syncthreads();
  if (tid == 0) {
    best = s_best[0];
    besti = s_besti[0];
    for (int w = 1; w < 4; ++w)
      if (s_best[w] > best || (s_best[w] == best && s_besti[w] < besti)) {
        best = s_best[w];
        besti = s_besti[w];
      }
    a.labels[(long long)b * n + qi] = besti;  // torch.argmax: first index of the maximum
  }
}

// The same per-query work with ONE WAVE per (clip, query) and no LDS or barriers, for windows of at most 16 x 16 patches and at
// most CTX context frames (the training protocol: radius 6 -> 13 x 13).  A lane owns window column lane & 15 and rows
// (lane >> 4) + 4 r of every context (4 CTX candidate slots, no integer division); the k rounds of arg-max are wave shuffles, the
// kept sources are broadcast lane by lane in (slot, lane) order, a lane owns the label channels k = lane + 64 t.  Same arithmetic
// as label_prop_kernel (fp32 affinities, fp32 column sum, fp64 maps); the column sum and the fp64 accumulation visit the kept
// sources in a different fixed order.  Measured on C2's launch (32 clips x 196 queries, 3 contexts, K = 200): 48.6 us for the
// workgroup-per-query kernel, 24.5 us for this one - which is bound by its instruction count (every wave of the launch is
// resident at once; compacting the kept list through LDS to batch the label-row loads made it slower, 32.7 us).
template <int CTX, int KT>
__global__ __launch_bounds__(256) void label_prop_wave_kernel(LpArgs a) {
  constexpr int CPL = 4 * CTX;
  const int lane = threadIdx.x & 63;
  const int n = a.g * a.g;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long long)a.bs * n) return;
  const int b = (int)(item / n), qi = (int)(item - (long long)b * n);
  const int qy = qi / a.g, qx = qi - qy * a.g;
  const int y0 = max(0, qy - a.radius), y1 = min(a.g - 1, qy + a.radius);
  const int x0 = max(0, qx - a.radius), x1 = min(a.g - 1, qx + a.radius);
  const int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
  const int total = ww * wh * a.c;
  const int lx = lane & 15, ly = lane >> 4;

  float val[CPL];
  int src[CPL];   // context << 12 | source patch, -1 = no candidate
#pragma unroll
  for (int j = 0; j < CTX; ++j) {
    const float* row = a.sims + (((long long)b * a.cs + j) * n + qi) * n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = j * 4 + r, y = ly + 4 * r;
      val[i] = -1.f;
      src[i] = -1;
      if (j < a.c && lx < ww && y < wh) {
        const int sp = (y0 + y) * a.g + x0 + lx;
        val[i] = expf(row[sp] / a.temp);
        src[i] = (j << 12) | sp;
      }
    }
  }
  // k-th largest with multiplicity
  float thr = 0.f;
  unsigned taken = 0u;
  const int rounds = min(a.topk, total);
  for (int rd = 0; rd < rounds; ++rd) {
    float bv = -2.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
      if (!((taken >> i) & 1u) && src[i] >= 0 && val[i] > bv) {   // (slots ascend: the first of equal values has the lowest id)
        bv = val[i];
        bi = lane + 64 * i;
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    thr = bv;
    if ((bi & 63) == lane) taken |= 1u << (bi >> 6);
  }
  if (total < a.topk) thr = 0.f;
  float mysum = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i)
    if (src[i] >= 0 && val[i] >= thr) mysum += val[i];
  const float colsum = wave_sum(mysum);

  // seg_tar[:, q] = sum_s segs[:, s] * aff[s, q] in fp64 (mask_propagation.py:442-444)
  double acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = 0.0;
  const long long fstride = (long long)a.bs * n * a.K;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    unsigned long long mask = __ballot(src[i] >= 0 && val[i] >= thr);
    while (mask) {
      const int l = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int s_ = __shfl(src[i], l, 64);
      const double w = (double)(__shfl(val[i], l, 64) / colsum);   // aff / aff.sum(0) in fp32 (mask_propagation.py:436)
      const int fr = a.ctx_frame[s_ >> 12];
      const long long off = ((long long)b * n + (s_ & 4095)) * a.K;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int k = lane + 64 * t;
        if (k < a.K) acc[t] += ((fr == 0) ? (double)a.seg0[off + k] : a.seg_prev[(long long)(fr - 1) * fstride + off + k]) * w;
      }
    }
  }
  double best = -1.0;
  int besti = 0x7fffffff;
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int k = lane + 64 * t;
    if (k < a.K) {
      a.seg_out[((long long)b * n + qi) * a.K + k] = acc[t];
      if (acc[t] > best) {
        best = acc[t];
        besti = k;
      }
    }
  }
  if (!a.labels) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) {
      best = ov;
      besti = oi;
    }
  }
  if (lane == 0) a.labels[(long long)b * n + qi] = besti;   // torch.argmax: first index of the maximum
}

__global__ void f64_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) dst[i] = src[i];
}

static int lp_cmax(int fs, int n_last) {
  int c = 1 + (fs - 2 < n_last ? fs - 2 : n_last);
  return c < 1 ? 1 : c;
}

// target frames whose similarities are held at once (all of them unless that needs more than LP_SIMS_CAP bytes).  256 MB holds every
// target frame of C2 (44 MB) and all but one of C4's (138 MB at 16 clips x 8 frames); the evaluation protocol's long clips on the
// 28 x 28 grid are chunked (there a few more similarity launches do not matter - the per-query kernel dominates).  The cost is
// workspace the caller allocates: see INTEGRATION.md "Workspaces".
constexpr size_t LP_SIMS_CAP = 256ull << 20;
static int lp_chunk(int bs, int fs, int n, int n_last) {
  const size_t per_t = (size_t)bs * lp_cmax(fs, n_last) * n * n * sizeof(float);
  const char* e = getenv("TT_LP_SIMS_CAP_MB");   // test aid: a small cap exercises the chunked path on small inputs
  size_t T = (e ? (size_t)atoll(e) << 20 : LP_SIMS_CAP) / per_t;
  if (T < 1) T = 1;
  if (T > (size_t)(fs - 1)) T = fs > 1 ? fs - 1 : 1;
  return (int)T;
}
static size_t lp_workspace(int T, int bs, int fs, size_t n, int K, int n_last) {
  const size_t sims = (size_t)T * bs * lp_cmax(fs, n_last) * n * n * sizeof(float);
  const size_t segs = (size_t)(fs > 1 ? fs - 1 : 1) * bs * n * K * sizeof(double);
  return ((sims + 255) / 256) * 256 + segs;
}

}  // namespace tt

using namespace tt;

extern "C" size_t tt_label_propagate_workspace_bytes(int bs, int fs, int g, int D, int K, int n_last_frames) {
  (void)D;
  const size_t n = (size_t)g * g;
  return lp_workspace(lp_chunk(bs, fs, (int)n, n_last_frames), bs, fs, n, K, n_last_frames);
}

static int lp_run(const char* who, const float* xn, const float* seg0, int64_t* labels, double* pmap_last, double* pmap_all, int bs,
                  int fs, int g, int D, int K, int n_last_frames, int radius, int topk, float temperature, int precision, void* workspace,
                  size_t workspace_bytes, tt_stream_t stream, int phase = 0) {
  // phase: 0 the whole propagation; 1 the similarities only (tt_label_propagate_sims: they do not depend on seg0 - a caller may compute them
  // beside other work); 2 the propagation from similarities phase 1 left in `workspace`.  1 / 2 need every target frame in ONE chunk.
  TT_REQUIRE(xn && (seg0 || phase == 1) && workspace, "%s: null pointer", who);
  TT_REQUIRE(precision >= TT_PRECISION_F32 && precision <= TT_PRECISION_BF16, "%s: precision must be 0, 1 or 2 (got %d)", who, precision);
  const int sims_bf16 = precision == TT_PRECISION_BF16;   // the "bf16" mode: the cosine similarities on bf16 MFMA (what torch.autocast makes of them)
  TT_REQUIRE(bs > 0 && fs >= 2 && g > 0 && D > 0 && K > 0, "%s: need fs >= 2 and positive sizes", who);
  TT_REQUIRE(n_last_frames >= 0 && n_last_frames + 1 <= LP_MAXC, "%s: n_last_frames must be <= %d", who, LP_MAXC - 1);
  TT_REQUIRE(radius > 0, "%s: size_mask_neighborhood must be > 0 (the unrestricted variant is not built)", who);
  TT_REQUIRE(topk >= 1, "%s: topk >= 1", who);
  const int win = (2 * radius + 1 < g ? 2 * radius + 1 : g);
  const long long cand_max = (long long)win * win * lp_cmax(fs, n_last_frames);
  TT_REQUIRE(cand_max <= 256LL * LP_CAND_MAX, "%s: window %dx%d with %d context frames exceeds %d candidates per query", who, win, win,
             lp_cmax(fs, n_last_frames), 256 * LP_CAND_MAX);
  TT_REQUIRE(D % 4 == 0, "%s: feature dim must be a multiple of 4", who);
  hipStream_t s = as_stream(stream);
  const int n = g * g;
  // The chunk length follows the workspace that was actually handed over, not a second reading of the cap (the environment may have
  // changed between the caller's size query and this call): the largest chunk, up to the cap's, that fits.
  int T = lp_chunk(bs, fs, n, n_last_frames);
  while (T > 1 && lp_workspace(T, bs, fs, (size_t)n, K, n_last_frames) > workspace_bytes) --T;
  TT_REQUIRE(workspace_bytes >= lp_workspace(T, bs, fs, (size_t)n, K, n_last_frames), "%s: workspace too small", who);
  const int cmax = lp_cmax(fs, n_last_frames);
  const size_t sims_bytes = (((size_t)T * bs * cmax * n * n * sizeof(float)) + 255) / 256 * 256;
  float* sims = static_cast<float*>(workspace);
  // the fp64 maps of frames 1..fs-1: the caller's buffer when all of them are wanted, the workspace otherwise
  double* segs = pmap_all ? pmap_all : reinterpret_cast<double*>(static_cast<char*>(workspace) + sims_bytes);
  const long long fstride = (long long)bs * n * K;
  const long long nn = (long long)n * n, frame = (long long)bs * n * D, per_t = (long long)bs * cmax * nn;
  if (phase != 0 && T < fs - 1) {
    if (phase == 1) return 1;   // (more than one chunk: nothing written - the caller runs the whole propagation in one call)
    set_error("%s: the similarities of %d target frames do not fit one chunk of this workspace (chunk %d)", who, fs - 1, T);
    return TT_EINVAL;
  }
  for (int t0 = 1; t0 < fs; t0 += T) {
    const int t1 = t0 + T < fs ? t0 + T : fs;
    // ---- cosine similarities of the chunk, sims[t - t0][b][j] = xn[t][b] @ xn[ctx_j(t)][b]^T with ctx(t) = {0} + [lo_t, t),
    //      lo_t = max(1, t - n_last)  (mask_propagation.py:480-487: the first frame and the queue of the last n_last frames)
    // slot 0, the pairs (t, 0) of every t in the chunk: inner batch = clip, outer = t
    int rc = phase == 2 ? TT_OK : launch_gemm_plain2(xn + t0 * frame, xn, sims, n, n, D, D, D, n, bs, t1 - t0, (long long)n * D, (long long)n * D,
                                                     (long long)cmax * nn, frame, 0, per_t, sims_bf16, s);
    if (rc != TT_OK) return rc;
    for (int d = 1; d <= n_last_frames && phase != 2; ++d) {
      // pairs (t, t - d), t - d >= 1.  While the queue is still filling (t <= n_last + 1: lo_t = 1) the slot is j = t - d and
      // moves with t; afterwards (lo_t = t - n_last) it is j = 1 + n_last - d
      const int a0 = t0 > d + 1 ? t0 : d + 1;
      const int a1 = t1 < n_last_frames + 2 ? t1 : n_last_frames + 2;
      if (a1 > a0) {
        rc = launch_gemm_plain2(xn + a0 * frame, xn + (a0 - d) * frame, sims + (a0 - t0) * per_t + (a0 - d) * nn, n, n, D, D, D, n, bs,
                                a1 - a0, (long long)n * D, (long long)n * D, (long long)cmax * nn, frame, frame, per_t + nn, sims_bf16, s);
        if (rc != TT_OK) return rc;
      }
      const int b0 = a0 > n_last_frames + 2 ? a0 : n_last_frames + 2;
      if (t1 > b0) {
        rc = launch_gemm_plain2(xn + b0 * frame, xn + (b0 - d) * frame, sims + (b0 - t0) * per_t + (1 + n_last_frames - d) * nn, n, n, D, D,
                                D, n, bs, t1 - b0, (long long)n * D, (long long)n * D, (long long)cmax * nn, frame, frame, per_t, sims_bf16, s);
        if (rc != TT_OK) return rc;
      }
    }
    if (phase == 1) return TT_OK;
    // ---- the maps, frame by frame: frame t's map is context for frame t + 1
    for (int t = t0; t < t1; ++t) {
      LpArgs a{};
      int c = 0;
      a.ctx_frame[c++] = 0;  // the first frame is always context (mask_propagation.py:482-483)
      const int lo = (t - n_last_frames > 1) ? t - n_last_frames : 1;
      for (int fr = lo; fr < t; ++fr) a.ctx_frame[c++] = fr;
      a.sims = sims + (t - t0) * per_t;
      a.seg0 = seg0;
      a.seg_prev = segs;
      a.seg_out = segs + (long long)(t - 1) * fstride;
      a.labels = (t == fs - 1) ? labels : nullptr;
      a.c = c; a.cs = cmax; a.bs = bs; a.g = g; a.K = K; a.radius = radius; a.topk = topk; a.temp = temperature;
      const unsigned wgrid = (unsigned)(((long long)bs * n + 3) / 4);
      static const bool wave_env = [] { const char* e = getenv("TT_LP_WAVE"); return !e || atoi(e) != 0; }();   // tuning aid
      const bool wave_kernel = wave_env && win <= 16 && n <= 4096 && K <= 512;   // (source patch packed into 12 bits)
      if (wave_kernel && c <= 3 && K <= 256)
        hipLaunchKernelGGL((label_prop_wave_kernel<3, 4>), dim3(wgrid), dim3(256), 0, s, a);
      else if (wave_kernel && c <= 3)
        hipLaunchKernelGGL((label_prop_wave_kernel<3, 8>), dim3(wgrid), dim3(256), 0, s, a);
      else if (wave_kernel && K <= 256)
        hipLaunchKernelGGL((label_prop_wave_kernel<8, 4>), dim3(wgrid), dim3(256), 0, s, a);
      else if (wave_kernel)
        hipLaunchKernelGGL((label_prop_wave_kernel<8, 8>), dim3(wgrid), dim3(256), 0, s, a);
      else if (cand_max <= 256LL * 8)
        hipLaunchKernelGGL((label_prop_kernel<8>), dim3(n, bs), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL((label_prop_kernel<LP_CAND_MAX>), dim3(n, bs), dim3(256), 0, s, a);
      TT_CHECK_LAUNCH(who);
    }
  }
  if (pmap_last) {
    const long long cnt = fstride;
    hipLaunchKernelGGL(f64_copy_kernel, dim3(1024), dim3(256), 0, s, segs + (long long)(fs - 2) * fstride, pmap_last, cnt);
    TT_CHECK_LAUNCH(who);
  }
  return TT_OK;
}

extern "C" int tt_label_propagate(const float* xn, const float* seg0, int64_t* labels, double* pmap_last, int bs, int fs, int g,
                                  int D, int K, int n_last_frames, int radius, int topk, float temperature, int precision, void* workspace,
                                  size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(labels, "label_propagate: null pointer");
  return lp_run("label_propagate", xn, seg0, labels, pmap_last, nullptr, bs, fs, g, D, K, n_last_frames, radius, topk, temperature,
                precision, workspace, workspace_bytes, stream);
}

extern "C" int tt_label_propagate_sims(const float* xn, int bs, int fs, int g, int D, int K, int n_last_frames, int precision, void* workspace,
                                       size_t workspace_bytes, tt_stream_t stream) {
  return lp_run("label_propagate_sims", xn, nullptr, nullptr, nullptr, nullptr, bs, fs, g, D, K, n_last_frames, 1, 1, 1.0f, precision, workspace,
                workspace_bytes, stream, 1);
}

extern "C" int tt_label_propagate_from_sims(const float* xn, const float* seg0, int64_t* labels, double* pmap_last, int bs, int fs, int g, int D,
                                            int K, int n_last_frames, int radius, int topk, float temperature, void* workspace,
                                            size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(labels, "label_propagate_from_sims: null pointer");
  return lp_run("label_propagate_from_sims", xn, seg0, labels, pmap_last, nullptr, bs, fs, g, D, K, n_last_frames, radius, topk, temperature,
                TT_PRECISION_F32, workspace, workspace_bytes, stream, 2);
}

extern "C" int tt_label_propagate_maps(const float* xn, const float* seg0, double* pmap_all, int bs, int fs, int g, int D, int K,
                                       int n_last_frames, int radius, int topk, float temperature, int precision, void* workspace,
                                       size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(pmap_all, "label_propagate_maps: null pointer");
  return lp_run("label_propagate_maps", xn, seg0, nullptr, nullptr, pmap_all, bs, fs, g, D, K, n_last_frames, radius, topk, temperature,
                precision, workspace, workspace_bytes, stream);
}

// ---- evaluation tail of mask_propagation.py:826-829: bilinear upsample (align_corners=False) of the fp64 maps to the
// input resolution, then argmax over the label channel - fused, so the [M, K, R, R] fp64 tensor (77 MB per 25-frame clip
// at K = 8) is never written.
namespace tt {
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const double* __restrict__ maps, int64_t* __restrict__ out, int g, int K,
                                                              int R) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= R * R) return;
  const int m = blockIdx.y, oy = pix / R, ox = pix - oy * R;
  const double scale = (double)g / (double)R;  // area_pixel_compute_scale, align_corners = False
  double sy = scale * (oy + 0.5) - 0.5, sx = scale * (ox + 0.5) - 0.5;
  sy = sy < 0.0 ? 0.0 : sy;
  sx = sx < 0.0 ? 0.0 : sx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
  const double ly = sy - y0, lx = sx - x0, hy = 1.0 - ly, hx = 1.0 - lx;
  const double* base = maps + (long long)m * g * g * K;
  const double* p00 = base + (long long)(y0 * g + x0) * K;
  const double* p01 = base + (long long)(y0 * g + x1) * K;
  const double* p10 = base + (long long)(y1 * g + x0) * K;
  const double* p11 = base + (long long)(y1 * g + x1) * K;
  double best = -INFINITY;
  int besti = 0;
  for (int k = 0; k < K; ++k) {
    const double v = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
    if (v > best) {  // torch.max: first index of the maximum
      best = v;
      besti = k;
    }
  }
  out[(long long)m * R * R + pix] = besti;
}

// counts[gt * C + pred] += 1 over n pixels (labels outside [0, C) are ignored): the confusion matrix behind the
// Jaccard index of the propagated masks
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ gt, long long n,
                                                        int C, unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned int hist[];
  const int cells = C * C;
  for (int i = threadIdx.x; i < cells; i += 256) hist[i] = 0;
  __syncthreads();
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const long long p = pred[i], t = gt[i];
    if (p >= 0 && p < C && t >= 0 && t < C) atomicAdd(&hist[(int)t * C + (int)p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cells; i += 256)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

// more classes than an LDS histogram holds (many-to-one evaluation with hundreds of clusters): global integer atomics
__global__ __launch_bounds__(256) void confusion_global_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ gt, long long n,
                                                               int C, unsigned long long* __restrict__ counts) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const long long p = pred[i], t = gt[i];
    if (p >= 0 && p < C && t >= 0 && t < C) atomicAdd(&counts[t * C + p], 1ull);
  }
}
}  // namespace tt

extern "C" int tt_upsample_argmax(const double* maps, int64_t* labels_out, int M, int g, int K, int R, tt_stream_t stream) {
  TT_REQUIRE(maps && labels_out && M > 0 && g > 0 && K > 0 && R > 0, "upsample_argmax: bad arguments");
  hipLaunchKernelGGL(upsample_argmax_kernel, dim3((R * R + 255) / 256, M), dim3(256), 0, as_stream(stream), maps, labels_out, g, K, R);
  TT_CHECK_LAUNCH("upsample_argmax");
  return TT_OK;
}

extern "C" int tt_confusion_counts(const int64_t* pred, const int64_t* gt, long long n, int C, unsigned long long* counts,
                                   tt_stream_t stream) {
  TT_REQUIRE(pred && gt && counts && n > 0, "confusion_counts: bad arguments");
  TT_REQUIRE(C > 0 && C <= 4096, "confusion_counts: need 0 < classes <= 4096 (got %d)", C);
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(counts, 0, sizeof(unsigned long long) * C * C, s) != hipSuccess) {
    set_error("confusion_counts: memset failed");
    return TT_ELAUNCH;
  }
  long long blocks = (n + 256 * 16 - 1) / (256 * 16);
  blocks = blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks);
  if (C <= 96)
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), sizeof(unsigned int) * C * C, s, pred, gt, n, C, counts);
  else
    hipLaunchKernelGGL(confusion_global_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred, gt, n, C, counts);
  TT_CHECK_LAUNCH("confusion_counts");
  return TT_OK;
}

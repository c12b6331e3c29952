// Sinkhorn-Knopp optimal assignment (time_tuning.py:157-168 + my_utils.py:246-274) and the
// cross-entropy objective (time_tuning.py:296-302).
//
// Scaling-vector form.  With E[b][k] = exp(scores[b][k] / eps) the reference's in-place updates of
// Q = diag(a) E^T diag(b) only ever change the two vectors:
//   row step   a_k <- (1/K) / sum_b E[b][k] * b_b        (Q *= (r/u)[:,None],  u = Q.sum(1))
//   col step   b_b <- (1/B) / sum_k a_k * E[b][k]        (Q *= (c/Q.sum(0))[None,:])
//   output     q[b][k] = a_k E[b][k] / sum_k a_k E[b][k] (final Q / Q.sum(0), transposed)
// so E is written once and then only READ (one pass per iteration: the column step of iteration i and the
// row sums of iteration i+1 share a sweep), no transposes, no in-place N x K rewrites.  The initial
// Q /= sum(Q) cancels in the first row step and is not computed.  E stays [B][K] (a row = one patch's K
// scores = one coalesced wave read).
//
// The problem is small (5 MB at C2, L2 / Infinity-Cache resident) and each pass ends in a K-vector reduction,
// so it is latency-bound, not bandwidth-bound: a few FAT workgroups (1024 threads = 16 waves, one row per wave
// per step, two rows in flight) keep every pass short and the cross-workgroup fold tiny.  A workgroup owns a run
// of rows, keeps its per-prototype partial sums in registers and publishes partial[wg][k]; the next launch folds
// the partials in a fixed order (deterministic, no atomics).  Kernel boundaries (~1.5 us) are cheaper than a grid
// barrier (~4-7 us) on this chip, so each iteration is its own launch.  (Measured alternative: letting the last workgroup
// of a launch fold the partials into u[K] - one atomic ticket plus agent-scope release / acquire fences, needed because the
// XCDs' L2s are not coherent - so that the next launch reads K floats instead of every workgroup re-reading all partials:
// 287 us instead of 89 us per 10-iteration solve at B = 6272 and growing with the workgroup count; the fences write back /
// invalidate L2 around the 5 MB of E every launch.  The redundant fold of at most 96 x K floats is the cheaper evil.)
#include "common.hpp"
#include <cstdlib>

namespace tt {

constexpr int SK_KPL = 8;      // K <= 512
constexpr int SK_MAXWG = 256;  // buffer capacity
// Workgroup-count policy, measured (tools/sk_sweep.py, K = 200): the best count grows with the problem - 64-96 at B = 6272
// (one rank), 128 at 12544, 192 at 25088, 192-256 at 50176 (the 8-rank global problem, 337 -> 193 us) - because the
// per-launch fold of the partial sums costs O(workgroups) while the sweep over E shrinks as 1/workgroups.
// Round 6: the fold's loads go out in whole batches of SK_FOLD_BATCH = 64 partials (sk_iter_kernel), so the count is a multiple of 64: 128 at
// B = 6272 (67.9 us per solve against 69.4 with 64), 128 at 12544, 192 - 256 beyond.
static int sk_default_cap(int B) {
  int c = (B / 98 + 63) / 64 * 64;   // ~ 98 rows per workgroup, up to a multiple of 64
  if (B >= 4096 && c < 128) c = 128;
  return c < 64 ? 64 : (c > SK_MAXWG ? SK_MAXWG : c);
}
#ifndef SK_FOLD_BATCH
#define SK_FOLD_BATCH 64
#endif
constexpr int SK_THREADS = 1024;
constexpr int SK_WAVES = SK_THREADS / 64;

// launches one of the two instantiations (columns per lane 4 | 8) of a kernel templated on KPL
#define SK_LAUNCH_KPL(KERNEL4, KERNEL8, grid, ...)                                                              \
  do {                                                                                                           \
    if (K <= 256) hipLaunchKernelGGL(KERNEL4, grid, dim3(SK_THREADS), 0, s, __VA_ARGS__);                        \
    else hipLaunchKernelGGL(KERNEL8, grid, dim3(SK_THREADS), 0, s, __VA_ARGS__);                                 \
  } while (0)

static int sk_wgs(int B) {
  int w = (B + 2 * SK_WAVES - 1) / (2 * SK_WAVES);  // >= 2 rows per wave
  static const int cap_env = [] { const char* e = getenv("TT_SK_WGS"); return e ? atoi(e) : 0; }();  // tuning aid
  const int cap = (cap_env > 0 && cap_env <= SK_MAXWG) ? cap_env : sk_default_cap(B);
  return w > cap ? cap : (w < 1 ? 1 : w);
}

// KPL = columns per lane (lane owns k = lane + 64 i, i < KPL): 4 for K <= 256 - the training shapes (K = 200): half the predicated loads
// and registers of the general 8 (round 6) - 8 up to K = 512.
// block-wide fold of per-wave register partials acc[KPL] (lane owns k = lane + 64 i) into out[k]
template <int KPL>
__device__ __forceinline__ void fold_waves(const float (&acc)[KPL], float (*red)[64 * KPL], float* __restrict__ out, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < KPL; ++i)
    if (lane + 64 * i < K) red[wave][lane + 64 * i] = acc[i];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += SK_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) s += red[w][k];
    out[k] = s;
  }
}

// E = exp(scores/eps); partial[wg][k] = sum over this workgroup's rows of E[b][k]
template <int KPL>
__global__ __launch_bounds__(SK_THREADS) void sk_init_kernel(const float* __restrict__ scores, float* __restrict__ E,
                                                             float* __restrict__ partial, int B, int K, float eps, int rows_per_wg) {
  __shared__ float red[SK_WAVES][64 * KPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(B, r0 + rows_per_wg);
  float acc[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) acc[i] = 0.f;
  for (int b = r0 + wave; b < r1; b += SK_WAVES) {
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
      const int k = lane + 64 * i;
      if (k < K) {
        const float e = expf(scores[(long long)b * K + k] / eps);
        E[(long long)b * K + k] = e;
        acc[i] += e;
      }
    }
  }
  fold_waves<KPL>(acc, red, partial + (long long)blockIdx.x * K, K);
}

// The same from the positive matrix itself, Q[k][b] = exp(scores[b][k] / eps) as my_utils.sinkhorn receives it ([K][B], :246):
// E[b][k] = Q[k][b] (a transposing copy through LDS, 64 x 64 tiles), no exp / log round trip.
__global__ __launch_bounds__(256) void sk_transpose_kernel(const float* __restrict__ Q, float* __restrict__ E, int B, int K) {
  __shared__ float t[64][65];
  const int b0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  for (int r = r4; r < 64; r += 4)
    if (k0 + r < K && b0 + c < B) t[r][c] = Q[(long long)(k0 + r) * B + b0 + c];
  __syncthreads();
  for (int r = r4; r < 64; r += 4)
    if (b0 + r < B && k0 + c < K) E[(long long)(b0 + r) * K + k0 + c] = t[c][r];
}

template <int KPL>
__global__ __launch_bounds__(SK_THREADS) void sk_init_from_e_kernel(const float* __restrict__ E, float* __restrict__ partial, int B, int K,
                                                                    int rows_per_wg) {
  __shared__ float red[SK_WAVES][64 * KPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(B, r0 + rows_per_wg);
  float acc[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) acc[i] = 0.f;
  for (int b = r0 + wave; b < r1; b += SK_WAVES) {
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
      const int k = lane + 64 * i;
      if (k < K) acc[i] += E[(long long)b * K + k];
    }
  }
  fold_waves<KPL>(acc, red, partial + (long long)blockIdx.x * K, K);
}

// (Round 6 tried to cut this kernel's two latency chains - the fold of the nwg_in x K partials spread over all 1024 threads in four slices
// of the workgroup list, and four rows of a wave in flight instead of two: 88.9 us per solve against 79.8 for this form (one box,
// tools/sk_time.py: graph replays of the 12 launches, device time); each variant alone was slower too (profiles/r06_sinkhorn.txt).  An
// iteration is ~ 6.7 us of which the fold of 64 partials is ~ 0.8; the rest is the launch of 64 - 96 sixteen-wave workgroups, two passes
// over L2-resident rows and the drain.)
// One Sinkhorn iteration (row step + column step) or, with LAST, the final column normalisation + output.
template <bool LAST, int KPL>
__global__ __launch_bounds__(SK_THREADS) void sk_iter_kernel(const float* __restrict__ E, const float* __restrict__ partial_in,
                                                             float* __restrict__ partial_out, float* __restrict__ q_out, int B, int K,
                                                             int nwg_in, int rows_per_wg, int row0, int rows_out, int uniform_a, int b_norm) {
  __shared__ float a_s[64 * KPL];
  __shared__ float red[SK_WAVES][64 * KPL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // row step: a_k = (1/K) / u_k, u_k folded from the previous launch's partial sums in fixed order
  for (int k = threadIdx.x; k < K; k += SK_THREADS) {
    float a = 1.0f;
    if (!uniform_a) {
      // the fold of the previous launch's partials, in workgroup order (fixed: run-to-run bits) - the loads of 32 partials in flight at
      // a time (round 6: with 16 the fold of 64 partials was four dependent batches of L2 latency at the head of every launch)
      // (whole batches only: a zero-padded or clamped last batch measured 20 % slower than this - the workgroup counts are multiples of
      // SK_FOLD_BATCH, sk_wgs; other counts finish in batches of 16 and singly)
      float u = 0.f;
      int w = 0;
      for (; w + SK_FOLD_BATCH <= nwg_in; w += SK_FOLD_BATCH) {
        float v[SK_FOLD_BATCH];
#pragma unroll
        for (int j = 0; j < SK_FOLD_BATCH; ++j) v[j] = partial_in[(long long)(w + j) * K + k];
#pragma unroll
        for (int j = 0; j < SK_FOLD_BATCH; ++j) u += v[j];
      }
      for (; w + 16 <= nwg_in; w += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = partial_in[(long long)(w + j) * K + k];
#pragma unroll
        for (int j = 0; j < 16; ++j) u += v[j];
      }
      for (; w < nwg_in; ++w) u += partial_in[(long long)w * K + k];
      a = (1.0f / (float)K) / u;
    }
    a_s[k] = a;
  }
  __syncthreads();
  float a[KPL], acc[KPL];
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const int k = lane + 64 * i;
    a[i] = (k < K) ? a_s[k] : 0.f;
    acc[i] = 0.f;
  }
  int r0, r1;
  if (LAST) {
    r0 = row0 + blockIdx.x * rows_per_wg;
    r1 = min(row0 + rows_out, r0 + rows_per_wg);
  } else {
    r0 = blockIdx.x * rows_per_wg;
    r1 = min(B, r0 + rows_per_wg);
  }
  const float c = 1.0f / (float)b_norm;   // 1 / (B W): B rows here, b_norm rows over all ranks (my_utils.py:257)
  // two rows per wave in flight: their loads overlap each other's reduction latency
  for (int b = r0 + wave; b < r1; b += 2 * SK_WAVES) {
    const int b2 = b + SK_WAVES;
    const bool has2 = b2 < r1;
    float e[KPL], f[KPL];
    float t = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
      const int k = lane + 64 * i;
      e[i] = (k < K) ? E[(long long)b * K + k] : 0.f;
      f[i] = (k < K && has2) ? E[(long long)b2 * K + k] : 0.f;
      t += a[i] * e[i];
      t2 += a[i] * f[i];
    }
    t = wave_sum(t);
    t2 = wave_sum(t2);
    if (LAST) {
#pragma unroll
      for (int i = 0; i < KPL; ++i) {
        const int k = lane + 64 * i;
        if (k < K) {
          q_out[(long long)(b - row0) * K + k] = a[i] * e[i] / t;
          if (has2) q_out[(long long)(b2 - row0) * K + k] = a[i] * f[i] / t2;
        }
      }
    } else {
      const float bb = c / t;                 // column step
      const float bb2 = has2 ? c / t2 : 0.f;
#pragma unroll
      for (int i = 0; i < KPL; ++i) acc[i] += e[i] * bb + f[i] * bb2;
    }
  }
  if (LAST) return;
  fold_waves<KPL>(acc, red, partial_out + (long long)blockIdx.x * K, K);
}

// ---- the whole solve in ONE launch ----------------------------------------------------------------------------------------------
// SURVEY 2.4 k12's persistent form.  G workgroups (one per CU: 1024 threads and most of the LDS), each owning a run of rows whose
// E = exp(scores / eps) it keeps in LDS for the whole solve (6272 x 200 at C2: 37 workgroups); per iteration the only thing that crosses
// workgroups is the K-vector of row sums.
// Round 5 built the exchange from the expensive primitives (4-byte write-through stores, `s_waitcnt vmcnt(0)` + barrier, ONE arrival
// counter everybody polls, then the partials): ~ 8 us per iteration, 105 us per solve against 78 for a launch per iteration.  Round 6: the
// exchange is a TAGGED GRANULE all-gather (MI355X_MICROARCH.md "Valid forms": an aligned 8-byte {data, tag} written by ONE store is seen
// whole or not at all) - a workgroup's partial sum of column k travels as {float bits, iteration tag} in one 8-byte agent-scope store, a
// reader polls the granules it needs (all of its loads in flight, re-issuing only those whose tag is not this iteration's yet) and sums
// them in workgroup order once they are all there.  No counter, no fence, no wait for the stores, no barrier between publishing and
// gathering: one fabric write + one fabric read per iteration.  Tags are iteration + 1 and the granule buffers are zeroed by the launch
// function (a tag never matches memory that was not written by this solve); two granule buffers alternate - a workgroup publishes iteration
// i + 1 only after it has gathered iteration i, i.e. after EVERY workgroup has published i, i.e. after every workgroup has finished
// gathering i - 1 from the buffer that i + 1 overwrites.  The fold order is fixed (run-to-run bit equality); every spin is bounded (status
// word).  Needs every workgroup resident at once (G <= CUs: they wait for each other) and 2 G K granules in the workspace's partial region
// (G <= SK_MAXWG / 2); larger problems (the 8-rank global problem: 50176 rows) take the launch-per-iteration path.
struct SkpArgs {
  const float* scores;          // [B][K]
  unsigned long long* gran;     // [2][G][K] granules: float bits | (tag << 32)
  unsigned* status;             // [0]: 1 = a spin gave up (zeroed by the launch function)
  float* q_out;                 // [rows_out][K]
  int B, K, G, rows_per_wg, row0, rows_out, iters;
  float eps;
  int b_norm;
};
constexpr int SKP_LDS_FLOATS = 38400;   // 150 KB
__host__ __device__ inline int skp_kpad(int K) { return (K + 63) / 64 * 64; }
__host__ __device__ inline int skp_lds_rows(int K) { return (SKP_LDS_FLOATS - 64 * SK_KPL - SK_WAVES * skp_kpad(K)) / K; }

__global__ __launch_bounds__(SK_THREADS) void sk_persistent_kernel(SkpArgs g) {
  __shared__ float sm[SKP_LDS_FLOATS];
  const int K = g.K, KP = skp_kpad(K);
  float* a_s = sm;                              // [64 SK_KPL]
  float* red = sm + 64 * SK_KPL;                // [SK_WAVES][KP]
  float* El = red + SK_WAVES * KP;              // [rows_per_wg][K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wg = blockIdx.x;
  const int r0 = wg * g.rows_per_wg, r1 = min(g.B, r0 + g.rows_per_wg);
  auto e_ptr = [&](int b) -> float* { return El + (size_t)(b - r0) * K; };
  float acc[SK_KPL], a[SK_KPL];

  // fold the 16 waves' register partials (LDS) and publish the K sums of phase `ph` as granules
  auto publish = [&](int ph) {
#pragma unroll
    for (int i = 0; i < SK_KPL; ++i)
      if (lane + 64 * i < K) red[wave * KP + lane + 64 * i] = acc[i];
    __syncthreads();
    unsigned long long* out = g.gran + ((size_t)(ph & 1) * g.G + wg) * K;
    for (int k = threadIdx.x; k < K; k += SK_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) s += red[w * KP + k];
      const unsigned long long v = (unsigned long long)__float_as_uint(s) | ((unsigned long long)(unsigned)(ph + 1) << 32);
      __hip_atomic_store(out + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // a_k = (1 / K) / sum over the workgroups' phase-`ph` partials (fixed order), each taken when its granule carries this phase's tag
  auto gather = [&](int ph) {
    const unsigned tag = (unsigned)(ph + 1);
    const unsigned long long* in = g.gran + (size_t)(ph & 1) * g.G * K;
    const int NJ = SK_THREADS / KP;
    const int k = threadIdx.x % KP, j = threadIdx.x / KP;
    float u = 0.f;
    if (k < K && j < NJ) {
      // thread (k, j) takes the workgroups w = j, j + NJ, ... - sixteen granules in flight, re-polled until they are all this phase's
      for (int w0 = j; w0 < g.G; w0 += 16 * NJ) {
        float v[16];
        unsigned need = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          v[q] = 0.f;
          if (w0 + q * NJ < g.G) need |= 1u << q;
        }
        unsigned spins = 0;
        while (need) {
          unsigned long long raw[16];
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (need >> q & 1u) raw[q] = __hip_atomic_load(in + (size_t)(w0 + q * NJ) * K + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if ((need >> q & 1u) && (unsigned)(raw[q] >> 32) == tag) {
              v[q] = __uint_as_float((unsigned)raw[q]);
              need &= ~(1u << q);
            }
          if (need) {
            if (++spins > (1u << 22)) { __hip_atomic_store(g.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) u += v[q];
      }
    }
    __syncthreads();   // (red: publish's readers are done)
    if (j < NJ) red[j * KP + k] = u;
    __syncthreads();
    for (int k2 = threadIdx.x; k2 < K; k2 += SK_THREADS) {
      float t = 0.f;
      for (int j2 = 0; j2 < NJ; ++j2) t += red[j2 * KP + k2];
      a_s[k2] = (1.0f / (float)K) / t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SK_KPL; ++i) a[i] = (lane + 64 * i < K) ? a_s[lane + 64 * i] : 0.f;
  };

  // ---- phase 0: E = exp(scores / eps) for this workgroup's rows (kept in LDS), their column sums; four rows of a wave in flight
#pragma unroll
  for (int i = 0; i < SK_KPL; ++i) acc[i] = 0.f;
  for (int b = r0 + wave; b < r1; b += 4 * SK_WAVES) {
    float v[4][SK_KPL];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int bj = b + jj * SK_WAVES;
      const int bc = min(bj, r1 - 1);                   // (clamped, not predicated: the loads of a pass then issue back to back)
#pragma unroll
      for (int i = 0; i < SK_KPL; ++i) {
        const int k = lane + 64 * i;
        if (64 * i < K) v[jj][i] = g.scores[(long long)bc * K + min(k, K - 1)];
      }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int bj = b + jj * SK_WAVES;
      if (bj < r1) {
        float* er = e_ptr(bj);
#pragma unroll
        for (int i = 0; i < SK_KPL; ++i) {
          const int k = lane + 64 * i;
          if (64 * i < K && k < K) {
            const float e = expf(v[jj][i] / g.eps);
            er[k] = e;
            acc[i] += e;
          }
        }
      }
    }
  }
  if (g.iters > 0) publish(0);
  else __syncthreads();
  const float c = 1.0f / (float)g.b_norm;
  for (int it = 0; it < g.iters || it == 0; ++it) {
    if (g.iters > 0) gather(it);
    else {
#pragma unroll
      for (int i = 0; i < SK_KPL; ++i) a[i] = (lane + 64 * i < K) ? 1.0f : 0.f;
    }
    const bool last = it + 1 >= g.iters;
    if (last) {
      // the last row step's a, the final column normalisation, written for the requested rows this workgroup owns
      const int o0 = max(r0, g.row0), o1 = min(r1, g.row0 + g.rows_out);
      for (int b = o0 + wave; b < o1; b += SK_WAVES) {
        const float* er = e_ptr(b);
        float e[SK_KPL], t = 0.f;
#pragma unroll
        for (int i = 0; i < SK_KPL; ++i) {
          e[i] = (lane + 64 * i < K) ? er[lane + 64 * i] : 0.f;
          t += a[i] * e[i];
        }
        t = wave_sum(t);
#pragma unroll
        for (int i = 0; i < SK_KPL; ++i)
          if (lane + 64 * i < K) g.q_out[(long long)(b - g.row0) * K + lane + 64 * i] = a[i] * e[i] / t;
      }
      break;
    }
    // column step over this workgroup's rows (LDS), two rows per wave in flight; the row sums that follow
#pragma unroll
    for (int i = 0; i < SK_KPL; ++i) acc[i] = 0.f;
    for (int b = r0 + wave; b < r1; b += 2 * SK_WAVES) {
      const int b2 = b + SK_WAVES;
      const bool has2 = b2 < r1;
      const float* er = e_ptr(b);
      const float* fr = e_ptr(has2 ? b2 : b);
      float e[SK_KPL], f[SK_KPL], t = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < SK_KPL; ++i) {
        const int k = lane + 64 * i;
        e[i] = (k < K) ? er[k] : 0.f;
        f[i] = (k < K && has2) ? fr[k] : 0.f;
        t += a[i] * e[i];
        t2 += a[i] * f[i];
      }
      t = wave_sum(t);
      t2 = wave_sum(t2);
      const float bb = c / t, bb2 = has2 ? c / t2 : 0.f;
#pragma unroll
      for (int i = 0; i < SK_KPL; ++i) acc[i] += e[i] * bb + f[i] * bb2;
    }
    publish(it + 1);   // (red: the gather above ended with a barrier behind its last reader)
  }
}

// ---- cross entropy -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ scores, const int64_t* __restrict__ labels,
                                                 const float* __restrict__ row_weight, float* __restrict__ row_loss,
                                                 float* __restrict__ dscores, int rows, int K, float temp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float z[SK_KPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SK_KPL; ++i) {
    const int k = lane + 64 * i;
    z[i] = (k < K) ? scores[(long long)row * K + k] / temp : -INFINITY;
    mx = fmaxf(mx, z[i]);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SK_KPL; ++i) {
    z[i] = expf(z[i] - mx);  // exp(-inf) = 0 for the padding lanes
    s += z[i];
  }
  s = wave_sum(s);
  const int lab = (int)labels[row];
  const float zl = scores[(long long)row * K + lab] / temp;
  // reduction='none' * mask, then the mean over ALL rows (time_tuning.py:298-300); weight 1 without a mask
  const float wgt = row_weight ? row_weight[row] : 1.0f;
  if (lane == 0) row_loss[row] = ((mx + logf(s)) - zl) * wgt;
  if (dscores) {
    const float gscale = wgt / (temp * (float)rows);
#pragma unroll
    for (int i = 0; i < SK_KPL; ++i) {
      const int k = lane + 64 * i;
      if (k < K) dscores[(long long)row * K + k] = (z[i] / s - (k == lab ? 1.0f : 0.0f)) * gscale;
    }
  }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n);
}

}  // namespace tt

using namespace tt;

extern "C" size_t tt_sinkhorn_workspace_bytes(int B_total, int K) {
  return ((size_t)B_total * K + 2ull * SK_MAXWG * K) * sizeof(float) + 256;   // E | two partial buffers | the persistent kernel's counter block
}

static int sinkhorn_impl(const float* scores, const float* Q, int q_rows_are_columns, float* q_out, int B_total, int K, int row0,
                         int rows_out, float eps, int iters, void* workspace, size_t workspace_bytes, tt_stream_t stream);

extern "C" int tt_sinkhorn(const float* scores, float* q_out, int B_total, int K, int row0, int rows_out, float eps,
                           int iters, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(scores, "sinkhorn: null pointer");
  return sinkhorn_impl(scores, nullptr, 0, q_out, B_total, K, row0, rows_out, eps, iters, workspace, workspace_bytes, stream);
}

extern "C" int tt_sinkhorn_from_q(const float* Q, int transposed, float* q_out, int B_total, int K, int row0, int rows_out, int iters,
                                  void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(Q, "sinkhorn_from_q: null pointer");
  return sinkhorn_impl(nullptr, Q, transposed, q_out, B_total, K, row0, rows_out, 1.0f, iters, workspace, workspace_bytes, stream);
}

static int sinkhorn_impl(const float* scores, const float* Q, int q_rows_are_columns, float* q_out, int B_total, int K, int row0,
                         int rows_out, float eps, int iters, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(q_out && workspace, "sinkhorn: null pointer");
  TT_REQUIRE(B_total > 0 && K > 0 && K <= 64 * SK_KPL, "sinkhorn: need 0 < K <= %d (got %d)", 64 * SK_KPL, K);
  TT_REQUIRE(row0 >= 0 && rows_out > 0 && row0 + rows_out <= B_total, "sinkhorn: output rows [%d, %d) outside [0, %d)", row0,
             row0 + rows_out, B_total);
  TT_REQUIRE(iters >= 0 && eps > 0.f, "sinkhorn: bad iters/eps");
  TT_REQUIRE(workspace_bytes >= tt_sinkhorn_workspace_bytes(B_total, K), "sinkhorn: workspace too small");
  hipStream_t s = as_stream(stream);
  float* Ews = static_cast<float*>(workspace);
  float* part[2] = {Ews + (size_t)B_total * K, Ews + (size_t)B_total * K + (size_t)SK_MAXWG * K};
  // E [B][K]: built in the workspace, or - when the caller already holds the positive matrix in that layout - read in place
  const float* E = (Q && q_rows_are_columns) ? Q : Ews;
  // ---- one persistent launch (sk_persistent_kernel): from scores, every workgroup resident at once with its rows of E in LDS, the
  // granule buffers inside the workspace's partial region.  Knob TT_SK_PERSIST: 0 never, 1 whenever it applies.
  if (!Q && tuning_knob(KNOB_SK_PERSIST) != 0) {
    const int ncu = device_cu_count();
    const int cap = skp_lds_rows(K);
    static const int rows_env = [] { const char* e = getenv("TT_SKP_ROWS"); return e ? atoi(e) : 0; }();   // tuning aid
    int rows = rows_env > 0 ? rows_env : cap;
    if (rows > cap) rows = cap;
    const int G = rows > 0 ? (B_total + rows - 1) / rows : ncu + 1;
    if (G <= ncu && G <= SK_MAXWG / 2) {
      rows = (B_total + G - 1) / G;             // even shares
      unsigned long long* gran = reinterpret_cast<unsigned long long*>(part[0]);
      unsigned* status = reinterpret_cast<unsigned*>(Ews + (size_t)B_total * K + 2ull * SK_MAXWG * K);
      // tags start at 1: zeroed granules never match.  The status word sits right behind the partial region: one memset for both
      if (hipMemsetAsync(gran, 0, 2ull * SK_MAXWG * K * sizeof(float) + 16, s) != hipSuccess) { set_error("sinkhorn: hipMemsetAsync failed"); return TT_ELAUNCH; }
      SkpArgs a{scores, gran, status, q_out, B_total, K, G, rows, row0, rows_out, iters, eps, B_total};
      hipLaunchKernelGGL(sk_persistent_kernel, dim3(G), dim3(SK_THREADS), 0, s, a);
      TT_CHECK_LAUNCH("sinkhorn (persistent)");
      return TT_OK;
    }
  }
  const int wgs = sk_wgs(B_total);
  const int rpw = (B_total + wgs - 1) / wgs;
  if (Q) {
    if (!q_rows_are_columns) hipLaunchKernelGGL(sk_transpose_kernel, dim3((B_total + 63) / 64, (K + 63) / 64), dim3(256), 0, s, Q, Ews, B_total, K);
    SK_LAUNCH_KPL(sk_init_from_e_kernel<4>, sk_init_from_e_kernel<8>, dim3(wgs), E, part[0], B_total, K, rpw);
  } else {
    SK_LAUNCH_KPL(sk_init_kernel<4>, sk_init_kernel<8>, dim3(wgs), scores, Ews, part[0], B_total, K, eps, rpw);
  }
  int cur = 0;
  for (int it = 0; it + 1 < iters; ++it) {  // iterations 1 .. iters-1 (each prepares the next row step)
    SK_LAUNCH_KPL((sk_iter_kernel<false, 4>), (sk_iter_kernel<false, 8>), dim3(wgs), E, part[cur], part[cur ^ 1], (float*)nullptr, B_total, K, wgs,
                  rpw, 0, 0, 0, B_total);
    cur ^= 1;
  }
  // last iteration's row step + column normalisation, written straight to q for the requested rows
  // (the output launch leaves no partials for anybody to fold: as many workgroups as give a wave its two rows - SK_LAST_WIDE, round 6)
#ifndef SK_LAST_WIDE
#define SK_LAST_WIDE 1
#endif
  int owgs = sk_wgs(rows_out);
  if (SK_LAST_WIDE && rows_out <= 8192) {   // (6 272 rows: 71.1 -> 69.9 us per solve; 12 544: 83.2 -> 84.9; 50 176: no change)
    owgs = (rows_out + 2 * SK_WAVES - 1) / (2 * SK_WAVES);
    owgs = owgs > SK_MAXWG ? SK_MAXWG : (owgs < 1 ? 1 : owgs);   // (one workgroup per CU at most: 1 024 of them measured slower at 50 176 rows)
  }
  const int orpw = (rows_out + owgs - 1) / owgs;
  SK_LAUNCH_KPL((sk_iter_kernel<true, 4>), (sk_iter_kernel<true, 8>), dim3(owgs), E, part[cur], (float*)nullptr, q_out, B_total, K, wgs, orpw,
                row0, rows_out, iters == 0 ? 1 : 0, B_total);
  TT_CHECK_LAUNCH("sinkhorn");
  return TT_OK;
}

// ---- The reference's own distributed form (my_utils.py:250-272): the columns (patches) stay on their rank, only the K row sums are
// all-reduced, once per iteration.  One rank's share of a solve in steps; the caller all-reduces u between them:
//   begin   E = exp(scores / eps) (kept in the workspace), u_out[k] = sum over the LOCAL rows of E[b][k]
//   step    a = (1/K) / u_in (u_in: the all-reduced row sums), column step with c = 1 / B_total, u_out = the local row sums that follow
//   end     the last row step (u_in null: none, the 0-iteration case) + the final column normalisation -> q [B_loc][K]
// A solve of `iters` iterations = begin, (all-reduce, step) x (iters - 1), all-reduce, end: `iters` all-reduces of K floats (the
// reference issues two more: the total mass, which cancels in the first row step, and a last row-sum nobody reads).
__global__ __launch_bounds__(256) void sk_fold_kernel(const float* __restrict__ partial, float* __restrict__ u, int nwg, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;   // fixed order; the loads in whole batches of SK_FOLD_BATCH (sk_iter_kernel), then of 16, then singly
  int w = 0;
  for (; w + SK_FOLD_BATCH <= nwg; w += SK_FOLD_BATCH) {
    float v[SK_FOLD_BATCH];
#pragma unroll
    for (int j = 0; j < SK_FOLD_BATCH; ++j) v[j] = partial[(long long)(w + j) * K + k];
#pragma unroll
    for (int j = 0; j < SK_FOLD_BATCH; ++j) s += v[j];
  }
  for (; w + 16 <= nwg; w += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = partial[(long long)(w + j) * K + k];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
  }
  for (; w < nwg; ++w) s += partial[(long long)w * K + k];
  u[k] = s;
}

extern "C" size_t tt_sinkhorn_local_workspace_bytes(int B_loc, int K) { return tt_sinkhorn_workspace_bytes(B_loc, K); }

static int sk_local_check(const void* workspace, size_t workspace_bytes, int B_loc, int K) {
  TT_REQUIRE(workspace, "sinkhorn_local: null workspace");
  TT_REQUIRE(B_loc > 0 && K > 0 && K <= 64 * SK_KPL, "sinkhorn_local: need 0 < K <= %d (got %d), B_loc > 0", 64 * SK_KPL, K);
  TT_REQUIRE(workspace_bytes >= tt_sinkhorn_local_workspace_bytes(B_loc, K), "sinkhorn_local: workspace too small");
  return TT_OK;
}

extern "C" int tt_sinkhorn_local_begin(const float* scores, float* u_out, int B_loc, int K, float eps, void* workspace, size_t workspace_bytes,
                                       tt_stream_t stream) {
  TT_REQUIRE(scores && u_out && eps > 0.f, "sinkhorn_local_begin: null pointer / bad eps");
  if (const int rc = sk_local_check(workspace, workspace_bytes, B_loc, K)) return rc;
  hipStream_t s = as_stream(stream);
  float* E = static_cast<float*>(workspace);
  float* part = E + (size_t)B_loc * K;
  const int wgs = sk_wgs(B_loc), rpw = (B_loc + wgs - 1) / wgs;
  SK_LAUNCH_KPL(sk_init_kernel<4>, sk_init_kernel<8>, dim3(wgs), scores, E, part, B_loc, K, eps, rpw);
  hipLaunchKernelGGL(sk_fold_kernel, dim3((K + 255) / 256), dim3(256), 0, s, part, u_out, wgs, K);
  TT_CHECK_LAUNCH("sinkhorn_local_begin");
  return TT_OK;
}

extern "C" int tt_sinkhorn_local_step(const float* u_in, float* u_out, int B_loc, int B_total, int K, void* workspace, size_t workspace_bytes,
                                      tt_stream_t stream) {
  TT_REQUIRE(u_in && u_out && B_total >= B_loc, "sinkhorn_local_step: null pointer / B_total < B_loc");
  if (const int rc = sk_local_check(workspace, workspace_bytes, B_loc, K)) return rc;
  hipStream_t s = as_stream(stream);
  float* E = static_cast<float*>(workspace);
  float* part = E + (size_t)B_loc * K;
  const int wgs = sk_wgs(B_loc), rpw = (B_loc + wgs - 1) / wgs;
  SK_LAUNCH_KPL((sk_iter_kernel<false, 4>), (sk_iter_kernel<false, 8>), dim3(wgs), E, u_in, part, (float*)nullptr, B_loc, K, 1, rpw, 0, 0, 0, B_total);
  hipLaunchKernelGGL(sk_fold_kernel, dim3((K + 255) / 256), dim3(256), 0, s, part, u_out, wgs, K);
  TT_CHECK_LAUNCH("sinkhorn_local_step");
  return TT_OK;
}

extern "C" int tt_sinkhorn_local_end(const float* u_in, float* q_out, int B_loc, int rows_out, int K, void* workspace, size_t workspace_bytes,
                                     tt_stream_t stream) {
  TT_REQUIRE(q_out && rows_out > 0 && rows_out <= B_loc, "sinkhorn_local_end: null pointer / rows_out outside (0, B_loc]");
  if (const int rc = sk_local_check(workspace, workspace_bytes, B_loc, K)) return rc;
  hipStream_t s = as_stream(stream);
  const float* E = static_cast<const float*>(workspace);
  const int owgs = sk_wgs(rows_out), orpw = (rows_out + owgs - 1) / owgs;
  SK_LAUNCH_KPL((sk_iter_kernel<true, 4>), (sk_iter_kernel<true, 8>), dim3(owgs), E, u_in, (float*)nullptr, q_out, B_loc, K, 1, orpw, 0, rows_out,
                u_in ? 0 : 1, B_loc);
  TT_CHECK_LAUNCH("sinkhorn_local_end");
  return TT_OK;
}

extern "C" size_t tt_ce_workspace_bytes(int rows) { return (size_t)rows * sizeof(float); }

extern "C" int tt_ce_loss_fwd_bwd(const float* scores, const int64_t* labels, const float* row_weight, float* loss_out, float* dscores,
                                  int rows, int K, float temperature, void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(scores && labels && loss_out && workspace, "ce_loss: null pointer");
  TT_REQUIRE(rows > 0 && K > 0 && K <= 64 * SK_KPL && temperature > 0.f, "ce_loss: need 0 < K <= %d", 64 * SK_KPL);
  TT_REQUIRE(workspace_bytes >= tt_ce_workspace_bytes(rows), "ce_loss: workspace too small");
  hipStream_t s = as_stream(stream);
  float* row_loss = static_cast<float*>(workspace);
  hipLaunchKernelGGL(ce_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, scores, labels, row_weight, row_loss, dscores, rows, K, temperature);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, row_loss, loss_out, rows);
  TT_CHECK_LAUNCH("ce_loss");
  return TT_OK;
}

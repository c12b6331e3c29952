// Coarse entry points: tt_vit_forward, tt_mlp_head_forward, tt_scores_sinkhorn, tt_adamw_ema_step.
// Host code only: each one lays out its scratch in the caller's workspace and enqueues the op-level entry points of this
// library on the caller's stream - the same launches, in the same order, as timetuning_amd/engine.py issues one ctypes call at a
// time, so the results are bit-identical to the fine-grained sequence.  Nothing here allocates, synchronises or keeps state.
#include "common.hpp"

namespace tt {

struct Carver {   // hands out 256-byte aligned pieces of the caller's workspace
  unsigned char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<unsigned char*>(p)) {}
  void* take(size_t bytes) {
    void* r = base ? base + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return r;
  }
};

struct VitScratch {
  void* h;        // LayerNorm output: fp32 [M, D] or bf16 planes [P, M, D]
  void* big;      // qkv | attention output (| its planes), overlaid by the MLP activation
  void* ks;       // planes > 0: the K-split block of the persistent GEMMs (common.hpp KsplitWs), directly behind `big`
  size_t ks_bytes;
  size_t bytes;
  size_t qkv_bytes, att_bytes;
};

static VitScratch carve_vit(void* ws, long long M, int D, int hidden, int planes) {
  Carver c(ws);
  VitScratch s;
  const size_t P = planes > 0 ? planes : 0;
  s.h = c.take(P ? P * M * D * 2 : (size_t)M * D * 4);
  // attention phase: qkv fp32 [M, 3D] (bf16 in the planes = 1 fast path: fits), att fp32 [M, D], att planes [P, M, D]
  s.qkv_bytes = ((size_t)M * 3 * D * 4 + 255) / 256 * 256;
  s.att_bytes = ((size_t)M * D * 4 + 255) / 256 * 256;
  const size_t attn_phase = s.qkv_bytes + s.att_bytes + (P ? (P * M * D * 2 + 255) / 256 * 256 : 0);
  const size_t mlp_phase = P ? P * M * hidden * 2 : (size_t)M * hidden * 4;
  s.big = c.take(attn_phase > mlp_phase ? attn_phase : mlp_phase);
  s.ks_bytes = P ? ksplit_ws_bytes() : 0;
  s.ks = P ? c.take(s.ks_bytes) : nullptr;
  s.bytes = c.off;
  return s;
}

}  // namespace tt

using namespace tt;

extern "C" size_t tt_vit_forward_workspace_bytes(int F, int N, int D, int hidden, int planes) {
  if (F <= 0 || N <= 0 || D <= 0 || hidden <= 0 || planes < 0) return 0;
  return carve_vit(nullptr, (long long)F * N, D, hidden, planes).bytes;
}

#define TT_FORWARD(call)        \
  do {                          \
    const int rc__ = (call);    \
    if (rc__ != TT_OK) return rc__; \
  } while (0)

extern "C" int tt_vit_forward(const tt_vit_params* p, const float* img, const int32_t* frame_map, int F, int C, int H, int W,
                              float* tokens, float* normed, int drop_cls, float* last_qkv, float* last_probs, void* workspace,
                              size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(p && tokens, "vit_forward: null pointer");
  TT_REQUIRE(F > 0 && p->dim > 0 && p->heads > 0 && p->hidden > 0 && p->patch > 0 && p->n_blocks >= 0, "vit_forward: bad shape");
  TT_REQUIRE(H % p->patch == 0 && W % p->patch == 0, "vit_forward: %d x %d input is not a multiple of the patch size %d", H, W, p->patch);
  TT_REQUIRE(p->dim % p->heads == 0, "vit_forward: dim %d not divisible by %d heads", p->dim, p->heads);
  TT_REQUIRE(p->n_blocks == 0 || p->blocks, "vit_forward: null block table");
  TT_REQUIRE(p->planes >= 0 && p->planes <= 3, "vit_forward: planes must be 0, 1, 3 (bf16 planes) or 2 (fp16 pairs) (got %d)", p->planes);
  TT_REQUIRE(p->precision >= TT_PRECISION_F32 && p->precision <= TT_PRECISION_BF16, "vit_forward: precision must be 0, 1 or 2 (got %d)", p->precision);
  TT_REQUIRE(p->planes == 0 || p->dim % 64 == 0, "vit_forward: the plane path needs dim %% 64 == 0 (got %d)", p->dim);
  TT_REQUIRE(p->planes != 2 || p->hidden % 64 == 0, "vit_forward: the pair path needs hidden %% 64 == 0 (got %d)", p->hidden);
  const int D = p->dim, hd = D / p->heads, P = p->planes;
  const int N = 1 + (H / p->patch) * (W / p->patch);
  const long long M = (long long)F * N;
  TT_REQUIRE(M * 3 * D < (1ll << 31), "vit_forward: F * N * 3 D exceeds the int range of the op entry points");
  const VitScratch s = carve_vit(workspace, M, D, p->hidden, P);
  TT_REQUIRE(p->n_blocks == 0 || (workspace && workspace_bytes >= s.bytes), "vit_forward: workspace too small (%zu < %zu)", workspace_bytes,
             s.bytes);
  const float scale = 1.0f / sqrtf((float)hd);
  int* const rf = p->range_flag;
  void* const ks = (workspace && workspace_bytes >= s.bytes) ? s.ks : nullptr;   // (p->n_blocks == 0 without a workspace: no K-split)
  const size_t ksb = ks ? s.ks_bytes : 0;
  if (ks) TT_FORWARD(tt_linear_ksplit_workspace_init(ks, ksb, stream));           // the counters: zero before the first launch that counts

  if (img) {
    TT_REQUIRE(p->patch_w && p->patch_b && p->cls && p->pos, "vit_forward: null patch-embedding parameter");
    // the bf16 path (planes == 1) embeds the patches on bf16 operands too where the plane GEMM's shape rules hold and the scratch
    // (the blocks' qkv / MLP region, idle now) holds the im2col rows; anything else: the fp32 conv
    const int Kp = C * p->patch * p->patch;
    // the patch-embedding entry points take [im2col rows | K-split block] as ONE workspace: the rows are placed at the END of `big`, so that
    // the K-split block behind it is theirs too
    const size_t a_bytes = tt_patch_embed_planes_workspace_bytes(F, C, H, W, p->patch), q_bytes = tt_patch_embed_pairs_workspace_bytes(F, C, H, W, p->patch);
    // (C P P <= 9 D: the rows then fit the attention phase of the scratch, 18 M D bytes at planes == 1, whatever the MLP width - a
    // rule on shapes alone, so that callers sequencing the op-level entry points themselves can make the same decision)
    if (P == 1 && p->patch_wp && workspace && workspace_bytes >= s.bytes && p->patch % 4 == 0 && W % 4 == 0 && Kp % 64 == 0 && D % 64 == 0 &&
        Kp <= 9 * D) {
      TT_FORWARD(tt_patch_embed_fwd_planes(img, frame_map, p->patch_wp, p->patch_b, p->cls, p->pos, tokens, F, C, H, W, p->patch, D,
                                           static_cast<unsigned char*>(s.ks) - (a_bytes - s.ks_bytes), a_bytes, stream));
    } else if (P == 2 && p->patch_wp && workspace && workspace_bytes >= s.bytes && p->patch % 4 == 0 && W % 4 == 0 && Kp % 32 == 0 &&
               D % 64 == 0 && Kp <= 3 * D) {
      // the pair path likewise: the im2col rows in pairs (4 bytes per element) fit the qkv region of the scratch when C P P <= 3 D
      TT_FORWARD(tt_patch_embed_fwd_pairs(img, frame_map, p->patch_wp, p->patch_b, p->cls, p->pos, tokens, F, C, H, W, p->patch, D,
                                          static_cast<unsigned char*>(s.ks) - (q_bytes - s.ks_bytes), q_bytes, rf, stream));
    } else {
      TT_FORWARD(tt_patch_embed_fwd(img, frame_map, p->patch_w, p->patch_b, p->cls, p->pos, tokens, F, C, H, W, p->patch, D, stream));
    }
  }
  for (int i = 0; i < p->n_blocks; ++i) {
    const tt_vit_block_params& b = p->blocks[i];
    const bool last = i == p->n_blocks - 1;
    float* probs = last ? last_probs : nullptr;
    float* qkv_out = last ? last_qkv : nullptr;
    unsigned char* big = static_cast<unsigned char*>(s.big);
    if (P == 0) {
      float* h = static_cast<float*>(s.h);
      float* qkv = qkv_out ? qkv_out : reinterpret_cast<float*>(big);
      float* att = reinterpret_cast<float*>(big + s.qkv_bytes);
      float* act = reinterpret_cast<float*>(big);
      TT_FORWARD(tt_layernorm_fwd(tokens, b.norm1_w, b.norm1_b, h, nullptr, nullptr, (int)M, D, 1e-6f, 0, stream));
      TT_FORWARD(tt_linear_fwd(h, b.qkv_w, b.qkv_b, nullptr, qkv, nullptr, (int)M, 3 * D, D, 0, p->precision, stream));
      TT_FORWARD(tt_attention_fwd(qkv, att, nullptr, probs, F, N, p->heads, hd, scale, stream));
      TT_FORWARD(tt_linear_fwd(att, b.proj_w, b.proj_b, tokens, tokens, nullptr, (int)M, D, D, 0, p->precision, stream));
      TT_FORWARD(tt_layernorm_fwd(tokens, b.norm2_w, b.norm2_b, h, nullptr, nullptr, (int)M, D, 1e-6f, 0, stream));
      TT_FORWARD(tt_linear_fwd(h, b.fc1_w, b.fc1_b, nullptr, act, nullptr, (int)M, p->hidden, D, 1, p->precision, stream));
      TT_FORWARD(tt_linear_fwd(act, b.fc2_w, b.fc2_b, tokens, tokens, nullptr, (int)M, D, p->hidden, 0, p->precision, stream));
      continue;
    }
    // bf16-plane / fp16-pair operands: every Linear reads what its producer wrote; the residual stream stays fp32, in place
    TT_REQUIRE(b.qkv_wp && b.proj_wp && b.fc1_wp && b.fc2_wp, "vit_forward: block %d has no weight planes", i);
    const long long MD = M * D;
    void* hp = s.h;
    unsigned char* att_region = big + s.qkv_bytes;
    void* attp = big + s.qkv_bytes + s.att_bytes;
    void* actp = big;
    if (P == 2) {   // the "f16x3" mode: fp16 pairs (4 bytes per element, the layout of 2 planes x 2 bytes)
      float* qkv = qkv_out ? qkv_out : reinterpret_cast<float*>(big);
      float* att = reinterpret_cast<float*>(att_region);
      TT_FORWARD(tt_layernorm_fwd_pairs(tokens, b.norm1_w, b.norm1_b, hp, nullptr, nullptr, (int)M, D, 1e-6f, 0, rf, stream));
      if (!qkv_out && !probs && hd == 64) {
        // qkv in pairs [M][2 x 3 D] (the bytes of the fp32 qkv region), the pair attention kernel, its output in pairs (the fp32 att region)
        TT_FORWARD(tt_linear_fwd_pairs(hp, b.qkv_wp, b.qkv_b, nullptr, nullptr, nullptr, big, (int)M, 3 * D, D, 0, ks, ksb, rf, stream));
        TT_FORWARD(tt_attention_fwd_pairs(big, att_region, nullptr, nullptr, F, N, p->heads, hd, scale, stream));
        TT_FORWARD(tt_linear_fwd_pairs(att_region, b.proj_wp, b.proj_b, tokens, tokens, nullptr, nullptr, (int)M, D, D, 0, ks, ksb, rf, stream));
      } else {
        TT_FORWARD(tt_linear_fwd_pairs(hp, b.qkv_wp, b.qkv_b, nullptr, qkv, nullptr, nullptr, (int)M, 3 * D, D, 0, ks, ksb, rf, stream));
        TT_FORWARD(tt_attention_fwd(qkv, att, nullptr, probs, F, N, p->heads, hd, scale, stream));
        TT_FORWARD(tt_split_pairs(att, attp, MD, rf, stream));
        TT_FORWARD(tt_linear_fwd_pairs(attp, b.proj_wp, b.proj_b, tokens, tokens, nullptr, nullptr, (int)M, D, D, 0, ks, ksb, rf, stream));
      }
      TT_FORWARD(tt_layernorm_fwd_pairs(tokens, b.norm2_w, b.norm2_b, hp, nullptr, nullptr, (int)M, D, 1e-6f, 0, rf, stream));
      TT_FORWARD(tt_linear_fwd_pairs(hp, b.fc1_wp, b.fc1_b, nullptr, nullptr, nullptr, actp, (int)M, p->hidden, D, 1, ks, ksb, rf, stream));
      TT_FORWARD(tt_linear_fwd_pairs(actp, b.fc2_wp, b.fc2_b, tokens, tokens, nullptr, nullptr, (int)M, D, p->hidden, 0, ks, ksb, rf, stream));
      continue;
    }
    TT_FORWARD(tt_layernorm_fwd_planes(tokens, b.norm1_w, b.norm1_b, hp, MD, P, nullptr, nullptr, (int)M, D, 1e-6f, 0, stream));
    const void* proj_in;
    if (P == 1 && !qkv_out && !probs && N <= 256 && hd == 64) {
      void* qkvb = big;   // bf16 [M, 3D]
      TT_FORWARD(tt_linear_fwd_planes(hp, MD, b.qkv_wp, 3ll * D * D, 1, b.qkv_b, nullptr, nullptr, nullptr, qkvb, M * 3 * D, 1, (int)M, 3 * D, D,
                                      0, ks, ksb, stream));
      TT_FORWARD(tt_attention_fwd_bf16(qkvb, att_region, F, N, p->heads, hd, scale, stream));
      proj_in = att_region;
    } else {
      float* qkv = qkv_out ? qkv_out : reinterpret_cast<float*>(big);
      float* att = reinterpret_cast<float*>(att_region);
      TT_FORWARD(tt_linear_fwd_planes(hp, MD, b.qkv_wp, 3ll * D * D, P, b.qkv_b, nullptr, qkv, nullptr, nullptr, 0, 0, (int)M, 3 * D, D, 0,
                                      ks, ksb, stream));
      TT_FORWARD(tt_attention_fwd(qkv, att, nullptr, probs, F, N, p->heads, hd, scale, stream));
      TT_FORWARD(tt_split_planes(att, attp, MD, P, MD, stream));
      proj_in = attp;
    }
    TT_FORWARD(tt_linear_fwd_planes(proj_in, MD, b.proj_wp, (long long)D * D, P, b.proj_b, tokens, tokens, nullptr, nullptr, 0, 0, (int)M, D, D, 0,
                                    ks, ksb, stream));
    TT_FORWARD(tt_layernorm_fwd_planes(tokens, b.norm2_w, b.norm2_b, hp, MD, P, nullptr, nullptr, (int)M, D, 1e-6f, 0, stream));
    TT_FORWARD(tt_linear_fwd_planes(hp, MD, b.fc1_wp, (long long)p->hidden * D, P, b.fc1_b, nullptr, nullptr, nullptr, actp, M * p->hidden, P,
                                    (int)M, p->hidden, D, 1, ks, ksb, stream));
    TT_FORWARD(tt_linear_fwd_planes(actp, M * p->hidden, b.fc2_wp, (long long)p->hidden * D, P, b.fc2_b, tokens, tokens, nullptr, nullptr, 0, 0,
                                    (int)M, D, p->hidden, 0, ks, ksb, stream));
  }
  if (normed) {
    TT_REQUIRE(p->norm_w && p->norm_b, "vit_forward: null final-norm parameter");
    if (drop_cls) TT_FORWARD(tt_layernorm_fwd(tokens, p->norm_w, p->norm_b, normed, nullptr, nullptr, F * (N - 1), D, 1e-6f, N, stream));
    else TT_FORWARD(tt_layernorm_fwd(tokens, p->norm_w, p->norm_b, normed, nullptr, nullptr, (int)M, D, 1e-6f, 0, stream));
  }
  return TT_OK;
}

// ---- projection head -----------------------------------------------------------------------------------------------------
static int head_max_width(const tt_linear_params* layers, int n_layers) {
  int w = 0;
  for (int i = 0; i + 1 < n_layers; ++i) w = layers[i].out_features > w ? layers[i].out_features : w;
  return w;
}

extern "C" size_t tt_mlp_head_forward_workspace_bytes(int M, const tt_linear_params* layers, int n_layers) {
  if (M <= 0 || !layers || n_layers <= 1) return 0;
  const size_t one = ((size_t)M * head_max_width(layers, n_layers) * 4 + 255) / 256 * 256;
  return n_layers > 2 ? 2 * one : one;
}

extern "C" int tt_mlp_head_forward(const float* x, int M, const tt_linear_params* layers, int n_layers, float* out, int precision,
                                   void* workspace, size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(x && layers && out, "mlp_head_forward: null pointer");
  TT_REQUIRE(M > 0 && n_layers > 0, "mlp_head_forward: bad shape");
  for (int i = 0; i < n_layers; ++i) {
    TT_REQUIRE(layers[i].w && layers[i].out_features > 0 && layers[i].in_features > 0, "mlp_head_forward: bad layer %d", i);
    TT_REQUIRE(i == 0 || layers[i].in_features == layers[i - 1].out_features, "mlp_head_forward: layer %d takes %d features, layer %d gives %d", i,
               layers[i].in_features, i - 1, layers[i - 1].out_features);
  }
  const size_t need = tt_mlp_head_forward_workspace_bytes(M, layers, n_layers);
  TT_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "mlp_head_forward: workspace too small (%zu < %zu)", workspace_bytes, need);
  float* buf[2] = {static_cast<float*>(workspace), nullptr};
  if (n_layers > 2) buf[1] = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + need / 2);
  const float* cur = x;
  for (int i = 0; i < n_layers; ++i) {
    const bool last = i == n_layers - 1;
    float* dst = last ? out : buf[i & 1];
    TT_FORWARD(tt_linear_fwd(cur, layers[i].w, layers[i].b, nullptr, dst, nullptr, M, layers[i].out_features, layers[i].in_features, last ? 0 : 1,
                             precision, stream));
    cur = dst;
  }
  return TT_OK;
}

// ---- scores + assignment ---------------------------------------------------------------------------------------------------
extern "C" size_t tt_scores_sinkhorn_workspace_bytes(int B, int queue_rows, int K, int dim) {
  if (B <= 0 || queue_rows < 0 || K <= 0 || dim <= 0) return 0;
  const size_t zn = ((size_t)(B + queue_rows) * dim * 4 + 255) / 256 * 256;
  return zn + tt_sinkhorn_workspace_bytes(B + queue_rows, K);
}

extern "C" int tt_scores_sinkhorn(const float* z, int B, const float* queue, int queue_rows, const float* prototypes, int K, int dim,
                                  float* scores, float* q_out, int rows_out, float eps, int iters, int precision, void* workspace,
                                  size_t workspace_bytes, tt_stream_t stream) {
  TT_REQUIRE(z && prototypes && scores && q_out && workspace, "scores_sinkhorn: null pointer");
  TT_REQUIRE(B > 0 && K > 0 && dim > 0 && queue_rows >= 0, "scores_sinkhorn: bad shape");
  if (!queue) queue_rows = 0;
  const int total = B + queue_rows;
  TT_REQUIRE(rows_out > 0 && rows_out <= total, "scores_sinkhorn: rows_out %d outside [1, %d]", rows_out, total);
  const size_t need = tt_scores_sinkhorn_workspace_bytes(B, queue_rows, K, dim);
  TT_REQUIRE(workspace_bytes >= need, "scores_sinkhorn: workspace too small (%zu < %zu)", workspace_bytes, need);
  float* zn = static_cast<float*>(workspace);
  const size_t zn_bytes = ((size_t)total * dim * 4 + 255) / 256 * 256;
  void* sk_ws = static_cast<unsigned char*>(workspace) + zn_bytes;
  TT_FORWARD(tt_l2norm_fwd(z, dim, zn, nullptr, B, dim, stream));
  TT_FORWARD(tt_linear_fwd(zn, prototypes, nullptr, nullptr, scores, nullptr, B, K, dim, 0, precision, stream));
  if (queue_rows) {   // time_tuning.py:207-211: the queue rows are scored the same way and take part in the assignment
    float* qn = zn + (size_t)B * dim;
    TT_FORWARD(tt_l2norm_fwd(queue, dim, qn, nullptr, queue_rows, dim, stream));
    TT_FORWARD(tt_linear_fwd(qn, prototypes, nullptr, nullptr, scores + (size_t)B * K, nullptr, queue_rows, K, dim, 0, precision, stream));
  }
  return tt_sinkhorn(scores, q_out, total, K, 0, rows_out, eps, iters, sk_ws, workspace_bytes - zn_bytes, stream);
}

// ---- optimizer step + prototype renormalisation + EMA teacher ---------------------------------------------------------------
extern "C" int tt_adamw_ema_step(const tt_adamw_tensor* tensors, int count, int step, float beta1, float beta2, float eps, float* prototypes,
                                 int K, int dim, float* teacher_flat, const float* student_flat, long long n_flat, float* teacher_prototypes,
                                 double momentum, tt_stream_t stream) {
  TT_REQUIRE(count >= 0 && (count == 0 || tensors), "adamw_ema_step: null tensor table");
  for (int i = 0; i < count; i += TT_MAX_TENSORS) {
    const int n = count - i < TT_MAX_TENSORS ? count - i : TT_MAX_TENSORS;
    TT_FORWARD(tt_adamw_step(tensors + i, n, step, beta1, beta2, eps, stream));
  }
  if (prototypes) TT_FORWARD(tt_normalize_rows_inplace(prototypes, K, dim, stream));
  if (teacher_flat || teacher_prototypes) {
    TT_REQUIRE(n_flat == 0 || (teacher_flat && student_flat), "adamw_ema_step: null parameter buffer");
    if (n_flat > 0) TT_FORWARD(tt_ema_update(teacher_flat, student_flat, n_flat, momentum, stream));
    if (teacher_prototypes) {
      TT_REQUIRE(prototypes, "adamw_ema_step: teacher prototypes without student prototypes");
      TT_FORWARD(tt_ema_update(teacher_prototypes, prototypes, (long long)K * dim, momentum, stream));
      TT_FORWARD(tt_normalize_rows_inplace(teacher_prototypes, K, dim, stream));
    }
  }
  return TT_OK;
}

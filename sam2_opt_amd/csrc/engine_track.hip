// Per-frame tracking stages: memory attention, two-way mask decoder + SAM heads, memory encoder.
#include "engine.h"

static const float LOG2E = 1.4426950408889634f;

static inline int ceil32(int v) { return (v + 31) / 32 * 32; }

#ifdef SAM2MI_EXPERIMENTAL
// Launch-count experiment ("what bounds the overlapped tracking path"): SAM2MI_DUMMY_LAUNCHES=n adds n dependent
// kernels of the shape of the small tracking GEMMs (256 workgroups x 256 threads, 32 KB of LDS, ~2 us of work) to every frame.
namespace {
__global__ __launch_bounds__(256) void dummy_dependent_kernel(float* p) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = p[blockIdx.x & 1023];
  __syncthreads();
  float v = sm[(threadIdx.x + 1) & 255];
  for (int i = 0; i < 200; ++i) v = fmaf(v, 1.0001f, 0.5f);
  if (v == 123.456f) p[blockIdx.x & 1023] = v;
}
}  // namespace
static int dummy_launches(sam2mi_ctx* ctx, hipStream_t s) {
  static const int n = getenv("SAM2MI_DUMMY_LAUNCHES") ? atoi(getenv("SAM2MI_DUMMY_LAUNCHES")) : 0;
  for (int i = 0; i < n; ++i) dummy_dependent_kernel<<<dim3(256), dim3(256), 32768, s>>>(ctx->t_x);
  if (n) CHK(hipGetLastError());
  return 0;
}
#endif


// MemoryAttention.inference_memory_attention_torch (modeling/memory_attention.py:299-349) with
// MemoryAttentionLayer.forward (:93-109) and RoPEAttention.forward (sam/transformer.py:345-424), for N objects of one
// frame at once (N <= TRACK_MAX_N): every projection / FFN GEMM and LayerNorm runs on N * 4096 rows, the two attentions per
// object (each object has its own memory bank).  Inputs: curr / curr_pos [4096,256] f32 (shared by the objects);
// ctx->t_kin16 / t_vin16 + n * t_nk_cap * 64 hold f16(memory+pos) / f16(memory) of object n: Nk[n] keys of which the first
// n_rope[n] get RoPE.  Output [N,4096,256] f32.
// combine(flash partials) -> out-projection + residual -> LayerNorm of the next sub-block, one launch (gemm_rowln.hip)
static int attn_tail(sam2mi_ctx* ctx, hipStream_t s, const Flash256Params& f, const Lin16& out_proj, const Norm& ln, float* x, half_t* h16) {
  RowLnParams r;
  memset(&r, 0, sizeof(r));
  r.o_part = f.o_part; r.ml_part = f.ml_part; r.splits = f.splits; r.part_rows = f.Nq;
  r.w = out_proj.w; r.bias = out_proj.b; r.res = x; r.out32 = x; r.ln_w = ln.w; r.ln_b = ln.b; r.eps = 1e-5f; r.out16 = h16; r.ld16 = 256; r.M = f.Nq;
  r.kc = f.dv == 64 ? 64 : 256;
  if (out_proj.N != 256 || out_proj.K != r.kc) return sam2mi_set_error(ctx, "attn_tail", "out-projection is not 256 x 256 (256 x 64 with the values in the memory space)");
  return run_rowln(ctx, s, r);
}

int memattn_forward(sam2mi_ctx* ctx, hipStream_t s, const float* curr, const float* curr_pos, int N, const int* Nk, const int* n_rope,
                    float* out32) {
  const int S = 4096, C = 256;
  PlanGroup plan_group(GRP_MA);
  if (N < 1 || N > TRACK_MAX_N) return sam2mi_set_error(ctx, "memattn_forward", "object batch out of range (1..TRACK_MAX_N)");
  const int M = N * S;
  const size_t cap = (size_t)ctx->t_nk_cap;
  float* x = ctx->t_x;
  const bool fused_tail = ctx->use_rowln;      // f16x3: split operands, the three-kernel tail
  // values in the memory space: the cross-attention multiplies its probabilities with the 64-channel memory tokens themselves
  // (flash256 DV = 64) and gemm_rowln applies Wo Wv behind it - no V projection of the bank, a quarter of the P.V products
  const bool msv = fused_tail && ctx->use_mem_space_values && ctx->mal[0].cross_vo.w != nullptr;
  for (int n = 0; n < N; ++n) {
    if (Nk[n] <= 0 || ceil32(Nk[n]) > ctx->t_nk_cap) return sam2mi_set_error(ctx, "memattn_forward", "memory length out of range");
    // x = curr + 0.1 * curr_pos   (pos_enc_at_input, :319-321)
    CHK(cast_add_launch(curr, C, curr_pos, C, 0, 0.1f, S, C, nullptr, 0, x + (size_t)n * S * C, C, s, ctx->lo16));
    // K / V of the memory for all 4 layers at once: K_all [Nk, 4*256] (RoPE on rows < n_rope), V^T_all [4*256, NkP]
    GemmParams p = lin_params(ctx->t_kin16 + n * cap * 64, 64, Nk[n], ctx->cross_k_all);
    p.out16 = ctx->t_kall16 + n * cap * 1024; p.ld16 = 1024;
    p.rope_cos = ctx->rope_cos; p.rope_sin = ctx->rope_sin; p.rope_len = S; p.rope_rows = n_rope[n]; p.rope_cols = 1024; p.rope_dim = C;
    CHKI(run_gemm(ctx, s, p));
    if (msv) {
      CHK(transpose_rows64_f16_launch(ctx->t_vin16 + n * cap * 64, ctx->t_vinT16 + n * cap * 64, Nk[n], ceil32(Nk[n]), s));
    } else {
      GemmParams q = lin_params(ctx->t_vin16 + n * cap * 64, 64, Nk[n], ctx->cross_v_all);
      q.n_split = 0; q.outT16 = ctx->t_vTall16 + n * cap * 1024; q.ldT16 = ceil32(Nk[n]);
      CHKI(run_gemm(ctx, s, q));
    }
  }
  for (int l = 0; l < 4; ++l) {
    const MemAttnLayerW& L = ctx->mal[l];
    // ---- self attention
    CHK(layernorm_launch(x, C, L.n1.w, L.n1.b, 1e-5f, M, C, ctx->t_h16, C, nullptr, 0, 0, s, ctx->lo16));
    {
      GemmParams p = lin_params(ctx->t_h16, C, M, L.self_qkv);
      p.n_split = 512; p.out16 = ctx->t_qk16; p.ld16 = 512; p.outT16 = ctx->t_vT16; p.ldT16 = M;
      p.col_scale = ctx->qs_self;
      p.rope_cos = ctx->rope_cos; p.rope_sin = ctx->rope_sin; p.rope_len = S; p.rope_rows = M; p.rope_cols = 512; p.rope_dim = C;
      CHKI(run_gemm(ctx, s, p));
    }
    for (int n = 0; n < N; ++n) {
      Flash256Params f;
      memset(&f, 0, sizeof(f));
      f.q = ctx->t_qk16 + (size_t)n * S * 512; f.ldq = 512; f.k = f.q + 256; f.ldk = 512; f.vT = ctx->t_vT16 + (size_t)n * S; f.ldvT = M;
      f.Nq = S; f.Nk = S; f.splits = flash256_pick_splits(S, S); f.o_part = ctx->t_opart; f.ml_part = ctx->t_ml;
      f.out = fused_tail ? nullptr : ctx->t_o16 + (size_t)n * S * C; f.ldout = C; f.scale_log2e = LOG2E / 16.f; f.out_lo_off = ctx->lo16;
      CHKI(run_flash256(ctx, s, f));
      // x += out_proj(attention), h = norm2(x): with the combine of the flash partials in one kernel (gemm_rowln.hip)
      if (fused_tail) CHKI(attn_tail(ctx, s, f, L.self_out, L.n2, x + (size_t)n * S * C, ctx->t_h16 + (size_t)n * S * C));
    }
    if (!fused_tail) {
      GemmParams p = lin_params(ctx->t_o16, C, M, L.self_out);
      p.res = x; p.ldres = C; p.out32 = x; p.ld32 = C;
      CHKI(run_gemm(ctx, s, p));
      // ---- cross attention to the memory bank
      CHK(layernorm_launch(x, C, L.n2.w, L.n2.b, 1e-5f, M, C, ctx->t_h16, C, nullptr, 0, 0, s, ctx->lo16));
    }
    {
      GemmParams p = lin_params(ctx->t_h16, C, M, L.cross_q);
      p.out16 = ctx->t_q16; p.ld16 = C; p.col_scale = ctx->qs_cross;
      p.rope_cos = ctx->rope_cos; p.rope_sin = ctx->rope_sin; p.rope_len = S; p.rope_rows = M; p.rope_cols = C; p.rope_dim = C;
      CHKI(run_gemm(ctx, s, p));
    }
    for (int n = 0; n < N; ++n) {
      const int NkP = ceil32(Nk[n]);
      Flash256Params f;
      memset(&f, 0, sizeof(f));
      f.q = ctx->t_q16 + (size_t)n * S * C; f.ldq = C; f.k = ctx->t_kall16 + n * cap * 1024 + l * 256; f.ldk = 1024;
      f.vT = ctx->t_vTall16 + n * cap * 1024 + (size_t)l * 256 * NkP; f.ldvT = NkP;
      if (msv) { f.vT = ctx->t_vinT16 + n * cap * 64; f.dv = 64; }
      f.Nq = S; f.Nk = Nk[n]; f.splits = msv ? flash256_pick_splits_dv64(S, Nk[n]) : flash256_pick_splits(S, Nk[n]); f.o_part = ctx->t_opart; f.ml_part = ctx->t_ml;
      f.out = fused_tail ? nullptr : ctx->t_o16 + (size_t)n * S * C; f.ldout = C; f.scale_log2e = LOG2E / 16.f; f.out_lo_off = ctx->lo16;
      CHKI(run_flash256(ctx, s, f));
      if (fused_tail) CHKI(attn_tail(ctx, s, f, msv ? L.cross_vo : L.cross_out, L.n3, x + (size_t)n * S * C, ctx->t_h16 + (size_t)n * S * C));
    }
    if (!fused_tail) {
      GemmParams p = lin_params(ctx->t_o16, C, M, L.cross_out);
      p.res = x; p.ldres = C; p.out32 = x; p.ld32 = C;
      CHKI(run_gemm(ctx, s, p));
      // ---- FFN
      CHK(layernorm_launch(x, C, L.n3.w, L.n3.b, 1e-5f, M, C, ctx->t_h16, C, nullptr, 0, 0, s, ctx->lo16));
    }
    {
      GemmParams p = lin_params(ctx->t_h16, C, M, L.lin1);
      p.act = ACT_RELU; p.out16 = ctx->t_ff16; p.ld16 = 2048;
      CHKI(run_gemm(ctx, s, p));
      GemmParams q = lin_params(ctx->t_ff16, 2048, M, L.lin2);
      q.res = x; q.ldres = C; q.out32 = x; q.ld32 = C;
      CHKI(run_gemm(ctx, s, q));
    }
  }
  CHK(layernorm_launch(x, C, ctx->ma_norm.w, ctx->ma_norm.b, 1e-5f, M, C, nullptr, 0, out32, C, 0, s, ctx->lo16));
#ifdef SAM2MI_EXPERIMENTAL
  CHKI(dummy_launches(ctx, s));
#endif
  return 0;
}

// ------------------------------------------------------------------ two-way transformer + output heads
static int tok_linear(sam2mi_ctx* ctx, hipStream_t s, const float* x, int ldx, const Lin32& L, float* y, int ldy, int T, int act,
                      const float* res = nullptr, int ldres = 0) {
  CHK(small_linear_launch(x, ldx, L.w, L.b, y, ldy, res, ldres, T, L.N, L.K, act, s));
  return 0;
}
static SmallLin mk_lin(const float* x, int ldx, const Lin32& L, float* y, int ldy, int T, int act, const float* res = nullptr,
                       int ldres = 0, const float* x2 = nullptr) {
  SmallLin d{x, L.w, L.b, y, res, ldx, ldy, ldres, T, L.N, L.K, act};
  d.x2 = x2;                                       // the linear runs on x + x2 (token + positional embedding, transformer.py:196,:204,:218)
  return d;
}
static Mlp3Group mk_mlp3(const float* x, const Lin32* L, float* y, int sigmoid_out, long x_rep_stride = 0, long y_rep_stride = 0) {
  Mlp3Group g;
  g.x = x; g.y = y; g.n_out = L[2].N; g.sigmoid_out = sigmoid_out; g.x_rep_stride = x_rep_stride; g.y_rep_stride = y_rep_stride;
  for (int i = 0; i < 3; ++i) { g.W[i] = L[i].w; g.b[i] = L[i].b; }
  return g;
}
static bool mlp3_ok(const Lin32* L) { return L[0].K == 256 && L[0].N == 256 && L[1].K == 256 && L[1].N == 256 && L[2].K == 256; }
static int tok_ln(sam2mi_ctx* ctx, hipStream_t s, float* x, const Norm& n, int T) {
  CHK(layernorm_launch(x, 256, n.w, n.b, 1e-5f, T, 256, nullptr, 0, x, 256, 0, s, ctx->lo16));
  return 0;
}

// token -> image attention for N prompts: q tokens (+pe) against each prompt's image keys (f16 operands in d_kpe16 / d_keys16)
static int t2i_attention(sam2mi_ctx* ctx, hipStream_t s, const Lin32& Wq, const Lin16& Wk, const Lin16& Wv, const Lin32& Wo, int N, int T,
                         const float* query_pe) {
  float* q = ctx->d_tok;            // [N*T,256]
  const int R = N * T;
  // qq = q_proj(q + qpe)
  {
    SmallLinBatch B;
    B.n = 1;
    B.d[0] = mk_lin(q, 256, Wq, ctx->d_t1, 128, R, 0, nullptr, 0, query_pe);
    CHK(small_linear_batch_launch(B, s));
  }
  GemmParams pk = lin_params(ctx->d_kpe16, 256, N * 4096, Wk);
  pk.out32 = ctx->d_big1; pk.ld32 = 128;
  CHKI(run_gemm(ctx, s, pk));
  GemmParams pv = lin_params(ctx->d_keys16, 256, N * 4096, Wv);
  pv.out32 = ctx->d_big2; pv.ld32 = 128;
  CHKI(run_gemm(ctx, s, pv));
  CHK(small_attn_launch(ctx->d_t1, 128, ctx->d_big1, 128, ctx->d_big2, 128, ctx->d_t2, 128, T, 4096, 8, 16, N, (size_t)T * 128,
                        (size_t)4096 * 128, (size_t)T * 128, s, ctx->d_t2i_part, ctx->d_t2i_part_floats));
  // q = q + out_proj(att)
  CHKI(tok_linear(ctx, s, ctx->d_t2, 128, Wo, q, 256, R, 0, q, 256));
  return 0;
}

// refresh the f16 image-side operands from ctx->d_keys [N*4096,256]: keys16 = f16(keys), kpe16 = f16(keys + pos)
static int refresh_key_operands(sam2mi_ctx* ctx, hipStream_t s, const float* pos_tok, bool pos_shared, int N) {
  CHK(cast_pair_launch(ctx->d_keys, pos_tok, pos_shared ? 4096 : 0, N * 4096, 256, ctx->d_keys16, ctx->d_kpe16, s, ctx->lo16));
  return 0;
}

// MaskDecoder.inference_predict_masks_torch (modeling/sam/mask_decoder.py:262-316) + TwoWayTransformer
// (sam/transformer.py:98-219), batched over N prompts / objects (N <= DEC_MAX_N): every image-side GEMM runs on
// M = N * 4096 rows and every token-side op on N * T rows.  in.keys_tok [4096,256] image embedding per prompt (token-major;
// keys_stride 0: one image shared by all prompts, the repeat_image case of sam2_image_predictor.py:564-579),
// in.dense_tok [dense_rows,256] dense prompt embedding added to it (dense_rows = 1: broadcast no_mask_embed, 4096: per token;
// null: none), in.pos_tok [4096,256] shared or [N,4096,256], in.tokens [N,T,256], hr0 [65536,32] / hr1 [16384,64] per prompt or shared.
// Results land in ctx->d_masks [N,4,65536], d_iou [N,4], d_obj [N]; the output tokens stay in ctx->d_tok [N,T,256] (hs: [0] object
// score token, [1] IoU token, [2..5] mask tokens; ctx->dec_T = T).
int decoder_forward(sam2mi_ctx* ctx, hipStream_t s, const DecoderIn& in, int N, int T) {
  PlanGroup plan_group(GRP_DEC);
  if (T < 6 || T > 64) return sam2mi_set_error(ctx, "decoder_forward", "token count out of range (6..64)");
  if (N < 1 || N > DEC_MAX_N) return sam2mi_set_error(ctx, "decoder_forward", "prompt batch out of range (1..DEC_MAX_N)");
  const int R = N * T, M = N * 4096;
  // src = image_embeddings + dense_prompt_embeddings (mask_decoder.py:216)
  for (int n = 0; n < N; ++n)
    CHK(cast_add_launch(in.keys_tok + (size_t)n * in.keys_stride, 256, in.dense_tok ? in.dense_tok + (size_t)n * in.dense_stride : nullptr, 256,
                        in.dense_rows >= 4096 ? 0 : 1, in.dense_tok ? 1.f : 0.f, 4096, 256, nullptr, 0, ctx->d_keys + (size_t)n * 4096 * 256, 256, s, ctx->lo16));
  const float* query_pe = in.tokens;             // the prompt tokens as they came in (transformer.py:128: query_pe = point_embedding), read-only
  float* q = ctx->d_tok;                         // first written by the output projection of layer 0's self attention (which REPLACES the queries)
  ctx->dec_T = T;
  for (int l = 0; l < 2; ++l) {
    const DecLayerW& L = ctx->dec[l];
    // ---- token self attention (layer 0: no pe, output replaces the queries; transformer.py:186-193)
    const float* qpe = l > 0 ? query_pe : nullptr;          // q = k = queries + query_pe from the second layer on
    {
      SmallLinBatch B;
      B.n = 3;
      const float* qin = l == 0 ? in.tokens : q;       // layer 0 reads the prompt tokens where they are (no copy into d_tok)
      B.d[0] = mk_lin(qin, 256, L.self_attn.q, ctx->d_t1, 256, R, 0, nullptr, 0, qpe);
      B.d[1] = mk_lin(qin, 256, L.self_attn.k, ctx->d_t2, 256, R, 0, nullptr, 0, qpe);
      B.d[2] = mk_lin(qin, 256, L.self_attn.v, ctx->d_t3, 256, R, 0);
      CHK(small_linear_batch_launch(B, s));
    }
    CHK(small_attn_launch(ctx->d_t1, 256, ctx->d_t2, 256, ctx->d_t3, 256, ctx->d_t4, 256, T, T, 8, 32, N, (size_t)T * 256, (size_t)T * 256,
                          (size_t)T * 256, s));
    CHKI(tok_linear(ctx, s, ctx->d_t4, 256, L.self_attn.o, q, 256, R, 0, l > 0 ? q : nullptr, 256));
    CHKI(tok_ln(ctx, s, q, L.n1, R));
    // ---- tokens attend to the image
    CHKI(refresh_key_operands(ctx, s, in.pos_tok, in.pos_shared, N));
    CHKI(t2i_attention(ctx, s, L.t2i_q, L.t2i_k, L.t2i_v, L.t2i_o, N, T, query_pe));
    CHKI(tok_ln(ctx, s, q, L.n2, R));
    // ---- MLP on tokens
    CHKI(tok_linear(ctx, s, q, 256, L.mlp1, ctx->d_t1, 2048, R, 2));
    CHKI(tok_linear(ctx, s, ctx->d_t1, 2048, L.mlp2, q, 256, R, 0, q, 256));
    CHKI(tok_ln(ctx, s, q, L.n3, R));
    // ---- image attends to the tokens: q = (keys+pe) Wq, k = (tokens+pe) Wk, v = tokens Wv
    {
      GemmParams p = lin_params(ctx->d_kpe16, 256, M, L.i2t_q);
      p.out32 = ctx->d_big1; p.ld32 = 128;
      CHKI(run_gemm(ctx, s, p));
      {
        SmallLinBatch B;
        B.n = 2;
        B.d[0] = mk_lin(q, 256, L.i2t_k, ctx->d_t1, 128, R, 0, nullptr, 0, query_pe);
        B.d[1] = mk_lin(q, 256, L.i2t_v, ctx->d_t2, 128, R, 0);
        CHK(small_linear_batch_launch(B, s));
      }
      CHK(small_attn_launch(ctx->d_big1, 128, ctx->d_t1, 128, ctx->d_t2, 128, nullptr, 128, 4096, T, 8, 16, N, (size_t)4096 * 128,
                            (size_t)T * 128, (size_t)4096 * 128, s, nullptr, 0, ctx->d_big16, ctx->lo16));
      GemmParams o = lin_params(ctx->d_big16, 128, M, L.i2t_o);
      o.res = ctx->d_keys; o.ldres = 256; o.out32 = ctx->d_keys; o.ld32 = 256;
      CHKI(run_gemm(ctx, s, o));
      CHK(layernorm_launch(ctx->d_keys, 256, L.n4.w, L.n4.b, 1e-5f, M, 256, nullptr, 0, ctx->d_keys, 256, 0, s, ctx->lo16));
    }
  }
  // ---- final token -> image attention + LN (transformer.py:134-139)
  CHKI(refresh_key_operands(ctx, s, in.pos_tok, in.pos_shared, N));
  CHKI(t2i_attention(ctx, s, ctx->fin_q, ctx->fin_k, ctx->fin_v, ctx->fin_o, N, T, query_pe));
  CHKI(tok_ln(ctx, s, q, ctx->fin_norm, R));
  // hs = q: per prompt [0] obj score token, [1] iou token, [2..5] mask tokens (read in place by the heads below and by select_mask)
  // ---- upscaling: ConvT(256->64) + hr1 -> LN2d -> GELU -> ConvT(64->32) + hr0 -> GELU  (mask_decoder.py:283-288)
  {
    GemmParams p = lin_params(ctx->d_keys16, 256, M, ctx->dc1);
    p.out32 = ctx->d_g; p.ld32 = 256;
    CHKI(run_gemm(ctx, s, p));
    CHK(upscale_glue_launch(ctx->d_g, 64, 64, ctx->dc1_b, in.hr1_tok, ctx->up_ln.w, ctx->up_ln.b, ctx->d_up1_16, N, in.hr1_stride, s, ctx->lo16));
    GemmParams p2 = lin_params(ctx->d_up1_16, 64, N * 16384, ctx->dc2);
    p2.out32 = ctx->d_g; p2.ld32 = 128;
    CHKI(run_gemm(ctx, s, p2));
    CHK(upscale_glue_launch(ctx->d_g, 128, 32, ctx->dc2_b, in.hr0_tok, nullptr, nullptr, ctx->d_up2_16, N, in.hr0_stride, s, ctx->lo16));
  }
  // ---- hyper-network MLPs on the 4 mask tokens -> [N,4,32], IoU head (sigmoid) and object-score head: one launch
  {
    if (!mlp3_ok(ctx->hyper[0]) || !mlp3_ok(ctx->iou_head) || !mlp3_ok(ctx->obj_head))
      return sam2mi_set_error(ctx, "decoder_forward", "output MLPs must be 256-256-256-n");
    Mlp3Batch B;
    B.n = 6;
    B.reps = N;
    for (int i = 0; i < 4; ++i) B.g[i] = mk_mlp3(q + (size_t)(2 + i) * 256, ctx->hyper[i], ctx->d_hyper + i * 32, 0, (long)T * 256, 4 * 32);
    B.g[4] = mk_mlp3(q + 256, ctx->iou_head, ctx->d_iou, 1, (long)T * 256, 4);
    B.g[5] = mk_mlp3(q, ctx->obj_head, ctx->d_obj, 0, (long)T * 256, 1);
    CHK(mlp3_launch(B, s));
  }
  CHK(cast_add_launch(ctx->d_hyper, 32, nullptr, 0, 0, 0.f, N * 4, 32, ctx->d_hyper16, 32, nullptr, 0, s, ctx->lo16));
  // masks[n, i, pix] = sum_c hyper[n, i, c] * up[n, pix, c]   (GEMM over pixels, stored transposed)
  for (int n = 0; n < N; ++n) {
    GemmParams p = gemm_params_zero();
    p.A = ctx->d_up2_16 + (size_t)n * 65536 * 32; p.lda = 32; p.W = ctx->d_hyper16 + (size_t)n * 4 * 32; p.ldw = 32; p.M = 65536; p.N = 4; p.K = 32;
    p.n_split = 0; p.outT32 = ctx->d_masks + (size_t)n * 4 * 65536; p.ldT32 = 65536;
    CHKI(run_gemm(ctx, s, p));
  }
  return 0;
}

// ------------------------------------------------------------------ memory encoder
// MemoryEncoder.inference_memory_torch (modeling/memory_encoder.py:233-241): feat2_tok [4096,256] raw
// vision features, mask1024 [1024*1024] already sigmoid-scaled -> out_tok64 [4096,64] f32.
int memenc_forward(sam2mi_ctx* ctx, hipStream_t s, const float* feat2_tok, const float* mask1024, float* out_tok64, const float* low256, int binarize) {
  PlanGroup plan_group(GRP_MENC);
  // MaskDownSampler: 4 x (conv3x3 s2 + LN2d + GELU), then 1x1
  if (mask1024) CHK(conv3x3s2_ln_gelu_launch(mask1024, 1024, 1, 4, ctx->md_w[0], ctx->md_b[0], ctx->md_ln[0].w, ctx->md_ln[0].b, ctx->m_c1, nullptr, s, ctx->lo16));
  else          // fused video path: bilinear x4 + sigmoid / binarise + scale folded into the first conv
    CHK(conv3x3s2_ln_gelu_from_low_launch(low256, binarize, 20.f, -10.f, ctx->md_w[0], ctx->md_b[0], ctx->md_ln[0].w, ctx->md_ln[0].b, ctx->m_c1, s));
  CHK(conv3x3s2_ln_gelu_launch(ctx->m_c1, 512, 4, 16, ctx->md_w[1], ctx->md_b[1], ctx->md_ln[1].w, ctx->md_ln[1].b, nullptr, ctx->m_c2_16, s, ctx->lo16));
  // conv 16 -> 64 as im2col + MFMA GEMM (K = 144), then LayerNorm2d + GELU
  CHK(im2col3x3s2_launch(ctx->m_c2_16, 256, 16, ctx->m_col16, s, ctx->lo16));
  {
    GemmParams p = lin_params(ctx->m_col16, 144, 16384, ctx->md_conv3);
    p.out32 = ctx->m_c4; p.ld32 = 64;
    CHKI(run_gemm(ctx, s, p));
  }
  CHK(layernorm_launch(ctx->m_c4, 64, ctx->md_ln[2].w, ctx->md_ln[2].b, 1e-6f, 16384, 64, ctx->m_c3_16, 64, nullptr, 0, 1, s, ctx->lo16));
  CHK(im2col3x3s2_launch(ctx->m_c3_16, 128, 64, ctx->m_col16, s, ctx->lo16));
  {
    GemmParams p = lin_params(ctx->m_col16, 576, 4096, ctx->md_conv4);
    p.out32 = ctx->m_c4; p.ld32 = 256;
    CHKI(run_gemm(ctx, s, p));
  }
  CHK(layernorm_launch(ctx->m_c4, 256, ctx->md_ln[3].w, ctx->md_ln[3].b, 1e-6f, 4096, 256, ctx->m_c4_16, 256, nullptr, 0, 1, s, ctx->lo16));
  {
    GemmParams p = lin_params(ctx->m_c4_16, 256, 4096, ctx->md_proj);
    p.out32 = ctx->m_emb; p.ld32 = 256;
    CHKI(run_gemm(ctx, s, p));
  }
  // x = pix_feat_proj(pix_feat) + mask embedding
  CHK(cast_add_launch(feat2_tok, 256, nullptr, 0, 0, 0.f, 4096, 256, ctx->m_pix16, 256, nullptr, 0, s, ctx->lo16));
  {
    GemmParams p = lin_params(ctx->m_pix16, 256, 4096, ctx->pix_proj);
    p.res = ctx->m_emb; p.ldres = 256; p.out32 = ctx->m_x; p.ld32 = 256;
    CHKI(run_gemm(ctx, s, p));
  }
  // Fuser: 2 x CXBlock (dwconv7 -> LN2d -> 256->1024 GELU -> 1024->256 -> gamma -> + x)
  for (int l = 0; l < 2; ++l) {
    CHK(dwconv7_launch(ctx->m_x, 64, 256, ctx->cx[l].dw_w, ctx->cx[l].dw_b, ctx->m_dw, s));
    CHK(layernorm_launch(ctx->m_dw, 256, ctx->cx[l].ln.w, ctx->cx[l].ln.b, 1e-6f, 4096, 256, ctx->m_ln16, 256, nullptr, 0, 0, s, ctx->lo16));
    GemmParams p = lin_params(ctx->m_ln16, 256, 4096, ctx->cx[l].pw1);
    p.act = ACT_GELU; p.out16 = ctx->m_h16; p.ld16 = 1024;
    CHKI(run_gemm(ctx, s, p));
    GemmParams q = lin_params(ctx->m_h16, 1024, 4096, ctx->cx[l].pw2);
    q.col_scale = ctx->cx[l].gamma; q.res = ctx->m_x; q.ldres = 256; q.out32 = ctx->m_x; q.ld32 = 256;
    CHKI(run_gemm(ctx, s, q));
  }
  CHK(cast_add_launch(ctx->m_x, 256, nullptr, 0, 0, 0.f, 4096, 256, ctx->m_ln16, 256, nullptr, 0, s, ctx->lo16));
  {
    GemmParams p = lin_params(ctx->m_ln16, 256, 4096, ctx->me_out);
    p.out32 = out_tok64; p.ld32 = 64;
    CHKI(run_gemm(ctx, s, p));
  }
  return 0;
}

// Image encoder orchestration: Hiera trunk + FPN neck + conv_s0/conv_s1.
// Reference: SAM2Base.inference_image_torch (modeling/sam2_base_official.py:566-582),
// Hiera.forward (modeling/backbones/hieradet.py:283-299), MultiScaleBlock.forward (:134-166),
// FpnNeck.forward (modeling/backbones/image_encoder.py:102-134).
//
// Layout: the residual stream is kept in WINDOW-MAJOR token order inside each stage (all tokens of a
// window are contiguous rows), so window partition/unpartition (backbones/utils.py:16-60) cost nothing:
// attention has no positional bias, so any consistent token permutation inside a stage is exact.
// Q-pooling halves the window edge (8->4 matches stage 2, 16->8 matches stage 4); only the
// stage-2 -> stage-3 transition (2 -> 16) needs one explicit re-ordering.
#include "engine.h"

static const float LOG2E = 1.4426950408889634f;

// The selective-split plan of the f16s precision mode for one Hiera block (tools/precision_shares.py and
// tools/precision_plan_video.py are the CPU experiments behind it).  What carries the f16-mode error of the encoder is, in this
// order: the rounding of WEIGHTS (coherent over all tokens: 4.6x the variance of all activation rounding together), of the
// attention output that feeds the projection (near-uniform attention makes it coherent inside a window), of q / k; stages 1-2
// and the attention linears of stage 3 matter, the MLP of stage 3 (2/3 of the encoder's FLOPs) and stage 4 hardly do.
static int plan_prec(const sam2mi_ctx* ctx, const HieraBlockW& b, int kind) {
  if (!ctx->selective) return PREC_AUTO;
  const int stage = b.dim_out >= 1152 ? 4 : b.dim_out >= 576 ? 3 : b.dim_out >= 288 ? 2 : 1;
  if (b.idx >= 0 && b.idx < 64 && ctx->plan_blk[b.idx][kind] >= 0) return ctx->plan_blk[b.idx][kind];
  return ctx->plan[stage][kind];
}

int hiera_block_forward(sam2mi_ctx* ctx, hipStream_t s, const HieraBlockW& b, int B, int& H, int& W, int& wcur) {
  const int M = B * H * W;
  const int C = b.dim, Co = b.dim_out;
  float* x = ctx->ws_x;
  // lo planes of the two LayerNorm outputs: only where a consumer splits its activation operand
  const size_t ln1_lo = (ctx->selective && plan_prec(ctx, b, LIN_QKV) != PREC_FULL && !(b.q_pool && plan_prec(ctx, b, LIN_SC) == PREC_FULL)) ? 0 : ctx->lo16;
  const size_t ln2_lo = (ctx->selective && plan_prec(ctx, b, LIN_FC1) != PREC_FULL) ? 0 : ctx->lo16;
  // 1. LN1 - as a kernel of its own only when the QKV projection cannot normalise its operand rows itself
  GemmParams qkv_p = lin_params(ctx->ws_a16, C, M, b.qkv);
  bool fuse1 = false;
  if ((ctx->ln_fuse || C <= ctx->ln1_fuse_maxc) && b.qkv.xs_ln_pack) {
    GemmParams t = qkv_p;                     // the launch below, with the LN-fused operand: does the X-stationary kernel take it?
    t.ln_x32 = x; t.ln_ld = C; t.ln_eps = 1e-6f; t.xs_pack = b.qkv.xs_ln_pack; t.bias = b.qkv.b_ln;
    t.n_split = 2 * Co; t.col_scale = b.qscale; t.xs_scale_cols = Co; t.out16 = ctx->ws_qk16; t.ld16 = 2 * Co; t.outT16 = ctx->ws_vT16; t.ldT16 = M;
    fuse1 = xs_eligible(ctx, t);
  }
  if (!fuse1) CHK(layernorm_launch(x, C, b.n1.w, b.n1.b, 1e-6f, M, C, ctx->ws_a16, C, nullptr, 0, 0, s, ln1_lo));
  int Mq = M;
  float* xres = x;           // residual target of the attention projection
  if (b.q_pool) {
    // shortcut = maxpool2x2(proj(LN(x)))   (hieradet.py:139-140)
    GemmParams p = lin_params(ctx->ws_a16, C, M, b.sc);
    p.prec = plan_prec(ctx, b, LIN_SC);
    Mq = M / 4;
    if (wcur <= 16 && (M & 31) == 0) {            // the pool runs in the GEMM epilogue: the unpooled [M, 2C] f32 tensor never exists
      p.pool_w = wcur; p.out32 = x; p.ld32 = Co;            // x is re-used: LN already consumed it
      CHKI(run_gemm(ctx, s, p));
    } else {
      p.out32 = ctx->ws_x2; p.ld32 = Co;
      CHKI(run_gemm(ctx, s, p));
      CHK(pool_tokens_f32_launch(ctx->ws_x2, Co, x, Co, M / (wcur * wcur), wcur, Co, s));
    }
  }
  // f16s: attention takes q / k as f16 hi + lo planes (three products for the scores) and V^T as plain f16 (attn_hiera.hip, SPLIT)
  const int astage = b.dim_out >= 1152 ? 4 : b.dim_out >= 576 ? 3 : b.dim_out >= 288 ? 2 : 1;
  const bool plain_attn = ctx->selective && ctx->split_attn && (M & 7) == 0 && !b.q_pool && plan_prec(ctx, b, LIN_PROJ) != PREC_FULL &&
                          ((b.window == 0 && !ctx->split_attn_global) || !ctx->split_attn_stage[astage]);      // plain f16 q / k where the plan says so
  const bool split_attn = ctx->selective && ctx->split_attn && (M & 7) == 0 && !plain_attn;
  // 2. QKV projection: q|k row-major, v transposed (attention consumes V^T tiles)
  {
    GemmParams p = qkv_p;
    if (fuse1) { p.ln_x32 = x; p.ln_ld = C; p.ln_eps = 1e-6f; p.xs_pack = b.qkv.xs_ln_pack; p.bias = b.qkv.b_ln; }
    p.n_split = 2 * Co;
    p.col_scale = b.qscale;                 // q pre-scaled (f32, before the f16 rounding) for the exp2-domain softmax
    p.xs_scale_cols = Co;                   // k / v columns have scale 1
    p.prec = plan_prec(ctx, b, LIN_QKV);
    p.outT_hi_only = split_attn || plain_attn;
    p.no_out_lo = plain_attn;
    if (ctx->precise && !split_attn && !plain_attn) {      // f16x3 mode: the attention kernel takes f32 q / k / V^T and splits them itself
      p.out32 = ctx->ws_qk32; p.ld32 = 2 * Co;
      p.outT32 = ctx->ws_vT32; p.ldT32 = M;
    } else {
      p.out16 = ctx->ws_qk16; p.ld16 = 2 * Co;
      p.outT16 = ctx->ws_vT16; p.ldT16 = M;
    }
    CHKI(run_gemm(ctx, s, p));
  }
  // 3. attention
  if (ctx->precise && !split_attn && !plain_attn) {
    PreciseAttnParams a;
    memset(&a, 0, sizeof(a));
    a.k = ctx->ws_qk32 + Co; a.ldk = 2 * Co;
    a.vT = ctx->ws_vT32; a.ldvT = M;
    a.o = ctx->ws_att16; a.ldo = Co; a.o_lo_off = ctx->lo16;
    a.heads = b.heads;
    const int win = (b.window > 0) ? wcur : 0;
    if (!b.q_pool) {
      a.q = ctx->ws_qk32; a.ldq = 2 * Co;
      if (win == 0) { a.GQ = a.GK = H * W; a.wq = a.GQ; a.wk = a.GK; a.num_groups = B; }
      else {
        const int n = win * win;
        if (n >= 32) { a.GQ = a.GK = n; a.wq = a.wk = n; a.num_groups = M / n; }
        else { a.GQ = a.GK = 32; a.wq = a.wk = n; a.num_groups = M / 32; }
      }
    } else {
      CHK(pool_tokens_f32_launch(ctx->ws_qk32, 2 * Co, ctx->ws_qp32, Co, M / (wcur * wcur), wcur, Co, s));
      a.q = ctx->ws_qp32; a.ldq = Co;
      const int nk = wcur * wcur, nq = nk / 4;
      if (nq >= 32) { a.GQ = nq; a.GK = nk; a.wq = nq; a.wk = nk; a.num_groups = M / nk; }
      else { const int pack = 32 / nq; a.GQ = 32; a.GK = pack * nk; a.wq = nq; a.wk = nk; a.num_groups = M / a.GK; }
    }
    CHKI(run_precise_attn(ctx, s, a));
  } else {
  HieraAttnParams a;
  memset(&a, 0, sizeof(a));
  a.k = ctx->ws_qk16 + Co; a.ldk = 2 * Co;
  a.vT = ctx->ws_vT16; a.ldvT = M;
  a.o = ctx->ws_att16; a.ldo = Co;
  a.heads = b.heads;
  a.scale_log2e = LOG2E / sqrtf(72.f);
  if (split_attn) a.qk_lo_off = a.o_lo_off = ctx->lo16;
  const int win = (b.window > 0) ? wcur : 0;
  if (!b.q_pool) {
    a.q = ctx->ws_qk16; a.ldq = 2 * Co;
    if (win == 0) { a.GQ = a.GK = H * W; a.wq = a.GQ; a.wk = a.GK; a.num_groups = B; }
    else {
      const int n = win * win;
      if (n >= 32) { a.GQ = a.GK = n; a.wq = a.wk = n; a.num_groups = M / n; }
      else { a.GQ = a.GK = 32; a.wq = a.wk = n; a.num_groups = M / 32; }          // pack 32/n windows, block-diagonal mask
    }
  } else {
    // Q max-pooled inside each window (hieradet.py:64-67)
    if (split_attn) CHK(pool_tokens_split_launch(ctx->ws_qk16, ctx->lo16, 2 * Co, ctx->ws_qp16, ctx->lo16, Co, M / (wcur * wcur), wcur, Co, s));
    else CHK(pool_tokens_f16_launch(ctx->ws_qk16, 2 * Co, ctx->ws_qp16, Co, M / (wcur * wcur), wcur, Co, s));
    a.q = ctx->ws_qp16; a.ldq = Co;
    const int nk = wcur * wcur, nq = nk / 4;
    if (nq >= 32) { a.GQ = nq; a.GK = nk; a.wq = nq; a.wk = nk; a.num_groups = M / nk; }
    else { const int pack = 32 / nq; a.GQ = 32; a.GK = pack * nk; a.wq = nq; a.wk = nk; a.num_groups = M / a.GK; }
  }
  CHKI(run_hiera_attn(ctx, s, a));
  }
  // 4. output projection + residual (+ norm2 in the same kernel where a workgroup can own whole rows: stages 1-3, f16 mode)
  const int pprec = plan_prec(ctx, b, LIN_PROJ);
  const bool proj_ln = ctx->use_projln && !ctx->ln_fuse && b.proj_pack && (Mq & 31) == 0 &&
                       (!ctx->selective || (b.proj_pack_lo && (pprec == PREC_WSPLIT || (pprec == PREC_FULL && split_attn))));
  if (proj_ln) {
    ProjLnParams q{ctx->ws_att16, Co, b.proj_pack, b.proj.b, xres, x, b.n2.w, b.n2.b, 1e-6f, ctx->ws_a16, Co, Mq, Co, nullptr, 0};
    if (ctx->selective) { q.wpack_lo = b.proj_pack_lo; q.a_lo_off = pprec == PREC_FULL ? ctx->lo16 : 0; }
    CHKI(run_projln(ctx, s, q));
  } else {
    GemmParams p = lin_params(ctx->ws_att16, Co, Mq, b.proj);
    p.res = xres; p.ldres = Co; p.out32 = x; p.ld32 = Co;
    p.prec = plan_prec(ctx, b, LIN_PROJ);
    CHKI(run_gemm(ctx, s, p));
  }
  if (b.q_pool) { H /= 2; W /= 2; wcur /= 2; }
  // 5-7. MLP (LN2 inside the consumer's operand load where it reads whole rows)
  if (ctx->use_fused_mlp && b.mlp_pack) {
    // stages 1-2: fc1 -> GELU -> fc2 -> +x in one kernel, the 4C-wide hidden never leaves the CU (mlp_fused.hip)
    if (ctx->ln_fuse && b.mlp_ln_pack) {
      MlpFusedParams m{nullptr, Co, b.mlp_ln_pack, b.fc1.b_ln, b.fc2.b, x, Co, Mq, 1e-6f};
      CHKI(run_mlp_fused(ctx, s, m, Co));
      return 0;
    }
    if (!proj_ln) CHK(layernorm_launch(x, Co, b.n2.w, b.n2.b, 1e-6f, Mq, Co, ctx->ws_a16, Co, nullptr, 0, 0, s, ln2_lo));
    MlpFusedParams m{ctx->ws_a16, Co, b.mlp_pack, b.fc1.b, b.fc2.b, x, Co, Mq};
    CHKI(run_mlp_fused(ctx, s, m, Co));
    return 0;
  }
  {
    GemmParams p = lin_params(ctx->ws_a16, Co, Mq, b.fc1);
    p.act = ACT_GELU; p.out16 = ctx->ws_h16; p.ld16 = 4 * Co;
    p.prec = plan_prec(ctx, b, LIN_FC1);
    p.no_out_lo = plan_prec(ctx, b, LIN_FC2) != PREC_FULL;      // the hidden tensor: lo plane only for a consumer that splits it
    bool fuse2 = false;
    if (ctx->ln_fuse && b.fc1.xs_ln_pack) {
      GemmParams t = p;
      t.ln_x32 = x; t.ln_ld = Co; t.ln_eps = 1e-6f; t.xs_pack = b.fc1.xs_ln_pack; t.bias = b.fc1.b_ln;
      if ((fuse2 = xs_eligible(ctx, t))) p = t;
    }
    if (!fuse2 && !proj_ln) CHK(layernorm_launch(x, Co, b.n2.w, b.n2.b, 1e-6f, Mq, Co, ctx->ws_a16, Co, nullptr, 0, 0, s, ln2_lo));
    CHKI(run_gemm(ctx, s, p));
  }
  {
    GemmParams p = lin_params(ctx->ws_h16, 4 * Co, Mq, b.fc2);
    p.res = x; p.ldres = Co; p.out32 = x; p.ld32 = Co;
    p.prec = plan_prec(ctx, b, LIN_FC2);
    CHKI(run_gemm(ctx, s, p));
  }
  return 0;
}

static inline int pad32(int v) { return (v + 31) / 32 * 32; }
// rows one window occupies in the window layout: its w*w tokens, packed (32 % T == 0: several windows per 32-row group, separated
// by the block-diagonal mask) or padded to a multiple of 32 (the padding rows are masked out as keys, dropped as queries)
static inline int window_rows(int T) { return (T % 32 == 0 || 32 % T == 0) ? T : pad32(T); }

// One MultiScaleBlock (hieradet.py:134-166) of the padded-window model sizes on plain row-major tokens [B, H, W, C]; see
// hiera_generic.hip for the window layout.
int hiera_block_forward_generic(sam2mi_ctx* ctx, hipStream_t s, const HieraBlockW& b, int B, int& H, int& W) {
  const int M = B * H * W, C = b.dim, Co = b.dim_out, hd = ctx->head_dim;
  float* x = ctx->ws_x;
  if (H != W) return sam2mi_set_error(ctx, "hiera_block_forward_generic", "square token grids only");
  CHK(layernorm_launch(x, C, b.n1.w, b.n1.b, 1e-6f, M, C, ctx->ws_a16, C, nullptr, 0, 0, s));
  if (b.q_pool) {
    // shortcut = maxpool2x2(proj(LN(x)))   (hieradet.py:139-140); the whole image is one "window" of the pooling kernel
    GemmParams p = lin_params(ctx->ws_a16, C, M, b.sc);
    p.out32 = ctx->ws_x2; p.ld32 = Co;
    CHKI(run_gemm(ctx, s, p));
    CHK(pool_tokens_f32_launch(ctx->ws_x2, Co, x, Co, B, W, Co, s));
  }
  // ---- windows
  const int w = b.window;
  const bool global = w == 0;
  const int nW = global ? 1 : (H + w - 1) / w;
  const int T = global ? H * W : w * w, wk = global ? T : window_rows(T);
  const int nwin = B * nW * nW, Mw = nwin * wk;
  if (b.q_pool && (global || (w & 1))) return sam2mi_set_error(ctx, "hiera_block_forward_generic", "query pooling needs an even window");
  const half_t* aw = ctx->ws_a16;
  if (!global) {
    CHK(window_gather_launch(ctx->ws_a16, ctx->ws_w16, B, H, W, C, w, nW, wk, s));
    aw = ctx->ws_w16;
  }
  {
    GemmParams p = lin_params(aw, C, Mw, b.qkv);
    p.n_split = 2 * Co; p.col_scale = b.qscale; p.xs_scale_cols = Co;
    p.out16 = ctx->ws_qk16; p.ld16 = 2 * Co; p.outT16 = ctx->ws_vT16; p.ldT16 = Mw;
    CHKI(run_gemm(ctx, s, p));
  }
  GenericAttnParams a;
  memset(&a, 0, sizeof(a));
  a.k = ctx->ws_qk16 + Co; a.ldk = 2 * Co; a.vT = ctx->ws_vT16; a.ldvT = Mw; a.heads = b.heads; a.vk = T;
  int wq = wk, we = global ? W : w;                 // rows per window of the query side, edge of the query window
  if (!b.q_pool) {
    a.q = ctx->ws_qk16; a.ldq = 2 * Co;
  } else {
    const int hw = w / 2;
    wq = window_rows(hw * hw);
    we = hw;
    CHK(window_pool_q_launch(ctx->ws_qk16, 2 * Co, ctx->ws_qp16, Co, nwin, w, wk, wq, s));
    a.q = ctx->ws_qp16; a.ldq = Co;
  }
  if (wq >= 32) { a.GQ = wq; a.GK = wk; a.num_groups = nwin; }
  else {
    const int pack = 32 / wq;
    if (nwin % pack) return sam2mi_set_error(ctx, "hiera_block_forward_generic", "window count not divisible by the packing factor");
    a.GQ = 32; a.GK = pack * wk; a.num_groups = nwin / pack;
  }
  a.wq = wq; a.wk = wk;
  const int Hq = b.q_pool ? H / 2 : H, Mq = B * Hq * Hq;
  a.o = global ? ctx->ws_att16 : ctx->ws_o16; a.ldo = Co;
  CHK(generic_attn_launch(a, hd, s));
  if (!global) CHK(window_scatter_launch(ctx->ws_o16, ctx->ws_att16, B, Hq, Hq, Co, we, nW, wq, s));
  {
    GemmParams p = lin_params(ctx->ws_att16, Co, Mq, b.proj);
    p.res = x; p.ldres = Co; p.out32 = x; p.ld32 = Co;
    CHKI(run_gemm(ctx, s, p));
  }
  if (b.q_pool) { H /= 2; W /= 2; }
  CHK(layernorm_launch(x, Co, b.n2.w, b.n2.b, 1e-6f, Mq, Co, ctx->ws_a16, Co, nullptr, 0, 0, s));
  {
    GemmParams p = lin_params(ctx->ws_a16, Co, Mq, b.fc1);
    p.act = ACT_GELU; p.out16 = ctx->ws_h16; p.ld16 = 4 * Co;
    CHKI(run_gemm(ctx, s, p));
  }
  {
    GemmParams p = lin_params(ctx->ws_h16, 4 * Co, Mq, b.fc2);
    p.res = x; p.ldres = Co; p.out32 = x; p.ld32 = Co;
    CHKI(run_gemm(ctx, s, p));
  }
  return 0;
}

// One stage-end: lateral 1x1 conv of the FPN on the stage's output (image_encoder.py:113-114).  `row0`: first token row of the frames in
// flight inside the level's full-batch lateral buffer (sub-batched stages 1-2).
static int stage_lateral(sam2mi_ctx* ctx, hipStream_t s, const HieraBlockW& b, int M, int level, size_t row0) {
  CHK(cast_add_launch(ctx->ws_x, b.dim_out, nullptr, 0, 0, 0.f, M, b.dim_out, ctx->ws_a16, b.dim_out, nullptr, 0, s, ctx->lo16));
  // levels 0 / 1: the lateral composed with conv_s0 / conv_s1 (32 / 64 channels, compact in ws_lat[level]); levels 2 / 3: lateral
  const Lin16& L = level == 0 ? ctx->neck_s0 : level == 1 ? ctx->neck_s1 : ctx->neck[level];
  GemmParams p = lin_params(ctx->ws_a16, b.dim_out, M, L);
  p.out32 = ctx->ws_lat[level] + row0 * L.N; p.ld32 = L.N;
  CHKI(run_gemm(ctx, s, p));
  return 0;
}

// Runs patch embedding + all blocks; fills ctx->ws_lat[0..3] (neck laterals, window-major) and returns
// the window size of each level's token order in wlev[].
//
// Optional (SAM2MI_ENC_SUB = n, off by default): stages 1-2 in SUB-BATCHES of n frames, hiera-large layout only.  Their kernels are
// HBM-bound launches of 2048+ workgroups that last 250-540 us at batch 8; two frames at a time the same kernels last 65-135 us, which
// was meant to let the kernels of the tracking stream in more often.  MEASURED, NO GAIN (round 3, same box: 202.0 whole batch, 201.6 /
// 202.2 frames/s with n = 2 / 4, 195.7 with n = 1): the encoder stream is the critical path and never idle (tools/event_timeline.py,
// profiles/*_event_timeline.txt: passes back to back, 37 ms each beside the tracking against 31 ms alone), the rate follows the
// SUM of the two streams' work, not how it is cut.  Outputs are bitwise those of the whole-batch pass (tests/test_plugs_gpu.py).
// Stage 1 / 2 outputs of a sub-batch sit at the sub-batch's stage-1 offset of ws_x; they are moved up to their place in the compact
// stage-2 layout before stage 3 runs on the whole batch.
static int trunk_forward(sam2mi_ctx* ctx, hipStream_t s, const float* img, const uint8_t* img_u8, int B, int wlev[4]) {
  const sam2mi_config& c = ctx->cfg;
  const int G = c.image_size / 4, E = c.embed_dim;
  const size_t nblk = ctx->blocks.size();
  // the sub-batched prefix: the blocks in front of the first one that leaves stage 2 (dim_out > 2 E), if no window re-ordering falls
  // inside it (none does for hiera-large: 8x8 windows, halved to 4x4 by the query pooling of block 2)
  size_t nprefix = 0;
  int sub = B;
  if (!ctx->generic && ctx->enc_sub > 0 && ctx->enc_sub < B) {
    while (nprefix < nblk && ctx->blocks[nprefix].dim_out <= 2 * E) ++nprefix;
    int w = 8;
    bool ok = nprefix > 0 && nprefix < nblk;
    for (size_t i = 0; ok && i < nprefix; ++i) {
      if (ctx->blocks[i].q_pool) w /= 2;
      const int wn = ctx->blocks[i + 1].window;
      if (wn > 0 && wn != w) ok = false;
    }
    if (ok) sub = ctx->enc_sub; else nprefix = 0;
  }
  float* const base_x = ctx->ws_x;
  int H = G, W = G, wcur = 8, level = 0;
  for (int f0 = 0; f0 < B; f0 += sub) {
    const int Bs = std::min(sub, B - f0);
    H = G; W = G; wcur = 8; level = 0;
    ctx->ws_x = base_x + (size_t)f0 * G * G * E;
    // patch embed (conv 7x7 s4 p3 as im2col GEMM) + position table, written in window-major order
    const size_t img_off = (size_t)f0 * 3 * c.image_size * c.image_size;
    if (img_u8) CHK(im2col_patch_u8_launch(img_u8 + img_off, Bs, c.image_size, ctx->ws_a16, s, ctx->lo16, ctx->generic));
    else CHK(im2col_patch_launch(img + img_off, Bs, c.image_size, ctx->ws_a16, s, ctx->lo16, ctx->generic));
    if (ctx->generic) wcur = G;                    // generic sizes: row-major tokens = one "window" as wide as the grid
    {
      GemmParams p = lin_params(ctx->ws_a16, 160, Bs * G * G, ctx->patch);
      p.res = ctx->pos_tab; p.ldres = E; p.res_mod = G * G;
      p.out32 = ctx->ws_x; p.ld32 = E;
      CHKI(run_gemm(ctx, s, p));
    }
    for (size_t i = 0; i < nprefix; ++i) {
      const HieraBlockW& b = ctx->blocks[i];
      CHKI(hiera_block_forward(ctx, s, b, Bs, H, W, wcur));
      if (b.stage_end) {
        CHKI(stage_lateral(ctx, s, b, Bs * H * W, level, (size_t)f0 * H * W));
        wlev[level] = wcur;
        ++level;
      }
    }
    if (nprefix > 0 && f0 > 0) {                   // up to the compact layout of the prefix's output (destination ends below the source: f0 >= sub)
      const size_t per_frame = (size_t)H * W * ctx->blocks[nprefix - 1].dim_out;
      CHK(hipMemcpyAsync(base_x + (size_t)f0 * per_frame, ctx->ws_x, (size_t)Bs * per_frame * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (ctx->ws_x != base_x + (size_t)f0 * G * G * E) { ctx->ws_x = base_x; return sam2mi_set_error(ctx, "trunk_forward", "unexpected buffer swap inside the sub-batched prefix"); }
  }
  ctx->ws_x = base_x;
  for (size_t i = nprefix; i < nblk; ++i) {
    const HieraBlockW& b = ctx->blocks[i];
    if (ctx->generic) {
      CHKI(hiera_block_forward_generic(ctx, s, b, B, H, W));
      wcur = W;
    } else CHKI(hiera_block_forward(ctx, s, b, B, H, W, wcur));
    // window size expected by the next block (hieradet.py:243-256: the window lags one block)
    if (!ctx->generic && i + 1 < nblk) {
      const int wn = ctx->blocks[i + 1].window;
      if (wn > 0 && wn != wcur) {
        CHK(permute_tokens_launch(ctx->ws_x, ctx->ws_x2, B, H, W, b.dim_out, wcur, wn, nullptr, 0, s));
        std::swap(ctx->ws_x, ctx->ws_x2);                 // both buffers have the same capacity (engine_core.hip)
        wcur = wn;
      }
    }
    if (b.stage_end) {
      CHKI(stage_lateral(ctx, s, b, B * H * W, level, 0));
      wlev[level] = wcur;
      ++level;
    }
  }
  return level == 4 ? 0 : sam2mi_set_error(ctx, "trunk_forward", "expected 4 stage outputs");
}

int encoder_forward(sam2mi_ctx* ctx, hipStream_t s, const float* img, int B, const EncOut* outs, const uint8_t* img_u8) {
  PlanGroup plan_group(GRP_NECK);
  const sam2mi_config& c = ctx->cfg;
  if (!ctx->finalized) return sam2mi_set_error(ctx, "encoder_forward", "weights not finalized");
  if (B <= 0 || B > c.max_batch) return sam2mi_set_error(ctx, "encoder_forward", "batch exceeds cfg.max_batch");
  const int G = c.image_size / 4;
  int wlev[4];
  if (!img && !img_u8) return sam2mi_set_error(ctx, "encoder_forward", "no input frames");
  CHKI(trunk_forward(ctx, s, img, img_u8, B, wlev));
  // level 2 (64x64): lateral + nearest-2x of level 3, to row-major tokens  (fpn_top_down_levels [2,3], scalp 1); levels 1 (128x128,
  // 64 channels) and 0 (256x256, 32 channels) are conv_s1 / conv_s0 outputs already.  Each frame goes straight to its own slot.
  if (B > PermuteDst::MAX_B) return sam2mi_set_error(ctx, "encoder_forward", "batch exceeds the permute destination table");
  float* d2[PermuteDst::MAX_B]; float* d1[PermuteDst::MAX_B]; float* d0[PermuteDst::MAX_B];
  for (int b = 0; b < B; ++b) { d2[b] = outs[b].feat2; d1[b] = outs[b].fpn1; d0[b] = outs[b].fpn0; }
  CHK(permute_tokens_launch(ctx->ws_lat[2], nullptr, B, G / 4, G / 4, 256, wlev[2], G / 4, ctx->ws_lat[3], wlev[3], s, d2));
  CHK(permute_tokens_launch(ctx->ws_lat[1], nullptr, B, G / 2, G / 2, 64, wlev[1], G / 2, nullptr, 0, s, d1));
  CHK(permute_tokens_launch(ctx->ws_lat[0], nullptr, B, G, G, 32, wlev[0], G, nullptr, 0, s, d0));
  return 0;
}

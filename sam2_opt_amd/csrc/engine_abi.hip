// extern "C" entry points: plug-level (reference tensor layouts) and the fused video path.
#include "engine.h"

static inline hipStream_t S(void* stream) { return (hipStream_t)stream; }

#define REQUIRE_READY()                                                                     \
  do {                                                                                      \
    if (!ctx) return 1;                                                                     \
    if (!ctx->finalized) return sam2mi_set_error(ctx, __func__, "weights not finalized");   \
  } while (0)

// ------------------------------------------------------------------ image encoder plugs
static int encode_to_scratch(sam2mi_ctx* ctx, hipStream_t s, const float* img, int B, std::vector<EncOut>& outs) {
  // results go to feature slots [0, B) of the cache used as scratch when called through the plug API
  if (B > (int)ctx->feats.size()) return sam2mi_set_error(ctx, "image_encoder", "B exceeds feat_slots");
  outs.resize(B);
  for (int b = 0; b < B; ++b) outs[b] = {ctx->feats[b].feat2, ctx->feats[b].fpn1, ctx->feats[b].fpn0};
  return encoder_forward(ctx, s, img, B, outs.data());
}

extern "C" int sam2mi_image_encoder(sam2mi_ctx* ctx, void* stream, const float* img, int B, float* const out[7]) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_enc, S(stream));
  hipStream_t s = S(stream);
  std::vector<EncOut> outs;
  CHKI(encode_to_scratch(ctx, s, img, B, outs));
  for (int b = 0; b < B; ++b) {
    if (out[0]) CHK(transpose_f32_launch(outs[b].feat2, out[0] + (size_t)b * 256 * 4096, 1, 4096, 256, s));
    if (out[6]) CHK(transpose_f32_launch(outs[b].feat2, out[6] + (size_t)b * 256 * 4096, 1, 4096, 256, s));
    if (out[4]) CHK(transpose_f32_launch(outs[b].fpn0, out[4] + (size_t)b * 32 * 65536, 1, 65536, 32, s));
    if (out[5]) CHK(transpose_f32_launch(outs[b].fpn1, out[5] + (size_t)b * 64 * 16384, 1, 16384, 64, s));
    for (int i = 0; i < 3; ++i) {
      const size_t n = (size_t)256 * (65536 >> (2 * i));
      if (out[1 + i]) CHK(hipMemcpyAsync(out[1 + i] + b * n, ctx->sine_pe[i], n * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
  }
  return 0;
}

namespace {
__global__ void normalize_img_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n_per_chan, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * 3 * n_per_chan) return;
  const int c = (int)((i / n_per_chan) % 3);
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  out[i] = (in[i] - mean[c]) / stdv[c];
}
__global__ void bilinear_kernel(const float* __restrict__ in, int C, int Hin, int Win, float* __restrict__ out, int Hout, int Wout) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)C * Hout * Wout) return;
  const int x = (int)(i % Wout), y = (int)((i / Wout) % Hout), c = (int)(i / ((size_t)Wout * Hout));
  const float sh = (float)Hin / Hout, sw = (float)Win / Wout;
  float sy = fmaxf(sh * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(sw * (x + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)sy, Hin - 1), x0 = min((int)sx, Win - 1);
  const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
  const float ly = sy - y0, lx = sx - x0;
  const float* p = in + (size_t)c * Hin * Win;
  out[i] = (1.f - ly) * ((1.f - lx) * p[y0 * Win + x0] + lx * p[y0 * Win + x1]) +
           ly * ((1.f - lx) * p[y1 * Win + x0] + lx * p[y1 * Win + x1]);
}
}  // namespace

extern "C" int sam2mi_set_image_e2e(sam2mi_ctx* ctx, void* stream, const float* img01, int B, float* feat0, float* feat1, float* feat2) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_enc, S(stream));
  hipStream_t s = S(stream);
  const size_t n = (size_t)ctx->cfg.image_size * ctx->cfg.image_size;
  if (B > ctx->cfg.max_batch) return sam2mi_set_error(ctx, __func__, "batch exceeds cfg.max_batch");
  float* norm = ctx->ws_x2;    // [B,3,S,S] fits: ws_x2 holds max_batch * 65536 * 288 floats
  normalize_img_kernel<<<dim3((unsigned)((B * 3 * n + 255) / 256)), dim3(256), 0, s>>>(img01, norm, n, B);
  CHK(hipGetLastError());
  // NOTE: ws_x2 is also used inside the trunk (shortcut buffer) but only after the im2col has consumed the image
  std::vector<EncOut> outs;
  CHKI(encode_to_scratch(ctx, s, norm, B, outs));
  for (int b = 0; b < B; ++b) {
    CHK(add_rowvec_launch(outs[b].feat2, 256, ctx->no_mem_embed, 4096, 256, nullptr, s));
    if (feat2) CHK(transpose_f32_launch(outs[b].feat2, feat2 + (size_t)b * 256 * 4096, 1, 4096, 256, s));
    if (feat0) CHK(transpose_f32_launch(outs[b].fpn0, feat0 + (size_t)b * 32 * 65536, 1, 65536, 32, s));
    if (feat1) CHK(transpose_f32_launch(outs[b].fpn1, feat1 + (size_t)b * 64 * 16384, 1, 16384, 64, s));
  }
  return 0;
}

// ------------------------------------------------------------------ memory attention plug
extern "C" int sam2mi_memory_attention(sam2mi_ctx* ctx, void* stream, const float* curr, const float* memory, const float* curr_pos,
                                       const float* memory_pos, const float* memory_exclude, const float* memory_pos_exclude,
                                       int L, int P, int N, float* out) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  if (N != 1) return sam2mi_set_error(ctx, __func__, "only N == 1 (one object per call) is implemented");
  hipStream_t s = S(stream);
  const int n_rope = L * 4096, Nk = n_rope + P;
  if (Nk <= 0 || (Nk + 31) / 32 * 32 > ctx->t_nk_cap) return sam2mi_set_error(ctx, __func__, "memory length out of range");
  // kin = f16(memory + memory_pos), vin = f16(memory); with N == 1 the (L,4096,1,64) tensors are [L*4096, 64] rows
  CHK(cast_add_launch(memory, 64, memory_pos, 64, 0, 1.f, n_rope, 64, ctx->t_kin16, 64, nullptr, 0, s, ctx->lo16));
  CHK(cast_add_launch(memory, 64, nullptr, 0, 0, 0.f, n_rope, 64, ctx->t_vin16, 64, nullptr, 0, s, ctx->lo16));
  if (P > 0) {
    CHK(cast_add_launch(memory_exclude, 64, memory_pos_exclude, 64, 0, 1.f, P, 64, ctx->t_kin16 + (size_t)n_rope * 64, 64, nullptr, 0, s, ctx->lo16));
    CHK(cast_add_launch(memory_exclude, 64, nullptr, 0, 0, 0.f, P, 64, ctx->t_vin16 + (size_t)n_rope * 64, 64, nullptr, 0, s, ctx->lo16));
  }
  return memattn_forward(ctx, s, curr, curr_pos, 1, &Nk, &n_rope, out);
}

// ------------------------------------------------------------------ mask decoder plug
extern "C" int sam2mi_mask_decoder(sam2mi_ctx* ctx, void* stream, const float* src, const float* tokens, const float* pos_src,
                                   const float* hr0, const float* hr1, int N, int T, float* masks, float* iou_pred,
                                   float* mask_tokens_out, float* object_score_logits) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  for (int n0 = 0; n0 < N; n0 += DEC_MAX_N) {
    const int nb = std::min(DEC_MAX_N, N - n0);
    // NCHW -> token-major, nb prompts back to back
    CHK(transpose_f32_launch(src + (size_t)n0 * 256 * 4096, ctx->p_d, nb, 256, 4096, s));
    CHK(transpose_f32_launch(hr0 + (size_t)n0 * 32 * 65536, ctx->p_a, nb, 32, 65536, s));
    CHK(transpose_f32_launch(hr1 + (size_t)n0 * 64 * 16384, ctx->p_c, nb, 64, 16384, s));
    CHK(transpose_f32_launch(pos_src + (size_t)n0 * 256 * 4096, ctx->d_big3, nb, 256, 4096, s));
    DecoderIn in{ctx->p_d, (size_t)4096 * 256, nullptr, 0, 0, ctx->d_big3, false, tokens + (size_t)n0 * T * 256,
                 ctx->p_a, (size_t)65536 * 32, ctx->p_c, (size_t)16384 * 64};
    CHKI(decoder_forward(ctx, s, in, nb, T));
    if (masks) CHK(hipMemcpyAsync(masks + (size_t)n0 * 4 * 65536, ctx->d_masks, (size_t)nb * 4 * 65536 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (iou_pred) CHK(hipMemcpyAsync(iou_pred + n0 * 4, ctx->d_iou, (size_t)nb * 4 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (mask_tokens_out)      // hs[:, 2:6] of every prompt (mask_decoder.py:280-281), straight from the token buffer
      CHK(hipMemcpy2DAsync(mask_tokens_out + (size_t)n0 * 1024, (size_t)4 * 256 * sizeof(float), ctx->d_tok + 2 * 256, (size_t)T * 256 * sizeof(float),
                           (size_t)4 * 256 * sizeof(float), nb, hipMemcpyDeviceToDevice, s));
    if (object_score_logits) CHK(hipMemcpyAsync(object_score_logits + n0, ctx->d_obj, (size_t)nb * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// ------------------------------------------------------------------ memory encoder plug
extern "C" int sam2mi_memory_encoder(sam2mi_ctx* ctx, void* stream, const float* pix_feat, const float* masks, int N, float* x, float* pos) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  for (int n = 0; n < N; ++n) {
    CHK(transpose_f32_launch(pix_feat + (size_t)n * 256 * 4096, ctx->p_d, 1, 256, 4096, s));
    CHKI(memenc_forward(ctx, s, ctx->p_d, masks + (size_t)n * 1024 * 1024, ctx->m_out));
    if (x) CHK(transpose_f32_launch(ctx->m_out, x + (size_t)n * 64 * 4096, 1, 4096, 64, s));
    if (pos) CHK(hipMemcpyAsync(pos + (size_t)n * 64 * 4096, ctx->mem_pos_nchw, (size_t)64 * 4096 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// ------------------------------------------------------------------ prompt encoder plug
extern "C" int sam2mi_prompt_encoder(sam2mi_ctx* ctx, void* stream, const float* coords, const int32_t* labels, int B, int Np,
                                     float* sparse, float* dense) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  for (int b = 0; b < B; ++b) {
    if (sparse)
      CHK(point_embed_launch(coords + (size_t)b * Np * 2, (const int*)labels + (size_t)b * Np, Np, ctx->gauss, ctx->point_emb4,
                             ctx->not_a_point, (float)ctx->cfg.image_size, sparse + (size_t)b * (Np + 1) * 256, s));
    if (dense) {
      // no_mask_embed broadcast to (256, 64, 64): fill token-major then transpose
      CHK(fill_f32_launch(ctx->p_d, 0.f, (size_t)4096 * 256, s));
      CHK(add_rowvec_launch(ctx->p_d, 256, ctx->no_mask_embed, 4096, 256, nullptr, s));
      CHK(transpose_f32_launch(ctx->p_d, dense + (size_t)b * 256 * 4096, 1, 4096, 256, s));
    }
  }
  return 0;
}

extern "C" int sam2mi_prompt_encoder_ex(sam2mi_ctx* ctx, void* stream, const float* coords, const int32_t* labels, int B, int Np, int pad,
                                        const float* masks, float* sparse, float* dense) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  if (B < 1 || Np < 0 || Np + 1 > 64 || (pad != 0 && pad != 1)) return sam2mi_set_error(ctx, __func__, "bad B / Np / pad (Np <= 63)");
  for (int b = 0; b < B; ++b) {
    if (sparse && Np + pad > 0) {
      // the embedding kernel always appends the padding point: build Np + 1 rows in scratch, keep Np + pad of them
      float* tmp = ctx->d_sparse;
      CHK(point_embed_launch(Np > 0 ? coords + (size_t)b * Np * 2 : nullptr, Np > 0 ? (const int*)labels + (size_t)b * Np : nullptr, Np, ctx->gauss,
                             ctx->point_emb4, ctx->not_a_point, (float)ctx->cfg.image_size, tmp, s));
      CHK(hipMemcpyAsync(sparse + (size_t)b * (Np + pad) * 256, tmp, (size_t)(Np + pad) * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (dense) {
      if (masks) {
        CHK(mask_embed_launch(masks + (size_t)b * 65536, ctx->mask_embed, ctx->d_dense, s));
        CHK(transpose_f32_launch(ctx->d_dense, dense + (size_t)b * 256 * 4096, 1, 4096, 256, s));
      } else {
        CHK(fill_f32_launch(ctx->p_d, 0.f, (size_t)4096 * 256, s));
        CHK(add_rowvec_launch(ctx->p_d, 256, ctx->no_mask_embed, 4096, 256, nullptr, s));
        CHK(transpose_f32_launch(ctx->p_d, dense + (size_t)b * 256 * 4096, 1, 4096, 256, s));
      }
    }
  }
  return 0;
}

extern "C" int sam2mi_dense_pe(sam2mi_ctx* ctx, void* stream, float* out) {
  REQUIRE_READY();
  CHK(transpose_f32_launch(ctx->dense_pe, out, 1, 4096, 256, S(stream)));
  return 0;
}

// HIP stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask).  `reserve` CUs are left OUT of the mask, spread
// evenly over the 8 XCDs whichever way the mask bits map to them (bit 16 k + k % 16, k = 0 .. reserve-1 cleared per 256 bits):
// kernels on this stream never occupy them, so small kernels of another stream always find free CUs.
extern "C" int sam2mi_stream_create_reserved(sam2mi_ctx* ctx, int reserve, void** stream_out) {
  if (!ctx || !stream_out) return 1;
  hipDeviceProp_t prop;
  int dev = 0;
  CHK(hipGetDevice(&dev));
  CHK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  if (reserve < 0 || reserve > 16 || ncu < 256) return sam2mi_set_error(ctx, __func__, "reserve must be 0..16 on a 256-CU device");
  const int nwords = (ncu + 31) / 32;
  std::vector<uint32_t> mask(nwords, 0xFFFFFFFFu);
  if (ncu % 32) mask[nwords - 1] = (1u << (ncu % 32)) - 1u;
  for (int k = 0; k < reserve; ++k) {
    const int bit = 16 * k + (k % 16);
    mask[bit / 32] &= ~(1u << (bit % 32));
  }
  hipStream_t st = nullptr;
  CHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask.data()));
  *stream_out = (void*)st;
  return 0;
}
extern "C" int sam2mi_stream_destroy(sam2mi_ctx* ctx, void* stream) {
  if (!ctx) return 1;
  if (stream) CHK(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

extern "C" int sam2mi_resize_bilinear(sam2mi_ctx* ctx, void* stream, const float* in, int C, int Hin, int Win, float* out, int Hout, int Wout) {
  if (!ctx) return 1;
  const size_t n = (size_t)C * Hout * Wout;
  bilinear_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream)>>>(in, C, Hin, Win, out, Hout, Wout);
  CHK(hipGetLastError());
  return 0;
}

// ============================================================================================ fused video path
extern "C" int sam2mi_video_encode(sam2mi_ctx* ctx, void* stream, const float* frames, int B, const int32_t* feat_slots) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_enc, S(stream));
  std::vector<EncOut> outs(B);
  for (int b = 0; b < B; ++b) {
    const int sl = feat_slots[b];
    if (sl < 0 || sl >= (int)ctx->feats.size()) return sam2mi_set_error(ctx, __func__, "feature slot out of range");
    outs[b] = {ctx->feats[sl].feat2, ctx->feats[sl].fpn1, ctx->feats[sl].fpn0};
  }
  return encoder_forward(ctx, S(stream), frames, B, outs.data());
}

extern "C" int sam2mi_set_fill_hole_area(sam2mi_ctx* ctx, int max_area) {
  if (!ctx) return 1;
  if (max_area < 0 || max_area > FILL_HOLES_MAX_AREA) return sam2mi_set_error(ctx, __func__, "fill_hole_area out of range (0..63)");
  ctx->fill_hole_area = max_area;
  return 0;
}

extern "C" int sam2mi_fill_holes(sam2mi_ctx* ctx, void* stream, const float* masks_in, float* masks_out, int N, int H, int W, int max_area) {
  if (!ctx) return 1;
  if (!masks_in || !masks_out || masks_in == masks_out) return sam2mi_set_error(ctx, __func__, "in / out must be distinct non-null buffers");
  if (max_area < 1 || max_area > FILL_HOLES_MAX_AREA) return sam2mi_set_error(ctx, __func__, "max_area out of range (1..63)");
  if (N <= 0 || H <= 0 || W <= 0 || (long)H * W > (1L << 30)) return sam2mi_set_error(ctx, __func__, "bad mask shape");
  CHK(fill_holes_launch(masks_in, masks_out, N, H, W, max_area, S(stream)));
  return 0;
}

extern "C" int sam2mi_video_encode_u8(sam2mi_ctx* ctx, void* stream, const uint8_t* frames_hwc, int B, const int32_t* feat_slots) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_enc, S(stream));
  if (!frames_hwc) return sam2mi_set_error(ctx, __func__, "null frames");
  std::vector<EncOut> outs(B);
  for (int b = 0; b < B; ++b) {
    const int sl = feat_slots[b];
    if (sl < 0 || sl >= (int)ctx->feats.size()) return sam2mi_set_error(ctx, __func__, "feature slot out of range");
    outs[b] = {ctx->feats[sl].feat2, ctx->feats[sl].fpn1, ctx->feats[sl].fpn0};
  }
  return encoder_forward(ctx, S(stream), nullptr, B, outs.data(), frames_hwc);
}

// SAM heads after the decoder: select, obj_ptr; writes bank slot + optional outputs
// `fill_before_outputs`: hole filling of the stored low-res mask (fill_holes_in_mask_scores on pred_masks,
// sam2_video_predictor_official.py:889-894) happens here; a tracked frame passes `mem_feat_slot >= 0` to run its
// memory encoder on the UNFILLED mask first, as track_step does (sam2_base_official.py:1151-1166 precedes :889).
// `n`: which prompt / object of the batched decoder pass (results in ctx->d_masks [n], d_iou [n], d_obj [n], tokens 2..5 of d_tok [n]).
// The caller's copies of the selected mask, the pointer and the score are written by the kernels that produce them (no copy launches
// on the per-frame path, where every launch of the tracking stream queues behind a kernel of the encoder stream).
static int sam_heads_finish(sam2mi_ctx* ctx, hipStream_t s, int multimask, int bank_slot, const sam2mi_frame_out* out,
                            int mem_feat_slot = -1, int n = 0) {
  sam2mi_ctx::BankSlot& bk = ctx->bank[bank_slot];
  const float* d_obj = ctx->d_obj + n;
  const bool fill = ctx->fill_hole_area > 0;
  CHK(select_mask_launch(ctx->d_masks + (size_t)n * 4 * 65536, ctx->d_iou + n * 4, d_obj, ctx->d_tok + ((size_t)n * ctx->dec_T + 2) * 256, multimask, ctx->d_best + 2,
                         0.05f, 0.98f, ctx->d_low_multi, bk.low_mask, ctx->d_tok_sel, ctx->d_best, ctx->d_iou_sel, s,
                         (out && !fill) ? out->low_res_masks : nullptr));
  // obj_ptr = MLP3(token) gated by the object score (sam2_base_official.py:474-484)
  {
    Mlp3Batch B;
    B.n = 1;
    B.reps = 1;
    Mlp3Group& g = B.g[0];
    g.x = ctx->d_tok_sel; g.y = bk.obj_ptr; g.n_out = 256; g.sigmoid_out = 0; g.x_rep_stride = 0; g.y_rep_stride = 0;
    for (int i = 0; i < 3; ++i) { g.W[i] = ctx->ptr_proj[i].w; g.b[i] = ctx->ptr_proj[i].b; }
    CHK(mlp3_launch(B, s));
  }
  CHK(gate_obj_ptr_launch(bk.obj_ptr, ctx->no_obj_ptr, d_obj, 256, s, out ? out->obj_ptr : nullptr, bk.obj_score, out ? out->object_score_logits : nullptr));
  if (mem_feat_slot >= 0) CHKI(sam2mi_video_encode_memory(ctx, (void*)s, mem_feat_slot, bank_slot, 0));
  if (fill) {
    CHK(fill_holes_launch(bk.low_mask, ctx->d_fill_tmp, 1, 256, 256, ctx->fill_hole_area, s));
    CHK(hipMemcpyAsync(bk.low_mask, ctx->d_fill_tmp, 65536 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  if (out) {
    const int nm = multimask ? 3 : 1;
    if (out->low_res_masks && fill) CHK(hipMemcpyAsync(out->low_res_masks, bk.low_mask, 65536 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out->low_res_multimasks) CHK(hipMemcpyAsync(out->low_res_multimasks, ctx->d_low_multi, (size_t)nm * 65536 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out->ious) CHK(hipMemcpyAsync(out->ious, ctx->d_iou_sel, nm * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out->best_idx) CHK(hipMemcpyAsync(out->best_idx, ctx->d_best, sizeof(int), hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// tokens[n] = [obj_score, iou, mask x4] ++ sparse(points of prompt n + pad)   (mask_decoder.py:186-202), N prompts of Np points
// pad = 1: the prompt encoder's padding point is appended (points without boxes, prompt_encoder.py:220); pad = 0 with Np = 0: no
// sparse prompt at all (prompt-free prediction: the six output tokens only)
static int build_tokens(sam2mi_ctx* ctx, hipStream_t s, const float* coords, const int32_t* labels, int N, int Np, int& T, int pad = 1) {
  if (Np + 1 + 6 > 64) return sam2mi_set_error(ctx, "build_tokens", "too many points");
  if (N < 1 || N > DEC_MAX_N) return sam2mi_set_error(ctx, "build_tokens", "prompt batch out of range");
  T = 6 + Np + pad;
  if (Np > 0) {
    CHK(hipMemcpyAsync(ctx->d_pts, coords, (size_t)N * Np * 2 * sizeof(float), hipMemcpyDefault, s));
    CHK(hipMemcpyAsync(ctx->d_labels, labels, (size_t)N * Np * sizeof(int), hipMemcpyDefault, s));
  }
  for (int n = 0; n < N; ++n) {
    float* tok = ctx->d_sparse + (size_t)n * T * 256;
    CHK(hipMemcpyAsync(tok, ctx->out_tokens, 6 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (pad)      // the embedding kernel writes Np + 1 rows (the padding point last)
      CHK(point_embed_launch(Np > 0 ? ctx->d_pts + (size_t)n * Np * 2 : nullptr, Np > 0 ? ctx->d_labels + (size_t)n * Np : nullptr, Np, ctx->gauss,
                             ctx->point_emb4, ctx->not_a_point, (float)ctx->cfg.image_size, tok + 6 * 256, s));
    else if (Np > 0) return sam2mi_set_error(ctx, "build_tokens", "points without the padding point are not a prompt the predictors produce");
  }
  return 0;
}

// one image (token-major pix features + its high-res features), N prompt token sets in ctx->d_sparse
static DecoderIn one_image_in(sam2mi_ctx* ctx, const sam2mi_ctx::FeatSlot& f) {
  return DecoderIn{ctx->t_pix, 0, ctx->no_mask_embed, 1, 0, ctx->dense_pe, true, ctx->d_sparse, f.fpn0, 0, f.fpn1, 0};
}
// same with a mask prompt: the dense embedding is PromptEncoder._embed_masks(mask256) instead of no_mask_embed
static int one_image_in_mask(sam2mi_ctx* ctx, hipStream_t s, const sam2mi_ctx::FeatSlot& f, const float* keys, const float* mask256, DecoderIn& in) {
  CHK(mask_embed_launch(mask256, ctx->mask_embed, ctx->d_dense, s));
  in = DecoderIn{keys, 0, ctx->d_dense, 4096, 0, ctx->dense_pe, true, ctx->d_sparse, f.fpn0, 0, f.fpn1, 0};
  return 0;
}

extern "C" int sam2mi_video_click(sam2mi_ctx* ctx, void* stream, int feat_slot, const float* coords, const int32_t* labels, int Np,
                                  const float* mask_logits, int multimask, int bank_slot, const sam2mi_frame_out* out) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  if (feat_slot < 0 || feat_slot >= (int)ctx->feats.size() || bank_slot < 0 || bank_slot >= (int)ctx->bank.size())
    return sam2mi_set_error(ctx, __func__, "slot out of range");
  const sam2mi_ctx::FeatSlot& f = ctx->feats[feat_slot];
  // pix_feat = feat + no_mem_embed (directly_add_no_mem_embed, sam2_base_official.py:953-957)
  CHK(cast_add_launch(f.feat2, 256, ctx->no_mem_embed, 256, 1, 1.f, 4096, 256, nullptr, 0, ctx->t_pix, 256, s, ctx->lo16));
  int T = 0;
  CHKI(build_tokens(ctx, s, coords, labels, 1, Np, T));
  DecoderIn in = one_image_in(ctx, f);
  if (mask_logits) CHKI(one_image_in_mask(ctx, s, f, ctx->t_pix, mask_logits, in));     // previous mask logits as dense prompt (:1136-1142)
  CHKI(decoder_forward(ctx, s, in, 1, T));
  if (out && out->pix_feat) CHK(hipMemcpyAsync(out->pix_feat, ctx->t_pix, (size_t)4096 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
  return sam_heads_finish(ctx, s, multimask, bank_slot, out);
}

// SAM2Base._use_mask_as_output (sam2_base_official.py:496-546): a binary mask input IS the output; the SAM decoder only
// supplies the object pointer.  mask1024: {0,1} float [image_size^2] (already resized / thresholded like
// sam2_video_predictor_official.py add_new_mask).  Stores low-res logits, pointer and +-10 score in `bank_slot`.
extern "C" int sam2mi_video_mask(sam2mi_ctx* ctx, void* stream, int feat_slot, const float* mask1024, int bank_slot, const sam2mi_frame_out* out) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  if (feat_slot < 0 || feat_slot >= (int)ctx->feats.size() || bank_slot < 0 || bank_slot >= (int)ctx->bank.size())
    return sam2mi_set_error(ctx, __func__, "slot out of range");
  if (!mask1024) return sam2mi_set_error(ctx, __func__, "null mask");
  const sam2mi_ctx::FeatSlot& f = ctx->feats[feat_slot];
  sam2mi_ctx::BankSlot& bk = ctx->bank[bank_slot];
  const int S1 = ctx->cfg.image_size;
  // object pointer: SAM heads on the RAW frame features (track_step :1120-1131) with mask_downsample(mask) as dense prompt, no points
  CHK(conv4x4s4_launch(mask1024, S1, ctx->mds_w, ctx->mds_b, ctx->d_mask256, ctx->d_flag, s));
  int T = 0;
  CHKI(build_tokens(ctx, s, nullptr, nullptr, 1, 0, T));
  CHK(hipMemcpyAsync(ctx->d_sparse + (size_t)T * 256, ctx->d_sparse + (size_t)(T - 1) * 256, 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
  T += 1;
  DecoderIn in;
  CHKI(one_image_in_mask(ctx, s, f, f.feat2, ctx->d_mask256, in));
  CHKI(decoder_forward(ctx, s, in, 1, T));
  CHKI(sam_heads_finish(ctx, s, 0, bank_slot, nullptr));           // single-mask path: token 0 -> obj_ptr, gated by the decoder's score
  // the mask decides whether the object is there (:527-535): score = +-10, pointer gated once more
  CHK(flag_to_score_launch(ctx->d_flag, 10.f, -10.f, ctx->d_pm10, s));
  CHK(gate_obj_ptr_launch(bk.obj_ptr, ctx->no_obj_ptr, ctx->d_pm10, 256, s, out ? out->obj_ptr : nullptr, bk.obj_score, out ? out->object_score_logits : nullptr));
  // low-res output = antialiased 4x down-sampling of mask * 20 - 10 (:503-511)
  CHK(aa_down4_launch(mask1024, S1, 20.f, -10.f, bk.low_mask, s));
  if (ctx->fill_hole_area > 0) {
    CHK(fill_holes_launch(bk.low_mask, ctx->d_fill_tmp, 1, 256, 256, ctx->fill_hole_area, s));
    CHK(hipMemcpyAsync(bk.low_mask, ctx->d_fill_tmp, 65536 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  if (out) {
    if (out->low_res_masks) CHK(hipMemcpyAsync(out->low_res_masks, bk.low_mask, 65536 * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

extern "C" int sam2mi_image_predict_ex(sam2mi_ctx* ctx, void* stream, int feat_slot, const float* coords, const int32_t* labels, int N, int Np,
                                       const float* mask_inputs, int multimask, float* masks_out, float* iou_out) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  if (feat_slot < 0 || feat_slot >= (int)ctx->feats.size()) return sam2mi_set_error(ctx, __func__, "slot out of range");
  if (N < 1) return sam2mi_set_error(ctx, __func__, "no prompts");
  const sam2mi_ctx::FeatSlot& f = ctx->feats[feat_slot];
  CHK(cast_add_launch(f.feat2, 256, ctx->no_mem_embed, 256, 1, 1.f, 4096, 256, nullptr, 0, ctx->t_pix, 256, s, ctx->lo16));
  const int pad = Np > 0 ? 1 : 0;          // SAM2ImagePredictor._predict passes boxes as points (labels 2 / 3): padded whenever points exist
  // N independent prompts on ONE image (repeat_image, sam2_image_predictor.py:564-579): batched through the decoder
  for (int n0 = 0; n0 < N; n0 += DEC_MAX_N) {
    const int nb = std::min(DEC_MAX_N, N - n0);
    int T = 0;
    CHKI(build_tokens(ctx, s, Np > 0 ? coords + (size_t)n0 * Np * 2 : nullptr, Np > 0 ? labels + (size_t)n0 * Np : nullptr, nb, Np, T, pad));
    DecoderIn in = one_image_in(ctx, f);
    if (mask_inputs) {                     // dense prompt = PromptEncoder._embed_masks(mask_input) per prompt (prompt_encoder.py:178-181)
      for (int n = 0; n < nb; ++n)
        CHK(mask_embed_launch(mask_inputs + (size_t)(n0 + n) * 65536, ctx->mask_embed, ctx->d_dense + (size_t)n * 4096 * 256, s));
      in.dense_tok = ctx->d_dense; in.dense_rows = 4096; in.dense_stride = (size_t)4096 * 256;
    }
    CHKI(decoder_forward(ctx, s, in, nb, T));
    if (multimask) {
      if (masks_out)
        CHK(hipMemcpy2DAsync(masks_out + (size_t)n0 * 3 * 65536, (size_t)3 * 65536 * sizeof(float), ctx->d_masks + 65536, (size_t)4 * 65536 * sizeof(float),
                             (size_t)3 * 65536 * sizeof(float), nb, hipMemcpyDeviceToDevice, s));
      if (iou_out)
        CHK(hipMemcpy2DAsync(iou_out + (size_t)n0 * 3, 3 * sizeof(float), ctx->d_iou + 1, 4 * sizeof(float), 3 * sizeof(float), nb, hipMemcpyDeviceToDevice, s));
    } else {
      // MaskDecoder._dynamic_multimask_via_stability (mask_decoder.py:346-382); "object present" forced on: no gating here
      CHK(fill_f32_launch(ctx->d_t1, 1.f, 1, s));
      for (int n = 0; n < nb; ++n) {
        CHK(select_mask_launch(ctx->d_masks + (size_t)n * 4 * 65536, ctx->d_iou + n * 4, ctx->d_t1, ctx->d_tok + ((size_t)n * T + 2) * 256, 0, ctx->d_best + 2, 0.05f,
                               0.98f, nullptr, ctx->d_low_sel, ctx->d_tok_sel, ctx->d_best, ctx->d_iou_sel, s));
        if (masks_out) CHK(hipMemcpyAsync(masks_out + (size_t)(n0 + n) * 65536, ctx->d_low_sel, 65536 * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (iou_out) CHK(hipMemcpyAsync(iou_out + (n0 + n), ctx->d_iou_sel, sizeof(float), hipMemcpyDeviceToDevice, s));
      }
    }
  }
  return 0;
}

extern "C" int sam2mi_image_predict(sam2mi_ctx* ctx, void* stream, int feat_slot, const float* coords, const int32_t* labels, int N, int Np,
                                    int multimask, float* masks_out, float* iou_out) {
  return sam2mi_image_predict_ex(ctx, stream, feat_slot, coords, labels, N, Np, nullptr, multimask, masks_out, iou_out);
}

extern "C" int sam2mi_video_encode_memory(sam2mi_ctx* ctx, void* stream, int feat_slot, int bank_slot, int is_mask_from_pts) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  if (feat_slot < 0 || feat_slot >= (int)ctx->feats.size() || bank_slot < 0 || bank_slot >= (int)ctx->bank.size())
    return sam2mi_set_error(ctx, __func__, "slot out of range");
  sam2mi_ctx::BankSlot& bk = ctx->bank[bank_slot];
  // mask_for_mem = (binarize ? mask > 0 : sigmoid(mask)) * 20 - 10 on the 1024^2 bilinear upsampling (:1000-1010): evaluated
  // inside the first conv of the mask down-sampler, the 4-MB tensor is never materialised
  CHKI(memenc_forward(ctx, s, ctx->feats[feat_slot].feat2, nullptr, ctx->m_out, bk.low_mask, is_mask_from_pts ? 1 : 0));
  // + (1 - appearing) * no_obj_embed_spatial, then the bf16 rounding of the memory bank
  CHK(add_rowvec_round_bf16_launch(ctx->m_out, ctx->no_obj_embed_spatial, 4096, 64, bk.obj_score, bk.mem, s));
  return 0;
}

// memory bank of one object -> ctx->t_kin16 / t_vin16 (+ object-pointer tokens) at workspace lane `n`
static int assemble_object_memory(sam2mi_ctx* ctx, hipStream_t s, const sam2mi_mem_select* sel, int n, int& Nk, int& n_rope) {
  if (!sel || sel->num_mem <= 0 || sel->num_mem > 8 || sel->num_ptr < 0 || sel->num_ptr > 32)
    return sam2mi_set_error(ctx, "video_track", "memory selection out of range (1..8 memories, 0..32 pointers)");
  const int L = sel->num_mem, P = 4 * sel->num_ptr;
  float* ptr_tok = ctx->t_ptr_tok + (size_t)n * 128 * 64;
  float* ptr_pos = ctx->t_ptr_pos + (size_t)n * 128 * 64;
  // object-pointer tokens + their temporal position encoding
  if (sel->num_ptr > 0) {
    PtrTokParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.n = sel->num_ptr;
    for (int i = 0; i < pp.n; ++i) {
      const int sl = sel->ptr_slot[i];
      if (sl < 0 || sl >= (int)ctx->bank.size()) return sam2mi_set_error(ctx, "video_track", "pointer slot out of range");
      pp.ptr[i] = ctx->bank[sl].obj_ptr;
      pp.dt[i] = sel->ptr_dt[i];
    }
    pp.tmax = sel->ptr_tmax; pp.Wt = ctx->tpos_proj.w; pp.bt = ctx->tpos_proj.b; pp.tok = ptr_tok; pp.pos = ptr_pos;
    CHK(ptr_tokens_launch(pp, s));
  }
  MemAssembleParams ma;
  memset(&ma, 0, sizeof(ma));
  ma.L = L;
  for (int i = 0; i < L; ++i) {
    const int sl = sel->mem_slot[i];
    if (sl < 0 || sl >= (int)ctx->bank.size() || sel->mem_tpos[i] < 0 || sel->mem_tpos[i] >= 7)
      return sam2mi_set_error(ctx, "video_track", "memory slot / tpos out of range");
    ma.feat[i] = ctx->bank[sl].mem;
    ma.tpos[i] = ctx->tpos_enc + sel->mem_tpos[i] * 64;
  }
  ma.pos = ctx->mem_pos; ma.ptr_tok = ptr_tok; ma.ptr_pos = ptr_pos; ma.P = P;
  ma.kin = ctx->t_kin16 + (size_t)n * ctx->t_nk_cap * 64; ma.vin = ctx->t_vin16 + (size_t)n * ctx->t_nk_cap * 64;
  ma.lo_off = ctx->lo16;
  CHK(mem_assemble_launch(ma, s));
  Nk = L * 4096 + P;
  n_rope = L * 4096;
  return 0;
}

extern "C" int sam2mi_video_track(sam2mi_ctx* ctx, void* stream, int feat_slot, const sam2mi_mem_select* sel, const sam2mi_prompt* prompt,
                                  int bank_slot, int run_mem_encoder, const sam2mi_frame_out* out) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  if (feat_slot < 0 || feat_slot >= (int)ctx->feats.size() || bank_slot < 0 || bank_slot >= (int)ctx->bank.size())
    return sam2mi_set_error(ctx, __func__, "slot out of range");
  const sam2mi_ctx::FeatSlot& f = ctx->feats[feat_slot];
  int Nk = 0, n_rope = 0;
  CHKI(assemble_object_memory(ctx, s, sel, 0, Nk, n_rope));
  CHKI(memattn_forward(ctx, s, f.feat2, ctx->sine_pe_tok64, 1, &Nk, &n_rope, ctx->t_pix));
  if (out && out->pix_feat) CHK(hipMemcpyAsync(out->pix_feat, ctx->t_pix, (size_t)4096 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
  int T = 0;
  const bool has_pts = prompt && prompt->num_points > 0;
  if (has_pts) {
    // correction clicks on a tracked frame: memory-conditioned features + the user's points (+ previous mask logits)
    CHKI(build_tokens(ctx, s, prompt->coords, prompt->labels, 1, prompt->num_points, T));
  } else {
    // no prompt: _forward_sam_heads pads with ONE (0,0)/-1 point and the prompt encoder appends another pad point (:395-401,
    // prompt_encoder.py:133-137) - both are the not_a_point embedding, so the 8 tokens are a constant built at weight-load time
    T = 8;
  }
  DecoderIn in = one_image_in(ctx, f);
  if (!has_pts) in.tokens = ctx->track_tokens;
  if (prompt && prompt->mask_logits) CHKI(one_image_in_mask(ctx, s, f, ctx->t_pix, prompt->mask_logits, in));
  CHKI(decoder_forward(ctx, s, in, 1, T));
  CHKI(sam_heads_finish(ctx, s, prompt ? prompt->multimask : 1, bank_slot, out, run_mem_encoder ? feat_slot : -1));
  return 0;
}

// N objects of one frame in one pass (plain propagation, no prompts): the reference loops objects with B = 1
// (sam2_video_predictor_official.py:691-725); here the memory-attention GEMMs / LayerNorms run on N * 4096 rows and the mask
// decoder on N "prompts", only the attentions over each object's own memory bank and the memory encoder stay per object.
extern "C" int sam2mi_video_track_batch(sam2mi_ctx* ctx, void* stream, int feat_slot, int N, const sam2mi_mem_select* sels,
                                        const int32_t* bank_slots, int run_mem_encoder, const sam2mi_frame_out* outs) {
  REQUIRE_READY();
  DomainGuard guard_(ctx->dom_track, S(stream));
  hipStream_t s = S(stream);
  if (N < 1 || N > TRACK_MAX_N) return sam2mi_set_error(ctx, __func__, "object batch out of range (1..8)");
  if (feat_slot < 0 || feat_slot >= (int)ctx->feats.size()) return sam2mi_set_error(ctx, __func__, "slot out of range");
  for (int n = 0; n < N; ++n)
    if (bank_slots[n] < 0 || bank_slots[n] >= (int)ctx->bank.size()) return sam2mi_set_error(ctx, __func__, "bank slot out of range");
  const sam2mi_ctx::FeatSlot& f = ctx->feats[feat_slot];
  int Nk[TRACK_MAX_N], n_rope[TRACK_MAX_N];
  for (int n = 0; n < N; ++n) CHKI(assemble_object_memory(ctx, s, sels + n, n, Nk[n], n_rope[n]));
  CHKI(memattn_forward(ctx, s, f.feat2, ctx->sine_pe_tok64, N, Nk, n_rope, ctx->t_pix));
  const int T = 8;                                               // no prompt: the constant token set, the same for every object
  DecoderIn in{ctx->t_pix, (size_t)4096 * 256, ctx->no_mask_embed, 1, 0, ctx->dense_pe, true, ctx->track_tokens, f.fpn0, 0, f.fpn1, 0};
  CHKI(decoder_forward(ctx, s, in, N, T));
  for (int n = 0; n < N; ++n) {
    const sam2mi_frame_out* out = outs ? outs + n : nullptr;
    if (out && out->pix_feat)
      CHK(hipMemcpyAsync(out->pix_feat, ctx->t_pix + (size_t)n * 4096 * 256, (size_t)4096 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    CHKI(sam_heads_finish(ctx, s, 1, bank_slots[n], out, run_mem_encoder ? feat_slot : -1, n));
  }
  return 0;
}

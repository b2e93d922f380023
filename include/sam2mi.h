/* sam2mi - C ABI of the MI355X-native SAM 2.1 backend (libsam2mi.so).
 *
 * This is the drop-in boundary for the reference's backend plug layer.  The reference swaps six
 * function-pointer attributes with `predictor.speedup(backend)` / `module.set_runtime_backend()`
 * and expects an executor object with `Inference(list[Tensor]) -> list[Tensor]`
 * (`from ytools.executor import ModelExectuor`, un-vendored submodule).  Each entry point below
 * replaces one of those plugs; the Python adapter in sam2_opt_amd/plugin.py binds them with ctypes
 * and installs them on the reference's modules.
 *
 * Conventions: every pointer is a DEVICE pointer to contiguous fp32 data in the reference's own
 * layout (NCHW / sequence-first) unless stated otherwise; `stream` is a hipStream_t (pass
 * torch.cuda.current_stream().cuda_stream); inputs are borrowed for the call, outputs are caller
 * allocated; nothing synchronises the host.  Returns 0 on success, non-zero on error
 * (message: sam2mi_last_error).  No C++ exceptions cross this boundary.
 * Threading: one context may be driven from several host threads, each on its own stream (the reference's contract,
 * /root/reference/video_multi_thread.py:36-87).  Entry points serialise their host side per workspace domain and order
 * consecutive calls that arrive on different streams with an event, so results never depend on the interleaving; calls of
 * the image-encoder domain and of the tracking domain still overlap.  For full concurrency create one context per thread.
 *
 * Paths below are relative to /root/reference/sam2/sam2/.
 */
#ifndef SAM2MI_H
#define SAM2MI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sam2mi_ctx sam2mi_ctx;

/* Hiera / model hyper-parameters (configs/sam2.1/sam2.1_hiera_l.yaml:1-120). */
typedef struct sam2mi_config {
  int embed_dim;             /* 144 */
  int num_heads;             /* 2 */
  int stages[4];             /* 2, 6, 36, 4 */
  int global_att_blocks[8];  /* 23, 33, 43, -1 ... */
  int window_spec[4];        /* 8, 4, 16, 8 */
  int image_size;            /* 1024 */
  int max_batch;             /* largest B of sam2mi_image_encoder / sam2mi_video_encode (workspace size) */
  int bank_slots;            /* capacity of the device-resident memory bank of the video path */
  int feat_slots;            /* capacity of the device-resident frame-feature cache of the video path */
  int precision;             /* SAM2MI_PRECISION_F16 (0, default): f16 MFMA operands, f32 accumulate - masks within ~2e-3 of the
                              * reference's fp32 torch path; SAM2MI_PRECISION_F16X3 (1): every MFMA operand is carried as a
                              * 2-term f16 split (hi + lo) and every product costs three MFMAs - the north-star "within 1e-3"
                              * class (measured <= 1e-4 per plug), about 2-3x the MFMA work;
                              * SAM2MI_PRECISION_F16S (2): SELECTIVE split - the same "within 1e-3" class at 0.87x the f16 rate:
                              * only the operands whose rounding carries the error are split (weights of the encoder's QKV and
                              * output projections, q / k of the stage-1 attention, patch embedding / neck / mask decoder), the
                              * rest runs as in the f16 mode (measured plan: engine_core.hip f16s_plan_init) */
} sam2mi_config;
#define SAM2MI_PRECISION_F16 0
#define SAM2MI_PRECISION_F16X3 1
#define SAM2MI_PRECISION_F16S 2

int sam2mi_abi_version(void);
int sam2mi_create(const sam2mi_config* cfg, sam2mi_ctx** out);
void sam2mi_destroy(sam2mi_ctx* ctx);
/* Message of the last failed call OF THE CALLING THREAD (errno-style, thread-local: a context may be driven from several host
 * threads at once); ctx may be NULL (creation errors).  Valid until the thread's next failing call. */
const char* sam2mi_last_error(sam2mi_ctx* ctx);

/* Weight loading - replaces build_sam._load_checkpoint (build_sam.py:164-174): call once per
 * state_dict entry with a HOST pointer to fp32 data, then sam2mi_finalize_weights (packs f16 MFMA
 * operands, fused projection matrices and the input-independent tables). Unknown keys are an error,
 * missing keys are reported by finalize (strict, like load_state_dict). */
int sam2mi_load_weight(sam2mi_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim);
int sam2mi_finalize_weights(sam2mi_ctx* ctx);

/* ---- plug: SAM2Base.inference_image (modeling/sam2_base_official.py:548-582)
 * img [B,3,1024,1024] normalised -> out[0..6] = vision_features (B,256,64,64), vision_pos_enc0..2
 * (B,256,{256,128,64}^2), backbone_fpn0 (B,32,256,256), backbone_fpn1 (B,64,128,128),
 * backbone_fpn2 (B,256,64,64).  Any out[i] may be NULL (skipped). */
int sam2mi_image_encoder(sam2mi_ctx* ctx, void* stream, const float* img, int B, float* const out[7]);

/* ---- plug: SAM2ImagePredictor.set_image_e2e (sam2_image_predictor.py:252-266)
 * img01 [B,3,1024,1024] in [0,1] -> feat0 (B,32,256,256), feat1 (B,64,128,128), feat2 (B,256,64,64)
 * (Normalize and "+ no_mem_embed" included). */
int sam2mi_set_image_e2e(sam2mi_ctx* ctx, void* stream, const float* img01, int B, float* feat0, float* feat1, float* feat2);

/* ---- plug: MemoryAttention.inference_memory_attention_{none,exclude} (modeling/memory_attention.py:294-349)
 * curr (4096,N,256), memory (L,4096,N,64), curr_pos (4096,N,256), memory_pos (L,4096,N,64),
 * memory_exclude (P,N,64), memory_pos_exclude (P,N,64) -> out (4096,N,256).  N must be 1. */
int sam2mi_memory_attention(sam2mi_ctx* ctx, void* stream, const float* curr, const float* memory, const float* curr_pos,
                            const float* memory_pos, const float* memory_exclude, const float* memory_pos_exclude,
                            int L, int P, int N, float* out);

/* ---- plug: MaskDecoder.inference_predict_masks (modeling/sam/mask_decoder.py:222-316)
 * src (N,256,64,64), tokens (N,T,256), pos_src (N,256,64,64), hr0 (N,32,256,256), hr1 (N,64,128,128)
 * -> masks (N,4,256,256), iou_pred (N,4), mask_tokens_out (N,4,256), object_score_logits (N,1). */
int sam2mi_mask_decoder(sam2mi_ctx* ctx, void* stream, const float* src, const float* tokens, const float* pos_src,
                        const float* hr0, const float* hr1, int N, int T, float* masks, float* iou_pred,
                        float* mask_tokens_out, float* object_score_logits);

/* ---- plug: MemoryEncoder.inference_memory (modeling/memory_encoder.py:228-241)
 * pix_feat (N,256,64,64), masks (N,1,1024,1024) already sigmoid-scaled -> x (N,64,64,64), pos (N,64,64,64). */
int sam2mi_memory_encoder(sam2mi_ctx* ctx, void* stream, const float* pix_feat, const float* masks, int N, float* x, float* pos);

/* ---- plug: PromptEncoder.inference_prompt (modeling/sam/prompt_encoder.py:215-231), points only
 * coords (B,Np,2) px, labels (B,Np) int32 -> sparse (B,Np+1,256), dense (B,256,64,64) [no_mask_embed]. */
int sam2mi_prompt_encoder(sam2mi_ctx* ctx, void* stream, const float* coords, const int32_t* labels, int B, int Np,
                          float* sparse, float* dense);
/* The full plug signature inference_prompt(points, boxes, masks) (prompt_encoder.py:215-231).  The caller lays the sparse
 * prompts out as the reference concatenates them: point coordinates first, then the two corners of each box with labels 2 / 3
 * (_embed_boxes :168-176 = _embed_points on the corners without padding); `pad` = 1 appends the padding point exactly when the
 * reference does (boxes is None, :220).  coords (B,Np,2) px, labels (B,Np) -> sparse (B,Np+pad,256); Np may be 0.
 * masks: (B,1,256,256) mask prompts -> dense = _embed_masks(masks) (:178-181), or NULL -> no_mask_embed. dense (B,256,64,64). */
int sam2mi_prompt_encoder_ex(sam2mi_ctx* ctx, void* stream, const float* coords, const int32_t* labels, int B, int Np, int pad,
                             const float* masks, float* sparse, float* dense);
/* PromptEncoder.get_dense_pe (prompt_encoder.py:113-122) -> (1,256,64,64) */
int sam2mi_dense_pe(sam2mi_ctx* ctx, void* stream, float* out);

/* ---- frame ingest: the reference's two resizers, on the device.  `in`: decoded RGB frame [H,W,3] uint8.
 * sam2mi_resize_u8_pil_bicubic   - load_video_frames_from_jpg_images / _load_img_as_tensor (utils/misc.py:92-101): PIL
 *   Image.resize((S,S)) = separable bicubic (a = -0.5, support stretched when shrinking), 22-bit fixed-point coefficients, a uint8
 *   rounding after each pass; out [S,S,3] uint8, bit-exact against Pillow.  Feed it to sam2mi_video_encode_u8.
 * sam2mi_resize_image_aa_bilinear - SAM2Transforms (utils/transforms.py:27-41): ToTensor (/255) + torchvision Resize on a float
 *   tensor = antialiased bilinear (aten _upsample_bilinear2d_aa); out [3,S,S] f32 in [0,1] (what set_image_e2e takes).
 * The FIRST call with a new (in, out) size pair allocates its coefficient tables (hipMalloc: may synchronise the device once) and
 * uploads them asynchronously on `stream`; a frame larger than any before re-allocates the scratch buffer (hipFree: waits for
 * the device).  Every other call only launches two kernels on `stream`. */
int sam2mi_resize_u8_pil_bicubic(sam2mi_ctx* ctx, void* stream, const uint8_t* in, int H, int W, uint8_t* out, int S);
int sam2mi_resize_image_aa_bilinear(sam2mi_ctx* ctx, void* stream, const uint8_t* in, int H, int W, float* out, int S);

/* ============================================================================================
 * Fused video path: device-resident frame features and memory bank, no host sync, no layout
 * round-trips between the plugs.  Host code (sam2_opt_amd/video_predictor.py) keeps the
 * reference's state machine (sam2_video_predictor_official.py:651-736, sam2_base_official.py:797-976)
 * and passes slot indices.
 * ============================================================================================ */

/* Encode B frames (B <= max_batch) and keep their features in feature-cache slots feat_slot[i]. */
int sam2mi_video_encode(sam2mi_ctx* ctx, void* stream, const float* frames, int B, const int32_t* feat_slots);
/* Same from DECODED frames: frames_hwc uint8 [B, image_size, image_size, 3] (device memory).  The /255, -mean, /std of
 * load_video_frames (utils/misc.py:270-276) is applied in f32 inside the patch-embedding gather, so the result is
 * bit-identical to sam2mi_video_encode on the normalised f32 frames and the 12.6 MB f32 frame is never materialised. */
int sam2mi_video_encode_u8(sam2mi_ctx* ctx, void* stream, const uint8_t* frames_hwc, int B, const int32_t* feat_slots);

/* Which memories / object pointers a tracked frame attends to (SAM2Base._prepare_memory_conditioned_features). */
typedef struct sam2mi_mem_select {
  int num_mem;               /* L: spatial memories, in concatenation order */
  int mem_slot[16];          /* bank slot of each */
  int mem_tpos[16];          /* index into maskmem_tpos_enc (= num_maskmem - t_pos - 1) */
  int num_ptr;               /* object pointers, in concatenation order */
  int ptr_slot[32];          /* bank slot of each */
  float ptr_dt[32];          /* signed temporal distance (frame_idx - t) */
  float ptr_tmax;            /* max_obj_ptrs_in_encoder - 1 */
} sam2mi_mem_select;

/* Outputs of one frame of the fused path (all device pointers, any may be NULL). */
typedef struct sam2mi_frame_out {
  float* low_res_masks;      /* (1,1,256,256) selected mask logits */
  float* low_res_multimasks; /* (1,3,256,256) or (1,1,256,256) */
  float* ious;               /* (1,3) or (1,1) */
  float* obj_ptr;            /* (1,256) */
  float* object_score_logits;/* (1,1) */
  float* pix_feat;           /* (4096,1,256) memory-conditioned features (debug/parity) */
  int32_t* best_idx;         /* (1) chosen multimask candidate */
} sam2mi_frame_out;

/* Conditioning frame with point prompts (add_new_points_or_box -> track_step, is_init_cond_frame):
 * SAM heads on feat + no_mem_embed; stores obj_ptr / score / low-res mask in bank slot `bank_slot`. */
int sam2mi_video_click(sam2mi_ctx* ctx, void* stream, int feat_slot, const float* coords, const int32_t* labels, int Np,
                       const float* mask_logits /* device [256*256] previous low-res logits as dense prompt, or NULL */,
                       int multimask, int bank_slot, const sam2mi_frame_out* out);

/* Mask prompt on a frame (add_new_mask -> track_step with mask_inputs -> SAM2Base._use_mask_as_output,
 * sam2_base_official.py:496-546): the binary mask IS the output - low-res logits = antialiased 4x down-sampling of
 * mask*20-10, object score +-10 by "any pixel set" - and the SAM decoder, fed mask_downsample(mask) as dense prompt on the raw
 * frame features, only supplies the object pointer.  mask1024: device {0,1} float [image_size^2]. */
int sam2mi_video_mask(sam2mi_ctx* ctx, void* stream, int feat_slot, const float* mask1024, int bank_slot, const sam2mi_frame_out* out);

/* Hole filling (fill_holes_in_mask_scores, utils/misc.py:312-338; replaces the reference's CUDA extension
 * csrc/connected_components.cu:62-282, bound at utils/misc.py:59-62): background (score <= 0) 8-connected components
 * of at most max_area pixels get score 0.1.  masks_in / masks_out: [N, H, W] f32 device buffers, must not alias;
 * 1 <= max_area <= 63. */
int sam2mi_fill_holes(sam2mi_ctx* ctx, void* stream, const float* masks_in, float* masks_out, int N, int H, int W, int max_area);
/* Apply it inside the fused video path to every stored low-res mask (SAM2Base.fill_hole_area, build_sam.py:129;
 * applied where sam2_video_predictor_official.py:889-894 does: after the frame's own memory encoding on tracked
 * frames, before it on prompted frames).  0 (default) = off. */
int sam2mi_set_fill_hole_area(sam2mi_ctx* ctx, int max_area);

/* Memory encoder for a bank slot (propagate_in_video_preflight / _encode_new_memory): uses the low-res
 * mask and object score stored in the slot, writes the bf16-rounded memory features into the slot. */
int sam2mi_video_encode_memory(sam2mi_ctx* ctx, void* stream, int feat_slot, int bank_slot, int is_mask_from_pts);

/* Prompt on a tracked frame (correction clicks: add_new_points_or_box on a frame that already has memories,
 * sam2_video_predictor_official.py:337-366 -> track_step :1136-1142). */
typedef struct sam2mi_prompt {
  const float* coords;        /* [num_points,2] pixels at image_size (host or device memory) */
  const int32_t* labels;      /* [num_points] */
  int32_t num_points;
  int32_t multimask;          /* 1: 3 candidates, best by IoU; 0: one mask with the dynamic stability fallback */
  const float* mask_logits;   /* device [256*256]: previous low-res logits (clamped to +-32 by the caller) as dense prompt, or NULL */
} sam2mi_prompt;

/* Tracked frame: memory attention over `sel`, SAM heads, memory encoder; result in `bank_slot`.
 * prompt == NULL: plain propagation (no prompt, multimask).  Otherwise the user's points / previous mask are the prompt. */
int sam2mi_video_track(sam2mi_ctx* ctx, void* stream, int feat_slot, const sam2mi_mem_select* sel, const sam2mi_prompt* prompt,
                       int bank_slot, int run_mem_encoder, const sam2mi_frame_out* out);

/* N objects of one frame in one pass (plain propagation, no prompts; 1 <= N <= 8).  The reference loops objects with B = 1
 * over shared frame features (sam2_video_predictor_official.py:691-725); here the memory-attention projections / FFN /
 * LayerNorms run on N*4096 rows and the mask decoder on N prompts - only the two attentions over each object's own memory
 * bank and the memory encoder stay per object.  sels[N], bank_slots[N], outs[N] (or NULL).  Results are identical to N
 * sam2mi_video_track calls. */
int sam2mi_video_track_batch(sam2mi_ctx* ctx, void* stream, int feat_slot, int N, const sam2mi_mem_select* sels,
                             const int32_t* bank_slots, int run_mem_encoder, const sam2mi_frame_out* outs);

/* Image predictor: prompt encoder + mask decoder on a cached frame (SAM2ImagePredictor._predict,
 * sam2_image_predictor.py:487-589: features + no_mem_embed, no object-score gating).  N independent prompts of Np points each
 * on the SAME image (coords [N,Np,2] pixels at image_size, labels [N,Np]: 0/1 points, 2/3 box corners) - the repeat_image case
 * (:564-579) and the 64-prompt batches of the automatic mask generator - run as one batched decoder pass (chunks of 16).
 * multimask: masks_out (N,3,256,256) + iou_out (N,3); otherwise the dynamic stability fallback picks one per prompt:
 * masks_out (N,1,256,256) + iou_out (N,1). */
int sam2mi_image_predict(sam2mi_ctx* ctx, void* stream, int feat_slot, const float* coords, const int32_t* labels, int N, int Np,
                         int multimask, float* masks_out, float* iou_out);
/* The same with the remaining prompt kinds of SAM2ImagePredictor._predict (sam2_image_predictor.py:487-589): mask_inputs
 * (N,1,256,256) low-res logits of a previous prediction as dense prompt (or NULL), and Np == 0 = no sparse prompt at all
 * (prompt-free / mask-only prediction: the six output tokens alone). */
int sam2mi_image_predict_ex(sam2mi_ctx* ctx, void* stream, int feat_slot, const float* coords, const int32_t* labels, int N, int Np,
                            const float* mask_inputs, int multimask, float* masks_out, float* iou_out);

/* Bilinear resize (align_corners=False) of a (H_in,W_in) fp32 map, F.interpolate semantics
 * (_get_orig_video_res_output, sam2_video_predictor_official.py:489-509). */
int sam2mi_resize_bilinear(sam2mi_ctx* ctx, void* stream, const float* in, int C, int Hin, int Win, float* out, int Hout, int Wout);

/* A HIP stream whose kernels never run on `reserve` (0..16) of the CUs, 2 per XCD at 16 (hipExtStreamCreateWithCUMask): the
 * encoder prefetch stream of the video predictor, so that the small kernels of the tracking path always find free CUs. */
int sam2mi_stream_create_reserved(sam2mi_ctx* ctx, int reserve, void** stream_out);
int sam2mi_stream_destroy(sam2mi_ctx* ctx, void* stream);

/* Profiling hook for bench.py: when enabled, every MFMA GEMM launch is bracketed by HIP events on its
 * own stream; totals are read back with sam2mi_profile_read (synchronises). */
int sam2mi_profile_enable(sam2mi_ctx* ctx, int on);
int sam2mi_profile_read(sam2mi_ctx* ctx, double* gemm_ms, double* gemm_flops, int64_t* gemm_launches,
                        double* attn_ms, double* attn_flops, int64_t* attn_launches);

/* Same for the fused Hiera MLP kernel (mlp_fused_kernel: fc1 + GELU + fc2 + residual; flops = 2 * 2 * M * C * 4C per launch). */
int sam2mi_profile_read_mlp(sam2mi_ctx* ctx, double* ms, double* flops, int64_t* launches);
/* Same for the X-stationary short-K GEMM (gemm_xs_kernel: QKV / projection / fc1 of Hiera stages 1-3; flops = 2 M N K). */
int sam2mi_profile_read_xs(sam2mi_ctx* ctx, double* ms, double* flops, int64_t* launches);
/* Same for the accumulator-stationary N = 576 GEMM (gemm_ks_kernel: projection / fc2 of Hiera stage 3). */
int sam2mi_profile_read_ks(sam2mi_ctx* ctx, double* ms, double* flops, int64_t* launches);

/* The GEMM-family launches again, per kernel instantiation: one line "name\tms\tflops\tlaunches\n" each, names as rocprofv3
 * prints them (e.g. "gemm_xs_kernel<576, true, false, 0>").  Returns the bytes written (NUL-terminated), -1 if cap is too small. */
int sam2mi_profile_read_kernels(sam2mi_ctx* ctx, char* out, int cap);

/* Debug/test entry points: single kernels behind the C ABI (used by tests/test_kernels_gpu.py). */
int sam2mi_debug_gemm(sam2mi_ctx* ctx, void* stream, const float* A, const float* W, const float* bias, int M, int N, int K,
                      int act, const float* residual, float* out);
int sam2mi_debug_hiera_attention(sam2mi_ctx* ctx, void* stream, const float* q, const float* k, const float* v, int groups,
                                 int heads, int GQ, int GK, int wq, int wk, float* out);
int sam2mi_debug_rowln(sam2mi_ctx* ctx, void* stream, const float* a_or_parts, const float* ml, int splits, const float* W, const float* bias,
                       float* x, const float* ln_w, const float* ln_b, int M, float* h);
int sam2mi_debug_projln(sam2mi_ctx* ctx, void* stream, const float* a, const float* W, const float* bias, float* x, const float* ln_w,
                        const float* ln_b, int M, int C, float* h);
int sam2mi_debug_flash256(sam2mi_ctx* ctx, void* stream, const float* q, const float* k, const float* v, int Nq, int Nk, float* out);
int sam2mi_debug_hiera_block(sam2mi_ctx* ctx, void* stream, int block_idx, const float* x_nhwc, int B, float* out_nhwc);
/* Hiera MLP x += fc2(GELU(fc1(xn))) on its own (f32 in, f16 MFMA operands): fused kernel (fused != 0) or the two-GEMM path;
 * iters > 0 also times that many launches (ms per launch).  MultiScaleBlock.forward, modeling/backbones/hieradet.py:163-165. */
int sam2mi_debug_mlp(sam2mi_ctx* ctx, void* stream, const float* xn, const float* W1, const float* b1, const float* W2,
                     const float* b2, float* x, int M, int C, int fused, int iters, float* ms_out);
int sam2mi_debug_gemm_bench(sam2mi_ctx* ctx, void* stream, int M, int N, int K, int iters, int mode, float* ms_out);
/* time `iters` launches of the d=256 flash attention (+ combine) on random f16 operands; ms per launch */
int sam2mi_debug_flash_bench(sam2mi_ctx* ctx, void* stream, int Nq, int Nk, int iters, float* ms_out);
int sam2mi_debug_read(sam2mi_ctx* ctx, void* stream, const char* name, float* out, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* SAM2MI_H */

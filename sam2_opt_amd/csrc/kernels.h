// Launchers for the non-GEMM, non-attention kernels (elementwise.hip, convs.hip, heads.hip).
// All take raw device pointers; token-major ("tok") tensors are [rows, channels] row-major.
// `lo_off` (f16 producers): 0, or the element offset of the lo array of the split-f16 precision mode (common.h) - the
// kernel then writes hi at the given pointer and lo at pointer + lo_off.
#pragma once
#include "common.h"

// ---------------------------------------------------------------- elementwise.hip
// LayerNorm over the last dim of x [M, C] (ldx) -> y16 (f16, ldy16) and/or y32 (f32, ldy32); act: 0 none, 1 GELU
hipError_t layernorm_launch(const float* x, int ldx, const float* w, const float* b, float eps, int M, int C,
                            half_t* y16, int ldy16, float* y32, int ldy32, int act, hipStream_t s, size_t lo_off = 0);
// y16[m, c] = f16(a[m, c] + sb * b[(m % bmod), c]);  b may be null; bmod == 0 -> m.  Optional y32 copy.
hipError_t cast_pair_launch(const float* a, const float* b, int bmod, int M, int C, half_t* y16, half_t* z16, hipStream_t s, size_t lo_off = 0);   // y16 = f16(a), z16 = f16(a + b[m % bmod]); contiguous [M, C]
hipError_t cast_add_launch(const float* a, int lda, const float* b, int ldb, int bmod, float sb, int M, int C,
                           half_t* y16, int ldy16, float* y32, int ldy32, hipStream_t s, size_t lo_off = 0);
// patch-embed im2col: img [B,3,S,S] f32 -> A [B*(S/4)^2, 160] f16, rows in window-major (w=8) token order,
// column k = c*49 + ky*7 + kx (conv 7x7, stride 4, pad 3); columns 147..159 are zero.
hipError_t im2col_patch_launch(const float* img, int B, int S, half_t* A, hipStream_t s, size_t lo_off = 0, int row_major = 0);   // row_major: plain token order
// same from decoded frames: uint8 [B, S, S, 3] HWC, normalised ((v/255 - mean)/std, ImageNet constants) on the fly
hipError_t im2col_patch_u8_launch(const uint8_t* img_hwc, int B, int S, half_t* A, hipStream_t s, size_t lo_off = 0, int row_major = 0);
// 2x2 max-pool inside w x w windows of window-major tokens: in [nwin*w*w, C] -> out [nwin*(w/2)^2, C]
hipError_t pool_tokens_f32_launch(const float* in, int ldin, float* out, int ldout, int nwin, int w, int C, hipStream_t s);
// in [R, 64] f16 -> out [64, ld] (out[c][r] = in[r][c]; columns R .. ld - 1 are written as zeros)
hipError_t transpose_rows64_f16_launch(const half_t* in, half_t* out, int R, int ld, hipStream_t s);
hipError_t pool_tokens_split_launch(const half_t* in, size_t in_lo, int ldin, half_t* out, size_t out_lo, int ldout, int nwin, int w, int C, hipStream_t s);
hipError_t pool_tokens_f16_launch(const half_t* in, int ldin, half_t* out, int ldout, int nwin, int w, int C, hipStream_t s);
// reorder tokens of B grids (H x W) from window size w_in to window size w_out (w == W means row-major);
// optional add of a nearest-2x-upsampled coarser grid `up` (H/2 x W/2, window size w_up).
struct PermuteDst { static constexpr int MAX_B = 32; float* p[MAX_B]; };     // per-frame destinations of permute_tokens (p[0] null: contiguous `out`)
hipError_t permute_tokens_launch(const float* in, float* out, int B, int H, int W, int C, int w_in, int w_out,
                                 const float* up, int w_up, hipStream_t s, float* const* dst = nullptr);
// batched 2-D transpose f32: in [batch, R, Cc] -> out [batch, Cc, R]
hipError_t transpose_f32_launch(const float* in, float* out, int batch, int R, int Cc, hipStream_t s);
// out[m, c] += v[c] * (flag_ptr ? (1 - (flag[0] > 0)) : 1)
hipError_t add_rowvec_launch(float* x, int ld, const float* v, int M, int C, const float* flag, hipStream_t s);
// round-to-nearest-even to bf16 precision, kept as f32 (the reference stores the memory bank as bf16)
hipError_t round_bf16_launch(const float* in, float* out, size_t n, hipStream_t s);
hipError_t add_rowvec_round_bf16_launch(const float* x, const float* v, int M, int C, const float* flag, float* out, hipStream_t s);   // x contiguous [M, C]
// fill f16 / f32
hipError_t fill_f32_launch(float* p, float v, size_t n, hipStream_t s);

// ---------------------------------------------------------------- hiera_generic.hip (window layouts of the padded-window model sizes)
// src [B,H,W,C] f16 -> dst [B*nW*nW*wk, C]: window (wy,wx) owns wk rows, the first w*w are its tokens (zero outside the image), the rest zero
hipError_t window_gather_launch(const half_t* src, half_t* dst, int B, int H, int W, int C, int w, int nW, int wk, hipStream_t s);
// 2x2 max-pool of q inside every window: q [nwin*wk, ldq] -> out [nwin*wq, C], rows >= (w/2)^2 of a window zero
hipError_t window_pool_q_launch(const half_t* q, int ldq, half_t* out, int C, int nwin, int w, int wk, int wq, hipStream_t s);
// src [B*nW*nW*wq, C] (windows of edge `we`) -> dst [B,H,W,C], cropping the padding
hipError_t window_scatter_launch(const half_t* src, half_t* dst, int B, int H, int W, int C, int we, int nW, int wq, hipStream_t s);

// ---------------------------------------------------------------- convs.hip (memory encoder)
// bilinear x4 upsample (align_corners=False) of low [256*256] + (binarize ? (x>0) : sigmoid(x)) * scale + bias -> out [1024*1024]
hipError_t mask_prep_launch(const float* low, float* out, int binarize, float scale, float bias, hipStream_t s);
// direct conv 3x3 stride 2 pad 1 on NHWC f32 + bias + LayerNorm2d(eps 1e-6) + GELU; in [Hin*Hin, CIN] -> out [Hout*Hout, COUT]
hipError_t conv3x3s2_ln_gelu_launch(const float* in, int Hin, int CIN, int COUT, const float* w, const float* b,
                                    const float* lnw, const float* lnb, float* out32, half_t* out16, hipStream_t s, size_t lo_off = 0);
// first MaskDownSampler stage straight from the 256^2 low-res logits: mask_prep (above) is evaluated per input tap, the 1024^2
// mask_for_mem tensor never exists; out32 [512*512, 4].  Bit-identical to mask_prep_launch + conv3x3s2_ln_gelu_launch(1 -> 4).
hipError_t conv3x3s2_ln_gelu_from_low_launch(const float* low256, int binarize, float scale, float bias, const float* w, const float* b,
                                             const float* lnw, const float* lnb, float* out32, hipStream_t s);
// im2col for a 3x3 s2 p1 conv on NHWC f16: in [Hin*Hin, CIN] -> A [Hout*Hout, 9*CIN], column (ky*3+kx)*CIN + c
hipError_t im2col3x3s2_launch(const half_t* in, int Hin, int CIN, half_t* A, hipStream_t s, size_t lo_off = 0);
// depth-wise 7x7 pad 3 on NHWC f32 [H*H, C]; w [C, 49]
hipError_t dwconv7_launch(const float* in, int H, int C, const float* w, const float* b, float* out, hipStream_t s);

// ---------------------------------------------------------------- postproc.hip
// fill_holes_in_mask_scores (utils/misc.py:312-338): background (score <= 0) 8-connected components of at most max_area
// pixels get score 0.1.  in / out: [N, H, W] f32, must not alias; 1 <= max_area <= FILL_HOLES_MAX_AREA.
constexpr int FILL_HOLES_MAX_AREA = 63;      // visited list of max_area + 1 ints per thread in LDS (64 KiB per workgroup)
hipError_t fill_holes_launch(const float* in, float* out, int N, int H, int W, int max_area, hipStream_t s);

// mask prompts: antialiased 4x bilinear down-sampling of in * a + b ([S,S] -> [S/4,S/4]); SAM2Base.mask_downsample (4x4 s4 conv,
// also sets *any_pos when a pixel is > 0); PromptEncoder._embed_masks on a [256,256] prompt -> dense [4096,256] token-major
struct MaskEmbedW { const float *w1, *b1, *ln1w, *ln1b, *w2, *b2, *ln2w, *ln2b, *w3, *b3; };
hipError_t aa_down4_launch(const float* in, int S, float a, float b, float* out, hipStream_t s);
hipError_t conv4x4s4_launch(const float* in, int S, const float* w, const float* bias, float* out, int* any_pos, hipStream_t s);
hipError_t mask_embed_launch(const float* mask256, const MaskEmbedW& W, float* dense_tok, hipStream_t s);
hipError_t flag_to_score_launch(const int* flag, float on, float off, float* out, hipStream_t s);   // out[0] = flag ? on : off

// ---------------------------------------------------------------- heads.hip (prompt encoder, mask decoder glue)
// y[t, n] = act(sum_k x[t, k] W[n, k] + b[n]) (+ res[t, n]);  f32 everywhere, T <= 64.  act: 0 none, 2 relu, 3 sigmoid
hipError_t small_linear_launch(const float* x, int ldx, const float* W, const float* b, float* y, int ldy,
                               const float* res, int ldres, int T, int N, int K, int act, hipStream_t s);
// up to 4 independent small linears in one launch (same semantics per entry)
struct SmallLin {
  const float* x; const float* W; const float* b; float* y; const float* res;
  int ldx, ldy, ldres, T, N, K, act;
  const float* x2 = nullptr;     // optional second operand, same layout as x: the linear runs on x + x2 (token + positional embedding)
};
struct SmallLinBatch { SmallLin d[4]; int n; };
hipError_t small_linear_batch_launch(const SmallLinBatch& B, hipStream_t s);
// up to 8 fused 3-layer MLPs (256 -> 256 -> 256 -> n_out, ReLU, ReLU, optional sigmoid) on one row each, one launch:
//   y[0..n_out) = act(W2 relu(W1 relu(W0 x + b0) + b1) + b2);  W row-major [out, 256] f32
// The launch is repeated `reps` times (grid.y): repetition r reads x + r * x_rep_stride and writes y + r * y_rep_stride
// (same weights) - one repetition per prompt / object of a batched decoder call.
struct Mlp3Group { const float* x; const float* W[3]; const float* b[3]; float* y; int n_out; int sigmoid_out; long x_rep_stride; long y_rep_stride; };
struct Mlp3Batch { Mlp3Group g[8]; int n; int reps; };
hipError_t mlp3_launch(const Mlp3Batch& B, hipStream_t s);
// sparse point embeddings (PromptEncoder._embed_points): pts [Np,2] px, labels [Np] -> out [Np+1, 256] (pad point appended)
hipError_t point_embed_launch(const float* pts, const int* labels, int Np, const float* gauss, const float* point_emb4,
                              const float* not_a_point, float image_size, float* out, hipStream_t s);
// dense positional encoding grid (PromptEncoder.get_dense_pe) -> out [64*64, 256] token-major
hipError_t dense_pe_launch(const float* gauss, int S, float* out, hipStream_t s);
// ConvTranspose 2x2 s2 glue.  g [Hin*Hin, 4*C] (GEMM output, column pos*C + c, pos = dy*2+dx) + bias[c] + hr [ (2Hin)^2, C ]
// -> (LayerNorm2d if lnw) -> GELU -> out16 [(2Hin)^2, C]
// `batch` images back to back in g / out16; hr advances by hr_bstride floats per image (0: shared)
hipError_t upscale_glue_launch(const float* g, int Hin, int C, const float* bias, const float* hr, const float* lnw,
                               const float* lnb, half_t* out16, int batch, size_t hr_bstride, hipStream_t s, size_t lo_off = 0);
// SAM-head selection on device (SAM2Base._forward_sam_heads :440-484, MaskDecoder.forward :151-169):
//  masks [4, 65536], iou [4], obj [1], tokens [4,256]  ->  low_sel [65536] (NO_OBJ filled when obj<=0),
//  tok_sel [256], best_idx [1].  multimask: argmax-IoU over candidates 1..3; otherwise candidate 0 with the
//  dynamic stability fallback (stab_counts: 2-int scratch; null disables the fallback).
hipError_t select_mask_launch(const float* masks, const float* iou, const float* obj, const float* tokens, int multimask,
                              int* stab_counts, float stab_delta, float stab_thresh, float* low_multi, float* low_sel,
                              float* tok_sel, int* best_idx, float* iou_out, hipStream_t s, float* low_sel2 = nullptr);
// obj_ptr = lam * ptr + (1 - lam) * no_obj_ptr, lam = obj > 0
hipError_t gate_obj_ptr_launch(float* ptr, const float* no_obj_ptr, const float* obj, int C, hipStream_t s, float* ptr_out = nullptr,
                               float* score_out = nullptr, float* score_out2 = nullptr);
// memory-bank K/V operand assembly: for slot s (0..L-1) rows [s*4096, (s+1)*4096):
//   kin = f16(feat_s + pos + tpos_s), vin = f16(feat_s); pointer tokens appended after the L frames.
struct MemAssembleParams {
  const float* feat[8]; const float* tpos[8]; int L;        // feat_s [4096,64] f32 (bf16-rounded), tpos_s [64]
  const float* pos;                                          // maskmem_pos_enc [4096,64]
  const float* ptr_tok; const float* ptr_pos; int P;         // [P,64] each (already split into 64-wide tokens)
  half_t* kin; half_t* vin;                                  // [L*4096 + P (padded), 64]
  size_t lo_off;                                             // split-f16 mode: lo planes of kin / vin (0: off)
  float* mem32; float* mempos32;                             // optional f32 copies (plug-format debugging)
};
hipError_t mem_assemble_launch(const MemAssembleParams& p, hipStream_t s);
// object-pointer tokens: ptr[i] [256] f32 -> tok rows [4*i + j] = ptr[i][64*j .. 64*j+64);
// pos rows 4*i+j = Linear_{256->64}( [sin(dt_i/tmax / dim_t), cos(...)] )   (get_1d_sine_pe, sam2_utils.py:64-74)
struct PtrTokParams {
  const float* ptr[32]; float dt[32]; int n;
  float tmax;
  const float* Wt; const float* bt;     // obj_ptr_tpos_proj (64, 256) f32
  float* tok; float* pos;               // [4n, 64] each
};
hipError_t ptr_tokens_launch(const PtrTokParams& p, hipStream_t s);

// Internal engine state behind the C ABI (include/sam2mi.h).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/sam2mi.h"
#include "attn.h"
#include "gemm.h"
#include "kernels.h"
#include "gemm_xs.h"
#include "gemm_ks.h"
#include "mlp_fused.h"

struct HostW {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct Lin16 {            // MFMA operand: f16 weight [N, K] (nn.Linear layout), f32 bias
  half_t* w = nullptr;
  float* b = nullptr;
  int N = 0, K = 0;
  size_t lo_off = 0;           // f16x3 mode: the lo plane of w sits lo_off (= N * K) elements behind it; 0 in the f16 mode
  half_t* xs_pack = nullptr;   // same weight in the piece order of the X-stationary GEMM (Hiera stages 1-3, K <= 576)
  half_t* xs_wpack = nullptr;  // ... as a 2-term split, [W_hi | W_lo] chunk pairs (f16s mode: linears planned as weight split; gemm_xs_wsplit_pack)
  half_t* xs_ln_pack = nullptr;   // ... with the preceding LayerNorm's gain folded in (W diag(g)), for the LN-fused operand load
  float* b_ln = nullptr;          //     and its bias: W b_LN + b
  half_t* ks_pack = nullptr;   // ... of the accumulator-stationary GEMM (N = 576: stage-3 projection and fc2)
};
struct Lin32 {            // tiny fp32 linear for the token-side heads
  float* w = nullptr;
  float* b = nullptr;
  int N = 0, K = 0;
};
struct Norm {
  float* w = nullptr;
  float* b = nullptr;
  int C = 0;
};

struct HieraBlockW {
  int idx, dim, dim_out, heads, window;
  bool q_pool, stage_end;
  Norm n1, n2;
  Lin16 qkv, proj, fc1, fc2, sc;   // sc: dim-change shortcut projection (blocks 2, 8, 44)
  half_t* proj_pack_lo = nullptr;  // ... of the weight's lo plane (f16s mode: split projection + norm2 in one launch, stages 1-2)
  half_t* proj_pack = nullptr;     // proj in the X-stationary piece order, for gemm_projln_kernel (proj + residual + norm2 in one launch)
  half_t* mlp_pack = nullptr;      // fc1 + fc2 in the fused MLP kernel's piece order (dim_out <= 288), mlp_fused_pack
  half_t* mlp_ln_pack = nullptr;   // the same with norm2 folded into fc1 (LN-fused operand load); fc1 bias = fc1.b_ln
  float* qscale = nullptr;         // [3*dim_out] column scale of the QKV GEMM: q columns *= 72^-0.5*log2(e), k/v columns 1
};

struct MemAttnLayerW {
  Norm n1, n2, n3;
  Lin16 self_qkv;   // [768, 256] = q_proj | k_proj | v_proj
  Lin16 self_out, cross_q, cross_out, lin1, lin2;
  Lin16 cross_vo;   // [256, 64] = out_proj . v_proj of the cross-attention (bias: Wo bv + bo), composed in f64 at weight-load time: the
                    // value projection applied BEHIND the attention (attn_flash256.hip DV = 64, gemm_rowln.hip KC = 64)
};

struct AttnW32 { Lin32 q, k, v, o; };          // token-side projections (fp32)
struct DecLayerW {
  AttnW32 self_attn;                            // 256 -> 256
  Lin32 t2i_q, t2i_o;                           // token side of token->image attention
  Lin16 t2i_k, t2i_v;                           // image side [128, 256]
  Lin16 i2t_q;                                  // image side query projection [128, 256]
  Lin32 i2t_k, i2t_v;                           // token side
  Lin16 i2t_o;                                  // [256, 128]
  Lin32 mlp1, mlp2;
  Norm n1, n2, n3, n4;
};

struct ProfAcc {
  double ms = 0, flops = 0, bytes = 0;     // bytes: ALGORITHMIC HBM bytes (operands read once + outputs written once)
  int64_t launches = 0;
  struct Pending { hipEvent_t first, second; ProfAcc* named; };   // named: the per-instantiation accumulator fed by the same events
  std::vector<Pending> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};

// One context may be driven from several host threads, each on its own HIP stream (the reference's contract:
// /root/reference/video_multi_thread.py:36-87).  The workspaces are shared, so every entry point takes the guard of the
// workspace domain it uses: the host side of the call is serialised by the mutex, and when the previous call of the domain was
// enqueued on a DIFFERENT stream the new stream first waits for an event recorded at the end of that call.  Calls that stay
// on one stream pay one hipEventRecord.  The encoder domain (ws_*) and the tracking domain (t_*, d_*, m_*, p_*) are
// independent, which is what lets the video predictor run its encoder stream beside the tracking stream.
struct WsDomain {
  std::recursive_mutex mu;
  hipStream_t last = nullptr;
  hipEvent_t ev = nullptr;
  bool used = false;
};
struct DomainGuard {
  WsDomain& d;
  hipStream_t s;
  DomainGuard(WsDomain& dom, hipStream_t stream) : d(dom), s(stream) {
    d.mu.lock();
    if (!d.ev) hipEventCreateWithFlags(&d.ev, hipEventDisableTiming);
    if (d.used && d.last != s) hipStreamWaitEvent(s, d.ev, 0);
  }
  ~DomainGuard() {
    hipEventRecord(d.ev, s);
    d.last = s;
    d.used = true;
    d.mu.unlock();
  }
};

struct ResizeTable {                 // resize.hip: per (kind, in, out) size triple; the host copies back the asynchronous upload
  int* bounds = nullptr; void* coef = nullptr; int ksize = 0;
  std::vector<int> h_bounds; std::vector<char> h_coef;
};

enum { LIN_QKV = 0, LIN_SC = 1, LIN_PROJ = 2, LIN_FC1 = 3, LIN_FC2 = 4 };     // linear kinds of a Hiera block (f16s plan)
// linears outside the Hiera blocks, by the forward function that launches them (f16s plan): patch embedding + FPN neck,
// memory attention, mask decoder (+ prompt encoder), memory encoder.  The running function declares its group with a PlanGroup
// guard (thread-local: the encoder and the tracking domain may run on two host threads).
enum { GRP_NECK = 0, GRP_MA = 1, GRP_DEC = 2, GRP_MENC = 3 };
extern thread_local int tl_plan_group;
struct PlanGroup {
  int prev;
  explicit PlanGroup(int g) : prev(tl_plan_group) { tl_plan_group = g; }
  ~PlanGroup() { tl_plan_group = prev; }
};

struct sam2mi_ctx {
  sam2mi_config cfg;
  WsDomain dom_enc, dom_track;
  std::unordered_map<uint64_t, ResizeTable> resize_tables;
  void* resize_tmp = nullptr;
  size_t resize_tmp_bytes = 0;
  std::mutex misc_mu;               // ctx->allocs (run-time allocations may come from both domains)
  std::mutex prof_mu;               // the profiling accumulators below
  bool finalized = false;
  std::unordered_map<std::string, HostW> hw;     // host copies until finalize
  std::vector<void*> allocs;

  // ---- image encoder
  std::vector<HieraBlockW> blocks;
  Lin16 patch;                 // [embed, 160] (K padded from 147)
  float* pos_tab = nullptr;    // [G*G, embed] f32, window-major (w = window_spec[0]) token order
  Lin16 neck[4];               // index = level (0: stride 4 ... 3: stride 32)
  Lin16 conv_s0, conv_s1;
  Lin16 neck_s0, neck_s1;      // conv_s0 o neck[0] (144 -> 32) and conv_s1 o neck[1] (288 -> 64) composed at load time (both are 1x1 convs with
                               // nothing in between: levels 0 / 1 take no top-down term, fpn_top_down_levels = [2, 3])
  float* sine_pe[3] = {nullptr, nullptr, nullptr};   // NCHW [256, S, S] for S = 256, 128, 64
  float* sine_pe_tok64 = nullptr;                    // the 64x64 table token-major [4096, 256]
  float* no_mem_embed = nullptr;                     // [256]
  // ---- memory attention
  std::vector<MemAttnLayerW> mal;
  Lin16 cross_k_all, cross_v_all;    // [4*256, 64]
  half_t* t_vinT16 = nullptr;        // [TRACK_MAX_N][64, t_nk_cap] memory tokens transposed (V^T operand of the cross-attention, DV = 64)
  bool use_mem_space_values = true;  // the cross-attention multiplies the probabilities with the memory tokens, Wo Wv behind it (fused tail only)
  Norm ma_norm;
  float* rope_cos = nullptr;         // [4096, 128]
  float* rope_sin = nullptr;
  float* qs_self = nullptr;          // [768] q columns *= 256^-0.5*log2(e) (softmax runs in the exp2 domain)
  float* qs_cross = nullptr;         // [256]
  // ---- SAM heads
  std::vector<DecLayerW> dec;
  Lin32 fin_q, fin_o;
  Lin16 fin_k, fin_v;
  Norm fin_norm;
  float* out_tokens = nullptr;       // [6, 256] obj_score | iou | 4 mask tokens
  Lin16 dc1, dc2;                    // ConvTranspose as GEMM: [4*64, 256], [4*32, 64]
  float* dc1_b = nullptr; float* dc2_b = nullptr;
  Norm up_ln;
  Lin32 hyper[4][3], iou_head[3], obj_head[3], ptr_proj[3];
  Lin32 tpos_proj;                   // obj_ptr_tpos_proj (64, 256)
  float* no_obj_ptr = nullptr;
  float* gauss = nullptr;            // [2, 128]
  float* point_emb4 = nullptr;       // [4, 256]
  float* not_a_point = nullptr;      // [256]
  float* no_mask_embed = nullptr;    // [256]
  float* dense_pe = nullptr;         // [4096, 256] token-major
  // ---- memory encoder
  float* md_w[3]; float* md_b[3]; Norm md_ln[4];   // direct convs 1->4, 4->16, 16->64
  Lin16 md_conv3;                     // [64, 144]  (k = (ky*3+kx)*16 + c)
  Lin16 md_conv4;                     // [256, 576] (k = (ky*3+kx)*64 + c)
  Lin16 md_proj, pix_proj, me_out;    // 1x1 convs
  struct CX { float* dw_w; float* dw_b; Norm ln; Lin16 pw1, pw2; float* gamma; } cx[2];
  float* mem_pos = nullptr;           // sine PE 64 feats, token-major [4096, 64]
  float* mem_pos_nchw = nullptr;      // [64, 64, 64]
  float* tpos_enc = nullptr;          // [7, 64]
  float* no_obj_embed_spatial = nullptr;   // [64]

  // ---- precision mode (sam2mi_config.precision).  f16x3: every f16 activation buffer below lives in ONE arena whose second
  // half holds the lo planes, so `lo16` (elements) is the hi -> lo distance of all of them; 0 in the default f16 mode.
  bool precise = false;        // f16x3 or f16s: lo planes exist (arena, packed weights)
  bool selective = false;      // f16s: per-linear plan (GemmParams.prec); everything not planned runs as plan_other
  // the f16s plan: PREC_* per (Hiera stage 1-4, linear kind LIN_*) and for every linear outside the trunk; split_attn: stage-3 attention on
  // split q / k (attn_hiera.hip) instead of attn_precise.hip.  Defaults in engine_core.hip (sam2mi_create); SAM2MI_F16S_PLAN overrides
  // entries for precision experiments ("s3.qkv=f16,s12.all=w,other=f16,attn=0"; tools/f16s_plan_sweep.sh).
  int plan[5][5] = {};
  signed char plan_blk[64][5];      // per-block overrides of plan[][] (-1: none); plan keys "b<lo>-<hi>.<kind>"
  int plan_grp[4] = {3, 3, 3, 3};      // GRP_*: linears outside the Hiera blocks
  bool split_attn = true;
  bool split_attn_global = false;   // ... also in the three global-attention blocks (plan key "gattn"; off: 4,096-key soft-maxes average the q / k rounding out)
  bool split_attn_stage[5] = {true, true, false, false, false};    // ... per Hiera stage (plan keys "attn1" .. "attn4"): off = plain f16 q / k
                                    // (stages 3-4 off: 256-key windows average the rounding out - 7.59e-4 / 6.69e-4 vs 7.59e-4 / 6.38e-4 on the
                                    // golden, 6.07e-4 / 5.28e-4 on the second one, +4.7 % frames/s; stages 1-2 keep it: 16- and 64-key windows)
  size_t lo16 = 0;
  float* ws_qk32 = nullptr;    // f16x3 mode: q|k and V^T of the Hiera blocks in f32 (operands of attn_precise.hip)
  float* ws_vT32 = nullptr;
  float* ws_qp32 = nullptr;

  // ---- model family: hiera-large (window spec 8/4/16/8, head_dim 72: windows tile the grid - the window-major path of
  // engine_encoder.hip) or the padded-window sizes tiny / small / base+ (8/4/14/7, head_dim 96 / 56: hiera_generic.hip)
  bool generic = false;
  int head_dim = 72;
  half_t* ws_w16 = nullptr;    // generic path: LN output gathered into the (padded) window layout
  half_t* ws_o16 = nullptr;    // generic path: attention output in the window layout

  // ---- workspaces (sized for cfg.max_batch frames)
  float* ws_x = nullptr;        // residual stream f32
  float* ws_x2 = nullptr;       // second f32 buffer (shortcut / permute target)
  half_t* ws_a16 = nullptr;     // LN output / generic f16 operand
  half_t* ws_qk16 = nullptr;    // [M, 2C]
  half_t* ws_vT16 = nullptr;    // [C, M]
  half_t* ws_att16 = nullptr;   // [M, C]
  half_t* ws_h16 = nullptr;     // [M, 4C]
  half_t* ws_qp16 = nullptr;    // pooled q
  float* ws_lat[4] = {nullptr, nullptr, nullptr, nullptr};   // neck outputs f32: [M_0, 32] (conv_s0), [M_1, 64] (conv_s1), laterals [M_2, 256], [M_3, 256]
  size_t ws_tokens = 0;         // max tokens (max_batch * G*G)

  // per-frame (B = 1) tracking workspaces
  float* t_x = nullptr; half_t* t_h16 = nullptr; half_t* t_qk16 = nullptr; half_t* t_vT16 = nullptr;
  half_t* t_o16 = nullptr; half_t* t_q16 = nullptr; half_t* t_ff16 = nullptr;
  half_t* t_kin16 = nullptr; half_t* t_vin16 = nullptr; half_t* t_kall16 = nullptr; half_t* t_vTall16 = nullptr;
  float* t_opart = nullptr; float* t_ml = nullptr;
  int t_nk_cap = 0;             // padded key capacity
  float* t_ptr_tok = nullptr; float* t_ptr_pos = nullptr;
  float* t_pix = nullptr;       // memory-conditioned features [4096, 256]
  // decoder
  float* d_keys = nullptr; half_t* d_keys16 = nullptr; half_t* d_kpe16 = nullptr;
  float* d_t2i_part = nullptr; size_t d_t2i_part_floats = 0;      // split partials of the token -> image attention (attn_small.hip)
  float* d_tok = nullptr; float* d_tokpe = nullptr; float* d_t1 = nullptr; float* d_t2 = nullptr; float* d_t3 = nullptr;
  float* d_t4 = nullptr; float* d_big1 = nullptr; float* d_big2 = nullptr; float* d_big3 = nullptr; half_t* d_big16 = nullptr;
  float* d_tokens_in = nullptr; float* d_sparse = nullptr;
  float* track_tokens = nullptr;   // [TRACK_MAX_N, 8, 256] constant decoder tokens of a prompt-free tracked frame (engine_core.hip)
  half_t* d_up1_16 = nullptr; half_t* d_up2_16 = nullptr; float* d_g = nullptr;
  float* d_hyper = nullptr; half_t* d_hyper16 = nullptr;
  float* d_fill_tmp = nullptr;     // [65536] hole-filling scratch
  // mask prompts (add_new_mask / correction clicks)
  MaskEmbedW mask_embed{};         // sam_prompt_encoder.mask_downscaling.{0,1,3,4,6}
  float* mds_w = nullptr; float* mds_b = nullptr;     // SAM2Base.mask_downsample (Conv2d 1->1, k4 s4)
  float* d_mask256 = nullptr;      // [65536] mask prompt at the prompt encoder's input size
  float* d_dense = nullptr;        // [4096,256] dense prompt embedding of a mask prompt
  float* d_pm10 = nullptr;         // [2]: {+10 / -10 object score of a mask input, scratch}
  int* d_flag = nullptr;
  int fill_hole_area = 0;          // sam2mi_set_fill_hole_area: 0 = off (SAM2Base.fill_hole_area, build_sam.py:129)
  float* d_masks = nullptr; float* d_iou = nullptr; float* d_obj = nullptr;
  int dec_T = 0;                   // tokens per prompt of the last decoder pass (layout of d_tok)
  float* d_low_multi = nullptr; float* d_low_sel = nullptr; float* d_tok_sel = nullptr; int* d_best = nullptr; float* d_iou_sel = nullptr;
  float* d_ptr = nullptr; float* d_pts = nullptr; int* d_labels = nullptr;
  // memory encoder
  float* m_mask = nullptr; float* m_c1 = nullptr; float* m_c2 = nullptr; half_t* m_c2_16 = nullptr; half_t* m_c3_16 = nullptr; half_t* m_col16 = nullptr;
  float* m_c4 = nullptr; half_t* m_c4_16 = nullptr; float* m_emb = nullptr; float* m_x = nullptr; float* m_dw = nullptr;
  half_t* m_ln16 = nullptr; half_t* m_h16 = nullptr; float* m_out = nullptr; half_t* m_pix16 = nullptr;
  // plug-boundary scratch (layout conversion)
  float* p_a = nullptr; float* p_b = nullptr; float* p_c = nullptr; float* p_d = nullptr;

  // ---- video state
  struct FeatSlot { float* feat2; float* fpn1; float* fpn0; };   // token-major row-major
  std::vector<FeatSlot> feats;
  struct BankSlot { float* mem; float* obj_ptr; float* obj_score; float* low_mask; };
  std::vector<BankSlot> bank;

  // ---- profiling
  bool prof_on = false;
  ProfAcc prof_gemm, prof_attn, prof_mlp, prof_xs, prof_ks;
  std::map<std::string, ProfAcc> prof_by_kernel;   // the GEMM-family launches again, keyed by kernel instantiation (rocprofv3 names)
  bool use_ks = false;             // accumulator-stationary GEMM for stage-3 fc2 (experimental, SAM2MI_KS=1; parity-tested, not faster end to end)
  bool use_xs = true;              // X-stationary GEMM for K <= 576 linears of the encoder (SAM2MI_NO_XS=1: tiled kernel)
  int enc_sub = 0;                 // frames per sub-batch of Hiera stages 1-2 inside an encoder pass (0: whole batch, the default - no gain measured; SAM2MI_ENC_SUB)
  bool ln_fuse = false;            // LN1 / LN2 of Hiera blocks computed inside the consumer's operand load (opt-in: SAM2MI_LN_FUSE=1; no end-to-end gain)
  int ln1_fuse_maxc = 0;           // norm1 inside the operand load of the X-stationary QKV kernel for blocks with dim <= this (SAM2MI_LN1_FUSE_MAXC; 0: off)
  bool use_projln = true;          // Hiera stages 1-3: out-projection + residual + norm2 in one kernel (SAM2MI_NO_PROJLN=1: GEMM + LayerNorm)
  bool use_rowln = true;           // memory attention: combine + out-projection + residual + next LayerNorm in one kernel (SAM2MI_NO_ROWLN=1: three kernels)
  bool use_fused_mlp = true;       // stages with C <= 288: one fused fc1-GELU-fc2 kernel (SAM2MI_NO_FUSED_MLP=1: two GEMMs, for A/B runs)
};

int sam2mi_set_error(sam2mi_ctx* ctx, const char* what, const char* detail);
#define CHK(expr)                                                               \
  do {                                                                          \
    hipError_t _e = (expr);                                                     \
    if (_e != hipSuccess) return sam2mi_set_error(ctx, #expr, hipGetErrorString(_e)); \
  } while (0)
#define CHKI(expr)                 \
  do {                             \
    int _r = (expr);               \
    if (_r != 0) return _r;        \
  } while (0)

// engine_core.hip
void* dalloc(sam2mi_ctx* ctx, size_t bytes);
void* dalloc_raw(sam2mi_ctx* ctx, size_t bytes);       // not cleared; may be released with dfree before sam2mi_destroy
void dfree(sam2mi_ctx* ctx, void* p);
int run_gemm(sam2mi_ctx* ctx, hipStream_t s, const GemmParams& p);                 // with profiling
int run_rowln(sam2mi_ctx* ctx, hipStream_t s, const RowLnParams& p);               // gemm_rowln.hip, with profiling
int run_projln(sam2mi_ctx* ctx, hipStream_t s, const ProjLnParams& p);             // gemm_projln.hip, with profiling
bool xs_eligible(const sam2mi_ctx* ctx, const GemmParams& p);                      // will run_gemm take the X-stationary kernel?
int run_hiera_attn(sam2mi_ctx* ctx, hipStream_t s, const HieraAttnParams& p);
int run_mlp_fused(sam2mi_ctx* ctx, hipStream_t s, const MlpFusedParams& p, int C);     // with profiling
int run_flash256(sam2mi_ctx* ctx, hipStream_t s, const Flash256Params& p);
int run_precise_attn(sam2mi_ctx* ctx, hipStream_t s, const PreciseAttnParams& p);
GemmParams lin_params(const half_t* A, int lda, int M, const Lin16& L);            // bias + W filled, n_split = N

// engine_encoder.hip
struct EncOut { float* feat2; float* fpn1; float* fpn0; };   // token-major [B, HW, C]
// img: normalised f32 NCHW, or (img == nullptr) img_u8: decoded uint8 HWC frames normalised on the fly
int encoder_forward(sam2mi_ctx* ctx, hipStream_t s, const float* img, int B, const EncOut* outs /*[B]*/, const uint8_t* img_u8 = nullptr);
int hiera_block_forward(sam2mi_ctx* ctx, hipStream_t s, const HieraBlockW& blk, int B, int& H, int& W, int& wcur);
int hiera_block_forward_generic(sam2mi_ctx* ctx, hipStream_t s, const HieraBlockW& blk, int B, int& H, int& W);   // row-major tokens

// engine_track.hip
constexpr int TRACK_MAX_N = 8;    // objects per batched tracking call (workspace size)
int memattn_forward(sam2mi_ctx* ctx, hipStream_t s, const float* curr, const float* curr_pos, int N, const int* Nk, const int* n_rope,
                    float* out32);
constexpr int DEC_MAX_N = 16;     // prompts / objects per batched mask-decoder call (workspace size)
struct DecoderIn {
  const float* keys_tok; size_t keys_stride;                    // floats between prompts; 0: one image for all (repeat_image)
  const float* dense_tok; int dense_rows; size_t dense_stride;  // dense prompt embedding (null: none)
  const float* pos_tok; bool pos_shared;                         // [4096,256] shared by all prompts, or [N,4096,256]
  const float* tokens;                                           // [N,T,256]
  const float* hr0_tok; size_t hr0_stride;                       // [65536,32] per prompt, stride 0: shared
  const float* hr1_tok; size_t hr1_stride;                       // [16384,64]
};
int decoder_forward(sam2mi_ctx* ctx, hipStream_t s, const DecoderIn& in, int N, int T);
// mask1024 != null: explicit (already sigmoid-scaled) 1024^2 mask (plug); else low256 + binarize: the fused video path
int memenc_forward(sam2mi_ctx* ctx, hipStream_t s, const float* feat2_tok, const float* mask1024, float* out_tok64,
                   const float* low256 = nullptr, int binarize = 0);

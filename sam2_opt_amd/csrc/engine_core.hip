// Context, weight packing, input-independent tables and profiled launch wrappers.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "engine.h"

// Error text is per THREAD (errno-style): one context may be driven from several host threads (encoder and tracking domains run
// concurrently by design), and the thread whose call returned != 0 is the one that asks for the message.
static thread_local std::string t_last_error;

int sam2mi_set_error(sam2mi_ctx* ctx, const char* what, const char* detail) {
  (void)ctx;
  t_last_error = std::string(what) + ": " + detail;
  return 1;
}

void* dalloc(sam2mi_ctx* ctx, size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  hipMemset(p, 0, bytes);     // finite contents everywhere (pad regions are read by MFMA tiles)
  std::lock_guard<std::mutex> lk(ctx->misc_mu);      // run-time allocations (resize.hip) may come from two domains at once
  ctx->allocs.push_back(p);
  return p;
}

// run-time allocations (resize.hip): no clearing pass, releasable before sam2mi_destroy
void* dalloc_raw(sam2mi_ctx* ctx, size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(ctx->misc_mu);
  ctx->allocs.push_back(p);
  return p;
}
void dfree(sam2mi_ctx* ctx, void* p) {
  {
    std::lock_guard<std::mutex> lk(ctx->misc_mu);
    for (size_t i = 0; i < ctx->allocs.size(); ++i)
      if (ctx->allocs[i] == p) { ctx->allocs[i] = ctx->allocs.back(); ctx->allocs.pop_back(); break; }
  }
  hipFree(p);          // waits for the device: nothing still reads the buffer
}

template <typename T>
static T* dupload(sam2mi_ctx* ctx, const std::vector<T>& v) {
  T* p = (T*)dalloc(ctx, v.size() * sizeof(T));
  if (p && !v.empty()) hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

// ------------------------------------------------------------------ profiling wrappers
static void prof_drain(ProfAcc& a) {          // synchronises; profiling mode only
  for (auto& pr : a.pending) {
    hipEventSynchronize(pr.second);
    float ms = 0;
    hipEventElapsedTime(&ms, pr.first, pr.second);
    a.ms += ms;
    if (pr.named) pr.named->ms += ms;
    a.pool.push_back({pr.first, pr.second});
  }
  a.pending.clear();
}
// The accumulators are shared by the encoder and the tracking domain (two host threads may profile at once): every access
// below holds ctx->prof_mu.  prof_begin .. prof_end* of one launch are two short critical sections, not one.
static void prof_begin(sam2mi_ctx* ctx, ProfAcc& a, hipStream_t s, hipEvent_t& e0, hipEvent_t& e1) {
  std::lock_guard<std::mutex> lk(ctx->prof_mu);
  if (a.pool.empty()) {
    // drain pending (synchronises; profiling mode only)
    prof_drain(a);
    if (a.pool.empty()) {
      for (int i = 0; i < 2048; ++i) {
        hipEvent_t x, y;
        hipEventCreate(&x);
        hipEventCreate(&y);
        a.pool.push_back({x, y});
      }
    }
  }
  auto pr = a.pool.back();
  a.pool.pop_back();
  e0 = pr.first;
  e1 = pr.second;
  hipEventRecord(e0, s);
}
static void prof_end(sam2mi_ctx* ctx, ProfAcc& a, hipStream_t s, hipEvent_t e0, hipEvent_t e1, double flops) {
  std::lock_guard<std::mutex> lk(ctx->prof_mu);
  hipEventRecord(e1, s);
  a.pending.push_back({e0, e1, nullptr});
  a.flops += flops;
  a.launches += 1;
}

// second accumulator keyed by the kernel instantiation: same events, drained together in sam2mi_profile_read_kernels
static void prof_end_named(sam2mi_ctx* ctx, ProfAcc& a, const std::string& name, hipStream_t s, hipEvent_t e0, hipEvent_t e1, double flops,
                           double bytes = 0) {
  std::lock_guard<std::mutex> lk(ctx->prof_mu);
  ProfAcc& k = ctx->prof_by_kernel[name];       // std::map: references stay valid
  k.flops += flops;
  k.bytes += bytes;
  k.launches += 1;
  hipEventRecord(e1, s);
  a.pending.push_back({e0, e1, &k});
  a.flops += flops;
  a.launches += 1;
}

// X-stationary kernel for the encoder's short-K linears when the operands allow it
bool xs_eligible(const sam2mi_ctx* ctx, const GemmParams& p) {
  static const int min_k = getenv("SAM2MI_XS_MINK") ? atoi(getenv("SAM2MI_XS_MINK")) : 0;      // A/B aid
  static const int min_m = getenv("SAM2MI_XS_MINM") ? atoi(getenv("SAM2MI_XS_MINM")) : 16384;      // A/B aid: 8192 takes batch-2 encoder calls from 9.67 to 9.27 ms (batch 1, M = 4096: 6.24 -> 6.39), but then a 2-frame and a 1-frame pass of the same frame differ in the last bits
  if (p.w_lo_off) {            // weight split (f16s): the QKV shape only - f16 outputs, no activation, the V^T consumer reads the hi plane
    if (!p.xs_wpack || p.act != ACT_NONE || p.ln_x32 || (p.outT16 && p.out_lo_off && !p.outT_hi_only)) return false;
  } else if (p.out_lo_off || !p.xs_pack) return false;
  return ctx->use_xs && !p.a_lo_off && p.pool_w == 0 && p.K >= min_k && p.tile_hint == 0 && p.M >= min_m && (p.ln_x32 ? p.ln_ld == p.K : p.lda == p.K) && gemm_xs_supported(p.N, p.K) &&
         (p.act == ACT_NONE || p.act == ACT_GELU) && p.rope_cols == 0 && p.res_mod == 0 && !p.outT32 && (p.n_split >= p.N || (p.n_split & 31) == 0) &&
         !(p.out32 && p.out16) && (p.out32 || p.out16) && (!p.res || p.out32) && p.bias &&
         (!p.col_scale || (p.xs_scale_cols > 0 && p.xs_scale_cols <= 576 && (p.xs_scale_cols + 31) / 32 * 32 <= p.n_split));
}
static bool ks_eligible(const sam2mi_ctx* ctx, const GemmParams& p) {
  return ctx->use_ks && p.pool_w == 0 && p.ks_pack && p.tile_hint == 0 && p.M >= 16384 && gemm_ks_supported(p.N, p.K) && p.act == ACT_NONE &&
         !p.col_scale && p.rope_cols == 0 && p.res_mod == 0 && p.out32 && !p.out16 && !p.outT16 && !p.outT32 && p.n_split >= p.N && p.bias;
}
// algorithmic HBM bytes of one linear: both operands read once (x2 planes in the split mode), every output written once, the
// f32 residual read once
static double gemm_algo_bytes(const GemmParams& p) {
  const double mk = (double)p.M * p.K, nk = (double)p.N * p.K, mn = (double)p.M * p.N;
  const double planes = p.a_lo_off ? 2.0 : 1.0;
  double b = (p.ln_x32 ? 4.0 * mk : 2.0 * mk * planes) + 2.0 * nk * planes;
  const double row_cols = std::min(p.N, p.n_split), t_cols = p.N - row_cols;
  const double out_rows = p.pool_w ? p.M / 4.0 : (double)p.M;       // fused 2x2 max-pool: a quarter of the rows is written
  if (p.out32) b += 4.0 * out_rows * row_cols;
  if (p.out16) b += 2.0 * out_rows * row_cols * (p.out_lo_off ? 2.0 : 1.0);
  if (p.outT16) b += 2.0 * p.M * t_cols * (p.out_lo_off ? 2.0 : 1.0);
  if (p.outT32) b += 4.0 * p.M * t_cols;
  if (p.res) b += 4.0 * mn;
  return b;
}

thread_local int tl_plan_group = GRP_NECK;

int run_gemm(sam2mi_ctx* ctx, hipStream_t s, const GemmParams& p_in) {
  GemmParams p = p_in;
  if (ctx->precise) {            // split operands (activations: arena lo plane; weights: packed [hi | lo], or an arena buffer)
    const int prec = !ctx->selective ? PREC_FULL : (p.prec != PREC_AUTO ? p.prec : ctx->plan_grp[tl_plan_group]);      // f16x3: every linear fully split
    if (prec == PREC_FULL) {
      if (!p.a_lo_off) p.a_lo_off = ctx->lo16;
      if (!p.w_lo_off) p.w_lo_off = ctx->lo16;
    } else {
      p.a_lo_off = 0;
      if (prec == PREC_F16) p.w_lo_off = 0;
      else if (!p.w_lo_off) p.w_lo_off = ctx->lo16;
    }
    p.out_lo_off = ((p.out16 || p.outT16) && !(ctx->selective && p.no_out_lo)) ? ctx->lo16 : 0;
  }
  if (ks_eligible(ctx, p)) {
    GemmKsParams k{p.A, p.lda, p.ks_pack, p.bias, p.res, p.ldres, p.out32, p.ld32, p.M, p.K};
    hipEvent_t e0, e1;
    if (ctx->prof_on) prof_begin(ctx, ctx->prof_ks, s, e0, e1);
    CHK(gemm_ks_launch(k, s));
    if (ctx->prof_on) prof_end_named(ctx, ctx->prof_ks, "gemm_ks_kernel<0>", s, e0, e1, 2.0 * p.M * (double)p.N * p.K, gemm_algo_bytes(p));
    return 0;
  }
  if (xs_eligible(ctx, p)) {
    const bool ws = p.w_lo_off != 0;
    GemmXsParams x{p.A, p.lda, ws ? p.xs_wpack : p.xs_pack, p.bias, p.col_scale, p.xs_scale_cols, p.act, p.M, p.N, p.n_split, p.out16, p.ld16, p.outT16, p.ldT16,
                   p.out32, p.ld32, p.res, p.ldres, 0, p.ln_x32, p.ln_ld, p.ln_eps, ws ? 1 : 0, p.out_lo_off};
    hipEvent_t e0, e1;
    if (ctx->prof_on) prof_begin(ctx, ctx->prof_xs, s, e0, e1);
    CHK(gemm_xs_launch(x, p.K, s));
    if (ctx->prof_on) {
      char nm[96];
      snprintf(nm, sizeof(nm), "gemm_xs_kernel<%d, %s, %s, 0%s>", p.K, p.act == ACT_GELU ? "true" : "false", p.out32 ? "true" : "false", ws ? ", true" : "");
      prof_end_named(ctx, ctx->prof_xs, nm, s, e0, e1, 2.0 * p.M * (double)p.N * p.K, gemm_algo_bytes(p));
    }
    return 0;
  }
  if (p.ln_x32) return sam2mi_set_error(ctx, "run_gemm", "LayerNorm-fused operand on a shape the X-stationary kernel does not take");
  hipEvent_t e0, e1;
  if (ctx->prof_on) prof_begin(ctx, ctx->prof_gemm, s, e0, e1);
  CHK(gemm_launch(p, s));
  if (ctx->prof_on) prof_end_named(ctx, ctx->prof_gemm, gemm_v2_kernel_name(p), s, e0, e1, 2.0 * p.M * (double)p.N * p.K, gemm_algo_bytes(p));
  return 0;
}
int run_mlp_fused(sam2mi_ctx* ctx, hipStream_t s, const MlpFusedParams& p, int C) {
  hipEvent_t e0, e1;
  if (ctx->prof_on) prof_begin(ctx, ctx->prof_mlp, s, e0, e1);
  CHK(mlp_fused_launch(p, C, s));
  // fused MLP: X (f16, or the f32 residual row when LayerNorm is fused) + the f32 residual in and out + both weight matrices
  if (ctx->prof_on) prof_end_named(ctx, ctx->prof_mlp, mlp_fused_kernel_name(C), s, e0, e1, 2.0 * 2.0 * p.M * (double)C * (4.0 * C),
                                   (p.ln_eps > 0.f ? 0.0 : 2.0 * p.M * (double)C) + 8.0 * p.M * (double)C + 2.0 * 8.0 * C * (double)C);
  return 0;
}
int run_hiera_attn(sam2mi_ctx* ctx, hipStream_t s, const HieraAttnParams& p) {
  hipEvent_t e0, e1;
  if (ctx->prof_on) prof_begin(ctx, ctx->prof_attn, s, e0, e1);
  CHK(hiera_attn_launch(p, s));
  // algorithmic flops: 4 * Nq * Nk_visible * 72 per head
  const double nk_vis = (p.wq >= p.GQ) ? p.GK : p.wk;
  // algorithmic bytes: q, k (two planes each in the split mode), V^T in, the output (hi + lo in the split mode) out - all f16
  const double cols = 72.0 * p.heads, planes = p.qk_lo_off ? 2.0 : 1.0;
  const double bytes = 2.0 * p.num_groups * cols * (planes * (p.GQ + p.GK) + p.GK + (p.o_lo_off ? 2.0 : 1.0) * p.GQ);
  if (ctx->prof_on) prof_end_named(ctx, ctx->prof_attn, hiera_attn_kernel_name(p), s, e0, e1, 4.0 * p.num_groups * (double)p.GQ * nk_vis * 72.0 * p.heads, bytes);
  return 0;
}
int run_precise_attn(sam2mi_ctx* ctx, hipStream_t s, const PreciseAttnParams& p) {
  hipEvent_t e0, e1;
  if (ctx->prof_on) prof_begin(ctx, ctx->prof_attn, s, e0, e1);
  CHK(precise_attn_launch(p, s));
  const double nk_vis = (p.wq >= p.GQ) ? p.GK : p.wk;
  if (ctx->prof_on) prof_end(ctx, ctx->prof_attn, s, e0, e1, 4.0 * p.num_groups * (double)p.GQ * nk_vis * 72.0 * p.heads);
  return 0;
}
int run_rowln(sam2mi_ctx* ctx, hipStream_t s, const RowLnParams& p) {
  hipEvent_t e0, e1;
  if (ctx->prof_on) prof_begin(ctx, ctx->prof_gemm, s, e0, e1);
  CHK(gemm_rowln_launch(p, s));
  // algorithmic bytes: partials (or the f16 operand) + weights + residual in and out + f16 LayerNorm output
  const double kc = p.kc == 64 ? 64.0 : 256.0;
  const double bytes = (p.o_part ? (double)p.splits * p.M * (kc * 4 + 8) : p.M * kc * 2.0) + 256.0 * kc * 2 + p.M * 256.0 * (4 + 4 + 2);
  if (ctx->prof_on) prof_end_named(ctx, ctx->prof_gemm, p.kc == 64 ? "gemm_rowln_kernel<64>" : "gemm_rowln_kernel<256>", s, e0, e1, 2.0 * p.M * 256.0 * kc, bytes);
  return 0;
}

int run_projln(sam2mi_ctx* ctx, hipStream_t s, const ProjLnParams& p) {
  hipEvent_t e0, e1;
  if (ctx->prof_on) prof_begin(ctx, ctx->prof_gemm, s, e0, e1);
  CHK(gemm_projln_launch(p, s));
  if (ctx->prof_on) {
    char nm[64];
    snprintf(nm, sizeof(nm), p.wpack_lo ? "gemm_projln_kernel<%d, %d>" : "gemm_projln_kernel<%d>", p.C, p.a_lo_off ? 3 : 2);
    // algorithmic bytes: f16 operand + weights + f32 residual in and out + f16 LayerNorm output
    prof_end_named(ctx, ctx->prof_gemm, nm, s, e0, e1, 2.0 * p.M * (double)p.C * p.C, (double)p.M * p.C * (2 + 4 + 4 + 2) + 2.0 * p.C * p.C);
  }
  return 0;
}

int run_flash256(sam2mi_ctx* ctx, hipStream_t s, const Flash256Params& p) {
  hipEvent_t e0, e1;
  if (ctx->prof_on) prof_begin(ctx, ctx->prof_attn, s, e0, e1);
  CHK(flash256_launch(p, s));
  // algorithmic bytes: q, K, V^T in (f16), the un-normalised f32 partial outputs + their (max, sum) pairs out (one set per KV split)
  if (ctx->prof_on) {
    const double dv = p.dv == 64 ? 64.0 : 256.0;
    const char* nm = flash256_kernel_name(p);
    prof_end_named(ctx, ctx->prof_attn, nm, s, e0, e1, 2.0 * p.Nq * (double)p.Nk * (256.0 + dv),
                   512.0 * p.Nq + 2.0 * p.Nk * (256.0 + dv) + (double)p.splits * p.Nq * (dv * 4 + 8));
  }
  return 0;
}

GemmParams lin_params(const half_t* A, int lda, int M, const Lin16& L) {
  GemmParams p = gemm_params_zero();
  p.A = A; p.lda = lda; p.W = L.w; p.ldw = L.K; p.M = M; p.N = L.N; p.K = L.K; p.bias = L.b; p.n_split = L.N; p.xs_pack = L.xs_pack; p.xs_wpack = L.xs_wpack; p.ks_pack = L.ks_pack;
  p.w_lo_off = L.lo_off;
  return p;
}

// ------------------------------------------------------------------ weight access helpers
namespace {
struct Packer {
  sam2mi_ctx* ctx;
  bool ok = true;
  std::string missing;
  const HostW* get(const std::string& k) {
    auto it = ctx->hw.find(k);
    if (it == ctx->hw.end()) {
      ok = false;
      if (missing.size() < 400) missing += k + " ";
      return nullptr;
    }
    return &it->second;
  }
  float* f32(const std::string& k) {
    const HostW* w = get(k);
    return w ? dupload(ctx, w->data) : nullptr;
  }
  Norm norm(const std::string& p) {
    Norm n;
    const HostW* w = get(p + ".weight");
    n.w = f32(p + ".weight");
    n.b = f32(p + ".bias");
    n.C = w ? (int)w->data.size() : 0;
    return n;
  }
  Lin16 lin16_raw(const std::vector<float>& W, const std::vector<float>& b, int N, int K) {
    Lin16 l;
    const size_t n = (size_t)N * K;
    std::vector<half_t> h(ctx->precise ? 2 * n : n);
    for (size_t i = 0; i < n; ++i) {
      h[i] = (half_t)W[i];
      if (ctx->precise) h[n + i] = (half_t)((W[i] - (float)h[i]) * 2048.0f);      // lo plane (common.h: SPLIT_SCALE)
    }
    if (ctx->precise) l.lo_off = n;
    l.w = dupload(ctx, h);
    l.b = dupload(ctx, b);
    l.N = N;
    l.K = K;
    return l;
  }
  // nn.Linear / 1x1 conv: weight (N, K[,1,1])
  Lin16 lin16(const std::string& p) {
    const HostW* w = get(p + ".weight");
    const HostW* b = get(p + ".bias");
    if (!w || !b) return Lin16();
    const int N = (int)w->shape[0];
    const int K = (int)(w->data.size() / N);
    return lin16_raw(w->data, b->data, N, K);
  }
  Lin32 lin32(const std::string& p) {
    Lin32 l;
    const HostW* w = get(p + ".weight");
    if (!w) return l;
    l.N = (int)w->shape[0];
    l.K = (int)(w->data.size() / l.N);
    l.w = f32(p + ".weight");
    l.b = f32(p + ".bias");
    return l;
  }
  Lin16 concat16(const std::vector<std::string>& ps) {
    std::vector<float> W, B;
    int K = 0, N = 0;
    for (auto& p : ps) {
      const HostW* w = get(p + ".weight");
      const HostW* b = get(p + ".bias");
      if (!w || !b) return Lin16();
      K = (int)(w->data.size() / w->shape[0]);
      N += (int)w->shape[0];
      W.insert(W.end(), w->data.begin(), w->data.end());
      B.insert(B.end(), b->data.begin(), b->data.end());
    }
    return lin16_raw(W, B, N, K);
  }
};

inline float cubic1(float x, float A) { return ((A + 2) * x - (A + 3)) * x * x + 1; }
inline float cubic2(float x, float A) { return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A; }

// F.interpolate(mode="bicubic", align_corners=False) of in [C, Hin, Win] -> out [C, Hout, Wout]
void bicubic_resize(const float* in, int C, int Hin, int Win, float* out, int Hout, int Wout) {
  const float A = -0.75f;
  auto prep = [&](int insz, int outsz, std::vector<int>& idx, std::vector<float>& wt) {
    idx.resize((size_t)outsz * 4);
    wt.resize((size_t)outsz * 4);
    const float scale = (float)insz / (float)outsz;
    for (int o = 0; o < outsz; ++o) {
      const float real = scale * (o + 0.5f) - 0.5f;
      int i0 = (int)std::floor(real);
      i0 = std::min(i0, insz - 1);
      float t = std::min(std::max(real - (float)i0, 0.f), 1.f);
      const float c[4] = {cubic2(t + 1.f, A), cubic1(t, A), cubic1(1.f - t, A), cubic2(2.f - t, A)};
      for (int j = 0; j < 4; ++j) {
        idx[o * 4 + j] = std::max(std::min(i0 + j - 1, insz - 1), 0);
        wt[o * 4 + j] = c[j];
      }
    }
  };
  std::vector<int> iy, ix;
  std::vector<float> wy, wx;
  prep(Hin, Hout, iy, wy);
  prep(Win, Wout, ix, wx);
  for (int c = 0; c < C; ++c)
    for (int y = 0; y < Hout; ++y)
      for (int x = 0; x < Wout; ++x) {
        float acc = 0.f;
        for (int i = 0; i < 4; ++i) {
          float row = 0.f;
          for (int j = 0; j < 4; ++j) row += wx[x * 4 + j] * in[((size_t)c * Hin + iy[y * 4 + i]) * Win + ix[x * 4 + j]];
          acc += wy[y * 4 + i] * row;
        }
        out[((size_t)c * Hout + y) * Wout + x] = acc;
      }
}

// PositionEmbeddingSine._pe (position_encoding.py:90-125): NCHW [F, H, W], F = num_pos_feats
std::vector<float> sine_pe_nchw(int H, int W, int F) {
  const int n = F / 2;
  std::vector<float> out((size_t)F * H * W);
  const float two_pi = 2.f * (float)M_PI;
  std::vector<float> dim_t(n);
  for (int j = 0; j < n; ++j) dim_t[j] = std::pow(10000.f, 2.f * (float)(j / 2) / (float)n);
  for (int y = 0; y < H; ++y) {
    const float ye = (float)(y + 1) / ((float)H + 1e-6f) * two_pi;
    for (int x = 0; x < W; ++x) {
      const float xe = (float)(x + 1) / ((float)W + 1e-6f) * two_pi;
      for (int j = 0; j < n; ++j) {
        const float py = ye / dim_t[j], px = xe / dim_t[j];
        out[((size_t)j * H + y) * W + x] = (j & 1) ? std::cos(py) : std::sin(py);
        out[((size_t)(n + j) * H + y) * W + x] = (j & 1) ? std::cos(px) : std::sin(px);
      }
    }
  }
  return out;
}
}  // namespace

extern "C" int sam2mi_abi_version(void) { return 2; }      // 2: sam2mi_config.precision

// The selective-split plan of the f16s precision mode (tools/precision_shares.py and tools/precision_plan_video.py are the
// CPU experiments behind it, tools/f16s_plan_sweep.sh the on-device one).  What carries the f16-mode error of the encoder is, in
// this order: the rounding of WEIGHTS (coherent over all tokens: 4.6x the variance of all activation rounding together), of the
// attention output that feeds the projection (near-uniform attention makes it coherent inside a window), of q / k; stages 1-2
// and the attention linears of stage 3 matter, the MLP of stage 3 (2/3 of the encoder's FLOPs) and stage 4 hardly do.
static bool f16s_plan_init(sam2mi_ctx* ctx) {
  memset(ctx->plan_blk, -1, sizeof(ctx->plan_blk));
  for (int st = 1; st <= 4; ++st)
    for (int k = 0; k < 5; ++k) {
      int v;
      if (st == 4) v = PREC_F16;
      else if (k == LIN_FC1 || k == LIN_FC2) v = PREC_F16;               // the MLPs (2/3 of the encoder's FLOPs) as in the f16 mode
      else if (k == LIN_PROJ && st == 1) v = PREC_FULL;                  // attention output of stage 1: x and W (W only: 8.2e-4 / 7.2e-4)
      else v = PREC_WSPLIT;                                              // QKV, stage-3 projection, transition shortcuts: W
      ctx->plan[st][k] = v;
    }
  // outside the trunk: patch embedding + neck and the mask decoder on full splits (either in f16 alone breaks the 1e-3 bar: 1.3e-3 /
  // 1.1e-3 max-abs on the 24-frame golden), memory attention and memory encoder in f16 (+5e-5 / +2e-5 of rel L2)
  ctx->plan_grp[GRP_NECK] = ctx->plan_grp[GRP_DEC] = PREC_FULL;
  ctx->plan_grp[GRP_MA] = ctx->plan_grp[GRP_MENC] = PREC_F16;
  ctx->split_attn = true;
  ctx->split_attn_global = false;         // measured: 7.59e-4 / 6.38e-4 vs 7.53e-4 / 6.47e-4 with it, +2.1 % frames/s
  // q / k split: stage 1 only (8 x 8 windows of 64 keys feeding a fully split projection); stage 2 without it and with a W-only
  // projection measured 7.47e-4 / 6.32e-4 (second golden 6.31e-4 / 5.51e-4) against 7.59e-4 / 6.69e-4 with both
  ctx->split_attn_stage[1] = true;
  ctx->split_attn_stage[2] = ctx->split_attn_stage[3] = ctx->split_attn_stage[4] = false;
  const char* e = getenv("SAM2MI_F16S_PLAN");
  if (!e) return true;
  std::string str(e);
  size_t pos = 0;
  while (pos < str.size()) {
    size_t end = str.find(',', pos);
    if (end == std::string::npos) end = str.size();
    const std::string tok = str.substr(pos, end - pos);
    pos = end + 1;
    if (tok.empty()) continue;
    const size_t eq = tok.find('=');
    if (eq == std::string::npos) return false;
    const std::string key = tok.substr(0, eq), val = tok.substr(eq + 1);
    if (key == "attn") { ctx->split_attn = val != "0"; continue; }
    if (key == "gattn") { ctx->split_attn_global = val != "0"; continue; }
    if (key.size() == 5 && key.compare(0, 4, "attn") == 0 && key[4] >= '1' && key[4] <= '4') { ctx->split_attn_stage[key[4] - '0'] = val != "0"; continue; }
    const int v = val == "f16" ? PREC_F16 : val == "w" ? PREC_WSPLIT : val == "full" ? PREC_FULL : -1;
    if (v < 0) return false;
    if (key == "other") { for (int g = 0; g < 4; ++g) ctx->plan_grp[g] = v; continue; }
    if (key == "neck") { ctx->plan_grp[GRP_NECK] = v; continue; }
    if (key == "ma") { ctx->plan_grp[GRP_MA] = v; continue; }
    if (key == "dec") { ctx->plan_grp[GRP_DEC] = v; continue; }
    if (key == "menc") { ctx->plan_grp[GRP_MENC] = v; continue; }
    const size_t dot = key.find('.');
    if (dot == std::string::npos) return false;
    const std::string sc = key.substr(0, dot), kd = key.substr(dot + 1);
    int s0, s1;
    int b0 = -1, b1 = -1;
    if (sc.size() >= 2 && sc[0] == 'b' && isdigit((unsigned char)sc[1])) {             // b<lo>-<hi> / b<n>: block range
      const size_t dash = sc.find('-');
      b0 = atoi(sc.c_str() + 1);
      b1 = dash == std::string::npos ? b0 : atoi(sc.c_str() + dash + 1);
      if (b0 < 0 || b1 < b0 || b1 >= 64) return false;
      s0 = s1 = 0;
    } else
    if (sc == "s1") s0 = s1 = 1;
    else if (sc == "s2") s0 = s1 = 2;
    else if (sc == "s12") { s0 = 1; s1 = 2; }
    else if (sc == "s3") s0 = s1 = 3;
    else if (sc == "s4") s0 = s1 = 4;
    else return false;
    int k0, k1;
    if (kd == "qkv") k0 = k1 = LIN_QKV;
    else if (kd == "sc") k0 = k1 = LIN_SC;
    else if (kd == "proj") k0 = k1 = LIN_PROJ;
    else if (kd == "fc1") k0 = k1 = LIN_FC1;
    else if (kd == "fc2") k0 = k1 = LIN_FC2;
    else if (kd == "mlp") { k0 = LIN_FC1; k1 = LIN_FC2; }
    else if (kd == "all") { k0 = 0; k1 = 4; }
    else return false;
    if (b0 >= 0) {
      for (int bi = b0; bi <= b1; ++bi)
        for (int k = k0; k <= k1; ++k) ctx->plan_blk[bi][k] = (signed char)v;
      continue;
    }
    for (int st = s0; st <= s1; ++st)
      for (int k = k0; k <= k1; ++k) ctx->plan[st][k] = v;
  }
  return true;
}

extern "C" int sam2mi_create(const sam2mi_config* cfg, sam2mi_ctx** out) {
  if (!cfg || !out) return sam2mi_set_error(nullptr, "sam2mi_create", "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return sam2mi_set_error(nullptr, "sam2mi_create", "no HIP device visible (this backend has no CPU fallback)");
  sam2mi_ctx* ctx = new sam2mi_ctx();
  ctx->cfg = *cfg;
  if (ctx->cfg.max_batch <= 0) ctx->cfg.max_batch = 1;
  if (ctx->cfg.bank_slots <= 0) ctx->cfg.bank_slots = 64;
  if (ctx->cfg.feat_slots <= 0) ctx->cfg.feat_slots = 16;
  if (ctx->cfg.precision != SAM2MI_PRECISION_F16 && ctx->cfg.precision != SAM2MI_PRECISION_F16X3 && ctx->cfg.precision != SAM2MI_PRECISION_F16S) {
    sam2mi_set_error(nullptr, "sam2mi_create", "unknown precision (0: f16, 1: f16x3, 2: f16s)");
    delete ctx;
    return 1;
  }
  ctx->precise = ctx->cfg.precision != SAM2MI_PRECISION_F16;
  ctx->selective = ctx->cfg.precision == SAM2MI_PRECISION_F16S;
  if (ctx->selective && !f16s_plan_init(ctx)) {
    sam2mi_set_error(nullptr, "sam2mi_create", "SAM2MI_F16S_PLAN: expected <s1|s2|s12|s3|s4>.<qkv|sc|proj|fc1|fc2|mlp|all>=<f16|w|full>, <other|neck|ma|dec|menc>=<...>, attn=<0|1>");
    delete ctx;
    return 1;
  }
  // the X-stationary / fused-MLP / accumulator-stationary kernels take plain f16 operands: the f16x3 mode runs every linear
  // on the split-operand instantiation of the tiled kernel (gemm2.hip)
  ctx->use_fused_mlp = (!ctx->precise || ctx->selective) && getenv("SAM2MI_NO_FUSED_MLP") == nullptr;      // f16s: where the plan has both MLP linears in f16
  ctx->use_xs = (!ctx->precise || ctx->selective) && getenv("SAM2MI_NO_XS") == nullptr;      // f16s: the linears planned as plain f16
  ctx->use_rowln = (!ctx->precise || (ctx->selective && ctx->plan_grp[GRP_MA] == PREC_F16)) && getenv("SAM2MI_NO_ROWLN") == nullptr;
  ctx->use_projln = (!ctx->precise || ctx->selective) && getenv("SAM2MI_NO_PROJLN") == nullptr;      // f16s: the split instantiations, stages 1-2
  ctx->use_mem_space_values = getenv("SAM2MI_NO_MEM_SPACE_VALUES") == nullptr;      // A/B switch (needs the fused tail)
  // norm1 inside the operand load of the X-stationary QKV kernel: pays in stage 1 only (C = 144: the QKV launch goes 162 -> 200 us and the
  // 107-us LayerNorm launch disappears; same box, 8-frame pass 28.14 -> 27.89 ms with 144, 27.98 with 288, worse with 576)
  ctx->ln1_fuse_maxc = ctx->precise ? 0 : (getenv("SAM2MI_LN1_FUSE_MAXC") ? atoi(getenv("SAM2MI_LN1_FUSE_MAXC")) : 144);
  // LayerNorm inside the operand load of the X-stationary / fused-MLP kernels: parity-tested, but measured EQUAL end to end
  // (205.6 vs 205.7 frames/s): the row is read twice as f32 by every column split, which costs what the separate LayerNorm
  // kernel cost (it runs at 5.5 TB/s) and moves more bytes past the L2.  Opt-in for A/B runs.
  ctx->ln_fuse = !ctx->precise && getenv("SAM2MI_LN_FUSE") != nullptr;
  if (getenv("SAM2MI_ENC_SUB")) ctx->enc_sub = atoi(getenv("SAM2MI_ENC_SUB"));
  ctx->use_ks = !ctx->precise && getenv("SAM2MI_KS") != nullptr;     // experimental (no end-to-end gain over the tiled kernel on fc2): opt-in
  hipError_t e = gemm_init();
  if (e == hipSuccess) e = flash256_init();
  if (e == hipSuccess) e = mlp_fused_init();
  if (e == hipSuccess) e = gemm_xs_init();
  if (e == hipSuccess) e = gemm_projln_init();
  if (e == hipSuccess) e = gemm_ks_init();
  if (e != hipSuccess) {
    sam2mi_set_error(nullptr, "gemm_init", hipGetErrorString(e));
    delete ctx;
    return 1;
  }
  *out = ctx;
  return 0;
}

extern "C" void sam2mi_destroy(sam2mi_ctx* ctx) {
  if (!ctx) return;
  hipDeviceSynchronize();
  for (void* p : ctx->allocs) hipFree(p);
  for (WsDomain* d : {&ctx->dom_enc, &ctx->dom_track})
    if (d->ev) hipEventDestroy(d->ev);
  for (ProfAcc* a : {&ctx->prof_gemm, &ctx->prof_attn, &ctx->prof_mlp, &ctx->prof_xs, &ctx->prof_ks}) {
    for (auto& pr : a->pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto& pr : a->pending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  }
  delete ctx;
}

extern "C" const char* sam2mi_last_error(sam2mi_ctx* ctx) { (void)ctx; return t_last_error.c_str(); }

extern "C" int sam2mi_load_weight(sam2mi_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim) {
  if (!ctx || !key || !host_data) return sam2mi_set_error(ctx, "sam2mi_load_weight", "null argument");
  if (ctx->finalized) return sam2mi_set_error(ctx, "sam2mi_load_weight", "weights already finalized");
  HostW w;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    w.shape.push_back(shape[i]);
    n *= (size_t)shape[i];
  }
  w.data.assign(host_data, host_data + n);
  ctx->hw[key] = std::move(w);
  return 0;
}

extern "C" int sam2mi_profile_enable(sam2mi_ctx* ctx, int on) {
  std::lock_guard<std::mutex> lk(ctx->prof_mu);
  ctx->prof_on = on != 0;
  if (on) {
    for (ProfAcc* a : {&ctx->prof_gemm, &ctx->prof_attn, &ctx->prof_mlp, &ctx->prof_xs, &ctx->prof_ks}) { a->ms = 0; a->flops = 0; a->launches = 0; }
    ctx->prof_by_kernel.clear();
  }
  return 0;
}

extern "C" int sam2mi_profile_read(sam2mi_ctx* ctx, double* gemm_ms, double* gemm_flops, int64_t* gemm_launches,
                                   double* attn_ms, double* attn_flops, int64_t* attn_launches) {
  std::lock_guard<std::mutex> lk(ctx->prof_mu);
  for (ProfAcc* a : {&ctx->prof_gemm, &ctx->prof_attn, &ctx->prof_mlp, &ctx->prof_xs, &ctx->prof_ks}) {
    prof_drain(*a);
  }
  if (gemm_ms) *gemm_ms = ctx->prof_gemm.ms;
  if (gemm_flops) *gemm_flops = ctx->prof_gemm.flops;
  if (gemm_launches) *gemm_launches = ctx->prof_gemm.launches;
  if (attn_ms) *attn_ms = ctx->prof_attn.ms;
  if (attn_flops) *attn_flops = ctx->prof_attn.flops;
  if (attn_launches) *attn_launches = ctx->prof_attn.launches;
  return 0;
}

static int prof_read_one(ProfAcc& a, double* ms, double* flops, int64_t* launches);
// One line per GEMM-family kernel instantiation: "name\tms\tflops\tlaunches\n" (names as rocprofv3 prints them).
// Call after sam2mi_profile_read (which drains the events).  Returns the number of bytes written, or -1.
extern "C" int sam2mi_profile_read_kernels(sam2mi_ctx* ctx, char* out, int cap) {
  if (!ctx || !out || cap <= 0) return -1;
  std::lock_guard<std::mutex> lk(ctx->prof_mu);
  for (ProfAcc* a : {&ctx->prof_gemm, &ctx->prof_mlp, &ctx->prof_xs, &ctx->prof_ks}) prof_drain(*a);
  std::string sout;
  for (auto& kv : ctx->prof_by_kernel) {
    char line[256];
    snprintf(line, sizeof(line), "%s\t%.9f\t%.17g\t%lld\t%.17g\n", kv.first.c_str(), kv.second.ms, kv.second.flops, (long long)kv.second.launches,
             kv.second.bytes);
    sout += line;
  }
  if ((int)sout.size() + 1 > cap) return -1;
  memcpy(out, sout.c_str(), sout.size() + 1);
  return (int)sout.size();
}
extern "C" int sam2mi_profile_read_xs(sam2mi_ctx* ctx, double* ms, double* flops, int64_t* launches) {
  return ctx ? prof_read_one(ctx->prof_xs, ms, flops, launches) : 1;
}
extern "C" int sam2mi_profile_read_ks(sam2mi_ctx* ctx, double* ms, double* flops, int64_t* launches) {
  return ctx ? prof_read_one(ctx->prof_ks, ms, flops, launches) : 1;
}
extern "C" int sam2mi_profile_read_mlp(sam2mi_ctx* ctx, double* ms, double* flops, int64_t* launches) {
  return ctx ? prof_read_one(ctx->prof_mlp, ms, flops, launches) : 1;
}
static int prof_read_one(ProfAcc& a, double* ms, double* flops, int64_t* launches) {
  prof_drain(a);        // callers: single-threaded read-out after the profiled pass
  if (ms) *ms = a.ms;
  if (flops) *flops = a.flops;
  if (launches) *launches = a.launches;
  return 0;
}

// ------------------------------------------------------------------ finalize: pack everything
static int alloc_workspaces(sam2mi_ctx* ctx);

extern "C" int sam2mi_finalize_weights(sam2mi_ctx* ctx) {
  if (!ctx) return 1;
  if (ctx->finalized) return 0;
  const sam2mi_config& c = ctx->cfg;
  Packer pk{ctx};
  const int E = c.embed_dim;
  const int G = c.image_size / 4;
  ctx->head_dim = c.num_heads > 0 ? E / c.num_heads : 0;
  const bool large_family = c.window_spec[0] == 8 && c.window_spec[1] == 4 && c.window_spec[2] == 16 && c.window_spec[3] == 8 && ctx->head_dim == 72;
  ctx->generic = !large_family;
  if (ctx->generic) {
    if (c.num_heads <= 0 || E % c.num_heads || (ctx->head_dim != 56 && ctx->head_dim != 72 && ctx->head_dim != 96) || E % 16)
      return sam2mi_set_error(ctx, "sam2mi_finalize_weights", "head_dim must be 56, 72 or 96 (hiera tiny / small / base+ / large)");
    for (int i = 0; i < 4; ++i)
      if (c.window_spec[i] < 2 || c.window_spec[i] > 16) return sam2mi_set_error(ctx, "sam2mi_finalize_weights", "window sizes 2..16 are supported");
    if (ctx->precise) return sam2mi_set_error(ctx, "sam2mi_finalize_weights", "the split-operand precision modes (f16x3, f16s) are implemented for hiera-large only");
  }

  // ---- Hiera block table (hieradet.py:243-268)
  {
    int depth = 0, stage_ends[4], acc = 0;
    for (int i = 0; i < 4; ++i) { acc += c.stages[i]; stage_ends[i] = acc - 1; }
    depth = acc;
    int dim = E, heads = c.num_heads, cur_stage = 1;
    auto is_end = [&](int i) { for (int k = 0; k < 4; ++k) if (stage_ends[k] == i) return true; return false; };
    for (int i = 0; i < depth; ++i) {
      HieraBlockW b;
      b.idx = i;
      b.dim = dim;
      int dim_out = dim;
      int window = c.window_spec[cur_stage - 1];
      for (int k = 0; k < 8; ++k) if (c.global_att_blocks[k] == i) window = 0;
      b.q_pool = false;
      if (i > 0 && is_end(i - 1)) {
        dim_out = dim * 2;
        heads *= 2;
        cur_stage += 1;
        b.q_pool = true;
      }
      b.dim_out = dim_out;
      b.heads = heads;
      b.window = window;
      b.stage_end = is_end(i);
      const std::string p = "image_encoder.trunk.blocks." + std::to_string(i) + ".";
      b.n1 = pk.norm(p + "norm1");
      b.n2 = pk.norm(p + "norm2");
      b.qkv = pk.lin16(p + "attn.qkv");
      b.proj = pk.lin16(p + "attn.proj");
      b.fc1 = pk.lin16(p + "mlp.layers.0");
      b.fc2 = pk.lin16(p + "mlp.layers.1");
      for (Lin16* L : {&b.qkv, &b.fc1}) {       // stages 1-3: QKV and fc1 also in the X-stationary kernel's piece order (the projection, N = K, is not faster there)
        if (!pk.ok || (ctx->precise && !ctx->selective) || !L->w || !gemm_xs_supported(L->N, L->K) || (L == &b.fc1 && mlp_fused_supported(b.dim_out) && !ctx->precise)) continue;
        L->xs_pack = (half_t*)dalloc(ctx, gemm_xs_pack_bytes(L->N, L->K));
        if (!L->xs_pack || gemm_xs_pack(L->w, L->N, L->K, L->K, L->xs_pack, nullptr) != hipSuccess) pk.ok = false;
      }
      for (int kind : {LIN_QKV, LIN_PROJ}) {       // f16s: the linears planned as weight split also as [W_hi | W_lo] image of the X-stationary kernel
        Lin16* L = kind == LIN_QKV ? &b.qkv : &b.proj;
        if (!pk.ok || !ctx->selective || !L->w || !L->lo_off || !gemm_xs_supported(L->N, L->K) ||
            ctx->plan[dim_out >= 1152 ? 4 : dim_out >= 576 ? 3 : dim_out >= 288 ? 2 : 1][kind] != PREC_WSPLIT) continue;
        const size_t scratch_b = (size_t)2 * ((L->N + 31) / 32 * 32) * L->K * sizeof(half_t);
        half_t* scratch = (half_t*)dalloc_raw(ctx, scratch_b);
        L->xs_wpack = (half_t*)dalloc(ctx, gemm_xs_wsplit_pack_bytes(L->N, L->K));
        if (!scratch || !L->xs_wpack || gemm_xs_wsplit_pack(L->w, L->w + L->lo_off, L->N, L->K, L->xs_wpack, scratch, nullptr) != hipSuccess) pk.ok = false;
        if (scratch) { hipStreamSynchronize(nullptr); dfree(ctx, scratch); }
      }
      // stages 1-2 only by default: at C = 576 a 32-row workgroup streams the whole 663-KB weight from L2 with 9 KB per wave in flight
      // and takes 89 us where GEMM + LayerNorm take 55 + 21 (C = 144: 184 vs 198 + 107 us, C = 288: 94 vs 106 + 41 us)
      static const int projln_max_c = getenv("SAM2MI_PROJLN_MAXC") ? atoi(getenv("SAM2MI_PROJLN_MAXC")) : 288;
      const int pst = dim_out >= 1152 ? 4 : dim_out >= 576 ? 3 : dim_out >= 288 ? 2 : 1;
      const bool projln_split = ctx->selective && dim_out <= 288 && b.proj.lo_off && ctx->plan[pst][LIN_PROJ] >= PREC_WSPLIT;      // f16s: stages 1-2
      if (pk.ok && (!ctx->precise || projln_split) && b.proj.w && b.proj.N == dim_out && b.proj.K == dim_out && gemm_projln_supported(dim_out) && dim_out <= projln_max_c) {
        b.proj_pack = (half_t*)dalloc(ctx, gemm_xs_pack_bytes(dim_out, dim_out));      // out-projection + residual + norm2 in one kernel
        if (!b.proj_pack || gemm_xs_pack(b.proj.w, dim_out, dim_out, dim_out, b.proj_pack, nullptr) != hipSuccess) pk.ok = false;
        if (projln_split) {
          b.proj_pack_lo = (half_t*)dalloc(ctx, gemm_xs_pack_bytes(dim_out, dim_out));
          if (!b.proj_pack_lo || gemm_xs_pack(b.proj.w + b.proj.lo_off, dim_out, dim_out, dim_out, b.proj_pack_lo, nullptr) != hipSuccess) pk.ok = false;
        }
      }
      for (Lin16* L : {&b.fc2}) {       // stage 3 (N = 576, K = 2304): fc2 in the accumulator-stationary kernel's order (the projection, K = 576, is faster tiled)
        if (!pk.ok || ctx->precise || !L->w || !gemm_ks_supported(L->N, L->K)) continue;
        L->ks_pack = (half_t*)dalloc(ctx, gemm_ks_pack_bytes(L->N, L->K));
        if (!L->ks_pack || gemm_ks_pack(L->w, L->N, L->K, L->K, L->ks_pack, nullptr) != hipSuccess) pk.ok = false;
      }
      // LN folded into the consumers that load their operand row-wise (X-stationary QKV / fc1, fused MLP):
      //   W (g * xhat + b_LN) + b = (W diag g) xhat + (W b_LN + b)       (hieradet.py:137,:163: norm1 -> attn.qkv, norm2 -> mlp)
      auto fold_ln = [&](const std::string& lin, const std::string& nrm, Lin16& folded) {
        const HostW* w = pk.get(lin + ".weight"); const HostW* bb = pk.get(lin + ".bias");
        const HostW* g = pk.get(nrm + ".weight"); const HostW* be = pk.get(nrm + ".bias");
        if (!w || !bb || !g || !be) return;
        const int N = (int)w->shape[0], K = (int)(w->data.size() / N);
        std::vector<float> W((size_t)N * K), B(N);
        for (int n = 0; n < N; ++n) {
          double acc = bb->data[n];
          for (int k = 0; k < K; ++k) {
            W[(size_t)n * K + k] = w->data[(size_t)n * K + k] * g->data[k];
            acc += (double)w->data[(size_t)n * K + k] * be->data[k];
          }
          B[n] = (float)acc;
        }
        folded = pk.lin16_raw(W, B, N, K);
      };
      Lin16 fc1_folded;
      if (pk.ok && (ctx->ln_fuse || dim <= ctx->ln1_fuse_maxc)) {
        if (b.qkv.xs_pack && dim == dim_out) {                 // not on dim-change blocks: their LN1 output also feeds the shortcut projection
          Lin16 f;
          fold_ln(p + "attn.qkv", p + "norm1", f);
          b.qkv.xs_ln_pack = (half_t*)dalloc(ctx, gemm_xs_pack_bytes(f.N, f.K));
          if (!f.w || !b.qkv.xs_ln_pack || gemm_xs_pack(f.w, f.N, f.K, f.K, b.qkv.xs_ln_pack, nullptr) != hipSuccess) pk.ok = false;
          b.qkv.b_ln = f.b;
        }
        if (b.fc1.xs_pack || mlp_fused_supported(b.dim_out)) fold_ln(p + "mlp.layers.0", p + "norm2", fc1_folded);
        if (b.fc1.xs_pack && fc1_folded.w) {
          b.fc1.xs_ln_pack = (half_t*)dalloc(ctx, gemm_xs_pack_bytes(fc1_folded.N, fc1_folded.K));
          if (!b.fc1.xs_ln_pack || gemm_xs_pack(fc1_folded.w, fc1_folded.N, fc1_folded.K, fc1_folded.K, b.fc1.xs_ln_pack, nullptr) != hipSuccess) pk.ok = false;
          b.fc1.b_ln = fc1_folded.b;
        }
      }
      if (pk.ok && ctx->ln_fuse && mlp_fused_supported(b.dim_out) && fc1_folded.w) {
        b.mlp_ln_pack = (half_t*)dalloc(ctx, mlp_fused_pack_bytes(b.dim_out));
        if (!b.mlp_ln_pack || mlp_fused_pack(fc1_folded.w, b.fc2.w, b.dim_out, b.mlp_ln_pack, nullptr) != hipSuccess) pk.ok = false;
        b.fc1.b_ln = fc1_folded.b;
      }
      const int pstage = dim_out >= 1152 ? 4 : dim_out >= 576 ? 3 : dim_out >= 288 ? 2 : 1;
      const bool mlp_f16 = !ctx->precise || (ctx->selective && ctx->plan[pstage][LIN_FC1] == PREC_F16 && ctx->plan[pstage][LIN_FC2] == PREC_F16);
      if (pk.ok && mlp_f16 && mlp_fused_supported(b.dim_out)) {     // stages 1-2: weights also in the fused MLP kernel's piece order
        b.mlp_pack = (half_t*)dalloc(ctx, mlp_fused_pack_bytes(b.dim_out));
        if (!b.mlp_pack || mlp_fused_pack(b.fc1.w, b.fc2.w, b.dim_out, b.mlp_pack, nullptr) != hipSuccess) pk.ok = false;
      }
      if (dim != dim_out) b.sc = pk.lin16(p + "proj");
      // shapes derived from the config: a checkpoint of another model size must fail here, not read out of bounds later
      auto shape_ok = [&](const Lin16& L, int N, int K) { return !L.w || (L.N == N && L.K == K); };
      if (!shape_ok(b.qkv, 3 * dim_out, dim) || !shape_ok(b.proj, dim_out, dim_out) || !shape_ok(b.fc1, 4 * dim_out, dim_out) ||
          !shape_ok(b.fc2, dim_out, 4 * dim_out) || !shape_ok(b.sc, dim_out, dim) || (b.n1.w && b.n1.C != dim) || (b.n2.w && b.n2.C != dim_out))
        return sam2mi_set_error(ctx, "sam2mi_finalize_weights: tensor shapes do not match the configured architecture at", p.c_str());
      {
        std::vector<float> qs((size_t)3 * dim_out, 1.0f);
        for (int k = 0; k < dim_out; ++k) qs[k] = 1.4426950408889634f / std::sqrt((float)ctx->head_dim);
        b.qscale = dupload(ctx, qs);
      }
      ctx->blocks.push_back(b);
      dim = dim_out;
    }
  }
  // ---- patch embed (K 147 -> 160) and the position table in window-major order
  if (const HostW* w = pk.get("image_encoder.trunk.patch_embed.proj.weight")) {
    const HostW* b = pk.get("image_encoder.trunk.patch_embed.proj.bias");
    std::vector<float> W((size_t)E * 160, 0.f);
    for (int o = 0; o < E; ++o)
      for (int k = 0; k < 147; ++k) W[(size_t)o * 160 + k] = w->data[(size_t)o * 147 + k];
    if (b) ctx->patch = pk.lin16_raw(W, b->data, E, 160);
  }
  {
    const HostW* pe = pk.get("image_encoder.trunk.pos_embed");
    const HostW* pw = pk.get("image_encoder.trunk.pos_embed_window");
    if (pe && pw) {
      const int ph = (int)pe->shape[2], pwid = (int)pe->shape[3], ws = (int)pw->shape[2];
      std::vector<float> big((size_t)E * G * G);
      bicubic_resize(pe->data.data(), E, ph, pwid, big.data(), G, G);
      std::vector<float> tab((size_t)G * G * E);
      for (int y = 0; y < G; ++y)
        for (int x = 0; x < G; ++x) {
          const int t = ctx->generic ? y * G + x : tok_of_yx(y, x, G, 8);       // generic sizes: plain row-major tokens
          for (int ch = 0; ch < E; ++ch)
            tab[(size_t)t * E + ch] = big[((size_t)ch * G + y) * G + x] + pw->data[((size_t)ch * ws + (y % ws)) * ws + (x % ws)];
        }
      ctx->pos_tab = dupload(ctx, tab);
    }
  }
  // ---- neck (image_encoder.py:113-114: convs[n - i] serves level i)
  for (int lvl = 0; lvl < 4; ++lvl) ctx->neck[lvl] = pk.lin16("image_encoder.neck.convs." + std::to_string(3 - lvl) + ".conv");
  ctx->conv_s0 = pk.lin16("sam_mask_decoder.conv_s0");
  ctx->conv_s1 = pk.lin16("sam_mask_decoder.conv_s1");
  // FPN levels 0 and 1 receive no top-down term (fpn_top_down_levels [2, 3], image_encoder.py:115-125) and only feed
  // conv_s0 / conv_s1 (sam2_base_official.py:560-565): lateral 1x1 conv and conv_s* are composed in f32 into one linear
  // map, W = W_s W_lat, b = W_s b_lat + b_s, so the 256-channel 256^2 / 128^2 laterals (537 MB f32 per 8 frames) never exist.
  {
    auto compose = [&](const std::string& lat, const std::string& cs, Lin16& out) {
      const HostW* wl = pk.get(lat + ".weight"); const HostW* bl = pk.get(lat + ".bias");
      const HostW* ws = pk.get(cs + ".weight");  const HostW* bs = pk.get(cs + ".bias");
      if (!wl || !bl || !ws || !bs) return;
      const int mid = (int)wl->shape[0], K = (int)(wl->data.size() / mid), N = (int)ws->shape[0];
      if ((int)(ws->data.size() / N) != mid) { pk.ok = false; pk.missing += cs + ".weight(shape) "; return; }
      std::vector<float> W((size_t)N * K), B(N);
      for (int n = 0; n < N; ++n) {
        double bacc = bs->data[n];
        for (int m = 0; m < mid; ++m) bacc += (double)ws->data[(size_t)n * mid + m] * bl->data[m];
        B[n] = (float)bacc;
        for (int k = 0; k < K; ++k) {
          double acc = 0;
          for (int m = 0; m < mid; ++m) acc += (double)ws->data[(size_t)n * mid + m] * wl->data[(size_t)m * K + k];
          W[(size_t)n * K + k] = (float)acc;
        }
      }
      out = pk.lin16_raw(W, B, N, K);
    };
    compose("image_encoder.neck.convs.3.conv", "sam_mask_decoder.conv_s0", ctx->neck_s0);
    compose("image_encoder.neck.convs.2.conv", "sam_mask_decoder.conv_s1", ctx->neck_s1);
  }
  for (int i = 0; i < 3; ++i) {
    const int S = G >> i;
    std::vector<float> pe = sine_pe_nchw(S, S, 256);
    ctx->sine_pe[i] = dupload(ctx, pe);
    if (i == 2) {
      std::vector<float> tok((size_t)S * S * 256);
      for (int ch = 0; ch < 256; ++ch)
        for (int t = 0; t < S * S; ++t) tok[(size_t)t * 256 + ch] = pe[(size_t)ch * S * S + t];
      ctx->sine_pe_tok64 = dupload(ctx, tok);
    }
  }
  ctx->no_mem_embed = pk.f32("no_mem_embed");

  // ---- memory attention
  {
    std::vector<std::string> ks, vs;
    for (int l = 0; l < 4; ++l) {
      const std::string p = "memory_attention.layers." + std::to_string(l) + ".";
      MemAttnLayerW m;
      m.n1 = pk.norm(p + "norm1");
      m.n2 = pk.norm(p + "norm2");
      m.n3 = pk.norm(p + "norm3");
      m.self_qkv = pk.concat16({p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"});
      m.self_out = pk.lin16(p + "self_attn.out_proj");
      m.cross_q = pk.lin16(p + "cross_attn_image.q_proj");
      m.cross_out = pk.lin16(p + "cross_attn_image.out_proj");
      m.lin1 = pk.lin16(p + "linear1");
      m.lin2 = pk.lin16(p + "linear2");
      {                          // Wo Wv [256, 64] and Wo bv + bo: softmax(q k^T) (m Wv^T + bv) Wo^T + bo = (softmax(q k^T) m) (Wo Wv)^T + (Wo bv + bo)
        const HostW* wo = pk.get(p + "cross_attn_image.out_proj.weight"); const HostW* bo = pk.get(p + "cross_attn_image.out_proj.bias");
        const HostW* wv = pk.get(p + "cross_attn_image.v_proj.weight"); const HostW* bv = pk.get(p + "cross_attn_image.v_proj.bias");
        if (wo && bo && wv && bv && wo->data.size() == (size_t)256 * 256 && wv->data.size() == (size_t)256 * 64) {
          std::vector<float> W((size_t)256 * 64), B(256);
          for (int n = 0; n < 256; ++n) {
            double bb = bo->data[n];
            for (int j = 0; j < 256; ++j) bb += (double)wo->data[(size_t)n * 256 + j] * bv->data[j];
            B[n] = (float)bb;
            for (int k = 0; k < 64; ++k) {
              double a = 0.0;
              for (int j = 0; j < 256; ++j) a += (double)wo->data[(size_t)n * 256 + j] * wv->data[(size_t)j * 64 + k];
              W[(size_t)n * 64 + k] = (float)a;
            }
          }
          m.cross_vo = pk.lin16_raw(W, B, 256, 64);
        }
      }
      ctx->mal.push_back(m);
      ks.push_back(p + "cross_attn_image.k_proj");
      vs.push_back(p + "cross_attn_image.v_proj");
    }
    ctx->cross_k_all = pk.concat16(ks);
    ctx->cross_v_all = pk.concat16(vs);
    ctx->ma_norm = pk.norm("memory_attention.norm");
    // axial RoPE table (position_encoding_fix.py:172-205): pair p<64 -> x * theta^(-4p/256), else y * ...
    std::vector<float> rc((size_t)4096 * 128), rs((size_t)4096 * 128);
    for (int t = 0; t < 4096; ++t) {
      const float tx = (float)(t % 64), ty = (float)(t / 64);
      for (int p = 0; p < 128; ++p) {
        const int fi = 2 * (p % 64);                       // index into freqs (step 2 of the [..., ::2] slice)
        const float freq = 1.0f / std::pow(10000.f, (float)(2 * fi) / 256.f);
        const float ang = (p < 64 ? tx : ty) * freq;
        rc[(size_t)t * 128 + p] = std::cos(ang);
        rs[(size_t)t * 128 + p] = std::sin(ang);
      }
    }
    ctx->rope_cos = dupload(ctx, rc);
    ctx->rope_sin = dupload(ctx, rs);
    std::vector<float> qs(768, 1.0f), qc(256, 1.4426950408889634f / 16.0f);
    for (int k = 0; k < 256; ++k) qs[k] = 1.4426950408889634f / 16.0f;
    ctx->qs_self = dupload(ctx, qs);
    ctx->qs_cross = dupload(ctx, qc);
  }

  // ---- SAM heads
  {
    const std::string d = "sam_mask_decoder.";
    for (int l = 0; l < 2; ++l) {
      const std::string p = d + "transformer.layers." + std::to_string(l) + ".";
      DecLayerW L;
      L.self_attn = {pk.lin32(p + "self_attn.q_proj"), pk.lin32(p + "self_attn.k_proj"), pk.lin32(p + "self_attn.v_proj"),
                     pk.lin32(p + "self_attn.out_proj")};
      L.t2i_q = pk.lin32(p + "cross_attn_token_to_image.q_proj");
      L.t2i_k = pk.lin16(p + "cross_attn_token_to_image.k_proj");
      L.t2i_v = pk.lin16(p + "cross_attn_token_to_image.v_proj");
      L.t2i_o = pk.lin32(p + "cross_attn_token_to_image.out_proj");
      L.i2t_q = pk.lin16(p + "cross_attn_image_to_token.q_proj");
      L.i2t_k = pk.lin32(p + "cross_attn_image_to_token.k_proj");
      L.i2t_v = pk.lin32(p + "cross_attn_image_to_token.v_proj");
      L.i2t_o = pk.lin16(p + "cross_attn_image_to_token.out_proj");
      L.mlp1 = pk.lin32(p + "mlp.layers.0");
      L.mlp2 = pk.lin32(p + "mlp.layers.1");
      L.n1 = pk.norm(p + "norm1");
      L.n2 = pk.norm(p + "norm2");
      L.n3 = pk.norm(p + "norm3");
      L.n4 = pk.norm(p + "norm4");
      ctx->dec.push_back(L);
    }
    const std::string f = d + "transformer.final_attn_token_to_image.";
    ctx->fin_q = pk.lin32(f + "q_proj");
    ctx->fin_k = pk.lin16(f + "k_proj");
    ctx->fin_v = pk.lin16(f + "v_proj");
    ctx->fin_o = pk.lin32(f + "out_proj");
    ctx->fin_norm = pk.norm(d + "transformer.norm_final_attn");
    {
      const HostW* a = pk.get(d + "obj_score_token.weight");
      const HostW* b = pk.get(d + "iou_token.weight");
      const HostW* m = pk.get(d + "mask_tokens.weight");
      if (a && b && m) {
        std::vector<float> t;
        t.insert(t.end(), a->data.begin(), a->data.end());
        t.insert(t.end(), b->data.begin(), b->data.end());
        t.insert(t.end(), m->data.begin(), m->data.end());
        ctx->out_tokens = dupload(ctx, t);
      }
    }
    // ConvTranspose2d(k=2, s=2) as a GEMM: Wg[(dy*2+dx)*O + o][i] = W[i][o][dy][dx]
    auto convT = [&](const std::string& key, Lin16& out, float*& bias) {
      const HostW* w = pk.get(key + ".weight");
      const HostW* b = pk.get(key + ".bias");
      if (!w || !b) return;
      const int I = (int)w->shape[0], O = (int)w->shape[1];
      std::vector<float> Wg((size_t)4 * O * I), zero((size_t)4 * O, 0.f);
      for (int i = 0; i < I; ++i)
        for (int o = 0; o < O; ++o)
          for (int pos = 0; pos < 4; ++pos) Wg[((size_t)pos * O + o) * I + i] = w->data[((size_t)i * O + o) * 4 + pos];
      out = pk.lin16_raw(Wg, zero, 4 * O, I);
      bias = dupload(ctx, b->data);
    };
    convT(d + "output_upscaling.0", ctx->dc1, ctx->dc1_b);
    convT(d + "output_upscaling.3", ctx->dc2, ctx->dc2_b);
    ctx->up_ln = pk.norm(d + "output_upscaling.1");
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j)
        ctx->hyper[i][j] = pk.lin32(d + "output_hypernetworks_mlps." + std::to_string(i) + ".layers." + std::to_string(j));
    for (int j = 0; j < 3; ++j) {
      ctx->iou_head[j] = pk.lin32(d + "iou_prediction_head.layers." + std::to_string(j));
      ctx->obj_head[j] = pk.lin32(d + "pred_obj_score_head.layers." + std::to_string(j));
      ctx->ptr_proj[j] = pk.lin32("obj_ptr_proj.layers." + std::to_string(j));
    }
    ctx->tpos_proj = pk.lin32("obj_ptr_tpos_proj");
    ctx->no_obj_ptr = pk.f32("no_obj_ptr");
    const std::string pe = "sam_prompt_encoder.";
    ctx->gauss = pk.f32(pe + "pe_layer.positional_encoding_gaussian_matrix");
    {
      std::vector<float> t;
      bool all = true;
      for (int i = 0; i < 4; ++i) {
        const HostW* w = pk.get(pe + "point_embeddings." + std::to_string(i) + ".weight");
        if (!w) { all = false; break; }
        t.insert(t.end(), w->data.begin(), w->data.end());
      }
      if (all) ctx->point_emb4 = dupload(ctx, t);
    }
    ctx->not_a_point = pk.f32(pe + "not_a_point_embed.weight");
    {
      // decoder tokens of a tracked frame without prompts, one copy per object of a batched pass: the six output tokens + two
      // not-a-point rows (the (0,0)/-1 dummy point of _forward_sam_heads and the prompt encoder's padding point)
      const HostW* a = pk.get(d + "obj_score_token.weight");
      const HostW* b = pk.get(d + "iou_token.weight");
      const HostW* m = pk.get(d + "mask_tokens.weight");
      const HostW* nap = pk.get(pe + "not_a_point_embed.weight");
      if (a && b && m && nap) {
        std::vector<float> one;
        one.insert(one.end(), a->data.begin(), a->data.end());
        one.insert(one.end(), b->data.begin(), b->data.end());
        one.insert(one.end(), m->data.begin(), m->data.end());
        one.insert(one.end(), nap->data.begin(), nap->data.end());
        one.insert(one.end(), nap->data.begin(), nap->data.end());
        std::vector<float> t;
        for (int n = 0; n < TRACK_MAX_N; ++n) t.insert(t.end(), one.begin(), one.end());
        ctx->track_tokens = dupload(ctx, t);
      }
    }
    ctx->no_mask_embed = pk.f32(pe + "no_mask_embed.weight");
    ctx->mask_embed = MaskEmbedW{pk.f32(pe + "mask_downscaling.0.weight"), pk.f32(pe + "mask_downscaling.0.bias"),
                                 pk.f32(pe + "mask_downscaling.1.weight"), pk.f32(pe + "mask_downscaling.1.bias"),
                                 pk.f32(pe + "mask_downscaling.3.weight"), pk.f32(pe + "mask_downscaling.3.bias"),
                                 pk.f32(pe + "mask_downscaling.4.weight"), pk.f32(pe + "mask_downscaling.4.bias"),
                                 pk.f32(pe + "mask_downscaling.6.weight"), pk.f32(pe + "mask_downscaling.6.bias")};
    ctx->mds_w = pk.f32("mask_downsample.weight");
    ctx->mds_b = pk.f32("mask_downsample.bias");
  }

  // ---- memory encoder
  {
    const std::string m = "memory_encoder.";
    for (int i = 0; i < 3; ++i) {
      ctx->md_w[i] = pk.f32(m + "mask_downsampler.encoder." + std::to_string(3 * i) + ".weight");
      ctx->md_b[i] = pk.f32(m + "mask_downsampler.encoder." + std::to_string(3 * i) + ".bias");
    }
    for (int i = 0; i < 4; ++i) ctx->md_ln[i] = pk.norm(m + "mask_downsampler.encoder." + std::to_string(3 * i + 1));
    if (const HostW* w = pk.get(m + "mask_downsampler.encoder.6.weight")) {
      const HostW* b = pk.get(m + "mask_downsampler.encoder.6.bias");
      const int O = (int)w->shape[0], I = (int)w->shape[1];
      std::vector<float> Wg((size_t)O * 9 * I);
      for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
          for (int t = 0; t < 9; ++t) Wg[((size_t)o * 9 + t) * I + i] = w->data[((size_t)o * I + i) * 9 + t];
      if (b) ctx->md_conv3 = pk.lin16_raw(Wg, b->data, O, 9 * I);
    }
    if (const HostW* w = pk.get(m + "mask_downsampler.encoder.9.weight")) {
      const HostW* b = pk.get(m + "mask_downsampler.encoder.9.bias");
      const int O = (int)w->shape[0], I = (int)w->shape[1];
      std::vector<float> Wg((size_t)O * 9 * I);
      for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
          for (int t = 0; t < 9; ++t) Wg[((size_t)o * 9 + t) * I + i] = w->data[((size_t)o * I + i) * 9 + t];
      if (b) ctx->md_conv4 = pk.lin16_raw(Wg, b->data, O, 9 * I);
    }
    ctx->md_proj = pk.lin16(m + "mask_downsampler.encoder.12");
    ctx->pix_proj = pk.lin16(m + "pix_feat_proj");
    ctx->me_out = pk.lin16(m + "out_proj");
    for (int l = 0; l < 2; ++l) {
      const std::string p = m + "fuser.layers." + std::to_string(l) + ".";
      ctx->cx[l].dw_w = pk.f32(p + "dwconv.weight");
      ctx->cx[l].dw_b = pk.f32(p + "dwconv.bias");
      ctx->cx[l].ln = pk.norm(p + "norm");
      ctx->cx[l].pw1 = pk.lin16(p + "pwconv1");
      ctx->cx[l].pw2 = pk.lin16(p + "pwconv2");
      ctx->cx[l].gamma = pk.f32(p + "gamma");
    }
    std::vector<float> pe = sine_pe_nchw(64, 64, 64);
    ctx->mem_pos_nchw = dupload(ctx, pe);
    std::vector<float> tok((size_t)4096 * 64);
    for (int ch = 0; ch < 64; ++ch)
      for (int t = 0; t < 4096; ++t) tok[(size_t)t * 64 + ch] = pe[(size_t)ch * 4096 + t];
    ctx->mem_pos = dupload(ctx, tok);
    ctx->tpos_enc = pk.f32("maskmem_tpos_enc");
    ctx->no_obj_embed_spatial = pk.f32("no_obj_embed_spatial");
  }

  if (!pk.ok) return sam2mi_set_error(ctx, "sam2mi_finalize_weights: missing state_dict keys", pk.missing.c_str());
  if (hipDeviceSynchronize() != hipSuccess) return sam2mi_set_error(ctx, "sam2mi_finalize_weights", "device error while uploading");
  CHKI(alloc_workspaces(ctx));
  // dense positional encoding of the prompt encoder (constant)
  CHK(dense_pe_launch(ctx->gauss, 64, ctx->dense_pe, 0));
  CHK(hipDeviceSynchronize());
  ctx->hw.clear();
  ctx->finalized = true;
  return 0;
}

static int alloc_workspaces(sam2mi_ctx* ctx) {
  const sam2mi_config& c = ctx->cfg;
  const int G = c.image_size / 4;
  const int E = c.embed_dim;
  const size_t T0 = (size_t)c.max_batch * G * G;      // stage-1 tokens
  ctx->ws_tokens = T0;
  // per-token element counts are maximal in stage 1 (tokens / 4 and channels * 2 per stage)
#define ALLOC(ptr, type, count)                                                     \
  do {                                                                              \
    ptr = (type*)dalloc(ctx, (size_t)(count) * sizeof(type));                       \
    if (!ptr) return sam2mi_set_error(ctx, "hipMalloc", #ptr);                      \
  } while (0)
  // every f16 activation buffer comes out of ONE arena (256-B aligned pieces); the f16x3 mode doubles it and keeps the lo
  // plane of each buffer `ctx->lo16` elements behind its hi plane (one offset for all of them)
  std::vector<std::pair<half_t**, size_t>> arena16;
  size_t arena_elems = 0;
#define ARENA16(ptr, count)                                                         \
  do {                                                                              \
    arena16.push_back({&(ptr), arena_elems});                                       \
    arena_elems += ((size_t)(count) + 127) / 128 * 128;                             \
  } while (0)
  ALLOC(ctx->ws_x, float, T0 * 2 * E);               // same capacity as ws_x2: the two are swapped after a window re-ordering
  ALLOC(ctx->ws_x2, float, T0 * 2 * E);              // shortcut projection output (unpooled, 2C)
  ARENA16(ctx->ws_a16, T0 * 160);              // LN output (<= 144 ch) or im2col patches (160)
  ARENA16(ctx->ws_qk16, T0 * 4 * E);           // [M, 2*Cout], Cout up to 2E at stage-1 tokens (block 2)
  ARENA16(ctx->ws_vT16, T0 * 2 * E);
  ARENA16(ctx->ws_att16, T0 * E);
  ARENA16(ctx->ws_h16, T0 * 4 * E);
  ARENA16(ctx->ws_qp16, T0 / 4 * 2 * E);
  if (ctx->generic) {
    ARENA16(ctx->ws_w16, T0 * 2 * E);
    ARENA16(ctx->ws_o16, T0 * 2 * E);
  }
  for (int l = 0; l < 4; ++l) ALLOC(ctx->ws_lat[l], float, (T0 >> (2 * l)) * (l == 0 ? 32 : l == 1 ? 64 : 256));   // levels 0 / 1: conv_s0 / conv_s1 outputs

  // tracking (B = 1)
  const int NKCAP = 7 * 4096 + 64 * 4 + 4096;         // generous: many conditioning frames are rejected above this
  ctx->t_nk_cap = NKCAP;
  ALLOC(ctx->t_x, float, (size_t)TRACK_MAX_N * 4096 * 256);
  ARENA16(ctx->t_h16, (size_t)TRACK_MAX_N * 4096 * 256);
  ARENA16(ctx->t_qk16, (size_t)TRACK_MAX_N * 4096 * 512);
  ARENA16(ctx->t_vT16, (size_t)TRACK_MAX_N * 256 * 4096);
  ARENA16(ctx->t_o16, (size_t)TRACK_MAX_N * 4096 * 256);
  ARENA16(ctx->t_q16, (size_t)TRACK_MAX_N * 4096 * 256);
  ARENA16(ctx->t_ff16, (size_t)TRACK_MAX_N * 4096 * 2048);
  ARENA16(ctx->t_kin16, (size_t)TRACK_MAX_N * NKCAP * 64);
  ARENA16(ctx->t_vin16, (size_t)TRACK_MAX_N * NKCAP * 64);
  ARENA16(ctx->t_kall16, (size_t)TRACK_MAX_N * NKCAP * 1024);
  ARENA16(ctx->t_vTall16, (size_t)TRACK_MAX_N * 1024 * NKCAP);
  ARENA16(ctx->t_vinT16, (size_t)TRACK_MAX_N * 64 * NKCAP);
  ALLOC(ctx->t_opart, float, (size_t)16 * 4096 * 256);
  ALLOC(ctx->d_fill_tmp, float, (size_t)65536);
  ALLOC(ctx->d_mask256, float, (size_t)65536);
  ALLOC(ctx->d_dense, float, (size_t)DEC_MAX_N * 4096 * 256);      // dense prompt embeddings of mask prompts, one per prompt of a decoder batch
  ALLOC(ctx->d_pm10, float, 2);
  ALLOC(ctx->d_flag, int, 2);
  ALLOC(ctx->t_ml, float, (size_t)16 * 4096 * 2);
  ALLOC(ctx->t_ptr_tok, float, (size_t)TRACK_MAX_N * 128 * 64);
  ALLOC(ctx->t_ptr_pos, float, (size_t)TRACK_MAX_N * 128 * 64);
  ALLOC(ctx->t_pix, float, (size_t)TRACK_MAX_N * 4096 * 256);
  // decoder
  ALLOC(ctx->d_keys, float, (size_t)DEC_MAX_N * 4096 * 256);
  ARENA16(ctx->d_keys16, (size_t)DEC_MAX_N * 4096 * 256);
  ARENA16(ctx->d_kpe16, (size_t)DEC_MAX_N * 4096 * 256);
  ALLOC(ctx->d_tok, float, (size_t)DEC_MAX_N * 64 * 256);
  ALLOC(ctx->d_tokpe, float, (size_t)DEC_MAX_N * 64 * 256);
  ALLOC(ctx->d_t1, float, (size_t)DEC_MAX_N * 64 * 2048);
  ALLOC(ctx->d_t2, float, (size_t)DEC_MAX_N * 64 * 2048);
  ctx->d_t2i_part_floats = (size_t)DEC_MAX_N * 8 * 8 * 8 * 18;               // prompts x heads x 8 key splits x T <= 8 x (o[16], m, l)
  ALLOC(ctx->d_t2i_part, float, ctx->d_t2i_part_floats);
  ALLOC(ctx->d_t3, float, (size_t)DEC_MAX_N * 64 * 2048);
  ALLOC(ctx->d_t4, float, (size_t)DEC_MAX_N * 64 * 2048);
  ALLOC(ctx->d_big1, float, (size_t)DEC_MAX_N * 4096 * 256);
  ALLOC(ctx->d_big2, float, (size_t)DEC_MAX_N * 4096 * 256);
  ALLOC(ctx->d_big3, float, (size_t)DEC_MAX_N * 4096 * 256);
  ARENA16(ctx->d_big16, (size_t)DEC_MAX_N * 4096 * 256);
  ALLOC(ctx->d_tokens_in, float, (size_t)DEC_MAX_N * 64 * 256);
  ALLOC(ctx->d_sparse, float, (size_t)DEC_MAX_N * 64 * 256);
  ARENA16(ctx->d_up1_16, (size_t)DEC_MAX_N * 16384 * 64);
  ARENA16(ctx->d_up2_16, (size_t)DEC_MAX_N * 65536 * 32);
  ALLOC(ctx->d_g, float, (size_t)DEC_MAX_N * 16384 * 128);
  ALLOC(ctx->d_hyper, float, (size_t)DEC_MAX_N * 4 * 32);
  ARENA16(ctx->d_hyper16, (size_t)(DEC_MAX_N * 4 + 32) * 32);
  ALLOC(ctx->d_masks, float, (size_t)DEC_MAX_N * 4 * 65536);
  ALLOC(ctx->d_iou, float, DEC_MAX_N * 4 + 8);
  ALLOC(ctx->d_obj, float, DEC_MAX_N + 8);
  ALLOC(ctx->d_low_multi, float, 3 * 65536);
  ALLOC(ctx->d_low_sel, float, 65536);
  ALLOC(ctx->d_tok_sel, float, 256);
  ALLOC(ctx->d_best, int, 4);
  ALLOC(ctx->d_iou_sel, float, 4);
  ALLOC(ctx->d_ptr, float, 256);
  ALLOC(ctx->d_pts, float, (size_t)DEC_MAX_N * 64 * 2);
  ALLOC(ctx->d_labels, int, (size_t)DEC_MAX_N * 64);
  ALLOC(ctx->dense_pe, float, 4096 * 256);
  // memory encoder
  ALLOC(ctx->m_mask, float, 1024 * 1024);
  ALLOC(ctx->m_c1, float, 512 * 512 * 4);
  ALLOC(ctx->m_c2, float, 256 * 256 * 16);
  ARENA16(ctx->m_c2_16, 256 * 256 * 16);
  ARENA16(ctx->m_c3_16, 128 * 128 * 64);
  ARENA16(ctx->m_col16, 4096 * 576);
  ALLOC(ctx->m_c4, float, 4096 * 256);
  ARENA16(ctx->m_c4_16, 4096 * 256);
  ALLOC(ctx->m_emb, float, 4096 * 256);
  ALLOC(ctx->m_x, float, 4096 * 256);
  ALLOC(ctx->m_dw, float, 4096 * 256);
  ARENA16(ctx->m_ln16, 4096 * 256);
  ARENA16(ctx->m_h16, 4096 * 1024);
  ALLOC(ctx->m_out, float, 4096 * 64);
  ARENA16(ctx->m_pix16, 4096 * 256);
  // plug scratch
  ALLOC(ctx->p_a, float, (size_t)DEC_MAX_N * 65536 * 32);
  ALLOC(ctx->p_b, float, (size_t)65536 * 32);
  ALLOC(ctx->p_c, float, (size_t)DEC_MAX_N * 16384 * 64);
  ALLOC(ctx->p_d, float, (size_t)DEC_MAX_N * 4096 * 256);
  // video caches
  ctx->feats.resize(c.feat_slots);
  for (auto& f : ctx->feats) {
    ALLOC(f.feat2, float, 4096 * 256);
    ALLOC(f.fpn1, float, 16384 * 64);
    ALLOC(f.fpn0, float, 65536 * 32);
  }
  ctx->bank.resize(c.bank_slots);
  for (auto& b : ctx->bank) {
    ALLOC(b.mem, float, 4096 * 64);
    ALLOC(b.obj_ptr, float, 256);
    ALLOC(b.obj_score, float, 4);
    ALLOC(b.low_mask, float, 65536);
  }
  {
    half_t* base = (half_t*)dalloc(ctx, arena_elems * sizeof(half_t) * (ctx->precise ? 2 : 1));
    if (!base) return sam2mi_set_error(ctx, "hipMalloc", "f16 activation arena");
    for (auto& a : arena16) *a.first = base + a.second;
    ctx->lo16 = ctx->precise ? arena_elems : 0;
  }
  if (ctx->precise) {
    ALLOC(ctx->ws_qk32, float, T0 * 4 * E);
    ALLOC(ctx->ws_vT32, float, T0 * 2 * E);
    ALLOC(ctx->ws_qp32, float, T0 / 4 * 2 * E);
  }
#undef ALLOC
#undef ARENA16
  return 0;
}

// Single-kernel entry points behind the C ABI, used by the GPU parity tests (tests/test_kernels_gpu.py).
#include "engine.h"
#include <cstdlib>

namespace {
struct Tmp {
  std::vector<void*> ptrs;
  ~Tmp() { for (void* p : ptrs) hipFree(p); }
  template <typename T> T* get(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(n, 4) * sizeof(T)) != hipSuccess) return nullptr;
    hipMemset(p, 0, std::max<size_t>(n, 4) * sizeof(T));
    ptrs.push_back(p);
    return (T*)p;
  }
};
__global__ void transpose_to_f16_kernel(const float* __restrict__ in, half_t* __restrict__ out, int R, int Cc, int ldo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * Cc) return;
  const int r = (int)(i / Cc), c = (int)(i % Cc);
  out[(size_t)c * ldo + r] = (half_t)in[i];
}
__global__ void f16_to_f32_kernel(const half_t* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
__global__ void split_to_f32_kernel(const half_t* __restrict__ in, size_t lo_off, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i] + (float)in[i + lo_off] * SPLIT_INV;
}
__global__ void transpose_f32_scale_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc, int ldo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R * Cc) return;
  const int r = (int)(i / Cc), c = (int)(i % Cc);
  out[(size_t)c * ldo + r] = in[i];
}
}  // namespace

extern "C" int sam2mi_debug_gemm(sam2mi_ctx* ctx, void* stream, const float* A, const float* W, const float* bias, int M, int N, int K,
                                 int act, const float* residual, float* out) {
  if (!ctx) return 1;
  hipStream_t s = (hipStream_t)stream;
  Tmp t;
  const bool split = ctx->precise;       // f16x3 context: operands as hi + lo planes, lo right behind hi
  const size_t alo = split ? (size_t)M * K : 0, wlo = split ? (size_t)N * K : 0;
  half_t* a16 = t.get<half_t>((size_t)M * K * (split ? 2 : 1));
  half_t* w16 = t.get<half_t>((size_t)N * K * (split ? 2 : 1));
  if (!a16 || !w16) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
  CHK(cast_add_launch(A, K, nullptr, 0, 0, 0.f, M, K, a16, K, nullptr, 0, s, alo));
  CHK(cast_add_launch(W, K, nullptr, 0, 0, 0.f, N, K, w16, K, nullptr, 0, s, wlo));
  GemmParams p = gemm_params_zero();
  p.a_lo_off = alo; p.w_lo_off = wlo;
  p.A = a16; p.lda = K; p.W = w16; p.ldw = K; p.M = M; p.N = N; p.K = K; p.bias = bias; p.act = act & 0xFF; p.n_split = N;
  p.tile_hint = (act >> 8) & 0xFF;   // tests: force a tile / kernel variant
  p.pool_w = act >> 16;              // fused 2x2 max-pool of the output rows (window side; out then has M / 4 rows)
  p.res = residual; p.ldres = N; p.out32 = out; p.ld32 = N;
  if (p.tile_hint == 31) {               // accumulator-stationary kernel (N = 576, K % 64 == 0)
    if (!gemm_ks_supported(N, K) || (act & 0xFF)) return sam2mi_set_error(ctx, __func__, "gemm_ks needs N == 576, K % 64 == 0, no activation");
    half_t* wp = t.get<half_t>(gemm_ks_pack_bytes(N, K) / 2);
    if (!wp) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
    CHK(gemm_ks_pack(w16, N, K, K, wp, s));
    GemmKsParams k{a16, K, wp, bias, residual, N, out, N, M, K};
    CHK(gemm_ks_launch(k, s));
    CHK(hipStreamSynchronize(s));
    return 0;
  }
  if (p.tile_hint == 32) {               // weight-split X-stationary kernel (f16s): A in f16, W as hi + lo; f16 hi + lo output folded back to f32
    if (!gemm_xs_supported(N, K) || !split || (act & 0xFF) || residual) return sam2mi_set_error(ctx, __func__, "gemm_xs wsplit: split context, K in {144,288,576}, no activation / residual");
    half_t* wp = t.get<half_t>(gemm_xs_wsplit_pack_bytes(N, K) / 2);
    half_t* scratch = t.get<half_t>((size_t)2 * ((N + 31) / 32 * 32) * K);
    half_t* o16 = t.get<half_t>((size_t)M * N * 2);
    if (!wp || !scratch || !o16) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
    CHK(gemm_xs_wsplit_pack(w16, w16 + wlo, N, K, wp, scratch, s));
    GemmXsParams x{a16, K, wp, bias, nullptr, 0, ACT_NONE, M, N, N, o16, N, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0.f, 1, (size_t)M * N};
    CHK(gemm_xs_launch(x, K, s));
    split_to_f32_kernel<<<dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, s>>>(o16, (size_t)M * N, out, (size_t)M * N);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(s));
    return 0;
  }
  if (p.tile_hint == 30) {               // X-stationary kernel (K = 144 / 288 / 576): pack W, then the production dispatch path
    if (!gemm_xs_supported(N, K)) return sam2mi_set_error(ctx, __func__, "gemm_xs needs K in {144,288,576}, N % 8 == 0");
    half_t* wp = t.get<half_t>(gemm_xs_pack_bytes(N, K) / 2);
    if (!wp) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
    CHK(gemm_xs_pack(w16, N, K, K, wp, s));
    GemmXsParams x{a16, K, wp, bias, nullptr, 0, p.act, M, N, N, nullptr, 0, nullptr, 0, out, N, residual, N, 0};
    CHK(gemm_xs_launch(x, K, s));
    CHK(hipStreamSynchronize(s));
    return 0;
  }
  CHKI(run_gemm(ctx, s, p));
  CHK(hipStreamSynchronize(s));
  return 0;
}

// q [groups*GQ, heads*72], k/v [groups*GK, heads*72] f32 -> out [groups*GQ, heads*72] f32
extern "C" int sam2mi_debug_hiera_attention(sam2mi_ctx* ctx, void* stream, const float* q, const float* k, const float* v, int groups,
                                            int heads, int GQ, int GK, int wq, int wk, float* out) {
  if (!ctx) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int C = heads * 72, Mq = groups * GQ, Mk = groups * GK;
  Tmp t;
  if (ctx->selective && ctx->split_attn && (Mk & 7) == 0) {
    // f16s context: q / k as hi + lo planes, V^T in f16, output folded back from hi + lo
    half_t* q16 = t.get<half_t>((size_t)std::max(Mq, Mk) * C * 2);
    half_t* k16 = t.get<half_t>((size_t)std::max(Mq, Mk) * C * 2);
    half_t* vT = t.get<half_t>((size_t)C * Mk);
    half_t* o = t.get<half_t>((size_t)std::max(Mq, Mk) * C * 2);
    if (!q16 || !k16 || !vT || !o) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
    const size_t lo = (size_t)std::max(Mq, Mk) * C;      // one plane distance for q and k (the kernel takes one qk_lo_off)
    CHK(cast_add_launch(q, C, q, C, 0, 1.4426950408889634f / sqrtf(72.f) - 1.f, Mq, C, q16, C, nullptr, 0, s, lo));   // pre-scaled q
    CHK(cast_add_launch(k, C, nullptr, 0, 0, 0.f, Mk, C, k16, C, nullptr, 0, s, lo));
    transpose_to_f16_kernel<<<dim3((unsigned)(((size_t)Mk * C + 255) / 256)), dim3(256), 0, s>>>(v, vT, Mk, C, Mk);
    CHK(hipGetLastError());
    HieraAttnParams a;
    memset(&a, 0, sizeof(a));
    a.q = q16; a.ldq = C; a.k = k16; a.ldk = C; a.vT = vT; a.ldvT = Mk; a.o = o; a.ldo = C; a.heads = heads;
    a.GQ = GQ; a.GK = GK; a.wq = wq; a.wk = wk; a.num_groups = groups; a.scale_log2e = 1.4426950408889634f / sqrtf(72.f);
    a.qk_lo_off = lo; a.o_lo_off = lo;
    CHKI(run_hiera_attn(ctx, s, a));
    split_to_f32_kernel<<<dim3((unsigned)(((size_t)Mq * C + 255) / 256)), dim3(256), 0, s>>>(o, lo, out, (size_t)Mq * C);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(s));
    return 0;
  }
  if (ctx->precise) {                    // f16x3 context: the split-operand kernel on f32 inputs
    float* q32 = t.get<float>((size_t)Mq * C);
    float* vT32 = t.get<float>((size_t)C * Mk);
    half_t* o = t.get<half_t>((size_t)Mq * C * 2);
    if (!q32 || !vT32 || !o) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
    CHK(cast_add_launch(q, C, q, C, 0, 1.4426950408889634f / sqrtf(72.f) - 1.f, Mq, C, nullptr, 0, q32, C, s));   // pre-scaled q
    transpose_f32_scale_kernel<<<dim3((unsigned)(((size_t)Mk * C + 255) / 256)), dim3(256), 0, s>>>(v, vT32, Mk, C, Mk);
    CHK(hipGetLastError());
    PreciseAttnParams a;
    memset(&a, 0, sizeof(a));
    a.q = q32; a.ldq = C; a.k = k; a.ldk = C; a.vT = vT32; a.ldvT = Mk; a.o = o; a.ldo = C; a.o_lo_off = (size_t)Mq * C; a.heads = heads;
    a.GQ = GQ; a.GK = GK; a.wq = wq; a.wk = wk; a.num_groups = groups;
    CHKI(run_precise_attn(ctx, s, a));
    split_to_f32_kernel<<<dim3((unsigned)(((size_t)Mq * C + 255) / 256)), dim3(256), 0, s>>>(o, (size_t)Mq * C, out, (size_t)Mq * C);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(s));
    return 0;
  }
  half_t* q16 = t.get<half_t>((size_t)Mq * C);
  half_t* k16 = t.get<half_t>((size_t)Mk * C);
  half_t* vT = t.get<half_t>((size_t)C * Mk);
  half_t* o16 = t.get<half_t>((size_t)Mq * C);
  if (!q16 || !k16 || !vT || !o16) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
  CHK(cast_add_launch(q, C, q, C, 0, 1.4426950408889634f / sqrtf(72.f) - 1.f, Mq, C, q16, C, nullptr, 0, s));   // pre-scaled q
  CHK(cast_add_launch(k, C, nullptr, 0, 0, 0.f, Mk, C, k16, C, nullptr, 0, s));
  transpose_to_f16_kernel<<<dim3((unsigned)(((size_t)Mk * C + 255) / 256)), dim3(256), 0, s>>>(v, vT, Mk, C, Mk);
  CHK(hipGetLastError());
  HieraAttnParams a;
  memset(&a, 0, sizeof(a));
  a.q = q16; a.ldq = C; a.k = k16; a.ldk = C; a.vT = vT; a.ldvT = Mk; a.o = o16; a.ldo = C; a.heads = heads;
  a.GQ = GQ; a.GK = GK; a.wq = wq; a.wk = wk; a.num_groups = groups; a.scale_log2e = 1.4426950408889634f / sqrtf(72.f);
  CHKI(run_hiera_attn(ctx, s, a));
  f16_to_f32_kernel<<<dim3((unsigned)(((size_t)Mq * C + 255) / 256)), dim3(256), 0, s>>>(o16, out, (size_t)Mq * C);
  CHK(hipGetLastError());
  CHK(hipStreamSynchronize(s));
  return 0;
}

// q [Nq,256], k/v [Nk,256] f32 (RoPE already applied by the caller) -> out [Nq,256] f32
extern "C" int sam2mi_debug_flash256(sam2mi_ctx* ctx, void* stream, const float* q, const float* k, const float* v, int Nq, int Nk, float* out) {
  if (!ctx) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int NkP = (Nk + 31) / 32 * 32;
  Tmp t;
  half_t* q16 = t.get<half_t>((size_t)Nq * 256);
  half_t* k16 = t.get<half_t>((size_t)NkP * 256);
  half_t* vT = t.get<half_t>((size_t)256 * NkP);
  half_t* o16 = t.get<half_t>((size_t)Nq * 256);
  const int splits = flash256_pick_splits(Nq, Nk);
  float* opart = t.get<float>((size_t)splits * Nq * 256);
  float* ml = t.get<float>((size_t)splits * Nq * 2);
  if (!q16 || !k16 || !vT || !o16 || !opart || !ml) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
  CHK(cast_add_launch(q, 256, q, 256, 0, 1.4426950408889634f / 16.f - 1.f, Nq, 256, q16, 256, nullptr, 0, s));   // pre-scaled q
  CHK(cast_add_launch(k, 256, nullptr, 0, 0, 0.f, Nk, 256, k16, 256, nullptr, 0, s));
  transpose_to_f16_kernel<<<dim3((unsigned)(((size_t)Nk * 256 + 255) / 256)), dim3(256), 0, s>>>(v, vT, Nk, 256, NkP);
  CHK(hipGetLastError());
  Flash256Params f;
  memset(&f, 0, sizeof(f));
  f.q = q16; f.ldq = 256; f.k = k16; f.ldk = 256; f.vT = vT; f.ldvT = NkP; f.Nq = Nq; f.Nk = Nk; f.splits = splits;
  f.o_part = opart; f.ml_part = ml; f.out = o16; f.ldout = 256; f.scale_log2e = 1.4426950408889634f / 16.f;
  CHKI(run_flash256(ctx, s, f));
  f16_to_f32_kernel<<<dim3((unsigned)(((size_t)Nq * 256 + 255) / 256)), dim3(256), 0, s>>>(o16, out, (size_t)Nq * 256);
  CHK(hipGetLastError());
  CHK(hipStreamSynchronize(s));
  return 0;
}

// gemm_rowln_kernel on its own: partials [splits, M, 256] + ml [splits, M, 2] (or, splits == 0, a plain operand a [M, 256] f32 that is
// rounded to f16), W [256, 256] f32, bias, residual x [M, 256] (updated in place), LayerNorm weights -> h [M, 256] f32 (from f16)
extern "C" int sam2mi_debug_rowln(sam2mi_ctx* ctx, void* stream, const float* a_or_parts, const float* ml, int splits, const float* W,
                                  const float* bias, float* x, const float* ln_w, const float* ln_b, int M, float* h) {
  if (!ctx) return 1;
  hipStream_t s = (hipStream_t)stream;
  Tmp t;
  half_t* w16 = t.get<half_t>(256 * 256);
  half_t* a16 = t.get<half_t>((size_t)M * 256);
  half_t* h16 = t.get<half_t>((size_t)M * 256);
  if (!w16 || !a16 || !h16) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
  CHK(cast_add_launch(W, 256, nullptr, 0, 0, 0.f, 256, 256, w16, 256, nullptr, 0, s));
  RowLnParams r;
  memset(&r, 0, sizeof(r));
  if (splits > 0) { r.o_part = a_or_parts; r.ml_part = ml; r.splits = splits; r.part_rows = M; }
  else { CHK(cast_add_launch(a_or_parts, 256, nullptr, 0, 0, 0.f, M, 256, a16, 256, nullptr, 0, s)); r.a16 = a16; r.lda = 256; }
  r.w = w16; r.bias = bias; r.res = x; r.out32 = x; r.ln_w = ln_w; r.ln_b = ln_b; r.eps = 1e-5f; r.out16 = h16; r.ld16 = 256; r.M = M;
  CHKI(run_rowln(ctx, s, r));
  f16_to_f32_kernel<<<dim3((unsigned)(((size_t)M * 256 + 255) / 256)), dim3(256), 0, s>>>(h16, h, (size_t)M * 256);
  CHK(hipGetLastError());
  CHK(hipStreamSynchronize(s));
  return 0;
}

// gemm_projln_kernel on its own: a [M, C] f32 (rounded to f16), W [C, C] f32 (rounded, packed), bias, residual x [M, C] (updated in place),
// LayerNorm weights (eps 1e-6) -> h [M, C] f32 (from f16).  C in {144, 288, 576}, M % 32 == 0.
extern "C" int sam2mi_debug_projln(sam2mi_ctx* ctx, void* stream, const float* a, const float* W, const float* bias, float* x, const float* ln_w,
                                   const float* ln_b, int M, int C, float* h) {
  if (!ctx) return 1;
  if (!gemm_projln_supported(C)) return sam2mi_set_error(ctx, __func__, "C must be 144, 288 or 576");
  hipStream_t s = (hipStream_t)stream;
  Tmp t;
  half_t* w16 = t.get<half_t>((size_t)C * C);
  half_t* wpk = t.get<half_t>(gemm_xs_pack_bytes(C, C) / 2);
  half_t* a16 = t.get<half_t>((size_t)M * C);
  half_t* h16 = t.get<half_t>((size_t)M * C);
  if (!w16 || !wpk || !a16 || !h16) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
  CHK(cast_add_launch(W, C, nullptr, 0, 0, 0.f, C, C, w16, C, nullptr, 0, s));
  CHK(gemm_xs_pack(w16, C, C, C, wpk, s));
  CHK(cast_add_launch(a, C, nullptr, 0, 0, 0.f, M, C, a16, C, nullptr, 0, s));
  ProjLnParams q{a16, C, wpk, bias, x, x, ln_w, ln_b, 1e-6f, h16, C, M, C};
  CHKI(run_projln(ctx, s, q));
  f16_to_f32_kernel<<<dim3((unsigned)(((size_t)M * C + 255) / 256)), dim3(256), 0, s>>>(h16, h, (size_t)M * C);
  CHK(hipGetLastError());
  CHK(hipStreamSynchronize(s));
  return 0;
}

// One Hiera block on x [B, H, W, C] (row-major NHWC, H = W = the grid of that block's stage) -> out NHWC
extern "C" int sam2mi_debug_hiera_block(sam2mi_ctx* ctx, void* stream, int block_idx, const float* x_nhwc, int B, float* out_nhwc) {
  if (!ctx || !ctx->finalized) return sam2mi_set_error(ctx, __func__, "weights not finalized");
  if (block_idx < 0 || block_idx >= (int)ctx->blocks.size()) return sam2mi_set_error(ctx, __func__, "bad block index");
  hipStream_t s = (hipStream_t)stream;
  const HieraBlockW& b = ctx->blocks[block_idx];
  int G = ctx->cfg.image_size / 4;
  for (int i = 0; i < block_idx; ++i) if (ctx->blocks[i].q_pool) G /= 2;
  int H = G, W = G;
  int w = b.window > 0 ? b.window : 16;
  if (B > ctx->cfg.max_batch) return sam2mi_set_error(ctx, __func__, "batch exceeds max_batch");
  if (ctx->generic) {                      // padded-window sizes: row-major tokens in and out
    CHK(hipMemcpyAsync(ctx->ws_x, x_nhwc, (size_t)B * H * W * b.dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    CHKI(hiera_block_forward_generic(ctx, s, b, B, H, W));
    CHK(hipMemcpyAsync(out_nhwc, ctx->ws_x, (size_t)B * H * W * b.dim_out * sizeof(float), hipMemcpyDeviceToDevice, s));
    CHK(hipStreamSynchronize(s));
    return 0;
  }
  CHK(permute_tokens_launch(x_nhwc, ctx->ws_x, B, H, W, b.dim, W, w, nullptr, 0, s));
  CHKI(hiera_block_forward(ctx, s, b, B, H, W, w));
  CHK(permute_tokens_launch(ctx->ws_x, out_nhwc, B, H, W, b.dim_out, w, W, nullptr, 0, s));
  CHK(hipStreamSynchronize(s));
  return 0;
}

// copy a named internal buffer (tests only)
extern "C" int sam2mi_debug_read(sam2mi_ctx* ctx, void* stream, const char* name, float* out, int64_t count) {
  if (!ctx) return 1;
  const std::string n(name);
  const float* src = nullptr;
  if (n == "d_tok") src = ctx->d_tok;
  else if (n == "d_keys") src = ctx->d_keys;
  else if (n == "d_t1") src = ctx->d_t1;
  else if (n == "d_t2") src = ctx->d_t2;
  else if (n == "d_big1") src = ctx->d_big1;
  else if (n == "d_big2") src = ctx->d_big2;
  else if (n == "t_pix") src = ctx->t_pix;
  else if (n == "m_out") src = ctx->m_out;
  else if (n == "t_x") src = ctx->t_x;
  else if (n == "t_kall16") src = reinterpret_cast<const float*>(ctx->t_kall16);       // f16 buffers: raw bits, count = halfs / 2
  else if (n == "t_vTall16") src = reinterpret_cast<const float*>(ctx->t_vTall16);
  else if (n == "t_kin16") src = reinterpret_cast<const float*>(ctx->t_kin16);
  else if (n == "t_vin16") src = reinterpret_cast<const float*>(ctx->t_vin16);
  else return sam2mi_set_error(ctx, __func__, "unknown buffer");
  CHK(hipMemcpyAsync(out, src, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  CHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// Hiera MLP on its own:  x += fc2(GELU(fc1(xn)))  with xn [M,C], W1 [4C,C], W2 [C,4C] f32 (rounded to f16 here), x [M,C] f32 in/out.
// fused != 0: mlp_fused_kernel; 0: the two-GEMM path (fc1 + GELU epilogue -> f16 hidden -> fc2 + residual epilogue).
// iters > 0 additionally times that many launches on a scratch copy of x (ms per launch in *ms_out).
extern "C" int sam2mi_debug_mlp(sam2mi_ctx* ctx, void* stream, const float* xn, const float* W1, const float* b1, const float* W2,
                                const float* b2, float* x, int M, int C, int fused, int iters, float* ms_out) {
  if (!ctx) return 1;
  if (fused && !mlp_fused_supported(C)) return sam2mi_set_error(ctx, __func__, "fused MLP supports C = 144 / 288");
  hipStream_t s = (hipStream_t)stream;
  const int H4 = 4 * C;
  Tmp t;
  half_t* x16 = t.get<half_t>((size_t)M * C);
  half_t* w1 = t.get<half_t>((size_t)H4 * C);
  half_t* w2 = t.get<half_t>((size_t)C * H4);
  half_t* h16 = fused ? nullptr : t.get<half_t>((size_t)M * H4);
  half_t* wpk = fused ? t.get<half_t>(mlp_fused_pack_bytes(C) / 2) : nullptr;
  float* xs = iters > 0 ? t.get<float>((size_t)M * C) : nullptr;
  if (!x16 || !w1 || !w2 || (!fused && !h16) || (fused && !wpk) || (iters > 0 && !xs)) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
  CHK(cast_add_launch(xn, C, nullptr, 0, 0, 0.f, M, C, x16, C, nullptr, 0, s));
  CHK(cast_add_launch(W1, C, nullptr, 0, 0, 0.f, H4, C, w1, C, nullptr, 0, s));
  CHK(cast_add_launch(W2, H4, nullptr, 0, 0, 0.f, C, H4, w2, H4, nullptr, 0, s));
  if (fused) CHK(mlp_fused_pack(w1, w2, C, wpk, s));
  auto run = [&](float* xio) -> int {
    if (fused) {
      MlpFusedParams m{x16, C, wpk, b1, b2, xio, C, M};
      CHK(mlp_fused_launch(m, C, s));
    } else {
      GemmParams p = gemm_params_zero();
      p.A = x16; p.lda = C; p.W = w1; p.ldw = C; p.M = M; p.N = H4; p.K = C; p.n_split = H4; p.bias = b1; p.act = ACT_GELU;
      p.out16 = h16; p.ld16 = H4;
      CHK(gemm_launch(p, s));
      GemmParams q = gemm_params_zero();
      q.A = h16; q.lda = H4; q.W = w2; q.ldw = H4; q.M = M; q.N = C; q.K = H4; q.n_split = C; q.bias = b2;
      q.res = xio; q.ldres = C; q.out32 = xio; q.ld32 = C;
      CHK(gemm_launch(q, s));
    }
    return 0;
  };
  if (iters > 0) {
    CHK(hipMemcpyAsync(xs, x, (size_t)M * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int i = 0; i < 2; ++i) CHKI(run(xs));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) CHKI(run(xs));
    CHK(hipEventRecord(e1, s));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  CHKI(run(x));
  CHK(hipStreamSynchronize(s));
  return 0;
}

// time `iters` launches of the production GEMM on an [M,K] x [N,K]^T problem (random f16 operands); returns ms per launch
extern "C" int sam2mi_debug_gemm_bench(sam2mi_ctx* ctx, void* stream, int M, int N, int K, int iters, int mode, float* ms_out) {
  const int tile_hint = mode >> 4;
  mode &= 15;
  if (!ctx) return 1;
  hipStream_t s = (hipStream_t)stream;
  Tmp t;
  half_t* a16 = t.get<half_t>((size_t)M * K);
  half_t* w16 = t.get<half_t>((size_t)N * K);
  half_t* o16 = t.get<half_t>((size_t)M * N);
  float* o32 = t.get<float>((size_t)M * N);
  float* tmp = t.get<float>((size_t)std::max(M, N) * K);
  if (!a16 || !w16 || !o16 || !o32 || !tmp) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
  // pseudo-random fill via the bf16-round kernel on an iota-ish pattern is not needed: use cast of uninitialised zeros + pattern
  std::vector<float> h((size_t)std::max(M, N) * K);
  uint32_t x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 9) & 0xFFFF) / 32768.0f - 1.0f; }
  CHK(hipMemcpy(tmp, h.data(), (size_t)M * K * sizeof(float), hipMemcpyHostToDevice));
  CHK(cast_add_launch(tmp, K, nullptr, 0, 0, 0.f, M, K, a16, K, nullptr, 0, s));
  CHK(hipMemcpy(tmp, h.data(), (size_t)N * K * sizeof(float), hipMemcpyHostToDevice));
  CHK(cast_add_launch(tmp, K, nullptr, 0, 0, 0.f, N, K, w16, K, nullptr, 0, s));
  GemmParams p = gemm_params_zero();
  p.A = a16; p.lda = K; p.W = w16; p.ldw = K; p.M = M; p.N = N; p.K = K; p.n_split = N; p.tile_hint = tile_hint;
  if (getenv("SAM2MI_BENCH_NOMEM")) { p.lda = 0; p.ldw = 0; }   // tuning aid: every tile reads the same rows (cache-resident operands)
  if (mode == 0) { p.out16 = o16; p.ld16 = N; }            // f16 output (QKV / fc1 style)
  else if (mode == 1) { p.out32 = o32; p.ld32 = N; p.res = o32; p.ldres = N; }   // f32 in-place residual (proj / fc2 style)
  else if (mode == 3) { p.out16 = o16; p.ld16 = N; p.act = ACT_GELU; }          // f16 output through GELU (fc1 style)
  // mode 2: no output at all (main-loop-only timing, tuning aid)
  half_t* wpk = nullptr;
  GemmXsParams xsp{};
  if (tile_hint == 30) {
    if (!gemm_xs_supported(N, K)) return sam2mi_set_error(ctx, __func__, "gemm_xs needs K in {144,288,576}, N % 8 == 0");
    wpk = t.get<half_t>(gemm_xs_pack_bytes(N, K) / 2);
    float* bz = t.get<float>((size_t)N);
    if (!wpk || !bz) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
    CHK(hipMemsetAsync(bz, 0, (size_t)N * sizeof(float), s));
    CHK(gemm_xs_pack(w16, N, K, K, wpk, s));
    xsp = GemmXsParams{a16, K, wpk, bz, nullptr, 0, p.act, M, N, N, p.out16, p.ld16, nullptr, 0, p.out32, p.ld32, p.res, p.ldres, 0};
  }
  GemmKsParams ksp{};
  if (tile_hint == 31) {
    if (!gemm_ks_supported(N, K) || mode != 1) return sam2mi_set_error(ctx, __func__, "gemm_ks needs N == 576, K % 64 == 0, mode 1");
    wpk = t.get<half_t>(gemm_ks_pack_bytes(N, K) / 2);
    float* bz = t.get<float>((size_t)N);
    if (!wpk || !bz) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
    CHK(hipMemsetAsync(bz, 0, (size_t)N * sizeof(float), s));
    CHK(gemm_ks_pack(w16, N, K, K, wpk, s));
    ksp = GemmKsParams{a16, K, wpk, bz, o32, N, o32, N, M, K};
  }
  auto launch = [&]() -> hipError_t {
    return tile_hint == 30 ? gemm_xs_launch(xsp, K, s) : (tile_hint == 31 ? gemm_ks_launch(ksp, s) : gemm_launch(p, s));
  };
  for (int i = 0; i < 3; ++i) CHK(launch());
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CHK(launch());
  CHK(hipEventRecord(e1, s));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

// time `iters` launches of the d=256 flash attention + combine (pseudo-random f16 operands); ms per launch
extern "C" int sam2mi_debug_flash_bench(sam2mi_ctx* ctx, void* stream, int Nq, int Nk, int iters, float* ms_out) {
  if (!ctx) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int NkP = (Nk + 31) / 32 * 32;
  Tmp t;
  half_t* q16 = t.get<half_t>((size_t)Nq * 256);
  half_t* k16 = t.get<half_t>((size_t)NkP * 256);
  half_t* vT = t.get<half_t>((size_t)256 * NkP);
  half_t* o16 = t.get<half_t>((size_t)Nq * 256);
  const int splits = flash256_pick_splits(Nq, Nk);
  float* opart = t.get<float>((size_t)splits * Nq * 256);
  float* ml = t.get<float>((size_t)splits * Nq * 2);
  const size_t nmax = (size_t)std::max(Nq, NkP) * 256;
  float* tmp = t.get<float>(nmax);
  if (!q16 || !k16 || !vT || !o16 || !opart || !ml || !tmp) return sam2mi_set_error(ctx, __func__, "hipMalloc failed");
  std::vector<float> h(nmax);
  uint32_t x = 777u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (((x >> 9) & 0xFFFF) / 32768.0f - 1.0f) * 0.25f; }
  CHK(hipMemcpy(tmp, h.data(), nmax * sizeof(float), hipMemcpyHostToDevice));
  CHK(cast_add_launch(tmp, 256, nullptr, 0, 0, 0.f, Nq, 256, q16, 256, nullptr, 0, s));
  CHK(cast_add_launch(tmp, 256, nullptr, 0, 0, 0.f, NkP, 256, k16, 256, nullptr, 0, s));
  CHK(cast_add_launch(tmp, NkP, nullptr, 0, 0, 0.f, 256, NkP, vT, NkP, nullptr, 0, s));
  Flash256Params f;
  memset(&f, 0, sizeof(f));
  f.q = q16; f.ldq = 256; f.k = k16; f.ldk = 256; f.vT = vT; f.ldvT = NkP; f.Nq = Nq; f.Nk = Nk; f.splits = splits;
  f.o_part = opart; f.ml_part = ml; f.out = o16; f.ldout = 256; f.scale_log2e = 1.4426950408889634f / 16.f;
  for (int i = 0; i < 2; ++i) CHK(flash256_launch(f, s));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) CHK(flash256_launch(f, s));
  CHK(hipEventRecord(e1, s));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

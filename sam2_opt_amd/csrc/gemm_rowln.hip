// Tail of an attention sub-block of the memory attention in ONE kernel (gfx950):
//
//     o   = combine(flash partials)                  (or a plain f16 [M,256] operand)
//     x  += o W^T + b                                 out-projection + residual, f32 residual stream
//     h   = LayerNorm(x) * gamma + beta  -> f16       the operand of the NEXT projection (norm2 / norm3)
//
// Reference: MemoryAttentionLayer._forward_sa / _forward_ca, /root/reference/sam2/sam2/modeling/memory_attention.py:60-91
// (tgt = tgt + dropout(self_attn(...)) ; tgt2 = self.norm2(tgt) ...) and RoPEAttention's out_proj (sam/transformer.py:420-424).
//
// Why: at M = 4096 these were three dependent launches (flash256_combine_kernel 8 us, a 64x64 GEMM 8 us, layernorm 5 us), each
// one reading what the previous one had just written through the memory-side cache.  d_model = 256 is one MFMA-friendly row:
// a workgroup that owns 32 full rows can do all three steps without leaving the CU.
//   * 8 waves, 32 rows x 256 columns per workgroup; wave w owns columns [32 w, 32 w + 32) (one 32x32 MFMA tile, K = 256)
//   * W fragments (B operand) are read straight from global memory (128 KB, L2 resident, the same for every workgroup):
//     16 x 16-B loads per lane issued BEFORE the operand tile is built, so their latency is hidden behind the combine
//   * the combine keeps the partial loads of 8 splits in flight at once (32 x 16 B per thread): the partials have just been
//     written by the flash kernel and come from the memory-side cache, one round trip per split would dominate the kernel
//   * the combined / loaded operand tile goes to LDS as f16 [32][256] (16-B chunks XOR-ed with the row: conflict-free b128 reads)
//   * the accumulators (+ bias) are transposed through an f32 LDS tile; then one wave per row (4 rows per wave): residual add,
//     f32 store of x, two-pass mean / variance (the form layernorm_vec_kernel and PyTorch use), f16 store of h.
// The f16x3 precision mode does not use this kernel (its operands are split f16 pairs); engine_track.hip keeps the three-kernel
// path there.
#include "gemm.h"

namespace {
constexpr int RL_BM = 32, RL_C = 256;
constexpr int RL_XLD = 264;                               // f32 tile row stride: 4 rows apart = 32 banks apart
constexpr int RL_X_BYTES = RL_BM * RL_XLD * 4;            // 33 KiB

// KC: channels of the operand (K of the product): 256, or 64 = the flash partials of the cross-attention with the values in the
// memory space (attn_flash256.hip, DV = 64): W is then Wo Wv [256, 64] and the bias Wo bv + bo, composed at weight-load time
template <int KC>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(const RowLnParams p) {
  constexpr int RL_A_BYTES = RL_BM * KC * 2;              // 16 KiB / 4 KiB
  constexpr int ROWB = KC * 2;                            // operand tile row bytes
  constexpr int SWZ = KC / 8 < 16 ? KC / 8 - 1 : 15;      // 16-B chunks of a row are XOR-ed with (row & SWZ)
  constexpr int NG = KC / 64;                             // f32x4 groups per thread in the combine (16 threads per row)
  constexpr int KS = KC / 16;                             // k-steps
  __shared__ __attribute__((aligned(16))) char smem[RL_A_BYTES + RL_X_BYTES];
  char* sA = smem;
  float* sX = reinterpret_cast<float*>(smem + RL_A_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 8 waves: wave w owns columns [32 w, 32 w + 32)
  const int fr = lane & 31, fh = lane >> 5;
  const int m0 = blockIdx.x * RL_BM;

  // ---- W fragments: lane holds W[n = 32 w + fr][k = 16 s + 8 fh .. + 8)
  half8 wf[KS];
  {
    const half_t* wr = p.w + (size_t)(wave * 32 + fr) * KC + fh * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[s] = *reinterpret_cast<const half8*>(wr + s * 16);
  }
  // ---- operand tile -> LDS f16.  Thread: row = tid / 16, 16-B f32 groups (sub + 16 i) (coalesced 256-B segments per row)
  {
    const int row = tid >> 4, sub = tid & 15;
    const size_t m = (size_t)(m0 + row);
    if (p.o_part) {
      f32x4 acc[NG];
#pragma unroll
      for (int i = 0; i < NG; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      float mstar = -1e30f;
      for (int s = 0; s < p.splits; ++s) mstar = fmaxf(mstar, p.ml_part[((size_t)s * p.part_rows + m) * 2]);
      float L = 0.f;
      // 8 splits per round, their 32 loads all in flight (a round trip to the memory-side cache per split otherwise)
      for (int s0 = 0; s0 < p.splits; s0 += 8) {
        f32x4 v[8][NG];
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int s = min(s0 + u, p.splits - 1);
          const float* op = p.o_part + ((size_t)s * p.part_rows + m) * KC;
#pragma unroll
          for (int i = 0; i < NG; ++i) v[u][i] = *reinterpret_cast<const f32x4*>(op + (sub + 16 * i) * 4);
          const float* ml = p.ml_part + ((size_t)s * p.part_rows + m) * 2;
          w[u] = (s0 + u < p.splits) ? exp2f(ml[0] - mstar) : 0.f;
          L += w[u] * ml[1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < NG; ++i) {
            acc[i][0] += w[u] * v[u][i][0]; acc[i][1] += w[u] * v[u][i][1]; acc[i][2] += w[u] * v[u][i][2]; acc[i][3] += w[u] * v[u][i][3];
          }
      }
      const float inv = 1.f / L;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int g = sub + 16 * i;                                 // f32 group = 4 channels = half a 16-B f16 chunk
        const int chunk = g >> 1;
        const half4 h = {(half_t)(acc[i][0] * inv), (half_t)(acc[i][1] * inv), (half_t)(acc[i][2] * inv), (half_t)(acc[i][3] * inv)};
        *reinterpret_cast<half4*>(sA + row * ROWB + ((chunk ^ (row & SWZ)) << 4) + (g & 1) * 8) = h;
      }
    } else {
      const half_t* ar = p.a16 + m * p.lda;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int chunk = sub + 16 * i;
        if (chunk < KC / 8) {
          const half8 v = *reinterpret_cast<const half8*>(ar + chunk * 8);
          *reinterpret_cast<half8*>(sA + row * ROWB + ((chunk ^ (row & SWZ)) << 4)) = v;
        }
      }
    }
  }
  // ---- residual rows of the LayerNorm phase, in flight during the products (wave w: rows 4 w .. 4 w + 3, lane: 4 consecutive channels)
  f32x4 rv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rv[i] = *reinterpret_cast<const f32x4*>(p.res + (size_t)(m0 + wave * 4 + i) * RL_C + lane * 4);

  __syncthreads();

  // ---- 32 x 32 per wave, K = KC (two accumulation chains)
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; s += 2) {
    const half8 a0 = *reinterpret_cast<const half8*>(sA + fr * ROWB + (((2 * s + fh) ^ (fr & SWZ)) << 4));
    const half8 a1 = *reinterpret_cast<const half8*>(sA + fr * ROWB + (((2 * s + 2 + fh) ^ (fr & SWZ)) << 4));
    acc0 = mfma32(a0, wf[s], acc0);
    acc1 = mfma32(a1, wf[s + 1], acc1);
  }
  {
    const int n = wave * 32 + fr;
    const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sX[acc_row(r, lane) * RL_XLD + n] = acc0[r] + acc1[r] + b;
  }
  __syncthreads();

  // ---- residual + LayerNorm: one wave per row
  const f32x4 gw = *reinterpret_cast<const f32x4*>(p.ln_w + lane * 4), gb = *reinterpret_cast<const f32x4*>(p.ln_b + lane * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 4 + i;
    const size_t m = (size_t)(m0 + row);
    const f32x4 t = *reinterpret_cast<const f32x4*>(sX + row * RL_XLD + lane * 4);
    const f32x4 x = {t[0] + rv[i][0], t[1] + rv[i][1], t[2] + rv[i][2], t[3] + rv[i][3]};
    *reinterpret_cast<f32x4*>(p.out32 + m * RL_C + lane * 4) = x;
    const float mean = wave_sum((x[0] + x[1]) + (x[2] + x[3])) * (1.f / RL_C);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = x[e] - mean;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / RL_C) + p.eps);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (x[e] - mean) * rstd * gw[e] + gb[e];
    const half4 h = {(half_t)y[0], (half_t)y[1], (half_t)y[2], (half_t)y[3]};
    *reinterpret_cast<half4*>(p.out16 + m * p.ld16 + lane * 4) = h;
  }
}
}  // namespace

hipError_t gemm_rowln_launch(const RowLnParams& p, hipStream_t s) {
  if (p.M <= 0 || p.M % RL_BM || !p.w || !p.res || !p.out32 || !p.out16 || !p.ln_w || !p.ln_b || (p.ld16 & 3)) return hipErrorInvalidValue;
  if (p.o_part ? (!p.ml_part || p.splits < 1 || p.part_rows < p.M) : (!p.a16 || (p.lda & 7))) return hipErrorInvalidValue;
  if (p.kc == 64) gemm_rowln_kernel<64><<<dim3(p.M / RL_BM), dim3(512), 0, s>>>(p);
  else if (p.kc == 0 || p.kc == 256) gemm_rowln_kernel<256><<<dim3(p.M / RL_BM), dim3(512), 0, s>>>(p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

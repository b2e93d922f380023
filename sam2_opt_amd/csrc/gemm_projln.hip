// Output projection of a Hiera block + residual + the block's second LayerNorm in ONE kernel (gfx950):
//
//     x  += att W^T + b                      attention out-projection onto the f32 residual stream   (N = K = C)
//     h   = LayerNorm(x) * gamma + beta  -> f16   the operand of the MLP (norm2)
//
// Reference: MultiScaleBlock.forward, /root/reference/sam2/sam2/modeling/backbones/hieradet.py:161-165
// (x = shortcut + drop_path(x) after self.attn, whose last step is self.proj (:78-80); then self.mlp(self.norm2(x))).
//
// Why: the projection GEMM is HBM-bound (f32 residual in and out, K = N = C small) and the LayerNorm behind it re-reads the
// 4 C bytes per element the GEMM has just written.  A workgroup that owns 32 FULL rows (all C columns) normalises them on the way
// out: x is read once and written once, h is written once - 299 -> 224 MB per stage-3 launch pair.
//   * C = 144 / 288 / 576 (stages 1-3 of hiera-large); 32 rows x C columns per workgroup; wave w owns TPW column tiles of 32
//   * W comes from the X-stationary kernel's packed image (gemm_xs_pack: every 1-KiB piece is one MFMA fragment tile), read
//     straight from L2 into registers, three k-steps ahead of the MFMAs that use them
//   * the f16 operand tile [32][C] sits in LDS (16-B chunks XOR-ed with the row so that ds_read_b128 is conflict-free for the
//     288 / 576 / 1152-byte row strides); the f32 result tile re-uses that LDS after the products
//   * row phase: one wave per row, the layernorm_vec_kernel arithmetic (two-pass mean / variance), residual rows requested before
//     the barrier in front of it.
// SP (f16s precision mode): 2 = the weight as a 2-term f16 split (second packed image `wpack_lo`, its products in a second accumulator
// set folded in x 2^-11), 3 = the operand split as well (lo plane `a_lo_off` elements behind a16: three MFMAs per fragment pair) - the
// stage-1 / stage-2 projections of the f16s plan.  The f16x3 mode keeps GEMM + LayerNorm.
#include "gemm.h"
#include "gemm_xs.h"

namespace {
template <int C>
struct PL {
  static constexpr int KS = C / 16;                          // k-steps
  static constexpr int NT = (C + 31) / 32;                   // column tiles (C = 144: the fifth is half empty, zero rows in the pack)
  static constexpr int TPW = C == 576 ? 3 : 1;               // tiles per wave
  static constexpr int NW = (NT + TPW - 1) / TPW;            // 5 / 9 / 6 waves
  static constexpr int CPS = XS_STAGE_PIECES / KS;           // packed image: chunks of 32 columns per stage
  static constexpr int CH = C / 8;                           // 16-B chunks per operand row
  static constexpr int SWZ_SHIFT = C == 144 ? 3 : C == 288 ? 2 : 1, SWZ_MASK = C == 144 ? 1 : C == 288 ? 3 : 7;
  static constexpr int XLD = C + 8;                          // f32 tile row stride: 4 rows apart = 32 banks apart
  static constexpr int A_BYTES = 32 * C * 2, X_BYTES = 32 * XLD * 4;
  static constexpr int LDS_B = A_BYTES > X_BYTES ? A_BYTES : X_BYTES;
  static constexpr int LDS_B2 = 2 * A_BYTES > X_BYTES ? 2 * A_BYTES : X_BYTES;       // operand split as well: hi tile + lo tile
  static constexpr int VPL = (C / 4 + 63) / 64;              // float4 per lane in the row phase
  static constexpr int RPW = (32 + NW - 1) / NW;             // rows per wave (at most)
  static constexpr int G = 3;                                // k-steps per weight-prefetch group
  static_assert(KS % G == 0, "k-steps per group");
};

template <int C, int SP = 0>
__global__ __launch_bounds__(PL<C>::NW * 64) void gemm_projln_kernel(const ProjLnParams p) {
  using L = PL<C>;
  constexpr bool WS = SP >= 2, AS = SP == 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // L::LDS_B bytes (74,752 at C = 576: dynamic)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int m0 = blockIdx.x * 32;

  // ---- weight pieces of this wave's tiles: piece (tile t, k-step s) of the packed image, lane offset = its B-operand position
  const char* wp = reinterpret_cast<const char*>(p.wpack) + fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);
  auto piece = [&](int t, int s) -> const half8* {
    const int stage = t / L::CPS, q = (t % L::CPS) * L::KS + s;
    return reinterpret_cast<const half8*>(wp + ((size_t)stage * XS_STAGE_SLOTS + q) * 1024);
  };
  const long wlo = WS ? reinterpret_cast<const char*>(p.wpack_lo) - reinterpret_cast<const char*>(p.wpack) : 0;      // bytes from a hi piece to its lo piece
  struct WG { half8 h[L::G][L::TPW]; half8 l[WS ? L::G : 1][WS ? L::TPW : 1]; };
  WG wa, wb;
  auto load_group = [&](WG& w, int g) {
#pragma unroll
    for (int j = 0; j < L::G; ++j)
#pragma unroll
      for (int i = 0; i < L::TPW; ++i) {
        const half8* pp = piece(min(wave * L::TPW + i, L::NT - 1), g * L::G + j);
        w.h[j][i] = *pp;
        if constexpr (WS) w.l[j][i] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(pp) + wlo);
      }
  };
  load_group(wa, 0);

  // ---- operand tile -> LDS (coalesced 16-B pieces, chunk index XOR-ed with the row)
  for (int i = tid; i < 32 * L::CH; i += L::NW * 64) {
    const int row = i / L::CH, ch = i % L::CH;
    const half_t* ap = p.a16 + (size_t)(m0 + row) * p.lda + ch * 8;
    char* dst = smem + row * (C * 2) + ((ch ^ ((row >> L::SWZ_SHIFT) & L::SWZ_MASK)) << 4);
    *reinterpret_cast<half8*>(dst) = *reinterpret_cast<const half8*>(ap);
    if constexpr (AS) *reinterpret_cast<half8*>(dst + L::A_BYTES) = *reinterpret_cast<const half8*>(ap + p.a_lo_off);      // lo tile behind the hi tile
  }
  __syncthreads();

  f32x16 acc[L::TPW], accx[WS ? L::TPW : 1];
#pragma unroll
  for (int i = 0; i < L::TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[i][r] = 0.f;
      if constexpr (WS) accx[i][r] = 0.f;
    }
  const char* arow = smem + fr * (C * 2);
  const int aswz = (fr >> L::SWZ_SHIFT) & L::SWZ_MASK;
  auto mma_group = [&](const WG& w, int g) {
#pragma unroll
    for (int j = 0; j < L::G; ++j) {
      const int s = g * L::G + j;
      const half8 a = *reinterpret_cast<const half8*>(arow + (((2 * s + fh) ^ aswz) << 4));
      half8 al;
      if constexpr (AS) al = *reinterpret_cast<const half8*>(arow + L::A_BYTES + (((2 * s + fh) ^ aswz) << 4));
#pragma unroll
      for (int i = 0; i < L::TPW; ++i) {
        acc[i] = mfma32(a, w.h[j][i], acc[i]);
        if constexpr (WS) accx[i] = mfma32(a, w.l[j][i], accx[i]);
        if constexpr (AS) accx[i] = mfma32(al, w.h[j][i], accx[i]);
      }
    }
  };
  constexpr int NG = L::KS / L::G;
#pragma unroll
  for (int g = 0; g < NG; g += 2) {
    if (g + 1 < NG) load_group(wb, g + 1);
    mma_group(wa, g);
    if (g + 1 < NG) {
      if (g + 2 < NG) load_group(wa, g + 2);
      mma_group(wb, g + 1);
    }
  }

  // ---- residual rows of the row phase (wave w: rows w, w + NW, ...), requested before the barrier
  f32x4 rv[L::RPW][L::VPL];
#pragma unroll
  for (int i = 0; i < L::RPW; ++i) {
    const int row = min(wave + i * L::NW, 31);
#pragma unroll
    for (int k = 0; k < L::VPL; ++k) {
      const int v = min(lane + 64 * k, C / 4 - 1);
      rv[i][k] = *reinterpret_cast<const f32x4*>(p.res + (size_t)(m0 + row) * C + 4 * v);
    }
  }
  __syncthreads();                                            // every wave is done with the operand tile
  float* sX = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < L::TPW; ++i) {
    const int n = (wave * L::TPW + i) * 32 + fr;
    if (wave * L::TPW + i < L::NT && n < C) {
      const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sX[acc_row(r, lane) * L::XLD + n] = (WS ? fmaf(accx[i][r], SPLIT_INV, acc[i][r]) : acc[i][r]) + b;
    }
  }
  __syncthreads();

  // ---- residual + LayerNorm, one wave per row
  f32x4 gw[L::VPL], gb[L::VPL];
#pragma unroll
  for (int k = 0; k < L::VPL; ++k) {
    const int v = min(lane + 64 * k, C / 4 - 1);
    gw[k] = *reinterpret_cast<const f32x4*>(p.ln_w + 4 * v);
    gb[k] = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * v);
  }
#pragma unroll
  for (int i = 0; i < L::RPW; ++i) {
    const int row = wave + i * L::NW;
    if (row >= 32) break;                                     // wave-uniform
    const size_t m = (size_t)(m0 + row);
    f32x4 x[L::VPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < L::VPL; ++k) {
      const int v = lane + 64 * k;
      if (v < C / 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(sX + row * L::XLD + 4 * v);
        x[k] = f32x4{t[0] + rv[i][k][0], t[1] + rv[i][k][1], t[2] + rv[i][k][2], t[3] + rv[i][k][3]};
        *reinterpret_cast<f32x4*>(p.out32 + m * C + 4 * v) = x[k];
        s += (x[k][0] + x[k][1]) + (x[k][2] + x[k][3]);
      }
    }
    const float mean = wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < L::VPL; ++k)
      if (lane + 64 * k < C / 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = x[k][e] - mean;
          q += d * d;
        }
      }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / C) + p.eps);
#pragma unroll
    for (int k = 0; k < L::VPL; ++k) {
      const int v = lane + 64 * k;
      if (v < C / 4) {
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)((x[k][e] - mean) * rstd * gw[k][e] + gb[k][e]);
        *reinterpret_cast<half4*>(p.out16 + m * p.ld16 + 4 * v) = h;
      }
    }
  }
}
}  // namespace

bool gemm_projln_supported(int C) { return C == 144 || C == 288 || C == 576; }

hipError_t gemm_projln_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_projln_kernel<144>), hipFuncAttributeMaxDynamicSharedMemorySize, PL<144>::LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_projln_kernel<288>), hipFuncAttributeMaxDynamicSharedMemorySize, PL<288>::LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_projln_kernel<576>), hipFuncAttributeMaxDynamicSharedMemorySize, PL<576>::LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_projln_kernel<144, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PL<144>::LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_projln_kernel<144, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, PL<144>::LDS_B2);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_projln_kernel<288, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PL<288>::LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_projln_kernel<288, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, PL<288>::LDS_B2);
  return e;
}

hipError_t gemm_projln_launch(const ProjLnParams& p, hipStream_t s) {
  if (!gemm_projln_supported(p.C) || p.M <= 0 || (p.M & 31) || !p.a16 || (p.lda & 7) || !p.wpack || !p.res || !p.out32 || !p.ln_w || !p.ln_b ||
      !p.out16 || (p.ld16 & 3))
    return hipErrorInvalidValue;
  const dim3 grid(p.M / 32);
  if (p.wpack_lo) {                    // split modes: stages 1-2 only
    const bool as = p.a_lo_off != 0;
    if (p.C == 144) {
      if (as) gemm_projln_kernel<144, 3><<<grid, dim3(PL<144>::NW * 64), PL<144>::LDS_B2, s>>>(p);
      else gemm_projln_kernel<144, 2><<<grid, dim3(PL<144>::NW * 64), PL<144>::LDS_B, s>>>(p);
    } else if (p.C == 288) {
      if (as) gemm_projln_kernel<288, 3><<<grid, dim3(PL<288>::NW * 64), PL<288>::LDS_B2, s>>>(p);
      else gemm_projln_kernel<288, 2><<<grid, dim3(PL<288>::NW * 64), PL<288>::LDS_B, s>>>(p);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  switch (p.C) {
    case 144: gemm_projln_kernel<144><<<grid, dim3(PL<144>::NW * 64), PL<144>::LDS_B, s>>>(p); break;
    case 288: gemm_projln_kernel<288><<<grid, dim3(PL<288>::NW * 64), PL<288>::LDS_B, s>>>(p); break;
    default: gemm_projln_kernel<576><<<grid, dim3(PL<576>::NW * 64), PL<576>::LDS_B, s>>>(p); break;
  }
  return hipGetLastError();
}

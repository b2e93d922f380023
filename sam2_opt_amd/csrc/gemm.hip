// Dispatcher of the MFMA GEMMs (production kernel: gemm2.hip) + the first-session kernel, kept for A/B runs
// (-DSAM2MI_EXPERIMENTAL, SAM2MI_GEMM_V1=1):
// MFMA f16 GEMM for gfx950.  256 threads = 4 waves in a 2x2 grid, each wave owns a
// (BM/2)x(BN/2) sub-tile built from 32x32x16 MFMAs.  Global->register->LDS staging with
// the loads of tile k+1 issued before the MFMAs of tile k (one barrier per K-tile); LDS rows
// are padded by 8 halfs so that the ds_read_b128 fragment reads are bank-conflict free
// (row strides 80/112/144 B map the 16-lane read groups onto 16 distinct 16-B slots).
#include <cstdlib>

#include "gemm.h"

#ifdef SAM2MI_EXPERIMENTAL
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void gemm_f16_kernel(const GemmParams p) {
  constexpr int BKP = BK + 8;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KS = BK / 16;
  constexpr int CPR = BK / 8;
  constexpr int A_CH = BM * CPR, B_CH = BN * CPR;
  constexpr int A_IT = (A_CH + 255) / 256, B_IT = (B_CH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* sA = reinterpret_cast<half_t*>(smem);
  half_t* sB = sA + 2 * BM * BKP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  half8 ra[A_IT], rb[B_IT];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int c = tid + i * 256;
      if (A_CH % 256 == 0 || c < A_CH) {
        const int row = c / CPR, kc = c % CPR;
        const int gr = min(m0 + row, p.M - 1);
        ra[i] = *reinterpret_cast<const half8*>(p.A + (size_t)gr * p.lda + kt * BK + kc * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int c = tid + i * 256;
      if (B_CH % 256 == 0 || c < B_CH) {
        const int row = c / CPR, kc = c % CPR;
        const int gr = min(n0 + row, p.N - 1);
        rb[i] = *reinterpret_cast<const half8*>(p.W + (size_t)gr * p.ldw + kt * BK + kc * 8);
      }
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int c = tid + i * 256;
      if (A_CH % 256 == 0 || c < A_CH) {
        const int row = c / CPR, kc = c % CPR;
        *reinterpret_cast<half8*>(sA + (buf * BM + row) * BKP + kc * 8) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int c = tid + i * 256;
      if (B_CH % 256 == 0 || c < B_CH) {
        const int row = c / CPR, kc = c % CPR;
        *reinterpret_cast<half8*>(sB + (buf * BN + row) * BKP + kc * 8) = rb[i];
      }
    }
  };

  const int nk = p.K / BK;
  gload(0);
  swrite(0);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const half_t* a_base = sA + (cur * BM + wm * WTM + fr) * BKP + fh * 8;
    const half_t* b_base = sB + (cur * BN + wn * WTN + fr) * BKP + fh * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      half8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8*>(a_base + i * 32 * BKP + s * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8*>(b_base + j * 32 * BKP + s * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
    }
    if (kt + 1 < nk) swrite(cur ^ 1);
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  const bool has_rope = p.rope_cols > 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WTN + j * 32 + fr;
    const bool n_ok = n < p.N;
    const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
    const float cscale = (p.col_scale && n_ok) ? p.col_scale[n] : 1.f;
    const bool transposed = (n0 + wn * WTN + j * 32) >= p.n_split;   // wave-uniform (n_split % 32 == 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + wm * WTM + i * 32 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mb + 8 * g + r;
          float x = acc[i][j][4 * g + r] + bias;
          if (has_rope) {
            const float partner = __shfl_xor(x, 1, 64);
            if (n < p.rope_cols && m < p.rope_rows) {
              const int pr = (n % p.rope_dim) >> 1;
              const size_t ti = (size_t)(m % p.rope_len) * (p.rope_dim >> 1) + pr;
              const float c = p.rope_cos[ti], sn = p.rope_sin[ti];
              x = (n & 1) ? (partner * sn + x * c) : (x * c - partner * sn);
            }
          }
          if (p.act == ACT_GELU) x = gelu_erf(x);
          else if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
          else if (p.act == ACT_SIGMOID) x = 1.f / (1.f + __expf(-x));
          x *= cscale;
          if (p.res && n_ok && m < p.M) {
            const int rm = p.res_mod ? (m % p.res_mod) : m;
            x += p.res[(size_t)rm * p.ldres + n];
          }
          v[r] = x;
        }
        const int mg = mb + 8 * g;
        if (!transposed) {
          if (n_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int m = mg + r;
              if (m < p.M) {
                if (p.out32) p.out32[(size_t)m * p.ld32 + n] = v[r];
                if (p.out16) p.out16[(size_t)m * p.ld16 + n] = (half_t)v[r];
              }
            }
          }
        } else if (n_ok) {
          const int nt = n - p.n_split;
          if (mg + 3 < p.M) {
            if (p.outT16) {
              half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
              *reinterpret_cast<half4*>(p.outT16 + (size_t)nt * p.ldT16 + mg) = h;
            }
            if (p.outT32) {
              f32x4 f = {v[0], v[1], v[2], v[3]};
              *reinterpret_cast<f32x4*>(p.outT32 + (size_t)nt * p.ldT32 + mg) = f;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (mg + r < p.M) {
                if (p.outT16) p.outT16[(size_t)nt * p.ldT16 + mg + r] = (half_t)v[r];
                if (p.outT32) p.outT32[(size_t)nt * p.ldT32 + mg + r] = v[r];
              }
            }
          }
        }
      }
    }
  }
}

template <int BM, int BN, int BK>
static constexpr size_t gemm_smem() { return (size_t)2 * (BM + BN) * (BK + 8) * sizeof(half_t); }

template <int BM, int BN, int BK>
static hipError_t gemm_launch_t(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t smem = gemm_smem<BM, BN, BK>();
  gemm_f16_kernel<BM, BN, BK><<<dim3(tiles), dim3(256), smem, s>>>(p);
  return hipGetLastError();
}

template <int BM, int BN, int BK>
static hipError_t gemm_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<BM, BN, BK>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_smem<BM, BN, BK>());
}

#endif  // SAM2MI_EXPERIMENTAL

hipError_t gemm_init() {
#ifdef SAM2MI_EXPERIMENTAL
  hipError_t e[9] = {
      gemm_attr<128, 128, 64>(), gemm_attr<128, 128, 48>(), gemm_attr<128, 128, 32>(),
      gemm_attr<128, 64, 64>(),  gemm_attr<128, 64, 48>(),  gemm_attr<128, 64, 32>(),
      gemm_attr<64, 64, 64>(),   gemm_attr<64, 64, 48>(),   gemm_attr<64, 64, 32>()};
  for (int i = 0; i < 9; ++i)
    if (e[i] != hipSuccess) return e[i];
  hipError_t e3 = gemm_v3_init();
  if (e3 == hipSuccess) e3 = gemm_p4_init();
  if (e3 != hipSuccess) return e3;
#endif
  return gemm_v2_init();
}

#ifdef SAM2MI_EXPERIMENTAL
template <int BK>
static hipError_t gemm_dispatch_tile(const GemmParams& p, hipStream_t s) {
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const long t12864 = (long)((p.M + 127) / 128) * ((p.N + 63) / 64);
  if (t128 >= 384) return gemm_launch_t<128, 128, BK>(p, s);
  if (t12864 >= 256) return gemm_launch_t<128, 64, BK>(p, s);
  return gemm_launch_t<64, 64, BK>(p, s);
}
#endif

// Every linear of the model has K % 16 == 0 and runs on the LDS-DMA kernel (gemm2.hip).
hipError_t gemm_launch(const GemmParams& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if ((p.lda & 7) || (p.ldw & 7) || (p.n_split < p.N && (p.n_split & 31)) || p.K <= 0) return hipErrorInvalidValue;
#ifdef SAM2MI_EXPERIMENTAL
  static const bool use_v1 = getenv("SAM2MI_GEMM_V1") != nullptr;      // A/B switch: the register-staged first-session kernel
  if (use_v1 || (p.K & 15)) {
    if (p.a_lo_off) return hipErrorInvalidValue;
    if (p.K % 64 == 0) return gemm_dispatch_tile<64>(p, s);
    if (p.K % 48 == 0) return gemm_dispatch_tile<48>(p, s);
    if (p.K % 32 == 0) return gemm_dispatch_tile<32>(p, s);
    return hipErrorInvalidValue;
  }
#endif
  if (p.K & 15) return hipErrorInvalidValue;
  return gemm_v2_launch(p, s);
}

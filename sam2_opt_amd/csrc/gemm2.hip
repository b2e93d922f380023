// MFMA f16 GEMM v2 for gfx950: direct-to-LDS staging (global_load_lds_dwordx4), XOR-swizzled LDS,
// double-buffered K pipeline with one barrier per 64-deep K tile, and a coalescing epilogue.
//
//   * Staging: each wave issues 1-KiB LDS-DMA pieces (64 lanes x 16 B).  The LDS image is linear
//     [row][64 halfs] (128-B rows); the bank-conflict swizzle lives on the per-lane SOURCE address
//     (LDS-DMA destinations are lane-linear): LDS 16-B slot `pc` of row r holds logical chunk
//     pc ^ ((r >> 1) & 7).  The ds_read_b128 fragment reads apply the same XOR, which spreads every
//     16-lane read group over all 16 slots of the 256-B bank row (conflict-free).
//   * Pipeline: tile t+1 is in flight (LDS-DMA) while tile t is consumed from LDS; the compiler's
//     vmcnt(0) in front of __syncthreads() retires it ("2-phase minimum" structure of the CDNA guide).
//   * K may be any multiple of 16: the last K tile runs fewer 16-deep MFMA steps.
//   * Epilogue: each 32x32 accumulator tile goes through a wave-private 4-KiB LDS patch so that residual
//     loads and f32/f16 stores are 16-B / 8-B per lane over whole 128-B row segments (the accumulator layout
//     itself has one column per lane); transposed outputs are stored straight from the accumulators
//     (4 consecutive rows per lane).
#include "gemm.h"

#ifndef KUNROLL
#define KUNROLL 4
#endif

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_v2_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_CALLS = BM / 8 / NW, B_CALLS = BN / 8 / NW;     // 1-KiB pieces per wave
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 31, fh = lane >> 5;
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

  const int nk = (p.K + 63) / 64;
  const int srow = lane >> 3, spc = lane & 7;        // row-in-piece and physical 16-B slot of this lane

  // per-lane source offsets of every LDS-DMA piece (row clamp + swizzled chunk), hoisted out of the K loop
  const half_t* a_src[A_CALLS];
  const half_t* b_src[B_CALLS];
  int a_kc[A_CALLS], b_kc[B_CALLS];
#pragma unroll
  for (int j = 0; j < A_CALLS; ++j) {
    const int row = (wave * A_CALLS + j) * 8 + srow;
    a_kc[j] = (spc ^ ((row >> 1) & 7)) << 3;
    a_src[j] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + a_kc[j];
  }
#pragma unroll
  for (int j = 0; j < B_CALLS; ++j) {
    const int row = (wave * B_CALLS + j) * 8 + srow;
    b_kc[j] = (spc ^ ((row >> 1) & 7)) << 3;
    b_src[j] = p.W + (size_t)min(n0 + row, p.N - 1) * p.ldw + b_kc[j];
  }
  const bool k_tail = (p.K & 63) != 0;
  auto issue = [&](int stage, int kt) {
    char* sbase = smem + stage * STAGE;
    const bool last = k_tail && kt == nk - 1;
#pragma unroll
    for (int j = 0; j < A_CALLS; ++j) {
      const half_t* g = a_src[j] + kt * 64;
      if (last && kt * 64 + a_kc[j] >= p.K) g -= a_kc[j] + kt * 64;      // K tail: slot never consumed, read col 0 instead
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(sbase + (wave * A_CALLS + j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_CALLS; ++j) {
      const half_t* g = b_src[j] + kt * 64;
      if (last && kt * 64 + b_kc[j] >= p.K) g -= b_kc[j] + kt * 64;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(sbase + A_BYTES + (wave * B_CALLS + j) * 1024), 16, 0, 0);
    }
  };

  issue(0, 0);
  __syncthreads();
#pragma nounroll
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(cur ^ 1, kt + 1);
    const char* sA = smem + cur * STAGE;
    const char* sB = sA + A_BYTES;
    const int ksteps = (kt == nk - 1 && (p.K & 63)) ? ((p.K & 63) >> 4) : 4;
#pragma unroll KUNROLL
    for (int s = 0; s < ksteps; ++s) {
      {
        half8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * WTM + i * 32 + fr;
          af[i] = *reinterpret_cast<const half8*>(sA + row * 128 + (((2 * s + fh) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wn * WTN + j * 32 + fr;
          bf[j] = *reinterpret_cast<const half8*>(sB + row * 128 + (((2 * s + fh) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
      }
    }
    __syncthreads();          // retires the in-flight tile (vmcnt(0)) and frees `cur` for the next issue
  }

  // ------------------------------------------------------------------ epilogue
  float* patch = reinterpret_cast<float*>(smem) + wave * 1024;     // wave-private 32x32 f32
  const bool has_rope = p.rope_cols > 0;
  const bool vec_ok = (p.N & 3) == 0 && (p.ld32 & 3) == 0 && (p.ld16 & 3) == 0 && (p.ldres & 3) == 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nt0 = n0 + wn * WTN + j * 32;
    const int n = nt0 + fr;
    const bool n_ok = n < p.N;
    const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
    const float cscale = (p.col_scale && n_ok) ? p.col_scale[n] : 1.f;
    const bool transposed = nt0 >= p.n_split;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mt0 = m0 + wm * WTM + i * 32;
      if (mt0 >= p.M || nt0 >= p.N) continue;                       // wave-uniform
      // phase 1: bias / RoPE / activation / column scale on the accumulator layout
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt0 + acc_row(r, lane);
        float x = acc[i][j][r] + bias;
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
        v[r] = x * cscale;
      }
      if (transposed) {
        const int nt = n - p.n_split;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int mg = mt0 + 8 * g + 4 * fh;
          float w4[4] = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
          if (n_ok) {
            if (p.res) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (mg + r < p.M) w4[r] += p.res[(size_t)(p.res_mod ? (mg + r) % p.res_mod : (mg + r)) * p.ldres + n];
            }
            if (mg + 3 < p.M) {
              if (p.outT16) {
                const half4 h = {(half_t)w4[0], (half_t)w4[1], (half_t)w4[2], (half_t)w4[3]};
                *reinterpret_cast<half4*>(p.outT16 + (size_t)nt * p.ldT16 + mg) = h;
              }
              if (p.outT32) {
                const f32x4 f = {w4[0], w4[1], w4[2], w4[3]};
                *reinterpret_cast<f32x4*>(p.outT32 + (size_t)nt * p.ldT32 + mg) = f;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (mg + r < p.M) {
                  if (p.outT16) p.outT16[(size_t)nt * p.ldT16 + mg + r] = (half_t)w4[r];
                  if (p.outT32) p.outT32[(size_t)nt * p.ldT32 + mg + r] = w4[r];
                }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      // phase 2: through the wave-private LDS patch -> row-major 16-B accesses
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[acc_row(r, lane) * 32 + fr] = v[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int c4 = (lane & 7) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = (lane >> 3) + 8 * q;
        const f32x4 t = *reinterpret_cast<const f32x4*>(patch + rr * 32 + c4);
        const int m = mt0 + rr, nn = nt0 + c4;
        if (m < p.M && nn < p.N) {
          float o[4] = {t[0], t[1], t[2], t[3]};
          const size_t rrow = (size_t)(p.res_mod ? m % p.res_mod : m);
          if (vec_ok) {                                             // N % 4 == 0: the 4 columns are all valid
            if (p.res) {
              const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + rrow * p.ldres + nn);
              o[0] += rv[0]; o[1] += rv[1]; o[2] += rv[2]; o[3] += rv[3];
            }
            if (p.out32) {
              const f32x4 ov = {o[0], o[1], o[2], o[3]};
              *reinterpret_cast<f32x4*>(p.out32 + (size_t)m * p.ld32 + nn) = ov;
            }
            if (p.out16) {
              const half4 hv = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
              *reinterpret_cast<half4*>(p.out16 + (size_t)m * p.ld16 + nn) = hv;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (nn + e < p.N) {
                float x = o[e];
                if (p.res) x += p.res[rrow * p.ldres + nn + e];
                if (p.out32) p.out32[(size_t)m * p.ld32 + nn + e] = x;
                if (p.out16) p.out16[(size_t)m * p.ld16 + nn + e] = (half_t)x;
              }
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // patch is rewritten by the next tile
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int BM, int BN>
constexpr size_t v2_smem() { return (size_t)2 * (BM + BN) * 128; }

template <int BM, int BN, int WM, int WN>
hipError_t v2_launch(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t smem = v2_smem<BM, BN>();
  gemm_v2_kernel<BM, BN, WM, WN><<<dim3(tiles), dim3(WM * WN * 64), smem, s>>>(p);
  return hipGetLastError();
}
template <int BM, int BN, int WM, int WN>
hipError_t v2_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v2_kernel<BM, BN, WM, WN>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)v2_smem<BM, BN>());
}
}  // namespace

hipError_t gemm_v2_init() {
  hipError_t e[4] = {v2_attr<256, 128, 4, 2>(), v2_attr<128, 128, 2, 2>(), v2_attr<128, 64, 2, 2>(), v2_attr<64, 64, 2, 2>()};
  for (int i = 0; i < 4; ++i)
    if (e[i] != hipSuccess) return e[i];
  return hipSuccess;
}

static inline long tiles_of(const GemmParams& p, int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); }

// Tile choice (measured on MI355X, tools/gemm_bench.py): with K <= 4608 every output tile is a short K loop whose
// prologue (first LDS-DMA round trip) and epilogue are not overlapped inside a workgroup, so small tiles at high
// occupancy (3+ workgroups per CU hiding each other's prologue/epilogue) beat the 256x128 / 128x128 tiles here.
hipError_t gemm_v2_launch(const GemmParams& p, hipStream_t s) {
  const int force = p.tile_hint;
  if (force == 2) return v2_launch<256, 128, 4, 2>(p, s);
  if (force == 3) return v2_launch<128, 128, 2, 2>(p, s);
  if (force == 4 || (force == 0 && tiles_of(p, 128, 64) >= 2048)) return v2_launch<128, 64, 2, 2>(p, s);
  return v2_launch<64, 64, 2, 2>(p, s);
}

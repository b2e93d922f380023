// MFMA f16 GEMM, 256x256 tile, one workgroup per CU, staggered 4-phase pipeline (EXPERIMENTAL: tile_hint 20).
//
// Why: the production kernel (gemm2.hip) moves (BM + BN) * 128 B per 64-deep K tile through the CU's vector-memory path
// for BM * BN * 128 FLOP; at 128x128 that path (25-29 B/clk measured), not the MFMA pipe, bounds the loop.  A 256x256 tile
// halves the bytes per FLOP, but only pays off if the workgroup - alone on its CU - never stops the MFMA pipe to load.
//
// Structure (K % 64 == 0):
//   * 8 waves = 2 (M) x 4 (N), 128x64 of the tile per wave = 8 accumulator tiles (128 VGPRs).
//   * LDS 128 KiB = 2 K-tile stages x 4 half-tiles of 16 KiB, cut by WHEN a wave reads them:
//       A-r0 / A-r1 = the first / second 64 rows of every wave's 128 rows, B-c0 / B-c1 = the first / second 32 columns of
//       every wave's 64 columns.  A K tile runs 4 phases (r0,c0) (r0,c1) (r1,c1) (r1,c0); A fragments are kept over two
//       phases, B-c0 fragments over all four, so each half-tile is read in ONE phase and is free for the K tile after next
//       right after it.
//   * phase = LOAD [ds_read the operand half this phase introduces] barrier, MFMA [8 x 32x32x16 under s_setprio, one
//     half-tile of LDS-DMA (2 pieces per wave) issued between the 4th and 5th] barrier.
//   * the two wave rows run ONE BARRIER APART (row 1 takes an extra barrier up front, row 0 at the end): a SIMD hosts one
//     wave of each row, so while one loads the other multiplies.
//   * LDS-DMA is never drained in the loop: every LOAD starts with s_waitcnt vmcnt(8) (4 half-tiles stay in flight), placed
//     one phase before the first reader of the data because the rows are staggered (the partner row's pieces are only
//     known to have landed after ITS wait and a barrier both rows pass).
// DMA order per K tile j:  phase 0: A-r1(j+1)   phase 1: A-r0(j+2)   phase 2: B-c0(j+2)   phase 3: B-c1(j+2)
// (each into the buffer whose last reader finished at least two barriers earlier; tiles past the end re-load the last tile).
#include "gemm.h"

#include <type_traits>

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int P4_HALF = 128 * 128;          // bytes of one half-tile: 128 rows x 64 halfs
constexpr int P4_STAGE = 4 * P4_HALF;       // A-r0, A-r1, B-c0, B-c1
constexpr int P4_LDS = 2 * P4_STAGE;        // 131072

template <int N>
__device__ __forceinline__ void p4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void p4_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(512, 2) void gemm_p4_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;                 // wave row (0..1), wave column (0..3); waves w and w+4 share a SIMD
  const int fr = lane & 31, fh = lane >> 5;
  const int tiles_n = (p.N + 255) / 256;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * 256, n0 = (bid % tiles_n) * 256;
  const int nk = p.K / 64;

  // ---- LDS-DMA sources: this wave moves pieces 2w and 2w+1 (LDS rows 16w .. 16w+15) of every half-tile
  unsigned off[4][2];                                     // [A-r0, A-r1, B-c0, B-c1][piece]: byte offset from the tile origin
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int R = 16 * wave + 8 * q + (lane >> 3);        // LDS row of this lane inside a half-tile
    const int c = ((lane & 7) ^ ((R >> 1) & 7)) << 3;     // logical 8-half chunk held by this lane's physical slot
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int am = min(m0 + (R >> 6) * 128 + h * 64 + (R & 63), p.M - 1);
      const int bn = min(n0 + (R >> 5) * 64 + h * 32 + (R & 31), p.N - 1);
      off[h][q] = (unsigned)((am - m0) * p.lda + c) * 2u;
      off[2 + h][q] = (unsigned)((bn - n0) * p.ldw + c) * 2u;
    }
  }
  const char* baseA = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
  const char* baseB = reinterpret_cast<const char*>(p.W + (size_t)n0 * p.ldw);
  // half-tile `which` (0 A-r0, 1 A-r1, 2 B-c0, 3 B-c1) of K tile min(kt, nk-1) into stage kt & 1
  auto issue = [&](int which, int kt) {
    const int t = min(kt, nk - 1);
    const char* src = (which < 2 ? baseA : baseB) + (size_t)t * 128;
    char* dst = smem + (kt & 1) * P4_STAGE + which * P4_HALF + (2 * wave) * 1024;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off[which][0]), (lds_ptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off[which][1]), (lds_ptr_t)(dst + 1024), 16, 0, 0);
  };

  // ---- fragment reads: A rows wr*64 + i*32 + fr of A-r{h}; B rows wc*32 + fr of B-c{h}
  const int ra = wr * 64 + fr, rb = wc * 32 + fr;
  const int a_sw = (ra >> 1) & 7, a_sw1 = ((ra + 32) >> 1) & 7, b_sw = (rb >> 1) & 7;
  half8 Af[2][4], Bc0[4], Bc1[4];
  auto read_a = [&](int stage, int h) {
    const char* s = smem + stage * P4_STAGE + h * P4_HALF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Af[0][ks] = *reinterpret_cast<const half8*>(s + ra * 128 + (((2 * ks + fh) ^ a_sw) << 4));
      Af[1][ks] = *reinterpret_cast<const half8*>(s + (ra + 32) * 128 + (((2 * ks + fh) ^ a_sw1) << 4));
    }
  };
  auto read_b = [&](int stage, int h, half8* B) {
    const char* s = smem + stage * P4_STAGE + (2 + h) * P4_HALF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) B[ks] = *reinterpret_cast<const half8*>(s + rb * 128 + (((2 * ks + fh) ^ b_sw) << 4));
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: K tiles 0 and 1 completely
#pragma unroll
  for (int w = 0; w < 4; ++w) issue(w, 0);
#pragma unroll
  for (int w = 0; w < 4; ++w) issue(w, 1);
  p4_wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();              // stagger: wave row 1 runs one barrier behind row 0

// the compiler is free to move MFMAs (register-only) across s_barrier: pin every barrier so the phases stay phases
#define P4_BAR()                         \
  __builtin_amdgcn_sched_barrier(0);     \
  __builtin_amdgcn_s_barrier();          \
  __builtin_amdgcn_sched_barrier(0);
// 8 MFMAs of one phase with this phase's LDS-DMA half-tile issued in their middle: a DMA piece costs the issuing wave
// 60+ cycles - inside the LOAD section (which must fit under the partner row's 256-cycle MFMA section) that is too much,
// between MFMAs of this wave it only opens a short bubble
#define P4_MMA(I0, J, BF, WHICH, KT)                                                                \
  __builtin_amdgcn_s_setprio(1);                                                                    \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                \
    acc[I0][J] = mfma32(Af[0][ks], BF[ks], acc[I0][J]);                                             \
    acc[I0 + 1][J] = mfma32(Af[1][ks], BF[ks], acc[I0 + 1][J]);                                     \
  }                                                                                                 \
  asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]));                                        \
  issue(WHICH, KT);                                                                                 \
  _Pragma("unroll") for (int ks = 2; ks < 4; ++ks) {                                                \
    acc[I0][J] = mfma32(Af[0][ks], BF[ks], acc[I0][J]);                                             \
    acc[I0 + 1][J] = mfma32(Af[1][ks], BF[ks], acc[I0 + 1][J]);                                     \
  }                                                                                                 \
  /* MFMAs are pure: without a use HERE the optimiser sinks them into later phases */              \
  asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]));                                        \
  __builtin_amdgcn_s_setprio(0);

#pragma nounroll
  for (int j = 0; j < nk; ++j) {
    const int st = j & 1;
    // ---- phase 0: (r0, c0)
    p4_wait_vm<8>();
    read_a(st, 0);
    read_b(st, 0, Bc0);
    P4_BAR()
    P4_MMA(0, 0, Bc0, 1, j + 1)
    P4_BAR()
    // ---- phase 1: (r0, c1)
    p4_wait_vm<8>();
    read_b(st, 1, Bc1);
    P4_BAR()
    P4_MMA(0, 1, Bc1, 0, j + 2)
    P4_BAR()
    // ---- phase 2: (r1, c1)
    p4_wait_vm<8>();
    read_a(st, 1);
    P4_BAR()
    P4_MMA(2, 1, Bc1, 2, j + 2)
    P4_BAR()
    // ---- phase 3: (r1, c0), B-c0 fragments still in registers
    p4_wait_vm<8>();
    P4_BAR()
    P4_MMA(2, 0, Bc0, 3, j + 2)
    P4_BAR()
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();              // row 0 catches up with row 1's extra barrier
  p4_wait_vm<0>();                                         // the trailing (duplicate) DMA pieces must land before LDS is reused
  __syncthreads();

  // ------------------------------------------------------------------ epilogue (same fusions as gemm2.hip)
  float* patch = reinterpret_cast<float*>(smem) + wave * 1024;     // wave-private 32x32 f32
  const bool has_rope = p.rope_cols > 0;
  const bool vec_ok = (p.N & 3) == 0 && (p.ld32 & 3) == 0 && (p.ld16 & 3) == 0 && (p.ldres & 3) == 0;
  auto tile_epilogue = [&](auto tile_c) {
    constexpr int TILE = decltype(tile_c)::value;
    constexpr int i = TILE >> 1, j = TILE & 1;             // acc[i][j]: rows wr*128 + i*32, columns wc*64 + j*32
    const int mt0 = m0 + wr * 128 + i * 32, nt0 = n0 + wc * 64 + j * 32;
    if (mt0 >= p.M || nt0 >= p.N) return;                  // wave-uniform
    const int n = nt0 + fr;
    const bool n_ok = n < p.N;
    const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
    const float cscale = (p.col_scale && n_ok) ? p.col_scale[n] : 1.f;
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
    if (nt0 >= p.n_split) {                                // transposed store straight from the accumulator layout
      const int nt = n - p.n_split;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mg = mt0 + 8 * g + 4 * fh;
        float w4[4] = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        if (!n_ok) continue;
        if (p.res) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (mg + r < p.M) w4[r] += p.res[(size_t)(p.res_mod ? (mg + r) % p.res_mod : (mg + r)) * p.ldres + n];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mg + r < p.M) {
            if (p.outT16) p.outT16[(size_t)nt * p.ldT16 + mg + r] = (half_t)w4[r];
            if (p.outT32) p.outT32[(size_t)nt * p.ldT32 + mg + r] = w4[r];
          }
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[acc_row(r, lane) * 32 + fr] = v[r];
    p4_wait_lds();
    const int c4 = (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = (lane >> 3) + 8 * q;
      const f32x4 t = *reinterpret_cast<const f32x4*>(patch + rr * 32 + c4);
      const int m = mt0 + rr, nn = nt0 + c4;
      if (m < p.M && nn < p.N) {
        float o[4] = {t[0], t[1], t[2], t[3]};
        const size_t rrow = (size_t)(p.res_mod ? m % p.res_mod : m);
        if (vec_ok) {
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
    p4_wait_lds();                                         // the patch is rewritten by the next tile
    __builtin_amdgcn_sched_barrier(0);
  };
  tile_epilogue(std::integral_constant<int, 0>{});
  tile_epilogue(std::integral_constant<int, 1>{});
  tile_epilogue(std::integral_constant<int, 2>{});
  tile_epilogue(std::integral_constant<int, 3>{});
  tile_epilogue(std::integral_constant<int, 4>{});
  tile_epilogue(std::integral_constant<int, 5>{});
  tile_epilogue(std::integral_constant<int, 6>{});
  tile_epilogue(std::integral_constant<int, 7>{});
}
}  // namespace

hipError_t gemm_p4_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS);
}

hipError_t gemm_p4_launch(const GemmParams& p, hipStream_t s) {
  if (p.K < 64 || (p.K & 63) || (p.lda & 7) || (p.ldw & 7)) return hipErrorInvalidValue;
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  gemm_p4_kernel<<<dim3(tiles), dim3(512), P4_LDS, s>>>(p);
  return hipGetLastError();
}

// MFMA f16 GEMM v2 for gfx950: direct-to-LDS staging (global_load_lds_dwordx4), XOR-swizzled LDS,
// double-buffered K pipeline with one barrier per 64-deep K tile, and a coalescing epilogue.
//
//   * Staging: each wave issues 1-KiB LDS-DMA pieces (64 lanes x 16 B).  The LDS image is linear
//     [row][64 halfs] (128-B rows); the bank-conflict swizzle lives on the per-lane SOURCE address
//     (LDS-DMA destinations are lane-linear): LDS 16-B slot `pc` of row r holds logical chunk
//     pc ^ ((r >> 1) & 7).  The ds_read_b128 fragment reads apply the same XOR, which spreads every
//     16-lane read group over all 16 slots of the 256-B bank row (conflict-free).
//   * Pipeline: tile t+1 is in flight (LDS-DMA, wave-uniform scalar base + one 32-bit per-lane offset per piece) while
//     tile t is consumed from LDS; a counted vmcnt + raw s_barrier per K tile (STAGES - 1 tiles stay in flight).  The K loop
//     has ONE straight-line MFMA path with the fragments of k-step s+1 read while the MFMAs of step s run.
//   * K may be any multiple of 16: a K tail is peeled behind the loop (last tile loaded from columns [K-64, K)).
//   * Workgroups of 8 waves (128x128 or 128x64 tile, 2-3 per CU) on the large shapes, 4 waves (64x64) on M = 4096.
//   * Epilogue: each 32x32 accumulator tile goes through a wave-private 4-KiB LDS patch so that residual
//     loads and f32/f16 stores are 16-B / 8-B per lane over whole 128-B row segments (the accumulator layout
//     itself has one column per lane); transposed outputs are stored straight from the accumulators
//     (4 consecutive rows per lane).
#include "gemm.h"
#include <cstdlib>

#ifndef KUNROLL
#define KUNROLL 4
#endif

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// SPLIT = 1 (the "f16x3" precision mode): both operands arrive as two f16 arrays, hi = f16(v) and lo = f16((v - hi) * 2^11) at
// p.a_lo_off / p.w_lo_off elements behind the hi array.  A stage holds [A_hi | B_hi | B_lo | A_lo]; every fragment pair
// costs three MFMAs: hi*hi into the main accumulator, lo*hi + hi*lo into a second one that is folded in (x 2^-11) before the
// epilogue.  The dropped lo*lo term is 2^-22 relative; the 2^11 scale keeps the lo parts in the normal f16 range.
// SPLIT = 2 (weight split, the "f16s" selective mode): only W is split, stage = [A | B_hi | B_lo], two MFMAs per fragment pair
// (a*b_hi, a*b_lo).  The rounding of a WEIGHT is the same perturbation for every token, so it survives the averaging over
// tokens (attention, pooling) that the per-token rounding of an activation does not: on the encoder it carries 4.6x the
// variance of the activation rounding (tools/precision_shares.py) at a third less work than the full split.
// RS ("register-staged", K % 64 == 0, even piece split, 2 LDS stages): the K tiles do not arrive by LDS-DMA but through two
// register sets - global -> VGPR loads issued TWO tiles ahead, written to the LDS stage (same lane-linear piece layout, so the
// fragment reads are unchanged) one tile ahead.  Twice the bytes in flight of the 2-stage DMA ring at the same LDS footprint.
// MEASURED AND LOST (tools/gemm_bench.py, hints 41-43): 10-20 % slower than the LDS-DMA ring on every K >= 576
// shape (fc2 of stage 3: 132 us vs 111 us), equal at K = 144 - the extra VGPR -> LDS hop costs more than the deeper prefetch
// buys, i.e. the loop is not short of bytes in flight.  Only compiled with -DSAM2MI_EXPERIMENTAL.
template <int BM, int BN, int WM, int WN, int STAGES, int SPLIT = 0, bool RS = false>
__global__ __launch_bounds__(WM * WN * 64, RS ? 4 : (WM * WN > 8 || SPLIT == 1 || (SPLIT == 2 && BN > 64)) ? 1 : 2) void gemm_v2_kernel(const GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, HALF_STAGE = A_BYTES + B_BYTES;
  constexpr int STAGE = HALF_STAGE + (SPLIT ? B_BYTES : 0) + (SPLIT == 1 ? A_BYTES : 0);      // [A | B | B_lo | A_lo]
  constexpr bool ASPLIT = SPLIT == 1;
  constexpr int A_CALLS = BM / 8 / NW, B_CALLS = BN / 8 / NW;     // 1-KiB pieces per wave
  // EVEN: A and B pieces split evenly over the waves.  Otherwise (12-wave 256x192 workgroup) the PA + PB pieces of a K
  // tile are dealt round-robin, piece q = wave + j * NW (A pieces first); STAGES == 2 only (vmcnt is always drained to 0).
  constexpr bool EVEN = (BM / 8) % NW == 0 && (BN / 8) % NW == 0;
  constexpr int PA = BM / 8, PB = BN / 8, NJ = (PA + PB + NW - 1) / NW;
  static_assert(EVEN || STAGES == 2, "uneven piece split needs the 2-stage ring");
  static_assert(EVEN || !SPLIT, "split operands need the even piece split");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 31, fh = lane >> 5;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

  f32x16 acc[TM][TN];
  f32x16 accx[SPLIT ? TM : 1][SPLIT ? TN : 1];       // cross terms (scaled by 2^11)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if constexpr (SPLIT) accx[i][j][r] = 0.f;
      }

  const int nk = (p.K + 63) / 64;
  const int srow = lane >> 3, spc = lane & 7;        // row-in-piece and physical 16-B slot of this lane

  // LDS-DMA sources: wave-uniform base (tile origin + K offset, scalar registers) + a per-lane 32-bit byte offset per
  // piece (clamped row + swizzled chunk) computed once -> no vector address arithmetic inside the K loop.
  // K tail (K % 64 != 0, K >= 64): the last tile is loaded from columns [K-64, K) - always in bounds - and only its
  // last (K % 64) / 16 k-steps are consumed.  K < 64 (one partial tile): chunks past K are redirected to column 0.
  const char* baseA = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
  const char* baseB = reinterpret_cast<const char*>(p.W + (size_t)n0 * p.ldw);
  unsigned a_off[EVEN ? A_CALLS : 1], b_off[EVEN ? B_CALLS : 1], g_off[EVEN ? 1 : NJ];
  const bool short_k = p.K < 64;
  if constexpr (!EVEN) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = wave + j * NW;                                 // wave-uniform
      const bool isA = q < PA;
      const int row = (isA ? q : q - PA) * 8 + srow;
      int kc = (spc ^ ((row >> 1) & 7)) << 3;
      if (short_k && kc >= p.K) kc = 0;
      g_off[j] = isA ? (unsigned)((min(m0 + row, p.M - 1) - m0) * p.lda + kc) * 2u
                     : (unsigned)((min(n0 + row, p.N - 1) - n0) * p.ldw + kc) * 2u;
    }
  }
#pragma unroll
  for (int j = 0; j < (EVEN ? A_CALLS : 0); ++j) {
    const int row = (wave * A_CALLS + j) * 8 + srow;
    int kc = (spc ^ ((row >> 1) & 7)) << 3;
    if (short_k && kc >= p.K) kc = 0;
    a_off[j] = (unsigned)((min(m0 + row, p.M - 1) - m0) * p.lda + kc) * 2u;
  }
#pragma unroll
  for (int j = 0; j < (EVEN ? B_CALLS : 0); ++j) {
    const int row = (wave * B_CALLS + j) * 8 + srow;
    int kc = (spc ^ ((row >> 1) & 7)) << 3;
    if (short_k && kc >= p.K) kc = 0;
    b_off[j] = (unsigned)((min(n0 + row, p.N - 1) - n0) * p.ldw + kc) * 2u;
  }
  const int k_tail = short_k ? 0 : (p.K & 63);
  auto issue = [&](int stage, int kt) {
    char* sbase = smem + stage * STAGE;
    const int kcol = (k_tail && kt == nk - 1) ? p.K - 64 : kt * 64;       // scalar
    const char* ka = baseA + (size_t)kcol * 2;
    const char* kb = baseB + (size_t)kcol * 2;
    if constexpr (EVEN) {
#pragma unroll
      for (int j = 0; j < A_CALLS; ++j)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ka + a_off[j]), (lds_ptr_t)(sbase + (wave * A_CALLS + j) * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < B_CALLS; ++j)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kb + b_off[j]), (lds_ptr_t)(sbase + A_BYTES + (wave * B_CALLS + j) * 1024), 16, 0, 0);
      if constexpr (SPLIT) {
        const char* kbl = kb + p.w_lo_off * 2;
#pragma unroll
        for (int j = 0; j < B_CALLS; ++j)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kbl + b_off[j]), (lds_ptr_t)(sbase + HALF_STAGE + (wave * B_CALLS + j) * 1024), 16, 0, 0);
      }
      if constexpr (ASPLIT) {
        const char* kal = ka + p.a_lo_off * 2;
#pragma unroll
        for (int j = 0; j < A_CALLS; ++j)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kal + a_off[j]), (lds_ptr_t)(sbase + HALF_STAGE + B_BYTES + (wave * A_CALLS + j) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int q = wave + j * NW;
        if ((j + 1) * NW <= PA + PB || q < PA + PB)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)((q < PA ? ka : kb) + g_off[j]), (lds_ptr_t)(sbase + q * 1024), 16, 0, 0);
      }
    }
  };

  // ---- K pipeline: STAGES-1 tiles in flight, counted vmcnt + raw barrier (a __syncthreads() would drain the DMA queue)
  constexpr int NPW = A_CALLS + B_CALLS + (SPLIT ? B_CALLS : 0) + (ASPLIT ? A_CALLS : 0);           // LDS-DMA pieces this wave issues per K tile
  struct Frag { half8 a[TM]; half8 b[TN]; half8 al[ASPLIT ? TM : 1]; half8 bl[SPLIT ? TN : 1]; };
  auto ldfrag = [&](const char* sA, const char* sB, int s) {
    Frag f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * WTM + i * 32 + fr;
      const int off = row * 128 + (((2 * s + fh) ^ ((row >> 1) & 7)) << 4);
      f.a[i] = *reinterpret_cast<const half8*>(sA + off);
      if constexpr (ASPLIT) f.al[i] = *reinterpret_cast<const half8*>(sA + HALF_STAGE + B_BYTES + off);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * WTN + j * 32 + fr;
      const int off = row * 128 + (((2 * s + fh) ^ ((row >> 1) & 7)) << 4);
      f.b[j] = *reinterpret_cast<const half8*>(sB + off);
      if constexpr (SPLIT) f.bl[j] = *reinterpret_cast<const half8*>(sB + B_BYTES + off);
    }
    return f;
  };
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = mfma32(f.a[i], f.b[j], acc[i][j]);
        if constexpr (ASPLIT) accx[i][j] = mfma32(f.al[i], f.b[j], accx[i][j]);
        if constexpr (SPLIT) accx[i][j] = mfma32(f.a[i], f.bl[j], accx[i][j]);
      }
  };
  auto sync_tile = [&](int kt) {                   // tile kt landed & visible; slot of tile kt-1 free; keep the ring full
    const int later = min(STAGES - 2, nk - 1 - kt);
    if (STAGES >= 4 && later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
    else if (STAGES >= 3 && later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + STAGES - 1 < nk) issue((kt + STAGES - 1) % STAGES, kt + STAGES - 1);
  };
  if constexpr (RS) {
    static_assert(EVEN && !SPLIT && STAGES == 2, "register staging: even piece split, plain operands, 2 LDS stages");
    constexpr int NP = A_CALLS + B_CALLS;
    f32x4 ra[NP], rb[NP];                                    // two register sets, used alternately (never indexed dynamically)
    auto rs_load = [&](f32x4 (&r)[NP], int kt) {
      const char* ka = baseA + (size_t)kt * 128;             // 64 halfs per K tile
      const char* kb = baseB + (size_t)kt * 128;
#pragma unroll
      for (int j = 0; j < A_CALLS; ++j) r[j] = *reinterpret_cast<const f32x4*>(ka + a_off[j]);
#pragma unroll
      for (int j = 0; j < B_CALLS; ++j) r[A_CALLS + j] = *reinterpret_cast<const f32x4*>(kb + b_off[j]);
    };
    auto rs_commit = [&](const f32x4 (&r)[NP], int stage) {
      char* sbase = smem + stage * STAGE + lane * 16;
#pragma unroll
      for (int j = 0; j < A_CALLS; ++j) *reinterpret_cast<f32x4*>(sbase + (wave * A_CALLS + j) * 1024) = r[j];
#pragma unroll
      for (int j = 0; j < B_CALLS; ++j) *reinterpret_cast<f32x4*>(sbase + A_BYTES + (wave * B_CALLS + j) * 1024) = r[A_CALLS + j];
    };
    auto compute = [&](int stage) {
      const char* sA = smem + stage * STAGE;
      const char* sB = sA + A_BYTES;
      Frag f0 = ldfrag(sA, sB, 0);
      Frag f1 = ldfrag(sA, sB, 1);
      mma(f0);
      f0 = ldfrag(sA, sB, 2);
      mma(f1);
      f1 = ldfrag(sA, sB, 3);
      mma(f0);
      mma(f1);
    };
    auto lds_barrier = [&]() {                               // LDS writes visible to the workgroup; the global loads stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    rs_load(ra, 0);
    if (nk > 1) rs_load(rb, 1);
    rs_commit(ra, 0);
    if (nk > 2) rs_load(ra, 2);
    lds_barrier();
#pragma nounroll
    for (int kt = 0; kt < nk; kt += 2) {
      // tile kt from stage 0; tile kt + 1 goes from register set B to stage 1 meanwhile, set B is re-armed with tile kt + 3
      if (kt + 1 < nk) rs_commit(rb, 1);
      if (kt + 3 < nk) rs_load(rb, kt + 3);
      compute(0);
      lds_barrier();
      if (kt + 1 < nk) {
        if (kt + 2 < nk) rs_commit(ra, 0);
        if (kt + 4 < nk) rs_load(ra, kt + 4);
        compute(1);
        lds_barrier();
      }
    }
  } else {
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st)
    if (st < nk) issue(st, st);
  // full 64-deep tiles: ONE straight-line MFMA path in the loop (a conditional k-step makes the compiler copy all
  // accumulators at the loop back-edge), fragments of k-step s+1 are read from LDS while the MFMAs of step s run
  const int nfull = (k_tail || short_k) ? nk - 1 : nk;
#pragma nounroll
  for (int kt = 0; kt < nfull; ++kt) {
    sync_tile(kt);
    const char* sA = smem + (kt % STAGES) * STAGE;
    const char* sB = sA + A_BYTES;
    // Three k-steps of fragments are requested before the first MFMA and the fourth right behind it; the scheduling barriers pin
    // that order.  Left alone the machine scheduler sinks every ds_read to its use to save registers (86 VGPRs on the 128x192 tile)
    // and, with an LDS-DMA in flight, hipcc waits lgkmcnt(0) at every use: the loop became read -> wait -> ONE MFMA, twelve exposed
    // LDS latencies per K tile and wave.  Pinned, a wave waits once at the top of the tile (covered by the other waves of the SIMD) and
    // once, for reads issued two MFMA groups earlier, before the last k-step.
    if constexpr (TM * TN <= 4) {
      Frag f0 = ldfrag(sA, sB, 0);
      Frag f1 = ldfrag(sA, sB, 1);
      Frag f2 = ldfrag(sA, sB, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(f0);
      f0 = ldfrag(sA, sB, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma(f1);
      mma(f2);
      __builtin_amdgcn_sched_barrier(0);
      mma(f0);
    } else {                       // six accumulator tiles per wave (256x288): two fragment sets, each k-step's MFMAs cover the next reads
      Frag f0 = ldfrag(sA, sB, 0);
      Frag f1 = ldfrag(sA, sB, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(f0);
      f0 = ldfrag(sA, sB, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(f1);
      f1 = ldfrag(sA, sB, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma(f0);
      mma(f1);
    }
  }
  if (nfull < nk) {                                // K tail, executed once
    const int kt = nk - 1;
    sync_tile(kt);
    const char* sA = smem + (kt % STAGES) * STAGE;
    const char* sB = sA + A_BYTES;
    // K < 64: the first K/16 steps; otherwise the tile holds columns [K-64, K): its last (K % 64)/16 steps
    const int s_lo = short_k ? 0 : 4 - (k_tail >> 4), s_hi = short_k ? (p.K >> 4) : 4;
    for (int s = s_lo; s < s_hi; ++s) {
      const Frag f = ldfrag(sA, sB, s);
      mma(f);
    }
  }
  }
  if constexpr (SPLIT) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(accx[i][j][r], SPLIT_INV, acc[i][j][r]);
  }
  __syncthreads();              // every wave is done with the ring before the epilogue patches overwrite it

  // ------------------------------------------------------------------ epilogue
  float* patch = reinterpret_cast<float*>(smem) + wave * 1024;     // wave-private 32x32 f32
  // RoPE (host contract, gemm_v2_launch): rope_cols <= n_split, rope_cols % 4 == 0, rope_dim % 4 == 0, N % 4 == 0
  const bool has_rope = p.rope_cols > 0;
  const bool rope_pow2 = (p.rope_len & (p.rope_len - 1)) == 0;
  int rope_pr[TN];                                                  // pair index of the lane's first column in phase 2
#pragma unroll
  for (int j = 0; j < TN; ++j) rope_pr[j] = has_rope ? ((n0 + wn * WTN + j * 32 + (lane & 7) * 4) % p.rope_dim) >> 1 : 0;
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
        if (p.act == ACT_GELU) x = gelu_erf_fast(x);
        else if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
        else if (p.act == ACT_SIGMOID) x = 1.f / (1.f + __expf(-x));
        v[r] = has_rope ? x : x * cscale;             // RoPE: rotated in phase 2, the column scale follows the rotation
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
                const f32x4 f = {w4[0], w4[1], w4[2], w4[3]};
                store_h4(p.outT16 + (size_t)nt * p.ldT16 + mg, p.out_lo_off, f);
              }
              if (p.outT32) {
                const f32x4 f = {w4[0], w4[1], w4[2], w4[3]};
                *reinterpret_cast<f32x4*>(p.outT32 + (size_t)nt * p.ldT32 + mg) = f;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (mg + r < p.M) {
                  if (p.outT16) store_h1(p.outT16 + (size_t)nt * p.ldT16 + mg + r, p.out_lo_off, w4[r]);
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
      if (p.pool_w) {                                               // 32 rows -> 8 pooled rows, one 16-B store per lane
        const int w = p.pool_w, hw = w >> 1, pr = lane >> 3;
        const int t = (pr / hw) * 2 * w + 2 * (pr % hw);            // top-left row of the quad inside the tile
        const int m = mt0 + t, nn = nt0 + c4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(patch + t * 32 + c4), b = *reinterpret_cast<const f32x4*>(patch + (t + 1) * 32 + c4);
        const f32x4 c = *reinterpret_cast<const f32x4*>(patch + (t + w) * 32 + c4), d = *reinterpret_cast<const f32x4*>(patch + (t + w + 1) * 32 + c4);
        if (m < p.M && nn < p.N) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
          const int ww = w * w, in_w = m % ww;
          const size_t orow = (size_t)(m / ww) * (ww >> 2) + ((in_w / w) >> 1) * hw + ((in_w % w) >> 1);
          if (p.out32) *reinterpret_cast<f32x4*>(p.out32 + orow * p.ld32 + nn) = o;
          if (p.out16) store_h4(p.out16 + orow * p.ld16 + nn, p.out_lo_off, o);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = (lane >> 3) + 8 * q;
        const f32x4 t = *reinterpret_cast<const f32x4*>(patch + rr * 32 + c4);
        const int m = mt0 + rr, nn = nt0 + c4;
        if (m < p.M && nn < p.N) {
          float o[4] = {t[0], t[1], t[2], t[3]};
          if (has_rope) {                                           // a lane holds the pairs (nn, nn+1), (nn+2, nn+3) of row m
            if (nn < p.rope_cols && m < p.rope_rows) {
              const int tr = rope_pow2 ? (m & (p.rope_len - 1)) : (m % p.rope_len);
              const size_t ti = (size_t)tr * (p.rope_dim >> 1) + rope_pr[j];
              const float2 c = *reinterpret_cast<const float2*>(p.rope_cos + ti), sn = *reinterpret_cast<const float2*>(p.rope_sin + ti);
              const float a0 = o[0], a1 = o[1], a2 = o[2], a3 = o[3];
              // scalar FMAs on purpose (the asm barriers keep the SLP vectoriser from forming v_pk_mul/fma_f32 with op_sel here)
              float t0 = a1 * sn.x, t1 = a0 * sn.x, t2 = a3 * sn.y, t3 = a2 * sn.y;
              asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
              o[0] = fmaf(a0, c.x, -t0); o[1] = fmaf(a1, c.x, t1);
              o[2] = fmaf(a2, c.y, -t2); o[3] = fmaf(a3, c.y, t3);
              asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
            }
            if (p.col_scale) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (nn + e < p.N) o[e] *= p.col_scale[nn + e];
            }
          }
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
              const f32x4 ov = {o[0], o[1], o[2], o[3]};
              store_h4(p.out16 + (size_t)m * p.ld16 + nn, p.out_lo_off, ov);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (nn + e < p.N) {
                float x = o[e];
                if (p.res) x += p.res[rrow * p.ldres + nn + e];
                if (p.out32) p.out32[(size_t)m * p.ld32 + nn + e] = x;
                if (p.out16) store_h1(p.out16 + (size_t)m * p.ld16 + nn + e, p.out_lo_off, x);
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

template <int BM, int BN, int STAGES, int SPLIT = 0>
constexpr size_t v2_smem() { return (size_t)STAGES * ((BM + BN) * 128 + (SPLIT ? BN * 128 : 0) + (SPLIT == 1 ? BM * 128 : 0)); }

template <int BM, int BN, int WM, int WN, int STAGES, int SPLIT = 0, bool RS = false>
hipError_t v2_launch(const GemmParams& p, hipStream_t s) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t smem = v2_smem<BM, BN, STAGES, SPLIT>();
  gemm_v2_kernel<BM, BN, WM, WN, STAGES, SPLIT, RS><<<dim3(tiles), dim3(WM * WN * 64), smem, s>>>(p);
  return hipGetLastError();
}
template <int BM, int BN, int WM, int WN, int STAGES, int SPLIT = 0, bool RS = false>
hipError_t v2_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v2_kernel<BM, BN, WM, WN, STAGES, SPLIT, RS>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)v2_smem<BM, BN, STAGES, SPLIT>());
}
}  // namespace

hipError_t gemm_v2_init() {
  const hipError_t e[] = {
      v2_attr<128, 192, 4, 2, 2>(), v2_attr<128, 128, 4, 2, 2>(), v2_attr<128, 64, 4, 2, 2>(), v2_attr<64, 64, 2, 2, 2>(), v2_attr<64, 64, 2, 2, 4>(),
      v2_attr<128, 128, 4, 2, 2, 1>(), v2_attr<128, 64, 4, 2, 2, 1>(), v2_attr<64, 64, 2, 2, 2, 1>(),
      v2_attr<128, 128, 4, 2, 2, 2>(), v2_attr<128, 64, 4, 2, 2, 2>(), v2_attr<64, 64, 2, 2, 2, 2>(), v2_attr<128, 192, 4, 2, 2, 2>(),
#ifdef SAM2MI_EXPERIMENTAL
      v2_attr<128, 192, 4, 2, 2, 0, true>(), v2_attr<128, 128, 4, 2, 2, 0, true>(), v2_attr<128, 64, 4, 2, 2, 0, true>(),
      v2_attr<256, 192, 4, 2, 2>(), v2_attr<256, 256, 4, 4, 2>(), v2_attr<256, 192, 4, 3, 2>(), v2_attr<256, 128, 4, 2, 3>(), v2_attr<128, 128, 2, 2, 2>(),
      v2_attr<128, 64, 2, 2, 2>(), v2_attr<128, 64, 2, 2, 3>(), v2_attr<128, 128, 2, 2, 3>(), v2_attr<256, 128, 4, 2, 2>(),
      v2_attr<256, 64, 4, 2, 2>(), v2_attr<256, 288, 4, 3, 2>(),
#endif
  };
  for (hipError_t x : e)
    if (x != hipSuccess) return x;
  return hipSuccess;
}

static inline long tiles_of(const GemmParams& p, int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); }

// Tile choice (measured on MI355X, tools/gemm_bench.py; DESIGN.md "GEMM tuning log").  This is a one-barrier-per-K-tile
// structure at 2-3 workgroups per CU: its main loop tops out near 900 TFLOP/s even on cache-resident operands, and with
// K <= 4608 the un-overlapped prologue / epilogue of every output tile costs another 20-25 %.  What helps is waves:
// 8-wave workgroups beat 4-wave ones on every large shape; 128x128 (2 per CU) wins when N pads to 128 with <= 8 % waste,
// 128x64 (3 per CU) otherwise; 256x128 and 3/4-stage rings (1 workgroup per CU) lose 30-50 %, 16-wave 256x128 3-stage
// workgroups lose 10 %, a 256x256 8-wave tile on this loop structure loses 50 % (it needs the fine-grained multi-phase
// schedule, not more bytes per barrier).  The loop is bound by the per-CU L2->LDS rate (~25-29 B/clk) at these tiles.
// The losing tiles are only compiled with -DSAM2MI_EXPERIMENTAL (tile_hint, tools/gemm_bench.py).
hipError_t gemm_v2_launch(const GemmParams& p, hipStream_t s) {
  if (p.pool_w && ((p.pool_w != 2 && p.pool_w != 4 && p.pool_w != 8 && p.pool_w != 16) || (p.M & 31) || (p.N & 3) || (p.ld32 & 3) || (p.ld16 & 3) ||
                   p.res || p.rope_cols > 0 || p.n_split < p.N || p.outT16 || p.outT32))
    return hipErrorInvalidValue;                          // fused 2x2 max-pool: see gemm.h
  if (p.rope_cols > 0 && (p.rope_cols > p.n_split || (p.rope_cols & 3) || (p.rope_dim & 3) || (p.N & 3) || (p.ld32 & 3) || (p.ld16 & 3) ||
                          (p.ldres & 3) || p.rope_len <= 0))
    return hipErrorInvalidValue;                          // RoPE runs on the row-major 4-column phase of the epilogue
  if (p.a_lo_off) {                                      // split-f16 operands (f16x3 precision mode)
    if (!p.w_lo_off) return hipErrorInvalidValue;
    switch (gemm_v2_auto_tile(p)) {
      case 3:
      case 2: return v2_launch<128, 128, 4, 2, 2, 1>(p, s);
      case 1: return v2_launch<128, 64, 4, 2, 2, 1>(p, s);
      default: return v2_launch<64, 64, 2, 2, 2, 1>(p, s);
    }
  }
  if (p.w_lo_off) {                                      // weight split only (selective mode)
    int t = gemm_v2_auto_tile(p);
    if (p.tile_hint == 16) t = 3; else if (p.tile_hint == 10) t = 2; else if (p.tile_hint == 13) t = 1; else if (p.tile_hint == 5) t = 0;
    switch (t) {
      case 3: return v2_launch<128, 192, 4, 2, 2, 2>(p, s);
      case 2: return v2_launch<128, 128, 4, 2, 2, 2>(p, s);
      case 1: return v2_launch<128, 64, 4, 2, 2, 2>(p, s);
      default: return v2_launch<64, 64, 2, 2, 2, 2>(p, s);
    }
  }
#ifdef SAM2MI_EXPERIMENTAL
  const int force = p.tile_hint;
  if (force == 6) return gemm_v3_launch(p, s);
  if (force == 20 && (p.K & 63) == 0) return gemm_p4_launch(p, s);          // 256x256 staggered kernel (else: automatic)
  if (force == 2) return v2_launch<256, 128, 4, 2, 3>(p, s);
  if (force == 3) return v2_launch<128, 128, 2, 2, 2>(p, s);
  if (force == 7) return v2_launch<128, 64, 2, 2, 3>(p, s);
  if (force == 8) return v2_launch<128, 128, 2, 2, 3>(p, s);
  if (force == 11) return v2_launch<256, 128, 4, 2, 2>(p, s);
  if (force == 12) return v2_launch<256, 64, 4, 2, 2>(p, s);
  if (force == 17) return v2_launch<256, 192, 4, 2, 2>(p, s);
  if (force == 14) return v2_launch<256, 256, 4, 4, 2>(p, s);
  if (force == 15) return v2_launch<256, 192, 4, 3, 2>(p, s);
  if (force == 4) return v2_launch<128, 64, 2, 2, 2>(p, s);
  if (force == 18) return v2_launch<256, 288, 4, 3, 2>(p, s);
  // register-staged variants (hints 41..43): K % 64 == 0 only
  if (force >= 41 && force <= 43 && (p.K & 63) == 0) {
    if (force == 43) return v2_launch<128, 192, 4, 2, 2, 0, true>(p, s);
    if (force == 42) return v2_launch<128, 128, 4, 2, 2, 0, true>(p, s);
    return v2_launch<128, 64, 4, 2, 2, 0, true>(p, s);
  }
#endif
  int tile = gemm_v2_auto_tile(p);
  switch (p.tile_hint) {                                 // the four production tiles can be forced (benchmarks, parity tests)
    case 16: tile = 3; break;
    case 10: tile = 2; break;
    case 13: tile = 1; break;
    case 5: tile = 0; break;
    case 9: tile = 4; break;
    default: break;
  }
  switch (tile) {
    case 3: return v2_launch<128, 192, 4, 2, 2>(p, s);
    case 2: return v2_launch<128, 128, 4, 2, 2>(p, s);
    case 1: return v2_launch<128, 64, 4, 2, 2>(p, s);
    case 4: return v2_launch<64, 64, 2, 2, 4>(p, s);
    default: return v2_launch<64, 64, 2, 2, 2>(p, s);
  }
}

// automatic tile: 3 = 128x192, 2 = 128x128, 1 = 128x64 (8 waves), 0 = 64x64 (4 waves), 4 = 64x64 with a 4-stage ring
int gemm_v2_auto_tile(const GemmParams& p) {
  // long K, N a multiple of 192: fc2 of stages 3-4 (K >= 2048), QKV and fc1 of stage 4 (K = 1152, N = 3456 / 4608: 91 vs 96, 119 vs 132 us;
  // the stage-4 projection, N = K = 1152, stays on 128x64: 43 vs 50 us): the 128x192 tile re-reads the A panel N/192 instead of
  // N/64 times through the L2->LDS path that bounds this kernel (measured +14 % / +16 % on those two shapes) - when it still fills
  // the chip (a batch-1 encoder call has 96 such tiles: 36.5 us vs 25.0 us on 64x64 tiles)
  // (a 256x288 tile - fc2 of stage 3 at batch 8 as exactly one 12-wave workgroup per CU, 43 % fewer L2->LDS bytes per flop, 5 instead of 8
  // fragment reads per 6 MFMAs - was measured in round 3: 158 vs 110 us alone, 153 vs 123 us in the pipeline.  One workgroup per CU leaves
  // nobody to cover the vmcnt + barrier of every K tile; hint 18 in SAM2MI_EXPERIMENTAL builds.)
  if (p.N % 192 == 0 && (p.K >= 2048 || (p.K >= 1024 && p.N >= 2304)) && tiles_of(p, 128, 192) >= 320) return 3;      // stage-4 fc2 at batch 8: 384 tiles, 121 vs 138 us on 128x128
  const int n128 = ((p.N + 127) / 128) * 128;
  const long t128 = tiles_of(p, 128, 128);
  const bool fits128 = (n128 - p.N) * 100 <= 8 * p.N;              // N pads to 128 with at most 8 % waste
  if (fits128 && p.K > 64 && (t128 >= 1536 || (t128 >= 512 && p.K >= 2048))) return 2;      // K = 64 (one K tile: all prologue and epilogue): 128x64, more workgroups in flight
  if (tiles_of(p, 128, 64) >= 512) return 1;
  // small grids (the M = 4096 GEMMs of the tracking path, stage 4 of a batch-1 encoder call) run ~1 workgroup per CU with operands
  // that the previous kernel has just written, i.e. served by the memory-side cache, not by the XCD's L2: with one K tile in flight
  // the loop is latency-bound (ff2 of the memory attention, K = 2048: 26.6 us in the pipeline vs 13.9 us on L2-hot operands in
  // tools/gemm_bench.py).  A 4-stage ring (3 tiles in flight) takes it to 16 us in the pipeline.  Larger grids hide the latency with
  // 3-4 workgroups of the 2-stage kernel per CU instead (stage-3 fc2 at batch 1, 576 tiles: 25.0 vs 33.1 us).
  if (p.K >= 512 && tiles_of(p, 64, 64) <= 320) return 4;
  return 0;
}

// kernel name as rocprofv3 prints it (profiling by instantiation)
const char* gemm_v2_kernel_name(const GemmParams& p) {
  if (p.tile_hint != 0 && !p.w_lo_off) return "gemm_v2_kernel<forced tile>";
  static const char* names[5] = {"gemm_v2_kernel<64, 64, 2, 2, 2, 0, false>", "gemm_v2_kernel<128, 64, 4, 2, 2, 0, false>",
                                 "gemm_v2_kernel<128, 128, 4, 2, 2, 0, false>", "gemm_v2_kernel<128, 192, 4, 2, 2, 0, false>",
                                 "gemm_v2_kernel<64, 64, 2, 2, 4, 0, false>"};
  static const char* split_names[5] = {"gemm_v2_kernel<64, 64, 2, 2, 2, 1, false>", "gemm_v2_kernel<128, 64, 4, 2, 2, 1, false>",
                                       "gemm_v2_kernel<128, 128, 4, 2, 2, 1, false>", "gemm_v2_kernel<128, 128, 4, 2, 2, 1, false>",
                                       "gemm_v2_kernel<64, 64, 2, 2, 2, 1, false>"};
  static const char* wsplit_names[5] = {"gemm_v2_kernel<64, 64, 2, 2, 2, 2, false>", "gemm_v2_kernel<128, 64, 4, 2, 2, 2, false>",
                                        "gemm_v2_kernel<128, 128, 4, 2, 2, 2, false>", "gemm_v2_kernel<128, 192, 4, 2, 2, 2, false>",
                                        "gemm_v2_kernel<64, 64, 2, 2, 2, 2, false>"};
  return (p.a_lo_off ? split_names : p.w_lo_off ? wsplit_names : names)[gemm_v2_auto_tile(p)];
}

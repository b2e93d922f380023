// MFMA f16 GEMM v3 for gfx950: PERSISTENT workgroups with a 4-slot LDS-DMA ring that keeps 3 K-tiles in
// flight ACROSS output tiles.
//
// Why: the Hiera GEMMs have short K (576..4608), so with one output tile per workgroup the first LDS-DMA
// round trip (prologue) and the epilogue are exposed on every tile, and the small high-occupancy tiles that
// hide them (gemm2.hip, 128x64) run into the L2->LDS DMA bandwidth (measured ~17 TB/s = 626 TFLOP/s at 42.7 flop/B).
// Here one workgroup per CU (4 waves, one per SIMD, 128x128 tile = 64 flop/B) walks a static list of tiles; the
// K-tile stream is linearised over (tile, kt) and the loader runs 3 steps ahead of the MFMAs, so the next tile's
// first K-tiles land during the current tile's last MFMAs and its epilogue.
//
// Roles: waves 0-3 are MFMA consumers (64x64 each), waves 4-5 stream the two halves of A, waves 6-7 of B.  vmcnt is per wave and
// counts loads, stores and LDS-DMA together in issue order, so keeping the DMA in dedicated loader waves is what
// makes a COUNTED wait possible: the consumers' epilogue stores and residual loads never enter the loaders' queue.
// Ring protocol (per 64-deep K step g):
//   loader:   s_waitcnt vmcnt(8 * steps_in_flight_after)    // its 8 pieces of step g have landed (never 0 mid-stream)
//   all:      s_barrier                                     // step g visible to everyone; step g-1 consumed by everyone
//   loader:   issue LDS-DMA of step g+3 into slot (g+3)&3   // == slot of step g-1, free after the barrier
//   consumer: MFMAs of step g from slot g&3 (+ epilogue when it closes a tile)
// Same XOR-swizzled linear LDS image and the same fused epilogue as gemm2.hip; the epilogue patch lives
// outside the ring so it never races with in-flight DMA.
#include "gemm.h"

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int BM = 128, BN = 128;
constexpr int SLOT = (BM + BN) * 128;          // 32 KiB
constexpr int NSLOT = 4;
constexpr int RING = SLOT * NSLOT;             // 128 KiB
constexpr int PATCH = 4 * 4096;                // 4 waves x 32x32 f32
constexpr int PIECES = 8;                      // 1-KiB LDS-DMA pieces per loader wave per step (64 rows x 128 B)

__global__ __launch_bounds__(512, 1) void gemm_v3_kernel(const GemmParams p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int srow = lane >> 3, spc = lane & 7;
  const int ntiles = tiles_m * tiles_n;
  const int nk = (p.K + 63) >> 6;
  const int nwg = gridDim.x;
  const int my_tiles = (ntiles - (int)blockIdx.x + nwg - 1) / nwg;     // tiles blockIdx.x, +nwg, ...
  const int nsteps = my_tiles * nk;
  const bool k_tail = (p.K & 63) != 0;

  // tile id -> (m0, n0): n fastest so that concurrently running workgroups share A rows through L2
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int id = blockIdx.x + t * nwg;
    m0 = (id / tiles_n) * BM;
    n0 = (id % tiles_n) * BN;
  };

  if (wave >= 4) {
    // ================================================================== loader waves
    // waves 4,5: rows 0-63 / 64-127 of A; waves 6,7: rows 0-63 / 64-127 of B
    const bool isB = wave >= 6;
    const int half_id = wave & 1;
    const half_t* base = isB ? p.W : p.A;
    const int ld = isB ? p.ldw : p.lda;
    const int rows_max = (isB ? p.N : p.M) - 1;
    const int lds_off = (isB ? BM * 128 : 0) + half_id * 64 * 128;
    const half_t* src[PIECES];             // per-lane source pointer of each piece at kt = 0 of the current tile
    // pieces alternate between two swizzle phases: row = 8 j + srow -> (row >> 1) & 7 = (4 j + (srow >> 1)) & 7
    const int kc0 = (spc ^ ((srow >> 1) & 7)) << 3;
    const int kc1 = (spc ^ ((4 + (srow >> 1)) & 7)) << 3;
    int cur_tile = -1;
    auto prep_tile = [&](int t) {
      int m0, n0;
      tile_origin(t, m0, n0);
      const int r0 = (isB ? n0 : m0) + half_id * 64;
#pragma unroll
      for (int j = 0; j < PIECES; ++j)
        src[j] = base + (size_t)min(r0 + j * 8 + srow, rows_max) * ld + ((j & 1) ? kc1 : kc0);
      cur_tile = t;
    };
    auto issue = [&](int g) {
      const int t = g / nk, kt = g - t * nk;
      if (t != cur_tile) prep_tile(t);
      char* sb = smem + (g & (NSLOT - 1)) * SLOT + lds_off;
      const int koff = kt * 64;
      if (k_tail && kt == nk - 1) {
        const int back0 = (koff + kc0 >= p.K) ? koff + kc0 : 0;         // K tail: read column 0 instead (never consumed)
        const int back1 = (koff + kc1 >= p.K) ? koff + kc1 : 0;
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[j] + koff - ((j & 1) ? back1 : back0)), (lds_ptr_t)(sb + j * 1024), 16, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[j] + koff), (lds_ptr_t)(sb + j * 1024), 16, 0, 0);
      }
    };
    for (int g = 0; g < 3 && g < nsteps; ++g) issue(g);
#pragma nounroll
    for (int g = 0; g < nsteps; ++g) {
      const int later = min(2, nsteps - 1 - g);
      if (later == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (g + 3 < nsteps) issue(g + 3);
    }
    return;
  }

  // ==================================================================== consumer waves
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float* patch = reinterpret_cast<float*>(smem + RING) + wave * 1024;
  const bool has_rope = p.rope_cols > 0;
  const bool vec_ok = (p.N & 3) == 0 && (p.ld32 & 3) == 0 && (p.ld16 & 3) == 0 && (p.ldres & 3) == 0;

#pragma nounroll
  for (int g = 0; g < nsteps; ++g) {
    __builtin_amdgcn_s_barrier();          // the loaders retired step g before arriving here

    const int t = g / nk, kt = g - t * nk;
    const char* sA = smem + (g & (NSLOT - 1)) * SLOT;
    const char* sB = sA + BM * 128;
    const int ksteps = (kt == nk - 1 && k_tail) ? ((p.K & 63) >> 4) : 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < ksteps) {
        half8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 64 + i * 32 + fr;
          af[i] = *reinterpret_cast<const half8*>(sA + row * 128 + (((2 * s + fh) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = wn * 64 + j * 32 + fr;
          bf[j] = *reinterpret_cast<const half8*>(sB + row * 128 + (((2 * s + fh) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
      }
    }

    if (kt != nk - 1) continue;
    // ------------------------------------------------------------------ epilogue of tile t (loads of tile t+1 are in flight)
    int m0, n0;
    tile_origin(t, m0, n0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nt0 = n0 + wn * 64 + j * 32;
      const int n = nt0 + fr;
      const bool n_ok = n < p.N;
      const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
      const float cscale = (p.col_scale && n_ok) ? p.col_scale[n] : 1.f;
      const bool transposed = nt0 >= p.n_split;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int mt0 = m0 + wm * 64 + i * 32;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mt0 + acc_row(r, lane);
          float x = acc[i][j][r] + bias;
          acc[i][j][r] = 0.f;
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
        if (mt0 >= p.M || nt0 >= p.N) continue;
        if (transposed) {
          if (n_ok) {
            const int nt = n - p.n_split;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int mg = mt0 + 8 * q + 4 * fh;
              float w4[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
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
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[acc_row(r, lane) * 32 + fr] = v[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int c4 = (lane & 7) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rr = (lane >> 3) + 8 * q;
          const f32x4 tv = *reinterpret_cast<const f32x4*>(patch + rr * 32 + c4);
          const int m = mt0 + rr, nn = nt0 + c4;
          if (m < p.M && nn < p.N) {
            float o[4] = {tv[0], tv[1], tv[2], tv[3]};
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}
}  // namespace

hipError_t gemm_v3_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RING + PATCH);
}

hipError_t gemm_v3_launch(const GemmParams& p, hipStream_t s) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const int grid = ntiles < 256 ? ntiles : 256;          // one persistent workgroup per CU
  gemm_v3_kernel<<<dim3(grid), dim3(512), RING + PATCH, s>>>(p, tiles_m, tiles_n);
  return hipGetLastError();
}

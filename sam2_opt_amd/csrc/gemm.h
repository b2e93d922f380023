// MFMA GEMM with fused epilogues:  C[M,N] = epi( A[M,K] (f16) . W[N,K]^T (f16) ), f32 accumulate.
// Both operands are K-contiguous (activations row-major, weights in nn.Linear layout), which is
// exactly the gfx950 32x32x16 fragment layout (8 consecutive k per lane).
#pragma once
#include "common.h"

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3 };
enum { PREC_AUTO = 0, PREC_F16 = 1, PREC_WSPLIT = 2, PREC_FULL = 3 };

struct GemmParams {
  const half_t* A; int lda;      // [M, lda] f16, lda % 8 == 0
  const half_t* W; int ldw;      // [N, ldw] f16, ldw % 8 == 0
  int M, N, K;                   // K % 32 == 0 (or % 48)
  const float* bias;             // [N] or null
  int act;                       // ACT_*
  const float* col_scale;        // [N] or null: v = v * col_scale[n] (after act; CXBlock gamma)
  const float* res; int ldres;   // optional f32 residual added last; row index = res_mod ? m % res_mod : m
  int res_mod;
  float* out32; int ld32;        // optional f32 row-major output (columns < n_split)
  half_t* out16; int ld16;       // optional f16 row-major output (columns < n_split)
  int n_split;                   // columns >= n_split are stored transposed; multiple of 32; == N if unused
  half_t* outT16; int ldT16;     // outT16[(n - n_split) * ldT16 + m]  (ldT16 % 4 == 0)
  float* outT32; int ldT32;      // outT32[(n - n_split) * ldT32 + m]
  // axial RoPE fused on columns n < rope_cols of rows m < rope_rows (rope_cols == 0: off):
  //   the pair (2p, 2p+1), p = (n % rope_dim) / 2, is rotated by the angle in table row m % rope_len
  const float* rope_cos; const float* rope_sin;  // [rope_len, rope_dim/2]
  int rope_len, rope_rows, rope_cols, rope_dim;
  // 2x2 max-pool of the OUTPUT rows fused into the epilogue (gemm_v2 only; Hiera's shortcut  maxpool(proj(x)), hieradet.py:139-140):
  // rows are tokens in window-major order with pool_w x pool_w windows (pool_w in {2,4,8,16}, 0: off); out32 / out16 then have
  // M / 4 rows in window-major order with (pool_w/2)^2 windows.  Needs M % 32 == 0, N % 4 == 0, no residual / RoPE / transposed part.
  int pool_w;
  int tile_hint;                 // 0 = automatic tile choice; 1..5 force a v2 tile (benchmarks)
  const half_t* xs_pack;         // W in the piece order of the X-stationary kernel (gemm_xs.hip), or null
  const half_t* ks_pack;         // W in the piece order of the accumulator-stationary kernel (gemm_ks.hip: N = 576), or null
  int xs_scale_cols;             // col_scale is 1 from this column on (lets the X-stationary kernel keep only the q scale); 0: unknown
  // split-f16 operands (precision modes, common.h): lo arrays at these element offsets behind A / W.  Both set: full split (3 MFMAs
  // per product); w_lo_off alone: weight split (2 MFMAs); a_lo_off alone is not a mode.  out_lo_off != 0: out16 / outT16 are
  // written as hi + lo
  size_t a_lo_off, w_lo_off, out_lo_off;
  // what the caller asks of engine_core's run_gemm in a split-precision context (ignored by the kernels themselves):
  // PREC_AUTO = the context's default (f16 / full split), PREC_F16 / PREC_WSPLIT / PREC_FULL = this linear's entry in the
  // selective plan; no_out_lo: the f16 output feeds a consumer that reads the hi plane only - skip the lo plane
  int prec, no_out_lo;
  const half_t* xs_wpack;        // weight-split image of the X-stationary kernel (gemm_xs_wsplit_pack), or null
  int outT_hi_only;              // the consumer of outT16 reads its hi plane only (lets the weight-split X-stationary kernel take the launch)
  // LayerNorm fused into the operand load of the X-stationary kernel (gemm_xs.h): A is ignored, the operand is the normalised
  // row of ln_x32 [M, K] (ln_ld floats per row); xs_pack / bias must be the LN-folded ones.  X-stationary shapes only.
  const float* ln_x32; int ln_ld; float ln_eps;
};

static inline GemmParams gemm_params_zero() {
  GemmParams p;
  __builtin_memset(&p, 0, sizeof(p));
  return p;
}

// returns hipSuccess or an error; `flops`/`launches` accumulate statistics when non-null
hipError_t gemm_launch(const GemmParams& p, hipStream_t stream);
hipError_t gemm_init();   // sets the dynamic-LDS attributes once
// gemm2.hip: LDS-DMA pipelined kernel (K % 16 == 0); gemm_launch dispatches to it
hipError_t gemm_v2_launch(const GemmParams& p, hipStream_t stream);
hipError_t gemm_v2_init();
int gemm_v2_auto_tile(const GemmParams& p);              // the tile the automatic choice takes (0: 64x64 .. 3: 128x192)
const char* gemm_v2_kernel_name(const GemmParams& p);    // kernel name as rocprofv3 prints it
// gemm3.hip: persistent loader/consumer kernel (large M*N, K % 16 == 0)
// ---- out-projection + residual + LayerNorm of a d_model = 256 attention sub-block in one kernel (gemm_rowln.hip)
struct RowLnParams {
  const half_t* a16; int lda;        // operand [M, 256] f16 ...
  const float* o_part; const float* ml_part; int splits; int part_rows;   // ... or (o_part != nullptr) the flash256 partials to combine:
                                     //     o_part [splits, part_rows, 256] f32, ml_part [splits, part_rows, 2] f32 (attn.h, Flash256Params)
  const half_t* w; const float* bias;   // W [256, 256] f16 row-major (out, in), bias [256] or nullptr
  const float* res; float* out32;    // out32 = res + operand W^T + bias   [M, 256] f32 (may alias)
  const float* ln_w; const float* ln_b; float eps;
  half_t* out16; int ld16;           // LayerNorm(out32) * ln_w + ln_b as f16
  int M;                             // multiple of 32
  int kc;                            // operand channels: 0 / 256, or 64 (a16 [M, 64] / o_part [splits, part_rows, 64], W [256, 64])
};
hipError_t gemm_rowln_launch(const RowLnParams& p, hipStream_t stream);

// ---- Hiera out-projection + residual + norm2 in one kernel (gemm_projln.hip): N = K = C in {144, 288, 576}
struct ProjLnParams {
  const half_t* a16; int lda;        // attention output [M, C] f16
  const half_t* wpack;               // proj weight [C, C] in the X-stationary piece order (gemm_xs_pack)
  const float* bias;                 // [C]
  const float* res; float* out32;    // out32 = res + a16 W^T + bias   [M, C] f32, contiguous rows (may alias)
  const float* ln_w; const float* ln_b; float eps;
  half_t* out16; int ld16;           // LayerNorm(out32) * ln_w + ln_b as f16
  int M, C;                          // M % 32 == 0
  // f16s precision mode (C = 144 / 288): wpack_lo = the packed image of the weight's lo plane (2 MFMAs per fragment pair);
  // a_lo_off != 0: the operand's lo plane sits that many elements behind a16 (3 MFMAs)
  const half_t* wpack_lo; size_t a_lo_off;
};
bool gemm_projln_supported(int C);
hipError_t gemm_projln_launch(const ProjLnParams& p, hipStream_t stream);
hipError_t gemm_projln_init();       // dynamic-LDS attributes, once

hipError_t gemm_v3_launch(const GemmParams& p, hipStream_t stream);
hipError_t gemm_v3_init();
// gemm4.hip: 256x256 staggered 4-phase kernel, one workgroup per CU (experimental; K % 64 == 0)
hipError_t gemm_p4_launch(const GemmParams& p, hipStream_t stream);
hipError_t gemm_p4_init();

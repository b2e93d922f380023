// X-stationary GEMM for K = 144 / 288 / 576 (gemm_xs.hip): X rows live in registers, the packed weight streams through LDS.
#pragma once
#include <algorithm>
#include "gemm.h"

// packed weight image (gemm_xs_pack): 1-KiB pieces = MFMA fragment tiles of 32 rows x 16 k (unit l of a piece: row l / 2, k half
// (l & 1) ^ ((row >> 3) & 1)); a stage = XS_STAGE_PIECES pieces (576 / K chunks of 32 columns x K / 16 k-steps) in XS_STAGE_SLOTS slots
constexpr int XS_STAGE_PIECES = 36, XS_STAGE_SLOTS = 40;

struct GemmXsParams {
  const half_t* x16; int ldx;      // [M, K] f16 activations (K-contiguous), ldx % 8 == 0
  const half_t* wpack;             // weight [N, K] in piece order (gemm_xs_pack)
  const float* bias;               // [N]
  const float* col_scale;          // [>= scale_cols] or null: v *= col_scale[n] after bias / activation, columns n < scale_cols
  int scale_cols;                  //   (the q pre-scale of a QKV projection); <= 576, rounded up to 32 <= n_split
  int act;                         // ACT_NONE or ACT_GELU (row-major columns only)
  int M, N;                        // N % 8 == 0
  int n_split;                     // columns >= n_split go to outT16 (multiple of 32; == N: none)
  half_t* out16; int ld16;         // f16 row-major output of columns < n_split (or null), ld16 % 8 == 0
  half_t* outT16; int ldT16;       // outT16[(n - n_split) * ldT16 + m], ldT16 % 4 == 0
  float* out32; int ld32;          // f32 row-major output of columns < n_split (or null)
  const float* res; int ldres;     // f32 residual added to out32 (may alias out32)
  int splits;                      // 0: automatic column split
  // LayerNorm fused into the operand load (x16 ignored): X = (x32 - mean) * rstd per row, statistics over all K channels of
  // the row; the weight / bias must carry the affine part (Lin16::xs_ln_pack).  ldx32 % 4 == 0.
  const float* ln_x32; int ldx32; float ln_eps;
  // weight split (f16s precision mode): wpack is the image of gemm_xs_wsplit_pack ([W_hi | W_lo] chunk pairs), two MFMA chains per
  // output tile; no activation.  out_lo_off != 0: out16 is written as hi + lo planes (outT16: hi only)
  int wsplit; size_t out_lo_off;
};
bool gemm_xs_supported(int N, int K);
size_t gemm_xs_pack_bytes(int N, int K);
hipError_t gemm_xs_pack(const half_t* w, int N, int K, int ldw, half_t* wpack, hipStream_t s);
size_t gemm_xs_wsplit_pack_bytes(int N, int K);
// scratch: 2 * ceil32(N) * K halfs (the interleaved [hi | lo] matrix; may be released after the stream has run)
hipError_t gemm_xs_wsplit_pack(const half_t* w_hi, const half_t* w_lo, int N, int K, half_t* wpack, half_t* scratch, hipStream_t s);
hipError_t gemm_xs_launch(const GemmXsParams& p, int K, hipStream_t s);
hipError_t gemm_xs_init();

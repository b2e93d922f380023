// Accumulator-stationary GEMM for long K and N = 576 (gemm_ks.hip): out32[M,576] = X[M,K] . W[576,K]^T + bias (+ residual).
#pragma once
#include "gemm.h"

struct GemmKsParams {
  const half_t* x16; int ldx;      // [M, K] f16 activations, ldx % 8 == 0
  const half_t* wpack;             // weight [576, K] in piece order (gemm_ks_pack)
  const float* bias;               // [576]
  const float* res; int ldres;     // optional f32 residual (may alias out32)
  float* out32; int ld32;          // [M, 576] f32
  int M, K;                        // K % 64 == 0
};
#ifdef SAM2MI_EXPERIMENTAL      // parity-tested, equal end to end to the tiled kernel on stage-3 fc2: not in the default build
bool gemm_ks_supported(int N, int K);                 // N == 576, K % 64 == 0
size_t gemm_ks_pack_bytes(int N, int K);
hipError_t gemm_ks_pack(const half_t* w, int N, int K, int ldw, half_t* wpack, hipStream_t s);
hipError_t gemm_ks_launch(const GemmKsParams& p, hipStream_t s);
hipError_t gemm_ks_init();
#else
static inline bool gemm_ks_supported(int, int) { return false; }
static inline size_t gemm_ks_pack_bytes(int, int) { return 0; }
static inline hipError_t gemm_ks_pack(const half_t*, int, int, int, half_t*, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t gemm_ks_launch(const GemmKsParams&, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t gemm_ks_init() { return hipSuccess; }
#endif

// Attention kernels (see the .hip files for the algorithms).
#pragma once
#include "common.h"

// ---- Hiera head_dim-72 attention (attn_hiera.hip)
struct HieraAttnParams {
  const half_t* q; int ldq;     // [Mq, ldq] f16; head h at columns [h*72, h*72+72)
  const half_t* k; int ldk;     // [Mk, ldk] f16; same head layout
  const half_t* vT; int ldvT;   // V^T [heads*72, ldvT] f16: row h*72+d, column = key row index
  half_t* o; int ldo;           // [Mq, ldo] f16 output, same head layout
  int heads;
  int GQ, GK;                   // queries / keys per group; group g owns q rows [g*GQ, (g+1)*GQ), keys [g*GK, ...)
  int wq, wk;                   // query i sees key j (indices inside the group) iff i / wq == j / wk
  int num_groups;
  float scale_log2e;            // head_dim^-0.5 * log2(e)
  // selective-split mode (f16s): q and k carry a lo plane qk_lo_off elements behind the hi plane (scores from three products),
  // the output is written as hi + lo (o_lo_off, 0: hi only).  Stage-3 shapes only (unmasked groups, GQ % 128 == 0).
  size_t qk_lo_off, o_lo_off;
};
hipError_t hiera_attn_launch(const HieraAttnParams& p, hipStream_t stream);
const char* hiera_attn_kernel_name(const HieraAttnParams& p);

// ---- head_dim-72 attention of the f16x3 precision mode (attn_precise.hip): same grouping / masking as HieraAttnParams, but
// q, k, V^T arrive as f32 (the QKV GEMM's f32 outputs) and are split into hi + lo f16 MFMA operands in registers:
// S = K Q^T with 3 products, O += V^T P^T with V split (2 products), P in f16; output written as hi + lo at o / o + o_lo_off.
struct PreciseAttnParams {
  const float* q; int ldq;      // [Mq, ldq] f32, pre-scaled by head_dim^-0.5 * log2(e); head h at columns [h*72, h*72+72)
  const float* k; int ldk;      // [Mk, ldk] f32
  const float* vT; int ldvT;    // V^T [heads*72, ldvT] f32
  half_t* o; int ldo;           // [Mq, ldo] f16 hi plane
  size_t o_lo_off;              // lo plane offset (elements), must be non-zero
  int heads;
  int GQ, GK, wq, wk, num_groups;   // as HieraAttnParams
};
hipError_t precise_attn_launch(const PreciseAttnParams& p, hipStream_t stream);

// ---- Hiera attention for the sizes whose windows are zero-padded (hiera_generic.hip): any head_dim in {56, 72, 96}
struct GenericAttnParams {
  const half_t* q; int ldq;     // [num_groups * GQ, ldq] f16, pre-scaled by head_dim^-0.5 * log2(e); head h at columns [h*HD, (h+1)*HD)
  const half_t* k; int ldk;     // [num_groups * GK, ldk]
  const half_t* vT; int ldvT;   // V^T [heads*HD, ldvT], column = key row index
  half_t* o; int ldo;           // [num_groups * GQ, ldo]
  int heads;
  int GQ, GK;                   // rows per group (multiples of 32)
  int wq, wk, vk;               // query i sees key j (inside the group) iff i / wq == j / wk and j % wk < vk
  int num_groups;
};
hipError_t generic_attn_launch(const GenericAttnParams& p, int head_dim, hipStream_t stream);

// ---- single-head d=256 flash attention with split-KV (attn_flash256.hip)
struct Flash256Params {
  const half_t* q; int ldq;     // [Nq, ldq] f16, RoPE applied and PRE-SCALED by 256^-0.5 * log2(e); Nq % 128 == 0
  const half_t* k; int ldk;     // [>= ceil32(Nk), ldk] f16
  const half_t* vT; int ldvT;   // V^T [256, ldvT] f16, ldvT >= ceil32(Nk); pad columns must be finite
  int Nq, Nk;
  int splits;                   // KV splits (grid.y)
  float* o_part;                // [splits, Nq, 256] f32 un-normalised partial outputs
  float* ml_part;               // [splits, Nq, 2] f32 (running max in log2 domain, running sum)
  half_t* out; int ldout;       // [Nq, ldout] f16 final (written by the combine pass); nullptr: no combine pass, the consumer
                                //   combines the partials itself (gemm_rowln.hip)
  float scale_log2e;            // unused by the kernel (q is pre-scaled); kept for the debug entry point
  size_t out_lo_off;            // split-f16 mode: `out` is written as hi + lo (common.h); 0: off
  int dv;                       // channels of the values: 0 / 256, or 64: vT is [64, ldvT] (the memory tokens themselves, value projection
                                //   applied by the consumer), o_part [splits, Nq, 64]; partials only (out must be null)
};
hipError_t flash256_launch(const Flash256Params& p, hipStream_t stream);
const char* flash256_kernel_name(const Flash256Params& p);
hipError_t flash256_init();   // dynamic-LDS attribute, once
int flash256_pick_splits(int Nq, int Nk);
int flash256_pick_splits_dv64(int Nq, int Nk);   // the DV = 64 variant (two workgroups per CU)   // KV splits that fill the chip once (<= 16: the size of the partial buffers)

// ---- tiny fp32 attentions of the two-way mask decoder (attn_small.hip)
// scratch (optional): partial results of the token -> image kernel that handles all T <= 8 queries of a prompt per workgroup
// (batch * heads * Tk / 512 * Tq * 18 floats); without it the per-query kernel runs
// q [Tq, ldq], k/v [Tk, ld], heads x hd, out [Tq, ldo]; all f32.  softmax(q k^T / sqrt(hd)) v
hipError_t small_attn_launch(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                             float* out, int ldo, int Tq, int Tk, int heads, int hd, int batch,
                             size_t q_bstride, size_t kv_bstride, size_t o_bstride, hipStream_t stream, float* scratch = nullptr, size_t scratch_floats = 0,
                             half_t* out16 = nullptr, size_t out16_lo_off = 0);   // out16: image -> token case only, f16 output instead of `out`

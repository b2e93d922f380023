// X-stationary GEMM for short K (K = 144 / 288 / 576: the QKV, output-projection and fc1 linears of Hiera stages 1-3,
// /root/reference/sam2/sam2/modeling/backbones/hieradet.py:56-81 and :123-129):   out = epi( X[M,K] . W[N,K]^T ).
//
// Why a second GEMM kernel: in the tiled kernel (gemm2.hip) every 128x128 output tile re-fetches its A and W panels from
// L2 into LDS, and with K <= 576 a workgroup lives for 3-9 K tiles - the kernel is bound by the per-CU L2->LDS path and by
// prologue/epilogue that nothing overlaps.  Here the short K is used the other way round:
//   * a wave keeps its 32 token rows of X (all of K) in REGISTERS as MFMA operands for its whole life (K/16 fragments),
//     so X is read from HBM/L2 exactly once and never touches LDS;
//   * the weight matrix streams past in 36-KiB stages (576 / K chunks of 32 output columns) through a 2-slot LDS ring
//     by LDS-DMA, from an image packed at weight-load time so that every 1-KiB piece is one MFMA fragment tile
//     (32 rows x 32 B, 8 full cache lines per piece, halves swapped on rows with bit 3 set: conflict-free ds_read_b128);
//   * per chunk one accumulator tile: S^T = W_chunk . X^T (row-major outputs: W rows permuted by pi23 so that a lane
//     owns 8 consecutive output columns of its token -> 16-B f16 / 32-B f32 stores straight from the accumulator) or
//     S = X . W_chunk^T (transposed output V^T: a lane owns 4 consecutive tokens of its column -> 8-B stores);
//   * the epilogue (bias, erf-GELU, column scale, f32 residual) runs on that one tile while the other wave of the SIMD is in
//     its MFMA chain (two workgroups of 4 waves per CU).
// Grid = (M / 128 token blocks) x (column splits): the column range is split so that the chip holds two waves per SIMD.
#include "gemm_xs.h"
#include <type_traits>
#include <cstdlib>

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

static __device__ __forceinline__ int pi23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

static __device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

constexpr int NW = 8;                    // waves per workgroup = 8 token blocks of 32 (one workgroup per CU, two waves per SIMD)
constexpr int STAGE_PIECES = XS_STAGE_PIECES;     // 1-KiB fragment pieces per ring stage ...
constexpr int PPW = (STAGE_PIECES + NW - 1) / NW;     // ... 5 per wave: the packed image pads every stage with 4 zero pieces
constexpr int STAGE_SLOTS = PPW * NW;    // 40
static_assert(STAGE_SLOTS == XS_STAGE_SLOTS, "packed image layout (gemm_xs.h)");
constexpr int STAGE_B = STAGE_SLOTS * 1024;
constexpr int NST = 3;                   // two stages in flight while one is consumed
constexpr int MAX_COLS = 1152;           // output columns per workgroup (bias table in LDS)
constexpr int MAX_SCALE = 576;           // of which at most this many leading ones carry a column scale
constexpr int LDS_B = NST * STAGE_B + (MAX_COLS + MAX_SCALE) * 4;       // 129,792 B
// fragments per LDS read batch (double-buffered); TIGHT (K = 576, weight split, f32 output: 144 X registers + two accumulators + the
// prefetched residual) takes batches of 4 to stay inside the 256 registers of two waves per SIMD
template <int K, bool TIGHT = false> constexpr int frag_batch() { return K == 144 ? 5 : TIGHT ? 4 : 6; }

// GELU: erf-GELU on the row-major columns; F32: f32 output (+ residual) instead of f16 for the row-major columns
// ABL != 0: timing ablations (wrong results; tuning aid): 1 no LDS-DMA in the loop, 2 no MFMA, 3 no output stores
// WS (weight split, the f16s precision mode): the packed image holds W as a 2-term f16 split, chunk 2j = rows of W_hi, chunk 2j+1 =
// the same rows of W_lo (lo = f16((w - hi) * 2^11), common.h); a logical chunk takes both chains - X W_hi^T into the main
// accumulator, X W_lo^T into a second one that is folded in x 2^-11 - so the rounding of the weights (the coherent part of the
// f16-mode error) is gone at 2x the MFMA work and 2x the weight stream, X still read once.  K = 576 (one chunk per stage): the
// two halves of a logical chunk sit in consecutive ring stages and the stage loop advances by two.
template <int K, bool GELU, bool F32, int ABL = 0, bool WS = false>
__global__ __launch_bounds__(64 * NW, 1) void gemm_xs_kernel(const GemmXsParams p) {
  constexpr int KS = K / 16;             // k-steps = pieces per chunk
  constexpr int CPS = STAGE_PIECES / KS; // chunks (of 32 output columns) per stage
  constexpr int FB = frag_batch<K, (K == 576 && WS && F32)>();
  constexpr int LS = (WS && CPS == 1) ? 2 : 1;      // ring stages per loop iteration
  constexpr int LCS = WS ? 1 : 0;                   // log2(physical chunks per logical chunk)
  static_assert(KS * CPS == STAGE_PIECES, "K must divide 576");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int tok0 = (blockIdx.x * NW + wave) * 32;
  // this workgroup's stage range (column split)
  const int nchunks = ((p.N + 31) / 32) << LCS;             // physical chunks
  const int nstages = ((nchunks + CPS - 1) / CPS + LS - 1) / LS * LS;
  const int per = ((nstages / LS + gridDim.y - 1) / gridDim.y) * LS;
  const int st_lo = blockIdx.y * per, st_hi = min(nstages, st_lo + per);
  if (st_lo >= st_hi) return;

  // X fragments: X[token fr][16 s + 8 fh + e]; rows past M are clamped (never stored)
  half8 xf[KS];
  if (p.ln_x32) {                        // wave-uniform: LayerNorm of the f32 residual-stream row, fused into the load
    ln_row_fragments<KS>(p.ln_x32 + (size_t)min(tok0 + fr, p.M - 1) * p.ldx32 + fh * 8, p.ln_eps, xf);
  } else {
    const half_t* xp = p.x16 + (size_t)min(tok0 + fr, p.M - 1) * p.ldx + fh * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const half8*>(xp + s * 16);
  }
  // bias (and the column scale of the first scale_cols columns) of this workgroup's column range, in LDS: a global load
  // at the top of every chunk sits in front of the MFMA chain it initialises (measured: ~1 us per chunk under load)
  float* bias_lds = reinterpret_cast<float*>(smem + NST * STAGE_B);
  float* scale_lds = bias_lds + MAX_COLS;
  const int col_lo = (st_lo * CPS * 32) >> LCS, col_hi = min(p.N, (st_hi * CPS * 32) >> LCS);
  for (int i = tid; i < MAX_COLS; i += 64 * NW) bias_lds[i] = (col_lo + i < col_hi) ? p.bias[col_lo + i] : 0.f;
  for (int i = tid; i < MAX_SCALE; i += 64 * NW) scale_lds[i] = (p.col_scale && col_lo + i < p.scale_cols) ? p.col_scale[col_lo + i] : 1.f;
  __syncthreads();
  const char* wp = reinterpret_cast<const char*>(p.wpack);
  const unsigned lane_off = (unsigned)lane * 16u;
  // ring step st loads stage min(st, st_hi-1) into slot (st - st_lo) % NST (past the end: the last stage again, into a slot
  // nobody reads), so every wave issues exactly PPW pieces per iteration and the vmcnt waits can be counted
  auto issue = [&](int st) {
    char* sb = smem + ((st - st_lo) % NST) * STAGE_B;
    const char* cb = wp + (size_t)min(st, st_hi - 1) * STAGE_B;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int q = wave + NW * k;
      const char* src = sgpr_ptr(cb + q * 1024);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + lane_off), (lds_ptr_t)(sb + q * 1024), 16, 0, 0);
    }
  };
  const int r1 = pi23(fr);
  const int rd_perm = r1 * 32 + ((fh ^ ((r1 >> 3) & 1)) << 4);     // row-major outputs: MFMA row i <-> column pi23(i)
  const int rd_nat = fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);      // transposed outputs: natural order

  // One chunk = one 32x32 accumulator tile over all of K.  The two variants are separate straight-line bodies (a branch
  // inside the MFMA chain would cut the scheduling region at every fragment batch).
  // chain<SWAP>: acc = W_chunk . X^T (SWAP = false) or X . W_chunk^T (SWAP = true), fragments read in double-buffered batches
  auto chain = [&](const char* sW, f32x16& sa, auto swap) {
    constexpr bool SWAP = decltype(swap)::value;
    constexpr int NB = (KS + FB - 1) / FB;
    half8 cur[FB], nxt[FB];
#pragma unroll
    for (int j = 0; j < FB; ++j)
      if (j < KS) cur[j] = *reinterpret_cast<const half8*>(sW + j * 1024);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int j = 0; j < FB; ++j)
        if ((b + 1) * FB + j < KS) nxt[j] = *reinterpret_cast<const half8*>(sW + ((b + 1) * FB + j) * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < FB; ++j)
        if (b * FB + j < KS) {
          if (ABL != 2) sa = SWAP ? mfma32(xf[b * FB + j], cur[j], sa) : mfma32(cur[j], xf[b * FB + j], sa);
          else asm volatile("" ::"v"(cur[j]), "v"(xf[b * FB + j]));
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < FB; ++j) cur[j] = nxt[j];
    }
  };
  // row-major columns: S^T = W_chunk X^T; register 8 ks + e <-> column n0 + 16 ks + 8 fh + e of token fr
  auto row_init = [&](int n0) {
    f32x16 sa;
    const float* bp = bias_lds + (n0 - col_lo) + 8 * fh;       // the bias is the initial accumulator (zero past N)
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(bp + 16), b3 = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
    for (int e = 0; e < 4; ++e) { sa[e] = b0[e]; sa[4 + e] = b1[e]; sa[8 + e] = b2[e]; sa[12 + e] = b3[e]; }
    return sa;
  };
  // f32 residual of this wave's 32 tokens x 32 columns, loaded BEFORE the DMA issue and the chain, by inline-asm loads that the
  // compiler does not track, and retired by a COUNTED s_waitcnt in front of row_fin (res_wait<pieces issued since>).  With an
  // LDS-DMA in flight hipcc waits vmcnt(0) at the use of any ordinary global_load result (cdna_hip_programming.md, glds notes): the
  // loads inside row_fin drained the two ring stages in flight once per 8-column group - the projection kernel spent 51 % of its
  // wave cycles in s_waitcnt (profiles/r03d_util.md).  Addresses are clamped (rows past M / columns past N are never stored).
  struct Res { f32x4 v[4]; };
  auto res_load = [&](int n0) {
    Res r;
    const float* rp = p.res + (size_t)min(tok0 + fr, p.M - 1) * p.ldres;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* q = rp + min(n0 + 16 * ks + 8 * fh, p.N - 8);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.v[2 * ks]) : "v"(q) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r.v[2 * ks + 1]) : "v"(q) : "memory");
    }
    return r;
  };
  auto res_wait = [](Res& r, auto younger) {        // every vector-memory operation older than the youngest `younger` ones has retired
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]) : "n"(decltype(younger)::value) : "memory");
  };
  auto row_fin = [&](const f32x16& sa, int n0, const Res& res) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = GELU ? gelu_erf_fast(sa[r]) : sa[r];
    if (n0 < p.scale_cols) {                                   // wave-uniform; the table holds 1 from scale_cols on
      const float* cp = scale_lds + (n0 - col_lo) + 8 * fh;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
      const f32x4 c2 = *reinterpret_cast<const f32x4*>(cp + 16), c3 = *reinterpret_cast<const f32x4*>(cp + 20);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] *= c0[e]; v[4 + e] *= c1[e]; v[8 + e] *= c2[e]; v[12 + e] *= c3[e]; }
    }
    const int tok = tok0 + fr;
    if (ABL == 3) { asm volatile("" ::"v"(v[0]), "v"(v[15])); return; }
    if (tok < p.M) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int n = n0 + 16 * ks + 8 * fh;
        if (n >= p.N) continue;                                // N % 8 == 0: an 8-column group is valid or wholly past N
        if (F32) {
          float* op = p.out32 + (size_t)tok * p.ld32 + n;
          f32x4 a = {v[8 * ks], v[8 * ks + 1], v[8 * ks + 2], v[8 * ks + 3]};
          f32x4 b = {v[8 * ks + 4], v[8 * ks + 5], v[8 * ks + 6], v[8 * ks + 7]};
          if (p.res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += res.v[2 * ks][e]; b[e] += res.v[2 * ks + 1][e]; }
          }
          *reinterpret_cast<f32x4*>(op) = a;
          *reinterpret_cast<f32x4*>(op + 4) = b;
        } else {
          half8 h, l;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            h[e] = (half_t)v[8 * ks + e];
            if (WS) l[e] = split_lo(v[8 * ks + e], h[e]);
          }
          *reinterpret_cast<half8*>(p.out16 + (size_t)tok * p.ld16 + n) = h;
          if (WS && p.out_lo_off) *reinterpret_cast<half8*>(p.out16 + p.out_lo_off + (size_t)tok * p.ld16 + n) = l;
        }
      }
    }
  };
  // transposed columns (V^T): S = X W_chunk^T; register r <-> token tok0 + (r & 3) + 8 (r >> 2) + 4 fh of column n0 + fr
  auto trans_init = [&](int n0) {
    f32x16 sa;
    const float bt = bias_lds[n0 - col_lo + fr];
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = bt;
    return sa;
  };
  auto trans_fin = [&](const f32x16& sa, int n0) {
    if (n0 + fr < p.N) {
      half_t* op = p.outT16 + (size_t)(n0 - p.n_split + fr) * p.ldT16 + tok0 + 4 * fh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)sa[4 * g + e];
        if (tok0 + 8 * g + 4 * fh + 3 < p.M) *reinterpret_cast<half4*>(op + 8 * g) = h;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (tok0 + 8 * g + 4 * fh + e < p.M) op[8 * g + e] = h[e];
        }
      }
    }
  };

  auto zero16 = [] {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
  };
  auto fold = [](f32x16& sa, const f32x16& sc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = fmaf(sc[r], SPLIT_INV, sa[r]);
  };
  issue(st_lo);
  issue(st_lo + 1);
  if constexpr (WS && CPS == 1) {
    // K = 576, weight split: stage st = W_hi rows of logical chunk st / 2, stage st + 1 = their W_lo.  Vector-memory order of a
    // wave at the first wait (old -> young): pieces(st) | pieces(st+1) | stores(st-2, st-1: >= 2), at the second one
    // pieces(st+1) | stores | pieces(st+2): both may leave PPW + 2 operations outstanding (PPW where a wave may have stored nothing)
#pragma nounroll
    for (int st = st_lo; st < st_hi; st += 2) {
      const int n0 = (st >> 1) * 32;
      const bool row = n0 < p.n_split;
      const bool counted = st != st_lo && !(p.M & (32 * NW - 1));
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      __builtin_amdgcn_s_barrier();
      Res res;
      if (F32 && p.res && row) res = res_load(n0);
      issue(st + 2);
      f32x16 sa, sc = zero16();
      if (row) { sa = row_init(n0); chain(smem + ((st - st_lo) % NST) * STAGE_B + rd_perm, sa, std::false_type{}); }
      else { sa = trans_init(n0); chain(smem + ((st - st_lo) % NST) * STAGE_B + rd_nat, sa, std::true_type{}); }
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      __builtin_amdgcn_s_barrier();
      issue(st + 3);
      if (row) {
        chain(smem + ((st + 1 - st_lo) % NST) * STAGE_B + rd_perm, sc, std::false_type{});
        fold(sa, sc);
        if (F32 && p.res) res_wait(res, std::integral_constant<int, 2 * PPW>{});       // younger: the pieces of stages st + 2 and st + 3
        row_fin(sa, n0, res);
      }
      else { chain(smem + ((st + 1 - st_lo) % NST) * STAGE_B + rd_nat, sc, std::true_type{}); fold(sa, sc); trans_fin(sa, n0); }
    }
  } else {
  constexpr int SPS = WS ? CPS : 2 * CPS;                 // stores per stage a wave issues at least (2 per logical chunk)
#pragma nounroll
  for (int st = st_lo; st < st_hi; ++st) {
    // Stage st landed; stage st+1 stays in flight.  Vector-memory order of a wave (old -> young):
    //   pieces(st) | stores(st-2) | pieces(st+1) | stores(st-1)          (5 pieces per stage; >= 2 stores per logical chunk)
    // vmcnt retires in order, so with N outstanding allowed, N <= (operations younger than pieces(st)) keeps every piece of
    // stage st complete: 5 in the first iteration, 5 + SPS in the second, 5 + 2 SPS afterwards - without waiting for the
    // store acknowledgements.  M % 256 != 0: some waves store nothing -> 5.  (The residual loads of the f32 variants are further
    // younger operations: the counts stay valid.)
    if (st == st_lo || (p.M & (32 * NW - 1))) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else if (st == st_lo + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + SPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2 * SPS) : "memory");
    __builtin_amdgcn_s_barrier();                         // ... for every wave; everyone is past stage st-1 -> its slot is free
    Res res[F32 ? (CPS >> LCS) : 1];                      // residuals of every chunk of the stage, in front of the DMA issue
    if (F32 && p.res) {
#pragma unroll
      for (int c = 0; c < CPS; c += 1 << LCS) res[c >> LCS] = res_load(min(((st * CPS + c) >> LCS) * 32, p.N - 8));
    }
    if (ABL != 1) issue(st + 2);
    const char* sb = smem + ((st - st_lo) % NST) * STAGE_B;
#pragma unroll
    for (int c = 0; c < CPS; c += 1 << LCS) {
      const int n0 = ((st * CPS + c) >> LCS) * 32;
      if (n0 >= p.N) break;                               // wave-uniform (zero-padded tail of the packed image)
      if (n0 < p.n_split) {
        f32x16 sa = row_init(n0);
        chain(sb + c * KS * 1024 + rd_perm, sa, std::false_type{});
        if (WS) { f32x16 sc = zero16(); chain(sb + (c + 1) * KS * 1024 + rd_perm, sc, std::false_type{}); fold(sa, sc); }
        if (F32 && p.res) res_wait(res[c >> LCS], std::integral_constant<int, PPW>{});      // younger: the pieces of stage st + 2 (and stores)
        row_fin(sa, n0, res[F32 ? (c >> LCS) : 0]);
      } else {
        f32x16 sa = trans_init(n0);
        chain(sb + c * KS * 1024 + rd_nat, sa, std::true_type{});
        if (WS) { f32x16 sc = zero16(); chain(sb + (c + 1) * KS * 1024 + rd_nat, sc, std::true_type{}); fold(sa, sc); }
        trans_fin(sa, n0);
      }
    }
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing duplicate pieces land before the LDS is released
}

// one thread per 16-B unit of the packed image: stage = 36 pieces = CPS chunks x KS k-steps; piece (c, s) holds rows
// n = 32 (st CPS + c) + row of W, k = 16 s + 8 h .. + 7 at byte 32 row + 16 (h ^ ((row >> 3) & 1)); rows past N are zero
template <int K>
__global__ void gemm_xs_pack_kernel(const half_t* __restrict__ w, int N, int ldw, half_t* __restrict__ out, long units) {
  constexpr int KS = K / 16;
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const long slot = u >> 6;
  const int l = (int)(u & 63), row = l >> 1, h = (l & 1) ^ ((row >> 3) & 1);
  const long stage = slot / STAGE_SLOTS;
  const int q = (int)(slot - stage * STAGE_SLOTS);       // pieces 36..39 of a stage are zero padding
  const long chunk = stage * (STAGE_PIECES / KS) + q / KS;
  const int s = q % KS;
  const long n = chunk * 32 + row;
  half8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (half_t)0.f;
  if (q < STAGE_PIECES && n < N) v = *reinterpret_cast<const half8*>(w + n * ldw + 16 * s + 8 * h);
  *reinterpret_cast<half8*>(out + u * 8) = v;
}

// the virtual weight matrix of the weight-split image: row 64 j + r = W_hi row 32 j + r, row 64 j + 32 + r = W_lo row 32 j + r (zero past N)
__global__ void gemm_xs_interleave_kernel(const half_t* __restrict__ hi, const half_t* __restrict__ lo, int N, int K, half_t* __restrict__ out, long total) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= total) return;
  const long row = u / K;
  const int k = (int)(u - row * K);
  const long n = (row >> 6) * 32 + (row & 31);
  out[u] = n < N ? (((row >> 5) & 1) ? lo : hi)[n * K + k] : (half_t)0.f;
}

// (more column splits than that - 2 or 4 waves of workgroups, each 1/2 or 1/4 as long, so that a kernel of the tracking stream finds a
// free CU sooner - were measured in round 3: 205.9 -> 200.9 -> 192.5 frames/s; the X tile is re-read per split.)
template <int K>
hipError_t launch_k(const GemmXsParams& p, hipStream_t s) {
  constexpr int CPS = STAGE_PIECES / (K / 16);
  if (p.wsplit) {
    constexpr int LS = CPS == 1 ? 2 : 1;
    const int nchunks = 2 * ((p.N + 31) / 32);
    const int nstages = ((nchunks + CPS - 1) / CPS + LS - 1) / LS * LS;
    const int tb = (p.M + 32 * NW - 1) / (32 * NW);
    const int groups = nstages / LS;
    const int max_groups = 2 * MAX_COLS / (32 * CPS) / LS;              // bias table: at most MAX_COLS logical columns per workgroup
    int splits = p.splits > 0 ? p.splits : (256 + tb - 1) / tb;
    splits = std::max(splits, (groups + max_groups - 1) / max_groups);
    splits = std::max(1, std::min(splits, groups));
    if ((groups + splits - 1) / splits > max_groups) return hipErrorInvalidValue;
    const dim3 grid(tb, splits), block(64 * NW);
    if (p.act != ACT_NONE) return hipErrorInvalidValue;      // no activation: the QKV (f16 out) and output (f32 out + residual) projections
    if (p.out32) gemm_xs_kernel<K, false, true, 0, true><<<grid, block, LDS_B, s>>>(p);
    else gemm_xs_kernel<K, false, false, 0, true><<<grid, block, LDS_B, s>>>(p);
    return hipGetLastError();
  }
  const int nstages = (p.N + 32 * CPS - 1) / (32 * CPS);
  const int tb = (p.M + 32 * NW - 1) / (32 * NW);
  // column splits: fill the chip with two workgroups per CU (512 slots) at least once, never more splits than stages
  int splits = (256 + tb - 1) / tb;                                     // one workgroup (8 waves) per CU
  if (p.splits > 0) splits = p.splits;
  const int max_stages = MAX_COLS / (32 * CPS);                         // bias table: at most MAX_COLS columns per workgroup
  splits = std::max(splits, (nstages + max_stages - 1) / max_stages);
  splits = std::max(1, std::min(splits, nstages));
  if ((nstages + splits - 1) / splits > max_stages) return hipErrorInvalidValue;
  const dim3 grid(tb, splits), block(64 * NW);
  if (p.out32 && p.act == ACT_GELU) gemm_xs_kernel<K, true, true><<<grid, block, LDS_B, s>>>(p);      // tests only
  else if (p.out32) gemm_xs_kernel<K, false, true><<<grid, block, LDS_B, s>>>(p);
  else if (p.act == ACT_GELU) gemm_xs_kernel<K, true, false><<<grid, block, LDS_B, s>>>(p);
  else {
    static const int abl = getenv("SAM2MI_XS_ABL") ? atoi(getenv("SAM2MI_XS_ABL")) : 0;      // tuning aid (f16 / no-GELU variant only)
    if (abl == 1) gemm_xs_kernel<K, false, false, 1><<<grid, block, LDS_B, s>>>(p);
    else if (abl == 2) gemm_xs_kernel<K, false, false, 2><<<grid, block, LDS_B, s>>>(p);
    else if (abl == 3) gemm_xs_kernel<K, false, false, 3><<<grid, block, LDS_B, s>>>(p);
    else gemm_xs_kernel<K, false, false><<<grid, block, LDS_B, s>>>(p);
  }
  return hipGetLastError();
}
template <int K>
hipError_t attr_k() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, false, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, false, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, false, false, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_xs_kernel<K, false, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
  return e;
}
}  // namespace

hipError_t gemm_xs_init() {
  hipError_t e[3] = {attr_k<144>(), attr_k<288>(), attr_k<576>()};
  for (int i = 0; i < 3; ++i)
    if (e[i] != hipSuccess) return e[i];
  return hipSuccess;
}

bool gemm_xs_supported(int N, int K) { return (K == 144 || K == 288 || K == 576) && N % 8 == 0 && N >= 8; }

size_t gemm_xs_pack_bytes(int N, int K) {
  if (!gemm_xs_supported(N, K)) return 0;
  const int cps = STAGE_PIECES / (K / 16);
  const int nstages = (N + 32 * cps - 1) / (32 * cps);
  return (size_t)nstages * STAGE_B;
}

hipError_t gemm_xs_pack(const half_t* w, int N, int K, int ldw, half_t* wpack, hipStream_t s) {
  const long units = (long)(gemm_xs_pack_bytes(N, K) / 16);
  if (units == 0) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((units + 255) / 256)), block(256);
  if (K == 144) gemm_xs_pack_kernel<144><<<grid, block, 0, s>>>(w, N, ldw, wpack, units);
  else if (K == 288) gemm_xs_pack_kernel<288><<<grid, block, 0, s>>>(w, N, ldw, wpack, units);
  else gemm_xs_pack_kernel<576><<<grid, block, 0, s>>>(w, N, ldw, wpack, units);
  return hipGetLastError();
}

size_t gemm_xs_wsplit_pack_bytes(int N, int K) { return gemm_xs_pack_bytes(2 * ((N + 31) / 32) * 32, K); }

hipError_t gemm_xs_wsplit_pack(const half_t* w_hi, const half_t* w_lo, int N, int K, half_t* wpack, half_t* scratch, hipStream_t s) {
  const int rows = 2 * ((N + 31) / 32) * 32;
  const long total = (long)rows * K;
  gemm_xs_interleave_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(w_hi, w_lo, N, K, scratch, total);
  hipError_t e = hipGetLastError();
  return e != hipSuccess ? e : gemm_xs_pack(scratch, rows, K, K, wpack, s);
}

hipError_t gemm_xs_launch(const GemmXsParams& p, int K, hipStream_t s) {
  if (p.M <= 0) return hipSuccess;
  if (!gemm_xs_supported(p.N, K) || (!p.ln_x32 && (p.ldx & 7)) || (p.ln_x32 && (p.ldx32 & 3)) || (p.n_split < p.N && (p.n_split & 31)) || !p.bias) return hipErrorInvalidValue;
  if (p.n_split < p.N && (!p.outT16 || (p.ldT16 & 3))) return hipErrorInvalidValue;
  if (p.col_scale && (p.scale_cols <= 0 || p.scale_cols > MAX_SCALE || (p.scale_cols + 31) / 32 * 32 > p.n_split)) return hipErrorInvalidValue;
  if (p.out16 && (p.ld16 & 7)) return hipErrorInvalidValue;
  if (!p.out32 && !p.out16) return hipErrorInvalidValue;
  if (p.out32 && ((p.ld32 & 3) || (p.res && (p.ldres & 3)))) return hipErrorInvalidValue;
  switch (K) {
    case 144: return launch_k<144>(p, s);
    case 288: return launch_k<288>(p, s);
    default: return launch_k<576>(p, s);
  }
}

// Hiera multi-head attention for head_dim 72 (windowed, pooled-query and global blocks).
// Reference: MultiScaleAttention.forward, /root/reference/sam2/sam2/modeling/backbones/hieradet.py:56-81.
//
// One wave = one tile of 32 queries of one (group, head); a group is a window (or a pack of
// small windows with a block-diagonal mask).  Flash-style online softmax, swapped products:
//   S^T[key][q] = K . Q^T      (A = K fragment from LDS, B = Q fragment in registers)
//   O^T[d][q]  += V^T . P^T    (A = V^T fragment from LDS, B = P^T straight from the S^T
//                               accumulator: its rows are the keys the second product sums over)
// so softmax statistics are per lane (column = query) and P never touches LDS.
// head_dim 72 is padded in LDS/registers only: QK^T sums over 80 (5 MFMA k-steps, pad = 0),
// PV produces 96 rows (3 MFMA row tiles) of which 72 are stored.
#include "attn.h"
#include <cstdlib>

namespace {
constexpr int HD = 72;
constexpr int KROW = 88;              // K tile row stride in halfs (176 B: conflict-free ds_read_b128)
constexpr int VROW = 36;              // V^T tile row stride in halfs (72 B: conflict-free ds_read_b64)
constexpr int K_TILE = 32 * KROW;     // halfs
constexpr int V_TILE = 96 * VROW;     // halfs (rows 72..95 are never written; their products are discarded)
constexpr int REGION = K_TILE + V_TILE;

// SPLIT (f16s precision mode): q / k as 2-term f16 splits (lo plane qk_lo_off elements behind the hi plane), three products for
// the scores, output as hi + lo - see hiera_attn_v2_kernel<SPLIT> below; the K_lo tile sits behind the V^T tile of a region.
template <bool SHARE, bool MASK, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void hiera_attn_kernel(const HieraAttnParams p) {
  constexpr int REG = REGION + (SPLIT ? K_TILE : 0);
  __shared__ __attribute__((aligned(16))) half_t smem[(SHARE ? 1 : 4) * REG];
  constexpr int NTHR = SHARE ? 256 : 64;
  constexpr int K_IT = (32 * 9 + NTHR - 1) / NTHR;      // 16-B chunks of the K tile per thread
  constexpr int V_IT = (HD * 8 + NTHR - 1) / NTHR;      // 8-B chunks of the V^T tile per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int st = SHARE ? tid : lane;
  const int fr = lane & 31, fh = lane >> 5;
  const int qtiles = p.GQ / 32;
  const int total = p.num_groups * p.heads * qtiles;
  // XCD-aware order (blocks are dealt round-robin to the 8 XCDs): an XCD takes a contiguous range of tasks, so the heads and query
  // tiles of one group - which read the same K / V^T rows, 144-B head segments of shared 128-B lines - meet in ONE L2
  int task = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  const bool live = task < total;
  if (!live) task = total - 1;          // keep barrier participation uniform
  const int qt = task % qtiles;
  const int gh = task / qtiles;
  const int head = gh % p.heads, grp = gh / p.heads;

  half_t* sK = smem + (SHARE ? 0 : wave * REG);
  half_t* sV = sK + K_TILE;
  half_t* sKl = sK + REGION;            // SPLIT only
  // zero the head-dim pad (cols 72..87) of the K tile once; staging never touches it
  for (int i = st; i < 32 * 16; i += NTHR) {
    sK[(i >> 4) * KROW + HD + (i & 15)] = (half_t)0.f;
    if (SPLIT) sKl[(i >> 4) * KROW + HD + (i & 15)] = (half_t)0.f;
  }

  // ---- Q fragments (B operand): lane holds Q[q = fr][d = 16 s + 8 fh + j]; Q is pre-scaled by 72^-0.5 * log2(e)
  const size_t qrow = (size_t)grp * p.GQ + qt * 32 + fr;
  half8 qf[5], ql[SPLIT ? 5 : 1];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    if (s == 4 && fh == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (half_t)0.f;
      if (SPLIT) ql[SPLIT ? s : 0] = qf[s];
    } else {
      const half_t* qp = p.q + qrow * p.ldq + head * HD + s * 16 + fh * 8;
      qf[s] = *reinterpret_cast<const half8*>(qp);
      if (SPLIT) ql[SPLIT ? s : 0] = *reinterpret_cast<const half8*>(qp + p.qk_lo_off);
    }
  }

  const int q_in_grp = qt * 32 + fr;                 // this lane's query index inside the group
  const int nwin = p.wq >= 32 ? 1 : 32 / p.wq;
  const int kv_start = ((qt * 32) / p.wq) * p.wk;    // first key (inside group) this q-tile can see
  const int kv_tiles = (nwin * p.wk) / 32;
  const int q_win = q_in_grp / p.wq;

  f32x16 o[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const half_t* kbase = p.k + ((size_t)grp * p.GK) * p.ldk + head * HD;
  const half_t* vbase = p.vT + (size_t)head * HD * p.ldvT + (size_t)grp * p.GK;

  // register-staged tiles: the loads of tile t+1 are issued before the MFMAs of tile t
  half8 rk[K_IT], rkl[SPLIT ? K_IT : 1];
  half4 rv[V_IT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const int c = st + i * NTHR;
      if (c < 32 * 9) {
        const half_t* kp = kbase + (size_t)(k0 + c / 9) * p.ldk + (c % 9) * 8;
        rk[i] = *reinterpret_cast<const half8*>(kp);
        if (SPLIT) rkl[SPLIT ? i : 0] = *reinterpret_cast<const half8*>(kp + p.qk_lo_off);
      }
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      const int c = st + i * NTHR;
      if (c < HD * 8) rv[i] = *reinterpret_cast<const half4*>(vbase + (size_t)(c >> 3) * p.ldvT + k0 + (c & 7) * 4);
    }
  };
  auto swrite = [&]() {
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
      const int c = st + i * NTHR;
      if (c < 32 * 9) {
        *reinterpret_cast<half8*>(sK + (c / 9) * KROW + (c % 9) * 8) = rk[i];
        if (SPLIT) *reinterpret_cast<half8*>(sKl + (c / 9) * KROW + (c % 9) * 8) = rkl[SPLIT ? i : 0];
      }
    }
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
      const int c = st + i * NTHR;
      if (c < HD * 8) *reinterpret_cast<half4*>(sV + (c >> 3) * VROW + (c & 7) * 4) = rv[i];
    }
  };

  gload(kv_start);
  swrite();
  __syncthreads();
#pragma nounroll
  for (int kt = 0; kt < kv_tiles; ++kt) {
    const int k0 = kv_start + kt * 32;
    if (kt + 1 < kv_tiles) gload(k0 + 32);
    // ---- S^T = K Q^T  (32 keys x 32 queries), exp2 domain
    // the K (and K_lo) fragments of the tile are requested before the score chains, the V^T fragments between the chains and the
    // softmax arithmetic that covers their latency; the scheduling barriers pin that (left alone the scheduler sinks each ds_read
    // to its MFMA: one exposed LDS latency per MFMA)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    constexpr bool KL_UP = SPLIT && SHARE;       // per-wave tiles (!SHARE) stage 58 registers of the next tile: K_lo is read behind the first chain there
    half8 kf[5], kl[SPLIT ? 5 : 1], vf[3][2];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      kf[ks] = *reinterpret_cast<const half8*>(sK + fr * KROW + ks * 16 + fh * 8);
      if (KL_UP) kl[SPLIT ? ks : 0] = *reinterpret_cast<const half8*>(sKl + fr * KROW + ks * 16 + fh * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) s = mfma32(kf[ks], qf[ks], s);
    if (SPLIT) {                    // cross terms in their own chain, folded in x 2^-11
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        if (!KL_UP) kl[SPLIT ? ks : 0] = *reinterpret_cast<const half8*>(sKl + fr * KROW + ks * 16 + fh * 8);
        sc = mfma32(kf[ks], ql[SPLIT ? ks : 0], sc);
        sc = mfma32(kl[SPLIT ? ks : 0], qf[ks], sc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fmaf(sc[r], SPLIT_INV, s[r]);
    }
    constexpr bool V_UP = !(SPLIT && MASK && !SHARE);        // the masked split variant on per-wave tiles is 9 registers over 256 with the V^T prefetch
    auto read_v = [&]() {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const half_t* vr = sV + (t * 32 + fr) * VROW + ks * 16 + fh * 4;
          const half4 lo = *reinterpret_cast<const half4*>(vr);
          const half4 hi = *reinterpret_cast<const half4*>(vr + 8);
          vf[t][ks] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    if constexpr (V_UP) {
      __builtin_amdgcn_sched_barrier(0);
      read_v();
      __builtin_amdgcn_sched_barrier(0);
    }
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MASK && ((k0 + acc_row(r, lane)) / p.wk) != q_win) s[r] = -1e30f;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.f;
    half8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pv = __builtin_amdgcn_exp2f(s[r] - m_new);
      if (MASK && s[r] <= -1e30f) pv = 0.f;          // fully masked tile for this query: m_new may still be the sentinel
      psum += pv;
      pf[r >> 3][r & 7] = (half_t)pv;
    }
    psum += __shfl_xor(psum, 32, 64);
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    l_run += psum;
    // ---- O^T += V^T P^T
    if constexpr (!V_UP) read_v();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) o[t] = mfma32(vf[t][ks], pf[ks], o[t]);
    }
    __syncthreads();                                  // tile fully consumed
    if (kt + 1 < kv_tiles) swrite();
    __syncthreads();
  }

  if (live) {
    const float inv = 1.f / l_run;
    half_t* orow = p.o + qrow * p.ldo + head * HD;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = t * 32 + 8 * g + 4 * fh;
        if (d < HD) {
          half4 h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = o[t][4 * g + e] * inv;
            h[e] = (half_t)v;
            if (SPLIT) l[e] = split_lo(v, h[e]);
          }
          *reinterpret_cast<half4*>(orow + d) = h;
          if (SPLIT && p.o_lo_off) *reinterpret_cast<half4*>(orow + p.o_lo_off + d) = l;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// v2 (shared-tile, unmasked case: stage-3 windows and the global blocks): same swapped products, but staged like the
// d=256 kernel (attn_flash256.hip v3):
//   * K / V^T tiles arrive by LDS-DMA into a 4-stage ring (counted vmcnt, one barrier per tile, never drained);
//     K image dense [32 keys][9 x 16 B] (144-B rows are conflict-free for ds_read_b128 as they stand), V^T image
//     [72 d][4 x 16 B] with the chunk index XOR-ed by (d >> 2) & 3 on the DMA source address;
//   * K rows are read permuted (bits 2,3 of the key index swapped) so the V^T fragment that matches a lane's P^T
//     values is one contiguous ds_read_b128;
//   * iteration i runs S_{i+1} = K_{i+1} Q^T beside the exponentials of tile i, then O^T += V_i^T P_i^T;
//   * the O rescale is deferred to an outer loop (reference maximum raised only past a threshold).
//   * VALU diet (the d=72 products leave the softmax VALU-bound): the score accumulator starts at -m_ref, so the
//     probability is exp2 of the MFMA result with no subtraction; and the row sum l = sum_k p comes out of the MFMA too:
//     "row 72" of the V^T image is a row of ones, so O^T[72][q] accumulates exactly the f16 probabilities the numerator uses.
// head_dim pad: k-step 4 has only 8 real columns - lanes of the upper half re-read chunk 8 against a zero Q fragment;
// PV row tile 2 has 8 real rows + the ones row - the other lanes read the ones row too and their products are discarded.
constexpr int H2_NST = 4;
constexpr int H2_KT = 32 * 144;            // 4608 B
constexpr int H2_VT = HD * 64;             // 4608 B
// stage = [K | V^T | ones] (SPLIT: [K_hi | K_lo | V^T | ones]); the 64 B behind the V^T image are "row 72" = ones (never touched by the DMA)
template <bool SPLIT> constexpr int h2_stage() { return (SPLIT ? 2 : 1) * H2_KT + H2_VT + 64; }      // 9,280 / 13,888 B

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

static __device__ __forceinline__ int h2_pi23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
template <int N>
static __device__ __forceinline__ void h2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SPLIT (the selective-split precision mode, f16s): q and k arrive as 2-term f16 splits (hi plane + lo plane qk_lo_off elements
// behind it, lo = f16((v - hi) * 2^11), common.h) and the scores take three products - K_hi Q_hi^T into the main accumulator,
// K_lo Q_hi^T + K_hi Q_lo^T into a second one that is folded in x 2^-11 - because the f16 rounding of q and k is what carries
// the attention's share of the mask error (tools/precision_sim.py: 6.8e-4 of max|ref| against 1.9e-4 for p and v, which stay
// f16).  The K_lo image rides in the same ring stage; the output is written as hi + lo (o_lo_off) for the split projection.
template <bool SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 2 : 3) void hiera_attn_v2_kernel(const HieraAttnParams p) {
  constexpr int STAGE = h2_stage<SPLIT>();
  constexpr int VOFF = (SPLIT ? 2 : 1) * H2_KT;        // V^T image inside a stage
  constexpr int PPT = SPLIT ? 4 : 3;                   // DMA pieces per wave and tile
  __shared__ __attribute__((aligned(16))) char smem[H2_NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int qtiles = p.GQ / 32;                       // multiple of 4
  // the 4 waves: 4 consecutive query tiles of one (group, head); XCD-aware order as in hiera_attn_kernel: the 2 x heads workgroups of a
  // stage-3 window / the 32 x heads of a global block's frame share an L2 (PMC before: 2.3 - 2.8x the algorithmic bytes)
  const int task = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  const int qt = task % qtiles;
  const int gh = task / qtiles;
  const int head = gh % p.heads, grp = gh / p.heads;
  const int n = p.GK / 32;                            // key tiles (every query sees the whole group)

  // ---- Q fragments (B operand): lane holds Q[q = fr][d = 16 s + 8 fh + j]; pre-scaled by 72^-0.5 * log2(e)
  const size_t qrow = (size_t)grp * p.GQ + qt * 32 + fr;
  half8 qf[5], ql[SPLIT ? 5 : 1];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    if (s == 4 && fh == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (half_t)0.f;
      if (SPLIT) ql[SPLIT ? s : 0] = qf[s];
    } else {
      const half_t* qp = p.q + qrow * p.ldq + head * HD + s * 16 + fh * 8;
      qf[s] = *reinterpret_cast<const half8*>(qp);
      if (SPLIT) ql[SPLIT ? s : 0] = *reinterpret_cast<const half8*>(qp + p.qk_lo_off);
    }
  }

  // ---- LDS-DMA sources.  288 chunks per image and tile = 4.5 wave-instructions: wave w moves chunks [64 w, 64 w + 64) of
  // every image, and one half-wave piece: chunks 256..287 of K (waves 0, 2) or of V^T (waves 1, 3) - written twice with
  // identical bytes, so that every wave has exactly PPT pieces per tile in flight (SPLIT: K_hi, K_lo, V^T, K_hi again).
  const half_t* kbase = p.k + ((size_t)grp * p.GK) * p.ldk + head * HD;
  const half_t* vbase = p.vT + (size_t)head * HD * p.ldvT + (size_t)grp * p.GK;
  auto k_off = [&](int c) { return (c / 9) * p.ldk + (c % 9) * 8; };                              // halfs from kbase (+ k0 rows)
  auto v_off = [&](int c) { const int d = c >> 2, pc = c & 3; return d * p.ldvT + ((pc ^ ((d >> 2) & 3)) << 3); };
  const int c_main = wave * 64 + lane, c_tail = 256 + (lane & 31);
  const int ko_main = k_off(c_main), vo_main = v_off(c_main);
  const int tail_img = SPLIT ? (wave == 3 ? 0 : wave) : (wave & 1);         // SPLIT: 0 K_hi, 1 K_lo, 2 V^T; else 0 K, 1 V^T
  const bool tail_is_v = tail_img == (SPLIT ? 2 : 1);
  const size_t o_tail = tail_is_v ? (size_t)v_off(c_tail) : (size_t)k_off(c_tail) + (SPLIT && tail_img == 1 ? p.qk_lo_off : 0);
  const int tail_dst = tail_img * H2_KT + 4096;
  auto issue = [&](int i) {
    char* sb = smem + (i % H2_NST) * STAGE;
    const half_t* kb = kbase + (size_t)i * 32 * p.ldk;
    const half_t* vb = vbase + i * 32;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kb + ko_main), (lds_ptr_t)(sb + wave * 1024), 16, 0, 0);
    if (SPLIT) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kb + p.qk_lo_off + ko_main), (lds_ptr_t)(sb + H2_KT + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(vb + vo_main), (lds_ptr_t)(sb + VOFF + wave * 1024), 16, 0, 0);
    if (lane < 32)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)((tail_is_v ? vb : kb) + o_tail), (lds_ptr_t)(sb + tail_dst), 16, 0, 0);
  };

  // ---- fragment reads
  const int krow = h2_pi23(fr);
  const int k_base = krow * 144 + fh * 16, k_base4 = krow * 144 + 128;
  struct KF { half8 f[5]; half8 l[SPLIT ? 5 : 1]; };
  auto read_k = [&](int i) {
    const char* sK = smem + (i % H2_NST) * STAGE;
    KF k;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) k.f[ks] = *reinterpret_cast<const half8*>(sK + k_base + ks * 32);
    k.f[4] = *reinterpret_cast<const half8*>(sK + k_base4);
    if (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) k.l[SPLIT ? ks : 0] = *reinterpret_cast<const half8*>(sK + H2_KT + k_base + ks * 32);
      k.l[SPLIT ? 4 : 0] = *reinterpret_cast<const half8*>(sK + H2_KT + k_base4);
    }
    return k;
  };
  const int vsw = (fr >> 2) & 3;
  const int v_row2 = min(64 + fr, HD);                // row tile 2: row 72 = the ones row (l accumulates there), rows past it alias it
  const int v_sw2 = (v_row2 >> 2) & 3;
  struct VF { half8 f[6]; };
  auto read_v = [&](int i) {
    const char* sV = smem + (i % H2_NST) * STAGE + VOFF;
    VF v;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      v.f[2 * t] = *reinterpret_cast<const half8*>(sV + (t * 32 + fr) * 64 + (((0 + fh) ^ vsw) << 4));
      v.f[2 * t + 1] = *reinterpret_cast<const half8*>(sV + (t * 32 + fr) * 64 + (((2 + fh) ^ vsw) << 4));
    }
    v.f[4] = *reinterpret_cast<const half8*>(sV + v_row2 * 64 + (((0 + fh) ^ v_sw2) << 4));
    v.f[5] = *reinterpret_cast<const half8*>(sV + v_row2 * 64 + (((2 + fh) ^ v_sw2) << 4));
    return v;
  };
  auto qk = [&](const KF& k, float init) {
    f32x16 sa;                     // one chain: a single accumulation chain of this MFMA issues at full rate
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = init;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) sa = mfma32(k.f[ks], qf[ks], sa);
    if (SPLIT) {                   // cross terms in their own chain, folded in x 2^-11 (lo * lo = 2^-22 is dropped)
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        sc = mfma32(k.l[SPLIT ? ks : 0], qf[ks], sc);
        sc = mfma32(k.f[ks], ql[SPLIT ? ks : 0], sc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = fmaf(sc[r], SPLIT_INV, sa[r]);
    }
    return sa;
  };
  auto rowmax = [&](const f32x16& s) {
    float t = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) t = fmaxf(t, s[r]);
    return fmaxf(t, __shfl_xor(t, 32, 64));
  };

  f32x16 o[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  constexpr float RESCALE_THR = 8.f;
  // ones rows (one per ring stage), before the first barrier
  for (int j = tid; j < H2_NST * 32; j += 256)
    *reinterpret_cast<half_t*>(smem + (j >> 5) * STAGE + VOFF + H2_VT + (j & 31) * 2) = (half_t)1.f;

  issue(0);
  if (n > 1) issue(1);
  if (n > 2) issue(2);
  if (n > 2) h2_wait_vm<2 * PPT>();
  else if (n > 1) h2_wait_vm<PPT>();
  else h2_wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  // scores are kept RELATIVE to the reference maximum: s' = s - m_ref (first tile: subtract its own row maximum)
  f32x16 s;
  float m_ref;
  {
    const KF k0 = read_k(0);
    __builtin_amdgcn_sched_barrier(0);
    s = qk(k0, 0.f);
    m_ref = rowmax(s);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] -= m_ref;
  }
  float tmax = 0.f;                                      // row maximum of s', relative to m_ref
  int i = 0;
  for (;;) {
    {                                                    // raise the reference maximum by the excess (first entry: 0)
      const float delta = fmaxf(tmax, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;    // row 72 of tile 2 is l: rescaled with the rest
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] -= delta;
      m_ref += delta;
    }
    bool done = false;
#pragma nounroll
    for (;;) {
      // tile i+1 landed (tile i+2 may stay in flight); every wave is past tile i-1 -> its ring stage is free
      if (i + 2 < n) h2_wait_vm<PPT>();
      else h2_wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (i + 3 < n) issue(i + 3);
      const KF kn = read_k(i + 1);                       // past the last tile: stale ring data, result unused
      const VF vf = read_v(i);
      __builtin_amdgcn_sched_barrier(0);
      const f32x16 s_next = qk(kn, -m_ref);
      half8 pf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) pf[r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(s[r]);
#pragma unroll
      for (int t = 0; t < 3; ++t) o[t] = mfma32(vf.f[2 * t], pf[0], o[t]);
#pragma unroll
      for (int t = 0; t < 3; ++t) o[t] = mfma32(vf.f[2 * t + 1], pf[1], o[t]);
      s = s_next;
      tmax = rowmax(s);
      ++i;
      if (i >= n) { done = true; break; }
      if (__any(tmax > RESCALE_THR)) break;
    }
    if (done) break;
  }

  // l = O^T[72][q]: accumulator row 8 of row tile 2 = register 4 of the lower lane half
  const float l_run = __shfl(o[2][4], fr, 64);
  const float inv = 1.f / l_run;
  half_t* orow = p.o + qrow * p.ldo + head * HD;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = t * 32 + 8 * g + 4 * fh;
      if (d < HD) {
        half4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = o[t][4 * g + e] * inv;
          h[e] = (half_t)v;
          if (SPLIT) l[e] = split_lo(v, h[e]);
        }
        *reinterpret_cast<half4*>(orow + d) = h;
        if (SPLIT && p.o_lo_off) *reinterpret_cast<half4*>(orow + p.o_lo_off + d) = l;
      }
    }
  }
}
}  // namespace

// the instantiation hiera_attn_launch picks for these parameters, under the name rocprofv3 prints (profiling accumulators)
const char* hiera_attn_kernel_name(const HieraAttnParams& p) {
  const int qtiles = p.GQ / 32;
  const bool mask = !(p.wq >= p.GQ && p.wk >= p.GK);
  static const bool use_v1 = getenv("SAM2MI_HATTN_V1") != nullptr;
  const bool split = p.qk_lo_off != 0;
  if (qtiles % 4 == 0 && !mask && (split || !use_v1) && (p.ldvT & 7) == 0) return split ? "hiera_attn_v2_kernel<true>" : "hiera_attn_v2_kernel<false>";
  if (qtiles % 4 == 0) return mask ? (split ? "hiera_attn_kernel<true, true, true>" : "hiera_attn_kernel<true, true, false>")
                                   : (split ? "hiera_attn_kernel<true, false, true>" : "hiera_attn_kernel<true, false, false>");
  return mask ? (split ? "hiera_attn_kernel<false, true, true>" : "hiera_attn_kernel<false, true, false>")
              : (split ? "hiera_attn_kernel<false, false, true>" : "hiera_attn_kernel<false, false, false>");
}

hipError_t hiera_attn_launch(const HieraAttnParams& p, hipStream_t stream) {
  if (p.GQ % 32 || p.GK % 32 || p.num_groups <= 0) return hipErrorInvalidValue;
  if ((p.ldq & 7) || (p.ldk & 7) || (p.ldvT & 3) || (p.ldo & 3)) return hipErrorInvalidValue;
  if (p.wq < 32 && (32 % p.wq)) return hipErrorInvalidValue;
  if (p.wq >= 32 && (p.wq % 32 || p.wk % 32)) return hipErrorInvalidValue;
  if (p.wq < 32 && ((32 / p.wq) * p.wk) % 32) return hipErrorInvalidValue;
  const int qtiles = p.GQ / 32;
  const int total = p.num_groups * p.heads * qtiles;
  const int blocks = (total + 3) / 4;
  const bool mask = !(p.wq >= p.GQ && p.wk >= p.GK);
  static const bool use_v1 = getenv("SAM2MI_HATTN_V1") != nullptr;     // A/B switch: the register-staged kernel
  if (p.qk_lo_off) {                // split q / k (f16s precision mode)
    if ((p.qk_lo_off & 7) || (p.o_lo_off & 3)) return hipErrorInvalidValue;
    if (qtiles % 4 == 0 && !mask && (p.ldvT & 7) == 0) hiera_attn_v2_kernel<true><<<dim3(total / 4), dim3(256), 0, stream>>>(p);
    else if (qtiles % 4 == 0) {
      if (mask) hiera_attn_kernel<true, true, true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
      else hiera_attn_kernel<true, false, true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
    } else {
      if (mask) hiera_attn_kernel<false, true, true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
      else hiera_attn_kernel<false, false, true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
    }
  } else if (qtiles % 4 == 0 && !mask && !use_v1 && (p.ldvT & 7) == 0) {
    hiera_attn_v2_kernel<false><<<dim3(total / 4), dim3(256), 0, stream>>>(p);
  } else if (qtiles % 4 == 0) {
    if (mask) hiera_attn_kernel<true, true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
    else hiera_attn_kernel<true, false><<<dim3(blocks), dim3(256), 0, stream>>>(p);
  } else {
    if (mask) hiera_attn_kernel<false, true><<<dim3(blocks), dim3(256), 0, stream>>>(p);
    else hiera_attn_kernel<false, false><<<dim3(blocks), dim3(256), 0, stream>>>(p);
  }
  return hipGetLastError();
}

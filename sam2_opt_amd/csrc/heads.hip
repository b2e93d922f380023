// Prompt encoder, mask-decoder glue and memory-bank assembly kernels.  Everything here is tiny and
// latency-bound: fp32 SIMT, one wave per output row/column, no host synchronisation (decisions such
// as the IoU argmax or the object-score gate stay on the device).
#include "kernels.h"

namespace {
constexpr float NO_OBJ_SCORE = -1024.f;     // modeling/sam2_base_official.py:21
constexpr float TWO_PI = 6.283185307179586f;

// y[t, n] for all t: one wave per output column n, lanes over K (16-B loads); blockIdx.y picks one of up to 4
// independent linears so that e.g. the q/k/v projections of a token attention go out as one launch
__global__ __launch_bounds__(256) void small_linear_kernel(const SmallLinBatch B) {
  const SmallLin& D = B.d[blockIdx.y];
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= D.N) return;
  const float* wr = D.W + (size_t)n * D.K;
  const bool vec = ((D.K | D.ldx) & 3) == 0 && (((uintptr_t)D.x | (uintptr_t)D.W | (uintptr_t)D.x2) & 15) == 0;
  for (int t0 = 0; t0 < D.T; t0 += 8) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (vec) {
      // 4 k-steps of 256 per trip, every load unconditional (rows past T re-read row T-1, k-steps past K read column 0 against a zero
      // weight): with `if (t < T)` around each load the compiler kept one wait per token and k-step - 64 dependent round trips for
      // K = 2048, the second linear of the token MLP: 23 us for 2 MB of weights
      for (int k0 = lane * 4; k0 < D.K; k0 += 1024) {
        f32x4 wv[4];
        int ko[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = k0 + 256 * u < D.K;
          ko[u] = ok ? k0 + 256 * u : 0;
          wv[u] = *reinterpret_cast<const f32x4*>(wr + ko[u]);
          if (!ok) wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 xv[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const size_t ro = (size_t)min(t0 + j, D.T - 1) * D.ldx;
#pragma unroll
          for (int u = 0; u < 4; ++u) xv[j][u] = *reinterpret_cast<const f32x4*>(D.x + ro + ko[u]);
          if (D.x2) {                                               // wave-uniform
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const f32x4 e = *reinterpret_cast<const f32x4*>(D.x2 + ro + ko[u]);
              xv[j][u][0] += e[0]; xv[j][u][1] += e[1]; xv[j][u][2] += e[2]; xv[j][u][3] += e[3];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            acc[j] += wv[u][0] * xv[j][u][0] + wv[u][1] * xv[j][u][1] + wv[u][2] * xv[j][u][2] + wv[u][3] * xv[j][u][3];
      }
    } else {
      for (int k = lane; k < D.K; k += 64) {
        const float wv = wr[k];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (t0 + j < D.T) acc[j] += wv * (D.x[(size_t)(t0 + j) * D.ldx + k] + (D.x2 ? D.x2[(size_t)(t0 + j) * D.ldx + k] : 0.f));
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = wave_sum(acc[j]);
      if (lane == 0 && t0 + j < D.T) {
        float o = v + (D.b ? D.b[n] : 0.f);
        if (D.act == 2) o = fmaxf(o, 0.f);
        else if (D.act == 3) o = 1.f / (1.f + expf(-o));
        if (D.res) o += D.res[(size_t)(t0 + j) * D.ldres + n];
        D.y[(size_t)(t0 + j) * D.ldy + n] = o;
      }
    }
  }
}

// 3-layer MLP 256 -> 256 -> 256 -> n_out (ReLU between, optional sigmoid at the end) on ONE row per group, one
// 1024-thread workgroup per group (blockIdx.x): the hyper-network MLPs, the IoU / object-score heads and the
// object-pointer projection (sam/mask_decoder.py:290-303, sam2_base_official.py:474) are 3 dependent mat-vecs
// each - as separate launches they cost 3 launch latencies apiece, here one launch covers all of them.
__global__ __launch_bounds__(1024) void mlp3_kernel(const Mlp3Batch B) {
  const Mlp3Group& G = B.g[blockIdx.x];
  const float* gx = G.x + (size_t)blockIdx.y * G.x_rep_stride;         // blockIdx.y: repetition (one per prompt / object)
  float* gy = G.y + (size_t)blockIdx.y * G.y_rep_stride;
  __shared__ __attribute__((aligned(16))) float buf[2][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // 16 waves
  if (tid < 256) buf[0][tid] = gx[tid];
#pragma unroll 1
  for (int layer = 0; layer < 3; ++layer) {
    const float* W = G.W[layer];
    const float* b = G.b[layer];
    const int N = layer == 2 ? G.n_out : 256;
    // each wave: outputs wave, wave+16, ...; ALL 16 rows of W of this wave are requested - unconditionally, rows past N re-read row
    // N-1 - before the barrier that publishes the layer's input: one memory round trip per layer
    f32x4 w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = *reinterpret_cast<const f32x4*>(W + (size_t)min(wave + 16 * u, N - 1) * 256 + lane * 4);
    __syncthreads();
    const float* in = buf[layer & 1];
    const f32x4 xv = *reinterpret_cast<const f32x4*>(in + lane * 4);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int o = wave + 16 * u;
      float v = wave_sum(w[u][0] * xv[0] + w[u][1] * xv[1] + w[u][2] * xv[2] + w[u][3] * xv[3]);
      if (lane == 0 && o < N) {
        v += b ? b[o] : 0.f;
        if (layer < 2) buf[(layer + 1) & 1][o] = fmaxf(v, 0.f);
        else gy[o] = G.sigmoid_out ? 1.f / (1.f + expf(-v)) : v;
      }
    }
  }
}

// PromptEncoder._embed_points (+ pad point), prompt_encoder.py:124-166 / position_encoding.py:148-175
__global__ void point_embed_kernel(const float* __restrict__ pts, const int* __restrict__ labels, int Np,
                                   const float* __restrict__ gauss, const float* __restrict__ pe4,
                                   const float* __restrict__ nap, float image_size, float* __restrict__ out) {
  const int i = blockIdx.x;             // point index (Np = pad point)
  const int c = threadIdx.x;            // 0..255
  float px = 0.f, py = 0.f;
  int lab = -1;
  if (i < Np) {
    px = pts[2 * i] + 0.5f;
    py = pts[2 * i + 1] + 0.5f;
    lab = labels[i];
  }
  const float cx = 2.f * (px / image_size) - 1.f, cy = 2.f * (py / image_size) - 1.f;
  const int f = c & 127;
  const float ang = TWO_PI * (cx * gauss[f] + cy * gauss[128 + f]);
  float v = (c < 128) ? sinf(ang) : cosf(ang);
  if (lab == -1) v = nap[c];
  else if (lab >= 0 && lab < 4) v += pe4[lab * 256 + c];
  out[(size_t)i * 256 + c] = v;
}

// PromptEncoder.get_dense_pe: token (y, x) -> coords ((x+0.5)/S, (y+0.5)/S)
__global__ void dense_pe_kernel(const float* __restrict__ gauss, int S, float* __restrict__ out) {
  const int t = blockIdx.x, c = threadIdx.x;
  const int y = t / S, x = t % S;
  const float cx = 2.f * ((x + 0.5f) / S) - 1.f, cy = 2.f * ((y + 0.5f) / S) - 1.f;
  const int f = c & 127;
  const float ang = TWO_PI * (cx * gauss[f] + cy * gauss[128 + f]);
  out[(size_t)t * 256 + c] = (c < 128) ? sinf(ang) : cosf(ang);
}

// one wave per output pixel (2Hin x 2Hin grid), C <= 64 channels (lane = channel)
__global__ __launch_bounds__(256) void upscale_glue_kernel(const float* __restrict__ g, int Hin, int C, const float* __restrict__ bias,
                                                           const float* __restrict__ hr, const float* __restrict__ lnw,
                                                           const float* __restrict__ lnb, half_t* __restrict__ out16,
                                                           size_t hr_bstride, size_t lo_off) {
  const int Hout = 2 * Hin;
  g += (size_t)blockIdx.y * Hin * Hin * 4 * C;           // batch: contiguous inputs / outputs, hr shared when its stride is 0
  hr += blockIdx.y * hr_bstride;
  out16 += (size_t)blockIdx.y * Hout * Hout * C;
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pix >= Hout * Hout) return;
  const int Y = pix / Hout, X = pix % Hout;
  const int src = (Y >> 1) * Hin + (X >> 1), pos = (Y & 1) * 2 + (X & 1);
  const bool on = lane < C;
  float v = on ? g[(size_t)src * 4 * C + pos * C + lane] + bias[lane] + hr[(size_t)pix * C + lane] : 0.f;
  if (lnw) {
    const float mean = wave_sum(v) / C;
    const float d = on ? v - mean : 0.f;
    const float rstd = 1.f / sqrtf(wave_sum(d * d) / C + 1e-6f);
    v = on ? d * rstd * lnw[lane] + lnb[lane] : 0.f;
  }
  if (on) store_h1(out16 + (size_t)pix * C + lane, lo_off, gelu_erf(v));
}

// stability score inputs of MaskDecoder._get_stability_scores (mask_decoder.py:332-344) for mask 0
__global__ void stability_count_kernel(const float* __restrict__ mask0, float delta, int* counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float v = i < 65536 ? mask0[i] : -1e30f;
  const unsigned long long bi = __ballot(v > delta), bu = __ballot(v > -delta);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], __popcll(bi));
    atomicAdd(&counts[1], __popcll(bu));
  }
}

// SAM2Base._forward_sam_heads :440-484 + MaskDecoder.forward :151-169.
//  multimask: candidates 1..3, best by IoU, its token.  Otherwise candidate 0 unless its stability score
//  (counts from stability_count_kernel) is below `stab_thresh`, then the best of 1..3 (token stays token 0).
__global__ void select_mask_kernel(const float* __restrict__ masks, const float* __restrict__ iou, const float* __restrict__ obj,
                                   const float* __restrict__ tokens, int multimask, const int* __restrict__ counts,
                                   float stab_thresh, float* low_multi, float* low_sel, float* tok_sel, int* best_idx,
                                   float* iou_out, float* low_sel2) {
  const bool appearing = obj[0] > 0.f;
  int bm = 1;
  float bv = iou[1];
  if (iou[2] > bv) { bv = iou[2]; bm = 2; }
  if (iou[3] > bv) { bv = iou[3]; bm = 3; }
  int best, tok_idx;
  if (multimask) {
    best = bm;
    tok_idx = bm;
  } else {
    float stab = 1.f;
    if (counts && counts[1] > 0) stab = (float)counts[0] / (float)counts[1];
    best = (!counts || stab >= stab_thresh) ? 0 : bm;
    tok_idx = 0;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 65536) {
    const float sel = appearing ? masks[(size_t)best * 65536 + i] : NO_OBJ_SCORE;
    low_sel[i] = sel;
    if (low_sel2) low_sel2[i] = sel;                   // the caller's copy of the selected mask
    if (low_multi) {
      if (multimask) {
#pragma unroll
        for (int k = 0; k < 3; ++k) low_multi[(size_t)k * 65536 + i] = appearing ? masks[(size_t)(k + 1) * 65536 + i] : NO_OBJ_SCORE;
      } else {
        low_multi[i] = appearing ? masks[(size_t)best * 65536 + i] : NO_OBJ_SCORE;
      }
    }
  }
  if (i < 256) tok_sel[i] = tokens[tok_idx * 256 + i];
  if (i == 0) {
    best_idx[0] = multimask ? best - 1 : best;
    if (iou_out) {
      if (multimask) { iou_out[0] = iou[1]; iou_out[1] = iou[2]; iou_out[2] = iou[3]; }
      else iou_out[0] = iou[best];
    }
  }
}

// also hands the pointer / the score on to up to three more places (bank slot score, caller's outputs): no copy launches behind it
__global__ void gate_obj_ptr_kernel(float* ptr, const float* __restrict__ no_obj_ptr, const float* __restrict__ obj, int C, float* ptr_out,
                                    float* score_out, float* score_out2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = obj[0];
  const float lam = sc > 0.f ? 1.f : 0.f;
  const float v = lam * ptr[c] + (1.f - lam) * no_obj_ptr[c];
  ptr[c] = v;
  if (ptr_out) ptr_out[c] = v;
  if (c == 0) {
    if (score_out) score_out[0] = sc;
    if (score_out2) score_out2[0] = sc;
  }
}

__global__ void mem_assemble_kernel(const MemAssembleParams p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nframe = (size_t)p.L * 4096 * 64;
  const size_t total = nframe + (size_t)p.P * 64;
  if (i >= total) return;
  float f, ps;
  if (i < nframe) {
    const int s = (int)(i / (4096 * 64));
    const int r = (int)(i % (4096 * 64));
    f = p.feat[s][r];
    ps = p.pos[r] + p.tpos[s][r & 63];
  } else {
    const size_t r = i - nframe;
    f = p.ptr_tok[r];
    ps = p.ptr_pos[r];
  }
  store_h1(p.kin + i, p.lo_off, f + ps);
  store_h1(p.vin + i, p.lo_off, f);
  if (p.mem32) p.mem32[i] = f;
  if (p.mempos32) p.mempos32[i] = ps;
}

// one block (64 threads = 64 output features) per pointer
__global__ void ptr_tokens_kernel(const PtrTokParams p) {
  __shared__ float pe[256];
  const int i = blockIdx.x, t = threadIdx.x;
  const float pos = p.dt[i] / p.tmax;
  for (int j = t; j < 128; j += 64) {
    const float dim_t = powf(10000.f, 2.f * (float)(j / 2) / 128.f);
    const float e = pos / dim_t;
    pe[j] = sinf(e);
    pe[128 + j] = cosf(e);
  }
  __syncthreads();
  float acc = p.bt[t];
  const float* wr = p.Wt + (size_t)t * 256;
  for (int k = 0; k < 256; ++k) acc += wr[k] * pe[k];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p.pos[((size_t)4 * i + j) * 64 + t] = acc;
    p.tok[((size_t)4 * i + j) * 64 + t] = p.ptr[i][64 * j + t];
  }
}
}  // namespace

hipError_t small_linear_batch_launch(const SmallLinBatch& B, hipStream_t s) {
  if (B.n < 1 || B.n > 4) return hipErrorInvalidValue;
  int maxN = 0;
  for (int i = 0; i < B.n; ++i) maxN = B.d[i].N > maxN ? B.d[i].N : maxN;
  small_linear_kernel<<<dim3((maxN + 3) / 4, B.n), dim3(256), 0, s>>>(B);
  return hipGetLastError();
}
hipError_t small_linear_launch(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, const float* res,
                               int ldres, int T, int N, int K, int act, hipStream_t s) {
  SmallLinBatch B;
  B.n = 1;
  B.d[0] = SmallLin{x, W, b, y, res, ldx, ldy, ldres, T, N, K, act};
  return small_linear_batch_launch(B, s);
}
hipError_t mlp3_launch(const Mlp3Batch& B, hipStream_t s) {
  if (B.n < 1 || B.n > 8 || B.reps < 1 || B.reps > 65535) return hipErrorInvalidValue;
  for (int i = 0; i < B.n; ++i)
    if (B.g[i].n_out < 1 || B.g[i].n_out > 256) return hipErrorInvalidValue;
  mlp3_kernel<<<dim3(B.n, B.reps), dim3(1024), 0, s>>>(B);
  return hipGetLastError();
}
hipError_t point_embed_launch(const float* pts, const int* labels, int Np, const float* gauss, const float* point_emb4,
                              const float* not_a_point, float image_size, float* out, hipStream_t s) {
  point_embed_kernel<<<dim3(Np + 1), dim3(256), 0, s>>>(pts, labels, Np, gauss, point_emb4, not_a_point, image_size, out);
  return hipGetLastError();
}
hipError_t dense_pe_launch(const float* gauss, int S, float* out, hipStream_t s) {
  dense_pe_kernel<<<dim3(S * S), dim3(256), 0, s>>>(gauss, S, out);
  return hipGetLastError();
}
hipError_t upscale_glue_launch(const float* g, int Hin, int C, const float* bias, const float* hr, const float* lnw,
                               const float* lnb, half_t* out16, int batch, size_t hr_bstride, hipStream_t s, size_t lo_off) {
  if (C > 64 || batch < 1 || batch > 65535) return hipErrorInvalidValue;
  const int n = 4 * Hin * Hin;
  upscale_glue_kernel<<<dim3((n + 3) / 4, batch), dim3(256), 0, s>>>(g, Hin, C, bias, hr, lnw, lnb, out16, hr_bstride, lo_off);
  return hipGetLastError();
}
hipError_t select_mask_launch(const float* masks, const float* iou, const float* obj, const float* tokens, int multimask,
                              int* stab_counts, float stab_delta, float stab_thresh, float* low_multi, float* low_sel,
                              float* tok_sel, int* best_idx, float* iou_out, hipStream_t s, float* low_sel2) {
  if (!multimask && stab_counts) {
    hipError_t e = hipMemsetAsync(stab_counts, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return e;
    stability_count_kernel<<<dim3(256), dim3(256), 0, s>>>(masks, stab_delta, stab_counts);
  }
  select_mask_kernel<<<dim3(256), dim3(256), 0, s>>>(masks, iou, obj, tokens, multimask, multimask ? nullptr : stab_counts,
                                                     stab_thresh, low_multi, low_sel, tok_sel, best_idx, iou_out, low_sel2);
  return hipGetLastError();
}
hipError_t gate_obj_ptr_launch(float* ptr, const float* no_obj_ptr, const float* obj, int C, hipStream_t s, float* ptr_out, float* score_out,
                               float* score_out2) {
  gate_obj_ptr_kernel<<<dim3((C + 255) / 256), dim3(256), 0, s>>>(ptr, no_obj_ptr, obj, C, ptr_out, score_out, score_out2);
  return hipGetLastError();
}
hipError_t mem_assemble_launch(const MemAssembleParams& p, hipStream_t s) {
  const size_t total = (size_t)p.L * 4096 * 64 + (size_t)p.P * 64;
  mem_assemble_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(p);
  return hipGetLastError();
}
hipError_t ptr_tokens_launch(const PtrTokParams& p, hipStream_t s) {
  if (p.n <= 0) return hipSuccess;
  ptr_tokens_kernel<<<dim3(p.n), dim3(64), 0, s>>>(p);
  return hipGetLastError();
}

// afm.hip -- AFMLayer (attentional pooling of the pairwise products, reference interaction.py:251-325) on gfx950.
//
//   bi_k = e_i (.) e_j  for the P = F(F-1)/2 pairs (i < j, itertools.combinations order)
//   t_k  = relu(bi_k W + b)          W [D, A], b [A]
//   s_k  = t_k . h                   h [A]          a = softmax_k(s)
//   out  = sum_k a_k bi_k            [D]            afm = out . p          p [D]
// The reference materialises p, q = cat of 325 slices each ([B, 325, 16] x 3), two tensordots and a softmax: ~15
// launches and ~60 MB of traffic per call at the Criteo shape.  Here ONE wave owns a sample: its F*D embedding row,
// the pair scores and (backward) the per-pair gradient rows live in LDS, nothing but E, g and the results touches
// HBM.  The backward recomputes the forward (cheaper than saving [B, P] scores), is free of atomics, and sums the
// parameter gradients in a fixed order: per-lane partial sums over the workgroup's samples -> one partial row per
// workgroup -> k_afm_reduce adds the rows in workgroup order.  Limits: D <= 64, A <= 32, F <= 64 and the LDS budget.
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kW = 64;   // one wave per workgroup

struct AfmArgs {
  const float* E;     // [B, lde]: fields first
  int64_t lde;
  const float* W;     // [D, A]
  const float* bias;  // [A]
  const float* h;     // [A]
  const float* p;     // [D]
  int B, F, D, A, P;
  float* y;           // fwd: [B]
  const float* gy;    // bwd: [B]
  float* gE;          // bwd: [B, ldge] (only the F*D field columns are written)
  int64_t ldge;
  float* part;        // bwd: [n_wg][D*AP + 2*AP + D] partial parameter gradients
};

__device__ __forceinline__ int pair_index(int f, int j, int F) {   // f < j
  return f * F - f * (f + 1) / 2 + (j - f - 1);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
  return v;
}

// LDS layout (floats): Ws [D*AP] | bs [AP] | hs [AP] | ps [D] | pairs (int) [P] | es [F*D] | sc [P] | outs [D]
//                      | (bwd) gout [D] | gpre [P*AP] | gbi [P*D]
template <int AP, bool BWD>
__global__ __launch_bounds__(kW) void k_afm(AfmArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int F = a.F, D = a.D, P = a.P, lane = threadIdx.x;
  float* Ws = smem;
  float* bs = Ws + D * AP;
  float* hs = bs + AP;
  float* ps = hs + AP;
  int* pairs = reinterpret_cast<int*>(ps + D);
  float* es = reinterpret_cast<float*>(pairs + P);
  float* sc = es + F * D;
  float* outs = sc + P;
  float* gout = outs + D;
  float* gpre = gout + D;
  float* gbi = gpre + (BWD ? P * AP : 0);

  for (int e = lane; e < D * AP; e += kW) {
    const int d = e / AP, q = e - d * AP;
    Ws[e] = q < a.A ? ldg_f32(a.W + d * a.A + q) : 0.f;
  }
  for (int q = lane; q < AP; q += kW) {
    bs[q] = q < a.A ? ldg_f32(a.bias + q) : 0.f;
    hs[q] = q < a.A ? ldg_f32(a.h + q) : 0.f;
  }
  for (int d = lane; d < D; d += kW) ps[d] = ldg_f32(a.p + d);
  for (int k = lane; k < P; k += kW) {   // k -> (i, j), combinations order
    int i = 0, kk = k;
    while (kk >= F - 1 - i) {
      kk -= F - 1 - i;
      ++i;
    }
    pairs[k] = (i << 8) | (i + 1 + kk);
  }

  // parameter-gradient partials of this workgroup (registers, summed over its samples in sample order)
  constexpr int NW = (64 * AP + kW - 1) / kW;   // gW elements per lane (D <= 64)
  float gWp[NW], gbp[AP], ghp[AP], gpp = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) gWp[i] = 0.f;
#pragma unroll
  for (int q = 0; q < AP; ++q) gbp[q] = ghp[q] = 0.f;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    {  // the sample's embedding row: unconditional loads, 8 in flight
      const float* src = a.E + static_cast<int64_t>(b) * a.lde;
      const int n = F * D;
      for (int e0 = lane; e0 < n; e0 += 8 * kW) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ldg_f32(src + (e0 + u * kW < n ? e0 + u * kW : 0));
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (e0 + u * kW < n) es[e0 + u * kW] = v[u];
      }
    }
    __syncthreads();
    // scores
    float mx = -3.0e38f;
    for (int k = lane; k < P; k += kW) {
      const int i = pairs[k] >> 8, j = pairs[k] & 255;
      float acc[AP];
#pragma unroll
      for (int q = 0; q < AP; ++q) acc[q] = bs[q];
      for (int d = 0; d < D; ++d) {
        const float bi = es[i * D + d] * es[j * D + d];
#pragma unroll
        for (int q = 0; q < AP; ++q) acc[q] += bi * Ws[d * AP + q];
      }
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < AP; ++q) s += fmaxf(acc[q], 0.f) * hs[q];
      sc[k] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float z = 0.f;
    for (int k = lane; k < P; k += kW) {
      const float ex = expf(sc[k] - mx);
      sc[k] = ex;
      z += ex;
    }
    z = wave_sum(z);
    const float rz = 1.f / z;
    for (int k = lane; k < P; k += kW) sc[k] *= rz;     // normalised attention scores a_k
    __syncthreads();
    // out[d] = sum_k a_k bi_k[d]: lane = (k-group, d) with DP = pow2 >= D lanes per group
    int DP = 1;
    while (DP < D) DP <<= 1;
    {
      const int d = lane & (DP - 1), kg = lane / DP, ng = kW / DP;
      float o = 0.f;
      if (d < D)
        for (int k = kg; k < P; k += ng) {
          const int i = pairs[k] >> 8, j = pairs[k] & 255;
          o += sc[k] * (es[i * D + d] * es[j * D + d]);
        }
      for (int m = DP; m < kW; m <<= 1) o += __shfl_xor(o, m, kWave);
      if (lane < D) outs[lane] = o;
    }
    __syncthreads();
    if (!BWD) {
      float t = lane < D ? outs[lane] * ps[lane] : 0.f;
      t = wave_sum(t);
      if (lane == 0) a.y[b] = t;
      continue;
    }
    // ---- backward --------------------------------------------------------------------------------------------
    const float g = ldg_f32(a.gy + b);
    if (lane < D) {
      gout[lane] = g * ps[lane];
      gpp += g * outs[lane];
    }
    __syncthreads();
    // g_a_k = gout . bi_k ; dot = sum_k a_k g_a_k
    float dot = 0.f;
    for (int k = lane; k < P; k += kW) {
      const int i = pairs[k] >> 8, j = pairs[k] & 255;
      float ga = 0.f;
      for (int d = 0; d < D; ++d) ga += gout[d] * (es[i * D + d] * es[j * D + d]);
      gbi[k * D] = ga;                 // parked in the pair's gradient row (overwritten below by the same lane)
      dot += sc[k] * ga;
    }
    dot = wave_sum(dot);
    for (int k = lane; k < P; k += kW) {
      const int i = pairs[k] >> 8, j = pairs[k] & 255;
      const float ak = sc[k];
      const float gs = ak * (gbi[k * D] - dot);          // softmax backward
      float acc[AP];
#pragma unroll
      for (int q = 0; q < AP; ++q) acc[q] = bs[q];
      for (int d = 0; d < D; ++d) {
        const float bi = es[i * D + d] * es[j * D + d];
#pragma unroll
        for (int q = 0; q < AP; ++q) acc[q] += bi * Ws[d * AP + q];
      }
      float gp[AP];
#pragma unroll
      for (int q = 0; q < AP; ++q) {
        const float t = fmaxf(acc[q], 0.f);
        ghp[q] += gs * t;
        gp[q] = acc[q] > 0.f ? gs * hs[q] : 0.f;
        gbp[q] += gp[q];
        gpre[k * AP + q] = gp[q];
      }
      for (int d = 0; d < D; ++d) {
        float v = ak * gout[d];
#pragma unroll
        for (int q = 0; q < AP; ++q) v += gp[q] * Ws[d * AP + q];
        gbi[k * D + d] = v;                              // d loss / d bi_k[d]
      }
    }
    __syncthreads();
    // gW[d, q] += sum_k bi_k[d] gpre[k, q]: lane owns elements idx = lane + 64 * n of the [D, AP] tile
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const int idx = lane + kW * n;
      if (idx < D * AP) {
        const int d = idx / AP, q = idx - d * AP;
        float s = 0.f;
        for (int k = 0; k < P; ++k) {
          const int i = pairs[k] >> 8, j = pairs[k] & 255;
          s += (es[i * D + d] * es[j * D + d]) * gpre[k * AP + q];
        }
        gWp[n] += s;
      }
    }
    // gE[f, d] = sum_{j != f} gbi[pair(f, j)][d] * e_j[d]: lane owns elements e = lane + 64 * n of the row
    {
      float* dst = a.gE + static_cast<int64_t>(b) * a.ldge;
      for (int e = lane; e < F * D; e += kW) {
        const int f = e / D, d = e - f * D;
        float s = 0.f;
        for (int j = 0; j < F; ++j) {
          if (j == f) continue;
          const int k = f < j ? pair_index(f, j, F) : pair_index(j, f, F);
          s += gbi[k * D + d] * es[j * D + d];
        }
        stg_f32(dst + e, s);
      }
    }
  }
  if (BWD) {
    float* mine = a.part + static_cast<int64_t>(blockIdx.x) * (D * AP + 2 * AP + D);
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const int idx = lane + kW * n;
      if (idx < D * AP) mine[idx] = gWp[n];
    }
#pragma unroll
    for (int q = 0; q < AP; ++q) {
      const float sb = wave_sum(gbp[q]), sh = wave_sum(ghp[q]);
      if (lane == 0) {
        mine[D * AP + q] = sb;
        mine[D * AP + AP + q] = sh;
      }
    }
    if (lane < D) mine[D * AP + 2 * AP + lane] = gpp;
  }
}

// out[i] = sum_g part[g][i] in workgroup order; thread (o, sl) adds the groups sl, sl + 16, ..., slices added in order
__global__ __launch_bounds__(256) void k_afm_reduce(const float* __restrict__ part, int64_t stride, int groups, int D,
                                                    int A, int AP, float* __restrict__ gW, float* __restrict__ gb,
                                                    float* __restrict__ gh, float* __restrict__ gp) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + o;
  const int64_t ic = i < stride ? i : 0;
  float s = 0.f;
  for (int g0 = sl; g0 < groups; g0 += 16 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = g0 + 16 * u;
      v[u] = ldg_f32(part + static_cast<int64_t>(g < groups ? g : 0) * stride + ic);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (g0 + 16 * u < groups) s += v[u];
  }
  red[sl][o] = s;
  __syncthreads();
  if (sl == 0 && i < stride) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    const int nW = D * AP;
    if (i < nW) {
      const int d = static_cast<int>(i) / AP, q = static_cast<int>(i) - d * AP;
      if (q < A) gW[d * A + q] = t;
    } else if (i < nW + AP) {
      if (i - nW < A) gb[i - nW] = t;
    } else if (i < nW + 2 * AP) {
      if (i - nW - AP < A) gh[i - nW - AP] = t;
    } else {
      gp[i - nW - 2 * AP] = t;
    }
  }
}

int pad_a(int A) {
  int ap = 4;
  while (ap < A) ap <<= 1;
  return ap;
}

size_t lds_bytes(int F, int D, int AP, int P, bool bwd) {
  size_t n = static_cast<size_t>(D) * AP + 2 * AP + D + P + static_cast<size_t>(F) * D + P + D;
  if (bwd) n += D + static_cast<size_t>(P) * AP + static_cast<size_t>(P) * D;
  return n * sizeof(float);
}

int afm_groups(int B) { return B < 4096 ? B : 4096; }   // one wave each: about one round of the chip

template <bool BWD>
int launch(const AfmArgs& a, int AP, int groups, hipStream_t s) {
  const size_t lds = lds_bytes(a.F, a.D, AP, a.P, BWD);
  if (lds > 150u * 1024u) return DCTR_ENOSUP;
#define DCTR_AFM(AP_)                                                                                        \
  do {                                                                                                       \
    if (lds > 64u * 1024u)                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_afm<AP_, BWD>),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));          \
    k_afm<AP_, BWD><<<dim3(groups), dim3(kW), lds, s>>>(a);                                                  \
  } while (0)
  switch (AP) {
    case 4: DCTR_AFM(4); break;
    case 8: DCTR_AFM(8); break;
    case 16: DCTR_AFM(16); break;
    default: DCTR_AFM(32); break;
  }
#undef DCTR_AFM
  return launch_status();
}

int check(const float* E, int64_t ld_e, int B, int F, int D, int A, const float* W, const float* bias, const float* h,
          const float* p) {
  if (!E || !W || !bias || !h || !p || B < 0 || F < 2 || D <= 0 || A <= 0 || ld_e < static_cast<int64_t>(F) * D)
    return DCTR_EINVAL;
  if (D > 64 || A > 32 || F > 64) return DCTR_ENOSUP;
  return DCTR_OK;
}

}  // namespace

extern "C" size_t dctr_afm_bwd_workspace_floats(int32_t B, int32_t D, int32_t A) {
  if (B <= 0 || D <= 0 || A <= 0) return 0;
  const int AP = pad_a(A);
  return static_cast<size_t>(afm_groups(B)) * (static_cast<size_t>(D) * AP + 2 * AP + D);
}

extern "C" int dctr_afm_fwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t A, const float* W,
                            const float* bias, const float* h, const float* p, float* y, dctr_stream_t stream) {
  const int rc = check(E, ld_e, B, F, D, A, W, bias, h, p);
  if (rc != DCTR_OK) return rc;
  if (!y) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  AfmArgs a = {};
  a.E = E; a.lde = ld_e; a.W = W; a.bias = bias; a.h = h; a.p = p;
  a.B = B; a.F = F; a.D = D; a.A = A; a.P = F * (F - 1) / 2; a.y = y;
  const int groups = B < 4096 ? B : 4096;
  return launch<false>(a, pad_a(A), groups, static_cast<hipStream_t>(stream));
}

extern "C" int dctr_afm_bwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t A, const float* W,
                            const float* bias, const float* h, const float* p, const float* gy, float* gE,
                            int64_t ld_ge, float* gW, float* gbias, float* gh, float* gp, float* workspace,
                            dctr_stream_t stream) {
  const int rc = check(E, ld_e, B, F, D, A, W, bias, h, p);
  if (rc != DCTR_OK) return rc;
  if (!gy || !gE || !gW || !gbias || !gh || !gp || ld_ge < static_cast<int64_t>(F) * D) return DCTR_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) {
    (void)hipMemsetAsync(gW, 0, sizeof(float) * D * A, s);
    (void)hipMemsetAsync(gbias, 0, sizeof(float) * A, s);
    (void)hipMemsetAsync(gh, 0, sizeof(float) * A, s);
    (void)hipMemsetAsync(gp, 0, sizeof(float) * D, s);
    return DCTR_OK;
  }
  if (!workspace) return DCTR_EINVAL;
  const int AP = pad_a(A), groups = afm_groups(B);
  AfmArgs a = {};
  a.E = E; a.lde = ld_e; a.W = W; a.bias = bias; a.h = h; a.p = p;
  a.B = B; a.F = F; a.D = D; a.A = A; a.P = F * (F - 1) / 2;
  a.gy = gy; a.gE = gE; a.ldge = ld_ge; a.part = workspace;
  const int st = launch<true>(a, AP, groups, s);
  if (st != DCTR_OK) return st;
  const int64_t stride = static_cast<int64_t>(D) * AP + 2 * AP + D;
  k_afm_reduce<<<dim3(static_cast<unsigned>((stride + 15) / 16)), dim3(256), 0, s>>>(workspace, stride, groups, D, A,
                                                                                    AP, gW, gbias, gh, gp);
  return launch_status();
}

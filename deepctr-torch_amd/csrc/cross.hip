// cross.hip -- CrossNet, vector parameterisation (DCN-V; interaction.py:438-447) on gfx950.
//     x_{l+1} = x_0 * (x_l . w_l) + b_l + x_l          x in R^W per sample, L layers
// The reference runs tensordot + matmul + two adds per layer on [B, W, 1] tensors.  Here one wavefront owns a
// sample: x_0 and x_l live in registers (ceil(W/64) floats per lane), the dot product is a wave reduction (DPP
// shuffles), and all L layers run inside one launch -- the op is HBM-bound at 2*W*4 bytes per sample.
// Backward re-runs the recurrence (the per-layer inputs are kept in LDS), then walks it in reverse:
//     c_l = x_0 . g_{l+1};  g_l = g_{l+1} + w_l c_l;  g_{x0} += g_{l+1} s_l;  g_{w_l} += x_l c_l;  g_{b_l} += g_{l+1}
// Parameter gradients are reduced over a workgroup's samples in registers, then over workgroups by a fixed-order
// second pass (no float atomics: bit-reproducible).
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kT = 256;
constexpr int kNRMax = 32;  // W <= 2048

template <int NR>
__global__ __launch_bounds__(kT) void k_cross_fwd(const float* __restrict__ X, int64_t ldx, int B, int W, int L,
                                                  const float* __restrict__ Kw, const float* __restrict__ Kb,
                                                  float* __restrict__ Y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  // every load is unconditional on a clamped index and masked afterwards (a predicated load compiles to a branch with
  // its own s_waitcnt vmcnt(0): NR serial round trips per loop otherwise); lanes past W carry zeros
  float x0[NR], xl[NR];
  int ic[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int i = q * 64 + lane;
    ic[q] = i < W ? i : 0;
  }
#pragma unroll
  for (int q = 0; q < NR; ++q) x0[q] = ldg_f32(X + b * ldx + ic[q]);
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    x0[q] = (q * 64 + lane < W) ? x0[q] : 0.f;
    xl[q] = x0[q];
  }
  for (int l = 0; l < L; ++l) {
    float kw[NR], kb[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      kw[q] = ldg_f32(Kw + static_cast<int64_t>(l) * W + ic[q]);
      kb[q] = ldg_f32(Kb + static_cast<int64_t>(l) * W + ic[q]);
    }
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < NR; ++q) part += xl[q] * kw[q];           // xl is 0 past W
    const float s = wave_sum(part);
#pragma unroll
    for (int q = 0; q < NR; ++q) xl[q] = (q * 64 + lane < W) ? x0[q] * s + kb[q] + xl[q] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int i = q * 64 + lane;
    if (i < W) stg_f32(Y + b * ldy + i, xl[q]);
  }
}

// part: [gridDim.x * 4][2][L][W]  (gw then gb) -- per-WAVE partial parameter gradients
template <int NR>
__global__ __launch_bounds__(kT) void k_cross_bwd(const float* __restrict__ X, int64_t ldx, int B, int W, int L,
                                                  const float* __restrict__ Kw, const float* __restrict__ Kb,
                                                  const float* __restrict__ gY, int64_t ldg, int spw,
                                                  float* __restrict__ gX, int64_t ldgx, float* __restrict__ part) {
  extern __shared__ __align__(16) float smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* xs = smem + static_cast<size_t>(wv) * L * W;   // [L][W] inputs of every layer (this wave's sample)
  float* ss = smem + static_cast<size_t>(4) * L * W + wv * L;  // [L] the s_l
  // every WAVE owns one partial row of the workspace (plain stores, no read-modify-write, no barrier)
  float* mine = part + (static_cast<int64_t>(blockIdx.x) * 4 + wv) * 2 * L * W;
  for (int l = 0; l < L; ++l) {
    float gw[NR], gb[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) gw[q] = gb[q] = 0.f;
    // every wave walks its samples; the layer loop is outermost only for the parameter accumulators of layer l,
    // so the recurrence is recomputed per layer (L is 2-3): registers stay at 4*NR instead of (2L+2)*NR
    for (int sidx = 0; sidx < spw; ++sidx) {
      const int64_t b = (static_cast<int64_t>(blockIdx.x) * 4 + wv) * spw + sidx;
      if (b >= B) break;
      float x0[NR], xl[NR], g[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q) {          // unconditional loads on clamped indices, masked below
        const int i = q * 64 + lane, icq = i < W ? i : 0;
        x0[q] = ldg_f32(X + b * ldx + icq);
        g[q] = ldg_f32(gY + b * ldg + icq);
      }
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const bool in = q * 64 + lane < W;
        x0[q] = in ? x0[q] : 0.f;
        g[q] = in ? g[q] : 0.f;
        xl[q] = x0[q];
      }
      for (int k = 0; k < L; ++k) {  // forward again, remembering x_k and s_k
        float kw[NR], kb[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          const int i = q * 64 + lane, icq = i < W ? i : 0;
          kw[q] = ldg_f32(Kw + static_cast<int64_t>(k) * W + icq);
          kb[q] = ldg_f32(Kb + static_cast<int64_t>(k) * W + icq);
        }
        float p = 0.f;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          const int i = q * 64 + lane;
          if (i < W) xs[k * W + i] = xl[q];
          p += xl[q] * kw[q];                    // xl is 0 past W
        }
        const float s = wave_sum(p);
        if (lane == 0) ss[k] = s;
#pragma unroll
        for (int q = 0; q < NR; ++q) xl[q] = (q * 64 + lane < W) ? x0[q] * s + kb[q] + xl[q] : 0.f;
      }
      float gx0[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q) gx0[q] = 0.f;
      for (int k = L - 1; k >= l; --k) {  // reverse; layers below l do not matter for layer l's parameters
        float kwr[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          const int i = q * 64 + lane;
          kwr[q] = ldg_f32(Kw + static_cast<int64_t>(k) * W + (i < W ? i : 0));
        }
        float p = 0.f;
#pragma unroll
        for (int q = 0; q < NR; ++q) p += x0[q] * g[q];
        const float c = wave_sum(p);
        const float s = ss[k];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          const int i = q * 64 + lane;
          if (i < W) {
            if (k == l) {
              gw[q] += xs[k * W + i] * c;
              gb[q] += g[q];
            }
            gx0[q] += g[q] * s;
            g[q] += kwr[q] * c;
          }
        }
      }
      if (l == 0) {  // the full reverse walk was done: g = g_{x_l at 0}; total input gradient = g + gx0
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          const int i = q * 64 + lane;
          if (i < W) stg_f32(gX + b * ldgx + i, g[q] + gx0[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int i = q * 64 + lane;
      if (i < W) {
        stg_f32(mine + static_cast<int64_t>(l) * W + i, gw[q]);
        stg_f32(mine + static_cast<int64_t>(L + l) * W + i, gb[q]);
      }
    }
  }
}

// gK*[i] = sum_g part[g][i] in workgroup order: a workgroup owns 16 outputs, thread (o, sl) adds the groups sl, sl + 16,
// ... (eight loads in flight), the 16 slices are added in slice order.  (One thread per output walking the 256 groups
// one dependent load at a time was the slowest piece of the layer.)
__global__ __launch_bounds__(kT) void k_cross_reduce(const float* __restrict__ part, int64_t n, int groups,
                                                     float* __restrict__ gKw, float* __restrict__ gKb) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + o;
  const int64_t ic = i < 2 * n ? i : 0;
  float s = 0.f;
  for (int g0 = sl; g0 < groups; g0 += 16 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = g0 + 16 * u;
      v[u] = ldg_f32(part + static_cast<int64_t>(g < groups ? g : 0) * 2 * n + ic);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (g0 + 16 * u < groups) s += v[u];
  }
  red[sl][o] = s;
  __syncthreads();
  if (sl == 0 && i < 2 * n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    if (i < n) gKw[i] = t; else gKb[i - n] = t;
  }
}

int pick_nr(int W) {
  int nr = 1;
  while (nr * 64 < W) nr <<= 1;
  return nr;
}

int cross_groups(int B, int* spw) {
  // ~256 workgroups of 4 waves; every wave walks `spw` consecutive samples
  int s = (B + 1023) / 1024;
  if (s < 1) s = 1;
  *spw = s;
  return (B + 4 * s - 1) / (4 * s);
}

}  // namespace

#define DCTR_NR_SWITCH(nr, ...)                              \
  switch (nr) {                                              \
    case 1: { constexpr int NR = 1; __VA_ARGS__; } break;    \
    case 2: { constexpr int NR = 2; __VA_ARGS__; } break;    \
    case 4: { constexpr int NR = 4; __VA_ARGS__; } break;    \
    case 8: { constexpr int NR = 8; __VA_ARGS__; } break;    \
    case 16: { constexpr int NR = 16; __VA_ARGS__; } break;  \
    default: { constexpr int NR = 32; __VA_ARGS__; } break;  \
  }

extern "C" int dctr_crossnet_vec_fwd(const float* X, int64_t ld_x, int32_t B, int32_t W, int32_t L,
                                     const float* kernels, const float* bias, float* Y, int64_t ld_y,
                                     dctr_stream_t stream) {
  if (!X || !kernels || !bias || !Y || B < 0 || W <= 0 || L < 0 || ld_x < W || ld_y < W) return DCTR_EINVAL;
  if (W > 64 * kNRMax) return DCTR_ENOSUP;
  if (B == 0) return DCTR_OK;
  const int nr = pick_nr(W);
  hipStream_t s = static_cast<hipStream_t>(stream);
  DCTR_NR_SWITCH(nr, k_cross_fwd<NR><<<dim3((B + 3) / 4), dim3(kT), 0, s>>>(X, ld_x, B, W, L, kernels, bias, Y, ld_y));
  return launch_status();
}

extern "C" size_t dctr_crossnet_vec_bwd_workspace_floats(int32_t B, int32_t W, int32_t L) {
  int spw;
  return static_cast<size_t>(cross_groups(B > 0 ? B : 1, &spw)) * 4u * 2u * L * W;   // one row per wave
}

extern "C" int dctr_crossnet_vec_bwd(const float* X, int64_t ld_x, int32_t B, int32_t W, int32_t L,
                                     const float* kernels, const float* bias, const float* gY, int64_t ld_g,
                                     float* gX, int64_t ld_gx, float* g_kernels, float* g_bias, float* workspace,
                                     dctr_stream_t stream) {
  if (!X || !kernels || !bias || !gY || !gX || !g_kernels || !g_bias || !workspace || B < 0 || W <= 0 || L <= 0)
    return DCTR_EINVAL;
  if (W > 64 * kNRMax) return DCTR_ENOSUP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n = static_cast<int64_t>(L) * W;
  if (B == 0) {
    (void)hipMemsetAsync(g_kernels, 0, sizeof(float) * n, s);
    (void)hipMemsetAsync(g_bias, 0, sizeof(float) * n, s);
    return DCTR_OK;
  }
  int spw;
  const int groups = cross_groups(B, &spw);
  const size_t lds = (4u * L * W + 4u * L) * sizeof(float);
  if (lds > 150 * 1024) return DCTR_ENOSUP;
  const int nr = pick_nr(W);
  DCTR_NR_SWITCH(nr, {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cross_bwd<NR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    k_cross_bwd<NR><<<dim3(groups), dim3(kT), lds, s>>>(X, ld_x, B, W, L, kernels, bias, gY, ld_g, spw, gX, ld_gx,
                                                       workspace);
  });
  k_cross_reduce<<<dim3(static_cast<unsigned>((2 * n + 15) / 16)), dim3(kT), 0, s>>>(workspace, n, 4 * groups,
                                                                                         g_kernels, g_bias);
  return launch_status();
}

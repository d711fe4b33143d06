// fm.hip -- FM (interaction.py:26-34) on an explicit [B, F, D] tensor, forward and backward.
// HBM-bound streaming kernel: LPR lanes per sample, lane d walks the F fields of dimension d, so a
// wave reads 64/LPR samples x D contiguous floats per step.  The pairwise term is a group reduction
// done with wave shuffles.  (Inside DeepFM the same math is folded into embed.hip's gather/scatter.)
#include "common.hpp"

using namespace dctr;

namespace {

template <int LPR>
__global__ __launch_bounds__(256) void k_fm_fwd(const float* __restrict__ E, int64_t ldb, int B, int F,
                                                int D, float* __restrict__ y) {
  constexpr int SPB = 256 / LPR;
  const int grp = threadIdx.x / LPR, gl = threadIdx.x % LPR;
  const int b = blockIdx.x * SPB + grp;
  if (b >= B) return;
  const float* e = E + static_cast<int64_t>(b) * ldb;
  float t = 0.f;
  for (int d = gl; d < D; d += LPR) {
    float s = 0.f, q = 0.f;
    for (int f = 0; f < F; ++f) {
      const float v = e[f * D + d];
      s += v;
      q += v * v;
    }
    t += s * s - q;
  }
  t = group_sum<LPR>(t);
  if (gl == 0) y[b] = 0.5f * t;
}

template <int LPR>
__global__ __launch_bounds__(256) void k_fm_bwd(const float* __restrict__ E, int64_t ldb, int B, int F,
                                                int D, const float* __restrict__ gy,
                                                float* __restrict__ gE, int64_t ldg, int accumulate) {
  constexpr int SPB = 256 / LPR;
  const int grp = threadIdx.x / LPR, gl = threadIdx.x % LPR;
  const int b = blockIdx.x * SPB + grp;
  if (b >= B) return;
  const float* e = E + static_cast<int64_t>(b) * ldb;
  float* g = gE + static_cast<int64_t>(b) * ldg;
  const float gb = gy[b];
  for (int d = gl; d < D; d += LPR) {
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += e[f * D + d];
    for (int f = 0; f < F; ++f) {
      const float v = gb * (s - e[f * D + d]);
      if (accumulate) g[f * D + d] += v; else g[f * D + d] = v;
    }
  }
}

// BiInteractionPooling (interaction.py:54-61): the FM term BEFORE its sum over d,
//     bi[b, d] = 0.5 * ((sum_f e[b,f,d])^2 - sum_f e[b,f,d]^2),
// written straight into the NFM tower's input row [ bi (D) | dense (n_dense) ] (nfm.py:71: combined_dnn_input([bi_out],
// dense_value_list)) -- the dense values are copied from the gather's row, nothing is concatenated afterwards.
// G is the gather's output [B, ld_g]: row b = [ e_0 | ... | e_{F-1} | ... dense at dense_off ].  One lane per (b, d).
__global__ __launch_bounds__(256) void k_bi_fwd(const float* __restrict__ G, int64_t ldg, int B, int F, int D,
                                                int dense_off, int n_dense, float* __restrict__ out, int64_t ldo) {
  const int W = D + n_dense;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= static_cast<int64_t>(B) * W) return;
  const int64_t b = idx / W;
  const int c = static_cast<int>(idx - b * W);
  const float* row = G + b * ldg;
  if (c < D) {
    float s = 0.f, q = 0.f;
    for (int f = 0; f < F; ++f) {
      const float v = ldg_f32(row + f * D + c);
      s += v;
      q += v * v;
    }
    out[b * ldo + c] = 0.5f * (s * s - q);
  } else {
    out[b * ldo + c] = ldg_f32(row + dense_off + (c - D));
  }
}

// gG[b, f*D + d] = g[b, d] * (S[b, d] - e[b, f, d]) ;  gG[b, dense_off + j] = g[b, D + j]   (gG has G's layout)
__global__ __launch_bounds__(256) void k_bi_bwd(const float* __restrict__ G, int64_t ldg, int B, int F, int D,
                                                int dense_off, int n_dense, const float* __restrict__ go, int64_t ldgo,
                                                float* __restrict__ gG, int64_t ldgg) {
  const int W = D + n_dense;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= static_cast<int64_t>(B) * W) return;
  const int64_t b = idx / W;
  const int c = static_cast<int>(idx - b * W);
  const float g = ldg_f32(go + b * ldgo + c);
  if (c < D) {
    const float* row = G + b * ldg;
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += ldg_f32(row + f * D + c);
    for (int f = 0; f < F; ++f) gG[b * ldgg + f * D + c] = g * (s - ldg_f32(row + f * D + c));
  } else {
    gG[b * ldgg + dense_off + (c - D)] = g;
  }
}

int pick_lpr(int D) {
  int lpr = 1;
  while (lpr < D && lpr < 64) lpr <<= 1;
  return lpr;
}

#define FM_DISPATCH(lpr, ...)                                \
  switch (lpr) {                                              \
    case 1: { constexpr int LPR = 1; __VA_ARGS__; } break;           \
    case 2: { constexpr int LPR = 2; __VA_ARGS__; } break;           \
    case 4: { constexpr int LPR = 4; __VA_ARGS__; } break;           \
    case 8: { constexpr int LPR = 8; __VA_ARGS__; } break;           \
    case 16: { constexpr int LPR = 16; __VA_ARGS__; } break;         \
    case 32: { constexpr int LPR = 32; __VA_ARGS__; } break;         \
    default: { constexpr int LPR = 64; __VA_ARGS__; } break;         \
  }

}  // namespace

extern "C" int dctr_fm_fwd(const float* E, int64_t ld_b, int32_t B, int32_t F, int32_t D, float* y,
                           dctr_stream_t stream) {
  if (!E || !y || B < 0 || F <= 0 || D <= 0 || ld_b < static_cast<int64_t>(F) * D) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const int lpr = pick_lpr(D);
  const int spb = 256 / lpr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  FM_DISPATCH(lpr, k_fm_fwd<LPR><<<dim3((B + spb - 1) / spb), dim3(256), 0, s>>>(E, ld_b, B, F, D, y));
  return launch_status();
}

extern "C" int dctr_fm_bwd(const float* E, int64_t ld_b, int32_t B, int32_t F, int32_t D,
                           const float* gy, float* gE, int64_t ld_gb, int32_t accumulate,
                           dctr_stream_t stream) {
  if (!E || !gy || !gE || B < 0 || F <= 0 || D <= 0 || ld_b < static_cast<int64_t>(F) * D ||
      ld_gb < static_cast<int64_t>(F) * D)
    return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const int lpr = pick_lpr(D);
  const int spb = 256 / lpr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  FM_DISPATCH(lpr, k_fm_bwd<LPR><<<dim3((B + spb - 1) / spb), dim3(256), 0, s>>>(
                       E, ld_b, B, F, D, gy, gE, ld_gb, accumulate));
  return launch_status();
}

extern "C" int dctr_bi_pooling_fwd(const float* G, int64_t ld_g, int32_t B, int32_t F, int32_t D, int32_t dense_off,
                                   int32_t n_dense, float* out, int64_t ld_o, dctr_stream_t stream) {
  if (!G || !out || B < 0 || F <= 0 || D <= 0 || n_dense < 0 || ld_g < static_cast<int64_t>(F) * D ||
      ld_o < D + n_dense || (n_dense > 0 && (dense_off < F * D || ld_g < dense_off + n_dense)))
    return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const int64_t n = static_cast<int64_t>(B) * (D + n_dense);
  k_bi_fwd<<<dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      G, ld_g, B, F, D, dense_off, n_dense, out, ld_o);
  return launch_status();
}

extern "C" int dctr_bi_pooling_bwd(const float* G, int64_t ld_g, int32_t B, int32_t F, int32_t D, int32_t dense_off,
                                   int32_t n_dense, const float* gout, int64_t ld_go, float* gG, int64_t ld_gg,
                                   dctr_stream_t stream) {
  if (!G || !gout || !gG || B < 0 || F <= 0 || D <= 0 || n_dense < 0 || ld_g < static_cast<int64_t>(F) * D ||
      ld_gg < static_cast<int64_t>(F) * D || ld_go < D + n_dense ||
      (n_dense > 0 && (dense_off < F * D || ld_gg < dense_off + n_dense)))
    return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const int64_t n = static_cast<int64_t>(B) * (D + n_dense);
  k_bi_bwd<<<dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      G, ld_g, B, F, D, dense_off, n_dense, gout, ld_go, gG, ld_gg);
  return launch_status();
}

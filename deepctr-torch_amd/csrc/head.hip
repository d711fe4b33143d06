// head.hip -- the prediction head + loss of a binary CTR model, and the dense optimizer step (gfx950).
//
// k_bce_head replaces, per train step, the reference's chain (deepfm.py:78-86, core.py:154-160,
// basemodel.py:254-261):  logit adds, PredictionLayer (bias add + sigmoid), F.binary_cross_entropy(reduction=
// 'sum') and its autograd backward (binary_cross_entropy_backward, sigmoid_backward, the bias column-sum) -- about
// a dozen elementwise / reduce launches over 4096 floats, each 1.5-5 us of pure launch latency.
//
// k_dense_opt replaces torch.optim's foreach walk over the dense parameters (Adagrad: addcmul, sqrt, add, div /
// addcdiv = 5-8 launches) with one pass over one flat slab (deepctr_torch/_hip/dense.py re-seats the parameters
// as views of that slab).
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kTH = 1024;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();  // red may still be read from a previous call
  if (lane == 0) red[wv] = v;
  __syncthreads();
  float t = (threadIdx.x < kTH / 64) ? red[threadIdx.x] : 0.f;
  if (wv == 0) t = wave_sum(t);
  return t;  // valid in wave 0
}

__global__ __launch_bounds__(kTH) void k_bce_head(const float* __restrict__ p0, const float* __restrict__ p1,
                                                  const float* __restrict__ p2, const float* __restrict__ p3,
                                                  const float* __restrict__ bias, const float* __restrict__ y,
                                                  int B, float* __restrict__ y_pred, float* __restrict__ loss,
                                                  float* __restrict__ g_logit, float* __restrict__ g_bias) {
  __shared__ float red[kTH / 64];
  const float bv = bias ? ldg_f32(bias) : 0.f;
  float lsum = 0.f, gsum = 0.f;
  for (int b = threadIdx.x; b < B; b += kTH) {
    float z = 0.f;
    if (p0) z += ldg_f32(p0 + b);   // same association order as the reference: ((linear + fm) + dnn) + bias
    if (p1) z += ldg_f32(p1 + b);
    if (p2) z += ldg_f32(p2 + b);
    if (p3) z += ldg_f32(p3 + b);
    z += bv;
    const float p = 1.f / (1.f + expf(-z));                     // at::sigmoid
    const float t = ldg_f32(y + b);
    const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
    lsum += (t - 1.f) * l1p - t * lp;                            // at::binary_cross_entropy
    const float q = (1.f - p) * p;
    const float gp = (p - t) / fmaxf(q, 1e-12f);                 // binary_cross_entropy_backward (grad = 1)
    const float gz = gp * q;                                     // sigmoid_backward
    if (y_pred) stg_f32(y_pred + b, p);
    if (g_logit) stg_f32(g_logit + b, gz);
    gsum += gz;
  }
  const float L = block_sum(lsum, red);
  const float G = block_sum(gsum, red);
  if (threadIdx.x == 0) {
    if (loss) stg_f32(loss, L);
    if (g_bias) stg_f32(g_bias, G);
  }
}

template <int OPT>
__global__ __launch_bounds__(256) void k_dense_opt(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ st, int64_t n4, int64_t n, float lr,
                                                   float eps) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n4) {
    f32x4 pv = *(const DCTR_GLOBAL f32x4*)(p + 4 * i);
    const f32x4 gv = *(const DCTR_GLOBAL f32x4*)(g + 4 * i);
    if (OPT == DCTR_UPD_ADAGRAD) {
      f32x4 sv = *(const DCTR_GLOBAL f32x4*)(st + 4 * i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sv[k] = adagrad_sum(sv[k], gv[k]);      // (common.hpp: the one definition of the dense optimizers' arithmetic)
        pv[k] = adagrad_param(pv[k], gv[k], sv[k], lr, eps);
      }
      *(DCTR_GLOBAL f32x4*)(st + 4 * i) = sv;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) pv[k] = sgd_param(pv[k], gv[k], lr);
    }
    *(DCTR_GLOBAL f32x4*)(p + 4 * i) = pv;
  } else {
    const int64_t j = 4 * n4 + (i - n4);
    if (j < n) {
      const float gv = ldg_f32(g + j);
      float pv = ldg_f32(p + j);
      if (OPT == DCTR_UPD_ADAGRAD) {
        const float sv = adagrad_sum(ldg_f32(st + j), gv);
        stg_f32(st + j, sv);
        pv = adagrad_param(pv, gv, sv, lr, eps);
      } else {
        pv = sgd_param(pv, gv, lr);
      }
      stg_f32(p + j, pv);
    }
  }
}

// the same step on a LIST of tensors in one launch (torch.optim's foreach walk over an autograd-step model's
// parameters: 5 launches, 75-80 us per step for xDeepFM / FiBiNET / DCN).  Tensors are cut into chunks of kChunk
// elements; a block finds its tensor from the chunk prefix sums in the kernel arguments.
constexpr int kMulti = 48, kChunk = 4096;
struct MultiArgs {
  float* p[kMulti];
  const float* g[kMulti];
  float* st[kMulti];
  long long n[kMulti];
  float c2[kMulti];           // 2 * lambda of the tensor's L2 term (0: none): g += c2 * p before the step
  int blk0[kMulti + 1];
  int n_items;
  float lr, eps;
};

template <int OPT>
__global__ __launch_bounds__(256) void k_dense_opt_multi(MultiArgs A) {
  int t = 0;
  const int blk = blockIdx.x;
  while (t + 1 < A.n_items && blk >= A.blk0[t + 1]) ++t;
  float* __restrict__ p = A.p[t];
  const float* __restrict__ g = A.g[t];
  float* __restrict__ st = A.st[t];
  const long long n = A.n[t];
  const float c2 = A.c2[t];
  const long long base = static_cast<long long>(blk - A.blk0[t]) * kChunk;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                         reinterpret_cast<uintptr_t>(st)) & 15u) == 0;
#pragma unroll
  for (int r = 0; r < kChunk / 1024; ++r) {
    const long long i = base + r * 1024 + 4 * threadIdx.x;
    if (i >= n) break;
    if (aligned && i + 4 <= n) {
      f32x4 pv = *(const DCTR_GLOBAL f32x4*)(p + i);
      f32x4 gv = *(const DCTR_GLOBAL f32x4*)(g + i);
      if (c2 != 0.f) {            // autograd's d/dp of lambda * sum(p^2), added to the gradient: two roundings, no fma
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[k] = __fadd_rn(gv[k], __fmul_rn(c2, pv[k]));
      }
      if (OPT == DCTR_UPD_ADAGRAD) {
        f32x4 sv = *(const DCTR_GLOBAL f32x4*)(st + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sv[k] = adagrad_sum(sv[k], gv[k]);
          pv[k] = adagrad_param(pv[k], gv[k], sv[k], A.lr, A.eps);
        }
        *(DCTR_GLOBAL f32x4*)(st + i) = sv;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[k] = sgd_param(pv[k], gv[k], A.lr);
      }
      *(DCTR_GLOBAL f32x4*)(p + i) = pv;
    } else {
      for (long long j = i; j < i + 4 && j < n; ++j) {
        float gv = ldg_f32(g + j);
        float pv = ldg_f32(p + j);
        if (c2 != 0.f) gv = __fadd_rn(gv, __fmul_rn(c2, pv));
        if (OPT == DCTR_UPD_ADAGRAD) {
          const float sv = adagrad_sum(ldg_f32(st + j), gv);
          stg_f32(st + j, sv);
          pv = adagrad_param(pv, gv, sv, A.lr, A.eps);
        } else {
          pv = sgd_param(pv, gv, A.lr);
        }
        stg_f32(p + j, pv);
      }
    }
  }
}

// sum_i lambda_i * sum(p_i^2) of a tensor list: ONE workgroup, fixed order (thread t sums elements t, t + 1024, ...
// of every tensor in list order, then a tree): deterministic.  The logged value of the L2 terms that
// dctr_dense_opt_multi applies as gradients (get_regularization_loss, basemodel.py:412-428).
__global__ __launch_bounds__(kTH) void k_l2_value_multi(MultiArgs A, float* __restrict__ out) {
  __shared__ float red[kTH / 64];
  float acc = 0.f;
  for (int t = 0; t < A.n_items; ++t) {
    const float lam = 0.5f * A.c2[t];
    if (lam == 0.f) continue;
    const float* __restrict__ p = A.p[t];
    float s = 0.f;
    for (long long i = threadIdx.x; i < A.n[t]; i += kTH) {
      const float v = ldg_f32(p + i);
      s += v * v;
    }
    acc += lam * s;
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) stg_f32(out, A.lr != 0.f ? tot + ldg_f32(out) : tot);     // (lr: "a launch went before this one")
}


// ---- autograd glue of the models that run through torch.autograd (xDeepFM, FiBiNET, DCN, PNN, ...) as single launches ----

// out[b, 0:W) = a[b, :] (+ c[b, :]),  out[b, W:W+nd) = d[b, :],  out[b, W+nd:ldo) = 0 : the gradient of the gather's output
// [B, ld] from the gradients of its two views (the field block as [B, F*D], the dense block) -- as slices that is two
// copies and a fill (three launches, 17 us at the Criteo shape), plus an add when two consumers share the field block.
// W % 4 == 0, ldo % 4 == 0, rows 16-byte aligned: the field block moves as dwordx4.
__global__ __launch_bounds__(256) void k_rows_join(const float* __restrict__ a, int64_t lda, const float* __restrict__ c,
                                                   int64_t ldc, int W, const float* __restrict__ d, int64_t ldd, int nd,
                                                   float* __restrict__ out, int64_t ldo, int B) {
  const int q4 = static_cast<int>(ldo >> 2), w4 = W >> 2;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * q4) return;
  const int b = static_cast<int>(i / q4), q = static_cast<int>(i - static_cast<int64_t>(b) * q4);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (q < w4) {
    if (a) v = *(const DCTR_GLOBAL f32x4*)(a + b * lda + 4 * q);
    if (c) v += *(const DCTR_GLOBAL f32x4*)(c + b * ldc + 4 * q);
  } else if (d) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * q + k - W;
      if (col < nd) v[k] = ldg_f32(d + b * ldd + col);
    }
  }
  *(DCTR_GLOBAL f32x4*)(out + b * ldo + 4 * q) = v;
}

// out[b] = sum_j x[b, j] w[j]: a bias-free 1-unit Linear over narrow rows (xDeepFM's cin_linear, xdeepfm.py:72 / :97 -- 192
// columns at the Criteo shape, where the library GEMM is 12 us of launch + tile set-up for 0.8 MFLOP).  One wave per row,
// lanes stride the columns, a fixed butterfly adds the 64 partials: the same bits on every run.
__global__ __launch_bounds__(256) void k_rows_dot(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                  int B, int N, float* __restrict__ out) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* row = x + static_cast<int64_t>(b) * ldx;
  float s = 0.f;
  for (int j = lane; j < N; j += 64) s += ldg_f32(row + j) * ldg_f32(w + j);
  s = wave_sum(s);
  if (lane == 0) stg_f32(out + b, s);
}

// relu's backward on g [B, N] (mask h > 0, like aten::threshold_backward) fused with the bias gradient's column sums.
// Workgroup w takes kColRows rows; its 256 lanes are 4 row slices x 64 columns: lane (ty, tx) adds the rows ty, ty + 4, ...
// of the group for the columns tx, tx + 64, ... (every load of a lane independent of the others: one memory round trip per
// column tile -- the one-lane-per-column version walked 32 rows in four dependent rounds with half its lanes idle at
// N = 128: 16.6 us for 6 MB), the four slices meet in LDS in slice order, and k_colsum_finish adds the groups' sums in a
// fixed order.  Deterministic.  Replaces threshold_backward + sum(0) (a 2-stage ATen reduction).
constexpr int kColRows = 16;
template <bool WSUM>
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ g, int64_t ldg, const float* __restrict__ h,
                                                     int64_t ldh, const float* __restrict__ wt, int B, int N,
                                                     float* __restrict__ go, int64_t ldo, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int r0 = blockIdx.x * kColRows;
  for (int n0 = 0; n0 < N; n0 += 64) {
    const int n = n0 + tx;
    float s = 0.f;
    if (n < N) {
#pragma unroll
      for (int k = 0; k < kColRows / 4; ++k) {
        const int b = r0 + ty + 4 * k;
        if (b < B) {
          const float gv = ldg_f32(g + b * ldg + n);
          if (WSUM) {     // part[w][n] = sum_b wt[b] * x[b, n]  (g is x here)
            s = fmaf(ldg_f32(wt + b), gv, s);
          } else {
            const float v = (h == nullptr || ldg_f32(h + b * ldh + n) > 0.f) ? gv : 0.f;
            if (go) stg_f32(go + b * ldo + n, v);
            s += v;
          }
        }
      }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < N)
      stg_f32(part + static_cast<int64_t>(blockIdx.x) * N + n, ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx]);
    __syncthreads();
  }
}
// out[n] = sum_w part[w][n]: sixteen neighbouring lanes share a column, lane j adds the groups j, j + 16, ... ascending and
// the sixteen sums meet in a butterfly (xor 8, 4, 2, 1): a fixed order, a sixteenth of the one-lane-per-column chain.
__global__ __launch_bounds__(256) void k_colsum_finish(const float* __restrict__ part, int groups, int N,
                                                       float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int n = t >> 4, j = t & 15;
  const int nc = n < N ? n : N - 1;          // (whole 16-lane groups stay converged for the shuffles)
  float s = 0.f;
  for (int w = j; w < groups; w += 64) {     // four loads in flight per lane
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ldg_f32(part + static_cast<int64_t>(w + 16 * k < groups ? w + 16 * k : groups - 1) * N + nc);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (w + 16 * k < groups) s += v[k];
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    const float o = __shfl_xor(s, off, 64);
    s = (j & off) ? o + s : s + o;           // lower lane's sum first on both sides: the same bits on all sixteen lanes
  }
  if (n < N && j == 0) stg_f32(out + n, s);
}
}  // namespace

extern "C" int dctr_l2_value_multi(const dctr_dense_item_t* items, int32_t n_items, float* out, dctr_stream_t stream) {
  if (n_items < 0 || (n_items > 0 && !items) || !out) return DCTR_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_items == 0) (void)hipMemsetAsync(out, 0, sizeof(float), s);
  for (int i0 = 0; i0 < n_items; i0 += kMulti) {
    MultiArgs a;
    int k = 0;
    for (; k < kMulti && i0 + k < n_items; ++k) {
      const dctr_dense_item_t& it = items[i0 + k];
      if (it.n < 0 || (it.n > 0 && !it.p)) return DCTR_EINVAL;
      a.p[k] = it.p; a.g[k] = nullptr; a.st[k] = nullptr; a.n[k] = it.n; a.c2[k] = 2.f * it.l2;
    }
    a.n_items = k; a.eps = 0.f;
    a.lr = i0 > 0 ? 1.f : 0.f;                                 // the first launch stores, the later ones add
    k_l2_value_multi<<<dim3(1), dim3(kTH), 0, s>>>(a, out);    // (launches are stream-ordered)
    const int stt = launch_status();
    if (stt != DCTR_OK) return stt;
  }
  return DCTR_OK;
}

extern "C" int dctr_dense_opt_multi(const dctr_dense_item_t* items, int32_t n_items, int32_t opt, float lr, float eps,
                                    dctr_stream_t stream) {
  if (n_items < 0 || (n_items > 0 && !items)) return DCTR_EINVAL;
  if (opt != DCTR_UPD_SGD && opt != DCTR_UPD_ADAGRAD) return DCTR_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int i0 = 0; i0 < n_items;) {
    MultiArgs a;
    int k = 0;
    long long blocks = 0;
    for (; k < kMulti && i0 + k < n_items; ++k) {
      const dctr_dense_item_t& it = items[i0 + k];
      if (it.n < 0 || (it.n > 0 && (!it.p || !it.g || (opt == DCTR_UPD_ADAGRAD && !it.state)))) return DCTR_EINVAL;
      const long long nb = (it.n + kChunk - 1) / kChunk;
      if (blocks + nb > 0x7FFFFFF0ll) return DCTR_ENOSUP;
      a.p[k] = it.p; a.g[k] = it.g; a.st[k] = opt == DCTR_UPD_ADAGRAD ? it.state : it.p; a.n[k] = it.n;
      a.c2[k] = 2.f * it.l2;
      a.blk0[k] = static_cast<int>(blocks);
      blocks += nb;
    }
    a.blk0[k] = static_cast<int>(blocks);
    a.n_items = k; a.lr = lr; a.eps = eps;
    if (blocks > 0) {
      const dim3 grid(static_cast<unsigned>(blocks)), block(256);
      if (opt == DCTR_UPD_ADAGRAD) k_dense_opt_multi<DCTR_UPD_ADAGRAD><<<grid, block, 0, s>>>(a);
      else k_dense_opt_multi<DCTR_UPD_SGD><<<grid, block, 0, s>>>(a);
      const int stt = launch_status();
      if (stt != DCTR_OK) return stt;
    }
    i0 += k;
  }
  return DCTR_OK;
}

extern "C" int dctr_bce_head(const float* part0, const float* part1, const float* part2, const float* part3,
                             const float* bias, const float* y, int32_t B, float* y_pred, float* loss,
                             float* g_logit, float* g_bias, dctr_stream_t stream) {
  if (!y || B < 0 || (!part0 && !part1 && !part2 && !part3)) return DCTR_EINVAL;
  k_bce_head<<<dim3(1), dim3(kTH), 0, static_cast<hipStream_t>(stream)>>>(part0, part1, part2, part3, bias, y, B,
                                                                         y_pred, loss, g_logit, g_bias);
  return launch_status();
}

extern "C" int dctr_dense_opt(float* p, const float* g, float* state, int64_t n, int32_t opt, float lr, float eps,
                              dctr_stream_t stream) {
  if (!p || !g || n < 0) return DCTR_EINVAL;
  if (opt != DCTR_UPD_SGD && opt != DCTR_UPD_ADAGRAD) return DCTR_EINVAL;
  if (opt == DCTR_UPD_ADAGRAD && !state) return DCTR_EINVAL;
  if (n == 0) return DCTR_OK;
  const bool aligned = reinterpret_cast<uintptr_t>(p) % 16 == 0 && reinterpret_cast<uintptr_t>(g) % 16 == 0 &&
                       (!state || reinterpret_cast<uintptr_t>(state) % 16 == 0);
  const int64_t n4 = aligned ? n / 4 : 0;
  const int64_t items = n4 + (n - 4 * n4);
  const dim3 grid(static_cast<unsigned>((items + 255) / 256)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (opt == DCTR_UPD_ADAGRAD)
    k_dense_opt<DCTR_UPD_ADAGRAD><<<grid, block, 0, s>>>(p, g, state, n4, n, lr, eps);
  else
    k_dense_opt<DCTR_UPD_SGD><<<grid, block, 0, s>>>(p, g, state, n4, n, lr, eps);
  return launch_status();
}

extern "C" int dctr_rows_join(const float* a, int64_t ld_a, const float* c, int64_t ld_c, int32_t W, const float* d,
                              int64_t ld_d, int32_t n_d, float* out, int64_t ld_out, int32_t B, dctr_stream_t stream) {
  if (!out || B < 0 || W < 0 || n_d < 0 || ld_out < W + n_d || (n_d > 0 && !d)) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  if (W % 4 != 0 || ld_out % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return DCTR_EALIGN;
  if (a && (ld_a % 4 != 0 || reinterpret_cast<uintptr_t>(a) % 16 != 0 || ld_a < W)) return DCTR_EALIGN;
  if (c && (ld_c % 4 != 0 || reinterpret_cast<uintptr_t>(c) % 16 != 0 || ld_c < W)) return DCTR_EALIGN;
  const int64_t n = static_cast<int64_t>(B) * (ld_out / 4);
  k_rows_join<<<dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      a, ld_a, c, ld_c, W, d, ld_d, n_d, out, ld_out, B);
  return launch_status();
}

extern "C" int dctr_rows_dot(const float* x, int64_t ld_x, const float* w, int32_t B, int32_t N, float* out,
                             dctr_stream_t stream) {
  if (!x || !w || !out || B < 0 || N <= 0 || ld_x < N) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  k_rows_dot<<<dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(x, ld_x, w, B, N, out);
  return launch_status();
}

extern "C" int dctr_rows_tdot(const float* x, int64_t ld_x, const float* w, int32_t B, int32_t N, float* out,
                              float* workspace, dctr_stream_t stream) {
  if (!x || !w || !out || !workspace || B <= 0 || N <= 0 || ld_x < N) return DCTR_EINVAL;
  const int groups = (B + kColRows - 1) / kColRows;
  hipStream_t s = static_cast<hipStream_t>(stream);
  k_colsum_part<true><<<dim3(groups), dim3(256), 0, s>>>(x, ld_x, nullptr, 0, w, B, N, nullptr, 0, workspace);
  const int st = launch_status();
  if (st != DCTR_OK) return st;
  k_colsum_finish<<<dim3((16 * N + 255) / 256), dim3(256), 0, s>>>(workspace, groups, N, out);
  return launch_status();
}

extern "C" size_t dctr_relu_bwd_bias_workspace_floats(int32_t B, int32_t N) {
  if (B <= 0 || N <= 0) return 0;
  return static_cast<size_t>((B + kColRows - 1) / kColRows) * N;
}

extern "C" int dctr_relu_bwd_bias(const float* g, int64_t ld_g, const float* h, int64_t ld_h, int32_t B, int32_t N,
                                  float* g_out, int64_t ld_o, float* g_bias, float* workspace, dctr_stream_t stream) {
  if (!g || !g_bias || !workspace || B <= 0 || N <= 0 || ld_g < N || (h && ld_h < N) || (g_out && ld_o < N))
    return DCTR_EINVAL;
  const int groups = (B + kColRows - 1) / kColRows;
  hipStream_t s = static_cast<hipStream_t>(stream);
  k_colsum_part<false><<<dim3(groups), dim3(256), 0, s>>>(g, ld_g, h, ld_h, nullptr, B, N, g_out, ld_o, workspace);
  const int st = launch_status();
  if (st != DCTR_OK) return st;
  k_colsum_finish<<<dim3((16 * N + 255) / 256), dim3(256), 0, s>>>(workspace, groups, N, g_bias);
  return launch_status();
}

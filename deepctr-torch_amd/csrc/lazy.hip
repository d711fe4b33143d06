// lazy.hip -- the reference's REGULARISED / ADAM embedding update in O(batch), exactly (gfx950).
//
// With l2_reg_embedding / l2_reg_linear > 0 (the reference's defaults, basemodel.py:100-102,412-428) or with
// torch.optim.Adam (the examples' optimizer, basemodel.py:447-461) EVERY row of every table moves at every step:
// the L2 term contributes the gradient 2*lambda*w to all V rows, Adam's moments keep pushing rows whose data
// gradient is zero.  The reference pays O(vocabulary) for that (442 M parameters at the Criteo shape: 7 ms / step
// even on this GPU).  But a row's trajectory between two batches that touch it depends on nothing but the row
// itself:   g_t = 2*lambda*w_t ;  (w, state)_{t+1} = opt_step(w_t, state_t, g_t, t+1)
// so it can be replayed LAZILY, bit-for-bit the same recurrence, the next time the row is needed:
//   stamp[row] = number of optimizer steps already applied to the row;  *step = steps completed so far (t)
//   k_lazy_catchup  (before the gather of a train step, on the batch's ids): the lane group that wins
//                   atomicMax(stamp[row], t) replays the missed steps stamp..t-1 with g = 2*lambda*w -> the gather
//                   reads exactly the reference's w_t;
//   dctr_embed_update(OPT_ACCUM) (csrc/update.hip, deterministic, no atomics): gacc[row] = sum of the batch's data
//                   gradients of the row;
//   k_lazy_apply    (after the backward): the group that wins atomicMax(stamp[row], t+1) applies step t+1 with
//                   g = gacc[row] + 2*lambda*w and zeroes gacc[row];
//   k_lazy_flush    (before anything else reads the tables: predict / evaluate / state_dict): replays every row to t.
// Cost per step: O(batch * mean gap) optimizer steps (gap = V / B ~ 244 at the Criteo shape: ~0.1 ms of VALU work)
// instead of O(V) memory traffic.  Deterministic: whichever duplicate wins a row computes the same thing.
//
// Optimizer arithmetic = torch.optim's (single-tensor formulas, fp32; the step-dependent scalars in double like
// torch computes them on the host):
//   SGD      w -= lr * g
//   Adagrad  s += g*g ;  w -= lr * g / (sqrt(s) + eps)
//   Adam     m += (g - m) * (1 - b1) ;  v = v*b2 + (1 - b2)*g*g ;
//            w -= (lr / (1 - b1^T)) * m / (sqrt(v) / sqrt(1 - b2^T) + eps)                      T = step number
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kT = 256;

struct OptConst {
  int kind;   // DCTR_LAZY_SGD / ADAGRAD / ADAM / RMSPROP
  float lr, eps, beta1, beta2;
};

// step-dependent scalars of Adam for step number T (1-based), maintained incrementally in double
struct AdamClock {
  double p1, p2;      // beta1^T, beta2^T
  double b1, b2, lr;
  __device__ __forceinline__ void start(const OptConst& o, int T) {
    b1 = o.beta1; b2 = o.beta2; lr = o.lr;
    p1 = pow(b1, static_cast<double>(T));
    p2 = pow(b2, static_cast<double>(T));
  }
  __device__ __forceinline__ void next() { p1 *= b1; p2 *= b2; }
  __device__ __forceinline__ float step_size() const { return static_cast<float>(lr / (1.0 - p1)); }
  __device__ __forceinline__ float bc2_sqrt() const { return static_cast<float>(sqrt(1.0 - p2)); }
};

// one optimizer step on one element.  a = Adagrad sum | Adam exp_avg, b = Adam exp_avg_sq.
__device__ __forceinline__ void opt_step(const OptConst& o, float g, float& w, float& a, float& b, float step_size,
                                         float bc2s) {
  if (o.kind == DCTR_LAZY_ADAM) {
    a = a + (g - a) * (1.f - o.beta1);
    b = b * o.beta2 + (1.f - o.beta2) * g * g;
    const float denom = sqrtf(b) / bc2s + o.eps;
    w = w - step_size * (a / denom);
  } else if (o.kind == DCTR_LAZY_ADAGRAD) {
    a = a + g * g;
    w = w - o.lr * (g / (sqrtf(a) + o.eps));
  } else if (o.kind == DCTR_LAZY_RMSPROP) {
    // square_avg.mul_(alpha).addcmul_(g, g, value=1 - alpha); p.addcdiv_(g, square_avg.sqrt().add_(eps), value=-lr)
    // -- with the roundings of ATen's device kernels (each tensor op rounds; addcmul is a + (v * b) * c and addcdiv is
    // a + v * (b / c), their last multiply-add contracted): RMSprop divides by sqrt(square_avg) + 1e-8, so a row whose
    // accumulator is still ~g^2/100 moves by 10 lr whatever |g| is, and an ulp of difference in a sign-deciding value
    // shows up as 0.1 in the weight.  beta1 carries float(1 - alpha) as torch computes it (in double).
    const float a1 = __fmul_rn(a, o.beta2);
    a = __fmaf_rn(__fmul_rn(o.beta1, g), g, a1);
    w = __fmaf_rn(-o.lr, __fdiv_rn(g, __fadd_rn(__fsqrt_rn(a), o.eps)), w);
  } else {
    w = w - o.lr * g;
  }
}

// Replay the steps from+1 .. to of an UNTOUCHED element (g = 2*lambda*w) -- the reference's dense update of a row
// no sample of those batches referred to.
template <int VEC>
__device__ __forceinline__ void replay(const OptConst& o, float lam2, int from, int to, float (&w)[VEC],
                                       float (&a)[VEC], float (&b)[VEC]) {
  if (from >= to) return;
  // zero gradient: SGD / Adagrad do not move (RMSprop's square_avg still decays, Adam's moments too)
  if ((o.kind == DCTR_LAZY_SGD || o.kind == DCTR_LAZY_ADAGRAD) && lam2 == 0.f) return;
  AdamClock ck;
  if (o.kind == DCTR_LAZY_ADAM) ck.start(o, from + 1);
  for (int T = from + 1; T <= to; ++T) {
    float ss = 0.f, bc = 1.f;
    if (o.kind == DCTR_LAZY_ADAM) {
      ss = ck.step_size();
      bc = ck.bc2_sqrt();
      ck.next();
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) opt_step(o, lam2 * w[i], w[i], a[i], b[i], ss, bc);
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  if (!p) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = 0.f;
    return;
  }
  const Strip<VEC> s = strip_load<VEC>(p);
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = s.v[i];
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if (!p) return;
  Strip<VEC> s;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s.v[i] = v[i];
  strip_store<VEC>(p, s);
}

// The three passes share one body: MODE 0 catch-up (batch ids), 1 apply (batch ids), 2 flush (all rows).
// A lane group of `lpr` lanes (a power of two <= 64) owns one (unit, entry); lane gl handles the deep strip
// [gl*VEC, gl*VEC + VEC) and, when gl == 0, the wide element.
template <int VEC, int MODE>
__global__ __launch_bounds__(kT) void k_lazy(const dctr_lazy_unit_t* __restrict__ units, int n_units,
                                             const int32_t* __restrict__ ids_t, int64_t n_entries, int lpr_shift,
                                             const int32_t* __restrict__ step_ptr, OptConst o) {
  const int lpr = 1 << lpr_shift;
  const int64_t grp = (static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x) >> lpr_shift;
  const int gl = threadIdx.x & (lpr - 1);
  const int u = blockIdx.y;
  if (u >= n_units || grp >= n_entries) return;
  const dctr_lazy_unit_t un = units[u];
  const int t = *(const DCTR_GLOBAL int32_t*)step_ptr;
  int64_t row;
  if (MODE == 2) {
    row = grp;
    if (row >= un.vocab) return;
  } else {
    const int32_t id = ldg_i32(ids_t + static_cast<int64_t>(u) * n_entries + grp);
    row = (static_cast<uint64_t>(static_cast<int64_t>(id)) >= static_cast<uint64_t>(un.vocab)) ? 0 : id;
  }
  // The row's strips are loaded BEFORE the claim is known (a row has one claimant per launch except for duplicate ids,
  // whose losers simply drop what they loaded): the claim's atomic round trip and the row's HBM round trip overlap
  // instead of following each other.  A winner's early loads are valid: nobody else writes its row in this launch.
  const int e0 = gl * VEC;
  const bool deep_on = un.deep != nullptr && e0 < un.dim;
  const bool wide_on = un.wide != nullptr && gl == 0;
  const float lam2d = 2.f * un.l2_deep, lam2w = 2.f * un.l2_wide;
  // the table and its first state slab may be strided views of one interleaved slab (row strides in the unit);
  // the second state slab and the gradient slab are contiguous
  const int64_t off = deep_on ? row * un.dim + e0 : 0;
  const int64_t off_w = deep_on ? row * (un.ld_deep > 0 ? un.ld_deep : un.dim) + e0 : 0;
  const int64_t off_a = deep_on ? row * (un.ld_deep_s1 > 0 ? un.ld_deep_s1 : un.dim) + e0 : 0;
  const int64_t row_w = row * (un.ld_wide > 0 ? un.ld_wide : 1), row_a = row * (un.ld_wide_s1 > 0 ? un.ld_wide_s1 : 1);
  float w[VEC], a[VEC], b[VEC], g[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) w[i] = a[i] = b[i] = g[i] = 0.f;
  if (deep_on) {
    load_vec<VEC>(un.deep + off_w, w);
    if (un.deep_s1) load_vec<VEC>(un.deep_s1 + off_a, a);
    if (un.deep_s2) load_vec<VEC>(un.deep_s2 + off, b);
    if (MODE == 1) load_vec<VEC>(un.deep_g + off, g);
  }
  float ww[1] = {0.f}, wa[1] = {0.f}, wb[1] = {0.f}, wg = 0.f;
  if (wide_on) {
    ww[0] = ldg_f32(un.wide + row_w);
    if (un.wide_s1) wa[0] = ldg_f32(un.wide_s1 + row_a);
    if (un.wide_s2) wb[0] = ldg_f32(un.wide_s2 + row);
    if (MODE == 1) wg = ldg_f32(un.wide_g + row);
  }
  // claim the row: the winner is the only group that touches it in this launch
  const int target = (MODE == 1) ? t + 1 : t;
  int prev = 0;
  if (gl == 0) {
    if (MODE == 2) {
      prev = un.stamp[row];
      if (prev < target) un.stamp[row] = target;
    } else {
      prev = atomicMax(un.stamp + row, target);
    }
  }
  prev = __shfl(prev, (threadIdx.x & 63) & ~(lpr - 1), kWave);
  if (prev >= target) return;

  float ss1 = 0.f, bc1 = 1.f;        // Adam scalars of step t + 1 (apply)
  if (MODE == 1 && o.kind == DCTR_LAZY_ADAM) {
    AdamClock ck;
    ck.start(o, t + 1);
    ss1 = ck.step_size();
    bc1 = ck.bc2_sqrt();
  }
  if (deep_on) {
    replay<VEC>(o, lam2d, prev, t, w, a, b);          // (apply after a catch-up: prev == t, nothing to replay)
    if (MODE == 1) {
      float z[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) z[i] = 0.f;
      store_vec<VEC>(un.deep_g + off, z);             // zero at rest
#pragma unroll
      for (int i = 0; i < VEC; ++i) opt_step(o, g[i] + lam2d * w[i], w[i], a[i], b[i], ss1, bc1);
    }
    store_vec<VEC>(un.deep + off_w, w);
    if (un.deep_s1) store_vec<VEC>(un.deep_s1 + off_a, a);
    if (un.deep_s2) store_vec<VEC>(un.deep_s2 + off, b);
  }
  if (wide_on) {
    replay<1>(o, lam2w, prev, t, ww, wa, wb);
    if (MODE == 1) {
      stg_f32(un.wide_g + row, 0.f);
      opt_step(o, wg + lam2w * ww[0], ww[0], wa[0], wb[0], ss1, bc1);
    }
    stg_f32(un.wide + row_w, ww[0]);
    if (un.wide_s1) stg_f32(un.wide_s1 + row_a, wa[0]);
    if (un.wide_s2) stg_f32(un.wide_s2 + row, wb[0]);
  }
}

__global__ void k_lazy_inc(int32_t* step) { *step += 1; }

// the same optimizer step over a flat DENSE slab (tower weights, Linear.weight, the prediction bias), with an optional
// per-element lambda: g = grad + 2*lambda*p.  T = *step + 1.
__global__ __launch_bounds__(kT) void k_dense_opt_reg(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ s1, float* __restrict__ s2,
                                                      const float* __restrict__ lam, int64_t n,
                                                      const int32_t* __restrict__ step_ptr, OptConst o) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  if (i >= n) return;
  float ss = 0.f, bc = 1.f;
  if (o.kind == DCTR_LAZY_ADAM) {
    AdamClock ck;
    ck.start(o, *(const DCTR_GLOBAL int32_t*)step_ptr + 1);
    ss = ck.step_size();
    bc = ck.bc2_sqrt();
  }
  float w = ldg_f32(p + i);
  float a = s1 ? ldg_f32(s1 + i) : 0.f, b = s2 ? ldg_f32(s2 + i) : 0.f;
  const float gt = ldg_f32(g + i) + (lam ? 2.f * ldg_f32(lam + i) * w : 0.f);
  opt_step(o, gt, w, a, b, ss, bc);
  stg_f32(p + i, w);
  if (s1) stg_f32(s1 + i, a);
  if (s2) stg_f32(s2 + i, b);
}

int check(const dctr_lazy_unit_t* units, int n_units, const int32_t* step, const dctr_lazy_opt_t* opt, int vec,
          int max_dim) {
  if (!units || n_units <= 0 || !step || !opt) return DCTR_EINVAL;
  if (opt->kind != DCTR_LAZY_SGD && opt->kind != DCTR_LAZY_ADAGRAD && opt->kind != DCTR_LAZY_ADAM &&
      opt->kind != DCTR_LAZY_RMSPROP)
    return DCTR_EINVAL;
  if (vec != 1 && vec != 4) return DCTR_EINVAL;
  if (max_dim < 1 || max_dim > 64 * vec) return DCTR_ENOSUP;
  return DCTR_OK;
}

template <int MODE>
int launch(const dctr_lazy_unit_t* units, int n_units, const int32_t* ids_t, int64_t n_entries, const int32_t* step,
           const dctr_lazy_opt_t* opt, int vec, int max_dim, hipStream_t s) {
  const int rc = check(units, n_units, step, opt, vec, max_dim);
  if (rc != DCTR_OK) return rc;
  if (n_entries <= 0) return DCTR_OK;
  int lpr = 1, shift = 0;
  while (lpr * vec < max_dim) {
    lpr <<= 1;
    ++shift;
  }
  OptConst o;
  o.kind = opt->kind; o.lr = opt->lr; o.eps = opt->eps; o.beta1 = opt->beta1; o.beta2 = opt->beta2;
  const int64_t threads = n_entries << shift;
  const dim3 grid(static_cast<unsigned>((threads + kT - 1) / kT), static_cast<unsigned>(n_units));
  if (vec == 4)
    k_lazy<4, MODE><<<grid, dim3(kT), 0, s>>>(units, n_units, ids_t, n_entries, shift, step, o);
  else
    k_lazy<1, MODE><<<grid, dim3(kT), 0, s>>>(units, n_units, ids_t, n_entries, shift, step, o);
  return launch_status();
}

}  // namespace

extern "C" size_t dctr_sizeof_lazy_unit(void) { return sizeof(dctr_lazy_unit_t); }

extern "C" int dctr_lazy_catchup(const dctr_lazy_unit_t* units, int32_t n_units, const int32_t* ids_t, int32_t B,
                                 const int32_t* step, const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim,
                                 dctr_stream_t stream) {
  if (B < 0 || (B > 0 && !ids_t)) return DCTR_EINVAL;
  return launch<0>(units, n_units, ids_t, B, step, opt, vec, max_dim, static_cast<hipStream_t>(stream));
}

extern "C" int dctr_lazy_apply(const dctr_lazy_unit_t* units, int32_t n_units, const int32_t* ids_t, int32_t B,
                               const int32_t* step, const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim,
                               dctr_stream_t stream) {
  if (B < 0 || (B > 0 && !ids_t)) return DCTR_EINVAL;
  return launch<1>(units, n_units, ids_t, B, step, opt, vec, max_dim, static_cast<hipStream_t>(stream));
}

extern "C" int dctr_lazy_flush(const dctr_lazy_unit_t* units, int32_t n_units, int64_t max_vocab, const int32_t* step,
                               const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim, dctr_stream_t stream) {
  if (max_vocab < 0) return DCTR_EINVAL;
  return launch<2>(units, n_units, nullptr, max_vocab, step, opt, vec, max_dim, static_cast<hipStream_t>(stream));
}

extern "C" int dctr_dense_opt_reg(float* p, const float* g, float* s1, float* s2, const float* lam, int64_t n,
                                  const dctr_lazy_opt_t* opt, const int32_t* step, dctr_stream_t stream) {
  if (!p || !g || n < 0 || !opt || !step) return DCTR_EINVAL;
  if ((opt->kind == DCTR_LAZY_ADAGRAD || opt->kind == DCTR_LAZY_RMSPROP) && !s1) return DCTR_EINVAL;
  if (opt->kind == DCTR_LAZY_ADAM && (!s1 || !s2)) return DCTR_EINVAL;
  if (opt->kind != DCTR_LAZY_SGD && opt->kind != DCTR_LAZY_ADAGRAD && opt->kind != DCTR_LAZY_ADAM &&
      opt->kind != DCTR_LAZY_RMSPROP)
    return DCTR_EINVAL;
  if (n == 0) return DCTR_OK;
  OptConst o;
  o.kind = opt->kind; o.lr = opt->lr; o.eps = opt->eps; o.beta1 = opt->beta1; o.beta2 = opt->beta2;
  k_dense_opt_reg<<<dim3(static_cast<unsigned>((n + kT - 1) / kT)), dim3(kT), 0, static_cast<hipStream_t>(stream)>>>(
      p, g, s1, s2, lam, n, step, o);
  return launch_status();
}

extern "C" int dctr_lazy_step_inc(int32_t* step, dctr_stream_t stream) {
  if (!step) return DCTR_EINVAL;
  k_lazy_inc<<<dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream)>>>(step);
  return launch_status();
}

// lazy_opt.hpp -- the optimizer arithmetic of the lazily replayed table update (csrc/lazy.hip), shared with the sorted
// embedding update (csrc/update_kernels.hpp, DCTR_UPD_LAZY: round 6 -- the step that carries the batch's data gradient runs
// inside the update kernel instead of a gradient slab + a second pass).  See lazy.hip for the protocol.
#pragma once
#include "common.hpp"

namespace dctr_lazy {
using namespace dctr;

#ifdef DCTR_LAZY_IEEE_REPLAY
constexpr bool kFastReplay = false;   // (A/B build: the replay loop on IEEE division / square root)
#else
constexpr bool kFastReplay = true;
#endif

struct OptConst {
  int kind;   // DCTR_LAZY_SGD / ADAGRAD / ADAM / RMSPROP
  float lr, eps, beta1, beta2;
  // Adam's step-dependent scalars by step number (dctr_lazy_opt_t: adam_ss[T - 1], adam_bc[T - 1]; constant from the last
  // entry on), or NULL: computed in the kernel (AdamClock)
  const float* adam_ss;
  const float* adam_bc;
  const float* adam_rbc;   // 1 / adam_bc (n_bc entries) or NULL
  int n_ss, n_bc;
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

// Adam's scalars of step T from the host's tables.  At the Criteo shape a row sleeps V / B ~ 244 steps between two batches
// that touch it and its catch-up replays every one of them: with the scalars computed in the loop (two double
// multiplies, a double division and a double square root per step and lane) that arithmetic was ~40 % of k_lazy's
// 282 us; they depend on T alone, so the host tabulates them once -- with the same pow / division / sqrt in double that
// torch.optim.Adam performs per step on the host (adam.py: bias_correction = 1 - beta ** step).
__device__ __forceinline__ void adam_tab(const OptConst& o, int T, float& ss, float& bc) {
  ss = ldg_f32(o.adam_ss + ((T < o.n_ss ? T : o.n_ss) - 1));
  bc = ldg_f32(o.adam_bc + ((T < o.n_bc ? T : o.n_bc) - 1));
}
// the same for a wave-uniform T: plain loads the compiler can issue on the scalar unit (no per-lane address arithmetic)
__device__ __forceinline__ void adam_tab_uniform(const OptConst& o, int T, float& ss, float& bc) {
  ss = o.adam_ss[(T < o.n_ss ? T : o.n_ss) - 1];
  bc = o.adam_bc[(T < o.n_bc ? T : o.n_bc) - 1];
}
__device__ __forceinline__ void adam_scalars(const OptConst& o, int T, float& ss, float& bc) {
  if (o.adam_ss) {
    adam_tab(o, T, ss, bc);
  } else {
    AdamClock ck;
    ck.start(o, T);
    ss = ck.step_size();
    bc = ck.bc2_sqrt();
  }
}

// Division and square root of the REPLAY loop.  A row that slept k steps replays k optimizer steps whose only gradient is
// the L2 term; every row owes one such step per train step (whoever pays it: catch-up, sweep or flush), so at the Criteo
// shape this loop is 26 M row-steps x 16 elements per train step and its instruction count IS the cost of the
// default-kwargs step: the chip's vector ALUs are saturated by it (sweep alone: 231 us of a 0.44 ms step).
//   round 5: IEEE division / square root (v_div_scale x 2, v_rcp, five fmas, v_div_fmas, v_div_fixup each: ~30 of ~50
//            instructions per element) -> hardware reciprocal / reciprocal square root + ONE Newton correction through
//            the exact residual (<= 1 ulp): 0.62 -> 0.46 ms.
//   round 6: the Newton corrections go too (DCTR_LAZY_REPLAY_NR=1 at build time brings them back): q = a * rcp(d),
//            s = x * rsq(x).  v_rcp_f32 / v_rsq_f32 are accurate to 1 ulp, the product rounds once more: <= 1.5 ulp per
//            operation, three of them in a step's update lr_t * m / (sqrt(v) / bc + eps) -- a relative error of ~3e-7 of
//            a STEP (itself ~1e-3 of the weight's magnitude), unbiased, next to the 6e-8 every fp32 rounding of the step
//            contributes.  7 of the loop's 17 packed instructions per element pair: sweep 231 -> 195 us, step 0.437 ->
//            0.401 ms on one box (tools/runs/nonr_ab.sh).  Every test of the lazy update and every golden trajectory of
//            the reference with adam + L2 passes unchanged with either build (same bars).  Then the Adam step written out
//            on packed pairs (adam_replay_step below): 15 packed + 18 plain -> 22 packed instructions per trip of four
//            elements, 0.412 -> 0.397 ms (tools/runs/lazy_ab.sh).
// The operands of this loop are ordinary normal numbers (denominators >= eps, moments of magnitude (lambda w)^2); the
// result is the same on every run and for every schedule (who replays a step never changes what the step computes).
// The step that carries a DATA gradient (apply, the sorted update) and the dense slab keep the IEEE operations.
#ifndef DCTR_LAZY_REPLAY_NR
#define DCTR_LAZY_REPLAY_NR 0
#endif
__device__ __forceinline__ float div_nr(float a, float d, float rd) {   // rd = rcp(d)
  const float q = a * rd;
#if DCTR_LAZY_REPLAY_NR
  return fmaf(fmaf(-q, d, a), rd, q);
#else
  return q;
#endif
}
__device__ __forceinline__ float sqrt_nr(float x) {
  // (a moment is never negative; x = 0 stays exactly 0 -- 0 * finite -- without a select: the smallest normal number added
  // (a packed add; it vanishes in the rounding of any moment above 2^-102) or clamped to.  A moment below it comes out
  // smaller than its root: it is added to eps = 1e-8 either way)
#if DCTR_LAZY_REPLAY_NR
  const float r = __builtin_amdgcn_rsqf(fmaxf(x, 1.17549435e-38f));
  const float s = x * r;
  return fmaf(fmaf(-s, s, x), 0.5f * r, s);
#else
  return x * __builtin_amdgcn_rsqf(x + 1.17549435e-38f);
#endif
}

// One REPLAYED Adam step (gradient = the L2 term lam2 * w alone) on one element and on a packed pair: the same operations in
// the same order with every fused multiply-add written out and contraction off, so that whoever replays a (row, step) --
// catch-up, sweep, flush; one lane or a packed pair -- computes the same bits (the compiler contracts `a * b + c` in one
// kernel and not in the next: round-6 finding on the dense optimizer, common.hpp).  11 vector instructions + 2
// transcendentals per element (9 + 2 with the hardware square root instead of v * rsq(v + tiny)); packed: per PAIR.  Left to the compiler the loop's third that sits between the reciprocal
// square roots and the reciprocals (`+ tiny`, `x * rsq`, `fma(.., rbc, eps)`, the quotient) stayed unpacked: 15 packed + 18
// plain instructions per trip of four elements, now 22 packed.
//   g = lam2 w ;  m += (g - m)(1 - beta1) ;  v = v beta2 + ((1 - beta2) g) g ;  s = sqrt(v) (the 1-ulp instruction) ;
//   den = s / bc + eps  (as fma(s, 1 / bc, eps)) ;  w -= ss (m rcp(den))
template <typename T>
__device__ __forceinline__ T rsq_t(T x);
template <>
__device__ __forceinline__ float rsq_t<float>(float x) { return __builtin_amdgcn_rsqf(x); }
template <>
__device__ __forceinline__ f32x2 rsq_t<f32x2>(f32x2 x) { return f32x2{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
template <typename T>
__device__ __forceinline__ T sqrt_t(T x);
template <>
__device__ __forceinline__ float sqrt_t<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <>
__device__ __forceinline__ f32x2 sqrt_t<f32x2>(f32x2 x) { return f32x2{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)}; }
template <typename T>
__device__ __forceinline__ T rcp_t(T x);
template <>
__device__ __forceinline__ float rcp_t<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <>
__device__ __forceinline__ f32x2 rcp_t<f32x2>(f32x2 x) { return f32x2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f32x2 fma_t(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <typename T>
__device__ __forceinline__ void adam_replay_step(float c1, float c2, float beta2, float eps, T g, float ss, float rbc,
                                                 T& w, T& m, T& v) {
#pragma clang fp contract(off)
  m = fma_t(g - m, T(c1), m);
  v = fma_t(g * c2, g, v * beta2);
  const T s = sqrt_t<T>(v);        // (v_sqrt_f32: 1 ulp, exact 0 at 0 -- no clamp and no multiply as with v * rsq(v + tiny))
  const T den = fma_t(s, T(rbc), T(eps));
  const T q = m * rcp_t<T>(den);
  w = fma_t(T(-ss), q, w);
}

// one optimizer step on one element.  a = Adagrad sum | Adam exp_avg, b = Adam exp_avg_sq.  FAST: the replay loop's
// division / square root (above); rbc = rcp(bc2s), computed once per step by the caller.
template <bool FAST = false>
__device__ __forceinline__ void opt_step(const OptConst& o, float g, float& w, float& a, float& b, float step_size,
                                         float bc2s, float rbc = 0.f) {
  if (o.kind == DCTR_LAZY_ADAM && FAST && !DCTR_LAZY_REPLAY_NR) {
    adam_replay_step<float>(1.f - o.beta1, 1.f - o.beta2, o.beta2, o.eps, g, step_size, rbc, w, a, b);
  } else if (o.kind == DCTR_LAZY_ADAM) {
    a = a + (g - a) * (1.f - o.beta1);
    b = b * o.beta2 + (1.f - o.beta2) * g * g;
    if (FAST) {
      const float denom = div_nr(sqrt_nr(b), bc2s, rbc) + o.eps;
      w = w - step_size * div_nr(a, denom, __builtin_amdgcn_rcpf(denom));
    } else {
      const float denom = sqrtf(b) / bc2s + o.eps;
      w = w - step_size * (a / denom);
    }
  } else if (o.kind == DCTR_LAZY_ADAGRAD) {
    a = a + g * g;
    if (FAST) {
      const float denom = sqrt_nr(a) + o.eps;
      w = w - o.lr * div_nr(g, denom, __builtin_amdgcn_rcpf(denom));
    } else {
      w = w - o.lr * (g / (sqrtf(a) + o.eps));
    }
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


inline OptConst opt_const(const dctr_lazy_opt_t* opt) {
  OptConst o;
  o.kind = opt->kind; o.lr = opt->lr; o.eps = opt->eps; o.beta1 = opt->beta1; o.beta2 = opt->beta2;
  const bool tab = opt->kind == DCTR_LAZY_ADAM && opt->adam_ss && opt->adam_bc && opt->n_ss > 0 && opt->n_bc > 0;
  o.adam_ss = tab ? opt->adam_ss : nullptr;
  o.adam_bc = tab ? opt->adam_bc : nullptr;
  o.adam_rbc = tab ? opt->adam_rbc : nullptr;
  o.n_ss = tab ? opt->n_ss : 0;
  o.n_bc = tab ? opt->n_bc : 0;
  return o;
}

}  // namespace dctr_lazy

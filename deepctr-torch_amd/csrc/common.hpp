// common.hpp -- shared device helpers for the gfx950 kernels of libdctr_hip.so.
// Wave size is 64 on CDNA4; every kernel here is written for 64-lane wavefronts.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dctr.h"

namespace dctr {

constexpr int kWave = 64;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A short per-lane strip of VEC consecutive floats (VEC in {1,2,4,8}); loads/stores are one
// global_load/store_dword{,x2,x4} (two dwordx4 for VEC = 8).
template <int VEC>
struct Strip {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ Strip<VEC> strip_zero() {
  Strip<VEC> r;
#pragma unroll
  for (int i = 0; i < VEC; ++i) r.v[i] = 0.f;
  return r;
}

// Every pointer these helpers see addresses global (HBM) memory, but pointers that were parked in
// LDS come back as generic pointers and would be accessed with FLAT instructions, which tick the
// LGKM counter as well and so serialise against every later LDS read.  Casting to address space 1
// makes them global_load / global_store / global_atomic.
#define DCTR_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ float ldg_f32(const float* p) {
  return *(const DCTR_GLOBAL float*)p;
}
__device__ __forceinline__ int32_t ldg_i32(const int32_t* p) {
  return *(const DCTR_GLOBAL int32_t*)p;
}
__device__ __forceinline__ void stg_f32(float* p, float v) { *(DCTR_GLOBAL float*)p = v; }

template <int VEC>
__device__ __forceinline__ Strip<VEC> strip_load(const float* p) {
  Strip<VEC> r;
  if constexpr (VEC == 8) {
    f32x4 t = *(const DCTR_GLOBAL f32x4*)p;
    f32x4 u = *(const DCTR_GLOBAL f32x4*)(p + 4);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    r.v[4] = u.x; r.v[5] = u.y; r.v[6] = u.z; r.v[7] = u.w;
  } else if constexpr (VEC == 4) {
    f32x4 t = *(const DCTR_GLOBAL f32x4*)p;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else if constexpr (VEC == 2) {
    f32x2 t = *(const DCTR_GLOBAL f32x2*)p;
    r.v[0] = t.x; r.v[1] = t.y;
  } else {
    r.v[0] = *(const DCTR_GLOBAL float*)p;
  }
  return r;
}

template <int VEC>
__device__ __forceinline__ void strip_store(float* p, const Strip<VEC>& s) {
  if constexpr (VEC == 8) {
    f32x4 t = {s.v[0], s.v[1], s.v[2], s.v[3]};
    f32x4 u = {s.v[4], s.v[5], s.v[6], s.v[7]};
    *(DCTR_GLOBAL f32x4*)p = t;
    *(DCTR_GLOBAL f32x4*)(p + 4) = u;
  } else if constexpr (VEC == 4) {
    f32x4 t = {s.v[0], s.v[1], s.v[2], s.v[3]};
    *(DCTR_GLOBAL f32x4*)p = t;
  } else if constexpr (VEC == 2) {
    f32x2 t = {s.v[0], s.v[1]};
    *(DCTR_GLOBAL f32x2*)p = t;
  } else {
    *(DCTR_GLOBAL float*)p = s.v[0];
  }
}

// ---- write-through stores + the signal half of dctr_step_wait (include/dctr.h) -----------------------------------------
// sc0 sc1: the bytes leave this XCD's L2 for memory as they are stored, not at the end-of-kernel write-back; s_waitcnt
// vmcnt(0) then means they have arrived.  (Inline asm: the compiler's own waitcnt bookkeeping does not see these stores,
// which only makes its later waits conservative -- memory operations return in order.)
__device__ __forceinline__ void stg_wt(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void stg_wt(float* p, f32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void stg_wt(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// plain or write-through, by a wave-uniform flag
__device__ __forceinline__ void stg_f32(float* p, float v, bool wt) {
  if (wt) stg_wt(p, v);
  else *(DCTR_GLOBAL float*)p = v;
}
template <int VEC>
__device__ __forceinline__ void strip_store(float* p, const Strip<VEC>& s, bool wt) {
  if (!wt) {
    strip_store<VEC>(p, s);
  } else if constexpr (VEC == 8) {
    stg_wt(p, f32x4{s.v[0], s.v[1], s.v[2], s.v[3]});
    stg_wt(p + 4, f32x4{s.v[4], s.v[5], s.v[6], s.v[7]});
  } else if constexpr (VEC == 4) {
    stg_wt(p, f32x4{s.v[0], s.v[1], s.v[2], s.v[3]});
  } else if constexpr (VEC == 2) {
    stg_wt(p, f32x2{s.v[0], s.v[1]});
  } else {
    stg_wt(p, s.v[0]);
  }
}
// Every thread of the workgroup calls this once, behind its last write-through store: the stores have arrived, the
// workgroup counts itself done, and the launch's last workgroup advances the signal's generation.  (Relaxed agent-scope
// atomics: they execute beyond the XCD's L2; a release here would write back the whole L2, ~2-6 us per workgroup.)
__device__ __forceinline__ void step_signal(int32_t* sync, int signal) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t* gen = sync + 4 * signal;
    const int32_t n = static_cast<int32_t>(gridDim.x * gridDim.y * gridDim.z);
    if (__hip_atomic_fetch_add(gen + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1) {
      __hip_atomic_store(gen + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      gen[3] = static_cast<int32_t>(wall_clock64());     // (when: for tools/step_hops.py)
      __hip_atomic_fetch_add(gen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// fp32 hardware atomic add without return (global_atomic_add_f32; needs -munsafe-fp-atomics).
// Tables are ordinary coarse-grained hipMalloc memory owned by PyTorch, where it is valid.
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  __hip_atomic_fetch_add((DCTR_GLOBAL float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float atomic_xchg_f32(float* p, float v) {
  return __hip_atomic_exchange((DCTR_GLOBAL float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum over the LPR consecutive lanes that form one sample group (LPR is a power of two <= 64)
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// row strides of a field's table / optimizer state (0 in the descriptor = contiguous rows of `dim` floats)
__device__ __forceinline__ int64_t row_ld(const dctr_field_t& f) { return f.ld > 0 ? f.ld : f.dim; }
__device__ __forceinline__ int64_t state_ld(const dctr_field_t& f) { return f.ld_state > 0 ? f.ld_state : f.dim; }

// dctr_dense_step_t by value for kernels (kind < 0: no step) and the step itself on the element behind a gradient
struct DenseStepDev {
  int kind;
  float lr, eps;
  const float* grad_base;
  float* param_base;
  float* state_base;
};
inline DenseStepDev dense_step_dev(const dctr_dense_step_t* s) {
  DenseStepDev d;
  d.kind = s ? s->kind : -1;
  d.lr = s ? s->lr : 0.f;
  d.eps = s ? s->eps : 0.f;
  d.grad_base = s ? s->grad_base : nullptr;
  d.param_base = s ? s->param_base : nullptr;
  d.state_base = s ? s->state_base : nullptr;
  return d;
}
// The dense optimizers' arithmetic, ONE definition for every kernel that steps a dense parameter (k_mlp_reduce, the reduction
// folded into k_mlp_wgrad, k_sum_ranks, the update kernel's Linear.weight role, ...): products and sums rounded separately, in
// the order of the reference's formulas (torch.optim.Adagrad: s += g * g; p -= lr * g / (sqrt(s) + eps); SGD: p -= lr * g).
// Contraction is switched off on purpose: left to the compiler, `w - lr * q` became v_fma in one copy of this code and
// v_mul + v_sub in the next (even between the four elements of one thread: round 6, ISA of k_mlp_reduce), so two kernels
// that must agree bit for bit -- the engine's folded reduction and the separate launch -- did not.
__device__ __forceinline__ float adagrad_sum(float st, float g) {
#pragma clang fp contract(off)
  const float gg = g * g;
  return st + gg;
}
__device__ __forceinline__ float adagrad_param(float w, float g, float sn, float lr, float eps) {
#pragma clang fp contract(off)
  const float q = g / (sqrtf(sn) + eps);
  const float d = lr * q;
  return w - d;
}
// (SGD: ONE rounding, p + (-lr) * g fused -- what `p.add_(g, alpha=-lr)` of torch.optim.SGD computes on the GPU and, through
// Vectorized::fmadd, on the reference's CPU; tests/test_gpu_dense_multi.py holds this to 2e-7 of torch.optim)
__device__ __forceinline__ float sgd_param(float w, float g, float lr) { return __builtin_fmaf(-lr, g, w); }
__device__ __forceinline__ void dense_step_apply(const DenseStepDev& S, const float* gptr, float g) {
  if (S.kind < 0) return;
  const int64_t k = gptr - S.grad_base;
  float w = ldg_f32(S.param_base + k);
  if (S.kind == DCTR_UPD_ADAGRAD) {   // torch.optim.Adagrad: s += g*g ; p -= lr * g / (sqrt(s) + eps)
    const float st = adagrad_sum(ldg_f32(S.state_base + k), g);
    stg_f32(S.state_base + k, st);
    w = adagrad_param(w, g, st, S.lr, S.eps);
  } else {                            // torch.optim.SGD
    w = sgd_param(w, g, S.lr);
  }
  stg_f32(S.param_base + k, w);
}

inline int hip_status(hipError_t e) { return e == hipSuccess ? DCTR_OK : static_cast<int>(e); }

inline int launch_status() { return hip_status(hipGetLastError()); }


// Issue priority of the launches on a train step's critical chain (gather, tower, update, catch-up ...).  The lazily
// replayed tables' per-step SWEEP (lazy.hip, k_lazy<., 3>) is ~230 us of packed arithmetic that runs beside the chain on
// its own queue and leaves its waves at the default priority 0: a SIMD's arbiter then issues the chain's (latency-bound,
// few instructions between loads) waves ahead of it, and the sweep fills the cycles the chain leaves.  Without this
// every chain instruction queued behind up to six sweep waves and the step cost sweep + chain.  No effect when a launch
// has the SIMDs to itself.
// (the tower's weight gradients at a LOWER level than the embedding update they run beside: the update gets 1.7 us
// shorter, 21.6 -> 19.9, and the step 1.6 us longer, 0.0898 -> 0.0914 ms -- the two chains are balanced: tools/runs/lib_ab.sh)
#ifndef DCTR_WGRAD_PRIORITY
#define DCTR_WGRAD_PRIORITY 3
#endif
template <int LEVEL = 3>
__device__ __forceinline__ void step_priority() {
#ifndef DCTR_NO_STEP_PRIORITY
  __builtin_amdgcn_s_setprio(LEVEL);
#endif
}
}  // namespace dctr

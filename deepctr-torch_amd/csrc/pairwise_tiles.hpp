// pairwise_tiles.hpp -- what the pair-structured kernels share (pairwise.hip, bilinear_wide.hip): the 16-sample row
// tile in LDS, the tournament schedule entry, the raw 16x16 weight-tile operands of v_mfma_f32_16x16x4_f32.
#pragma once
#include <type_traits>

#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kT = 256;
constexpr int kSB = 16;  // samples per workgroup (= MFMA rows)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int row_stride(int F, int D) {
  // floats per sample row in LDS; (RS mod 32) == 16 spreads consecutive samples over both bank halves
  int rs = F * D;
  rs += (16 - (rs & 31)) & 31;
  return rs;
}

// copy the [16, F*D] tile of samples b0.. (b0 < B) into LDS (zeros past B).  Eight UNCONDITIONAL loads per thread are
// in flight at a time (rows past B are clamped to B-1 and masked afterwards): a predicated load compiles to a
// branch with its own s_waitcnt vmcnt(0), which made this copy 26 serial memory round trips.
// The same copy in dwordx4 pieces, split in its two halves so that a caller can put the loads of several tiles (and
// whatever else it needs from memory) in flight TOGETHER and pay one round trip: rows_load4 / rows_store4.  Needs
// rows_vec_ok(); U * kT float4 must cover the tile (U = 8: F*D <= 512 at 16 rows).  (Round 3: the scalar copy below
// takes ceil(F*D/128) dependent round trips per array -- 4 + 4 per tile at the Criteo shape, ~12 of the ~15 us
// k_bilinear_bwd_weight spent per 16-sample tile.)
__device__ __forceinline__ bool rows_vec_ok(int rows, int RS, const float* src, int64_t ld, int W, int U) {
  return (W & 3) == 0 && (ld & 3) == 0 && (RS & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
         rows * (W >> 2) <= U * kT;
}
template <int U>
__device__ __forceinline__ void rows_load4(const float* __restrict__ src, int64_t ld, int b0, int B, int rows, int W,
                                           f32x4 (&v)[U]) {
  const int w4 = W >> 2, n4 = rows * w4;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = threadIdx.x + u * kT;
    const int ec = e < n4 ? e : 0;
    const int r = ec / w4, c = ec - r * w4;
    const int rr = b0 + r < B ? b0 + r : B - 1;
    v[u] = *(const DCTR_GLOBAL f32x4*)(src + static_cast<int64_t>(rr) * ld + 4 * c);
  }
}
template <int U>
__device__ __forceinline__ void rows_store4(float* dst, int RS, int b0, int B, int rows, int W, const f32x4 (&v)[U]) {
  const int w4 = W >> 2, n4 = rows * w4;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = threadIdx.x + u * kT;
    if (e < n4) {
      const int r = e / w4, c = e - r * w4;
      *reinterpret_cast<f32x4*>(dst + r * RS + 4 * c) = (b0 + r < B) ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

template <int ROWS = kSB>
__device__ __forceinline__ void stage_rows(float* dst, int RS, const float* __restrict__ src, int64_t ld, int b0,
                                           int B, int W) {
  if (rows_vec_ok(ROWS, RS, src, ld, W, 8)) {      // (uniform)
    f32x4 v[8];
    rows_load4<8>(src, ld, b0, B, ROWS, W, v);
    rows_store4<8>(dst, RS, b0, B, ROWS, W, v);
    return;
  }
  const int n = ROWS * W;
  for (int e0 = threadIdx.x; e0 < n; e0 += 8 * kT) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * kT;
      const int ec = e < n ? e : 0;
      const int r = ec / W, c = ec - r * W;
      const int rr = b0 + r < B ? b0 + r : B - 1;
      v[u] = ldg_f32(src + static_cast<int64_t>(rr) * ld + c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * kT;
      if (e < n) {
        const int r = e / W, c = e - r * W;
        dst[r * RS + c] = (b0 + r < B) ? v[u] : 0.f;
      }
    }
  }
}

// One pair of the tournament schedule: {i, j, weight index, pair index k}; i < 0 = idle slot (its w and k are 0).
struct PairEnt {
  int i, j, wi, k;
};
typedef int i32x4 __attribute__((ext_vector_type(4)));
// entry q of the schedule as ONE 16-byte load; q past the end reads the last entry and is marked idle
__device__ __forceinline__ PairEnt load_pair(const int32_t* __restrict__ sched, int q, int n_sched) {
  const int qc = q < n_sched ? q : n_sched - 1;
  const i32x4 v = *(const DCTR_GLOBAL i32x4*)(sched + 4 * qc);
  PairEnt e;
  e.i = q < n_sched ? v.x : -1;
  e.j = v.y;
  e.wi = v.z;
  e.k = v.w;
  return e;
}
// raw weight-tile operands of a pair for lane (g, c): w[s] = W[e = c][d = 4g + s], wt[s] = W[e = 4g + s][d = c];
// lanes / steps outside D read element 0 of the tile and are masked by the consumer
__device__ __forceinline__ void load_w_raw(const float* __restrict__ Wf, const PairEnt& e, int D, int g, int c,
                                           float (&w)[4]) {
  const float* base = Wf + static_cast<int64_t>(e.wi) * D * D;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int d = 4 * g + s;
    w[s] = ldg_f32(base + ((c < D && d < D) ? c * D + d : 0));
  }
}
__device__ __forceinline__ void load_wt_raw(const float* __restrict__ Wf, const PairEnt& e, int D, int g, int c,
                                            float (&wt)[4]) {
  const float* base = Wf + static_cast<int64_t>(e.wi) * D * D;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int d = 4 * g + s;
    wt[s] = ldg_f32(base + ((c < D && d < D) ? d * D + c : 0));
  }
}


}  // namespace

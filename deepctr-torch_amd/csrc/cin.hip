// cin.hip -- Compressed Interaction Network layer (xDeepFM) on the gfx950 matrix cores, fp32 in / fp32 out.
//
// Reference (interaction.py:207-248), one layer:
//     Z[b, h*M + m, d] = H[b, h, d] * X0[b, m, d]                 (einsum 'bhd,bmd->bhmd', MATERIALISED: 436 MB at
//     Y[b, o, d]       = sum_k W[o, k] Z[b, k, d] + bias[o]        the Criteo shape for layer 1)
//     A                = relu(Y)
// GEMM view: columns c = (b, d) (B*D = 65536 of them), K = h*M (676 / 1664), rows o (128).  It is genuinely dense,
// so it runs on v_mfma_f32_32x32x2_f32 -- exact fp32 (the 1e-5 logit bar rules out bf16 inputs).
// Z is never formed in memory: the B operand of every MFMA is produced by ONE v_mul per lane,
//     Z[k = (h, m), c] = H[b, h, d] * X0[b, m, d],
// with the lane's column fixed for the whole kernel, H[b, h, d] held in a register per h and X0[b, :, d] in LDS.
//
// Forward mapping: workgroup = 4 waves = 256 columns (16 samples at D=16), one wave per SIMD; a wave owns 2 column
// tiles x OT row tiles of 32x32 (8 independent accumulators at O=128 -> the matrix pipe issues back to back).
// Loop over h: the 13 KB weight slice Wt[h] (prepared as [h][M_pad][O_pad] so rows are contiguous) is double
// buffered in LDS, fetched for h+1 while h computes; one barrier per h.  B=4096 -> 256 workgroups = one per CU.
#include <cstdlib>

#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kT = 256;       // threads per workgroup
constexpr int kCols = 256;    // columns per workgroup

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of accumulator register r held by a lane of half p (32x32 C/D layout)
__device__ __forceinline__ int acc_row(int r, int p) { return (r & 3) + 8 * (r >> 2) + 4 * p; }

// W [O, h*M] (conv1ds.k.weight squeezed) -> Wt [h][M_pad][O_pad], zero padded
// sym (layer 1: H IS X0, so Z[(h, m)] = Z[(m, h)]): the two weights of a field pair are folded onto its m > h entry,
//     Wt[h][m] = W[h][m] + W[m][h] (m > h),  W[h][h] (m == h),  0 (m < h)
// and the forward walks only m >= h: 351 of the 676 products at 26 fields.
__global__ __launch_bounds__(kT) void k_cin_prep_w(const float* __restrict__ W, int O, int h, int M, int M_pad,
                                                   int O_pad, float* __restrict__ Wt, int sym) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  const int64_t total = static_cast<int64_t>(h) * M_pad * O_pad;
  if (idx >= total) return;
  const int o = static_cast<int>(idx % O_pad);
  const int mm = static_cast<int>((idx / O_pad) % M_pad);
  const int hh = static_cast<int>(idx / (static_cast<int64_t>(O_pad) * M_pad));
  float w = 0.f;
  if (o < O && mm < M) {
    const float* row = W + static_cast<int64_t>(o) * h * M;
    if (!sym) w = row[hh * M + mm];
    else if (mm > hh) w = row[hh * M + mm] + row[mm * M + hh];
    else if (mm == hh) w = row[hh * M + hh];
  }
  Wt[idx] = w;
}

// CT column tiles per wave: 2 = four waves per workgroup (one per SIMD, 8 * OT / 2 ... accumulators each), 1 = EIGHT waves
// (two per SIMD, half the accumulators): the second wave on a SIMD issues its MFMAs while the first waits for its LDS
// operands at the top of a k-step or at the per-h barrier (round 3: at CT = 2 and OT = 4 the kernel holds 276 registers,
// i.e. one wave per SIMD, and the matrix pipe idles through every such wait: 70 % of its rate).
template <int OT, int CT>
__global__ __launch_bounds__(kT * 2 / CT, 1) void k_cin_fwd(const float* __restrict__ X0, int64_t ldx0,
                                                   const float* __restrict__ H, int64_t ldh, int h, int M,
                                                   int M_pad, int D, int B, const float* __restrict__ Wt,
                                                   int O_pad, int O, const float* __restrict__ bias, int relu,
                                                   float* __restrict__ A, int64_t lda, int sym) {
  constexpr int OB = OT * 32;
  constexpr int NT = kT * 2 / CT;       // threads per workgroup
  constexpr int WQ = (OT * kT + NT - 1) / NT;   // float4 of a weight slice per thread
  extern __shared__ __align__(16) float smem[];
  float* x0s = smem;                    // [M_pad][256]
  float* ws = x0s + M_pad * kCols;      // [2][M_pad][OB]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, p = lane >> 5, jl = lane & 31;
  const int64_t c_base = static_cast<int64_t>(blockIdx.x) * kCols;
  const int o_base = blockIdx.y * OB;      // (every y-chunk of a launch has this launch's OT row tiles)
  const int64_t ncol = static_cast<int64_t>(B) * D;
  const int chunk = M_pad * OB;

  int64_t hoff[CT];
  bool cv[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int64_t c = c_base + wv * (32 * CT) + t * 32 + jl;
    cv[t] = c < ncol;
    const int64_t b = cv[t] ? c / D : 0;
    const int d = cv[t] ? static_cast<int>(c - b * D) : 0;
    hoff[t] = b * ldh + d;
  }
  if (tid < kCols) {  // stage X0[b, :, d] of this thread's column
    const int64_t c = c_base + tid;
    const bool v = c < ncol;
    const int64_t b = v ? c / D : 0;
    const int d = v ? static_cast<int>(c - b * D) : 0;
    const float* src = X0 + b * ldx0 + d;
    // unconditional loads on clamped rows, 16 in flight, masked afterwards: a predicated load compiles to a branch
    // with its own s_waitcnt vmcnt(0), i.e. one memory round trip per field
#pragma unroll 1
    for (int m0 = 0; m0 < M_pad; m0 += 16) {
      float t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = ldg_f32(src + (m0 + i < M ? m0 + i : M - 1) * D);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (m0 + i < M_pad) x0s[(m0 + i) * kCols + tid] = (v && m0 + i < M) ? t[i] : 0.f;
    }
  }
  // weight slice of h = 0
  f32x4 wreg[WQ];
  auto fetch_w = [&](int hh) {
    const float* src = Wt + static_cast<int64_t>(hh) * M_pad * O_pad + o_base;
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int e = q * NT + tid;  // float4 index inside the slice
      const int row = e / (OB / 4), c4 = e - row * (OB / 4);
      wreg[q] = (row < M_pad) ? *(const DCTR_GLOBAL f32x4*)(src + static_cast<int64_t>(row) * O_pad + c4 * 4)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto park_w = [&](int buf) {
    float* dst = ws + buf * chunk;
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int e = q * NT + tid;
      const int row = e / (OB / 4);
      if (row < M_pad) *reinterpret_cast<f32x4*>(dst + e * 4) = wreg[q];
    }
  };
  fetch_w(0);
  park_w(0);
  // (columns past the end sit on b = 0, d = 0: their loads are valid, their results are never stored)
  float hv[CT], hnext[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) hv[t] = ldg_f32(H + hoff[t]);
  // hv must have ARRIVED before the loop: if its first use sits inside the loop, the s_waitcnt vmcnt(0) for it is
  // placed there and then also waits, in every iteration, for the prefetch of the next weight slice
#pragma unroll
  for (int t = 0; t < CT; ++t) asm volatile("" : "+v"(hv[t]));
  __syncthreads();

  f32x16 acc[OT][CT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ot][t][r] = 0.f;

  const int xoff = wv * (32 * CT) + jl;
  for (int hh = 0; hh < h; ++hh) {
    const bool more = hh + 1 < h;
    if (more) {
      fetch_w(hh + 1);
#pragma unroll
      for (int t = 0; t < CT; ++t) hnext[t] = ldg_f32(H + hoff[t] + static_cast<int64_t>(hh + 1) * D);
    }
    const float* wc = ws + (hh & 1) * chunk;
    for (int s = sym ? (hh >> 1) : 0; s < M_pad / 2; ++s) {   // (sym: the folded weights of m < h are zero)
      const int mm = 2 * s + p;
      float z[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) z[t] = hv[t] * x0s[mm * kCols + xoff + 32 * t];
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const float a = wc[mm * OB + ot * 32 + jl];
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[ot][t] = mfma32(a, z[t], acc[ot][t]);
      }
    }
    if (more) {
      park_w((hh + 1) & 1);
#pragma unroll
      for (int t = 0; t < CT; ++t) hv[t] = hnext[t];
    }
    __syncthreads();
  }

  // epilogue: + bias, activation, store A[b, o, d].  The bias goes through LDS (the loop's last barrier has retired
  // every read of x0s): 64 dependent scalar loads in this epilogue would be 64 serial round trips.
  if (tid < OB) x0s[tid] = (bias && o_base + tid < O) ? ldg_f32(bias + o_base + tid) : 0.f;
  __syncthreads();
  int64_t bcol[CT];
  int dcol[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int64_t c = c_base + wv * (32 * CT) + t * 32 + jl;
    bcol[t] = cv[t] ? c / D : 0;
    dcol[t] = cv[t] ? static_cast<int>(c - bcol[t] * D) : 0;
  }
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o_base + ot * 32 + acc_row(r, p);
      if (o < O) {
        const float bo = x0s[ot * 32 + acc_row(r, p)];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          if (cv[t]) {
            float y = acc[ot][t][r] + bo;
            if (relu) y = y > 0.f ? y : 0.f;
            stg_f32(A + bcol[t] * lda + static_cast<int64_t>(o) * D + dcol[t], y);
          }
        }
      }
    }
  }
}


// -------------------------------------------------------------------------------------------------------------
// backward, data side:  gZ[k, c] = sum_o W[o, k] gY[o, c]  (never stored), then per column c = (b, d)
//     gH[b, h, d]   = sum_m gZ[(h, m), c] X0[b, m, d]          gX0[b, m, d] += sum_h gZ[(h, m), c] H[b, h, d]
// MFMA roles: rows i = m (the 26 fields of one h, padded to 32), reduction = o, columns = c.  The lane's gY values
// (B operand, reused by every h) live in registers for the whole kernel; the weight slice W[:, h*M .. h*M+M) is
// double buffered in LDS as [o][32].  The product-rule tails are per-lane FMAs on the accumulator registers.
// -------------------------------------------------------------------------------------------------------------
// CT: column tiles per wave (see k_cin_fwd): 2 = four waves, 1 = eight waves of half the registers (at CT = 2, OT = 4 the
// lane's gY values alone are 128 registers and the kernel holds 356: one wave per SIMD)
template <int OT, int CT>
__global__ __launch_bounds__(kT * 2 / CT, 1) void k_cin_bwd_data(const float* __restrict__ gA, const float* __restrict__ Asv,
                                                        int64_t lda, int relu, const float* __restrict__ X0,
                                                        int64_t ldx0, const float* __restrict__ H, int64_t ldh,
                                                        int h, int M, int D, int B, const float* __restrict__ W,
                                                        int O, float* __restrict__ gH, int64_t ldgh, int acc_h,
                                                        float* __restrict__ gX0, int64_t ldgx, int acc_x) {
  constexpr int NT = kT * 2 / CT;                          // threads per workgroup
  constexpr int OB = OT * 32, NS = OT * 16, NW = OT * 32 * 32 / NT;  // NW dwords of the weight slice per thread
  extern __shared__ __align__(16) float smem[];
  float* wl = smem;  // [2][OB][32]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, p = lane >> 5, jl = lane & 31;
  const int64_t c_base = static_cast<int64_t>(blockIdx.x) * kCols;
  const int64_t ncol = static_cast<int64_t>(B) * D;
  const int K = h * M;

  int64_t bb[CT];
  int dd[CT];
  bool cv[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int64_t c = c_base + wv * (32 * CT) + t * 32 + jl;
    cv[t] = c < ncol;
    bb[t] = cv[t] ? c / D : 0;
    dd[t] = cv[t] ? static_cast<int>(c - bb[t] * D) : 0;
  }
  // gY of this lane's columns: o = 2*s + p.  Unconditional loads on clamped addresses, masked afterwards (a
  // predicated load is a branch with its own s_waitcnt vmcnt(0): 128 serial round trips in this prologue otherwise).
  float gy[CT][NS];
  const float* mask_src = relu ? Asv : gA;
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int64_t base = bb[t] * lda + dd[t];
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += 16) {
      float a[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = 2 * (s0 + i) + p;
        gy[t][s0 + i] = ldg_f32(gA + base + static_cast<int64_t>(o < O ? o : O - 1) * D);
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int o = 2 * (s0 + i) + p;
          a[i] = ldg_f32(mask_src + base + static_cast<int64_t>(o < O ? o : O - 1) * D);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 1.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = 2 * (s0 + i) + p;
        gy[t][s0 + i] = (cv[t] && o < O && a[i] > 0.f) ? gy[t][s0 + i] : 0.f;
      }
    }
  }
  // X0 rows matching this lane's accumulator rows
  float x0r[CT][16], gxa[CT][16];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = acc_row(r, p);
      const float v = ldg_f32(X0 + bb[t] * ldx0 + (mm < M ? mm : M - 1) * D + dd[t]);
      x0r[t][r] = (cv[t] && mm < M) ? v : 0.f;
      gxa[t][r] = 0.f;
    }

  float wreg[NW];
  auto fetch_w = [&](int hh) {
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int e = q * NT + tid;  // = o_local * 32 + m
      const int ol = e >> 5, mm = e & 31;
      wreg[q] = (ol < O && mm < M) ? ldg_f32(W + static_cast<int64_t>(ol) * K + hh * M + mm) : 0.f;
    }
  };
  auto park_w = [&](int buf) {
    float* dst = wl + buf * (OB * 32);
#pragma unroll
    for (int q = 0; q < NW; ++q) dst[q * NT + tid] = wreg[q];
  };
  fetch_w(0);
  park_w(0);
  float hv[CT], hnext[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) hv[t] = ldg_f32(H + bb[t] * ldh + dd[t]);
#pragma unroll
  for (int t = 0; t < CT; ++t) asm volatile("" : "+v"(hv[t]));   // arrived before the loop (see k_cin_fwd)
  __syncthreads();

  for (int hh = 0; hh < h; ++hh) {
    const bool more = hh + 1 < h;
    if (more) {
      fetch_w(hh + 1);
#pragma unroll
      for (int t = 0; t < CT; ++t) hnext[t] = ldg_f32(H + bb[t] * ldh + static_cast<int64_t>(hh + 1) * D + dd[t]);
    }
    const float* wc = wl + (hh & 1) * (OB * 32);
    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float a = wc[(2 * s + p) * 32 + jl];
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t] = mfma32(a, gy[t][s], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        part += acc[t][r] * x0r[t][r];
        gxa[t][r] += acc[t][r] * hv[t];
      }
      part += __shfl_xor(part, 32, kWave);
      if (p == 0 && cv[t]) {
        float* dst = gH + bb[t] * ldgh + static_cast<int64_t>(hh) * D + dd[t];
        stg_f32(dst, acc_h ? ldg_f32(dst) + part : part);
      }
    }
    if (more) {
      park_w((hh + 1) & 1);
#pragma unroll
      for (int t = 0; t < CT; ++t) hv[t] = hnext[t];
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = acc_row(r, p);
      if (cv[t] && mm < M) {
        float* dst = gX0 + bb[t] * ldgx + mm * D + dd[t];
        stg_f32(dst, acc_x ? ldg_f32(dst) + gxa[t][r] : gxa[t][r]);
      }
    }
}

// -------------------------------------------------------------------------------------------------------------
// backward, data side, FIRST layer (H IS X0).  With gZs[{a, b}] = sum_o (W[o, a, b] + W[o, b, a]) gY[o]  (a != b;
// W[o, a, a] on the diagonal) the gradient of the shared input is  G[x] = sum_{pairs {a, b}} gZs * (x == a) X0[b] +
// gZs * (x == b) X0[a]  (a diagonal pair counts twice, as the product rule says): only the M (M + 1) / 2 unordered pairs
// are needed, not M * M.  They are packed into ceil-half as many 32-row MFMA tiles as k_cin_bwd_data uses, with every
// accumulator register still tied to a FIXED field, so the tails stay per-lane FMAs on registers:
//   tile j (j < Hh = ceil(M / 2)):  row i >= j  <->  pair (j, i)                        "main":   G[j] += gZs X0[i],  G[i] += gZs X0[j]
//   tile j (1 <= j):                row i <  j  <->  pair {Hh + i, Hh - 1 + j} = {a, b} "mirror": G[b] += gZs X0[a],  G[a] += gZs X0[b]
// (the rows a tile's main pairs leave free hold the triangle of the upper half of the fields; M even needs one more tile,
// j = Hh, for b = M - 1).  M = 26: 14 tiles instead of 26.  Row i of a lane therefore needs X0[i] and X0[Hh + i] (static
// registers) and accumulates G[i] and G[Hh + i]; G[j] and G[b] of a tile come out of the same cross-row sums as gH did.
// Written to gH: the cross-row sums (every field exactly once); to gX0: the per-row accumulators -- the caller adds the
// two, as it does for the general layer.
// -------------------------------------------------------------------------------------------------------------
// -------------------------------------------------------------------------------------------------------------
// backward, data side, rows FLATTENED over (h, m): k_cin_bwd_data gives every h its own 32-row MFMA tile and leaves
// 32 - M rows of it empty (M = 26: 19 % of the matrix work of the layer's biggest kernel).  Here row i of tile T is the
// pair kf = 32 T + i = (h, m) = divmod(kf, M): K / 32 tiles instead of h (52 instead of 64 at h = 64, M = 26).  The price is
// that an accumulator row is no longer one field.  M is a template parameter and the tile loop is unrolled over the
// period after which the row -> (h, m) pattern repeats (TPER = M / gcd(32, M) tiles = 32 / gcd(32, M) values of h; 13
// tiles / 16 h at M = 26), so that every register index below is a compile-time constant: the lane keeps X0[b, :, d] and
// the gX0 accumulators of ALL M fields in registers (26 + 26) and the two candidates of a row -- its fields differ by 4
// between the lane halves -- are chosen with a select.  Per accumulator row: ~7 VALU operations instead of 2.
// -------------------------------------------------------------------------------------------------------------
constexpr int cin_gcd(int a, int b) { return b == 0 ? a : cin_gcd(b, a % b); }

template <int OT, int M>
__global__ __launch_bounds__(kT * 2, 1) void k_cin_bwd_data_flat(const float* __restrict__ gA, const float* __restrict__ Asv,
                                                        int64_t lda, int relu, const float* __restrict__ X0,
                                                        int64_t ldx0, const float* __restrict__ H, int64_t ldh,
                                                        int h, int D, int B, const float* __restrict__ W,
                                                        int O, float* __restrict__ gH, int64_t ldgh, int acc_h,
                                                        float* __restrict__ gX0, int64_t ldgx, int acc_x) {
  constexpr int NT = kT * 2;                               // eight waves, one column tile each
  constexpr int OB = OT * 32, NS = OT * 16, NW = OT * 32 * 32 / NT;
  constexpr int G = cin_gcd(32, M), TPER = M / G, HPER = 32 / G;
  extern __shared__ __align__(16) float smem[];
  float* wl = smem;  // [2][OB][32]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, p = lane >> 5, jl = lane & 31;
  const int64_t c_base = static_cast<int64_t>(blockIdx.x) * kCols;
  const int64_t ncol = static_cast<int64_t>(B) * D;
  const int K = h * M;
  const int ntiles = (K + 31) / 32;

  const int64_t c = c_base + wv * 32 + jl;
  const bool cv = c < ncol;
  const int64_t bb = cv ? c / D : 0;
  const int dd = cv ? static_cast<int>(c - bb * D) : 0;
  float gy[NS];
  {
    const float* mask_src = relu ? Asv : gA;
    const int64_t base = bb * lda + dd;
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += 16) {
      float a[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = 2 * (s0 + i) + p;
        gy[s0 + i] = ldg_f32(gA + base + static_cast<int64_t>(o < O ? o : O - 1) * D);
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int o = 2 * (s0 + i) + p;
          a[i] = ldg_f32(mask_src + base + static_cast<int64_t>(o < O ? o : O - 1) * D);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 1.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = 2 * (s0 + i) + p;
        gy[s0 + i] = (cv && o < O && a[i] > 0.f) ? gy[s0 + i] : 0.f;
      }
    }
  }
  float x0r[M], gxa[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float v = ldg_f32(X0 + bb * ldx0 + m * D + dd);
    x0r[m] = cv ? v : 0.f;
    gxa[m] = 0.f;
  }
  const float pm0 = p ? 0.f : 1.f, pm1 = p ? 1.f : 0.f;   // which of a row's two candidate fields this lane half holds

  float wreg[NW];
  auto fetch_w = [&](int T) {       // W[o][32 T + i]: 128-byte runs
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int e = q * NT + tid;  // = o_local * 32 + row
      const int ol = e >> 5, kf = 32 * T + (e & 31);
      const float v = ldg_f32(W + static_cast<int64_t>(ol < O ? ol : O - 1) * K + (kf < K ? kf : K - 1));
      wreg[q] = (ol < O && kf < K) ? v : 0.f;
    }
  };
  auto park_w = [&](int buf) {
    float* dst = wl + buf * (OB * 32);
#pragma unroll
    for (int q = 0; q < NW; ++q) dst[q * NT + tid] = wreg[q];
  };
  // H[b, hA .. hA + 2, d] of a tile (hA = the field of its first row; 32 rows touch at most three fields for M >= 16)
  static_assert(M >= 16 && M <= 32, "a tile of 32 rows must not span more than three fields");
  auto tile_h = [&](int T, float& ha, float& hb, float& hc) {
    const int hA = (32 * T) / M;
    const int h0c = hA < h ? hA : h - 1, h1c = hA + 1 < h ? hA + 1 : h - 1, h2c = hA + 2 < h ? hA + 2 : h - 1;
    ha = ldg_f32(H + bb * ldh + static_cast<int64_t>(h0c) * D + dd);
    hb = ldg_f32(H + bb * ldh + static_cast<int64_t>(h1c) * D + dd);
    hc = ldg_f32(H + bb * ldh + static_cast<int64_t>(h2c) * D + dd);
  };
  fetch_w(0);
  park_w(0);
  float hvA, hvB, hvC, hnA = 0.f, hnB = 0.f, hnC = 0.f;
  tile_h(0, hvA, hvB, hvC);
  asm volatile("" : "+v"(hvA));   // arrived before the loop (see k_cin_fwd)
  asm volatile("" : "+v"(hvB));
  asm volatile("" : "+v"(hvC));
  __syncthreads();

  float part[HPER + 1];
#pragma unroll
  for (int i = 0; i <= HPER; ++i) part[i] = 0.f;

  for (int Tb = 0; Tb < ntiles; Tb += TPER) {
    const int hbase = (Tb / TPER) * HPER;      // field of the period's first row
#pragma unroll
    for (int tm = 0; tm < TPER; ++tm) {
      const int T = Tb + tm;
      if (T < ntiles) {                        // (uniform)
        const bool more = T + 1 < ntiles;
        if (more) {
          fetch_w(T + 1);
          tile_h(T + 1, hnA, hnB, hnC);
        }
        const float* wc = wl + (T & 1) * (OB * 32);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) acc = mfma32(wc[(2 * s + p) * 32 + jl], gy[s], acc);
        // tails: everything below indexes registers with constants (tm, r are unrolled)
        const int hA = (32 * tm) / M;          // relative to hbase
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i0 = (r & 3) + 8 * (r >> 2);            // the row of lane half 0; half 1 holds row i0 + 4
          const int k0 = 32 * tm + i0, k1 = k0 + 4;
          const int h0 = k0 / M, m0 = k0 % M, h1 = k1 / M, m1 = k1 % M;
          const float z = acc[r];
          const float xs = p ? x0r[m1] : x0r[m0];
          const float hs0 = h0 == hA ? hvA : (h0 == hA + 1 ? hvB : hvC), hs1 = h1 == hA ? hvA : (h1 == hA + 1 ? hvB : hvC);
          const float hs = (h0 == h1) ? hs0 : (p ? hs1 : hs0);
          const float zh = z * hs, zx = z * xs;
          gxa[m0] = fmaf(zh, pm0, gxa[m0]);
          gxa[m1] = fmaf(zh, pm1, gxa[m1]);
          if (h0 == h1) {
            part[h0] += zx;
          } else {
            part[h0] = fmaf(zx, pm0, part[h0]);
            part[h1] = fmaf(zx, pm1, part[h1]);
          }
        }
        // fields whose last row lies in this tile are complete
#pragma unroll
        for (int hr = 0; hr < HPER; ++hr) {
          if (((hr + 1) * M - 1) / 32 == tm) {
            float v = part[hr];
            v += __shfl_xor(v, 32, kWave);
            part[hr] = 0.f;
            const int hq = hbase + hr;
            if (p == 0 && cv && hq < h) {
              float* dst = gH + bb * ldgh + static_cast<int64_t>(hq) * D + dd;
              stg_f32(dst, acc_h ? ldg_f32(dst) + v : v);
            }
          }
        }
        if (more) {
          park_w((T + 1) & 1);
          hvA = hnA;
          hvB = hnB;
          hvC = hnC;
        }
        __syncthreads();
      }
    }
  }
  // both lane halves hold partial sums of every field of their column
#pragma unroll
  for (int m = 0; m < M; ++m) {
    float v = gxa[m];
    v += __shfl_xor(v, 32, kWave);
    if (cv && (m & 1) == p) {
      float* dst = gX0 + bb * ldgx + m * D + dd;
      stg_f32(dst, acc_x ? ldg_f32(dst) + v : v);
    }
  }
}

// the folded weight slices of k_cin_bwd_data_sym: Ws[tile j][o < OB][row i < 32] (zero where a row holds no pair)
__global__ __launch_bounds__(kT) void k_cin_prep_wsym(const float* __restrict__ W, int O, int M, int OB, int ntiles,
                                                      float* __restrict__ Ws) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  if (idx >= static_cast<int64_t>(ntiles) * OB * 32) return;
  const int i = static_cast<int>(idx & 31), ol = static_cast<int>((idx >> 5) % OB), j = static_cast<int>(idx / (32 * OB));
  const int Hh = (M + 1) / 2;
  const bool main = i >= j;
  const int a = main ? j : Hh + i;
  const int b = main ? i : Hh - 1 + j;
  const bool valid = ol < O && (main ? (j < Hh && i < M) : (b < M));     // (mirror rows: a <= b by construction)
  float w = 0.f;
  if (valid) {
    const float* row = W + static_cast<int64_t>(ol) * M * M;
    w = a == b ? row[a * M + a] : row[a * M + b] + row[b * M + a];
  }
  Ws[idx] = w;
}

template <int OT, int CT>
__global__ __launch_bounds__(kT * 2 / CT, 1) void k_cin_bwd_data_sym(const float* __restrict__ gA, const float* __restrict__ Asv,
                                                        int64_t lda, int relu, const float* __restrict__ X0,
                                                        int64_t ldx0, int M, int D, int B, const float* __restrict__ Ws,
                                                        int O, float* __restrict__ gH, int64_t ldgh, int acc_h,
                                                        float* __restrict__ gX0, int64_t ldgx, int acc_x) {
  constexpr int NT = kT * 2 / CT;
  constexpr int OB = OT * 32, NS = OT * 16, NW = OT * 32 * 32 / NT;
  extern __shared__ __align__(16) float smem[];
  float* wl = smem;  // [2][OB][32]
  float* xu = smem + 2 * OB * 32;   // [16][kCols]: X0[Hh + i] of the workgroup's columns; at the end the upper-half accumulators
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, p = lane >> 5, jl = lane & 31;
  const int64_t c_base = static_cast<int64_t>(blockIdx.x) * kCols;
  const int64_t ncol = static_cast<int64_t>(B) * D;
  const int Hh = (M + 1) / 2;
  const int ntiles = (M & 1) ? Hh : Hh + 1;

  int64_t bb[CT];
  int dd[CT];
  bool cv[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int64_t c = c_base + wv * (32 * CT) + t * 32 + jl;
    cv[t] = c < ncol;
    bb[t] = cv[t] ? c / D : 0;
    dd[t] = cv[t] ? static_cast<int>(c - bb[t] * D) : 0;
  }
  float gy[CT][NS];
  const float* mask_src = relu ? Asv : gA;
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int64_t base = bb[t] * lda + dd[t];
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += 16) {
      float a[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = 2 * (s0 + i) + p;
        gy[t][s0 + i] = ldg_f32(gA + base + static_cast<int64_t>(o < O ? o : O - 1) * D);
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int o = 2 * (s0 + i) + p;
          a[i] = ldg_f32(mask_src + base + static_cast<int64_t>(o < O ? o : O - 1) * D);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 1.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = 2 * (s0 + i) + p;
        gy[t][s0 + i] = (cv[t] && o < O && a[i] > 0.f) ? gy[t][s0 + i] : 0.f;
      }
    }
  }
  // the two fields of a lane's accumulator row i: i and Hh + i.  Mirror rows are rows i < j <= Hh <= 16: only the
  // registers r < 8 (rows 0..15) ever hold one
  constexpr int RU = 8;
  float x0r[CT][16], gxa[CT][16], gxu[CT][RU];
  if (tid < kCols) {      // the upper half of the fields, by column, in LDS (two register sets of X0 would spill)
    const int64_t c = c_base + tid;
    const bool v = c < ncol;
    const int64_t b = v ? c / D : 0;
    const int d = v ? static_cast<int>(c - b * D) : 0;
    float t16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t16[i] = ldg_f32(X0 + b * ldx0 + (Hh + i < M ? Hh + i : M - 1) * D + d);
#pragma unroll
    for (int i = 0; i < 16; ++i) xu[i * kCols + tid] = (v && Hh + i < M) ? t16[i] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = acc_row(r, p);
      const float v = ldg_f32(X0 + bb[t] * ldx0 + (mm < M ? mm : M - 1) * D + dd[t]);
      x0r[t][r] = (cv[t] && mm < M) ? v : 0.f;
      gxa[t][r] = 0.f;
      if (r < RU) gxu[t][r] = 0.f;
    }

  float wreg[NW];
  auto fetch_w = [&](int j) {      // the folded slice of tile j, laid out [o][row] by k_cin_prep_wsym
    const float* src = Ws + static_cast<int64_t>(j) * (OB * 32);
#pragma unroll
    for (int q = 0; q < NW; ++q) wreg[q] = ldg_f32(src + q * NT + tid);
  };
  auto park_w = [&](int buf) {
    float* dst = wl + buf * (OB * 32);
#pragma unroll
    for (int q = 0; q < NW; ++q) dst[q * NT + tid] = wreg[q];
  };
  // X0[j] of a tile, for this lane's columns (X0[b], b >= Hh, is read from xu)
  auto tile_fields = [&](int j, float* hj) {
    const int jc = j < M ? j : M - 1;
#pragma unroll
    for (int t = 0; t < CT; ++t) hj[t] = ldg_f32(X0 + bb[t] * ldx0 + static_cast<int64_t>(jc) * D + dd[t]);
  };
  fetch_w(0);
  park_w(0);
  float hj[CT], hjn[CT];
  tile_fields(0, hj);
#pragma unroll
  for (int t = 0; t < CT; ++t) asm volatile("" : "+v"(hj[t]));   // arrived before the loop (see k_cin_fwd)
  __syncthreads();

  for (int j = 0; j < ntiles; ++j) {
    const bool more = j + 1 < ntiles;
    if (more) {
      fetch_w(j + 1);
      tile_fields(j + 1, hjn);
    }
    const float* wc = wl + (j & 1) * (OB * 32);
    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float a = wc[(2 * s + p) * 32 + jl];
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[t] = mfma32(a, gy[t][s], acc[t]);
    }
    const int bfield = Hh - 1 + j;
    const int brow = (j >= 1 && bfield < M) ? j - 1 : 0;      // row of X0[b] in xu (no mirror pairs: their weights are zero)
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int col = wv * (32 * CT) + t * 32 + jl;
      const float hb = xu[brow * kCols + col];
      float pm = 0.f, pu = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r < RU) {
          const bool main = acc_row(r, p) >= j;
          const float zm = main ? acc[t][r] : 0.f, zu = main ? 0.f : acc[t][r];
          pm += zm * x0r[t][r];
          gxa[t][r] += zm * hj[t];
          pu += zu * xu[acc_row(r, p) * kCols + col];
          gxu[t][r] += zu * hb;
        } else {            // rows 16..31: always a main pair (zero weights when there is none)
          pm += acc[t][r] * x0r[t][r];
          gxa[t][r] += acc[t][r] * hj[t];
        }
      }
      pm += __shfl_xor(pm, 32, kWave);
      pu += __shfl_xor(pu, 32, kWave);
      if (p == 0 && cv[t]) {
        if (j < Hh) {
          float* dst = gH + bb[t] * ldgh + static_cast<int64_t>(j) * D + dd[t];
          stg_f32(dst, acc_h ? ldg_f32(dst) + pm : pm);
        }
        if (j >= 1 && bfield < M) {
          float* dst = gH + bb[t] * ldgh + static_cast<int64_t>(bfield) * D + dd[t];
          stg_f32(dst, acc_h ? ldg_f32(dst) + pu : pu);
        }
      }
    }
    if (more) {
      park_w((j + 1) & 1);
#pragma unroll
      for (int t = 0; t < CT; ++t) hj[t] = hjn[t];
    }
    __syncthreads();
  }
  // the upper-half accumulators move to the lanes that own those fields' rows (same column: through LDS)
  float* up = xu;     // [16][kCols] (the loop's last barrier has retired every read of X0's upper half)
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < RU; ++r) up[acc_row(r, p) * kCols + wv * (32 * CT) + t * 32 + jl] = gxu[t][r];
  __syncthreads();
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = acc_row(r, p);
      if (cv[t] && mm < M) {
        float v = gxa[t][r];
        if (mm >= Hh && mm - Hh < 16) v += up[(mm - Hh) * kCols + wv * (32 * CT) + t * 32 + jl];
        float* dst = gX0 + bb[t] * ldgx + mm * D + dd[t];
        stg_f32(dst, acc_x ? ldg_f32(dst) + v : v);
      }
    }
}

// -------------------------------------------------------------------------------------------------------------
// backward, weight side:  gW[o, k = (h, m)] = sum_c gY[o, c] H[b, h, d] X0[b, m, d]        (c = (b, d))
// A GEMM with 128 rows (o), K = h*M columns and a reduction over the B*D = 65536 batch columns: 27.9 GFLOP for
// layer 1 of the Criteo shape.  MFMA roles (v_mfma_f32_32x32x2_f32): rows i = o, columns j = k, reduction = c.
//   * k runs over the FLATTENED (h, m) index, so M = 26 is not padded to 32 (52 column tiles instead of 64);
//     the B operand of an MFMA is still one v_mul per lane:  Z[c, k] = H[c, k / M] * X0[c, k % M].
//   * a wave owns kWgNT = 2 column tiles x OT row tiles (8 independent accumulators at O = 128: the matrix pipe
//     issues back to back); the 4 waves of a workgroup own 8 consecutive column tiles and share the staged gY block,
//     so gY (the big operand, B*O*D floats + its relu mask) is re-read ceil(K / 256) times instead of h / 4 times.
//   * workgroup (kx, q) walks the 64-column blocks q, q + Q, ...; Q is chosen so that two workgroups sit on every CU
//     (one stages while the other multiplies).  Staging: a thread owns ONE column (b, d) of the block and walks o /
//     m / h, a half-wave therefore writes 32 consecutive columns of an odd-pitch LDS row: conflict-free
//     ds_write_b32, and a wave's load covers 64-byte runs of gA.
//   * no atomics: every workgroup stores its partial [O, K] tile set to the workspace and k_cin_wgrad_reduce adds
//     the Q partials in a fixed order (deterministic; 2 x 60 MB of traffic at the Criteo shape, ~10 % of the GEMM).
// -------------------------------------------------------------------------------------------------------------
constexpr int kWgCB = 64;    // batch columns per staged block
constexpr int kWgP = 129;    // floats per gys row  [c][o]   (odd pitch)
constexpr int kWgPX = 33;    // floats per x0s row  [c][m]
constexpr int kWgNT = 2;     // column tiles per wave
constexpr int kWgKW = 4 * kWgNT * 32;   // k columns per workgroup

template <int OT>
__global__ __launch_bounds__(kT, 2) void k_cin_wgrad(const float* __restrict__ gA, const float* __restrict__ Asv,
                                                     int64_t lda, int relu, const float* __restrict__ X0,
                                                     int64_t ldx0, const float* __restrict__ H, int64_t ldh, int h,
                                                     int M, int D, int B, int O, int hspan, int PH,
                                                     float* __restrict__ part, float* __restrict__ bpart, int sym) {
  constexpr int OB = OT * 32;
  extern __shared__ __align__(16) float smem[];
  float* gys = smem;                  // [CB][kWgP]
  float* x0s = gys + kWgCB * kWgP;    // [CB][kWgPX]
  float* hs = x0s + kWgCB * kWgPX;    // [CB][PH]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, p = lane >> 5, jl = lane & 31;
  const int64_t ncol = static_cast<int64_t>(B) * D;
  const int64_t nblk = (ncol + kWgCB - 1) / kWgCB;
  // sym (H IS X0: gW[o, h, m] = gW[o, m, h]): the columns are the M (M + 1) / 2 pairs h <= m, row-major; both factors come
  // from the staged X0 block, k_cin_wgrad_reduce_sym writes every sum to its two places
  const int K = sym ? M * (M + 1) / 2 : h * M;
  const int k0 = blockIdx.x * kWgKW;
  const int h0 = sym ? 0 : k0 / M;

  // this lane's column of each of the wave's tiles
  int hk[kWgNT], mk[kWgNT], kk[kWgNT];
#pragma unroll
  for (int nt = 0; nt < kWgNT; ++nt) {
    const int k = k0 + (wv * kWgNT + nt) * 32 + jl;
    kk[nt] = k;
    const int kc = k < K ? k : k0;          // out-of-range columns read valid LDS and are never stored
    if (sym) {
      int hq = 0, rest = kc;                // row hq of the triangle holds M - hq pairs
      while (rest >= M - hq) { rest -= M - hq; ++hq; }
      hk[nt] = hq;
      mk[nt] = hq + rest;
    } else {
      const int hq = kc / M;
      hk[nt] = hq - h0;
      mk[nt] = kc - hq * M;
    }
  }
  const float* hl_base = sym ? x0s : hs;    // where the H factor of a column is staged
  const int hl_pitch = sym ? kWgPX : PH;
  const bool wave_active = (k0 + wv * kWgNT * 32) < K;

  f32x16 acc[OT][kWgNT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int nt = 0; nt < kWgNT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ot][nt][r] = 0.f;
  float bsum = 0.f;  // gbias partial of row o = tid (workgroup column 0 only)

  const int cl = tid & (kWgCB - 1), q4 = tid >> 6;   // staging role: column cl, rows q4, q4 + 4, ...
  for (int64_t blk = blockIdx.y; blk < nblk; blk += gridDim.y) {
    const int64_t c = blk * kWgCB + cl;
    const bool cvalid = c < ncol;
    const int64_t b = cvalid ? c / D : 0;
    const int d = cvalid ? static_cast<int>(c - b * D) : 0;
    // Every load below is UNCONDITIONAL on a clamped address and masked afterwards with a select: predicated loads
    // compile to a branch + s_waitcnt vmcnt(0) each, which serialises the staging into one memory round trip per row.
    // X0[b, :, d] and H[b, h0 .. h0 + hspan, d]: issued first, parked in LDS after the gY loop (one round trip shared)
    const float* xsrc = X0 + b * ldx0 + d;
    const float* hsrc = H + b * ldh + d;
    float x[8], hv[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int mm = q4 + 4 * i;
      x[i] = ldg_f32(xsrc + (mm < M ? mm : M - 1) * D);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hq = h0 + q4 + 4 * i;
      hv[i] = sym ? 0.f : ldg_f32(hsrc + static_cast<int64_t>(hq < h ? hq : h - 1) * D);
    }
    if (!relu) {  // gY^T as it is (the caller masked it: dctr_cin_pool_bwd with A): 16 rows per round trip, not 8 + 8
      constexpr int CH = 16;
      const float* ga = gA + b * lda + d;
#pragma unroll 1
      for (int i0 = 0; i0 < OB / 4; i0 += CH) {
        float g[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int o = q4 + 4 * (i0 + i);
          g[i] = ldg_f32(ga + static_cast<int64_t>(o < O ? o : O - 1) * D);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int o = q4 + 4 * (i0 + i);
          if (i0 + i < OB / 4) gys[cl * kWgP + o] = (cvalid && o < O) ? g[i] : 0.f;
        }
      }
    } else {  // gY^T (masked by the saved activation when relu)
      constexpr int CH = 8;   // 16 loads in flight per thread; 16 rows would spill at OT = 4
      const float* ga = gA + b * lda + d;
      const float* as = relu ? Asv + b * lda + d : ga;
#pragma unroll 1
      for (int i0 = 0; i0 < OB / 4; i0 += CH) {
        float g[CH], a[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int o = q4 + 4 * (i0 + i);
          const int oc = o < O ? o : O - 1;
          g[i] = ldg_f32(ga + static_cast<int64_t>(oc) * D);
        }
        if (relu) {
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const int o = q4 + 4 * (i0 + i);
            const int oc = o < O ? o : O - 1;
            a[i] = ldg_f32(as + static_cast<int64_t>(oc) * D);
          }
        } else {
#pragma unroll
          for (int i = 0; i < CH; ++i) a[i] = 1.f;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int o = q4 + 4 * (i0 + i);
          const bool keep = cvalid && o < O && a[i] > 0.f;
          gys[cl * kWgP + o] = keep ? g[i] : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x0s[cl * kWgPX + q4 + 4 * i] = cvalid ? x[i] : 0.f;   // rows m >= M are never read
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hl = q4 + 4 * i;
      if (!sym && hl < hspan) hs[cl * PH + hl] = (cvalid && h0 + hl < h) ? hv[i] : 0.f;
    }
    for (int hl = q4 + 16; !sym && hl < hspan; hl += 4) {     // only when M is small (hspan > 16)
      const int hq = h0 + hl < h ? h0 + hl : h - 1;
      const float v = ldg_f32(hsrc + static_cast<int64_t>(hq) * D);
      hs[cl * PH + hl] = (cvalid && h0 + hl < h) ? v : 0.f;
    }
    __syncthreads();
    if (bpart && blockIdx.x == 0 && tid < OB) {
#pragma unroll 8
      for (int cc = 0; cc < kWgCB; ++cc) bsum += gys[cc * kWgP + tid];
    }
    if (wave_active) {
#pragma unroll 4
      for (int ks = 0; ks < kWgCB / 2; ++ks) {
        const int cc = 2 * ks + p;
        float a[OT], z[kWgNT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) a[ot] = gys[cc * kWgP + ot * 32 + jl];
#pragma unroll
        for (int nt = 0; nt < kWgNT; ++nt) z[nt] = hl_base[cc * hl_pitch + hk[nt]] * x0s[cc * kWgPX + mk[nt]];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int nt = 0; nt < kWgNT; ++nt) acc[ot][nt] = mfma32(a[ot], z[nt], acc[ot][nt]);
      }
    }
    __syncthreads();
  }
  // partial tiles of this workgroup: part[q][o][k]
  float* dst = part + static_cast<int64_t>(blockIdx.y) * O * K;
  if (wave_active) {
#pragma unroll
    for (int nt = 0; nt < kWgNT; ++nt) {
      if (kk[nt] < K) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = ot * 32 + acc_row(r, p);
            if (o < O) stg_f32(dst + static_cast<int64_t>(o) * K + kk[nt], acc[ot][nt][r]);
          }
      }
    }
  }
  if (bpart && blockIdx.x == 0 && tid < O) stg_f32(bpart + static_cast<int64_t>(blockIdx.y) * O + tid, bsum);
}

// out[i] = sum_q part[q][i] in a FIXED order: four neighbouring lanes share an output, lane j of them adds the partials
// q = j, j + 4, ... ascending, the four sums are combined as (s0 + s1) + (s2 + s3) -- the order of the one-thread-per-output
// version it replaces, at a quarter of its dependent-load chain (Q = 170-256 partials: 12-21 us -> a few).
__device__ __forceinline__ float reduce_q4(const float* __restrict__ part, int Q, int64_t n, int64_t i, int j) {
  float s = 0.f;
  int q = j;
  for (; q + 12 < Q; q += 16) {       // four loads in flight per lane
    const float a = ldg_f32(part + (q + 0) * n + i), b = ldg_f32(part + (q + 4) * n + i);
    const float c = ldg_f32(part + (q + 8) * n + i), d = ldg_f32(part + (q + 12) * n + i);
    s += a; s += b; s += c; s += d;
  }
  for (; q < Q; q += 4) s += ldg_f32(part + q * n + i);
  const float s1 = __shfl_xor(s, 1);
  const float lo = (j & 1) ? s1 + s : s + s1;          // s_even + s_odd on both lanes of a pair
  const float hi = __shfl_xor(lo, 2);
  return (j & 2) ? hi + lo : lo + hi;                   // (s0 + s1) + (s2 + s3) on all four
}

__device__ __forceinline__ void reduce_wave_block(const float* __restrict__ part, int Q, int64_t n,
                                                  float* __restrict__ out, int blk);

__global__ __launch_bounds__(kT) void k_cin_wgrad_reduce(const float* __restrict__ part, int Q, int64_t n,
                                                         float* __restrict__ out, int nblk_w,
                                                         const float* __restrict__ bpart, int nb,
                                                         float* __restrict__ gbias) {
  if (static_cast<int>(blockIdx.x) >= nblk_w) {          // (uniform: the bias gradient's workgroups)
    reduce_wave_block(bpart, Q, nb, gbias, static_cast<int>(blockIdx.x) - nblk_w);
    return;
  }
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  const int64_t i = t >> 2;
  const int j = static_cast<int>(t & 3);
  const float v = reduce_q4(part, Q, n, i < n ? i : n - 1, j);     // (whole quads stay converged for the shuffles)
  if (i < n && j == 0) stg_f32(out + i, v);
}

// few outputs, many partials (the bias gradient: n = 128, Q up to 256): one wave per output, lane l adds q = l, l + 64, ...
// and the 64 sums meet in a butterfly (fixed order).  `blk`: the workgroup's index among the bias workgroups -- they ride
// behind the weight workgroups of k_cin_wgrad_reduce / _sym (a launch of their own was 5.7 us of latency per layer).
__device__ __forceinline__ void reduce_wave_block(const float* __restrict__ part, int Q, int64_t n,
                                                  float* __restrict__ out, int blk) {
  const int64_t i = static_cast<int64_t>(blk) * (kT / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.f;
  for (int q = lane; q < Q; q += 64) s += ldg_f32(part + q * n + i);
  s = wave_sum(s);
  if (lane == 0) stg_f32(out + i, s);
}

// the same sum for the symmetric layer: part[q][o][pair (h <= m)] -> out[o][h * M + m] and out[o][m * M + h]
__global__ __launch_bounds__(kT) void k_cin_wgrad_reduce_sym(const float* __restrict__ part, int Q, int O, int M,
                                                             float* __restrict__ out, int nblk_w,
                                                             const float* __restrict__ bpart, int nb,
                                                             float* __restrict__ gbias) {
  if (static_cast<int>(blockIdx.x) >= nblk_w) {          // (uniform: the bias gradient's workgroups)
    reduce_wave_block(bpart, Q, nb, gbias, static_cast<int>(blockIdx.x) - nblk_w);
    return;
  }
  const int KP = M * (M + 1) / 2;
  const int64_t n = static_cast<int64_t>(O) * KP;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  const int64_t i = t >> 2;
  const int j = static_cast<int>(t & 3);
  const float v = reduce_q4(part, Q, n, i < n ? i : n - 1, j);
  if (i >= n || j != 0) return;
  const int o = static_cast<int>(i / KP);
  int hq = 0, rest = static_cast<int>(i - static_cast<int64_t>(o) * KP);
  while (rest >= M - hq) { rest -= M - hq; ++hq; }
  const int mm = hq + rest;
  float* row = out + static_cast<int64_t>(o) * M * M;
  stg_f32(row + hq * M + mm, v);
  if (mm != hq) stg_f32(row + mm * M + hq, v);
}

// launch geometry of the weight-side kernels (shared by the workspace query and the launcher)
struct WgradGeom {
  int gx, q, hspan, ph;
  size_t lds;
};
inline WgradGeom wgrad_geom(int B, int h, int M, int D, bool sym = false) {
  WgradGeom g;
  const int K = sym ? M * (M + 1) / 2 : h * M;
  g.gx = (K + kWgKW - 1) / kWgKW;
  const int64_t nblk = (static_cast<int64_t>(B) * D + kWgCB - 1) / kWgCB;
  int64_t q = 512 / g.gx;
  if (q < 1) q = 1;
  if (q > nblk) q = nblk;
  if (q < 1) q = 1;
  g.q = static_cast<int>(q);
  int hs = (kWgKW - 1) / M + 2;
  if (hs > h) hs = h;
  g.hspan = hs;
  g.ph = hs | 1;
  g.lds = (static_cast<size_t>(kWgCB) * (kWgP + kWgPX + g.ph)) * sizeof(float);
  return g;
}

}  // namespace

// ---- the "direct connect" rows of a layer, as xDeepFM consumes them (interaction.py:226-246) ------------------------
// pooled[b, o] = sum_d A[b, pool_from + o, d]  (d ascending; rows of `pooled` ld_p apart: a layer's block of the CIN's
// [B, featuremap_num] output), and its adjoint together with the hidden rows' gradient:
//   gA[b, o, :] = (o < n_hidden ? g_hidden[b, o, :] : 0) + (o >= pool_from ? gp(b, o - pool_from) : 0)
//   gp(b, j)    = g_pooled[b * ld_gp + j]               (w_head == NULL)
//               = g_pooled[b * ld_gp] * w_head[j]       (the 1-unit projection xDeepFM puts on the CIN, xdeepfm.py:72:
//                                                        g_pooled is then the logit's gradient, one float per sample)
// One launch each (torch: a strided reduction at 0.8 TB/s forward; slice-copy + expand-copy (+ an outer-product GEMM)
// backward).
namespace {
__global__ __launch_bounds__(256) void k_cin_pool_fwd(const float* __restrict__ A, int64_t n_out, int O, int D,
                                                      int pool_from, float* __restrict__ pooled, int64_t ld_p) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n_out) return;
  const int nd = O - pool_from;
  const int64_t b = i / nd;
  const int o = static_cast<int>(i - b * nd);
  const float* src = A + (b * O + pool_from + o) * D;
  float s = 0.f;
  if ((D & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
    for (int d = 0; d < D; d += 4) {
      const f32x4 v = *(const DCTR_GLOBAL f32x4*)(src + d);
      s += v[0]; s += v[1]; s += v[2]; s += v[3];
    }
  } else {
    for (int d = 0; d < D; ++d) s += ldg_f32(src + d);
  }
  stg_f32(pooled + b * ld_p + o, s);
}

__device__ __forceinline__ float pool_grad(const float* __restrict__ g_pooled, int64_t ld_gp,
                                           const float* __restrict__ w_head, int64_t b, int j) {
  return w_head ? ldg_f32(g_pooled + b * ld_gp) * ldg_f32(w_head + j) : ldg_f32(g_pooled + b * ld_gp + j);
}

__global__ __launch_bounds__(256) void k_cin_pool_bwd(const float* __restrict__ g_hidden,
                                                      const float* __restrict__ g_pooled, int64_t ld_gp,
                                                      const float* __restrict__ w_head,
                                                      const float* __restrict__ Asv, int64_t n, int O, int D,
                                                      int n_hidden, int pool_from, float* __restrict__ gA) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;   // one element of gA [B, O, D]
  if (i >= n) return;
  const int64_t row = i / D;                 // b * O + o
  const int d = static_cast<int>(i - row * D);
  const int64_t b = row / O;
  const int o = static_cast<int>(row - b * O);
  float v = 0.f;
  if (o < n_hidden && g_hidden) v = ldg_f32(g_hidden + (b * n_hidden + o) * D + d);
  if (o >= pool_from && g_pooled) v += pool_grad(g_pooled, ld_gp, w_head, b, o - pool_from);
  if (Asv && !(ldg_f32(Asv + i) > 0.f)) v = 0.f;
  stg_f32(gA + i, v);
}
// the same, four elements of a row per thread (D % 4 == 0, 16-byte aligned operands)
__global__ __launch_bounds__(256) void k_cin_pool_bwd4(const float* __restrict__ g_hidden,
                                                       const float* __restrict__ g_pooled, int64_t ld_gp,
                                                       const float* __restrict__ w_head,
                                                       const float* __restrict__ Asv, int64_t n4, int O, int D4,
                                                       int n_hidden, int pool_from, float* __restrict__ gA) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;   // one float4 of gA [B, O, D]
  if (i >= n4) return;
  const int64_t row = i / D4;
  const int d4 = static_cast<int>(i - row * D4);
  const int64_t b = row / O;
  const int o = static_cast<int>(row - b * O);
  // (unconditional loads on valid addresses, selected afterwards)
  const f32x4 gh = g_hidden ? *(const DCTR_GLOBAL f32x4*)(g_hidden + ((b * n_hidden + (o < n_hidden ? o : 0)) * D4 + d4) * 4)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
  const float gp = g_pooled ? pool_grad(g_pooled, ld_gp, w_head, b, o >= pool_from ? o - pool_from : 0) : 0.f;
  f32x4 v = o < n_hidden ? gh : f32x4{0.f, 0.f, 0.f, 0.f};
  if (o >= pool_from) { v.x += gp; v.y += gp; v.z += gp; v.w += gp; }
  if (Asv) {
    const f32x4 a = *(const DCTR_GLOBAL f32x4*)(Asv + i * 4);
    v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
  }
  *(DCTR_GLOBAL f32x4*)(gA + i * 4) = v;
}
}  // namespace

extern "C" int dctr_cin_pool_fwd(const float* A, int32_t B, int32_t O, int32_t D, int32_t pool_from, float* pooled,
                                 int64_t ld_pooled, dctr_stream_t stream) {
  if (!A || !pooled || B < 0 || O <= 0 || D <= 0 || pool_from < 0 || pool_from >= O || ld_pooled < O - pool_from)
    return DCTR_EINVAL;
  const int64_t n = static_cast<int64_t>(B) * (O - pool_from);
  if (n == 0) return DCTR_OK;
  k_cin_pool_fwd<<<dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      A, n, O, D, pool_from, pooled, ld_pooled);
  return launch_status();
}

extern "C" int dctr_cin_pool_bwd(const float* g_hidden, const float* g_pooled, int64_t ld_gp, const float* w_head,
                                 const float* A_relu, int32_t B, int32_t O, int32_t D, int32_t n_hidden,
                                 int32_t pool_from, float* gA, dctr_stream_t stream) {
  if (!gA || B < 0 || O <= 0 || D <= 0 || n_hidden < 0 || n_hidden > O || pool_from < 0 || pool_from > O)
    return DCTR_EINVAL;
  if (g_pooled && pool_from < O && ld_gp < (w_head ? 1 : O - pool_from)) return DCTR_EINVAL;
  if (w_head && !g_pooled) return DCTR_EINVAL;
  const int64_t n = static_cast<int64_t>(B) * O * D;
  if (n == 0) return DCTR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  const float* gp = pool_from < O ? g_pooled : nullptr;
  // (g_hidden needs its row 0 readable when n_hidden == 0: only passed then as NULL)
  if ((D & 3) == 0 && al16(gA) && al16(g_hidden) && al16(A_relu) && (n_hidden > 0 || !g_hidden)) {
    const int64_t n4 = n / 4;
    k_cin_pool_bwd4<<<dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), 0, s>>>(
        n_hidden > 0 ? g_hidden : nullptr, gp, ld_gp, w_head, A_relu, n4, O, D / 4, n_hidden, pool_from, gA);
  } else {
    k_cin_pool_bwd<<<dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s>>>(
        n_hidden > 0 ? g_hidden : nullptr, gp, ld_gp, w_head, A_relu, n, O, D, n_hidden, pool_from, gA);
  }
  return launch_status();
}

extern "C" size_t dctr_cin_workspace_floats(int32_t h, int32_t M, int32_t O) {
  const size_t M_pad = (M + 1) / 2 * 2, O_pad = (O + 31) / 32 * 32;
  return static_cast<size_t>(h) * M_pad * O_pad;
}

extern "C" size_t dctr_cin_bwd_workspace_floats(int32_t B, int32_t h, int32_t M, int32_t D, int32_t O) {
  if (B <= 0 || h <= 0 || M <= 0 || D <= 0 || O <= 0) return 0;
  const WgradGeom g = wgrad_geom(B, h, M, D);
  const size_t o_chunk = O < 128 ? O : 128;
  size_t need = static_cast<size_t>(g.q) * o_chunk * (static_cast<size_t>(h) * M + 1);
  if (h == M) {   // (the symmetric layer's geometry: fewer columns, more partials)
    const WgradGeom gs = wgrad_geom(B, h, M, D, true);
    const size_t ns = static_cast<size_t>(gs.q) * o_chunk * (static_cast<size_t>(M) * (M + 1) / 2 + 1);
    if (ns > need) need = ns;
    // k_cin_prep_wsym stages the folded weight slices of k_cin_bwd_data_sym in the same workspace: [tiles][ot * 32][32].
    // (Small shapes -- 5 fields, 8 feature maps, 64 samples: 3072 floats against 832 of partials -- used to write past
    // the end of the buffer: zeros over whatever the allocator had placed behind it, found as an order-dependent
    // failure of tests/test_gpu_reference_matrix.py in the second session of round 4.)
    const size_t ntiles = (M & 1) ? (M + 1) / 2 : (M + 1) / 2 + 1;
    const size_t nw = ntiles * ((o_chunk + 31) / 32) * 32 * 32;
    if (nw > need) need = nw;
  }
  return need;
}

extern "C" int dctr_cin_layer_fwd(const float* H, int64_t ld_h, const float* X0, int64_t ld_x0, const float* W,
                                  const float* bias, int32_t B, int32_t h, int32_t M, int32_t D, int32_t O,
                                  int32_t relu, float* A, int64_t ld_a, float* workspace,
                                  dctr_stream_t stream) {
  if (!H || !X0 || !W || !A || !workspace || B < 0 || h <= 0 || M <= 0 || D <= 0 || O <= 0) return DCTR_EINVAL;
  if (ld_h < static_cast<int64_t>(h) * D || ld_x0 < static_cast<int64_t>(M) * D ||
      ld_a < static_cast<int64_t>(O) * D)
    return DCTR_EINVAL;
  if (M > 32) return DCTR_ENOSUP;
  if (B == 0) return DCTR_OK;
  const int M_pad = (M + 1) / 2 * 2, O_pad = (O + 31) / 32 * 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t total = static_cast<int64_t>(h) * M_pad * O_pad;
  // layer 1 of a CIN: the hidden state IS the field matrix -- the outer product is symmetric
  static const bool sym_ok = !(getenv("DCTR_CIN_SYM") && getenv("DCTR_CIN_SYM")[0] == '0');   // (A/B switch)
  const int sym = (sym_ok && H == X0 && ld_h == ld_x0 && h == M) ? 1 : 0;
  k_cin_prep_w<<<dim3(static_cast<unsigned>((total + kT - 1) / kT)), dim3(kT), 0, s>>>(W, O, h, M, M_pad, O_pad,
                                                                                       workspace, sym);
  const int64_t ncol = static_cast<int64_t>(B) * D;
  // outputs per workgroup: 128 (4 row tiles), or 64 -- two workgroups per CU then share the columns' work: twice the
  // waves per SIMD to hide the per-h barrier and the LDS waits (DCTR_CIN_FWD_CHUNK=64|128; A/B switch)
  static const int chunk_o = (getenv("DCTR_CIN_FWD_CHUNK") && atoi(getenv("DCTR_CIN_FWD_CHUNK")) == 64) ? 64 : 128;
  const int ychunks = (O_pad + chunk_o - 1) / chunk_o;
  const dim3 grid(static_cast<unsigned>((ncol + kCols - 1) / kCols), ychunks);
  // every y-chunk but the last has chunk_o / 32 row tiles; launch the last one separately if it is narrower
  const int ot_last = (O_pad - (ychunks - 1) * chunk_o) / 32;
  static const bool two_waves = !(getenv("DCTR_CIN_FWD_CT") && getenv("DCTR_CIN_FWD_CT")[0] == '2');   // (A/B switch)
  auto launch = [&](int ot, dim3 g, int ybase) {
    const size_t lds = (static_cast<size_t>(M_pad) * kCols + 2u * M_pad * ot * 32) * sizeof(float);
    const float* wt = workspace + ybase * chunk_o;
    const float* bs = bias ? bias + ybase * chunk_o : nullptr;
    float* a = A + static_cast<int64_t>(ybase) * chunk_o * D;
    const int o_here = O - ybase * chunk_o;
#define DCTR_CIN_FWD(OT_, CT_)                                                                                       \
  k_cin_fwd<OT_, CT_><<<g, dim3(kT * 2 / CT_), lds, s>>>(X0, ld_x0, H, ld_h, h, M, M_pad, D, B, wt, O_pad, o_here, bs, \
                                                         relu, a, ld_a, sym)
    // wide output tiles (3-4 row tiles = 96-128 accumulator registers per column tile): one column tile per wave, eight
    // waves; narrow ones: two column tiles per wave, four waves
    switch (ot) {
      case 1: DCTR_CIN_FWD(1, 2); break;
      case 2: if (two_waves && chunk_o == 64) DCTR_CIN_FWD(2, 1); else DCTR_CIN_FWD(2, 2); break;
      case 3: if (two_waves) DCTR_CIN_FWD(3, 1); else DCTR_CIN_FWD(3, 2); break;
      default: if (two_waves) DCTR_CIN_FWD(4, 1); else DCTR_CIN_FWD(4, 2); break;
    }
#undef DCTR_CIN_FWD
  };
  if (ot_last == chunk_o / 32) {
    launch(ot_last, dim3(grid.x, ychunks), 0);            // every chunk has the same width: one launch
  } else {
    if (ychunks > 1) launch(chunk_o / 32, dim3(grid.x, ychunks - 1), 0);
    launch(ot_last, dim3(grid.x, 1), ychunks - 1);
  }
  return launch_status();
}

extern "C" int dctr_cin_layer_bwd(const float* gA, const float* A, int64_t ld_a, int32_t relu, const float* H,
                                  int64_t ld_h, const float* X0, int64_t ld_x0, const float* W, int32_t B,
                                  int32_t h, int32_t M, int32_t D, int32_t O, float* gH, int64_t ld_gh,
                                  float* gX0, int64_t ld_gx, int32_t accumulate_x0, float* gW, float* gbias,
                                  float* workspace, dctr_stream_t stream) {
  if (!gA || !H || !X0 || !W || !gH || !gX0 || !gW || B < 0 || h <= 0 || M <= 0 || D <= 0 || O <= 0)
    return DCTR_EINVAL;
  if (B > 0 && !workspace) return DCTR_EINVAL;
  if (relu && !A) return DCTR_EINVAL;
  if (M > 32) return DCTR_ENOSUP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int K = h * M;
  if (B == 0) {   // an empty batch has zero gradients
    hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * static_cast<size_t>(O) * K, s);
    if (e != hipSuccess) return static_cast<int>(e);
    if (gbias) {
      e = hipMemsetAsync(gbias, 0, sizeof(float) * static_cast<size_t>(O), s);
      if (e != hipSuccess) return static_cast<int>(e);
    }
    return DCTR_OK;
  }
  static const bool sym_ok = !(getenv("DCTR_CIN_SYM") && getenv("DCTR_CIN_SYM")[0] == '0');   // (A/B switch)
  const bool sym = sym_ok && H == X0 && ld_h == ld_x0 && h == M;      // layer 1: see k_cin_prep_w
  const WgradGeom geo = wgrad_geom(B, h, M, D, sym);
  if (geo.lds > 160u * 1024u) return DCTR_ENOSUP;
  const int KW = sym ? M * (M + 1) / 2 : K;     // columns of a partial tile set
  const int64_t ncol = static_cast<int64_t>(B) * D;
  const int chunks = (O + 127) / 128;
  for (int ch = 0; ch < chunks; ++ch) {
    const int o0 = ch * 128;
    const int o_here = (O - o0) < 128 ? (O - o0) : 128;
    const int ot = (o_here + 31) / 32;
    const float* gA_c = gA + static_cast<int64_t>(o0) * D;
    const float* A_c = A ? A + static_cast<int64_t>(o0) * D : nullptr;
    const float* W_c = W + static_cast<int64_t>(o0) * K;
    // data side: the chunks of o add up in gH / gX0
    {
      const dim3 grid(static_cast<unsigned>((ncol + kCols - 1) / kCols));
      size_t lds = 2u * ot * 32 * 32 * sizeof(float);
      if (sym) lds += 16u * kCols * sizeof(float);      // X0's upper half by column
      const int acc_h = ch > 0, acc_x = (ch > 0) || accumulate_x0;
#define DCTR_CIN_BD(OT_, CT_)                                                                                  \
  k_cin_bwd_data<OT_, CT_><<<grid, dim3(kT * 2 / CT_), lds, s>>>(gA_c, A_c, ld_a, relu, X0, ld_x0, H, ld_h, h, M, D, B, \
                                                                W_c, o_here, gH, ld_gh, acc_h, gX0, ld_gx, acc_x)
      static const bool bd_two_waves = !(getenv("DCTR_CIN_BWD_CT") && getenv("DCTR_CIN_BWD_CT")[0] == '2');   // (A/B switch)
#define DCTR_CIN_BDS(OT_, CT_)                                                                                      \
  k_cin_bwd_data_sym<OT_, CT_><<<grid, dim3(kT * 2 / CT_), lds, s>>>(gA_c, A_c, ld_a, relu, X0, ld_x0, M, D, B, workspace, \
                                                                    o_here, gH, ld_gh, acc_h, gX0, ld_gx, acc_x)
      static const bool sym_data = !(getenv("DCTR_CIN_SYM_DATA") && getenv("DCTR_CIN_SYM_DATA")[0] == '0');   // (A/B switch)
      static const bool flat_ok = !(getenv("DCTR_CIN_FLAT") && getenv("DCTR_CIN_FLAT")[0] == '0');   // (A/B switch)
      if (!(sym && sym_data) && flat_ok && M == 26 && ot >= 3) {
        // the padded rows of the per-field tiles cost more than the flattened rows' bookkeeping (298 -> .. us at h = 64)
        if (ot == 3)
          k_cin_bwd_data_flat<3, 26><<<grid, dim3(kT * 2), lds, s>>>(gA_c, A_c, ld_a, relu, X0, ld_x0, H, ld_h, h, D, B, W_c,
                                                                    o_here, gH, ld_gh, acc_h, gX0, ld_gx, acc_x);
        else
          k_cin_bwd_data_flat<4, 26><<<grid, dim3(kT * 2), lds, s>>>(gA_c, A_c, ld_a, relu, X0, ld_x0, H, ld_h, h, D, B, W_c,
                                                                    o_here, gH, ld_gh, acc_h, gX0, ld_gx, acc_x);
      } else
      if (sym && sym_data) {
        const int ntiles = (M & 1) ? (M + 1) / 2 : (M + 1) / 2 + 1;
        const int64_t nws = static_cast<int64_t>(ntiles) * ot * 32 * 32;
        // (the workspace is free until the weight side, behind this launch on the stream, writes its partials)
        k_cin_prep_wsym<<<dim3(static_cast<unsigned>((nws + kT - 1) / kT)), dim3(kT), 0, s>>>(W_c, o_here, M, ot * 32, ntiles,
                                                                                              workspace);
        switch (ot) {
          case 1: DCTR_CIN_BDS(1, 2); break;
          case 2: DCTR_CIN_BDS(2, 2); break;
          case 3: if (bd_two_waves) DCTR_CIN_BDS(3, 1); else DCTR_CIN_BDS(3, 2); break;
          default: if (bd_two_waves) DCTR_CIN_BDS(4, 1); else DCTR_CIN_BDS(4, 2); break;
        }
      } else
      switch (ot) {
        case 1: DCTR_CIN_BD(1, 2); break;
        case 2: DCTR_CIN_BD(2, 2); break;
        case 3: if (bd_two_waves) DCTR_CIN_BD(3, 1); else DCTR_CIN_BD(3, 2); break;
        default: if (bd_two_waves) DCTR_CIN_BD(4, 1); else DCTR_CIN_BD(4, 2); break;
      }
#undef DCTR_CIN_BD
#undef DCTR_CIN_BDS
    }
    // weight side: Q partial [o_here, K] tile sets (+ bias partials) in the workspace, then a fixed-order sum
    {
      const dim3 grid(geo.gx, geo.q);
      float* part = workspace;
      float* bpart = gbias ? workspace + static_cast<size_t>(geo.q) * o_here * KW : nullptr;
#define DCTR_CIN_BW(OT_)                                                                                              \
  do {                                                                                                                \
    if (geo.lds > 64u * 1024u)                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cin_wgrad<OT_>),                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(geo.lds));               \
    k_cin_wgrad<OT_><<<grid, dim3(kT), geo.lds, s>>>(gA_c, A_c, ld_a, relu, X0, ld_x0, H, ld_h, h, M, D, B, o_here,   \
                                                     geo.hspan, geo.ph, part, bpart, sym ? 1 : 0);                    \
  } while (0)
      switch (ot) { case 1: DCTR_CIN_BW(1); break; case 2: DCTR_CIN_BW(2); break; case 3: DCTR_CIN_BW(3); break; default: DCTR_CIN_BW(4); break; }
#undef DCTR_CIN_BW
      const int64_t n = static_cast<int64_t>(o_here) * KW;
      const int nblk_w = static_cast<int>((4 * n + kT - 1) / kT);
      const int nblk_b = gbias ? (o_here + kT / 64 - 1) / (kT / 64) : 0;      // the bias sums ride in the same launch
      if (sym)
        k_cin_wgrad_reduce_sym<<<dim3(static_cast<unsigned>(nblk_w + nblk_b)), dim3(kT), 0, s>>>(
            part, geo.q, o_here, M, gW + static_cast<int64_t>(o0) * K, nblk_w, bpart, o_here, gbias ? gbias + o0 : nullptr);
      else
        k_cin_wgrad_reduce<<<dim3(static_cast<unsigned>(nblk_w + nblk_b)), dim3(kT), 0, s>>>(
            part, geo.q, n, gW + static_cast<int64_t>(o0) * K, nblk_w, bpart, o_here, gbias ? gbias + o0 : nullptr);
    }
  }
  return launch_status();
}

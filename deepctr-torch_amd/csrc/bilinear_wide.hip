// bilinear_wide.hip -- FiBiNET's bilinear pairs TOGETHER WITH the first tower layer behind them, backward direction
// (fibinet.py:82-99: dnn_input = [ Bilinear(senet) | Bilinear(raw) | dense ], h1 = relu(W0 dnn_input + b0);
//  interaction.py:140-156; core.py:123-133).
//
// The product slab P [B, 2 * 325 * 16] is 170 MB at the Criteo shape.  Up to round 5 its GRADIENT was a slab too: a library
// GEMM wrote gP = gh W0 (170 MB), k_bilinear_bwd_data_own read it twice and k_bilinear_bwd_weight a third time -- 349 us
// of a 0.83 ms step moving bytes that exist only between two kernels.  Here the gradient of a (16 samples x one pair)
// piece is MADE where it is consumed:
//     G[b][e]  = sum_h gh[b][h] W0[h][16 k + e]            32 x v_mfma_f32_16x16x4_f32 per pass, exact fp32
//     gX_j    += G (.) (x_i W_k^T)                          as k_bilinear_bwd_data
//     gX_i    += (G (.) x_j) W_k
//     gW_k    += (G (.) x_j)^T x_i                          as k_bilinear_bwd_weight, per-workgroup partials
// A workgroup owns 16 samples: their gh rows sit in registers as the A operand for the whole launch (32 values per
// lane), the pair's 16 columns of W0 arrive from L2 through a register ring as the B operand, in a layout packed once per
// step (k_wide_pack: one dwordx4 per lane and 16 hidden units, 1 KB per wave instruction).  The gradient slab is never
// written or read; the only slab-sized traffic left in the backward is the weight-gradient GEMM's read of P itself.
//
// Schedule: groups of four field-disjoint pairs, one per wave, a barrier per group (the rounds of the tournament hold 13
// pairs at 26 fields: 4 + 4 + 4 + 1 would leave three SIMDs idle for a quarter of every round).  The per-field gradient
// tiles in LDS are plain read-modify-writes in group order: no float atomics, bit-reproducible.
#include "pairwise_tiles.hpp"

namespace {

constexpr int kNQ = 8;          // groups of 16 hidden units: H <= 128
constexpr int kD = 16;          // embedding width of the fused route

inline size_t wide_tile_bytes(int F, int tiles) {
  int rs = F * kD;
  rs += (16 - (rs & 31)) & 31;
  return static_cast<size_t>(tiles) * kSB * rs * sizeof(float);
}

// W0 [H, ldw] (nn.Linear weight: row h = hidden unit) -> Wpk[kb][q][lane] (dwordx4): component s of lane (g, c) is
// W0[16 q + 4 g + s][16 kb + c], zero past H.  The MFMA contraction index of step 4 q + s in lane group g is that h.
__global__ __launch_bounds__(kT) void k_wide_pack(const float* __restrict__ W0, int64_t ldw, int H, int KB,
                                                  f32x4* __restrict__ Wpk) {
  __shared__ float t[16 * kNQ][65];
  const int kb0 = blockIdx.x * 4;
  const int tid = threadIdx.x, col = tid & 63, r0 = tid >> 6;
  const int ncol = (KB - kb0) * 16 < 64 ? (KB - kb0) * 16 : 64;
  const int cc = col < ncol ? col : 0;
  for (int h0 = 0; h0 < 16 * kNQ; h0 += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int h = h0 + r0 + 4 * u;
      v[u] = ldg_f32(W0 + static_cast<int64_t>(h < H ? h : 0) * ldw + kb0 * 16 + cc);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int h = h0 + r0 + 4 * u;
      t[h][col] = (h < H && col < ncol) ? v[u] : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kNQ; ++u) {
    const int idx = tid + kT * u;       // (kb, q, lane) of this block's 4 * kNQ * 64 pieces
    const int kbl = idx / (kNQ * 64), rem = idx - kbl * (kNQ * 64), q = rem >> 6, lane = rem & 63;
    const int g = lane >> 4, c = lane & 15;
    if (kb0 + kbl < KB) {
      f32x4 v;
      v.x = t[16 * q + 4 * g + 0][kbl * 16 + c];
      v.y = t[16 * q + 4 * g + 1][kbl * 16 + c];
      v.z = t[16 * q + 4 * g + 2][kbl * 16 + c];
      v.w = t[16 * q + 4 * g + 3][kbl * 16 + c];
      *(DCTR_GLOBAL f32x4*)(Wpk + (static_cast<int64_t>(kb0 + kbl) * kNQ + q) * 64 + lane) = v;
    }
  }
}

// barrier between groups: LDS traffic only -- the register ring's loads stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// sched4: [n_groups][4 waves][4] int32 = {i, j, weight index, pair index k}; i = -1: the wave idles in this group.
// part:   [tiles][P][16][16] per-workgroup partial of gW_k (row e, column d), every (tile, k) written exactly once.
template <int PD, int VAR = 0>
__global__ __launch_bounds__(kT) void k_bilinear_bwd_wide(const float* __restrict__ E, int64_t lde,
                                                          const float* __restrict__ V, int64_t ldv,
                                                          const float* __restrict__ Wf,
                                                          const int32_t* __restrict__ sched4, int n_groups, int P,
                                                          int F, int B, const float* __restrict__ gh, int64_t ldgh,
                                                          int H, const f32x4* __restrict__ Wpk,
                                                          float* __restrict__ gE, float* __restrict__ gV,
                                                          float* __restrict__ part) {
  extern __shared__ __align__(16) float smem[];
  const int RS = row_stride(F, kD), W = F * kD;
  float* xs0 = smem;               // V tile (pass 0: columns [0, 16 P) of the DNN input)
  float* xs1 = xs0 + kSB * RS;     // E tile (pass 1)
  float* gx0 = xs1 + kSB * RS;     // gV
  float* gx1 = gx0 + kSB * RS;     // gE
  float* tb = gx1 + kSB * RS;      // [4 waves][2 passes][16][17] layout-change scratch (wave-private)
  int32_t* sch = reinterpret_cast<int32_t*>(tb + 4 * 2 * 16 * 17);   // [n_groups][4][4]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kSB;
  for (int e = tid; e < 4 * n_groups; e += kT)
    *reinterpret_cast<i32x4*>(sch + 4 * e) = *(const DCTR_GLOBAL i32x4*)(sched4 + 4 * e);
  // this lane's A operand of G: gh[b = c][h = 16 q + 4 g + s] (zero past B / H)
  f32x4 ga[kNQ];
  {
    const int b = b0 + c;
    const float* row = gh + static_cast<int64_t>(b < B ? b : B - 1) * ldgh;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) {
      const int h = 16 * q + 4 * g;
      ga[q] = *(const DCTR_GLOBAL f32x4*)(row + (h < H ? h : 0));
    }
#pragma unroll
    for (int q = 0; q < kNQ; ++q)
      if (!(b < B && 16 * q + 4 * g < H)) ga[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  stage_rows(xs0, RS, V, ldv, b0, B, W);
  stage_rows(xs1, RS, E, lde, b0, B, W);
  for (int e = tid; e < 2 * kSB * RS; e += kT) gx0[e] = 0.f;
  __syncthreads();
  float* tb0 = tb + wv * (2 * 16 * 17);
  float* tb1 = tb0 + 16 * 17;
  auto entry = [&](int gi) {
    PairEnt e;
    const int gc = gi < n_groups ? gi : n_groups - 1;
    const i32x4 v = *reinterpret_cast<const i32x4*>(sch + 4 * (4 * gc + wv));
    e.i = gi < n_groups ? v.x : -1; e.j = v.y; e.wi = v.z; e.k = v.w;
    return e;
  };
  // ring: the two passes' 16 columns of W0 (8 dwordx4 each) and the pair's own weight tile in both operand layouts
  f32x4 w0[PD][kNQ], w1[PD][kNQ], wr[PD];
  float wtr[PD][4];
  auto issue = [&](int gi, int u) {
    const PairEnt e = entry(gi);
    const f32x4* p0 = Wpk + static_cast<int64_t>(e.k) * (kNQ * 64) + lane;
    const f32x4* p1 = Wpk + static_cast<int64_t>(P + e.k) * (kNQ * 64) + lane;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) {
      w0[u][q] = *(const DCTR_GLOBAL f32x4*)(p0 + 64 * q);
      w1[u][q] = *(const DCTR_GLOBAL f32x4*)(p1 + 64 * q);
    }
    const float* base = Wf + static_cast<int64_t>(e.wi) * (kD * kD);
    wr[u] = *(const DCTR_GLOBAL f32x4*)(base + c * kD + 4 * g);             // W[e = c][d = 4g + s]
#pragma unroll
    for (int s = 0; s < 4; ++s) wtr[u][s] = ldg_f32(base + (4 * g + s) * kD + c);   // W[e = 4g + s][d = c]
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    issue(u, u);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int64_t tile = blockIdx.x;
  auto body = [&](int gi, int u) {
    const PairEnt en = entry(gi);
    const bool live = en.i >= 0;
    const int i = live ? en.i : 0, j = live ? en.j : 0;
    // every LDS operand of the pair, read before the long MFMA block (an idle wave reads field 0: unused)
    f32x4 a[2];
    float xj[2][4], xi[2][4], gj[2][4], gi_[2][4];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const float* xs = ps ? xs1 : xs0;
      const float* gx = ps ? gx1 : gx0;
      a[ps] = *reinterpret_cast<const f32x4*>(xs + c * RS + i * kD + 4 * g);      // x_i[b = c][d = 4g + s]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xj[ps][r] = xs[(4 * g + r) * RS + j * kD + c];      // x_j[b = 4g + r][e = c]
        xi[ps][r] = xs[(4 * g + r) * RS + i * kD + c];      // x_i[b = 4g + r][d = c]
        gj[ps][r] = gx[(4 * g + r) * RS + j * kD + c];
        gi_[ps][r] = gx[(4 * g + r) * RS + i * kD + c];
      }
    }
    // G[b = 4g + r][e = c] of both passes: two independent chains of 32
    f32x4 G0 = {0.f, 0.f, 0.f, 0.f}, G1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < kNQ; ++q) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        G0 = mfma16(ga[q][s], w0[u][q][s], G0);
        G1 = mfma16(ga[q][s], w1[u][q][s], G1);
      }
    }
    const f32x4 wreg = wr[u];
    float wT[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wT[s] = wtr[u][s];
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & 4)) issue(gi + PD, u);
    __builtin_amdgcn_sched_barrier(0);
    // t[b = 4g + r][e = c] = (x_i W^T)
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      t0 = mfma16(a[0][s], wreg[s], t0);
      t1 = mfma16(a[1][s], wreg[s], t1);
    }
    // u = G (.) x_j in the accumulator layout: the A operand of gW as it is, of gX_i after a transposition in LDS
    float u0[4], u1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      u0[r] = G0[r] * xj[0][r];
      u1[r] = G1[r] * xj[1][r];
      tb0[(4 * g + r) * 17 + c] = u0[r];
      tb1[(4 * g + r) * 17 + c] = u1[r];
    }
    f32x4 aw = {0.f, 0.f, 0.f, 0.f};       // gW_k[e = 4g + r][d = c], both passes
#pragma unroll
    for (int s = 0; s < 4; ++s) aw = mfma16(u0[s], xi[0][s], aw);
#pragma unroll
    for (int s = 0; s < 4; ++s) aw = mfma16(u1[s], xi[1][s], aw);
    float ua0[4], ua1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ua0[s] = tb0[c * 17 + 4 * g + s];      // u[b = c][e = 4g + s]
      ua1[s] = tb1[c * 17 + 4 * g + s];
    }
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};   // (u W)[b = 4g + r][d = c]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      v0 = mfma16(ua0[s], wT[s], v0);
      v1 = mfma16(ua1[s], wT[s], v1);
    }
    if (live) {
      if (!(VAR & 8))
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gx0[(4 * g + r) * RS + j * kD + c] = gj[0][r] + G0[r] * t0[r];
        gx1[(4 * g + r) * RS + j * kD + c] = gj[1][r] + G1[r] * t1[r];
        gx0[(4 * g + r) * RS + i * kD + c] = gi_[0][r] + v0[r];
        gx1[(4 * g + r) * RS + i * kD + c] = gi_[1][r] + v1[r];
      }
      float* dst = part + (tile * P + en.k) * (kD * kD) + (4 * g) * kD + c;
      if (!(VAR & 2) || aw[0] == 12345.f)
#pragma unroll
      for (int r = 0; r < 4; ++r) stg_f32(dst + r * kD, aw[r]);
      if ((VAR & 8) && v0[0] + v1[0] + t0[0] + t1[0] == 12345.f) gx0[lane] = 1.f;
    }
    if (!(VAR & 1)) lds_barrier();       // the next group touches other fields' columns of gx
  };
  for (int gi0 = 0; gi0 < n_groups; gi0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (gi0 + u < n_groups) body(gi0 + u, u);      // (uniform over the workgroup)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the two gradient tiles leave in dwordx4 pieces (RS and W are multiples of 4)
  const int w4 = W >> 2;
  for (int e = tid; e < kSB * w4; e += kT) {
    const int r = e / w4, q = e - r * w4;
    if (b0 + r < B) {
      *(DCTR_GLOBAL f32x4*)(gV + static_cast<int64_t>(b0 + r) * W + 4 * q) =
          *reinterpret_cast<const f32x4*>(gx0 + r * RS + 4 * q);
      *(DCTR_GLOBAL f32x4*)(gE + static_cast<int64_t>(b0 + r) * W + 4 * q) =
          *reinterpret_cast<const f32x4*>(gx1 + r * RS + 4 * q);
    }
  }
}

// gW[pair_w[k]] = sum over the tiles' partials of pair k, fixed order: four slices of the tiles per workgroup, each
// lane 16 loads in flight, the slices combined in slice order.  (One weight per pair: the "interaction" type.)
__global__ __launch_bounds__(1024) void k_wide_reduce_w(const float* __restrict__ part, int tiles, int P,
                                                       const int32_t* __restrict__ pair_w,
                                                       float* __restrict__ gW) {
  __shared__ float red[4][kD * kD];
  const int k = blockIdx.x, el = threadIdx.x & 255, sl = threadIdx.x >> 8;
  const int per = (tiles + 3) / 4, t0 = sl * per, t1 = t0 + per < tiles ? t0 + per : tiles;
  float s = 0.f;
  for (int tb = t0; tb < t1; tb += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = tb + u < t1 ? tb + u : t1 - 1;
      v[u] = ldg_f32(part + (static_cast<int64_t>(t) * P + k) * (kD * kD) + el);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (tb + u < t1) s += v[u];
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0) gW[static_cast<int64_t>(ldg_i32(pair_w + k)) * (kD * kD) + el] = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
}

size_t wide_pack_floats(int P) { return static_cast<size_t>(2) * P * kNQ * 64 * 4; }

}  // namespace

extern "C" size_t dctr_bilinear_wide_bwd_workspace_floats(int32_t B, int32_t P) {
  const size_t tiles = static_cast<size_t>((B > 0 ? B : 1) + kSB - 1) / kSB;
  return wide_pack_floats(P > 0 ? P : 1) + tiles * (P > 0 ? P : 1) * kD * kD;
}

extern "C" int dctr_bilinear_wide_bwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                                      const int32_t* sched4, int32_t n_groups, const int32_t* pair_w, int32_t n_w,
                                      int32_t P, int32_t F, int32_t D, int32_t B, const float* gh, int64_t ld_gh,
                                      const float* W0, int64_t ld_w0, int32_t H, float* gE, float* gV, float* gW,
                                      float* workspace, dctr_stream_t stream) {
  if (!E || !V || !Wf || !sched4 || !pair_w || !gh || !W0 || !gE || !gV || !gW || !workspace || B < 0 || F < 2 ||
      P <= 0 || n_groups <= 0 || n_w <= 0 || H <= 0)
    return DCTR_EINVAL;
  // one weight per pair, 16-wide embeddings, at most 128 hidden units in dwordx4 pieces
  if (D != kD || n_w != P || H > 16 * kNQ || (H & 3) || (ld_gh & 3) || (reinterpret_cast<uintptr_t>(gh) & 15) ||
      (reinterpret_cast<uintptr_t>(Wf) & 15) || (reinterpret_cast<uintptr_t>(gE) & 15) ||
      (reinterpret_cast<uintptr_t>(gV) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return DCTR_ENOSUP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) {
    (void)hipMemsetAsync(gW, 0, sizeof(float) * n_w * kD * kD, s);
    return DCTR_OK;
  }
  const size_t lds = wide_tile_bytes(F, 4) + 4u * 2 * 16 * 17 * sizeof(float) + static_cast<size_t>(n_groups) * 64;
  if (lds > 158 * 1024) return DCTR_ENOSUP;
  f32x4* Wpk = reinterpret_cast<f32x4*>(workspace);
  float* part = workspace + wide_pack_floats(P);
  const int KB = 2 * P, tiles = (B + kSB - 1) / kSB;
  k_wide_pack<<<dim3((KB + 3) / 4), dim3(kT), 0, s>>>(W0, ld_w0, H, KB, Wpk);
#define DCTR_WIDE(...)                                                                                            \
  do {                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bilinear_bwd_wide<__VA_ARGS__>),                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));                 \
    k_bilinear_bwd_wide<__VA_ARGS__><<<dim3(tiles), dim3(kT), lds, s>>>(E, ld_e, V, ld_v, Wf, sched4, n_groups, P, \
                                                                        F, B, gh, ld_gh, H, Wpk, gE, gV, part);   \
  } while (0)
#ifdef DCTR_DIAG
  // timing variants (tools/probes/wide_bwd_probe.py; results are wrong for VAR != 0): DCTR_WIDE_VAR = "<PD><VAR>"
  const char* e = getenv("DCTR_WIDE_VAR");
  const int pd = e && e[0] ? e[0] - '0' : 2, var = e && e[0] && e[1] ? atoi(e + 1) : 0;
  if (pd == 1) DCTR_WIDE(1, 0);
  else if (pd == 3) DCTR_WIDE(3, 0);
  else if (var == 1) DCTR_WIDE(2, 1);
  else if (var == 2) DCTR_WIDE(2, 2);
  else if (var == 4) DCTR_WIDE(2, 4);
  else if (var == 8) DCTR_WIDE(2, 8);
  else if (var == 15) DCTR_WIDE(2, 15);
  else DCTR_WIDE(2, 0);
#else
  DCTR_WIDE(2, 0);
#endif
#undef DCTR_WIDE
  k_wide_reduce_w<<<dim3(P), dim3(1024), 0, s>>>(part, tiles, P, pair_w, gW);
  return launch_status();
}

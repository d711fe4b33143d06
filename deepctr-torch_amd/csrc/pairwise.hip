// pairwise.hip -- the pair-structured interaction layers of FiBiNET and PNN on gfx950:
//   SENETLayer          (interaction.py:93-101)   z = mean_d E ; a = relu(W2 relu(W1 z)) ; V = E * a
//   BilinearInteraction (interaction.py:140-156)  p_k = (x_i W_k^T) (.) x_j   for the F(F-1)/2 pairs, types all/each/interaction
//   InnerProductLayer   (interaction.py:557-577)  p_k = sum_d e_i e_j  (or the un-reduced product)
//
// The reference issues 325 tiny nn.Linear calls + 325 muls + a 650-way cat per Bilinear call (10.8 k launches per
// FiBiNET step, 62 % of its time in aten::cat).  Here a workgroup owns 16 samples, keeps their [16, F*D] embedding
// tile in LDS, and walks the pairs; the 16x16 weight tiles go through v_mfma_f32_16x16x4_f32 (exact fp32) with the
// 16 samples as the row dimension.  FiBiNET applies the SAME weights to the SENET output V and to the raw E
// (fibinet.py:82-83): both passes share one weight-tile load, and the kernel writes straight into the reference's
// DNN-input layout  [ V pairs | E pairs | dense ]  (fibinet.py:86-87) -- nothing is concatenated afterwards.
//
// Determinism: pairs are visited in round-robin-tournament order -- every round is a perfect matching, so the four
// waves of a workgroup never touch the same field inside a round and the per-field gradient tiles in LDS are
// accumulated with plain read-modify-writes in round order (no float atomics, bit-reproducible).  Parameter
// gradients (reductions over the batch) go to per-workgroup partial slabs that a second kernel sums in fixed order.
#include "pairwise_tiles.hpp"

namespace {

// ------------------------------------------------------------------------------------------------------------
// SENET
// ------------------------------------------------------------------------------------------------------------
constexpr int kSS = 8;  // samples per workgroup in the SENET kernels

__global__ __launch_bounds__(kT) void k_senet_fwd(const float* __restrict__ E, int64_t lde, int B, int F, int D,
                                                  const float* __restrict__ W1, const float* __restrict__ W2, int R,
                                                  float* __restrict__ V, float* __restrict__ a_out,
                                                  float* __restrict__ a1_out) {
  extern __shared__ __align__(16) float smem[];
  const int W = F * D;
  float* es = smem;             // [kSS][W]
  float* z = es + kSS * W;      // [kSS][F]
  float* a1 = z + kSS * F;      // [kSS][R]
  float* a = a1 + kSS * R;      // [kSS][F]
  float* w1s = a + kSS * F;     // [R][F]   both weight matrices once per workgroup: the two small products below walked
  float* w2s = w1s + R * F;     // [F][R]   them through 26 / 8 dependent global loads per output
  const int tid = threadIdx.x, b0 = blockIdx.x * kSS;
  for (int e = tid; e < R * F; e += kT) {
    w1s[e] = ldg_f32(W1 + e);
    w2s[e] = ldg_f32(W2 + e);
  }
  stage_rows<kSS>(es, W, E, lde, b0, B, W);
  __syncthreads();
  for (int e = tid; e < kSS * F; e += kT) {  // torch.mean(inputs, dim=-1)
    const int r = e / F, f = e - r * F;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += es[r * W + f * D + d];
    z[e] = s / static_cast<float>(D);
  }
  __syncthreads();
  for (int e = tid; e < kSS * R; e += kT) {  // relu(Linear(F -> R, no bias))
    const int r = e / R, q = e - r * R;
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += z[r * F + f] * w1s[q * F + f];
    a1[e] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < kSS * F; e += kT) {  // relu(Linear(R -> F, no bias))
    const int r = e / F, f = e - r * F;
    float s = 0.f;
    for (int q = 0; q < R; ++q) s += a1[r * R + q] * w2s[f * R + q];
    a[e] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < kSS * W; e += kT) {
    const int r = e / W, c = e - r * W;
    if (b0 + r < B) stg_f32(V + static_cast<int64_t>(b0 + r) * W + c, es[e] * a[r * F + c / D]);
  }
  for (int e = tid; e < kSS * F; e += kT) {
    const int r = e / F;
    if (b0 + r < B) stg_f32(a_out + static_cast<int64_t>(b0) * F + e, a[e]);
  }
  for (int e = tid; e < kSS * R; e += kT) {
    const int r = e / R;
    if (b0 + r < B) stg_f32(a1_out + static_cast<int64_t>(b0) * R + e, a1[e]);
  }
}

// gE = gV*a + (1/D) * W1^T ga1' ;  partial gW1 / gW2 of this workgroup's samples go to part[blockIdx.x]
__global__ __launch_bounds__(kT) void k_senet_bwd(const float* __restrict__ gV, const float* __restrict__ E,
                                                  int64_t lde, int B, int F, int D, const float* __restrict__ W1,
                                                  const float* __restrict__ W2, int R,
                                                  const float* __restrict__ a_in, const float* __restrict__ a1_in,
                                                  float* __restrict__ gE, float* __restrict__ part) {
  extern __shared__ __align__(16) float smem[];
  const int W = F * D;
  float* es = smem;              // [kSS][W]
  float* gv = es + kSS * W;      // [kSS][W]
  float* z = gv + kSS * W;       // [kSS][F]
  float* ga = z + kSS * F;       // [kSS][F]  masked
  float* ga1 = ga + kSS * F;     // [kSS][R]  masked
  float* gz = ga1 + kSS * R;     // [kSS][F]
  float* w1s = gz + kSS * F;     // [R][F]  (as in the forward)
  float* w2s = w1s + R * F;      // [F][R]
  const int tid = threadIdx.x, b0 = blockIdx.x * kSS;
  for (int e = tid; e < R * F; e += kT) {
    w1s[e] = ldg_f32(W1 + e);
    w2s[e] = ldg_f32(W2 + e);
  }
  stage_rows<kSS>(es, W, E, lde, b0, B, W);
  stage_rows<kSS>(gv, W, gV, W, b0, B, W);
  __syncthreads();
  for (int e = tid; e < kSS * F; e += kT) {
    const int r = e / F, f = e - r * F;
    float s = 0.f, g = 0.f;
    for (int d = 0; d < D; ++d) {
      s += es[r * W + f * D + d];
      g += gv[r * W + f * D + d] * es[r * W + f * D + d];
    }
    z[e] = s / static_cast<float>(D);
    const float av = (b0 + r < B) ? ldg_f32(a_in + static_cast<int64_t>(b0) * F + e) : 0.f;
    ga[e] = av > 0.f ? g : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < kSS * R; e += kT) {
    const int r = e / R, q = e - r * R;
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += ga[r * F + f] * w2s[f * R + q];
    const float a1v = (b0 + r < B) ? ldg_f32(a1_in + static_cast<int64_t>(b0) * R + e) : 0.f;
    ga1[e] = a1v > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < kSS * F; e += kT) {
    const int r = e / F, f = e - r * F;
    float s = 0.f;
    for (int q = 0; q < R; ++q) s += ga1[r * R + q] * w1s[q * F + f];
    gz[e] = s / static_cast<float>(D);
  }
  __syncthreads();
  for (int e = tid; e < kSS * W; e += kT) {
    const int r = e / W, c = e - r * W, f = c / D;
    if (b0 + r < B) {
      const float av = ldg_f32(a_in + static_cast<int64_t>(b0 + r) * F + f);
      stg_f32(gE + static_cast<int64_t>(b0 + r) * W + c, gv[e] * av + gz[r * F + f]);
    }
  }
  // parameter-gradient partials: gW1[q][f] = sum_r ga1[r][q] z[r][f] ; gW2[f][q] = sum_r ga[r][f] a1[r][q]
  float* mine = part + static_cast<int64_t>(blockIdx.x) * (2 * R * F);
  for (int e = tid; e < R * F; e += kT) {
    const int q = e / F, f = e - q * F;
    float s1 = 0.f;
    for (int r = 0; r < kSS; ++r) s1 += ga1[r * R + q] * z[r * F + f];
    mine[e] = s1;
  }
  for (int e = tid; e < F * R; e += kT) {
    const int f = e / R, q = e - f * R;
    float s2 = 0.f;
    for (int r = 0; r < kSS; ++r) {
      const float a1v = (b0 + r < B) ? ldg_f32(a1_in + static_cast<int64_t>(b0 + r) * R + q) : 0.f;
      s2 += ga[r * F + f] * a1v;
    }
    mine[R * F + e] = s2;
  }
}

// out[i] = sum_g part[g * stride + i], i < count, in a FIXED order: a workgroup owns 16 outputs; thread (o, sl) adds
// the groups sl, sl + 16, ... (eight loads in flight), the 16 slices are then added in slice order.  (One thread per
// output walking all 512 groups one dependent load at a time took 117 us for 208 outputs.)
__global__ __launch_bounds__(kT) void k_reduce_partials(const float* __restrict__ part, int64_t stride,
                                                        int64_t count, int groups, float* __restrict__ out,
                                                        int64_t split, float* __restrict__ out2) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + o;
  const int64_t ic = i < count ? i : 0;
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
  if (sl == 0 && i < count) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    if (i < split) out[i] = t;        // (two result tensors behind one partial layout: SENET's gW1 | gW2)
    else out2[i - split] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Bilinear.  sched: [n_sched][4] int32 = {i, j, weight index, pair index k}, i = -1 for an idle slot; entries are
// in round-robin order, `slots` entries per round (a perfect matching of the fields).
// MFMA contraction index: d = 4*g + s  (g = lane >> 4, s = step), identical on both operands.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void k_bilinear_fwd(const float* __restrict__ E, int64_t lde,
                                                     const float* __restrict__ V, int64_t ldv,
                                                     const float* __restrict__ Wf, const int32_t* __restrict__ sched,
                                                     int n_sched, int P, int F, int D, int B,
                                                     float* __restrict__ out, int64_t ldo,
                                                     const float* __restrict__ dense, int64_t ldd, int n_dense,
                                                     int dense_off) {
  extern __shared__ __align__(16) float smem[];
  const int RS = row_stride(F, D), W = F * D;
  float* xs0 = smem;               // V tile (pass 0), or E when V == nullptr
  float* xs1 = xs0 + kSB * RS;     // E tile (pass 1)
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kSB;
  const int npass = V ? 2 : 1;
  int32_t* sch = reinterpret_cast<int32_t*>(xs0 + 2 * kSB * RS);   // [n_sched][4]: the schedule
  for (int e = tid; e < n_sched; e += kT)
    *reinterpret_cast<i32x4*>(sch + 4 * e) = *(const DCTR_GLOBAL i32x4*)(sched + 4 * e);
  stage_rows(xs0, RS, V ? V : E, V ? ldv : lde, b0, B, W);
  if (V) stage_rows(xs1, RS, E, lde, b0, B, W);
  if (dense)
    for (int e = tid; e < kSB * n_dense; e += kT) {
      const int r = e / n_dense, q = e - r * n_dense;
      if (b0 + r < B)
        stg_f32(out + static_cast<int64_t>(b0 + r) * ldo + dense_off + q,
                ldg_f32(dense + static_cast<int64_t>(b0 + r) * ldd + q));
    }
  __syncthreads();
  // Software pipeline over this wave's pairs.  The schedule sits in LDS (staged above), so a pair's weight address
  // costs no memory round trip, and the weight tiles of the next kWD pairs are in flight while a pair computes.  What
  // bound this kernel at 1.35 TB/s (round 1) was not HBM but two latency chains, found in the ISA:
  //  * `d < D ? xs[..] : 0` compiles to a branch around the ds_read with its own lgkmcnt(0): ~10 dependent LDS round
  //    trips per pair.  Every LDS operand of a pair -- 4 A values and 4 x_j values per pass, both passes -- is now read
  //    unconditionally (clamped index, select afterwards) before the first MFMA;
  //  * loads and stores share one in-order counter on gfx950, and the compiler cannot count stores that sit behind a
  //    branch: the wait for a weight tile prefetched four pairs earlier became a wait for the PREVIOUS pair's stores
  //    (one write round trip per pair, ~0.7 us).  Full tiles (D == 16, 16 valid samples) therefore run a branch-free
  //    body over whole groups of kWD pairs -- every store unconditional, the waits exact -- and only the ragged
  //    rest takes the predicated body.  (Staging the products through LDS for 2 KB runs per sample was measured
  //    slower than these direct 64-byte pieces: with one workgroup per CU its store bursts do not overlap compute.)
  constexpr int kWD = 4;
  auto entry = [&](int q) {
    PairEnt e;
    const int qc = q < n_sched ? q : n_sched - 1;
    const i32x4 v = *reinterpret_cast<const i32x4*>(sch + 4 * qc);
    e.i = q < n_sched ? v.x : -1; e.j = v.y; e.wi = v.z; e.k = v.w;
    return e;
  };
  float wring[kWD][4];
#pragma unroll
  for (int u = 0; u < kWD; ++u) {
    load_w_raw(Wf, entry(wv + 4 * u), D, g, c, wring[u]);
    __builtin_amdgcn_sched_barrier(0);     // ring order: the loop's waits are derived from it
  }
  const bool cv = c < D;
  const int cc = cv ? c : 0;
  int dcl[4];
  bool dv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    dv[s] = 4 * g + s < D;
    dcl[s] = dv[s] ? 4 * g + s : 0;
  }
  auto pair = [&](auto full_tag, int q, float (&wslot)[4]) {
    constexpr bool FULL = decltype(full_tag)::value;
    const PairEnt en = entry(q);
    float wreg[4];  // B operand: W[e = c][d = 4g + s]
#pragma unroll
    for (int s = 0; s < 4; ++s) wreg[s] = (FULL || (cv && dv[s])) ? wslot[s] : 0.f;
    load_w_raw(Wf, entry(q + 4 * kWD), D, g, c, wslot);
    const bool live = FULL || en.i >= 0;
    const int i = live ? en.i : 0, j = live ? en.j : 0, k = en.k;
    float a[2][4], xj[2][4];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const float* xs = (ps && npass > 1) ? xs1 : xs0;
#pragma unroll
      for (int s = 0; s < 4; ++s) a[ps][s] = xs[c * RS + i * D + dcl[s]];          // A operand: x_i[b = c][d]
#pragma unroll
      for (int r = 0; r < 4; ++r) xj[ps][r] = xs[(4 * g + r) * RS + j * D + cc];   // x_j[b = 4g + r][e = c]
    }
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      t0 = mfma16((FULL || dv[s]) ? a[0][s] : 0.f, wreg[s], t0);
      t1 = mfma16((FULL || dv[s]) ? a[1][s] : 0.f, wreg[s], t1);
    }
    float* col = out + static_cast<int64_t>(k) * D + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // t[r] = (x_i W^T)[b = 4g + r][e = c]
      const int b = 4 * g + r;
      float* row = col + static_cast<int64_t>(b0 + b) * ldo;
      if (FULL) {
        stg_f32(row, t0[r] * xj[0][r]);
        stg_f32(row + static_cast<int64_t>(P) * D, t1[r] * xj[1][r]);
      } else if (live && cv && b0 + b < B) {
        stg_f32(row, t0[r] * xj[0][r]);
        if (npass > 1) stg_f32(row + static_cast<int64_t>(P) * D, t1[r] * xj[1][r]);
      }
    }
  };
  const int mine = (n_sched - wv + 3) / 4;                        // pairs of this wave: q = wv + 4 m
  const bool fast = (D == 16) && (b0 + kSB <= B) && (npass == 2);
  const int full_groups = fast ? mine / kWD : 0;
  int m = 0;
  auto group = [&]() {
#pragma unroll
    for (int u = 0; u < kWD; ++u) {
      pair(std::true_type{}, wv + 4 * (m + u), wring[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
    m += kWD;
  };
  // (the first group is peeled: the loop is then entered with the same loads and stores outstanding as its back edge
  // leaves -- the compiler takes the tighter of the two for the waits at the loop head)
  if (full_groups > 0) group();
  for (int gi = 1; gi < full_groups; ++gi) group();
  for (; m < mine + kWD; m += kWD) {      // the ragged rest (and everything of a partial tile)
#pragma unroll
    for (int u = 0; u < kWD; ++u) pair(std::false_type{}, wv + 4 * (m + u), wring[u]);
  }
}

// gradient w.r.t. the inputs: gX_j += gp (.) t ; gX_i += (gp (.) x_j) W    (per pass: V then E)
__global__ __launch_bounds__(kT) void k_bilinear_bwd_data(const float* __restrict__ E, int64_t lde,
                                                          const float* __restrict__ V, int64_t ldv,
                                                          const float* __restrict__ Wf,
                                                          const int32_t* __restrict__ sched, int n_sched, int slots,
                                                          int P, int F, int D, int B, const float* __restrict__ gout,
                                                          int64_t ldg, float* __restrict__ gE, float* __restrict__ gV,
                                                          int sch_lds) {
  extern __shared__ __align__(16) float smem[];
  const int RS = row_stride(F, D), W = F * D;
  float* xs0 = smem;
  float* xs1 = xs0 + kSB * RS;
  float* gx0 = xs1 + kSB * RS;
  float* gx1 = gx0 + kSB * RS;
  float* tb = gx1 + kSB * RS;  // [4 waves][16][17] layout-change scratch
  // [n_sched][4]: the schedule (16-byte aligned: RS and 4 * 16 * 17 are multiples of 4 floats)
  int32_t* sch = reinterpret_cast<int32_t*>(tb + 4 * 16 * 17);
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kSB;
  const int npass = V ? 2 : 1;
  if (sch_lds)
    for (int e = tid; e < n_sched; e += kT)
      *reinterpret_cast<i32x4*>(sch + 4 * e) = *(const DCTR_GLOBAL i32x4*)(sched + 4 * e);
  stage_rows(xs0, RS, V ? V : E, V ? ldv : lde, b0, B, W);
  if (V) stage_rows(xs1, RS, E, lde, b0, B, W);
  for (int e = tid; e < 2 * kSB * RS; e += kT) gx0[e] = 0.f;
  __syncthreads();
  float* mytb = tb + wv * (16 * 17);
  const int nrounds = (n_sched + slots - 1) / slots;
  // Flat iteration space of a wave: it -> (round rd = it / spw, slot sl = wv + 4 * (it % spw)); every wave runs the
  // same number of iterations, so the barrier after a round's last iteration is uniform.  One-ahead pipeline: while
  // pair `it` computes, the weight tiles and the 8 incoming-gradient values of pair it + 1 and the schedule entry of
  // pair it + 2 are in flight (all loads unconditional on clamped addresses, masked by the consumer; the gradient
  // slab is streamed from HBM, 170 MB at the Criteo shape -- with predicated loads every value was its own round trip).
  const int spw = (slots + 3) / 4, nit = nrounds * spw;
  auto pair_of = [&](int it) -> int {     // schedule index of iteration `it`, or n_sched (= idle) for an empty slot
    const int rd = it / spw, sl = wv + 4 * (it - rd * spw);
    const int q = rd * slots + sl;
    return (it < nit && sl < slots && q < n_sched) ? q : n_sched;
  };
  const int ccl = c < D ? c : 0;
  int64_t grow[4];                         // clamped gout row offsets of this lane's 4 samples
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * g + r;
    grow[r] = static_cast<int64_t>(b < B ? b : B - 1) * ldg + ccl;
  }
  auto load_gp = [&](const PairEnt& e, float (&gp)[2][4]) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        gp[ps][r] = ldg_f32(gout + grow[r] + (static_cast<int64_t>(ps < npass ? ps : 0) * P + e.k) * D);
  };
  // kPD-deep ring: the weight tiles and the 8 incoming-gradient values of the next kPD pairs are in flight while a pair
  // computes (round 1 kept ONE pair ahead: every iteration then cost one HBM round trip -- the gradient slab is
  // streamed, 170 MB at the Criteo shape -- and the kernel ran at 2.2 us per pair).  The schedule sits in LDS, so
  // the address of a pair kPD iterations ahead costs no memory round trip.
  constexpr int kPD = 4;    // (8 was measured slower: 207 vs 194 us -- registers, not latency, then bound the wave)
  auto entry_it = [&](int it) {
    const int q = pair_of(it);
    PairEnt e;
    const int qc = q < n_sched ? q : n_sched - 1;
    // (the widest field counts leave no LDS for the schedule: it is then read from global memory, L2-resident)
    const i32x4 v = sch_lds ? *reinterpret_cast<const i32x4*>(sch + 4 * qc) : *(const DCTR_GLOBAL i32x4*)(sched + 4 * qc);
    e.i = q < n_sched ? v.x : -1; e.j = v.y; e.wi = v.z; e.k = v.w;
    return e;
  };
  float wr[kPD][4], wtr[kPD][4], gpr[kPD][2][4];
#pragma unroll
  for (int u = 0; u < kPD; ++u) {
    const PairEnt e = entry_it(u);
    load_w_raw(Wf, e, D, g, c, wr[u]);
    load_wt_raw(Wf, e, D, g, c, wtr[u]);
    load_gp(e, gpr[u]);
    __builtin_amdgcn_sched_barrier(0);
  }
  const bool cv = c < D;
  int dcl[4];
  bool dv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    dv[s] = 4 * g + s < D;
    dcl[s] = dv[s] ? 4 * g + s : 0;
  }
  for (int it0 = 0; it0 < nit; it0 += kPD) {
#pragma unroll
    for (int u = 0; u < kPD; ++u) {
      const int it = it0 + u;
      if (it < nit) {                       // (uniform over the workgroup: nit is)
        const PairEnt en = entry_it(it);
        float wreg[4], wT[4], gpc[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bool in = cv && dv[s];
          wreg[s] = in ? wr[u][s] : 0.f;     // W[e = c][d]
          wT[s] = in ? wtr[u][s] : 0.f;      // W[e = d'][d = c]
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
          for (int r = 0; r < 4; ++r) gpc[ps][r] = (cv && b0 + 4 * g + r < B) ? gpr[u][ps][r] : 0.f;
        {
          const PairEnt ea = entry_it(it + kPD);
          load_w_raw(Wf, ea, D, g, c, wr[u]);
          load_wt_raw(Wf, ea, D, g, c, wtr[u]);
          load_gp(ea, gpr[u]);
        }
        const bool live = en.i >= 0;
        const int i = live ? en.i : 0, j = live ? en.j : 0;
        // every LDS operand of the pair is read unconditionally before the first MFMA (a predicated ds_read is a
        // branch with its own lgkmcnt(0))
        float a[2][4], xj[2][4], gj[2][4];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const float* xs = (ps && npass > 1) ? xs1 : xs0;
          const float* gx = (ps && npass > 1) ? gx1 : gx0;
#pragma unroll
          for (int s = 0; s < 4; ++s) a[ps][s] = xs[c * RS + i * D + dcl[s]];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            xj[ps][r] = xs[(4 * g + r) * RS + j * D + ccl];
            gj[ps][r] = gx[(4 * g + r) * RS + j * D + ccl];
          }
        }
        f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          t0 = mfma16(dv[s] ? a[0][s] : 0.f, wreg[s], t0);
          t1 = mfma16(dv[s] ? a[1][s] : 0.f, wreg[s], t1);
        }
        if (live) {
          for (int ps = 0; ps < npass; ++ps) {
            float* gx = ps ? gx1 : gx0;
            const f32x4 t = ps ? t1 : t0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int b = 4 * g + r;
              const float gp = ps ? gpc[1][r] : gpc[0][r];
              if (cv) gx[b * RS + j * D + c] = (ps ? gj[1][r] : gj[0][r]) + gp * t[r];   // gX_j[b][e]
              mytb[b * 17 + c] = cv ? gp * (ps ? xj[1][r] : xj[0][r]) : 0.f;             // g_t[b][e] in C layout
            }
            // g_xi[b][d] = sum_e g_t[b][e] W[e][d]: A operand g_t[b = c][e = 4g + s], B operand W[e = 4g + s][d = c]
            float ga[4], gi[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ga[s] = mytb[c * 17 + 4 * g + s];
#pragma unroll
            for (int r = 0; r < 4; ++r) gi[r] = gx[(4 * g + r) * RS + i * D + ccl];
            f32x4 uu = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) uu = mfma16(ga[s], wT[s], uu);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (cv) gx[(4 * g + r) * RS + i * D + c] = gi[r] + uu[r];
          }
        }
        if ((it + 1) % spw == 0) __syncthreads();  // next round touches other (field) columns of gx; rounds are perfect matchings
      }
    }
  }
  for (int e = tid; e < kSB * W; e += kT) {
    const int r = e / W, cc = e - r * W;
    if (b0 + r < B) {
      if (V) {
        stg_f32(gV + static_cast<int64_t>(b0 + r) * W + cc, gx0[r * RS + cc]);
        stg_f32(gE + static_cast<int64_t>(b0 + r) * W + cc, gx1[r * RS + cc]);
      } else {
        stg_f32(gE + static_cast<int64_t>(b0 + r) * W + cc, gx0[r * RS + cc]);
      }
    }
  }
}

// The same gradient, re-dealt BY OWNING FIELD (round 4; the default).  The kernel above walks the pairs in tournament
// rounds and read-modify-writes both fields' gradient rows in LDS: per pair a dependent chain LDS write -> LDS read -> 4
// dependent MFMAs -> LDS read-modify-write (~2 600 cycles for 512 of MFMA), a barrier per round (25 barriers = 32 of its
// 194 us), and 120 KB of LDS = one workgroup of one wave per SIMD per CU.  Here a wave OWNS a field f and walks f's F - 1
// partners in partner order, accumulating gX_f in registers:
//     f is the pair's j:  acc += gp (.) (x_i W^T)          (one MFMA group, as the forward's)
//     f is the pair's i:  acc += (gp (.) x_j) W            (layout change of gp (.) x_j through the wave's own LDS scratch)
// No LDS read-modify-write shared between waves, no barrier after the staging, 70 KB of LDS: two workgroups per CU
// (blockIdx.y splits the fields in two halves), eight waves per CU.  Every pair is visited twice (by its two owners), so
// the incoming gradient is read twice -- the second read comes from L2: both owners sit on the same CU within
// microseconds.  The owner table own[f][slot(partner)] = {partner, side, weight index, pair index} is rebuilt from the
// tournament schedule by every workgroup (no counters: a partner's slot is its index, so the order of additions is fixed).
template <int KPD>
__global__ __launch_bounds__(kT) void k_bilinear_bwd_data_own(const float* __restrict__ E, int64_t lde,
                                                              const float* __restrict__ V, int64_t ldv,
                                                              const float* __restrict__ Wf,
                                                              const int32_t* __restrict__ sched, int n_sched, int P,
                                                              int F, int D, int B, const float* __restrict__ gout,
                                                              int64_t ldg, float* __restrict__ gE,
                                                              float* __restrict__ gV, int nsplit) {
  extern __shared__ __align__(16) float smem[];
  const int RS = row_stride(F, D), W = F * D;
  float* xs0 = smem;
  float* xs1 = xs0 + kSB * RS;
  float* tb = xs1 + kSB * RS;   // [4 waves][16][17] layout-change scratch (wave-private)
  int32_t* own = reinterpret_cast<int32_t*>(tb + 4 * 16 * 17);   // [F][F - 1][4]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kSB;
  const int npass = V ? 2 : 1;
  const int npar = F - 1;
  for (int e = tid; e < n_sched; e += kT) {
    const i32x4 v = *(const DCTR_GLOBAL i32x4*)(sched + 4 * e);
    if (v.x >= 0) {     // pair (i = v.x < j = v.y): in i's row it sits at slot j - 1, in j's row at slot i
      *reinterpret_cast<i32x4*>(own + 4 * (v.x * npar + v.y - 1)) = i32x4{v.y, 1, v.z, v.w};
      *reinterpret_cast<i32x4*>(own + 4 * (v.y * npar + v.x)) = i32x4{v.x, 0, v.z, v.w};
    }
  }
  stage_rows(xs0, RS, V ? V : E, V ? ldv : lde, b0, B, W);
  if (V) stage_rows(xs1, RS, E, lde, b0, B, W);
  __syncthreads();
  float* mytb = tb + wv * (16 * 17);
  const int owner = static_cast<int>(blockIdx.y) * 4 + wv, stride = 4 * nsplit;
  const int nf = owner < F ? (F - 1 - owner) / stride + 1 : 0;   // fields owner, owner + stride, ...
  const int nvis = nf * npar;
  if (nvis == 0) return;
  const bool cv = c < D;
  const int ccl = cv ? c : 0;
  int dcl[4];
  bool dv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    dv[s] = 4 * g + s < D;
    dcl[s] = dv[s] ? 4 * g + s : 0;
  }
  int64_t grow[4];                         // clamped gout row offsets of this lane's 4 samples
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * g + r;
    grow[r] = static_cast<int64_t>(b < B ? b : B - 1) * ldg + ccl;
  }
  struct Vis { int f, p, side, wi, k; };
  auto visit = [&](int v) -> Vis {
    const int vc = v < nvis ? v : nvis - 1;
    const int fi = vc / npar, t = vc - fi * npar;
    Vis x;
    x.f = owner + fi * stride;
    const i32x4 e = *reinterpret_cast<const i32x4*>(own + 4 * (x.f * npar + t));
    x.p = e.x; x.side = e.y; x.wi = e.z; x.k = e.w;
    return x;
  };
  float wr[KPD][4], gpr[KPD][2][4];
  auto issue = [&](const Vis& x, float (&w)[4], float (&gp)[2][4]) {
    // the pair's weight tile as THIS visit's B operand: W[e = c][d = 4g + s] (f is j) or W[e = 4g + s][d = c] (f is i)
    const float* base = Wf + static_cast<int64_t>(x.wi) * D * D;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int d = 4 * g + s;
      const int idx = x.side ? d * D + c : c * D + d;
      w[s] = ldg_f32(base + ((c < D && d < D) ? idx : 0));
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        gp[ps][r] = ldg_f32(gout + grow[r] + (static_cast<int64_t>(ps < npass ? ps : 0) * P + x.k) * D);
  };
#pragma unroll
  for (int u = 0; u < KPD; ++u) {
    issue(visit(u), wr[u], gpr[u]);
    __builtin_amdgcn_sched_barrier(0);
  }
  float acc[2][4];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[ps][r] = 0.f;
  // one visit: no memory operation behind a branch (the two sides differ in LDS / MFMA work only), so the waits for the
  // ring stay exact
  auto body = [&](int v, float (&wslot)[4], float (&gslot)[2][4]) {
    const Vis x = visit(v);
    float wreg[4], gpc[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wreg[s] = (cv && dv[s]) ? wslot[s] : 0.f;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int r = 0; r < 4; ++r) gpc[ps][r] = (cv && b0 + 4 * g + r < B) ? gslot[ps][r] : 0.f;
    issue(visit(v + KPD), wslot, gslot);
    // every LDS operand read unconditionally before the first MFMA: the partner as A operand (f is j) and in the
    // accumulator layout (f is i)
    float xa[2][4], xp[2][4];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const float* xs = (ps && npass > 1) ? xs1 : xs0;
#pragma unroll
      for (int s = 0; s < 4; ++s) xa[ps][s] = xs[c * RS + x.p * D + dcl[s]];
#pragma unroll
      for (int r = 0; r < 4; ++r) xp[ps][r] = xs[(4 * g + r) * RS + x.p * D + ccl];
    }
    if (x.side == 0) {          // f = j: acc[b][e] += gp[b][e] * (x_i W^T)[b][e]
      f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        t0 = mfma16(dv[s] ? xa[0][s] : 0.f, wreg[s], t0);
        t1 = mfma16(dv[s] ? xa[1][s] : 0.f, wreg[s], t1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[0][r] += gpc[0][r] * t0[r];
        acc[1][r] += gpc[1][r] * t1[r];
      }
    } else {                    // f = i: acc[b][d] += sum_e (gp (.) x_j)[b][e] W[e][d]
      for (int ps = 0; ps < npass; ++ps) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mytb[(4 * g + r) * 17 + c] = cv ? gpc[ps][r] * xp[ps][r] : 0.f;   // C layout
        float ga[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) ga[s] = mytb[c * 17 + 4 * g + s];                             // A layout
        f32x4 uu = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) uu = mfma16(ga[s], wreg[s], uu);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ps][r] += uu[r];
      }
    }
  };
  auto flush = [&](int f) {     // the field is complete: out it goes
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = b0 + 4 * g + r;
      if (cv && b < B) {
        const int64_t o = static_cast<int64_t>(b) * W + f * D + c;
        if (V) {
          stg_f32(gV + o, acc[0][r]);
          stg_f32(gE + o, acc[1][r]);
        } else {
          stg_f32(gE + o, acc[0][r]);
        }
      }
      acc[0][r] = acc[1][r] = 0.f;
    }
  };
  // (host: KPD divides F - 1, so a field is a whole number of ring rounds and its stores sit BETWEEN the pipelined loops)
  const int groups = npar / KPD;
  for (int fi = 0; fi < nf; ++fi) {
    for (int gi = 0; gi < groups; ++gi) {
#pragma unroll
      for (int u = 0; u < KPD; ++u) {
        body((fi * groups + gi) * KPD + u, wr[u], gpr[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    flush(owner + fi * stride);
  }
}

// gradient w.r.t. the weights: gW_k[e][d] = sum_b (gp (.) x_j)[b][e] x_i[b][d], both passes.
// Workgroup (sample group sg, pair slice): partial[sg][k][e][d]; rows = e, columns = d, reduction = samples.
__global__ __launch_bounds__(kT) void k_bilinear_bwd_weight(const float* __restrict__ E, int64_t lde,
                                                            const float* __restrict__ V, int64_t ldv,
                                                            const int32_t* __restrict__ sched, int n_sched, int P,
                                                            int F, int D, int B, const float* __restrict__ gout,
                                                            int64_t ldg, int tiles_per_group,
                                                            float* __restrict__ part) {
  extern __shared__ __align__(16) float smem[];
  const int RS = row_stride(F, D), W = F * D;
  float* xs0 = smem;
  float* xs1 = xs0 + kSB * RS;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int sg = blockIdx.x;
  const int npass = V ? 2 : 1;
  // this wave's pairs: MAXQ CONSECUTIVE schedule entries (the schedule arrives in output order k: the 8 incoming-
  // gradient pieces of a sample are then 8 * D consecutive floats -- whole 128-byte lines; with entries dealt out
  // round-robin every line was fetched twice, by different workgroups at different times: 2x the HBM traffic of the
  // 170 MB slab); at most 8 live accumulators
  constexpr int MAXQ = 8;
  f32x4 acc[MAXQ];
  const int qstep = 4 * gridDim.y * MAXQ, q0 = (blockIdx.y * 4 + wv) * MAXQ;
  const int ccl = c < D ? c : 0;
  for (int base = 0; base < n_sched; base += qstep) {
    // this wave's (at most MAXQ) pairs of the sweep: schedule entries loaded ONCE, not once per tile
    PairEnt ent[MAXQ];
#pragma unroll
    for (int a = 0; a < MAXQ; ++a) {
      acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
      ent[a] = load_pair(sched, base + q0 + a, n_sched);
    }
    for (int tl = 0; tl < tiles_per_group; ++tl) {
      const int b0 = (sg * tiles_per_group + tl) * kSB;
      if (b0 >= B) break;                  // uniform: later tiles of the group are past B as well
      // the incoming gradients of every live pair of this tile, all in flight together (unconditional loads on
      // clamped addresses, masked below), issued BEFORE the tile is staged so that both share one round trip
      float gpr[MAXQ][2][4];
#pragma unroll
      for (int a = 0; a < MAXQ; ++a)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int b = b0 + 4 * g + r;
            // (an idle entry has k = 0 and a single-pass call re-reads pass 0: valid addresses, values unused)
            gpr[a][ps][r] = ldg_f32(gout + static_cast<int64_t>(b < B ? b : B - 1) * ldg +
                                    (static_cast<int64_t>(ps < npass ? ps : 0) * P + ent[a].k) * D + ccl);
          }
      const float* src0 = V ? V : E;
      const int64_t ld0 = V ? ldv : lde;
      if (rows_vec_ok(kSB, RS, src0, ld0, W, 8) && (!V || rows_vec_ok(kSB, RS, E, lde, W, 8))) {   // (uniform)
        // both row tiles leave with the gradients above: ONE memory round trip per tile
        f32x4 r0[8], r1[8];
        rows_load4<8>(src0, ld0, b0, B, kSB, W, r0);
        if (V) rows_load4<8>(E, lde, b0, B, kSB, W, r1);
        __syncthreads();                    // the previous tile's LDS reads are done
        rows_store4<8>(xs0, RS, b0, B, kSB, W, r0);
        if (V) rows_store4<8>(xs1, RS, b0, B, kSB, W, r1);
      } else {
        __syncthreads();
        stage_rows(xs0, RS, src0, ld0, b0, B, W);
        if (V) stage_rows(xs1, RS, E, lde, b0, B, W);
      }
      __syncthreads();
      // branch-free: every LDS operand is read unconditionally (clamped column / field 0 for an idle entry) and masked
      // by a select -- `c < D ? xs[..] : 0` compiled to a branch around each ds_read with its own lgkmcnt(0): 128
      // serial LDS round trips per tile
#pragma unroll
      for (int a = 0; a < MAXQ; ++a) {
        const bool live = ent[a].i >= 0;
        const int i = live ? ent[a].i : 0, j = live ? ent[a].j : 0;
        float xjv[2][4], xiv[2][4];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const float* xs = (ps && npass > 1) ? xs1 : xs0;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            xjv[ps][s] = xs[(4 * g + s) * RS + j * D + ccl];
            xiv[ps][s] = xs[(4 * g + s) * RS + i * D + ccl];
          }
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const bool on = live && c < D && ps < npass;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int b = 4 * g + s;  // reduction index = sample
            const float gp = (on && b0 + b < B) ? gpr[a][ps][s] : 0.f;
            acc[a] = mfma16(gp * xjv[ps][s], on ? xiv[ps][s] : 0.f, acc[a]);   // A: g_t[b][e = c], B: x_i[b][d = c]
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < MAXQ; ++a) {
      if (ent[a].i < 0) continue;
      float* dst = part + (static_cast<int64_t>(sg) * P + ent[a].k) * D * D;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * g + r;  // row = e, column = d = c
        if (e < D && c < D) stg_f32(dst + e * D + c, acc[a][r]);
      }
    }
  }
}

// gW[w][e][d] = sum over the pairs k with weight index w, over sample groups:  fixed order
__global__ __launch_bounds__(kT) void k_bilinear_reduce_w(const float* __restrict__ part, int groups, int P, int DD,
                                                          const int32_t* __restrict__ pair_w, int n_w,
                                                          float* __restrict__ gW) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  const int w = static_cast<int>(idx / DD), el = static_cast<int>(idx - static_cast<int64_t>(w) * DD);
  const bool valid = idx < static_cast<int64_t>(n_w) * DD;
  float s = 0.f;
  auto add_pair = [&](int k) {
    // (8 group partials per round trip, added in group order: the serial `s += load` was one trip per group)
    for (int g0 = 0; g0 < groups; g0 += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        v[q] = ldg_f32(part + (static_cast<int64_t>(g0 + q < groups ? g0 + q : groups - 1) * P + k) * DD + el);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (g0 + q < groups) s += v[q];
    }
  };
  if ((DD & 63) == 0) {
    // a wave shares its weight index: 64 table entries per load, the matching pairs as a ballot mask walked in ascending
    // order (one dependent load + branch per pair and thread was 325 serial L2 round trips: 26 of this kernel's 31 us)
    const int lane = threadIdx.x & 63;
    const int wq = valid ? w : -2;
    for (int c0 = 0; c0 < P; c0 += 64) {
      const int pv = ldg_i32(pair_w + (c0 + lane < P ? c0 + lane : P - 1));
      unsigned long long m = __ballot(c0 + lane < P && pv == __builtin_amdgcn_readfirstlane(wq));
      while (m) {
        const int k = c0 + __builtin_ctzll(m);
        m &= m - 1;
        if (valid) add_pair(k);
      }
    }
  } else {
    for (int k = 0; k < P; ++k)
      if (valid && ldg_i32(pair_w + k) == w) add_pair(k);
  }
  if (!valid) return;
  gW[idx] = s;
}

// ------------------------------------------------------------------------------------------------------------
// InnerProduct:  p[b, k] = sum_d e_i e_j  (reduce) or p[b, k, :] = e_i (.) e_j;  pair k = (i, j), i < j, i outer
// ------------------------------------------------------------------------------------------------------------
template <int kSB>
__global__ __launch_bounds__(kT) void k_inner_fwd(const float* __restrict__ E, int64_t lde, int B, int F, int D,
                                                  int reduce, float* __restrict__ out, int64_t ldo) {
  extern __shared__ __align__(16) float smem[];
  const int W = F * D, P = F * (F - 1) / 2;
  float* es = smem;  // [kSB][W + 1]
  const int b0 = blockIdx.x * kSB;
  for (int e = threadIdx.x; e < kSB * W; e += kT) {
    const int r = e / W, c = e - r * W;
    es[r * (W + 1) + c] = (b0 + r < B) ? ldg_f32(E + static_cast<int64_t>(b0 + r) * lde + c) : 0.f;
  }
  __syncthreads();
  const int per = reduce ? 1 : D;
  for (int e = threadIdx.x; e < kSB * P * per; e += kT) {
    const int r = e / (P * per), rem = e - r * (P * per), k = rem / per, d0 = rem - k * per;
    // pair index -> (i, j)
    int i = 0, kk = k;
    while (kk >= F - 1 - i) {
      kk -= F - 1 - i;
      ++i;
    }
    const int j = i + 1 + kk;
    const float* xi = es + r * (W + 1) + i * D;
    const float* xj = es + r * (W + 1) + j * D;
    float s;
    if (reduce) {
      s = 0.f;
      for (int d = 0; d < D; ++d) s += xi[d] * xj[d];
    } else {
      s = xi[d0] * xj[d0];
    }
    if (b0 + r < B) stg_f32(out + static_cast<int64_t>(b0 + r) * ldo + rem, s);
  }
}

// gE[b, f, d] = sum_{g != f} gp[b, pair(f, g)] (* per-d if not reduced) * E[b, g, d]
template <int kSB>
__global__ __launch_bounds__(kT) void k_inner_bwd(const float* __restrict__ E, int64_t lde, int B, int F, int D,
                                                  int reduce, const float* __restrict__ gp, int64_t ldg,
                                                  float* __restrict__ gE, int64_t ldge) {
  extern __shared__ __align__(16) float smem[];
  const int W = F * D, P = F * (F - 1) / 2, per = reduce ? 1 : D;
  float* es = smem;                    // [kSB][W + 1]
  float* gs = es + kSB * (W + 1);      // [kSB][P*per + 1]
  const int GS = P * per + 1;
  const int b0 = blockIdx.x * kSB;
  for (int e = threadIdx.x; e < kSB * W; e += kT) {
    const int r = e / W, c = e - r * W;
    es[r * (W + 1) + c] = (b0 + r < B) ? ldg_f32(E + static_cast<int64_t>(b0 + r) * lde + c) : 0.f;
  }
  for (int e = threadIdx.x; e < kSB * P * per; e += kT) {
    const int r = e / (P * per), c = e - r * (P * per);
    gs[r * GS + c] = (b0 + r < B) ? ldg_f32(gp + static_cast<int64_t>(b0 + r) * ldg + c) : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kSB * W; e += kT) {
    const int r = e / W, c = e - r * W, f = c / D, d = c - f * D;
    float s = 0.f;
    for (int o = 0; o < F; ++o) {
      if (o == f) continue;
      const int i = o < f ? o : f, j = o < f ? f : o;
      const int k = i * F - i * (i + 1) / 2 + (j - i - 1);
      s += gs[r * GS + k * per + (reduce ? 0 : d)] * es[r * (W + 1) + o * D + d];
    }
    if (b0 + r < B) stg_f32(gE + static_cast<int64_t>(b0 + r) * ldge + c, s);
  }
}

size_t tile_bytes(int F, int D, int tiles) {
  int rs = F * D;
  rs += (16 - (rs & 31)) & 31;
  return static_cast<size_t>(tiles) * kSB * rs * sizeof(float);
}

}  // namespace

extern "C" int dctr_senet_fwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, const float* W1,
                              const float* W2, int32_t R, float* V, float* a, float* a1, dctr_stream_t stream) {
  if (!E || !W1 || !W2 || !V || !a || !a1 || B < 0 || F <= 0 || D <= 0 || R <= 0 || ld_e < static_cast<int64_t>(F) * D)
    return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const size_t lds = (static_cast<size_t>(kSS) * F * D + 2u * kSS * F + static_cast<size_t>(kSS) * R + 2u * R * F) *
                     sizeof(float);
  if (lds > 64 * 1024) return DCTR_ENOSUP;
  k_senet_fwd<<<dim3((B + kSS - 1) / kSS), dim3(kT), lds, static_cast<hipStream_t>(stream)>>>(E, ld_e, B, F, D, W1, W2,
                                                                                            R, V, a, a1);
  return launch_status();
}

extern "C" size_t dctr_senet_bwd_workspace_floats(int32_t B, int32_t F, int32_t R) {
  return static_cast<size_t>((B + kSS - 1) / kSS) * 2u * R * F;
}

extern "C" int dctr_senet_bwd(const float* gV, const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D,
                              const float* W1, const float* W2, int32_t R, const float* a, const float* a1,
                              float* gE, float* gW1, float* gW2, float* workspace, dctr_stream_t stream) {
  if (!gV || !E || !W1 || !W2 || !a || !a1 || !gE || !gW1 || !gW2 || !workspace || B < 0 || F <= 0 || D <= 0 || R <= 0)
    return DCTR_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) {
    (void)hipMemsetAsync(gW1, 0, sizeof(float) * R * F, s);
    (void)hipMemsetAsync(gW2, 0, sizeof(float) * R * F, s);
    return DCTR_OK;
  }
  const int groups = (B + kSS - 1) / kSS;
  const size_t lds = (2u * kSS * F * D + 3u * kSS * F + static_cast<size_t>(kSS) * R + 2u * R * F) * sizeof(float);
  if (lds > 64 * 1024) return DCTR_ENOSUP;
  k_senet_bwd<<<dim3(groups), dim3(kT), lds, s>>>(gV, E, ld_e, B, F, D, W1, W2, R, a, a1, gE, workspace);
  // workspace[g] = [gW1 (R*F) | gW2 (F*R)]
  const int64_t n = 2LL * R * F, half = static_cast<int64_t>(R) * F;
  const dim3 rg(static_cast<unsigned>((n + 15) / 16));
  k_reduce_partials<<<rg, dim3(kT), 0, s>>>(workspace, n, n, groups, gW1, half, gW2);     // (one launch for both)
  return launch_status();
}

extern "C" int dctr_bilinear_fwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                                 const int32_t* sched, int32_t n_sched, int32_t P, int32_t F, int32_t D, int32_t B,
                                 float* out, int64_t ld_o, const float* dense, int64_t ld_d, int32_t n_dense,
                                 int32_t dense_off, dctr_stream_t stream) {
  if (!E || !Wf || !sched || !out || B < 0 || F < 2 || D <= 0 || P <= 0 || n_sched <= 0) return DCTR_EINVAL;
  if (D > 16) return DCTR_ENOSUP;
  if (B == 0) return DCTR_OK;
  const size_t lds = tile_bytes(F, D, 2) + static_cast<size_t>(n_sched) * 16;   // + the schedule
  if (lds > 150 * 1024) return DCTR_ENOSUP;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bilinear_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
  k_bilinear_fwd<<<dim3((B + kSB - 1) / kSB), dim3(kT), lds, static_cast<hipStream_t>(stream)>>>(
      E, ld_e, V, ld_v, Wf, sched, n_sched, P, F, D, B, out, ld_o, n_dense > 0 ? dense : nullptr, ld_d, n_dense,
      dense_off);
  return launch_status();
}

static int bilinear_groups(int B) {
  const int tiles = (B + kSB - 1) / kSB;
  return tiles < 32 ? tiles : 32;
}

extern "C" size_t dctr_bilinear_bwd_workspace_floats(int32_t B, int32_t P, int32_t D) {
  return static_cast<size_t>(bilinear_groups(B > 0 ? B : 1)) * P * D * D;
}

extern "C" int dctr_bilinear_bwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                                 const int32_t* sched, int32_t n_sched, int32_t slots, const int32_t* pair_w,
                                 int32_t n_w, int32_t P, int32_t F, int32_t D, int32_t B, const float* gout,
                                 int64_t ld_g, float* gE, float* gV, float* gW, float* workspace,
                                 const int32_t* sched_k, int32_t n_sched_k, dctr_stream_t stream) {
  if (!E || !Wf || !sched || !pair_w || !gout || !gE || !gW || !workspace || (V && !gV) || B < 0 || F < 2 || D <= 0 ||
      P <= 0 || n_sched <= 0 || slots <= 0 || n_w <= 0)
    return DCTR_EINVAL;
  if (D > 16) return DCTR_ENOSUP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) {
    (void)hipMemsetAsync(gW, 0, sizeof(float) * n_w * D * D, s);
    return DCTR_OK;
  }
  const size_t lds_own = tile_bytes(F, D, 2) + 4u * 16 * 17 * sizeof(float) + static_cast<size_t>(F) * (F - 1) * 16;
  bool by_owner = lds_own <= 158 * 1024;
#ifdef DCTR_DIAG
  if (const char* e = getenv("DCTR_BILINEAR_BWD")) by_owner = by_owner && e[0] != 't';   // "tournament": the round 1-3 kernel
#endif
  if (by_owner) {
    const int nsplit = (2 * lds_own <= 158 * 1024 && F > 4) ? 2 : 1;
    const dim3 grid((B + kSB - 1) / kSB, nsplit);
    const int npar = F - 1;      // the ring depth divides the partners per field (25 at the Criteo shape: 5)
#define DCTR_OWN(K)                                                                                               \
    do {                                                                                                          \
      if (lds_own > 64 * 1024)                                                                                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bilinear_bwd_data_own<K>),                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_own));         \
      k_bilinear_bwd_data_own<K><<<grid, dim3(kT), lds_own, s>>>(E, ld_e, V, ld_v, Wf, sched, n_sched, P, F, D, B, \
                                                                 gout, ld_g, gE, gV, nsplit);                     \
    } while (0)
    if (npar % 5 == 0) DCTR_OWN(5);
    else if (npar % 4 == 0) DCTR_OWN(4);
    else if (npar % 3 == 0) DCTR_OWN(3);
    else if (npar % 2 == 0) DCTR_OWN(2);
    else DCTR_OWN(1);
#undef DCTR_OWN
  } else {
    size_t lds = tile_bytes(F, D, 4) + 4u * 16 * 17 * sizeof(float);
    if (lds > 158 * 1024) return DCTR_ENOSUP;
    const int sch_lds = lds + static_cast<size_t>(n_sched) * 16 <= 158 * 1024;
    if (sch_lds) lds += static_cast<size_t>(n_sched) * 16;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bilinear_bwd_data),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    k_bilinear_bwd_data<<<dim3((B + kSB - 1) / kSB), dim3(kT), lds, s>>>(E, ld_e, V, ld_v, Wf, sched, n_sched, slots, P,
                                                                        F, D, B, gout, ld_g, gE, gV, sch_lds);
  }
  {
    const int tiles = (B + kSB - 1) / kSB;
    const int32_t* sw = (sched_k && n_sched_k > 0) ? sched_k : sched;
    const int nw_s = (sched_k && n_sched_k > 0) ? n_sched_k : n_sched;
    int py = (nw_s + 31) / 32;      // 4 waves x 8 consecutive pairs per workgroup row
    if (py > 16) py = 16;
    // The kernel holds one workgroup per CU (its registers): keep the launch to ONE round of the 256 CUs -- 32 x 11
    // workgroups at the Criteo shape were a full round plus a 96-workgroup tail, i.e. 16 tile times for the work of 11
    int groups = bilinear_groups(B);
    if (groups * py > 256) groups = 256 / py > 0 ? 256 / py : 1;
    const int tpg = (tiles + groups - 1) / groups;
    const size_t lds = tile_bytes(F, D, 2);
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bilinear_bwd_weight),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    // groups whose tiles all fall past B still own a slab: clear the workspace first so they contribute zeros (every other
    // slab is written in full by its group: no clear -- a 5.6 us node -- when the last group has a tile inside the batch)
    if (static_cast<int64_t>(groups - 1) * tpg * kSB >= B)
      (void)hipMemsetAsync(workspace, 0, sizeof(float) * static_cast<size_t>(groups) * P * D * D, s);
    k_bilinear_bwd_weight<<<dim3(groups, py), dim3(kT), lds, s>>>(E, ld_e, V, ld_v, sw, nw_s, P, F, D, B, gout,
                                                                 ld_g, tpg, workspace);
    const int64_t total = static_cast<int64_t>(n_w) * D * D;
    k_bilinear_reduce_w<<<dim3(static_cast<unsigned>((total + kT - 1) / kT)), dim3(kT), 0, s>>>(workspace, groups, P,
                                                                                              D * D, pair_w, n_w, gW);
  }
  return launch_status();
}

extern "C" int dctr_inner_product_fwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t reduce,
                                      float* out, int64_t ld_o, dctr_stream_t stream) {
  if (!E || !out || B < 0 || F < 2 || D <= 0) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const size_t row = static_cast<size_t>(F * D + 1) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (16 * row <= 60 * 1024) {
    k_inner_fwd<16><<<dim3((B + 15) / 16), dim3(kT), 16 * row, s>>>(E, ld_e, B, F, D, reduce, out, ld_o);
  } else if (row <= 60 * 1024) {
    k_inner_fwd<1><<<dim3(B), dim3(kT), row, s>>>(E, ld_e, B, F, D, reduce, out, ld_o);
  } else {
    return DCTR_ENOSUP;
  }
  return launch_status();
}

extern "C" int dctr_inner_product_bwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t reduce,
                                      const float* gp, int64_t ld_g, float* gE, int64_t ld_ge, dctr_stream_t stream) {
  if (!E || !gp || !gE || B < 0 || F < 2 || D <= 0) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const int P = F * (F - 1) / 2, per = reduce ? 1 : D;
  const size_t row = static_cast<size_t>(F * D + 1 + P * per + 1) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (16 * row <= 60 * 1024) {
    k_inner_bwd<16><<<dim3((B + 15) / 16), dim3(kT), 16 * row, s>>>(E, ld_e, B, F, D, reduce, gp, ld_g, gE, ld_ge);
  } else if (row <= 60 * 1024) {
    k_inner_bwd<1><<<dim3(B), dim3(kT), row, s>>>(E, ld_e, B, F, D, reduce, gp, ld_g, gE, ld_ge);
  } else {
    return DCTR_ENOSUP;
  }
  return launch_status();
}

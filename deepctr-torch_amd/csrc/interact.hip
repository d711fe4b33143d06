// interact.hip -- InteractingLayer of AutoInt (reference interaction.py:328-394): multi-head self-attention over the
// F field embeddings of one sample, with a residual projection and relu.
//
//   Q = E Wq, K = E Wk, V = E Wv   ([F, D] x [D, D]);   head n = columns [n*A, (n+1)*A), A = D / H
//   S_n = Q_n K_n^T (/ sqrt(A) if scaling);  P_n = softmax_rows(S_n);  O_n = P_n V_n
//   out = relu([O_0 | ... | O_{H-1}] (+ E Wr))
// The reference runs 4 tensordots, 3 stack/split, an einsum, a softmax, a matmul, a cat/split/squeeze and a relu per
// layer: ~20 launches and a dozen [B, F, D] / [H, B, F, F] round trips through HBM.  Here ONE wave owns a sample: E, Q,
// K, V, the H score matrices and the pre-activation live in LDS (6 KB + 5.4 KB at the Criteo shape), only E and the
// output touch HBM.  The backward recomputes the forward (no [H, B, F, F] tensor is ever saved), has no atomics, and
// sums the four weight gradients in a fixed order (per-lane partial sums over the workgroup's samples -> one partial
// row per workgroup -> k_interact_reduce in workgroup order).  Limits: D <= 32, F <= 64, H*F*F floats within LDS.
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kW = 64;

struct IntArgs {
  const float* E;
  int64_t lde;
  const float* Wq;
  const float* Wk;
  const float* Wv;
  const float* Wr;     // nullable (use_res = False)
  int B, F, D, H;
  float scale;         // 1 or 1 / sqrt(A)
  float* out;          // fwd [B, ldo]
  int64_t ldo;
  const float* g;      // bwd: d loss / d out [B, ldg]
  int64_t ldg;
  float* gE;           // bwd [B, ldge]
  int64_t ldge;
  float* part;         // bwd [n_wg][4 * D * D]
};

// LDS (floats): W [4][D*D] | es [F*D] | q | k | v | pre | sc [H*F*F] | (bwd) gq | gk | gv | gpre | gsc [H*F*F]
template <bool BWD>
__global__ __launch_bounds__(kW) void k_interact(IntArgs a) {
  extern __shared__ __align__(16) float smem[];
  const int F = a.F, D = a.D, H = a.H, A = D / H, lane = threadIdx.x;
  const int FD = F * D, DD = D * D, SS = H * F * F;
  float* Wq = smem;
  float* Wk = Wq + DD;
  float* Wv = Wk + DD;
  float* Wr = Wv + DD;
  float* es = Wr + DD;
  float* q = es + FD;
  float* k = q + FD;
  float* v = k + FD;
  float* pre = v + FD;
  float* sc = pre + FD;
  float* gq = sc + SS;
  float* gk = gq + (BWD ? FD : 0);
  float* gv = gk + (BWD ? FD : 0);
  float* gpre = gv + (BWD ? FD : 0);
  float* gsc = gpre + (BWD ? FD : 0);

  for (int e = lane; e < DD; e += kW) {
    Wq[e] = ldg_f32(a.Wq + e);
    Wk[e] = ldg_f32(a.Wk + e);
    Wv[e] = ldg_f32(a.Wv + e);
    Wr[e] = a.Wr ? ldg_f32(a.Wr + e) : 0.f;
  }
  // weight-gradient partials: lane owns elements idx = lane + 64 * n of every [D, D] matrix (D <= 32: 16 each)
  constexpr int NW = 16;
  float gWq[NW], gWk[NW], gWv[NW], gWr[NW];
#pragma unroll
  for (int n = 0; n < NW; ++n) gWq[n] = gWk[n] = gWv[n] = gWr[n] = 0.f;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    __syncthreads();
    {
      const float* src = a.E + static_cast<int64_t>(b) * a.lde;
      for (int e0 = lane; e0 < FD; e0 += 8 * kW) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ldg_f32(src + (e0 + u * kW < FD ? e0 + u * kW : 0));
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (e0 + u * kW < FD) es[e0 + u * kW] = t[u];
      }
    }
    __syncthreads();
    // projections: element (f, e) of Q, K, V and the residual
    for (int idx = lane; idx < FD; idx += kW) {
      const int f = idx / D, e = idx - f * D;
      float sq = 0.f, sk = 0.f, sv = 0.f, sr = 0.f;
      for (int d = 0; d < D; ++d) {
        const float x = es[f * D + d];
        sq += x * Wq[d * D + e];
        sk += x * Wk[d * D + e];
        sv += x * Wv[d * D + e];
        sr += x * Wr[d * D + e];
      }
      q[idx] = sq;
      k[idx] = sk;
      v[idx] = sv;
      pre[idx] = sr;
    }
    __syncthreads();
    // scores S[h, i, j] = Q_h[i] . K_h[j]
    for (int idx = lane; idx < SS; idx += kW) {
      const int h = idx / (F * F), r = idx - h * F * F, i = r / F, j = r - i * F;
      float s = 0.f;
      for (int t = 0; t < A; ++t) s += q[i * D + h * A + t] * k[j * D + h * A + t];
      sc[idx] = s * a.scale;
    }
    __syncthreads();
    // softmax over j of every row (h, i)
    for (int row = lane; row < H * F; row += kW) {
      float* p = sc + row * F;
      float mx = p[0];
      for (int j = 1; j < F; ++j) mx = fmaxf(mx, p[j]);
      float z = 0.f;
      for (int j = 0; j < F; ++j) {
        const float ex = expf(p[j] - mx);
        p[j] = ex;
        z += ex;
      }
      const float rz = 1.f / z;
      for (int j = 0; j < F; ++j) p[j] *= rz;
    }
    __syncthreads();
    // pre[i, e] += sum_j P[h(e), i, j] V[j, e]
    for (int idx = lane; idx < FD; idx += kW) {
      const int i = idx / D, e = idx - i * D, h = e / A;
      const float* p = sc + (h * F + i) * F;
      float s = pre[idx];
      for (int j = 0; j < F; ++j) s += p[j] * v[j * D + e];
      pre[idx] = s;
    }
    __syncthreads();
    if (!BWD) {
      float* dst = a.out + static_cast<int64_t>(b) * a.ldo;
      for (int idx = lane; idx < FD; idx += kW) stg_f32(dst + idx, fmaxf(pre[idx], 0.f));
      continue;
    }
    // ---- backward ------------------------------------------------------------------------------------------------
    {
      const float* gs = a.g + static_cast<int64_t>(b) * a.ldg;
      for (int idx = lane; idx < FD; idx += kW) gpre[idx] = pre[idx] > 0.f ? ldg_f32(gs + idx) : 0.f;
    }
    __syncthreads();
    // gS = P (.) (gP - rowsum(P (.) gP)) * scale, gP[h, i, j] = gpre[i, h-block] . V[j, h-block]
    for (int row = lane; row < H * F; row += kW) {
      const int h = row / F, i = row - h * F;
      const float* p = sc + row * F;
      float* gs = gsc + row * F;
      float dot = 0.f;
      for (int j = 0; j < F; ++j) {
        float gp = 0.f;
        for (int t = 0; t < A; ++t) gp += gpre[i * D + h * A + t] * v[j * D + h * A + t];
        gs[j] = gp;
        dot += p[j] * gp;
      }
      for (int j = 0; j < F; ++j) gs[j] = p[j] * (gs[j] - dot) * a.scale;
    }
    __syncthreads();
    // gV[j, e] = sum_i P[h, i, j] gpre[i, e];  gQ[i, e] = sum_j gS[h, i, j] K[j, e];  gK[j, e] = sum_i gS[h, i, j] Q[i, e]
    for (int idx = lane; idx < FD; idx += kW) {
      const int r = idx / D, e = idx - r * D, h = e / A;
      float sv = 0.f, sq = 0.f, sk = 0.f;
      for (int o = 0; o < F; ++o) {
        sv += sc[(h * F + o) * F + r] * gpre[o * D + e];
        sq += gsc[(h * F + r) * F + o] * k[o * D + e];
        sk += gsc[(h * F + o) * F + r] * q[o * D + e];
      }
      gv[idx] = sv;
      gq[idx] = sq;
      gk[idx] = sk;
    }
    __syncthreads();
    // gE[f, d] = sum_e gQ[f,e] Wq[d,e] + gK[f,e] Wk[d,e] + gV[f,e] Wv[d,e] + gpre[f,e] Wr[d,e]
    {
      float* dst = a.gE + static_cast<int64_t>(b) * a.ldge;
      for (int idx = lane; idx < FD; idx += kW) {
        const int f = idx / D, d = idx - f * D;
        float s = 0.f;
        for (int e = 0; e < D; ++e)
          s += gq[f * D + e] * Wq[d * D + e] + gk[f * D + e] * Wk[d * D + e] + gv[f * D + e] * Wv[d * D + e] +
               gpre[f * D + e] * Wr[d * D + e];
        stg_f32(dst + idx, s);
      }
    }
    // gW*[d, e] += sum_f E[f, d] g*[f, e]
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const int idx = lane + kW * n;
      if (idx < DD) {
        const int d = idx / D, e = idx - d * D;
        float sq = 0.f, sk = 0.f, sv = 0.f, sr = 0.f;
        for (int f = 0; f < F; ++f) {
          const float x = es[f * D + d];
          sq += x * gq[f * D + e];
          sk += x * gk[f * D + e];
          sv += x * gv[f * D + e];
          sr += x * gpre[f * D + e];
        }
        gWq[n] += sq;
        gWk[n] += sk;
        gWv[n] += sv;
        gWr[n] += sr;
      }
    }
  }
  if (BWD) {
    float* mine = a.part + static_cast<int64_t>(blockIdx.x) * (4 * DD);
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const int idx = lane + kW * n;
      if (idx < DD) {
        mine[idx] = gWq[n];
        mine[DD + idx] = gWk[n];
        mine[2 * DD + idx] = gWv[n];
        mine[3 * DD + idx] = gWr[n];
      }
    }
  }
}

// out_m[i] = sum_g part[g][m*DD + i] in workgroup order (thread (o, sl): groups sl, sl + 16, ...; slices added in order)
__global__ __launch_bounds__(256) void k_interact_reduce(const float* __restrict__ part, int64_t stride, int groups,
                                                         int DD, float* __restrict__ g0, float* __restrict__ g1,
                                                         float* __restrict__ g2, float* __restrict__ g3) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 16 + o;
  const int64_t ic = i < stride ? i : 0;
  float s = 0.f;
  for (int gg = sl; gg < groups; gg += 16 * 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = gg + 16 * u;
      t[u] = ldg_f32(part + static_cast<int64_t>(g < groups ? g : 0) * stride + ic);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (gg + 16 * u < groups) s += t[u];
  }
  red[sl][o] = s;
  __syncthreads();
  if (sl == 0 && i < stride) {
    float t = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) t += red[kk][o];
    const int m = static_cast<int>(i / DD), r = static_cast<int>(i - static_cast<int64_t>(m) * DD);
    float* dst = m == 0 ? g0 : (m == 1 ? g1 : (m == 2 ? g2 : g3));
    if (dst) dst[r] = t;
  }
}

size_t lds_bytes(int F, int D, int H, bool bwd) {
  size_t n = 4u * D * D + 5u * F * D + static_cast<size_t>(H) * F * F;
  if (bwd) n += 4u * F * D + static_cast<size_t>(H) * F * F;
  return n * sizeof(float);
}

// one wave per workgroup, ~13 resident per CU: 4096 workgroups are about one round of the chip; more samples than that
// are walked grid-stride (the backward's partial weight-gradient rows stay at 16 MB)
int groups_of(int B, bool) { return B < 4096 ? B : 4096; }

int check(const float* E, int64_t ld_e, int B, int F, int D, int H, const float* Wq, const float* Wk, const float* Wv) {
  if (!E || !Wq || !Wk || !Wv || B < 0 || F <= 0 || D <= 0 || H <= 0 || D % H != 0 ||
      ld_e < static_cast<int64_t>(F) * D)
    return DCTR_EINVAL;
  if (D > 32 || F > 64) return DCTR_ENOSUP;
  return DCTR_OK;
}

template <bool BWD>
int launch(const IntArgs& a, hipStream_t s) {
  const size_t lds = lds_bytes(a.F, a.D, a.H, BWD);
  if (lds > 150u * 1024u) return DCTR_ENOSUP;
  if (lds > 64u * 1024u)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_interact<BWD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  k_interact<BWD><<<dim3(groups_of(a.B, BWD)), dim3(kW), lds, s>>>(a);
  return launch_status();
}

}  // namespace

extern "C" size_t dctr_interacting_bwd_workspace_floats(int32_t B, int32_t D) {
  if (B <= 0 || D <= 0) return 0;
  return static_cast<size_t>(groups_of(B, true)) * 4u * D * D;
}

extern "C" int dctr_interacting_supported(int32_t F, int32_t D, int32_t H) {
  if (F <= 0 || D <= 0 || H <= 0 || D % H != 0 || D > 32 || F > 64) return 0;
  return lds_bytes(F, D, H, true) <= 150u * 1024u ? 1 : 0;
}

extern "C" int dctr_interacting_fwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t H,
                                    int32_t scaling, const float* Wq, const float* Wk, const float* Wv,
                                    const float* Wr, float* out, int64_t ld_o, dctr_stream_t stream) {
  const int rc = check(E, ld_e, B, F, D, H, Wq, Wk, Wv);
  if (rc != DCTR_OK) return rc;
  if (!out || ld_o < static_cast<int64_t>(F) * D) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  IntArgs a = {};
  a.E = E; a.lde = ld_e; a.Wq = Wq; a.Wk = Wk; a.Wv = Wv; a.Wr = Wr; a.B = B; a.F = F; a.D = D; a.H = H;
  a.scale = scaling ? 1.f / sqrtf(static_cast<float>(D / H)) : 1.f;
  a.out = out; a.ldo = ld_o;
  return launch<false>(a, static_cast<hipStream_t>(stream));
}

extern "C" int dctr_interacting_bwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t H,
                                    int32_t scaling, const float* Wq, const float* Wk, const float* Wv,
                                    const float* Wr, const float* gout, int64_t ld_g, float* gE, int64_t ld_ge,
                                    float* gWq, float* gWk, float* gWv, float* gWr, float* workspace,
                                    dctr_stream_t stream) {
  const int rc = check(E, ld_e, B, F, D, H, Wq, Wk, Wv);
  if (rc != DCTR_OK) return rc;
  if (!gout || !gE || !gWq || !gWk || !gWv || (Wr && !gWr) || ld_g < static_cast<int64_t>(F) * D ||
      ld_ge < static_cast<int64_t>(F) * D)
    return DCTR_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t dd = static_cast<size_t>(D) * D * sizeof(float);
  if (B == 0) {
    (void)hipMemsetAsync(gWq, 0, dd, s);
    (void)hipMemsetAsync(gWk, 0, dd, s);
    (void)hipMemsetAsync(gWv, 0, dd, s);
    if (gWr) (void)hipMemsetAsync(gWr, 0, dd, s);
    return DCTR_OK;
  }
  if (!workspace) return DCTR_EINVAL;
  IntArgs a = {};
  a.E = E; a.lde = ld_e; a.Wq = Wq; a.Wk = Wk; a.Wv = Wv; a.Wr = Wr; a.B = B; a.F = F; a.D = D; a.H = H;
  a.scale = scaling ? 1.f / sqrtf(static_cast<float>(D / H)) : 1.f;
  a.g = gout; a.ldg = ld_g; a.gE = gE; a.ldge = ld_ge; a.part = workspace;
  const int st = launch<true>(a, s);
  if (st != DCTR_OK) return st;
  const int64_t stride = 4LL * D * D;
  k_interact_reduce<<<dim3(static_cast<unsigned>((stride + 15) / 16)), dim3(256), 0, s>>>(
      workspace, stride, groups_of(B, true), D * D, gWq, gWk, gWv, Wr ? gWr : nullptr);
  return launch_status();
}

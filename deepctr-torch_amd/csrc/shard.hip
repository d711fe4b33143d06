// shard.hip -- the two "assemble" kernels of table-sharded multi-GPU training (deepctr_torch/parallel.py).
//
// Tables are sharded by table across the ranks of one node (rank q owns units q, q+N, q+2N, ...).  Per step
//   forward   every rank sends the id columns of its B samples to the owners (all-to-all #1), each owner gathers
//             the rows of ITS tables for all N*B samples with dctr_embed_fwd (csrc/embed.hip) and sends every
//             rank its B rows back (all-to-all #2);
//   backward  every rank sends the row gradients of its B samples to the owners (all-to-all #3 = the sparse
//             reduce-scatter) and each owner applies dctr_embed_update (csrc/update.hip) to its tables.
// What arrives from owner q is a chunk  R_q [B, ldc]  whose row b is
//     [ e(b, unit q) | e(b, unit q+N) | ... (slot j = unit q + j*N) | pad | wide partial sum of q's units | pad ]
// k_assemble_fwd turns the N chunks into exactly what dctr_embed_fwd produces on one GPU: the DNN-input row
// [ e_0 | ... | e_{F-1} | dense ] (inputs.py:126-138), the linear logit (basemodel.py:63-92), FM (interaction.py:
// 26-34) and the side output S = sum_f e.  k_assemble_bwd is its adjoint: it folds FM's backward into the row
// gradients, G[b, f] = g_out[b, f] + g_fm[b] * (S[b] - e[b, f]), and writes them in the chunk layout the owners'
// update kernel reads as one [N*B, ldc] matrix (wide column = g_wide[b]); an extra workgroup per dense column
// computes d loss / d Linear.weight like dctr_embed_update does on one GPU.
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kT = 256;

struct AsmArgs {
  const float* recv;     // forward: [N][B][ldc] chunks by owner;   backward: unused
  float* send;           // backward: [N][B][ldc] chunks by owner
  int64_t ldc;
  int32_t N, B, F, D;    // ranks, local batch, deep units (fields), embedding dim
  int32_t wide_col;      // column of the wide partial inside a chunk row, -1: no wide tables
  const float* X;        // [B, ldx] the rank's own input matrix (dense columns)
  int64_t ldx;
  const int32_t* dense_cols;   // [n_dense] X columns copied to out[:, dense_off ...)
  int32_t n_dense, dense_off;
  const int32_t* wdense_cols;  // [n_wdense] X columns of Linear's dense half
  const float* wdense_w;       // [n_wdense] Linear.weight
  int32_t n_wdense;
  float* out;            // [B, ldo]
  int64_t ldo;
  float* wide;           // [B] nullable
  float* fm;             // [B] nullable
  float* fm_s;           // [B, lds_] nullable
  int64_t lds_;
  // backward
  const float* g_out;    // [B, ldg] nullable
  int64_t ldg;
  const float* g_wide;   // [B] nullable
  const float* g_fm;     // [B] nullable
  float* g_wdense;       // [n_wdense] nullable
};

// One workgroup = 16 samples; thread (r = tid / 16, c = tid % 16) walks the row of sample r in 16-float steps.
__global__ __launch_bounds__(kT) void k_assemble_fwd(AsmArgs A) {
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 16 + r;
  const bool valid = b < A.B;
  const int64_t bb = valid ? b : 0;
  const int W = A.F * A.D;
  // deep slices: element e = f * D + d of the row comes from owner f % N, slot f / N
  for (int e = c; e < W; e += 16) {
    const int f = e / A.D, d = e - f * A.D;
    const int q = f % A.N, j = f / A.N;
    const float v = ldg_f32(A.recv + (static_cast<int64_t>(q) * A.B + bb) * A.ldc + j * A.D + d);
    if (valid) stg_f32(A.out + b * A.ldo + e, v);
  }
  for (int k = c; k < A.n_dense; k += 16)
    if (valid) stg_f32(A.out + b * A.ldo + A.dense_off + k, ldg_f32(A.X + b * A.ldx + ldg_i32(A.dense_cols + k)));
  // wide: sum of the owners' partial sums (owner order => deterministic) + dense . Linear.weight
  if (A.wide) {
    float w = 0.f;
    if (A.wide_col >= 0)
      for (int q = c; q < A.N; q += 16) w += ldg_f32(A.recv + (static_cast<int64_t>(q) * A.B + bb) * A.ldc + A.wide_col);
    for (int k = c; k < A.n_wdense; k += 16)
      w += ldg_f32(A.X + bb * A.ldx + ldg_i32(A.wdense_cols + k)) * ldg_f32(A.wdense_w + k);
    w = group_sum<16>(w);
    if (c == 0 && valid) stg_f32(A.wide + b, w);
  }
  // FM and S: lane c owns dimensions d = c, c + 16, ... (D <= 64)
  if (A.fm || A.fm_s) {
    float tot = 0.f;
    for (int d = c; d < A.D; d += 16) {
      float s = 0.f, sq = 0.f;
      for (int f = 0; f < A.F; ++f) {
        const int q = f % A.N, j = f / A.N;
        const float v = ldg_f32(A.recv + (static_cast<int64_t>(q) * A.B + bb) * A.ldc + j * A.D + d);
        s += v;
        sq += v * v;
      }
      if (A.fm_s && valid) stg_f32(A.fm_s + b * A.lds_ + d, s);
      tot += s * s - sq;
    }
    tot = group_sum<16>(tot);
    if (A.fm && c == 0 && valid) stg_f32(A.fm + b, 0.5f * tot);
  }
}

__global__ __launch_bounds__(kT) void k_assemble_bwd(AsmArgs A) {
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  const int nblk = (A.B + 15) / 16;
  if (static_cast<int>(blockIdx.x) >= nblk) {   // d loss / d Linear.weight, one workgroup per dense column
    __shared__ float red[kT / 64];
    const int j = static_cast<int>(blockIdx.x) - nblk;
    const int col = ldg_i32(A.wdense_cols + j);
    float acc = 0.f;
#pragma unroll 8
    for (int b = tid; b < A.B; b += kT) acc += ldg_f32(A.g_wide + b) * ldg_f32(A.X + static_cast<int64_t>(b) * A.ldx + col);
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < kT / 64; ++w) t += red[w];
      stg_f32(A.g_wdense + j, t);
    }
    return;
  }
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 16 + r;
  if (b >= A.B) return;
  const int W = A.F * A.D;
  const float gf = A.g_fm ? ldg_f32(A.g_fm + b) : 0.f;
  for (int e = c; e < W; e += 16) {
    const int f = e / A.D, d = e - f * A.D;
    const int q = f % A.N, j = f / A.N;
    float g = A.g_out ? ldg_f32(A.g_out + b * A.ldg + e) : 0.f;
    if (A.g_fm) g += gf * (ldg_f32(A.fm_s + b * A.lds_ + d) - ldg_f32(A.out + b * A.ldo + e));
    stg_f32(A.send + (static_cast<int64_t>(q) * A.B + b) * A.ldc + j * A.D + d, g);
  }
  if (A.wide_col >= 0) {
    const float gw = A.g_wide ? ldg_f32(A.g_wide + b) : 0.f;
    for (int q = c; q < A.N; q += 16) stg_f32(A.send + (static_cast<int64_t>(q) * A.B + b) * A.ldc + A.wide_col, gw);
  }
}

}  // namespace

extern "C" int dctr_shard_assemble_fwd(const float* recv, int64_t ld_chunk, int32_t n_ranks, int32_t B, int32_t F,
                                       int32_t D, int32_t wide_col, const float* X, int64_t ld_x,
                                       const int32_t* dense_cols, int32_t n_dense, int32_t dense_off,
                                       const int32_t* wdense_cols, const float* wdense_w, int32_t n_wdense,
                                       float* out, int64_t ld_out, float* wide, float* fm, float* fm_s,
                                       int64_t ld_s, dctr_stream_t stream) {
  if (!recv || !out || n_ranks <= 0 || B < 0 || F <= 0 || D <= 0 || D > 64) return DCTR_EINVAL;
  if ((n_dense > 0 || n_wdense > 0) && !X) return DCTR_EINVAL;
  if (n_dense > 0 && !dense_cols) return DCTR_EINVAL;
  if (n_wdense > 0 && (!wdense_cols || !wdense_w)) return DCTR_EINVAL;
  if ((fm || fm_s) && D > 64) return DCTR_ENOSUP;
  if (fm_s && ld_s < D) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  AsmArgs a = {};
  a.recv = recv; a.ldc = ld_chunk; a.N = n_ranks; a.B = B; a.F = F; a.D = D; a.wide_col = wide_col;
  a.X = X; a.ldx = ld_x; a.dense_cols = dense_cols; a.n_dense = n_dense; a.dense_off = dense_off;
  a.wdense_cols = wdense_cols; a.wdense_w = wdense_w; a.n_wdense = n_wdense;
  a.out = out; a.ldo = ld_out; a.wide = wide; a.fm = fm; a.fm_s = fm_s; a.lds_ = ld_s;
  k_assemble_fwd<<<dim3((B + 15) / 16), dim3(kT), 0, static_cast<hipStream_t>(stream)>>>(a);
  return launch_status();
}

extern "C" int dctr_shard_assemble_bwd(float* send, int64_t ld_chunk, int32_t n_ranks, int32_t B, int32_t F, int32_t D,
                                       int32_t wide_col, const float* g_out, int64_t ld_g, const float* g_wide,
                                       const float* g_fm, const float* out, int64_t ld_out, const float* fm_s,
                                       int64_t ld_s, const float* X, int64_t ld_x, const int32_t* wdense_cols,
                                       int32_t n_wdense, float* g_wdense, dctr_stream_t stream) {
  if (!send || n_ranks <= 0 || B < 0 || F <= 0 || D <= 0) return DCTR_EINVAL;
  if (g_fm && (!out || !fm_s)) return DCTR_EINVAL;
  if (g_wdense && (!X || !g_wide || !wdense_cols || n_wdense <= 0)) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  AsmArgs a = {};
  a.send = send; a.ldc = ld_chunk; a.N = n_ranks; a.B = B; a.F = F; a.D = D; a.wide_col = wide_col;
  a.g_out = g_out; a.ldg = ld_g; a.g_wide = g_wide; a.g_fm = g_fm; a.out = const_cast<float*>(out); a.ldo = ld_out;
  a.fm_s = const_cast<float*>(fm_s); a.lds_ = ld_s; a.X = X; a.ldx = ld_x; a.wdense_cols = wdense_cols;
  a.n_wdense = n_wdense; a.g_wdense = g_wdense;
  const unsigned nblk = (B + 15) / 16 + (g_wdense ? static_cast<unsigned>(n_wdense) : 0u);
  k_assemble_bwd<<<dim3(nblk), dim3(kT), 0, static_cast<hipStream_t>(stream)>>>(a);
  return launch_status();
}

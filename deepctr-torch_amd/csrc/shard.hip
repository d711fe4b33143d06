// shard.hip -- the two "assemble" kernels of table-sharded multi-GPU training (deepctr_torch/parallel.py).
//
// Tables are sharded by table across the ranks of one node (an owner / slot map per unit: deepctr_torch/parallel.py
// ShardLayout.assign balances unit counts and table bytes; without a map rank q owns units q, q+N, q+2N, ...).  Per step
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

// row b of the chunk for owner q: inside the local [N][B][ldc] staging array, or -- push-style exchange -- inside owner q's
// receive buffer
__device__ __forceinline__ float* chunk_row(float* send, const uint64_t* chunks, int q, int64_t B, int64_t b, int64_t ldc) {
  return chunks ? reinterpret_cast<float*>(chunks[q]) + b * ldc : send + (static_cast<int64_t>(q) * B + b) * ldc;
}

struct AsmArgs {
  const float* recv;     // forward: [N][B][ldc] chunks by owner;   backward: unused
  float* send;           // backward: [N][B][ldc] chunks by owner
  const uint64_t* send_chunks;   // or: chunk q at (float*)send_chunks[q] ([B][ldc], owner q's receive buffer: peer memory)
  int32_t carry_col, carry_n;    // send_chunks: columns [carry_col, +carry_n) of every local send row ride along (the next
                                 // batch's ids, staged there by dctr_shard_stage)
  int64_t ldc;
  int32_t N, B, F, D;    // ranks, local batch, deep units (fields), embedding dim
  const int32_t* owner_slot;   // [F] owner | slot << 16 of every unit, or NULL: owner f % N, slot f / N
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
  // round 6: the carried columns come straight from the NEXT batch's X (dctr_shard_assemble_bwd_next) -- carried value (q, j)
  // of sample b = x_next[b, carry_cols[q * carry_n + j]] -- instead of from the local send rows dctr_shard_stage filled
  const float* x_next;
  int64_t ldxn;
  const int32_t* carry_cols;
};

// unit f -> (owner rank q, slot j of that owner's chunk)
__device__ __forceinline__ void owner_of(const int32_t* map, int f, int N, int& q, int& j) {
  if (map) {
    const int32_t v = ldg_i32(map + f);
    q = v & 0xFFFF;
    j = v >> 16;
  } else {
    q = f % N;
    j = f / N;
  }
}

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
    int q, j;
    owner_of(A.owner_slot, f, A.N, q, j);
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
        int q, j;
        owner_of(A.owner_slot, f, A.N, q, j);
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
    int q, j;
    owner_of(A.owner_slot, f, A.N, q, j);
    float g = A.g_out ? ldg_f32(A.g_out + b * A.ldg + e) : 0.f;
    if (A.g_fm) g += gf * (ldg_f32(A.fm_s + b * A.lds_ + d) - ldg_f32(A.out + b * A.ldo + e));
    stg_f32(chunk_row(A.send, A.send_chunks, q, A.B, b, A.ldc) + j * A.D + d, g);
  }
  if (A.wide_col >= 0) {
    const float gw = A.g_wide ? ldg_f32(A.g_wide + b) : 0.f;
    for (int q = c; q < A.N; q += 16) stg_f32(chunk_row(A.send, A.send_chunks, q, A.B, b, A.ldc) + A.wide_col, gw);
  }
  if (A.send_chunks && A.carry_n > 0) {
    for (int t = c; t < A.N * A.carry_n; t += 16) {
      const int q = t / A.carry_n, j = t - q * A.carry_n;
      stg_f32(chunk_row(nullptr, A.send_chunks, q, A.B, b, A.ldc) + A.carry_col + j,
              A.x_next ? ldg_f32(A.x_next + b * A.ldxn + ldg_i32(A.carry_cols + t))
                       : ldg_f32(chunk_row(A.send, nullptr, q, A.B, b, A.ldc) + A.carry_col + j));
    }
  }
}

// ---- vectorised variants (D % 4 == 0, every pointer / leading dimension 16-byte aligned) -------------------------
// 32 lanes per sample: lane = (fl = lane / (D/4) ..., d4): each lane moves whole dwordx4 pieces of field rows, all of
// its loads in flight at once; the FM sums are reduced across the sample's lanes with DPP shuffles.
template <int D4>   // D / 4, a power of two <= 16
__global__ __launch_bounds__(kT) void k_assemble_fwd_v4(AsmArgs A) {
  constexpr int LPS = 32;                 // lanes per sample
  constexpr int FL = LPS / D4;            // fields walked in parallel
  constexpr int SPB = kT / LPS;           // samples per workgroup
  const int tid = threadIdx.x, sl = tid / LPS, l = tid % LPS, d4 = l % D4, fl = l / D4;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * SPB + sl;
  const bool valid = b < A.B;
  const int64_t bb = valid ? b : 0;
  const int D = 4 * D4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
  for (int f0 = 0; f0 < A.F; f0 += 4 * FL) {
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = f0 + k * FL + fl;
      const int fc = f < A.F ? f : 0;
      int q, j;
      owner_of(A.owner_slot, fc, A.N, q, j);
      v[k] = *(const DCTR_GLOBAL f32x4*)(A.recv + (static_cast<int64_t>(q) * A.B + bb) * A.ldc + j * D + 4 * d4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = f0 + k * FL + fl;
      if (f < A.F) {
        if (valid) *(DCTR_GLOBAL f32x4*)(A.out + b * A.ldo + f * D + 4 * d4) = v[k];
        s += v[k];
        sq += v[k] * v[k];
      }
    }
  }
  if (A.fm || A.fm_s) {
#pragma unroll
    for (int m = D4; m < LPS; m <<= 1) {     // across the FL field lanes that share d4
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s[c] += __shfl_xor(s[c], m, kWave);
        sq[c] += __shfl_xor(sq[c], m, kWave);
      }
    }
    if (A.fm_s && valid && fl == 0) *(DCTR_GLOBAL f32x4*)(A.fm_s + b * A.lds_ + 4 * d4) = s;
    float tot = (s[0] * s[0] - sq[0]) + (s[1] * s[1] - sq[1]) + (s[2] * s[2] - sq[2]) + (s[3] * s[3] - sq[3]);
#pragma unroll
    for (int m = 1; m < D4; m <<= 1) tot += __shfl_xor(tot, m, kWave);
    if (A.fm && valid && l == 0) stg_f32(A.fm + b, 0.5f * tot);
  }
  for (int k = l; k < A.n_dense; k += LPS)
    if (valid) stg_f32(A.out + b * A.ldo + A.dense_off + k, ldg_f32(A.X + b * A.ldx + ldg_i32(A.dense_cols + k)));
  if (A.wide) {
    float w = 0.f;
    if (A.wide_col >= 0)
      for (int q = l; q < A.N; q += LPS) w += ldg_f32(A.recv + (static_cast<int64_t>(q) * A.B + bb) * A.ldc + A.wide_col);
    for (int k = l; k < A.n_wdense; k += LPS)
      w += ldg_f32(A.X + bb * A.ldx + ldg_i32(A.wdense_cols + k)) * ldg_f32(A.wdense_w + k);
    w = group_sum<LPS>(w);
    if (l == 0 && valid) stg_f32(A.wide + b, w);
  }
}

template <int D4>
__global__ __launch_bounds__(kT) void k_assemble_bwd_v4(AsmArgs A) {
  constexpr int LPS = 32, FL = LPS / D4, SPB = kT / LPS;
  const int tid = threadIdx.x;
  const int nblk = (A.B + SPB - 1) / SPB;
  const int n_col = static_cast<int>(gridDim.x) - nblk;
  // d loss / d Linear.weight, one workgroup per dense column -- the FIRST workgroups of the grid: each walks all B samples
  // (two dependent rounds of strided loads + a reduction, ~6 us) and used to start last, as the launch's tail
  if (static_cast<int>(blockIdx.x) < n_col) {
    __shared__ float red[kT / 64];
    const int j = static_cast<int>(blockIdx.x);
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
  const int sl = tid / LPS, l = tid % LPS, d4 = l % D4, fl = l / D4;
  const int64_t b = static_cast<int64_t>(static_cast<int>(blockIdx.x) - n_col) * SPB + sl;
  if (b >= A.B) return;
  const int D = 4 * D4;
  const float gf = A.g_fm ? ldg_f32(A.g_fm + b) : 0.f;
  f32x4 S = {0.f, 0.f, 0.f, 0.f};
  if (A.g_fm) S = *(const DCTR_GLOBAL f32x4*)(A.fm_s + b * A.lds_ + 4 * d4);
  for (int f0 = 0; f0 < A.F; f0 += 4 * FL) {
    f32x4 g[4], e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = f0 + k * FL + fl;
      const int fc = f < A.F ? f : 0;
      g[k] = A.g_out ? *(const DCTR_GLOBAL f32x4*)(A.g_out + b * A.ldg + fc * D + 4 * d4) : f32x4{0.f, 0.f, 0.f, 0.f};
      e[k] = A.g_fm ? *(const DCTR_GLOBAL f32x4*)(A.out + b * A.ldo + fc * D + 4 * d4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = f0 + k * FL + fl;
      if (f < A.F) {
        int q, j;
        owner_of(A.owner_slot, f, A.N, q, j);
        f32x4 o = g[k];
        if (A.g_fm) {
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] += gf * (S[c] - e[k][c]);
        }
        *(DCTR_GLOBAL f32x4*)(chunk_row(A.send, A.send_chunks, q, A.B, b, A.ldc) + j * D + 4 * d4) = o;
      }
    }
  }
  if (A.wide_col >= 0) {
    const float gw = A.g_wide ? ldg_f32(A.g_wide + b) : 0.f;
    for (int q = l; q < A.N; q += LPS) stg_f32(chunk_row(A.send, A.send_chunks, q, A.B, b, A.ldc) + A.wide_col, gw);
  }
  if (A.send_chunks && A.carry_n > 0) {
    for (int t = l; t < A.N * A.carry_n; t += LPS) {
      const int q = t / A.carry_n, j = t - q * A.carry_n;
      stg_f32(chunk_row(nullptr, A.send_chunks, q, A.B, b, A.ldc) + A.carry_col + j,
              A.x_next ? ldg_f32(A.x_next + b * A.ldxn + ldg_i32(A.carry_cols + t))
                       : ldg_f32(chunk_row(A.send, nullptr, q, A.B, b, A.ldc) + A.carry_col + j));
    }
  }
}

// One launch in front of a sharded step: this rank's batch into the static buffers the captured step reads, and the NEXT
// batch's id columns into the id slots of the gradient chunks (the owners need them for step k + 1's gather; they ride in
// step k's gradient exchange).  Replaces five torch launches (two copies, index_select, a strided copy, an elementwise
// copy inside the step: ~25 us of kernel boundaries per step).
__global__ __launch_bounds__(kT) void k_shard_stage(const float* __restrict__ xb, int64_t ld_xb, const float* __restrict__ yb,
                                                    int B, int ncols, float* __restrict__ x_dst, int64_t ld_xd,
                                                    float* __restrict__ y_dst, const float* __restrict__ x_next,
                                                    int64_t ld_xn, const int32_t* __restrict__ id_cols, int N, int n_slots,
                                                    float* __restrict__ send, int64_t ldc, int ids_col) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  const int64_t nx = static_cast<int64_t>(B) * ncols;
  if (t < nx) {
    const int64_t b = t / ncols, c = t - b * ncols;
    stg_f32(x_dst + b * ld_xd + c, ldg_f32(xb + b * ld_xb + c));
  }
  if (t < B && y_dst) stg_f32(y_dst + t, ldg_f32(yb + t));
  const int64_t ni = x_next ? static_cast<int64_t>(B) * N * n_slots : 0;
  if (t < ni) {
    const int64_t b = t / (N * n_slots), k = t - b * (N * n_slots);
    const int q = static_cast<int>(k / n_slots), j = static_cast<int>(k - static_cast<int64_t>(q) * n_slots);
    stg_f32(send + (static_cast<int64_t>(q) * B + b) * ldc + ids_col + j, ldg_f32(x_next + b * ld_xn + ldg_i32(id_cols + k)));
  }
}

inline bool al16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace

extern "C" int dctr_shard_assemble_fwd(const float* recv, int64_t ld_chunk, int32_t n_ranks, int32_t B, int32_t F,
                                       int32_t D, const int32_t* owner_slot, int32_t wide_col, const float* X, int64_t ld_x,
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
  a.owner_slot = owner_slot;
  a.X = X; a.ldx = ld_x; a.dense_cols = dense_cols; a.n_dense = n_dense; a.dense_off = dense_off;
  a.wdense_cols = wdense_cols; a.wdense_w = wdense_w; a.n_wdense = n_wdense;
  a.out = out; a.ldo = ld_out; a.wide = wide; a.fm = fm; a.fm_s = fm_s; a.lds_ = ld_s;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = D % 4 == 0 && ld_chunk % 4 == 0 && ld_out % 4 == 0 && al16(recv) && al16(out) &&
                   (!fm_s || (ld_s % 4 == 0 && al16(fm_s)));
  const dim3 g8((B + 7) / 8), blk(kT);
  if (vec && D == 16) k_assemble_fwd_v4<4><<<g8, blk, 0, st>>>(a);
  else if (vec && D == 8) k_assemble_fwd_v4<2><<<g8, blk, 0, st>>>(a);
  else if (vec && D == 32) k_assemble_fwd_v4<8><<<g8, blk, 0, st>>>(a);
  else if (vec && D == 4) k_assemble_fwd_v4<1><<<g8, blk, 0, st>>>(a);
  else if (vec && D == 64) k_assemble_fwd_v4<16><<<g8, blk, 0, st>>>(a);
  else k_assemble_fwd<<<dim3((B + 15) / 16), blk, 0, st>>>(a);
  return launch_status();
}

extern "C" int dctr_shard_assemble_bwd(float* send, const uint64_t* send_chunks, int32_t carry_col, int32_t carry_n,
                                       int64_t ld_chunk, int32_t n_ranks, int32_t B, int32_t F, int32_t D,
                                       const int32_t* owner_slot, int32_t wide_col, const float* g_out, int64_t ld_g, const float* g_wide,
                                       const float* g_fm, const float* out, int64_t ld_out, const float* fm_s,
                                       int64_t ld_s, const float* X, int64_t ld_x, const int32_t* wdense_cols,
                                       int32_t n_wdense, float* g_wdense, dctr_stream_t stream) {
  return dctr_shard_assemble_bwd_next(send, send_chunks, carry_col, carry_n, ld_chunk, n_ranks, B, F, D, owner_slot, wide_col, g_out,
                                      ld_g, g_wide, g_fm, out, ld_out, fm_s, ld_s, X, ld_x, wdense_cols, n_wdense, g_wdense,
                                      nullptr, 0, nullptr, stream);
}

extern "C" int dctr_shard_assemble_bwd_next(float* send, const uint64_t* send_chunks, int32_t carry_col, int32_t carry_n,
                                            int64_t ld_chunk, int32_t n_ranks, int32_t B, int32_t F, int32_t D,
                                            const int32_t* owner_slot, int32_t wide_col, const float* g_out, int64_t ld_g,
                                            const float* g_wide, const float* g_fm, const float* out, int64_t ld_out,
                                            const float* fm_s, int64_t ld_s, const float* X, int64_t ld_x,
                                            const int32_t* wdense_cols, int32_t n_wdense, float* g_wdense,
                                            const float* x_next, int64_t ld_xn, const int32_t* carry_cols,
                                            dctr_stream_t stream) {
  if ((!send && !send_chunks) || n_ranks <= 0 || B < 0 || F <= 0 || D <= 0) return DCTR_EINVAL;
  if (g_fm && (!out || !fm_s)) return DCTR_EINVAL;
  if (x_next && (!carry_cols || !send_chunks || carry_n <= 0 || ld_xn <= 0)) return DCTR_EINVAL;
  if (carry_n < 0 || (carry_n > 0 && ((!send && !x_next) || !send_chunks || carry_col < 0 || carry_col + carry_n > ld_chunk))) return DCTR_EINVAL;
  if (g_wdense && (!X || !g_wide || !wdense_cols || n_wdense <= 0)) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  AsmArgs a = {};
  a.send = send; a.send_chunks = send_chunks; a.carry_col = carry_col; a.carry_n = carry_n; a.ldc = ld_chunk; a.N = n_ranks; a.B = B; a.F = F; a.D = D; a.wide_col = wide_col;
  a.owner_slot = owner_slot;
  a.g_out = g_out; a.ldg = ld_g; a.g_wide = g_wide; a.g_fm = g_fm; a.out = const_cast<float*>(out); a.ldo = ld_out;
  a.fm_s = const_cast<float*>(fm_s); a.lds_ = ld_s; a.X = X; a.ldx = ld_x; a.wdense_cols = wdense_cols;
  a.n_wdense = n_wdense; a.g_wdense = g_wdense;
  a.x_next = x_next; a.ldxn = ld_xn; a.carry_cols = carry_cols;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned extra = g_wdense ? static_cast<unsigned>(n_wdense) : 0u;
  // (send_chunks: the receive buffers are whole allocations, 16-byte aligned by construction -- the caller's contract)
  const bool vec = D % 4 == 0 && ld_chunk % 4 == 0 && (send_chunks || al16(send)) && (!g_out || (ld_g % 4 == 0 && al16(g_out))) &&
                   (!g_fm || (ld_out % 4 == 0 && al16(out) && ld_s % 4 == 0 && al16(fm_s)));
  const dim3 g8((B + 7) / 8 + extra), blk(kT);
  if (vec && D == 16) k_assemble_bwd_v4<4><<<g8, blk, 0, st>>>(a);
  else if (vec && D == 8) k_assemble_bwd_v4<2><<<g8, blk, 0, st>>>(a);
  else if (vec && D == 32) k_assemble_bwd_v4<8><<<g8, blk, 0, st>>>(a);
  else if (vec && D == 4) k_assemble_bwd_v4<1><<<g8, blk, 0, st>>>(a);
  else if (vec && D == 64) k_assemble_bwd_v4<16><<<g8, blk, 0, st>>>(a);
  else k_assemble_bwd<<<dim3((B + 15) / 16 + extra), blk, 0, st>>>(a);
  return launch_status();
}

extern "C" int dctr_shard_stage(const float* xb, int64_t ld_xb, const float* yb, int32_t B, int32_t ncols, float* x_dst,
                                int64_t ld_xd, float* y_dst, const float* x_next, int64_t ld_xn, const int32_t* id_cols,
                                int32_t n_ranks, int32_t n_slots, float* send, int64_t ld_chunk, int32_t ids_col,
                                dctr_stream_t stream) {
  if (!xb || !x_dst || B < 0 || ncols <= 0 || ld_xb < ncols || ld_xd < ncols) return DCTR_EINVAL;
  if (y_dst && !yb) return DCTR_EINVAL;
  if (x_next && (!id_cols || !send || n_ranks <= 0 || n_slots <= 0 || ids_col < 0 || ld_chunk < ids_col + n_slots))
    return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  int64_t n = static_cast<int64_t>(B) * ncols;
  const int64_t ni = x_next ? static_cast<int64_t>(B) * n_ranks * n_slots : 0;
  if (ni > n) n = ni;
  k_shard_stage<<<dim3(static_cast<unsigned>((n + kT - 1) / kT)), dim3(kT), 0, static_cast<hipStream_t>(stream)>>>(
      xb, ld_xb, yb, B, ncols, x_dst, ld_xd, y_dst, x_next, ld_xn, id_cols, n_ranks, n_slots, send, ld_chunk, ids_col);
  return launch_status();
}

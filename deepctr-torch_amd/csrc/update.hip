// update.hip -- deterministic, atomic-free embedding backward with the optimizer fused in (gfx950).
//
// Replaces autograd's aten::embedding_dense_backward x (n_deep + n_wide) + FM's backward + the dense
// optimizer walk over every table (basemodel.py:261-262, interaction.py:26-34) for plans made of
// fixed-length fields over distinct tables -- the Criteo shape.  (Plans with pooled VarLen fields or
// shared tables keep the atomic two-pass kernels of embed.hip.)
//
// Why not atomics: measured on MI355X at B=4096 (profiles/r01_*), the scatter (1.7 M dword atomics)
// took 36 us and the xchg-consume pass 59 us -- 9 % of the HBM roofline -- and float atomics make
// duplicate-row sums order-dependent, so data-parallel replicas drift apart.
//
// Mapping: a "unit" is one id column of X with the deep table and/or the wide (1-dim) table it
// feeds.  Workgroup (unit u, partition p) owns the rows {id : id mod P == p} of u's tables, so no two
// workgroups ever touch the same row:
//   1. scan   the unit's B ids (ids_t[u][0..B), contiguous int32 written by the forward kernel) and
//             collect the entries of this partition as 32-bit keys (id / P) << bbits | b in LDS;
//   2. sort   the keys (bitonic, in LDS; typically 64 of them) -> entries ordered by (id, b);
//   3. reduce tiles of G entries: every lane group fetches ITS entry's gradient strip
//             g = g_out[b, f] + g_fm[b] * (S[b] - e[b, f]) (all loads of a tile in flight at once) and
//             the table / state strips of its row, parks g in LDS; the last entry of each id segment
//             sums its segment in fixed order (carry across tiles) and applies the update with plain
//             coalesced 64-byte row stores.
// Every row is read-modify-written exactly once, by one lane group, in an order that depends only on
// (id, b): results are bit-reproducible run to run and rank to rank.
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kThreads = 256;

struct UpdArgs {
  const dctr_field_t* deep;
  const dctr_field_t* wide;
  const int32_t* units;  // [n_units][4] = {deep index | -1, wide index | -1, X column, 0}
  const int32_t* ids_t;  // [n_units][B] truncated ids
  const float* gout;     // [B, ldg]   d loss / d out (deep slices), nullable
  const float* out;      // [B, ldo]   forward output (e), needed with gfm
  const float* fm_s;     // [B, lds_]  S[b, :] = sum_f e[b, f, :], needed with gfm
  const float* gfm;      // [B] nullable
  const float* gwide;    // [B] (stride ldgw) nullable
  int64_t ldgw;
  int64_t ldg, ldo, lds_;
  int32_t n_units, B, log2p, bbits;
  int32_t gt;            // entries per tile (<= lane groups per workgroup; sized by the host so that 7 WGs fit a CU)
  float lr, eps;
  // optional extra role (last block): d loss / d Linear.weight = X_dense^T g_wide  (basemodel.py:88-90)
  const float* X;
  int64_t ldx;
  const int32_t* wdense_cols;
  int32_t n_wdense;
  float* g_wdense;
  unsigned long long* trace;  // diagnostics (tools/upd_trace.py): 8 timestamps per workgroup, or NULL
};

unsigned long long* g_trace = nullptr;  // host-side: set by dctr_dbg_update_trace
int g_force_log2p = -1;

#define DCTR_TRACE(slot)                                                     \
  do {                                                                       \
    if (A.trace && tid == 0) A.trace[blockIdx.x * 8ull + (slot)] = wall_clock64(); \
  } while (0)

__device__ __forceinline__ int32_t clamp_id(int32_t id, int64_t vocab) {
  return (static_cast<uint64_t>(static_cast<int64_t>(id)) >= static_cast<uint64_t>(vocab)) ? 0 : id;
}

// One optimizer step on a strip of a row.  OPT: 0 SGD, 1 Adagrad, 2 accumulate into gacc.
template <int VEC, int OPT>
__device__ __forceinline__ void apply_strip(const dctr_field_t& fd, int64_t off, const Strip<VEC>& G,
                                            const Strip<VEC>& w, const Strip<VEC>& s, float lr,
                                            float eps) {
  Strip<VEC> nw, ns;
  if (OPT == DCTR_UPD_ADAGRAD) {  // torch.optim.Adagrad: s += g*g ; p -= lr * g / (sqrt(s) + eps)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      ns.v[i] = s.v[i] + G.v[i] * G.v[i];
      nw.v[i] = w.v[i] - lr * (G.v[i] / (sqrtf(ns.v[i]) + eps));
    }
    strip_store<VEC>(fd.state + off, ns);
    strip_store<VEC>(fd.table + off, nw);
  } else if (OPT == DCTR_UPD_SGD) {  // torch.optim.SGD: p -= lr * g
#pragma unroll
    for (int i = 0; i < VEC; ++i) nw.v[i] = w.v[i] - lr * G.v[i];
    strip_store<VEC>(fd.table + off, nw);
  } else {  // dense-gradient semantics: gacc[row] += g   (w holds the gacc strip)
#pragma unroll
    for (int i = 0; i < VEC; ++i) nw.v[i] = w.v[i] + G.v[i];
    strip_store<VEC>(fd.gacc + off, nw);
  }
}

// (7 workgroups of 4 waves per CU: <= 72 VGPRs)
template <int VEC, int LPR, int OPT>
__global__ __launch_bounds__(kThreads, 7) void k_embed_update(UpdArgs A) {
  constexpr int GMAX = kThreads / LPR;  // lane groups per workgroup
  const int G = A.gt < GMAX ? A.gt : GMAX;  // entries per tile
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int n_sh;
  __shared__ int carry_id;
  const int tid = threadIdx.x;
  const int P = 1 << A.log2p;
  DCTR_TRACE(0);

  if (A.g_wdense && static_cast<int>(blockIdx.x) >= static_cast<int>(gridDim.x) - A.n_wdense) {
    // the dense half of Linear (basemodel.py:86-90): g_w[j] = sum_b g_wide[b] * X[b, col_j].  One extra
    // workgroup per dense column, hidden behind the row updates; per-thread partial sums over a strided row set,
    // then a fixed-order tree => deterministic.  (Kept tiny on purpose: its registers bound the whole kernel.)
    __shared__ float red[kThreads / 64];
    const int j = static_cast<int>(blockIdx.x) - (static_cast<int>(gridDim.x) - A.n_wdense);
    const int col = ldg_i32(A.wdense_cols + j);
    float acc = 0.f;
#pragma unroll 8
    for (int b = tid; b < A.B; b += kThreads)
      acc += ldg_f32(A.gwide + static_cast<int64_t>(b) * A.ldgw) * ldg_f32(A.X + static_cast<int64_t>(b) * A.ldx + col);
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < kThreads / 64; ++w) t += red[w];
      stg_f32(A.g_wdense + j, t);
    }
    return;
  }

  // Work item w = (unit, partition) in plain launch order: consecutive workgroups go to consecutive XCDs, so every
  // XCD gets the same number of working workgroups.  (Keeping a unit's partitions on one XCD -- to fetch its id row
  // into one L2 only -- left XCDs 0-1 with 4 units and the others with 3 at F = 26: a second round on two XCDs.)
  const int u = static_cast<int>(blockIdx.x) >> A.log2p, p = static_cast<int>(blockIdx.x) & (P - 1);
  if (u >= A.n_units) return;

  const int32_t* un = A.units + 4 * u;
  const int di = un[0], wi = un[1];
  dctr_field_t fd, fw;
  if (di >= 0) fd = A.deep[di];
  if (wi >= 0) fw = A.wide[wi];
  const int64_t vocab = (di >= 0) ? fd.vocab : fw.vocab;
  const int B = A.B;
  int cap = 2;
  while (cap < B) cap <<= 1;

  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);           // [cap]
  float* gbuf = reinterpret_cast<float*>(keys + cap);            // [G][LPR*VEC] deep gradient strips
  float* gwbuf = gbuf + G * LPR * VEC;                           // [G] wide gradients
  float* carry = gwbuf + G;                                      // [LPR*VEC + 4]
  float* gfbuf = carry + LPR * VEC + 4;                          // [G]  single-tile path: g_fm of the entry
  uint32_t* skeys = reinterpret_cast<uint32_t*>(gfbuf + G);      // [G]  single-tile path: keys in sorted order

  if (tid == 0) {
    n_sh = 0;
    carry_id = -1;
  }
  __syncthreads();

  // ---- 1. scan: collect this partition's entries -------------------------------------------------
  // All id loads of a chunk are issued before any is consumed (16 ids per thread in flight at B=4096):
  // the scan costs one L2 round trip, not one per iteration.
  const int32_t* ids = A.ids_t + static_cast<int64_t>(u) * B;
  auto take = [&](int32_t raw, int b) {
    const int32_t id = clamp_id(raw, vocab);
    if ((id & (P - 1)) == p) {
      const int slot = atomicAdd(&n_sh, 1);  // LDS atomic; the order is fixed by the sort below
      keys[slot] = (static_cast<uint32_t>(id >> A.log2p) << A.bbits) | static_cast<uint32_t>(b);
    }
  };
  if ((B & 3) == 0) {
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    const DCTR_GLOBAL i32x4* idv = (const DCTR_GLOBAL i32x4*)ids;
    const int nvec = B >> 2;
    for (int c0 = 0; c0 < nvec; c0 += 4 * kThreads) {
      i32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = c0 + q * kThreads + tid;
        v[q] = idv[idx < nvec ? idx : 0];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = c0 + q * kThreads + tid;
        if (idx < nvec) {
          take(v[q].x, 4 * idx);
          take(v[q].y, 4 * idx + 1);
          take(v[q].z, 4 * idx + 2);
          take(v[q].w, 4 * idx + 3);
        }
      }
    }
  } else {
    for (int b = tid; b < B; b += kThreads) take(ldg_i32(ids + b), b);
  }
  __syncthreads();
  const int n = n_sh;
  DCTR_TRACE(1);
  if (A.trace && tid == 0) A.trace[blockIdx.x * 8ull + 7] = static_cast<unsigned long long>(n);
  if (n == 0) return;

  if (n <= G) {
    // ---- single tile (the common case: P is chosen so that a partition holds ~G/2 entries) -------------------
    // Lane group i takes the i-th entry in SCAN order and issues its loads at once; the rank sort runs in their
    // shadow; gradients are parked in LDS at their SORTED position; the last entry of every id segment sums its
    // segment backwards (fixed order) and applies the update to the row strips it already holds.
    // FM's backward is folded algebraically: sum_seg [g + gf (S - e)] = sum_seg (g + gf S) - (sum_seg gf) e, and e
    // IS the table row this lane is about to update -- the forward's copy of it is not re-read.
    const int grp = tid / LPR, gl = tid % LPR, e0 = gl * VEC;
    const uint32_t bmask = (1u << A.bbits) - 1u;
    const bool deep_on = (di >= 0) && (A.gout || A.gfm);
    const bool wide_on = (wi >= 0) && A.gwide;
    const bool lane_on = deep_on && (e0 < fd.dim);
    const int goff = deep_on ? fd.out_off + (lane_on ? e0 : 0) : 0;
    const bool have = grp < n;
    const uint32_t key = have ? keys[grp] : 0xFFFFFFFFu;
    const int b = static_cast<int>(key & bmask);
    const int idq = static_cast<int>(key >> A.bbits);
    const int64_t row = (static_cast<int64_t>(idq) << A.log2p) | p;
    Strip<VEC> h = strip_zero<VEC>(), S = strip_zero<VEC>(), w = strip_zero<VEC>(), s = strip_zero<VEC>();
    Strip<VEC> e = strip_zero<VEC>();
    float gf = 0.f, gw = 0.f, ww = 0.f, sw = 0.f;
    if (have) {
      if (lane_on) {
        if (A.gout) h = strip_load<VEC>(A.gout + static_cast<int64_t>(b) * A.ldg + goff);
        if (A.gfm) {
          S = strip_load<VEC>(A.fm_s + static_cast<int64_t>(b) * A.lds_ + e0);
          gf = ldg_f32(A.gfm + b);
          if (OPT == DCTR_UPD_ACCUM) e = strip_load<VEC>(fd.table + row * fd.dim + e0);
        }
        const int64_t off = row * fd.dim + e0;
        w = strip_load<VEC>((OPT == DCTR_UPD_ACCUM ? fd.gacc : fd.table) + off);
        if (OPT == DCTR_UPD_ADAGRAD) s = strip_load<VEC>(fd.state + off);
      }
      if (wide_on && gl == 0) {
        gw = ldg_f32(A.gwide + static_cast<int64_t>(b) * A.ldgw);
        ww = ldg_f32((OPT == DCTR_UPD_ACCUM ? fw.gacc : fw.table) + row);
        if (OPT == DCTR_UPD_ADAGRAD) sw = ldg_f32(fw.state + row);
      }
    }
    int rank = 0;  // keys are unique: rank = number of smaller keys
#pragma unroll 8
    for (int q = 0; q < n; ++q) rank += (keys[q] < key) ? 1 : 0;
    DCTR_TRACE(2);
    if (have) {
      if (gl == 0) {
        skeys[rank] = key;
        gfbuf[rank] = gf;
        gwbuf[rank] = gw;
      }
      if (lane_on) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          float v = h.v[k] + gf * S.v[k];
          if (OPT == DCTR_UPD_ACCUM) v = h.v[k] + gf * (S.v[k] - e.v[k]);
          gbuf[rank * (LPR * VEC) + e0 + k] = v;
        }
      }
    }
    DCTR_TRACE(3);
    __syncthreads();
    DCTR_TRACE(4);
    if (have) {
      const bool seg_end = (rank == n - 1) || (static_cast<int>(skeys[rank + 1] >> A.bbits) != idq);
      if (seg_end) {
        Strip<VEC> acc = strip_zero<VEC>();
        float accf = 0.f, accw = 0.f;
        int r = rank;
        while (r >= 0 && static_cast<int>(skeys[r] >> A.bbits) == idq) {
          if (lane_on) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc.v[k] += gbuf[r * (LPR * VEC) + e0 + k];
          }
          accf += gfbuf[r];
          if (gl == 0) accw += gwbuf[r];
          --r;
        }
        if (lane_on) {
          if (OPT != DCTR_UPD_ACCUM && A.gfm) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc.v[k] -= accf * w.v[k];
          }
          apply_strip<VEC, OPT>(fd, row * fd.dim + e0, acc, w, s, A.lr, A.eps);
        }
        if (wide_on && gl == 0) {
          Strip<1> a1, w1, s1;
          a1.v[0] = accw;
          w1.v[0] = ww;
          s1.v[0] = sw;
          apply_strip<1, OPT>(fw, row, a1, w1, s1, A.lr, A.eps);
        }
      }
    }
    DCTR_TRACE(5);
    DCTR_TRACE(6);
    return;
  }

  // ---- 2. sort by (id, b) ------------------------------------------------------------------------
  if (n <= kThreads) {
    // rank sort: keys are unique, so rank = #smaller is a permutation; 2 barriers instead of ~21-36
    const uint32_t mine = tid < n ? keys[tid] : 0u;
    int rank = 0;
#pragma unroll 8
    for (int q = 0; q < n; ++q) rank += (keys[q] < mine) ? 1 : 0;  // broadcast LDS reads
    __syncthreads();
    if (tid < n) keys[rank] = mine;
    __syncthreads();
  } else {
    int m = 2;
    while (m < n) m <<= 1;
    for (int i = n + tid; i < m; i += kThreads) keys[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
      for (int s = k >> 1; s > 0; s >>= 1) {
        for (int i = tid; i < m; i += kThreads) {
          const int ixs = i ^ s;
          if (ixs > i) {
            const uint32_t a = keys[i], b2 = keys[ixs];
            const bool up = (i & k) == 0;
            if ((a > b2) == up) {
              keys[i] = b2;
              keys[ixs] = a;
            }
          }
        }
        __syncthreads();
      }
    }
  }

  DCTR_TRACE(2);
  // ---- 3. tiles of G entries ---------------------------------------------------------------------
  const int grp = tid / LPR, gl = tid % LPR;
  const int e0 = gl * VEC;
  const uint32_t bmask = (1u << A.bbits) - 1u;
  const bool deep_on = (di >= 0) && (A.gout || A.gfm);
  const bool wide_on = (wi >= 0) && A.gwide;
  const bool lane_on = deep_on && (e0 < fd.dim);
  const int goff = deep_on ? fd.out_off + (lane_on ? e0 : 0) : 0;

  for (int t0 = 0; t0 < n; t0 += G) {
    const int i = t0 + grp;
    const bool have = i < n && grp < G;
    const uint32_t key = have ? keys[i] : 0u;
    const int b = static_cast<int>(key & bmask);
    const int idq = static_cast<int>(key >> A.bbits);           // id / P
    const int64_t row = (static_cast<int64_t>(idq) << A.log2p) | p;

    // issue every load of this entry: gradient pieces, then the row strips it may update
    Strip<VEC> g = strip_zero<VEC>(), w = strip_zero<VEC>(), s = strip_zero<VEC>();
    float gw = 0.f, ww = 0.f, sw = 0.f;
    if (have) {
      if (lane_on) {
        if (A.gout) g = strip_load<VEC>(A.gout + static_cast<int64_t>(b) * A.ldg + goff);
        if (A.gfm) {
          const Strip<VEC> e = strip_load<VEC>(A.out + static_cast<int64_t>(b) * A.ldo + goff);
          const Strip<VEC> S = strip_load<VEC>(A.fm_s + static_cast<int64_t>(b) * A.lds_ + e0);
          const float gf = ldg_f32(A.gfm + b);
#pragma unroll
          for (int k = 0; k < VEC; ++k) g.v[k] += gf * (S.v[k] - e.v[k]);
        }
        const int64_t off = row * fd.dim + e0;
        w = strip_load<VEC>((OPT == DCTR_UPD_ACCUM ? fd.gacc : fd.table) + off);
        if (OPT == DCTR_UPD_ADAGRAD) s = strip_load<VEC>(fd.state + off);
      }
      if (wide_on && gl == 0) {
        gw = ldg_f32(A.gwide + static_cast<int64_t>(b) * A.ldgw);
        ww = ldg_f32((OPT == DCTR_UPD_ACCUM ? fw.gacc : fw.table) + row);
        if (OPT == DCTR_UPD_ADAGRAD) sw = ldg_f32(fw.state + row);
      }
    }
    if (lane_on && grp < G) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) gbuf[grp * (LPR * VEC) + e0 + k] = g.v[k];
    }
    if (gl == 0 && grp < G) gwbuf[grp] = gw;
    if (t0 == 0) DCTR_TRACE(3);
    __syncthreads();
    if (t0 == 0) DCTR_TRACE(4);

    // the last entry of an id segment (or of the tile) sums the segment's members inside this tile
    const bool last_of_tile = have && ((grp == G - 1) || (i == n - 1));
    const bool seg_end = have && ((i == n - 1) || (static_cast<int>(keys[i + 1] >> A.bbits) != idq));
    const bool summer = seg_end || last_of_tile;
    Strip<VEC> acc = strip_zero<VEC>();
    float accw = 0.f;
    if (summer) {
      int jj = grp;  // walk back: fixed order => deterministic
      while (jj >= 0 && static_cast<int>(keys[t0 + jj] >> A.bbits) == idq) {
        if (lane_on) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc.v[k] += gbuf[jj * (LPR * VEC) + e0 + k];
        }
        if (gl == 0) accw += gwbuf[jj];
        --jj;
      }
      if (jj < 0 && carry_id == idq) {  // the segment began in an earlier tile
        if (lane_on) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc.v[k] += carry[e0 + k];
        }
        if (gl == 0) accw += carry[LPR * VEC];
      }
      if (seg_end) {
        if (lane_on) apply_strip<VEC, OPT>(fd, row * fd.dim + e0, acc, w, s, A.lr, A.eps);
        if (wide_on && gl == 0) {
          Strip<1> a1, w1, s1;
          a1.v[0] = accw;
          w1.v[0] = ww;
          s1.v[0] = sw;
          apply_strip<1, OPT>(fw, row, a1, w1, s1, A.lr, A.eps);
        }
      }
    }
    __syncthreads();  // every read of gbuf / carry of this tile is done
    if (last_of_tile) {  // exactly one group: park an open segment's partial, or clear the carry
      if (!seg_end) {
        if (lane_on) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) carry[e0 + k] = acc.v[k];
        }
        if (gl == 0) {
          carry[LPR * VEC] = accw;
          carry_id = idq;
        }
      } else if (gl == 0) {
        carry_id = -1;
      }
    }
    __syncthreads();
    if (t0 == 0) DCTR_TRACE(5);
  }
  DCTR_TRACE(6);
}

// ---- X -> ids_t (standalone; the forward kernel fuses the same thing) -------------------------------
__global__ __launch_bounds__(kThreads) void k_embed_ids(const int32_t* __restrict__ units, int n_units,
                                                        const float* __restrict__ X, int64_t ldx, int B,
                                                        int32_t* __restrict__ ids_t) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= static_cast<int64_t>(n_units) * B) return;
  const int u = static_cast<int>(i / B), b = static_cast<int>(i - static_cast<int64_t>(u) * B);
  ids_t[i] = static_cast<int32_t>(X[static_cast<int64_t>(b) * ldx + units[4 * u + 2]]);
}

// ~64 entries (one tile) per workgroup up to B = 8192; beyond that every partition's workgroup would
// re-scan too many ids, so partitions grow to ~256 entries (4 tiles).
int pick_log2p(int B) {
  if (g_force_log2p >= 0) return g_force_log2p;
  const int per = B > 8192 ? 256 : 64;
  int l = 0;
  while ((B >> l) > per && l < 10) ++l;
  return l;
}

int ceil_log2(int64_t x) {
  int l = 0;
  while ((int64_t(1) << l) < x) ++l;
  return l;
}

}  // namespace

extern "C" int dctr_embed_ids(const int32_t* units, int32_t n_units, const float* X, int64_t ldx,
                              int32_t B, int32_t* ids_t, dctr_stream_t stream) {
  if (!units || !X || !ids_t || n_units <= 0 || B < 0) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const int64_t n = static_cast<int64_t>(n_units) * B;
  k_embed_ids<<<dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                static_cast<hipStream_t>(stream)>>>(units, n_units, X, ldx, B, ids_t);
  return launch_status();
}

// diagnostics: per-workgroup phase timestamps (8 x u64 per workgroup, wall_clock64 ticks) and a partition override
extern "C" void dctr_dbg_update_trace(unsigned long long* buf, int32_t force_log2p) {
  g_trace = buf;
  g_force_log2p = force_log2p;
}

extern "C" int dctr_embed_update_supported(const dctr_plan_t* plan, int64_t max_vocab, int32_t B) {
  if (!plan || B <= 0) return 0;
  if (plan->n_deep != plan->n_deep_fixed || plan->n_wide != plan->n_wide_fixed) return 0;
  if (plan->vec != 1 && plan->vec != 2 && plan->vec != 4) return 0;
  if (plan->n_deep > 0 && plan->max_dim > 64 * plan->vec) return 0;
  if (B > 32768) return 0;
  const int log2p = pick_log2p(B);
  const int bbits = ceil_log2(B < 2 ? 2 : B);
  if (ceil_log2(((max_vocab > 0 ? max_vocab : 1) >> log2p) + 1) + bbits > 32) return 0;
  return 1;
}

extern "C" int dctr_embed_update(const dctr_plan_t* plan, const int32_t* units, int32_t n_units,
                                 int64_t max_vocab, const int32_t* ids_t, int32_t B, const float* g_out,
                                 int64_t ld_g, const float* out, int64_t ld_out, const float* fm_s,
                                 int64_t ld_s, const float* g_fm, const float* g_wide, int64_t ld_gw,
                                 int32_t opt, float lr, float eps, const float* X, int64_t ld_x, float* g_wdense,
                                 dctr_stream_t stream) {
  if (!plan || !units || !ids_t || n_units <= 0 || B < 0) return DCTR_EINVAL;
  if (g_wide && ld_gw < 1) return DCTR_EINVAL;
  if (g_wdense && (!X || !g_wide || plan->n_wdense <= 0 || !plan->wdense_cols)) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  if (opt != DCTR_UPD_SGD && opt != DCTR_UPD_ADAGRAD && opt != DCTR_UPD_ACCUM) return DCTR_EINVAL;
  if (!dctr_embed_update_supported(plan, max_vocab, B)) return DCTR_ENOSUP;
  if (opt == DCTR_UPD_ACCUM && !(plan->flags & DCTR_PLAN_HAS_GACC)) return DCTR_EINVAL;
  if (opt == DCTR_UPD_ADAGRAD && !(plan->flags & DCTR_PLAN_HAS_STATE)) return DCTR_EINVAL;
  if (g_fm && (!out || !fm_s || plan->emb_dim <= 0)) return DCTR_EINVAL;
  int vec = plan->n_deep > 0 ? plan->vec : 1;
  const int avec = vec;  // alignment granule the caller guarantees
  if (vec == 4 && plan->emb_dim > 0 && plan->emb_dim % 8 == 0 && plan->emb_dim <= 64) vec = 8;  // two dwordx4 per lane
  if (avec > 1) {
    const int vec = avec;
    if (g_out && (ld_g % vec != 0 || reinterpret_cast<uintptr_t>(g_out) % (4 * vec) != 0)) return DCTR_EALIGN;
    if (g_fm && (ld_out % vec != 0 || reinterpret_cast<uintptr_t>(out) % (4 * vec) != 0 ||
                 ld_s % vec != 0 || reinterpret_cast<uintptr_t>(fm_s) % (4 * vec) != 0))
      return DCTR_EALIGN;
  }
  UpdArgs a;
  a.deep = plan->deep; a.wide = plan->wide; a.units = units; a.ids_t = ids_t;
  a.gout = g_out; a.out = out; a.fm_s = fm_s; a.gfm = g_fm; a.gwide = g_wide; a.ldgw = ld_gw;
  a.ldg = ld_g; a.ldo = ld_out; a.lds_ = ld_s;
  a.n_units = n_units; a.B = B;
  const int log2p = pick_log2p(B);
  a.log2p = log2p;
  a.bbits = ceil_log2(B < 2 ? 2 : B);
  a.lr = lr; a.eps = eps;
  a.trace = g_trace;
  a.X = X; a.ldx = ld_x; a.wdense_cols = plan->wdense_cols; a.n_wdense = plan->n_wdense; a.g_wdense = g_wdense;

  int lpr = 1;
  const int need = plan->n_deep > 0 ? (plan->max_dim + vec - 1) / vec : 1;
  while (lpr < need) lpr <<= 1;
  int cap = 2;
  while (cap < B) cap <<= 1;
  const int g = kThreads / lpr;
  auto lds_for = [&](int gt) {
    return static_cast<size_t>(cap) * 4 + (static_cast<size_t>(gt) * lpr * vec + 3 * gt + lpr * vec + 4) * 4;
  };
  // 160 KB of LDS / 7 workgroups: with all of a launch's workgroups resident at once the kernel is one round of
  // ~15 us workgroups instead of two.  Shrink the tile (never below half the lane groups) if that gets us there.
  int gt = g;
  constexpr size_t kBudget = 160 * 1024 / 7 - 64;
  if (lds_for(g) > kBudget) {
    int t = g;
    while (t - 8 >= g / 2 && lds_for(t) > kBudget) t -= 8;
    if (lds_for(t) <= kBudget) gt = t;
  }
  a.gt = gt;
  const size_t lds = lds_for(gt);
  if (lds > 150 * 1024) return DCTR_ENOSUP;
  const dim3 grid((static_cast<unsigned>(n_units) << log2p) + (g_wdense ? static_cast<unsigned>(plan->n_wdense) : 0u)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);

#define DCTR_UPD_LAUNCH(VEC_, LPR_)                                                                   \
  do {                                                                                                \
    if (opt == DCTR_UPD_ADAGRAD) {                                                                    \
      if (lds > 64 * 1024)                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_embed_update<VEC_, LPR_, 1>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
      k_embed_update<VEC_, LPR_, 1><<<grid, block, lds, s>>>(a);                                      \
    } else if (opt == DCTR_UPD_SGD) {                                                                 \
      if (lds > 64 * 1024)                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_embed_update<VEC_, LPR_, 0>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
      k_embed_update<VEC_, LPR_, 0><<<grid, block, lds, s>>>(a);                                      \
    } else {                                                                                          \
      if (lds > 64 * 1024)                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_embed_update<VEC_, LPR_, 2>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
      k_embed_update<VEC_, LPR_, 2><<<grid, block, lds, s>>>(a);                                      \
    }                                                                                                 \
  } while (0)

#define DCTR_UPD_LPR(VEC_)                      \
  switch (lpr) {                                \
    case 1: DCTR_UPD_LAUNCH(VEC_, 1); break;    \
    case 2: DCTR_UPD_LAUNCH(VEC_, 2); break;    \
    case 4: DCTR_UPD_LAUNCH(VEC_, 4); break;    \
    case 8: DCTR_UPD_LAUNCH(VEC_, 8); break;    \
    case 16: DCTR_UPD_LAUNCH(VEC_, 16); break;  \
    case 32: DCTR_UPD_LAUNCH(VEC_, 32); break;  \
    default: DCTR_UPD_LAUNCH(VEC_, 64); break;  \
  }

#define DCTR_UPD_LPR8()                       \
  switch (lpr) {                              \
    case 1: DCTR_UPD_LAUNCH(8, 1); break;     \
    case 2: DCTR_UPD_LAUNCH(8, 2); break;     \
    case 4: DCTR_UPD_LAUNCH(8, 4); break;     \
    default: DCTR_UPD_LAUNCH(8, 8); break;    \
  }
  if (vec == 8) { DCTR_UPD_LPR8() } else if (vec == 4) { DCTR_UPD_LPR(4) } else if (vec == 2) { DCTR_UPD_LPR(2) } else { DCTR_UPD_LPR(1) }
  return launch_status();
}

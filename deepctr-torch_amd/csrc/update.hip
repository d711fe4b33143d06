// update.hip -- deterministic, atomic-free embedding backward with the optimizer fused in (gfx950).
//
// Replaces autograd's aten::embedding_dense_backward x (n_deep + n_wide) + FM's backward + the dense
// optimizer walk over every table (basemodel.py:261-262, interaction.py:26-34) for plans made of
// fixed-length fields over distinct tables -- the Criteo shape.  (Plans with pooled VarLen fields or
// shared tables keep the atomic two-pass kernels of embed.hip.)
//
// Why not atomics: measured on MI355X at B=4096 (profiles/r01_*), the scatter (1.7 M dword atomics)
// took 36 us and the xchg-consume pass 59 us -- 9 % of the HBM roofline -- and float atomics make
// duplicate-row sums order-dependent, so replicas / shards drift apart.
//
// Mapping: a "unit" is one id column of X with the deep table and/or the wide (1-dim) table it feeds.
// Workgroup (unit u, partition p) owns the rows {id : id mod P == p} of u's tables, so no two workgroups
// ever touch the same row.  P is chosen so that a partition holds ~64 entries.  A workgroup
//   1. scans  the unit's B ids (ids_t[u][0..B), contiguous int32 written by the forward kernel) and collects
//             its entries as 32-bit keys (id / P) << bbits | b in LDS -- at most kCap of them;
//   2. if they fit one tile (<= G entries, the common case): every lane group takes one entry in SCAN order
//             and issues its loads at once (gradient strip, sum_f e strip, table + state row strips); the
//             rank sort of the keys runs in the shadow of those loads; gradients are parked in LDS at their
//             SORTED position; the last entry of every id segment sums its segment backwards and
//             read-modify-writes the row strips it already holds;
//      else if they fit kCap: bitonic / rank sort, then tiles of G sorted entries with a carry;
//   3. a partition with MORE than kCap entries (skewed ids) is split by the next bit of id / P and the two
//             halves are processed one after the other (re-scanning the ids; LDS stays small); a half that still
//             overflows but holds ONE id -- a hot id -- is summed by streaming over the batch in sample order.
// FM's backward is folded algebraically: sum_seg [g + gf (S - e)] = sum_seg (g + gf S) - (sum_seg gf) e, and e IS
// the table row about to be updated -- the forward's copy of it is not re-read.
// Every row is read-modify-written exactly once, by one lane group, in an order that depends only on the data
// (counts, ids, sample indices): results are bit-reproducible run to run and rank to rank.
//
// LDS per workgroup is ~12 KB whatever B is (the previous revision kept B keys -- 16 KB at B = 4096: six
// workgroups per CU and a second round; 128 KB at the 32 768-sample global batch of 8-GPU sharded training).
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kThreads = 256;
constexpr int kCap = 512;     // sort keys held in LDS
constexpr int kStack = 40;    // pending (id bits fixed, their value) splits of an overflowing partition
constexpr int kBucket = 512;  // keys per (unit, partition) bucket of the pre-pass (<= kCap; 2 per thread of its sort)
constexpr int kChunk = 4096;  // most samples of a unit one workgroup of the two-level pre-pass re-orders in LDS
constexpr int kFineMax = 16;  // most consecutive partitions per coarse bin (one workgroup of the second level sorts them all)
constexpr int kSortT = 256;   // threads of a second-level workgroup
constexpr int kBinsMax = 4096;   // coarse bins per unit (B <= 2^20: P <= 10923, 2731 bins)

struct UpdArgs {
  const dctr_field_t* deep;
  const dctr_field_t* wide;
  const int32_t* units;  // [n_units][4] = {deep index | -1, wide index | -1, X column, 0}
  const int32_t* ids_t;  // [n_units][B] truncated ids
  const uint16_t* parts_t;  // [n_units][B] clamp(id) mod P, written next to ids_t by the forward (nullable)
  const float* gout;     // [B, ldg]   d loss / d out (deep slices), nullable
  const float* fm_s;     // [B, lds_]  S[b, :] = sum_f e[b, f, :], needed with gfm
  const float* gfm;      // [B] nullable
  const float* gwide;    // [B] (stride ldgw) nullable
  int64_t ldgw;
  int64_t ldg, lds_;
  int32_t n_units, B, P, bbits;   // P partitions per unit (any positive number)
  uint64_t pmagic;                // floor(2^pshift / P) + 1: id / P == (id * pmagic) >> pshift for 0 <= id < 2^31
  int32_t pshift;
  float lr, eps;
  // optional extra role (last blocks): d loss / d Linear.weight = X_dense^T g_wide  (basemodel.py:88-90)
  const float* X;
  int64_t ldx;
  const int32_t* wdense_cols;
  int32_t n_wdense;
  float* g_wdense;
  DenseStepDev wd_step;       // kind >= 0: the extra workgroups also step Linear.weight
  unsigned long long* trace;  // diagnostics (tools/upd_trace.py): 8 timestamps per workgroup, or NULL
  // optional pre-bucketed entries (k_bucket): bcnt [n_units * P] (zero at rest), bkeys [n_units * P][kBucket]
  int32_t* bcnt;
  uint32_t* bkeys;
  int32_t presorted;   // the buckets hold keys already sorted by (id, sample) (dctr_embed_segments)
  // two-level pre-pass of large batches (k_prepass_bin / k_prepass_sort): every chunk of kChunk samples of a unit
  // re-ordered by coarse bin (`fine` consecutive partitions), with the chunk's bin offsets
  uint32_t* stage_keys;   // [n_units][n_chunks * chunk]
  uint16_t* stage_tags;   // [n_units][n_chunks * chunk] partition of the entry
  int32_t* stage_offs;    // [n_units][n_chunks][n_bins + 1]
  int32_t n_chunks, n_bins, chunk;   // chunk: samples per level-1 workgroup (a power of two <= kChunk)
  int32_t fine;                      // partitions per coarse bin (<= kFineMax)
};

// Diagnostics (per-workgroup phase stamps, partition override) exist only in the DCTR_DIAG build
// (`make diag` -> libdctr_hip_diag.so, used by tools/upd_trace.py): the shipped library keeps no mutable
// global state (include/dctr.h: re-entrant).
#ifdef DCTR_DIAG
unsigned long long* g_trace = nullptr;  // host-side: set by dctr_dbg_update_trace
int g_force_p = -1;
#define DCTR_TRACE(slot)                                                           \
  do {                                                                             \
    if (A.trace && tid == 0) A.trace[blockIdx.x * 8ull + (slot)] = wall_clock64(); \
  } while (0)
#else
#define DCTR_TRACE(slot) do { } while (0)
#endif

// Values that are the same for every lane of the workgroup (descriptor fields fetched through a pointer the
// compiler cannot prove uniform): pin them to scalar registers, 64-byte descriptors otherwise cost ~30 VGPRs.
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32));
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
  return reinterpret_cast<T*>(uni(static_cast<int64_t>(reinterpret_cast<uintptr_t>(p))));
}
__device__ __forceinline__ dctr_field_t uni_field(const dctr_field_t& f) {
  dctr_field_t r;
  r.table = uni(f.table);
  r.gacc = uni(f.gacc);
  r.state = uni(f.state);
  r.vocab = uni(f.vocab);
  r.dim = uni(f.dim);
  r.col = 0;
  r.len = 1;
  r.pool = 0;
  r.len_col = -1;
  r.out_off = uni(f.out_off);
  r.ld = uni(f.ld);
  r.ld_state = uni(f.ld_state);
  return r;
}

// id / P and id % P for a runtime P through a host-computed reciprocal (exact for 0 <= id < 2^31)
__device__ __forceinline__ uint32_t div_p(uint32_t id, uint64_t magic, int shift) {
  const uint64_t lo = (magic & 0xFFFFFFFFull) * id, hi = (magic >> 32) * id;
  return static_cast<uint32_t>(((lo >> 32) + hi) >> (shift - 32));
}

__device__ __forceinline__ int32_t clamp_id(int32_t id, int64_t vocab) {
  return (static_cast<uint64_t>(static_cast<int64_t>(id)) >= static_cast<uint64_t>(vocab)) ? 0 : id;
}

// One optimizer step on a strip of a row.  OPT: 0 SGD, 1 Adagrad, 2 accumulate into gacc.
// off_w / off_s / off_g: float offsets of the strip in the table, the state slab and the (contiguous) gacc slab.
template <int VEC, int OPT>
__device__ __forceinline__ void apply_strip(const dctr_field_t& fd, int64_t off_w, int64_t off_s, int64_t off_g,
                                            const Strip<VEC>& G, const Strip<VEC>& w, const Strip<VEC>& s, float lr,
                                            float eps) {
  Strip<VEC> nw, ns;
  if (OPT == DCTR_UPD_ADAGRAD) {  // torch.optim.Adagrad: s += g*g ; p -= lr * g / (sqrt(s) + eps)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      ns.v[i] = s.v[i] + G.v[i] * G.v[i];
      nw.v[i] = w.v[i] - lr * (G.v[i] / (sqrtf(ns.v[i]) + eps));
    }
    strip_store<VEC>(fd.state + off_s, ns);
    strip_store<VEC>(fd.table + off_w, nw);
  } else if (OPT == DCTR_UPD_SGD) {  // torch.optim.SGD: p -= lr * g
#pragma unroll
    for (int i = 0; i < VEC; ++i) nw.v[i] = w.v[i] - lr * G.v[i];
    strip_store<VEC>(fd.table + off_w, nw);
  } else {  // dense-gradient semantics: gacc[row] += g   (w holds the gacc strip)
#pragma unroll
    for (int i = 0; i < VEC; ++i) nw.v[i] = w.v[i] + G.v[i];
    strip_store<VEC>(fd.gacc + off_g, nw);
  }
}

// the dense half of Linear (basemodel.py:86-90): g_w[j] = sum_b g_wide[b] * X[b, col_j].  One extra workgroup per
// dense column, hidden behind the row updates; per-thread partial sums over a strided row set, then a fixed-order
// tree => deterministic.
__device__ __forceinline__ void wdense_column(const UpdArgs& A, int j) {
  __shared__ float red[kThreads / 64];
  const int tid = threadIdx.x;
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
    dense_step_apply(A.wd_step, A.g_wdense + j, t);
  }
}

// One (unit, partition): scan (or take the bucket), sort, segment sums, one read-modify-write per touched row.
// ---- where a tile's segments start, without a dependent LDS walk ---------------------------------------------------
// Every wave publishes which of its groups end a segment (one ballot, before the tile's barrier); a summing group then
// finds the first entry of its own segment from the flags strictly below it.  The walk over the segment becomes a
// counted loop whose LDS reads do not depend on each other -- what a hot id (hundreds of entries of one row: Zipf
// ids) needs; the order of the additions is unchanged.
template <int LPR>
__device__ __forceinline__ void publish_tails(unsigned long long* tails, bool tail, int tid) {
  const unsigned long long m = __ballot(tail && (tid % LPR == 0));
  if ((tid & 63) == 0) tails[tid >> 6] = m;
}
// The walk itself: entries grp, grp-1, ..., j0 of the tile, added in exactly that order (the result must not depend on
// which path summed it).  Branch-free and in small blocks whose LDS reads are issued together: one LDS latency per
// block instead of three per entry -- a hot id's segment fills whole tiles, and one group walks each of them alone.
// (Lanes past the row's width / other than lane 0 of the group sum values nobody reads.)
template <int VEC>
__device__ __forceinline__ void seg_walk(const float* gbuf, const float* gfbuf, const float* gwbuf, int RW, int e0,
                                         int grp, int j0, Strip<VEC>& acc, float& accf, float& accw) {
  constexpr int NB = VEC >= 8 ? 2 : 4;   // (register budget of the 6-workgroups-per-CU kernel)
  int jj = grp;
  for (; jj - (NB - 1) >= j0; jj -= NB) {
    Strip<VEC> t[NB];
    float tf[NB], tw[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) t[q].v[k] = gbuf[(jj - q) * RW + e0 + k];
      tf[q] = gfbuf[jj - q];
      tw[q] = gwbuf[jj - q];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc.v[k] += t[q].v[k];
      accf += tf[q];
      accw += tw[q];
    }
  }
  for (; jj >= j0; --jj) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc.v[k] += gbuf[jj * RW + e0 + k];
    accf += gfbuf[jj];
    accw += gwbuf[jj];
  }
}

template <int LPR>
__device__ __forceinline__ int seg_first(const unsigned long long* tails, int tid) {
  int wv = tid >> 6;
  const int lane0 = (tid & 63) / LPR * LPR;
  unsigned long long m = tails[wv] & ((1ull << lane0) - 1ull);
  while (m == 0ull && wv > 0) m = tails[--wv];
  return m ? wv * (64 / LPR) + (63 - __clzll(m)) / LPR + 1 : 0;
}

template <int VEC, int LPR, int OPT>
__device__ __forceinline__ void upd_partition(const UpdArgs& A, const int u, const int p) {
  constexpr int G = kThreads / LPR;   // lane groups per workgroup = entries per tile (a power of two)
  constexpr int RW = LPR * VEC;       // floats of one parked gradient row
  __shared__ uint32_t keys[kCap];     // this pass's entries; sorted in place by the tiled path
  __shared__ uint32_t skeys[G];       // single-tile path: keys in sorted order; streaming path: sample list
  __shared__ __align__(16) float gbuf[G * RW];  // gradient tile
  __shared__ float gfbuf[G];          // g_fm of the tile's entries
  __shared__ float gwbuf[G];          // wide gradient of the tile's entries
  __shared__ float carry[RW + 4];     // open segment of the tiled path: deep strip | g_fm sum | wide sum
  __shared__ int stack[kStack][2];
  __shared__ int n_sh, mn_sh, mx_sh, carry_id, sp_sh, wcnt[kThreads / 64];
  __shared__ unsigned long long tails[kThreads / 64];  // per wave: groups of the tile that end a segment
  const int tid = threadIdx.x;
  const int P = A.P;
  DCTR_TRACE(0);

  const int32_t* un = A.units + 4 * u;
  const int di = uni(un[0]), wi = uni(un[1]);
  dctr_field_t fd = {}, fw = {};
  if (di >= 0) fd = uni_field(A.deep[di]);
  if (wi >= 0) fw = uni_field(A.wide[wi]);
  const int64_t vocab = (di >= 0) ? fd.vocab : fw.vocab;
  const int B = A.B;
  const int32_t* ids = A.ids_t + static_cast<int64_t>(u) * B;

  const int grp = tid / LPR, gl = tid % LPR, e0 = gl * VEC;
  const uint32_t bmask = (1u << A.bbits) - 1u;
  const bool deep_on = (di >= 0) && (A.gout || A.gfm);
  const bool wide_on = (wi >= 0) && A.gwide;
  const bool lane_on = deep_on && (e0 < fd.dim);
  const int goff = deep_on ? fd.out_off + (lane_on ? e0 : 0) : 0;
  const bool fold = (A.gfm != nullptr);
  // row strides: a table and its Adagrad state may be strided views of one interleaved slab (dctr.h)
  const int64_t ld_dw = (di >= 0) ? row_ld(fd) : 1, ld_ds = (di >= 0) ? state_ld(fd) : 1;
  const int64_t ld_ww = (wi >= 0) ? row_ld(fw) : 1, ld_ws = (wi >= 0) ? state_ld(fw) : 1;

  // everything an entry contributes: h = g_out + g_fm * S (deep strip), g_fm, g_wide
  auto load_entry = [&](int b, Strip<VEC>& h, float& gf, float& gw) {
    h = strip_zero<VEC>();
    gf = 0.f;
    gw = 0.f;
    if (lane_on) {
      if (A.gout) h = strip_load<VEC>(A.gout + static_cast<int64_t>(b) * A.ldg + goff);
      if (fold) {
        const Strip<VEC> S = strip_load<VEC>(A.fm_s + static_cast<int64_t>(b) * A.lds_ + e0);
        gf = ldg_f32(A.gfm + b);
#pragma unroll
        for (int k = 0; k < VEC; ++k) h.v[k] += gf * S.v[k];
      }
    }
    if (wide_on && gl == 0) gw = ldg_f32(A.gwide + static_cast<int64_t>(b) * A.ldgw);
  };
  // the strips of a row this lane may update: w (table, or gacc in accumulate mode), s (Adagrad state), e (the
  // table strip FM's fold needs; = w unless accumulating)
  auto load_row = [&](int64_t row, Strip<VEC>& w, Strip<VEC>& s, Strip<VEC>& e, float& ww, float& sw) {
    w = strip_zero<VEC>();
    s = strip_zero<VEC>();
    e = strip_zero<VEC>();
    ww = 0.f;
    sw = 0.f;
    if (lane_on) {
      const int64_t off_w = row * ld_dw + e0;
      w = strip_load<VEC>(OPT == DCTR_UPD_ACCUM ? fd.gacc + row * fd.dim + e0 : fd.table + off_w);
      if (OPT == DCTR_UPD_ADAGRAD) s = strip_load<VEC>(fd.state + row * ld_ds + e0);
      if (OPT == DCTR_UPD_ACCUM) {
        if (fold) e = strip_load<VEC>(fd.table + off_w);
      } else {
        e = w;
      }
    }
    if (wide_on && gl == 0) {
      ww = ldg_f32(OPT == DCTR_UPD_ACCUM ? fw.gacc + row : fw.table + row * ld_ww);
      if (OPT == DCTR_UPD_ADAGRAD) sw = ldg_f32(fw.state + row * ld_ws);
    }
  };
  auto apply_row = [&](int64_t row, Strip<VEC> acc, float accf, float accw, const Strip<VEC>& w, const Strip<VEC>& s,
                       const Strip<VEC>& e, float ww, float sw) {
    if (lane_on) {
      if (fold) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc.v[k] -= accf * e.v[k];
      }
      apply_strip<VEC, OPT>(fd, row * ld_dw + e0, row * ld_ds + e0, row * fd.dim + e0, acc, w, s, A.lr, A.eps);
    }
    if (wide_on && gl == 0) {
      Strip<1> a1, w1, s1;
      a1.v[0] = accw;
      w1.v[0] = ww;
      s1.v[0] = sw;
      apply_strip<1, OPT>(fw, row * ld_ww, row * ld_ws, row, a1, w1, s1, A.lr, A.eps);
    }
  };

  if (tid == 0) {
    sp_sh = 1;
    stack[0][0] = 0;   // number of extra id bits fixed
    stack[0][1] = 0;   // their value
  }
  bool first_pass = true;
  for (;;) {
    __syncthreads();
    const int sp = sp_sh;
    if (sp == 0) break;
    const int mbits = stack[sp - 1][0], mres = stack[sp - 1][1];
    __syncthreads();
    if (tid == 0) {
      sp_sh = sp - 1;
      n_sh = 0;
      mn_sh = 0x7FFFFFFF;
      mx_sh = -1;
      carry_id = -1;
    }
    __syncthreads();

    const uint32_t mmask = (1u << mbits) - 1u;
    bool bucketed = false;
    if (first_pass && A.bcnt) {
      // the pre-pass (k_bucket) already collected this partition's keys: no scan over the unit's B ids.  The counter
      // is left at zero for the next launch.  A bucket that overflowed falls back to the scan.
      int32_t* cnt = A.bcnt + static_cast<int64_t>(u) * P + p;
      if (tid == 0) {
        const int nb = *(DCTR_GLOBAL int32_t*)cnt;
        *(DCTR_GLOBAL int32_t*)cnt = 0;
        n_sh = nb <= kBucket ? nb : -1;
      }
      __syncthreads();
      const int nb = n_sh;
      if (nb >= 0) {
        const uint32_t* src = A.bkeys + (static_cast<int64_t>(u) * P + p) * kBucket;
        for (int i = tid; i < nb; i += kThreads) keys[i] = *(const DCTR_GLOBAL uint32_t*)(src + i);
        bucketed = true;
      } else if (tid == 0) {
        n_sh = 0;
      }
      __syncthreads();
    }
    // ---- scan: collect the entries of (partition p, id/P mod 2^mbits == mres) ------------------------------------
    // All id loads of a chunk are issued before any is consumed: the scan costs one L2 round trip per chunk.
    if (!bucketed && A.parts_t) {
      // The forward stored clamp(id) mod P next to every id: the scan is a 16-bit compare per entry (a workgroup
      // keeps ~1/P of them), the exact division runs only for the entries kept.  Was: a 64-bit reciprocal multiply
      // per id, per workgroup -- 5.5 us of quarter-rate integer multiplies at B = 4096 (phase trace, round 1).
      const uint16_t* pt = A.parts_t + static_cast<int64_t>(u) * B;
      const uint32_t pp = static_cast<uint32_t>(p);
      auto keep = [&](int b) {
        const int32_t id = clamp_id(ldg_i32(ids + b), vocab);
        const uint32_t idq = div_p(static_cast<uint32_t>(id), A.pmagic, A.pshift);
        if ((idq & mmask) == static_cast<uint32_t>(mres)) {
          const int slot = atomicAdd(&n_sh, 1);  // LDS atomic; the order is fixed by the sort below
          if (slot < kCap) keys[slot] = (idq << A.bbits) | static_cast<uint32_t>(b);
        }
      };
      if ((B & 7) == 0) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const DCTR_GLOBAL u32x4* pv = (const DCTR_GLOBAL u32x4*)pt;
        const int nvec = B >> 3;
        for (int c0 = 0; c0 < nvec; c0 += 2 * kThreads) {
          u32x4 v[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int idx = c0 + q * kThreads + tid;
            v[q] = pv[idx < nvec ? idx : 0];
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int idx = c0 + q * kThreads + tid;
            if (idx < nvec) {
              const uint32_t d[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if ((d[j] & 0xFFFFu) == pp) keep(8 * idx + 2 * j);
                if ((d[j] >> 16) == pp) keep(8 * idx + 2 * j + 1);
              }
            }
          }
        }
      } else {
        for (int b = tid; b < B; b += kThreads)
          if (static_cast<uint32_t>(*(const DCTR_GLOBAL uint16_t*)(pt + b)) == pp) keep(b);
      }
    } else if (!bucketed) {
    auto take = [&](int32_t raw, int b) {
      const int32_t id = clamp_id(raw, vocab);
      const uint32_t idq = div_p(static_cast<uint32_t>(id), A.pmagic, A.pshift);
      if (static_cast<int>(static_cast<uint32_t>(id) - idq * static_cast<uint32_t>(P)) == p &&
          (idq & mmask) == static_cast<uint32_t>(mres)) {
        const int slot = atomicAdd(&n_sh, 1);  // LDS atomic; the order is fixed by the sort below
        if (slot < kCap) keys[slot] = (idq << A.bbits) | static_cast<uint32_t>(b);
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
    }
    __syncthreads();
    const int n = n_sh;
    if (first_pass) {
      DCTR_TRACE(1);
#ifdef DCTR_DIAG
      if (A.trace && tid == 0) A.trace[blockIdx.x * 8ull + 7] = static_cast<unsigned long long>(n);
#endif
    }
    if (n == 0) {
      first_pass = false;
      continue;
    }

    const bool sorted_in = bucketed && A.presorted;   // (the pre-pass also sorted: straight to the tiles)
    if (n <= G && !sorted_in) {
      // ---- single tile --------------------------------------------------------------------------------------------
      const bool have = grp < n;
      const uint32_t key = have ? keys[grp] : 0xFFFFFFFFu;
      const int b = static_cast<int>(key & bmask);
      const int idq = static_cast<int>(key >> A.bbits);
      const int64_t row = static_cast<int64_t>(idq) * P + p;
      Strip<VEC> h, w, s, e;
      float gf, gw, ww, sw;
      if (have) {
        load_entry(b, h, gf, gw);
        load_row(row, w, s, e, ww, sw);
      } else {
        h = w = s = e = strip_zero<VEC>();
        gf = gw = ww = sw = 0.f;
      }
      int rank = 0;  // keys are unique: rank = number of smaller keys
#pragma unroll 8
      for (int q = 0; q < n; ++q) rank += (keys[q] < key) ? 1 : 0;
      if (first_pass) DCTR_TRACE(2);
      if (have) {
        if (gl == 0) {
          skeys[rank] = key;
          gfbuf[rank] = gf;
          gwbuf[rank] = gw;
        }
        if (lane_on) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) gbuf[rank * RW + e0 + k] = h.v[k];
        }
      }
      if (first_pass) DCTR_TRACE(3);
      __syncthreads();
      if (first_pass) DCTR_TRACE(4);
      if (have) {
        const bool seg_end = (rank == n - 1) || (static_cast<int>(skeys[rank + 1] >> A.bbits) != idq);
        if (seg_end) {
          Strip<VEC> acc = strip_zero<VEC>();
          float accf = 0.f, accw = 0.f;
          int r = rank;  // walk back: fixed order => deterministic
          while (r >= 0 && static_cast<int>(skeys[r] >> A.bbits) == idq) {
            if (lane_on) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc.v[k] += gbuf[r * RW + e0 + k];
            }
            accf += gfbuf[r];
            if (gl == 0) accw += gwbuf[r];
            --r;
          }
          apply_row(row, acc, accf, accw, w, s, e, ww, sw);
        }
      }
      if (first_pass) {
        DCTR_TRACE(5);
        DCTR_TRACE(6);
      }
      first_pass = false;
      continue;
    }

    if (n <= kCap) {
      // ---- sort by (id, b), then tiles of G sorted entries with a carry ---------------------------------------------
      if (sorted_in) {
        // nothing to do
      } else {
        // rank sort: keys are unique, so rank = #smaller is a permutation; 2 barriers (kCap / kThreads keys a thread)
        constexpr int kSl = kCap / kThreads;
        uint32_t mine[kSl];
        int rank[kSl];
#pragma unroll
        for (int q = 0; q < kSl; ++q) {
          mine[q] = tid + q * kThreads < n ? keys[tid + q * kThreads] : 0u;
          rank[q] = 0;
        }
#pragma unroll 4
        for (int i = 0; i < n; ++i) {
          const uint32_t kv = keys[i];
#pragma unroll
          for (int q = 0; q < kSl; ++q) rank[q] += (kv < mine[q]) ? 1 : 0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kSl; ++q)
          if (tid + q * kThreads < n) keys[rank[q]] = mine[q];
        __syncthreads();
      }
      for (int t0 = 0; t0 < n; t0 += G) {
        const int i = t0 + grp;
        const bool have = i < n;
        const uint32_t key = have ? keys[i] : 0u;
        const int b = static_cast<int>(key & bmask);
        const int idq = static_cast<int>(key >> A.bbits);
        const int64_t row = static_cast<int64_t>(idq) * P + p;
        const bool last_of_tile = have && ((grp == G - 1) || (i == n - 1));
        const bool seg_end = have && ((i == n - 1) || (static_cast<int>(keys[i + 1] >> A.bbits) != idq));
        Strip<VEC> h, w, s, e;
        float gf, gw, ww, sw;
        h = w = s = e = strip_zero<VEC>();
        gf = gw = ww = sw = 0.f;
        if (have) {
          load_entry(b, h, gf, gw);
          if (seg_end) load_row(row, w, s, e, ww, sw);
        }
        if (lane_on) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) gbuf[grp * RW + e0 + k] = h.v[k];
        }
        if (gl == 0) {
          gfbuf[grp] = gf;
          gwbuf[grp] = gw;
        }
        publish_tails<LPR>(tails, seg_end, tid);
        __syncthreads();
        const bool summer = seg_end || last_of_tile;
        Strip<VEC> acc = strip_zero<VEC>();
        float accf = 0.f, accw = 0.f;
        if (summer) {
          const int j0 = seg_first<LPR>(tails, tid);
          seg_walk<VEC>(gbuf, gfbuf, gwbuf, RW, e0, grp, j0, acc, accf, accw);  // fixed order => deterministic
          if (j0 == 0 && carry_id == idq) {  // the segment began in an earlier tile
            if (lane_on) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc.v[k] += carry[e0 + k];
            }
            accf += carry[RW];
            if (gl == 0) accw += carry[RW + 1];
          }
          if (seg_end) apply_row(row, acc, accf, accw, w, s, e, ww, sw);
        }
        __syncthreads();  // every read of gbuf / carry of this tile is done
        if (last_of_tile) {  // exactly one group: park an open segment's partial, or clear the carry
          if (!seg_end) {
            if (lane_on) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) carry[e0 + k] = acc.v[k];
            }
            if (gl == 0) {
              carry[RW] = accf;
              carry[RW + 1] = accw;
              carry_id = idq;
            }
          } else if (gl == 0) {
            carry_id = -1;
          }
        }
        __syncthreads();
      }
      first_pass = false;
      continue;
    }

    // ---- more entries than LDS keys ----------------------------------------------------------------------------
    first_pass = false;
    {  // the id range of this pass decides what happens next (a second scan: only overflowing passes pay for it)
      int lo = 0x7FFFFFFF, hi = -1;
      for (int b = tid; b < B; b += kThreads) {
        const int32_t id = clamp_id(ldg_i32(ids + b), vocab);
        const uint32_t idq = div_p(static_cast<uint32_t>(id), A.pmagic, A.pshift);
        if (static_cast<int>(static_cast<uint32_t>(id) - idq * static_cast<uint32_t>(P)) == p &&
            (idq & mmask) == static_cast<uint32_t>(mres)) {
          lo = min(lo, static_cast<int>(idq));
          hi = max(hi, static_cast<int>(idq));
        }
      }
#pragma unroll
      for (int m2 = 32; m2 >= 1; m2 >>= 1) {
        lo = min(lo, __shfl_xor(lo, m2, kWave));
        hi = max(hi, __shfl_xor(hi, m2, kWave));
      }
      if ((tid & 63) == 0) {
        atomicMin(&mn_sh, lo);
        atomicMax(&mx_sh, hi);
      }
      __syncthreads();
    }
    if (mn_sh != mx_sh) {
      // several ids: fix one more bit of id / P and do the two halves one after the other
      if (tid == 0) {
        const int spn = sp_sh;
        if (spn + 2 <= kStack) {
          stack[spn][0] = mbits + 1;
          stack[spn][1] = mres | (1 << mbits);
          stack[spn + 1][0] = mbits + 1;
          stack[spn + 1][1] = mres;
          sp_sh = spn + 2;
        }
      }
      continue;
    }
    // a hot id: all n > kCap entries hit ONE row.  Stream over the batch in sample order, G matching samples at a
    // time; each tile is reduced by a fixed tree and added to the running sum kept by lane group 0.
    const int idq_hot = mn_sh;
    const int64_t row_hot = static_cast<int64_t>(idq_hot) * P + p;
    const int32_t id_hot = static_cast<int32_t>(row_hot);
    Strip<VEC> tot = strip_zero<VEC>();
    float totf = 0.f, totw = 0.f;
    auto flush_tile = [&](int cnt) {   // cnt sample indices sit in skeys[0..cnt)
      Strip<VEC> h;
      float gf, gw;
      if (grp < cnt) {
        load_entry(static_cast<int>(skeys[grp]), h, gf, gw);
      } else {
        h = strip_zero<VEC>();
        gf = gw = 0.f;
      }
      if (lane_on) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) gbuf[grp * RW + e0 + k] = h.v[k];
      }
      if (gl == 0) {
        gfbuf[grp] = gf;
        gwbuf[grp] = gw;
      }
      __syncthreads();
      for (int st = G >> 1; st > 0; st >>= 1) {
        if (grp < st) {
          if (lane_on) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) gbuf[grp * RW + e0 + k] += gbuf[(grp + st) * RW + e0 + k];
          }
          if (gl == 0) {
            gfbuf[grp] += gfbuf[grp + st];
            gwbuf[grp] += gwbuf[grp + st];
          }
        }
        __syncthreads();
      }
      if (grp == 0) {
        if (lane_on) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) tot.v[k] += gbuf[e0 + k];
        }
        totf += gfbuf[0];
        if (gl == 0) totw += gwbuf[0];
      }
      __syncthreads();
    };
    int pending = 0;   // matches parked in skeys[] (uniform over the workgroup)
    for (int c0 = 0; c0 < B; c0 += kThreads) {
      const int b = c0 + tid;
      const bool m = b < B && clamp_id(ldg_i32(ids + (b < B ? b : 0)), vocab) == id_hot;
      const unsigned long long bal = __ballot(m);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) wcnt[wv] = __popcll(bal);
      __syncthreads();
      int before = 0, total = 0;
      for (int w2 = 0; w2 < kThreads / 64; ++w2) {
        const int cw = wcnt[w2];
        if (w2 < wv) before += cw;
        total += cw;
      }
      const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));   // rank of this match in sample order
      int done = 0;  // matches of this chunk already parked
      while (done < total) {
        const int room = G - pending;
        const int takec = (total - done) < room ? (total - done) : room;
        if (m && pos >= done && pos < done + takec) skeys[pending + pos - done] = static_cast<uint32_t>(b);
        __syncthreads();
        pending += takec;
        done += takec;
        if (pending == G) {
          flush_tile(G);
          pending = 0;
        }
      }
      __syncthreads();   // wcnt is rewritten by the next chunk
    }
    if (pending > 0) flush_tile(pending);
    if (grp == 0) {
      Strip<VEC> w, s, e;
      float ww, sw;
      load_row(row_hot, w, s, e, ww, sw);
      apply_row(row_hot, tot, totf, totw, w, s, e, ww, sw);
    }
  }
}

// (5 workgroups of 4 waves per CU = 1280 resident, more than the ~1100 a launch has at batch 4096: <= 96 VGPRs, no
// spills; tighter bounds spill 8-40 registers in the Adagrad variant = +8 MB of scratch writes per launch)
template <int VEC, int LPR, int OPT>
__global__ __launch_bounds__(kThreads, 5) void k_embed_update(UpdArgs A) {
  if (A.g_wdense && static_cast<int>(blockIdx.x) >= static_cast<int>(gridDim.x) - A.n_wdense) {
    wdense_column(A, static_cast<int>(blockIdx.x) - (static_cast<int>(gridDim.x) - A.n_wdense));
    return;
  }
  // Work item (unit, partition) in plain launch order: consecutive workgroups go to consecutive XCDs, so every
  // XCD gets the same number of working workgroups.
  const int u = static_cast<int>(blockIdx.x) / A.P, p = static_cast<int>(blockIdx.x) - u * A.P;
  if (u >= A.n_units) return;
  upd_partition<VEC, LPR, OPT>(A, u, p);
}

// ---- the update proper, given the pre-pass's sorted keys ------------------------------------------------------------
// What is left on the step's critical chain once dctr_embed_segments has run: TWO memory round trips per workgroup --
// {entry count, sorted keys} (one coalesced read), then {gradient strips, row strips} -- a segmented sum through LDS
// and one read-modify-write per touched row.  No scan, no sort, no stack: 8 workgroups per CU (the general kernel
// below: 5), so a saturating launch keeps ~1.6x more rows in flight.  Same tiling, same summation order as the general
// kernel's tiled path: bit-identical results.  A partition whose count exceeds kBucket (a hot id: the pre-pass could not
// sort it) takes the general path right here (scan, split by id bits, streaming of a hot id), and the last n_wdense
// workgroups of the launch do the dense half of Linear.  (Both used to be a second launch, k_embed_update_overflow: 8 us
// of launch + boundary + counter round trip on the step's critical chain for work that is almost always empty -- round 3.)
template <int VEC, int LPR, int OPT>
__global__ __launch_bounds__(kThreads, 5) void k_embed_apply_sorted(UpdArgs A) {
  constexpr int G = kThreads / LPR;   // entries per tile
  constexpr int RW = LPR * VEC;
  __shared__ unsigned long long tails[kThreads / 64];  // per wave: groups of the tile that end a segment
  __shared__ __align__(16) float gbuf[G * RW];
  __shared__ float gfbuf[G], gwbuf[G];
  __shared__ float carry[RW + 4];
  __shared__ int carry_id;
  const int tid = threadIdx.x;
  const int P = A.P;
  // the dense half of Linear goes FIRST in the grid: its workgroups sweep the whole batch (two round trips + a
  // reduction) and must not queue behind a second round of partition workgroups
  const int n_lin = A.g_wdense ? A.n_wdense : 0;
  if (static_cast<int>(blockIdx.x) < n_lin) {
    wdense_column(A, static_cast<int>(blockIdx.x));
    return;
  }
  const int wg = static_cast<int>(blockIdx.x) - n_lin;
  const int u = wg / P, p = wg - u * P;
  if (u >= A.n_units) return;
  int32_t* cnt = A.bcnt + static_cast<int64_t>(u) * P + p;
  const uint32_t* src = A.bkeys + (static_cast<int64_t>(u) * P + p) * kBucket;
  const int grp = tid / LPR, gl = tid % LPR, e0 = gl * VEC;
  DCTR_TRACE(0);
  // round trip 1: the count and the first tile's keys leave together (slots past the count hold stale keys of an
  // earlier step: inside the bucket's own kBucket slots, never used)
  const int n_raw = *(const DCTR_GLOBAL int32_t*)cnt;
  uint32_t key = *(const DCTR_GLOBAL uint32_t*)(src + (grp < kBucket ? grp : kBucket - 1));
  uint32_t knext = *(const DCTR_GLOBAL uint32_t*)(src + (grp + 1 < kBucket ? grp + 1 : kBucket - 1));

  const int32_t* un = A.units + 4 * u;
  const int di = uni(un[0]), wi = uni(un[1]);
  dctr_field_t fd = {}, fw = {};
  if (di >= 0) fd = uni_field(A.deep[di]);
  if (wi >= 0) fw = uni_field(A.wide[wi]);
  const uint32_t bmask = (1u << A.bbits) - 1u;
  const bool deep_on = (di >= 0) && (A.gout || A.gfm);
  const bool wide_on = (wi >= 0) && A.gwide;
  const bool lane_on = deep_on && (e0 < fd.dim);
  const int goff = deep_on ? fd.out_off + (lane_on ? e0 : 0) : 0;
  const bool fold = (A.gfm != nullptr);
  const int64_t ld_dw = (di >= 0) ? row_ld(fd) : 1, ld_ds = (di >= 0) ? state_ld(fd) : 1;
  const int64_t ld_ww = (wi >= 0) ? row_ld(fw) : 1, ld_ws = (wi >= 0) ? state_ld(fw) : 1;

  const int n = uni(n_raw);
  if (n <= 0) return;                  // (uniform: every thread read the same counter)
  if (n > kBucket) {                   // a hot partition: the general path (it takes the counter and re-scans the ids)
    upd_partition<VEC, LPR, OPT>(A, u, p);
    return;
  }
  DCTR_TRACE(1);
#ifdef DCTR_DIAG
  if (A.trace && tid == 0) A.trace[blockIdx.x * 8ull + 7] = static_cast<unsigned long long>(n);
#endif
  __syncthreads();                     // every wave has read the counter ...
  if (tid == 0) {
    *(DCTR_GLOBAL int32_t*)cnt = 0;    // ... before it is handed back zeroed for the next pre-pass
    carry_id = -1;
  }

  uint32_t key_n = 0u, knext_n = 0u;
  for (int t0 = 0; t0 < n; t0 += G) {
    const int i = t0 + grp;
    if (t0 + G < n) {  // the next tile's keys leave ahead of this tile's strips: one round trip per tile, not two
      const int i2 = i + G;
      key_n = *(const DCTR_GLOBAL uint32_t*)(src + (i2 < kBucket ? i2 : kBucket - 1));
      knext_n = *(const DCTR_GLOBAL uint32_t*)(src + (i2 + 1 < kBucket ? i2 + 1 : kBucket - 1));
    }
    const bool have = i < n;
    const int b = static_cast<int>(key & bmask);
    const int idq = static_cast<int>(key >> A.bbits);
    const int64_t row = static_cast<int64_t>(idq) * P + p;
    const bool seg_end = have && ((i == n - 1) || (static_cast<int>(knext >> A.bbits) != idq));
    const bool last_of_tile = have && ((grp == G - 1) || (i == n - 1));
    // round trip 2: everything this entry contributes, and (segment ends only) the row it lands on
    // (nothing is USED inside the branches that guard these loads: with `h += gf * S` right behind its loads the compiler
    // waited for them there, and the row loads below left one round trip later -- three dependent trips per tile where
    // two were meant; round 3)
    Strip<VEC> h = strip_zero<VEC>(), w = strip_zero<VEC>(), s = strip_zero<VEC>(), e = strip_zero<VEC>();
    Strip<VEC> S = strip_zero<VEC>();
    float gf = 0.f, gw = 0.f, ww = 0.f, sw = 0.f;
    if (have) {
      if (lane_on) {
        if (A.gout) h = strip_load<VEC>(A.gout + static_cast<int64_t>(b) * A.ldg + goff);
        if (fold) {
          S = strip_load<VEC>(A.fm_s + static_cast<int64_t>(b) * A.lds_ + e0);
          gf = ldg_f32(A.gfm + b);
        }
      }
      if (wide_on && gl == 0) gw = ldg_f32(A.gwide + static_cast<int64_t>(b) * A.ldgw);
      if (seg_end) {
        if (lane_on) {
          const int64_t off_w = row * ld_dw + e0;
          w = strip_load<VEC>(OPT == DCTR_UPD_ACCUM ? fd.gacc + row * fd.dim + e0 : fd.table + off_w);
          if (OPT == DCTR_UPD_ADAGRAD) s = strip_load<VEC>(fd.state + row * ld_ds + e0);
          if (OPT == DCTR_UPD_ACCUM) {
            if (fold) e = strip_load<VEC>(fd.table + off_w);
          } else {
            e = w;
          }
        }
        if (wide_on && gl == 0) {
          ww = ldg_f32(OPT == DCTR_UPD_ACCUM ? fw.gacc + row : fw.table + row * ld_ww);
          if (OPT == DCTR_UPD_ADAGRAD) sw = ldg_f32(fw.state + row * ld_ws);
        }
      }
    }
    if (lane_on) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) gbuf[grp * RW + e0 + k] = h.v[k] + gf * S.v[k];   // (gf = 0, S = 0 without FM)
    }
    if (gl == 0) {
      gfbuf[grp] = gf;
      gwbuf[grp] = gw;
    }
    publish_tails<LPR>(tails, seg_end, tid);
    __syncthreads();
    if (t0 == 0) DCTR_TRACE(2);
    const bool summer = seg_end || last_of_tile;
    Strip<VEC> acc = strip_zero<VEC>();
    float accf = 0.f, accw = 0.f;
    if (summer) {
      const int j0 = seg_first<LPR>(tails, tid);
      seg_walk<VEC>(gbuf, gfbuf, gwbuf, RW, e0, grp, j0, acc, accf, accw);  // (the general kernel's order)
      if (j0 == 0 && carry_id == idq) {  // the segment began in an earlier tile
        if (lane_on) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc.v[k] += carry[e0 + k];
        }
        accf += carry[RW];
        if (gl == 0) accw += carry[RW + 1];
      }
      if (seg_end) {
        if (lane_on) {
          Strip<VEC> a2 = acc;
          if (fold) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) a2.v[k] -= accf * e.v[k];
          }
          apply_strip<VEC, OPT>(fd, row * ld_dw + e0, row * ld_ds + e0, row * fd.dim + e0, a2, w, s, A.lr, A.eps);
        }
        if (wide_on && gl == 0) {
          Strip<1> a1, w1, s1;
          a1.v[0] = accw;
          w1.v[0] = ww;
          s1.v[0] = sw;
          apply_strip<1, OPT>(fw, row * ld_ww, row * ld_ws, row, a1, w1, s1, A.lr, A.eps);
        }
      }
    }
    if (t0 == 0) DCTR_TRACE(3);
    if (t0 + G >= n) break;   // single tile (the common case): no carry to park
    __syncthreads();  // every read of gbuf / carry of this tile is done
    if (last_of_tile) {  // exactly one group: park an open segment's partial, or clear the carry
      if (!seg_end) {
        if (lane_on) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) carry[e0 + k] = acc.v[k];
        }
        if (gl == 0) {
          carry[RW] = accf;
          carry[RW + 1] = accw;
          carry_id = idq;
        }
      } else if (gl == 0) {
        carry_id = -1;
      }
    }
    __syncthreads();
    key = key_n;
    knext = knext_n;
  }
  DCTR_TRACE(6);
}

// ---- optional pre-pass: bucket the (unit, sample) entries by partition ---------------------------------------------
// One thread per entry; a bucket's fill order is whatever the atomics give (the update kernel sorts the keys).
// Pays off when a workgroup's scan over the unit's B ids is the expensive part, i.e. for large (global) batches.
__global__ __launch_bounds__(kThreads) void k_bucket(UpdArgs A) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= static_cast<int64_t>(A.n_units) * A.B) return;
  const int u = static_cast<int>(i / A.B), b = static_cast<int>(i - static_cast<int64_t>(u) * A.B);
  const int32_t* un = A.units + 4 * u;
  const int di = un[0], wi = un[1];
  const int64_t vocab = (di >= 0) ? A.deep[di].vocab : A.wide[wi].vocab;
  const int32_t id = clamp_id(ldg_i32(A.ids_t + i), vocab);
  const uint32_t idq = div_p(static_cast<uint32_t>(id), A.pmagic, A.pshift);
  const int p = static_cast<int>(static_cast<uint32_t>(id) - idq * static_cast<uint32_t>(A.P));
  const int64_t bucket = static_cast<int64_t>(u) * A.P + p;
  const int slot = atomicAdd(A.bcnt + bucket, 1);
  if (slot < kBucket) A.bkeys[bucket * kBucket + slot] = (idq << A.bbits) | static_cast<uint32_t>(b);
}

// ---- pre-pass: every (unit, partition)'s entries, sorted by (id, sample), parked in the bucket arrays ---------------
// What a workgroup of k_embed_update otherwise does first -- find its entries among the unit's B, sort them -- depends
// on the ids alone, which exist as soon as the forward has run: this kernel does it THEN, in the shadow of the tower,
// and the update kernel (on the step's critical chain once the gradients exist) starts with one coalesced read of its
// sorted keys: no scan, no sort, no dependent id loads.  FROM_BUCKETS: k_bucket collected the keys (large batches).
// A partition with more than kBucket entries is left to the update kernel's own scan (the counter says so).
template <bool FROM_BUCKETS>
__global__ __launch_bounds__(kThreads) void k_embed_segments(UpdArgs A) {
  constexpr int kSl = kBucket / kThreads;   // bucket slots per thread
  static_assert(kBucket % kThreads == 0, "whole slots per thread");
  __shared__ uint32_t keys[kBucket];
  __shared__ int n_sh;
  const int tid = threadIdx.x;
  const int P = A.P;
  const int u = static_cast<int>(blockIdx.x) / P, p = static_cast<int>(blockIdx.x) - u * P;
  if (u >= A.n_units) return;
  int32_t* cnt = A.bcnt + static_cast<int64_t>(u) * P + p;
  uint32_t* dst = A.bkeys + (static_cast<int64_t>(u) * P + p) * kBucket;
  const int B = A.B;
  int n;
  if (FROM_BUCKETS) {
    n = uni(*(const DCTR_GLOBAL int32_t*)cnt);
    if (n > kBucket || n <= 0) return;
#pragma unroll
    for (int q = 0; q < kSl; ++q)
      if (tid + q * kThreads < n) keys[tid + q * kThreads] = *(const DCTR_GLOBAL uint32_t*)(dst + tid + q * kThreads);
    __syncthreads();
  } else {
    const int32_t* un = A.units + 4 * u;
    const int di = uni(un[0]), wi = uni(un[1]);
    const int64_t vocab = uni((di >= 0) ? A.deep[di].vocab : A.wide[wi].vocab);
    const int32_t* ids = A.ids_t + static_cast<int64_t>(u) * B;
    const uint16_t* pt = A.parts_t + static_cast<int64_t>(u) * B;
    const uint32_t pp = static_cast<uint32_t>(p);
    if (tid == 0) n_sh = 0;
    __syncthreads();
    // phase 1: the samples whose partition tag is mine (no dependent loads inside the divergent branches)
    auto keep = [&](int b) {
      const int slot = atomicAdd(&n_sh, 1);
      if (slot < kBucket) keys[slot] = static_cast<uint32_t>(b);
    };
    if ((B & 7) == 0) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const DCTR_GLOBAL u32x4* pv = (const DCTR_GLOBAL u32x4*)pt;
      const int nvec = B >> 3;
      for (int c0 = 0; c0 < nvec; c0 += 2 * kThreads) {
        u32x4 v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int idx = c0 + q * kThreads + tid;
          v[q] = pv[idx < nvec ? idx : 0];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int idx = c0 + q * kThreads + tid;
          if (idx < nvec) {
            const uint32_t d[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if ((d[j] & 0xFFFFu) == pp) keep(8 * idx + 2 * j);
              if ((d[j] >> 16) == pp) keep(8 * idx + 2 * j + 1);
            }
          }
        }
      }
    } else {
      for (int b = tid; b < B; b += kThreads)
        if (static_cast<uint32_t>(*(const DCTR_GLOBAL uint16_t*)(pt + b)) == pp) keep(b);
    }
    __syncthreads();
    n = n_sh;
    if (tid == 0) *(DCTR_GLOBAL int32_t*)cnt = n;
    if (n > kBucket || n == 0) return;
    // phase 2: one id load per kept entry, all in flight together
    uint32_t key[kSl];
#pragma unroll
    for (int q = 0; q < kSl; ++q) {
      key[q] = 0u;
      if (tid + q * kThreads < n) {
        const int b = static_cast<int>(keys[tid + q * kThreads]);
        const int32_t id = clamp_id(ldg_i32(ids + b), vocab);
        key[q] = (div_p(static_cast<uint32_t>(id), A.pmagic, A.pshift) << A.bbits) | static_cast<uint32_t>(b);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kSl; ++q)
      if (tid + q * kThreads < n) keys[tid + q * kThreads] = key[q];
    __syncthreads();
  }
  // rank sort: keys are unique, so rank = #smaller is a permutation (n reads per key: 4 us at 512 keys -- the general
  // kernel's bitonic sort of the same 512 keys is 45 barrier stages, ~45 us: what made Zipf-distributed ids slow)
  uint32_t mine[kSl];
  int rank[kSl];
#pragma unroll
  for (int q = 0; q < kSl; ++q) {
    mine[q] = tid + q * kThreads < n ? keys[tid + q * kThreads] : 0u;
    rank[q] = 0;
  }
#pragma unroll 4
  for (int i = 0; i < n; ++i) {
    const uint32_t kv = keys[i];
#pragma unroll
    for (int q = 0; q < kSl; ++q) rank[q] += (kv < mine[q]) ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < kSl; ++q)
    if (tid + q * kThreads < n) *(DCTR_GLOBAL uint32_t*)(dst + rank[q]) = mine[q];
}

// ---- two-level pre-pass of LARGE batches (global batches of the sharded trainer, saturating launches) -----------------
// What k_embed_segments computes -- every (unit, partition)'s keys sorted by (id, sample) in its bucket, the count beside it
// -- for batches where "every workgroup scans the unit's B tags" (B / 96 workgroups x B tags per unit: quadratic) or "one
// global atomic + one scattered 4-byte store per entry" (k_bucket; 540 us at B = 262 144) is the expensive part:
//   level 1  k_prepass_bin   workgroup (unit, chunk of <= kChunk samples): LDS counting sort of the chunk's entries by COARSE
//                            BIN (`fine` consecutive partitions): keys + 16-bit partition tags written back in bin order,
//                            coalesced, with the chunk's bin offsets.  No global atomics, no capacity: exact for any ids.
//   level 2  k_prepass_sort  workgroup (unit, coarse bin): collects the bin's run of every chunk (B / n_bins entries on
//                            average) into `fine` LDS buckets, rank-sorts each bucket (keys are unique), writes the
//                            sorted keys and the counts.  A partition with more than kBucket entries only gets its count
//                            (the update kernel's general path takes it, as after k_embed_segments).
// The result does not depend on the order the LDS atomics hand out slots: the rank sort fixes it.
__global__ __launch_bounds__(kThreads) void k_prepass_bin(UpdArgs A) {
  // dynamic LDS, sized by the launch (prepass_bin_lds): keys [chunk] | cnt [n_bins + 1] (counts, then exclusive starts) |
  // cur [n_bins] | tags [chunk] -- 30 KB at B = 262 144 (five workgroups per CU) instead of 56 KB for the largest shapes
  extern __shared__ __align__(16) uint32_t pp_lds[];
  __shared__ int wsum[kThreads / 64];
  const int tid = threadIdx.x;
  uint32_t* keys = pp_lds;
  int* cnt = reinterpret_cast<int*>(keys + A.chunk);
  int* cur = cnt + (A.n_bins + 1);
  uint16_t* tags = reinterpret_cast<uint16_t*>(cur + A.n_bins);
  const int u = static_cast<int>(blockIdx.x) / A.n_chunks, ck = static_cast<int>(blockIdx.x) - u * A.n_chunks;
  const int32_t* un = A.units + 4 * u;
  const int di = uni(un[0]), wi = uni(un[1]);
  const int64_t vocab = uni((di >= 0) ? A.deep[di].vocab : A.wide[wi].vocab);
  const int B = A.B, NB = A.n_bins;
  const int chunk = A.chunk;
  const uint32_t fine = static_cast<uint32_t>(A.fine);
  const int b0 = ck * chunk;
  const int n = (B - b0) < chunk ? (B - b0) : chunk;
  const int32_t* ids = A.ids_t + static_cast<int64_t>(u) * B + b0;
  for (int i = tid; i < NB; i += kThreads) {
    cnt[i] = 0;
    cur[i] = 0;
  }
  __syncthreads();
  // pass A: bin sizes (the ids stay in L2 for pass B; 16 registers of keys per thread would cost more than re-reading)
#pragma unroll 4
  for (int e = tid; e < chunk; e += kThreads) {
    if (e < n) {
      const int32_t id = clamp_id(ldg_i32(ids + e), vocab);
      const uint32_t idq = div_p(static_cast<uint32_t>(id), A.pmagic, A.pshift);
      const uint32_t p = static_cast<uint32_t>(id) - idq * static_cast<uint32_t>(A.P);
      atomicAdd(&cnt[p / fine], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the NB counts (NB <= kBinsMax: kSc per thread, wave scan, four wave totals)
  {
    constexpr int kSc = kBinsMax / kThreads;
    int v[kSc], t = 0;
#pragma unroll
    for (int j = 0; j < kSc; ++j) {
      const int i = kSc * tid + j;
      v[j] = i < NB ? cnt[i] : 0;
      t += v[j];
    }
    int incl = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if ((tid & 63) >= off) incl += o;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    int run = base + incl - t;
#pragma unroll
    for (int j = 0; j < kSc; ++j) {
      const int i = kSc * tid + j;
      if (i < NB) cnt[i] = run;
      run += v[j];
    }
    if (tid == kThreads - 1) cnt[NB] = run;     // (= n; bins past kSc * kThreads do not exist: NB <= kBinsMax)
  }
  __syncthreads();
  // pass B: scatter (key, tag) to its bin's run
#pragma unroll 4
  for (int e = tid; e < chunk; e += kThreads) {
    if (e < n) {
      const int32_t id = clamp_id(ldg_i32(ids + e), vocab);
      const uint32_t idq = div_p(static_cast<uint32_t>(id), A.pmagic, A.pshift);
      const uint32_t p = static_cast<uint32_t>(id) - idq * static_cast<uint32_t>(A.P);
      const int c = static_cast<int>(p / fine);
      const int pos = cnt[c] + atomicAdd(&cur[c], 1);
      keys[pos] = (idq << A.bbits) | static_cast<uint32_t>(b0 + e);
      tags[pos] = static_cast<uint16_t>(p);
    }
  }
  __syncthreads();
  const int64_t so = (static_cast<int64_t>(u) * A.n_chunks + ck) * chunk;
  for (int e = tid; e < n; e += kThreads) {
    *(DCTR_GLOBAL uint32_t*)(A.stage_keys + so + e) = keys[e];
    *(DCTR_GLOBAL uint16_t*)(A.stage_tags + so + e) = tags[e];
  }
  int32_t* offs = A.stage_offs + (static_cast<int64_t>(u) * A.n_chunks + ck) * (NB + 1);
  for (int i = tid; i <= NB; i += kThreads) *(DCTR_GLOBAL int32_t*)(offs + i) = cnt[i];
}

__global__ __launch_bounds__(kSortT) void k_prepass_sort(UpdArgs A) {
  extern __shared__ __align__(16) uint32_t bk_lds[];      // [fine][kBucket]
  __shared__ int cnt[kFineMax], n_f[kFineMax], start[kFineMax + 1];
  const int fine = A.fine;
  const int tid = threadIdx.x;
  const int NB = A.n_bins, P = A.P;
  const int u = static_cast<int>(blockIdx.x) / NB, c = static_cast<int>(blockIdx.x) - u * NB;
  if (tid < fine) cnt[tid] = 0;
  __syncthreads();
  // eight lanes per chunk walk the bin's run of that chunk -- TWO chunks per lane group at a time, their offsets requested
  // together and their entries requested together: the walk is two dependent round trips per pass (offsets, entries), and at
  // B = 262 144 (64 chunks, ~6 entries per run) a workgroup is nothing but those round trips
  const int sub = tid & 7;
  const uint32_t p0 = static_cast<uint32_t>(c) * static_cast<uint32_t>(fine);
  constexpr int kGroups = kSortT / 8;
#ifdef DCTR_DIAG
  const int dbg = A.presorted;     // timing experiments of tools/prepass_bench.py (wrong results): 1 no stores, 2 no sort, 4 no walk
#else
  constexpr int dbg = 0;
#endif
  for (int ck0 = tid >> 3; ck0 < ((dbg & 4) ? 0 : A.n_chunks); ck0 += 2 * kGroups) {
    int s0[2], s1[2];
    const uint32_t* kk[2];
    const uint16_t* tt[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ck = ck0 + h * kGroups;
      const int ckc = ck < A.n_chunks ? ck : ck0;
      const int64_t row = static_cast<int64_t>(u) * A.n_chunks + ckc;
      const int32_t* offs = A.stage_offs + row * (NB + 1) + c;
      s0[h] = ldg_i32(offs);
      s1[h] = ldg_i32(offs + 1);
      kk[h] = A.stage_keys + row * A.chunk;
      tt[h] = A.stage_tags + row * A.chunk;
    }
    if (ck0 + kGroups >= A.n_chunks) s1[1] = s0[1];     // (no second chunk: an empty run)
    // four entries per lane and chunk in flight (a hot id's run is hundreds of entries long: one dependent round trip per
    // entry made the hot bin's workgroup the launch's tail)
    const int len0 = s1[0] - s0[0], len1 = s1[1] - s0[1];
    const int len = len0 > len1 ? len0 : len1;
    for (int i = sub; i < len; i += 32) {
      uint32_t key[2][4];
      uint32_t tg[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int lh = h ? len1 : len0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int ii = i + 8 * q;
          ii = ii < lh ? ii : (lh > 0 ? lh - 1 : 0);
          ii += s0[h];
          ii = lh > 0 ? ii : 0;                          // (an empty run: any staged entry, masked below)
          key[h][q] = *(const DCTR_GLOBAL uint32_t*)(kk[h] + ii);
          tg[h][q] = *(const DCTR_GLOBAL uint16_t*)(tt[h] + ii);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int lh = h ? len1 : len0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (i + 8 * q < lh) {
            const int f = static_cast<int>(tg[h][q] - p0);
            const int slot = atomicAdd(&cnt[f], 1);
            if (slot < kBucket) bk_lds[f * kBucket + slot] = key[h][q];
          }
        }
      }
    }
  }
  __syncthreads();
  // counts out; prefix of the sortable partitions' sizes for the flattened rank sort (in LDS: indexed per entry)
  if (tid == 0) {
    int run = 0;
    for (int f = 0; f < fine; ++f) {
      const int nf = cnt[f];
      start[f] = run;
      n_f[f] = (nf <= kBucket && static_cast<int>(p0) + f < P) ? nf : 0;
      run += n_f[f];
    }
    start[fine] = run;
  }
  if (tid < fine && static_cast<int>(p0) + tid < P)
    *(DCTR_GLOBAL int32_t*)(A.bcnt + static_cast<int64_t>(u) * P + p0 + tid) = cnt[tid];
  __syncthreads();
  const int total = start[fine];
  for (int e = tid; e < total; e += kSortT) {
    int f = 0;
    for (int g = 1; g < fine; ++g) f += (e >= start[g]) ? 1 : 0;
    const int nf = n_f[f];
    const uint32_t* src = bk_lds + f * kBucket;
    const uint32_t mine = src[e - start[f]];
    // (one broadcast LDS read per key: ~45 of the kernel's ~80 us at B = 262 144; dwordx4 reads with masked compares
    // measured slower, 137 against 112 us for both levels)
    int rank = 0;
#pragma unroll 4
    for (int i = 0; i < ((dbg & 2) ? 1 : nf); ++i) rank += (src[i] < mine) ? 1 : 0;
    if (!(dbg & 1) || rank == 12345678)
      *(DCTR_GLOBAL uint32_t*)(A.bkeys + (static_cast<int64_t>(u) * P + p0 + f) * kBucket + rank) = mine;
  }
}

// ---- X -> ids_t (+ parts_t) (standalone; the forward kernel fuses the same thing) -------------------
__global__ __launch_bounds__(kThreads) void k_embed_ids(const dctr_field_t* __restrict__ deep,
                                                        const dctr_field_t* __restrict__ wide,
                                                        const int32_t* __restrict__ units, int n_units,
                                                        const float* __restrict__ X, int64_t ldx, int B,
                                                        int32_t* __restrict__ ids_t, uint16_t* __restrict__ parts_t,
                                                        int n_parts) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= static_cast<int64_t>(n_units) * B) return;
  const int u = static_cast<int>(i / B), b = static_cast<int>(i - static_cast<int64_t>(u) * B);
  const int32_t id = static_cast<int32_t>(X[static_cast<int64_t>(b) * ldx + units[4 * u + 2]]);
  ids_t[i] = id;
  if (parts_t) {
    const int di = units[4 * u], wi = units[4 * u + 1];
    const int64_t vocab = (di >= 0) ? deep[di].vocab : wide[wi].vocab;
    parts_t[i] = static_cast<uint16_t>(static_cast<uint32_t>(clamp_id(id, vocab)) % static_cast<uint32_t>(n_parts));
  }
}

// Partitions per unit: ~3/4 of a tile per workgroup (a partition's size is Poisson-like: mean 96 of 128 leaves
// 3.3 sigma of head room), whatever the batch.  P need not be a power of two.
int pick_p(int B, int tile) {
#ifdef DCTR_DIAG
  if (g_force_p > 0) return g_force_p;
#endif
  int per = tile * 3 / 4;
  if (per < 1) per = 1;
  int P = (B + per - 1) / per;
  return P < 1 ? 1 : P;
}

int ceil_log2(int64_t x) {
  int l = 0;
  while ((int64_t(1) << l) < x) ++l;
  return l;
}

// lanes of a workgroup that share one row (a power of two) and the floats each of them moves
void lane_layout(const dctr_plan_t* plan, int* vec_out, int* lpr_out) {
  int vec = plan->n_deep > 0 ? plan->vec : 1;
  if (vec == 4 && plan->emb_dim > 0 && plan->emb_dim % 8 == 0 && plan->emb_dim <= 64) vec = 8;  // two dwordx4 per lane
  int lpr = 1;
  const int need = plan->n_deep > 0 ? (plan->max_dim + vec - 1) / vec : 1;
  while (lpr < need) lpr <<= 1;
  *vec_out = vec;
  *lpr_out = lpr;
}

}  // namespace

// partitions per unit dctr_embed_update uses for this plan / batch: what the forward's parts_t side output is taken
// modulo (0 on bad arguments)
extern "C" int32_t dctr_embed_update_partitions(const dctr_plan_t* plan, int32_t B) {
  if (!plan || B <= 0) return 0;
  int vec, lpr;
  lane_layout(plan, &vec, &lpr);
  return pick_p(B, kThreads / lpr);
}

extern "C" int dctr_embed_ids(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, const float* X,
                              int64_t ldx, int32_t B, int32_t* ids_t, uint16_t* parts_t, dctr_stream_t stream) {
  if (!units || !X || !ids_t || n_units <= 0 || B < 0) return DCTR_EINVAL;
  if (parts_t && !plan) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const int n_parts = parts_t ? dctr_embed_update_partitions(plan, B) : 0;
  if (parts_t && (n_parts <= 0 || n_parts > 65535)) return DCTR_ENOSUP;
  const int64_t n = static_cast<int64_t>(n_units) * B;
  k_embed_ids<<<dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                static_cast<hipStream_t>(stream)>>>(plan ? plan->deep : nullptr, plan ? plan->wide : nullptr, units,
                                                    n_units, X, ldx, B, ids_t, parts_t, n_parts);
  return launch_status();
}

// diagnostics: per-workgroup phase timestamps (8 x u64 per workgroup, wall_clock64 ticks) and a partition override
#ifdef DCTR_DIAG
extern "C" void dctr_dbg_update_trace(unsigned long long* buf, int32_t force_p) {
  g_trace = buf;
  g_force_p = force_p;
}
#endif

extern "C" int dctr_embed_update_supported(const dctr_plan_t* plan, int64_t max_vocab, int32_t B) {
  if (!plan || B <= 0) return 0;
  if (plan->n_deep != plan->n_deep_fixed || plan->n_wide != plan->n_wide_fixed) return 0;
  if (plan->vec != 1 && plan->vec != 2 && plan->vec != 4) return 0;
  if (plan->n_deep > 0 && plan->max_dim > 64 * plan->vec) return 0;
  if (B > (1 << 20)) return 0;
  if (max_vocab >= (int64_t(1) << 31)) return 0;
  const int P = pick_p(B, kThreads);   // the smallest P any lane layout would use: the widest keys
  const int bbits = ceil_log2(B < 2 ? 2 : B);
  if (ceil_log2((max_vocab > 0 ? max_vocab : 1) / P + 2) + bbits > 32) return 0;
  return 1;
}

// ints of the optional bucket workspace of dctr_embed_update for this plan / batch (must be zero before its first use;
// the kernels leave the counters at zero)
// Staging of the two-level pre-pass behind the bucket arrays: keys [n_units][n_chunks * chunk] u32, tags (u16, two per
// int), chunk offsets [n_units][n_chunks][n_bins + 1].  Used from kTwoLevelMin samples: below it a workgroup's scan over
// the unit's B partition tags is as cheap (uniform ids: 17.6 against 24 us at B = 8192) or cheaper (Zipf ids: 39 against
// 26 us -- four partitions of several hundred entries rank-sorted by ONE workgroup); profiles/r04_prepass_two_level.jsonl.
constexpr int kTwoLevelMin = 16384;
struct StageLayout {
  int n_chunks, n_bins, chunk, fine;
  int64_t keys_off, tags_off, offs_off, total;   // int32 offsets inside the workspace
};
StageLayout stage_layout(int32_t n_units, int32_t B, int P) {
  StageLayout L;
  // samples per level-1 workgroup: as large as LDS allows once the launch has ~1000 workgroups, never below 512
  L.chunk = kChunk;
  while (L.chunk > 512 && static_cast<int64_t>(n_units) * B / L.chunk < 1024) L.chunk >>= 1;
  L.n_chunks = (B + L.chunk - 1) / L.chunk;
  // partitions per coarse bin: the second level reads a (chunk, bin) run of chunk * fine / P ~ 96 * fine * chunk / B entries
  // per three lines it touches (keys, tags, offsets) and is bound by that request rate at large batches (B = 262 144,
  // fine 4: 192 lines per 380 entries, 80 us); more partitions per workgroup make the runs longer -- and the rank sort
  // of a skewed bin more serial, hence only as many as the batch needs
  L.fine = 4;      // (measured at B = 262 144: 4 -> 110, 16 -> 126 us for both levels; profiles/r04_prepass_two_level.jsonl)
  L.n_bins = (P + L.fine - 1) / L.fine;
  const int64_t ne = static_cast<int64_t>(n_units) * L.n_chunks * L.chunk;
  L.keys_off = static_cast<int64_t>(n_units) * P * (1 + kBucket);
  L.tags_off = L.keys_off + ne;
  L.offs_off = L.tags_off + (ne + 1) / 2;
  L.total = L.offs_off + static_cast<int64_t>(n_units) * L.n_chunks * (L.n_bins + 1);
  return L;
}

// the two launches of the two-level pre-pass on `a` (units, ids, P, magic, bcnt / bkeys set by the caller)
int launch_two_level(UpdArgs a, const StageLayout& SL, int32_t* workspace, hipStream_t s) {
  a.stage_keys = reinterpret_cast<uint32_t*>(workspace + SL.keys_off);
  a.stage_tags = reinterpret_cast<uint16_t*>(workspace + SL.tags_off);
  a.stage_offs = workspace + SL.offs_off;
  a.n_chunks = SL.n_chunks;
  a.n_bins = SL.n_bins;
  a.chunk = SL.chunk;
  a.fine = SL.fine;
  a.presorted = 0;
#ifdef DCTR_DIAG
  if (const char* e = getenv("DCTR_PREPASS_DBG")) a.presorted = atoi(e);
#endif
  const size_t lds1 = static_cast<size_t>(SL.chunk) * 6 + static_cast<size_t>(2 * SL.n_bins + 1) * 4;
  k_prepass_bin<<<dim3(static_cast<unsigned>(a.n_units) * SL.n_chunks), dim3(kThreads), lds1, s>>>(a);
  const int st = launch_status();
  if (st != DCTR_OK) return st;
  k_prepass_sort<<<dim3(static_cast<unsigned>(a.n_units) * SL.n_bins), dim3(kSortT),
                   static_cast<size_t>(SL.fine) * kBucket * 4, s>>>(a);
  return launch_status();
}
bool two_level_applies(const StageLayout& SL, int32_t B, int P, int64_t workspace_ints) {
  int two_level_min = kTwoLevelMin;
#ifdef DCTR_DIAG
  if (const char* e = getenv("DCTR_PREPASS_TWO_LEVEL_MIN")) two_level_min = atoi(e);   // tools/prepass_bench.py sweeps it
#endif
  return B >= kTwoLevelMin && B >= two_level_min && SL.n_bins <= kBinsMax && P <= 65535 && workspace_ints >= SL.total;
}

extern "C" int64_t dctr_embed_update_workspace_ints(const dctr_plan_t* plan, int32_t n_units, int32_t B) {
  if (!plan || n_units <= 0 || B <= 0) return 0;
  const int P = dctr_embed_update_partitions(plan, B);
  if (B >= kTwoLevelMin && stage_layout(n_units, B, P).n_bins <= kBinsMax) return stage_layout(n_units, B, P).total;
  return static_cast<int64_t>(n_units) * P * (1 + kBucket);
}

extern "C" int dctr_embed_segments(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, int64_t max_vocab,
                                   const int32_t* ids_t, const uint16_t* parts_t, int32_t B, int32_t* workspace,
                                   int64_t workspace_ints, dctr_stream_t stream) {
  if (!plan || !units || !ids_t || !workspace || n_units <= 0 || B < 0) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  if (!dctr_embed_update_supported(plan, max_vocab, B)) return DCTR_ENOSUP;
  int vec, lpr;
  lane_layout(plan, &vec, &lpr);
  const int P = pick_p(B, kThreads / lpr);
  const int64_t nbuckets = static_cast<int64_t>(n_units) * P;
  if (workspace_ints < nbuckets * (1 + kBucket)) return DCTR_EINVAL;
  UpdArgs a = {};
  a.wd_step.kind = -1;
  a.deep = plan->deep; a.wide = plan->wide; a.units = units; a.ids_t = ids_t; a.parts_t = parts_t;
  a.n_units = n_units; a.B = B; a.P = P;
  a.bbits = ceil_log2(B < 2 ? 2 : B);
  a.pshift = 32 + ceil_log2(P);
  a.pmagic = static_cast<uint64_t>((static_cast<unsigned __int128>(1) << a.pshift) / static_cast<unsigned>(P)) + 1;
  a.bcnt = workspace;
  a.bkeys = reinterpret_cast<uint32_t*>(workspace + nbuckets);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(nbuckets)), block(kThreads);
  // small batches: every workgroup finds its entries by comparing the forward's 16-bit partition tags (B / 8 vector
  // loads per workgroup).  Large batches: the two-level pre-pass (chunks re-ordered by coarse bin, then one workgroup
  // per bin), when the caller's workspace has room for its staging; else (or without tags): one global atomic per entry.
  const StageLayout SL = stage_layout(n_units, B, P);
  if (two_level_applies(SL, B, P, workspace_ints)) {
    return launch_two_level(a, SL, workspace, s);
  } else if (parts_t && P <= 65535 && B <= 32768) {
    k_embed_segments<false><<<grid, block, 0, s>>>(a);
  } else {
    const int64_t ne = static_cast<int64_t>(n_units) * B;
    k_bucket<<<dim3(static_cast<unsigned>((ne + kThreads - 1) / kThreads)), block, 0, s>>>(a);
    const int st = launch_status();
    if (st != DCTR_OK) return st;
    k_embed_segments<true><<<grid, block, 0, s>>>(a);
  }
  return launch_status();
}

extern "C" int dctr_embed_update(const dctr_plan_t* plan, const int32_t* units, int32_t n_units,
                                 int64_t max_vocab, const int32_t* ids_t, const uint16_t* parts_t, int32_t B,
                                 const float* g_out,
                                 int64_t ld_g, const float* out, int64_t ld_out, const float* fm_s,
                                 int64_t ld_s, const float* g_fm, const float* g_wide, int64_t ld_gw,
                                 int32_t opt, float lr, float eps, const float* X, int64_t ld_x, float* g_wdense,
                                 const dctr_dense_step_t* wdense_step, int32_t* workspace, int64_t workspace_ints,
                                 int32_t presorted, dctr_stream_t stream) {
  (void)out;
  (void)ld_out;  // kept in the signature: the forward's rows are no longer re-read (FM is folded algebraically)
  if (!plan || !units || !ids_t || n_units <= 0 || B < 0) return DCTR_EINVAL;
  if (g_wide && ld_gw < 1) return DCTR_EINVAL;
  if (g_wdense && (!X || !g_wide || plan->n_wdense <= 0 || !plan->wdense_cols)) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  if (opt != DCTR_UPD_SGD && opt != DCTR_UPD_ADAGRAD && opt != DCTR_UPD_ACCUM) return DCTR_EINVAL;
  if (!dctr_embed_update_supported(plan, max_vocab, B)) return DCTR_ENOSUP;
  if (opt == DCTR_UPD_ACCUM && !(plan->flags & DCTR_PLAN_HAS_GACC)) return DCTR_EINVAL;
  if (opt == DCTR_UPD_ADAGRAD && !(plan->flags & DCTR_PLAN_HAS_STATE)) return DCTR_EINVAL;
  if (g_fm && (!fm_s || plan->emb_dim <= 0)) return DCTR_EINVAL;
  int vec, lpr;
  lane_layout(plan, &vec, &lpr);
  const int avec = plan->n_deep > 0 ? plan->vec : 1;  // alignment granule the caller guarantees
  if (avec > 1) {
    if (g_out && (ld_g % avec != 0 || reinterpret_cast<uintptr_t>(g_out) % (4 * avec) != 0)) return DCTR_EALIGN;
    if (g_fm && (ld_s % avec != 0 || reinterpret_cast<uintptr_t>(fm_s) % (4 * avec) != 0)) return DCTR_EALIGN;
  }
  UpdArgs a;
  a.deep = plan->deep; a.wide = plan->wide; a.units = units; a.ids_t = ids_t;
  a.parts_t = nullptr;
  a.gout = g_out; a.fm_s = fm_s; a.gfm = g_fm; a.gwide = g_wide; a.ldgw = ld_gw;
  a.ldg = ld_g; a.lds_ = ld_s;
  a.n_units = n_units; a.B = B;
  a.bbits = ceil_log2(B < 2 ? 2 : B);
  a.lr = lr; a.eps = eps;
#ifdef DCTR_DIAG
  a.trace = g_trace;
#else
  a.trace = nullptr;
#endif
  a.X = X; a.ldx = ld_x; a.wdense_cols = plan->wdense_cols; a.n_wdense = plan->n_wdense; a.g_wdense = g_wdense;
  a.wd_step = dense_step_dev(g_wdense ? wdense_step : nullptr);

  const int P = pick_p(B, kThreads / lpr);
  a.P = P;
  if (parts_t && P <= 65535) a.parts_t = parts_t;   // (taken modulo the same P: dctr_embed_update_partitions)
  a.pshift = 32 + ceil_log2(P);
  a.pmagic = static_cast<uint64_t>((static_cast<unsigned __int128>(1) << a.pshift) / static_cast<unsigned>(P)) + 1;
  const dim3 grid(static_cast<unsigned>(n_units) * static_cast<unsigned>(P) +
                      (g_wdense ? static_cast<unsigned>(plan->n_wdense) : 0u)),
      block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  a.bcnt = nullptr;
  a.bkeys = nullptr;
  a.presorted = 0;
  const int64_t nbuckets = static_cast<int64_t>(n_units) * P;
  if (presorted && !(workspace && workspace_ints >= nbuckets * (1 + kBucket))) return DCTR_EINVAL;
  if (workspace && workspace_ints >= nbuckets * (1 + kBucket)) {
    a.bcnt = workspace;
    a.bkeys = reinterpret_cast<uint32_t*>(workspace + nbuckets);
    const StageLayout SL = stage_layout(n_units, B, P);
    if (presorted) {
      a.presorted = 1;    // dctr_embed_segments filled (and sorted) the buckets on these very ids
    } else if (two_level_applies(SL, B, P, workspace_ints)) {
      // a large batch without a pre-pass (the owners' update of the table-sharded step): the two-level pre-pass here, in
      // line, then the lean pre-sorted kernel -- instead of one global atomic per entry and every workgroup's own sort
      const int st = launch_two_level(a, SL, workspace, s);
      if (st != DCTR_OK) return st;
      a.presorted = 1;
    } else {
      const int64_t ne = static_cast<int64_t>(n_units) * B;
      k_bucket<<<dim3(static_cast<unsigned>((ne + kThreads - 1) / kThreads)), block, 0, s>>>(a);
      const int st = launch_status();
      if (st != DCTR_OK) return st;
    }
  }

  // presorted: one launch -- the lean path for every partition the pre-pass could sort, the general path inside the same
  // kernel for an overflowing one, the dense half of Linear in the last workgroups (`grid` above already counts them)
#define DCTR_UPD_LAUNCH(VEC_, LPR_)                                                              \
  do {                                                                                           \
    if (opt == DCTR_UPD_ADAGRAD) {                                                               \
      if (a.presorted) {                                                                         \
        k_embed_apply_sorted<VEC_, LPR_, 1><<<grid, block, 0, s>>>(a);                           \
      } else {                                                                                   \
        k_embed_update<VEC_, LPR_, 1><<<grid, block, 0, s>>>(a);                                 \
      }                                                                                          \
    } else if (opt == DCTR_UPD_SGD) {                                                            \
      if (a.presorted) {                                                                         \
        k_embed_apply_sorted<VEC_, LPR_, 0><<<grid, block, 0, s>>>(a);                           \
      } else {                                                                                   \
        k_embed_update<VEC_, LPR_, 0><<<grid, block, 0, s>>>(a);                                 \
      }                                                                                          \
    } else {                                                                                     \
      if (a.presorted) {                                                                         \
        k_embed_apply_sorted<VEC_, LPR_, 2><<<grid, block, 0, s>>>(a);                           \
      } else {                                                                                   \
        k_embed_update<VEC_, LPR_, 2><<<grid, block, 0, s>>>(a);                                 \
      }                                                                                          \
    }                                                                                            \
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

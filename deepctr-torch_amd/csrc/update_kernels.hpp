// update.hip -- deterministic, atomic-free embedding backward with the optimizer fused in (gfx950).
//
// Replaces autograd's aten::embedding_dense_backward x (n_deep + n_wide) + FM's backward + the dense
// optimizer walk over every table (basemodel.py:261-262, interaction.py:26-34): fixed-length fields over distinct
// tables (the Criteo shape) and -- since round 5, GEN = true below -- pooled VarLen fields and tables shared through
// embedding_name.  (The atomic two-pass kernels of embed.hip only take a unit of more than 128 X columns.)
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
//
// Round 5: the kernels are templates on GEN.  GEN = false is the simple case above, compiled exactly as before.
// GEN = true runs GENERAL units (include/dctr.h, dctr_plan_ext_t): a unit is a (deep table, wide table) pair with every X
// column that feeds it -- the positions of a pooled VarLenSparseFeat (inputs.py:141-155, sequence.py:49-77: the
// "EmbeddingBag backward") and every column that shares the table through `embedding_name` (inputs.py:158-180).  An entry is
// named by v = slot * B + b; what it contributes is the field's gradient slice times the pooling weight (1, 1 / (count +
// 1e-8), or the arg-max mask); everything behind the key -- sort, segment sums, one read-modify-write per row -- is the same
// code, so pooled and shared tables get the same deterministic, atomic-free update.
// This header is included by update.hip (GEN = false instantiations + the pre-pass kernels + the C ABI) and by
// update_gen.hip (GEN = true instantiations): two translation units that compile in parallel.
#pragma once
#include "common.hpp"
#include "lazy_opt.hpp"

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
  // general units (GEN kernels; NULL / 0 otherwise).  n_units above counts VUNITS then (the grid is n_units * P).
  int32_t n_vcols;                   // virtual columns of ids_t / parts_t (= n_units in the simple case)
  const dctr_uslot_t* slots;         // [n_vcols]
  const dctr_vunit_t* vunits;        // [n_units]
  const float* out;                  // [B, ldo] the forward's rows: a pooled field's pooled value (FM's backward)
  int64_t ldo;
  const float* den_t;                // [n_den][B] mean pooling's divisor
  const uint8_t* amax;               // [B, ld_am] max pooling's arg-max positions
  int64_t ld_am;
  uint64_t bmagic;                   // v / B == (v * bmagic) >> bshift
  int32_t bshift;
  // DCTR_UPD_LAZY (round 6; simple units only): the lazily regularised / Adam tables' step that carries the batch's data
  // gradient -- g = G + 2 lambda w, one step of csrc/lazy.hip's optimizer on (w, s1, s2), stamp = t + 1 -- at the row, where
  // the sorted update has the gradient's sum in registers (it used to go through a gradient slab and a second pass)
  const dctr_lazy_unit_t* lz;        // [n_units] device
  dctr_lazy::OptConst lzo;
  const int32_t* lz_step;            // device: optimizer steps completed so far
};

constexpr int kMaxSlots = DCTR_MAX_UNIT_SLOTS;   // slots of one unit (their descriptors are staged in LDS)

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

// a unit's lazily updated tables, wave-uniform (DCTR_UPD_LAZY)
struct LazyCtx {
  float* deep; float* s1; float* s2; float* wide; float* ws1; float* ws2; int32_t* stamp;
  int64_t ld_d, ld_s1, ld_w, ld_ws1;
  int dim;
  float lam2d, lam2w, ss1, bc1;
  int t;
};
__device__ __forceinline__ LazyCtx lazy_ctx(const UpdArgs& A, int u) {
  LazyCtx Z;
  const dctr_lazy_unit_t* un = A.lz + u;
  Z.deep = uni(un->deep); Z.s1 = uni(un->deep_s1); Z.s2 = uni(un->deep_s2);
  Z.wide = uni(un->wide); Z.ws1 = uni(un->wide_s1); Z.ws2 = uni(un->wide_s2);
  Z.stamp = uni(un->stamp);
  Z.dim = uni(un->dim);
  const int ldd = uni(un->ld_deep), lds1 = uni(un->ld_deep_s1), ldw = uni(un->ld_wide), ldws = uni(un->ld_wide_s1);
  Z.ld_d = ldd > 0 ? ldd : Z.dim;
  Z.ld_s1 = lds1 > 0 ? lds1 : Z.dim;
  Z.ld_w = ldw > 0 ? ldw : 1;
  Z.ld_ws1 = ldws > 0 ? ldws : 1;
  Z.lam2d = 2.f * __builtin_bit_cast(float, uni(__builtin_bit_cast(int32_t, un->l2_deep)));
  Z.lam2w = 2.f * __builtin_bit_cast(float, uni(__builtin_bit_cast(int32_t, un->l2_wide)));
  Z.t = uni(*(const DCTR_GLOBAL int32_t*)A.lz_step);
  Z.ss1 = 0.f;
  Z.bc1 = 1.f;
  if (A.lzo.kind == DCTR_LAZY_ADAM) dctr_lazy::adam_scalars(A.lzo, Z.t + 1, Z.ss1, Z.bc1);
  return Z;
}
// the step on one strip: G = the batch's gradient sum for the row; w / a / b = the row's strips of the table and the
// optimizer's two state slabs (zeros where a slab is absent).  lazy.hip k_lazy<., 1>'s arithmetic (opt_step, IEEE division).
template <int VEC>
__device__ __forceinline__ void lazy_apply_strip(const UpdArgs& A, const LazyCtx& Z, float lam2, float* pw, float* pa,
                                                 float* pb, const Strip<VEC>& G, const Strip<VEC>& w, const Strip<VEC>& a,
                                                 const Strip<VEC>& b) {
  Strip<VEC> nw = w, na = a, nb = b;
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    dctr_lazy::opt_step(A.lzo, G.v[i] + lam2 * w.v[i], nw.v[i], na.v[i], nb.v[i], Z.ss1, Z.bc1);
  strip_store<VEC>(pw, nw);
  if (pa) strip_store<VEC>(pa, na);
  if (pb) strip_store<VEC>(pb, nb);
}

// id / P and id % P for a runtime P through a host-computed reciprocal (exact for 0 <= id < 2^31)
__device__ __forceinline__ uint32_t div_p(uint32_t id, uint64_t magic, int shift) {
  const uint64_t lo = (magic & 0xFFFFFFFFull) * id, hi = (magic >> 32) * id;
  return static_cast<uint32_t>(((lo >> 32) + hi) >> (shift - 32));
}

__device__ __forceinline__ int32_t clamp_id(int32_t id, int64_t vocab) {
  return (static_cast<uint64_t>(static_cast<int64_t>(id)) >= static_cast<uint64_t>(vocab)) ? 0 : id;
}

// What a workgroup knows about its unit (all wave-uniform).  Simple case: unit u = virtual column u, one slot, k = 1.
struct UnitCtx {
  int di, wi;        // fields that supply the unit's tables
  int c0, ns;        // virtual columns [c0, c0 + ns)
  int k;             // the unit has k * P partitions
  int pu;            // this workgroup's partition of the unit (j * P + p)
  int vbits;         // bits of v = slot * B + b in a key
  int kshift;
  uint64_t kmagic;
};
template <bool GEN>
__device__ __forceinline__ UnitCtx unit_ctx(const UpdArgs& A, int vu, int p) {
  UnitCtx U;
  if constexpr (GEN) {
    const dctr_vunit_t* d = A.vunits + vu;
    U.di = uni(d->di);
    U.wi = uni(d->wi);
    U.c0 = uni(d->c0);
    U.ns = uni(d->n_slots);
    U.k = uni(d->k);
    U.pu = uni(d->j) * A.P + p;
    U.kshift = uni(d->kshift);
    U.kmagic = static_cast<uint64_t>(uni(static_cast<int64_t>(d->kmagic)));
    const int nv = U.ns * A.B;
    U.vbits = 32 - __builtin_clz(static_cast<unsigned>((nv < 2 ? 2 : nv) - 1));
  } else {
    const int32_t* un = A.units + 4 * vu;
    U.di = uni(un[0]);
    U.wi = uni(un[1]);
    U.c0 = vu;
    U.ns = 1;
    U.k = 1;
    U.pu = p;
    U.vbits = A.bbits;
    U.kshift = 32;
    U.kmagic = 0;
  }
  return U;
}
// clamped id -> (id / (k P), id mod (k P)): id = q1 P + r1, q1 = idq k + r2  =>  id = idq (k P) + (r2 P + r1)
template <bool GEN>
__device__ __forceinline__ void split_id(const UpdArgs& A, const UnitCtx& U, uint32_t id, uint32_t& idq, uint32_t& pu) {
  const uint32_t q1 = div_p(id, A.pmagic, A.pshift);
  const uint32_t r1 = id - q1 * static_cast<uint32_t>(A.P);
  if (!GEN || U.k == 1) {
    idq = q1;
    pu = r1;
  } else {
    idq = div_p(q1, U.kmagic, U.kshift);
    pu = r1 + static_cast<uint32_t>(A.P) * (q1 - idq * static_cast<uint32_t>(U.k));
  }
}

// One entry of a general unit, decoded: sample, the deep field's slice, the pooling weight's ingredients.
struct EntryDesc {
  int b, goff, pool, t, den, am_deep, am_wide;
};
template <bool GEN>
__device__ __forceinline__ EntryDesc entry_desc(const UpdArgs& A, const dctr_uslot_t* sl, int v, int goff_simple) {
  EntryDesc E;
  if constexpr (GEN) {
    const int s = static_cast<int>(div_p(static_cast<uint32_t>(v), A.bmagic, A.bshift));
    const dctr_uslot_t& d = sl[s];
    E.b = v - s * A.B;
    E.goff = d.goff;
    E.pool = d.pool;
    E.t = d.t;
    E.den = d.den;
    E.am_deep = d.am_deep;
    E.am_wide = d.am_wide;
  } else {
    E.b = v;
    E.goff = goff_simple;
    E.pool = 0;
    E.t = 0;
    E.den = -1;
    E.am_deep = -1;
    E.am_wide = -1;
  }
  return E;
}
// the unit's slot descriptors -> LDS (GEN; before the first barrier of the caller)
__device__ __forceinline__ void stage_slots(const UpdArgs& A, const UnitCtx& U, dctr_uslot_t* sl, int tid, int nthreads) {
  constexpr int kW = sizeof(dctr_uslot_t) / 4;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(A.slots + U.c0);
  uint32_t* dst = reinterpret_cast<uint32_t*>(sl);
  const int n = (U.ns < kMaxSlots ? U.ns : kMaxSlots) * kW;
  for (int i = tid; i < n; i += nthreads) dst[i] = *(const DCTR_GLOBAL uint32_t*)(src + i);
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

// What one entry of a GENERAL unit contributes (see EntryDesc): the deep strip h, its share gf of the g_fm that is folded at
// the row (fixed slots only), the wide gradient gw.  The arithmetic of a pooled slot follows autograd through
// SequencePoolingLayer (sequence.py:61-77) in the order the reference's backward applies it: FM's backward on the POOLED
// value first (interaction.py:26-34: g + g_fm (S - pooled)), then the pooling's own backward -- x 1 (sum), / (count + 1e-8)
// (mean: the division's backward, same rounding), or routed to the arg-max position per element (max).
template <int VEC>
__device__ __forceinline__ void gen_entry(const UpdArgs& A, const EntryDesc& E, bool lane_on, bool wide_lane, bool fold,
                                          int e0, Strip<VEC>& h, float& gf, float& gw) {
  const int64_t b = E.b;
  Strip<VEC> S = strip_zero<VEC>(), pv = strip_zero<VEC>();
  float gfl = 0.f, den = 1.f;
  uint32_t amd[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) amd[k] = 0u;
  uint32_t amw = 0u;
  const bool deep = lane_on && E.goff >= 0;
  // loads first (nothing is consumed between them: one round trip), then the arithmetic
  if (deep) {
    if (A.gout) h = strip_load<VEC>(A.gout + b * A.ldg + E.goff + e0);
    if (fold) {
      S = strip_load<VEC>(A.fm_s + b * A.lds_ + e0);
      gfl = ldg_f32(A.gfm + b);
      if (E.pool != DCTR_POOL_NONE) pv = strip_load<VEC>(A.out + b * A.ldo + E.goff + e0);
    }
    if (E.pool == DCTR_POOL_MAX) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) amd[k] = *(const DCTR_GLOBAL uint8_t*)(A.amax + b * A.ld_am + E.am_deep + e0 + k);
    }
  }
  if (wide_lane) {
    gw = ldg_f32(A.gwide + b * A.ldgw);
    if (E.pool == DCTR_POOL_MAX) amw = *(const DCTR_GLOBAL uint8_t*)(A.amax + b * A.ld_am + E.am_wide);
  }
  if (E.pool == DCTR_POOL_MEAN && (deep || wide_lane)) den = ldg_f32(A.den_t + static_cast<int64_t>(E.den) * A.B + b);
  if (deep) {
    if (E.pool == DCTR_POOL_NONE) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) h.v[k] += gfl * S.v[k];      // (- g_fm e is folded at the row: e IS the table row)
      gf = gfl;
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) h.v[k] += gfl * (S.v[k] - pv.v[k]);
      if (E.pool == DCTR_POOL_MEAN) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) h.v[k] = h.v[k] / den;
      } else if (E.pool == DCTR_POOL_MAX) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) h.v[k] = (static_cast<int>(amd[k]) == E.t) ? h.v[k] : 0.f;
      }
    }
  }
  if (wide_lane) {
    if (E.pool == DCTR_POOL_MEAN) gw = gw / den;
    else if (E.pool == DCTR_POOL_MAX) gw = (static_cast<int>(amw) == E.t) ? gw : 0.f;
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

template <int VEC, int LPR, int OPT, bool GEN>
__device__ __forceinline__ void upd_partition(const UpdArgs& A, const int u, const int p, const dctr_uslot_t* sl) {
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

  const UnitCtx U = unit_ctx<GEN>(A, u, p);
  const int di = U.di, wi = U.wi;
  dctr_field_t fd = {}, fw = {};
  if (di >= 0) fd = uni_field(A.deep[di]);
  if (wi >= 0) fw = uni_field(A.wide[wi]);
  const int64_t vocab = (di >= 0) ? fd.vocab : fw.vocab;
  const int B = GEN ? U.ns * A.B : A.B;     // the unit's entries: v = slot * B + b in [0, n_slots * B)
  const int32_t* ids = A.ids_t + static_cast<int64_t>(U.c0) * A.B;
  const int64_t Pu = static_cast<int64_t>(U.k) * P;   // rows of this workgroup: id mod Pu == U.pu

  const int grp = tid / LPR, gl = tid % LPR, e0 = gl * VEC;
  const uint32_t bmask = (1u << U.vbits) - 1u;
  const bool deep_on = (di >= 0) && (A.gout || A.gfm);
  const bool wide_on = (wi >= 0) && A.gwide;
  const bool lane_on = deep_on && (e0 < fd.dim);
  const int goff = deep_on ? fd.out_off + (lane_on ? e0 : 0) : 0;
  const bool fold = (A.gfm != nullptr);
  // row strides: a table and its Adagrad state may be strided views of one interleaved slab (dctr.h)
  const int64_t ld_dw = (di >= 0) ? row_ld(fd) : 1, ld_ds = (di >= 0) ? state_ld(fd) : 1;
  const int64_t ld_ww = (wi >= 0) ? row_ld(fw) : 1, ld_ws = (wi >= 0) ? state_ld(fw) : 1;

  // everything an entry contributes: h = g_out + g_fm * S (deep strip), g_fm, g_wide
  // (GEN: entry v = slot * B + b; a pooled slot's strip is the pooled field's gradient g_out + g_fm (S - pooled value) times
  // the pooling weight -- nothing of it is folded at the row, so its g_fm share is 0 -- see gen_entry below)
  auto load_entry = [&](int v, Strip<VEC>& h, float& gf, float& gw) {
    h = strip_zero<VEC>();
    gf = 0.f;
    gw = 0.f;
    if constexpr (GEN) {
      gen_entry<VEC>(A, entry_desc<true>(A, sl, v, 0), lane_on, wide_on && gl == 0, fold, e0, h, gf, gw);
      return;
    }
    const int b = v;
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
  LazyCtx Z = {};
  if constexpr (OPT == DCTR_UPD_LAZY) Z = lazy_ctx(A, u);
  auto load_row = [&](int64_t row, Strip<VEC>& w, Strip<VEC>& s, Strip<VEC>& e, float& ww, float& sw) {
    w = strip_zero<VEC>();
    s = strip_zero<VEC>();
    e = strip_zero<VEC>();
    ww = 0.f;
    sw = 0.f;
    if constexpr (OPT == DCTR_UPD_LAZY) {      // (s carries the first state slab; the second is fetched at the step)
      if (lane_on) {
        w = strip_load<VEC>(Z.deep + row * Z.ld_d + e0);
        if (Z.s1) s = strip_load<VEC>(Z.s1 + row * Z.ld_s1 + e0);
        e = w;
      }
      if (wide_on && gl == 0) {
        ww = ldg_f32(Z.wide + row * Z.ld_w);
        if (Z.ws1) sw = ldg_f32(Z.ws1 + row * Z.ld_ws1);
      }
      return;
    }
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
    if constexpr (OPT == DCTR_UPD_LAZY) {
      if (lane_on) {
        if (fold) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc.v[k] -= accf * e.v[k];
        }
        const Strip<VEC> b2 = Z.s2 ? strip_load<VEC>(Z.s2 + row * Z.dim + e0) : strip_zero<VEC>();
        lazy_apply_strip<VEC>(A, Z, Z.lam2d, Z.deep + row * Z.ld_d + e0, Z.s1 ? Z.s1 + row * Z.ld_s1 + e0 : nullptr,
                              Z.s2 ? Z.s2 + row * Z.dim + e0 : nullptr, acc, w, s, b2);
      }
      if (wide_on && gl == 0) {
        Strip<1> a1, w1, s1, b1;
        a1.v[0] = accw;
        w1.v[0] = ww;
        s1.v[0] = sw;
        b1.v[0] = Z.ws2 ? ldg_f32(Z.ws2 + row) : 0.f;
        lazy_apply_strip<1>(A, Z, Z.lam2w, Z.wide + row * Z.ld_w, Z.ws1 ? Z.ws1 + row * Z.ld_ws1 : nullptr,
                            Z.ws2 ? Z.ws2 + row : nullptr, a1, w1, s1, b1);
      }
      if (gl == 0) *(DCTR_GLOBAL int32_t*)(Z.stamp + row) = Z.t + 1;     // the row has seen step t + 1
      return;
    }
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
      const uint16_t* pt = A.parts_t + static_cast<int64_t>(U.c0) * A.B;
      const uint32_t pp = static_cast<uint32_t>(U.pu);
      auto keep = [&](int b) {
        const int32_t id = clamp_id(ldg_i32(ids + b), vocab);
        uint32_t idq, pu_;
        split_id<GEN>(A, U, static_cast<uint32_t>(id), idq, pu_);
        if ((idq & mmask) == static_cast<uint32_t>(mres)) {
          const int slot = atomicAdd(&n_sh, 1);  // LDS atomic; the order is fixed by the sort below
          if (slot < kCap) keys[slot] = (idq << U.vbits) | static_cast<uint32_t>(b);
        }
      };
      if ((A.B & 7) == 0) {
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
    // (no partition tags: simple units only -- a general unit's tags also say which positions are masked out, the host
    // always supplies them)
    auto take = [&](int32_t raw, int b) {
      const int32_t id = clamp_id(raw, vocab);
      const uint32_t idq = div_p(static_cast<uint32_t>(id), A.pmagic, A.pshift);
      if (static_cast<int>(static_cast<uint32_t>(id) - idq * static_cast<uint32_t>(P)) == p &&
          (idq & mmask) == static_cast<uint32_t>(mres)) {
        const int slot = atomicAdd(&n_sh, 1);  // LDS atomic; the order is fixed by the sort below
        if (slot < kCap) keys[slot] = (idq << U.vbits) | static_cast<uint32_t>(b);
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
      const int idq = static_cast<int>(key >> U.vbits);
      const int64_t row = static_cast<int64_t>(idq) * Pu + U.pu;
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
        const bool seg_end = (rank == n - 1) || (static_cast<int>(skeys[rank + 1] >> U.vbits) != idq);
        if (seg_end) {
          Strip<VEC> acc = strip_zero<VEC>();
          float accf = 0.f, accw = 0.f;
          int r = rank;  // walk back: fixed order => deterministic
          while (r >= 0 && static_cast<int>(skeys[r] >> U.vbits) == idq) {
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
        const int idq = static_cast<int>(key >> U.vbits);
        const int64_t row = static_cast<int64_t>(idq) * Pu + U.pu;
        const bool last_of_tile = have && ((grp == G - 1) || (i == n - 1));
        const bool seg_end = have && ((i == n - 1) || (static_cast<int>(keys[i + 1] >> U.vbits) != idq));
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
        uint32_t idq, pu_;
        split_id<GEN>(A, U, static_cast<uint32_t>(id), idq, pu_);
        bool mine = static_cast<int>(pu_) == U.pu && (idq & mmask) == static_cast<uint32_t>(mres);
        if constexpr (GEN)   // (a masked-out position is no entry: its tag says so)
          mine = mine && *(const DCTR_GLOBAL uint16_t*)(A.parts_t + static_cast<int64_t>(U.c0) * A.B + b) != 0xFFFFu;
        if (mine) {
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
    const int64_t row_hot = static_cast<int64_t>(idq_hot) * Pu + U.pu;
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
      bool m = b < B && clamp_id(ldg_i32(ids + (b < B ? b : 0)), vocab) == id_hot;
      if constexpr (GEN)
        m = m && *(const DCTR_GLOBAL uint16_t*)(A.parts_t + static_cast<int64_t>(U.c0) * A.B + (b < B ? b : 0)) != 0xFFFFu;
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
template <int VEC, int LPR, int OPT, bool GEN>
__global__ __launch_bounds__(kThreads, 5) void k_embed_update(UpdArgs A) {
  step_priority();
  __shared__ dctr_uslot_t sl[GEN ? kMaxSlots : 1];
  if (A.g_wdense && static_cast<int>(blockIdx.x) >= static_cast<int>(gridDim.x) - A.n_wdense) {
    wdense_column(A, static_cast<int>(blockIdx.x) - (static_cast<int>(gridDim.x) - A.n_wdense));
    return;
  }
  // Work item (unit, partition) in plain launch order: consecutive workgroups go to consecutive XCDs, so every
  // XCD gets the same number of working workgroups.
  const int u = static_cast<int>(blockIdx.x) / A.P, p = static_cast<int>(blockIdx.x) - u * A.P;
  if (u >= A.n_units) return;
  if constexpr (GEN) stage_slots(A, unit_ctx<true>(A, u, p), sl, threadIdx.x, kThreads);   // (upd_partition's first barrier)
  upd_partition<VEC, LPR, OPT, GEN>(A, u, p, sl);
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
template <int VEC, int LPR, int OPT, bool GEN>
__global__ __launch_bounds__(kThreads, 5) void k_embed_apply_sorted(UpdArgs A) {
  step_priority();
  constexpr int G = kThreads / LPR;   // entries per tile
  constexpr int RW = LPR * VEC;
  __shared__ dctr_uslot_t sl[GEN ? kMaxSlots : 1];     // the unit's slot descriptors (general units)
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

  const UnitCtx U = unit_ctx<GEN>(A, u, p);
  if constexpr (GEN) stage_slots(A, U, sl, tid, kThreads);   // (visible behind the barrier in front of the first tile)
  const int di = U.di, wi = U.wi;
  const int64_t Pu = static_cast<int64_t>(U.k) * P;
  dctr_field_t fd = {}, fw = {};
  if (di >= 0) fd = uni_field(A.deep[di]);
  if (wi >= 0) fw = uni_field(A.wide[wi]);
  const uint32_t bmask = (1u << U.vbits) - 1u;
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
    upd_partition<VEC, LPR, OPT, GEN>(A, u, p, sl);
    return;
  }
  LazyCtx Z = {};
  if constexpr (OPT == DCTR_UPD_LAZY) Z = lazy_ctx(A, u);
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
    const int idq = static_cast<int>(key >> U.vbits);
    const int64_t row = static_cast<int64_t>(idq) * Pu + U.pu;
    const bool seg_end = have && ((i == n - 1) || (static_cast<int>(knext >> U.vbits) != idq));
    const bool last_of_tile = have && ((grp == G - 1) || (i == n - 1));
    // round trip 2: everything this entry contributes, and (segment ends only) the row it lands on
    // (nothing is USED inside the branches that guard these loads: with `h += gf * S` right behind its loads the compiler
    // waited for them there, and the row loads below left one round trip later -- three dependent trips per tile where
    // two were meant; round 3)
    Strip<VEC> h = strip_zero<VEC>(), w = strip_zero<VEC>(), s = strip_zero<VEC>(), e = strip_zero<VEC>();
    Strip<VEC> S = strip_zero<VEC>();
    Strip<VEC> s2 = strip_zero<VEC>();      // (DCTR_UPD_LAZY: the second state slab's strip)
    float gf = 0.f, gw = 0.f, ww = 0.f, sw = 0.f, sw2 = 0.f;
    if (have) {
      if constexpr (GEN) {
        // (h comes back complete -- pooled value, pooling weight and g_fm S applied; S stays 0 for the sum below)
        gen_entry<VEC>(A, entry_desc<true>(A, sl, b, 0), lane_on, wide_on && gl == 0, fold, e0, h, gf, gw);
      } else {
      if (lane_on) {
        if (A.gout) h = strip_load<VEC>(A.gout + static_cast<int64_t>(b) * A.ldg + goff);
        if (fold) {
          S = strip_load<VEC>(A.fm_s + static_cast<int64_t>(b) * A.lds_ + e0);
          gf = ldg_f32(A.gfm + b);
        }
      }
      if (wide_on && gl == 0) gw = ldg_f32(A.gwide + static_cast<int64_t>(b) * A.ldgw);
      }
      if (seg_end) {
        if constexpr (OPT == DCTR_UPD_LAZY) {
          if (lane_on) {
            w = strip_load<VEC>(Z.deep + row * Z.ld_d + e0);
            if (Z.s1) s = strip_load<VEC>(Z.s1 + row * Z.ld_s1 + e0);
            if (Z.s2) s2 = strip_load<VEC>(Z.s2 + row * Z.dim + e0);
            e = w;
          }
          if (wide_on && gl == 0) {
            ww = ldg_f32(Z.wide + row * Z.ld_w);
            if (Z.ws1) sw = ldg_f32(Z.ws1 + row * Z.ld_ws1);
            if (Z.ws2) sw2 = ldg_f32(Z.ws2 + row);
          }
        } else {
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
    }
    if (lane_on) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) gbuf[grp * RW + e0 + k] = GEN ? h.v[k] : h.v[k] + gf * S.v[k];   // (gf = 0, S = 0 without FM)
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
        if constexpr (OPT == DCTR_UPD_LAZY) {
          if (lane_on) {
            Strip<VEC> a2 = acc;
            if (fold) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) a2.v[k] -= accf * e.v[k];
            }
            lazy_apply_strip<VEC>(A, Z, Z.lam2d, Z.deep + row * Z.ld_d + e0, Z.s1 ? Z.s1 + row * Z.ld_s1 + e0 : nullptr,
                                  Z.s2 ? Z.s2 + row * Z.dim + e0 : nullptr, a2, w, s, s2);
          }
          if (wide_on && gl == 0) {
            Strip<1> a1, w1, s1, b1;
            a1.v[0] = accw;
            w1.v[0] = ww;
            s1.v[0] = sw;
            b1.v[0] = sw2;
            lazy_apply_strip<1>(A, Z, Z.lam2w, Z.wide + row * Z.ld_w, Z.ws1 ? Z.ws1 + row * Z.ld_ws1 : nullptr,
                                Z.ws2 ? Z.ws2 + row : nullptr, a1, w1, s1, b1);
          }
          if (gl == 0) *(DCTR_GLOBAL int32_t*)(Z.stamp + row) = Z.t + 1;   // the row has seen step t + 1
        } else {
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

}  // namespace

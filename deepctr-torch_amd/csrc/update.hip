// update.hip -- the C ABI of the deterministic embedding update (the update kernels themselves: update_kernels.hpp), the
// pre-pass kernels that find and sort every (unit, partition)'s entries, and the GEN = false (simple units) instantiations.
#include "update_kernels.hpp"

namespace dctr {
// update_gen.hip: the GEN = true instantiations (general units: pooled VarLen fields, shared tables).  `args`: an UpdArgs
// (the struct lives in each translation unit's anonymous namespace, hence the untyped pointer).
int launch_update_gen(const void* args, int vec, int lpr, int opt, unsigned grid_x, hipStream_t s);
}

namespace {

// ---- optional pre-pass: bucket the (unit, sample) entries by partition ---------------------------------------------
// One thread per entry; a bucket's fill order is whatever the atomics give (the update kernel sorts the keys).
// Pays off when a workgroup's scan over the unit's B ids is the expensive part, i.e. for large (global) batches.
// (GEN: one thread per (virtual column, sample); a masked-out position -- tag 0xFFFF -- is no entry; the entry goes to the
// bucket (first vunit of its unit) * P + its partition of the unit: a unit's vunits are consecutive)
template <bool GEN>
__global__ __launch_bounds__(kThreads) void k_bucket(UpdArgs A) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= static_cast<int64_t>(A.n_vcols) * A.B) return;
  const int u = static_cast<int>(i / A.B), b = static_cast<int>(i - static_cast<int64_t>(u) * A.B);
  if constexpr (GEN) {
    if (*(const DCTR_GLOBAL uint16_t*)(A.parts_t + i) == 0xFFFFu) return;
    const int vu0 = A.slots[u].vu0;
    const dctr_vunit_t d = A.vunits[vu0];
    UnitCtx U;
    U.di = d.di; U.wi = d.wi; U.c0 = d.c0; U.ns = d.n_slots; U.k = d.k; U.pu = 0; U.kshift = d.kshift; U.kmagic = d.kmagic;
    const int nv = U.ns * A.B;
    U.vbits = 32 - __builtin_clz(static_cast<unsigned>((nv < 2 ? 2 : nv) - 1));
    const int64_t vocab = (d.di >= 0) ? A.deep[d.di].vocab : A.wide[d.wi].vocab;
    const int32_t id = clamp_id(ldg_i32(A.ids_t + i), vocab);
    uint32_t idq, pu;
    split_id<true>(A, U, static_cast<uint32_t>(id), idq, pu);
    const int64_t bucket = static_cast<int64_t>(vu0) * A.P + pu;
    const int slot = atomicAdd(A.bcnt + bucket, 1);
    if (slot < kBucket)
      A.bkeys[bucket * kBucket + slot] = (idq << U.vbits) | static_cast<uint32_t>((u - U.c0) * A.B + b);
    return;
  }
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
template <bool FROM_BUCKETS, bool GEN>
__global__ __launch_bounds__(kThreads) void k_embed_segments(UpdArgs A) {
  step_priority();
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
  const UnitCtx U = unit_ctx<GEN>(A, u, p);
  const int B = GEN ? U.ns * A.B : A.B;      // the unit's entries v = slot * B + b
  int n;
  if (FROM_BUCKETS) {
    n = uni(*(const DCTR_GLOBAL int32_t*)cnt);
    if (n > kBucket || n <= 0) return;
#pragma unroll
    for (int q = 0; q < kSl; ++q)
      if (tid + q * kThreads < n) keys[tid + q * kThreads] = *(const DCTR_GLOBAL uint32_t*)(dst + tid + q * kThreads);
    __syncthreads();
  } else {
    const int di = U.di, wi = U.wi;
    const int64_t vocab = uni((di >= 0) ? A.deep[di].vocab : A.wide[wi].vocab);
    const int32_t* ids = A.ids_t + static_cast<int64_t>(U.c0) * A.B;
    const uint16_t* pt = A.parts_t + static_cast<int64_t>(U.c0) * A.B;
    const uint32_t pp = static_cast<uint32_t>(U.pu);
    if (tid == 0) n_sh = 0;
    __syncthreads();
    // phase 1: the samples whose partition tag is mine (no dependent loads inside the divergent branches)
    auto keep = [&](int b) {
      const int slot = atomicAdd(&n_sh, 1);
      if (slot < kBucket) keys[slot] = static_cast<uint32_t>(b);
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
        uint32_t idq, pu_;
        split_id<GEN>(A, U, static_cast<uint32_t>(id), idq, pu_);
        key[q] = (idq << U.vbits) | static_cast<uint32_t>(b);
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
  step_priority();
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
  step_priority();
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
  step_priority();
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

// The same for GENERAL units (dctr_plan_ext_t): one thread per (virtual column, sample).  A position that the pooling masks
// out (sum / mean: id == 0 in mask mode, t >= length in length mode -- inputs.py:146, sequence.py:56-59) gets the tag 0xFFFF
// and is no entry of the update; a max-pooled position always is one (its gradient is masked per element by the forward's
// arg-max).  The thread of a mean-pooled field's first position also writes the divisor count + 1e-8 (sequence.py:72-74).
__global__ __launch_bounds__(kThreads) void k_embed_ids_gen(const dctr_field_t* __restrict__ deep,
                                                            const dctr_field_t* __restrict__ wide,
                                                            const dctr_uslot_t* __restrict__ slots,
                                                            const dctr_vunit_t* __restrict__ vunits, int n_vcols,
                                                            const float* __restrict__ X, int64_t ldx, int B,
                                                            int32_t* __restrict__ ids_t, uint16_t* __restrict__ parts_t,
                                                            float* __restrict__ den_t, UpdArgs A) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= static_cast<int64_t>(n_vcols) * B) return;
  const int c = static_cast<int>(i / B), b = static_cast<int>(i - static_cast<int64_t>(c) * B);
  const dctr_uslot_t s = slots[c];
  const float* xr = X + static_cast<int64_t>(b) * ldx;
  const int32_t rid = static_cast<int32_t>(xr[s.col]);
  ids_t[i] = rid;
  const bool by_len = s.len_col >= 0;
  const int32_t len_i = by_len ? static_cast<int32_t>(xr[s.len_col]) : 0;
  bool valid = true;
  if (s.pool == DCTR_POOL_SUM || s.pool == DCTR_POOL_MEAN) valid = by_len ? (s.t < len_i) : (rid != 0);
  if (den_t && s.den >= 0 && s.t == 0) {
    float cnt = 0.f;
    if (by_len) {
      cnt = static_cast<float>(len_i);
    } else {
      for (int t = 0; t < s.len; ++t) cnt += (static_cast<int32_t>(xr[s.col + t]) != 0) ? 1.f : 0.f;
    }
    den_t[static_cast<int64_t>(s.den) * B + b] = cnt + 1e-8f;
  }
  if (parts_t) {
    uint32_t tag = 0xFFFFu;
    if (valid) {
      const dctr_vunit_t d = vunits[s.vu0];
      UnitCtx U;
      U.di = d.di; U.wi = d.wi; U.c0 = d.c0; U.ns = d.n_slots; U.k = d.k; U.pu = 0; U.vbits = 0; U.kshift = d.kshift;
      U.kmagic = d.kmagic;
      const int64_t vocab = (d.di >= 0) ? deep[d.di].vocab : wide[d.wi].vocab;
      uint32_t idq;
      split_id<true>(A, U, static_cast<uint32_t>(clamp_id(rid, vocab)), idq, tag);
    }
    parts_t[i] = static_cast<uint16_t>(tag);
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

namespace {
// q / d == (q * magic) >> shift for 0 <= q < 2^31 (the scheme of div_p)
void magic_of(unsigned d, uint64_t* magic, int* shift) {
  *shift = 32 + ceil_log2(d);
  *magic = static_cast<uint64_t>((static_cast<unsigned __int128>(1) << *shift) / d) + 1;
}
// the partition arithmetic every kernel of a launch shares
void fill_partition_args(UpdArgs& a, int B, int P) {
  a.B = B;
  a.P = P;
  a.bbits = ceil_log2(B < 2 ? 2 : B);
  magic_of(static_cast<unsigned>(P), &a.pmagic, &a.pshift);
  magic_of(static_cast<unsigned>(B), &a.bmagic, &a.bshift);
}
// general units: the ext block's pointers -> kernel arguments; n_units / n_vcols = what the grid runs over
void fill_ext_args(UpdArgs& a, const dctr_plan_t* plan, int32_t n_units) {
  const dctr_plan_ext_t* x = plan->ext;
  a.slots = x ? x->slots : nullptr;
  a.vunits = x ? x->vunits : nullptr;
  a.den_t = x ? x->den_t : nullptr;
  a.amax = x ? x->amax : nullptr;
  a.ld_am = x ? x->ld_amax : 0;
  a.n_units = x ? x->n_vunits : n_units;
  a.n_vcols = x ? x->n_vcols : n_units;
}
}  // namespace

extern "C" int dctr_embed_ids(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, const float* X,
                              int64_t ldx, int32_t B, int32_t* ids_t, uint16_t* parts_t, dctr_stream_t stream) {
  if (!units || !X || !ids_t || n_units <= 0 || B < 0) return DCTR_EINVAL;
  if (parts_t && !plan) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  const int n_parts = parts_t ? dctr_embed_update_partitions(plan, B) : 0;
  if (parts_t && (n_parts <= 0 || n_parts > 65535)) return DCTR_ENOSUP;
  if (plan && plan->ext) {      // general units: ids_t / parts_t are [n_vcols, B]; den_t rides along
    const dctr_plan_ext_t* x = plan->ext;
    if (n_units != x->n_vunits || !x->slots || !x->vunits || x->n_vcols <= 0) return DCTR_EINVAL;
    if (x->n_den > 0 && !x->den_t) return DCTR_EINVAL;
    UpdArgs a = {};
    fill_partition_args(a, B, parts_t ? n_parts : 1);
    const int64_t n = static_cast<int64_t>(x->n_vcols) * B;
    k_embed_ids_gen<<<dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                      static_cast<hipStream_t>(stream)>>>(plan->deep, plan->wide, x->slots, x->vunits, x->n_vcols, X, ldx, B,
                                                          ids_t, parts_t, x->den_t, a);
    return launch_status();
  }
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
  if (plan->vec != 1 && plan->vec != 2 && plan->vec != 4) return 0;
  if (plan->n_deep > 0 && plan->max_dim > 64 * plan->vec) return 0;
  if (B > (1 << 20)) return 0;
  if (max_vocab >= (int64_t(1) << 31)) return 0;
  if (plan->ext) {
    // general units (pooled fields, shared tables): every vunit's keys (id / (k P), slot * B + b) must fit 32 bits, its
    // partitions 16-bit tags, its slots the LDS staging
    const dctr_plan_ext_t* x = plan->ext;
    if (!x->h_vunits || !x->h_vocab || x->n_vunits <= 0 || x->max_unit_slots > kMaxSlots) return 0;
    const int P = dctr_embed_update_partitions(plan, B);
    for (int i = 0; i < x->n_vunits; ++i) {
      const dctr_vunit_t& d = x->h_vunits[i];
      const int64_t nv = static_cast<int64_t>(d.n_slots) * B;
      const int64_t Pu = static_cast<int64_t>(d.k) * P;
      if (d.n_slots < 1 || d.n_slots > kMaxSlots || d.k < 1 || nv > (int64_t(1) << 24) || Pu > 65535) return 0;
      if (ceil_log2(x->h_vocab[i] / Pu + 2) + ceil_log2(nv < 2 ? 2 : nv) > 32) return 0;
    }
    return 1;
  }
  // without the ext block a unit is one X column over its own tables
  if (plan->n_deep != plan->n_deep_fixed || plan->n_wide != plan->n_wide_fixed) return 0;
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
  if (plan->ext) return static_cast<int64_t>(plan->ext->n_vunits) * P * (1 + kBucket);   // (no two-level staging)
  if (B >= kTwoLevelMin && stage_layout(n_units, B, P).n_bins <= kBinsMax) return stage_layout(n_units, B, P).total;
  return static_cast<int64_t>(n_units) * P * (1 + kBucket);
}

extern "C" int dctr_embed_segments(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, int64_t max_vocab,
                                   const int32_t* ids_t, const uint16_t* parts_t, int32_t B, int32_t* workspace,
                                   int64_t workspace_ints, dctr_stream_t stream) {
  if (!plan || !units || !ids_t || !workspace || n_units <= 0 || B < 0) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  if (!dctr_embed_update_supported(plan, max_vocab, B)) return DCTR_ENOSUP;
  const bool gen = plan->ext != nullptr;
  if (gen && (n_units != plan->ext->n_vunits || !parts_t)) return DCTR_EINVAL;
  int vec, lpr;
  lane_layout(plan, &vec, &lpr);
  const int P = pick_p(B, kThreads / lpr);
  const int64_t nbuckets = static_cast<int64_t>(n_units) * P;
  if (workspace_ints < nbuckets * (1 + kBucket)) return DCTR_EINVAL;
  UpdArgs a = {};
  a.wd_step.kind = -1;
  a.deep = plan->deep; a.wide = plan->wide; a.units = units; a.ids_t = ids_t; a.parts_t = parts_t;
  fill_partition_args(a, B, P);
  fill_ext_args(a, plan, n_units);
  a.bcnt = workspace;
  a.bkeys = reinterpret_cast<uint32_t*>(workspace + nbuckets);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(nbuckets)), block(kThreads);
  if (gen) {
    // general units: tag scan while a unit's n_slots * B tags are a short read, else one global atomic per entry
    if (P <= 65535 && static_cast<int64_t>(plan->ext->max_unit_slots) * B <= 65536) {
      k_embed_segments<false, true><<<grid, block, 0, s>>>(a);
    } else {
      const int64_t ne = static_cast<int64_t>(a.n_vcols) * B;
      k_bucket<true><<<dim3(static_cast<unsigned>((ne + kThreads - 1) / kThreads)), block, 0, s>>>(a);
      const int st = launch_status();
      if (st != DCTR_OK) return st;
      k_embed_segments<true, true><<<grid, block, 0, s>>>(a);
    }
    return launch_status();
  }
  // small batches: every workgroup finds its entries by comparing the forward's 16-bit partition tags (B / 8 vector
  // loads per workgroup).  Large batches: the two-level pre-pass (chunks re-ordered by coarse bin, then one workgroup
  // per bin), when the caller's workspace has room for its staging; else (or without tags): one global atomic per entry.
  const StageLayout SL = stage_layout(n_units, B, P);
  if (two_level_applies(SL, B, P, workspace_ints)) {
    return launch_two_level(a, SL, workspace, s);
  } else if (parts_t && P <= 65535 && B <= 32768) {
    k_embed_segments<false, false><<<grid, block, 0, s>>>(a);
  } else {
    const int64_t ne = static_cast<int64_t>(n_units) * B;
    k_bucket<false><<<dim3(static_cast<unsigned>((ne + kThreads - 1) / kThreads)), block, 0, s>>>(a);
    const int st = launch_status();
    if (st != DCTR_OK) return st;
    k_embed_segments<true, false><<<grid, block, 0, s>>>(a);
  }
  return launch_status();
}

namespace {
int embed_update_impl(const dctr_plan_t* plan, const int32_t* units, int32_t n_units,
                      int64_t max_vocab, const int32_t* ids_t, const uint16_t* parts_t, int32_t B,
                      const float* g_out,
                      int64_t ld_g, const float* out, int64_t ld_out, const float* fm_s,
                      int64_t ld_s, const float* g_fm, const float* g_wide, int64_t ld_gw,
                      int32_t opt, float lr, float eps, const float* X, int64_t ld_x, float* g_wdense,
                      const dctr_dense_step_t* wdense_step, int32_t* workspace, int64_t workspace_ints,
                      int32_t presorted, dctr_stream_t stream, const dctr_lazy_unit_t* lz, const int32_t* lz_step,
                      const dctr_lazy_opt_t* lz_opt);
}  // namespace

extern "C" int dctr_embed_update(const dctr_plan_t* plan, const int32_t* units, int32_t n_units,
                                 int64_t max_vocab, const int32_t* ids_t, const uint16_t* parts_t, int32_t B,
                                 const float* g_out,
                                 int64_t ld_g, const float* out, int64_t ld_out, const float* fm_s,
                                 int64_t ld_s, const float* g_fm, const float* g_wide, int64_t ld_gw,
                                 int32_t opt, float lr, float eps, const float* X, int64_t ld_x, float* g_wdense,
                                 const dctr_dense_step_t* wdense_step, int32_t* workspace, int64_t workspace_ints,
                                 int32_t presorted, dctr_stream_t stream) {
  if (opt == DCTR_UPD_LAZY) return DCTR_EINVAL;       // (dctr_embed_update_lazy)
  return embed_update_impl(plan, units, n_units, max_vocab, ids_t, parts_t, B, g_out, ld_g, out, ld_out, fm_s, ld_s, g_fm,
                           g_wide, ld_gw, opt, lr, eps, X, ld_x, g_wdense, wdense_step, workspace, workspace_ints, presorted,
                           stream, nullptr, nullptr, nullptr);
}

extern "C" int dctr_embed_update_lazy(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, int64_t max_vocab,
                                      const int32_t* ids_t, const uint16_t* parts_t, int32_t B, const float* g_out,
                                      int64_t ld_g, const float* out, int64_t ld_out, const float* fm_s, int64_t ld_s,
                                      const float* g_fm, const float* g_wide, int64_t ld_gw, const float* X, int64_t ld_x,
                                      float* g_wdense, int32_t* workspace, int64_t workspace_ints,
                                      const dctr_lazy_unit_t* lazy_units, const int32_t* step, const dctr_lazy_opt_t* lazy_opt,
                                      dctr_stream_t stream) {
  if (!lazy_units || !step || !lazy_opt) return DCTR_EINVAL;
  if (plan && plan->ext) return DCTR_ENOSUP;            // (simple units only, like csrc/lazy.hip)
  return embed_update_impl(plan, units, n_units, max_vocab, ids_t, parts_t, B, g_out, ld_g, out, ld_out, fm_s, ld_s, g_fm,
                           g_wide, ld_gw, DCTR_UPD_LAZY, 0.f, 0.f, X, ld_x, g_wdense, nullptr, workspace, workspace_ints, 1,
                           stream, lazy_units, step, lazy_opt);
}

namespace {
int embed_update_impl(const dctr_plan_t* plan, const int32_t* units, int32_t n_units,
                      int64_t max_vocab, const int32_t* ids_t, const uint16_t* parts_t, int32_t B,
                      const float* g_out,
                      int64_t ld_g, const float* out, int64_t ld_out, const float* fm_s,
                      int64_t ld_s, const float* g_fm, const float* g_wide, int64_t ld_gw,
                      int32_t opt, float lr, float eps, const float* X, int64_t ld_x, float* g_wdense,
                      const dctr_dense_step_t* wdense_step, int32_t* workspace, int64_t workspace_ints,
                      int32_t presorted, dctr_stream_t stream, const dctr_lazy_unit_t* lz, const int32_t* lz_step,
                      const dctr_lazy_opt_t* lz_opt) {
  // (out / ld_out: the forward's rows.  Fixed-length fields never re-read them -- FM is folded algebraically at the row --
  // a POOLED field's FM backward needs its pooled value, which is no table row: general units read it there)
  if (!plan || !units || !ids_t || n_units <= 0 || B < 0) return DCTR_EINVAL;
  const bool gen = plan->ext != nullptr;
  if (gen) {
    const dctr_plan_ext_t* x = plan->ext;
    if (n_units != x->n_vunits || !parts_t || !x->slots || !x->vunits) return DCTR_EINVAL;
    if (g_fm && !out) return DCTR_EINVAL;
    if (x->n_den > 0 && !x->den_t) return DCTR_EINVAL;
    if (x->ld_amax > 0 && !x->amax) return DCTR_EINVAL;
  }
  if (g_wide && ld_gw < 1) return DCTR_EINVAL;
  if (g_wdense && (!X || !g_wide || plan->n_wdense <= 0 || !plan->wdense_cols)) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  if (opt != DCTR_UPD_SGD && opt != DCTR_UPD_ADAGRAD && opt != DCTR_UPD_ACCUM && opt != DCTR_UPD_LAZY) return DCTR_EINVAL;
  if (opt == DCTR_UPD_LAZY) {
    if (!lz || !lz_step || !lz_opt || gen || !presorted) return DCTR_EINVAL;
    if (lz_opt->kind != DCTR_LAZY_SGD && lz_opt->kind != DCTR_LAZY_ADAGRAD && lz_opt->kind != DCTR_LAZY_ADAM &&
        lz_opt->kind != DCTR_LAZY_RMSPROP)
      return DCTR_EINVAL;
  }
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
  UpdArgs a = {};
  a.deep = plan->deep; a.wide = plan->wide; a.units = units; a.ids_t = ids_t;
  a.parts_t = nullptr;
  a.gout = g_out; a.fm_s = fm_s; a.gfm = g_fm; a.gwide = g_wide; a.ldgw = ld_gw;
  a.ldg = ld_g; a.lds_ = ld_s;
  a.out = out; a.ldo = ld_out;
  a.lr = lr; a.eps = eps;
#ifdef DCTR_DIAG
  a.trace = g_trace;
#else
  a.trace = nullptr;
#endif
  a.X = X; a.ldx = ld_x; a.wdense_cols = plan->wdense_cols; a.n_wdense = plan->n_wdense; a.g_wdense = g_wdense;
  a.wd_step = dense_step_dev(g_wdense ? wdense_step : nullptr);
  a.lz = lz; a.lz_step = lz_step;
  if (lz_opt) a.lzo = dctr_lazy::opt_const(lz_opt);

  const int P = pick_p(B, kThreads / lpr);
  fill_partition_args(a, B, P);
  fill_ext_args(a, plan, n_units);
  if (parts_t && P <= 65535) a.parts_t = parts_t;   // (taken modulo the same P: dctr_embed_update_partitions)
  if (gen && !a.parts_t) return DCTR_ENOSUP;
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
    } else if (gen) {
      // general units without a pre-pass: bucket the entries (one global atomic each); the update kernel sorts
      const int64_t ne = static_cast<int64_t>(a.n_vcols) * B;
      k_bucket<true><<<dim3(static_cast<unsigned>((ne + kThreads - 1) / kThreads)), block, 0, s>>>(a);
      const int st = launch_status();
      if (st != DCTR_OK) return st;
    } else if (two_level_applies(SL, B, P, workspace_ints)) {
      // a large batch without a pre-pass (the owners' update of the table-sharded step): the two-level pre-pass here, in
      // line, then the lean pre-sorted kernel -- instead of one global atomic per entry and every workgroup's own sort
      const int st = launch_two_level(a, SL, workspace, s);
      if (st != DCTR_OK) return st;
      a.presorted = 1;
    } else {
      const int64_t ne = static_cast<int64_t>(n_units) * B;
      k_bucket<false><<<dim3(static_cast<unsigned>((ne + kThreads - 1) / kThreads)), block, 0, s>>>(a);
      const int st = launch_status();
      if (st != DCTR_OK) return st;
    }
  }
  if (gen) return dctr::launch_update_gen(&a, vec, lpr, opt, grid.x, s);    // (update_gen.hip: the GEN = true instantiations)

#define DCTR_UPD_GEN false
#include "update_launch.inc"
#undef DCTR_UPD_GEN
  return launch_status();
}
}  // namespace

// embed.hip -- fused multi-table embedding lookup / scatter for gfx950 (MI355X).
//
// What the reference does with 2 x n_fields aten::embedding calls, n cat/sum/pow/mul launches and a
// dense [V, D] embedding_dense_backward per table (inputs.py:141-227, basemodel.py:63-92,354-380,
// interaction.py:26-34, sequence.py:49-77) is one gather kernel and one scatter kernel here.
//
// Mapping (HBM-bound, latency-critical at batch 4096):
//   * a workgroup is 4 waves (one per SIMD of a CU) that share SPB = 64 / LPR consecutive samples;
//   * LPR lanes of a wave form a "sample group"; lane g of the group owns floats [g*VEC, g*VEC+VEC)
//     of every embedding row, so a D=16 row is 4 lanes x dwordx4 = one 64-byte HBM burst;
//   * the 4 waves split the FIELDS (wave w takes fields w, w+4, ...): the per-wave dependent chain
//     (descriptor -> id -> address -> load) is 4x shorter and all 4 SIMDs issue loads;
//   * the workgroup first copies its X tile (ids + dense values, contiguous in memory) and the field
//     descriptors into LDS with coalesced loads -> one HBM round trip for all ids;
//   * every lane then issues ALL its row loads (branch-free, CH fields) before consuming any, so the
//     gather costs about one more HBM round trip;
//   * FM / wide-logit partials of the 4 waves meet in LDS; the final reduction inside a sample group
//     uses DPP shuffles.
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kNW = 4;  // waves per workgroup
constexpr int kThreads = kNW * kWave;
constexpr int kFieldWords = sizeof(dctr_field_t) / 4;
static_assert(sizeof(dctr_field_t) == 64, "dctr_field_t must be 64 bytes");
static_assert(sizeof(dctr_plan_t) == 120, "dctr_plan_t layout changed: update the Python binding");

struct Tile {
  const dctr_field_t* deep;
  const dctr_field_t* wide;
  const float* xs;  // [SPB][n_xcols]
  float* red;       // [kNW][kWave][RED] cross-wave reduction scratch
};

// Copy descriptors and the X tile of samples [b0, b0+nrows) into LDS (whole workgroup).
__device__ __forceinline__ Tile stage_tile(const dctr_plan_t& P, const float* __restrict__ X,
                                           int64_t ldx, int b0, int nrows, int spb,
                                           unsigned char* smem) {
  const int tid = threadIdx.x;
  uint32_t* w = reinterpret_cast<uint32_t*>(smem);
  const int nd = P.n_deep * kFieldWords, nw = P.n_wide * kFieldWords;
  const uint32_t* gd = reinterpret_cast<const uint32_t*>(P.deep);
  const uint32_t* gw = reinterpret_cast<const uint32_t*>(P.wide);
  for (int i = tid; i < nd; i += kThreads) w[i] = gd[i];
  for (int i = tid; i < nw; i += kThreads) w[nd + i] = gw[i];
  float* xs = reinterpret_cast<float*>(w + nd + nw);
  const int nc = P.n_xcols;
  const int n = nrows * nc;
  if (ldx == nc) {
    const float* src = X + static_cast<int64_t>(b0) * ldx;
    for (int i = tid; i < n; i += kThreads) xs[i] = src[i];
  } else {
    for (int i = tid; i < n; i += kThreads) {
      const int r = i / nc, c = i - r * nc;
      xs[i] = X[static_cast<int64_t>(b0 + r) * ldx + c];
    }
  }
  __syncthreads();
  Tile t;
  t.deep = reinterpret_cast<const dctr_field_t*>(w);
  t.wide = reinterpret_cast<const dctr_field_t*>(w + nd);
  t.xs = xs;
  t.red = xs + ((spb * nc + 3) & ~3);
  return t;
}

// id = Tensor.long() of the float in X: truncation toward zero (basemodel.py:369).  float32 holds
// integers exactly only below 2^24, so a 32-bit convert (one v_cvt_i32_f32) is exact
// for every id the reference can represent.
__device__ __forceinline__ int64_t raw_id(const float* xr, int col) {
  return static_cast<int64_t>(static_cast<int32_t>(xr[col]));
}

// Out-of-range ids read row 0 and raise `bad`; the caller ORs it into the error word once.
__device__ __forceinline__ int64_t checked(int64_t id, int64_t vocab, int& bad) {
  const bool oob = static_cast<uint64_t>(id) >= static_cast<uint64_t>(vocab);
  bad |= oob ? 1 : 0;
  return oob ? 0 : id;
}

// Pool one VarLen field for this lane's strip of the row.  Mirrors SequencePoolingLayer.forward
// (sequence.py:49-77) as called from get_varlen_pooling_list (inputs.py:141-155):
//   mask mode   (len_col < 0): m_t = (id_t != 0); length = sum_t m_t
//   length mode (len_col >= 0): m_t = (t < length)
//   sum : sum_t m_t e_t        mean: that / (length + 1e-8)        max: max_t (e_t - (1 - m_t) * 1e9)
//   am (nullable; max pooling): this sample's arg-max bytes of the field at e0 -- the position of the FIRST maximum per
//   element (torch.max's backward routes the gradient there), the side output dctr_embed_update reads
template <int VEC>
__device__ __forceinline__ Strip<VEC> pool_field(const dctr_field_t& fd, const float* xr, int e0,
                                                 bool act, int& bad, uint8_t* am = nullptr) {
  const bool by_len = fd.len_col >= 0;
  const int64_t len_i = by_len ? raw_id(xr, fd.len_col) : 0;
  Strip<VEC> acc;
  int arg[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    acc.v[i] = (fd.pool == DCTR_POOL_MAX) ? -INFINITY : 0.f;
    arg[i] = 0;
  }
  float cnt = 0.f;
  for (int t = 0; t < fd.len; ++t) {
    const int64_t rid = raw_id(xr, fd.col + t);
    const bool m = by_len ? (static_cast<int64_t>(t) < len_i) : (rid != 0);
    const int64_t id = checked(rid, fd.vocab, bad);
    Strip<VEC> row = act ? strip_load<VEC>(fd.table + id * row_ld(fd) + e0) : strip_zero<VEC>();
    if (fd.pool == DCTR_POOL_MAX) {
      const float pen = m ? 0.f : 1e9f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float v = row.v[i] - pen;
        arg[i] = (v > acc.v[i]) ? t : arg[i];
        acc.v[i] = (v > acc.v[i]) ? v : acc.v[i];
      }
    } else if (m) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.v[i] += row.v[i];
    }
    cnt += m ? 1.f : 0.f;
  }
  if (fd.pool == DCTR_POOL_MEAN) {
    const float den = (by_len ? static_cast<float>(len_i) : cnt) + 1e-8f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.v[i] = acc.v[i] / den;
  }
  if (am && act && fd.pool == DCTR_POOL_MAX) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) *(DCTR_GLOBAL uint8_t*)(am + i) = static_cast<uint8_t>(arg[i]);
  }
  return acc;
}

// a strip into an LDS row (16-byte aligned when VEC == 4)
template <int VEC>
__device__ __forceinline__ void strip_put(float* p, const Strip<VEC>& s) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{s.v[0], s.v[1], s.v[2], s.v[3]};
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = s.v[i];
  }
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
template <int VEC, int LPR>
__global__ __launch_bounds__(kThreads) void k_embed_fwd(dctr_plan_t P, const float* __restrict__ X,
                                                        int64_t ldx, int B, float* __restrict__ out,
                                                        int64_t ldo, float* __restrict__ wide,
                                                        int64_t ldw, float* __restrict__ fm, int32_t* err,
                                                        const int32_t* __restrict__ units, int n_units,
                                                        int32_t* __restrict__ ids_t,
                                                        uint16_t* __restrict__ parts_t, int n_parts,
                                                        float* __restrict__ fm_s, int64_t lds_, int stage_off,
                                                        uint8_t* __restrict__ amax, int64_t ld_am,
                                                        const int32_t* __restrict__ am_deep_off,
                                                        const int32_t* __restrict__ am_wide_off) {
  // No fused multiply-adds in this body: the fused train launch (csrc/mlp.hip, dctr_embed_tower_train_step) computes the
  // same linear logit / FM term / sum_f e inside the tower kernel and must land on the same bits -- with contraction left
  // to the compiler the two bodies were contracted differently (round 4: losses equal for 30 steps, then off by one ulp).
#pragma clang fp contract(off)
  step_priority();
  constexpr int SPB = kWave / LPR;
  constexpr int CH = 8;   // row loads in flight per lane and per pass (x4 waves = 32 fields)
  constexpr int WCH = 2;  // wide loads in flight per lane and per pass
  constexpr int RED = 2 * VEC + 1;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wv_id = tid >> 6, lane = tid & 63;
  const int grp = lane / LPR, gl = lane % LPR;
  const int b0 = blockIdx.x * SPB;
  const int nrows = min(SPB, B - b0);
  // P.step_sync: the tower that consumes this launch's outputs runs on another queue behind a dctr_step_wait; they are
  // signalled complete at the end of the kernel and must have left the chip's caches by then.
  //  * stage_off != 0: the tile's output rows are assembled in LDS (zero-filled first: the padding columns too) and
  //    written out as whole rows -- 1 KB per wave-instruction -- with write-through stores.  (Write-through stores of the
  //    64-byte field slices themselves, 16 rows x 64 bytes per instruction, took the kernel from 12 to 27 us.)
  //  * otherwise plain stores and an agent-scope release (this XCD's L2 written back) in front of the signal.
  const bool thru = P.step_sync != nullptr, staged = stage_off != 0;
  float* lrows = reinterpret_cast<float*>(smem + stage_off);
  if (staged) {
    f32x4* z = reinterpret_cast<f32x4*>(lrows);
    for (int i = tid; i < SPB * static_cast<int>(ldo >> 2); i += kThreads) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const Tile T = stage_tile(P, X, ldx, b0, nrows, SPB, smem);
  const bool valid = grp < nrows;
  const int g = valid ? grp : 0;  // idle groups shadow group 0; their stores are masked
  const int b = b0 + g;
  const float* xr = T.xs + g * P.n_xcols;
  const int e0 = gl * VEC;
  int bad = 0;
  // P.out_chunks: output rows go to per-rank buffers (peer memory: the sharded step's push-style rows exchange)
  const auto row_of = [&](int row) -> float* {
    if (!P.out_chunks) return out + static_cast<int64_t>(row) * ldo;
    const int r = row / P.chunk_rows;
    return reinterpret_cast<float*>(P.out_chunks[r]) + static_cast<int64_t>(row - r * P.chunk_rows) * ldo;
  };
  float* orow = out ? row_of(b) : nullptr;
  float* lrow = lrows + g * ldo;

  // side output for dctr_embed_update: the ids of this tile, transposed to [unit][b] (64-byte runs)
  // (+ parts_t: the partition of dctr_embed_update each entry belongs to, so that its workgroups compare 16-bit tags
  // instead of dividing every id of the unit again)
  if (ids_t) {
    for (int k = tid; k < n_units * nrows; k += kThreads) {
      const int u = k / nrows, r = k - u * nrows;
      const int32_t id = static_cast<int32_t>(T.xs[r * P.n_xcols + ldg_i32(units + 4 * u + 2)]);
      ids_t[static_cast<int64_t>(u) * B + b0 + r] = id;
      if (parts_t) {
        const int di = ldg_i32(units + 4 * u), wi = ldg_i32(units + 4 * u + 1);
        const int64_t vocab = (di >= 0) ? T.deep[di].vocab : T.wide[wi].vocab;
        const uint32_t cid = (static_cast<uint64_t>(static_cast<int64_t>(id)) >= static_cast<uint64_t>(vocab))
                                 ? 0u : static_cast<uint32_t>(id);
        parts_t[static_cast<int64_t>(u) * B + b0 + r] = static_cast<uint16_t>(cid % static_cast<uint32_t>(n_parts));
      }
    }
  }

  // ---- wide (1-dim) tables: (wave, lane-in-group) pairs split the fields; loads issued first ----
  // Branch-free: slots past the last field re-read the last field and are masked when summed, so
  // each pass is one basic block and the scheduler can overlap every LDS/HBM access.
  float ws = 0.f;
  const int nwf = wide ? P.n_wide_fixed : 0;
  float wval[WCH];
#pragma unroll
  for (int k = 0; k < WCH; ++k) wval[k] = 0.f;
  if (nwf > 0) {
#pragma unroll
    for (int k = 0; k < WCH; ++k) {
      const int f = (k * kNW + wv_id) * LPR + gl;
      const dctr_field_t& fd = T.wide[min(f, nwf - 1)];
      wval[k] = ldg_f32(fd.table + checked(raw_id(xr, fd.col), fd.vocab, bad) * row_ld(fd));
    }
  }

  // ---- deep fixed-length fields: issue every row load of a pass, then consume -------------------
  Strip<VEC> S = strip_zero<VEC>(), Q = strip_zero<VEC>();
  const int nfix = out ? P.n_deep_fixed : 0;
  for (int f0 = wv_id; f0 < nfix; f0 += kNW * CH) {
    Strip<VEC> r[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const dctr_field_t& fd = T.deep[min(f0 + k * kNW, nfix - 1)];
      const int64_t id = checked(raw_id(xr, fd.col), fd.vocab, bad);
      // lanes past the row width re-read the row's first strip; their value is never used
      r[k] = strip_load<VEC>(fd.table + id * row_ld(fd) + ((e0 < fd.dim) ? e0 : 0));
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const dctr_field_t& fd = T.deep[min(f0 + k * kNW, nfix - 1)];
      const bool live = (f0 + k * kNW < nfix) && (e0 < fd.dim);
      if (live && valid) {
        if (staged) strip_put<VEC>(lrow + fd.out_off + e0, r[k]);
        else strip_store<VEC>(orow + fd.out_off + e0, r[k]);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float v = live ? r[k].v[i] : 0.f;
        S.v[i] += v;
        Q.v[i] += v * v;
      }
    }
  }

  // ---- deep VarLen fields ----------------------------------------------------------------------
  if (out) {
    for (int f = P.n_deep_fixed + wv_id; f < P.n_deep; f += kNW) {
      const dctr_field_t& fd = T.deep[f];
      const bool act = e0 < fd.dim;
      uint8_t* am = nullptr;
      if (amax && valid && fd.pool == DCTR_POOL_MAX) {
        const int off = ldg_i32(am_deep_off + f);
        if (off >= 0) am = amax + static_cast<int64_t>(b) * ld_am + off + e0;
      }
      const Strip<VEC> p = pool_field<VEC>(fd, xr, e0, act, bad, am);
      if (act) {
        if (valid) {
          if (staged) strip_put<VEC>(lrow + fd.out_off + e0, p);
          else strip_store<VEC>(orow + fd.out_off + e0, p);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          S.v[i] += p.v[i];
          Q.v[i] += p.v[i] * p.v[i];
        }
      }
    }
    // dense block of combined_dnn_input (inputs.py:126-138)
    if (P.dense_off >= 0 && valid) {
      for (int j = wv_id * LPR + gl; j < P.n_dense; j += kNW * LPR) {
        if (staged) lrow[P.dense_off + j] = xr[ldg_i32(P.dense_cols + j)];
        else stg_f32(orow + P.dense_off + j, xr[ldg_i32(P.dense_cols + j)]);
      }
    }
  }

  if (wide) {
#pragma unroll
    for (int k = 0; k < WCH; ++k) ws += ((k * kNW + wv_id) * LPR + gl < nwf) ? wval[k] : 0.f;
    for (int f = (WCH * kNW + wv_id) * LPR + gl; f < nwf; f += kNW * LPR) {  // > 8*LPR wide fields
      const dctr_field_t& fd = T.wide[f];
      ws += ldg_f32(fd.table + checked(raw_id(xr, fd.col), fd.vocab, bad) * row_ld(fd));
    }
    for (int f = P.n_wide_fixed + wv_id * LPR + gl; f < P.n_wide; f += kNW * LPR) {  // pooled VarLen
      const dctr_field_t& fd = T.wide[f];
      uint8_t* am = nullptr;
      if (amax && valid && fd.pool == DCTR_POOL_MAX) {
        const int off = ldg_i32(am_wide_off + f);
        if (off >= 0) am = amax + static_cast<int64_t>(b) * ld_am + off;
      }
      ws += pool_field<1>(fd, xr, 0, true, bad, am).v[0];
    }
    if (P.wdense_w)
      for (int j = wv_id * LPR + gl; j < P.n_wdense; j += kNW * LPR)
        ws += xr[ldg_i32(P.wdense_cols + j)] * ldg_f32(P.wdense_w + j);
  }

  // ---- the 4 waves' partials meet in LDS; wave 0 finishes --------------------------------------
  if (fm || wide || fm_s) {
    float* mine = T.red + (wv_id * kWave + lane) * RED;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mine[i] = S.v[i];
      mine[VEC + i] = Q.v[i];
    }
    mine[2 * VEC] = ws;
    __syncthreads();
    if (wv_id == 0) {
      float st[VEC], qt[VEC], wt = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) st[i] = qt[i] = 0.f;
#pragma unroll
      for (int w = 0; w < kNW; ++w) {
        const float* o = T.red + (w * kWave + lane) * RED;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          st[i] += o[i];
          qt[i] += o[VEC + i];
        }
        wt += o[2 * VEC];
      }
      if (fm_s && valid && e0 < P.emb_dim) {
        Strip<VEC> sv;
#pragma unroll
        for (int i = 0; i < VEC; ++i) sv.v[i] = st[i];
        strip_store<VEC>(fm_s + static_cast<int64_t>(b) * lds_ + e0, sv, staged);
      }
      if (fm) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) t += st[i] * st[i] - qt[i];
        t = group_sum<LPR>(t);
        if (gl == 0 && valid) stg_f32(fm + b, 0.5f * t, staged);
      }
      if (wide) {
        wt = group_sum<LPR>(wt);
        if (gl == 0 && valid) stg_f32(P.out_chunks ? orow + (wide - out) : wide + static_cast<int64_t>(b) * ldw, wt, staged);
      }
    }
  }
  if (err && bad) atomicOr(err, 1);
  if (staged) {
    __syncthreads();
    const int q4 = static_cast<int>(ldo >> 2);
    for (int r = wv_id; r < nrows; r += kNW) {
      const f32x4* src = reinterpret_cast<const f32x4*>(lrows + r * ldo);
      float* dst = row_of(b0 + r);
      for (int c = lane; c < q4; c += kWave) stg_wt(dst + 4 * c, src[c]);
    }
  } else if (thru) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
  if (thru) step_signal(P.step_sync, DCTR_SYNC_GATHER);
}

// -------------------------------------------------------------------------------------------------
// backward: scatter-add of the row gradients (+ FM backward folded in)
// -------------------------------------------------------------------------------------------------
template <int VEC>
__device__ __forceinline__ void scatter_strip(float* dst, const Strip<VEC>& g, float scale) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) atomic_add_f32(dst + i, scale * g.v[i]);
}

// gradient of one pooled VarLen field routed back to its rows (sequence.py:49-77 under autograd)
template <int VEC, bool SGD>
__device__ __forceinline__ void unpool_field(const dctr_field_t& fd, const float* xr, int e0, bool act,
                                             const Strip<VEC>& gp, float scale) {
  if (!act) return;
  int bad = 0;  // ids were range-checked (and flagged) by the forward pass
  float* base = SGD ? fd.table : fd.gacc;
  const int64_t bld = SGD ? row_ld(fd) : fd.dim;   // gacc is contiguous
  const bool by_len = fd.len_col >= 0;
  const int64_t len_i = by_len ? raw_id(xr, fd.len_col) : 0;
  if (fd.pool == DCTR_POOL_MAX) {
    // arg-max per element, first maximum wins (torch.max on CPU); ties happen when every position
    // is masked -- then the gradient lands on position 0's row.
    float best[VEC];
    int64_t bid[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      best[i] = -INFINITY;
      bid[i] = 0;
    }
    for (int t = 0; t < fd.len; ++t) {
      const int64_t rid = raw_id(xr, fd.col + t);
      const bool m = by_len ? (static_cast<int64_t>(t) < len_i) : (rid != 0);
      const int64_t id = checked(rid, fd.vocab, bad);
      const Strip<VEC> row = strip_load<VEC>(fd.table + id * row_ld(fd) + e0);
      const float pen = m ? 0.f : 1e9f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float v = row.v[i] - pen;
        if (v > best[i]) {
          best[i] = v;
          bid[i] = id;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) atomic_add_f32(base + bid[i] * bld + e0 + i, scale * gp.v[i]);
    return;
  }
  float cnt = 0.f;
  if (fd.pool == DCTR_POOL_MEAN && !by_len)
    for (int t = 0; t < fd.len; ++t) cnt += (raw_id(xr, fd.col + t) != 0) ? 1.f : 0.f;
  Strip<VEC> gs = gp;
  if (fd.pool == DCTR_POOL_MEAN) {
    const float den = (by_len ? static_cast<float>(len_i) : cnt) + 1e-8f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) gs.v[i] = gp.v[i] / den;  // same rounding as autograd's div backward
  }
  for (int t = 0; t < fd.len; ++t) {
    const int64_t rid = raw_id(xr, fd.col + t);
    const bool m = by_len ? (static_cast<int64_t>(t) < len_i) : (rid != 0);
    if (!m) continue;
    const int64_t id = checked(rid, fd.vocab, bad);
    scatter_strip<VEC>(base + id * bld + e0, gs, scale);
  }
}

template <int VEC, int LPR, bool SGD>
__global__ __launch_bounds__(kThreads) void k_embed_bwd(dctr_plan_t P, const float* __restrict__ X,
                                                        int64_t ldx, int B,
                                                        const float* __restrict__ gout, int64_t ldg,
                                                        const float* __restrict__ out, int64_t ldo,
                                                        const float* __restrict__ gfm,
                                                        const float* __restrict__ gwide, float lr) {
  constexpr int SPB = kWave / LPR;
  constexpr int CH = 8;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wv_id = tid >> 6, lane = tid & 63;
  const int grp = lane / LPR, gl = lane % LPR;
  const int b0 = blockIdx.x * SPB;
  const int nrows = min(SPB, B - b0);
  const Tile T = stage_tile(P, X, ldx, b0, nrows, SPB, smem);
  if (grp >= nrows) return;  // no barrier below: whole groups may leave
  const int b = b0 + grp;
  const float* xr = T.xs + grp * P.n_xcols;
  const int e0 = gl * VEC;
  const float scale = SGD ? -lr : 1.f;
  int bad = 0;  // ids were range-checked (and flagged) by the forward pass

  // ---- wide tables: d wide[b] / d w_f[id] = 1 --------------------------------------------------
  if (gwide) {
    const float gw = ldg_f32(gwide + b);
    for (int f = wv_id * LPR + gl; f < P.n_wide_fixed; f += kNW * LPR) {
      const dctr_field_t& fd = T.wide[f];
      const int64_t id = checked(raw_id(xr, fd.col), fd.vocab, bad);
      atomic_add_f32((SGD ? fd.table + id * row_ld(fd) : fd.gacc + id), scale * gw);
    }
    for (int f = P.n_wide_fixed + wv_id * LPR + gl; f < P.n_wide; f += kNW * LPR) {
      const dctr_field_t& fd = T.wide[f];
      Strip<1> g1;
      g1.v[0] = gw;
      unpool_field<1, SGD>(fd, xr, 0, true, g1, scale);
    }
  }
  if (!gout && !gfm) return;
  if (P.n_deep <= 0) return;

  // ---- FM backward needs S[d] = sum_f e[f][d]; e comes from the saved forward output (every wave
  //      rebuilds S from the L2-resident row rather than synchronising with its siblings) --------
  const float gf = gfm ? ldg_f32(gfm + b) : 0.f;
  const float* orow = out ? out + static_cast<int64_t>(b) * ldo : nullptr;
  const float* grow = gout ? gout + static_cast<int64_t>(b) * ldg : nullptr;
  Strip<VEC> S = strip_zero<VEC>();
  if (gfm) {
    for (int f0 = 0; f0 < P.n_deep; f0 += CH) {
      Strip<VEC> r[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const dctr_field_t& fd = T.deep[min(f0 + k, P.n_deep - 1)];
        r[k] = strip_load<VEC>(orow + fd.out_off + ((e0 < fd.dim) ? e0 : 0));
      }
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const dctr_field_t& fd = T.deep[min(f0 + k, P.n_deep - 1)];
        const bool live = (f0 + k < P.n_deep) && (e0 < fd.dim);
#pragma unroll
        for (int i = 0; i < VEC; ++i) S.v[i] += live ? r[k].v[i] : 0.f;
      }
    }
  }

  for (int f0 = wv_id; f0 < P.n_deep; f0 += kNW * CH) {
    Strip<VEC> g[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const dctr_field_t& fd = T.deep[min(f0 + k * kNW, P.n_deep - 1)];
      const int eo = fd.out_off + ((e0 < fd.dim) ? e0 : 0);
      g[k] = grow ? strip_load<VEC>(grow + eo) : strip_zero<VEC>();
      if (gfm) {
        const Strip<VEC> e = strip_load<VEC>(orow + eo);
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[k].v[i] += gf * (S.v[i] - e.v[i]);
      }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int f = f0 + k * kNW;
      if (f < P.n_deep) {
        const dctr_field_t& fd = T.deep[f];
        const bool act = e0 < fd.dim;
        if (f < P.n_deep_fixed) {
          if (act) {
            const int64_t id = checked(raw_id(xr, fd.col), fd.vocab, bad);
            scatter_strip<VEC>((SGD ? fd.table + id * row_ld(fd) : fd.gacc + id * fd.dim) + e0, g[k], scale);
          }
        } else {
          unpool_field<VEC, SGD>(fd, xr, e0, act, g[k], scale);
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// pass 2: consume gacc rows of this batch, apply the optimizer, leave gacc zero
// -------------------------------------------------------------------------------------------------
template <int OPT>
__device__ __forceinline__ void apply_elem(const dctr_field_t& fd, int64_t row, int e, float lr, float eps) {
  const float G = atomic_xchg_f32(fd.gacc + row * fd.dim + e, 0.f);
  if (G == 0.f) return;  // untouched, already consumed by a duplicate, or a genuinely zero gradient
  float* w = fd.table + row * row_ld(fd) + e;
  if (OPT == DCTR_OPT_ADAGRAD) {
    float* st = fd.state + row * state_ld(fd) + e;
    const float s = ldg_f32(st) + G * G;
    stg_f32(st, s);
    stg_f32(w, ldg_f32(w) - lr * (G / (sqrtf(s) + eps)));
  } else {
    stg_f32(w, ldg_f32(w) - lr * G);
  }
}

template <int VEC, int LPR, int OPT>
__global__ __launch_bounds__(kThreads) void k_embed_apply(dctr_plan_t P, const float* __restrict__ X,
                                                          int64_t ldx, int B, float lr, float eps) {
  constexpr int SPB = kWave / LPR;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, wv_id = tid >> 6, lane = tid & 63;
  const int grp = lane / LPR, gl = lane % LPR;
  const int b0 = blockIdx.x * SPB;
  const int nrows = min(SPB, B - b0);
  const Tile T = stage_tile(P, X, ldx, b0, nrows, SPB, smem);
  if (grp >= nrows) return;
  const float* xr = T.xs + grp * P.n_xcols;
  const int e0 = gl * VEC;
  int bad = 0;
  for (int f = wv_id; f < P.n_deep; f += kNW) {
    const dctr_field_t& fd = T.deep[f];
    if (e0 >= fd.dim) continue;
    for (int t = 0; t < fd.len; ++t) {
      const int64_t id = checked(raw_id(xr, fd.col + t), fd.vocab, bad);
#pragma unroll
      for (int i = 0; i < VEC; ++i) apply_elem<OPT>(fd, id, e0 + i, lr, eps);
    }
  }
  for (int f = wv_id * LPR + gl; f < P.n_wide; f += kNW * LPR) {
    const dctr_field_t& fd = T.wide[f];
    for (int t = 0; t < fd.len; ++t)
      apply_elem<OPT>(fd, checked(raw_id(xr, fd.col + t), fd.vocab, bad), 0, lr, eps);
  }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
int lanes_per_row(const dctr_plan_t* p, int vec) {
  int need = (p->max_dim + vec - 1) / vec;
  int lpr = 1;
  while (lpr < need) lpr <<= 1;
  return lpr;
}

size_t tile_bytes(const dctr_plan_t* p, int lpr, int vec) {
  const size_t spb = kWave / lpr;
  return static_cast<size_t>(p->n_deep + p->n_wide) * sizeof(dctr_field_t) +
         ((spb * p->n_xcols + 3) & ~size_t(3)) * sizeof(float) +
         static_cast<size_t>(kNW) * kWave * (2 * vec + 1) * sizeof(float);
}

int check_plan(const dctr_plan_t* p, const float* X, int64_t ldx, int32_t B) {
  if (!p || !X || B < 0 || p->n_xcols <= 0 || ldx < p->n_xcols) return DCTR_EINVAL;
  if (p->n_deep < 0 || p->n_wide < 0 || p->n_deep_fixed > p->n_deep || p->n_wide_fixed > p->n_wide)
    return DCTR_EINVAL;
  if ((p->n_deep && !p->deep) || (p->n_wide && !p->wide)) return DCTR_EINVAL;
  if (p->vec != 1 && p->vec != 2 && p->vec != 4) return DCTR_EINVAL;
  if (p->max_dim > 64 * p->vec) return DCTR_ENOSUP;
  return DCTR_OK;
}

#define DCTR_DISPATCH_LPR(VEC_, lpr, ...)                                \
  switch (lpr) {                                                         \
    case 1: { constexpr int VEC = VEC_, LPR = 1; __VA_ARGS__; } break;   \
    case 2: { constexpr int VEC = VEC_, LPR = 2; __VA_ARGS__; } break;   \
    case 4: { constexpr int VEC = VEC_, LPR = 4; __VA_ARGS__; } break;   \
    case 8: { constexpr int VEC = VEC_, LPR = 8; __VA_ARGS__; } break;   \
    case 16: { constexpr int VEC = VEC_, LPR = 16; __VA_ARGS__; } break; \
    case 32: { constexpr int VEC = VEC_, LPR = 32; __VA_ARGS__; } break; \
    default: { constexpr int VEC = VEC_, LPR = 64; __VA_ARGS__; } break; \
  }

// Dynamic LDS above the 64 KB default needs the kernel's attribute raised first (gfx950: 160 KB per workgroup).  The
// tile of a plan with very many input columns (hundreds of VarLen positions, thousands of fields) goes up to kMaxTile.
constexpr size_t kMaxTile = 156 * 1024;
#define DCTR_LAUNCH(kernel, grid, block, lds, stream, ...)                                                       \
  do {                                                                                                           \
    auto kfn_ = kernel;                                                                                          \
    if ((lds) > 64 * 1024)                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn_), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                static_cast<int>(lds));                                                          \
    kfn_<<<grid, block, lds, stream>>>(__VA_ARGS__);                                                             \
  } while (0)

#define DCTR_DISPATCH(vec, lpr, ...)                              \
  if ((vec) == 4) { DCTR_DISPATCH_LPR(4, lpr, __VA_ARGS__) }      \
  else if ((vec) == 2) { DCTR_DISPATCH_LPR(2, lpr, __VA_ARGS__) } \
  else { DCTR_DISPATCH_LPR(1, lpr, __VA_ARGS__) }

}  // namespace

extern "C" int dctr_embed_fwd(const dctr_plan_t* plan, const float* X, int64_t ldx, int32_t B,
                              float* out, int64_t ld_out, float* wide, int64_t ld_wide, float* fm,
                              int32_t* err, const int32_t* units, int32_t n_units, int32_t* ids_t,
                              uint16_t* parts_t, float* fm_s, int64_t ld_s, dctr_stream_t stream) {
  if (int rc = check_plan(plan, X, ldx, B)) return rc;
  if (wide && ld_wide < 1) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  if (fm && (plan->emb_dim <= 0 || !out)) return DCTR_EINVAL;  // FM needs the deep rows
  if (fm_s && (plan->emb_dim <= 0 || !out || ld_s < plan->emb_dim)) return DCTR_EINVAL;
  if (ids_t && (!units || n_units <= 0)) return DCTR_EINVAL;
  if (parts_t && !ids_t) return DCTR_EINVAL;
  if (plan->out_chunks && (plan->chunk_rows <= 0 || !out || (wide && (ld_wide != ld_out || wide < out || wide - out >= ld_out))))
    return DCTR_EINVAL;
  const int n_parts = parts_t ? dctr_embed_update_partitions(plan, B) : 0;
  if (parts_t && (n_parts <= 0 || n_parts > 65535)) return DCTR_ENOSUP;
  if (fm_s && plan->vec > 1 &&
      (ld_s % plan->vec != 0 || reinterpret_cast<uintptr_t>(fm_s) % (4 * plan->vec) != 0))
    return DCTR_EALIGN;
  const int vec = plan->vec;
  if (out && vec > 1 && (ld_out % vec != 0 || reinterpret_cast<uintptr_t>(out) % (4 * vec) != 0))
    return DCTR_EALIGN;
  const int lpr = lanes_per_row(plan, vec);
  size_t lds = tile_bytes(plan, lpr, vec);
  if (lds > kMaxTile) return DCTR_ENOSUP;
  const int spb = kWave / lpr;
  // signalled launches assemble their output rows in LDS when they fit (see the kernel)
  int stage_off = 0;
  if (plan->step_sync && out && ld_out % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) {
    const size_t off = (lds + 15) & ~size_t(15), need = off + static_cast<size_t>(spb) * ld_out * sizeof(float);
    if (need <= 64 * 1024) {
      stage_off = static_cast<int>(off);
      lds = need;
    }
  }
  const dim3 grid((B + spb - 1) / spb), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // general update units (dctr_plan_ext_t): the ids / tags side outputs come from dctr_embed_ids (a unit spans several X
  // columns); what the forward contributes is the arg-max side output of max-pooled fields
  const dctr_plan_ext_t* x = plan->ext;
  if (x && ids_t) return DCTR_EINVAL;
  if (x && x->amax && (x->ld_amax <= 0 || !x->am_deep_off || !x->am_wide_off)) return DCTR_EINVAL;
  DCTR_DISPATCH(vec, lpr, DCTR_LAUNCH((k_embed_fwd<VEC, LPR>), grid, block, lds, s, *plan, X, ldx, B, out, ld_out,
                                      wide, ld_wide, fm, err, units, n_units, ids_t, parts_t, n_parts, fm_s, ld_s,
                                      stage_off, x ? x->amax : nullptr, x ? x->ld_amax : 0,
                                      x ? x->am_deep_off : nullptr, x ? x->am_wide_off : nullptr));
  return launch_status();
}

extern "C" int dctr_embed_bwd(const dctr_plan_t* plan, const float* X, int64_t ldx, int32_t B,
                              const float* g_out, int64_t ld_g, const float* out, int64_t ld_out,
                              const float* g_fm, const float* g_wide, int32_t mode, float lr,
                              dctr_stream_t stream) {
  if (int rc = check_plan(plan, X, ldx, B)) return rc;
  if (B == 0) return DCTR_OK;
  if (mode != DCTR_BWD_ACCUM && mode != DCTR_BWD_SGD) return DCTR_EINVAL;
  if (mode == DCTR_BWD_ACCUM && !(plan->flags & DCTR_PLAN_HAS_GACC)) return DCTR_EINVAL;
  if (mode == DCTR_BWD_SGD && (plan->flags & DCTR_PLAN_HAS_MAXPOOL)) return DCTR_ENOSUP;
  if (g_fm && (!out || plan->emb_dim <= 0)) return DCTR_EINVAL;
  const int vec = plan->vec;
  if (vec > 1) {
    if (g_out && (ld_g % vec != 0 || reinterpret_cast<uintptr_t>(g_out) % (4 * vec) != 0))
      return DCTR_EALIGN;
    if (g_fm && (ld_out % vec != 0 || reinterpret_cast<uintptr_t>(out) % (4 * vec) != 0))
      return DCTR_EALIGN;
  }
  const int lpr = lanes_per_row(plan, vec);
  const size_t lds = tile_bytes(plan, lpr, vec);
  if (lds > kMaxTile) return DCTR_ENOSUP;
  const int spb = kWave / lpr;
  const dim3 grid((B + spb - 1) / spb), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == DCTR_BWD_SGD) {
    DCTR_DISPATCH(vec, lpr, DCTR_LAUNCH((k_embed_bwd<VEC, LPR, true>), grid, block, lds, s,
                                        *plan, X, ldx, B, g_out, ld_g, out, ld_out, g_fm, g_wide, lr));
  } else {
    DCTR_DISPATCH(vec, lpr, DCTR_LAUNCH((k_embed_bwd<VEC, LPR, false>), grid, block, lds, s,
                                        *plan, X, ldx, B, g_out, ld_g, out, ld_out, g_fm, g_wide, lr));
  }
  return launch_status();
}

extern "C" int dctr_embed_apply(const dctr_plan_t* plan, const float* X, int64_t ldx, int32_t B,
                                int32_t opt, float lr, float eps, dctr_stream_t stream) {
  if (int rc = check_plan(plan, X, ldx, B)) return rc;
  if (B == 0) return DCTR_OK;
  if (!(plan->flags & DCTR_PLAN_HAS_GACC)) return DCTR_EINVAL;
  if (opt != DCTR_OPT_SGD && opt != DCTR_OPT_ADAGRAD) return DCTR_EINVAL;
  if (opt == DCTR_OPT_ADAGRAD && !(plan->flags & DCTR_PLAN_HAS_STATE)) return DCTR_EINVAL;
  const int vec = plan->vec;
  const int lpr = lanes_per_row(plan, vec);
  const size_t lds = tile_bytes(plan, lpr, vec);
  if (lds > kMaxTile) return DCTR_ENOSUP;
  const int spb = kWave / lpr;
  const dim3 grid((B + spb - 1) / spb), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (opt == DCTR_OPT_ADAGRAD) {
    DCTR_DISPATCH(vec, lpr, DCTR_LAUNCH((k_embed_apply<VEC, LPR, DCTR_OPT_ADAGRAD>), grid, block, lds, s,
                                        *plan, X, ldx, B, lr, eps));
  } else {
    DCTR_DISPATCH(vec, lpr, DCTR_LAUNCH((k_embed_apply<VEC, LPR, DCTR_OPT_SGD>), grid, block, lds, s,
                                        *plan, X, ldx, B, lr, eps));
  }
  return launch_status();
}

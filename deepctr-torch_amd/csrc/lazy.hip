// lazy.hip -- the reference's REGULARISED / ADAM embedding update without its O(vocabulary) memory traffic (gfx950).
//
// With l2_reg_embedding / l2_reg_linear > 0 (the reference's defaults, basemodel.py:100-102,412-428) or with
// torch.optim.Adam (the examples' optimizer, basemodel.py:447-461) EVERY row of every table moves at every step:
// the L2 term contributes the gradient 2*lambda*w to all V rows, Adam's moments keep pushing rows whose data
// gradient is zero.  The reference pays O(vocabulary) for that (442 M parameters at the Criteo shape: 7 ms / step
// even on this GPU).  But a row's trajectory between two batches that touch it depends on nothing but the row
// itself:   g_t = 2*lambda*w_t ;  (w, state)_{t+1} = opt_step(w_t, state_t, g_t, t+1)
// so it can be replayed LAZILY -- the same recurrence, step by step, in fp32 -- the next time the row is needed:
//   stamp[row] = number of optimizer steps already applied to the row;  *step = steps completed so far (t)
//   k_lazy_catchup  (before the gather of a train step, on the batch's ids): the lane group that wins
//                   atomicMax(stamp[row], t) replays the missed steps stamp..t-1 with g = 2*lambda*w -> the gather
//                   reads the reference's w_t;
//   dctr_embed_update(OPT_ACCUM) (csrc/update.hip, deterministic, no atomics): gacc[row] = sum of the batch's data
//                   gradients of the row;
//   k_lazy_apply    (after the backward): the group that wins atomicMax(stamp[row], t+1) applies step t+1 with
//                   g = gacc[row] + 2*lambda*w and zeroes gacc[row];
//   k_lazy_flush    (before anything else reads the tables: predict / evaluate / state_dict): replays every row to t;
//   dctr_lazy_sweep (round 5; once per step BEHIND the catch-up, finished before the step counter moves -- never beside a
//   catch-up, the stamps are not atomic: include/dctr.h): the flush of the (t mod K)-th of K windows of rows -- no row
//                   sleeps longer than K steps (a replay is a sequential chain: a batch's longest sleeper used to set the
//                   launch's time, and rows nobody drew left their whole history to the next flush).
// The arithmetic owed stays what the reference does -- one optimizer step per row per step, ~0.3 ms of VALU work at the
// Criteo shape -- but not its 10.6 GB of row traffic.  Deterministic: whichever duplicate wins a row computes the same
// thing, and each (row, step) is applied exactly once, in order, whoever applies it.  The replayed (zero-data-gradient)
// steps divide and take roots through the hardware reciprocal / reciprocal square root (div_nr / sqrt_nr in lazy_opt.hpp:
// within 1.5 ulp of the IEEE operations; the bound, the cost and the build switch are stated there); the step that carries
// a data gradient uses the IEEE operations.
//
// Optimizer arithmetic = torch.optim's (single-tensor formulas, fp32; the step-dependent scalars in double like
// torch computes them on the host):
//   SGD      w -= lr * g
//   Adagrad  s += g*g ;  w -= lr * g / (sqrt(s) + eps)
//   Adam     m += (g - m) * (1 - b1) ;  v = v*b2 + (1 - b2)*g*g ;
//            w -= (lr / (1 - b1^T)) * m / (sqrt(v) / sqrt(1 - b2^T) + eps)                      T = step number
#include "common.hpp"
#include "lazy_opt.hpp"

using namespace dctr;
using namespace dctr_lazy;

namespace {

constexpr int kT = 256;

// Replay the steps from+1 .. to of an UNTOUCHED row (g = 2*lambda*w) -- the reference's dense update of a row no sample
// of those batches referred to: the lane's strip of the deep row and, on the lane that holds it, the wide weight, in ONE
// walk over the steps (the wide weight used to take a second walk by a quarter of the lanes while the others waited).
// Adam with the host's tables, the WHOLE WAVE in step: every lane walks T = from_min + 1 .. to (from_min = the earliest
// stamp among the wave's rows) and sits out the steps its own row has already seen.  All lanes then read the SAME table
// entry per trip -- one cache line per load instruction.  Walking each row from its own stamp (the loop below) made the 16
// lane groups of a wave read 16 different lines per trip, twice: with rows that slept differently long (any real batch)
// the CU's L1 spent three times the loop's arithmetic on tag lookups (1 670 cycles per trip against 577 when every row had
// slept the same 64 steps).  Must be called by all 64 lanes (rows that need nothing: from = to).
template <int VEC, bool FAST>
__device__ __forceinline__ void replay_in_step(const OptConst& o, float lam2d, float lam2w, bool deep_on, bool wide_on,
                                               int from, int to, float (&w)[VEC], float (&a)[VEC], float (&b)[VEC],
                                               float& ww, float& wa, float& wb) {
  int from_min = from;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int other = __shfl_xor(from_min, off, kWave);
    from_min = other < from_min ? other : from_min;
  }
  from_min = __builtin_amdgcn_readfirstlane(from_min);
  if (from_min >= to) return;
  float ss, bc;
  adam_tab_uniform(o, from_min + 1, ss, bc);
  // (the divisor's reciprocal from the host's third table when there is one: a v_rcp_f32 per trip is 16 of the loop's
  // ~250 issue cycles)
  const bool tab_r = o.adam_rbc != nullptr;
  float rb = tab_r ? o.adam_rbc[(from_min + 1 < o.n_bc ? from_min + 1 : o.n_bc) - 1] : 0.f;
  for (int T = from_min + 1; T <= to; ++T) {
    float ssn, bcn;
    adam_tab_uniform(o, T + 1, ssn, bcn);       // (the next step's pair is in flight while this step computes; clamped: valid)
    const float rbn = tab_r ? o.adam_rbc[(T + 1 < o.n_bc ? T + 1 : o.n_bc) - 1] : 0.f;
    if (T > from) {
      const float rbc = tab_r ? rb : __builtin_amdgcn_rcpf(bc);
      if (deep_on) {
        if (FAST && !DCTR_LAZY_REPLAY_NR && VEC % 2 == 0) {
          // (this loop is Adam with the host's tables by construction: packed pairs, lazy_opt.hpp adam_replay_step)
#pragma unroll
          for (int i = 0; i < VEC; i += 2) {
            f32x2 w2 = {w[i], w[i + 1]}, a2 = {a[i], a[i + 1]}, b2 = {b[i], b[i + 1]};
            adam_replay_step<f32x2>(1.f - o.beta1, 1.f - o.beta2, o.beta2, o.eps, w2 * lam2d, ss, rbc, w2, a2, b2);
            w[i] = w2.x; w[i + 1] = w2.y; a[i] = a2.x; a[i + 1] = a2.y; b[i] = b2.x; b[i + 1] = b2.y;
          }
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) opt_step<FAST>(o, lam2d * w[i], w[i], a[i], b[i], ss, bc, rbc);
        }
      }
      if (wide_on) opt_step<FAST>(o, lam2w * ww, ww, wa, wb, ss, bc, rbc);
    }
    ss = ssn;
    bc = bcn;
    rb = rbn;
  }
}

template <int VEC, bool FAST>
__device__ __forceinline__ void replay(const OptConst& o, float lam2d, float lam2w, bool deep_on, bool wide_on, int from,
                                       int to, float (&w)[VEC], float (&a)[VEC], float (&b)[VEC], float& ww, float& wa,
                                       float& wb) {
  if (from >= to) return;
  // zero gradient: SGD / Adagrad do not move (RMSprop's square_avg still decays, Adam's moments too)
  const bool still = o.kind == DCTR_LAZY_SGD || o.kind == DCTR_LAZY_ADAGRAD;
  const bool do_d = deep_on && !(still && lam2d == 0.f), do_w = wide_on && !(still && lam2w == 0.f);
  if (!do_d && !do_w) return;
  AdamClock ck;
  if (o.kind == DCTR_LAZY_ADAM) ck.start(o, from + 1);
  for (int T = from + 1; T <= to; ++T) {
    float ss = 0.f, bc = 1.f;
    if (o.kind == DCTR_LAZY_ADAM) {
      ss = ck.step_size();
      bc = ck.bc2_sqrt();
      ck.next();
    }
    const float rbc = __builtin_amdgcn_rcpf(bc);
    if (do_d) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) opt_step<FAST>(o, lam2d * w[i], w[i], a[i], b[i], ss, bc, rbc);
    }
    if (do_w) opt_step<FAST>(o, lam2w * ww, ww, wa, wb, ss, bc, rbc);
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  if (!p) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = 0.f;
    return;
  }
  const Strip<VEC> s = strip_load<VEC>(p);
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = s.v[i];
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if (!p) return;
  Strip<VEC> s;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s.v[i] = v[i];
  strip_store<VEC>(p, s);
}

// The passes share one body: MODE 0 catch-up (batch ids), 1 apply (batch ids), 2 flush (all rows), 3 sweep (the
// (t mod K)-th of K windows of every table's rows: see dctr_lazy_sweep).
// A lane group of `lpr` lanes (a power of two <= 64) owns one (unit, entry); lane gl handles the deep strip
// [gl*VEC, gl*VEC + VEC) and, when gl == 0, the wide element.
template <int VEC, int MODE>
__device__ __forceinline__ void lazy_block(const dctr_lazy_unit_t* __restrict__ units, int n_units,
                                           const int32_t* __restrict__ ids_t, int64_t n_entries, int lpr_shift,
                                           const int32_t* __restrict__ step_ptr, const OptConst& o,
                                           const int32_t* __restrict__ order, int sweep_k, int part, int64_t block_x,
                                           int u, int tid) {
  const int lpr = 1 << lpr_shift;
  const int64_t grp = (block_x * kT + tid) >> lpr_shift;
  const int gl = tid & (lpr - 1);
  if (u >= n_units) return;
  // (lane groups past the end stay in the wave: the replay walks the steps with all 64 lanes, replay_in_step)
  bool dead = grp >= n_entries;
  const dctr_lazy_unit_t un = units[u];
  const int t = *(const DCTR_GLOBAL int32_t*)step_ptr;
  int64_t row = 0;
  if (MODE == 2) {
    row = grp;
    if (row >= un.vocab) {
      dead = true;
      row = 0;
    }
  } else if (MODE == 3) {
    const int64_t wlen = (un.vocab + sweep_k - 1) / sweep_k;      // this table's window
    row = static_cast<int64_t>(t % sweep_k) * wlen + grp;
    if (grp >= wlen || row >= un.vocab) {
      dead = true;
      row = 0;
    }
  } else if (!dead) {
    // (catch-up: entries dealt by gap, k_lazy_order -- a wave's rows then sleep about equally long)
    const int64_t ent = (MODE == 0 && order) ? ldg_i32(order + static_cast<int64_t>(u) * n_entries + grp) : grp;
    const int32_t id = ldg_i32(ids_t + static_cast<int64_t>(u) * n_entries + ent);
    row = (static_cast<uint64_t>(static_cast<int64_t>(id)) >= static_cast<uint64_t>(un.vocab)) ? 0 : id;
  }
  // The row's strips are loaded BEFORE the claim is known (a row has one claimant per launch except for duplicate ids,
  // whose losers simply drop what they loaded): the claim's atomic round trip and the row's HBM round trip overlap
  // instead of following each other.  A winner's early loads are valid: nobody else writes its row in this launch.
  const int e0 = gl * VEC;
  // part (sweep / flush, whose rows have no second claimant): 0 = the whole row; 1 = the wide weight only, one LANE per row
  // and the stamp left alone; 2 = the deep row only, then the stamp.  Pass 1 in front of pass 2 replays the wide weights of
  // 64 rows per wavefront trip instead of 16 (a quarter of the lanes carried them while the others waited: 27 of a trip's
  // 96 instructions).
  const bool deep_on = !dead && un.deep != nullptr && e0 < un.dim && part != 1;
  const bool wide_on = !dead && un.wide != nullptr && gl == 0 && part != 2;
  if (part == 1 && un.wide == nullptr) return;
  const float lam2d = 2.f * un.l2_deep, lam2w = 2.f * un.l2_wide;
  // the table and its first state slab may be strided views of one interleaved slab (row strides in the unit);
  // the second state slab and the gradient slab are contiguous
  const int64_t off = deep_on ? row * un.dim + e0 : 0;
  const int64_t off_w = deep_on ? row * (un.ld_deep > 0 ? un.ld_deep : un.dim) + e0 : 0;
  const int64_t off_a = deep_on ? row * (un.ld_deep_s1 > 0 ? un.ld_deep_s1 : un.dim) + e0 : 0;
  const int64_t row_w = row * (un.ld_wide > 0 ? un.ld_wide : 1), row_a = row * (un.ld_wide_s1 > 0 ? un.ld_wide_s1 : 1);
  float w[VEC], a[VEC], b[VEC], g[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) w[i] = a[i] = b[i] = g[i] = 0.f;
  if (deep_on) {
    load_vec<VEC>(un.deep + off_w, w);
    if (un.deep_s1) load_vec<VEC>(un.deep_s1 + off_a, a);
    if (un.deep_s2) load_vec<VEC>(un.deep_s2 + off, b);
    if (MODE == 1) load_vec<VEC>(un.deep_g + off, g);
  }
  float ww[1] = {0.f}, wa[1] = {0.f}, wb[1] = {0.f}, wg = 0.f;
  if (wide_on) {
    ww[0] = ldg_f32(un.wide + row_w);
    if (un.wide_s1) wa[0] = ldg_f32(un.wide_s1 + row_a);
    if (un.wide_s2) wb[0] = ldg_f32(un.wide_s2 + row);
    if (MODE == 1) wg = ldg_f32(un.wide_g + row);
  }
  // claim the row: the winner is the only group that touches it in this launch
  const int target = (MODE == 1) ? t + 1 : t;
  int prev = target;
  if (gl == 0 && !dead) {
    if (MODE >= 2) {
      prev = un.stamp[row];
      if (prev < target && part != 1) un.stamp[row] = target;
    } else {
      prev = atomicMax(un.stamp + row, target);
    }
  }
  prev = __shfl(prev, (tid & 63) & ~(lpr - 1), kWave);
  const bool live = prev < target;
  if (MODE != 1 && o.kind == DCTR_LAZY_ADAM && o.adam_ss) {
    replay_in_step<VEC, kFastReplay>(o, lam2d, lam2w, deep_on && live, wide_on && live, live ? prev : t, t, w, a, b, ww[0],
                                     wa[0], wb[0]);
    if (!live) return;
  } else {
    if (!live) return;
    replay<VEC, kFastReplay>(o, lam2d, lam2w, deep_on, wide_on, prev, t, w, a, b, ww[0], wa[0], wb[0]);   // (apply after a catch-up: prev == t)
  }
  float ss1 = 0.f, bc1 = 1.f;        // Adam scalars of step t + 1 (apply)
  if (MODE == 1 && o.kind == DCTR_LAZY_ADAM) adam_scalars(o, t + 1, ss1, bc1);
  if (deep_on) {
    if (MODE == 1) {
      float z[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) z[i] = 0.f;
      store_vec<VEC>(un.deep_g + off, z);             // zero at rest
#pragma unroll
      for (int i = 0; i < VEC; ++i) opt_step(o, g[i] + lam2d * w[i], w[i], a[i], b[i], ss1, bc1);
    }
    store_vec<VEC>(un.deep + off_w, w);
    if (un.deep_s1) store_vec<VEC>(un.deep_s1 + off_a, a);
    if (un.deep_s2) store_vec<VEC>(un.deep_s2 + off, b);
  }
  if (wide_on) {
    if (MODE == 1) {
      stg_f32(un.wide_g + row, 0.f);
      opt_step(o, wg + lam2w * ww[0], ww[0], wa[0], wb[0], ss1, bc1);
    }
    stg_f32(un.wide + row_w, ww[0]);
    if (un.wide_s1) stg_f32(un.wide_s1 + row_a, wa[0]);
    if (un.wide_s2) stg_f32(un.wide_s2 + row, wb[0]);
  }
}

template <int VEC, int MODE>
__global__ __launch_bounds__(kT) void k_lazy(const dctr_lazy_unit_t* __restrict__ units, int n_units,
                                             const int32_t* __restrict__ ids_t, int64_t n_entries, int lpr_shift,
                                             const int32_t* __restrict__ step_ptr, OptConst o,
                                             const int32_t* __restrict__ order, int sweep_k, int part) {
  if (MODE != 3) step_priority();
  lazy_block<VEC, MODE>(units, n_units, ids_t, n_entries, lpr_shift, step_ptr, o, order, sweep_k, part, blockIdx.x,
                        blockIdx.y, threadIdx.x);
}

// ---- catch-up: the batch's entries ordered by how long their rows slept -----------------------------------------------
// k_lazy gives every lane group one row and walks its missed steps: a wave runs until the LONGEST gap among its 16 rows
// is done.  The gaps of a big table's rows are geometric (mean V / B = 244 steps at the Criteo shape) and the expected
// maximum of 16 of them is 3.4 x their mean -- 70 % of the lanes' cycles went to waiting (measured: 0.29 ms per step while
// every row came back after exactly 64 steps, the bench's 64-batch cycle; 1.45 ms on fresh batches).  One workgroup per
// unit deals the entries into 256 buckets of gap (width = mean gap / 32: eight means end to end, everything longer in
// the last one), longest first: order[u][k] = the k-th entry to process.  Rows of one wave then differ by a bucket's
// width.  Counting sort through LDS; the order inside a bucket depends on the atomics' arrival but no result does (every
// row is replayed by itself).
constexpr int kOrderT = 1024;
__global__ __launch_bounds__(kOrderT) void k_lazy_order(const dctr_lazy_unit_t* __restrict__ units,
                                                        const int32_t* __restrict__ ids_t, int n_entries,
                                                        const int32_t* __restrict__ step_ptr, int32_t* __restrict__ order) {
  step_priority();
  __shared__ int hist[256];
  __shared__ unsigned long long total;
  __shared__ int red[kOrderT / 64];
  const int u = blockIdx.x;
  const dctr_lazy_unit_t un = units[u];
  const int t = *(const DCTR_GLOBAL int32_t*)step_ptr;
  const int32_t* ids = ids_t + static_cast<int64_t>(u) * n_entries;
  int32_t* out = order + static_cast<int64_t>(u) * n_entries;
  auto gap_of = [&](int i) {
    const int32_t id = ldg_i32(ids + i);
    const int64_t row = (static_cast<uint64_t>(static_cast<int64_t>(id)) >= static_cast<uint64_t>(un.vocab)) ? 0 : id;
    const int g = t - ldg_i32(un.stamp + row);
    return g > 0 ? g : 0;
  };
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) total = 0ull;
  __syncthreads();
  // mean gap -> bucket width
  unsigned long long mine = 0ull;
  for (int i = threadIdx.x; i < n_entries; i += kOrderT) mine += static_cast<unsigned long long>(gap_of(i));
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, kWave);
  if ((threadIdx.x & 63) == 0) atomicAdd(&total, mine);
  __syncthreads();
  const int mean = static_cast<int>(total / static_cast<unsigned long long>(n_entries));
  const int width = mean / 32 > 0 ? mean / 32 : 1;
  auto bucket_of = [&](int g) {          // bucket 0 = the longest gaps
    const int b = g / width;
    return 255 - (b < 255 ? b : 255);
  };
  for (int i = threadIdx.x; i < n_entries; i += kOrderT) atomicAdd(&hist[bucket_of(gap_of(i))], 1);
  __syncthreads();
  // exclusive scan of the 256 counts (wave 0..3 each scan 64, then add the waves' totals)
  if (threadIdx.x < 256) {
    const int c = hist[threadIdx.x];
    int x = c;
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off, kWave);
      if ((threadIdx.x & 63) >= off) x += y;
    }
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = x;
    hist[threadIdx.x] = x - c;           // exclusive within the wave
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    int base = 0;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) base += red[w];
    hist[threadIdx.x] += base;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_entries; i += kOrderT) out[atomicAdd(&hist[bucket_of(gap_of(i))], 1)] = i;
}

__global__ void k_lazy_inc(int32_t* step) { *step += 1; }

// the same optimizer step over a flat DENSE slab (tower weights, Linear.weight, the prediction bias), with an optional
// per-element lambda: g = grad + 2*lambda*p.  T = *step + 1.
__global__ __launch_bounds__(kT) void k_dense_opt_reg(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ s1, float* __restrict__ s2,
                                                      const float* __restrict__ lam, int64_t n,
                                                      const int32_t* __restrict__ step_ptr, OptConst o) {
  step_priority();
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kT + threadIdx.x;
  if (i >= n) return;
  float ss = 0.f, bc = 1.f;
  if (o.kind == DCTR_LAZY_ADAM) adam_scalars(o, *(const DCTR_GLOBAL int32_t*)step_ptr + 1, ss, bc);
  float w = ldg_f32(p + i);
  float a = s1 ? ldg_f32(s1 + i) : 0.f, b = s2 ? ldg_f32(s2 + i) : 0.f;
  const float gt = ldg_f32(g + i) + (lam ? 2.f * ldg_f32(lam + i) * w : 0.f);
  opt_step(o, gt, w, a, b, ss, bc);
  stg_f32(p + i, w);
  if (s1) stg_f32(s1 + i, a);
  if (s2) stg_f32(s2 + i, b);
}

int check(const dctr_lazy_unit_t* units, int n_units, const int32_t* step, const dctr_lazy_opt_t* opt, int vec,
          int max_dim) {
  if (!units || n_units <= 0 || !step || !opt) return DCTR_EINVAL;
  if (opt->kind != DCTR_LAZY_SGD && opt->kind != DCTR_LAZY_ADAGRAD && opt->kind != DCTR_LAZY_ADAM &&
      opt->kind != DCTR_LAZY_RMSPROP)
    return DCTR_EINVAL;
  if (vec != 1 && vec != 4) return DCTR_EINVAL;
  if (max_dim < 1 || max_dim > 64 * vec) return DCTR_ENOSUP;
  return DCTR_OK;
}

template <int MODE>
int launch(const dctr_lazy_unit_t* units, int n_units, const int32_t* ids_t, int64_t n_entries, const int32_t* step,
           const dctr_lazy_opt_t* opt, int vec, int max_dim, hipStream_t s, int32_t* order = nullptr, int sweep_k = 0) {
  const int rc = check(units, n_units, step, opt, vec, max_dim);
  if (rc != DCTR_OK) return rc;
  if (n_entries <= 0) return DCTR_OK;
  int lpr = 1, shift = 0;
  while (lpr * vec < max_dim) {
    lpr <<= 1;
    ++shift;
  }
  const OptConst o = opt_const(opt);
  const int64_t threads = n_entries << shift;
  static const bool ordered = !(getenv("DCTR_LAZY_ORDER") && getenv("DCTR_LAZY_ORDER")[0] == '0');   // (A/B switch)
  // zero-gradient steps of SGD / Adagrad without an L2 term move nothing: no replay to balance
  const bool replays = !((o.kind == DCTR_LAZY_SGD || o.kind == DCTR_LAZY_ADAGRAD) && !opt->any_l2);
  if (MODE == 0 && order && ordered && replays && n_entries >= 64 && n_entries <= (1 << 18)) {
    k_lazy_order<<<dim3(static_cast<unsigned>(n_units)), dim3(kOrderT), 0, s>>>(units, ids_t, static_cast<int>(n_entries),
                                                                                step, order);
  } else {
    order = nullptr;
  }
  static const bool split = !(getenv("DCTR_LAZY_SPLIT_WIDE") && getenv("DCTR_LAZY_SPLIT_WIDE")[0] == '0');   // (A/B switch)
  // (round 6 measured the sweep as a PERSISTENT launch -- one 1024-lane workgroup per CU, <= 80 VGPRs, so that a tower
  // workgroup (192 VGPRs on all four SIMDs of a CU) always finds room beside it: slower, 0.52 ms against 0.44.  At four
  // waves per SIMD the replay chain is latency-bound (the plain grid holds five to six), and what the chain's launches gain
  // is less than that.  The step is bound by the SUM of its arithmetic, not by who gets the registers.)
  auto run = [&](int64_t n_threads, int sh, int part) {
    const int64_t blocks = (n_threads + kT - 1) / kT;
    const dim3 g(static_cast<unsigned>(blocks), static_cast<unsigned>(n_units));
    if (vec == 4)
      k_lazy<4, MODE><<<g, dim3(kT), 0, s>>>(units, n_units, ids_t, n_entries, sh, step, o, order, sweep_k, part);
    else
      k_lazy<1, MODE><<<g, dim3(kT), 0, s>>>(units, n_units, ids_t, n_entries, sh, step, o, order, sweep_k, part);
  };
  int part = 0;
  if (MODE >= 2 && split && shift > 0) {
    run(n_entries, 0, 1);     // the wide weights first, one lane per row (the stamps are written by the second pass)
    part = 2;
  }
  run(threads, shift, part);
  return launch_status();
}

}  // namespace

extern "C" size_t dctr_sizeof_lazy_unit(void) { return sizeof(dctr_lazy_unit_t); }
extern "C" size_t dctr_sizeof_lazy_opt(void) { return sizeof(dctr_lazy_opt_t); }

extern "C" int dctr_lazy_catchup(const dctr_lazy_unit_t* units, int32_t n_units, const int32_t* ids_t, int32_t B,
                                 const int32_t* step, const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim,
                                 int32_t* order_ws, dctr_stream_t stream) {
  if (B < 0 || (B > 0 && !ids_t)) return DCTR_EINVAL;
  return launch<0>(units, n_units, ids_t, B, step, opt, vec, max_dim, static_cast<hipStream_t>(stream), order_ws);
}

extern "C" int dctr_lazy_apply(const dctr_lazy_unit_t* units, int32_t n_units, const int32_t* ids_t, int32_t B,
                               const int32_t* step, const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim,
                               dctr_stream_t stream) {
  if (B < 0 || (B > 0 && !ids_t)) return DCTR_EINVAL;
  return launch<1>(units, n_units, ids_t, B, step, opt, vec, max_dim, static_cast<hipStream_t>(stream));
}

extern "C" int dctr_lazy_flush(const dctr_lazy_unit_t* units, int32_t n_units, int64_t max_vocab, const int32_t* step,
                               const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim, dctr_stream_t stream) {
  if (max_vocab < 0) return DCTR_EINVAL;
  return launch<2>(units, n_units, nullptr, max_vocab, step, opt, vec, max_dim, static_cast<hipStream_t>(stream));
}

extern "C" int dctr_lazy_sweep(const dctr_lazy_unit_t* units, int32_t n_units, int64_t max_vocab, int32_t K,
                               const int32_t* step, const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim,
                               dctr_stream_t stream) {
  if (max_vocab < 0 || K <= 0) return DCTR_EINVAL;
  const int64_t wlen = (max_vocab + K - 1) / K;
  return launch<3>(units, n_units, nullptr, wlen, step, opt, vec, max_dim, static_cast<hipStream_t>(stream), nullptr, K);
}

extern "C" int dctr_dense_opt_reg(float* p, const float* g, float* s1, float* s2, const float* lam, int64_t n,
                                  const dctr_lazy_opt_t* opt, const int32_t* step, dctr_stream_t stream) {
  if (!p || !g || n < 0 || !opt || !step) return DCTR_EINVAL;
  if ((opt->kind == DCTR_LAZY_ADAGRAD || opt->kind == DCTR_LAZY_RMSPROP) && !s1) return DCTR_EINVAL;
  if (opt->kind == DCTR_LAZY_ADAM && (!s1 || !s2)) return DCTR_EINVAL;
  if (opt->kind != DCTR_LAZY_SGD && opt->kind != DCTR_LAZY_ADAGRAD && opt->kind != DCTR_LAZY_ADAM &&
      opt->kind != DCTR_LAZY_RMSPROP)
    return DCTR_EINVAL;
  if (n == 0) return DCTR_OK;
  const OptConst o = opt_const(opt);
  k_dense_opt_reg<<<dim3(static_cast<unsigned>((n + kT - 1) / kT)), dim3(kT), 0, static_cast<hipStream_t>(stream)>>>(
      p, g, s1, s2, lam, n, step, o);
  return launch_status();
}

extern "C" int dctr_lazy_step_inc(int32_t* step, dctr_stream_t stream) {
  if (!step) return DCTR_EINVAL;
  k_lazy_inc<<<dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream)>>>(step);
  return launch_status();
}

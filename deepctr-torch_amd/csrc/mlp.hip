// mlp.hip -- the DNN tower (layers/core.py:120-134: Linear -> ReLU per hidden layer, no BN, dropout inactive)
// plus the [N_last -> 1] projection `dnn_linear` (deepfm.py:61,84) on the gfx950 matrix cores, fp32 in / fp32 out.
//
// The reference issues, per train step, 3 GEMMs forward and 6 backward through ATen (hipBLASLt picks a
// different macro-tile per shape: 9 GEMM launches + bias / relu / threshold / column-sum launches, ~25 kernels,
// ~250 us at batch 4096 where the math is 3.5 GFLOP = 22 us at the fp32 MFMA peak).  Here the whole tower is
//     k_mlp_fwd       1 launch: every layer + the projection, activations handed on through LDS
//     k_mlp_bwd_data  1 launch: d loss / d pre-activation of every layer and d loss / d input
//     k_mlp_wgrad     1 launch: every dW, dbias, dw_out as split-batch partials (no atomics)
//     k_mlp_reduce    1 launch: fixed-order sum of the partials -> the gradient tensors
// v_mfma_f32_16x16x4_f32 / 32x32x2_f32 are exact fp32 (an fmaf chain), so the 1e-5 logit bar holds.
//
// Forward / backward-data mapping: a workgroup (8 waves) owns 16 samples -- the MFMA row dimension -- for the
// whole tower, so a layer's output never leaves the CU: it is written to LDS in the accumulator layout and read
// back as the next layer's A operand.  The weights are the B operand and are used by exactly one wave of the
// workgroup, so they are not staged: each lane fetches them from L2 straight into registers as dwordx4,
//   forward   W[n][k..k+3]   (4 consecutive k of one output column; the k -> (MFMA step, lane group) assignment
//                             is a permutation of the reduction index, applied to A and B alike)
//   backward  W[n][c..c+3]   (4 consecutive OUTPUT columns: the lane's four accumulators are four interleaved
//                             16-column tiles, written back as one dwordx4)
// Both read the native nn.Linear layout [N, K]; no transposed copy exists.  B=4096 -> 256 workgroups, one per CU.
//
// Weight-gradient mapping: dW_l = dH_l^T . In_l is a [N_l, K_l] GEMM reduced over the batch.  A wave owns a
// 64x64 output tile (2x2 MFMA 32x32x2) over a slice of the batch; the four waves of a workgroup take four batch
// slices of the same tile and are summed through LDS in wave order; S workgroups per tile write S partial slabs
// that k_mlp_reduce adds in slab order: bit-reproducible, which keeps data-parallel replicas identical.
#include "common.hpp"

using namespace dctr;

namespace {

constexpr int kTM = 16;        // samples per workgroup (forward / backward-data)
constexpr int kT = 512;        // threads per workgroup (forward / backward-data)
constexpr int kWaves = kT / 64;
constexpr int kKC = 512;       // columns of the tower input staged in LDS at a time
constexpr int kNTMax = 4;      // output tiles a wave carries at once (forward)
constexpr int kTW = 256;       // threads per workgroup (wgrad)
constexpr int kMaxL = DCTR_MLP_MAX_LAYERS;
// Padding floats behind every LDS tile row (row strides are a multiple of 16 plus this).  The A operand of
// v_mfma_f32_16x16x4 is read as one ds_read_b128 per lane -- lane (g = lane / 16, c = lane % 16) takes the 16 bytes at
// row c, column 4 g of the K block -- and that instruction is served in four FIXED 16-lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md, LDS) over 16 slots of 16 bytes.  With a row pitch of
// s slots the lane's slot is (c s + g) mod 16: for odd s (the +4 padding of rounds 1-3: s = 13 and 1) every group has two
// lanes on one slot -- SQ_LDS_BANK_CONFLICT was 40 % of the tower's LDS cycles; for s = 2 mod 4 the eight rows of a group
// that share g land on eight distinct even (g = 0, 2) or odd (g = 1, 3) slots: conflict-free.  s = 2 mod 4 <=> pitch = 8 mod 16.
constexpr int kPad = 8;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row32(int r, int p) { return (r & 3) + 8 * (r >> 2) + 4 * p; }

__host__ __device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct LayerDev {
  const float* W;
  const float* bias;
  float* h;
  float* dh;
  int K, N, ldw, ldh, relu;
};

struct MlpArgs {
  LayerDev L[kMaxL];
  int n_layers;
  int B;
  const float* x;
  int64_t ldx;
  const float* w_out;
  float* logit;      // forward: [B] (with w_out)
  const float* g;    // backward: [B] (with w_out) or [B, ldg]
  int64_t ldg;
  float* gx;         // backward: [B, ldgx] nullable
  int64_t ldgx;
  int rsx, rsh;      // LDS row strides (floats) of the forward
  int kc;            // columns of the tower input staged in LDS at a time (<= kKC; smaller for wide towers)
  int rsd;           // LDS row stride of the backward-data pass
  int fast;          // 1: the tower fits the fast bodies (mlp_fwd_fast / mlp_bwd_fast)
  int32_t* sync;     // nullable: k_mlp_train signals DCTR_SYNC_TOWER when gx and g_logit have left the chip (dctr.h)
  int wt;            // != 0: what the weight-gradient launch reads (h, dh, the gathered rows, g_logit, the head's partials) is
                     // stored write-through (fast bodies only): that launch waits for DCTR_SYNC_T_GEN, not for this one's end
  uint32_t wmask;    // diagnostics: AND mask on the weight byte offsets (0xffffffff normally; DCTR_MLP_WMASK in the diag build
                     // folds the weight stream onto a few KB that stay in L1 -- timing experiment, wrong results)
  unsigned long long* trace;
};

// head + loss of the fused train kernel (csrc/head.hip has the stand-alone version)
struct HeadArgs {
  const float* part0;   // [B] logit parts added before the tower's (nullable): linear, FM / CIN
  const float* part1;
  const float* bias;    // [1] nullable
  const float* y;       // [B]
  float* y_pred;        // [B]
  float* g_logit;       // [B]
  float* part_loss;     // [n workgroups] per-workgroup partial sums
  float* part_gbias;    // [n workgroups]
};

__device__ __forceinline__ f32x4 ldg_f4(const float* p) { return *(const DCTR_GLOBAL f32x4*)p; }

// The embedding lookup as the fused train kernel's input stage (dctr_embed_tower_train_step; round 4): a workgroup gathers
// the rows of its own 16 samples straight from the tables while it would otherwise wait for the gather kernel's output --
// what k_embed_fwd (csrc/embed.hip) computes for a plan of fixed-length fields with one embedding_dim, in its order of
// additions: the [16, K0] tower input tile in LDS (and out to `out`, for the weight gradients), the linear logit and the FM
// term (kept in LDS for the head), sum_f e (`fm_s`, for FM's backward in the update kernel).
struct GatherArgs {
  const dctr_field_t* deep;    // [n_deep] fixed-length fields, dim == D
  const dctr_field_t* wide;    // [n_wide] fixed-length fields, dim == 1
  const int32_t* dense_cols;   // [n_dense]
  const int32_t* wdense_cols;  // [n_wdense]
  const float* wdense_w;       // [n_wdense] nullable
  const float* X;              // [B, ldx]
  int64_t ldx;
  float* out;                  // [B, ldo] the gathered tower input (combined_dnn_input layout)
  int64_t ldo;
  float* fm_s;                 // [B, lds] nullable
  int64_t lds;
  int32_t* err;                // nullable
  int n_deep, n_wide, n_dense, n_wdense, nc, dense_off, D, lpr_shift, want_fm, scratch_off;
  // pooled VarLen fields (sum / mean; round 5): deep fields [n_deep_fixed, n_deep) and wide fields [n_wide_fixed, n_wide) are
  // pooled over their positions; gsd / gsw list those positions in field order, entry = (field << 16) | position
  int n_deep_fixed, n_wide_fixed, n_gsd, n_gsw;
  const int32_t* gsd;
  const int32_t* gsw;
  // max pooling: the position of the first maximum per element goes to amax [B, ld_am] (what dctr_embed_fwd writes: the
  // update routes the gradient there); byte offsets per field from the plan's ext block (-1: not max-pooled)
  uint8_t* amax;
  int64_t ld_am;
  const int32_t* am_deep_off;
  const int32_t* am_wide_off;
  // round 6 (dctr_embed_tower_train_step_sync): the launch may START before the previous step's weight-gradient launch has
  // stepped the dense parameters -- it gathers its rows first and waits here, in the kernel, until sync[DCTR_SYNC_W_GEN] has
  // reached sync[DCTR_SYNC_T_GEN] (towers finished so far = weight steps that must have happened); NULL: no wait
  int32_t* wsync;
  unsigned long long wtimeout;   // s_memrealtime ticks (100 MHz)
};


// diagnostics (tools/mlp_trace.py): 16 wall_clock64 stamps per workgroup, or NULL -- only in the DCTR_DIAG build
// (libdctr_hip_diag.so); the shipped library keeps no mutable global state
#ifdef DCTR_DIAG
unsigned long long* g_mlp_trace = nullptr;
#define MLP_TRACE(T, slot)                                                                   \
  do {                                                                                       \
    if ((T) && threadIdx.x == 0) (T)[blockIdx.x * 16ull + (slot)] = wall_clock64();          \
  } while (0)
// The fast bodies stamp into LDS (every lane of the wave stores the same 32-bit s_memrealtime value to the slot: no
// branch, no register pressure; the last wave through a point wins, i.e. the slot holds the slowest wave's time) and the
// kernel writes the slots out at its very end, so a stamp does not split a software-pipelined region.
// History (round 3): the diag k_mlp_train used to run 55-65 % slower than the shipped one and its timeline described
// another kernel.  The cause was one line, `A.trace += ...` before the generic backward: WRITING to the by-value
// argument struct makes the compiler keep the whole struct in scratch (784 B, every A.x a scratch load, 223 VGPRs).
// The trace pointer is now passed beside A; both builds have identical register counts and run at the same speed.
#define FT_DECL                                                                              \
  __shared__ uint32_t ft_sh_[32];                                                            \
  uint32_t* ft_ = ft_sh_;                                                                    \
  if (threadIdx.x < 32) ft_sh_[threadIdx.x] = 0;                                             \
  __syncthreads()
#define FT(slot) (ft_[slot] = static_cast<uint32_t>(wall_clock64()))
#define FT_ARG , uint32_t* ft_
#define FT_PASS , ft_
#define FT_DECL_B uint32_t* fb_ = ft_sh_ + 16
#define FT_PASS_B , fb_
#define FT_FLUSH_B(T)                                                                                       \
  do {                                                                                                      \
    if ((T) && threadIdx.x == 0)                                                                            \
      for (int i_ = 0; i_ < 16; ++i_) (T)[(4096ull + blockIdx.x) * 16ull + i_] = fb_[i_];                    \
  } while (0)
#define FT_FLUSH(T)                                                                                         \
  do {                                                                                                      \
    __syncthreads();                                                                                        \
    if ((T) && threadIdx.x == 0)                                                                            \
      for (int i_ = 0; i_ < 16; ++i_) (T)[blockIdx.x * 16ull + i_] = ft_[i_];                               \
  } while (0)
#else
static unsigned long long* const g_mlp_trace = nullptr;
#define MLP_TRACE(T, slot) do { } while (0)
#define FT_DECL do { } while (0)
#define FT(slot) do { } while (0)
#define FT_ARG
#define FT_PASS
#define FT_DECL_B do { } while (0)
#define FT_PASS_B
#define FT_FLUSH_B(T) do { } while (0)
#define FT_FLUSH(T) do { } while (0)
#endif

// the fast bodies (defined behind the general ones)
template <bool GATHER>
__device__ __forceinline__ const float* mlp_fwd_fast(const MlpArgs& A, float* smem, float* logit_lds, const GatherArgs& G,
                                                     float* p0s, float* p1s, const float* bias_p, float& h_bias FT_ARG);
__device__ __forceinline__ void mlp_bwd_fast(const MlpArgs& A, float* smem, const float* g_lds, const float* hb0,
                                             const float* hb1, int rs_h FT_ARG);

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
// NT output tiles (16 columns each) of one layer over the K range [kg0, kg0 + klen) whose A rows sit in LDS.
//  * No load in the loop is predicated (a predicated load becomes a branch and serialises the loop on memory
//    latency): columns past N re-read row N-1 (their results are dropped by the epilogue) and a dwordx4 that
//    would leave the row is pulled back inside it (its A elements are zero, and weights are finite).
//  * The weight stream runs kPD-1 iterations ahead of the matrix pipe in a register ring (one iteration is
//    4*NT MFMAs = 128*NT cycles, so fewer tiles => deeper ring to cover the L2 latency).
//  * Addresses are a uniform base + one 32-bit lane offset per tile (+ a per-iteration byte offset shared by the
//    tiles): ~NT+2 vector ALU instructions per iteration next to 4*NT MFMAs.
//  * A tile's k-steps alternate between KS accumulators so that at least four independent MFMA chains are in
//    flight per wave (a dependent 16x16x4 chain leaves the matrix pipe idle between issues).
//  * `between()` runs after the ring prologue has been issued: the kernel stages the A chunk there, so the
//    first weight loads overlap the staging's own memory latency.
template <int NT, typename Between>
__device__ __forceinline__ void fwd_tiles(const float* As, int rs, int kg0, int klen, const LayerDev& Ld,
                                          int tile0, f32x4* acc, int g, int c, Between between) {
  constexpr int kPD = NT == 1 ? 12 : (NT == 2 ? 8 : (NT == 3 ? 6 : 4));
  constexpr int KS = NT >= 4 ? 1 : (NT >= 2 ? 2 : 4);
  const DCTR_GLOBAL char* wbase = (const DCTR_GLOBAL char*)Ld.W;
  // first column this lane reads, pulled back inside the row when the (16-wide, zero-padded in LDS) K range is wider
  // than the weight row itself (K < 12: the lane's A elements are zero there).  Was computed unsigned: for tiny K the
  // offset wrapped and the last row's loads left the allocation (round 2, tools/uninit_probe.py).
  const int col0 = (kg0 + 4 * g) < (Ld.ldw - 4) ? (kg0 + 4 * g) : (Ld.ldw - 4);
  uint32_t voff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int n = (tile0 + t * kWaves) * 16 + c;
    n = n < Ld.N ? n : Ld.N - 1;
    voff[t] = (static_cast<uint32_t>(n) * static_cast<uint32_t>(Ld.ldw) + static_cast<uint32_t>(col0)) * 4u;
  }
  const float* ap = As + c * rs + 4 * g;
  const int n_it = klen >> 4;
  const uint32_t omax = static_cast<uint32_t>(Ld.ldw - 4 - col0) * 4u;  // largest in-row byte offset (>= 0)
  auto woff = [&](int it) -> uint32_t {
    it = it < n_it ? it : n_it - 1;                       // scalar: `it` is wave-uniform
    const uint32_t o = static_cast<uint32_t>(it) << 6;
    return o < omax ? o : omax;
  };
  f32x4 ring[kPD][NT];
#pragma unroll
  for (int d = 0; d < kPD - 1; ++d) {
    const uint32_t o = woff(d);
#pragma unroll
    for (int t = 0; t < NT; ++t) ring[d][t] = *(const DCTR_GLOBAL f32x4*)(wbase + (voff[t] + o));
  }
  between();
  f32x4 accs[NT][KS];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    accs[t][0] = acc[t];
#pragma unroll
    for (int k = 1; k < KS; ++k) accs[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int n_grp = n_it / kPD, rem = n_it - n_grp * kPD;
  f32x4 a_nxt = *reinterpret_cast<const f32x4*>(ap);
  for (int gi = 0; gi < n_grp; ++gi) {
#pragma unroll
    for (int d = 0; d < kPD; ++d) {
      const int it = gi * kPD + d;
      const uint32_t o = woff(it + kPD - 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) ring[(d + kPD - 1) % kPD][t] = *(const DCTR_GLOBAL f32x4*)(wbase + (voff[t] + o));
      const f32x4 a4 = a_nxt;
      const int itn = it + 1 < n_it ? it + 1 : it;
      a_nxt = *reinterpret_cast<const f32x4*>(ap + (itn << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) accs[t][j % KS] = mfma16(a4[j], ring[d][t][j], accs[t][j % KS]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < kPD - 1; ++d) {
    if (d < rem) {
      const int it = n_grp * kPD + d;
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + (it << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) accs[t][j % KS] = mfma16(a4[j], ring[d][t][j], accs[t][j % KS]);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4 r = accs[t][0];
#pragma unroll
    for (int k = 1; k < KS; ++k) r += accs[t][k];
    acc[t] = r;
  }
}

template <typename Between>
__device__ __forceinline__ void fwd_dispatch(int nt, const float* As, int rs, int kg0, int klen, const LayerDev& Ld,
                                             int tile0, f32x4* acc, int g, int c, Between between) {
  switch (nt) {
    case 1: fwd_tiles<1>(As, rs, kg0, klen, Ld, tile0, acc, g, c, between); break;
    case 2: fwd_tiles<2>(As, rs, kg0, klen, Ld, tile0, acc, g, c, between); break;
    case 3: fwd_tiles<3>(As, rs, kg0, klen, Ld, tile0, acc, g, c, between); break;
    case 4: fwd_tiles<4>(As, rs, kg0, klen, Ld, tile0, acc, g, c, between); break;
    default: between(); break;   // a wave without tiles still takes part in the staging barriers
  }
}

// the forward of one 16-sample row tile; `logit_lds` (nullable): [16] LDS floats that receive the projection.
// CROSS: the layers are the matrix form of CrossNet (interaction.py:448-451) instead of Linear + activation:
//     u_l = x_l W_l^T + b_l ;  x_{l+1} = x_0 (.) u_l + x_l          (every layer W x W, x_0 = the staged input tile)
// u_l is parked in the layer's `dh` buffer (the backward needs it and overwrites it with its own d loss / d u_l).
// Returns the LDS tile [16][rsh] that holds the top layer's output when the body ends.
template <bool CROSS = false>
__device__ __forceinline__ const float* mlp_fwd_body(const MlpArgs& A, float* smem, float* logit_lds) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kTM;
  const int rsx = A.rsx, rsh = A.rsh;
  float* xs = smem;               // [16][rsx]  chunk of the tower input
  float* hb0 = xs + kTM * rsx;    // [16][rsh]  ping
  float* hb1 = hb0 + kTM * rsh;   // [16][rsh]  pong
  const int K0 = A.L[0].K, K0p = round_up(K0, 16);
  const int kcw = K0p < A.kc ? K0p : A.kc;
  MLP_TRACE(A.trace, 0);
  // the projection's weights, requested now and used after the last layer (it used to wait for them there)
  float wo_pre[4] = {0.f, 0.f, 0.f, 0.f};
  if (A.w_out && (A.logit || logit_lds)) {
    const int ntop = A.L[A.n_layers - 1].N;
#pragma unroll
    for (int i = 0; i < 4; ++i) wo_pre[i] = ldg_f32(A.w_out + ((lane + 64 * i) < ntop ? (lane + 64 * i) : ntop - 1));
  }

  const float* in = nullptr;
  for (int l = 0; l < A.n_layers; ++l) {
    const LayerDev& Ld = A.L[l];
    const int ntile = (Ld.N + 15) >> 4;
    float* outb = (l & 1) ? hb1 : hb0;
    for (int tbase = 0; tbase < ntile; tbase += kWaves * kNTMax) {
      const int tile0 = tbase + wv;
      int nt = 0;
#pragma unroll
      for (int t = 0; t < kNTMax; ++t) nt += (tile0 + t * kWaves < ntile) ? 1 : 0;
      f32x4 acc[kNTMax];
#pragma unroll
      for (int t = 0; t < kNTMax; ++t) {
        const int n = (tile0 + t * kWaves) * 16 + c;
        const float bv = (t < nt && n < Ld.N && Ld.bias) ? ldg_f32(Ld.bias + n) : 0.f;
        acc[t] = f32x4{bv, bv, bv, bv};
      }
      if (l == 0) {
        for (int kc = 0; kc < K0p; kc += kcw) {
          const int klen = (K0p - kc) < kcw ? (K0p - kc) : kcw;
          auto stage = [&]() {
            __syncthreads();  // the previous chunk (or pass) is consumed
            // No load here is predicated (a predicated load is a branch around the load with its own vmcnt(0): the
            // loop used to be one memory round trip per iteration, 4.9 us for the 27 KB tile of the DeepFM tower --
            // round 3): rows past B re-read row B-1, a dwordx4 that would leave the row is pulled back inside it
            // (ld_x % 4 == 0), and what lies past K0 or B is zeroed by a select on the way to LDS.
            const int q4 = klen >> 2, n_e = kTM * q4;
            const int64_t blast = A.B - 1;
            for (int e0 = 0; e0 < n_e; e0 += 4 * kT) {
              f32x4 v[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                int e = e0 + i * kT + tid;
                e = e < n_e ? e : n_e - 1;
                const int r = e / q4, q = e - r * q4;
                const int k = kc + 4 * q;
                const int64_t b = (b0 + r) < blast ? (b0 + r) : blast;
                const int64_t kk = k < A.ldx - 4 ? k : A.ldx - 4;
                v[i] = ldg_f4(A.x + b * A.ldx + kk);
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int e = e0 + i * kT + tid;
                if (e < n_e) {
                  const int r = e / q4, q = e - r * q4;
                  const int k = kc + 4 * q;
                  const bool rv = b0 + r < A.B && k <= A.ldx - 4;
                  f32x4 w;
                  w.x = (rv && k < K0) ? v[i].x : 0.f;
                  w.y = (rv && k + 1 < K0) ? v[i].y : 0.f;
                  w.z = (rv && k + 2 < K0) ? v[i].z : 0.f;
                  w.w = (rv && k + 3 < K0) ? v[i].w : 0.f;
                  *reinterpret_cast<f32x4*>(xs + r * rsx + 4 * q) = w;
                }
              }
            }
            __syncthreads();
            if (kc == 0 && tbase == 0) MLP_TRACE(A.trace, 1);
          };
          fwd_dispatch(nt, xs, rsx, kc, klen, Ld, tile0, acc, g, c, stage);
        }
      } else {
        fwd_dispatch(nt, in, rsh, 0, round_up(Ld.K, 16), Ld, tile0, acc, g, c, []() {});
      }
      if (tbase == 0) MLP_TRACE(A.trace, 2 + 3 * l);
      // epilogue: activation; keep the tile in LDS for the next layer, save it for the backward
#pragma unroll
      for (int t = 0; t < kNTMax; ++t) {
        if (t < nt) {
          const int n = (tile0 + t * kWaves) * 16 + c;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            float v = acc[t][r];
            if (CROSS) {
              const float x0v = xs[row * rsx + n];                  // (K0 <= kKC: the whole input tile is staged)
              const float xlv = (l == 0) ? x0v : in[row * rsh + n];
              if (Ld.dh && n < Ld.N && b0 + row < A.B) stg_f32(Ld.dh + static_cast<int64_t>(b0 + row) * Ld.ldh + n, v);
              v = x0v * v + xlv;
            } else if (Ld.relu) {
              v = v > 0.f ? v : 0.f;
            }
            if (n >= Ld.N) v = 0.f;
            outb[row * rsh + n] = v;
            if (Ld.h && n < Ld.N && b0 + row < A.B) stg_f32(Ld.h + static_cast<int64_t>(b0 + row) * Ld.ldh + n, v);
          }
        }
      }
    }
    MLP_TRACE(A.trace, 3 + 3 * l);
    __syncthreads();
    MLP_TRACE(A.trace, 4 + 3 * l);
    in = outb;
  }
  if (A.w_out && (A.logit || logit_lds)) {  // dnn_linear: logit[b] = h_last[b, :] . w_out
    const LayerDev& Lt = A.L[A.n_layers - 1];
    for (int row = wv; row < kTM; row += kWaves) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < Lt.N) s += in[row * rsh + lane + 64 * i] * wo_pre[i];
      for (int n = lane + 256; n < Lt.N; n += 64) s += in[row * rsh + n] * ldg_f32(A.w_out + n);
      s = wave_sum(s);
      if (lane == 0) {
        if (A.logit && b0 + row < A.B) stg_f32(A.logit + b0 + row, s);
        if (logit_lds) logit_lds[row] = s;
      }
    }
  }
  MLP_TRACE(A.trace, 15);
  return in;
}

__global__ __launch_bounds__(kT) void k_mlp_fwd(MlpArgs A) {
  extern __shared__ __align__(16) float smem[];
  FT_DECL;
#ifdef DCTR_FAST_ONLY   // (ISA reading aid: compile the fast body alone)
  float nb_ = 0.f;
  mlp_fwd_fast<false>(A, smem, nullptr, GatherArgs{}, nullptr, nullptr, nullptr, nb_ FT_PASS);
#else
  if (A.fast) {
    float nb_ = 0.f;
    mlp_fwd_fast<false>(A, smem, nullptr, GatherArgs{}, nullptr, nullptr, nullptr, nb_ FT_PASS);
    FT_FLUSH(A.trace);
  } else {
    mlp_fwd_body<false>(A, smem, nullptr);
  }
#endif
}

__global__ __launch_bounds__(kT) void k_cross_mat_fwd(MlpArgs A) {
  extern __shared__ __align__(16) float smem[];
  mlp_fwd_body<true>(A, smem, nullptr);
}

// ------------------------------------------------------------------------------------------------------------
// backward, data path: dH_l (gradient w.r.t. the pre-activation of layer l) for every layer, then d/d input
// ------------------------------------------------------------------------------------------------------------
// Q consecutive output columns per lane (a group of 16 Q columns per wave pass) of d loss / d input of layer l:
//     out[16 rows][16 Q cols] = din[16][Np] . W_l[Np][cols]          (reduction over the layer's N outputs)
// then the epilogue: relu mask of the layer below, into LDS (`dout`) for the next pass and into its `dh`; at l = 0 into gx.
//  * Q = 4: 64-column groups, dwordx4 weight loads; Q = 2: 32-column groups, dwordx2 -- chosen when the 64-column groups
//    would leave waves without work (a 256-wide layer has 4 of them for 8 waves: round 3).
//  * ldw % 4 == 0: a load at col0 < ldw stays inside the row.  Columns past it re-read column 0 and rows past N re-read
//    row N-1 (the A operand is zero there): no predicated loads in the loop.
//  * The relu mask of the layer below (its saved output h) is requested BEFORE the weight ring and the MFMA loop and
//    consumed in the epilogue (it used to be loaded there: one exposed round trip per pass).
template <int Q>
__device__ __forceinline__ void bwd_cols(const MlpArgs& A, int l, const float* din, float* dout, int rs, int gb, int b0,
                                         int g, int c) {
  typedef float vecq __attribute__((ext_vector_type(Q)));
  const LayerDev& Ld = A.L[l];
  const int Np = round_up(Ld.N, 16);      // reduction length
  const int Kp = round_up(Ld.K, 16);      // output columns kept in LDS for the next (lower) layer
  const int col0 = 16 * Q * gb + Q * c;
  const int colc = col0 < Ld.ldw ? col0 : 0;
  const int lp = l > 0 ? l - 1 : 0;
  const LayerDev& Lp = A.L[lp];           // (l == 0: only its buffer is borrowed as a valid address)
  vecq hpre[4];
  {
    const int cc = col0 < Lp.ldh - Q ? col0 : Lp.ldh - Q;
    const int64_t blast = A.B - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t b = (b0 + 4 * g + r) < blast ? (b0 + 4 * g + r) : blast;
      hpre[r] = *(const DCTR_GLOBAL vecq*)(Lp.h + b * Lp.ldh + cc);
    }
  }
  f32x4 acc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = din + c * rs + 4 * g;
  const int n_it = Np >> 4;
  // uniform base + 32-bit lane offsets: row n = 16 it + 4 g + j of W starts at byte (n * ldw + colc) * 4
  const DCTR_GLOBAL char* wbase = (const DCTR_GLOBAL char*)Ld.W;
  const uint32_t ldw4 = static_cast<uint32_t>(Ld.ldw) * 4u;
  const uint32_t vlast = static_cast<uint32_t>(Ld.N - 1) * ldw4 + static_cast<uint32_t>(colc) * 4u;
  uint32_t vrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) vrow[j] = static_cast<uint32_t>(4 * g + j) * ldw4 + static_cast<uint32_t>(colc) * 4u;
  auto wld = [&](int it, vecq* dst) {
    it = it < n_it ? it : n_it - 1;                               // scalar
    const uint32_t so = static_cast<uint32_t>(it) * 16u * ldw4;  // scalar
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t o = vrow[j] + so;
      o = o < vlast ? o : vlast;                                  // rows past N re-read row N-1 (A is 0 there)
      dst[j] = *(const DCTR_GLOBAL vecq*)(wbase + o);
    }
  };
  constexpr int PD = Q == 4 ? 5 : 8;      // an iteration is 4 Q MFMAs: fewer columns => deeper ring to cover the L2 latency
  vecq ring[PD][4];
#pragma unroll
  for (int d = 0; d < PD - 1; ++d) wld(d, ring[d]);
  const int n_grp = n_it / PD, rem = n_it - n_grp * PD;
  f32x4 a_nxt = *reinterpret_cast<const f32x4*>(ap);
  for (int gi = 0; gi < n_grp; ++gi) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int it = gi * PD + d;
      wld(it + PD - 1, ring[(d + PD - 1) % PD]);
      const f32x4 a4 = a_nxt;
      const int itn = it + 1 < n_it ? it + 1 : it;
      a_nxt = *reinterpret_cast<const f32x4*>(ap + (itn << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = mfma16(a4[j], ring[d][j][q], acc[q]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < PD - 1; ++d) {
    if (d < rem) {
      const int it = n_grp * PD + d;
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + (it << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = mfma16(a4[j], ring[d][j][q], acc[q]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * g + r;
    const int64_t b = b0 + row;
    vecq v;
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = acc[q][r];
    if (l > 0) {
      if (col0 < Kp) {    // Lp.N == Ld.K
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          float o = v[q];
          if (Lp.relu && !(hpre[r][q] > 0.f)) o = 0.f;
          if (col0 + q >= Lp.N || b >= A.B) o = 0.f;
          v[q] = o;
        }
        *reinterpret_cast<vecq*>(dout + row * rs + col0) = v;
        if (Lp.dh && b < A.B) {
          if (col0 + Q - 1 < Lp.N) *(DCTR_GLOBAL vecq*)(Lp.dh + b * Lp.ldh + col0) = v;
          else
            for (int q = 0; q < Q; ++q)
              if (col0 + q < Lp.N) stg_f32(Lp.dh + b * Lp.ldh + col0 + q, v[q]);
        }
      }
    } else if (b < A.B) {
      // columns [K, ldgx) of gx are padding: written as zeros so that no garbage is ever handed on
      if (col0 + Q - 1 < A.ldgx) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (col0 + q >= Ld.K) v[q] = 0.f;
        *(DCTR_GLOBAL vecq*)(A.gx + b * A.ldgx + col0) = v;
      } else {
        for (int q = 0; q < Q; ++q)
          if (col0 + q < A.ldgx) stg_f32(A.gx + b * A.ldgx + col0 + q, col0 + q < Ld.K ? v[q] : 0.f);
      }
    }
  }
}

// the backward-data pass of one row tile; `g_lds` (nullable): [16] LDS floats holding d loss / d logit of the tile's
// rows (the fused train kernel) instead of A.g; `htop` (nullable): the top layer's output tile still in LDS
// ([16][rs_h], the fused train kernel) instead of its saved copy in global memory
__device__ __forceinline__ void mlp_bwd_body(const MlpArgs& A, float* smem, const float* g_lds, const float* htop,
                                             int rs_h, unsigned long long* tr) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kTM;
  const int rs = A.rsd;
  float* d0 = smem;
  float* d1 = d0 + kTM * rs;
  const int top = A.n_layers - 1;
  MLP_TRACE(tr, 0);
  {
    // d loss / d pre-activation of the top layer.  Four elements per thread and round trip, every load unconditional
    // from a clamped address (this loop was one round trip per element: 3.5 us for 16 x 128 -- round 3)
    const LayerDev& Lt = A.L[top];
    const int Np = round_up(Lt.N, 16);
    const int n_e = kTM * Np;
    const int64_t blast = A.B - 1;
    for (int e0 = 0; e0 < n_e; e0 += 4 * kT) {
      float gv[4], hv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int e = e0 + i * kT + tid;
        e = e < n_e ? e : n_e - 1;
        const int r = e / Np, n = e - r * Np;
        const int nn = n < Lt.N ? n : Lt.N - 1;
        const int64_t b = (b0 + r) < blast ? (b0 + r) : blast;
        if (A.w_out) gv[i] = (g_lds ? g_lds[r] : ldg_f32(A.g + b)) * ldg_f32(A.w_out + nn);
        else gv[i] = ldg_f32(A.g + b * A.ldg + nn);
        hv[i] = htop ? htop[r * rs_h + nn] : ldg_f32(Lt.h + b * Lt.ldh + nn);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = e0 + i * kT + tid;
        if (e < n_e) {
          const int r = e / Np, n = e - r * Np;
          const int64_t b = b0 + r;
          float v = 0.f;
          if (b < A.B && n < Lt.N) {
            v = gv[i];
            if (Lt.relu) v = hv[i] > 0.f ? v : 0.f;
            if (Lt.dh) stg_f32(Lt.dh + b * Lt.ldh + n, v);
          }
          d0[r * rs + n] = v;
        }
      }
    }
  }
  __syncthreads();
  MLP_TRACE(tr, 1);
  float* din = d0;
  float* dout = d1;
  for (int l = top; l >= 0; --l) {
    const LayerDev& Ld = A.L[l];
    if (l > 0 || A.gx) {
      if (Ld.K <= 32 * kWaves) {
        const int ngroups = (Ld.K + 31) >> 5;
        for (int gb = wv; gb < ngroups; gb += kWaves) bwd_cols<2>(A, l, din, dout, rs, gb, b0, g, c);
      } else {
        const int ngroups = (Ld.K + 63) >> 6;
        for (int gb = wv; gb < ngroups; gb += kWaves) bwd_cols<4>(A, l, din, dout, rs, gb, b0, g, c);
      }
    }
    MLP_TRACE(tr, 2 + 2 * (top - l));
    __syncthreads();
    MLP_TRACE(tr, 3 + 2 * (top - l));
    float* t = din;
    din = dout;
    dout = t;
  }
  MLP_TRACE(tr, 15);
}


// ------------------------------------------------------------------------------------------------------------
// fast path (round 3): towers whose layers are at most 512 wide and whose input is one staged chunk -- every
// BASELINE configuration.  Same arithmetic as the general bodies above (same MFMA order per output element), other
// schedule: the weight ring of the NEXT phase is requested before the current phase's epilogue and barrier, the tower
// input before the first weight ring, biases / relu masks / projection weights a phase ahead.  In the general
// bodies every phase starts with an exposed L2 round trip of every wave (~1-2 us, 6-8 phases per launch).
// ------------------------------------------------------------------------------------------------------------
// A 16-row tile [16][rs] in LDS -> rows b0 .. b0 + 15 of a [B, ld] global matrix, N columns (ld % 4 == 0, N <= 512), as
// dwordx4 stores by all 512 threads: 32 threads cover 512 contiguous bytes of a row.  Called behind the barrier that
// completes the tile.  (The epilogues used to store their accumulators themselves, one dword per lane and matrix row: 8
// store instructions per wave that took ~3 us to ISSUE -- measured with a stamp behind them, round 3.)
// (wt: write-through stores -- the reader is a kernel on ANOTHER queue that waits for a word in memory, not for this launch's
// end: dctr_mlp_train_wgrad_sync behind dctr_embed_tower_train_step_sync)
__device__ __forceinline__ void tile_store(const float* tile, int rs, float* dst, int64_t ld, int N, int b0, int B,
                                           bool wt = false) {
  const int tid = threadIdx.x, r = tid >> 5, q0 = tid & 31;
  const int n4 = (N + 3) >> 2;
  if (b0 + r < B) {
    float* drow = dst + static_cast<int64_t>(b0 + r) * ld;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = q0 + 32 * i;
      if (q < n4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * rs + 4 * q);
        if (wt) stg_wt(drow + 4 * q, v);
        else *(DCTR_GLOBAL f32x4*)(drow + 4 * q) = v;
      }
    }
  }
}

// Forward: a wave always carries TWO weight streams (ring slots 2 d + t) and two accumulator pairs:
//   * layers of more than 128 outputs: the streams are two 16-column tiles (wv + 16 pass, wv + 8 + 16 pass), up to two
//     passes (512 outputs);
//   * layers of at most 128 outputs (one tile per wave): the streams are the two HALVES OF K of that tile, summed at the end.
// One code path for every layer -- no template dispatch on the tile count.  (With one instantiation per tile count behind a
// switch, the ring registers were copied at every arm's entry, and a copy of a register that a load is still filling
// is an s_waitcnt vmcnt(0): every phase drained the loads it was meant to overlap.)
constexpr int kFPD = 8;                    // ring depth in iterations
constexpr int kRingSlots = 2 * kFPD;       // dwordx4 weight loads a wave keeps in flight (forward): 14 + the 2 being used

struct FwdPlan {      // one pass of one layer, wave-uniform
  int col_t1;         // tile index distance of stream 1 (8: column mode, 0: K-split)
  int k_t1;           // first K iteration of stream 1 (0: column mode, h: K-split)
  int n_it;           // iterations of the pass (column mode: all of K; K-split: h = ceil(all / 2))
  int n_all;          // iterations of the whole K range
};
__device__ __forceinline__ FwdPlan fwd_plan(const LayerDev& Ld, int klen) {
  FwdPlan P;
  const int ntile = (Ld.N + 15) >> 4;
  P.n_all = klen >> 4;
  if (ntile > kWaves || (P.n_all & 1)) {
    // (one tile per wave but an odd K iteration count: stream 1 computes a second copy of column N-1's tile and is
    // dropped -- towers have even K / 16 in practice)
    P.col_t1 = kWaves; P.k_t1 = 0; P.n_it = P.n_all;
  } else {
    P.col_t1 = 0; P.n_it = P.n_all >> 1; P.k_t1 = P.n_it;
  }
  return P;
}

struct FwdAddr {
  const DCTR_GLOBAL char* wbase;
  uint32_t voff[2];
  uint32_t omax;
};
__device__ __forceinline__ FwdAddr fwd_addr(const LayerDev& Ld, const FwdPlan& P, int tile0, int g, int c) {
  FwdAddr a;
  a.wbase = (const DCTR_GLOBAL char*)Ld.W;
  const int col0 = (4 * g) < (Ld.ldw - 4) ? (4 * g) : (Ld.ldw - 4);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int n = (tile0 + t * P.col_t1) * 16 + c;
    n = n < Ld.N ? n : Ld.N - 1;
    a.voff[t] = (static_cast<uint32_t>(n) * static_cast<uint32_t>(Ld.ldw) + static_cast<uint32_t>(col0)) * 4u;
  }
  a.omax = static_cast<uint32_t>(Ld.ldw - 4 - col0) * 4u;
  return a;
}
// byte offset of K iteration `it` of stream t inside the weight row (clamped into the row: what lies past K meets zeros)
__device__ __forceinline__ uint32_t fwd_woff(const FwdAddr& a, const FwdPlan& P, int it, int t) {
  it = it < P.n_it ? it : P.n_it - 1;                       // scalar
  it += t * P.k_t1;
  it = it < P.n_all ? it : P.n_all - 1;
  const uint32_t o = static_cast<uint32_t>(it) << 6;
  return o < a.omax ? o : a.omax;
}

__device__ __forceinline__ void fwd_fill(f32x4 (&ring)[kRingSlots], const LayerDev& Ld, int klen, int tile0, int g, int c,
                                         uint32_t wmask) {
  const FwdPlan P = fwd_plan(Ld, klen);
  const FwdAddr a = fwd_addr(Ld, P, tile0, g, c);
#pragma unroll
  for (int d = 0; d < kFPD - 1; ++d)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      ring[2 * d + t] = *(const DCTR_GLOBAL f32x4*)(a.wbase + ((a.voff[t] + fwd_woff(a, P, d, t)) & wmask));
}

// acc[0], acc[1]: the two tiles (column mode) or the tile's sum in acc[0] (K-split; acc[1] unused)
__device__ __forceinline__ void fwd_run(f32x4 (&ring)[kRingSlots], const float* As, int rs, const LayerDev& Ld, int klen,
                                        int tile0, int g, int c, f32x4* acc, uint32_t wmask) {
  constexpr int PD = kFPD;
  const FwdPlan P = fwd_plan(Ld, klen);
  const FwdAddr a = fwd_addr(Ld, P, tile0, g, c);
  const float* ap = As + c * rs + 4 * g;
  const int n_it = P.n_it;
  const bool split = P.col_t1 == 0;
  f32x4 accs[2][2];
  accs[0][0] = acc[0];
  accs[1][0] = split ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[1];
  accs[0][1] = accs[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // A fragments: stream 1 reads the same rows of the tile, k_t1 iterations further right (K-split; its range may end one
  // iteration early: that fragment is zeroed)
  auto a_of = [&](int it, int t) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(ap + ((it + t * P.k_t1) << 4));
  };
  const int n_grp = n_it / PD, rem = n_it - n_grp * PD;
  f32x4 a0n = a_of(0, 0), a1n = a_of(0, 1);
  for (int gi = 0; gi < n_grp; ++gi) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int it = gi * PD + d;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        ring[2 * ((d + PD - 1) % PD) + t] =
            *(const DCTR_GLOBAL f32x4*)(a.wbase + ((a.voff[t] + fwd_woff(a, P, it + PD - 1, t)) & wmask));
      const f32x4 a0 = a0n, a1 = a1n;
      const int itn = it + 1 < n_it ? it + 1 : it;
      a0n = a_of(itn, 0);
      a1n = a_of(itn, 1);
      // (grouped, not interleaved: with the iteration's address arithmetic, loads and LDS reads spread between the MFMAs
      // by sched_group_barrier the first layer's phase took 8.8 us against 8.5.  Round 3 also tried, without effect on
      // the kernel's time: warming each XCD's L2 with the weights from the workgroups' first instructions, and
      // write-through (sc0 sc1) stores of the saved activations / gradients)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        accs[0][j & 1] = mfma16(a0[j], ring[2 * d][j], accs[0][j & 1]);
        accs[1][j & 1] = mfma16(a1[j], ring[2 * d + 1][j], accs[1][j & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < PD - 1; ++d) {
    if (d < rem) {
      const int it = n_grp * PD + d;
      const f32x4 a0 = a_of(it, 0), a1 = a_of(it, 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        accs[0][j & 1] = mfma16(a0[j], ring[2 * d][j], accs[0][j & 1]);
        accs[1][j & 1] = mfma16(a1[j], ring[2 * d + 1][j], accs[1][j & 1]);
      }
    }
  }
  const f32x4 r0 = accs[0][0] + accs[0][1], r1 = accs[1][0] + accs[1][1];
  acc[0] = split ? r0 + r1 : r0;
  acc[1] = r1;
}

// bias of the wave's (up to two) tiles of a pass: the RAW loads, unconditional from a clamped address (Ld.W stands in for a
// missing bias vector); fwd_bias_sel() turns them into the values (0 where there is no bias / no tile) at the point of use,
// so that nothing waits for them earlier
__device__ __forceinline__ void fwd_bias_ld(const LayerDev& Ld, int tile0, int c, float* raw) {
  const float* bp = Ld.bias ? Ld.bias : Ld.W;
  const int col_t1 = fwd_plan(Ld, round_up(Ld.K, 16)).col_t1;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = (tile0 + t * col_t1) * 16 + c;
    raw[t] = ldg_f32(bp + (n < Ld.N ? n : Ld.N - 1));
  }
}
__device__ __forceinline__ void fwd_bias_sel(const LayerDev& Ld, int tile0, int c, const float* raw, float* bv) {
  const int col_t1 = fwd_plan(Ld, round_up(Ld.K, 16)).col_t1;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = (tile0 + t * col_t1) * 16 + c;
    bv[t] = (n < Ld.N && Ld.bias && (t == 0 || col_t1)) ? raw[t] : 0.f;
  }
}

// LDS scratch of the gather stage (it lies in the backward's image, which nobody touches before the head is done)
struct GatherLds {
  const dctr_field_t* deep;   // staged descriptors
  const dctr_field_t* wide;
  const float* x;             // [16][nc] the X tile (rows past B repeat row B-1)
  const int32_t* dcol;        // [n_dense]
  const int32_t* wcol;        // [n_wdense]
  const float* ww;            // [n_wdense]
  const int32_t* gsd;         // [n_gsd] pooled deep positions
  const int32_t* gsw;         // [n_gsw] pooled wide positions
  float* wv;                  // [16][32] the wide tables' values (0 past n_wide / B)
  float* st;                  // [16][D] sum_f e
  float* tm;                  // [16][D] (sum_f e)^2 - sum_f e^2
  float* pb;                  // [16][n_gsd][D] the rows of the pooled deep fields' positions (0 where masked out)
  float* pw;                  // [16][n_gsw] the pooled wide fields' values per position
};
__host__ __device__ __forceinline__ int gather_stage_words(int n_deep, int n_wide, int nc, int n_dense, int n_wdense,
                                                           int n_gsd = 0, int n_gsw = 0) {
  return 16 * (n_deep + n_wide) + kTM * nc + n_dense + 2 * n_wdense + n_gsd + n_gsw;
}
__host__ __device__ __forceinline__ int gather_lds_words(int n_deep, int n_wide, int nc, int n_dense, int n_wdense, int D,
                                                         int n_gsd = 0, int n_gsw = 0) {
  return round_up(gather_stage_words(n_deep, n_wide, nc, n_dense, n_wdense, n_gsd, n_gsw), 4) + kTM * 32 + 2 * kTM * D +
         kTM * n_gsd * D + round_up(kTM * n_gsw, 4);
}
__device__ __forceinline__ GatherLds gather_lds(const GatherArgs& G, float* base) {
  GatherLds S;
  uint32_t* w = reinterpret_cast<uint32_t*>(base);
  S.deep = reinterpret_cast<const dctr_field_t*>(w); w += 16 * G.n_deep;
  S.wide = reinterpret_cast<const dctr_field_t*>(w); w += 16 * G.n_wide;
  S.x = reinterpret_cast<const float*>(w); w += kTM * G.nc;
  S.dcol = reinterpret_cast<const int32_t*>(w); w += G.n_dense;
  S.wcol = reinterpret_cast<const int32_t*>(w); w += G.n_wdense;
  S.ww = reinterpret_cast<const float*>(w); w += G.n_wdense;
  S.gsd = reinterpret_cast<const int32_t*>(w); w += G.n_gsd;
  S.gsw = reinterpret_cast<const int32_t*>(w); w += G.n_gsw;
  w = reinterpret_cast<uint32_t*>(base) +
      round_up(gather_stage_words(G.n_deep, G.n_wide, G.nc, G.n_dense, G.n_wdense, G.n_gsd, G.n_gsw), 4);
  S.wv = reinterpret_cast<float*>(w); w += kTM * 32;
  S.st = reinterpret_cast<float*>(w); w += kTM * G.D;
  S.tm = reinterpret_cast<float*>(w); w += kTM * G.D;
  S.pb = reinterpret_cast<float*>(w); w += kTM * G.n_gsd * G.D;
  S.pw = reinterpret_cast<float*>(w);
  return S;
}

template <bool GATHER>
__device__ __forceinline__ const float* mlp_fwd_fast(const MlpArgs& A, float* smem, float* logit_lds, const GatherArgs& G,
                                                     float* p0s, float* p1s, const float* bias_p, float& h_bias FT_ARG) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kTM;
  const int rsx = A.rsx, rsh = A.rsh;
  float* xs = smem;               // [16][rsx]  the tower input
  float* hb0 = xs + kTM * rsx;    // [16][rsh]  ping
  float* hb1 = hb0 + kTM * rsh;   // [16][rsh]  pong
  const int K0 = A.L[0].K, K0p = round_up(K0, 16);
  FT((0) & 15);
  // (loads return in order: what is consumed first is requested first -- input tile, bias, then the weight ring)
  // the tower input: thread (row tid / 32, dwordx4 columns tid % 32 + 32 i); nothing predicated -- rows past B re-read
  // row B-1, columns past the row are pulled back inside it, both are zeroed on the way to LDS
  const int q4 = K0p >> 2;
  const int xr = tid >> 5, xq = tid & 31;
  f32x4 xv[4];
  f32x4 ring[kRingSlots];
  float braw[2];
  float wo_pre[4] = {0.f, 0.f, 0.f, 0.f};
  int bias_l = 0, bias_tile0 = wv;        // which layer / tile the raw bias values in flight belong to
  GatherLds S{};
  if constexpr (GATHER) {
    // ---- round trip 1: the field descriptors, the X tile and the dense-column tables -> LDS (<= 4 words per thread, all in
    // flight together; one concatenated index space, no load behind a branch)
    S = gather_lds(G, smem + G.scratch_off);
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem + G.scratch_off);
    const int e0 = 16 * G.n_deep, e1 = e0 + 16 * G.n_wide, e2 = e1 + kTM * G.nc, e3 = e2 + G.n_dense, e4 = e3 + G.n_wdense,
              e4b = e4 + G.n_wdense, e4c = e4b + G.n_gsd, e5 = e4c + G.n_gsw;
    const int64_t blast = A.B - 1;
    uint32_t sv[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      int idx = tid + kT * s_;
      idx = idx < e5 ? idx : e5 - 1;
      const int j = idx - e1;
      const int r = j / G.nc, cc = j - r * G.nc;
      const int64_t b = (b0 + r) < blast ? (b0 + r) : blast;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(G.deep) + idx;
      src = idx >= e0 ? reinterpret_cast<const uint32_t*>(G.wide) + (idx - e0) : src;
      src = idx >= e1 ? reinterpret_cast<const uint32_t*>(G.X + b * G.ldx + cc) : src;
      src = idx >= e2 ? reinterpret_cast<const uint32_t*>(G.dense_cols) + (idx - e2) : src;
      src = idx >= e3 ? reinterpret_cast<const uint32_t*>(G.wdense_cols) + (idx - e3) : src;
      src = idx >= e4 ? reinterpret_cast<const uint32_t*>(G.wdense_w) + (idx - e4) : src;
      src = idx >= e4b ? reinterpret_cast<const uint32_t*>(G.gsd) + (idx - e4b) : src;
      src = idx >= e4c ? reinterpret_cast<const uint32_t*>(G.gsw) + (idx - e4c) : src;
      sv[s_] = *(const DCTR_GLOBAL uint32_t*)src;
    }
    // the tile's columns that no table row fills (dense block, padding up to K0p) start as zeros
    const int nqd = G.n_deep << G.lpr_shift;          // quads of the row that table rows fill ...
    const int nqf = G.n_deep_fixed << G.lpr_shift;    // ... of which the fixed-length fields' (one row each)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = xq + 32 * i;
      if (q >= nqd && q < q4) *reinterpret_cast<f32x4*>(xs + xr * rsx + 4 * q) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
      if (tid + kT * s_ < e5) stage[tid + kT * s_] = sv[s_];
    __syncthreads();
    FT(10);
    // ---- round trip 2: every table row of the tile (16 x n_deep x D / 4 dwordx4 pieces + 16 x n_wide floats), then the
    // bias and the first layer's weight ring behind them
    const bool rvalid = b0 + xr < A.B;
    const float* xrow = S.x + xr * G.nc;
    int bad = 0;
    const int lmask = (1 << G.lpr_shift) - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = xq + 32 * i;
      const int qq = q < nqf ? q : nqf - 1;
      const dctr_field_t& fd = S.deep[qq >> G.lpr_shift];
      const int32_t rid = static_cast<int32_t>(xrow[fd.col]);            // Tensor.long(): truncation (basemodel.py:369)
      const bool oob = static_cast<uint32_t>(rid) >= static_cast<uint32_t>(fd.vocab);
      bad |= (oob && q < nqf) ? 1 : 0;
      const int64_t id = oob ? 0 : rid;
      xv[i] = ldg_f4(fd.table + id * row_ld(fd) + 4 * (qq & lmask));
    }
    float wval = 0.f;
    if (G.n_wide_fixed > 0) {
      const dctr_field_t& fw = S.wide[xq < G.n_wide_fixed ? xq : G.n_wide_fixed - 1];
      const int32_t rid = static_cast<int32_t>(xrow[fw.col]);
      const bool oob = static_cast<uint32_t>(rid) >= static_cast<uint32_t>(fw.vocab);
      bad |= (oob && xq < G.n_wide_fixed) ? 1 : 0;
      wval = ldg_f32(fw.table + (oob ? 0 : static_cast<int64_t>(rid)) * row_ld(fw));
    }
    // pooled VarLen fields (inputs.py:141-155, sequence.py:49-77): every position's row, up to four 16-byte pieces and one
    // wide value per thread, in flight with the rows above; a masked-out position (id == 0, or t >= length) parks zeros
    f32x4 pv[4];
    int pdst[4];
    float pwv = 0.f;
    int pwdst = -1;
    if (G.n_gsd > 0) {
      const int per_row = G.n_gsd << G.lpr_shift, ni = kTM * per_row;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int w_ = tid + kT * i;
        const int wc = w_ < ni ? w_ : ni - 1;
        const int r = wc / per_row, rem = wc - r * per_row;
        const int sl_ = rem >> G.lpr_shift, piece = rem & lmask;
        const int ft = S.gsd[sl_];
        const dctr_field_t& fd = S.deep[ft >> 16];
        const int t = ft & 0xFFFF;
        const float* xr_ = S.x + r * G.nc;
        const int32_t rid = static_cast<int32_t>(xr_[fd.col + t]);
        const bool m = fd.len_col >= 0 ? (t < static_cast<int32_t>(xr_[fd.len_col])) : (rid != 0);
        const bool oob = static_cast<uint32_t>(rid) >= static_cast<uint32_t>(fd.vocab);
        const bool live = w_ < ni && b0 + r < A.B;
        bad |= (oob && live) ? 1 : 0;
        pv[i] = ldg_f4(fd.table + (oob ? 0 : static_cast<int64_t>(rid)) * row_ld(fd) + 4 * piece);
        // (max pooling looks at every position -- a masked-out one as row - 1e9, sequence.py:65-68 -- so its rows are parked
        // as they are; sum / mean park zeros for a masked-out position)
        const bool keep = live && (m || fd.pool == DCTR_POOL_MAX);
        pdst[i] = w_ < ni ? ((r * G.n_gsd + sl_) * G.D + 4 * piece) | (keep ? 0 : (1 << 30)) : -1;
      }
    }
    if (G.n_gsw > 0 && tid < kTM * G.n_gsw) {
      const int r = tid / G.n_gsw, sl_ = tid - r * G.n_gsw;
      const int ft = S.gsw[sl_];
      const dctr_field_t& fw = S.wide[ft >> 16];
      const int t = ft & 0xFFFF;
      const float* xr_ = S.x + r * G.nc;
      const int32_t rid = static_cast<int32_t>(xr_[fw.col + t]);
      const bool m = fw.len_col >= 0 ? (t < static_cast<int32_t>(xr_[fw.len_col])) : (rid != 0);
      const bool oob = static_cast<uint32_t>(rid) >= static_cast<uint32_t>(fw.vocab);
      bad |= (oob && b0 + r < A.B) ? 1 : 0;
      pwv = ldg_f32(fw.table + (oob ? 0 : static_cast<int64_t>(rid)) * row_ld(fw));
      pwdst = tid | (((m || fw.pool == DCTR_POOL_MAX) && b0 + r < A.B) ? 0 : (1 << 30));
    }
    // (G.wsync: the dense parameters may still be under the previous step's optimizer step -- nothing of them is requested
    // before the wait below)
    if (!G.wsync) {
      fwd_bias_ld(A.L[0], wv, c, braw);
      __builtin_amdgcn_sched_barrier(0);
      fwd_fill(ring, A.L[0], K0p, wv, g, c, A.wmask);
      if (A.w_out && (A.logit || logit_lds)) {
        const int ntop = A.L[A.n_layers - 1].N;
#pragma unroll
        for (int i = 0; i < 4; ++i) wo_pre[i] = ldg_f32(A.w_out + ((lane + 64 * i) < ntop ? (lane + 64 * i) : ntop - 1));
      }
    }
    // the dense block of combined_dnn_input (inputs.py:126-138): scalars of the X tile, LDS to LDS
    for (int e = tid; e < kTM * G.n_dense; e += kT) {
      const int r = e / G.n_dense, j = e - r * G.n_dense;
      xs[r * rsx + G.dense_off + j] = (b0 + r < A.B) ? S.x[r * G.nc + S.dcol[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = xq + 32 * i;
      if (q < nqf) *reinterpret_cast<f32x4*>(xs + xr * rsx + 4 * q) = rvalid ? xv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    S.wv[xr * 32 + xq] = (rvalid && xq < G.n_wide_fixed) ? wval : 0.f;
    if (G.n_gsd > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (pdst[i] >= 0)
          *reinterpret_cast<f32x4*>(S.pb + (pdst[i] & ~(1 << 30))) = (pdst[i] >> 30) ? f32x4{0.f, 0.f, 0.f, 0.f} : pv[i];
    }
    if (pwdst >= 0) S.pw[pwdst & ~(1 << 30)] = (pwdst >> 30) ? 0.f : pwv;
    if (bad && G.err) atomicOr(G.err, 1);
    if (G.wsync) {
      // ---- the rows are here; now the weights must be: ONE wave polls the generation word (relaxed, past the L1), everybody
      // meets at the barrier, then the bias, the first weight ring and the projection are requested -- what the launch does
      // right behind its row loads when it has nothing to wait for.  No acquire fence: this launch has not read a dense
      // parameter yet (its start invalidated the L1), the reducers never leave one in an L1 (sc1 loads, write-through
      // stores), and lines of ordinary device memory are never stale in an L2 (memory probes).  A wait that runs out raises
      // bit 2 of the plan's error word and goes on: wrong numbers and a raised flag, not a hang.
      if (wv == 0) {
        const int32_t want = __hip_atomic_load(G.wsync + DCTR_SYNC_T_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        for (;;) {
          const int32_t have = __hip_atomic_load(G.wsync + DCTR_SYNC_W_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (have - want >= 0) break;
          if (wall_clock64() - t0 > G.wtimeout) {
            if (lane == 0 && G.err) atomicOr(G.err, 4);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();
      h_bias = ldg_f32(bias_p);
      fwd_bias_ld(A.L[0], wv, c, braw);
      __builtin_amdgcn_sched_barrier(0);
      fwd_fill(ring, A.L[0], K0p, wv, g, c, A.wmask);
      if (A.w_out && (A.logit || logit_lds)) {
        const int ntop = A.L[A.n_layers - 1].N;
#pragma unroll
        for (int i = 0; i < 4; ++i) wo_pre[i] = ldg_f32(A.w_out + ((lane + 64 * i) < ntop ? (lane + 64 * i) : ntop - 1));
      }
    } else {
      __syncthreads();
    }
    if (G.n_deep > G.n_deep_fixed || G.n_wide > G.n_wide_fixed) {
      // ---- the pooling itself, in pool_field's order (csrc/embed.hip): positions added one after the other, a masked-out
      // position adds nothing; mean divides by (count | length) + 1e-8 (sequence.py:72-74)
#pragma clang fp contract(off)
      const int lpr = 1 << G.lpr_shift;
      const int npd = G.n_deep - G.n_deep_fixed, npw = G.n_wide - G.n_wide_fixed;
      for (int e = tid; e < kTM * (npd * lpr + npw); e += kT) {
        const int r = e / (npd * lpr + npw), rem = e - r * (npd * lpr + npw);
        const bool deep_item = rem < npd * lpr;
        const int fi = deep_item ? G.n_deep_fixed + (rem >> G.lpr_shift) : G.n_wide_fixed + (rem - npd * lpr);
        const int piece = deep_item ? (rem & lmask) : 0;
        const dctr_field_t& fd = deep_item ? S.deep[fi] : S.wide[fi];
        int base = 0;     // the field's first position in the flattened list
        for (int g2 = deep_item ? G.n_deep_fixed : G.n_wide_fixed; g2 < fi; ++g2) base += (deep_item ? S.deep[g2] : S.wide[g2]).len;
        const float* xr_ = S.x + r * G.nc;
        const bool by_len = fd.len_col >= 0;
        float den = 1.f;
        if (fd.pool == DCTR_POOL_MEAN) {
          float cnt = 0.f;
          if (by_len) cnt = static_cast<float>(static_cast<int32_t>(xr_[fd.len_col]));
          else
            for (int t = 0; t < fd.len; ++t) cnt += (static_cast<int32_t>(xr_[fd.col + t]) != 0) ? 1.f : 0.f;
          den = cnt + 1e-8f;
        }
        if (fd.pool == DCTR_POOL_MAX) {
          // max_t (e_t - (1 - m_t) 1e9), the FIRST maximum's position per element (pool_field in csrc/embed.hip: strict >)
          const int32_t len_i = by_len ? static_cast<int32_t>(xr_[fd.len_col]) : 0;
          const int nel = deep_item ? 4 : 1;
          float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          int arg[4] = {0, 0, 0, 0};
          for (int t = 0; t < fd.len; ++t) {
            const bool m = by_len ? (t < len_i) : (static_cast<int32_t>(xr_[fd.col + t]) != 0);
            const float pen = m ? 0.f : 1e9f;
            float vals[4] = {0.f, 0.f, 0.f, 0.f};
            if (deep_item) {
              const f32x4 row = *reinterpret_cast<const f32x4*>(S.pb + (r * G.n_gsd + base + t) * G.D + 4 * piece);
              vals[0] = row.x; vals[1] = row.y; vals[2] = row.z; vals[3] = row.w;
            } else {
              vals[0] = S.pw[r * G.n_gsw + base + t];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = vals[i] - pen;
              arg[i] = (v > best[i]) ? t : arg[i];
              best[i] = (v > best[i]) ? v : best[i];
            }
          }
          const bool rv = b0 + r < A.B;
          if (deep_item) *reinterpret_cast<f32x4*>(xs + r * rsx + fi * G.D + 4 * piece) =
              rv ? f32x4{best[0], best[1], best[2], best[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
          else S.wv[r * 32 + fi] = rv ? best[0] : 0.f;
          if (G.amax && rv) {
            const int off = ldg_i32((deep_item ? G.am_deep_off : G.am_wide_off) + fi);
            if (off >= 0) {
              uint8_t* dst = G.amax + static_cast<int64_t>(b0 + r) * G.ld_am + off + (deep_item ? 4 * piece : 0);
              for (int i = 0; i < nel; ++i) *(DCTR_GLOBAL uint8_t*)(dst + i) = static_cast<uint8_t>(arg[i]);
            }
          }
        } else if (deep_item) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
          for (int t = 0; t < fd.len; ++t) acc += *reinterpret_cast<const f32x4*>(S.pb + (r * G.n_gsd + base + t) * G.D + 4 * piece);
          if (fd.pool == DCTR_POOL_MEAN) acc = f32x4{acc.x / den, acc.y / den, acc.z / den, acc.w / den};
          *reinterpret_cast<f32x4*>(xs + r * rsx + fi * G.D + 4 * piece) = (b0 + r < A.B) ? acc : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
          float acc = 0.f;
          for (int t = 0; t < fd.len; ++t) acc += S.pw[r * G.n_gsw + base + t];
          if (fd.pool == DCTR_POOL_MEAN) acc = acc / den;
          S.wv[r * 32 + fi] = (b0 + r < A.B) ? acc : 0.f;
        }
      }
      __syncthreads();
    }
    // ---- FM's per-sample sums in k_embed_fwd's order of additions: wave w of that kernel takes fields w, w + 4, ...; its
    // wave 0 then adds the four partial sums in wave order
    for (int e = tid; e < kTM * G.D; e += kT) {
#pragma clang fp contract(off)      // (k_embed_fwd's roundings: products and sums rounded separately)
      const int r = e / G.D, d = e - r * G.D;
      float st = 0.f, qt = 0.f;
      for (int w = 0; w < 4; ++w) {
        float sw = 0.f, qw = 0.f;
        for (int f = w; f < G.n_deep_fixed; f += 4) {
          const float v = xs[r * rsx + f * G.D + d];
          sw += v;
          qw += v * v;
        }
        for (int f = G.n_deep_fixed + w; f < G.n_deep; f += 4) {     // (k_embed_fwd: the wave's pooled fields follow)
          const float v = xs[r * rsx + f * G.D + d];
          sw += v;
          qw += v * v;
        }
        st += sw;
        qt += qw;
      }
      S.st[e] = st;
      S.tm[e] = st * st - qt;
    }
  } else {
    {
      const int64_t blast = A.B - 1;
      const int64_t b = (b0 + xr) < blast ? (b0 + xr) : blast;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 4 * (xq + 32 * i);
        const int64_t kk = k < A.ldx - 4 ? k : A.ldx - 4;
        xv[i] = ldg_f4(A.x + b * A.ldx + kk);
      }
    }
    fwd_bias_ld(A.L[0], wv, c, braw);
    __builtin_amdgcn_sched_barrier(0);      // (keep the issue order: the compiler otherwise puts the ring in front)
    fwd_fill(ring, A.L[0], K0p, wv, g, c, A.wmask);
    if (A.w_out && (A.logit || logit_lds)) {
      const int ntop = A.L[A.n_layers - 1].N;
#pragma unroll
      for (int i = 0; i < 4; ++i) wo_pre[i] = ldg_f32(A.w_out + ((lane + 64 * i) < ntop ? (lane + 64 * i) : ntop - 1));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = xq + 32 * i;
      if (q < q4) {
        const int k = 4 * q;
        const bool rv = b0 + xr < A.B && k <= A.ldx - 4;
        f32x4 w;
        w.x = (rv && k < K0) ? xv[i].x : 0.f;
        w.y = (rv && k + 1 < K0) ? xv[i].y : 0.f;
        w.z = (rv && k + 2 < K0) ? xv[i].z : 0.f;
        w.w = (rv && k + 3 < K0) ? xv[i].w : 0.f;
        *reinterpret_cast<f32x4*>(xs + xr * rsx + k) = w;
      }
    }
    __syncthreads();
  }
  FT((1) & 15);
  const float* in = xs;
  int rs_in = rsx;
  for (int l = 0; l < A.n_layers; ++l) {
    const LayerDev& Ld = A.L[l];
    const int ntile = (Ld.N + 15) >> 4;
    const int klen = round_up(Ld.K, 16);
    const int col_t1 = fwd_plan(Ld, klen).col_t1;
    float* outb = (l & 1) ? hb1 : hb0;
    for (int tile0 = wv; tile0 < ntile || tile0 == wv; tile0 += 2 * kWaves) {     // (every wave runs the first pass)
      f32x4 acc[2];
      {
        float bv[2];
        fwd_bias_sel(A.L[bias_l], bias_tile0, c, braw, bv);
        acc[0] = f32x4{bv[0], bv[0], bv[0], bv[0]};
        acc[1] = f32x4{bv[1], bv[1], bv[1], bv[1]};
      }
      fwd_run(ring, in, rs_in, Ld, klen, tile0, g, c, acc, A.wmask);
      if (tile0 == wv) FT((2 + 3 * l) & 15);
      // the epilogue's stores go first: the memory pipeline is in order, behind a 14 KB burst of weight requests per
      // wave they waited ~5 us (measured with a stamp between the two: round 3)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int tile = tile0 + t * col_t1;
        if (tile < ntile && (t == 0 || col_t1)) {
          const int n = tile * 16 + c;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            float v = acc[t][r];
            if (Ld.relu) v = v > 0.f ? v : 0.f;
            if (n >= Ld.N) v = 0.f;
            outb[row * rsh + n] = v;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (l == 0) FT(9);
      // what this wave multiplies next -- its tiles of the second pass, or of the next layer -- and the bias that goes
      // with it: in flight across the barrier
      if (tile0 + 2 * kWaves < ntile) {
        bias_l = l; bias_tile0 = tile0 + 2 * kWaves;
        fwd_bias_ld(Ld, bias_tile0, c, braw);
        __builtin_amdgcn_sched_barrier(0);
        fwd_fill(ring, Ld, klen, bias_tile0, g, c, A.wmask);
      } else if (l + 1 < A.n_layers) {
        const LayerDev& Ln = A.L[l + 1];
        bias_l = l + 1; bias_tile0 = wv;
        fwd_bias_ld(Ln, wv, c, braw);
        __builtin_amdgcn_sched_barrier(0);
        fwd_fill(ring, Ln, round_up(Ln.K, 16), wv, g, c, A.wmask);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    FT((3 + 3 * l) & 15);
    __syncthreads();
    FT((4 + 3 * l) & 15);
    if (Ld.h) tile_store(outb, rsh, Ld.h, Ld.ldh, Ld.N, b0, A.B, A.wt != 0);   // the saved activation, for the backward kernels
    if constexpr (GATHER) {
      if (l == 0) {
        // the gathered tile (the first layer's input): its copy for the weight-gradient kernel, padding columns as zeros
        tile_store(xs, rsx, G.out, G.ldo, K0p < static_cast<int>(G.ldo) ? K0p : static_cast<int>(G.ldo), b0, A.B, A.wt != 0);
        // the linear logit and the FM term, finished by lpr lanes per sample the way k_embed_fwd's wave 0 does it
        // (lane gl owns the strip [4 gl, 4 gl + 4) of the row and the wide fields gl, gl + lpr, ... of each "wave" w)
        const int lpr = 1 << G.lpr_shift;
        if (tid < kTM * lpr) {
#pragma clang fp contract(off)
          const int r = tid >> G.lpr_shift, gl = tid & (lpr - 1);
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) t += S.tm[r * G.D + 4 * gl + i];
          float wt = 0.f;
          for (int w = 0; w < 4; ++w) {
            float pw = 0.f;
            for (int f = w * lpr + gl; f < G.n_wide_fixed; f += 4 * lpr) pw += S.wv[r * 32 + f];
            for (int f = G.n_wide_fixed + w * lpr + gl; f < G.n_wide; f += 4 * lpr) pw += S.wv[r * 32 + f];   // pooled
            if (G.wdense_w)
              for (int j = w * lpr + gl; j < G.n_wdense; j += 4 * lpr)
                pw += S.x[r * G.nc + S.wcol[j]] * S.ww[j];
            wt += pw;
          }
          for (int m = lpr >> 1; m >= 1; m >>= 1) {
            t += __shfl_xor(t, m, kWave);
            wt += __shfl_xor(wt, m, kWave);
          }
          if (gl == 0) {
            p0s[r] = wt;
            p1s[r] = 0.5f * t;
          }
          if (G.fm_s && b0 + r < A.B)
            *(DCTR_GLOBAL f32x4*)(G.fm_s + static_cast<int64_t>(b0 + r) * G.lds + 4 * gl) =
                *reinterpret_cast<const f32x4*>(S.st + r * G.D + 4 * gl);
        }
      }
    }
    in = outb;
    rs_in = rsh;
  }
  if (A.w_out && (A.logit || logit_lds)) {  // dnn_linear: logit[b] = h_last[b, :] . w_out
    const LayerDev& Lt = A.L[A.n_layers - 1];
    for (int row = wv; row < kTM; row += kWaves) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < Lt.N) s += in[row * rsh + lane + 64 * i] * wo_pre[i];
      for (int n = lane + 256; n < Lt.N; n += 64) s += in[row * rsh + n] * ldg_f32(A.w_out + n);
      s = wave_sum(s);
      if (lane == 0) {
        if (A.logit && b0 + row < A.B) stg_f32(A.logit + b0 + row, s);
        if (logit_lds) logit_lds[row] = s;
      }
    }
  }
  FT((15) & 15);
  return in;
}

// ---- backward-data, fast path ---------------------------------------------------------------------------------
template <int Q>
struct BwdCfg {
  static constexpr int PD = Q == 4 ? 6 : 10;   // ring depth in iterations of 4 weight rows x Q columns
};
template <int Q>
struct BwdRing {
  typedef float vecq __attribute__((ext_vector_type(Q)));
  vecq r[BwdCfg<Q>::PD][4];
};

struct BwdAddr {
  const DCTR_GLOBAL char* wbase;
  uint32_t vrow[4];
  uint32_t vlast, ldw4, wmask;
  int n_it;
};
template <int Q>
__device__ __forceinline__ BwdAddr bwd_addr(const LayerDev& Ld, int gb, int g, int c, uint32_t wmask) {
  BwdAddr a;
  a.wmask = wmask;
  const int col0 = 16 * Q * gb + Q * c;
  const int colc = col0 < Ld.ldw ? col0 : 0;
  a.wbase = (const DCTR_GLOBAL char*)Ld.W;
  a.ldw4 = static_cast<uint32_t>(Ld.ldw) * 4u;
  a.vlast = static_cast<uint32_t>(Ld.N - 1) * a.ldw4 + static_cast<uint32_t>(colc) * 4u;
#pragma unroll
  for (int j = 0; j < 4; ++j) a.vrow[j] = static_cast<uint32_t>(4 * g + j) * a.ldw4 + static_cast<uint32_t>(colc) * 4u;
  a.n_it = round_up(Ld.N, 16) >> 4;
  return a;
}
template <int Q>
__device__ __forceinline__ void bwd_wld(const BwdAddr& a, int it, typename BwdRing<Q>::vecq* dst) {
  typedef typename BwdRing<Q>::vecq vecq;
  it = it < a.n_it ? it : a.n_it - 1;                               // scalar
  const uint32_t so = static_cast<uint32_t>(it) * 16u * a.ldw4;    // scalar
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t o = a.vrow[j] + so;
    o = o < a.vlast ? o : a.vlast;                                  // rows past N re-read row N-1 (A is 0 there)
    dst[j] = *(const DCTR_GLOBAL vecq*)(a.wbase + (o & a.wmask));
  }
}
template <int Q>
__device__ __forceinline__ void bwd_fill(BwdRing<Q>& R, const LayerDev& Ld, int gb, int g, int c, uint32_t wmask) {
  const BwdAddr a = bwd_addr<Q>(Ld, gb, g, c, wmask);
#pragma unroll
  for (int d = 0; d < BwdCfg<Q>::PD - 1; ++d) bwd_wld<Q>(a, d, R.r[d]);
}
template <int Q>
__device__ __forceinline__ void bwd_run(BwdRing<Q>& R, const LayerDev& Ld, const float* din, int rs, int gb, int g, int c,
                                        f32x4* acc, uint32_t wmask) {
  constexpr int PD = BwdCfg<Q>::PD;
  const BwdAddr a = bwd_addr<Q>(Ld, gb, g, c, wmask);
  const float* ap = din + c * rs + 4 * g;
  const int n_it = a.n_it;
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n_grp = n_it / PD, rem = n_it - n_grp * PD;
  f32x4 a_nxt = *reinterpret_cast<const f32x4*>(ap);
  for (int gi = 0; gi < n_grp; ++gi) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int it = gi * PD + d;
      bwd_wld<Q>(a, it + PD - 1, R.r[(d + PD - 1) % PD]);
      const f32x4 a4 = a_nxt;
      const int itn = it + 1 < n_it ? it + 1 : it;
      a_nxt = *reinterpret_cast<const f32x4*>(ap + (itn << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = mfma16(a4[j], R.r[d][j][q], acc[q]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < PD - 1; ++d) {
    if (d < rem) {
      const int it = n_grp * PD + d;
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + (it << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = mfma16(a4[j], R.r[d][j][q], acc[q]);
    }
  }
}
// relu mask source of the layer below (its saved output), rows 4 g .. 4 g + 3 at the lane's columns: from its tile in LDS
// when the forward left it there (`hl`, row stride rs_h: the fused train kernel), else from its copy in global memory
template <int Q>
__device__ __forceinline__ void bwd_mask(const MlpArgs& A, int l, int gb, int b0, int g, int c, const float* hl, int rs_h,
                                         typename BwdRing<Q>::vecq* hpre) {
  typedef typename BwdRing<Q>::vecq vecq;
  const LayerDev& Lp = A.L[l > 0 ? l - 1 : 0];   // (l == 0: only its buffer is borrowed as a valid address)
  const int col0 = 16 * Q * gb + Q * c;
  if (hl) {
    const int cc = col0 < rs_h - kPad - Q ? col0 : rs_h - kPad - Q;     // (the tile is rs_h - kPad columns wide)
#pragma unroll
    for (int r = 0; r < 4; ++r) hpre[r] = *reinterpret_cast<const vecq*>(hl + (4 * g + r) * rs_h + cc);
    return;
  }
  const int cc = col0 < Lp.ldh - Q ? col0 : Lp.ldh - Q;
  const int64_t blast = A.B - 1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t b = (b0 + 4 * g + r) < blast ? (b0 + 4 * g + r) : blast;
    hpre[r] = *(const DCTR_GLOBAL vecq*)(Lp.h + b * Lp.ldh + cc);
  }
}
template <int Q>
__device__ __forceinline__ void bwd_epilogue(const MlpArgs& A, int l, float* dout, int rs, int gb, int b0, int g, int c,
                                             const f32x4* acc, const typename BwdRing<Q>::vecq* hpre) {
  typedef typename BwdRing<Q>::vecq vecq;
  const LayerDev& Ld = A.L[l];
  const LayerDev& Lp = A.L[l > 0 ? l - 1 : 0];
  const int Kp = round_up(Ld.K, 16);
  const int col0 = 16 * Q * gb + Q * c;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * g + r;
    const int64_t b = b0 + row;
    vecq v;
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = acc[q][r];
    if (l > 0) {
      if (col0 < Kp) {    // Lp.N == Ld.K
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          float o = v[q];
          if (Lp.relu && !(hpre[r][q] > 0.f)) o = 0.f;
          if (col0 + q >= Lp.N || b >= A.B) o = 0.f;
          v[q] = o;
        }
        *reinterpret_cast<vecq*>(dout + row * rs + col0) = v;     // (its copy in Lp.dh: tile_store behind the barrier)
      }
    } else if (b < A.B) {
      if (col0 + Q - 1 < A.ldgx) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (col0 + q >= Ld.K) v[q] = 0.f;
        if (A.sync) stg_wt(A.gx + b * A.ldgx + col0, v);      // (the update waits for a signal, not for this kernel's end)
        else *(DCTR_GLOBAL vecq*)(A.gx + b * A.ldgx + col0) = v;
      } else {
        for (int q = 0; q < Q; ++q)
          if (col0 + q < A.ldgx) stg_f32(A.gx + b * A.ldgx + col0 + q, col0 + q < Ld.K ? v[q] : 0.f, A.sync != nullptr);
      }
    }
  }
}
// ONE column-group width per launch (two ring types alive across the layer loop spilled): Q = 2 (32-column groups) when
// every layer has at most 256 inputs -- 8 groups keep all 8 waves busy -- else Q = 4 (64-column groups, one pass up to 512)
template <int Q>
__device__ __forceinline__ int bwd_groups(const LayerDev& Ld) { return (Ld.K + 16 * Q - 1) / (16 * Q); }

// hb0 / hb1 (nullable, row stride rs_h): the forward's ping / pong activation tiles when they are still in LDS (the fused
// train kernel with both LDS images): layer j's output then sits in (j & 1 ? hb1 : hb0) for j >= n_layers - 2
template <int Q>
__device__ __forceinline__ void mlp_bwd_fast_q(const MlpArgs& A, float* smem, const float* g_lds, const float* hb0,
                                               const float* hb1, int rs_h FT_ARG) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kTM;
  const int rs = A.rsd;
  float* d0 = smem;
  float* d1 = d0 + kTM * rs;
  const int top = A.n_layers - 1;
  auto h_lds = [&](int j) -> const float* {
    return (hb0 && j >= 0 && j >= top - 1) ? ((j & 1) ? hb1 : hb0) : nullptr;
  };
  const float* htop = h_lds(top);
  FT((0) & 15);
  BwdRing<Q> R;
  {
    const LayerDev& Lt = A.L[top];
    const int Np = round_up(Lt.N, 16);
    const int n_e = kTM * Np;
    const int64_t blast = A.B - 1;
    for (int e0 = 0; e0 < n_e; e0 += 4 * kT) {
      float gv[4], hv[4], wv4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int e = e0 + i * kT + tid;
        e = e < n_e ? e : n_e - 1;
        const int r = e / Np, n = e - r * Np;
        const int nn = n < Lt.N ? n : Lt.N - 1;
        const int64_t b = (b0 + r) < blast ? (b0 + r) : blast;
        if (A.w_out) {
          gv[i] = g_lds ? g_lds[r] : ldg_f32(A.g + b);
          wv4[i] = ldg_f32(A.w_out + nn);
        } else {
          gv[i] = ldg_f32(A.g + b * A.ldg + nn);
          wv4[i] = 1.f;
        }
        hv[i] = htop ? htop[r * rs_h + nn] : ldg_f32(Lt.h + b * Lt.ldh + nn);
      }
      // the top layer's first weights: requested behind the (few) loads of its incoming gradient -- loads return in
      // order, the ring in front would make the staging wait for all of it
      if (e0 == 0 && (top > 0 || A.gx) && wv < bwd_groups<Q>(A.L[top])) bwd_fill<Q>(R, A.L[top], wv, g, c, A.wmask);
#pragma unroll
      for (int i = 0; i < 4; ++i) gv[i] = A.w_out ? gv[i] * wv4[i] : gv[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = e0 + i * kT + tid;
        if (e < n_e) {
          const int r = e / Np, n = e - r * Np;
          const int64_t b = b0 + r;
          float v = 0.f;
          if (b < A.B && n < Lt.N) {
            v = gv[i];
            if (Lt.relu) v = hv[i] > 0.f ? v : 0.f;
          }
          d0[r * rs + n] = v;
        }
      }
    }
  }
  __syncthreads();
  FT((1) & 15);
  if (A.L[top].dh) tile_store(d0, rs, A.L[top].dh, A.L[top].ldh, A.L[top].N, b0, A.B, A.wt != 0);
  float* din = d0;
  float* dout = d1;
  for (int l = top; l >= 0; --l) {
    const LayerDev& Ld = A.L[l];
    if (l > 0 || A.gx) {
      const int ngroups = bwd_groups<Q>(Ld);
      const bool below = l > 0 && (l - 1 > 0 || A.gx);         // a layer below whose weights can be requested early
      const float* hl = h_lds(l - 1);
      for (int gb = wv; gb < ngroups; gb += kWaves) {
        f32x4 acc[Q];
        typename BwdRing<Q>::vecq msk[4];
        bwd_mask<Q>(A, l, gb, b0, g, c, hl, rs_h, msk);
        bwd_run<Q>(R, Ld, din, rs, gb, g, c, acc, A.wmask);
        bwd_epilogue<Q>(A, l, dout, rs, gb, b0, g, c, acc, msk);
        __builtin_amdgcn_sched_barrier(0);
        // what this wave multiplies next, requested behind the epilogue's stores (the memory pipeline is in order: in
        // front of them the burst delays the stores by microseconds) and in flight across the barrier
        if (gb + kWaves < ngroups) bwd_fill<Q>(R, Ld, gb + kWaves, g, c, A.wmask);
        else if (below && wv < bwd_groups<Q>(A.L[l - 1])) bwd_fill<Q>(R, A.L[l - 1], wv, g, c, A.wmask);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (a wave without a group in this layer still opens the next one)
      if (wv >= ngroups && below && wv < bwd_groups<Q>(A.L[l - 1])) bwd_fill<Q>(R, A.L[l - 1], wv, g, c, A.wmask);
    }
    FT((2 + 2 * (top - l)) & 15);
    __syncthreads();
    FT((3 + 2 * (top - l)) & 15);
    // d loss / d pre-activation of the layer below, complete in `dout`: its copy for the weight-gradient kernel
    if (l > 0 && A.L[l - 1].dh) tile_store(dout, rs, A.L[l - 1].dh, A.L[l - 1].ldh, A.L[l - 1].N, b0, A.B, A.wt != 0);
    float* t = din;
    din = dout;
    dout = t;
  }
  FT((15) & 15);
}

__device__ __forceinline__ void mlp_bwd_fast(const MlpArgs& A, float* smem, const float* g_lds, const float* hb0,
                                             const float* hb1, int rs_h FT_ARG) {
  if (A.fast == 2) mlp_bwd_fast_q<2>(A, smem, g_lds, hb0, hb1, rs_h FT_PASS);
  else mlp_bwd_fast_q<4>(A, smem, g_lds, hb0, hb1, rs_h FT_PASS);
}

__global__ __launch_bounds__(kT) void k_mlp_bwd_data(MlpArgs A) {
  extern __shared__ __align__(16) float smem[];
  FT_DECL;
  if (A.fast) {
    mlp_bwd_fast(A, smem, nullptr, nullptr, nullptr, 0 FT_PASS);
    FT_FLUSH(A.trace);
  } else {
    mlp_bwd_body(A, smem, nullptr, nullptr, 0, A.trace);
  }
}

// ------------------------------------------------------------------------------------------------------------
// CrossNet, matrix form: backward-data of a 16-sample row tile through all layers
// ------------------------------------------------------------------------------------------------------------
// With g = d loss / d x_{l+1}:   a_l = g (.) x_0  (= d loss / d u_l: what the weight-gradient kernel consumes, stored in
// the layer's `dh`, over the u_l the forward parked there);   d loss / d x_l = g + a_l W_l;   and x_0 collects
// sum_l g (.) u_l on the side (kept in registers: a thread owns the same elements of the tile in every layer).
// LDS: x_0 | g (ping) | g (pong) | a_l, [16][rs] floats each.
__global__ __launch_bounds__(kT) void k_cross_mat_bwd(MlpArgs A) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kTM;
  const int rs = A.rsd;
  const int W = A.L[0].K;
  const int Wp = round_up(W, 16);
  float* x0s = smem;
  float* gin = x0s + kTM * rs;
  float* gout = gin + kTM * rs;
  float* as = gout + kTM * rs;
  constexpr int kPer = kTM * kKC / kT;      // elements of the tile a thread owns (W <= kKC; unrolled: registers)
  float side[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) side[k] = 0.f;
  const int n_el = kTM * Wp;
  for (int e = tid; e < n_el; e += kT) {
    const int r = e / Wp, n = e - r * Wp;
    const int64_t b = b0 + r;
    const bool ok = b < A.B && n < W;
    x0s[r * rs + n] = ok ? ldg_f32(A.x + b * A.ldx + n) : 0.f;
    gin[r * rs + n] = ok ? ldg_f32(A.g + b * A.ldg + n) : 0.f;
  }
  __syncthreads();
  for (int l = A.n_layers - 1; l >= 0; --l) {
    const LayerDev& Ld = A.L[l];
    // a_l = g (.) x_0 -> LDS (the A operand) and the layer's dh (over u_l, which feeds the side term first)
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int e = tid + k * kT;
      if (e < n_el) {
        const int r = e / Wp, n = e - r * Wp;
        const int64_t b = b0 + r;
        const float gv = gin[r * rs + n];
        const float av = gv * x0s[r * rs + n];
        as[r * rs + n] = av;
        if (b < A.B && n < W) {
          float* up = Ld.dh + b * Ld.ldh + n;
          side[k] += gv * ldg_f32(up);
          stg_f32(up, av);
        }
      }
    }
    __syncthreads();
    // d loss / d x_l = g + a_l W_l : 64-column groups over the waves, reduction over n (rows of W_l)
    const int ngroups = (W + 63) >> 6;
    for (int gb = wv; gb < ngroups; gb += kWaves) {
      const int col0 = 64 * gb + 4 * c;
      const int colc = col0 < Ld.ldw ? col0 : 0;
      f32x4 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* ap = as + c * rs + 4 * g;
      const int n_it = Wp >> 4;
      const DCTR_GLOBAL char* wbase = (const DCTR_GLOBAL char*)Ld.W;
      const uint32_t ldw4 = static_cast<uint32_t>(Ld.ldw) * 4u;
      const uint32_t vlast = static_cast<uint32_t>(Ld.N - 1) * ldw4 + static_cast<uint32_t>(colc) * 4u;
      uint32_t vrow[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) vrow[j] = static_cast<uint32_t>(4 * g + j) * ldw4 + static_cast<uint32_t>(colc) * 4u;
      auto wld = [&](int it, f32x4* dst) {
        it = it < n_it ? it : n_it - 1;
        const uint32_t so = static_cast<uint32_t>(it) * 16u * ldw4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t o = vrow[j] + so;
          o = o < vlast ? o : vlast;                                  // rows past N re-read row N-1 (A is 0 there)
          dst[j] = *(const DCTR_GLOBAL f32x4*)(wbase + o);
        }
      };
      constexpr int PD = 5;
      f32x4 ring[PD][4];
#pragma unroll
      for (int d = 0; d < PD - 1; ++d) wld(d, ring[d]);
      const int n_grp = n_it / PD, rem = n_it - n_grp * PD;
      f32x4 a_nxt = *reinterpret_cast<const f32x4*>(ap);
      for (int gi = 0; gi < n_grp; ++gi) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          const int it = gi * PD + d;
          wld(it + PD - 1, ring[(d + PD - 1) % PD]);
          const f32x4 a4 = a_nxt;
          const int itn = it + 1 < n_it ? it + 1 : it;
          a_nxt = *reinterpret_cast<const f32x4*>(ap + (itn << 4));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma16(a4[j], ring[d][j][q], acc[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int d = 0; d < PD - 1; ++d) {
        if (d < rem) {
          const int it = n_grp * PD + d;
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + (it << 4));
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma16(a4[j], ring[d][j][q], acc[q]);
        }
      }
      if (col0 < Wp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * g + r;
          f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
          const f32x4 gprev = *reinterpret_cast<const f32x4*>(gin + row * rs + col0);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = (col0 + q < W) ? v[q] + gprev[q] : 0.f;
          *reinterpret_cast<f32x4*>(gout + row * rs + col0) = v;
        }
      }
    }
    __syncthreads();
    float* t = gin;
    gin = gout;
    gout = t;
  }
  // d loss / d x_0 = what came down the chain + the side term
  if (A.gx) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int e = tid + k * kT;
      if (e < n_el) {
        const int r = e / Wp, n = e - r * Wp;
        const int64_t b = b0 + r;
        if (b < A.B && n < A.ldgx) stg_f32(A.gx + b * A.ldgx + n, n < W ? gin[r * rs + n] + side[k] : 0.f);
      }
    }
    // columns [Wp, ldgx) of gx (padding) are written as zeros as well
    for (int e = tid; e < kTM * 4; e += kT) {
      const int r = e >> 2, n = Wp + (e & 3);
      const int64_t b = b0 + r;
      if (b < A.B && n < A.ldgx) stg_f32(A.gx + b * A.ldgx + n, 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// CrossNetMix (DCN-Mix): mixture of low-rank experts per cross layer, as three dense layers each (see dctr.h)
// ------------------------------------------------------------------------------------------------------------
// one dense layer of a 16-sample tile whose input sits in LDS: bias-initialised accumulators, the forward's MFMA
// loops, then epi(row, n, value) per output element
template <typename Epi>
__device__ __forceinline__ void mix_layer(const LayerDev& Ld, const float* in, int rs, int klen, int wv, int g, int c,
                                          Epi epi) {
  const int ntile = (Ld.N + 15) >> 4;
  for (int tbase = 0; tbase < ntile; tbase += kWaves * kNTMax) {
    const int tile0 = tbase + wv;
    int nt = 0;
#pragma unroll
    for (int t = 0; t < kNTMax; ++t) nt += (tile0 + t * kWaves < ntile) ? 1 : 0;
    f32x4 acc[kNTMax];
#pragma unroll
    for (int t = 0; t < kNTMax; ++t) {
      const int n = (tile0 + t * kWaves) * 16 + c;
      const float bv = (t < nt && n < Ld.N && Ld.bias) ? ldg_f32(Ld.bias + n) : 0.f;
      acc[t] = f32x4{bv, bv, bv, bv};
    }
    fwd_dispatch(nt, in, rs, 0, klen, Ld, tile0, acc, g, c, []() {});
#pragma unroll
    for (int t = 0; t < kNTMax; ++t) {
      if (t < nt) {
        const int n = (tile0 + t * kWaves) * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(4 * g + r, n, acc[t][r]);
      }
    }
  }
}

__global__ __launch_bounds__(kT) void k_cross_mix_fwd(MlpArgs A, int E, int R) {
  extern __shared__ __align__(16) float smem[];
  __shared__ float sc[kTM][8];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kTM;
  const int W = A.L[0].K, Wp = round_up(W, 16);
  const int ER = E * R, N1 = ER + E;
  const int rsw = Wp + 4, rsv = round_up(N1, 16) + 4;
  float* xs = smem;                 // x_0
  float* xa = xs + kTM * rsw;       // x_l / x_{l+1}: ping
  float* xb = xa + kTM * rsw;       //                pong
  float* v1 = xb + kTM * rsw;       // [v1 | scores]
  float* v2 = v1 + kTM * rsv;       // s (.) v2
  for (int e = tid; e < kTM * Wp; e += kT) {
    const int r = e / Wp, n = e - r * Wp;
    const int64_t b = b0 + r;
    xs[r * rsw + n] = (b < A.B && n < W) ? ldg_f32(A.x + b * A.ldx + n) : 0.f;
  }
  __syncthreads();
  const float* xl = xs;
  float* xn = xa;
  const int n_cross = A.n_layers / 3;
  for (int lc = 0; lc < n_cross; ++lc) {
    const LayerDev& L1 = A.L[3 * lc];
    const LayerDev& L2 = A.L[3 * lc + 1];
    const LayerDev& L3 = A.L[3 * lc + 2];
    // ---- project to the experts' rank spaces (+ the gating scores as E more output columns)
    mix_layer(L1, xl, rsw, Wp, wv, g, c, [&](int row, int n, float v) {
      // (the scores go to their own array: the next layer reads v1 over round16(E*R) columns and whatever sits
      // beyond E*R must be zero -- its weight loads are pulled back inside the row there)
      if (n >= ER && n < N1) sc[row][n - ER] = v;
      v = (n < ER) ? tanhf(v) : 0.f;
      v1[row * rsv + n] = v;
      if (n < ER && b0 + row < A.B) stg_f32(L1.h + static_cast<int64_t>(b0 + row) * L1.ldh + n, v);
    });
    __syncthreads();
    if (tid < kTM) {                                     // softmax over the E scores of a sample (torch.softmax, dim=1)
      float m = -INFINITY;
      for (int i = 0; i < E; ++i) m = fmaxf(m, sc[tid][i]);
      float ex[8], sum = 0.f;
      for (int i = 0; i < E; ++i) {
        ex[i] = expf(sc[tid][i] - m);
        sum += ex[i];
      }
      for (int i = 0; i < E; ++i) {
        const float si = ex[i] / sum;
        sc[tid][i] = si;
        if (b0 + tid < A.B) stg_f32(L1.h + static_cast<int64_t>(b0 + tid) * L1.ldh + ER + i, si);
      }
    }
    __syncthreads();
    // ---- the experts' r x r maps (one block-diagonal layer); the mixture weights go onto the result
    mix_layer(L2, v1, rsv, round_up(ER, 16), wv, g, c, [&](int row, int n, float v) {
      float t = 0.f, ts = 0.f;
      if (n < ER) {
        t = tanhf(v);
        ts = t * sc[row][n / R];
        if (b0 + row < A.B) {
          stg_f32(L2.dh + static_cast<int64_t>(b0 + row) * L2.ldh + n, t);     // parked for the backward
          stg_f32(L2.h + static_cast<int64_t>(b0 + row) * L2.ldh + n, ts);
        }
      }
      v2[row * rsv + n] = ts;
    });
    __syncthreads();
    // ---- back to R^W, bias, cross with x_0, residual
    mix_layer(L3, v2, rsv, round_up(ER, 16), wv, g, c, [&](int row, int n, float u) {
      float v = 0.f;
      if (n < W) {
        v = xs[row * rsw + n] * u + xl[row * rsw + n];
        if (b0 + row < A.B) {
          stg_f32(L3.dh + static_cast<int64_t>(b0 + row) * L3.ldh + n, u);     // parked for the backward
          stg_f32(L3.h + static_cast<int64_t>(b0 + row) * L3.ldh + n, v);
        }
      }
      xn[row * rsw + n] = v;
    });
    __syncthreads();
    xl = xn;
    xn = (xn == xa) ? xb : xa;
  }
}

// out[row, k] = sum_{n < N} a[row, n] W[n, k] for k in [0, ncols): the backward-data product of one dense layer on a
// 16-sample tile (a in LDS, zero beyond N); 64-column groups over the waves; epi(row, col0, f32x4) per 4 columns.
template <typename Epi>
__device__ __forceinline__ void mix_bwd_product(const float* as, int rs, int Nred, const LayerDev& Ld, int ncols,
                                                int wv, int g, int c, Epi epi) {
  const int ngroups = (ncols + 63) >> 6;
  const int ncp = round_up(ncols, 16);
  for (int gb = wv; gb < ngroups; gb += kWaves) {
    const int col0 = 64 * gb + 4 * c;
    const int colc = col0 < Ld.ldw ? col0 : 0;
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = as + c * rs + 4 * g;
    const int n_it = Nred >> 4;
    const DCTR_GLOBAL char* wbase = (const DCTR_GLOBAL char*)Ld.W;
    const uint32_t ldw4 = static_cast<uint32_t>(Ld.ldw) * 4u;
    const uint32_t vlast = static_cast<uint32_t>(Ld.N - 1) * ldw4 + static_cast<uint32_t>(colc) * 4u;
    uint32_t vrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vrow[j] = static_cast<uint32_t>(4 * g + j) * ldw4 + static_cast<uint32_t>(colc) * 4u;
    auto wld = [&](int it, f32x4* dst) {
      it = it < n_it ? it : n_it - 1;
      const uint32_t so = static_cast<uint32_t>(it) * 16u * ldw4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t o = vrow[j] + so;
        o = o < vlast ? o : vlast;                                  // rows past N re-read row N-1 (a is 0 there)
        dst[j] = *(const DCTR_GLOBAL f32x4*)(wbase + o);
      }
    };
    constexpr int PD = 5;
    f32x4 ring[PD][4];
#pragma unroll
    for (int d = 0; d < PD - 1; ++d) wld(d, ring[d]);
    const int n_grp = n_it / PD, rem = n_it - n_grp * PD;
    f32x4 a_nxt = *reinterpret_cast<const f32x4*>(ap);
    for (int gi = 0; gi < n_grp; ++gi) {
#pragma unroll
      for (int d = 0; d < PD; ++d) {
        const int it = gi * PD + d;
        wld(it + PD - 1, ring[(d + PD - 1) % PD]);
        const f32x4 a4 = a_nxt;
        const int itn = it + 1 < n_it ? it + 1 : it;
        a_nxt = *reinterpret_cast<const f32x4*>(ap + (itn << 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = mfma16(a4[j], ring[d][j][q], acc[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int d = 0; d < PD - 1; ++d) {
      if (d < rem) {
        const int it = n_grp * PD + d;
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + (it << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = mfma16(a4[j], ring[d][j][q], acc[q]);
      }
    }
    if (col0 < ncp) {
#pragma unroll
      for (int r = 0; r < 4; ++r) epi(4 * g + r, col0, f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]});
    }
  }
}

// backward-data of a 16-sample tile through all cross layers.  Per layer, with g = d loss / d x_{l+1}:
//   a3 = g (.) x_0  (-> dh of layer 3l+2, over the parked u; x_0 collects g (.) u on the side)
//   p3 = a3 W3;  d s_e = sum_r p3[e, r] v2[e, r];  a2 = p3 (.) s_e (.) (1 - v2^2)  (-> dh of layer 3l+1, over the parked v2)
//   softmax backward: d score_e = s_e (d s_e - sum_e' s_e' d s_e')
//   a1 = [ (a2 W2) (.) (1 - v1^2) | d score ]  (-> dh of layer 3l);   d loss / d x_l = g + a1 W1
__global__ __launch_bounds__(kT) void k_cross_mix_bwd(MlpArgs A, int E, int R) {
  extern __shared__ __align__(16) float smem[];
  __shared__ float dsc[kTM][8];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * kTM;
  const int W = A.L[0].K, Wp = round_up(W, 16);
  const int ER = E * R, N1 = ER + E;
  const int ERp = round_up(ER, 16), N1p = round_up(N1, 16);
  const int rsw = round_up(W, 64) + 4, rsv = round_up(N1, 64) + 4;
  float* x0s = smem;
  float* gin = x0s + kTM * rsw;
  float* gout = gin + kTM * rsw;
  float* a3 = gout + kTM * rsw;
  float* p3 = a3 + kTM * rsw;       // [16][rsv]: p3, then a2 in place
  float* a1 = p3 + kTM * rsv;       // [16][rsv]
  constexpr int kPer = kTM * kKC / kT;
  float side[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) side[k] = 0.f;
  const int n_el = kTM * Wp;
  for (int e = tid; e < n_el; e += kT) {
    const int r = e / Wp, n = e - r * Wp;
    const int64_t b = b0 + r;
    const bool ok = b < A.B && n < W;
    x0s[r * rsw + n] = ok ? ldg_f32(A.x + b * A.ldx + n) : 0.f;
    gin[r * rsw + n] = ok ? ldg_f32(A.g + b * A.ldg + n) : 0.f;
  }
  for (int e = tid; e < kTM * rsv; e += kT) {
    p3[e] = 0.f;
    a1[e] = 0.f;
  }
  __syncthreads();
  const int n_cross = A.n_layers / 3;
  for (int lc = n_cross - 1; lc >= 0; --lc) {
    const LayerDev& L1 = A.L[3 * lc];
    const LayerDev& L2 = A.L[3 * lc + 1];
    const LayerDev& L3 = A.L[3 * lc + 2];
    // ---- a3 = g (.) x_0; the side term for x_0
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int e = tid + k * kT;
      if (e < n_el) {
        const int r = e / Wp, n = e - r * Wp;
        const int64_t b = b0 + r;
        const float gv = gin[r * rsw + n];
        const float av = gv * x0s[r * rsw + n];
        a3[r * rsw + n] = av;
        if (b < A.B && n < W) {
          float* up = L3.dh + b * L3.ldh + n;
          side[k] += gv * ldg_f32(up);
          stg_f32(up, av);
        }
      }
    }
    __syncthreads();
    // ---- p3 = a3 W3  ([16, W] x [W, ER])
    mix_bwd_product(a3, rsw, Wp, L3, ER, wv, g, c, [&](int row, int col0, f32x4 v) {
#pragma unroll
      for (int q = 0; q < 4; ++q) p3[row * rsv + col0 + q] = (col0 + q < ER) ? v[q] : 0.f;
    });
    __syncthreads();
    // ---- d s_e (a reduction over the expert's R columns), then a2 in place
    if (tid < kTM * E) {
      const int row = tid / E, e = tid - row * E;
      const int64_t b = b0 + row;
      float acc = 0.f;
      if (b < A.B)
        for (int r = 0; r < R; ++r) acc += p3[row * rsv + e * R + r] * ldg_f32(L2.dh + b * L2.ldh + e * R + r);
      dsc[row][e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < kTM * ERp; e += kT) {
      const int row = e / ERp, n = e - row * ERp;
      const int64_t b = b0 + row;
      float av = 0.f;
      if (b < A.B && n < ER) {
        const float t = ldg_f32(L2.dh + b * L2.ldh + n);                     // unscaled v2 (parked by the forward)
        const float s = ldg_f32(L1.h + b * L1.ldh + ER + n / R);
        av = p3[row * rsv + n] * s * (1.f - t * t);
        stg_f32(L2.dh + b * L2.ldh + n, av);
      }
      p3[row * rsv + n] = av;
    }
    if (tid < kTM) {                                                          // softmax backward
      const int64_t b = b0 + tid;
      float dot = 0.f, sv[8];
      for (int i = 0; i < E; ++i) {
        sv[i] = b < A.B ? ldg_f32(L1.h + b * L1.ldh + ER + i) : 0.f;
        dot += sv[i] * dsc[tid][i];
      }
      for (int i = 0; i < E; ++i) {
        const float ds = sv[i] * (dsc[tid][i] - dot);
        a1[tid * rsv + ER + i] = ds;
        if (b < A.B) stg_f32(L1.dh + b * L1.ldh + ER + i, ds);
      }
    }
    __syncthreads();
    // ---- a1[:, :ER] = (a2 W2) (.) (1 - v1^2)
    mix_bwd_product(p3, rsv, ERp, L2, ER, wv, g, c, [&](int row, int col0, f32x4 v) {
      const int64_t b = b0 + row;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = col0 + q;
        float av = 0.f;
        if (b < A.B && n < ER) {
          const float t = ldg_f32(L1.h + b * L1.ldh + n);
          av = v[q] * (1.f - t * t);
          stg_f32(L1.dh + b * L1.ldh + n, av);
        }
        if (n < ER) a1[row * rsv + n] = av;
      }
    });
    __syncthreads();
    // ---- d loss / d x_l = g + a1 W1  ([16, N1] x [N1, W])
    mix_bwd_product(a1, rsv, N1p, L1, W, wv, g, c, [&](int row, int col0, f32x4 v) {
      const f32x4 gprev = *reinterpret_cast<const f32x4*>(gin + row * rsw + col0);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (col0 + q < W) ? v[q] + gprev[q] : 0.f;
      *reinterpret_cast<f32x4*>(gout + row * rsw + col0) = v;
    });
    __syncthreads();
    float* t = gin;
    gin = gout;
    gout = t;
  }
  if (A.gx) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int e = tid + k * kT;
      if (e < n_el) {
        const int r = e / Wp, n = e - r * Wp;
        const int64_t b = b0 + r;
        if (b < A.B && n < A.ldgx) stg_f32(A.gx + b * A.ldgx + n, n < W ? gin[r * rsw + n] + side[k] : 0.f);
      }
    }
    for (int e = tid; e < kTM * 4; e += kT) {
      const int r = e >> 2, n = Wp + (e & 3);
      const int64_t b = b0 + r;
      if (b < A.B && n < A.ldgx) stg_f32(A.gx + b * A.ldgx + n, 0.f);
    }
  }
}

// forward + prediction head + BCE(sum) + backward-data of one row tile in ONE launch (the fused train step): the
// logits never leave the workgroup, d loss / d logit goes to the backward through LDS, and the head's own launch,
// the backward's staging round trip and two kernel boundaries disappear.  Per-workgroup partial sums of the loss and
// of d loss / d bias are reduced in fixed order by k_mlp_reduce.
template <bool GATHER>
__device__ __forceinline__ void mlp_train_body(const MlpArgs& A, const HeadArgs& Hd, int bwd_off, const GatherArgs& G,
                                               float* smem) {
  __shared__ float zl[kTM], gl[kTM], p0s[kTM], p1s[kTM];
  const int tid = threadIdx.x;
  // the head's inputs, requested before the tower runs (they used to cost a round trip between forward and backward);
  // absent logit parts borrow y's address and are dropped by a select
  const int64_t hb = static_cast<int64_t>(blockIdx.x) * kTM + (tid & (kTM - 1));
  const int64_t hbc = hb < A.B ? hb : A.B - 1;
  float h_p0 = 0.f, h_p1 = 0.f;
  if constexpr (!GATHER) {
    h_p0 = ldg_f32((Hd.part0 ? Hd.part0 : Hd.y) + hbc);
    h_p1 = ldg_f32((Hd.part1 ? Hd.part1 : Hd.y) + hbc);
  }
  // (out.bias is a dense parameter: with G.wsync it is requested behind the wait for the weights, inside mlp_fwd_fast)
  const float* bias_p = Hd.bias ? Hd.bias : Hd.y;
  float h_bias = 0.f;
  bool early_bias = true;
  if constexpr (GATHER) early_bias = G.wsync == nullptr;
  if (early_bias) h_bias = ldg_f32(bias_p);
  const float h_y = ldg_f32(Hd.y + hbc);
  FT_DECL;
  FT_DECL_B;
  const float* htop = nullptr;
  if constexpr (GATHER) {
    htop = mlp_fwd_fast<true>(A, smem, zl, G, p0s, p1s, bias_p, h_bias FT_PASS);
  } else {
    htop = A.fast ? mlp_fwd_fast<false>(A, smem, zl, G, p0s, p1s, bias_p, h_bias FT_PASS) : mlp_fwd_body<false>(A, smem, zl);
  }
  __syncthreads();
  if (tid < 64) {
    const int64_t b = hb;
    float li = 0.f, gz = 0.f;
    if (tid < kTM && b < A.B) {
      float z = 0.f;                       // ((linear + fm) + dnn) + bias: the reference's order of additions
      if constexpr (GATHER) {
        z += p0s[tid];
        if (G.want_fm) z += p1s[tid];
      } else {
        if (Hd.part0) z += h_p0;
        if (Hd.part1) z += h_p1;
      }
      z += zl[tid];
      if (Hd.bias) z += h_bias;
      const float p = 1.f / (1.f + expf(-z));                     // at::sigmoid
      const float t = h_y;
      const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
      li = (t - 1.f) * l1p - t * lp;                              // at::binary_cross_entropy
      const float q = (1.f - p) * p;
      gz = ((p - t) / fmaxf(q, 1e-12f)) * q;                      // bce backward (grad 1) x sigmoid backward
      stg_f32(Hd.y_pred + b, p);
      stg_f32(Hd.g_logit + b, gz, A.sync != nullptr || A.wt != 0);
    }
    if (tid < kTM) gl[tid] = gz;
    li = group_sum<16>(li);
    float gs = group_sum<16>(gz);
    if (tid == 0) {
      stg_f32(Hd.part_loss + blockIdx.x, li, A.wt != 0);
      stg_f32(Hd.part_gbias + blockIdx.x, gs, A.wt != 0);
    }
  }
  __threadfence_block();   // the saved activations written above are re-read as relu masks below
  __syncthreads();
  // bwd_off > 0: the backward's two gradient tiles lie BEHIND the forward's LDS image, so the top layer's output tile is
  // still there and its relu mask needs no global round trip; 0: they alias it (towers too wide for both images)
  if (GATHER || A.fast) {
    float* hb0 = smem + kTM * A.rsx;
    mlp_bwd_fast(A, smem + bwd_off, gl, bwd_off > 0 ? hb0 : nullptr, bwd_off > 0 ? hb0 + kTM * A.rsh : nullptr, A.rsh
                 FT_PASS_B);
    FT_FLUSH(A.trace);         // (diag build: the forward's stamps to region 0, the backward's to region 1)
    FT_FLUSH_B(A.trace);
  } else {
    // (diag build: the backward stamps region 1.  Through a local pointer -- writing to A would put the whole argument
    // struct in scratch)
    mlp_bwd_body(A, smem + bwd_off, gl, bwd_off > 0 ? htop : nullptr, A.rsh, A.trace ? A.trace + 16ull * 4096 : nullptr);
    // (the generic body's gx stores are plain: an agent-scope release writes this XCD's dirty lines back first)
    if (A.sync) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
  if (A.sync) step_signal(A.sync, DCTR_SYNC_TOWER);
  if constexpr (GATHER) {
    // one more tower launch done (the next one waits until as many weight steps have happened): the launch's last workgroup
    // advances sync[DCTR_SYNC_T_GEN].  Plain ordering suffices -- the reader is a LATER launch on this queue.
    if (G.wsync) {
      __syncthreads();
      if (threadIdx.x == 0) {
        const int32_t n = static_cast<int32_t>(gridDim.x);
        if (__hip_atomic_fetch_add(G.wsync + DCTR_SYNC_T_ARR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1) {
          __hip_atomic_store(G.wsync + DCTR_SYNC_T_ARR, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(G.wsync + DCTR_SYNC_T_GEN, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

__global__ __launch_bounds__(kT) void k_mlp_train(MlpArgs A, HeadArgs Hd, int bwd_off) {
  step_priority();
  extern __shared__ __align__(16) float smem[];
  mlp_train_body<false>(A, Hd, bwd_off, GatherArgs{}, smem);
}

// the same launch with the embedding lookup as its input stage (GatherArgs): gather + wide + FM + tower + head + BCE +
// backward-data of a 16-sample tile; fast-path towers only
__global__ __launch_bounds__(kT) void k_embed_tower_train(MlpArgs A, HeadArgs Hd, int bwd_off, GatherArgs G) {
  step_priority();
  extern __shared__ __align__(16) float smem[];
  mlp_train_body<true>(A, Hd, bwd_off, G, smem);
}

// ------------------------------------------------------------------------------------------------------------
// weight gradients
// ------------------------------------------------------------------------------------------------------------
struct WgradArgs {
  LayerDev L[kMaxL];
  int n_layers, B, S, bs;      // S batch splits of bs rows each (bs % 8 == 0)
  int P, pbs;                  // projection (d w_out) workgroups, pbs rows each; they fill the slabs j, j + P, ... < S
  int blk0[kMaxL + 2];         // first block of layer l; [n_layers] = projection blocks; [n_layers + 1] = end
  int64_t off_w[kMaxL], off_b[kMaxL], off_o, slab;  // float offsets inside one partial slab; slab = its size
  const float* x;
  int64_t ldx;
  const float* w_out;
  const float* g;  // [B] when w_out
  float* part;     // [S][slab]
  unsigned long long* trace;
  // ---- the reduction folded into this launch (dctr_mlp_train_wgrad_sync, round 6); cnt == NULL: off (k_mlp_reduce follows).
  // A tile's S workgroups store their partials WRITE-THROUGH, drain them and take a ticket of the tile's counter; the one that
  // draws the last ticket sums the S slabs in slab order (k_mlp_reduce's arithmetic, element for element), stores the gradient
  // and steps the parameter behind it.  The projection's P workgroups do the same for d w_out and -- in the same last
  // arriver -- the head's partial sums (loss, d bias).  Every reducer stores the parameters write-through, drains, and counts
  // itself done; the launch's last reducer advances sync[DCTR_SYNC_W_GEN]: a tower launch that waits for it
  // (dctr_embed_tower_train_step_sync) may then read every dense parameter.
  int32_t* cnt;        // [n_red] arrivals per reducer (zero at rest)
  int32_t* sync;       // the model's sync block (include/dctr.h)
  int n_red;           // GEMM tiles (+ 1 with a projection)
  float* gW[kMaxL];    // gradient tensors the sums go to
  float* gb[kMaxL];    // nullable
  float* g_wo;
  const float* head_loss;   // [n_head] nullable (fused train step)
  const float* head_gbias;
  int n_head;
  float* loss;
  float* g_bias;
  DenseStepDev step;
};

// optimizer step on four consecutive elements whose gradients `acc` were just finished: k_mlp_reduce's arithmetic, expression
// for expression (the two paths must agree bit for bit: tests/test_gpu_step_engine.py)
__device__ __forceinline__ void fold_step4(const DenseStepDev& S, float* d, const f32x4 acc, const f32x4 w, const f32x4 st) {
  stg_wt(d, acc);
  if (S.kind < 0) return;
  const int64_t k = d - S.grad_base;
  f32x4 wn, sn4;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float g = acc[c];
    if (S.kind == DCTR_UPD_ADAGRAD) {
      const float sn = adagrad_sum(st[c], g);
      sn4[c] = sn;
      wn[c] = adagrad_param(w[c], g, sn, S.lr, S.eps);
    } else {
      sn4[c] = 0.f;
      wn[c] = sgd_param(w[c], g, S.lr);
    }
  }
  if (S.kind == DCTR_UPD_ADAGRAD) stg_wt(S.state_base + k, sn4);
  stg_wt(S.param_base + k, wn);
}
__device__ __forceinline__ void fold_step1(const DenseStepDev& S, float* d, const float g) {
  stg_wt(d, g);
  if (S.kind < 0) return;
  const int64_t k = d - S.grad_base;
  const float w = __hip_atomic_load(S.param_base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (sc1: past the L1)
  if (S.kind == DCTR_UPD_ADAGRAD) {
    const float sn = adagrad_sum(__hip_atomic_load(S.state_base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), g);
    stg_wt(S.state_base + k, sn);
    stg_wt(S.param_base + k, adagrad_param(w, g, sn, S.lr, S.eps));
  } else {
    stg_wt(S.param_base + k, sgd_param(w, g, S.lr));
  }
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// a 16-byte load that bypasses this CU's L1 (sc1): the partial slabs were stored write-through by other workgroups
__device__ __forceinline__ f32x4 ld_slab4(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}
__device__ __forceinline__ float ld_slab1(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 16));
}
// every thread of the workgroup, behind its last write-through store: the stores have arrived; thread 0 takes a ticket of
// `cnt` (relaxed, agent scope: the atomic executes beyond the XCD's L2) -> true in the workgroup that drew the last of `n`
// (which hands the counter back zeroed).  `flag`: one LDS word.
__device__ __forceinline__ bool fold_arrive(int32_t* cnt, int n, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == n - 1) ? 1 : 0;
    if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
// a reducer is done (its parameter stores drained): the launch's last one advances the weights' generation
__device__ __forceinline__ void fold_done(int32_t* sync, int n_red) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(sync + DCTR_SYNC_W_ARR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == n_red - 1) {
      __hip_atomic_store(sync + DCTR_SYNC_W_ARR, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(sync + DCTR_SYNC_W_GEN, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One workgroup = one 64x64 tile of one layer's dW over one batch slice; its four waves take the slice's rows in four
// contiguous quarters (multiples of 8 rows) and are summed through LDS in wave order.
//  * Lane (p, jl) owns the COLUMN PAIR (2 jl, 2 jl + 1) of the tile in both operands and the row parity p of a
//    k-step: one global_load_dwordx2 per operand per k-step (two rows x 256 contiguous bytes), i.e. 2 loads per 4
//    MFMAs.  The four accumulators are the (even | odd n) x (even | odd k) sub-tiles; the LDS park un-permutes them.
//  * Loads and MFMAs are interleaved one for one (round 3: the previous loop issued 16 dword loads, then 16 MFMAs --
//    the matrix pipe idled ~340 cycles per group behind the load block's issue time: 40 ns per MFMA against the
//    pipe's 30).  A load issued between two MFMAs hides under the 64 cycles the first one occupies the pipe.
//  * The ring keeps PD-1 groups of U k-steps in flight (~1.4 us of MFMA time: an L2 / MALL round trip under load).
//  * No predicated load anywhere: column pairs past the row are pulled back inside it (they only reach output rows /
//    columns that are dropped or zeroed at the store), the ragged last rows of a slice are loaded from a clamped
//    row and masked to zero afterwards.
__global__ __launch_bounds__(kTW, 2) void k_mlp_wgrad(WgradArgs A) {
  step_priority<DCTR_WGRAD_PRIORITY>();
  extern __shared__ __align__(16) float wsm[];
  float* red = wsm;                    // [4 waves][64 n][64 k] (64 KB)
  float* redb = wsm + 4 * 4096;        // [4][64] bias partials
  const int tid = threadIdx.x, lane = tid & 63, p = lane >> 5, jl = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x;
  MLP_TRACE(A.trace, 0);
  int l = 0;
  while (l < A.n_layers && blk >= A.blk0[l + 1]) ++l;
  const int local = blk - A.blk0[l];
  const int s = local % A.S;
  const int rb0 = s * A.bs;
  const int rb1 = (rb0 + A.bs < A.B) ? rb0 + A.bs : A.B;
  float* part = A.part + static_cast<int64_t>(s) * A.slab;
  // the slice's rows in four quarters (multiples of 8 rows: whole k-step groups), one per wave
  const int nslice = rb1 > rb0 ? rb1 - rb0 : 0;
  const int bq = round_up((nslice + 3) / 4, 8);
  const int wb0 = rb0 + wv * bq;
  const int wb1 = (wb0 + bq < rb1) ? wb0 + bq : rb1;
  const int nrow = wb1 > wb0 ? wb1 - wb0 : 0;

  if (l == A.n_layers) {
    // d w_out[n] = sum_b g[b] * h_top[b, n] over the rows [j * pbs, (j + 1) * pbs) of projection workgroup j.  A half-wave
    // reads one row as dwordx4 (128 columns per pass), the 8 half-waves take every 8th row, 8 rows in flight per lane;
    // the 8 partial sums per column are added in a fixed order.  Few, long workgroups on purpose: the launch is sized so
    // that GEMM tiles + projection workgroups fit the 256 CUs in one round (plan_wgrad).
    const LayerDev& Lt = A.L[A.n_layers - 1];
    const int j = local;
    const int pr0 = j * A.pbs, pr1 = (pr0 + A.pbs < A.B) ? pr0 + A.pbs : A.B;
    const int hw = wv * 2 + p;                       // half-wave 0..7
    float* redp = wsm;                               // [8][128]
    for (int n0 = 0; n0 < Lt.N; n0 += 128) {
      const int n = n0 + 4 * jl;
      const int nc = n < Lt.ldh - 4 ? n : Lt.ldh - 4;    // ldh % 4 == 0: the dwordx4 stays inside the row
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int b = pr0 + hw; b < pr1; b += 64) {
        f32x4 hv[8];
        float gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int bb = b + 8 * i < pr1 ? b + 8 * i : pr1 - 1;
          hv[i] = ldg_f4(Lt.h + static_cast<int64_t>(bb) * Lt.ldh + nc);
          gv[i] = ldg_f32(A.g + bb);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float gq = (b + 8 * i < pr1) ? gv[i] : 0.f;
          acc += gq * hv[i];
        }
      }
      __syncthreads();
      *reinterpret_cast<f32x4*>(redp + hw * 128 + 4 * jl) = acc;
      __syncthreads();
      if (tid < 128 && n0 + tid < Lt.N) {
        float t = redp[tid];
#pragma unroll
        for (int h = 1; h < 8; ++h) t += redp[h * 128 + tid];
        for (int sl = j; sl < A.S; sl += A.P)
          stg_f32(A.part + static_cast<int64_t>(sl) * A.slab + A.off_o + n0 + tid, sl == j ? t : 0.f, A.cnt != nullptr);
      }
    }
    if (!A.cnt) return;
    // ---- folded reduction: the last of the P projection workgroups sums d w_out over the S slabs and finishes the head
    int* flag = reinterpret_cast<int*>(wsm + 4 * 4096 + 256);
    if (!fold_arrive(A.cnt + (A.n_red - 1), A.P, flag)) return;
    {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(A.part, 0, static_cast<int>(A.S * A.slab * 4), 0x00020000);
      for (int n = tid; n < Lt.N; n += kTW) {
        float acc = 0.f;
        for (int s0 = 0; s0 < A.S; s0 += 8) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int sj = s0 + q < A.S ? s0 + q : A.S - 1;
            v[q] = ld_slab1(rs, static_cast<uint32_t>((static_cast<int64_t>(sj) * A.slab + A.off_o + n) * 4));
          }
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (s0 + q < A.S) acc += v[q];
        }
        if (A.g_wo) fold_step1(A.step, A.g_wo + n, acc);
      }
      if (A.head_loss) {   // k_mlp_reduce's head workgroup: fixed-order tree over the row tiles' partial sums
        __syncthreads();
        float* hred = wsm;     // [2][4]
        float l = 0.f, gsum = 0.f;
        for (int k = tid; k < A.n_head; k += 256) {
          l += ldg_f32(A.head_loss + k);
          gsum += ldg_f32(A.head_gbias + k);
        }
        l = wave_sum(l);
        gsum = wave_sum(gsum);
        if ((tid & 63) == 0) {
          hred[tid >> 6] = l;
          hred[4 + (tid >> 6)] = gsum;
        }
        __syncthreads();
        if (tid == 0) {
          stg_f32(A.loss, ((hred[0] + hred[1]) + hred[2]) + hred[3]);
          if (A.g_bias) fold_step1(A.step, A.g_bias, ((hred[4] + hred[5]) + hred[6]) + hred[7]);
        }
      }
    }
    fold_done(A.sync, A.n_red);
    return;
  }

  const LayerDev& Ld = A.L[l];
  const float* in = (l == 0) ? A.x : A.L[l - 1].h;
  const int ldi = static_cast<int>((l == 0) ? A.ldx : A.L[l - 1].ldh);
  const int kt = (Ld.K + 63) >> 6;
  const int tile = local / A.S;
  const int m0 = (tile / kt) * 64, k0 = (tile % kt) * 64;
  const bool want_bias = (k0 == 0);

  f32x16 c00, c01, c10, c11;   // (n parity, k parity)
#pragma unroll
  for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = 0.f;
  float sa0 = 0.f, sa1 = 0.f;
  // column pair of this lane, pulled back inside the row (ldh, ldi are multiples of 4)
  const int mc = (m0 + 2 * jl) < (Ld.ldh - 2) ? (m0 + 2 * jl) : (Ld.ldh - 2);
  const int kc = (k0 + 2 * jl) < (ldi - 2) ? (k0 + 2 * jl) : (ldi - 2);
  const DCTR_GLOBAL char* dbase = (const DCTR_GLOBAL char*)Ld.dh;
  const DCTR_GLOBAL char* ibase = (const DCTR_GLOBAL char*)in;
  const uint32_t ldh4 = static_cast<uint32_t>(Ld.ldh) * 4u, ldi4 = static_cast<uint32_t>(ldi) * 4u;
  const uint32_t va = static_cast<uint32_t>(p) * ldh4 + 4u * static_cast<uint32_t>(mc);
  const uint32_t vx = static_cast<uint32_t>(p) * ldi4 + 4u * static_cast<uint32_t>(kc);
  constexpr int U = 4, PD = 3;
  const int n_full = nrow / (2 * U);                  // full groups of U k-steps (2 rows each)
  const uint32_t row0 = static_cast<uint32_t>(wb0);
  auto lda = [&](int gidx, int u) -> f32x2 {          // group index -> k-step u's rows (clamped to the last full group)
    gidx = gidx < n_full ? gidx : n_full - 1;          // scalar
    const uint32_t r = row0 + static_cast<uint32_t>(gidx * 2 * U + 2 * u);
    return *(const DCTR_GLOBAL f32x2*)(dbase + (va + r * ldh4));
  };
  auto ldx = [&](int gidx, int u) -> f32x2 {
    gidx = gidx < n_full ? gidx : n_full - 1;
    const uint32_t r = row0 + static_cast<uint32_t>(gidx * 2 * U + 2 * u);
    return *(const DCTR_GLOBAL f32x2*)(ibase + (vx + r * ldi4));
  };
  if (n_full > 0) {
    f32x2 ra[PD][U], rx[PD][U];
#pragma unroll
    for (int d = 0; d < PD - 1; ++d)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ra[d][u] = lda(d, u);
        rx[d][u] = ldx(d, u);
      }
    // whole rounds of PD groups (no exit inside: the compiler's vmcnt bookkeeping stays exact -- with a break per
    // group it waited for all but the last 6 loads, i.e. a prefetch distance of 3 k-steps instead of 12), then the
    // < PD left-over groups, whose operands the last round (or the prologue) has already requested
    const int n_round = n_full / PD, rem = n_full - n_round * PD;
    for (int rd = 0; rd < n_round; ++rd) {
#pragma unroll
      for (int dd = 0; dd < PD; ++dd) {
        const int gi = rd * PD + dd;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const f32x2 a = ra[dd][u], x = rx[dd][u];
          __builtin_amdgcn_sched_barrier(0);
          c00 = mfma32(a.x, x.x, c00);
          // the bias column sums, pinned HERE: written as `sa0 += a.x` the adds are sunk across the scheduling
          // barriers, `a` outlives the reload of its ring slot, the ring is rotated with v_mov copies at the loop
          // head -- and those copies wait for every load in flight (the ring drained once per round)
          asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(sa0), "+v"(sa1) : "v"(a.x), "v"(a.y));
          __builtin_amdgcn_sched_barrier(0);
          ra[(dd + PD - 1) % PD][u] = lda(gi + PD - 1, u);
          __builtin_amdgcn_sched_barrier(0);
          c01 = mfma32(a.x, x.y, c01);
          __builtin_amdgcn_sched_barrier(0);
          rx[(dd + PD - 1) % PD][u] = ldx(gi + PD - 1, u);
          __builtin_amdgcn_sched_barrier(0);
          c10 = mfma32(a.y, x.x, c10);
          c11 = mfma32(a.y, x.y, c11);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int d = 0; d < PD - 1; ++d) {
      if (d < rem) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const f32x2 a = ra[d][u], x = rx[d][u];
          c00 = mfma32(a.x, x.x, c00);
          c01 = mfma32(a.x, x.y, c01);
          c10 = mfma32(a.y, x.x, c10);
          c11 = mfma32(a.y, x.y, c11);
          sa0 += a.x;
          sa1 += a.y;
        }
      }
    }
  }
  if (nrow > 2 * U * n_full) {   // ragged tail (< 8 rows): clamped rows, masked to zero, loaded in one round trip
    f32x2 ta[U], tx[U];
    const int t0 = wb0 + 2 * U * n_full;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = t0 + 2 * u + p;
      const uint32_t bc = static_cast<uint32_t>(b < wb1 ? b : wb0);
      ta[u] = *(const DCTR_GLOBAL f32x2*)(dbase + (4u * static_cast<uint32_t>(mc) + bc * ldh4));
      tx[u] = *(const DCTR_GLOBAL f32x2*)(ibase + (4u * static_cast<uint32_t>(kc) + bc * ldi4));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool vr = t0 + 2 * u + p < wb1;
      const float a0 = vr ? ta[u].x : 0.f, a1 = vr ? ta[u].y : 0.f;   // (the clamped row holds finite data: 0 * x = 0)
      c00 = mfma32(a0, tx[u].x, c00);
      c01 = mfma32(a0, tx[u].y, c01);
      c10 = mfma32(a1, tx[u].x, c10);
      c11 = mfma32(a1, tx[u].y, c11);
      sa0 += a0;
      sa1 += a1;
    }
  }
  MLP_TRACE(A.trace, 1);
  // park the wave's tile at its logical coordinates: accumulator (pa, pb), register r, lane (p, jl) is
  // dW[m0 + 2 acc_row32(r, p) + pa][k0 + 2 jl + pb]
  {
    float* dst = red + wv * 4096 + 2 * jl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 2 * acc_row32(r, p);
      *reinterpret_cast<f32x2*>(dst + n * 64) = f32x2{c00[r], c01[r]};
      *reinterpret_cast<f32x2*>(dst + (n + 1) * 64) = f32x2{c10[r], c11[r]};
    }
  }
  sa0 += __shfl_xor(sa0, 32, kWave);
  sa1 += __shfl_xor(sa1, 32, kWave);
  if (want_bias && p == 0) *reinterpret_cast<f32x2*>(redb + wv * 64 + 2 * jl) = f32x2{sa0, sa1};
  __syncthreads();
  MLP_TRACE(A.trace, 2);
  // the four waves' tiles summed in wave order; thread t stores 4 x (4 consecutive k of one row)
  float* pw = part + A.off_w[l];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e4 = tid + kTW * i;
    const int n = e4 >> 4, k4 = (e4 & 15) << 2;
    const float* src = red + n * 64 + k4;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(src), q1 = *reinterpret_cast<const f32x4*>(src + 4096),
                q2 = *reinterpret_cast<const f32x4*>(src + 2 * 4096), q3 = *reinterpret_cast<const f32x4*>(src + 3 * 4096);
    f32x4 q = ((q0 + q1) + q2) + q3;
    const int row = m0 + n, col = k0 + k4;
    if (row < Ld.N && col < Ld.ldw) {   // ldw % 4 == 0: the four columns are inside the row or all outside it
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (col + c >= Ld.K) q[c] = 0.f;
      if (A.cnt) stg_wt(pw + static_cast<int64_t>(row) * Ld.ldw + col, q);
      else *(DCTR_GLOBAL f32x4*)(pw + static_cast<int64_t>(row) * Ld.ldw + col) = q;
    }
  }
  if (want_bias && tid < 64 && m0 + tid < Ld.N)
    stg_f32(part + A.off_b[l] + m0 + tid, ((redb[tid] + redb[64 + tid]) + redb[128 + tid]) + redb[192 + tid], A.cnt != nullptr);
  MLP_TRACE(A.trace, 3);
  if (A.cnt) {
    // ---- folded reduction: the tile's last workgroup to arrive sums its S slabs (slab order) and steps the parameters
    int* flag = reinterpret_cast<int*>(wsm + 4 * 4096 + 256);
    if (!fold_arrive(A.cnt + (A.blk0[l] / A.S + tile), A.S, flag)) return;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(A.part, 0, static_cast<int>(A.S * A.slab * 4), 0x00020000);
    const uint32_t slab4 = static_cast<uint32_t>(A.slab) * 4u;
    const auto rs_p = __builtin_amdgcn_make_buffer_rsrc(A.step.param_base ? A.step.param_base : A.part, 0, 0x7FFFFFF0, 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(A.step.state_base ? A.step.state_base : A.part, 0, 0x7FFFFFF0, 0x00020000);
    float* gWl = A.gW[l];
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {      // two rounds of two element quads: (S + 2) x 2 loads in flight per thread
      f32x4 acc[2], w4[2], st4[2];
      bool ok[2];
      float* dptr[2];
      f32x4 v[2][8];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int e4 = tid + kTW * (2 * ih + ii);
        const int n = e4 >> 4, k4 = (e4 & 15) << 2;
        const int row = m0 + n, col = k0 + k4;
        ok[ii] = row < Ld.N && col < Ld.ldw;
        const int64_t o = ok[ii] ? static_cast<int64_t>(row) * Ld.ldw + col : 0;
        dptr[ii] = gWl + o;
        const uint32_t b0_ = static_cast<uint32_t>(A.off_w[l] + o) * 4u;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[ii][q] = ld_slab4(rs, b0_ + static_cast<uint32_t>(q < A.S ? q : A.S - 1) * slab4);
        // (sc1 loads: a parameter line must never sit in this CU's L1 -- a tower workgroup that waits for this launch's
        // signal may run on the same CU and read the line right after it was stepped)
        const uint32_t kb = static_cast<uint32_t>(dptr[ii] - A.step.grad_base) * 4u;
        w4[ii] = A.step.kind >= 0 ? ld_slab4(rs_p, kb) : f32x4{0.f, 0.f, 0.f, 0.f};
        st4[ii] = A.step.kind == DCTR_UPD_ADAGRAD ? ld_slab4(rs_s, kb) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        acc[ii] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < A.S) acc[ii] += v[ii][q];
        for (int s0 = 8; s0 < A.S; s0 += 8) {    // (more than 8 slices: never at the BASELINE shapes)
          const uint32_t b0_ = static_cast<uint32_t>(A.off_w[l] + (dptr[ii] - gWl)) * 4u;
          for (int q = 0; q < 8 && s0 + q < A.S; ++q) acc[ii] += ld_slab4(rs, b0_ + static_cast<uint32_t>(s0 + q) * slab4);
        }
        if (ok[ii] && gWl) fold_step4(A.step, dptr[ii], acc[ii], w4[ii], st4[ii]);
      }
    }
    if (want_bias && tid < 64 && m0 + tid < Ld.N && A.gb[l]) {
      const uint32_t b0_ = static_cast<uint32_t>(A.off_b[l] + m0 + tid) * 4u;
      float acc = 0.f;
      for (int s0 = 0; s0 < A.S; s0 += 8) {
        float vb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) vb[q] = ld_slab1(rs, b0_ + static_cast<uint32_t>(s0 + q < A.S ? s0 + q : A.S - 1) * slab4);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (s0 + q < A.S) acc += vb[q];
      }
      fold_step1(A.step, A.gb[l] + m0 + tid, acc);
    }
    fold_done(A.sync, A.n_red);
  }
#ifdef DCTR_DIAG
  if (A.trace && threadIdx.x == 0) A.trace[blockIdx.x * 16ull + 14] = static_cast<unsigned long long>(l);
#endif
}

struct ReduceArgs {
  int n_seg, S;
  int64_t slab;
  int64_t seg_off[2 * kMaxL + 2];  // float offset of segment i inside a slab; [n_seg] = end
  int64_t seg_len[2 * kMaxL + 1];  // floats of the segment that belong to the destination tensor
  float* seg_dst[2 * kMaxL + 1];   // nullable: skipped
  const float* part;
  // fused train step: one extra workgroup sums the per-row-tile partials of the head
  const float* head_loss;          // [n_head] nullable
  const float* head_gbias;         // [n_head]
  int n_head;
  float* loss;                     // [1]
  float* g_bias;                   // [1] nullable
  DenseStepDev step;               // kind >= 0: also step the parameter behind every gradient element written
};

// (Folding this reduction into k_mlp_wgrad -- the last of a tile's S workgroups to arrive sums the partials -- was
// measured in round 2: correct, and TWICE the step time (0.223 vs 0.113 ms).  The partials must cross XCDs, i.e. every
// workgroup needs an agent-scope release fence, and on this part that is a write-back of the whole per-XCD L2, issued
// 296 times beside an embedding update that keeps the L2 full of dirty table lines.  A kernel boundary does it once.)
__global__ __launch_bounds__(256) void k_mlp_reduce(ReduceArgs A) {
  step_priority<DCTR_WGRAD_PRIORITY>();
  if (A.head_loss && blockIdx.x == gridDim.x - 1) {   // fixed-order tree over the row tiles' partial sums
    __shared__ float red[2][4];
    float l = 0.f, gsum = 0.f;
    for (int k = threadIdx.x; k < A.n_head; k += 256) {
      l += ldg_f32(A.head_loss + k);
      gsum += ldg_f32(A.head_gbias + k);
    }
    l = wave_sum(l);
    gsum = wave_sum(gsum);
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = l;
      red[1][threadIdx.x >> 6] = gsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      stg_f32(A.loss, ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]);
      if (A.g_bias) {
        const float gb = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        stg_f32(A.g_bias, gb);
        dense_step_apply(A.step, A.g_bias, gb);
      }
    }
    return;
  }
  // 4 consecutive floats of the slab per thread (every segment starts at a multiple of 4 floats: a float4 never
  // straddles two tensors).  The S partials are fetched 8 at a time, all loads of a batch in flight together (round 3:
  // the loop used to be `acc += load` with a runtime trip count -- S dependent round trips, 5.5 us for 5.7 MB), and
  // added in slab order.
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i >= A.seg_off[A.n_seg]) return;
  int sg = 0;
  while (i >= A.seg_off[sg + 1]) ++sg;
  float* dst = A.seg_dst[sg];
  const int64_t o = i - A.seg_off[sg];
  if (!dst || o >= A.seg_len[sg]) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < A.S; s0 += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sj = s0 + j < A.S ? s0 + j : A.S - 1;
      v[j] = ldg_f4(A.part + static_cast<int64_t>(sj) * A.slab + i);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (s0 + j < A.S) acc += v[j];
  }
  const int nv = (A.seg_len[sg] - o) < 4 ? static_cast<int>(A.seg_len[sg] - o) : 4;
  float* d = dst + o;
  // (row padding of a weight: gradient 0, parameter 0, Adagrad state 0 -- the step leaves all three alone)
  if (A.step.kind < 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < nv) stg_f32(d + c, acc[c]);
    return;
  }
  // the optimizer step on the elements just finished: parameter and state loads of all four first, then the stores
  const int64_t k = d - A.step.grad_base;
  float w[4], st[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int cc = c < nv ? c : 0;
    w[c] = ldg_f32(A.step.param_base + k + cc);
    st[c] = A.step.kind == DCTR_UPD_ADAGRAD ? ldg_f32(A.step.state_base + k + cc) : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < nv) {
      const float g = acc[c];
      stg_f32(d + c, g);
      if (A.step.kind == DCTR_UPD_ADAGRAD) {   // torch.optim.Adagrad: s += g*g ; p -= lr * g / (sqrt(s) + eps)
        const float sn = adagrad_sum(st[c], g);
        stg_f32(A.step.state_base + k + c, sn);
        stg_f32(A.step.param_base + k + c, adagrad_param(w[c], g, sn, A.step.lr, A.step.eps));
      } else {                                 // torch.optim.SGD
        stg_f32(A.step.param_base + k + c, sgd_param(w[c], g, A.step.lr));
      }
    }
  }
}

// ---- host helpers ---------------------------------------------------------------------------------------------
int check_mlp(const dctr_mlp_t* m, int32_t B) {
  if (!m || B < 0 || m->n_layers <= 0 || m->n_layers > kMaxL) return DCTR_EINVAL;
  for (int l = 0; l < m->n_layers; ++l) {
    const dctr_mlp_layer_t& L = m->layer[l];
    if (!L.W || L.K <= 0 || L.N <= 0 || L.ld_w < L.K) return DCTR_EINVAL;
    if (L.ld_w % 4 != 0 || reinterpret_cast<uintptr_t>(L.W) % 16 != 0) return DCTR_EALIGN;
    if (l > 0 && L.K != m->layer[l - 1].N) return DCTR_EINVAL;
    if (L.N > 2048) return DCTR_ENOSUP;
    if (static_cast<int64_t>(L.N) * L.ld_w * 4 >= (int64_t(1) << 31)) return DCTR_ENOSUP;  // 32-bit weight offsets
  }
  return DCTR_OK;
}

void fill_layers(const dctr_mlp_t* m, LayerDev* L) {
  for (int l = 0; l < m->n_layers; ++l) {
    const dctr_mlp_layer_t& s = m->layer[l];
    L[l].W = s.W; L[l].bias = s.bias; L[l].h = s.h; L[l].dh = s.dh;
    L[l].K = s.K; L[l].N = s.N; L[l].ldw = s.ld_w; L[l].ldh = s.ld_h; L[l].relu = s.relu;
  }
}

// Columns of the tower input staged at a time: as many as leave room for the two activation tiles of the widest layer
// (1024-wide towers: 256; the chunk only bounds how often the first layer's K loop re-stages).
int pick_kc(int K0p, int rsh) {
  int kc = kKC;
  while (kc > 64 && static_cast<size_t>(kTM) * ((K0p < kc ? K0p : kc) + kPad + 2 * rsh) * 4 > 150 * 1024) kc >>= 1;
  return kc;
}

// > 0 when the tower fits the fast bodies: one staged input chunk of at most 512 columns, every layer at most 512 wide (one
// pass of <= 4 output tiles per wave forward, <= 8 column groups backward)
int tower_fast(const dctr_mlp_t* m, int kc) {
  const int K0p = round_up(m->layer[0].K, 16);
  if (K0p > kc || K0p > 512) return 0;
  int narrow = 1;
  for (int l = 0; l < m->n_layers; ++l) {
    if (m->layer[l].N > 512 || m->layer[l].K > 512) return 0;
    if (m->layer[l].K > 256) narrow = 0;
  }
  return narrow ? 2 : 1;     // 2: the backward uses 32-column groups
}

uint32_t diag_wmask() {
#ifdef DCTR_DIAG
  if (const char* e = getenv("DCTR_MLP_WMASK")) return static_cast<uint32_t>(strtoul(e, nullptr, 0));
#endif
  return 0xffffffffu;
}

int max_width(const dctr_mlp_t* m) {
  int w = 0;
  for (int l = 0; l < m->n_layers; ++l) w = m->layer[l].N > w ? m->layer[l].N : w;
  return w;
}

struct WgradPlan {
  int S, bs, P, pbs;
  int blk0[kMaxL + 2];
  int64_t off_w[kMaxL], off_b[kMaxL], off_o, slab;
};

WgradPlan plan_wgrad(const dctr_mlp_t* m, int32_t B) {
  WgradPlan P;
  // Batch slices per 64x64 output tile.  The weight-gradient kernel saturates the matrix pipe with ONE wave per SIMD
  // (32x32x2 issues back to back from a single accumulator chain), so the aim is one 4-wave workgroup per CU, all
  // resident at once and all of equal length: S ~ 256 CUs / tiles (DeepFM tower: 36 tiles -> 7 slices of ~585 rows,
  // 252 workgroups).  Fewer slices also mean fewer partial slabs for k_mlp_reduce.  Slices stay >= 64 rows.
  // (Round 2 ran 10 slices = 360 workgroups, i.e. 104 CUs with two and 152 with one: a 1.4-wave tail.)
  int tiles = 0;
  for (int l = 0; l < m->n_layers; ++l) tiles += ((m->layer[l].N + 63) / 64) * ((m->layer[l].K + 63) / 64);
  // cost model: workgroups spread evenly over 256 CUs, a CU's workgroups run back to back (two co-resident ones share
  // its matrix pipe, which is the same thing), each costs its rows + ~150 rows' worth of start-up / LDS combine.  One
  // CU is kept for the projection's workgroup(s): at 7 slices of the DeepFM tower 252 + 7 workgroups was 3 more than the
  // chip has CUs -- three CUs ran two GEMM workgroups back to back and the launch took twice as long (round 3).
  int S = 1;
  {
    int64_t best = -1;
    for (int s = 1; s <= 16 && s <= (B >= 64 ? B / 64 : 1); ++s) {
      const int64_t rounds = (static_cast<int64_t>(tiles) * s + (m->w_out ? 1 : 0) + 255) / 256;
      const int64_t cost = rounds * ((B + s - 1) / s + 150);
      if (best < 0 || cost < best) {
        best = cost;
        S = s;
      }
    }
  }
#ifdef DCTR_DIAG
  if (const char* e = getenv("DCTR_WGRAD_SLICES")) S = atoi(e);   // tools/tower_bench.py sweeps it
#endif
  if (S > B / 64) S = B / 64;
  if (S > 16) S = 16;
  if (S < 1) S = 1;
  P.S = S;
  P.bs = round_up((B + S - 1) / S, 8);
  int64_t off = 0;
  int blk = 0;
  for (int l = 0; l < m->n_layers; ++l) {
    const dctr_mlp_layer_t& L = m->layer[l];
    P.blk0[l] = blk;
    blk += ((L.N + 63) / 64) * ((L.K + 63) / 64) * S;
    P.off_w[l] = off;
    off += static_cast<int64_t>(L.N) * L.ld_w;
    P.off_b[l] = off;
    off += round_up(L.N, 4);
  }
  P.blk0[m->n_layers] = blk;
  P.off_o = off;
  P.P = 0;
  P.pbs = 0;
  if (m->w_out) {
    // projection workgroups: as many as CUs are left in the last round of GEMM workgroups, between 1 and S
    int spare = 256 - blk % 256;
    if (blk % 256 == 0) spare = 0;
    P.P = spare < 1 ? 1 : (spare > S ? S : spare);
    if (P.P > 4 && B / P.P < 512) P.P = B / 512 > 0 ? (B / 512 < P.P ? B / 512 : P.P) : 1;   // no point in slivers
    P.pbs = round_up((B + P.P - 1) / P.P, 8);
    blk += P.P;
    off += round_up(m->layer[m->n_layers - 1].N, 4);
  }
  P.blk0[m->n_layers + 1] = blk;
  P.slab = off;
  return P;
}

}  // namespace

// diagnostics: buf holds 3 x 4096 x 16 u64 (forward | backward-data | wgrad workgroups); NULL switches it off
#ifdef DCTR_DIAG
extern "C" void dctr_dbg_mlp_trace(unsigned long long* buf) { g_mlp_trace = buf; }
#endif

extern "C" int dctr_mlp_fwd(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, float* logit,
                            dctr_stream_t stream) {
  const int rc = check_mlp(m, B);
  if (rc != DCTR_OK) return rc;
  if (!x || ld_x < m->layer[0].K) return DCTR_EINVAL;
  if (ld_x % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0) return DCTR_EALIGN;
  if (m->w_out && !logit) return DCTR_EINVAL;
  if (!m->w_out && !m->layer[m->n_layers - 1].h) return DCTR_EINVAL;  // nowhere to put the result
  for (int l = 0; l < m->n_layers; ++l)
    if (m->layer[l].h && (m->layer[l].ld_h < m->layer[l].N)) return DCTR_EINVAL;
  if (B == 0) return DCTR_OK;
  MlpArgs a;
  fill_layers(m, a.L);
  a.n_layers = m->n_layers; a.B = B; a.x = x; a.ldx = ld_x; a.w_out = m->w_out; a.logit = logit;
  a.g = nullptr; a.ldg = 0; a.gx = nullptr; a.ldgx = 0;
  a.trace = g_mlp_trace;
  const int K0p = round_up(m->layer[0].K, 16);
  a.rsh = round_up(max_width(m), 16) + kPad;
  a.kc = pick_kc(K0p, a.rsh);
  a.rsx = (K0p < a.kc ? K0p : a.kc) + kPad;
  a.rsd = 0;
  a.fast = tower_fast(m, a.kc); a.wmask = diag_wmask(); a.sync = nullptr; a.wt = 0;
  const size_t lds = static_cast<size_t>(kTM) * (a.rsx + 2 * a.rsh) * 4;
  if (lds > 150 * 1024) return DCTR_ENOSUP;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
  k_mlp_fwd<<<dim3((B + kTM - 1) / kTM), dim3(kT), lds, static_cast<hipStream_t>(stream)>>>(a);
  return launch_status();
}

extern "C" size_t dctr_mlp_bwd_workspace_floats(const dctr_mlp_t* m, int32_t B) {
  if (check_mlp(m, B) != DCTR_OK) return 0;
  const WgradPlan P = plan_wgrad(m, B);
  return static_cast<size_t>(P.slab) * P.S;
}

namespace {

int check_bwd(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* gx, int64_t ld_gx) {
  if (!x || ld_x < m->layer[0].K) return DCTR_EINVAL;
  if (gx && (ld_gx < m->layer[0].K)) return DCTR_EINVAL;
  if (gx && (ld_gx % 4 != 0 || reinterpret_cast<uintptr_t>(gx) % 16 != 0)) return DCTR_EALIGN;
  for (int l = 0; l < m->n_layers; ++l) {
    const dctr_mlp_layer_t& L = m->layer[l];
    if (!L.h || !L.dh || L.ld_h < L.N) return DCTR_EINVAL;
    if (L.ld_h % 4 != 0 || reinterpret_cast<uintptr_t>(L.h) % 16 != 0 || reinterpret_cast<uintptr_t>(L.dh) % 16 != 0)
      return DCTR_EALIGN;
    // k_mlp_wgrad addresses its operands with 32-bit byte offsets
    if (static_cast<int64_t>(B) * L.ld_h * 4 >= (int64_t(1) << 32)) return DCTR_ENOSUP;
  }
  if (static_cast<int64_t>(B) * ld_x * 4 >= (int64_t(1) << 32)) return DCTR_ENOSUP;
  return DCTR_OK;
}

int bwd_stride(const dctr_mlp_t* m) {
  int w = max_width(m);
  for (int l = 1; l < m->n_layers; ++l) w = m->layer[l].K > w ? m->layer[l].K : w;
  return round_up(w, 64) + kPad;
}

// weight gradients (split-batch partials) + their fixed-order reduction (+ the head's partials, fused step only)
// (sync + cnt: ONE launch -- the reduction and the optimizer step folded into k_mlp_wgrad's last arrivers, the weights'
// generation advanced at the end: dctr_mlp_train_wgrad_sync)
int launch_wgrad_reduce(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* g,
                        float* workspace, const float* head_loss, const float* head_gbias, int n_head, float* loss,
                        float* g_bias, const dctr_dense_step_t* step, hipStream_t s, int32_t* sync = nullptr,
                        int32_t* cnt = nullptr) {
  const WgradPlan P = plan_wgrad(m, B);
  if (cnt && static_cast<int64_t>(P.S) * P.slab * 4 >= (int64_t(1) << 31)) return DCTR_ENOSUP;
  {
    WgradArgs a;
    fill_layers(m, a.L);
    a.n_layers = m->n_layers; a.B = B; a.S = P.S; a.bs = P.bs; a.P = P.P; a.pbs = P.pbs;
    for (int i = 0; i < kMaxL + 2; ++i) a.blk0[i] = i <= m->n_layers + 1 ? P.blk0[i] : 0;
    for (int l = 0; l < kMaxL; ++l) {
      a.off_w[l] = l < m->n_layers ? P.off_w[l] : 0;
      a.off_b[l] = l < m->n_layers ? P.off_b[l] : 0;
    }
    a.off_o = P.off_o; a.slab = P.slab; a.x = x; a.ldx = ld_x; a.w_out = m->w_out; a.g = g; a.part = workspace;
    a.trace = g_mlp_trace ? g_mlp_trace + 16ull * 8192 : nullptr;
    a.cnt = cnt; a.sync = sync;
    a.n_red = P.blk0[m->n_layers] / P.S + (m->w_out ? 1 : 0);
    for (int l = 0; l < kMaxL; ++l) {
      a.gW[l] = l < m->n_layers ? m->layer[l].gW : nullptr;
      a.gb[l] = (l < m->n_layers && m->layer[l].bias) ? m->layer[l].gbias : nullptr;
    }
    a.g_wo = m->w_out ? m->g_w_out : nullptr;
    a.head_loss = head_loss; a.head_gbias = head_gbias; a.n_head = n_head; a.loss = loss; a.g_bias = g_bias;
    a.step = dense_step_dev(step);
    constexpr int kWgLds = (4 * 4096 + 256 + 16) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kWgLds);
    k_mlp_wgrad<<<dim3(P.blk0[m->n_layers + 1]), dim3(kTW), kWgLds, s>>>(a);
    const int st = launch_status();
    if (st != DCTR_OK || cnt) return st;
  }
  ReduceArgs r;
  r.S = P.S; r.slab = P.slab; r.part = workspace;
  int ns = 0;
  for (int l = 0; l < m->n_layers; ++l) {
    r.seg_off[ns] = P.off_w[l]; r.seg_len[ns] = static_cast<int64_t>(m->layer[l].N) * m->layer[l].ld_w;
    r.seg_dst[ns++] = m->layer[l].gW;
    r.seg_off[ns] = P.off_b[l]; r.seg_len[ns] = m->layer[l].N;
    r.seg_dst[ns++] = m->layer[l].bias ? m->layer[l].gbias : nullptr;
  }
  if (m->w_out) {
    r.seg_off[ns] = P.off_o; r.seg_len[ns] = m->layer[m->n_layers - 1].N;
    r.seg_dst[ns++] = m->g_w_out;
  }
  r.seg_off[ns] = P.slab;
  r.n_seg = ns;
  r.head_loss = head_loss; r.head_gbias = head_gbias; r.n_head = n_head; r.loss = loss; r.g_bias = g_bias;
  r.step = dense_step_dev(step);
  const unsigned nblk = static_cast<unsigned>((P.slab / 4 + 255) / 256) + (head_loss ? 1u : 0u);
  k_mlp_reduce<<<dim3(nblk), dim3(256), 0, s>>>(r);
  return launch_status();
}

}  // namespace

extern "C" int dctr_mlp_bwd(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* g,
                            int64_t ld_g, float* gx, int64_t ld_gx, float* workspace, dctr_stream_t stream) {
  const int rc = check_mlp(m, B);
  if (rc != DCTR_OK) return rc;
  if (!g || !workspace) return DCTR_EINVAL;
  if (!m->w_out && ld_g < m->layer[m->n_layers - 1].N) return DCTR_EINVAL;
  const int rb = check_bwd(m, x, ld_x, B, gx, ld_gx);
  if (rb != DCTR_OK) return rb;
  if (B == 0) return DCTR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    MlpArgs a;
    fill_layers(m, a.L);
    a.n_layers = m->n_layers; a.B = B; a.x = x; a.ldx = ld_x; a.w_out = m->w_out; a.logit = nullptr;
    a.g = g; a.ldg = ld_g; a.gx = gx; a.ldgx = ld_gx;
    a.trace = g_mlp_trace ? g_mlp_trace + 16ull * 4096 : nullptr;
    a.rsx = 0; a.rsh = 0;
    a.rsd = bwd_stride(m);
    a.fast = tower_fast(m, kKC); a.wmask = diag_wmask(); a.sync = nullptr; a.wt = 0;
    const size_t lds = static_cast<size_t>(kTM) * 2 * a.rsd * 4;
    if (lds > 160 * 1024) return DCTR_ENOSUP;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_bwd_data),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    k_mlp_bwd_data<<<dim3((B + kTM - 1) / kTM), dim3(kT), lds, s>>>(a);
    const int st = launch_status();
    if (st != DCTR_OK) return st;
  }
  return launch_wgrad_reduce(m, x, ld_x, B, g, workspace, nullptr, nullptr, 0, nullptr, nullptr, nullptr, s);
}

extern "C" size_t dctr_mlp_train_workspace_floats(const dctr_mlp_t* m, int32_t B) {
  if (check_mlp(m, B) != DCTR_OK) return 0;
  const WgradPlan P = plan_wgrad(m, B);
  return static_cast<size_t>(P.slab) * P.S + 2 * static_cast<size_t>((B + kTM - 1) / kTM);
}

extern "C" int dctr_mlp_train_step(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* part0,
                                   const float* part1, const float* bias, const float* y, float* y_pred, float* loss,
                                   float* g_logit, float* g_bias, float* gx, int64_t ld_gx, float* workspace,
                                   int32_t defer_wgrad, const dctr_dense_step_t* step, dctr_stream_t stream) {
  const int rc = check_mlp(m, B);
  if (rc != DCTR_OK) return rc;
  if (!m->w_out || !y || !y_pred || !loss || !g_logit || !workspace) return DCTR_EINVAL;
  if (ld_x % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0) return DCTR_EALIGN;
  const int rb = check_bwd(m, x, ld_x, B, gx, ld_gx);
  if (rb != DCTR_OK) return rb;
  if (B == 0) return DCTR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const WgradPlan P = plan_wgrad(m, B);
  const int n_tiles = (B + kTM - 1) / kTM;
  float* part_loss = workspace + static_cast<size_t>(P.slab) * P.S;
  float* part_gb = part_loss + n_tiles;
  {
    MlpArgs a;
    fill_layers(m, a.L);
    a.n_layers = m->n_layers; a.B = B; a.x = x; a.ldx = ld_x; a.w_out = m->w_out; a.logit = nullptr;
    a.g = nullptr; a.ldg = 0; a.gx = gx; a.ldgx = ld_gx;
    a.trace = g_mlp_trace;
    const int K0p = round_up(m->layer[0].K, 16);
    a.rsh = round_up(max_width(m), 16) + kPad;
    a.kc = pick_kc(K0p, a.rsh);
    a.rsx = (K0p < a.kc ? K0p : a.kc) + kPad;
    a.rsd = bwd_stride(m);
    a.fast = tower_fast(m, a.kc); a.wmask = diag_wmask(); a.sync = gx ? m->step_sync : nullptr; a.wt = 0;
    const size_t lds_f = static_cast<size_t>(kTM) * (a.rsx + 2 * a.rsh) * 4;
    const size_t lds_b = static_cast<size_t>(kTM) * 2 * a.rsd * 4;
    size_t lds = lds_f > lds_b ? lds_f : lds_b;
    int bwd_off = 0;
    if (lds_f + lds_b <= 150 * 1024) {     // both images fit: the backward keeps the forward's activations in LDS
      bwd_off = static_cast<int>(lds_f / 4);
      lds = lds_f + lds_b;
    }
    if (lds > 150 * 1024) return DCTR_ENOSUP;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_train), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds));
    HeadArgs hd;
    hd.part0 = part0; hd.part1 = part1; hd.bias = bias; hd.y = y; hd.y_pred = y_pred; hd.g_logit = g_logit;
    hd.part_loss = part_loss; hd.part_gbias = part_gb;
    k_mlp_train<<<dim3(n_tiles), dim3(kT), lds, s>>>(a, hd, bwd_off);
    const int st = launch_status();
    if (st != DCTR_OK) return st;
  }
  if (defer_wgrad) return DCTR_OK;   // the caller enqueues dctr_mlp_train_wgrad (possibly on another stream)
  return launch_wgrad_reduce(m, x, ld_x, B, g_logit, workspace, part_loss, part_gb, n_tiles, loss, g_bias, step, s);
}

namespace {
// launch geometry shared by dctr_mlp_train_step and dctr_embed_tower_train_step
struct TrainGeom {
  MlpArgs a;
  size_t lds;
  int bwd_off;
};
int train_geom(const dctr_mlp_t* m, int32_t B, TrainGeom* T) {
  MlpArgs& a = T->a;
  fill_layers(m, a.L);
  a.n_layers = m->n_layers; a.B = B; a.w_out = m->w_out; a.logit = nullptr;
  a.g = nullptr; a.ldg = 0;
  a.trace = g_mlp_trace;
  const int K0p = round_up(m->layer[0].K, 16);
  a.rsh = round_up(max_width(m), 16) + kPad;
  a.kc = pick_kc(K0p, a.rsh);
  a.rsx = (K0p < a.kc ? K0p : a.kc) + kPad;
  a.rsd = bwd_stride(m);
  a.fast = tower_fast(m, a.kc); a.wmask = diag_wmask(); a.wt = 0;
  const size_t lds_f = static_cast<size_t>(kTM) * (a.rsx + 2 * a.rsh) * 4;
  const size_t lds_b = static_cast<size_t>(kTM) * 2 * a.rsd * 4;
  T->lds = lds_f > lds_b ? lds_f : lds_b;
  T->bwd_off = 0;
  if (lds_f + lds_b <= 150 * 1024) {     // both images fit: the backward keeps the forward's activations in LDS
    T->bwd_off = static_cast<int>(lds_f / 4);
    T->lds = lds_f + lds_b;
  }
  return T->lds > 150 * 1024 ? DCTR_ENOSUP : DCTR_OK;
}

int lpr_shift_of(int D) {
  switch (D) {
    case 4: return 0;
    case 8: return 1;
    case 16: return 2;
    case 32: return 3;
    case 64: return 4;
    default: return -1;
  }
}

// the plans / towers the fused gather stage takes (everything else keeps dctr_embed_fwd + dctr_mlp_train_step)
int gather_envelope(const dctr_plan_t* p, const dctr_mlp_t* m, int32_t B, const TrainGeom& T) {
  if (!p || !p->deep || p->n_deep < 1 || p->n_deep_fixed < 1) return DCTR_ENOSUP;
  // pooled VarLen fields (sum / mean / max): their positions listed in the ext block, at most four 16-byte pieces and one
  // wide value per thread of the 16-sample tile
  const bool pooled = p->n_deep != p->n_deep_fixed || p->n_wide != p->n_wide_fixed;
  const int n_gsd = (pooled && p->ext) ? p->ext->n_gslot_deep : 0, n_gsw = (pooled && p->ext) ? p->ext->n_gslot_wide : 0;
  if (pooled) {
    if (!p->ext || !p->ext->gslot_deep || !p->ext->gslot_wide) return DCTR_ENOSUP;
    if ((p->flags & DCTR_PLAN_HAS_MAXPOOL) && (p->ext->ld_amax <= 0 || !p->ext->am_deep_off || !p->ext->am_wide_off)) return DCTR_ENOSUP;
    if ((p->n_deep != p->n_deep_fixed) != (n_gsd > 0) || (p->n_wide != p->n_wide_fixed) != (n_gsw > 0)) return DCTR_ENOSUP;
    if (p->emb_dim <= 0 || kTM * n_gsd * (p->emb_dim / 4) > 4 * kT || kTM * n_gsw > kT) return DCTR_ENOSUP;
  }
  if (p->n_wide > 32 || p->n_wide < 0 || (p->n_wide && !p->wide)) return DCTR_ENOSUP;
  if (p->vec != 4 || lpr_shift_of(p->emb_dim) < 0) return DCTR_ENOSUP;
  const int width = p->n_deep * p->emb_dim;
  if (p->n_dense < 0 || (p->n_dense > 0 && p->dense_off != width) || (p->n_dense == 0 && p->dense_off >= 0)) return DCTR_ENOSUP;
  if (m->layer[0].K != width + p->n_dense) return DCTR_ENOSUP;
  if (!T.a.fast || T.bwd_off <= 0) return DCTR_ENOSUP;
  const int n_wdense = p->wdense_w ? p->n_wdense : 0;
  if (gather_stage_words(p->n_deep, p->n_wide, p->n_xcols, p->n_dense, n_wdense, n_gsd, n_gsw) > 4 * kT) return DCTR_ENOSUP;
  if (gather_lds_words(p->n_deep, p->n_wide, p->n_xcols, p->n_dense, n_wdense, p->emb_dim, n_gsd, n_gsw) > 2 * kTM * T.a.rsd)
    return DCTR_ENOSUP;
  (void)B;
  return DCTR_OK;
}
}  // namespace

extern "C" int dctr_embed_tower_train_supported(const dctr_plan_t* plan, const dctr_mlp_t* m, int32_t B) {
  if (check_mlp(m, B) != DCTR_OK || !m->w_out) return 0;
  TrainGeom T;
  if (train_geom(m, B, &T) != DCTR_OK) return 0;
  return gather_envelope(plan, m, B, T) == DCTR_OK ? 1 : 0;
}

namespace {
int embed_tower_train_step(const dctr_plan_t* plan, const float* X, int64_t ldx, const dctr_mlp_t* m,
                           int32_t B, int32_t want_fm, const float* bias, const float* y, float* y_pred,
                           float* g_logit, float* gx, int64_t ld_gx, float* out, int64_t ld_out,
                           float* fm_s, int64_t ld_s, int32_t* err, float* workspace, int32_t* wsync, int32_t timeout_us,
                           dctr_stream_t stream);
}  // namespace

extern "C" int dctr_embed_tower_train_step(const dctr_plan_t* plan, const float* X, int64_t ldx, const dctr_mlp_t* m,
                                           int32_t B, int32_t want_fm, const float* bias, const float* y, float* y_pred,
                                           float* g_logit, float* gx, int64_t ld_gx, float* out, int64_t ld_out,
                                           float* fm_s, int64_t ld_s, int32_t* err, float* workspace,
                                           dctr_stream_t stream) {
  return embed_tower_train_step(plan, X, ldx, m, B, want_fm, bias, y, y_pred, g_logit, gx, ld_gx, out, ld_out, fm_s, ld_s, err,
                                workspace, nullptr, 0, stream);
}

extern "C" int dctr_embed_tower_train_step_sync(const dctr_plan_t* plan, const float* X, int64_t ldx, const dctr_mlp_t* m,
                                                int32_t B, int32_t want_fm, const float* bias, const float* y, float* y_pred,
                                                float* g_logit, float* gx, int64_t ld_gx, float* out, int64_t ld_out,
                                                float* fm_s, int64_t ld_s, int32_t* err, float* workspace, int32_t* sync,
                                                int32_t timeout_us, dctr_stream_t stream) {
  if (!sync || timeout_us <= 0 || B <= 0) return DCTR_EINVAL;      // (B == 0 would not advance the tower generation)
  return embed_tower_train_step(plan, X, ldx, m, B, want_fm, bias, y, y_pred, g_logit, gx, ld_gx, out, ld_out, fm_s, ld_s, err,
                                workspace, sync, timeout_us, stream);
}

namespace {
int embed_tower_train_step(const dctr_plan_t* plan, const float* X, int64_t ldx, const dctr_mlp_t* m,
                           int32_t B, int32_t want_fm, const float* bias, const float* y, float* y_pred,
                           float* g_logit, float* gx, int64_t ld_gx, float* out, int64_t ld_out,
                           float* fm_s, int64_t ld_s, int32_t* err, float* workspace, int32_t* wsync, int32_t timeout_us,
                           dctr_stream_t stream) {
  const int rc = check_mlp(m, B);
  if (rc != DCTR_OK) return rc;
  if (!plan || !X || !m->w_out || !y || !y_pred || !g_logit || !workspace || !out || !gx) return DCTR_EINVAL;
  if (ldx < plan->n_xcols || ld_out < m->layer[0].K) return DCTR_EINVAL;
  if (ld_out % 4 != 0 || reinterpret_cast<uintptr_t>(out) % 16 != 0) return DCTR_EALIGN;
  if (want_fm && (!fm_s || ld_s < plan->emb_dim)) return DCTR_EINVAL;
  if (fm_s && (ld_s % 4 != 0 || reinterpret_cast<uintptr_t>(fm_s) % 16 != 0)) return DCTR_EALIGN;
  const int rb = check_bwd(m, out, ld_out, B, gx, ld_gx);
  if (rb != DCTR_OK) return rb;
  if (B == 0) return DCTR_OK;
  TrainGeom T;
  const int rg = train_geom(m, B, &T);
  if (rg != DCTR_OK) return rg;
  const int re = gather_envelope(plan, m, B, T);
  if (re != DCTR_OK) return re;
  MlpArgs& a = T.a;
  a.x = out; a.ldx = ld_out; a.gx = gx; a.ldgx = ld_gx; a.sync = nullptr; a.wt = 0;
  const WgradPlan P = plan_wgrad(m, B);
  const int n_tiles = (B + kTM - 1) / kTM;
  HeadArgs hd;
  hd.part0 = nullptr; hd.part1 = nullptr; hd.bias = bias; hd.y = y; hd.y_pred = y_pred; hd.g_logit = g_logit;
  hd.part_loss = workspace + static_cast<size_t>(P.slab) * P.S;
  hd.part_gbias = hd.part_loss + n_tiles;
  GatherArgs G;
  G.deep = plan->deep; G.wide = plan->wide; G.dense_cols = plan->dense_cols; G.wdense_cols = plan->wdense_cols;
  G.wdense_w = plan->wdense_w; G.X = X; G.ldx = ldx; G.out = out; G.ldo = ld_out; G.fm_s = fm_s; G.lds = ld_s; G.err = err;
  G.n_deep = plan->n_deep; G.n_wide = plan->n_wide; G.n_dense = plan->n_dense;
  G.n_wdense = plan->wdense_w ? plan->n_wdense : 0;
  G.nc = plan->n_xcols; G.dense_off = plan->dense_off; G.D = plan->emb_dim; G.lpr_shift = lpr_shift_of(plan->emb_dim);
  G.want_fm = want_fm ? 1 : 0; G.scratch_off = T.bwd_off;
  G.n_deep_fixed = plan->n_deep_fixed; G.n_wide_fixed = plan->n_wide_fixed;
  const bool pooled = plan->n_deep != plan->n_deep_fixed || plan->n_wide != plan->n_wide_fixed;
  G.n_gsd = pooled ? plan->ext->n_gslot_deep : 0; G.n_gsw = pooled ? plan->ext->n_gslot_wide : 0;
  G.gsd = pooled ? plan->ext->gslot_deep : nullptr; G.gsw = pooled ? plan->ext->gslot_wide : nullptr;
  const bool maxp = pooled && (plan->flags & DCTR_PLAN_HAS_MAXPOOL);
  if (maxp && !plan->ext->amax) return DCTR_EINVAL;       // (the caller points ext->amax at this step's arg-max buffer)
  G.amax = maxp ? plan->ext->amax : nullptr; G.ld_am = maxp ? plan->ext->ld_amax : 0;
  G.am_deep_off = maxp ? plan->ext->am_deep_off : nullptr; G.am_wide_off = maxp ? plan->ext->am_wide_off : nullptr;
  G.wsync = wsync; G.wtimeout = static_cast<unsigned long long>(timeout_us > 0 ? timeout_us : 0) * 100ull;
  if (T.lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_embed_tower_train),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(T.lds));
  k_embed_tower_train<<<dim3(n_tiles), dim3(kT), T.lds, static_cast<hipStream_t>(stream)>>>(a, hd, T.bwd_off, G);
  return launch_status();
}
}  // namespace

extern "C" int dctr_mlp_train_wgrad(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* g_logit,
                                    float* workspace, float* loss, float* g_bias, const dctr_dense_step_t* step,
                                    dctr_stream_t stream) {
  const int rc = check_mlp(m, B);
  if (rc != DCTR_OK) return rc;
  if (!m->w_out || !x || !loss || !g_logit || !workspace) return DCTR_EINVAL;
  if (ld_x % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0) return DCTR_EALIGN;
  if (B == 0) return DCTR_OK;
  const WgradPlan P = plan_wgrad(m, B);
  const int n_tiles = (B + kTM - 1) / kTM;
  float* part_loss = workspace + static_cast<size_t>(P.slab) * P.S;
  float* part_gb = part_loss + n_tiles;
  return launch_wgrad_reduce(m, x, ld_x, B, g_logit, workspace, part_loss, part_gb, n_tiles, loss, g_bias, step,
                             static_cast<hipStream_t>(stream));
}

extern "C" size_t dctr_mlp_train_wgrad_counters(const dctr_mlp_t* m, int32_t B) {
  if (check_mlp(m, B) != DCTR_OK) return 0;
  const WgradPlan P = plan_wgrad(m, B);
  return static_cast<size_t>(P.blk0[m->n_layers] / P.S + 1);
}

extern "C" int dctr_mlp_train_wgrad_sync(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* g_logit,
                                         float* workspace, float* loss, float* g_bias, const dctr_dense_step_t* step,
                                         int32_t* sync, int32_t* counters, dctr_stream_t stream) {
  const int rc = check_mlp(m, B);
  if (rc != DCTR_OK) return rc;
  if (!m->w_out || !x || !loss || !g_logit || !workspace || !sync || !counters) return DCTR_EINVAL;
  if (ld_x % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0) return DCTR_EALIGN;
  for (int l = 0; l < m->n_layers; ++l)
    if (!m->layer[l].gW || reinterpret_cast<uintptr_t>(m->layer[l].gW) % 16 != 0) return DCTR_EALIGN;
  if (step && (reinterpret_cast<uintptr_t>(step->param_base) % 16 != 0 || reinterpret_cast<uintptr_t>(step->grad_base) % 16 != 0 ||
               reinterpret_cast<uintptr_t>(step->state_base) % 16 != 0))
    return DCTR_EALIGN;
  if (B == 0) return DCTR_EINVAL;       // (the generation must advance once per call)
  const WgradPlan P = plan_wgrad(m, B);
  const int n_tiles = (B + kTM - 1) / kTM;
  float* part_loss = workspace + static_cast<size_t>(P.slab) * P.S;
  float* part_gb = part_loss + n_tiles;
  return launch_wgrad_reduce(m, x, ld_x, B, g_logit, workspace, part_loss, part_gb, n_tiles, loss, g_bias, step,
                             static_cast<hipStream_t>(stream), sync, counters);
}

// ---- CrossNet, matrix parameterisation (interaction.py:448-451) ------------------------------------------------------
namespace {
int check_cross(const dctr_mlp_t* m, int32_t B) {
  const int rc = check_mlp(m, B);
  if (rc != DCTR_OK) return rc;
  if (m->w_out) return DCTR_EINVAL;
  const int W = m->layer[0].K;
  for (int l = 0; l < m->n_layers; ++l) {
    const dctr_mlp_layer_t& L = m->layer[l];
    if (L.K != W || L.N != W || !L.h || !L.dh || L.ld_h < W) return DCTR_EINVAL;
    if (L.ld_h % 4 != 0 || reinterpret_cast<uintptr_t>(L.h) % 16 != 0 || reinterpret_cast<uintptr_t>(L.dh) % 16 != 0)
      return DCTR_EALIGN;
  }
  return DCTR_OK;
}
}  // namespace

extern "C" int dctr_crossnet_mat_supported(int32_t W, int32_t n_layers) {
  if (W <= 0 || n_layers <= 0 || n_layers > kMaxL) return 0;
  const int Wp = round_up(W, 16);
  if (Wp > kKC) return 0;                                              // the forward keeps x_0 as ONE staged chunk
  const size_t lds_f = static_cast<size_t>(kTM) * ((Wp + 4) + 2 * (Wp + 4)) * 4;
  const size_t lds_b = static_cast<size_t>(kTM) * 4 * (round_up(W, 64) + 4) * 4;
  return (lds_f <= 150 * 1024 && lds_b <= 150 * 1024) ? 1 : 0;
}

extern "C" int dctr_crossnet_mat_fwd(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B,
                                     dctr_stream_t stream) {
  const int rc = check_cross(m, B);
  if (rc != DCTR_OK) return rc;
  const int W = m->layer[0].K;
  if (!x || ld_x < W) return DCTR_EINVAL;
  if (ld_x % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0) return DCTR_EALIGN;
  if (!dctr_crossnet_mat_supported(W, m->n_layers)) return DCTR_ENOSUP;
  if (B == 0) return DCTR_OK;
  MlpArgs a;
  fill_layers(m, a.L);
  a.n_layers = m->n_layers; a.B = B; a.x = x; a.ldx = ld_x; a.w_out = nullptr; a.logit = nullptr;
  a.g = nullptr; a.ldg = 0; a.gx = nullptr; a.ldgx = 0; a.trace = nullptr;
  const int Wp = round_up(W, 16);
  a.kc = kKC;
  a.rsx = Wp + kPad;
  a.rsh = Wp + kPad;
  a.rsd = 0;
  a.fast = 0; a.wmask = 0xffffffffu; a.sync = nullptr; a.wt = 0;
  const size_t lds = static_cast<size_t>(kTM) * (a.rsx + 2 * a.rsh) * 4;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cross_mat_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
  k_cross_mat_fwd<<<dim3((B + kTM - 1) / kTM), dim3(kT), lds, static_cast<hipStream_t>(stream)>>>(a);
  return launch_status();
}

extern "C" size_t dctr_crossnet_mat_bwd_workspace_floats(const dctr_mlp_t* m, int32_t B) {
  if (check_mlp(m, B) != DCTR_OK) return 0;
  const WgradPlan P = plan_wgrad(m, B);
  return static_cast<size_t>(P.slab) * P.S;
}

extern "C" int dctr_crossnet_mat_bwd(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* gY,
                                     int64_t ld_g, float* gx, int64_t ld_gx, float* workspace, dctr_stream_t stream) {
  const int rc = check_cross(m, B);
  if (rc != DCTR_OK) return rc;
  const int W = m->layer[0].K;
  if (!x || !gY || !workspace || ld_x < W || ld_g < W) return DCTR_EINVAL;
  const int rb = check_bwd(m, x, ld_x, B, gx, ld_gx);
  if (rb != DCTR_OK) return rb;
  if (!dctr_crossnet_mat_supported(W, m->n_layers)) return DCTR_ENOSUP;
  if (B == 0) return DCTR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    MlpArgs a;
    fill_layers(m, a.L);
    a.n_layers = m->n_layers; a.B = B; a.x = x; a.ldx = ld_x; a.w_out = nullptr; a.logit = nullptr;
    a.g = gY; a.ldg = ld_g; a.gx = gx; a.ldgx = ld_gx; a.trace = nullptr;
    a.rsx = 0; a.rsh = 0; a.fast = 0; a.wmask = 0xffffffffu; a.sync = nullptr; a.wt = 0;
    a.rsd = round_up(W, 64) + kPad;
    const size_t lds = static_cast<size_t>(kTM) * 4 * a.rsd * 4;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cross_mat_bwd),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    k_cross_mat_bwd<<<dim3((B + kTM - 1) / kTM), dim3(kT), lds, s>>>(a);
    const int st = launch_status();
    if (st != DCTR_OK) return st;
  }
  // d W_l = a_l^T x_l, d b_l = column sums of a_l: the tower's weight-gradient kernels, as they are
  return launch_wgrad_reduce(m, x, ld_x, B, nullptr, workspace, nullptr, nullptr, 0, nullptr, nullptr, nullptr, s);
}

// ---- CrossNetMix (interaction.py:499-534) -----------------------------------------------------------------------------
namespace {
int check_mix(const dctr_mlp_t* m, int32_t E, int32_t R, int32_t B) {
  // (not check_mlp: the dense layers of a cross layer do not chain by width -- layer 3l+1 reads the first E*R of
  // layer 3l's E*R + E outputs)
  if (!m || B < 0 || m->n_layers <= 0 || m->n_layers > kMaxL) return DCTR_EINVAL;
  for (int l = 0; l < m->n_layers; ++l) {
    const dctr_mlp_layer_t& L = m->layer[l];
    if (!L.W || L.K <= 0 || L.N <= 0 || L.ld_w < L.K) return DCTR_EINVAL;
    if (L.ld_w % 4 != 0 || reinterpret_cast<uintptr_t>(L.W) % 16 != 0) return DCTR_EALIGN;
    if (L.N > 2048) return DCTR_ENOSUP;
  }
  if (m->w_out || E <= 0 || R <= 0 || m->n_layers % 3 != 0) return DCTR_EINVAL;
  const int W = m->layer[0].K, ER = E * R;
  for (int l = 0; l < m->n_layers; ++l) {
    const dctr_mlp_layer_t& L = m->layer[l];
    const int k = l % 3;
    const int wantK = k == 0 ? W : ER, wantN = k == 0 ? ER + E : (k == 1 ? ER : W);
    if (L.K != wantK || L.N != wantN || !L.h || !L.dh || L.ld_h < L.N) return DCTR_EINVAL;
    if (L.ld_h % 4 != 0 || reinterpret_cast<uintptr_t>(L.h) % 16 != 0 || reinterpret_cast<uintptr_t>(L.dh) % 16 != 0)
      return DCTR_EALIGN;
  }
  return DCTR_OK;
}
size_t mix_lds_fwd(int W, int N1) {
  return static_cast<size_t>(kTM) * (3 * (round_up(W, 16) + 4) + 2 * (round_up(N1, 16) + 4)) * 4;
}
size_t mix_lds_bwd(int W, int N1) {
  return static_cast<size_t>(kTM) * (4 * (round_up(W, 64) + 4) + 2 * (round_up(N1, 64) + 4)) * 4;
}
}  // namespace

extern "C" int dctr_crossnet_mix_supported(int32_t W, int32_t n_cross_layers, int32_t E, int32_t R) {
  if (W <= 0 || n_cross_layers <= 0 || 3 * n_cross_layers > kMaxL || E <= 0 || E > 8 || R <= 0) return 0;
  const int N1 = E * R + E;
  if (round_up(W, 16) > kKC || N1 > 512) return 0;
  return (mix_lds_fwd(W, N1) <= 150 * 1024 && mix_lds_bwd(W, N1) <= 150 * 1024) ? 1 : 0;
}

extern "C" int dctr_crossnet_mix_fwd(const dctr_mlp_t* m, int32_t E, int32_t R, const float* x, int64_t ld_x, int32_t B,
                                     dctr_stream_t stream) {
  const int rc = check_mix(m, E, R, B);
  if (rc != DCTR_OK) return rc;
  const int W = m->layer[0].K;
  if (!x || ld_x < W) return DCTR_EINVAL;
  if (!dctr_crossnet_mix_supported(W, m->n_layers / 3, E, R)) return DCTR_ENOSUP;
  if (B == 0) return DCTR_OK;
  MlpArgs a;
  fill_layers(m, a.L);
  a.n_layers = m->n_layers; a.B = B; a.x = x; a.ldx = ld_x; a.w_out = nullptr; a.logit = nullptr;
  a.g = nullptr; a.ldg = 0; a.gx = nullptr; a.ldgx = 0; a.trace = nullptr;
  a.kc = kKC; a.rsx = 0; a.rsh = 0; a.rsd = 0; a.fast = 0; a.wmask = 0xffffffffu; a.sync = nullptr; a.wt = 0;
  const size_t lds = mix_lds_fwd(W, E * R + E);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cross_mix_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
  k_cross_mix_fwd<<<dim3((B + kTM - 1) / kTM), dim3(kT), lds, static_cast<hipStream_t>(stream)>>>(a, E, R);
  return launch_status();
}

extern "C" size_t dctr_crossnet_mix_bwd_workspace_floats(const dctr_mlp_t* m, int32_t B) {
  if (!m || B < 0 || m->n_layers <= 0 || m->n_layers > kMaxL) return 0;
  const WgradPlan P = plan_wgrad(m, B);
  return static_cast<size_t>(P.slab) * P.S;
}

extern "C" int dctr_crossnet_mix_bwd(const dctr_mlp_t* m, int32_t E, int32_t R, const float* x, int64_t ld_x, int32_t B,
                                     const float* gY, int64_t ld_g, float* gx, int64_t ld_gx, float* workspace,
                                     dctr_stream_t stream) {
  const int rc = check_mix(m, E, R, B);
  if (rc != DCTR_OK) return rc;
  const int W = m->layer[0].K;
  if (!x || !gY || !workspace || ld_x < W || ld_g < W) return DCTR_EINVAL;
  const int rb = check_bwd(m, x, ld_x, B, gx, ld_gx);
  if (rb != DCTR_OK) return rb;
  if (!dctr_crossnet_mix_supported(W, m->n_layers / 3, E, R)) return DCTR_ENOSUP;
  if (B == 0) return DCTR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    MlpArgs a;
    fill_layers(m, a.L);
    a.n_layers = m->n_layers; a.B = B; a.x = x; a.ldx = ld_x; a.w_out = nullptr; a.logit = nullptr;
    a.g = gY; a.ldg = ld_g; a.gx = gx; a.ldgx = ld_gx; a.trace = nullptr;
    a.kc = kKC; a.rsx = 0; a.rsh = 0; a.rsd = 0; a.fast = 0; a.wmask = 0xffffffffu; a.sync = nullptr; a.wt = 0;
    const size_t lds = mix_lds_bwd(W, E * R + E);
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cross_mix_bwd),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    k_cross_mix_bwd<<<dim3((B + kTM - 1) / kTM), dim3(kT), lds, s>>>(a, E, R);
    const int st = launch_status();
    if (st != DCTR_OK) return st;
  }
  // d W of the three dense layers per cross layer (= packed gV | gG, gC blocks, gU) and d b: the tower's kernels
  return launch_wgrad_reduce(m, x, ld_x, B, nullptr, workspace, nullptr, nullptr, 0, nullptr, nullptr, nullptr, s);
}

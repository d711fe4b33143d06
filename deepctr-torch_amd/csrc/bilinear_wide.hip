// bilinear_wide.hip -- FiBiNET's bilinear pairs TOGETHER WITH the first tower layer behind them, both directions (the
// backward first: it is where the traffic was; the forward, k_bilinear_fwd_wide, follows it in this file)
// (fibinet.py:82-99: dnn_input = [ Bilinear(senet) | Bilinear(raw) | dense ], h1 = relu(W0 dnn_input + b0);
//  interaction.py:140-156; core.py:123-133).
//
// The product slab P [B, 2 * 325 * 16] is 170 MB at the Criteo shape.  Up to round 5 its GRADIENT was a slab too: a library
// GEMM wrote gP = gh W0 (170 MB), k_bilinear_bwd_data_own read it twice and k_bilinear_bwd_weight a third time -- 349 us
// of a 0.83 ms step moving bytes that exist only between two kernels.  Here the gradient of a (16 samples x one pair)
// piece is MADE where it is consumed:
//     G[b][e]  = sum_h gh[b][h] W0[h][16 k + e]            32 x v_mfma_f32_16x16x4_f32 per pass, exact fp32
//     gX_j    += G (.) (x_i W_k^T)                          as k_bilinear_bwd_data
//     gX_i    += (G (.) x_j) W_k
//     gW_k    += (G (.) x_j)^T x_i                          as k_bilinear_bwd_weight, per-workgroup partials
// A workgroup owns 32 samples of ONE of the two inputs (blockIdx.y: V or E): their gh rows sit in registers as the A
// operands for the whole launch (2 x 32 values per lane), the pair's 16 columns of W0 arrive from L2 through a register
// ring as the B operand -- shared by the two 16-sample halves, which halves the L2 stream against one workgroup per 16
// samples of both inputs (0.68 GB per launch) --, in a layout packed once per step (k_wide_pack: one dwordx4 per lane and
// 16 hidden units, 1 KB per wave instruction).  The gradient slab is never
// written or read; the only slab-sized traffic left in the backward is the weight-gradient GEMM's read of P itself.
//
// Schedule: groups of eight field-disjoint pairs, one per wave of a 512-thread workgroup, a barrier per group (the
// rounds of the tournament hold 13 pairs at 26 fields: 8 + 5 would idle three waves in every round).  TWO waves per SIMD:
// with one, every LDS / memory / vector instruction of a wave is an issue slot its own MFMAs cannot use -- measured 55 %
// matrix-pipe occupancy with four waves, whatever the order of the instructions.  The per-field gradient tiles in LDS
// are plain read-modify-writes in group order: no float atomics, bit-reproducible.
#include "pairwise_tiles.hpp"

namespace {

constexpr int kNQ = 8;          // groups of 16 hidden units: H <= 128
constexpr int kNW = 8;          // waves per workgroup = pairs per group
constexpr int kD = 16;          // embedding width of the fused route

// W0 [H, ldw] (nn.Linear weight: row h = hidden unit) -> Wpk[kb][q][lane] (dwordx4): component s of lane (g, c) is
// W0[16 q + 4 g + s][16 kb + c], zero past H.  The MFMA contraction index of step 4 q + s in lane group g is that h.
// FWD: the forward's layout instead -- component s of lane (g, c) of piece (kb, nb) is W0[16 nb + c][16 kb + 4 g + s]
// (B operand of y[b][h] += sum_e p[b][e] W0[h][16 kb + e] with the contraction index e = 4 g + s).
// One more workgroup (FWD, Wd != NULL): the n_dense columns behind the pairs as a compact [128][32] block -- read where
// they lie (one 52-byte piece per hidden unit, rows 41 KB apart) by every workgroup of the finishing launch, the same
// 128 lines were requested 256 times over: 11 us for a 4 us launch.
template <bool FWD>
// Wpk2 (FWD, nullable): the backward's layout too, from the same staged tile -- the forward's pack launch then serves
// the backward of the same step (W0 does not move between the two: autograd's version check guards it).
__global__ __launch_bounds__(kT) void k_wide_pack(const float* __restrict__ W0, int64_t ldw, int H, int KB,
                                                  f32x4* __restrict__ Wpk, float* __restrict__ Wd, int n_dense,
                                                  f32x4* __restrict__ Wpk2) {
  __shared__ float t[16 * kNQ][65];
  if (FWD && Wd && blockIdx.x == gridDim.x - 1) {
    const int n = 16 * kNQ * 32;
    for (int e0 = threadIdx.x; e0 < n; e0 += 8 * kT) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * kT, h = e >> 5, q = e & 31;
        v[u] = ldg_f32(W0 + static_cast<int64_t>(h < H ? h : 0) * ldw + KB * 16 + (q < n_dense ? q : 0));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * kT, h = e >> 5, q = e & 31;
        Wd[e] = (h < H && q < n_dense) ? v[u] : 0.f;
      }
    }
    return;
  }
  const int kb0 = blockIdx.x * 4;
  const int tid = threadIdx.x, col = tid & 63, r0 = tid >> 6;
  const int ncol = (KB - kb0) * 16 < 64 ? (KB - kb0) * 16 : 64;
  const int cc = col < ncol ? col : 0;
  for (int h0 = 0; h0 < 16 * kNQ; h0 += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int h = h0 + r0 + 4 * u;
      v[u] = ldg_f32(W0 + static_cast<int64_t>(h < H ? h : 0) * ldw + kb0 * 16 + cc);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int h = h0 + r0 + 4 * u;
      t[h][col] = (h < H && col < ncol) ? v[u] : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kNQ; ++u) {
    const int idx = tid + kT * u;       // (kb, q, lane) of this block's 4 * kNQ * 64 pieces
    const int kbl = idx / (kNQ * 64), rem = idx - kbl * (kNQ * 64), q = rem >> 6, lane = rem & 63;
    const int g = lane >> 4, c = lane & 15;
    if (kb0 + kbl < KB) {
      f32x4 v;
      if (FWD) {
        v.x = t[16 * q + c][kbl * 16 + 4 * g + 0];
        v.y = t[16 * q + c][kbl * 16 + 4 * g + 1];
        v.z = t[16 * q + c][kbl * 16 + 4 * g + 2];
        v.w = t[16 * q + c][kbl * 16 + 4 * g + 3];
      } else {
        v.x = t[16 * q + 4 * g + 0][kbl * 16 + c];
        v.y = t[16 * q + 4 * g + 1][kbl * 16 + c];
        v.z = t[16 * q + 4 * g + 2][kbl * 16 + c];
        v.w = t[16 * q + 4 * g + 3][kbl * 16 + c];
      }
      *(DCTR_GLOBAL f32x4*)(Wpk + (static_cast<int64_t>(kb0 + kbl) * kNQ + q) * 64 + lane) = v;
      if (FWD && Wpk2) {
        f32x4 w;
        w.x = t[16 * q + 4 * g + 0][kbl * 16 + c];
        w.y = t[16 * q + 4 * g + 1][kbl * 16 + c];
        w.z = t[16 * q + 4 * g + 2][kbl * 16 + c];
        w.w = t[16 * q + 4 * g + 3][kbl * 16 + c];
        *(DCTR_GLOBAL f32x4*)(Wpk2 + (static_cast<int64_t>(kb0 + kbl) * kNQ + q) * 64 + lane) = w;
      }
    }
  }
}

// barrier between groups: LDS traffic only -- the register ring's loads stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// row stride of the LDS tiles: (RS mod 8) == 4 puts the rows 4g + r of the two lane groups g that share an LDS cycle
// on disjoint bank halves (the 32-mod-16 stride of pairwise.hip leaves every ds_read_b32 of a [4g + r][c] piece 2-way
// conflicted: 40 % of the LDS cycles in the first version's counters, SQ_LDS_BANK_CONFLICT 11.4 M -> 3.4 M)
__device__ __host__ inline int wide_rs(int F) { return F * kD + 4; }

// sched4: [n_groups][8 waves][4] int32 = {i, j, weight index, pair index k}; i = -1: the wave idles in this group.
// part:   [2 inputs x tiles][P][16][16] per-workgroup partial of gW_k (row e, column d), every one written exactly once.
//
// How the kernel got here (one-box timings of dctr_bilinear_wide_bwd, pack + main + reduce; DESIGN.md §3 has the table):
// four waves in step, ring two groups deep, program order 240 us -> ring re-loaded in place behind the MFMAs that read
// it, barrier behind the MFMA block, conflict-free stride 217 -> eight waves 203 -> 32 samples of one input per workgroup
// 193 -> two parts per group with the waves of a SIMD half a group apart (below), compile-time field count, x_i W^T
// behind part 1 ~167.  The group's own structure is described at the lambda below.
// FC: the field count as a compile-time constant (0: read F).  With it every LDS address of a group is one of two
// per-lane registers (the pair's i / j column) plus an immediate; without, the row and tile offsets are vector adds --
// 51 of the 83 vector instructions of part 2, each of which waits for a slot between the other wave's MFMAs.
template <int VAR = 0, int FC = 0>
__global__ __launch_bounds__(64 * kNW) void k_bilinear_bwd_wide(const float* __restrict__ E, int64_t lde,
                                                          const float* __restrict__ V, int64_t ldv,
                                                          const float* __restrict__ Wf,
                                                          const int32_t* __restrict__ sched4, int n_groups, int P,
                                                          int F, int B, const float* __restrict__ gh, int64_t ldgh,
                                                          int H, const f32x4* __restrict__ Wpk,
                                                          float* __restrict__ gE, float* __restrict__ gV,
                                                          float* __restrict__ part) {
  extern __shared__ __align__(16) float smem[];
  uint64_t k0 = 0, k1 = 0, k2 = 0;
  if (VAR & 16) k0 = __builtin_amdgcn_s_memtime();
  const int RS = FC ? wide_rs(FC) : wide_rs(F), W = FC ? FC * kD : F * kD;
  // blockIdx.y: 0 = the V input (columns [0, 16 P) of the DNN input), 1 = the E input (columns [16 P, 32 P))
  const int pass = blockIdx.y;
  const float* X = pass ? E : V;
  const int64_t ldx = pass ? lde : ldv;
  float* gX = pass ? gE : gV;
  float* xs0 = smem;               // samples b0 .. b0 + 15
  float* xs1 = xs0 + kSB * RS;     // samples b0 + 16 .. b0 + 31
  float* gx0 = xs1 + kSB * RS;     // their gradient tiles
  float* gx1 = gx0 + kSB * RS;
  float* tb = gx1 + kSB * RS;      // [waves][2 passes][16][17] layout-change scratch (wave-private)
  int32_t* sch = reinterpret_cast<int32_t*>(tb + kNW * 2 * 16 * 17);   // [n_groups][waves][4]
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * (2 * kSB);
  for (int e = tid; e < kNW * n_groups; e += 64 * kNW)
    *reinterpret_cast<i32x4*>(sch + 4 * e) = *(const DCTR_GLOBAL i32x4*)(sched4 + 4 * e);
  // this lane's A operands of G: gh[b = c (+ 16)][h = 16 q + 4 g + s] (zero past B / H)
  f32x4 ga0[kNQ], ga1[kNQ];
  {
    const int ba = b0 + c, bb = b0 + kSB + c;
    const float* rowa = gh + static_cast<int64_t>(ba < B ? ba : B - 1) * ldgh;
    const float* rowb = gh + static_cast<int64_t>(bb < B ? bb : B - 1) * ldgh;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) {
      const int h = 16 * q + 4 * g;
      ga0[q] = *(const DCTR_GLOBAL f32x4*)(rowa + (h < H ? h : 0));
      ga1[q] = *(const DCTR_GLOBAL f32x4*)(rowb + (h < H ? h : 0));
    }
#pragma unroll
    for (int q = 0; q < kNQ; ++q) {
      if (!(ba < B && 16 * q + 4 * g < H)) ga0[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!(bb < B && 16 * q + 4 * g < H)) ga1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  {
    // each half of the workgroup stages one 16-sample tile (zeros past B): eight UNCONDITIONAL loads per thread in flight
    // on clamped rows, masked at the LDS write (a predicated load is a branch with its own wait: one round trip each)
    const int t2 = tid & (kT - 1), half = tid >> 8;
    float* dst = half ? xs1 : xs0;
    const int bs = b0 + half * kSB;
    if ((ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {      // (uniform)
      const int w4 = W >> 2, n4 = kSB * w4;
      for (int e0 = t2; e0 < n4; e0 += 8 * kT) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kT, ec = e < n4 ? e : 0;
          const int r = ec / w4, q4 = ec - r * w4;
          const int rr = bs + r < B ? bs + r : B - 1;
          v[u] = *(const DCTR_GLOBAL f32x4*)(X + static_cast<int64_t>(rr) * ldx + 4 * q4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kT;
          if (e < n4) {
            const int r = e / w4, q4 = e - r * w4;
            *reinterpret_cast<f32x4*>(dst + r * RS + 4 * q4) = (bs + r < B) ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    } else {
      const int n = kSB * W;
      for (int e0 = t2; e0 < n; e0 += 8 * kT) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kT, ec = e < n ? e : 0;
          const int r = ec / W, cc = ec - r * W;
          const int rr = bs + r < B ? bs + r : B - 1;
          v[u] = ldg_f32(X + static_cast<int64_t>(rr) * ldx + cc);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kT;
          if (e < n) {
            const int r = e / W, cc = e - r * W;
            dst[r * RS + cc] = (bs + r < B) ? v[u] : 0.f;
          }
        }
      }
    }
  }
  for (int e = tid; e < 2 * kSB * RS; e += 64 * kNW) gx0[e] = 0.f;
  __syncthreads();
  float* tb0 = tb + wv * (2 * 16 * 17);
  float* tb1 = tb0 + 16 * 17;
  auto entry = [&](int gi) {
    PairEnt e;
    const int gc = gi < n_groups ? gi : n_groups - 1;
    const i32x4 v = *reinterpret_cast<const i32x4*>(sch + 4 * (kNW * gc + wv));
    e.i = gi < n_groups ? v.x : -1; e.j = v.y; e.wi = v.z; e.k = v.w;
    return e;
  };
  // the pair's 16 columns of W0 (8 dwordx4) and the pair's own weight tile in both operand layouts
  f32x4 w0[kNQ], wr;
  float wtr[4];
  auto load_tile = [&](const PairEnt& e) {
    const float* base = Wf + static_cast<int64_t>(e.wi) * (kD * kD);
    wr = *(const DCTR_GLOBAL f32x4*)(base + c * kD + 4 * g);                     // W[e = c][d = 4g + s]
#pragma unroll
    for (int s = 0; s < 4; ++s) wtr[s] = ldg_f32(base + (4 * g + s) * kD + c);   // W[e = 4g + s][d = c]
  };
  {
    const PairEnt e = entry(0);
    const f32x4* p0 = Wpk + static_cast<int64_t>(pass * P + e.k) * (kNQ * 64) + lane;
#pragma unroll
    for (int q = 0; q < kNQ; ++q) w0[q] = *(const DCTR_GLOBAL f32x4*)(p0 + 64 * q);
    load_tile(e);
  }
  const int64_t tile = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
  uint64_t tG = 0, tB = 0, tD = 0;     // (VAR & 16: shader cycles in the MFMA block / at the barrier / behind it)
  // FULL: this wave's pair is live (no branch in the body: a store behind a branch is one the compiler cannot count, and
  // its waits for the ring then fall back to "everything older").
  //
  // A group of a wave is two parts, and the two waves of a SIMD run HALF A GROUP APART (below): between two barriers one
  // wave is in part 1, the other in part 2.
  //   part 1  64 MFMAs back to back (G of both sample halves) with nothing between them but the ring's re-loads -- one
  //           dwordx4 behind every 8 MFMAs, into the registers those 8 have just read (in place, a whole group ahead) --
  //           then t = x_i W^T (2 LDS reads, 8 MFMAs: it needs nothing of G and nothing behind the barrier)
  //   part 2  everything with latency in it: the pair's other LDS operands in the order of need, the transposition of
  //           G (.) x_j through LDS, gW / gX_i MFMAs (16), the LDS read-modify-writes of the two fields' gradient rows,
  //           the partial's stores, the next pair's own weight tile
  // Phase stamps (DCTR_WIDE_VAR=216): part 1 2 750-3 100 cycles, part 2 3 450-3 700, for 2 816 cycles of MFMA per
  // interval and SIMD.  (4 waves in step: matrix pipe 55 % busy whatever the instruction order -- with one wave per SIMD
  // every LDS / memory / vector instruction is an issue slot its own MFMAs cannot use; 8 waves in step: 79 % in part 1,
  // 75 % in part 2.)
  PairEnt cur = entry(0), nxt = entry(1);
  auto group = [&](auto full_tag, int gi) {
    constexpr bool FULL = decltype(full_tag)::value;
    // (VAR & 64: the schedule entries carried in registers from the previous group instead -- measured 3-4 % slower)
    const PairEnt en = (VAR & 64) ? cur : entry(gi), nx = (VAR & 64) ? nxt : entry(gi + 1);
    const bool live = FULL || en.i >= 0;
    const int i = live ? en.i : 0, j = live ? en.j : 0;
    // (VAR & 32: every group re-reads the first pair's columns -- the same instructions out of the CU's L1)
    const f32x4* p0 = Wpk + static_cast<int64_t>((VAR & 32) ? 0 : pass * P + nx.k) * (kNQ * 64) + lane;
    uint64_t c0 = 0, c1 = 0, c2 = 0;
    if (VAR & 16) c0 = __builtin_amdgcn_s_memtime();
    // ---- part 1: G[b = 4g + r][e = c] of both sample halves, two independent chains of 32 over the same B operands
    const int oi = 4 * g * RS + i * kD + c, oj = 4 * g * RS + j * kD + c;
    f32x4 a[2];
    float xj[2][4], xi[2][4], gj[2][4], gi_[2][4];
    f32x4 G0 = {0.f, 0.f, 0.f, 0.f}, G1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < kNQ; ++q) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        G0 = mfma16(ga0[q][s], w0[q][s], G0);
        G1 = mfma16(ga1[q][s], w0[q][s], G1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(VAR & 4)) w0[q] = *(const DCTR_GLOBAL f32x4*)(p0 + 64 * q);
      __builtin_amdgcn_sched_barrier(0);
    }
    // t[b = 4g + r][e = c] = (x_i W^T): needs nothing of G and nothing behind the barrier -- 8 more MFMAs for this part,
    // one LDS round trip and 8 dependent MFMAs less for the long part 2
    a[0] = *reinterpret_cast<const f32x4*>(xs0 + c * RS + i * kD + 4 * g);      // x_i[b = c][d = 4g + s]
    a[1] = *reinterpret_cast<const f32x4*>(xs1 + c * RS + i * kD + 4 * g);
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      t0 = mfma16(a[0][s], wr[s], t0);
      t1 = mfma16(a[1][s], wr[s], t1);
    }
    // (an opaque use: the mfma intrinsics are pure, and the compiler otherwise sinks the block's last MFMAs below the
    // re-loads of their own operands -- overlapping live ranges, a copy at the loop's end that waits for every load)
    asm volatile("" : "+v"(G0), "+v"(G1), "+v"(t0), "+v"(t1));
    if (VAR & 16) c1 = __builtin_amdgcn_s_memtime();
    if (!(VAR & 1)) lds_barrier();     // the other half of the waves has finished its part 2: its gradient rows are in LDS
    if (VAR & 16) c2 = __builtin_amdgcn_s_memtime();
    // ---- part 2  (measured and dropped: a raised issue priority for it, 195 against 183 us; the read-only operands
    // read behind part 1's last MFMAs instead, 18 LDS instructions that lengthen part 1 by 1 000 cycles)
    if (VAR & 128) __builtin_amdgcn_s_setprio(3);
    // LDS reads in the order of need: x_j (for u, whose transposition is the longest chain), x_i (gW), the gradient rows
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xj[0][r] = xs0[r * RS + oj];                     // x_j[b = 4g + r][e = c]
      xj[1][r] = xs1[r * RS + oj];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xi[0][r] = xs0[r * RS + oi];                     // x_i[b = 4g + r][d = c]
      xi[1][r] = xs1[r * RS + oi];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gj[0][r] = gx0[r * RS + oj];
      gj[1][r] = gx1[r * RS + oj];
      gi_[0][r] = gx0[r * RS + oi];
      gi_[1][r] = gx1[r * RS + oi];
    }
    // u = G (.) x_j in the accumulator layout: the A operand of gW as it is, of gX_i after a transposition in LDS
    float u0[4], u1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      u0[r] = G0[r] * xj[0][r];
      u1[r] = G1[r] * xj[1][r];
      tb0[(4 * g + r) * 17 + c] = u0[r];
      tb1[(4 * g + r) * 17 + c] = u1[r];
    }
    f32x4 aw0 = {0.f, 0.f, 0.f, 0.f}, aw1 = {0.f, 0.f, 0.f, 0.f};       // gW_k[e = 4g + r][d = c], per sample half
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      aw0 = mfma16(u0[s], xi[0][s], aw0);
      aw1 = mfma16(u1[s], xi[1][s], aw1);
    }
    float ua0[4], ua1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ua0[s] = tb0[c * 17 + 4 * g + s];      // u[b = c][e = 4g + s]
      ua1[s] = tb1[c * 17 + 4 * g + s];
    }
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};   // (u W)[b = 4g + r][d = c]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      v0 = mfma16(ua0[s], wtr[s], v0);
      v1 = mfma16(ua1[s], wtr[s], v1);
    }
    // the pair's own tile of the next group, into the registers the last MFMAs above have just read (loading them any
    // earlier makes the loop-carried copy of the old values wait for every load in flight at the loop's head).  Opaque
    // use + compiler-level memory barrier: the loads are otherwise hoisted above the MFMAs at IR level, where
    // sched_barrier does not exist.
    asm volatile("" : "+v"(v0), "+v"(v1) : : "memory");
    if (!(VAR & 4)) load_tile(nx);
    __builtin_amdgcn_sched_barrier(0);
    if (live) {
      if (!(VAR & 8)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gx0[r * RS + oj] = gj[0][r] + G0[r] * t0[r];
          gx1[r * RS + oj] = gj[1][r] + G1[r] * t1[r];
          gx0[r * RS + oi] = gi_[0][r] + v0[r];
          gx1[r * RS + oi] = gi_[1][r] + v1[r];
        }
      }
      float* dst = part + (tile * P + en.k) * (kD * kD) + (4 * g) * kD + c;
      if (!(VAR & 2) || aw0[0] == 12345.f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) stg_f32(dst + r * kD, aw0[r] + aw1[r]);
      }
      if ((VAR & 8) && gj[0][0] + gj[1][1] + gi_[0][2] + gi_[1][3] + t0[0] + t1[1] + v0[2] + v1[3] == 12345.f) gx0[lane] = 1.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (VAR & 16) {
      const uint64_t c3 = __builtin_amdgcn_s_memtime();
      tG += c1 - c0; tB += c2 - c1; tD += c3 - c2;
    }
    if (VAR & 128) __builtin_amdgcn_s_setprio(0);
    if (VAR & 64) {
      cur = nxt;
      nxt = entry(gi + 2);
    }
    if (!(VAR & 1)) lds_barrier();     // this wave's gradient rows are in LDS; the other half passes to ITS part 2
  };
  // this wave's leading live groups, 64 groups per ballot (a loop of dependent LDS reads took 2 us of the prologue)
  int n_live = n_groups;
  for (int g0 = 0; g0 < n_groups && n_live == n_groups; g0 += 64) {
    const int gq = g0 + lane;
    const bool idle = gq < n_groups && sch[4 * (kNW * (gq < n_groups ? gq : 0) + wv)] < 0;
    const unsigned long long m = __ballot(idle);
    if (m) n_live = g0 + __builtin_ctzll(m);
  }
  if (VAR & 16) k1 = __builtin_amdgcn_s_memtime();
  // half a group apart: read-modify-writes of the two halves never share an interval between two barriers, and each
  // half's four pairs are field-disjoint
  if (wv >= kNW / 2 && !(VAR & 1)) lds_barrier();
  int gi = 0;
  for (; gi < n_live; ++gi) group(std::true_type{}, gi);
  for (; gi < n_groups; ++gi) group(std::false_type{}, gi);
  if (VAR & 16) k2 = __builtin_amdgcn_s_memtime();
  if (wv < kNW / 2 && !(VAR & 1)) lds_barrier();
  __syncthreads();
  if ((VAR & 16) && lane == 0) {       // per-wave cycle sums, in the first floats of this tile's partials (diagnostics only)
    float* d = part + tile * P * (kD * kD) + 8 * wv;
    d[0] = static_cast<float>(tG); d[1] = static_cast<float>(tB); d[2] = static_cast<float>(tD);
    d[3] = static_cast<float>(n_groups);
    d[4] = static_cast<float>(k1 - k0); d[5] = static_cast<float>(k2 - k1);
    d[6] = static_cast<float>(__builtin_amdgcn_s_memtime() - k2);
  }
  // the two gradient tiles leave in dwordx4 pieces (RS and W are multiples of 4)
  const int w4 = W >> 2;
  for (int e = tid; e < 2 * kSB * w4; e += 64 * kNW) {
    const int r = e / w4, q = e - r * w4;      // r < 32: gx1 follows gx0 in LDS
    if (b0 + r < B)
      *(DCTR_GLOBAL f32x4*)(gX + static_cast<int64_t>(b0 + r) * W + 4 * q) =
          *reinterpret_cast<const f32x4*>(gx0 + r * RS + 4 * q);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Forward: h = act(W0 [ pairs(V) | pairs(E) | dense ] + b0) with the pairs made where they are consumed.
//   t^T[e][b] = sum_d W_k[e][d] x_i[b][d]     (A = the weight tile, B = x_i: the product lands in the A-operand layout
//   p[b][e]   = t[b][e] x_j[b][e]              of the next MFMA -- lane (g, c) holds p[b = c][e = 4g + s] -- no transposition)
//   y[b][h]  += sum_e p[b][e] W0[h][16 k + e]  8 column blocks x 4 steps x 2 inputs = 64 MFMAs per pair
// A workgroup owns 16 samples, a wave every eighth pair (output order) and its own partial y [16, 128] in registers; the
// eight partials meet once, in LDS, in wave order (fixed order: bit-reproducible).  No barrier and no LDS write inside
// the pair loop: the two waves of a SIMD drift apart by themselves.  The product slab x is still WRITTEN (one dwordx4
// per lane, pair and input) for the weight gradient's GEMM of the backward -- but never read by this launch, and the
// [B, 10 413] x [10 413, 128] library GEMM of the forward is gone.
template <int FC, int NWF>
__global__ __launch_bounds__(64 * NWF) void k_bilinear_fwd_wide(const float* __restrict__ E, int64_t lde,
                                                                const float* __restrict__ V, int64_t ldv,
                                                                const float* __restrict__ Wf,
                                                                const int32_t* __restrict__ sched_k, int P, int F, int B,
                                                                const f32x4* __restrict__ Wpk,
                                                                const float* __restrict__ dense, int64_t ldd, int n_dense,
                                                                float* __restrict__ x, int64_t ldx,
                                                                float* __restrict__ ypart, int Bp) {
  extern __shared__ __align__(16) float smem[];
  const int RS = FC ? wide_rs(FC) : wide_rs(F), W = FC ? FC * kD : F * kD;
  // blockIdx.y: 0 = the V input (columns [0, 16 P) of the DNN input), 1 = the E input (columns [16 P, 32 P))
  const int pass = blockIdx.y;
  const float* X = pass ? E : V;
  const int64_t ldq = pass ? lde : ldv;
  float* xs0 = smem;               // samples b0 .. b0 + 15
  float* xs1 = xs0 + kSB * RS;     // samples b0 + 16 .. b0 + 31
  int32_t* sch = reinterpret_cast<int32_t*>(xs1 + kSB * RS);   // [P][4] = {i, j, weight index, k}
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int b0 = blockIdx.x * (2 * kSB);
  for (int e = tid; e < P; e += 64 * NWF)
    *reinterpret_cast<i32x4*>(sch + 4 * e) = *(const DCTR_GLOBAL i32x4*)(sched_k + 4 * e);
  if (tid < 2 * kT) {              // (two groups of 256 threads stage the two tiles)
    const int t2 = tid & (kT - 1), half = tid >> 8;
    float* dst = half ? xs1 : xs0;
    const int bs = b0 + half * kSB;
    if ((ldq & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {      // (uniform)
      const int w4 = W >> 2, n4 = kSB * w4;
      for (int e0 = t2; e0 < n4; e0 += 8 * kT) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kT, ec = e < n4 ? e : 0;
          const int r = ec / w4, q4 = ec - r * w4;
          const int rr = bs + r < B ? bs + r : B - 1;
          v[u] = *(const DCTR_GLOBAL f32x4*)(X + static_cast<int64_t>(rr) * ldq + 4 * q4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kT;
          if (e < n4) {
            const int r = e / w4, q4 = e - r * w4;
            *reinterpret_cast<f32x4*>(dst + r * RS + 4 * q4) = (bs + r < B) ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    } else {
      const int n = kSB * W;
      for (int e0 = t2; e0 < n; e0 += 8 * kT) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kT, ec = e < n ? e : 0;
          const int r = ec / W, cc = ec - r * W;
          const int rr = bs + r < B ? bs + r : B - 1;
          v[u] = ldg_f32(X + static_cast<int64_t>(rr) * ldq + cc);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kT;
          if (e < n) {
            const int r = e / W, cc = e - r * W;
            dst[r * RS + cc] = (bs + r < B) ? v[u] : 0.f;
          }
        }
      }
    }
  }
  // the dense columns of the DNN input (fibinet.py:86-87): copied behind the pairs (by the V launch row)
  if (dense && pass == 0)
    for (int e = tid; e < 2 * kSB * n_dense; e += 64 * NWF) {
      const int r = e / n_dense, q = e - r * n_dense;
      if (b0 + r < B)
        stg_f32(x + static_cast<int64_t>(b0 + r) * ldx + 2 * P * kD + q, ldg_f32(dense + static_cast<int64_t>(b0 + r) * ldd + q));
    }
  __syncthreads();
  const int npw = wv < P ? (P - 1 - wv) / NWF + 1 : 0;      // this wave's pairs: k = wv, wv + NWF, ...
  auto entry = [&](int m) {
    const int k = wv + NWF * (m < npw ? m : (npw > 0 ? npw - 1 : 0));
    const i32x4 v = *reinterpret_cast<const i32x4*>(sch + 4 * (k < P ? k : 0));
    PairEnt e;
    e.i = v.x; e.j = v.y; e.wi = v.z; e.k = v.w;
    return e;
  };
  // ring: the pair's 16 columns of W0 (8 dwordx4, shared by the two sample halves), the pair's own weight tile as an A operand
  f32x4 w0[kNQ], wr;
  f32x4 y0[kNQ], y1[kNQ];
#pragma unroll
  for (int q = 0; q < kNQ; ++q) y0[q] = y1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (npw > 0) {
    PairEnt en = entry(0);
    {
      const f32x4* p0 = Wpk + static_cast<int64_t>(pass * P + en.k) * (kNQ * 64) + lane;
#pragma unroll
      for (int q = 0; q < kNQ; ++q) w0[q] = *(const DCTR_GLOBAL f32x4*)(p0 + 64 * q);
      wr = *(const DCTR_GLOBAL f32x4*)(Wf + static_cast<int64_t>(en.wi) * (kD * kD) + c * kD + 4 * g);   // W[e = c][d = 4g + s]
    }
    // x_i as the B operand (x_i[b = c][d = 4g + s]) and x_j in the A-operand layout of the product (x_j[b = c][e = 4g + s])
    f32x4 a0 = *reinterpret_cast<const f32x4*>(xs0 + c * RS + en.i * kD + 4 * g);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(xs1 + c * RS + en.i * kD + 4 * g);
    f32x4 j0 = *reinterpret_cast<const f32x4*>(xs0 + c * RS + en.j * kD + 4 * g);
    f32x4 j1 = *reinterpret_cast<const f32x4*>(xs1 + c * RS + en.j * kD + 4 * g);
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};     // t^T[e = 4g + r][b = c]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      t0 = mfma16(wr[s], a0[s], t0);
      t1 = mfma16(wr[s], a1[s], t1);
    }
    f32x4 pa = t0 * j0, pb = t1 * j1;
    // One pair of a wave: 64 MFMAs into y in four steps of two column blocks x two sample halves (an accumulator is hit
    // every fourth MFMA); between the steps, in the shadow of the matrix pipe: the ring's re-loads in place, and the
    // NEXT pair's products -- its LDS operands after step 0, its 8 MFMAs after step 1, its two multiplies after step 2.
    PairEnt nx = entry(1);
    for (int m = 0; m < npw; ++m) {
      // (the schedule entry after the next: read now, needed at the end of this pair -- read at the head of the pair that
      // needs it, its LDS round trip stood in front of the pair's first MFMAs)
      const PairEnt nn = entry(m + 2);
      const f32x4* p0 = Wpk + static_cast<int64_t>(pass * P + nx.k) * (kNQ * 64) + lane;
      // the pair's pieces of the DNN input, for the backward's weight-gradient GEMM.  Unconditional: x holds whole
      // 32-row tiles (rows past B receive zeros) -- a store behind a branch is one the compiler cannot count, and its
      // waits for the ring then fall back to "everything older"
      {
        float* row = x + static_cast<int64_t>(b0 + c) * ldx + (static_cast<int64_t>(pass) * P + en.k) * kD + 4 * g;
        *(DCTR_GLOBAL f32x4*)row = pa;
        *(DCTR_GLOBAL f32x4*)(row + kSB * ldx) = pb;
      }
      wr = *(const DCTR_GLOBAL f32x4*)(Wf + static_cast<int64_t>(nx.wi) * (kD * kD) + c * kD + 4 * g);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 pan = pa, pbn = pb;
#pragma unroll
      for (int q = 0; q < kNQ; q += 2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          y0[q] = mfma16(pa[s], w0[q][s], y0[q]);
          y1[q] = mfma16(pb[s], w0[q][s], y1[q]);
          y0[q + 1] = mfma16(pa[s], w0[q + 1][s], y0[q + 1]);
          y1[q + 1] = mfma16(pb[s], w0[q + 1][s], y1[q + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        w0[q] = *(const DCTR_GLOBAL f32x4*)(p0 + 64 * q);
        w0[q + 1] = *(const DCTR_GLOBAL f32x4*)(p0 + 64 * (q + 1));
        if (q == 0) {
          a0 = *reinterpret_cast<const f32x4*>(xs0 + c * RS + nx.i * kD + 4 * g);
          a1 = *reinterpret_cast<const f32x4*>(xs1 + c * RS + nx.i * kD + 4 * g);
          j0 = *reinterpret_cast<const f32x4*>(xs0 + c * RS + nx.j * kD + 4 * g);
          j1 = *reinterpret_cast<const f32x4*>(xs1 + c * RS + nx.j * kD + 4 * g);
        }
        if (q == 2) {
          t0 = f32x4{0.f, 0.f, 0.f, 0.f};
          t1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            t0 = mfma16(wr[s], a0[s], t0);
            t1 = mfma16(wr[s], a1[s], t1);
          }
        }
        if (q == 4) {
          pan = t0 * j0;
          pbn = t1 * j1;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("" : "+v"(y0[0]), "+v"(y0[1]), "+v"(y0[2]), "+v"(y0[3]), "+v"(y0[4]), "+v"(y0[5]), "+v"(y0[6]), "+v"(y0[7]));
      asm volatile("" : "+v"(y1[0]), "+v"(y1[1]), "+v"(y1[2]), "+v"(y1[3]), "+v"(y1[4]), "+v"(y1[5]), "+v"(y1[6]), "+v"(y1[7]));
      en = nx;
      nx = nn;
      pa = pan;
      pb = pbn;
    }
  }
  __syncthreads();                 // every wave is done with the row tiles: their LDS becomes the partials' meeting place
  float* red = smem;               // [waves][16][128 + 4], one sample half at a time
  constexpr int RP = 16 * kNQ + 4;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    if (hf) __syncthreads();
#pragma unroll
    for (int q = 0; q < kNQ; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wv * kSB + 4 * g + r) * RP + 16 * q + c] = hf ? y1[q][r] : y0[q][r];
    __syncthreads();
    // this input's share of the pre-activation, waves in order (fixed order); k_wide_fwd_finish adds the two inputs' shares
    for (int o = tid; o < kSB * 16 * kNQ; o += 64 * NWF) {
      const int b = o / (16 * kNQ), h = o - b * (16 * kNQ);
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NWF; ++w) v += red[(w * kSB + b) * RP + h];
      const int bb = b0 + hf * kSB + b;
      if (bb < Bp) stg_f32(ypart + (static_cast<int64_t>(pass) * Bp + bb) * (16 * kNQ) + h, v);
    }
  }
}

// h = act((share of V's pairs + share of E's pairs) + dense columns + bias); Wd: the dense columns' weights, [128][32]
// A thread keeps its hidden unit's 32 dense weights in registers (zeros past n_dense) and walks 8 of the workgroup's 16
// samples, whose dense values it reads as broadcast dwordx4 pieces from LDS (as two LDS reads per product this launch
// was LDS-bound at 10 us).
constexpr int kFinRows = 16, kFinND = 32;
__global__ __launch_bounds__(kT) void k_wide_fwd_finish(const float* __restrict__ ypart, int Bp, int B, int H,
                                                        const float* __restrict__ dense, int64_t ldd, int n_dense,
                                                        const float* __restrict__ Wd, const float* __restrict__ b0v,
                                                        int relu, float* __restrict__ hout, int64_t ldh) {
  __shared__ __align__(16) float dn[kFinRows][kFinND + 4];
  const int tid = threadIdx.x, r0 = blockIdx.x * kFinRows;
  const int nd = n_dense < kFinND ? n_dense : kFinND;       // (host: n_dense <= kFinND)
  const int h = tid & (16 * kNQ - 1), rh = tid >> 7;        // (kT = 256 threads: two samples per pass)
  f32x4 w[kFinND / 4];
#pragma unroll
  for (int j = 0; j < kFinND / 4; ++j) w[j] = *(const DCTR_GLOBAL f32x4*)(Wd + h * kFinND + 4 * j);
  {
    // (unconditional loads on clamped addresses: a predicated load is a branch with its own wait)
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * kT, r = e >> 5, q = e & 31;
      v[u] = nd > 0 ? ldg_f32(dense + static_cast<int64_t>(r0 + r < B ? r0 + r : B - 1) * ldd + (q < nd ? q : 0)) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * kT, r = e >> 5, q = e & 31;
      dn[r][q] = (r0 + r < B && q < nd) ? v[u] : 0.f;
    }
  }
  __syncthreads();
  const float bias = (b0v && h < H) ? ldg_f32(b0v + h) : 0.f;
#pragma unroll
  for (int k = 0; k < kFinRows / 2; ++k) {
    const int r = rh + 2 * k, b = r0 + r;
    const int bc = b < Bp ? b : Bp - 1;
    float v = ldg_f32(ypart + static_cast<int64_t>(bc) * (16 * kNQ) + h) +
              ldg_f32(ypart + (static_cast<int64_t>(Bp) + bc) * (16 * kNQ) + h);
#pragma unroll
    for (int j = 0; j < kFinND / 4; ++j) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(&dn[r][4 * j]);
      v += d.x * w[j].x;
      v += d.y * w[j].y;
      v += d.z * w[j].z;
      v += d.w * w[j].w;
    }
    v += bias;
    if (relu) v = v > 0.f ? v : 0.f;
    if (b < B && h < H) stg_f32(hout + static_cast<int64_t>(b) * ldh + h, v);
  }
}

// gW[pair_w[k]] = sum over the tiles' partials of pair k, fixed order: four slices of the tiles per workgroup, each
// lane 16 loads in flight, the slices combined in slice order.  (One weight per pair: the "interaction" type.)
__global__ __launch_bounds__(1024) void k_wide_reduce_w(const float* __restrict__ part, int tiles, int P,
                                                       const int32_t* __restrict__ pair_w,
                                                       float* __restrict__ gW) {
  __shared__ float red[4][kD * kD];
  const int k = blockIdx.x, el = threadIdx.x & 255, sl = threadIdx.x >> 8;
  const int per = (tiles + 3) / 4, t0 = sl * per, t1 = t0 + per < tiles ? t0 + per : tiles;
  float s = 0.f;
  for (int tb = t0; tb < t1; tb += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = tb + u < t1 ? tb + u : t1 - 1;
      v[u] = ldg_f32(part + (static_cast<int64_t>(t) * P + k) * (kD * kD) + el);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (tb + u < t1) s += v[u];
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0) gW[static_cast<int64_t>(ldg_i32(pair_w + k)) * (kD * kD) + el] = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
}

size_t wide_pack_floats(int P) { return static_cast<size_t>(2) * P * kNQ * 64 * 4; }

}  // namespace

extern "C" size_t dctr_bilinear_wide_bwd_workspace_floats(int32_t B, int32_t P) {
  const size_t tiles = 2 * (static_cast<size_t>((B > 0 ? B : 1) + 2 * kSB - 1) / (2 * kSB));
  return wide_pack_floats(P > 0 ? P : 1) + tiles * (P > 0 ? P : 1) * kD * kD;
}

extern "C" int dctr_bilinear_wide_bwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                                      const int32_t* sched4, int32_t n_groups, const int32_t* pair_w, int32_t n_w,
                                      int32_t P, int32_t F, int32_t D, int32_t B, const float* gh, int64_t ld_gh,
                                      const float* W0, int64_t ld_w0, int32_t H, float* gE, float* gV, float* gW,
                                      float* workspace, const float* wpk, dctr_stream_t stream) {
  if (!E || !V || !Wf || !sched4 || !pair_w || !gh || !W0 || !gE || !gV || !gW || !workspace || B < 0 || F < 2 ||
      P <= 0 || n_groups <= 0 || n_w <= 0 || H <= 0)
    return DCTR_EINVAL;
  // one weight per pair, 16-wide embeddings, at most 128 hidden units in dwordx4 pieces
  if (D != kD || n_w != P || H > 16 * kNQ || (H & 3) || (ld_gh & 3) || (reinterpret_cast<uintptr_t>(gh) & 15) ||
      (reinterpret_cast<uintptr_t>(Wf) & 15) || (reinterpret_cast<uintptr_t>(gE) & 15) ||
      (reinterpret_cast<uintptr_t>(gV) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15) ||
      (reinterpret_cast<uintptr_t>(wpk) & 15))
    return DCTR_ENOSUP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) {
    (void)hipMemsetAsync(gW, 0, sizeof(float) * n_w * kD * kD, s);
    return DCTR_OK;
  }
  const size_t lds = static_cast<size_t>(4) * kSB * wide_rs(F) * sizeof(float) +
                     static_cast<size_t>(kNW) * 2 * 16 * 17 * sizeof(float) + static_cast<size_t>(n_groups) * kNW * 16;
  if (lds > 158 * 1024) return DCTR_ENOSUP;
  f32x4* Wpk = reinterpret_cast<f32x4*>(workspace);
  float* part = workspace + wide_pack_floats(P);
  const int KB = 2 * P, tiles = (B + 2 * kSB - 1) / (2 * kSB);
  if (wpk) Wpk = reinterpret_cast<f32x4*>(const_cast<float*>(wpk));      // (packed by the forward of this step)
  else k_wide_pack<false><<<dim3((KB + 3) / 4), dim3(kT), 0, s>>>(W0, ld_w0, H, KB, Wpk, nullptr, 0, nullptr);
#define DCTR_WIDE_F(VAR, FC)                                                                                      \
  do {                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bilinear_bwd_wide<VAR, FC>),                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));                 \
    k_bilinear_bwd_wide<VAR, FC><<<dim3(tiles, 2), dim3(64 * kNW), lds, s>>>(E, ld_e, V, ld_v, Wf, sched4, n_groups, P, F, \
                                                                    B, gh, ld_gh, H, Wpk, gE, gV, part);          \
  } while (0)
  // (26 fields: the Criteo shape of BASELINE.json; any other count takes the same kernel with F read at run time)
#define DCTR_WIDE(VAR)                                                                                            \
  do {                                                                                                            \
    if (F == 26 && !fc_off) DCTR_WIDE_F(VAR, 26);                                                                 \
    else DCTR_WIDE_F(VAR, 0);                                                                                     \
  } while (0)
  bool fc_off = false;
#ifdef DCTR_DIAG
  // timing variants (tools/probes/wide_bwd_probe.py; results are wrong for VAR != 0): DCTR_WIDE_VAR = "2<VAR>",
  // "1<VAR>" = the same with F read at run time
  const char* e = getenv("DCTR_WIDE_VAR");
  fc_off = e && e[0] == '1';
  const int var = e && e[0] && e[1] ? atoi(e + 1) : 0;
  if (var == 15) DCTR_WIDE(15);        // (every other single variant was measured while the kernel was built:
  else if (var == 16) DCTR_WIDE(16);   //  profiles/r06_bilinear_wide_variants.txt; instantiate them here to repeat that)
  else DCTR_WIDE(0);
#else
  DCTR_WIDE(0);
#endif
#undef DCTR_WIDE
#undef DCTR_WIDE_F
  k_wide_reduce_w<<<dim3(P), dim3(1024), 0, s>>>(part, 2 * tiles, P, pair_w, gW);
  return launch_status();
}

extern "C" size_t dctr_bilinear_wide_pack_floats(int32_t P) { return wide_pack_floats(P > 0 ? P : 1); }

extern "C" size_t dctr_bilinear_wide_fwd_workspace_floats(int32_t B, int32_t P) {
  const size_t Bp = (static_cast<size_t>(B > 0 ? B : 1) + 2 * kSB - 1) / (2 * kSB) * (2 * kSB);
  return wide_pack_floats(P > 0 ? P : 1) + 2 * Bp * 16 * kNQ + 16 * kNQ * 32;
}

extern "C" int dctr_bilinear_wide_fwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                                      const int32_t* sched_k, int32_t P, int32_t F, int32_t D, int32_t B,
                                      const float* dense, int64_t ld_d, int32_t n_dense, const float* W0, int64_t ld_w0,
                                      int32_t H, const float* b0, int32_t relu, float* x, int64_t ld_x, float* h,
                                      int64_t ld_h, float* workspace, float* wpk_bwd, dctr_stream_t stream) {
  if (!E || !V || !Wf || !sched_k || !W0 || !x || !h || !workspace || B < 0 || F < 2 || P <= 0 || H <= 0 || n_dense < 0 ||
      (n_dense > 0 && !dense))
    return DCTR_EINVAL;
  if (D != kD || H > 16 * kNQ || n_dense > kFinND || (ld_x & 3) || (reinterpret_cast<uintptr_t>(x) & 15) ||
      (reinterpret_cast<uintptr_t>(Wf) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15) ||
      (reinterpret_cast<uintptr_t>(wpk_bwd) & 15))
    return DCTR_ENOSUP;
  if (B == 0) return DCTR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t lds = static_cast<size_t>(2) * kSB * wide_rs(F) * sizeof(float) + static_cast<size_t>(P) * 16;
  constexpr int NWF = 8;           // (12 = three waves per SIMD, 150 registers per lane allow it: 135.1 us against 128.5)
  const size_t lds_red = static_cast<size_t>(NWF) * kSB * (16 * kNQ + 4) * sizeof(float);
  if (lds < lds_red) lds = lds_red;
  if (lds > 158 * 1024) return DCTR_ENOSUP;
  f32x4* Wpk = reinterpret_cast<f32x4*>(workspace);
  float* ypart = workspace + wide_pack_floats(P);
  const int KB = 2 * P, tiles = (B + 2 * kSB - 1) / (2 * kSB), Bp = tiles * 2 * kSB;
  float* Wd = ypart + static_cast<size_t>(2) * Bp * 16 * kNQ;
  k_wide_pack<true><<<dim3((KB + 3) / 4 + 1), dim3(kT), 0, s>>>(W0, ld_w0, H, KB, Wpk, Wd, n_dense,
                                                               reinterpret_cast<f32x4*>(wpk_bwd));
#define DCTR_WIDE_FWD(FC)                                                                                         \
  do {                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bilinear_fwd_wide<FC, NWF>),                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));                 \
    k_bilinear_fwd_wide<FC, NWF><<<dim3(tiles, 2), dim3(64 * NWF), lds, s>>>(E, ld_e, V, ld_v, Wf, sched_k, P, F, B, Wpk, \
                                                                       n_dense > 0 ? dense : nullptr, ld_d, n_dense, x, \
                                                                       ld_x, ypart, Bp);                          \
  } while (0)
  if (F == 26) DCTR_WIDE_FWD(26);
  else DCTR_WIDE_FWD(0);
#undef DCTR_WIDE_FWD
  k_wide_fwd_finish<<<dim3((B + kFinRows - 1) / kFinRows), dim3(kT), 0, s>>>(
      ypart, Bp, B, H, n_dense > 0 ? dense : nullptr, ld_d, n_dense, Wd, b0, relu, h, ld_h);
  return launch_status();
}

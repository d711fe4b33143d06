// fp32 MFMA issue-rate microbenchmark for gfx950: how many core cycles does one v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32 cost a wave, as a function of (a) independent accumulators, (b) waves per SIMD,
// (c) VALU / global loads issued between the MFMAs?  Core clock from s_memtime against the 100 MHz s_memrealtime.
// Decides what the tower weight-gradient kernel (k_mlp_wgrad) can reach.
//   hipcc --offload-arch=gfx950 -O3 -o mfmabench mfmabench.hip && ./mfmabench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Stamp { unsigned long long core, wall; };

// MODE 0: 32x32x2, NACC accumulators, operands in registers
// MODE 1: 16x16x4, NACC accumulators
// MODE 2: 32x32x2, 4 accumulators, every group of 4 MFMAs preceded by two 8-byte global loads (the wgrad pattern),
//         loads consumed LOOK groups later
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k_mfma(const float* __restrict__ src, int ld, int iters, float* sink,
                                              Stamp* stamps, float base, float slope) {
  const int lane = threadIdx.x & 63;
  // operand values: base = 1, slope = 1e-3 is "live" data; base = slope = 0 multiplies zeros -- the clock the chip holds
  // under MFMA load depends on the data's switching activity (MI355X_MICROARCH.md, DVFS give-back)
  float a = base + lane * slope, b = base - lane * slope;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  float out = 0.f;
  if (MODE == 0) {
    f32x16 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; it += 8 * NACC) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) out += c[i][0] + c[i][15];
  } else if (MODE == 1) {
    f32x4 c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[i][r] = 0.f;
    // (round 3: unrolled -- the rolled loop carried an s_nop 7 and accumulator moves per 4-8 MFMAs and read 40-48
    // cycles per 16x16x4 where the pipe issues one every 32)
    for (int it = 0; it < iters; it += 8 * NACC) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) out += c[i][0] + c[i][3];
  } else {
    constexpr int LOOK = 6;
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    const int p = lane >> 5, jl = lane & 31;
    const float* base = src + (size_t)blockIdx.x * 64 + 2 * jl + (size_t)p * ld + (size_t)(threadIdx.x >> 6) * 128 * ld;
    f32x2 ra[LOOK], rb[LOOK];
#pragma unroll
    for (int d = 0; d < LOOK - 1; ++d) {
      ra[d] = *(const f32x2*)(base + (size_t)(2 * d) * ld);
      rb[d] = *(const f32x2*)(base + (size_t)(2 * d) * ld + 4096);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int g0 = 0; g0 < iters / 4; g0 += LOOK) {
#pragma unroll
      for (int dd = 0; dd < LOOK; ++dd) {
        const int g = g0 + dd + LOOK - 1;
        const int row = (2 * g) & 127;
        ra[(dd + LOOK - 1) % LOOK] = *(const f32x2*)(base + (size_t)row * ld);
        rb[(dd + LOOK - 1) % LOOK] = *(const f32x2*)(base + (size_t)row * ld + 4096);
        __builtin_amdgcn_sched_barrier(0);
        c[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[dd].x, rb[dd].x, c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[dd].x, rb[dd].y, c[1], 0, 0, 0);
        c[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[dd].y, rb[dd].x, c[2], 0, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[dd].y, rb[dd].y, c[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out += c[i][0] + c[i][15];
  }
  asm volatile("" : "+v"(out));
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (lane == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[w].core = c1 - c0;
    stamps[w].wall = w1 - w0;
  }
  if (out == 12345.678f) *sink = out;
}

template <int MODE, int NACC>
void run(const char* name, int blocks, int threads, int iters, const float* src, int ld, float* sink, Stamp* d_st,
         float base = 1.f, float slope = 1e-3f) {
  const int nw = blocks * threads / 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  k_mfma<MODE, NACC><<<blocks, threads>>>(src, ld, iters, sink, d_st, base, slope);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_mfma<MODE, NACC><<<blocks, threads>>>(src, ld, iters, sink, d_st, base, slope);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Stamp> st(nw);
  CK(hipMemcpy(st.data(), d_st, nw * sizeof(Stamp), hipMemcpyDeviceToHost));
  double core = 0, wall = 0;
  for (auto& s : st) { core += s.core; wall += s.wall; }
  core /= nw; wall /= nw;
  const double flop_per = MODE == 1 ? 2048.0 : 4096.0;
  printf("{\"case\": \"%s\", \"blocks\": %d, \"waves_per_block\": %d, \"mfma_per_wave\": %d, \"core_cycles_per_mfma\": %.1f, "
         "\"ns_per_mfma\": %.2f, \"core_clock_ghz\": %.3f, \"kernel_us\": %.1f, \"tflops\": %.1f}\n",
         name, blocks, threads / 64, iters, core / iters, wall * 10.0 / iters, core / (wall * 10.0), ms * 1e3,
         flop_per * iters * nw / (ms * 1e-3) / 1e12);
}

int main() {
  const int ld = 8192 + 64;
  float *src, *sink;
  Stamp* st;
  CK(hipMalloc(&src, (size_t)ld * 1024 * 4 + (1 << 20)));
  CK(hipMemset(src, 0, (size_t)ld * 1024 * 4 + (1 << 20)));
  CK(hipMalloc(&sink, 4));
  CK(hipMalloc(&st, sizeof(Stamp) * 8192));
  const int it = 4096;
  run<0, 1>("32x32x2 1 acc, 1 wave/SIMD", 256, 256, it, src, ld, sink, st);
  run<0, 2>("32x32x2 2 acc, 1 wave/SIMD", 256, 256, it, src, ld, sink, st);
  run<0, 4>("32x32x2 4 acc, 1 wave/SIMD", 256, 256, it, src, ld, sink, st);
  run<0, 4>("32x32x2 4 acc, 2 waves/SIMD", 512, 256, it, src, ld, sink, st);
  run<0, 4>("32x32x2 4 acc, 1 wave on 1 SIMD per CU", 256, 64, it, src, ld, sink, st);
  run<1, 1>("16x16x4 1 acc, 1 wave/SIMD", 256, 256, it, src, ld, sink, st);
  run<1, 4>("16x16x4 4 acc, 1 wave/SIMD", 256, 256, it, src, ld, sink, st);
  run<1, 8>("16x16x4 8 acc, 1 wave/SIMD", 256, 256, it, src, ld, sink, st);
  run<1, 8>("16x16x4 8 acc, 2 waves/SIMD", 512, 256, it, src, ld, sink, st);
  run<2, 4>("32x32x2 4 acc + 2 dwordx2 loads per 4 MFMA (L2-resident), 1 wave/SIMD", 256, 256, 1024, src, ld, sink, st);
  run<2, 4>("32x32x2 4 acc + 2 dwordx2 loads per 4 MFMA (L2-resident), 2 waves/SIMD", 512, 256, 1024, src, ld, sink, st);
  // the same pipes on all-zero operands: the instruction rate is unchanged, the clock is not
  run<0, 4>("ZERO operands: 32x32x2 4 acc, 1 wave/SIMD", 256, 256, it, src, ld, sink, st, 0.f, 0.f);
  run<1, 8>("ZERO operands: 16x16x4 8 acc, 2 waves/SIMD", 512, 256, it, src, ld, sink, st, 0.f, 0.f);
  // long run (8x the MFMAs): does the clock sag further once the power budget bites?
  run<0, 4>("32x32x2 4 acc, 1 wave/SIMD, 32768 MFMAs per wave", 256, 256, 8 * it, src, ld, sink, st);
  return 0;
}

// sync.hip -- the waiting half of the device-side dependency between the two queues of a train step (gfx950).
//
// include/dctr.h (dctr_step_wait) has the protocol; common.hpp (step_signal) the signalling half, called by k_embed_fwd
// and k_mlp_train.  Measured on MI355X (tools/micro/hopbench.hip, profiles/r03_step_topologies.json): producer's last
// workgroup -> consumer's first workgroup 4.6 us through the word in memory + this kernel, 11.3 us through a hipGraph
// edge between two queues.
#include "common.hpp"

using namespace dctr;

namespace {

__global__ __launch_bounds__(64) void k_step_wait(int32_t* sync, int signal, unsigned long long timeout_ticks) {
  if (threadIdx.x != 0) return;
  int32_t* gen = sync + 4 * signal;
  const int32_t epoch = gen[1] + 1;          // (this kernel is the epoch counter's only reader and writer)
  gen[1] = epoch;
  const unsigned long long t0 = wall_clock64();
  sync[16 + 2 * signal] = static_cast<int32_t>(t0);       // (when this wait began / ended: tools/step_hops.py)
  for (;;) {
    const int32_t g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g - epoch >= 0) {                    // (wrap-safe: generations and epochs advance together)
      sync[17 + 2 * signal] = static_cast<int32_t>(wall_clock64());
      return;
    }
    if (wall_clock64() - t0 > timeout_ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_fetch_or(sync + DCTR_SYNC_ERR, 1 << signal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the signalling half as a launch of its own: behind a producer kernel on the producer's queue (the kernel boundary in front
// of it has written the producer's stores back), for producers that cannot signal from inside -- the embedding update
// re-reads rows it has just written within one launch, write-through stores would break that
__global__ __launch_bounds__(64) void k_step_signal(int32_t* sync, int signal) {
  if (threadIdx.x != 0) return;
  int32_t* gen = sync + 4 * signal;
  gen[3] = static_cast<int32_t>(wall_clock64());
  __hip_atomic_fetch_add(gen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- direct exchange between the ranks of a table-sharded job (deepctr_torch/parallel.py DirectExchange) -------------
// Every rank owns receive buffers other ranks' copy engines write into (IPC-mapped device memory) and one arrival word per
// (exchange, sender).  A step's exchanges are numbered by a counter every rank keeps on its own device and advances in
// lock-step; a sender POSTS the counter's value into its word on every receiver behind its copies (stream order: the
// copies have completed), a receiver WAITS until all n words have reached its own counter.  Words only grow: a sender
// that is a whole exchange ahead cannot be mistaken for the current one, and no word is ever reset.
__global__ __launch_bounds__(64) void k_exch_post(int32_t* const* __restrict__ peer_words, int n, int my_index,
                                                  const int32_t* __restrict__ step) {
  const int r = threadIdx.x;
  if (r >= n) return;
  const int32_t v = __hip_atomic_load(step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(peer_words[r] + my_index, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void k_exch_wait(const int32_t* __restrict__ words, int n, const int32_t* __restrict__ step,
                                                  unsigned long long timeout_ticks, int32_t* err) {
  const int r = threadIdx.x;
  const int32_t want = __hip_atomic_load(step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long t0 = wall_clock64();
  bool ok = r >= n;
  while (!ok) {
    const int32_t v = __hip_atomic_load(words + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    ok = v - want >= 0;
    if (!ok) {
      if (wall_clock64() - t0 > timeout_ticks) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (!ok && err) __hip_atomic_fetch_or(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// post + wait (+ the counter's advance behind the step's last exchange) as ONE launch: a kernel boundary costs ~4.7 us in
// a hipGraph and the direct-exchange step had nine of them for its three exchanges (profiles/r04_sharded_direct_*)
__global__ __launch_bounds__(64) void k_exch_sync(int32_t* const* __restrict__ peer_words, const int32_t* __restrict__ words,
                                                  int n, int my_index, int32_t* step, int advance,
                                                  unsigned long long timeout_ticks, int32_t* err) {
  const int r = threadIdx.x;
  const int32_t want = __hip_atomic_load(step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool ok = r >= n;
  if (r < n) __hip_atomic_store(peer_words[r] + my_index, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t0 = wall_clock64();
  while (!ok) {
    const int32_t v = __hip_atomic_load(words + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    ok = v - want >= 0;
    if (!ok) {
      if (wall_clock64() - t0 > timeout_ticks) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (!ok && err) __hip_atomic_fetch_or(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (every lane has read `want` before any lane gets here: one wave, the loads above are complete)
  __builtin_amdgcn_wave_barrier();
  if (advance && r == 0) __hip_atomic_store(step, want + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// *dst = the device's 100 MHz wall clock now: a one-thread launch between two kernels of a queue dates the boundary (bench.py's
// graph-replayed kernel durations: HIP events cannot be read back out of a hipGraph replay)
__global__ __launch_bounds__(64) void k_stamp(unsigned long long* dst) {
  if (threadIdx.x == 0) *dst = wall_clock64();
}

__global__ void k_exch_next(int32_t* step) {
  if (threadIdx.x == 0) *step = *step + 1;
}

// dst[i] = sum over the n ranks' slabs, in rank order (every rank computes the same bits): the dense gradients' all-reduce
// as all-gather by copy engines + this local sum -- or, tbl != NULL, as a PULL: rank r's slab is read where it lies
// ((const float*)tbl[r], peer memory) and nothing is copied.  store == 0: the sum is only consumed by the optimizer step
// (dst then just names the gradients' positions in the slab; it may alias a source that peers are still reading).
__global__ __launch_bounds__(256) void k_sum_ranks(float* __restrict__ dst, const float* __restrict__ src,
                                                   const uint64_t* __restrict__ tbl, int n_ranks, int64_t n, int64_t ld,
                                                   DenseStepDev S, int store) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const auto slab = [&](int r) -> const float* { return tbl ? reinterpret_cast<const float*>(tbl[r]) : src + r * ld; };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i + 3 < n) {
    for (int r = 0; r < n_ranks; ++r) acc += *(const DCTR_GLOBAL f32x4*)(slab(r) + i);
    if (store) *(DCTR_GLOBAL f32x4*)(dst + i) = acc;
#pragma unroll
    for (int c = 0; c < 4; ++c) dense_step_apply(S, dst + i + c, acc[c]);
  } else {
    for (int64_t k = i; k < n; ++k) {
      float a = 0.f;
      for (int r = 0; r < n_ranks; ++r) a += ldg_f32(slab(r) + k);
      if (store) stg_f32(dst + k, a);
      dense_step_apply(S, dst + k, a);
    }
  }
}

}  // namespace

// The exchange's copies and its set-up as plain HIP calls (no torch cross-device logic inside a stream capture):
// dctr_copy_async = hipMemcpyAsync(device to device) on the caller's stream -- the destination may be memory of another
// device mapped through IPC; dctr_enable_peer_access(d) lets the CURRENT device's kernels and copies reach device d's
// memory (hipDeviceEnablePeerAccess; "already enabled" is success; d == current device is a no-op).
extern "C" int dctr_copy_async(void* dst, const void* src, size_t bytes, dctr_stream_t stream) {
  if (!dst || !src) return DCTR_EINVAL;
  if (bytes == 0) return DCTR_OK;
  return hip_status(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
}

extern "C" int dctr_enable_peer_access(int32_t peer_device) {
  int cur = -1;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return hip_status(e);
  if (peer_device < 0) return DCTR_EINVAL;
  if (peer_device == cur) return DCTR_OK;
  int can = 0;
  e = hipDeviceCanAccessPeer(&can, cur, peer_device);
  if (e != hipSuccess) return hip_status(e);
  if (!can) return DCTR_ENOSUP;
  e = hipDeviceEnablePeerAccess(peer_device, 0);
  if (e == hipErrorPeerAccessAlreadyEnabled) {
    (void)hipGetLastError();
    return DCTR_OK;
  }
  return hip_status(e);
}

extern "C" int dctr_exchange_post(int32_t* const* peer_words, int32_t n, int32_t my_index, const int32_t* step,
                                  dctr_stream_t stream) {
  if (!peer_words || !step || n <= 0 || n > 64 || my_index < 0) return DCTR_EINVAL;
  k_exch_post<<<dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream)>>>(peer_words, n, my_index, step);
  return launch_status();
}

extern "C" int dctr_exchange_wait(const int32_t* words, int32_t n, const int32_t* step, int32_t timeout_us, int32_t* err,
                                  dctr_stream_t stream) {
  if (!words || !step || n <= 0 || n > 64 || timeout_us <= 0) return DCTR_EINVAL;
  k_exch_wait<<<dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream)>>>(
      words, n, step, static_cast<unsigned long long>(timeout_us) * 100ull, err);
  return launch_status();
}

extern "C" int dctr_exchange_next(int32_t* step, dctr_stream_t stream) {
  if (!step) return DCTR_EINVAL;
  k_exch_next<<<dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream)>>>(step);
  return launch_status();
}

extern "C" int dctr_sum_ranks(float* dst, const float* src, const uint64_t* src_tbl, int32_t n_ranks, int64_t n, int64_t ld,
                              const dctr_dense_step_t* step, int32_t store, dctr_stream_t stream) {
  if (!dst || (!src && !src_tbl) || n_ranks <= 0 || n < 0 || (!src_tbl && ld < n)) return DCTR_EINVAL;
  if (!store && !step) return DCTR_EINVAL;      // a sum nobody receives
  if ((!src_tbl && ld % 4 != 0) || reinterpret_cast<uintptr_t>(dst) % 16 != 0 || reinterpret_cast<uintptr_t>(src) % 16 != 0)
    return DCTR_EALIGN;
  if (n == 0) return DCTR_OK;
  k_sum_ranks<<<dim3(static_cast<unsigned>((n / 4 + 256) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      dst, src, src_tbl, n_ranks, n, ld, dense_step_dev(step), store);
  return launch_status();
}

extern "C" int dctr_exchange_sync(int32_t* const* peer_words, const int32_t* words, int32_t n, int32_t my_index,
                                  int32_t* step, int32_t advance, int32_t timeout_us, int32_t* err, dctr_stream_t stream) {
  if (!peer_words || !words || !step || n <= 0 || n > 64 || my_index < 0 || timeout_us <= 0) return DCTR_EINVAL;
  k_exch_sync<<<dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream)>>>(
      peer_words, words, n, my_index, step, advance, static_cast<unsigned long long>(timeout_us) * 100ull, err);
  return launch_status();
}

extern "C" int dctr_stamp(uint64_t* dst, dctr_stream_t stream) {
  if (!dst) return DCTR_EINVAL;
  k_stamp<<<dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream)>>>(reinterpret_cast<unsigned long long*>(dst));
  return launch_status();
}

extern "C" int dctr_step_signal(int32_t* sync, int32_t signal, dctr_stream_t stream) {
  if (!sync || signal < 0 || signal > 2) return DCTR_EINVAL;
  k_step_signal<<<dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream)>>>(sync, signal);
  return launch_status();
}

extern "C" int dctr_step_wait(int32_t* sync, int32_t signal, int32_t timeout_us, dctr_stream_t stream) {
  if (!sync || signal < 0 || signal > 2 || timeout_us <= 0) return DCTR_EINVAL;
  k_step_wait<<<dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream)>>>(
      sync, signal, static_cast<unsigned long long>(timeout_us) * 100ull);      // s_memrealtime: 100 MHz
  return launch_status();
}

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

}  // namespace

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

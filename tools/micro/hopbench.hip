// What does a dependency between two kernels cost on gfx950, by mechanism?  (round 3: the DeepFM step's kernels add up to
// ~75 us on their critical chain and the step takes 96-99 us; the rest is dependency latency.)
//   hipcc --offload-arch=gfx950 -O3 -o hopbench hopbench.hip && ./hopbench
// Every kernel busy-waits a fixed time on s_memrealtime (100 MHz, one clock for the whole device) and stamps its first
// workgroup's start / last workgroup's end, so gaps between kernels are read directly from the stamps.
//   S1  one graph, two streams:  A: K1 -> K3,  B: (event) K2 after K1, joined          -> cross-queue edge, graph boundary
//   S2  one graph, one stream:   K1 -> K2 -> K3                                        -> same-queue edge, graph boundary
//   S3  two graphs on two streams, never joined; K1's workgroups count themselves done in a flag word and a one-wave
//       spinner kernel in front of K2 polls it                                         -> flag hop
//   S4  as S1 but plain stream launches (no graph)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIters = 64, kKernels = 8;
struct Stamps {
    unsigned long long t[kKernels][kIters][2];
    unsigned int iter[kKernels], done[kKernels];
    unsigned int flag[4];      // [0]: workgroups of K1 finished, ever   [1]: spinner epochs   [2]: spinner time-outs
};

__device__ __forceinline__ void enter(Stamps* S, int kid, unsigned& it) {
    it = __hip_atomic_load(&S->iter[kid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % kIters;
    if (threadIdx.x == 0) atomicMin(&S->t[kid][it][0], wall_clock64());
}
__device__ __forceinline__ void leave(Stamps* S, int kid, unsigned it) {
    if (threadIdx.x == 0) {
        atomicMax(&S->t[kid][it][1], wall_clock64());
        if (atomicAdd(&S->done[kid], 1u) == gridDim.x - 1) {
            S->done[kid] = 0;
            __hip_atomic_fetch_add(&S->iter[kid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void k_work(Stamps* S, int kid, int ticks, int set_flag) {
    unsigned it;
    enter(S, kid, it);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < static_cast<unsigned long long>(ticks)) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (set_flag && threadIdx.x == 0) __hip_atomic_fetch_add(&S->flag[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    leave(S, kid, it);
}

// one wave: waits until the producers of this epoch have all counted themselves done (bounded: 20 ms)
__global__ void k_spin(Stamps* S, int kid, unsigned per_epoch) {
    unsigned it;
    enter(S, kid, it);
    if (threadIdx.x == 0) {
        const unsigned epoch = S->flag[1] + 1;
        const unsigned long long t0 = wall_clock64();
        bool ok = false;
        while (wall_clock64() - t0 < 2000000ull) {
            if (__hip_atomic_load(&S->flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= epoch * per_epoch) { ok = true; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        if (!ok) S->flag[2] += 1;
        S->flag[1] = epoch;
    }
    leave(S, kid, it);
}

static void reset(Stamps* d) {
    std::vector<char> z(sizeof(Stamps), 0);
    Stamps* h = reinterpret_cast<Stamps*>(z.data());
    for (int k = 0; k < kKernels; ++k)
        for (int i = 0; i < kIters; ++i) h->t[k][i][0] = ~0ull;
    CK(hipMemcpy(d, h, sizeof(Stamps), hipMemcpyHostToDevice));
}
static double med(std::vector<double> v) {
    if (v.empty()) return -1;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}
// gap from kernel a's end to kernel b's start (b of iteration i + shift), microseconds, median over iterations
static double gap(const Stamps& h, int a, int b, int n, int shift = 0) {
    std::vector<double> v;
    for (int i = 4; i + shift < n; ++i)
        v.push_back((static_cast<double>(h.t[b][i + shift][0]) - static_cast<double>(h.t[a][i][1])) * 0.01);
    return med(v);
}
static double dur(const Stamps& h, int a, int n) {
    std::vector<double> v;
    for (int i = 4; i < n; ++i) v.push_back((static_cast<double>(h.t[a][i][1]) - static_cast<double>(h.t[a][i][0])) * 0.01);
    return med(v);
}

int main() {
    Stamps* S;
    CK(hipMalloc(&S, sizeof(Stamps)));
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t e1, e2, e0;
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    const int G = 256, T = 256, n = 48;
    Stamps h;
    auto body_two_streams = [&]() {
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, A, S, 1, 4000, 0);
        CK(hipEventRecord(e1, A));
        CK(hipStreamWaitEvent(B, e1, 0));
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, B, S, 2, 2000, 0);
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, A, S, 3, 2000, 0);
        CK(hipEventRecord(e2, B));
        CK(hipStreamWaitEvent(A, e2, 0));
    };
    {   // S1
        reset(S);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
        body_two_streams();
        CK(hipStreamEndCapture(A, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < n; ++i) CK(hipGraphLaunch(ge, A));
        CK(hipStreamSynchronize(A));
        CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
        printf("{\"case\": \"S1 graph, two streams\", \"K1_us\": %.2f, \"cross_queue_edge_us\": %.2f, \"same_queue_edge_us\": %.2f, "
               "\"boundary_after_K3_us\": %.2f, \"boundary_after_K2_us\": %.2f}\n",
               dur(h, 1, n), gap(h, 1, 2, n), gap(h, 1, 3, n), gap(h, 3, 1, n, 1), gap(h, 2, 1, n, 1));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    {   // S2
        reset(S);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, A, S, 1, 4000, 0);
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, A, S, 2, 2000, 0);
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, A, S, 3, 2000, 0);
        CK(hipStreamEndCapture(A, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < n; ++i) CK(hipGraphLaunch(ge, A));
        CK(hipStreamSynchronize(A));
        CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
        printf("{\"case\": \"S2 graph, one stream\", \"edge_K1_K2_us\": %.2f, \"edge_K2_K3_us\": %.2f, \"boundary_us\": %.2f}\n",
               gap(h, 1, 2, n), gap(h, 2, 3, n), gap(h, 3, 1, n, 1));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    {   // S3
        reset(S);
        hipGraph_t ga, gb; hipGraphExec_t gea, geb;
        CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, A, S, 1, 4000, 1);
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, A, S, 3, 2000, 0);
        CK(hipStreamEndCapture(A, &ga));
        CK(hipGraphInstantiate(&gea, ga, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(B, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, B, S, 4, static_cast<unsigned>(G));
        hipLaunchKernelGGL(k_work, dim3(G), dim3(T), 0, B, S, 2, 2000, 0);
        CK(hipStreamEndCapture(B, &gb));
        CK(hipGraphInstantiate(&geb, gb, nullptr, nullptr, 0));
        for (int i = 0; i < n; ++i) { CK(hipGraphLaunch(gea, A)); CK(hipGraphLaunch(geb, B)); }
        CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
        printf("{\"case\": \"S3 two graphs, flag + one-wave spinner\", \"flag_hop_K1_end_to_K2_start_us\": %.2f, "
               "\"spinner_end_to_K2_start_us\": %.2f, \"same_queue_edge_K1_K3_us\": %.2f, \"boundary_A_us\": %.2f, "
               "\"boundary_B_K2_to_spinner_us\": %.2f, \"spinner_timeouts\": %u}\n",
               gap(h, 1, 2, n), gap(h, 4, 2, n), gap(h, 1, 3, n), gap(h, 3, 1, n, 1), gap(h, 2, 4, n, 1), h.flag[2]);
    }
    {   // S4
        reset(S);
        for (int i = 0; i < n; ++i) body_two_streams();
        CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        CK(hipMemcpy(&h, S, sizeof(h), hipMemcpyDeviceToHost));
        printf("{\"case\": \"S4 stream launches, two streams\", \"cross_queue_edge_us\": %.2f, \"same_queue_edge_us\": %.2f, "
               "\"boundary_after_K3_us\": %.2f}\n", gap(h, 1, 2, n), gap(h, 1, 3, n), gap(h, 3, 1, n, 1));
    }
    return 0;
}

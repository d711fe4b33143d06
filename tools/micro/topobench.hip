// Which stream-capture recipe gives the DeepFM step's dependency graph the shortest period under hipGraph on gfx950?
// (round 3.)  hipGraph maps captured nodes to hardware queues by topology, not by the stream they were captured on, and a
// dependency that crosses queues costs ~11 us against ~4 us inside a queue (tools/micro/hopbench.hip).  The step:
//     ids -> segments -> apply          gather -> tower -> { wgrad -> reduce , apply }
//     apply(n) -> gather(n+1)           reduce(n) -> tower(n+1)
// Kernels here only busy-wait their measured body times (no memory traffic, no contention between co-running kernels):
// the periods below isolate what the dependency edges cost.
//   hipcc --offload-arch=gfx950 -O3 -o topobench topobench.hip && ./topobench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

enum { IDS, SEG, GATHER, TOWER, WGRAD, REDUCE, APPLY, WGRED, NK };
static const char* kName[NK] = {"ids", "segments", "gather", "tower", "wgrad", "reduce", "apply", "wgrad+reduce"};
static const int kTicks[NK] = {150, 900, 800, 3900, 1800, 250, 1500, 2000};      // body times, 10 ns ticks
constexpr int kMaxIter = 256;
struct Stamps {
    unsigned long long t[NK][kMaxIter][2];
    unsigned int iter[NK], done[NK];
};

__global__ void k_work(Stamps* S, int kid, int ticks) {
    const unsigned it = __hip_atomic_load(&S->iter[kid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % kMaxIter;
    if (threadIdx.x == 0) atomicMin(&S->t[kid][it][0], wall_clock64());
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < static_cast<unsigned long long>(ticks)) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&S->t[kid][it][1], wall_clock64());
        if (atomicAdd(&S->done[kid], 1u) == gridDim.x - 1) {
            S->done[kid] = 0;
            __hip_atomic_fetch_add(&S->iter[kid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static Stamps* gS;
static void launch(hipStream_t s, int kid) { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, gS, kid, kTicks[kid]); }
static hipEvent_t ev() {
    hipEvent_t e;
    CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return e;
}
static void edge(hipStream_t from, hipStream_t to) {      // `to` waits for everything enqueued on `from` so far
    hipEvent_t e = ev();
    CK(hipEventRecord(e, from));
    CK(hipStreamWaitEvent(to, e, 0));
}

static void reset() {
    std::vector<char> z(sizeof(Stamps), 0);
    Stamps* h = reinterpret_cast<Stamps*>(z.data());
    for (int k = 0; k < NK; ++k)
        for (int i = 0; i < kMaxIter; ++i) h->t[k][i][0] = ~0ull;
    CK(hipMemcpy(gS, h, sizeof(Stamps), hipMemcpyHostToDevice));
}

// a recipe enqueues `steps` steps on the streams (inside a capture that began on M)
typedef std::function<void(hipStream_t M, hipStream_t S, hipStream_t T, int steps)> Recipe;

static void run(const char* name, const Recipe& r, int steps_per_graph, int replays) {
    hipStream_t M, S, T;
    CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&T, hipStreamNonBlocking));
    reset();
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(M, hipStreamCaptureModeThreadLocal));
    r(M, S, T, steps_per_graph);
    CK(hipStreamEndCapture(M, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, M));
    CK(hipStreamSynchronize(M));
    static Stamps h;
    CK(hipMemcpy(&h, gS, sizeof(h), hipMemcpyDeviceToHost));
    const int n = steps_per_graph * replays;
    // period: tower start to tower start, inside a graph (not across a replay boundary) and across it
    std::vector<double> in, across;
    for (int i = steps_per_graph; i + 1 < n && i + 1 < kMaxIter; ++i) {
        const double d = (static_cast<double>(h.t[TOWER][i + 1][0]) - static_cast<double>(h.t[TOWER][i][0])) * 0.01;
        ((i + 1) % steps_per_graph == 0 ? across : in).push_back(d);
    }
    auto med = [](std::vector<double> v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double total = (static_cast<double>(h.t[TOWER][std::min(n, kMaxIter) - 1][0]) -
                          static_cast<double>(h.t[TOWER][steps_per_graph][0])) * 0.01 / (std::min(n, kMaxIter) - 1 - steps_per_graph);
    printf("{\"recipe\": \"%s\", \"steps_per_graph\": %d, \"period_in_graph_us\": %.1f, \"period_across_replays_us\": %.1f, "
           "\"period_mean_us\": %.1f, \"step\": {", name, steps_per_graph, med(in), med(across), total);
    const int i0 = steps_per_graph + 1;           // a step in the middle of the second replay
    const double t0 = static_cast<double>(h.t[TOWER][i0][0]);
    for (int k = 0; k < NK; ++k)
        printf("\"%s\": [%.1f, %.1f]%s", kName[k], (static_cast<double>(h.t[k][i0][0]) - t0) * 0.01,
               (static_cast<double>(h.t[k][i0][1]) - t0) * 0.01, k + 1 < NK ? ", " : "");
    printf("}}\n");
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(M)); CK(hipStreamDestroy(S)); CK(hipStreamDestroy(T));
}

int main(int argc, char** argv) {
    CK(hipMalloc(&gS, sizeof(Stamps)));
    const int spg = argc > 1 ? atoi(argv[1]) : 4, rep = 12;

    // V0: today's default ("update_side").  M: gather, tower, wgrad, reduce.  S: ids, segments, [tower] apply.
    Recipe v0 = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        edge(M, S);
        for (int n = 0; n < steps; ++n) {
            launch(S, IDS); launch(S, SEG);
            launch(M, GATHER); launch(M, TOWER);
            edge(M, S);                       // the update may start once the tower is done
            launch(M, WGRAD); launch(M, REDUCE);
            launch(S, APPLY);
            edge(S, M);                       // next gather needs the updated rows
        }
    };
    // V1: weight gradients on S behind the pre-pass, update on M; the update is enqueued FIRST after the tower
    Recipe v1 = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        edge(M, S);
        hipEvent_t red = nullptr;
        for (int n = 0; n < steps; ++n) {
            launch(S, IDS); launch(S, SEG);
            hipEvent_t seg = ev(); CK(hipEventRecord(seg, S));
            launch(M, GATHER);
            if (red) CK(hipStreamWaitEvent(M, red, 0));     // the tower needs the stepped weights
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            CK(hipStreamWaitEvent(M, seg, 0));
            launch(M, APPLY);
            CK(hipStreamWaitEvent(S, tw, 0));
            launch(S, WGRAD); launch(S, REDUCE);
            red = ev(); CK(hipEventRecord(red, S));
        }
        edge(S, M);
    };
    // V2: as V1 but the weight gradients are enqueued before the update
    Recipe v2 = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        edge(M, S);
        hipEvent_t red = nullptr;
        for (int n = 0; n < steps; ++n) {
            launch(S, IDS); launch(S, SEG);
            hipEvent_t seg = ev(); CK(hipEventRecord(seg, S));
            launch(M, GATHER);
            if (red) CK(hipStreamWaitEvent(M, red, 0));
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            CK(hipStreamWaitEvent(S, tw, 0));
            launch(S, WGRAD); launch(S, REDUCE);
            red = ev(); CK(hipEventRecord(red, S));
            CK(hipStreamWaitEvent(M, seg, 0));
            launch(M, APPLY);
        }
        edge(S, M);
    };
    // V3: three streams.  M: gather, tower, apply.  S: ids, segments.  T: wgrad, reduce.
    Recipe v3 = [](hipStream_t M, hipStream_t S, hipStream_t T, int steps) {
        edge(M, S); edge(M, T);
        hipEvent_t red = nullptr;
        for (int n = 0; n < steps; ++n) {
            launch(S, IDS); launch(S, SEG);
            hipEvent_t seg = ev(); CK(hipEventRecord(seg, S));
            launch(M, GATHER);
            if (red) CK(hipStreamWaitEvent(M, red, 0));
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            CK(hipStreamWaitEvent(M, seg, 0));
            launch(M, APPLY);
            CK(hipStreamWaitEvent(T, tw, 0));
            launch(T, WGRAD); launch(T, REDUCE);
            red = ev(); CK(hipEventRecord(red, T));
        }
        edge(S, M); edge(T, M);
    };
    // V4: the pre-pass on the main queue IN FRONT of the gather (serial), update on M, weight gradients on S: only the
    // tower cycle crosses queues
    Recipe v4 = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        edge(M, S);
        hipEvent_t red = nullptr;
        for (int n = 0; n < steps; ++n) {
            launch(M, IDS); launch(M, SEG); launch(M, GATHER);
            if (red) CK(hipStreamWaitEvent(M, red, 0));
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            launch(M, APPLY);
            CK(hipStreamWaitEvent(S, tw, 0));
            launch(S, WGRAD); launch(S, REDUCE);
            red = ev(); CK(hipEventRecord(red, S));
        }
        edge(S, M);
    };
    // V5: everything on one stream (no cross-queue edge at all)
    Recipe v5 = [](hipStream_t M, hipStream_t, hipStream_t, int steps) {
        for (int n = 0; n < steps; ++n) {
            launch(M, IDS); launch(M, SEG); launch(M, GATHER); launch(M, TOWER); launch(M, WGRAD); launch(M, REDUCE);
            launch(M, APPLY);
        }
    };
    // V6: V1 with the pre-pass of step n enqueued behind the weight gradients of step n-1 on S and waited for by the
    // UPDATE only (same as V1) -- but the tower's wait for `reduce` placed in front of the gather
    Recipe v6 = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        edge(M, S);
        hipEvent_t red = nullptr;
        for (int n = 0; n < steps; ++n) {
            launch(S, IDS); launch(S, SEG);
            hipEvent_t seg = ev(); CK(hipEventRecord(seg, S));
            if (red) CK(hipStreamWaitEvent(M, red, 0));
            launch(M, GATHER);
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            CK(hipStreamWaitEvent(M, seg, 0));
            launch(M, APPLY);
            CK(hipStreamWaitEvent(S, tw, 0));
            launch(S, WGRAD); launch(S, REDUCE);
            red = ev(); CK(hipEventRecord(red, S));
        }
        edge(S, M);
    };
    // V7: V1 with the reduction (and the optimizer step) done by the weight-gradient kernel's last workgroups
    Recipe v7 = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        edge(M, S);
        hipEvent_t red = nullptr;
        for (int n = 0; n < steps; ++n) {
            launch(S, IDS); launch(S, SEG);
            hipEvent_t seg = ev(); CK(hipEventRecord(seg, S));
            launch(M, GATHER);
            if (red) CK(hipStreamWaitEvent(M, red, 0));
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            CK(hipStreamWaitEvent(M, seg, 0));
            launch(M, APPLY);
            CK(hipStreamWaitEvent(S, tw, 0));
            launch(S, WGRED);
            red = ev(); CK(hipEventRecord(red, S));
        }
        edge(S, M);
    };
    // V8: V7 with the pre-pass one step ahead: S runs ids/segments of step n+1 in front of the weight gradients of step
    // n (i.e. under tower n), so the tower's wait for the stepped weights covers the pre-pass and the update has ONE parent
    Recipe v8 = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        edge(M, S);
        launch(S, IDS); launch(S, SEG);                 // step 0's pre-pass
        hipEvent_t red = ev(); CK(hipEventRecord(red, S));
        for (int n = 0; n < steps; ++n) {
            launch(M, GATHER);
            CK(hipStreamWaitEvent(M, red, 0));          // stepped weights of n-1 (and, in stream order, the pre-pass of n)
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            launch(M, APPLY);
            if (n + 1 < steps) { launch(S, IDS); launch(S, SEG); }
            CK(hipStreamWaitEvent(S, tw, 0));
            launch(S, WGRED);
            red = ev(); CK(hipEventRecord(red, S));
        }
        edge(S, M);
    };
    // V9: V8 without fusing the reduction
    Recipe v9 = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        edge(M, S);
        launch(S, IDS); launch(S, SEG);
        hipEvent_t red = ev(); CK(hipEventRecord(red, S));
        for (int n = 0; n < steps; ++n) {
            launch(M, GATHER);
            CK(hipStreamWaitEvent(M, red, 0));
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            launch(M, APPLY);
            if (n + 1 < steps) { launch(S, IDS); launch(S, SEG); }
            CK(hipStreamWaitEvent(S, tw, 0));
            launch(S, WGRAD); launch(S, REDUCE);
            red = ev(); CK(hipEventRecord(red, S));
        }
        edge(S, M);
    };
    // V1b: V1 plus the edge the real capture has: the pre-pass of step n waits for the update of step n-1 (it rewrites the
    // bucket workspace the update reads)
    Recipe v1b = [](hipStream_t M, hipStream_t S, hipStream_t, int steps) {
        hipEvent_t red = nullptr;
        for (int n = 0; n < steps; ++n) {
            edge(M, S);
            launch(S, IDS); launch(S, SEG);
            launch(M, GATHER);
            if (red) CK(hipStreamWaitEvent(M, red, 0));
            launch(M, TOWER);
            hipEvent_t tw = ev(); CK(hipEventRecord(tw, M));
            edge(S, M);                                    // the update waits for the side stream's tail = the pre-pass
            launch(M, APPLY);
            CK(hipStreamWaitEvent(S, tw, 0));
            launch(S, WGRAD); launch(S, REDUCE);
            red = ev(); CK(hipEventRecord(red, S));
        }
        edge(S, M);
    };
    run("V1b V1 + update(n-1) -> pre-pass(n)", v1b, spg, rep);
    run("V0 update_side (today)", v0, spg, rep);
    run("V7 V1 + reduction fused into wgrad", v7, spg, rep);
    run("V8 V7 + pre-pass one step ahead", v8, spg, rep);
    run("V9 V1 + pre-pass one step ahead", v9, spg, rep);
    run("V1 wgrad on S behind pre-pass, update first on M", v1, spg, rep);
    run("V2 as V1, wgrad enqueued first", v2, spg, rep);
    run("V3 three streams", v3, spg, rep);
    run("V4 pre-pass serial on M, wgrad on S", v4, spg, rep);
    run("V5 one stream", v5, spg, rep);
    run("V6 as V1, reduce waited for before the gather", v6, spg, rep);
    return 0;
}

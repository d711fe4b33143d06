// Random-row HBM microbenchmark for gfx950: what does a 64-byte row cost against a 128-byte row, read-only and
// read-modify-write, at a saturating launch?  Decides the table layout.
//   hipcc --offload-arch=gfx950 -O3 -o rowbench rowbench.hip && ./rowbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// LPR lanes of 16 B per row: row = LPR*16 bytes.  Each lane group handles rows g, g+stride, ... ; UNROLL rows in flight.
template <int LPR, int UNROLL, bool RMW, int NARR>
__global__ void k_rows(float4* __restrict__ t0, float4* __restrict__ t1, const int* __restrict__ ids, int n, float4* sink) {
    int lane = threadIdx.x % LPR;
    long g = (long)(blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    long G = (long)gridDim.x * blockDim.x / LPR;
    float4 acc = {0, 0, 0, 0};
    for (long r = g; r < n; r += G * UNROLL) {
        float4 v[UNROLL][NARR];
        long row[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            long rr = r + u * G;
            row[u] = rr < n ? ids[rr] : 0;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            v[u][0] = t0[row[u] * LPR + lane];
            if (NARR > 1) v[u][NARR - 1] = t1[row[u] * LPR + lane];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (RMW) {
                if (r + u * G < n) {
                    float4 a = v[u][0];
                    a.x += 1.f; a.y += 1.f; a.z += 1.f; a.w += 1.f;
                    t0[row[u] * LPR + lane] = a;
                    if (NARR > 1) { float4 b = v[u][NARR - 1]; b.x += 1.f; t1[row[u] * LPR + lane] = b; }
                }
            } else {
                acc.x += v[u][0].x; if (NARR > 1) acc.y += v[u][NARR - 1].y;
            }
        }
    }
    if (!RMW && acc.x == 123.456f) sink[0] = acc;
}

__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long G = (long)gridDim.x * blockDim.x;
    for (; i < n; i += G) b[i] = a[i];
}
// U dwordx4 loads in flight per lane before the first store (round 3: the one-load-per-iteration copy above reached
// 4.5 TB/s where MI355X_MICROARCH.md measures 6.3 for a float4 copy -- every ceiling derived from it was ~30 % low)
template <int U>
__global__ void k_copy_u(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long G = (long)gridDim.x * blockDim.x;
    for (; i + (U - 1) * G < n; i += U * G) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * G];
#pragma unroll
        for (int u = 0; u < U; ++u) b[i + u * G] = v[u];
    }
    for (; i < n; i += G) b[i] = a[i];
}

// Round 4 (verdict: the copies above stop at 4.3-4.5 TB/s where the guide measures 6.29 for a float4 copy): every
// workgroup copies ONE contiguous chunk (DRAM pages are walked in order instead of 2^k-strided by the whole grid), U
// dwordx4 in flight per lane, optionally with non-temporal loads / stores (the data is touched once).
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void k_copy_chunk(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    long i = lo + threadIdx.x;
    for (; i + (long)(U - 1) * blockDim.x < hi; i += (long)U * blockDim.x) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) {
                vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(a + i + u * blockDim.x));
                v[u] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                v[u] = a[i + u * blockDim.x];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(vf4{v[u].x, v[u].y, v[u].z, v[u].w}, reinterpret_cast<vf4*>(b + i + u * blockDim.x));
            else b[i + u * blockDim.x] = v[u];
        }
    }
    for (; i < hi; i += blockDim.x) b[i] = a[i];
}
// read-only stream (what a float4 "copy" figure that counts the read side alone would be)
template <int U>
__global__ void k_read_chunk(const float4* __restrict__ a, float4* __restrict__ sink, long n) {
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = lo + threadIdx.x; i + (long)(U - 1) * blockDim.x < hi; i += (long)U * blockDim.x) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].w; }
    }
    if (acc.x == 123.456f) sink[0] = acc;
}

template <class F> float timeit(F f, int it = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / it * 1e3f;
}

int main() {
    const long V = 26L * 1000000;            // rows (26 tables of 1M)
    const int N = 26 * 262144;               // distinct-ish random rows per launch (saturating launch)
    float4 *t64, *s64, *t128, *sink; int *ids, *ids_small;
    CK(hipMalloc(&t64, V * 64)); CK(hipMalloc(&s64, V * 64)); CK(hipMalloc(&t128, V * 128)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(t64, 0, V * 64)); CK(hipMemset(s64, 0, V * 64)); CK(hipMemset(t128, 0, V * 128));
    // a permutation prefix: distinct rows (an RMW of duplicate rows would race; the real kernel de-duplicates)
    std::vector<int> h(N);
    { std::vector<int> perm(V); for (long i = 0; i < V; ++i) perm[i] = (int)i; uint64_t s = 88172645463325252ull;
      for (long i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; long j = i + s % (V - i); std::swap(perm[i], perm[j]); h[i] = perm[i]; } }
    CK(hipMalloc(&ids, N * 4)); CK(hipMemcpy(ids, h.data(), N * 4, hipMemcpyHostToDevice));
    const int NB = 26 * 4096;
    CK(hipMalloc(&ids_small, NB * 4)); CK(hipMemcpy(ids_small, h.data(), NB * 4, hipMemcpyHostToDevice));
    printf("{\n");
    {   long n4 = V * 4; float us = timeit([&] { k_copy<<<4096, 256>>>(t64, s64, n4); });
        printf(" \"stream_copy_1.66GB\": {\"us\": %.1f, \"GBs_rd_plus_wr\": %.0f},\n", us, 2.0 * V * 64 / us / 1e3);
        const int grids[4] = {2048, 4096, 8192, 16384};
        for (int gi = 0; gi < 4; ++gi) {
            const int g = grids[gi];
            float u4 = timeit([&] { k_copy_u<4><<<g, 256>>>(t64, s64, n4); });
            float u8 = timeit([&] { k_copy_u<8><<<g, 256>>>(t64, s64, n4); });
            printf(" \"stream_copy_u4_grid%d\": {\"us\": %.1f, \"GBs_rd_plus_wr\": %.0f},\n", g, u4, 2.0 * V * 64 / u4 / 1e3);
            printf(" \"stream_copy_u8_grid%d\": {\"us\": %.1f, \"GBs_rd_plus_wr\": %.0f},\n", g, u8, 2.0 * V * 64 / u8 / 1e3);
        }
        const int cg[3] = {1024, 4096, 16384};
        for (int gi = 0; gi < 3; ++gi) {
            const int g = cg[gi];
            float c8 = timeit([&] { k_copy_chunk<8, false><<<g, 256>>>(t64, s64, n4); });
            float n8 = timeit([&] { k_copy_chunk<8, true><<<g, 256>>>(t64, s64, n4); });
            float r8 = timeit([&] { k_read_chunk<8><<<g, 256>>>(t64, sink, n4); });
            printf(" \"chunk_copy_u8_grid%d\": {\"us\": %.1f, \"GBs_rd_plus_wr\": %.0f},\n", g, c8, 2.0 * V * 64 / c8 / 1e3);
            printf(" \"chunk_copy_nt_u8_grid%d\": {\"us\": %.1f, \"GBs_rd_plus_wr\": %.0f},\n", g, n8, 2.0 * V * 64 / n8 / 1e3);
            printf(" \"chunk_read_u8_grid%d\": {\"us\": %.1f, \"GBs_rd\": %.0f},\n", g, r8, 1.0 * V * 64 / r8 / 1e3);
        } }
#define RUN(tag, LPR, UNR, RMW, NARR, A, B2, IDS, NN, bytes_per_row) { \
        int rows_per_wg = 256 / LPR; int grid = (NN + rows_per_wg * UNR - 1) / (rows_per_wg * UNR); if (grid > 256 * 32) grid = 256 * 32; \
        float us = timeit([&] { k_rows<LPR, UNR, RMW, NARR><<<grid, 256>>>(A, B2, IDS, NN, sink); }); \
        printf(" \"%s\": {\"rows\": %d, \"us\": %.1f, \"Mrows_per_s\": %.0f, \"useful_GBs\": %.0f},\n", tag, NN, us, NN / us, (double)NN * bytes_per_row / us / 1e3); }
    // read-only
    RUN("read64_sat", 4, 4, false, 1, t64, s64, ids, N, 64)
    RUN("read64_sat_u8", 4, 8, false, 1, t64, s64, ids, N, 64)
    RUN("read128_sat", 8, 4, false, 1, t128, t128, ids, N, 128)
    RUN("read128_sat_u8", 8, 8, false, 1, t128, t128, ids, N, 128)
    RUN("read64x2arrays_sat", 4, 4, false, 2, t64, s64, ids, N, 128)
    // read-modify-write
    RUN("rmw64_sat", 4, 4, true, 1, t64, s64, ids, N, 128)
    RUN("rmw64x2arrays_sat", 4, 4, true, 2, t64, s64, ids, N, 256)
    RUN("rmw128_sat", 8, 4, true, 1, t128, t128, ids, N, 256)
    RUN("rmw128_sat_u8", 8, 8, true, 1, t128, t128, ids, N, 256)
    // the bench's launch size (B = 4096): latency regime
    RUN("read64_b4096", 4, 1, false, 1, t64, s64, ids_small, NB, 64)
    RUN("read128_b4096", 8, 1, false, 1, t128, t128, ids_small, NB, 128)
    RUN("rmw64x2arrays_b4096", 4, 1, true, 2, t64, s64, ids_small, NB, 256)
    RUN("rmw128_b4096", 8, 1, true, 1, t128, t128, ids_small, NB, 256)
    printf(" \"note\": \"rows distinct, uniform over 26M rows; useful_GBs counts row bytes read (+ written for rmw)\"\n}\n");
    return 0;
}

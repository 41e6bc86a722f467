// membw.hip -- the copy / read / write floor of the memory system, re-calibrated: persistent grids
// (k x 256 CUs), U 16-byte loads in flight per lane before the first store, non-temporal variants; from HBM (4 GB
// streams) and between buffers small enough to stay in the 256-MB Infinity Cache.  MI355X_MICROARCH.md quotes 6.29 TB/s
// for a float4 copy; round 2's grid-stride k_copy<<<2048, 256>>> (one load in flight per lane) reached 4.93.
// hipcc --offload-arch=gfx950 -O3 membw.hip -o membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, bool NTL, bool NTS, int THR>
__global__ void __launch_bounds__(THR) k_copy(const v4f* __restrict__ s, v4f* __restrict__ d, size_t ntile) {
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const v4f* sp = s + t * (size_t)(THR * U) + threadIdx.x;
        v4f* dp = d + t * (size_t)(THR * U) + threadIdx.x;
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(sp + u * THR) : sp[u * THR];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v[u], dp + u * THR); else dp[u * THR] = v[u]; }
    }
}
template <int U, bool NTL, int THR>
__global__ void __launch_bounds__(THR) k_read(const v4f* __restrict__ s, float* sink, size_t ntile) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const v4f* sp = s + t * (size_t)(THR * U) + threadIdx.x;
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(sp + u * THR) : sp[u * THR];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = acc.x;
}
template <int U, bool NTS, int THR>
__global__ void __launch_bounds__(THR) k_write(v4f* __restrict__ d, size_t ntile) {
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        v4f* dp = d + t * (size_t)(THR * U) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v, dp + u * THR); else dp[u * THR] = v; }
    }
}
__global__ void k_copy_gs(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n) {  // round 2's form
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t BIG = (size_t)4 << 30;
    char *a, *b; float* sink;
    CK(hipMalloc(&a, BIG)); CK(hipMalloc(&b, BIG)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, BIG)); CK(hipMemset(b, 0, BIG));
    for (int pass = 0; pass < 2; ++pass) {
        const size_t bytes = pass == 0 ? BIG : ((size_t)64 << 20);
        const int reps = pass == 0 ? 3 : 30;
        printf("== %s: %zu MB -> %zu MB (best of %d; GB/s counts read + written bytes for copies)\n", pass == 0 ? "HBM" : "Infinity-Cache resident", bytes >> 20, bytes >> 20, reps);
        { const size_t n = bytes / 16; float t = timeit([&] { k_copy_gs<<<2048, 256>>>((const v4f*)a, (v4f*)b, n); }, reps);
          printf("copy grid-stride <<<2048,256>>> (round 2):            %7.0f GB/s\n", 2.0 * bytes / t / 1e6); }
#define COPY(U, NTL, NTS, THR, G) do { const size_t nt = bytes / 16 / (THR * U); \
        float t = timeit([&] { k_copy<U, NTL, NTS, THR><<<G, THR>>>((const v4f*)a, (v4f*)b, nt); }, reps); \
        printf("copy U=%2d ntl=%d nts=%d thr=%4d grid=%5d: %7.0f GB/s\n", U, (int)NTL, (int)NTS, THR, G, 2.0 * bytes / t / 1e6); } while (0)
        COPY(4, false, false, 256, 2048); COPY(8, false, false, 256, 2048); COPY(16, false, false, 256, 2048);
        COPY(8, false, false, 256, 1024); COPY(8, false, false, 256, 512); COPY(8, false, false, 256, 4096); COPY(8, false, false, 256, 65536);
        COPY(8, false, false, 512, 1024); COPY(8, false, false, 1024, 512); COPY(16, false, false, 512, 512);
        COPY(8, true, false, 256, 2048); COPY(8, false, true, 256, 2048); COPY(8, true, true, 256, 2048); COPY(16, true, true, 256, 2048); COPY(16, true, true, 512, 1024);
#define READ(U, NTL, THR, G) do { const size_t nt = bytes / 16 / (THR * U); \
        float t = timeit([&] { k_read<U, NTL, THR><<<G, THR>>>((const v4f*)a, sink, nt); }, reps); \
        printf("read U=%2d ntl=%d thr=%4d grid=%5d:       %7.0f GB/s\n", U, (int)NTL, THR, G, 1.0 * bytes / t / 1e6); } while (0)
        READ(8, false, 256, 2048); READ(16, false, 256, 2048); READ(16, true, 256, 2048); READ(16, false, 512, 1024); READ(8, false, 256, 65536);
#define WRITE(U, NTS, THR, G) do { const size_t nt = bytes / 16 / (THR * U); \
        float t = timeit([&] { k_write<U, NTS, THR><<<G, THR>>>((v4f*)b, nt); }, reps); \
        printf("write U=%2d nts=%d thr=%4d grid=%5d:      %7.0f GB/s\n", U, (int)NTS, THR, G, 1.0 * bytes / t / 1e6); } while (0)
        WRITE(8, false, 256, 2048); WRITE(8, true, 256, 2048); WRITE(16, true, 512, 1024); WRITE(8, false, 256, 65536);
    }
    return 0;
}

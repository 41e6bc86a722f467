// mall.hip -- what stays in the 256-MB Infinity Cache?  (1) write a 64-MB buffer A with policy P, stream Y MB of other
// traffic of kind K through the chip, then time a read of A: is A still served from the cache, and which kinds of traffic
// evict it;  (2) steady-state bandwidth of cache-resident copies (many sweeps inside ONE launch: no launch ramp).
// hipcc --offload-arch=gfx950 -O3 mall.hip -o mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE: 0 plain, 1 nt, 2 sc1 (buffer aux 16), 3 sc0 sc1 (aux 17)
template <int MODE> __device__ __forceinline__ v4f ld(const v4f* base, size_t i) {
    if (MODE == 1) return __builtin_nontemporal_load(base + i);
    if (MODE >= 2) { __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4f*>(base + (i & ~(size_t)0xffffff)), 0, 0x7fffffff, 0x00020000);
                     return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((i & 0xffffff) * 16), 0, MODE == 2 ? 16 : 17); }
    return base[i];
}
template <int MODE> __device__ __forceinline__ void st(v4f* base, size_t i, v4f v) {
    if (MODE == 1) { __builtin_nontemporal_store(v, base + i); return; }
    if (MODE >= 2) { __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (i & ~(size_t)0xffffff), 0, 0x7fffffff, 0x00020000);
                     __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)((i & 0xffffff) * 16), 0, MODE == 2 ? 16 : 17); return; }
    base[i] = v;
}
template <int MODE> __global__ void __launch_bounds__(256) k_read(const v4f* s, float* sink, size_t ntile, int sweeps) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int sw = 0; sw < sweeps; ++sw)
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld<MODE>(s, t * 2048 + u * 256 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = acc.x;
}
template <int MODE> __global__ void __launch_bounds__(256) k_write(v4f* d, size_t ntile, int sweeps) {
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (int sw = 0; sw < sweeps; ++sw)
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
#pragma unroll
        for (int u = 0; u < 8; ++u) st<MODE>(d, t * 2048 + u * 256 + threadIdx.x, v);
    }
}
template <int LM, int SM> __global__ void __launch_bounds__(256) k_copy(const v4f* s, v4f* d, size_t ntile, int sweeps) {
    for (int sw = 0; sw < sweeps; ++sw)
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld<LM>(s, t * 2048 + u * 256 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < 8; ++u) st<SM>(d, t * 2048 + u * 256 + threadIdx.x, v[u]);
    }
}

static hipEvent_t e0, e1;
template <typename F> float timed(F f) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t AB = (size_t)64 << 20, BB = (size_t)2 << 30;
    char *A, *A2, *B; float* sink;
    CK(hipMalloc(&A, AB)); CK(hipMalloc(&A2, AB)); CK(hipMalloc(&B, BB)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(A, 0, AB)); CK(hipMemset(A2, 0, AB)); CK(hipMemset(B, 0, BB));
    const size_t nA = AB / 16 / 2048;
    const char* names[4] = {"plain", "nt", "sc1", "sc0sc1"};
    printf("== (1) write A (64 MB, policy P), stream Y MB of traffic K over another buffer, read A (plain): us for the read of A\n");
    printf("        (an HBM read of 64 MB takes ~10.5 us at 6.3 TB/s; the kernel's launch ramp is in both)\n");
    for (int P = 0; P < 4; ++P) {
        for (int K = 0; K < 6; ++K) {   // 0 read plain, 1 read nt, 2 write plain, 3 write nt, 4 read sc1, 5 write sc1
            printf("A written %-6s, then %-11s:", names[P], K == 0 ? "reads" : K == 1 ? "nt reads" : K == 2 ? "writes" : K == 3 ? "nt writes" : K == 4 ? "sc1 reads" : "sc1 writes");
            for (size_t Y : {0, 64, 128, 192, 256, 384, 1024}) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    switch (P) { case 0: k_write<0><<<2048, 256>>>((v4f*)A, nA, 1); break; case 1: k_write<1><<<2048, 256>>>((v4f*)A, nA, 1); break;
                                 case 2: k_write<2><<<2048, 256>>>((v4f*)A, nA, 1); break; default: k_write<3><<<2048, 256>>>((v4f*)A, nA, 1); }
                    const size_t nB = (Y << 20) / 16 / 2048;
                    if (nB) switch (K) {
                        case 0: k_read<0><<<2048, 256>>>((const v4f*)B, sink, nB, 1); break;
                        case 1: k_read<1><<<2048, 256>>>((const v4f*)B, sink, nB, 1); break;
                        case 2: k_write<0><<<2048, 256>>>((v4f*)B, nB, 1); break;
                        case 3: k_write<1><<<2048, 256>>>((v4f*)B, nB, 1); break;
                        case 4: k_read<2><<<2048, 256>>>((const v4f*)B, sink, nB, 1); break;
                        default: k_write<2><<<2048, 256>>>((v4f*)B, nB, 1); }
                    const float ms = timed([&] { k_read<0><<<2048, 256>>>((const v4f*)A, sink, nA, 1); });
                    if (ms < best) best = ms;
                }
                printf("  Y=%4zu: %5.1f", Y, best * 1e3);
            }
            printf("\n");
        }
    }
    printf("== (2) steady state, 20 sweeps in one launch, GB/s (copies count read + written bytes)\n");
    for (size_t mb : {32, 64, 96, 128, 192, 512}) {
        const size_t n = (mb << 20) / 16 / 2048;
        const int SW = 20;
        float r = timed([&] { k_read<0><<<2048, 256>>>((const v4f*)B, sink, n, SW); });
        float w = timed([&] { k_write<0><<<2048, 256>>>((v4f*)B, n, SW); });
        float c = timed([&] { k_copy<0, 0><<<2048, 256>>>((const v4f*)B, (v4f*)(B + (mb << 20)), n, SW); });
        float cn = timed([&] { k_copy<0, 1><<<2048, 256>>>((const v4f*)B, (v4f*)(B + (mb << 20)), n, SW); });
        float cs = timed([&] { k_copy<0, 2><<<2048, 256>>>((const v4f*)B, (v4f*)(B + (mb << 20)), n, SW); });
        const double bytes = (double)(mb << 20) * SW;
        printf("%4zu MB: read %6.0f  write %6.0f  copy %6.0f  copy(nt st) %6.0f  copy(sc1 st) %6.0f\n", mb, bytes / r / 1e6, bytes / w / 1e6, 2 * bytes / c / 1e6, 2 * bytes / cn / 1e6, 2 * bytes / cs / 1e6);
    }
    // (3) HBM stream beside a cache-resident stream: two kernels on two streams
    {
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        const size_t nb = ((size_t)1 << 30) / 16 / 2048, ns = ((size_t)64 << 20) / 16 / 2048;
        CK(hipDeviceSynchronize());
        float t_h = timed([&] { k_copy<0, 1><<<1024, 256, 0, 0>>>((const v4f*)B, (v4f*)(B + ((size_t)1 << 30)), nb, 2); });
        float t_m = timed([&] { k_copy<0, 0><<<1024, 256, 0, 0>>>((const v4f*)A, (v4f*)A2, ns, 64); });
        hipEvent_t a, b, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, s1)); CK(hipStreamWaitEvent(s2, a, 0));
        k_copy<0, 1><<<1024, 256, 0, s1>>>((const v4f*)B, (v4f*)(B + ((size_t)1 << 30)), nb, 2);
        k_copy<0, 0><<<1024, 256, 0, s2>>>((const v4f*)A, (v4f*)A2, ns, 64);
        CK(hipEventRecord(b, s1)); CK(hipEventRecord(c, s2)); CK(hipEventSynchronize(b)); CK(hipEventSynchronize(c));
        float tb, tc; CK(hipEventElapsedTime(&tb, a, b)); CK(hipEventElapsedTime(&tc, a, c));
        printf("== (3) alone: HBM copy 2 x 1 GB (nt stores) %.0f GB/s; cache-resident copy 64 x 64 MB %.0f GB/s\n", 4.0 * (1 << 30) / t_h / 1e6, 2.0 * 64 * (64 << 20) / t_m / 1e6);
        printf("       together on two streams: HBM copy done after %.3f ms (alone %.3f), cache copy after %.3f ms (alone %.3f): sum of rates %.0f GB/s\n", tb, t_h, tc, t_m,
               (4.0 * (1 << 30) + 2.0 * 64 * (64 << 20)) / (tb > tc ? tb : tc) / 1e6);
    }
    return 0;
}

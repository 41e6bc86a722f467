// yfirst.hip -- memory-pattern skeletons of the two-pass "y first" pipeline, no arithmetic:
//   pass 1: a workgroup owns COLS adjacent columns x of a row-major float32 slab [ny][nx] (reads COLS*4-byte row segments,
//           16 bytes per lane, 4 lanes per 64 bytes), writes full 128-byte lines of W2[ky/2][x/8][ky%2][x%8] (complex64)
//   pass 2: a workgroup owns one row pair of W2 (one contiguous 64-KB read at nx = 4096) and writes four complete output rows
//           (two direct, two mirrored) of the row-major float32 result
// Answers: what do 32 / 64 / 128-byte strided row segments cost on the read side, does an XCD-aware unit order matter,
// and how close to a plain copy does each pass run.  Slabs are cycled (32 x 64 MB) so that nothing is cache-resident.
// hipcc --offload-arch=gfx950 -O3 yfirst.hip -o yfirst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NY = 4096, NX = 4096, NPAIR = NY / 4 + 1;  // ky = 0..ny/2 in pairs: 1025
constexpr size_t W2_SLAB = (size_t)NPAIR * (NX / 8) * 8 /*float4 per line*/;  // float4 units

// COLS in {8, 16, 32}; 1024 threads; lane = (u, g): g = tid % (COLS/4) fastest
template <int COLS, bool XCD, bool NTL = false>
__global__ void __launch_bounds__(1024) k_pass1(const float* __restrict__ in, float4* __restrict__ w2, int nslab) {
    constexpr int LPR = COLS / 4, RPR = 1024 / LPR, NQ = NY / RPR, UPS = NX / COLS;
    extern __shared__ float lds_fp[];  // the real kernel's LDS footprint decides how the two passes can share a CU
    if (nslab < 0) lds_fp[threadIdx.x] = 0.f;
    const int tid = threadIdx.x, g = tid % LPR, u = tid / LPR;
    int slab, xb;
    if (XCD) {  // blocks b, b+8, ... share an XCD: give each XCD a contiguous range of column blocks
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slab = j / (UPS / 8);
        xb = xcd * (UPS / 8) + j % (UPS / 8);
    } else {
        slab = blockIdx.x / UPS; xb = blockIdx.x % UPS;
    }
    if (slab >= nslab) return;
    const float* src = in + (size_t)slab * NY * NX + (size_t)xb * COLS + 4 * g;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
    for (int q = 0; q < NQ; ++q) {
        const float4* ap = reinterpret_cast<const float4*>(src + (size_t)(u + RPR * q) * NX);
        float4 v;
        if (NTL) { const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(ap)); v = make_float4(t.x, t.y, t.z, t.w); } else v = *ap;
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    // COLS/8 lines (128 B = 8 float4) per pair, adjacent in memory
    constexpr int F4PP = COLS;  // float4 per pair from this workgroup: COLS/8 lines * 8
    float4* dst = w2 + (size_t)slab * W2_SLAB + (size_t)xb * F4PP;
    for (int e = tid; e < NPAIR * F4PP; e += 1024) {
        const int p = e / F4PP, r = e % F4PP;
        dst[(size_t)p * (NX / 8) * 8 + r] = acc;
    }
}

// 512 threads; unit = one row pair p < ny/4 of one slab: 64-KB contiguous read, four 16-KB rows written
template <bool NTS = false, bool NTLD = false>
__global__ void __launch_bounds__(512) k_pass2(const float4* __restrict__ w2, float* __restrict__ out, int nslab) {
    extern __shared__ float lds_fp[];
    if (nslab < 0) lds_fp[threadIdx.x] = 0.f;
    const int tid = threadIdx.x;
    const int slab = blockIdx.x / (NY / 4), p = blockIdx.x % (NY / 4);
    const float4* src = w2 + (size_t)slab * W2_SLAB + (size_t)p * (NX / 8) * 8;
    float4 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (NTLD) { const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + tid + 512 * r)); v[r] = make_float4(t.x, t.y, t.z, t.w); }
        else v[r] = src[tid + 512 * r];
    }
    float4 s = v[0];
#pragma unroll
    for (int r = 1; r < 8; ++r) { s.x += v[r].x; s.y += v[r].y; s.z += v[r].z; s.w += v[r].w; }
    float* o = out + (size_t)slab * NY * NX;
    const int ky0 = 2 * p, ky1 = 2 * p + 1;
    const int rows[4] = {(ky0 + NY / 2) & (NY - 1), (ky1 + NY / 2) & (NY - 1), ((NY - ky0) + NY / 2) & (NY - 1), ((NY - ky1) + NY / 2) & (NY - 1)};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float4* d = reinterpret_cast<float4*>(o + (size_t)rows[r] * NX);
        if (NTS) { v4f t0 = {s.x, s.y, s.z, s.w}, t1 = {v[r].x, v[r].y, v[r].z, v[r].w}; __builtin_nontemporal_store(t0, reinterpret_cast<v4f*>(d + tid)); __builtin_nontemporal_store(t1, reinterpret_cast<v4f*>(d + tid + 512)); }
        else { d[tid] = s; d[tid + 512] = v[r]; }
    }
}

__global__ void k_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const int NS = 32;
    float* in; float4* w2; float* out;
    CK(hipMalloc(&in, (size_t)NS * NY * NX * 4)); CK(hipMemset(in, 0, (size_t)NS * NY * NX * 4));
    CK(hipMalloc(&w2, (size_t)NS * W2_SLAB * 16)); CK(hipMemset(w2, 0, (size_t)NS * W2_SLAB * 16));
    CK(hipMalloc(&out, (size_t)NS * NY * NX * 4)); CK(hipMemset(out, 0, (size_t)NS * NY * NX * 4));
    const double in_mb = NY * (double)NX * 4 / 1e6, w2_mb = W2_SLAB * 16.0 / 1e6;
    printf("per slab: in %.1f MB, W2 %.1f MB, out %.1f MB\n", in_mb, w2_mb, in_mb);
    {
        const size_t n = (size_t)NS * NY * NX / 4;
        float t = timeit([&] { k_copy<<<2048, 256>>>((const float4*)in, (float4*)out, n); }, 5);
        printf("plain copy in->out:            %6.1f us / slab  (%.0f GB/s r+w)\n", t * 1e3 / NS, 2 * in_mb * NS / t / 1e3);
    }
#define P1(C, X) do { \
        float t = timeit([&] { k_pass1<C, X><<<NS * (NX / C), 1024>>>(in, w2, NS); }, 5); \
        printf("pass1 cols=%2d (%3d-B segments) xcd-aware=%d: %6.1f us / slab  (%.0f GB/s r+w)\n", C, C * 4, (int)X, t * 1e3 / NS, (in_mb + w2_mb) * NS / t / 1e3); \
    } while (0)
    P1(8, false); P1(8, true); P1(16, false); P1(16, true); P1(32, false); P1(32, true);
    {
        float t = timeit([&] { k_pass2<false><<<NS * (NY / 4), 512>>>(w2, out, NS); }, 5);
        printf("pass2 (64-KB pair read, 4 rows written): %6.1f us / slab  (%.0f GB/s r+w)\n", t * 1e3 / NS, (in_mb + w2_mb) * NS / t / 1e3);
    }
    {   // both passes back to back per group of 32 slabs, as the plan would launch them
        float t = timeit([&] { k_pass1<16, true><<<NS * (NX / 16), 1024>>>(in, w2, NS); k_pass2<false><<<NS * (NY / 4), 512>>>(w2, out, NS); }, 5);
        printf("pass1(16, xcd) + pass2: %6.1f us / slab\n", t * 1e3 / NS);
    }
    // ---- the same two passes software-pipelined on two streams: pass 1 of group k+1 runs beside pass 2 of group k, the
    // intermediate cycles through a ring of R groups (is it served from the Infinity Cache, and do reads and writes overlap?)
    {
        const size_t L1 = 139264, L2 = 71680;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass1<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L1));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pass2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L2));
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        for (int withlds = 0; withlds < 2; ++withlds)
        for (int G : {1, 2, 4, 8}) for (int R : {2, 3}) {
            if (G * R > NS) continue;
            const int ngroups = 64;  // groups per repetition (inputs / outputs cycle over the 32 slabs)
            std::vector<hipEvent_t> done1(ngroups), done2(ngroups);
            for (auto& ev : done1) CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            for (auto& ev : done2) CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            const size_t l1 = withlds ? L1 : 0, l2 = withlds ? L2 : 0;
            auto run = [&] {
                for (int k = 0; k < ngroups; ++k) {
                    const int slot = k % R, s0 = (k * G) % NS;
                    if (k >= R) CK(hipStreamWaitEvent(s1, done2[k - R], 0));  // the slot's previous contents have been consumed
                    k_pass1<16, true><<<G * (NX / 16), 1024, l1, s1>>>(in + (size_t)s0 * NY * NX, w2 + (size_t)slot * G * W2_SLAB, G);
                    CK(hipEventRecord(done1[k], s1));
                    CK(hipStreamWaitEvent(s2, done1[k], 0));
                    k_pass2<false><<<G * (NY / 4), 512, l2, s2>>>(w2 + (size_t)slot * G * W2_SLAB, out + (size_t)s0 * NY * NX, G);
                    CK(hipEventRecord(done2[k], s2));
                }
                CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            };
            run();
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, s1));
            CK(hipStreamWaitEvent(s2, a, 0));
            for (int rep = 0; rep < 3; ++rep) run();
            CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("two streams, lds=%d, group %d slabs, ring %d (%4.0f MB of W2): %6.1f us / slab\n", withlds, G, R, G * R * w2_mb, ms * 1e3 / (3.0 * ngroups * G));
        }
    }
    // ---- one stream, small groups: the intermediate of a group is written and read back at once (Infinity-Cache resident?)
    for (int G : {1, 2, 4, 8, 32}) {
        float t = timeit([&] {
            for (int s0 = 0; s0 < NS; s0 += G) {
                k_pass1<16, true><<<G * (NX / 16), 1024>>>(in + (size_t)s0 * NY * NX, w2, G);
                k_pass2<false><<<G * (NY / 4), 512>>>(w2, out + (size_t)s0 * NY * NX, G);
            } }, 5);
        float t1 = timeit([&] { for (int s0 = 0; s0 < NS; s0 += G) k_pass1<16, true><<<G * (NX / 16), 1024>>>(in + (size_t)s0 * NY * NX, w2, G); }, 5);
        float t2 = timeit([&] { for (int s0 = 0; s0 < NS; s0 += G) k_pass2<false><<<G * (NY / 4), 512>>>(w2, out + (size_t)s0 * NY * NX, G); }, 5);
        printf("one stream, group %2d (W2 %4.0f MB re-used): %6.1f us / slab;  pass1 alone %5.1f, pass2 alone %5.1f\n", G, G * w2_mb, t * 1e3 / NS, t1 * 1e3 / NS, t2 * 1e3 / NS);
    }
    // ---- non-temporal variants (what leaves the intermediate in the Infinity Cache?): A = nt stores of the output only,
    // B = A + nt loads of the intermediate in pass 2, C = B + nt loads of the input in pass 1
#define NTV(NAME, L1, S2, L2) \
    for (int G : {1, 2, 3, 4, 6, 8, 16, 32}) { \
        float t = timeit([&] { \
            for (int s0 = 0; s0 + G <= NS; s0 += G) { \
                k_pass1<16, true, L1><<<G * (NX / 16), 1024>>>(in + (size_t)s0 * NY * NX, w2, G); \
                k_pass2<S2, L2><<<G * (NY / 4), 512>>>(w2, out + (size_t)s0 * NY * NX, G); \
            } }, 5); \
        printf("%s, one stream, group %2d (W2 %4.0f MB re-used): %6.1f us / slab\n", NAME, G, G * w2_mb, t * 1e3 / (NS / G * G)); \
    }
    NTV("plain", false, false, false)
    NTV("A (nt out)", false, true, false)
    NTV("B (nt out, nt W2 loads)", false, true, true)
    NTV("C (all nt)", true, true, true)
    // ---- copies between two small buffers that stay in the Infinity Cache: is read + write faster there than from HBM?
    for (size_t mb : {8, 16, 32, 64, 128, 512}) {
        const size_t n = (mb << 20) / 16;
        float t = timeit([&] { k_copy<<<2048, 256>>>((const float4*)in, (float4*)out, n); }, 50);
        printf("copy %4zu MB -> %4zu MB, same buffers every time: %7.0f GB/s r+w\n", mb, mb, 2.0 * (mb << 20) / t / 1e6);
    }
    return 0;
}

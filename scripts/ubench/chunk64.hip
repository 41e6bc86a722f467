// chunk64.hip -- the output pattern of a 16-column column pass: every workgroup (ka, tile) writes, for 1024 rows
// ky = ka + 4 k', a 64-byte chunk [ky][16 t .. 16 t + 15] and the mirrored chunk [(-ky)][4096 - 16 t - 15 .. 4096 - 16 t].
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct __attribute__((aligned(4))) F4u { float x, y, z, w; };
template <int MODE>  // 0 direct only, 1 direct + mirror as 4 unaligned 16-byte stores, 2 direct + mirror as 3 aligned float4 + 4 scalars
__global__ void __launch_bounds__(1024) k(float* __restrict__ out, int ntile) {
    const int unit = blockIdx.x;          // (ka, tile)
    const int ka = unit & 3, t = unit >> 2;
    if (t >= ntile) return;
    const int lane4 = threadIdx.x & 3;    // quarter of the 64-byte chunk
    for (int r = 0; r < 4; ++r) {
        const int kp = (threadIdx.x >> 2) + 256 * r;
        const int ky = ka + 4 * kp;
        const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
        if (t < 128) *reinterpret_cast<float4*>(out + (size_t)ky * 4096 + 16 * t + 4 * lane4) = v;   // direct (kx < 2048)
        if (MODE == 0) continue;
        const int mrow = (4096 - ky) & 4095;
        const int m0 = 4096 - 16 * t - 15;    // first mirrored column (kx = 16t+15 .. 16t -> cols m0 .. m0+15), t >= 0; kx = 0 has no mirror
        float* row = out + (size_t)mrow * 4096;
        if (MODE == 1) {
            int c = m0 + 4 * lane4;
            if (c + 3 < 4096) { F4u u; u.x = 1.f; u.y = 2.f; u.z = 3.f; u.w = 4.f; *reinterpret_cast<F4u*>(row + c) = u; }
            else { for (int i = 0; i < 4 && c + i < 4096; ++i) row[c + i] = 1.f; }
        } else {
            // aligned quads inside [m0, m0+15]: m0 = 1 mod 4 -> aligned quads start at m0+3, m0+7, m0+11; pieces: m0..m0+2 and m0+15
            if (lane4 < 3) *reinterpret_cast<float4*>(row + m0 + 3 + 4 * lane4) = v;
            else { row[m0] = 1.f; row[m0 + 1] = 2.f; row[m0 + 2] = 3.f; if (m0 + 15 < 4096) row[m0 + 15] = 4.f; }
        }
    }
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
    char* buf; CK(hipMalloc(&buf, (size_t)8 << 30)); CK(hipMemset(buf, 0, (size_t)8 << 30));
    int s = 0;
    auto slab = [&]() { return (float*)(buf + ((size_t)(s++ % 100) << 26)); };
    float t0 = timeit([&] { k<0><<<4 * 128, 1024>>>(slab(), 128); }, 100);
    float t1 = timeit([&] { k<1><<<4 * 129, 1024>>>(slab(), 129); }, 100);
    float t2 = timeit([&] { k<2><<<4 * 129, 1024>>>(slab(), 129); }, 100);
    printf("direct half only (33.5 MB): %6.1f us  %5.0f GB/s\n", t0 * 1e3, 33.5e6 / t0 / 1e6);
    printf("direct + mirror, unaligned 16-B stores (67 MB): %6.1f us  %5.0f GB/s\n", t1 * 1e3, 67.1e6 / t1 / 1e6);
    printf("direct + mirror, aligned quads + scalar pieces (67 MB): %6.1f us  %5.0f GB/s\n", t2 * 1e3, 67.1e6 / t2 / 1e6);
    return 0;
}

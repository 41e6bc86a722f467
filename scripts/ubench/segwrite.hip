// segwrite.hip -- can the column pass write its |F|^2 output as narrow per-tile segments?  Each block owns a tile of
// T columns of a [4096][4096] f32 slab and writes every row's T*4-byte segment (and optionally the mirrored one).
// Variants: tile->block mapping (naive vs XCD-aware so that the blocks sharing a 128-B line sit on one XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int T>  // T columns (floats) per tile; thread t writes rows t, t+nthreads, ...
__global__ void k_tilewrite(float* __restrict__ out, int ntiles, int mapping, int mirror) {
    int b = blockIdx.x;
    int tile;
    const int per_line = 32 / T;  // tiles sharing one 128-B line
    if (mapping == 0) tile = b;
    else {  // blocks b, b+8, b+16, ... (same XCD) take consecutive tiles
        int x = b % 8, j = b / 8;
        int grp = j / per_line, w = j % per_line;
        tile = (grp * 8 + x) * per_line + w;
    }
    if (tile >= ntiles) return;
    for (int row = threadIdx.x; row < 4096; row += blockDim.x) {
        float* p = out + (size_t)row * 4096 + (size_t)tile * T;
        if (T == 4) *reinterpret_cast<float4*>(p) = make_float4(1.f, 2.f, 3.f, 4.f);
        else if (T == 2) *reinterpret_cast<float2*>(p) = make_float2(1.f, 2.f);
        else for (int c = 0; c < T; c += 4) *reinterpret_cast<float4*>(p + c) = make_float4(1.f, 2.f, 3.f, 4.f);
        if (mirror) {
            int mrow = (4096 - row) & 4095;
            float* q = out + (size_t)mrow * 4096 + (4096 - T - (size_t)tile * T);
            if (T == 4) *reinterpret_cast<float4*>(q) = make_float4(4.f, 3.f, 2.f, 1.f);
            else if (T == 2) *reinterpret_cast<float2*>(q) = make_float2(1.f, 2.f);
            else for (int c = 0; c < T; c += 4) *reinterpret_cast<float4*>(q + c) = make_float4(1.f, 2.f, 3.f, 4.f);
        }
    }
}
// lanes-across-columns variant: a wave writes 64/T4 rows x T floats with lane = (row_in_wave, col4)
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
    for (int mirror = 0; mirror < 2; ++mirror)
        for (int mapping = 0; mapping < 2; ++mapping) {
            int s = 0;
            auto slab = [&]() { return (float*)(buf + ((size_t)(s++ % 100) << 26)); };
            const double bytes = 4096.0 * 4096 * 4 * (mirror ? 1.0 : 1.0);  // with mirror each block covers half the columns
            float t2 = timeit([&] { k_tilewrite<2><<<mirror ? 1024 : 2048, 256>>>(slab(), mirror ? 1024 : 2048, mapping, mirror); }, 100);
            float t4 = timeit([&] { k_tilewrite<4><<<mirror ? 512 : 1024, 1024>>>(slab(), mirror ? 512 : 1024, mapping, mirror); }, 100);
            float t8 = timeit([&] { k_tilewrite<8><<<mirror ? 256 : 512, 1024>>>(slab(), mirror ? 256 : 512, mapping, mirror); }, 100);
            float t16 = timeit([&] { k_tilewrite<16><<<mirror ? 128 : 256, 1024>>>(slab(), mirror ? 128 : 256, mapping, mirror); }, 100);
            float t32 = timeit([&] { k_tilewrite<32><<<mirror ? 64 : 128, 1024>>>(slab(), mirror ? 64 : 128, mapping, mirror); }, 100);
            printf("mirror=%d mapping=%s : T=2 %6.1f us (%5.0f GB/s) | T=4 %6.1f us (%5.0f GB/s) | T=8 %6.1f us (%5.0f) | T=16 %6.1f us (%5.0f) | T=32 %6.1f us (%5.0f)\n",
                   mirror, mapping ? "xcd  " : "naive", t2 * 1e3, bytes / t2 / 1e6, t4 * 1e3, bytes / t4 / 1e6, t8 * 1e3, bytes / t8 / 1e6, t16 * 1e3, bytes / t16 / 1e6, t32 * 1e3, bytes / t32 / 1e6);
        }
    return 0;
}

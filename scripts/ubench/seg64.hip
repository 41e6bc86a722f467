// seg64.hip -- scattered segment writes of 32/64/128/256 bytes (aligned), many segments in flight per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// buffer viewed as [ntile][4096 rows][seg bytes]; block (row group) writes for every tile a segment at its rows:
// block b owns ROWS consecutive rows; segment bytes = SEG; per tile the block writes ROWS*SEG contiguous bytes.
template <int SEG, int ROWS>
__global__ void k_w(char* __restrict__ buf, int ntile) {
    const int lanes_per_seg = SEG * ROWS / 16;  // 16-byte lanes per contiguous chunk
    const size_t tile_stride = (size_t)4096 * SEG;
    const size_t base = (size_t)blockIdx.x * ROWS * SEG;
    for (int e = threadIdx.x; e < ntile * lanes_per_seg; e += blockDim.x) {
        const int t = e / lanes_per_seg, l = e % lanes_per_seg;
        *reinterpret_cast<float4*>(buf + t * tile_stride + base + (size_t)l * 16) = make_float4(1.f, 2.f, 3.f, 4.f);
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
    auto slab = [&]() { return buf + ((size_t)(s++ % 60) << 27); };
    // total bytes per launch = 4096 rows * ntile * SEG ; choose ntile so total = 64 MiB
#define RUN(SEG, ROWS) { int ntile = (64 << 20) / (4096 * SEG); float t = timeit([&] { k_w<SEG, ROWS><<<4096 / ROWS, 256>>>(slab(), ntile); }, 50); \
    printf("chunk %4d B (seg %3d B x %d rows): %6.1f us  %6.0f GB/s\n", SEG * ROWS, SEG, ROWS, t * 1e3, (double)(64 << 20) / t / 1e6); }
    RUN(16, 1) RUN(16, 2) RUN(16, 4) RUN(16, 8)
    RUN(32, 1) RUN(32, 2) RUN(32, 4)
    RUN(64, 1) RUN(64, 2)
    RUN(128, 1) RUN(256, 1)
    return 0;
}

// membench.hip -- memory-system microbenchmarks that shaped the kernel design (DESIGN.md section "measurements").
// hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_read(const float4* __restrict__ p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i]; acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) *sink = acc;
}
__global__ void k_write(float4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// strided-segment read: array [rows][rowlen floats]; a block reads SEGS segments of seg_floats (4-byte lanes),
// segment s of block b = row (b_i2 + stride_rows*s), columns [b_col*seg_floats, ...)
__global__ void k_seg_read(const float* __restrict__ p, int rows, int rowlen, int seg_floats, int stride_rows, int nseg, float* sink) {
    const int tiles_x = rowlen / seg_floats;
    const int bcol = blockIdx.x % tiles_x, bi = blockIdx.x / tiles_x;
    float acc = 0.f;
    const int per = seg_floats;  // lanes per segment
    for (int e = threadIdx.x; e < nseg * per; e += blockDim.x) {
        const int s = e / per, c = e % per;
        const size_t row = (size_t)bi + (size_t)stride_rows * s;
        acc += p[row * rowlen + (size_t)bcol * seg_floats + c];
    }
    if (acc == 1.2345f) *sink = acc;
}
// scattered-segment write of float2: block writes nseg segments of seg_elems float2 at rows (bi + stride*s)
__global__ void k_seg_write(float2* __restrict__ p, int rowlen, int seg_elems, int stride_rows, int nseg) {
    const int tiles_x = rowlen / seg_elems;
    const int bcol = blockIdx.x % tiles_x, bi = blockIdx.x / tiles_x;
    for (int e = threadIdx.x; e < nseg * seg_elems; e += blockDim.x) {
        const int s = e / seg_elems, c = e % seg_elems;
        const size_t row = (size_t)bi + (size_t)stride_rows * s;
        p[row * rowlen + (size_t)bcol * seg_elems + c] = make_float2(1.f, 2.f);
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
    float* sink; CK(hipMalloc(&sink, 4));
    const size_t MAXB = (size_t)8 << 30;
    char* buf; CK(hipMalloc(&buf, MAXB)); CK(hipMemset(buf, 0, MAXB));
    char* buf2; CK(hipMalloc(&buf2, (size_t)2 << 30)); CK(hipMemset(buf2, 0, (size_t)2 << 30));
    printf("== streaming, same buffer re-used every repetition (fits Infinity Cache when small)\n");
    for (size_t mb : {16, 32, 64, 128, 192, 256, 512, 2048}) {
        size_t n = (mb << 20) / 16;
        float r = timeit([&] { k_read<<<2048, 256>>>((const float4*)buf, n, sink); }, 20);
        float w = timeit([&] { k_write<<<2048, 256>>>((float4*)buf, n); }, 20);
        float c = timeit([&] { k_copy<<<2048, 256>>>((const float4*)buf, (float4*)buf2, n); }, 20);
        printf("size %5zu MB: read %7.1f GB/s  write %7.1f GB/s  copy(r+w) %7.1f GB/s\n", mb, (mb << 20) / r / 1e6, (mb << 20) / w / 1e6, 2.0 * (mb << 20) / c / 1e6);
    }
    printf("== streaming over 8 GB in 64 MB slabs, one launch per slab (HBM every time)\n");
    {
        size_t n = ((size_t)64 << 20) / 16;
        int i = 0;
        float r = timeit([&] { for (int s = 0; s < 128; ++s) k_read<<<2048, 256>>>((const float4*)(buf + ((size_t)s << 26)), n, sink); }, 3);
        printf("read  64MB slabs: %7.1f GB/s (%.1f us per slab)\n", 128.0 * (64 << 20) / r / 1e6, r * 1e3 / 128);
        float w = timeit([&] { for (int s = 0; s < 128; ++s) k_write<<<2048, 256>>>((float4*)(buf + ((size_t)s << 26)), n); }, 3);
        printf("write 64MB slabs: %7.1f GB/s (%.1f us per slab)\n", 128.0 * (64 << 20) / w / 1e6, w * 1e3 / 128);
        (void)i;
    }
    printf("== producer/consumer through a 64 MB intermediate: write slab then read it back (x128), vs 1 GB intermediate\n");
    for (size_t mb : {64, 128, 256, 1024}) {
        size_t n = (mb << 20) / 16;
        float t = timeit([&] { k_write<<<2048, 256>>>((float4*)buf2, n); k_read<<<2048, 256>>>((const float4*)buf2, n, sink); }, 20);
        printf("intermediate %5zu MB: write+read %7.1f GB/s aggregate (%.1f us)\n", mb, 2.0 * (mb << 20) / t / 1e6, t * 1e3);
    }
    printf("== pipeline of 3: read new 64MB input slab (HBM) -> write 64MB intermediate -> read intermediate -> write new 64MB output slab\n");
    {
        size_t n = ((size_t)64 << 20) / 16;
        float t = timeit([&] {
            for (int s = 0; s < 32; ++s) {
                k_copy<<<2048, 256>>>((const float4*)(buf + ((size_t)s << 26)), (float4*)buf2, n);
                k_copy<<<2048, 256>>>((const float4*)buf2, (float4*)(buf + ((size_t)(64 + s) << 26)), n);
            } }, 3);
        printf("2-pass copy pipeline: %.1f us per slab (ideal HBM-only traffic 128 MB -> %.1f GB/s compulsory)\n", t * 1e3 / 32, 32.0 * 2 * (64 << 20) / t / 1e6);
    }
    printf("== segment reads on [4096][4096] f32 slabs (64 MB): block reads 128 segments of 128 B, row stride 64 (K1 pattern)\n");
    {
        const int rows = 4096, rowlen = 4096;
        for (int seg : {16, 32, 64, 128}) {
            int nseg = 128, stride = 32;  // 2 i2 x 64 i1 -> rows bi + 32*s covers 4096 rows for bi<32
            int blocks = (rowlen / seg) * 32;
            // warm: same slab repeatedly (MALL/L2); cold: cycle through 64 slabs
            float warm = timeit([&] { k_seg_read<<<blocks, 256>>>((const float*)buf, rows, rowlen, seg, stride, nseg, sink); }, 20);
            int s = 0;
            float cold = timeit([&] { k_seg_read<<<blocks, 256>>>((const float*)(buf + ((size_t)(s++ % 100) << 26)), rows, rowlen, seg, stride, nseg, sink); }, 100);
            printf("seg %4d B: warm %7.1f GB/s   cold(HBM) %7.1f GB/s\n", seg * 4, 67.1e6 / warm / 1e3, 67.1e6 / cold / 1e3);
        }
    }
    printf("== segment writes of float2 into [2112][4096] c64 (69 MB): block writes 66 segments, row stride 64 (K1 output pattern)\n");
    {
        for (int seg : {16, 32, 64}) {
            int rowlen = 4096, nseg = 66, stride = 32;
            int blocks = (rowlen / seg) * 32;
            float warm = timeit([&] { k_seg_write<<<blocks, 256>>>((float2*)buf2, rowlen, seg, stride, nseg); }, 20);
            printf("seg %4d B: %7.1f GB/s\n", seg * 8, (double)blocks * nseg * seg * 8 / warm / 1e6);
        }
    }
    return 0;
}

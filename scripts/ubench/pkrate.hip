// Issue rate of packed vs scalar FP32 FMA on gfx950: 16 scalar v_fma_f32 vs 8 v_pk_fma_f32 per iteration (same flops).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_scalar(float* out, int iters) {
    float a[16], b = 1.0001f, c = 0.5f;
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_packed(float* out, int iters) {
    f2 a[8], b = {1.0001f, 1.0001f}, c = {0.5f, 0.5f};
    for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 1e-3f + i + 8};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_packed_add(float* out, int iters) {
    f2 a[8], b = {1.0001f, 1.0001f};
    for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 1e-3f + i + 8};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_scalar_add(float* out, int iters) {
    float a[16], b = 1.0001f;
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K> float run(K k, float* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<256 * 8, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(a); k<<<256 * 8, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 20000;
    const double flops = 2.0 * 16 * iters * 256.0 * 8 * 256;
    float t1 = run(k_scalar, d, iters), t2 = run(k_packed, d, iters), t3 = run(k_scalar_add, d, iters), t4 = run(k_packed_add, d, iters);
    printf("scalar fma  %.3f ms  %.1f TFLOP/s\npacked fma  %.3f ms  %.1f TFLOP/s\nscalar add  %.3f ms (%.1f Gop/s x2)\npacked add  %.3f ms\n", t1, flops / t1 * 1e-9, t2, flops / t2 * 1e-9, t3, flops / 2 / t3 * 1e-6, t4);
    return 0;
}

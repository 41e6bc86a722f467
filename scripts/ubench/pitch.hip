// Does the power-of-two row pitch of a (4096, 4096) float32 slab cost pass 1 its input over-fetch?  The read pattern of fasty_cols_kernel<4096>
// (512 threads, 64 KB of LDS = two workgroups per CU, 8 columns = 32-byte row segments, 16 rows per thread, XCD-contiguous column blocks, the four
// sharers of a 128-byte line in consecutive workgroups) on slabs whose rows are PITCH floats apart: 4096 (the real case), 4096 + 32, 4096 + 288.
// Prints the time per slab; run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  for the bytes (x2, gfx950: profiles/r02_pmc_ubench_calibration.txt).
// Build: hipcc --offload-arch=gfx950 -O3 pitch.hip -o pitch
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NY = 4096, NX = 4096, CW = 8;
template <int PITCH>
__global__ void __launch_bounds__(512) k_read(const float* __restrict__ in, float* __restrict__ sink, int nslab) {
    extern __shared__ float lds_fp[];
    if (nslab < 0) lds_fp[threadIdx.x] = 0.f;
    const int tid = threadIdx.x, g = tid & 1, u = tid >> 1;  // 2 float4 per row segment, 256 row slots
    const int nxb = NX / CW, xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nxb >> 3;
    const int slab = j / per, xb = xcd * per + j % per;
    const float* src = in + (size_t)slab * NY * PITCH + (size_t)xb * CW + 4 * g;
    float4 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const float4*>(src + (size_t)(u + 256 * q) * PITCH);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v[q].x + v[q].y + v[q].z + v[q].w;
    if (s == 1.2345f) sink[blockIdx.x] = s;
}
template <int PITCH> int run(const char* name) {
    const int NS = 32;
    float* in; float* sink;
    CK(hipMalloc(&in, (size_t)NS * NY * PITCH * 4)); CK(hipMemset(in, 0, (size_t)NS * NY * PITCH * 4));
    CK(hipMalloc(&sink, 1 << 20));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_read<PITCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const dim3 grid(NS * (NX / CW));
    hipLaunchKernelGGL(k_read<PITCH>, grid, dim3(512), 65536, 0, in, sink, NS);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_read<PITCH>, grid, dim3(512), 65536, 0, in, sink, NS);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s %6.2f us per slab = %5.2f TB/s on the 67.1 MB a slab holds\n", name, ms / 5 * 1e3 / NS, 67.1 * NS / (ms / 5) / 1e3);
    CK(hipFree(in)); CK(hipFree(sink));
    return 0;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (run<4096>("pitch 4096 floats (16 KB)")) return 1;
    if (run<4096 + 32>("pitch 4096 + 32 (one line)")) return 1;
    if (run<4096 + 288>("pitch 4096 + 288 (9 lines)")) return 1;
    return 0;
}

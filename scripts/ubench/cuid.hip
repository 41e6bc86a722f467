// Which CU does workgroup b of a launch land on?  (csrc/fasty.h maps the four column blocks that share a 128-byte input line to
// consecutive workgroups of one XCD; if consecutive workgroups of an XCD go to different CUs, sharers 32 apart would share a CU.)
// Every workgroup (512 threads, 64 KB of LDS: two per CU, as fasty_cols_kernel<4096>) records XCC_ID, HW_ID and its start time, then
// spins for a while.  Build: hipcc --offload-arch=gfx950 -O2 cuid.hip -o cuid ; run: ./cuid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(512) probe(unsigned* out, long long spin) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        const long long t0 = wall_clock64();
        out[3 * blockIdx.x] = x; out[3 * blockIdx.x + 1] = h; out[3 * blockIdx.x + 2] = (unsigned)t0;
        smem[0] = 1;
        while (wall_clock64() - t0 < spin) {}
    }
    __syncthreads();
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int nb = 2048;
    unsigned* d; hipMalloc(&d, nb * 12);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 65536, 0, d, 2000ll + 37 * rep);  // 100 MHz clock: 20 us
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(nb * 3);
    hipMemcpy(h.data(), d, nb * 12, hipMemcpyDeviceToHost);
    printf("# block xcc se sh cu  (HW_ID fields: cu [11:8], sh [12], se [15:13]) start-time\n");
    for (int b = 0; b < nb; ++b) {
        const unsigned x = h[3 * b] & 0xf, w = h[3 * b + 1];
        if (b < 640 || (b % 8) == 0) printf("%4d  xcc %u  se %u sh %u cu %2u  simd %u  t %u\n", b, x, (w >> 13) & 7, (w >> 12) & 1, (w >> 8) & 15, (w >> 4) & 3, h[3 * b + 2] - h[2]);
    }
    return 0;
}

// A plain C-ABI client of libxrft_hip.so: no Python, no torch -- raw device pointers in, spectrum out.
// Power spectrum of two 256 x 384 float32 slabs (linear detrend, no window), checked on the host with closed forms:
// the plane is gone (the k = 0 bin holds only rounding), and Parseval holds for the detrended field.
// Build: hipcc --offload-arch=gfx950 ps_example.cpp -I../../include -L../../xrft_amd -lxrft_hip -o ps_example
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "xrft_hip.h"

#define CK(x) do { int rc_ = (x); if (rc_) { std::fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, xrfthip_strerror(rc_)); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int64_t nb = 2, ny = 256, nx = 384;
    const size_t n = (size_t)nb * ny * nx;
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t i = 0; i < ny; ++i)
            for (int64_t j = 0; j < nx; ++j) {
                s = s * 1664525u + 1013904223u;
                const float noise = (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f;
                h[((size_t)b * ny + i) * nx + j] = noise + 0.01f * (float)i - 0.02f * (float)j + 3.0f;
            }
    xrfthip_desc d = {};
    d.struct_size = sizeof d; d.ndim = 2; d.batch = nb; d.ny = ny; d.nx = nx;
    d.dtype = XRFTHIP_F32; d.out_mode = XRFTHIP_OUT_POWER; d.detrend = XRFTHIP_DETREND_LINEAR;
    d.flags = 0; d.scale = 1.0;
    xrfthip_plan* plan = nullptr;
    CK(xrfthip_plan_create(&plan, &d));
    const size_t wsb = xrfthip_workspace_bytes(plan);
    float *d_in = nullptr, *d_out = nullptr;
    void* d_ws = nullptr;
    HK(hipMalloc(&d_in, n * sizeof(float)));
    HK(hipMalloc(&d_out, n * sizeof(float)));
    HK(hipMalloc(&d_ws, wsb ? wsb : 256));
    HK(hipMemcpy(d_in, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t st;
    HK(hipStreamCreate(&st));
    CK(xrfthip_exec(plan, d_in, nullptr, d_out, nullptr, d_ws, wsb, st));
    HK(hipStreamSynchronize(st));
    std::vector<float> ps(n);
    HK(hipMemcpy(ps.data(), d_out, n * sizeof(float), hipMemcpyDeviceToHost));
    // the stand-alone detrend gives the field whose energy Parseval must reproduce
    std::vector<float> det(n);
    float* d_det = nullptr;
    HK(hipMalloc(&d_det, n * sizeof(float)));
    void* d_ws2 = nullptr;
    const size_t ws2 = xrfthip_detrend_workspace_bytes(nb);
    HK(hipMalloc(&d_ws2, ws2));
    CK(xrfthip_detrend(XRFTHIP_F32, 2, nb, ny, nx, XRFTHIP_DETREND_LINEAR, d_in, d_det, d_ws2, ws2, st));
    HK(hipStreamSynchronize(st));
    HK(hipMemcpy(det.data(), d_det, n * sizeof(float), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int64_t b = 0; b < nb; ++b) {
        double e_x = 0.0, e_k = 0.0;
        for (size_t e = 0; e < (size_t)ny * nx; ++e) {
            const double v = det[(size_t)b * ny * nx + e];
            e_x += v * v;
            e_k += ps[(size_t)b * ny * nx + e];
        }
        e_k /= (double)(ny * nx);  // sum |F|^2 = N sum x^2
        const double dc = ps[(size_t)b * ny * nx];  // unshifted: k = 0 first
        std::printf("slab %lld: sum x^2 = %.6e, sum |F|^2 / N = %.6e, |F(0)|^2 = %.3e\n", (long long)b, e_x, e_k, dc);
        if (std::fabs(e_k - e_x) > 1e-4 * e_x || dc > 1e-3 * e_x) bad = 1;
    }
    CK(xrfthip_plan_destroy(plan));
    hipFree(d_in); hipFree(d_out); hipFree(d_ws); hipFree(d_det); hipFree(d_ws2);
    std::puts(bad ? "FAIL" : "OK");
    return bad;
}

// selftest.h -- the memory floor of the headline path, MEASURED by the library itself (xrfthip_selftest_floor): what the two passes of
// a 4096^2 float32 power spectrum (fasty.h) cost when nothing but their memory accesses is left.  Three kernels, no arithmetic:
//   selftest_copy_kernel   a plain 16-byte copy (the read + write rate of this box)
//   selftest_cols_kernel   pass 1's accesses: 512 threads own 8 adjacent columns (32-byte row segments, 16 rows per thread, the
//                          XCD-aware unit order), and write the half-spectrum intermediate as 16-byte pieces of full 128-byte lines,
//                          non-temporal -- fasty_cols_kernel<4096> with the transforms removed
//   selftest_rows_kernel   pass 2's accesses: 512 threads read 4 rows of the intermediate (one contiguous block, 8 bytes per lane and
//                          load) and write them twice, rotated and mirrored, as whole rows of 16-byte non-temporal stores
// Each is launched with the real kernel's workgroup size and dynamic LDS, so as many workgroups share a CU as in the product.
// bench.py times them in the run it reports (roofline.measured_floor): the claim "the kernels run at the floor of their access
// patterns" is then a number of that run, not of a committed profile.   (xrft.power_spectrum: xrft/xrft.py:685-750)
#pragma once
#include "fasty.h"

namespace xrft {

static __global__ void __launch_bounds__(256) selftest_copy_kernel(const F4* __restrict__ src, F4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        F4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) xrft_store_nt(reinterpret_cast<float*>(dst + i + k * stride), v[k]);
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// the geometry of the 4096-point passes (fasty.h: YCols<4096>, YRows<4096>)
struct SelfGeom {
    static constexpr int N = 4096, THR = 512, NT = 256, GY = 2, CW = 8, RK = 4, LBS = 16, NROW_PAD = 2052, RPU = 4;
};

static __global__ void __launch_bounds__(512, 4) selftest_cols_kernel(const float* __restrict__ in, cf* __restrict__ w2, int nslab) {
    typedef SelfGeom S;
    XRFT_DYN_SMEM(smem_raw);
    if (nslab < 0) smem_raw[threadIdx.x] = 0;  // (keeps the dynamic LDS allocation: it is what sets the workgroups per CU)
    const int tid = threadIdx.x, g = tid % S::GY, u = tid / S::GY;
    const int nxb = S::N / S::CW, xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nxb >> 3;
    const int slab = j / per, xb = xcd * per + j % per;
    const char* __restrict__ src = reinterpret_cast<const char*>(in + (size_t)slab * S::N * S::N + (size_t)xb * S::CW);
    const unsigned off0 = ((unsigned)u * (unsigned)S::N + 4u * (unsigned)g) * 4u, rstep = (unsigned)S::NT * (unsigned)S::N * 4u;
    F4 raw[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) raw[q] = *reinterpret_cast<const F4*>(src + (off0 + rstep * (unsigned)q));
    char* __restrict__ w2s = reinterpret_cast<char*>(w2 + (size_t)slab * S::NROW_PAD * S::N);
#pragma unroll
    for (int set = 0; set < 2; ++set)
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int k = u + S::NT * q;
            if (q < 8 || u == 0) {
                const unsigned off = ((((unsigned)(k / S::RK) * (unsigned)nxb + (unsigned)xb) * 2u + set) * S::LBS) + (k % S::RK) * (2 * S::GY) + 2 * g;
                F4 o = raw[q & 15];
                o.x += raw[(q + 8 * set) & 15].y;
                xrft_store_nt(reinterpret_cast<float*>(w2s + off * 8u), o);
            }
        }
}

static __global__ void __launch_bounds__(512, 4) selftest_rows_kernel(const cf* __restrict__ w2, float* __restrict__ out, int nslab) {
    typedef SelfGeom S;
    XRFT_DYN_SMEM(smem_raw);
    if (nslab < 0) smem_raw[threadIdx.x] = 0;
    const int tid = threadIdx.x;
    const int upr = S::NROW_PAD / S::RPU, slab = (int)blockIdx.x / upr, unit = (int)blockIdx.x % upr, ky0 = unit * S::RPU, nyh = S::N / 2;
    // the unit's four rows are one contiguous 128-KB block of the intermediate (RK = 4 rows per 128-byte line); the lanes address it as
    // fasty_rows_kernel does: lane (u, g), transform A = row ky0 + g, B = row ky0 + 2 + g, x = u + 256 q, 8 bytes per lane and load
    const int g = tid % 2, u = tid / 2, nxb = S::N / S::CW;
    auto w2off = [&](int ky, int x) -> unsigned {
        const unsigned blk = (((unsigned)ky >> 2) * (unsigned)nxb + ((unsigned)x >> 3)) * 2u + (((unsigned)x >> 1) & 1u);
        return (blk << 4) + (((unsigned)ky & 3u) << 2) + ((((unsigned)x & 7u) >> 2) << 1) + ((unsigned)x & 1u);
    };
    const char* __restrict__ w2s = reinterpret_cast<const char*>(w2 + (size_t)slab * S::NROW_PAD * S::N);
    const int kyA = min(ky0 + g, nyh), kyB = min(ky0 + 2 + g, nyh);
    const unsigned offA = w2off(kyA, u) * 8u, offB = w2off(kyB, u) * 8u, qstr = (unsigned)(((S::NT >> 3) * 2) << 4) * 8u;
    cf a[16], b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        a[q] = *reinterpret_cast<const cf*>(w2s + (offA + qstr * (unsigned)q));
        b[q] = *reinterpret_cast<const cf*>(w2s + (offB + qstr * (unsigned)q));
    }
    float* __restrict__ outs = out + (size_t)slab * S::N * S::N;
    constexpr int CPR = S::N / 4;  // 16-byte chunks per row
#pragma unroll
    for (int it = 0; it < S::RPU * 2 * CPR / S::THR; ++it) {  // every valid row leaves twice: rotated (direct) and reversed + rotated (mirror)
        const int e = tid + S::THR * it, chunk = e % CPR, rr = e / CPR, rl = rr >> 1, mir = rr & 1;
        const int ky = ky0 + rl;
        if (ky > nyh || (mir && (ky == 0 || ky == nyh))) continue;
        const int orow = mir ? ((S::N - ky) + nyh) & (S::N - 1) : (ky + nyh) & (S::N - 1);
        F4 o; o.x = a[it].re; o.y = a[it].im; o.z = b[it].re; o.w = b[it].im;
        xrft_store_nt(outs + ((size_t)orow * S::N + 4 * (mir ? CPR - 1 - chunk : chunk)), o);
    }
}

}  // namespace xrft

// aux_kernels.h -- small bandwidth-bound helper kernels around the FFT passes:
//   slab_moments      centred first moments of every slab (for detrend)      xrft/detrend.py:54-55, 64-71, 100-113
//   finalize_coef     moments -> trend coefficients c0 + c1*i + c2*j
//   detrend_apply     out = in - trend                                        xrft.detrend as a stand-alone op
//   radial_binsum     isotropize of an existing spectrum                      xrft/xrft.py:895-906, 993-1004
#pragma once
#include "tile_fft.h"

namespace xrft {

// Block-wide sum of NV doubles per thread through LDS; result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum(double* v, double* red /* >= NV*blockDim doubles */) {
    const int tid = threadIdx.x, n = blockDim.x;
    for (int k = 0; k < NV; ++k) red[k * n + tid] = v[k];
    __syncthreads();
    for (int s = n / 2; s > 0; s >>= 1) {
        if (tid < s)
            for (int k = 0; k < NV; ++k) red[k * n + tid] += red[k * n + tid + s];
        __syncthreads();
    }
    if (tid == 0)
        for (int k = 0; k < NV; ++k) v[k] = red[k * n];
}

// part[slab][chunk][6] = { sum x.re, sum x.im, sum (i-ibar) x.re, .. .im, sum (j-jbar) x.re, .. .im } over the rows dealt to
// block `chunk` of the slab; grid = (chunks, batch).  The finalize kernel adds the chunks in order: no atomics, the result
// does not depend on the order in which blocks finish (bit-identical from run to run).
template <typename T, bool CPLX>
__global__ void __launch_bounds__(256) slab_moments_kernel(const void* in, long long ny, long long nx, long long slab_stride,
                                                          long long row_stride, double* acc) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);
    const long long b = blockIdx.y;
    const T ibar = (T)(0.5 * (double)(ny - 1)), jbar = (T)(0.5 * (double)(nx - 1));
    double s[6] = {0, 0, 0, 0, 0, 0};
    // rows are dealt round-robin to the blocks of a slab; inside a row lanes stride along j (no integer division)
    for (long long i = blockIdx.x; i < ny; i += gridDim.x) {
        const long long base = b * slab_stride + i * row_stride;
        T p0r = 0, p0i = 0, pjr = 0, pji = 0;
        int cnt = 0;
        // four loads of a thread in flight at once (one at a time the pass ran at 2.9 TB/s on 3000 x 3000 float64 slabs); the sums in the same order as ever
        constexpr int U = 4;
        for (long long j0 = threadIdx.x; j0 < nx; j0 += (long long)U * blockDim.x) {
            T vr[U], vi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long j = j0 + (long long)u * blockDim.x;
                vr[u] = (T)0; vi[u] = (T)0;
                if (j < nx) {
                    if (CPLX) { C2<T> v = reinterpret_cast<const C2<T>*>(in)[base + j]; vr[u] = v.re; vi[u] = v.im; }
                    else vr[u] = reinterpret_cast<const T*>(in)[base + j];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long j = j0 + (long long)u * blockDim.x;
                if (j >= nx) break;
                const T xr = vr[u], xi = vi[u];
                const T dj = (T)j - jbar;
                p0r += xr; pjr += dj * xr;
                if (CPLX) { p0i += xi; pji += dj * xi; }
                if (++cnt == 64) {  // bound the length of the working-precision partial sums
                    s[0] += (double)p0r; s[1] += (double)p0i; s[4] += (double)pjr; s[5] += (double)pji;
                    s[2] += (double)(((T)i - ibar) * p0r); s[3] += (double)(((T)i - ibar) * p0i);
                    p0r = p0i = pjr = pji = (T)0; cnt = 0;
                }
            }
        }
        s[0] += (double)p0r; s[1] += (double)p0i; s[4] += (double)pjr; s[5] += (double)pji;
        s[2] += (double)(((T)i - ibar) * p0r); s[3] += (double)(((T)i - ibar) * p0i);
    }
    block_sum<6>(s, red);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; ++k) acc[(b * gridDim.x + blockIdx.x) * 6 + k] = s[k];
}

// One wave per slab: lane l adds the chunks l, l + 64, ... in order, the lanes meet in a fixed shuffle tree (one thread per slab walked
// its chunks alone: 75 us for the 7 slabs of a (16, 3000, 3000) group).  Least squares on a full regular grid: the centred regressors
// (i-ibar), (j-jbar) are orthogonal to each other and to 1, so the plane fit of detrend.py:100-113 (normal equations on [1, i+1, j+1])
// and the line fit of scipy.signal.detrend (detrend.py:64-71) reduce to three independent ratios.
static __global__ void __launch_bounds__(64) finalize_coef_kernel(const double* part, double* coef, long long batch, long long ny, long long nx, int kind, int nchunk) {
    const long long b = blockIdx.x;
    if (b >= batch) return;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int ch = threadIdx.x; ch < nchunk; ch += 64)
        for (int k = 0; k < 6; ++k) acc[k] += part[(b * nchunk + ch) * 6 + k];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
        for (int k = 0; k < 6; ++k) acc[k] += __shfl_xor(acc[k], m);
    if (threadIdx.x != 0) return;
    const double n = (double)ny * (double)nx;
    const double ibar = 0.5 * (double)(ny - 1), jbar = 0.5 * (double)(nx - 1);
    const double sii = (double)nx * (double)ny * ((double)ny * (double)ny - 1.0) / 12.0;
    const double sjj = (double)ny * (double)nx * ((double)nx * (double)nx - 1.0) / 12.0;
    for (int c = 0; c < 2; ++c) {
        const double mean = acc[c] / n;
        double c1 = 0.0, c2 = 0.0;
        if (kind == 2) {
            if (ny > 1) c1 = acc[2 + c] / sii;
            if (nx > 1) c2 = acc[4 + c] / sjj;
        }
        coef[b * 6 + c] = mean - c1 * ibar - c2 * jbar;
        coef[b * 6 + 2 + c] = c1;
        coef[b * 6 + 4 + c] = c2;
    }
}

// XRFTHIP_AXIS_Y: one least-squares line per COLUMN of [slab][ny][nx] (scipy.signal.detrend along y, xrft/detrend.py:64-71,
// or the column mean, :54-55).  Lanes run along x (coalesced); fixed summation order: deterministic.
// coef[(slab * nx + j) * 6] = { c0.re, c0.im, c1.re, c1.im, 0, 0 }, trend = c0 + c1 * i.
template <typename T, bool CPLX>
__global__ void __launch_bounds__(256) column_fit_kernel(const void* in, long long ny, long long nx, double* coef, int kind) {
    // 64 columns x 4 row parts per workgroup (one thread per column left half of the CUs idle on (16, 4096, 2048) and ran at
    // 0.66 TB/s); a part's rows in batches of U loads; the four partial sums are added in part order: deterministic
    constexpr int CX = 64, RY = 4, U = 8;
    XRFT_DYN_SMEM(smem_raw);  // RY * 4 * CX doubles
    double (*red)[4][CX] = reinterpret_cast<double (*)[4][CX]>(smem_raw);
    const int cx = threadIdx.x % CX, ry = threadIdx.x / CX;
    const long long b = blockIdx.y, j = (long long)blockIdx.x * CX + cx;
    const double ibar = 0.5 * (double)(ny - 1);
    const long long per = (ny + RY - 1) / RY, r0 = ry * per, r1 = r0 + per < ny ? r0 + per : ny;
    double s0r = 0, s0i = 0, s1r = 0, s1i = 0;
    if (j < nx) {
        for (long long i0 = r0; i0 < r1; i0 += U) {
            double xr[U], xi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xr[u] = 0.0; xi[u] = 0.0;
                if (i0 + u < r1) {
                    const long long off = (b * ny + i0 + u) * nx + j;
                    if (CPLX) { const C2<T> v = reinterpret_cast<const C2<T>*>(in)[off]; xr[u] = (double)v.re; xi[u] = (double)v.im; }
                    else xr[u] = (double)reinterpret_cast<const T*>(in)[off];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u >= r1) break;
                const double di = (double)(i0 + u) - ibar;
                s0r += xr[u]; s1r = fma(di, xr[u], s1r);
                if (CPLX) { s0i += xi[u]; s1i = fma(di, xi[u], s1i); }
            }
        }
    }
    red[ry][0][cx] = s0r; red[ry][1][cx] = s0i; red[ry][2][cx] = s1r; red[ry][3][cx] = s1i;
    __syncthreads();
    if (ry != 0 || j >= nx) return;
    s0r = s0i = s1r = s1i = 0.0;
#pragma unroll
    for (int r = 0; r < RY; ++r) { s0r += red[r][0][cx]; s0i += red[r][1][cx]; s1r += red[r][2][cx]; s1i += red[r][3][cx]; }
    const double sii = (double)ny * ((double)ny * (double)ny - 1.0) / 12.0;
    const double c1r = (kind == 2 && ny > 1) ? s1r / sii : 0.0, c1i = (kind == 2 && ny > 1) ? s1i / sii : 0.0;
    double* c = coef + (b * nx + j) * 6;
    c[0] = s0r / (double)ny - c1r * ibar; c[1] = s0i / (double)ny - c1i * ibar;
    c[2] = c1r; c[3] = c1i; c[4] = 0.0; c[5] = 0.0;
}

template <typename T, bool CPLX>
__global__ void __launch_bounds__(256) detrend_apply_kernel(const void* in, void* out, long long ny, long long nx, const double* coef) {
    const long long b = blockIdx.y;
    const long long total = ny * nx;
    const double* c = coef + b * 6;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e / nx;
        const long long j = e - i * nx;
        const long long off = b * total + e;
        if (CPLX) {
            C2<T> v = reinterpret_cast<const C2<T>*>(in)[off];
            v.re = (T)((double)v.re - (c[0] + c[2] * (double)i + c[4] * (double)j));
            v.im = (T)((double)v.im - (c[1] + c[3] * (double)i + c[5] * (double)j));
            reinterpret_cast<C2<T>*>(out)[off] = v;
        } else {
            T v = reinterpret_cast<const T*>(in)[off];
            v = (T)((double)v - (c[0] + c[2] * (double)i + c[4] * (double)j));
            reinterpret_cast<T*>(out)[off] = v;
        }
    }
}

// ---- 3-D blocks (xrft/detrend.py:116-138: least-squares hyperplane a0 + a1 i + a2 j + a3 k over a (n0, n1, n2) block).
// acc[block][8] += { sum x, sum (i-ibar) x, sum (j-jbar) x, sum (k-kbar) x } as (re, im) pairs; rows r = i*n1 + j of
// length n2 are dealt round-robin to the blocks of the grid's x dimension.
template <typename T, bool CPLX>
__global__ void __launch_bounds__(256) block3_moments_kernel(const void* in, long long n0, long long n1, long long n2, double* acc) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);
    const long long b = blockIdx.y, rows = n0 * n1;
    const double ibar = 0.5 * (double)(n0 - 1), jbar = 0.5 * (double)(n1 - 1);
    const T kbar = (T)(0.5 * (double)(n2 - 1));
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const long long i = r / n1, j = r - i * n1;
        const long long base = (b * rows + r) * n2;
        T p0r = 0, p0i = 0, pkr = 0, pki = 0;
        double q0r = 0, q0i = 0;
        int cnt = 0;
        for (long long k = threadIdx.x; k < n2; k += blockDim.x) {
            T xr, xi = (T)0;
            if (CPLX) { C2<T> v = reinterpret_cast<const C2<T>*>(in)[base + k]; xr = v.re; xi = v.im; }
            else xr = reinterpret_cast<const T*>(in)[base + k];
            const T dk = (T)k - kbar;
            p0r += xr; pkr += dk * xr;
            if (CPLX) { p0i += xi; pki += dk * xi; }
            if (++cnt == 64) {
                q0r += (double)p0r; q0i += (double)p0i; s[6] += (double)pkr; s[7] += (double)pki;
                p0r = p0i = pkr = pki = (T)0; cnt = 0;
            }
        }
        q0r += (double)p0r; q0i += (double)p0i; s[6] += (double)pkr; s[7] += (double)pki;
        s[0] += q0r; s[1] += q0i;
        s[2] += ((double)i - ibar) * q0r; s[3] += ((double)i - ibar) * q0i;
        s[4] += ((double)j - jbar) * q0r; s[5] += ((double)j - jbar) * q0i;
    }
    block_sum<8>(s, red);
    if (threadIdx.x == 0)  // per-chunk partial sums, added in order by the finalize kernel (deterministic)
        for (int k = 0; k < 8; ++k) acc[(b * gridDim.x + blockIdx.x) * 8 + k] = s[k];
}

// the centred regressors of a full grid are mutually orthogonal: four independent ratios (constant: only the mean)
static __global__ void finalize_coef3_kernel(const double* part, double* coef, long long batch, long long n0, long long n1, long long n2, int kind, int nchunk) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ch = 0; ch < nchunk; ++ch)
        for (int k = 0; k < 8; ++k) acc[k] += part[(b * nchunk + ch) * 8 + k];
    const double n = (double)n0 * (double)n1 * (double)n2;
    const double bar[3] = {0.5 * (double)(n0 - 1), 0.5 * (double)(n1 - 1), 0.5 * (double)(n2 - 1)};
    const double len[3] = {(double)n0, (double)n1, (double)n2};
    for (int c = 0; c < 2; ++c) {
        double c0 = acc[c] / n;
        for (int a = 0; a < 3; ++a) {
            double sl = 0.0;
            if (kind == 2 && len[a] > 1.0) sl = acc[2 + 2 * a + c] / (n * (len[a] * len[a] - 1.0) / 12.0);
            coef[b * 8 + 2 + 2 * a + c] = sl;
            c0 -= sl * bar[a];
        }
        coef[b * 8 + c] = c0;
    }
}

template <typename T, bool CPLX>
__global__ void __launch_bounds__(256) detrend3_apply_kernel(const void* in, void* out, long long n0, long long n1, long long n2, const double* coef) {
    const long long b = blockIdx.y, rows = n0 * n1;
    const double* c = coef + b * 8;
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const long long i = r / n1, j = r - i * n1;
        const double tr = c[0] + c[2] * (double)i + c[4] * (double)j, ti = c[1] + c[3] * (double)i + c[5] * (double)j;
        const long long base = (b * rows + r) * n2;
        for (long long k = threadIdx.x; k < n2; k += blockDim.x) {
            if (CPLX) {
                C2<T> v = reinterpret_cast<const C2<T>*>(in)[base + k];
                v.re = (T)((double)v.re - (tr + c[6] * (double)k));
                v.im = (T)((double)v.im - (ti + c[7] * (double)k));
                reinterpret_cast<C2<T>*>(out)[base + k] = v;
            } else {
                const T v = reinterpret_cast<const T*>(in)[base + k];
                reinterpret_cast<T*>(out)[base + k] = (T)((double)v - (tr + c[6] * (double)k));
            }
        }
    }
}

// |a|^2 * scale (real result) or a * conj(b) * scale (complex result) of already transformed fields: the tail of
// power_spectrum / cross_spectrum (xrft.py:740, 825) for transforms composed of several plans (more than two axes)
template <typename T, bool CROSS>
__global__ void __launch_bounds__(256) spectrum_tail_kernel(const C2<T>* a, const C2<T>* b, void* out, long long n, double scale) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const C2<T> x = a[e];
        if (CROSS) reinterpret_cast<C2<T>*>(out)[e] = cscale(cmulc(x, b[e]), (T)scale);
        else reinterpret_cast<T*>(out)[e] = (x.re * x.re + x.im * x.im) * (T)scale;
    }
}

// the same with the real-dim factor [1, 2, ..., 2, (1 if the real axis is even)] along one axis of [outer][na][inner] (xrft.py:673-682)
template <typename T, bool CROSS>
__global__ void __launch_bounds__(256) spectrum_tail_axis_kernel(const C2<T>* a, const C2<T>* b, void* out, long long n, double scale,
                                                                 long long na, long long inner, int last_is_one) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const long long k = (e / inner) % na;
        const T f = (T)((k == 0 || (last_is_one && k == na - 1)) ? scale : 2.0 * scale);
        const C2<T> x = a[e];
        if (CROSS) reinterpret_cast<C2<T>*>(out)[e] = cscale(cmulc(x, b[e]), f);
        else reinterpret_cast<T*>(out)[e] = (x.re * x.re + x.im * x.im) * f;
    }
}

// out[o][i][j] = in[o][src(i)][j] over [outer][n_out][inner]; src(i) = index[i], or (i - roll) mod n_in (numpy.roll) -- the
// fftshift / ifftshift of the backend object (xrft.py:446-447, 617-621) and the re-ordering of spectra that are not stored
// in fftshift order (xrft.ifft).  E = element type by size (4, 8, 16 bytes).
template <typename E>
__global__ void __launch_bounds__(256) gather_axis_kernel(const E* in, E* out, long long outer, long long n_out, long long inner, long long n_in,
                                                          const long long* index, long long roll) {
    const long long total = outer * n_out * inner;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long j = e % inner, r = e / inner, i = r % n_out, o = r / n_out;
        long long si;
        if (index) si = index[i];
        else { si = (i - roll) % n_in; if (si < 0) si += n_in; }
        out[e] = in[(o * n_in + si) * inner + j];
    }
}

// out[b][j] = (j < n_in ? in[b][j] : 0) * table[j], j < n_out: the pointwise steps of Bluestein's algorithm run through global
// memory (chirp multiply with zero padding, product with the chirp's spectrum, chirp multiply with truncation) for lengths whose
// in-tile Bluestein does not fit the LDS.  `in` real (CIN = false) or complex, table and out complex.
template <typename T, bool CIN>
__global__ void __launch_bounds__(256) table_mul_kernel(const void* in, const C2<T>* __restrict__ table, C2<T>* __restrict__ out, long long batch, long long n_in, long long n_out) {
    const long long total = batch * n_out;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long b = e / n_out, j = e - b * n_out;
        C2<T> v = mk<T>((T)0, (T)0);
        if (j < n_in) {  // (the table has min(n_in, n_out) entries)
            if (CIN) v = reinterpret_cast<const C2<T>*>(in)[b * n_in + j];
            else v = mk<T>(reinterpret_cast<const T*>(in)[b * n_in + j], (T)0);
            v = cmul(v, table[j]);
        }
        out[e] = v;
    }
}

// Detrending with the independent elements INNERMOST: data [batch][ny][nx][inner2] (inner2 counts real components: 2 per complex
// sample), one mean / least-squares plane over (ny, nx) per (batch, inner2) element (xrft/detrend.py:54-55, 100-113 for two adjacent
// axes that are not the trailing ones).  A workgroup = IB consecutive inner2 indices (lanes: contiguous) x XS columns; rows are dealt
// round-robin to the gridDim.y chunks of a slab; part[b][chunk][i2][3] = { sum d, sum (i - ibar) d, sum (j - jbar) d }, added in
// chunk order by plane_inner_finalize_kernel (no atomics: bit-reproducible).
// (round 3: ib = min(inner2, 1024) lanes across the inner index and xs = 1024 / ib column slots, chosen at launch -- fixed at 32 x 8, a
// 16-element inner dimension left half of every wave idle on 64-byte pieces: 0.86 ms for a 268-MB array)
constexpr int kInnerThreads = 1024;  // (a (y, x, t) array with few inner elements is ONE tile: at most kInnerMaxChunks = 256 workgroups share it, so they are large)
template <typename T>
__global__ void __launch_bounds__(kInnerThreads) plane_inner_moments_kernel(const T* __restrict__ in, long long ny, long long nx, long long inner2, double* part, int ib, int xsn, long long mid, long long b0) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);  // [xsn][3][ib]
    const int li = threadIdx.x % ib, xs = threadIdx.x / ib;
    const long long i2 = (long long)blockIdx.x * ib + li, b = b0 + blockIdx.z;  // grid = (tiles of the inner index, row chunks, batch elements b0 .. of this launch)
    const double ibar = 0.5 * (double)(ny - 1), jbar = 0.5 * (double)(nx - 1);
    const bool live = xs < xsn && i2 < inner2;
    double s0 = 0.0, si = 0.0, sj = 0.0;
    if (live) {
        for (long long i = blockIdx.y; i < ny; i += gridDim.y) {
            // (mid > 1: [batch / mid][ny][mid][nx][inner2] -- the independent elements between the two axes; element b = (outer, m))
            const T* row = in + ((((b / mid) * ny + i) * mid + b % mid) * nx) * inner2 + i2;
            // four independent loads per trip (one load in flight per thread ran the pass at a fifth of the copy rate); the four partial
            // sums are combined in a fixed order
            double r0[4] = {0.0, 0.0, 0.0, 0.0}, rj[4] = {0.0, 0.0, 0.0, 0.0};
            for (long long j0 = xs; j0 < nx; j0 += 4 * xsn) {
                T v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const long long j = j0 + (long long)u * xsn; v[u] = j < nx ? row[j * inner2] : (T)0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    r0[u] += (double)v[u];
                    rj[u] = fma((double)(j0 + (long long)u * xsn) - jbar, (double)v[u], rj[u]);
                }
            }
            const double r0s = (r0[0] + r0[1]) + (r0[2] + r0[3]), rjs = (rj[0] + rj[1]) + (rj[2] + rj[3]);
            s0 += r0s; sj += rjs;
            si = fma((double)i - ibar, r0s, si);
        }
    }
    if (xs < xsn) { red[(xs * 3 + 0) * ib + li] = s0; red[(xs * 3 + 1) * ib + li] = si; red[(xs * 3 + 2) * ib + li] = sj; }
    __syncthreads();
    if (xs == 0 && i2 < inner2) {
        for (int k = 1; k < xsn; ++k) { s0 += red[(k * 3 + 0) * ib + li]; si += red[(k * 3 + 1) * ib + li]; sj += red[(k * 3 + 2) * ib + li]; }
        double* dst = part + ((b * gridDim.y + blockIdx.y) * inner2 + i2) * 3;
        dst[0] = s0; dst[1] = si; dst[2] = sj;
    }
}
// coef[b][i2] = { c0, c1, c2 }: trend = c0 + c1 i + c2 j (kind 1: the mean)
static __global__ void plane_inner_finalize_kernel(const double* part, double* coef, long long batch, long long ny, long long nx, long long inner2, int nchunk, int kind) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * inner2) return;
    const long long b = e / inner2, i2 = e - b * inner2;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) {
        const double* src = part + ((b * nchunk + ch) * inner2 + i2) * 3;
        a0 += src[0]; a1 += src[1]; a2 += src[2];
    }
    const double n = (double)ny * (double)nx, ibar = 0.5 * (double)(ny - 1), jbar = 0.5 * (double)(nx - 1);
    const double sii = (double)nx * (double)ny * ((double)ny * (double)ny - 1.0) / 12.0, sjj = (double)ny * (double)nx * ((double)nx * (double)nx - 1.0) / 12.0;
    double c1 = 0.0, c2 = 0.0;
    if (kind == 2) { if (ny > 1) c1 = a1 / sii; if (nx > 1) c2 = a2 / sjj; }
    coef[e * 3] = a0 / n - c1 * ibar - c2 * jbar;
    coef[e * 3 + 1] = c1;
    coef[e * 3 + 2] = c2;
}
// A workgroup walks a CONTIGUOUS range of rows (b, i); the coefficients of the batch element it is in sit in LDS (three float64 per inner
// index: read from memory per element they were six times the data), a thread's position (j, i2) advances by additions (no division per
// element), four loads in flight per thread.  lds_coef = 0: inner2 too large for the LDS, the coefficients come from memory.
template <typename T>
__global__ void __launch_bounds__(256) plane_inner_apply_kernel(const T* __restrict__ in, T* __restrict__ out, const double* __restrict__ coef, long long batch, long long ny, long long nx, long long inner2, int lds_coef, long long mid) {
    XRFT_DYN_SMEM(smem_raw);
    double* cl = reinterpret_cast<double*>(smem_raw);  // [inner2][3]
    const long long rowlen = nx * inner2;  // (beyond 2^32 for a long inner extent: detrend along time of a (1000, 4096, 2048) array)
    const unsigned in2 = (unsigned)inner2;
    const long long rows = batch * ny, per = (rows + gridDim.x - 1) / gridDim.x;
    const long long r_lo = (long long)blockIdx.x * per, r_hi = r_lo + per < rows ? r_lo + per : rows;
    const unsigned dj = 256u / in2, di = 256u % in2;  // a step of 256 elements in (j, i2)
    long long bcur = -1;
    for (long long r = r_lo; r < r_hi; ++r) {  // memory rows: (outer, i, m) with mid > 1 (batch counts (outer, m) pairs), else (b, i)
        const long long rq = r / mid, m_ = r - rq * mid, bo = rq / ny;
        const long long b = bo * mid + m_;
        const double fi = (double)(rq - bo * ny);
        const double* cb = coef + b * inner2 * 3;
        if (lds_coef && b != bcur) {
            __syncthreads();  // (the previous batch element's rows are done with the table)
            for (unsigned k = threadIdx.x; k < 3u * in2; k += 256) cl[k] = cb[k];
            __syncthreads();
            bcur = b;
        }
        const double* ct = lds_coef ? cl : cb;
        const T* src = in + r * rowlen;
        T* dst = out + r * rowlen;
        unsigned j = threadIdx.x / in2, i2 = threadIdx.x - j * in2;
        for (long long e0 = threadIdx.x; e0 < rowlen; e0 += 4 * 256) {
            T v[4];
            unsigned jj[4], ii[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long e = e0 + u * 256;
                jj[u] = j; ii[u] = i2;
                v[u] = e < rowlen ? src[e] : (T)0;
                j += dj; i2 += di;
                if (i2 >= in2) { i2 -= in2; ++j; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long e = e0 + u * 256;
                if (e < rowlen) {
                    const double* c = ct + (size_t)ii[u] * 3;
                    dst[e] = (T)((double)v[u] - (c[0] + c[1] * fi + c[2] * (double)jj[u]));
                }
            }
        }
    }
}

// out[e] = (B) in[e]: float32 <-> float64 (complex data: twice the elements)
template <typename A, typename B>
__global__ void __launch_bounds__(256) convert_kernel(const A* __restrict__ in, B* __restrict__ out, long long n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) out[e] = (B)in[e];
}

// out[e] = arg(a[e]) in [-pi, pi] (numpy.angle of a stored cross spectrum: xrft.cross_phase, xrft/xrft.py:838-874, where the fused
// plans cannot take the phase in their epilogue)
template <typename T>
__global__ void __launch_bounds__(256) angle_kernel(const C2<T>* __restrict__ a, T* __restrict__ out, long long n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        out[e] = (T)atan2((double)a[e].im, (double)a[e].re);
}

// out[o][i] = scale * sum_k in[o][k][i] over [outer][n][inner] (complex data: inner counts real components), the terms added in
// the order k = 0, 1, ... in float64: a mean / sum over a batch dimension whose result does not depend on how the launch was
// scheduled (the reference's users average isotropic spectra over the batch: xrft/tests/test_xrft.py:1011-1013, `.mean("d0")`).
template <typename T>
__global__ void __launch_bounds__(256) reduce_axis_kernel(const T* __restrict__ in, T* __restrict__ out, long long outer, long long n, long long inner, double scale) {
    const long long total = outer * inner;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long o = e / inner, i = e - o * inner;
        const T* src = in + o * n * inner + i;
        double acc = 0.0;
        for (long long k = 0; k < n; ++k) acc += (double)src[k * inner];
        out[e] = (T)(acc * scale);
    }
}

// Radial bin sums of a stored spectrum (xrft.isotropize, xrft.py:948-1010; _groupby_bins_agg / _binned_agg :877-945),
// BIT-REPRODUCIBLE: floating-point atomics would make a sum depend on the order in which waves arrive.  A workgroup owns one
// contiguous chunk of a slab and makes two sweeps over it: (1) the largest exponent per bin (atomicMax on the high word of the
// float64 magnitude: order-independent), (2) every value converted to int64 fixed point FR bits below its bin's exponent bound
// and added with INTEGER atomics (exact, order-independent).  The chunk's sums go to part[slab][chunk][bin]; the chunks are
// added in order by iso_reduce_kernel.  Bins outside [b0, b0 + nb) are skipped (a window of the bins per launch when the
// tables do not fit the LDS).  The spectrum is stored with rows / columns rotated by sy / sx (fftshift); binmap is indexed by
// unshifted frequencies.  A chunk holds <= 2^17 elements: |sum| < 2^(FR + 1 + 17) = 2^62 (also summed over the copies).
constexpr int kIsoFR = 44;
// rounds to nearest (a truncating shift biases a sum of positive powers by up to 2^-45 of the bin's bound per sample);
// non-finite values give 0 here: they are recorded in the bin's flag word instead (iso_nonfinite_flags)
__device__ __forceinline__ long long iso_fixed(double v, int eb) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const int ex = (int)((bits >> 52) & 0x7ffull);
    if (ex == 0 || ex == 0x7ff) return 0;  // zero / denormal; inf / nan
    const long long m = (long long)((bits & 0xfffffffffffffull) | 0x10000000000000ull);  // |v| = m 2^(ex - 1075)
    const int sh = eb - ex + (52 - kIsoFR);                                               // q = m >> sh, sh >= 8 (ex <= eb)
    const long long q = sh < 62 ? ((m + (1ll << (sh - 1))) >> sh) : 0;
    return (bits >> 63) ? -q : q;
}
// what a non-finite sample does to an IEEE sum: bit 0 +inf seen, bit 1 -inf seen, bit 2 nan seen (0 for a finite value)
__device__ __forceinline__ unsigned iso_nonfinite_flags(double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (((bits >> 52) & 0x7ffull) != 0x7ffull) return 0u;
    if (bits & 0xfffffffffffffull) return 4u;
    return (bits >> 63) ? 2u : 1u;
}
// the finite part of a sum combined with the flags of its non-finite members, as IEEE addition would have given it
__device__ __forceinline__ double iso_apply_flags(double finite_sum, unsigned fl) {
    if (fl == 0u) return finite_sum;
    if ((fl & 4u) || (fl & 3u) == 3u) return __longlong_as_double(0x7ff8000000000000ll);
    return __longlong_as_double((fl & 1u) ? 0x7ff0000000000000ll : (long long)0xfff0000000000000ull);
}

// Neighbouring samples of a row mostly fall into the same radial bin (a 64-lane wave touches ~4 bins at 1440 x 720: 16-way
// serialised LDS atomics), so the tables exist in `ncopy` copies (a power of two, as many as fit 64 KB), lane l uses copy
// l % ncopy; the copies' maxima are merged before the second sweep, their (exact, integer) sums after it.
template <typename T, bool CPLX>
__global__ void __launch_bounds__(256) radial_binsum_det_kernel(const void* in, const int* __restrict__ binmap, long long total, int nxo, int ny,
                                                                int sy, int sx, int b0, int nb, int nbins, int ncopy, double* part) {
    XRFT_DYN_SMEM(smem_raw);
    constexpr int HW = CPLX ? 2 : 1;
    unsigned long long* acc_all = reinterpret_cast<unsigned long long*>(smem_raw);       // [ncopy][nb][HW]
    unsigned* bmax_all = reinterpret_cast<unsigned*>(acc_all + (size_t)ncopy * nb * HW);  // [ncopy][nb]: high word of the largest FINITE magnitude
    unsigned* nfl = bmax_all + (size_t)ncopy * nb;                                        // [nb]: non-finite members (3 bits per component), one copy: rare
    for (int i = threadIdx.x; i < ncopy * nb * HW; i += blockDim.x) acc_all[i] = 0ull;
    for (int i = threadIdx.x; i < ncopy * nb; i += blockDim.x) bmax_all[i] = 0u;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) nfl[i] = 0u;
    const int cp = (int)(threadIdx.x & (unsigned)(ncopy - 1));
    unsigned long long* acc = acc_all + (size_t)cp * nb * HW;
    unsigned* bmax = bmax_all + (size_t)cp * nb;
    __syncthreads();
    const long long b = blockIdx.y;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    const long long e0 = (long long)blockIdx.x * per;
    const long long e1 = e0 + per < total ? e0 + per : total;
    const T* __restrict__ src = reinterpret_cast<const T*>(in) + b * total * HW;
    // (row, column) of the thread's first sample by one division, then advanced by the block size: no division per sample
    const long long ef = e0 + threadIdx.x;
    const int rf = (int)(ef / nxo), cf = (int)(ef - (long long)rf * nxo), step_r = (int)(blockDim.x / nxo), step_c = (int)(blockDim.x % nxo);
    auto bin_at = [&](int r, int c) -> int {  // r, c: storage position; the map is indexed by unshifted frequencies
        r -= sy; if (r < 0) r += ny;
        c -= sx; if (c < 0) c += nxo;
        const int bin = binmap[(long long)r * nxo + c] - b0;
        return (bin >= 0 && bin < nb) ? bin : -1;
    };
    // both sweeps in batches of U samples per thread: the U bin-map loads and the U value loads of a batch are independent and in
    // flight together (one sample at a time, map load -> value load, the kernel was latency-bound: 5.5 us per 1440 x 720 slab)
    constexpr int U = 8;
    for (int sweep = 0; sweep < 2; ++sweep) {
        int r = rf, c = cf;
        for (long long eb0 = ef; eb0 < e1; eb0 += (long long)U * blockDim.x) {
            int bins[U];
            double vr[U], vi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long e = eb0 + (long long)u * blockDim.x;
                bins[u] = -1; vr[u] = 0.0; vi[u] = 0.0;
                if (e < e1) {
                    bins[u] = bin_at(r, c);
                    if (CPLX) { vr[u] = (double)src[2 * e]; vi[u] = (double)src[2 * e + 1]; } else vr[u] = (double)src[e];
                }
                r += step_r; c += step_c;
                if (c >= nxo) { c -= nxo; ++r; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int bin = bins[u];
                if (bin < 0) continue;
                if (sweep == 0) {
                    // inf / nan members do not set the scale of the bin's finite sum: they go to the flag word and are put back
                    // at the end the way IEEE addition treats them (xrft.py:895-906 sums in floating point: a NaN poisons its bin)
                    const unsigned fl = iso_nonfinite_flags(vr[u]) | (CPLX ? iso_nonfinite_flags(vi[u]) << 3 : 0u);
                    if (fl) atomicOr(&nfl[bin], fl);
                    const double mr = (fl & 7u) ? 0.0 : fabs(vr[u]), mi = (!CPLX || (fl >> 3)) ? 0.0 : fabs(vi[u]);
                    atomicMax(&bmax[bin], (unsigned)((unsigned long long)__double_as_longlong(mr + mi) >> 32));
                } else {
                    const int eb = (int)(bmax_all[bin] >> 20);  // biased exponent of the bound: |v| <= mag < 2^(eb - 1022)
                    atomicAdd(&acc[HW * bin], (unsigned long long)iso_fixed(vr[u], eb));
                    if (CPLX) atomicAdd(&acc[2 * bin + 1], (unsigned long long)iso_fixed(vi[u], eb));
                }
            }
        }
        __syncthreads();
        if (sweep == 0 && ncopy > 1) {  // one bound per bin for every copy: the maximum over the copies, kept in copy 0
            for (int i = threadIdx.x; i < nb; i += blockDim.x) {
                unsigned m = bmax_all[i];
                for (int k = 1; k < ncopy; ++k) m = max(m, bmax_all[(size_t)k * nb + i]);
                bmax_all[i] = m;
            }
            __syncthreads();
        }
    }
    double* dst = part + ((size_t)b * gridDim.x + blockIdx.x) * (size_t)nbins * HW + (size_t)b0 * HW;
    for (int i = threadIdx.x; i < nb * HW; i += blockDim.x) {
        long long sum = 0;
        for (int k = 0; k < ncopy; ++k) sum += (long long)acc_all[(size_t)k * nb * HW + i];
        dst[i] = iso_apply_flags(ldexp((double)sum, (int)(bmax_all[i / HW] >> 20) - 1023 - kIsoFR), (nfl[i / HW] >> (3 * (i % HW))) & 7u);
    }
}

// iso[slab][bin] = sum over the partial tables of the slab's units in a FIXED order (bit-reproducible): 256 threads = 4 segments
// of units x 64 bins; every segment adds its units in order, the four segment sums are combined in order
static __global__ void __launch_bounds__(256) iso_reduce_kernel(const double* __restrict__ part, double* __restrict__ iso, int upr, int nb,
                                                         const unsigned* __restrict__ tunits, int hw) {
    XRFT_DYN_SMEM(smem_raw);
    double (*seg)[64] = reinterpret_cast<double (*)[64]>(smem_raw);  // [4][64]
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6, i = blockIdx.x * 64 + lane, slab = blockIdx.y;
    // tunits (may be null): only the units tunits[bin] & 0xffff .. (tunits[bin] >> 16) - 1 wrote this bin (the gather of
    // fasty_rows_kernel leaves out the bins a unit's rows do not reach)
    int ulo = 0, uhi = upr;
    if (tunits && i < nb) { const unsigned w = tunits[i / hw]; ulo = (int)(w & 0xffffu); uhi = (int)(w >> 16); }
    // the 64 bins of a wave walk ONE range of units -- the union of theirs (neighbouring bins' ranges nearly coincide) -- so that a
    // load is 64 adjacent doubles of one unit's table; a lane skips the units outside its own range (nothing was written there)
    int wlo = (tunits && i >= nb) ? upr : ulo, whi = (tunits && i >= nb) ? 0 : uhi;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { wlo = min(wlo, (int)__shfl_xor((double)wlo, m)); whi = max(whi, (int)__shfl_xor((double)whi, m)); }
    const int per = (max(whi - wlo, 0) + 3) / 4, u0 = wlo + sg * per, u1 = min(whi, u0 + per);
    double s = 0.0;
    if (i < nb) {
        const double* src = part + (size_t)slab * upr * nb + i;
        int un = u0;
        for (; un + 8 <= u1; un += 8) {  // eight loads in flight, added in unit order (one at a time the kernel was latency-bound: 2 us per 4096^2 slab)
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (un + k >= ulo && un + k < uhi) ? src[(size_t)(un + k) * nb] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; un < u1; ++un) s += (un >= ulo && un < uhi) ? src[(size_t)un * nb] : 0.0;
    }
    seg[sg][lane] = s;
    __syncthreads();
    if (sg == 0 && i < nb) iso[(size_t)slab * nb + i] = ((seg[0][lane] + seg[1][lane]) + seg[2][lane]) + seg[3][lane];
}

}  // namespace xrft

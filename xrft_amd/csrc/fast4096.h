// fast4096.h -- the specialised kernels for BASELINE.json's headline shape: 2-D power spectrum of (nt, 4096, 4096)
// float32 with detrend + window (xrft.power_spectrum, reference xrft/xrft.py:685-750 -> fft :307-476).
//
// Design, from measurements on MI355X (scripts/ubench/*.hip, DESIGN.md "measurements"):
//   * HBM streams ~6.0 TB/s read / ~5.1 TB/s write and the Infinity Cache adds almost no bandwidth on top, so the
//     number of passes over a slab is what counts: 2 passes (rows, then columns), never 3.
//   * scattered FULL 128-byte lines write at streaming speed -> the row pass stores the half spectrum in a tiled
//     layout W[slab][tile = kx/4][i/4][kx%4][i%4] (4 columns x 4 rows x 8 B = one line per workgroup and tile), which the
//     column pass then reads as one contiguous 128 KiB block per tile.
//   * a 4096-point column of complex64 is 32 KiB, so only 4 columns fit one CU's LDS: the column pass writes
//     16-byte output segments.  With the tile -> workgroup mapping arranged so that the 8 workgroups sharing a
//     128-byte output line run on the same XCD at the same time, the XCD's L2 merges them (3.0 TB/s measured vs
//     1.0 TB/s with the naive mapping).
// Core: a 4096-point complex FFT by 256 threads, 16 points per thread held in registers, three radix-16 passes
// (decimation in frequency) with two padded, bank-conflict-free LDS exchanges; twiddles W^(u k), k = 1..15, are
// generated in registers from one table load W^u by a depth-4 product tree (no strided table gathers).
#pragma once
#include "aux_kernels.h"

namespace xrft {

typedef C2<float> cf;
struct alignas(16) F4 { float x, y, z, w; };
struct __attribute__((aligned(4))) F4u { float x, y, z, w; };  // 16-byte access at 4-byte alignment (mirror segments)

// a[k] *= w1^k for k = 1..15, powers built by a product tree of depth <= 4 (error ~ 4 ulp)
template <typename T> __device__ __forceinline__ void twiddle16(C2<T>* a, C2<T> w1) {
    C2<T> w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
    C2<T> w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
    a[1] = cmul(a[1], w1); a[2] = cmul(a[2], w2); a[3] = cmul(a[3], w3); a[4] = cmul(a[4], w4);
    a[5] = cmul(a[5], w5); a[6] = cmul(a[6], w6); a[7] = cmul(a[7], w7); a[8] = cmul(a[8], w8);
    a[9] = cmul(a[9], cmul(w8, w1)); a[10] = cmul(a[10], cmul(w8, w2)); a[11] = cmul(a[11], cmul(w8, w3));
    a[12] = cmul(a[12], cmul(w8, w4)); a[13] = cmul(a[13], cmul(w8, w5)); a[14] = cmul(a[14], cmul(w8, w6));
    a[15] = cmul(a[15], cmul(w8, w7));
}

// LDS elements one 4096-point sequence needs (16 blocks of 256 + 16 padding, = 16 x 16 runs of 17)
#define XRFT_F4096_LDS 4352
__device__ __forceinline__ int nat4096(int k) { return k + (k >> 4); }  // natural-order slot of frequency k (1 pad per 16)

// 4096-point forward FFT by a 256-thread group.  In: a[q] = x[u + 256 q].  Out: a[k3] = X[k1 + 16 k2 + 256 k3] with
// k1 = u >> 4, k2 = u & 15.  `lds` = this group's XRFT_F4096_LDS elements; every thread of the workgroup must call it
// (it contains __syncthreads()); the buffer may be reused by the caller after the trailing barrier.
__device__ __forceinline__ void fft4096_group(cf* a, int u, cf* lds, const cf* __restrict__ tw) {
    dft16(a);
    twiddle16(a, tw[u]);  // W_4096^(u k)
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * 272 + u] = a[k];
    __syncthreads();
    const int k1 = u >> 4, v = u & 15;
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = lds[k1 * 272 + v + 16 * q];
    __syncthreads();
    dft16(a);
    twiddle16(a, tw[16 * v]);  // W_256^(v k)
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k1 * 272 + k * 17 + v] = a[k];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = lds[k1 * 272 + v * 17 + q];
    dft16(a);
    __syncthreads();
}

struct Fast4096 {  // parameters shared by the two passes
    const float* in;         // [slab][4096][4096] float32
    cf* w;                   // tiled intermediate [slab][513][1024 lines][col(4)][row(4)]
    float* pt;               // line-tiled half power spectrum [slab][ky/8 (512)][tile (513)][ky%8][4]
    float* out;              // [slab][4096][4096] float32 power spectrum
    const cf* tw;            // W_4096^k
    const float* win_y;      // nullable
    const float* win_x;      // nullable
    double* rowfit;          // [slab][4096][2]: per-row mean and slope found by the row pass (detrend != none), float64
    const float* corr;       // [slab][4096][2]: wy[i] * (row fit - plane fit) as (offset, slope), from fast4096_fit_kernel
    const cf* what0;         // FFT_x(wx)[kx], kx <= 2048 (2052 entries)
    const cf* what1;         // FFT_x(wx * (j - 2047.5))[kx]
    int detrend;             // 0 none, 1 constant, 2 linear
    int nslab;
    int shift_y, shift_x;    // 0 or 2048
    float scale;
};

#define XRFT_F4096_TILES 513

// ------------------------------------------------------------------------------------------------
// row pass: 512 threads = 2 groups; group g transforms rows 4w+2g (real part) and 4w+2g+1 (imaginary part) packed
// into one complex sequence, splits the two half spectra and the workgroup stores 4 rows x 2049 columns as 513
// full 128-byte lines of the tiled intermediate.      detrend/window: xrft.py:425-433
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) fast4096_rows_kernel(Fast4096 p) {
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x, g = tid >> 8, u = tid & 255;
    const int slab = blockIdx.x >> 10, wrow = blockIdx.x & 1023;
    const int rA = 4 * wrow + 2 * g, rB = rA + 1;
    cf* mine = lds + g * XRFT_F4096_LDS;
    const float* __restrict__ srcA = p.in + ((size_t)slab * 4096 + rA) * 4096;
    const float* __restrict__ srcB = srcA + 4096;
    const float wA = p.win_y[rA], wB = p.win_y[rB];
    float xa[16], xb[16], wx[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { xa[q] = srcA[u + 256 * q]; xb[q] = srcB[u + 256 * q]; wx[q] = p.win_x[u + 256 * q]; }
    // ---- detrend, fused: no pre-pass over the slab.  Every row's own least-squares line m + s*(j - 2047.5) is found
    // here (the whole row is in this group's registers) and subtracted; it differs from the slab's plane
    // a + b*(i - 2047.5) + c*(j - 2047.5)  (xrft/detrend.py:100-113) only by a noise-sized (offset, slope) pair per row,
    // which the column pass adds back in the spectral domain:  wy[i] * (alpha_i * What0[kx] + gamma_i * What1[kx])
    // with What0 = FFT(wx), What1 = FFT(wx * (j - 2047.5)).  Because the large part of the trend is removed exactly in
    // x-space, nothing cancels catastrophically in float32; the same float32 (m, s) are used on both sides.
    // The row sums are accumulated in float64: the lowest bins see the plane through a gain of ~1e9 (sum of the window
    // times |What1[1]|), so the slope must be good to ~1e-11 -- float32 sums leave 5e-5 of max there, float64 1e-6.
    float mA = 0.f, sA = 0.f, mB = 0.f, sB = 0.f;
    if (p.detrend) {
        double p0a = 0.0, p1a = 0.0, p0b = 0.0, p1b = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double jc = (double)(u + 256 * q) - 2047.5;
            const double da = (double)xa[q], db = (double)xb[q];
            p0a += da; p1a = fma(jc, da, p1a);
            p0b += db; p1b = fma(jc, db, p1b);
        }
        struct alignas(16) D4 { double a, b, c, d; };
        D4* red = reinterpret_cast<D4*>(mine);  // 256 + 16 entries of this group's (still unused) FFT buffer
        D4 t; t.a = p0a; t.b = p1a; t.c = p0b; t.d = p1b;
        red[u] = t;
        __syncthreads();
        if (u < 16) {
            D4 acc; acc.a = acc.b = acc.c = acc.d = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { const D4 v = red[u * 16 + k]; acc.a += v.a; acc.b += v.b; acc.c += v.c; acc.d += v.d; }
            red[256 + u] = acc;
        }
        __syncthreads();
        D4 tot; tot.a = tot.b = tot.c = tot.d = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const D4 v = red[256 + k]; tot.a += v.a; tot.b += v.b; tot.c += v.c; tot.d += v.d; }
        const double inv_n = 1.0 / 4096.0, inv_sjj = 1.0 / 5726622720.0;  // sum_j (j - 2047.5)^2 = n (n^2 - 1) / 12
        const double mAd = tot.a * inv_n, mBd = tot.c * inv_n;
        const double sAd = p.detrend == 2 ? tot.b * inv_sjj : 0.0, sBd = p.detrend == 2 ? tot.d * inv_sjj : 0.0;
        mA = (float)mAd; mB = (float)mBd; sA = (float)sAd; sB = (float)sBd;  // the float32 values are what gets subtracted
        if (u == 0) {
            double* rf = p.rowfit + ((size_t)slab * 4096 + rA) * 2;
            rf[0] = mAd; rf[1] = sAd; rf[2] = mBd; rf[3] = sBd;
        }
        __syncthreads();  // the reduction scratch aliases the FFT buffer written next
    }
    // local trend (m - s*2047.5) + s*j, subtracted in float32 with hi/lo splits whose hi parts lie on a coarse
    // power-of-two grid G (G ~ 2^-20 of the trend's magnitude): x - th and the FMA with the exact product sh*j are then
    // error-free, and the lo parts are applied to the already noise-sized value, so every remaining rounding depends on
    // the data's own low bits -- no error that is coherent along a row or a column (a plain float32 evaluation leaves
    // 6e-4 of max in the ky = 0 / kx = 0 bins; this form 1e-6, like float64, at 4 float32 operations per sample).
    const double tA = (double)mA - (double)sA * 2047.5, tB = (double)mB - (double)sB * 2047.5;
    int geA, geB;
    (void)frexp(fabs(tA) + fabs((double)sA) * 4096.0, &geA);
    (void)frexp(fabs(tB) + fabs((double)sB) * 4096.0, &geB);
    const double GA = ldexp(1.0, geA - 20), rGA = ldexp(1.0, 20 - geA), GB = ldexp(1.0, geB - 20), rGB = ldexp(1.0, 20 - geB);
    const double tAq = rint(tA * rGA) * GA, tBq = rint(tB * rGB) * GB, sAq = rint((double)sA * rGA) * GA, sBq = rint((double)sB * rGB) * GB;
    const float tAh = (float)tAq, tAl = (float)(tA - tAq), sAh = (float)sAq, sAl = (float)((double)sA - sAq);
    const float tBh = (float)tBq, tBl = (float)(tB - tBq), sBh = (float)sBq, sBl = (float)((double)sB - sBq);
    cf a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float jf = (float)(u + 256 * q);
        const float va = fmaf(-sAl, jf, fmaf(-sAh, jf, xa[q] - tAh) - tAl);
        const float vb = fmaf(-sBl, jf, fmaf(-sBh, jf, xb[q] - tBh) - tBl);
        a[q] = mk<float>(va * (wx[q] * wA), vb * (wx[q] * wB));
    }
    fft4096_group(a, u, mine, p.tw);
    {
        const int k1 = u >> 4, k2 = u & 15;
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) mine[nat4096(k1 + 16 * k2 + 256 * k3)] = a[k3];
    }
    __syncthreads();
    // split: Ra[k] = (Z[k] + conj Z[N-k]) / 2, Rb[k] = (Z[k] - conj Z[N-k]) / (2i), k = u + 256 q (q < 8), and k = 2048
    cf ra[9], rb[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int k = u + 256 * q;
        if (q < 8 || u == 0) {
            const cf zk = mine[nat4096(k & 4095)];
            const cf zc = cconj(mine[nat4096((4096 - k) & 4095)]);
            ra[q] = cscale(zk + zc, 0.5f);
            rb[q] = cscale(mul_mi(zk - zc), 0.5f);
        }
    }
    __syncthreads();
    // stage the 4 x 2049 outputs as [tile][col(4)][row(4)], then write full lines
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int k = u + 256 * q;
        if (q < 8 || u == 0) {
            // line layout [col(4)][row(4)]: the column pass then reads 4 consecutive rows of its column as one 32-byte sector.
            // slot = (col ^ (tile & 3)): 4 consecutive lanes (cols of one tile) x 4 consecutive tiles hit 16 different bank pairs
            const int tl = k >> 2, sw = tl & 3;
            lds[tl * 16 + (((k & 3) ^ sw) << 2) + 2 * g] = ra[q];
            lds[tl * 16 + (((k & 3) ^ sw) << 2) + 2 * g + 1] = rb[q];
        }
    }
    if (tid < 12) {  // the 3 padding columns of tile 512 (kx = 2049..2051): keep the intermediate deterministic
        const int r = tid / 3, c = 1 + tid % 3;
        lds[512 * 16 + c * 4 + r] = mk<float>(0.f, 0.f);  // tile 512: swizzle (512 & 3) = 0, layout [col][row]
    }
    __syncthreads();
    F4* __restrict__ dst = reinterpret_cast<F4*>(p.w + ((size_t)slab * XRFT_F4096_TILES * 4096 + 4 * wrow) * 4);
    const F4* stg = reinterpret_cast<const F4*>(lds);
    for (int e = tid; e < XRFT_F4096_TILES * 8; e += 512) {
        const int tile = e >> 3, part = e & 7;  // part = col * 2 + (row pair)
        dst[(size_t)tile * (4096 * 2) + part] = stg[tile * 8 + ((((part >> 1) ^ (tile & 3)) << 1) | (part & 1))];  // tile stride = 4096*2 F4
    }
}

// ------------------------------------------------------------------------------------------------
// column pass: 1024 threads = 4 groups, one column of the tile each; persistent over tiles; |F|^2 * scale is stored
// line-tiled (full 128-byte lines); fast4096_untile_kernel turns that into the row-major, shifted, mirrored output.
// (Writing the 16-byte-per-row segments straight into the output relies on L2 write-combining, which collapses when
// 256 CUs x 128 KiB of partial lines = the whole L2 are in flight: measured 2.4x write amplification, 53% store stalls.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) fast4096_cols_kernel(Fast4096 p) {
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    float* stg = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, g = tid >> 8, u = tid & 255;
    cf* mine = lds + g * XRFT_F4096_LDS;
    const long long ntiles = (long long)p.nslab * XRFT_F4096_TILES;
    // blocks b, b+8, b+16, ... sit on one XCD (round-robin dispatch): give each run of 8 of them 8 consecutive tiles
    const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int per_round = gridDim.x;  // multiple of 64
    const long long first = (long long)((bj >> 3) * 8 + bx) * 8 + (bj & 7);
    cf a[16];
    if (first < ntiles) {
        const cf* __restrict__ src = p.w + (size_t)first * 4096 * 4 + (u >> 2) * 16 + g * 4 + (u & 3);
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = src[q * 1024];  // row u + 256 q of column g: line (i >> 2), slot [g][i & 3]
    }
    for (long long T = first; T < ntiles; T += per_round) {
        const int slab = (int)(T / XRFT_F4096_TILES), tile = (int)(T - (long long)slab * XRFT_F4096_TILES);
        if (p.detrend) {  // add back wy[i] * (row fit - plane fit) in the spectral domain (see fast4096_rows_kernel)
            const cf w0 = p.what0[4 * tile + g], w1 = p.what1[4 * tile + g];
            const float* __restrict__ cr = p.corr + ((size_t)slab * 4096 + u) * 2;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float al = cr[512 * q], ga = cr[512 * q + 1];
                a[q].re = fmaf(al, w0.re, fmaf(ga, w1.re, a[q].re));
                a[q].im = fmaf(al, w0.im, fmaf(ga, w1.im, a[q].im));
            }
        }
        fft4096_group(a, u, mine, p.tw);
        {   // power, staged column-major [g][ky] with the conflict-free 17/16 padding
            const int k1 = u >> 4, k2 = u & 15;
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) {
                const int ky = k1 + 16 * k2 + 256 * k3;
                stg[g * XRFT_F4096_LDS + nat4096(ky)] = (a[k3].re * a[k3].re + a[k3].im * a[k3].im) * p.scale;
            }
        }
        if (T + per_round < ntiles) {  // late prefetch: the FFT registers are dead; the next tile loads while this one is stored
            const cf* __restrict__ src = p.w + (size_t)(T + per_round) * 4096 * 4 + (u >> 2) * 16 + g * 4 + (u & 3);
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = src[q * 1024];
        }
        __syncthreads();
        // line-tiled store: 8 consecutive lanes (rows ky..ky+7 of this tile) fill one 128-byte line
        F4* __restrict__ pt = reinterpret_cast<F4*>(p.pt) + (size_t)slab * 512 * XRFT_F4096_TILES * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ky = tid + 1024 * r;
            const int s = nat4096(ky);
            F4 d;
            d.x = stg[s]; d.y = stg[XRFT_F4096_LDS + s]; d.z = stg[2 * XRFT_F4096_LDS + s]; d.w = stg[3 * XRFT_F4096_LDS + s];
            pt[((size_t)(ky >> 3) * XRFT_F4096_TILES + tile) * 8 + (ky & 7)] = d;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// untile + fftshift + Hermitian mirror: a workgroup owns 8 rows ky0..ky0+7 of the half spectrum (one contiguous
// 65.7 KB read), writes them as the direct part of output rows ky (kx = 0..2048) and, reversed, as the mirror part
// of output rows -ky (kx = 4095..2049); every run is contiguous, aligned quads go out as 16-byte stores.
//   xrft.py:446-447 (fftshift); the mirror is the Hermitian symmetry of the transform of a real field.
// ------------------------------------------------------------------------------------------------
#define XRFT_UNTILE_LD 2052  // floats per staged row (2049 used)
__global__ void __launch_bounds__(256) fast4096_untile_kernel(Fast4096 p) {
    XRFT_DYN_SMEM(smem_raw);
    float* rows = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x;
    const int slab = blockIdx.x >> 9, kb = blockIdx.x & 511;
    const F4* __restrict__ src = reinterpret_cast<const F4*>(p.pt) + ((size_t)slab * 512 + kb) * XRFT_F4096_TILES * 8;
    for (int e = tid; e < XRFT_F4096_TILES * 8; e += 256) {
        const F4 v = src[e];
        const int tile = e >> 3, r = e & 7;
        float* d = rows + r * XRFT_UNTILE_LD + tile * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    float* __restrict__ out = p.out + (size_t)slab * 4096 * 4096;
    const int sx = p.shift_x;
    for (int r = 0; r < 8; ++r) {
        const int ky = kb * 8 + r;
        const float* row = rows + r * XRFT_UNTILE_LD;
        float* drow = out + (size_t)((ky + p.shift_y) & 4095) * 4096;
        float* mrow = out + (size_t)(((4096 - ky) + p.shift_y) & 4095) * 4096;
        // direct: destination column c = (kx + sx) & 4095 for kx = 0..2048
        for (int qd = tid; qd < 512; qd += 256) {  // kx = 4 qd .. 4 qd + 3 (kx <= 2047): aligned quads on both sides
            F4 v; v.x = row[4 * qd]; v.y = row[4 * qd + 1]; v.z = row[4 * qd + 2]; v.w = row[4 * qd + 3];
            *reinterpret_cast<F4*>(drow + ((4 * qd + sx) & 4095)) = v;
        }
        if (tid == 0) drow[(2048 + sx) & 4095] = row[2048];
        // mirror: value at kx goes to column (4096 - kx + sx) & 4095, kx = 1..2047.  Destination quads
        // c0 = (4096 - 4 m - 3 + sx) .. +3 hold kx = 4m+3, 4m+2, 4m+1, 4m  -> aligned when taken as kx = 4m+1..4m+4 instead:
        // columns (4096 - (4m+4) + sx) .. (4096 - (4m+1) + sx) are an aligned quad for m = 0..510 (kx <= 2044)
        for (int m = tid; m < 511; m += 256) {
            F4 v; v.x = row[4 * m + 4]; v.y = row[4 * m + 3]; v.z = row[4 * m + 2]; v.w = row[4 * m + 1];
            *reinterpret_cast<F4*>(mrow + ((4096 - (4 * m + 4) + sx) & 4095)) = v;
        }
        if (tid < 3) { const int kx = 2045 + tid; mrow[(4096 - kx + sx) & 4095] = row[kx]; }
    }
}

// ------------------------------------------------------------------------------------------------
// plane fit from the per-row fits (one 256-thread block per slab, float64): a = mean(m_i), b = slope of m_i over i,
// c = mean(s_i)  (the centred regressors of a full grid are orthogonal, so this IS the least-squares plane of
// xrft/detrend.py:100-113).  Output: corr[i] = wy[i] * (m_i - a - b (i - 2047.5),  s_i - c).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fast4096_fit_kernel(const double* rowfit, const float* win_y, float* corr, int detrend) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);
    const int slab = blockIdx.x, tid = threadIdx.x;
    const double* rf = rowfit + (size_t)slab * 4096 * 2;
    double s[3] = {0.0, 0.0, 0.0};
    for (int i = tid; i < 4096; i += 256) {
        const double m = rf[2 * i], sl = rf[2 * i + 1];
        s[0] += m;
        s[1] += ((double)i - 2047.5) * m;
        s[2] += sl;
    }
    block_sum<3>(s, red);
    __syncthreads();
    if (tid == 0) { red[0] = s[0]; red[1] = s[1]; red[2] = s[2]; }
    __syncthreads();
    const double a = red[0] / 4096.0;
    const double b = detrend == 2 ? red[1] / 5726622720.0 : 0.0;
    const double c = detrend == 2 ? red[2] / 4096.0 : 0.0;
    float* out = corr + (size_t)slab * 4096 * 2;
    for (int i = tid; i < 4096; i += 256) {
        const double wy = win_y[i];
        // what the row pass subtracted is the float32-rounded row fit; what must be subtracted is the plane
        out[2 * i] = (float)(wy * ((double)(float)rf[2 * i] - a - b * ((double)i - 2047.5)));
        out[2 * i + 1] = (float)(wy * ((double)(float)rf[2 * i + 1] - c));
    }
}

}  // namespace xrft

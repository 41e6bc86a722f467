// fastp2.h -- the specialised kernels for 2-D float32 spectra whose two transform lengths are 256, 512, 1024, 2048 or
// 4096 (BASELINE.json's headline shape (nt, 4096, 4096) and the other power-of-two slabs), with detrend + window
// (xrft.power_spectrum, reference xrft/xrft.py:685-750 -> fft :307-476).
//
// Design, from measurements on MI355X (scripts/ubench/*.hip, DESIGN.md "measurements"):
//   * HBM streams ~6.0 TB/s read / ~5.1 TB/s write and the Infinity Cache adds almost no bandwidth on top, so the
//     number of passes over a slab is what counts: 2 FFT passes (rows, then columns), never 3.
//   * scattered FULL 128-byte lines write at streaming speed, anything narrower does not -> the row pass stores the
//     half spectrum in a tiled layout W[slab][tile = kx/4][i/4][kx%4][i%4] (4 columns x 4 rows x 8 B = one line), which
//     the column pass reads as contiguous blocks; the column pass stores |F|^2 line-tiled and a streaming kernel
//     produces the row-major, shifted, mirrored output.
// Core: an N-point complex FFT (N = 256 R3, R3 = 1, 2, 4, 8, 16; R3 = 1 keeps the exchange and skips the butterfly) by N/16 threads, 16 points per thread held in registers,
// radix 16 x 16 x R3 (decimation in frequency) with two padded LDS exchanges; twiddles W^(u k), k = 1..15, are
// generated in registers from one table load W^u by a depth-4 product tree (no strided table gathers).
#pragma once
#include <type_traits>
#include "aux_kernels.h"

namespace xrft {

typedef C2<float> cf;
struct alignas(16) F4 { float x, y, z, w; };

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

template <int N> struct P2 {
    static_assert(N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096, "radix 16 x 16 x {1, 2, 4, 8, 16}");
    static constexpr int NT = N / 16;    // threads per sequence
    static constexpr int R3 = N / 256;   // last radix
    static constexpr int NB = 16 / R3;   // last-pass butterflies per thread
    static constexpr int S1 = NT + R3;   // exchange 1: 16 blocks of NT, padded so that the strided reads spread over the banks
    static constexpr int RP = R3 + 1;    // exchange 2: runs of R3, padded to an odd length
    static constexpr int S2 = 16 * RP;   // = 16 (mod 32): two half-waves read disjoint bank sets
    static constexpr int LDS = N + 256;  // elements one sequence needs (>= 16 S1, = 16 S2, > natural-order slots)
};
__device__ __forceinline__ int nat16(int k) { return k + (k >> 4); }  // natural-order slot of frequency k (1 pad per 16)

// N-point forward FFT by an N/16-thread group.  In: a[q] = x[u + NT q].  Out: a[b R3 + k3] = X[k1 + 16 k2 + 256 k3] with
// k1 = pr >> 4, k2 = pr & 15, pr = u + NT b  (b < 16 / R3).  `lds` = this group's P2<N>::LDS elements; every thread of
// the workgroup must call it (it contains __syncthreads()); the buffer may be reused after the trailing barrier.
// tw2: LDS table of the second stage's factors, tw2[k * R3 + v] = W_(N/16)^(v k) (15 LDS reads instead of a 56-instruction
// product tree per thread; the passes are VALU-bound, the LDS pipe is 15 % busy)
template <int N> __device__ __forceinline__ void fft_p2_group(cf* a, int u, cf* lds, const cf* __restrict__ tw, const cf* tw2) {
    typedef P2<N> G;
    dft16(a);
    twiddle16(a, tw[u]);  // W_N^(u k)
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * G::S1 + u] = a[k];
    __syncthreads();
    const int k1 = u / G::R3, v = u % G::R3;
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = lds[k1 * G::S1 + v + G::R3 * q];
    __syncthreads();
    dft16(a);
    if (G::R3 > 1) {
        if (tw2) {
#pragma unroll
            for (int k = 1; k < 16; ++k) a[k] = cmul(a[k], tw2[k * G::R3 + v]);  // W_(N/16)^(v k)
        } else {
            twiddle16(a, tw[16 * v]);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k1 * G::S2 + k * G::RP + v] = a[k];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < G::NB; ++b) {
        const int pr = u + G::NT * b;
        const cf* s = lds + (pr >> 4) * G::S2 + (pr & 15) * G::RP;
#pragma unroll
        for (int e = 0; e < G::R3; ++e) a[b * G::R3 + e] = s[e];
        dft_r<float, G::R3>(a + b * G::R3);
    }
    __syncthreads();
}

// fill the stage-2 table (16 * R3 entries) from the W_N^k table; visible after the next barrier
template <int N> __device__ __forceinline__ void fill_tw2(cf* tw2, const cf* __restrict__ tw, int tid, int nthreads) {
    typedef P2<N> G;
    for (int e = tid; e < 16 * G::R3; e += nthreads) {
        const int k = e / G::R3, v = e % G::R3;
        tw2[e] = tw[16 * v * k];  // W_N^(16 v k), 16 v k < N
    }
}

struct FastP2 {  // parameters shared by the passes
    const float* in;         // [slab][ny][nx] float32
    cf* w;                   // tiled intermediate [slab][ntile_pad][ny/4 lines][col(4)][row(4)]
    float* pt;               // line-tiled half power spectrum [slab][ky/8][ntile_pad][ky%8][4]
    float* out;              // [slab][ny][nx] float32 power spectrum
    const cf* tw_x;          // W_nx^k
    const cf* tw_y;          // W_ny^k
    const float* win_y;      // never null (ones when there is no window)
    const float* win_x;
    double* rowfit;          // [slab][ny][2]: per-row mean and slope found by the row pass (detrend != none), float64
    const float* corr;       // [slab][ny][2]: wy[i] * (row fit - plane fit) as (offset, slope), from fastp2_fit_kernel
    const cf* what0;         // FFT_x(wx)[kx], kx <= nx/2, zero-padded to 4 * ntile_pad entries
    const cf* what1;         // FFT_x(wx * (j - (nx-1)/2))[kx]
    const cf* ph_y;          // complex modes: true-phase factor per unshifted ky (times (-1)^ky for an ifftshifted input), never null
    const cf* ph_x;
    const unsigned* tcodes;  // radial bins in the column pass's own order [unit][slot(16)][column][u]: (direct + 1) | (mirror + 1) << 16
    double* iso;             // [slab][nbins] per-bin sums (ISO), zeroed by the caller
    int nbins;
    int ny, nx;
    int ntile;               // nx/8 + 1 tiles of 4 columns hold kx = 0..nx/2
    int ntile_pad;           // ntile rounded up to what one column workgroup covers
    int detrend;             // 0 none, 1 constant, 2 linear
    int nslab;
    int shift_y, shift_x;    // 0 or n/2
    int half;                // real_dim: only kx = 0..nx/2 is stored, rows of nx/2 + 1 samples, no mirror (xrft.py:400-404)
    int realdim2;            // ... and 0 < kx < nx/2 counts twice (xrft.py:673-682)
    int phase_out;           // cross_phase (xrft.py:838-874): the untile pass writes arg(F0 conj(F1) * phase) as float32
    float scale;
};

// ------------------------------------------------------------------------------------------------
// row pass: THR threads = GX groups; group g transforms rows RW w + 2g (real part) and RW w + 2g + 1 (imaginary part)
// packed into one complex sequence, splits the two half spectra, and the workgroup stores RW rows x (NX/2 + 1) columns
// as RW/4 consecutive full 128-byte lines per tile of the intermediate.      detrend/window: xrft.py:425-433
// ------------------------------------------------------------------------------------------------
template <int NX, int THR>
__global__ void __launch_bounds__(THR) fastp2_rows_kernel(FastP2 p) {
    typedef P2<NX> G;
    constexpr int NT = G::NT, GX = THR / NT, RW = 2 * GX, LB = RW / 4, NTILE = NX / 8 + 1;
    static_assert(LB >= 1, "a workgroup must own whole lines");
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x, g = tid / NT, u = tid % NT;
    const int wpr = p.ny / RW;  // workgroups per slab
    const int slab = blockIdx.x / wpr, wrow = blockIdx.x % wpr;
    const int rA = RW * wrow + 2 * g, rB = rA + 1;
    cf* mine = lds + g * G::LDS;
    constexpr int FFTW = GX * G::LDS, STGW = NTILE * LB * 16;
    cf* tw2 = lds + (FFTW > STGW ? FFTW : STGW);
    fill_tw2<NX>(tw2, p.tw_x, tid, THR);
    const float* __restrict__ srcA = p.in + ((size_t)slab * p.ny + rA) * NX;
    const float* __restrict__ srcB = srcA + NX;
    const float wA = p.win_y[rA], wB = p.win_y[rB];
    float xa[16], xb[16], wx[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { xa[q] = srcA[u + NT * q]; xb[q] = srcB[u + NT * q]; wx[q] = p.win_x[u + NT * q]; }
    // ---- detrend, fused: no pre-pass over the slab.  Every row's own least-squares line m + s*(j - jbar) is found
    // here (the whole row is in this group's registers) and subtracted; it differs from the slab's plane
    // a + b*(i - ibar) + c*(j - jbar)  (xrft/detrend.py:100-113) only by a noise-sized (offset, slope) pair per row,
    // which the column pass adds back in the spectral domain:  wy[i] * (alpha_i * What0[kx] + gamma_i * What1[kx])
    // with What0 = FFT(wx), What1 = FFT(wx * (j - jbar)).  Because the large part of the trend is removed exactly in
    // x-space, nothing cancels catastrophically in float32; the same float32 (m, s) are used on both sides.
    // The row sums are accumulated in float64: the lowest bins see the plane through a gain of ~1e9 (sum of the window
    // times |What1[1]|), so the slope must be good to ~1e-11 -- float32 sums leave 5e-5 of max there, float64 1e-6.
    constexpr double JBAR = 0.5 * (NX - 1);
    float mA = 0.f, sA = 0.f, mB = 0.f, sB = 0.f;
    if (p.detrend) {
        double p0a = 0.0, p1a = 0.0, p0b = 0.0, p1b = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double jc = (double)(u + NT * q) - JBAR;
            const double da = (double)xa[q], db = (double)xb[q];
            p0a += da; p1a = fma(jc, da, p1a);
            p0b += db; p1b = fma(jc, db, p1b);
        }
        struct alignas(16) D4 { double a, b, c, d; };
        D4* red = reinterpret_cast<D4*>(mine);  // NT + 16 entries of this group's (still unused) FFT buffer
        D4 t; t.a = p0a; t.b = p1a; t.c = p0b; t.d = p1b;
        red[u] = t;
        __syncthreads();
        if (u < 16) {
            D4 acc; acc.a = acc.b = acc.c = acc.d = 0.0;
#pragma unroll
            for (int k = 0; k < NT / 16; ++k) { const D4 v = red[u * (NT / 16) + k]; acc.a += v.a; acc.b += v.b; acc.c += v.c; acc.d += v.d; }
            red[NT + u] = acc;
        }
        __syncthreads();
        D4 tot; tot.a = tot.b = tot.c = tot.d = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const D4 v = red[NT + k]; tot.a += v.a; tot.b += v.b; tot.c += v.c; tot.d += v.d; }
        constexpr double inv_n = 1.0 / NX, inv_sjj = 12.0 / ((double)NX * ((double)NX * NX - 1.0));  // sum_j (j - jbar)^2 = n (n^2 - 1) / 12
        const double mAd = tot.a * inv_n, mBd = tot.c * inv_n;
        const double sAd = p.detrend == 2 ? tot.b * inv_sjj : 0.0, sBd = p.detrend == 2 ? tot.d * inv_sjj : 0.0;
        mA = (float)mAd; mB = (float)mBd; sA = (float)sAd; sB = (float)sBd;  // the float32 values are what gets subtracted
        if (u == 0) {
            double* rf = p.rowfit + ((size_t)slab * p.ny + rA) * 2;
            rf[0] = mAd; rf[1] = sAd; rf[2] = mBd; rf[3] = sBd;
        }
        __syncthreads();  // the reduction scratch aliases the FFT buffer written next
    }
    // local trend (m - s*jbar) + s*j, subtracted in float32 with hi/lo splits whose hi parts lie on a coarse
    // power-of-two grid (~ 2^-20 of the trend's magnitude): x - th and the FMA with the exact product sh*j are then
    // error-free, and the lo parts are applied to the already noise-sized value, so every remaining rounding depends on
    // the data's own low bits -- no error that is coherent along a row or a column (a plain float32 evaluation leaves
    // 6e-4 of max in the ky = 0 / kx = 0 bins; this form 1e-6, like float64, at 4 float32 operations per sample).
    const double tA = (double)mA - (double)sA * JBAR, tB = (double)mB - (double)sB * JBAR;
    int geA, geB;
    (void)frexp(fabs(tA) + fabs((double)sA) * (double)NX, &geA);
    (void)frexp(fabs(tB) + fabs((double)sB) * (double)NX, &geB);
    const double GA = ldexp(1.0, geA - 20), rGA = ldexp(1.0, 20 - geA), GB = ldexp(1.0, geB - 20), rGB = ldexp(1.0, 20 - geB);
    const double tAq = rint(tA * rGA) * GA, tBq = rint(tB * rGB) * GB, sAq = rint((double)sA * rGA) * GA, sBq = rint((double)sB * rGB) * GB;
    const float tAh = (float)tAq, tAl = (float)(tA - tAq), sAh = (float)sAq, sAl = (float)((double)sA - sAq);
    const float tBh = (float)tBq, tBl = (float)(tB - tBq), sBh = (float)sBq, sBl = (float)((double)sB - sBq);
    cf a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float jf = (float)(u + NT * q);
        const float va = fmaf(-sAl, jf, fmaf(-sAh, jf, xa[q] - tAh) - tAl);
        const float vb = fmaf(-sBl, jf, fmaf(-sBh, jf, xb[q] - tBh) - tBl);
        a[q] = mk<float>(va * (wx[q] * wA), vb * (wx[q] * wB));
    }
    fft_p2_group<NX>(a, u, mine, p.tw_x, tw2);
#pragma unroll
    for (int b = 0; b < G::NB; ++b) {
        const int pr = u + NT * b;
#pragma unroll
        for (int k3 = 0; k3 < G::R3; ++k3) mine[nat16((pr >> 4) + 16 * (pr & 15) + 256 * k3)] = a[b * G::R3 + k3];
    }
    __syncthreads();
    // split: Ra[k] = (Z[k] + conj Z[N-k]) / 2, Rb[k] = (Z[k] - conj Z[N-k]) / (2i), k = u + NT q (q < 8), and k = NX/2
    cf ra[9], rb[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int k = u + NT * q;
        if (q < 8 || u == 0) {
            const cf zk = mine[nat16(k & (NX - 1))];
            const cf zc = cconj(mine[nat16((NX - k) & (NX - 1))]);
            ra[q] = cscale(zk + zc, 0.5f);
            rb[q] = cscale(mul_mi(zk - zc), 0.5f);
        }
    }
    __syncthreads();
    // stage the RW x (NX/2 + 1) outputs as [tile][line (LB)][col(4)][row(4)], then write full lines.  The column pass reads
    // 4 consecutive rows of its column as one 32-byte sector.  slot = col ^ (tile & 3) spreads the staging writes.
    const int lb = g >> 1, r0 = 2 * (g & 1);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int k = u + NT * q;
        if (q < 8 || u == 0) {
            const int tl = k >> 2, sw = tl & 3;
            cf* d = lds + (tl * LB + lb) * 16 + (((k & 3) ^ sw) << 2) + r0;
            d[0] = ra[q];
            d[1] = rb[q];
        }
    }
    if (tid < 3 * RW) {  // the 3 padding columns of the last tile (kx = NX/2 + 1..3): keep the intermediate deterministic
        const int r = tid / 3, c = 1 + tid % 3;
        lds[((NTILE - 1) * LB + (r >> 2)) * 16 + c * 4 + (r & 3)] = mk<float>(0.f, 0.f);  // (NTILE - 1) & 3 == 0: no swizzle
    }
    __syncthreads();
    const size_t tile_stride = (size_t)p.ny * 2;  // F4 per tile: ny/4 lines of 8
    F4* __restrict__ dst = reinterpret_cast<F4*>(p.w) + (size_t)slab * p.ntile_pad * tile_stride + (size_t)wrow * LB * 8;
    const F4* stg = reinterpret_cast<const F4*>(lds);
    for (int e = tid; e < NTILE * LB * 8; e += THR) {
        const int tile = e / (LB * 8), rem = e % (LB * 8), part = rem & 7;  // part = col * 2 + (row pair)
        dst[(size_t)tile * tile_stride + rem] = stg[(e & ~7) + ((((part >> 1) ^ (tile & 3)) << 1) | (part & 1))];
    }
}

// ------------------------------------------------------------------------------------------------
// column pass: THR threads (1024; 768 when the 1024-point pass also needs room for the histogram) = GY groups, one column
// each (GY/4 tiles, one contiguous read); persistent over tile
// groups; adds the residual trend back in the spectral domain; |F|^2 * scale is stored line-tiled (full 128-byte lines).
// (Writing 16-byte-per-row segments straight into the output relies on L2 write-combining, which collapses when
// 256 CUs x 128 KiB of partial lines = the whole L2 are in flight: measured 2.4x write amplification, 53% store stalls.)
// ------------------------------------------------------------------------------------------------
// MODE 1: |F|^2 * scale (float).  MODE 0: F * scale (complex; the true-phase factors are applied by the untile kernel).
// MODE 2: cross spectrum, second of two passes -- the field-0 pass (MODE 0, scale 1) left F0 in `pt`; this pass transforms
// field 1 and replaces F0 by F0 conj(F1) * scale in place (xrft.py:825).  ISO: radial sums, MODE 1 in transform order
// straight from the registers, MODE 2 in the store loop (complex bins; the mirror contributes the conjugate).
template <int NY, int THR, int MODE, bool ISO>
__global__ void __launch_bounds__(THR) fastp2_cols_kernel(FastP2 p) {
    static_assert(MODE == 1 || !ISO || MODE == 2, "complex output has no radial reduce");
    typedef P2<NY> G;
    constexpr int NT = G::NT, GY = THR / NT, TPU = GY / 4;  // tiles per unit of work
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    float* stg = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, g = tid / NT, u = tid % NT;
    cf* mine = lds + g * G::LDS;
    const int upr = p.ntile_pad / TPU;  // units per slab
    const long long nunits = (long long)p.nslab * upr;
    // blocks b, b+8, b+16, ... sit on one XCD (round-robin dispatch): give each run of 8 of them 8 consecutive units
    const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int per_round = gridDim.x;  // multiple of 64
    const long long first = (long long)((bj >> 3) * 8 + bx) * 8 + (bj & 7);
    // unit U = tiles TPU*U .. of the [slab][ntile_pad] sequence; row i of column g: tile g>>2, line i>>2, slot [g&3][i&3]
    const size_t lane_off = (size_t)(g >> 2) * NY * 4 + (u >> 2) * 16 + (g & 3) * 4 + (u & 3);
    // radial sums (xrft.py:895-906): per-workgroup float64 histogram behind the FFT buffers, flushed when the slab changes
    double* hist = reinterpret_cast<double*>(lds + GY * G::LDS);
    // the column pass keeps the product tree for its second-stage twiddles: with one workgroup per CU the 15 extra LDS reads
    // cost more (+1 us / slab measured) than the 56 VALU instructions they replace; the row pass gains 1.2 us from the table
    const cf* tw2 = nullptr;
    int cur_slab = -1;
    constexpr int HW = MODE == 2 ? 2 : 1;  // doubles per bin
    if (ISO) for (int i = tid; i < p.nbins * HW; i += THR) hist[i] = 0.0;  // ordered before the first add by the FFT's barriers
    cf a[16];
    if (first < nunits) {
        const cf* __restrict__ src = p.w + (size_t)first * TPU * NY * 4 + lane_off;
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = src[q * NT * 4];
    }
    for (long long U = first; U < nunits; U += per_round) {
        const int slab = (int)(U / upr), unit = (int)(U - (long long)slab * upr), tile0 = unit * TPU;
        if (ISO && slab != cur_slab) {
            if (cur_slab >= 0) {
                __syncthreads();
                for (int i = tid; i < p.nbins * HW; i += THR) {
                    const double v = hist[i];
                    if (v != 0.0) { atomicAdd(&p.iso[(size_t)cur_slab * p.nbins * HW + i], v); hist[i] = 0.0; }
                }
                __syncthreads();
            }
            cur_slab = slab;
        }
        if (p.detrend) {  // add back wy[i] * (row fit - plane fit) in the spectral domain (see fastp2_rows_kernel)
            const cf w0 = p.what0[4 * tile0 + g], w1 = p.what1[4 * tile0 + g];
            const float* __restrict__ cr = p.corr + ((size_t)slab * NY + u) * 2;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float al = cr[2 * NT * q], ga = cr[2 * NT * q + 1];
                a[q].re = fmaf(al, w0.re, fmaf(ga, w1.re, a[q].re));
                a[q].im = fmaf(al, w0.im, fmaf(ga, w1.im, a[q].im));
            }
        }
        fft_p2_group<NY>(a, u, mine, p.tw_y, tw2);
        if constexpr (MODE == 1) {
            if (ISO) {  // value at (ky, kx) goes to its bin, and once more to the bin of (-ky, -kx) (Hermitian mirror of a real field)
                const unsigned* __restrict__ tc = p.tcodes + ((size_t)unit * 16 * GY + g) * NT + u;
                unsigned code[16];
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) code[sl] = tc[sl * GY * NT];
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) {
                    const float v = (a[sl].re * a[sl].re + a[sl].im * a[sl].im) * p.scale;
                    const unsigned cd = code[sl] & 0xffffu, cm = code[sl] >> 16;
                    if (cd == cm) { if (cd) atomicAdd(&hist[cd - 1], 2.0 * (double)v); }
                    else {
                        if (cd) atomicAdd(&hist[cd - 1], (double)v);
                        if (cm) atomicAdd(&hist[cm - 1], (double)v);
                    }
                }
            }
            const bool want_p = !ISO || p.pt != nullptr;
            // power, staged column-major [g][ky] with the conflict-free 17/16 padding
            if (want_p) {
#pragma unroll
                for (int b = 0; b < G::NB; ++b) {
                    const int pr = u + NT * b;
#pragma unroll
                    for (int k3 = 0; k3 < G::R3; ++k3) {
                        const cf v = a[b * G::R3 + k3];
                        stg[g * G::LDS + nat16((pr >> 4) + 16 * (pr & 15) + 256 * k3)] = (v.re * v.re + v.im * v.im) * p.scale;
                    }
                }
            }
            if (U + per_round < nunits) {  // late prefetch: the FFT registers are dead; the next unit loads while this one is stored
                const cf* __restrict__ src = p.w + (size_t)(U + per_round) * TPU * NY * 4 + lane_off;
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = src[q * NT * 4];
            }
            if (!want_p) continue;
            __syncthreads();
            // line-tiled store: 8 consecutive lanes (rows ky..ky+7 of one tile) fill one 128-byte line
            F4* __restrict__ pt = reinterpret_cast<F4*>(p.pt) + (size_t)slab * (NY / 8) * p.ntile_pad * 8;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int item = tid + THR * r, ky = item % NY, wt = item / NY;
                const float* s = stg + (4 * wt) * G::LDS + nat16(ky);
                F4 d;
                d.x = s[0]; d.y = s[G::LDS]; d.z = s[2 * G::LDS]; d.w = s[3 * G::LDS];
                pt[((size_t)(ky >> 3) * p.ntile_pad + tile0 + wt) * 8 + (ky & 7)] = d;
            }
            __syncthreads();
        } else {
            // complex result, staged in the group's own FFT buffer in natural order
            const float sc0 = MODE == 0 ? p.scale : 1.0f;
#pragma unroll
            for (int b = 0; b < G::NB; ++b) {
                const int pr = u + NT * b;
#pragma unroll
                for (int k3 = 0; k3 < G::R3; ++k3)
                    mine[nat16((pr >> 4) + 16 * (pr & 15) + 256 * k3)] = cscale(a[b * G::R3 + k3], sc0);
            }
            if (U + per_round < nunits) {
                const cf* __restrict__ src = p.w + (size_t)(U + per_round) * TPU * NY * 4 + lane_off;
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = src[q * NT * 4];
            }
            __syncthreads();
            // line-tiled store, 32 bytes per row and tile: 8 consecutive lanes fill two 128-byte lines
            F4* __restrict__ pt = reinterpret_cast<F4*>(p.pt) + (size_t)slab * (NY / 8) * p.ntile_pad * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int item = tid + THR * r, ky = item % NY, wt = item / NY;
                const cf* s = lds + (4 * wt) * G::LDS + nat16(ky);
                cf v[4];
                v[0] = s[0]; v[1] = s[G::LDS]; v[2] = s[2 * G::LDS]; v[3] = s[3 * G::LDS];
                F4* dst = pt + (((size_t)(ky >> 3) * p.ntile_pad + tile0 + wt) * 8 + (ky & 7)) * 2;
                if constexpr (MODE == 2) {
                    const F4 o0 = dst[0], o1 = dst[1];
                    v[0] = cscale(cmulc(mk<float>(o0.x, o0.y), v[0]), p.scale); v[1] = cscale(cmulc(mk<float>(o0.z, o0.w), v[1]), p.scale);
                    v[2] = cscale(cmulc(mk<float>(o1.x, o1.y), v[2]), p.scale); v[3] = cscale(cmulc(mk<float>(o1.z, o1.w), v[3]), p.scale);
                    if (ISO) {
                        const unsigned* __restrict__ tc = p.tcodes + ((size_t)unit * (NY * TPU) + item) * 4;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const unsigned code = tc[c], cd = code & 0xffffu, cm = code >> 16;
                            if (cd == cm) { if (cd) atomicAdd(&hist[2 * (cd - 1)], 2.0 * (double)v[c].re); }  // V + conj V
                            else {
                                if (cd) { atomicAdd(&hist[2 * (cd - 1)], (double)v[c].re); atomicAdd(&hist[2 * (cd - 1) + 1], (double)v[c].im); }
                                if (cm) { atomicAdd(&hist[2 * (cm - 1)], (double)v[c].re); atomicAdd(&hist[2 * (cm - 1) + 1], -(double)v[c].im); }
                            }
                        }
                    }
                }
                if (MODE == 0 || p.out != nullptr) {
                    F4 d0, d1;
                    d0.x = v[0].re; d0.y = v[0].im; d0.z = v[1].re; d0.w = v[1].im;
                    d1.x = v[2].re; d1.y = v[2].im; d1.z = v[3].re; d1.w = v[3].im;
                    dst[0] = d0; dst[1] = d1;
                }
            }
            __syncthreads();
        }
    }
    if (ISO && cur_slab >= 0) {
        __syncthreads();
        for (int i = tid; i < p.nbins * HW; i += THR) {
            const double v = hist[i];
            if (v != 0.0) atomicAdd(&p.iso[(size_t)cur_slab * p.nbins * HW + i], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// untile + fftshift + Hermitian mirror: a workgroup owns 8 rows ky0..ky0+7 of the half spectrum (one contiguous
// read), writes them as the direct part of output rows ky (kx = 0..nx/2) and, reversed, as the mirror part
// of output rows -ky (kx = nx-1..nx/2+1); every run is contiguous, aligned quads go out as 16-byte stores.
//   xrft.py:446-447 (fftshift); the mirror is the Hermitian symmetry of the transform of a real field.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fastp2_untile_kernel(FastP2 p) {
    XRFT_DYN_SMEM(smem_raw);
    float* rows = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x;
    const int nx = p.nx, ny = p.ny, nxh = nx >> 1, ld = nxh + 4, mx = nx - 1, my = ny - 1;
    const int gpr = ny >> 3;  // 8-row groups per slab
    const int slab = blockIdx.x / gpr, kb = blockIdx.x % gpr;
    const F4* __restrict__ src = reinterpret_cast<const F4*>(p.pt) + ((size_t)slab * gpr + kb) * p.ntile_pad * 8;
    for (int e = tid; e < p.ntile * 8; e += 256) {
        const F4 v = src[e];
        const int tile = e >> 3, r = e & 7;
        float* d = rows + r * ld + tile * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    if (p.half) {  // rows of nx/2 + 1 samples: an odd row length, so plain 4-byte stores (still whole lines per wave)
        const int w = nxh + 1;
        float* __restrict__ outh = p.out + (size_t)slab * ny * w;
        for (int e = tid; e < 8 * w; e += 256) {
            const int r = e / w, kx = e - r * w, ky = kb * 8 + r;
            float v = rows[r * ld + kx];
            if (p.realdim2 && kx != 0 && kx != nxh) v *= 2.0f;
            outh[(size_t)((ky + p.shift_y) & my) * w + kx] = v;
        }
        return;
    }
    float* __restrict__ out = p.out + (size_t)slab * ny * nx;
    const int sx = p.shift_x;
    for (int r = 0; r < 8; ++r) {
        const int ky = kb * 8 + r;
        const float* row = rows + r * ld;
        float* drow = out + (size_t)((ky + p.shift_y) & my) * nx;
        float* mrow = out + (size_t)(((ny - ky) + p.shift_y) & my) * nx;
        // direct: destination column c = (kx + sx) & mx for kx = 0..nx/2
        for (int qd = tid; qd < (nx >> 3); qd += 256) {  // kx = 4 qd .. 4 qd + 3 (kx < nx/2): aligned quads on both sides
            F4 v; v.x = row[4 * qd]; v.y = row[4 * qd + 1]; v.z = row[4 * qd + 2]; v.w = row[4 * qd + 3];
            *reinterpret_cast<F4*>(drow + ((4 * qd + sx) & mx)) = v;
        }
        if (tid == 0) drow[(nxh + sx) & mx] = row[nxh];
        // mirror: value at kx goes to column (nx - kx + sx) & mx, kx = 1..nx/2-1.  Taken as kx = 4m+1..4m+4 the destination
        // columns (nx - (4m+4) + sx) .. (nx - (4m+1) + sx) are an aligned quad for m = 0..nx/8-2 (kx <= nx/2-4)
        for (int m = tid; m < (nx >> 3) - 1; m += 256) {
            F4 v; v.x = row[4 * m + 4]; v.y = row[4 * m + 3]; v.z = row[4 * m + 2]; v.w = row[4 * m + 1];
            *reinterpret_cast<F4*>(mrow + ((nx - (4 * m + 4) + sx) & mx)) = v;
        }
        if (tid < 3) { const int kx = nxh - 3 + tid; mrow[(nx - kx + sx) & mx] = row[kx]; }
    }
}

// ------------------------------------------------------------------------------------------------
// complex variant (fft / cross_spectrum): a workgroup owns 4 rows (half of an 8-row group of the line-tiled
// intermediate, still full 128-byte lines); the mirror half is the conjugate; the true-phase factors
// exp(-i 2 pi k lag) (xrft.py:462-469) are applied here, to direct and mirrored samples alike.
// ------------------------------------------------------------------------------------------------
template <bool ANGLE>
__global__ void __launch_bounds__(256) fastp2_untile_c_kernel(FastP2 p) {
    typedef typename std::conditional<ANGLE, float, cf>::type OutT;  // one angle, or the complex sample
    auto fin = [](cf v) -> OutT { if constexpr (ANGLE) return (float)atan2((double)v.im, (double)v.re); else return v; };
    XRFT_DYN_SMEM(smem_raw);
    cf* rows = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x;
    const int nx = p.nx, ny = p.ny, nxh = nx >> 1, ld = nxh + 4, mx = nx - 1, my = ny - 1;
    const int gpr = ny >> 2;  // 4-row groups per slab
    const int slab = blockIdx.x / gpr, kq = blockIdx.x % gpr, kb = kq >> 1, half = kq & 1;
    const F4* __restrict__ src = reinterpret_cast<const F4*>(p.pt) + ((size_t)slab * (ny >> 3) + kb) * p.ntile_pad * 16 + half * 8;
    for (int e = tid; e < p.ntile * 8; e += 256) {
        const int tile = e >> 3, j = e & 7;  // j = row * 2 + column pair
        const F4 v = src[tile * 16 + j];
        cf* d = rows + (j >> 1) * ld + tile * 4 + 2 * (j & 1);
        d[0] = mk<float>(v.x, v.y); d[1] = mk<float>(v.z, v.w);
    }
    __syncthreads();
    if (p.half) {
        const int w = nxh + 1;
        OutT* __restrict__ outh = reinterpret_cast<OutT*>(p.out) + (size_t)slab * ny * w;
        for (int e = tid; e < 4 * w; e += 256) {
            const int r = e / w, kx = e - r * w, ky = kb * 8 + half * 4 + r;
            cf v = cmul(cmul(rows[r * ld + kx], p.ph_y[ky]), p.ph_x[kx]);
            if (p.realdim2 && kx != 0 && kx != nxh) v = cscale(v, 2.0f);
            outh[(size_t)((ky + p.shift_y) & my) * w + kx] = fin(v);
        }
        return;
    }
    OutT* __restrict__ out = reinterpret_cast<OutT*>(p.out) + (size_t)slab * ny * nx;
    const int sx = p.shift_x;
    for (int r = 0; r < 4; ++r) {
        const int ky = kb * 8 + half * 4 + r, nky = (ny - ky) & my;
        const cf* row = rows + r * ld;
        const cf py = p.ph_y[ky], pmy = p.ph_y[nky];
        OutT* drow = out + (size_t)((ky + p.shift_y) & my) * nx;
        OutT* mrow = out + (size_t)((nky + p.shift_y) & my) * nx;
        struct alignas(2 * sizeof(OutT)) Pair { OutT a, b; };  // two neighbours: one aligned store
        for (int m = tid; m < (nxh >> 1); m += 256) {  // direct, kx = 2m, 2m+1
            const cf v0 = cmul(cmul(row[2 * m], py), p.ph_x[2 * m]), v1 = cmul(cmul(row[2 * m + 1], py), p.ph_x[2 * m + 1]);
            Pair o; o.a = fin(v0); o.b = fin(v1);
            *reinterpret_cast<Pair*>(drow + ((2 * m + sx) & mx)) = o;
        }
        if (tid == 0) drow[(nxh + sx) & mx] = fin(cmul(cmul(row[nxh], py), p.ph_x[nxh]));
        for (int m = tid; m < (nxh >> 1) - 1; m += 256) {  // mirror of kx = 2m+2, 2m+1 at columns nx - kx: again an aligned pair
            const cf v2 = cmul(cmul(cconj(row[2 * m + 2]), pmy), p.ph_x[nx - (2 * m + 2)]);
            const cf v1 = cmul(cmul(cconj(row[2 * m + 1]), pmy), p.ph_x[nx - (2 * m + 1)]);
            Pair o; o.a = fin(v2); o.b = fin(v1);
            *reinterpret_cast<Pair*>(mrow + ((nx - (2 * m + 2) + sx) & mx)) = o;
        }
        if (tid == 0) { const int kx = nxh - 1; mrow[(nx - kx + sx) & mx] = fin(cmul(cmul(cconj(row[kx]), pmy), p.ph_x[nx - kx])); }
    }
}

// ------------------------------------------------------------------------------------------------
// plane fit from the per-row fits (one 256-thread block per slab, float64): a = mean(m_i), b = slope of m_i over i,
// c = mean(s_i)  (the centred regressors of a full grid are orthogonal, so this IS the least-squares plane of
// xrft/detrend.py:100-113).  Output: corr[i] = wy[i] * (m_i - a - b (i - ibar),  s_i - c).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fastp2_fit_kernel(const double* rowfit, const float* win_y, float* corr, int ny, int detrend) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);
    const int slab = blockIdx.x, tid = threadIdx.x;
    const double* rf = rowfit + (size_t)slab * ny * 2;
    const double ibar = 0.5 * (ny - 1), sii = (double)ny * ((double)ny * ny - 1.0) / 12.0;
    double s[3] = {0.0, 0.0, 0.0};
    for (int i = tid; i < ny; i += 256) {
        const double m = rf[2 * i], sl = rf[2 * i + 1];
        s[0] += m;
        s[1] += ((double)i - ibar) * m;
        s[2] += sl;
    }
    block_sum<3>(s, red);
    __syncthreads();
    if (tid == 0) { red[0] = s[0]; red[1] = s[1]; red[2] = s[2]; }
    __syncthreads();
    const double a = red[0] / ny;
    const double b = detrend == 2 ? red[1] / sii : 0.0;
    const double c = detrend == 2 ? red[2] / ny : 0.0;
    float* out = corr + (size_t)slab * ny * 2;
    for (int i = tid; i < ny; i += 256) {
        const double wy = win_y[i];
        // what the row pass subtracted is the float32-rounded row fit; what must be subtracted is the plane
        out[2 * i] = (float)(wy * ((double)(float)rf[2 * i] - a - b * ((double)i - ibar)));
        out[2 * i + 1] = (float)(wy * ((double)(float)rf[2 * i + 1] - c));
    }
}

}  // namespace xrft

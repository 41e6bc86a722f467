// fasty.h -- the two-pass "y first" pipeline for float32 power spectra of power-of-two slabs (256..4096 per axis):
//     pass 1  fasty_cols_kernel   FFT along y of the real columns (detrend + window fused), half spectra ky = 0..ny/2
//     [fit]   fastp2_fit_kernel   plane from the per-column fits (the same kernel as the x-first path, axes swapped)
//     pass 2  fasty_rows_kernel   FFT along x of the rows ky = 0..ny/2, |F|^2 * scale, and BOTH output rows ky and -ky
// (xrft.power_spectrum, reference xrft/xrft.py:685-750 -> fft :307-476, detrend.py:100-113, window :96-103.)
//
// Why y first: the last pass then owns complete rows of the result, so fftshift is a rotation and the Hermitian mirror
// (row -ky = row ky reversed) a reversed copy of a row that is already in LDS -- the result leaves as full, aligned
// 16-KB rows and the x-first path's third pass (untile + mirror, 101 MB of HBM traffic per 4096^2 slab) does not exist.
// The price is on the read side of pass 1: a workgroup owns CW adjacent columns (CW * 4-byte row segments).  Measured
// (scripts/ubench/yfirst.hip, profiles/r02_ubench_yfirst.txt): with the column blocks of one XCD adjacent, 32-byte
// segments stream as fast as a plain copy (27.8 vs 28.4 us per 4096^2 slab; 45.9 us with a naive block order).
//
// Every thread runs TWO transforms (one float4 load = four real columns = two packed complex columns in pass 1; two
// rows in pass 2) through one LDS buffer: the exchange of one overlaps the butterflies of the other.  Lane order inside a workgroup is (u, g) with the transform index g fastest, so that the lanes of one
// wave cover whole row segments (pass 1) / whole 128-byte lines (pass 2).
//
// Intermediate W2 (complex64), written by pass 1 in full 128-byte lines straight from the registers:
//     W2[slab][ky / RK][x / CW][set(2)][ky % RK][2 GY]      GY = pass-1 transforms per workgroup / 2, CW = 4 GY columns,
//     RK = max(1, 8 / GY) rows per line, set = (x >> 1) & 1, position in the line's row = 2 ((x % CW) / 4) + (x & 1)
// i.e. a line holds RK consecutive ky of the 2 GY columns of one set.  Pass 2 reads 2 GX >= RK consecutive ky per
// workgroup, a contiguous block.  ky runs to nrow_pad (ny/2 + 1 rounded up to a pass-2 unit); the padding rows are
// never written and never read.
#pragma once
#include "fastp2.h"

namespace xrft {

struct FastY {
    const float* in;         // [slab][ny][nx] float32
    cf* w2;                  // intermediate, see above
    float* out;              // [slab][ny][nx] float32 power spectrum (may be null with ISO)
    const cf* tw_x;          // W_nx^k
    const cf* tw_y;          // W_ny^k
    const float* win_y;      // never null (ones when there is no window)
    const float* win_x;
    double* colfit;          // [slab][nx][2]: per-column mean and slope found by pass 1 (detrend != none), float64
    const float* corr;       // [slab][nx][2]: wx[x] * (column fit - plane fit) as (offset, slope), from fastp2_fit_kernel
    const cf* what0;         // FFT_y(wy)[ky], ky < nrow_pad (zero beyond ny/2)
    const cf* what1;         // FFT_y(wy * (i - (ny-1)/2))[ky]
    const unsigned* tcodes;  // radial bins in pass 2's register order: (direct + 1) | (mirror + 1) << 16
    double* iso;             // [slab][nbins] per-bin sums (ISO), zeroed by the caller
    int nbins;
    int ny, nx;
    int nrow_pad;            // rows of W2 per slab
    int l_cw, l_rk, l_2gy;   // log2 of CW, RK, 2 GY (layout of W2, fixed by ny)
    int detrend;             // 0 none, 1 constant, 2 linear
    int nslab;
    int shift_y, shift_x;    // 0 or n/2
    float scale;
};

// phase-ablation bits for profiling builds (scripts/gpu_ablate_yf.sh compiles variants with -DXRFT_YDBG=bits); 0 in the product
#ifndef XRFT_YDBG
#define XRFT_YDBG 0
#endif

// geometry of pass 1 for NY-point columns
template <int NY> struct YCols {
    static constexpr int THR = NY >= 2048 ? 512 : 256;
    static constexpr int NT = NY / 16;
    static constexpr int GY = THR / NT;              // lockstep transform pairs per workgroup: 2, 4, 4, 8, 16
    static constexpr int CW = 4 * GY;                // real columns per workgroup
    static constexpr int RK = GY >= 8 ? 1 : 8 / GY;  // rows per line of W2
    static constexpr int LBS = RK * 2 * GY;          // complex per (row block, column block, set): 16, or 32 when GY = 16
};
constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

// per-transform LDS stride (complex): P2<N>::LDS plus a pad that puts the GX transforms of one wave on different banks
// (lanes (u, g), g fastest: without it all g collide).  ds_write_b64 and ds_read2_b64 -- what the exchanges compile to -- are
// serviced 16 lanes at a time over 32 banks, so 16 lanes = 16/GX values of u x GX transforms must cover 32 dwords:
// pad = 16/GX complex (scripts/lds_conflicts.py; PMC: 52 % of the LDS cycles were conflicts with pad = 32/GX)
template <int N, int GX> struct YLds {
    static constexpr int PAD = GX >= 16 ? 1 : 16 / GX;  // complex
    static constexpr int GSTR = P2<N>::LDS + PAD;
};

// Two N-point forward FFTs by one N/16-thread group, through ONE LDS buffer.  The exchanges alternate and are placed so
// that one transform is parked in LDS whenever the other is inside a butterfly (a butterfly needs 32 temporaries on top
// of its 32 data registers: with both transforms live the kernel would not fit 128 VGPRs).  In / out conventions per
// transform as fft_p2_group.
template <int N> __device__ __forceinline__ void fft_p2_pair(cf* a, cf* b, int u, cf* lds, const cf* __restrict__ tw, const cf* tw2) {
    typedef P2<N> G;
    const int k1 = u / G::R3, v = u % G::R3;
    const cf w1 = tw[u];  // W_N^u
    dft16(a);
    twiddle16(a, w1);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * G::S1 + u] = a[k];
    dft16(b);
    twiddle16(b, w1);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = lds[k1 * G::S1 + v + G::R3 * q];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * G::S1 + u] = b[k];
    dft16(a);
    if (G::R3 > 1) {
#pragma unroll
        for (int k = 1; k < 16; ++k) a[k] = cmul(a[k], tw2[k * G::R3 + v]);  // W_(N/16)^(v k)
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) b[q] = lds[k1 * G::S1 + v + G::R3 * q];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k1 * G::S2 + k * G::RP + v] = a[k];
    dft16(b);
    if (G::R3 > 1) {
#pragma unroll
        for (int k = 1; k < 16; ++k) b[k] = cmul(b[k], tw2[k * G::R3 + v]);
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < G::NB; ++bb) {
        const int pr = u + G::NT * bb;
        const cf* s = lds + (pr >> 4) * G::S2 + (pr & 15) * G::RP;
#pragma unroll
        for (int e = 0; e < G::R3; ++e) a[bb * G::R3 + e] = s[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k1 * G::S2 + k * G::RP + v] = b[k];
#pragma unroll
    for (int bb = 0; bb < G::NB; ++bb) dft_r<float, G::R3>(a + bb * G::R3);
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < G::NB; ++bb) {
        const int pr = u + G::NT * bb;
        const cf* s = lds + (pr >> 4) * G::S2 + (pr & 15) * G::RP;
#pragma unroll
        for (int e = 0; e < G::R3; ++e) b[bb * G::R3 + e] = s[e];
        dft_r<float, G::R3>(b + bb * G::R3);
    }
    __syncthreads();
}

// slot of frequency k held as element (bb, k3) by thread u after fft_p2_group / fft_p2_pair
template <int N> __device__ __forceinline__ int held_k(int u, int bb, int k3) {
    const int pr = u + P2<N>::NT * bb;
    return (pr >> 4) + 16 * (pr & 15) + 256 * k3;
}

// ------------------------------------------------------------------------------------------------
// pass 1: THR threads = GY groups (lane order (u, g), g fastest); group g packs the real columns x0 + 4g, +1 into transform
// A and +2, +3 into transform B (one float4 per row), splits the half spectra and stores them as 16-byte (column pair)
// pieces: 8 consecutive lanes fill one 128-byte line of W2.          detrend/window: xrft.py:425-433
// ------------------------------------------------------------------------------------------------
template <int NY>
__global__ void __launch_bounds__(YCols<NY>::THR, YCols<NY>::THR / 128) fasty_cols_kernel(FastY p) {
    typedef P2<NY> G;
    typedef YCols<NY> Y;
    constexpr int NT = G::NT, GY = Y::GY, THR = Y::THR, GSTR = YLds<NY, GY>::GSTR;
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x, g = tid % GY, u = tid / GY;
    cf* mine = lds + g * GSTR;
    cf* tw2 = lds + GY * GSTR;
    fill_tw2<NY>(tw2, p.tw_y, tid, THR);
    // unit = (slab, column block).  Blocks b, b+8, b+16, ... run on one XCD (round-robin dispatch): give each XCD a
    // contiguous range of column blocks, so that the workgroups sharing a 128-byte line of the input share an L2.
    const int nxb = p.nx / Y::CW;
    int slab, xb;
    if ((nxb & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nxb >> 3;
        slab = j / per;
        xb = xcd * per + j % per;
    } else {
        slab = blockIdx.x / nxb;
        xb = blockIdx.x % nxb;
    }
    const int x0 = xb * Y::CW + 4 * g;
    // uniform 64-bit base + one 32-bit per-lane byte offset (a slab is < 4 GB): scalar-base loads, no 64-bit address per row
    const char* __restrict__ src = reinterpret_cast<const char*>(p.in + (size_t)slab * NY * p.nx + (size_t)xb * Y::CW);
    const unsigned off0 = ((unsigned)u * (unsigned)p.nx + 4u * (unsigned)g) * 4u, rstep = (unsigned)NT * (unsigned)p.nx * 4u;
    F4 raw[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) raw[q] = *reinterpret_cast<const F4*>(src + (off0 + rstep * (unsigned)q));
    const F4 wx = *reinterpret_cast<const F4*>(p.win_x + x0);
    // ---- detrend, fused (cf. fastp2_rows_kernel with the axes swapped): every column's own least-squares line
    // m + s (i - ibar) is found here and subtracted in y-space; the plane of xrft/detrend.py:100-113 differs from it by a
    // noise-sized (offset, slope) pair per column, which pass 2 adds back in the spectral domain:
    // wx[x] * (alpha_x What0[ky] + gamma_x What1[ky]),  What0 = FFT(wy), What1 = FFT(wy (i - ibar)).  float64 sums.
    constexpr double IBAR = 0.5 * (NY - 1);
    float th[4] = {0.f, 0.f, 0.f, 0.f}, tl[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, sl[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.detrend && !(XRFT_YDBG & 16)) {
        struct alignas(16) D8 { double s0[4], s1[4]; };
        // a thread's 16 rows are summed in float32 (16 terms: ~1e-7 relative, random over the NT threads of a column), the
        // rest of the reduction runs in float64.  sum (i - ibar) d over i = u + NT q is (u - ibar) S0 + NT sum q d.
        // (float64 from the first term costs 64 conversions per thread and 128 live VGPRs when the scheduler hoists them.)
        float f0[4] = {0.f, 0.f, 0.f, 0.f}, f1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float fq = (float)q;
            f0[0] += raw[q].x; f1[0] = fmaf(fq, raw[q].x, f1[0]);
            f0[1] += raw[q].y; f1[1] = fmaf(fq, raw[q].y, f1[1]);
            f0[2] += raw[q].z; f1[2] = fmaf(fq, raw[q].z, f1[2]);
            f0[3] += raw[q].w; f1[3] = fmaf(fq, raw[q].w, f1[3]);
        }
        D8 t;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t.s0[c] = (double)f0[c];
            t.s1[c] = fma((double)u - IBAR, (double)f0[c], (double)NT * (double)f1[c]);
        }
        // three levels through LDS (the scratch aliases the still unused FFT buffers), every access unit-stride over the
        // lanes: (1) red[c][tid], c = 8 sums; (2) 8 J GY threads (c, jj, g) add NT / J entries each; (3) everyone adds the
        // J partial sums of its group
        constexpr int J = 8 * 4 * GY <= THR ? 4 : THR / (8 * GY), RS = THR + 8;  // the pad spreads the 8 rows over the banks
        double* red = reinterpret_cast<double*>(lds);
        double* mid = red + 8 * RS;
#pragma unroll
        for (int c = 0; c < 4; ++c) { red[c * RS + tid] = t.s0[c]; red[(4 + c) * RS + tid] = t.s1[c]; }
        __syncthreads();
        if (tid < 8 * J * GY) {
            const int c = tid / (J * GY), r = tid % (J * GY);
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < NT / J; ++k) acc += red[c * RS + k * (J * GY) + r];
            mid[tid] = acc;
        }
        __syncthreads();
        D8 tot;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tot.s0[c] = 0.0; tot.s1[c] = 0.0;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) { tot.s0[c] += mid[c * (J * GY) + jj * GY + g]; tot.s1[c] += mid[(4 + c) * (J * GY) + jj * GY + g]; }
        }
        constexpr double inv_n = 1.0 / NY, inv_sii = 12.0 / ((double)NY * ((double)NY * NY - 1.0));  // sum (i - ibar)^2 = n (n^2 - 1) / 12
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double md = tot.s0[c] * inv_n, sd = p.detrend == 2 ? tot.s1[c] * inv_sii : 0.0;
            if (u == 0) {
                double* cfp = p.colfit + ((size_t)slab * p.nx + x0 + c) * 2;
                cfp[0] = md; cfp[1] = sd;
            }
            // local trend (m - s ibar) + s i, subtracted in float32 with hi/lo splits whose hi parts lie on a coarse
            // power-of-two grid: x - th and the FMA with the exact product sh * i are error-free, the lo parts are
            // applied to the already noise-sized value (see fastp2_rows_kernel: 1e-6 of max instead of 6e-4)
            const float mf = (float)md, sf = (float)sd;  // the float32 values are what gets subtracted
            const double tt = (double)mf - (double)sf * IBAR;
            // grid 2^(e-20), 2^e <= |t| + |s| ny < 2^(e+1): adding and subtracting C = 1.5 * 2^(e+3) rounds to that grid
            const float mag = fabsf((float)tt) + fabsf(sf) * (float)NY;
            const float C = __uint_as_float((__float_as_uint(mag) & 0x7f800000u) + (3u << 23)) * 1.5f;
            const float thc = ((float)tt + C) - C, shc = (sf + C) - C;
            th[c] = thc; tl[c] = (float)(tt - (double)thc); sh[c] = shc; sl[c] = sf - shc;
        }
        __syncthreads();  // the reduction scratch aliases the FFT buffer written next
    }
    cf a[16], b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float fi = (float)(u + NT * q);
        const float wy = p.win_y[u + NT * q];
        const float v0 = fmaf(-sl[0], fi, fmaf(-sh[0], fi, raw[q].x - th[0]) - tl[0]);
        const float v1 = fmaf(-sl[1], fi, fmaf(-sh[1], fi, raw[q].y - th[1]) - tl[1]);
        const float v2 = fmaf(-sl[2], fi, fmaf(-sh[2], fi, raw[q].z - th[2]) - tl[2]);
        const float v3 = fmaf(-sl[3], fi, fmaf(-sh[3], fi, raw[q].w - th[3]) - tl[3]);
        a[q] = mk<float>(v0 * (wy * wx.x), v1 * (wy * wx.y));
        b[q] = mk<float>(v2 * (wy * wx.z), v3 * (wy * wx.w));
    }
    if (!(XRFT_YDBG & 32)) fft_p2_pair<NY>(a, b, u, mine, p.tw_y, tw2);
    // split the packed transforms: Ra[k] = (Z[k] + conj Z[N-k]) / 2, Rb[k] = (Z[k] - conj Z[N-k]) / (2i), k = u + NT q (q < 8)
    // and k = NY/2; (Ra, Rb) = two adjacent columns = one 16-byte store; lanes (u..u+RK-1, all g) complete a line
    cf* __restrict__ w2s = p.w2 + (size_t)slab * p.nrow_pad * p.nx;
#pragma unroll
    for (int set = 0; set < 2; ++set) {
        const cf* z = set == 0 ? a : b;
#pragma unroll
        for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
            for (int k3 = 0; k3 < G::R3; ++k3) mine[nat16(held_k<NY>(u, bb, k3))] = z[bb * G::R3 + k3];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int k = u + NT * q;
            if (q < 8 || u == 0) {
                const cf zk = mine[nat16(k & (NY - 1))];
                const cf zc = cconj(mine[nat16((NY - k) & (NY - 1))]);
                const cf ra = cscale(zk + zc, 0.5f), rb = cscale(mul_mi(zk - zc), 0.5f);
                const size_t off = ((((size_t)(k / Y::RK) * nxb + xb) * 2 + set) * Y::LBS) + (k % Y::RK) * (2 * GY) + 2 * g;
                F4 o; o.x = ra.re; o.y = ra.im; o.z = rb.re; o.w = rb.im;
                if (!(XRFT_YDBG & 64) || o.x == 1.2345f) *reinterpret_cast<F4*>(w2s + off) = o;
            }
        }
        if (set == 0) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: THR threads = GX groups (lane order (u, g), g fastest); a workgroup owns RPU = 2 GX consecutive rows ky0.. of
// W2 (group g: rows ky0 + g and ky0 + GX + g, in lockstep), adds the residual trend back, transforms along x, stages
// |F|^2 * scale in LDS and writes every valid row twice: as output row ky (rotated by the fftshift) and, reversed, as
// output row -ky (Hermitian mirror of the spectrum of a real field); ky = 0 and ny/2 are their own mirrors.
//   xrft.py:446-447 (fftshift), :740-748 (|F|^2 and the scalings, folded into `scale`), :895-906 (ISO: radial sums)
// ------------------------------------------------------------------------------------------------
template <int NX> struct YRows {
    static constexpr int THR = NX >= 2048 ? 512 : 256;
    static constexpr int NT = NX / 16;
    static constexpr int GX = THR / NT;  // 2, 4, 4, 8, 16
    static constexpr int RPU = 2 * GX;   // rows per workgroup
    static constexpr int RS = NX + NX / 16;  // floats per staged row (nat16 padding)
};

__device__ __forceinline__ size_t w2_offset(const FastY& p, int ky, int x) {
    const int nxb = p.nx >> p.l_cw;
    const size_t blk = ((size_t)(ky >> p.l_rk) * nxb + (x >> p.l_cw)) * 2 + ((x >> 1) & 1);
    return (blk << (p.l_rk + p.l_2gy)) + ((ky & ((1 << p.l_rk) - 1)) << p.l_2gy) + (((x & ((1 << p.l_cw) - 1)) >> 2) << 1) + (x & 1);
}

template <int NX, bool ISO>
__global__ void __launch_bounds__(YRows<NX>::THR, YRows<NX>::THR / 128) fasty_rows_kernel(FastY p) {
    typedef P2<NX> G;
    typedef YRows<NX> R;
    constexpr int NT = G::NT, GX = R::GX, THR = R::THR, RPU = R::RPU, GSTR = YLds<NX, GX>::GSTR;
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    float* stg = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, g = tid % GX, u = tid / GX;
    cf* mine = lds + g * GSTR;
    cf* tw2 = lds + GX * GSTR;
    double* hist = reinterpret_cast<double*>(tw2 + 16 * G::R3);
    fill_tw2<NX>(tw2, p.tw_x, tid, THR);
    if (ISO) for (int i = tid; i < p.nbins; i += THR) hist[i] = 0.0;  // ordered before the first add by the FFT's barriers
    const int upr = p.nrow_pad / RPU;  // units per slab
    const int slab = blockIdx.x / upr, unit = blockIdx.x % upr, ky0 = unit * RPU;
    const int nyh = p.ny >> 1;
    // rows beyond ny/2 (padding of the last unit) are computed on row ny/2's data and never stored or binned
    const int kyA = min(ky0 + g, nyh), kyB = min(ky0 + GX + g, nyh);
    const cf* __restrict__ w2s = p.w2 + (size_t)slab * p.nrow_pad * NX;
    cf a[16], b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int x = u + NT * q;
        a[q] = w2s[w2_offset(p, kyA, x)];
        b[q] = w2s[w2_offset(p, kyB, x)];
    }
    if (p.detrend && !(XRFT_YDBG & 1)) {  // add back wx[x] * (column fit - plane fit) in the spectral domain (see fasty_cols_kernel)
        const cf a0 = p.what0[kyA], a1 = p.what1[kyA], b0 = p.what0[kyB], b1 = p.what1[kyB];
        const float* __restrict__ cr = p.corr + ((size_t)slab * NX + u) * 2;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float al = (XRFT_YDBG & 2) ? a0.re : cr[2 * NT * q], ga = (XRFT_YDBG & 2) ? a0.im : cr[2 * NT * q + 1];
            a[q].re = fmaf(al, a0.re, fmaf(ga, a1.re, a[q].re));
            a[q].im = fmaf(al, a0.im, fmaf(ga, a1.im, a[q].im));
            b[q].re = fmaf(al, b0.re, fmaf(ga, b1.re, b[q].re));
            b[q].im = fmaf(al, b0.im, fmaf(ga, b1.im, b[q].im));
        }
    }
    if (!(XRFT_YDBG & 4)) fft_p2_pair<NX>(a, b, u, mine, p.tw_x, tw2);
    if (ISO) {  // value at (ky, kx) goes to its bin, and once more to the bin of (-ky, -kx)
        const unsigned* __restrict__ tc = p.tcodes + ((size_t)unit * 32) * THR + tid;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const cf v = e < 16 ? a[e] : b[e - 16];
            const unsigned code = tc[(size_t)e * THR];
            const float pw = (v.re * v.re + v.im * v.im) * p.scale;
            const unsigned cd = code & 0xffffu, cm = code >> 16;
            if (cd == cm) { if (cd) atomicAdd(&hist[cd - 1], 2.0 * (double)pw); }
            else {
                if (cd) atomicAdd(&hist[cd - 1], (double)pw);
                if (cm) atomicAdd(&hist[cm - 1], (double)pw);
            }
        }
    }
    if (p.out != nullptr) {
        // power, staged row-major [row][kx] in natural order with the conflict-free 17/16 padding
#pragma unroll
        for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
            for (int k3 = 0; k3 < G::R3; ++k3) {
                const int s = nat16(held_k<NX>(u, bb, k3));
                const cf va = a[bb * G::R3 + k3], vb = b[bb * G::R3 + k3];
                stg[g * R::RS + s] = (va.re * va.re + va.im * va.im) * p.scale;
                stg[(GX + g) * R::RS + s] = (vb.re * vb.re + vb.im * vb.im) * p.scale;
            }
        __syncthreads();
        // every staged row leaves twice: rotated (direct) and reversed + rotated (mirror); 16-byte stores, whole rows
        float* __restrict__ outs = p.out + (size_t)slab * p.ny * NX;
        const int mx = NX - 1, my = p.ny - 1, sx = p.shift_x;
        constexpr int CPR = NX / 4;  // float4 chunks per row
        for (int e = tid; e < RPU * 2 * CPR; e += THR) {
            const int chunk = e % CPR, rr = e / CPR, r = rr >> 1, mir = rr & 1;
            const int ky = ky0 + r;
            if (ky > nyh || (mir && (ky == 0 || ky == nyh))) continue;
            const float* row = stg + r * R::RS;
            const int c = 4 * chunk;
            F4 v;
            if (!mir) {
                const int kx = (c - sx) & mx;
                v.x = row[nat16(kx)]; v.y = row[nat16(kx + 1)]; v.z = row[nat16(kx + 2)]; v.w = row[nat16(kx + 3)];
            } else {  // output column c holds kx = (nx - (c - sx)) mod nx
                const int kx = (NX - c + sx) & mx;
                v.x = row[nat16(kx)]; v.y = row[nat16((kx - 1) & mx)]; v.z = row[nat16((kx - 2) & mx)]; v.w = row[nat16((kx - 3) & mx)];
            }
            const int orow = mir ? ((p.ny - ky) + p.shift_y) & my : (ky + p.shift_y) & my;
            if (!(XRFT_YDBG & 8) || v.x == 1.2345f) *reinterpret_cast<F4*>(outs + (size_t)orow * NX + c) = v;
        }
    }
    if (ISO) {
        __syncthreads();
        for (int i = tid; i < p.nbins; i += THR) {
            const double v = hist[i];
            if (v != 0.0) atomicAdd(&p.iso[(size_t)slab * p.nbins + i], v);
        }
    }
}

}  // namespace xrft

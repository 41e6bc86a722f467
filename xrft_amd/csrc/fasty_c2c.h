// fasty_c2c.h -- the two-pass "y first" pipeline of fasty.h for COMPLEX float32 slabs (both lengths a power of two, 256 .. 4096):
//     pass 1  fastyc_cols_kernel   FFT along y of the complex columns (input rotation / phase / window / conjugation fused on load)
//     pass 2  fastyc_rows_kernel   FFT along x of the rows, scale / conjugation / output phase / fftshift fused on store
// xrft.ifft over two axes (reference xrft/xrft.py:586-621: ifftshift, numpy.fft.ifftn, fftshift, the lag's phase) and xrft.fft of complex
// data (:439-447).  The inverse is conj(FFT(conj z)) / N: conjugate on load in pass 1, on store in pass 2.
//
// Round 5 ran these as two one-axis plans over the array where it lies (fastm_yonly + fastm_xonly: 32 B per point through memory at
// 2 TB/s, (16, 4096, 4096) 62 GFFT/s): a column workgroup there owns 4 complex columns and WRITES 32-byte row segments -- four
// workgroups fill one 128-byte line at four different times.  Here, as in fasty.h, pass 1 writes a tiled intermediate in whole lines
//     W2[slab][ky / RK][x / CW][ky % RK][CW]     CW = 2 GY complex columns of a pass-1 workgroup, RK = max(1, 16 / CW) rows per line
// and pass 2 owns complete rows of the result.  A thread runs two transforms through one LDS buffer (fft_p2_pair): two adjacent columns
// (one 16-byte load per row) in pass 1, two rows in pass 2.  32 B per point through memory either way; the passes run at the copy-like
// rate of fasty's.
#pragma once
#include "fasty.h"

namespace xrft {

struct FastYC {
    const cf* in;        // [slab][ny][nx] complex64
    cf* w2;              // the tiled intermediate, see above
    void* out;           // [slab][ny][nx] complex64, or float32 |F|^2 (power)
    const cf* tw_x;      // W_nx^k
    const cf* tw_y;      // W_ny^k
    const float* win_y;  // never null (ones when there is no window)
    const float* win_x;
    const cf* ph_y;      // phase tables: on the INPUT samples (ph_in: by source index) or on the output (ph_on: by unshifted frequency index)
    const cf* ph_x;
    int ph_in, ph_on;
    int inv;             // inverse transform: conjugate in, conjugate out
    int ishift_y, ishift_x;  // the input is rotated by n/2 on load (an fftshifted spectrum, xrft.py:612-617): 0 | 1
    int shift_y, shift_x;    // fftshift of the output: 0 | n/2
    int ny, nx, nslab;
    int l_cw, l_rk;      // log2 of CW, RK (layout of W2, fixed by ny)
    int power;           // |F|^2 * scale as float32 instead of the complex result
    int win_on;          // a window is set (the tables are read)
    long long nrows;     // fastyc_rows_kernel alone on ROW-MAJOR complex rows (one transform axis, the contiguous one; `w2` = the input, l_cw = log2 nx, l_rk = 0):
                         // the number of rows (0: pass 2 of the two-pass pipeline); the input-side options of pass 1 then apply to the rows here
    float scale;
};

// element offset of (ky, x) inside one slab of W2 (< 2^24 elements)
__device__ __forceinline__ unsigned w2c_offset(const FastYC& p, int ky, int x) {
    const unsigned nxb = (unsigned)p.nx >> p.l_cw;
    return ((((unsigned)ky >> p.l_rk) * nxb + ((unsigned)x >> p.l_cw)) << (p.l_rk + p.l_cw)) + (((unsigned)ky & ((1u << p.l_rk) - 1u)) << p.l_cw) + ((unsigned)x & ((1u << p.l_cw) - 1u));
}

// ------------------------------------------------------------------------------------------------
// pass 1: THR threads = GY groups (lane order (u, g), g fastest); group g owns the complex columns x0 + 2g (transform A) and x0 + 2g + 1
// (transform B): one 16-byte load per row.  The spectra leave in natural order as 16-byte (column pair) pieces: the lanes (u .. u + RK - 1,
// all g) complete a 128-byte line of W2.
// ------------------------------------------------------------------------------------------------
template <int NY>
__global__ void __launch_bounds__(YCols<NY>::THR, (YCols<NY>::THR >= 512 ? 4 : YCols<NY>::THR / 128)) fastyc_cols_kernel(FastYC p) {
    typedef P2<NY> G;
    typedef YCols<NY> Y;
    constexpr int NT = G::NT, GY = Y::GY, THR = Y::THR, GSTR = YLds<NY, GY>::GSTR, CW = 2 * GY;
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x, g = tid % GY, u = tid / GY;
    cf* mine = lds + g * GSTR;
    cf* tw2 = lds + GY * GSTR;
    fill_tw2<NY>(tw2, p.tw_y, tid, THR);
    // unit = (slab, column block); blocks b, b + 8, ... run on one XCD: each XCD gets a contiguous range of column blocks, so that the
    // workgroups sharing a 128-byte line of the input share an L2 (fasty_cols_kernel)
    const int nxb = p.nx / CW;
    int slab, xb;
    if ((nxb & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nxb >> 3;
        slab = j / per;
        xb = xcd * per + j % per;
    } else {
        slab = blockIdx.x / nxb;
        xb = blockIdx.x % nxb;
    }
    // the block of SOURCE columns (an fftshifted input: rotated by nx/2 = nxb/2 blocks) and the source rows u + NT q (+ ny/2: q + 8)
    const int xbs = p.ishift_x ? (xb + (nxb >> 1)) % nxb : xb;
    const int qrot = p.ishift_y ? 8 : 0;
    const char* __restrict__ src = reinterpret_cast<const char*>(p.in + (size_t)slab * NY * p.nx + (size_t)xbs * CW);
    const unsigned off0 = ((unsigned)u * (unsigned)p.nx + 2u * (unsigned)g) * 8u, rstep = (unsigned)NT * (unsigned)p.nx * 8u;
    cf a[16], b[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const F4 v = *reinterpret_cast<const F4*>(src + (off0 + rstep * (unsigned)((q + qrot) & 15)));
        a[q] = mk<float>(v.x, v.y);
        b[q] = mk<float>(v.z, v.w);
    }
    const int xs0 = xbs * CW + 2 * g;  // source column of transform A
    if (p.ph_in) {  // the lag's phase factor on the source samples: ph_y[row] ph_x[column]
        const cf pxa = p.ph_x[xs0], pxb = p.ph_x[xs0 + 1];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const cf py = p.ph_y[u + NT * ((q + qrot) & 15)];
            a[q] = cmul(a[q], cmul(py, pxa));
            b[q] = cmul(b[q], cmul(py, pxb));
        }
    }
    if (p.win_on) {  // (indexed by the transform's own sample index, as the one-axis kernels do)
        const int x0 = xb * CW + 2 * g;
        const float wxa = p.win_x[x0], wxb = p.win_x[x0 + 1];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float wy = p.win_y[u + NT * q];
            a[q] = cscale(a[q], wy * wxa);
            b[q] = cscale(b[q], wy * wxb);
        }
    }
    if (p.inv) {
#pragma unroll
        for (int q = 0; q < 16; ++q) { a[q].im = -a[q].im; b[q].im = -b[q].im; }
    }
    fft_p2_pair<NY>(a, b, u, mine, p.tw_y, tw2);
    // natural order through the group's LDS buffer, one transform at a time; then (A[k], B[k]) = one 16-byte piece, k = u + NT q
    int tid2 = threadIdx.x;
    XRFT_OPAQUE(tid2);
    const int g2 = tid2 % GY, u2 = tid2 / GY;
    cf* mine2 = lds + g2 * GSTR;
#pragma unroll
    for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
        for (int k3 = 0; k3 < G::R3; ++k3) mine2[nat16(held_k<NY>(u2, bb, k3))] = a[bb * G::R3 + k3];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = mine2[nat16(u2 + NT * q)];
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
        for (int k3 = 0; k3 < G::R3; ++k3) mine2[nat16(held_k<NY>(u2, bb, k3))] = b[bb * G::R3 + k3];
    __syncthreads();
    char* __restrict__ w2s = reinterpret_cast<char*>(p.w2 + (size_t)slab * NY * p.nx);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = u2 + NT * q;
        const cf zb = mine2[nat16(k)];
        const unsigned nxbu = (unsigned)nxb;
        const unsigned off = ((((unsigned)k >> p.l_rk) * nxbu + (unsigned)xb) << (p.l_rk + p.l_cw)) + (((unsigned)k & ((1u << p.l_rk) - 1u)) << p.l_cw) + 2u * (unsigned)g2;
        F4 o; o.x = a[q].re; o.y = a[q].im; o.z = zb.re; o.w = zb.im;
        xrft_store_nt(reinterpret_cast<float*>(w2s + off * 8u), o);  // (the next reader is another kernel, a whole group of slabs later)
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: THR threads = GX groups; a workgroup owns 2 GX consecutive rows ky0 .. of W2 (group g: rows ky0 + g and ky0 + GX + g), transforms
// them along x, and stores them as whole rows of the result: conjugated (inverse), scaled, times the output phase, rotated by the fftshift.
// ------------------------------------------------------------------------------------------------
template <int NX>
__global__ void __launch_bounds__((YRows<NX>::THR), (YRows<NX>::THR / 128 < 1 ? 1 : YRows<NX>::THR / 128)) fastyc_rows_kernel(FastYC p) {
    typedef P2<NX> G;
    typedef YRows<NX> R;
    constexpr int NT = G::NT, GX = R::GX, THR = R::THR, RPU = 2 * GX, GSTR = YLds<NX, GX>::GSTR;
    constexpr int RSC = NX + NX / 16;
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x, g = tid % GX, u = tid / GX;
    cf* mine = lds + g * GSTR;
    cf* tw2 = lds + GX * GSTR;
    fill_tw2<NX>(tw2, p.tw_x, tid, THR);
    const bool alone = p.nrows > 0;  // one transform axis: the rows of the input itself
    const int upr = p.ny / RPU;
    const int slab = alone ? 0 : (int)blockIdx.x / upr;
    const long long ky0 = alone ? (long long)blockIdx.x * RPU : (long long)(((int)blockIdx.x % upr) * RPU);
    const long long kyA = alone ? min(ky0 + g, p.nrows - 1) : ky0 + g, kyB = alone ? min(ky0 + GX + g, p.nrows - 1) : ky0 + GX + g;
    const char* __restrict__ w2s = reinterpret_cast<const char*>(p.w2 + (size_t)slab * p.ny * NX);
    cf a[16], b[16];
    if (alone) {  // row-major rows: 8 bytes per lane, 64 consecutive lanes = 512 contiguous bytes; the input-side options of pass 1
        const int qrot = p.ishift_x ? 8 : 0;  // an fftshifted input: x + nx/2 = u + NT (q + 8)
        const cf* __restrict__ ra = reinterpret_cast<const cf*>(p.w2) + (size_t)kyA * NX + u;
        const cf* __restrict__ rb = reinterpret_cast<const cf*>(p.w2) + (size_t)kyB * NX + u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = ra[NT * ((q + qrot) & 15)];
            b[q] = rb[NT * ((q + qrot) & 15)];
        }
        if (p.ph_in) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf f = p.ph_x[u + NT * ((q + qrot) & 15)];
                a[q] = cmul(a[q], f);
                b[q] = cmul(b[q], f);
            }
        }
        if (p.win_on) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float w = p.win_x[u + NT * q];
                a[q] = cscale(a[q], w);
                b[q] = cscale(b[q], w);
            }
        }
        if (p.inv) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { a[q].im = -a[q].im; b[q].im = -b[q].im; }
        }
    } else if (NT >= (1 << p.l_cw)) {  // x = u + NT q advances by whole column blocks: constant stride
        const unsigned offA = w2c_offset(p, (int)kyA, u) * 8u, offB = w2c_offset(p, (int)kyB, u) * 8u;
        const unsigned qstr = (unsigned)((NT >> p.l_cw) << (p.l_rk + p.l_cw)) * 8u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = *reinterpret_cast<const cf*>(w2s + (offA + qstr * (unsigned)q));
            b[q] = *reinterpret_cast<const cf*>(w2s + (offB + qstr * (unsigned)q));
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = *reinterpret_cast<const cf*>(w2s + w2c_offset(p, (int)kyA, u + NT * q) * 8u);
            b[q] = *reinterpret_cast<const cf*>(w2s + w2c_offset(p, (int)kyB, u + NT * q) * 8u);
        }
    }
    fft_p2_pair<NX>(a, b, u, mine, p.tw_x, tw2);
    // GX rows at a time staged in natural order (a round fills the transforms' LDS exactly): transform A's rows, then B's
    const int mx = NX - 1, my = p.ny - 1, sx = p.shift_x;
    cf* cstg = lds;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (round) __syncthreads();
#pragma unroll
        for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
            for (int k3 = 0; k3 < G::R3; ++k3) cstg[g * RSC + nat16(held_k<NX>(u, bb, k3))] = cscale(round ? b[bb * G::R3 + k3] : a[bb * G::R3 + k3], p.power ? 1.0f : p.scale);
        __syncthreads();
        if (p.power) {  // |F|^2 * scale: four samples per 16-byte store
            float* __restrict__ outs = reinterpret_cast<float*>(p.out) + (size_t)slab * p.ny * NX;
            constexpr int CPR = NX / 4;
            for (int e = tid; e < GX * CPR; e += THR) {
                const int chunk = e % CPR, rl = e / CPR, c = 4 * chunk, kx = (c - sx) & mx;
                const long long ky = ky0 + round * GX + rl;
                if (alone && ky >= p.nrows) break;  // (rl grows with e)
                const cf* row = cstg + rl * RSC;
                const cf v0 = row[nat16(kx)], v1 = row[nat16(kx + 1)], v2 = row[nat16(kx + 2)], v3 = row[nat16(kx + 3)];
                F4 o; o.x = (v0.re * v0.re + v0.im * v0.im) * p.scale; o.y = (v1.re * v1.re + v1.im * v1.im) * p.scale; o.z = (v2.re * v2.re + v2.im * v2.im) * p.scale; o.w = (v3.re * v3.re + v3.im * v3.im) * p.scale;
                xrft_store_nt(outs + ((size_t)(alone ? ky : ((ky + p.shift_y) & my)) * NX + c), o);
            }
            continue;
        }
        cf* __restrict__ outs = reinterpret_cast<cf*>(p.out) + (size_t)slab * p.ny * NX;
        constexpr int CPR = NX / 2;  // pairs of samples per row
        for (int e = tid; e < GX * CPR; e += THR) {
            const int chunk = e % CPR, rl = e / CPR, c = 2 * chunk;
            const long long ky = ky0 + round * GX + rl;
            if (alone && ky >= p.nrows) break;
            const int fx0 = (c - sx) & mx, fx1 = (c + 1 - sx) & mx;  // unshifted frequency indices of the two output columns
            const cf* row = cstg + rl * RSC;
            cf v0 = row[nat16(fx0)], v1 = row[nat16(fx1)];
            if (p.inv) { v0.im = -v0.im; v1.im = -v1.im; }
            if (p.ph_on) {
                const cf py = alone ? mk<float>(1.f, 0.f) : p.ph_y[ky];
                v0 = cmul(v0, cmul(py, p.ph_x[fx0]));
                v1 = cmul(v1, cmul(py, p.ph_x[fx1]));
            }
            xrft_store_nt2(outs + ((size_t)(alone ? ky : ((ky + p.shift_y) & my)) * NX + c), v0, v1);
        }
    }
}

}  // namespace xrft

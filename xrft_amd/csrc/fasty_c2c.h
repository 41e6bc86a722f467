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
    int c2r;             // the half spectrum of a real field back to real samples (irfftn, xrft.py:612-616): the input holds nx/2 + 1 complex columns
                         // (rows of in_pitch elements), pass 1 transforms them along y -- the nx/2 regular column blocks and ONE extra block for the Nyquist
                         // column -- and fastyc_rows_c2r_kernel<nx/2> turns every row into nx real samples
    int in_pitch;        // complex elements per input row (nx; c2r: nx/2 + 1)
    int w2_nxb;          // column blocks per row block of W2 (nx / CW; c2r: nx/2 / CW + 1)
    const cf* tw_big;    // c2r: W_nx^k, k < nx/2 / 16 (the split's twiddle of lane u; times W_32^q in registers)
    int fs;              // "four-step": a slab is ONE complex sequence of N = ny * nx points (xrft.ifft / fft of complex data along one long axis), n = nx i1 + i2:
                         // pass 1 as it is (the sums over i1), pass 2 multiplies row k1 by W_N^(i2 k1) (tw_big: W_N^j, j < N / 16) before its transform and stores
                         // TRANSPOSED, X[k1 + ny k2]: the rows of a unit are consecutive samples for a given k2; ph_x of an output phase is indexed by the sample k
    long long nrows;     // fastyc_rows_kernel alone on ROW-MAJOR complex rows (one transform axis, the contiguous one; `w2` = the input, l_cw = log2 nx, l_rk = 0):
                         // the number of rows (0: pass 2 of the two-pass pipeline); the input-side options of pass 1 then apply to the rows here
    float scale;
};

// element offset of (ky, x) inside one slab of W2 (< 2^24 elements)
__device__ __forceinline__ unsigned w2c_offset(const FastYC& p, int ky, int x) {
    const unsigned nxb = (unsigned)p.w2_nxb;
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
    const int nxb_full = p.c2r ? (p.nx / 2) / CW : p.nx / CW;  // whole column blocks; c2r: + ONE block whose first column is the Nyquist column
    const int nxb = p.w2_nxb;
    const bool tailb = p.c2r && (int)blockIdx.x >= p.nslab * nxb_full;
    int slab, xb;
    if (tailb) {
        slab = (int)blockIdx.x - p.nslab * nxb_full;
        xb = nxb_full;
    } else if ((nxb_full & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nxb_full >> 3;
        slab = j / per;
        xb = xcd * per + j % per;
    } else {
        slab = blockIdx.x / nxb_full;
        xb = blockIdx.x % nxb_full;
    }
    // the block of SOURCE columns (an fftshifted input: rotated by nx/2 = nxb/2 blocks) and the source rows u + NT q (+ ny/2: q + 8)
    const int xbs = p.ishift_x ? (xb + (nxb_full >> 1)) % nxb_full : xb;
    const int qrot = p.ishift_y ? 8 : 0;
    const char* __restrict__ src = reinterpret_cast<const char*>(p.in + (size_t)slab * NY * p.in_pitch + (size_t)xbs * CW);
    const unsigned off0 = ((unsigned)u * (unsigned)p.in_pitch + 2u * (unsigned)g) * 8u, rstep = (unsigned)NT * (unsigned)p.in_pitch * 8u;
    cf a[16], b[16];
    if (!tailb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#ifdef XRFT_EMULATE
            const float* v = reinterpret_cast<const float*>(src + (off0 + rstep * (unsigned)((q + qrot) & 15)));
            a[q] = mk<float>(v[0], v[1]);
            b[q] = mk<float>(v[2], v[3]);
#else
            typedef float v4f_a8 __attribute__((ext_vector_type(4), aligned(8)));  // (rows of an odd number of complex values: 8-byte aligned, still one 16-byte load)
            const v4f_a8 v = *reinterpret_cast<const v4f_a8*>(src + (off0 + rstep * (unsigned)((q + qrot) & 15)));
            a[q] = mk<float>(v.x, v.y);
            b[q] = mk<float>(v.z, v.w);
#endif
        }
    } else {  // the Nyquist column alone: transform A of group 0; everything else of the block is zero
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = g == 0 ? *reinterpret_cast<const cf*>(src + (off0 + rstep * (unsigned)((q + qrot) & 15))) : mk<float>(0.f, 0.f);
            b[q] = mk<float>(0.f, 0.f);
        }
    }
    const int xs0 = xbs * CW + 2 * g;  // source column of transform A
    if (p.ph_in) {  // the lag's phase factor on the source samples: ph_y[row] ph_x[column]
        const cf pxa = p.ph_x[tailb ? p.nx / 2 : xs0], pxb = tailb ? mk<float>(1.f, 0.f) : p.ph_x[xs0 + 1];
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
    char* __restrict__ w2s = reinterpret_cast<char*>(p.w2 + (size_t)slab * NY * ((size_t)nxb * CW));
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
    const char* __restrict__ w2s = reinterpret_cast<const char*>(p.w2 + (size_t)slab * p.ny * ((size_t)p.w2_nxb << p.l_cw));
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
    if (p.fs) {  // x W_N^(i2 k1), i2 = u + NT q: W^(k1 u) (W^(k1 NT))^q -- two table loads and a product tree per row (k1 u, k1 NT < N / 16)
        const cf wa = p.tw_big[(int)kyA * u], wb = p.tw_big[(int)kyB * u];
        twiddle16(a, p.tw_big[(int)kyA * NT]);
        twiddle16(b, p.tw_big[(int)kyB * NT]);
#pragma unroll
        for (int q = 0; q < 16; ++q) { a[q] = cmul(a[q], wa); b[q] = cmul(b[q], wb); }
    }
    fft_p2_pair<NX>(a, b, u, mine, p.tw_x, tw2);
    // GX rows at a time staged in natural order (a round fills the transforms' LDS exactly): transform A's rows, then B's
    const int mx = NX - 1, my = p.ny - 1, sx = p.shift_x;
    cf* cstg = lds;
    if (p.fs) {
        // transposed: sample k = k1 + ny k2 of the sequence (fftshift of the sequence by N/2: k2 + nx/2); the GX rows of a round are GX consecutive samples.
        // Rows staged one element apart from the natural pitch (RSC + 1: the lanes of a run read GX different banks)
        constexpr int RSF = RSC + 1;
        const size_t nseq = (size_t)p.ny * NX;
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if (round) __syncthreads();
#pragma unroll
            for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
                for (int k3 = 0; k3 < G::R3; ++k3) cstg[g * RSF + nat16(held_k<NX>(u, bb, k3))] = round ? b[bb * G::R3 + k3] : a[bb * G::R3 + k3];
            __syncthreads();
            const int k1b = (int)ky0 + round * GX;  // first row of the round
            for (int e = tid; e < GX * NX; e += THR) {
                const int rl = e % GX, oc = e / GX, k2 = (oc - sx) & mx;  // oc: the position along k2 in the (shifted) result
                cf v = cstg[rl * RSF + nat16(k2)];
                const size_t pos = (size_t)slab * nseq + (size_t)oc * p.ny + (size_t)(k1b + rl);
                if (p.power) {
                    reinterpret_cast<float*>(p.out)[pos] = (v.re * v.re + v.im * v.im) * p.scale;
                } else {
                    v = cscale(v, p.scale);
                    if (p.inv) v.im = -v.im;
                    if (p.ph_on) v = cmul(v, p.ph_x[(k1b + rl) + p.ny * k2]);  // (indexed by the unshifted sample index)
                    reinterpret_cast<cf*>(p.out)[pos] = v;
                }
            }
        }
        return;
    }
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


// ------------------------------------------------------------------------------------------------
// pass 2 of a c2r plan: rows ky of W2 hold X[ky, k], k = 0 .. M (M = nx/2; the Nyquist sample in the extra column block); the row's nx real
// samples are the packed inverse transform  z[n] = x[2n] + i x[2n+1] = IFFT_M(Z),  Z[k] = E[k] + i O[k],
//     E[k] = (X[k] + conj X[M-k]) / 2,   O[k] = (X[k] - conj X[M-k]) conj(W_nx^k) / 2,
// computed as conj(FFT_M(conj Z)): a thread holds k = u + NT q of two rows, fetches the partners X[M - k] itself (they are another lane's
// samples: a second read of the row, from L2), W_nx^k = W_nx^u W_32^q (nx = 32 NT).  Output: whole rows of nx float32 samples, 16-byte stores.
// `alone` (nrows > 0): the rows of a row-major half spectrum [rows][M + 1] themselves -- xrft.ifft with real_dim along ONE axis, the contiguous one.
// ------------------------------------------------------------------------------------------------
template <int M, bool ALONE>
__global__ void __launch_bounds__((YRows<M>::THR), (YRows<M>::THR >= 256 ? 2 : 1)) fastyc_rows_c2r_kernel(FastYC p) {  // (about 150 VGPRs: three waves per SIMD; held to 128 the M = 2048 form spilled 46-112)
    typedef P2<M> G;
    typedef YRows<M> R;
    constexpr int NT = G::NT, GX = R::GX, THR = R::THR, RPU = 2 * GX, GSTR = YLds<M, GX>::GSTR;
    constexpr int RSC = M + M / 16;
    // W_32^q = cos - i sin of 2 pi q / 32
    constexpr float C32[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                               0.38268343236508977173f, 0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
                               -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float S32[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f, 0.83146961230254523708f,
                               0.92387953251128675613f, 0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                               0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x, g = tid % GX, u = tid / GX;
    cf* mine = lds + g * GSTR;
    cf* tw2 = lds + GX * GSTR;
    fill_tw2<M>(tw2, p.tw_x, tid, THR);
    constexpr bool alone = ALONE;  // (nrows > 0: the rows of the row-major input itself)
    const int upr = p.ny / RPU;
    const int slab = alone ? 0 : (int)blockIdx.x / upr;
    const long long ky0 = alone ? (long long)blockIdx.x * RPU : (long long)(((int)blockIdx.x % upr) * RPU);
    const long long kyr[2] = {alone ? min(ky0 + g, p.nrows - 1) : ky0 + g, alone ? min(ky0 + GX + g, p.nrows - 1) : ky0 + GX + g};
    const cf wu = p.tw_big[u];  // W_nx^u
    const cf* __restrict__ w2s = p.w2 + (size_t)slab * p.ny * ((size_t)p.w2_nxb << p.l_cw);
    // the rows' samples k = u + NT q once from memory (both rows in flight together; a uniform 64-bit base + 32-bit per-lane byte offsets: scalar-base loads); the
    // partners X[M - k] are other lanes' samples: they come through the group's LDS buffer (a second read of the row from memory took the pass from 25 to 59 us per
    // 4096-row slab); the Nyquist sample M rides with lane u = 0
    cf a[16], b[16];
    cf nyqa = mk<float>(0.f, 0.f), nyqb = nyqa;  // (read by lane u = 0 only; they reach their row through the LDS slot of sample M)
    if (alone) {  // the row-major input itself: rows of M + 1 complex values
        const char* __restrict__ base = reinterpret_cast<const char*>(p.w2 + (size_t)ky0 * (M + 1));
        const unsigned oa = (unsigned)((kyr[0] - ky0) * (M + 1) + u) * 8u, ob = (unsigned)((kyr[1] - ky0) * (M + 1) + u) * 8u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = *reinterpret_cast<const cf*>(base + (oa + (unsigned)(NT * q) * 8u));
            b[q] = *reinterpret_cast<const cf*>(base + (ob + (unsigned)(NT * q) * 8u));
        }
        if (u == 0) { nyqa = *reinterpret_cast<const cf*>(base + (oa + (unsigned)M * 8u)); nyqb = *reinterpret_cast<const cf*>(base + (ob + (unsigned)M * 8u)); }
        if (p.ph_in) {  // the lag's phase on the source samples, four factors at a time
#pragma unroll
            for (int q4 = 0; q4 < 16; q4 += 4) {
                cf f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = p.ph_x[u + NT * (q4 + j)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[q4 + j] = cmul(a[q4 + j], f[j]);
                    b[q4 + j] = cmul(b[q4 + j], f[j]);
                }
#ifndef XRFT_EMULATE
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            if (u == 0) { const cf f = p.ph_x[M]; nyqa = cmul(nyqa, f); nyqb = cmul(nyqb, f); }
        }
    } else {  // the tiled intermediate: pass 1 of an inverse plan left conj(IFFT_y X) (FFT_y of the conjugated input)
        const char* __restrict__ w2c = reinterpret_cast<const char*>(w2s);
        const unsigned oa = w2c_offset(p, (int)kyr[0], u) * 8u, ob = w2c_offset(p, (int)kyr[1], u) * 8u;
        if (NT >= (1 << p.l_cw)) {
            const unsigned qstr = (unsigned)((NT >> p.l_cw) << (p.l_rk + p.l_cw)) * 8u;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                a[q] = cconj(*reinterpret_cast<const cf*>(w2c + (oa + qstr * (unsigned)q)));
                b[q] = cconj(*reinterpret_cast<const cf*>(w2c + (ob + qstr * (unsigned)q)));
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                a[q] = cconj(*reinterpret_cast<const cf*>(w2c + w2c_offset(p, (int)kyr[0], u + NT * q) * 8u));
                b[q] = cconj(*reinterpret_cast<const cf*>(w2c + w2c_offset(p, (int)kyr[1], u + NT * q) * 8u));
            }
        }
        if (u == 0) { nyqa = cconj(w2s[w2c_offset(p, (int)kyr[0], M)]); nyqb = cconj(w2s[w2c_offset(p, (int)kyr[1], M)]); }
    }
    __syncthreads();  // (the twiddle table's writers and, in a later use, the buffer's previous readers)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        cf* z = t ? b : a;
        cf wut = wu;  // (opaque per row: shared between the two rows, the sixteen twiddles W_nx^k stayed live beside both rows' 64 registers -- 112 spilled at M = 2048)
        XRFT_OPAQUE(wut.re);
        XRFT_OPAQUE(wut.im);
#pragma unroll
        for (int q = 0; q < 16; ++q) mine[nat16(u + NT * q)] = z[q];
        if (u == 0) mine[nat16(M)] = t ? nyqb : nyqa;  // (slot M + M/16: inside the group's buffer of M + 256 elements)
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // (eight partners at a time: sixteen beside the two rows' 64 registers spilled 38 at M = 2048)
            cf pr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = u + NT * (8 * h + j);
                pr[j] = mine[nat16(M - k)];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = 8 * h + j;
                const cf x = z[q], pc = cconj(pr[j]);
                const cf e = mk<float>(x.re + pc.re, x.im + pc.im), d = mk<float>(x.re - pc.re, x.im - pc.im);  // 2E, X - conj P
                // conj(W_nx^k) = conj(W_nx^u W_32^q)
                const cf w = cconj(cmul(wut, mk<float>(C32[q], -S32[q])));
                const cf o = cmul(d, w);                                  // 2 O
                const cf zz = mk<float>(e.re - o.im, e.im + o.re);        // 2 (E + i O)
                z[q] = cconj(zz);                                         // the inverse transform as conj(FFT(conj Z))
            }
#ifndef XRFT_EMULATE
            __builtin_amdgcn_sched_barrier(0);  // (nothing is scheduled across: the batches' partners and twiddles do not pile up beside the rows' 64 registers)
#endif
        }
        __syncthreads();
    }
    fft_p2_pair<M>(a, b, u, mine, p.tw_x, tw2);
    // GX rows at a time staged in natural order; z[n] = conj(result[n]): x[2n] = re, x[2n + 1] = -im
    const int mm = M - 1, my = p.ny - 1, sxh = p.shift_x >> 1;  // (the output fftshift by nx/2 samples = M/2 packed values)
    cf* cstg = lds;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (round) __syncthreads();
#pragma unroll
        for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
            for (int k3 = 0; k3 < G::R3; ++k3) cstg[g * RSC + nat16(held_k<M>(u, bb, k3))] = round ? b[bb * G::R3 + k3] : a[bb * G::R3 + k3];
        __syncthreads();
        float* __restrict__ outs = reinterpret_cast<float*>(p.out) + (size_t)slab * p.ny * (2 * M);
        constexpr int CPR = M / 2;  // 16-byte chunks (two packed values = four samples) per row
        for (int e = tid; e < GX * CPR; e += THR) {
            const int chunk = e % CPR, rl = e / CPR, n0 = (2 * chunk - sxh) & mm;
            const long long ky = ky0 + round * GX + rl;
            if (alone && ky >= p.nrows) break;
            const cf* row = cstg + rl * RSC;
            const cf v0 = row[nat16(n0)], v1 = row[nat16((n0 + 1) & mm)];
            F4 o; o.x = v0.re * p.scale; o.y = -v0.im * p.scale; o.z = v1.re * p.scale; o.w = -v1.im * p.scale;
            xrft_store_nt(outs + ((size_t)(alone ? ky : ((ky + p.shift_y) & my)) * (2 * M) + 4 * chunk), o);
        }
    }
}

}  // namespace xrft

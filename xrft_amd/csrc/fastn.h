// fastn.h -- the two-pass "y first" pipeline of fastm.h with the LENGTHS AS DATA: one column kernel and one row kernel per precision serve every
// real slab whose two lengths are products of the butterflies {2 ... 16, 18, 20} (7, 11, 13 included: numpy's pocketfft hard-codes 7 and 11) in up to
// six passes -- and, for the columns, any other length through a chirp (Bluestein) convolution inside the tile (721 = 7 x 103 latitudes of the ERA5 grid).
// Before it a large slab off fastm.h's table of ~50 lengths took the generic tile passes (tile_fft.h): a moments pass, x rows, y in one or two steps --
// four trips through memory at 1.9 TB/s each.           (xrft.power_spectrum / fft / cross_spectrum / isotropic_*: xrft/xrft.py:307-476, 685-1187;
// any length: numpy.fft.fftn / rfftn behind xrft.py:439-444; detrend: xrft/detrend.py:100-113)
//
// Same plan as fastm.h (read its header first):
//   pass 1  fastn_cols_kernel  FFT along y of packed column pairs, window fused, half spectra ky = 0 .. ny/2 into the line-blocked intermediate W2,
//                              exact column sums on the side (float32: an in-pass estimate of the trend subtracted first, fasty.h / fastm.h)
//   [fit]   fastm_fit_kernel   the plane from the column sums
//   pass 2  fastn_rows_kernel  plane added back in the spectral domain, FFT along x of rows ky, every row stored as ky and (reversed) -ky;
//                              power / complex / cross / cross phase, full or half rows, radial sums gathered per bin
// and the same LDS scheme: the first pass of a transform runs on operands loaded straight from global memory, the others in place in LDS
// (decimation in frequency), the last one leaves NATURAL order -- but the radices, the strides, the paddings that keep the LDS accesses
// conflict-free, the sequences per workgroup and the thread count come from a parameter block (NGeo), and a thread takes as many butterflies per
// pass as the geometry needs (only the last pass, which permutes the whole sequence, is one butterfly per thread: the host picks the largest radix for it).
// Either pass of a plan may instead be the table kernel of fastm.h when its length is in the table: the intermediate's layout (FastM::l_cw, l_rk)
// is the contract between them.
#pragma once
#include "fastm.h"
#include "fastg.h"  // (the column passes of the Rader form)

namespace xrft {

constexpr int kNMaxPass = 6;
template <typename T> constexpr int fastn_max_threads() { return sizeof(T) == 4 ? 1024 : 512; }

struct NGeo {
    int n;                    // points of the transform (pass 1 of a Bluestein plan: the convolution length m)
    int np;                   // radix passes, 2 .. kNMaxPass
    int r[kNMaxPass];         // radices, pass 0 first
    float inv_r[kNMaxPass];
    int m[kNMaxPass];         // m[p] = L_p / r[p], L_p = n / (r[0] .. r[p-1]): the operands of a butterfly of pass p are m[p] points apart
    int step[kNMaxPass];      // ... which in the intermediate LDS layout is m[p] + m[p] / pdq positions
    int two[kNMaxPass];       // middle passes (1 <= p <= np - 2): offset of W_{L_p}^(j k) at [j r[p] + k] in the staged twiddle table
    int twn;                  // entries of that table
    int g, lg;                // sequences per workgroup (a power of two) and its log2
    int thr;                  // threads per workgroup
    int str;                  // LDS elements between sequences
    float inv_pdq;            // intermediate layout: pd(i) = i + i / pdq, pdq = r[np-1] when that is even (0: no padding) -- the last pass reads runs of r[np-1]
    float inv_pnq;            // natural layout:      pn(k) = k + k / pnq, pnq = r[0] when that is even -- the last pass writes with a lane stride of r[0] ...
    int pn_r0;                // ... and then pn(k0 + r0 rest) = k0 + r0 rest + rest needs no division
    int wlast;                // r[1] .. r[np-2]: the weight of the last pass's output index in `rest`
};

// The geometry is read through the CONSTANT address space: uniform, read-only for the kernel's lifetime, so every field is a scalar load (s_load, scalar
// cache) -- through a plain global pointer the compiler cannot rule out aliasing with the kernel's own stores and fetches each field with a vector load, an
// L2 round trip on the critical path between two barriers (measured: 2-3 x the table kernels' VMEM read instructions, SQ busy cycles 2 x).
#ifdef XRFT_EMULATE
typedef const NGeo* NGeoPtr;
typedef const NGeo& NGeoRef;
#else
typedef const NGeo __attribute__((address_space(4)))* NGeoPtr;
typedef const NGeo __attribute__((address_space(4)))& NGeoRef;
#endif

// the prime-factor / Rader form of pass 1 (fastg.h, FastGY::rad_p: ny = q p, ONE prime p with a smooth p - 1): its radices, in device memory like NGeo
struct RGeo {
    int p, q, nrq, nrp;
    int rq[kNMaxPass + 2], rp[kNMaxPass + 2];
};
#ifdef XRFT_EMULATE
typedef const RGeo* RGeoPtr;
typedef const RGeo& RGeoRef;
#else
typedef const RGeo __attribute__((address_space(4)))* RGeoPtr;
typedef const RGeo __attribute__((address_space(4)))& RGeoRef;
#endif

struct FastN {
    FastM f;              // the pipeline's parameter block exactly as the table kernels take it (incl. the intermediate's layout l_cw, l_rk)
    NGeoPtr g;            // the transform of THIS pass (device memory, uploaded once per plan: a by-value copy in the kernel arguments is dynamically
                          // indexed by the pass number, which made the compiler park the whole argument block in scratch memory)
    const void* twm;      // twiddles of the middle passes, [g.twn] complex T (staged in LDS)
    int pitch;            // complex elements per row of the intermediate: nxb column blocks of CW columns (a ragged last block is padded)
    int nxb;
    int pair_ok;          // pass 1: a column pair is one aligned 2 T-wide load (nx even)
    const void* blue_c;   // pass 1, Bluestein (f.ny points as a circular convolution of g.n): c[k] = exp(i pi k^2 / ny), k < ny
    const void* blue_b;   // FFT_m(chirp) / m in natural order
    int vec_ok;           // pass 2: the rows leave 16 bytes per lane (the row length divides)
    int rpu;              // pass 2: rows ky per workgroup (two fields: g.g = 2 rpu sequences)
    int dbg;              // ablation switches of the measuring scripts (XRFTHIP_FASTN_DBG; 0 in production): 1 no LDS passes, 2 no stores, 4 no first pass
    // pass 1, the Rader form (FORM 2): g.n = f.ny, g.str = f.ny (the tile is [ny][G], lanes along the sequences), twm = W_q then W_(p-1)
    RGeoPtr rg;
    const unsigned short* rad_pin;   // row of input sample i
    const unsigned short* rad_pout;  // row of frequency k
    const void* rad_b;               // FFT_(p-1)(W_p^(g^m)) / (p - 1) at the row the forward passes leave each frequency
};

__device__ __forceinline__ int n_pad(int i, float inv) { return i + (int)(((float)i + 0.5f) * inv); }  // i + i / q, inv = 1 / q (or 0)

// (CAP: the largest radix the enclosing kernel variant carries -- 16 or 20: the 18- and 20-point butterflies cost the others ~20 registers)
#define XRFT_N_SWITCH(RR, F_)                                                                                                         \
    switch (RR) {                                                                                                                     \
        case 2: F_(2); break; case 3: F_(3); break; case 4: F_(4); break; case 5: F_(5); break; case 6: F_(6); break;                 \
        case 7: F_(7); break; case 8: F_(8); break; case 9: F_(9); break; case 10: F_(10); break; case 11: if (CAP >= 11) { F_(11); } break;             \
        case 12: if (CAP >= 12) { F_(12); } break; case 13: if (CAP >= 13) { F_(13); } break; case 14: if (CAP >= 14) { F_(14); } break; case 15: if (CAP >= 15) { F_(15); } break;                               \
        case 18: if (sizeof(T) == 4 && CAP >= 18) { F_(18); } break; case 20: if (sizeof(T) == 4 && CAP >= 20) { F_(20); } break;                               \
        default: if (CAP >= 16) { F_(16); } break;                                                                                                       \
    }
// (float64: radices up to 16 -- the 18- and 20-point butterflies want more than the 168 registers that leave three waves on a SIMD)
template <typename T> constexpr int fastn_max_radix() { return sizeof(T) == 4 ? 20 : 16; }

// a[k] *= w0^k, k = 1 .. R-1 (the first pass's twiddles W_n^(j k) as powers of ONE table load: fastm.h)
template <typename T, int R> __device__ __forceinline__ void n_chain(C2<T>* a, C2<T> w0) {
    C2<T> w = w0;
#pragma unroll
    for (int k = 1; k < R; ++k) {
        a[k] = cmul(a[k], w);
        if (k + 1 < R) w = cmul(w, w0);
    }
}

// Pass p < np - 1 over the G sequences of a workgroup, in place in the intermediate layout (sequence t at lds + t str).  p == 0 (the transforms of a
// Bluestein convolution: operands in LDS) takes its twiddles as powers of W_n^j, staged at two[0]; the others W_{L_p}^(j k) from the staged table.
template <typename T, int R>
__device__ __forceinline__ void n_pass_mid(C2<T>* lds, NGeoRef g, int p, int tid, int nthr, const C2<T>* twl, const C2<T>* __restrict__ twg) {
    const int m = g.m[p], L = m * R, bps = g.n / R, nb = bps << g.lg, st = g.step[p];
    const float inv_bps = 1.0f / (float)bps, inv_m = 1.0f / (float)m;
    const C2<T>* twp = twl + g.two[p];
    for (int w = tid; w < nb; w += nthr) {
        const int t = fdiv(w, inv_bps), gg = w - t * bps, blk = fdiv(gg, inv_m), j = gg - blk * m;
        C2<T>* s = lds + t * g.str + n_pad(blk * L + j, g.inv_pdq);
        C2<T> a[R];
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = s[q * st];
        dft_r<T, R>(a);
        if (p == 0) n_chain<T, R>(a, twp[j]);  // (a Bluestein plan: W_n^j staged at two[0])
        else {
#pragma unroll
            for (int k = 1; k < R; ++k) a[k] = cmul(a[k], twp[j * R + k]);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) s[k * st] = a[k];
    }
}

// The last pass: runs of R in the intermediate layout -> natural order (frequency k at pn(k)); ONE butterfly per thread, everyone reads before
// anyone writes.  Run blk = ((k_0 r_1 + k_1) r_2 + ...) + k_{np-2} holds the frequencies k_0 + r_0 (k_1 + r_1 (k_2 + ... + r_{np-2} k_{np-1})).
// Starts and ends with a barrier.
template <typename T, int R>
__device__ __forceinline__ void n_pass_last(C2<T>* lds, NGeoRef g, int tid) {
    const int bps = g.n / R, nb = bps << g.lg;
    const bool on = tid < nb;
    const int t = on ? fdiv(tid, 1.0f / (float)bps) : 0, blk = on ? tid - t * bps : 0;
    C2<T>* s = lds + t * g.str;
    C2<T> a[R];
    __syncthreads();
    if (on) {
        const int src = n_pad(blk * R, g.inv_pdq);  // (pdq is R or nothing: the run is contiguous)
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = s[src + q];
    }
    __syncthreads();
    if (on) {
        dft_r<T, R>(a);
        int rem = blk, rest = 0;
        for (int p = g.np - 2; p >= 1; --p) {
            const int q = fdiv(rem, g.inv_r[p]);
            rest = rest * g.r[p] + (rem - q * g.r[p]);
            rem = q;
        }
#pragma unroll
        for (int k2 = 0; k2 < R; ++k2) {
            const int rk = rest + g.wlast * k2, kk = rem + g.r[0] * rk;
            s[g.pn_r0 ? kk + rk : n_pad(kk, g.inv_pnq)] = a[k2];
        }
    }
    __syncthreads();
}

// passes 1 .. np - 1 (the first one has put its results into LDS); ends with the result in natural order, after a barrier
template <typename T, int CAP>
__device__ __forceinline__ void n_fft_tail(C2<T>* lds, NGeoRef g, int tid, int nthr, const C2<T>* twl) {
    for (int p = 1; p + 1 < g.np; ++p) {
        __syncthreads();
#define NM_(RR) n_pass_mid<T, RR>(lds, g, p, tid, nthr, twl, nullptr)
        XRFT_N_SWITCH(g.r[p], NM_)
#undef NM_
    }
#define NL_(RR) n_pass_last<T, RR>(lds, g, tid)
    XRFT_N_SWITCH(g.r[g.np - 1], NL_)
#undef NL_
}

// ------------------------------------------------------------------------------------------------
// pass 1 (fastm_cols_kernel with the geometry as data): a workgroup owns CW = 2 G adjacent real columns of one slab; columns 2g, 2g+1 are the real
// and imaginary part of sequence g.  A thread (g, j) takes the first-pass butterflies j, j + thr / G, ... of its sequence.
// ------------------------------------------------------------------------------------------------
template <typename T> struct NColsCtx {
    const char* src;     // the slab's column block (bytes)
    unsigned rowb;       // bytes per row
    unsigned coff;       // byte offset of this thread's column pair in a row
    unsigned coff1;      // ... of its second column (an odd nx: two loads)
    bool pair_ok, has0, has1;
    const T* wy;
    C2<T> wx;
    float Tl[2], Sl[2];  // float32: the line subtracted here, T + S i (fastm.h)
    bool pre, det;
    double ibar;
    double s[4];
    const C2<T>* blue_c;
    int ny;
};

// UNCONDITIONAL loads (a per-lane branch around a load makes the compiler wait for each one, and all of a butterfly's loads must be in flight together):
// an even nx makes every column pair one aligned 2 T-wide load; an odd nx (pair_ok = 0, a UNIFORM switch) two T-wide loads.  A column beyond the ragged
// edge re-reads column 0 and is zeroed by its window factor (c.wx).
template <typename T> __device__ __forceinline__ C2<T> n_load_pair(const NColsCtx<T>& c, unsigned rowoff) {
    if (c.pair_ok) return *reinterpret_cast<const C2<T>*>(c.src + (rowoff + c.coff));
    return mk<T>(*reinterpret_cast<const T*>(c.src + (rowoff + c.coff)), *reinterpret_cast<const T*>(c.src + (rowoff + c.coff1)));
}

template <typename T, int R, bool BLUE>
__device__ __forceinline__ void n_first_cols(NColsCtx<T>& c, NGeoRef g, C2<T>* seq, int j, const C2<T>* __restrict__ tw) {
    typedef C2<T> CT;
    const int M0 = g.m[0];
    CT a[R];
    T wyv[R];
    const CT w0 = tw[j];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int row = j + q * M0;
        a[q] = mk<T>((T)0, (T)0); wyv[q] = (T)0;
        if (!BLUE || row < c.ny) {
            a[q] = n_load_pair<T>(c, c.rowb * (unsigned)row);
            wyv[q] = c.wy[row];
        }
    }
    if (c.det) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const double ri = (double)(j + q * M0) - c.ibar;  // (Bluestein: the rows beyond ny hold zeros)
            c.s[0] += (double)a[q].re; c.s[1] += (double)a[q].im;
            c.s[2] = fma(ri, (double)a[q].re, c.s[2]); c.s[3] = fma(ri, (double)a[q].im, c.s[3]);
        }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int row = j + q * M0;
        if (c.pre && (!BLUE || row < c.ny)) {
            const float fi = (float)row;
            a[q] = mk<T>((T)((float)a[q].re - fmaf(c.Sl[0], fi, c.Tl[0])), (T)((float)a[q].im - fmaf(c.Sl[1], fi, c.Tl[1])));
        }
        a[q] = mk<T>(a[q].re * (wyv[q] * c.wx.re), a[q].im * (wyv[q] * c.wx.im));
        if (BLUE) { if (row < c.ny) a[q] = cmulc(a[q], c.blue_c[row]); }
    }
    dft_r<T, R>(a);
    n_chain<T, R>(a, w0);
    CT* s = seq + n_pad(j, g.inv_pdq);
    const int st = g.step[0];
#pragma unroll
    for (int k = 0; k < R; ++k) s[k * st] = a[k];
}

// FORM 0: the radix passes; 1: Bluestein's chirp convolution; 2: the prime-factor form with Rader's algorithm along the prime (fastg.h)
template <typename T, int FORM, int CAP>
__global__ void __launch_bounds__(fastn_max_threads<T>(), (sizeof(T) == 4 ? 4 : 3)) fastn_cols_kernel(FastN P) {
    constexpr bool BLUE = FORM == 1, RADER = FORM == 2;
    typedef C2<T> CT;
    NGeoRef g = *P.g;
    const FastM& p = P.f;
    XRFT_DYN_SMEM(smem_raw);
    CT* lds = reinterpret_cast<CT*>(smem_raw);
    CT* twl = lds + g.g * g.str;
    double* part = reinterpret_cast<double*>(twl + g.twn);  // [wave][g][4]
    const int tid = threadIdx.x, nthr = g.thr, G = g.g, gi = tid & (G - 1), r0 = tid >> g.lg, RQ = nthr >> g.lg, CW = 2 * G;
    const int ny = p.ny, nx = p.nx, nyh = ny >> 1;
    // unit = (slab, column block); every XCD gets a contiguous range of units: the workgroups sharing the input's 128-byte lines share an L2 (fastm.h)
    const int per = (p.nunits + 7) >> 3, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int unit = xcd * per + jb;
    if (jb >= per || unit >= p.nunits) return;
    const int nxb = P.nxb, slab = unit / nxb, xb = unit - slab * nxb;
    unsigned short* pin = reinterpret_cast<unsigned short*>(part + (nthr >> 6) * G * 4);
    unsigned short* pout = pin + ((ny + 7) & ~7);
    const int col0 = xb * CW + 2 * gi;  // this thread's column pair
    NColsCtx<T> c;
    c.src = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.in) + (size_t)slab * ny * nx + (size_t)xb * CW);
    c.rowb = (unsigned)nx * (unsigned)sizeof(T);
    c.coff = (unsigned)gi * (unsigned)sizeof(CT);
    c.pair_ok = P.pair_ok != 0;  // (uniform)
    c.has0 = col0 < nx; c.has1 = col0 + 1 < nx;
    c.coff1 = c.has1 ? c.coff + (unsigned)sizeof(T) : 0u;
    if (!c.has0) c.coff = 0;  // (a column beyond the ragged edge: column 0 again, times a zero window)
    c.wy = reinterpret_cast<const T*>(p.win_y);
    {
        const T* __restrict__ wxp = reinterpret_cast<const T*>(p.win_x);
        const T w0 = wxp[min(col0, nx - 1)], w1 = wxp[min(col0 + 1, nx - 1)];
        c.wx = mk<T>(c.has0 ? w0 : (T)0, c.has1 ? w1 : (T)0);
    }
    c.det = p.detrend != 0;
    c.pre = c.det && sizeof(T) == 4;
    c.ibar = 0.5 * (ny - 1);
    c.s[0] = c.s[1] = c.s[2] = c.s[3] = 0.0;
    c.blue_c = reinterpret_cast<const CT*>(P.blue_c);
    c.ny = ny;
    c.Tl[0] = c.Tl[1] = c.Sl[0] = c.Sl[1] = 0.f;
    if (c.pre) {
        // float32: what is subtracted here only has to take the bulk of the trend out (nothing may cancel in float32): a line per column from the
        // medians of three adjacent rows around ny/4 and around 3 ny/4, rounded to a power-of-two grid on which T + S i is exact; pass 2 corrects
        // whatever was subtracted (fastm.h, fasty.h)
        const int ITOP = ny / 4, IBOT = (3 * ny) / 4;
        CT rt[3], rb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rt[k] = n_load_pair<T>(c, c.rowb * (unsigned)(ITOP - 1 + k));
            rb[k] = n_load_pair<T>(c, c.rowb * (unsigned)(IBOT - 1 + k));
        }
        // (the tables are staged while these six rows are in flight: one memory latency, not two, before the first pass)
        for (int e = tid; e < g.twn; e += nthr) twl[e] = reinterpret_cast<const CT*>(P.twm)[e];
        if (RADER) for (int e = tid; e < ny; e += nthr) { pin[e] = P.rad_pin[e]; pout[e] = P.rad_pout[e]; }
        auto med3 = [](float x, float y, float z) { return fmaxf(fminf(x, y), fminf(fmaxf(x, y), z)); };
        const float mt[2] = {med3((float)rt[0].re, (float)rt[1].re, (float)rt[2].re), med3((float)rt[0].im, (float)rt[1].im, (float)rt[2].im)};
        const float mb[2] = {med3((float)rb[0].re, (float)rb[1].re, (float)rb[2].re), med3((float)rb[0].im, (float)rb[1].im, (float)rb[2].im)};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const float top = mt[cc], bot = mb[cc];
            const float Se = p.detrend == 2 ? (bot - top) / (float)(IBOT - ITOP) : 0.f;
            const float Te = p.detrend == 2 ? top - Se * (float)ITOP : 0.5f * (top + bot);
            const float mag = fabsf(Te) + fabsf(Se) * (float)ny;
            const float C = __uint_as_float((__float_as_uint(mag) & 0x7f800000u) + (3u << 23)) * 1.5f;  // rounds to 2^(e-20), 2^e <= mag
            c.Tl[cc] = (Te + C) - C; c.Sl[cc] = (Se + C) - C;
        }
    } else {
        for (int e = tid; e < g.twn; e += nthr) twl[e] = reinterpret_cast<const CT*>(P.twm)[e];
        if (RADER) for (int e = tid; e < ny; e += nthr) { pin[e] = P.rad_pin[e]; pout[e] = P.rad_pout[e]; }
    }
    if (c.det && r0 == 0) {  // what is subtracted, as (offset at ibar, slope)
        double* cfp = p.colfit + ((size_t)slab * nx + col0) * 4;
        if (c.has0) { cfp[2] = (double)c.Tl[0] + (double)c.Sl[0] * c.ibar; cfp[3] = (double)c.Sl[0]; }
        if (c.has1) { cfp[6] = (double)c.Tl[1] + (double)c.Sl[1] * c.ibar; cfp[7] = (double)c.Sl[1]; }
    }
    if (RADER) __syncthreads();  // (the row tables)
    if (BLUE || RADER) {
        // Bluestein: the column pair's ny rows are staged in LDS -- detrended, windowed, times conj(c[i]) -- behind them zeros up to the convolution
        // length, and ALL passes run from LDS (the natural layout of a Bluestein plan is its intermediate layout).  U rows per thread in flight.
        // Rader: the rows are staged at the rows of the prime-factor / generator order, tile [ny][G].
        CT* seq = lds + gi * g.str;
        const CT* __restrict__ ch = reinterpret_cast<const CT*>(P.blue_c);
        constexpr int U = 4;  // (eight in flight: no faster, profiles/r05_rader_cols.txt)
        for (int i0 = r0; i0 < ny && !(P.dbg & 4); i0 += U * RQ) {
            CT v[U], cc_[U];
            T wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ic = min(i0 + u * RQ, ny - 1);
                v[u] = n_load_pair<T>(c, c.rowb * (unsigned)ic);
                wv[u] = c.wy[ic];
                if (BLUE) cc_[u] = ch[ic];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * RQ;
                if (i < ny) {
                    if (c.det) {
                        const double ri = (double)i - c.ibar;
                        c.s[0] += (double)v[u].re; c.s[1] += (double)v[u].im;
                        c.s[2] = fma(ri, (double)v[u].re, c.s[2]); c.s[3] = fma(ri, (double)v[u].im, c.s[3]);
                    }
                    CT z = v[u];
                    if (c.pre) {
                        const float fi = (float)i;
                        z = mk<T>((T)((float)z.re - fmaf(c.Sl[0], fi, c.Tl[0])), (T)((float)z.im - fmaf(c.Sl[1], fi, c.Tl[1])));
                    }
                    z = mk<T>(z.re * (wv[u] * c.wx.re), z.im * (wv[u] * c.wx.im));
                    if (BLUE) seq[n_pad(i, g.inv_pdq)] = cmulc(z, cc_[u]);
                    else lds[(int)pin[i] * G + gi] = z;
                }
            }
        }
        if (BLUE) {
            for (int i = ny + r0; i < g.n; i += RQ) seq[n_pad(i, g.inv_pdq)] = mk<T>((T)0, (T)0);
            __syncthreads();
#define NB_(RR) n_pass_mid<T, RR>(lds, g, 0, tid, nthr, twl, nullptr)
            if (!(P.dbg & 1)) { XRFT_N_SWITCH(g.r[0], NB_) }
#undef NB_
        }
    } else {
        CT* seq = lds + gi * g.str;
        const CT* __restrict__ tw = reinterpret_cast<const CT*>(p.tw_y);
        const int M0 = g.m[0];
        for (int j = r0; j < M0 && !(P.dbg & 4); j += RQ) {
#define NF_(RR) n_first_cols<T, RR, false>(c, g, seq, j, tw)
            XRFT_N_SWITCH(g.r[0], NF_)
#undef NF_
        }
    }
    if (c.det) {
#pragma unroll
        for (int m = G; m < 64; m <<= 1)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) c.s[cc] += __shfl_xor(c.s[cc], m);
        if ((tid & 63) < G) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) part[((tid >> 6) * G + gi) * 4 + cc] = c.s[cc];
        }
    }
    if (RADER && (P.dbg & 1)) __syncthreads();
    else if (RADER) {
        // along q inside every block of q rows (the tail passes of a length-ny transform), then Rader's cyclic convolution of p - 1 points across the first
        // p - 1 blocks (q G sequences side by side): forward passes, * the transformed kernel with the two frequency-0 exchanges, inverse passes (fastg.h)
        constexpr bool X17 = true;
        RGeoRef rg = *P.rg;
        const CT* twp = twl + rg.q;  // (the staged table: W_q, then W_(p-1))
        __syncthreads();
        int L = rg.q;
        for (int ps = 0; ps < rg.nrq; ++ps) {
            fastg_cols_pass<T>(lds, G, ny, G, rg.rq[ps], L, tid, nthr, twl, rg.q / L);
            L /= rg.rq[ps];
            __syncthreads();
        }
        const int P1 = rg.p - 1, qg = rg.q * G;
        L = P1;
        for (int ps = 0; ps + 1 < rg.nrp; ++ps) {  // (the last forward pass runs inside fastg_cols_pass_inv_first)
            fastg_cols_pass<T, X17>(lds, qg, P1, qg, rg.rp[ps], L, tid, nthr, twp);
            L /= rg.rp[ps];
            __syncthreads();
        }
        const CT* __restrict__ bh = reinterpret_cast<const CT*>(P.rad_b);
        fastg_cols_pass_inv_first<T, X17>(lds, qg, P1, qg, rg.rp[rg.nrp - 1], tid, nthr, bh, P1 * qg);
        __syncthreads();
        int Li = rg.rp[rg.nrp - 1];
        for (int ip = rg.nrp - 2; ip >= 0; --ip) {
            Li *= rg.rp[ip];
            fastg_cols_pass_inv<T, X17>(lds, qg, P1, qg, rg.rp[ip], Li, tid, nthr, twp);
            __syncthreads();
        }
    } else if (!(P.dbg & 1)) n_fft_tail<T, CAP>(lds, g, tid, nthr, twl); else __syncthreads();
    if (BLUE && !(P.dbg & 1)) {
        // circular convolution with the chirp: Z1 B, conjugated (the inverse transform is conj FFT conj; 1 / m rides on B), a second forward transform
        // whose first pass finds its operands in LDS -- the Bluestein plan's natural layout IS its intermediate layout --, and Z[k] = conj(res[k] c[k])
        const CT* __restrict__ bh = reinterpret_cast<const CT*>(P.blue_b);
        const int mlen = g.n, tot = mlen << g.lg;
        const float inv_m = 1.0f / (float)mlen;
        for (int e = tid; e < tot; e += nthr) {
            const int t = fdiv(e, inv_m), k = e - t * mlen;
            CT* z = lds + t * g.str + n_pad(k, g.inv_pnq);
            const CT v = cmul(*z, bh[k]);
            *z = mk<T>(v.re, -v.im);
        }
        __syncthreads();
#define NB_(RR) n_pass_mid<T, RR>(lds, g, 0, tid, nthr, twl, nullptr)
        XRFT_N_SWITCH(g.r[0], NB_)
#undef NB_
        n_fft_tail<T, CAP>(lds, g, tid, nthr, twl);
    }
    if (c.det && tid < 4 * G) {  // (sum d, sum (i - ibar) d) per column, the waves' partial sums in wave order
        const int cc = tid >> g.lg, gg = tid & (G - 1);  // cc: 0, 1 = sum d of columns 2gg, 2gg+1; 2, 3 = the first moments
        double acc = 0.0;
        for (int w = 0; w < (nthr >> 6); ++w) acc += part[(w * G + gg) * 4 + cc];
        const int col = xb * CW + 2 * gg + (cc & 1);
        if (col < nx) {
            double* cfp = p.colfit + ((size_t)slab * nx + col) * 4;
            cfp[cc >> 1] = acc;
            if (!c.pre) cfp[2 + (cc >> 1)] = 0.0;
        }
    }
    // split the packed spectra: Ra[k] = (Z[k] + conj Z[N-k]) / 2, Rb[k] = (Z[k] - conj Z[N-k]) / (2i); lanes (ky, column): CW consecutive lanes write the
    // CW columns of a row, RK rows complete a line of the intermediate
    const int rk = 1 << p.l_rk, lcw = g.lg + 1;
    char* __restrict__ w2s = reinterpret_cast<char*>(reinterpret_cast<CT*>(p.w2) + (size_t)slab * p.nrow_pad * P.pitch);
    const int nst = (P.dbg & 2) ? 0 : CW * (nyh + 1);
    for (int l = tid; l < nst; l += nthr) {
        const int col = l & (CW - 1), k = l >> lcw, km = k == 0 ? 0 : ny - k;
        const CT* z = lds + (col >> 1) * g.str;
        CT zk, zc;
        if (RADER) { zk = lds[(int)pout[k] * G + (col >> 1)]; zc = lds[(int)pout[km] * G + (col >> 1)]; }
        else { zk = z[n_pad(k, g.inv_pnq)]; zc = z[n_pad(km, g.inv_pnq)]; }
        if (BLUE) {
            zk = cmul(zk, reinterpret_cast<const CT*>(P.blue_c)[k]); zk.im = -zk.im;
            zc = cmul(zc, reinterpret_cast<const CT*>(P.blue_c)[km]);  // conj(conj(res c)) = res c
        } else {
            zc.im = -zc.im;
        }
        const CT o = (col & 1) ? cscale(mul_mi(zk - zc), (T)0.5) : cscale(zk + zc, (T)0.5);
        const unsigned off = ((((unsigned)(k >> p.l_rk) * (unsigned)nxb + (unsigned)xb) << p.l_rk) + (unsigned)(k & (rk - 1))) * (unsigned)CW + (unsigned)col;
        mr_store_ct_nt<T>(w2s + (size_t)off * sizeof(CT), o);
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2 (fastm_rows_kernel with the geometry as data): a workgroup owns `rpu` consecutive rows ky0.. of the intermediate, adds the plane back,
// transforms along x and writes every row twice: as output row ky (rotated by the fftshift) and, reversed, as row -ky.
// MODE = xrfthip_out_mode: 1 power, 0 complex (fft), 2 cross / 3 cross phase (sequences rpu.. are the same rows of field 1).
// ISO: the radial sums of a RADIAL bin map gathered per bin from the spectra in LDS (fastm.h: no atomics, fixed order); any other map is summed
// from the stored spectrum by radial_binsum_det_kernel.
// ------------------------------------------------------------------------------------------------
template <typename T, int R>
__device__ __forceinline__ void n_first_rows(const FastN& P, C2<T>* lds, int w, int ky0, int slab, bool two) {
    typedef C2<T> CT;
    NGeoRef g = *P.g;
    const FastM& p = P.f;
    const int M0 = g.m[0], rk = 1 << p.l_rk, cwm = (1 << p.l_cw) - 1, nyh = p.ny >> 1;
    const int xq = w >> p.l_rk, pairi = fdiv(xq, 1.0f / (float)M0), j = xq - pairi * M0, t = (pairi << p.l_rk) + (w & (rk - 1));
    const int f = two && t >= P.rpu ? 1 : 0, row = t - f * P.rpu, ky = ky0 + row;
    const bool live = ky <= nyh;  // (padding rows of the last unit were never written by pass 1)
    const CT w0 = reinterpret_cast<const CT*>(p.tw_x)[j];
    const CT* __restrict__ blk = reinterpret_cast<const CT*>(f ? p.w2b : p.w2) + ((size_t)slab * p.nrow_pad + (size_t)((ky >> p.l_rk) << p.l_rk)) * P.pitch;
    const CT* __restrict__ cr = reinterpret_cast<const CT*>(f ? p.corr_b : p.corr) + (size_t)slab * p.nx;
    const bool addback = p.detrend != 0;
    CT a[R], c[R];
    CT h0 = mk<T>((T)0, (T)0), h1 = h0;
    if (addback && live) { h0 = reinterpret_cast<const CT*>(p.what0)[ky]; h1 = reinterpret_cast<const CT*>(p.what1)[ky]; }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        a[q] = mk<T>((T)0, (T)0); c[q] = a[q];
        if (live) {
            const int x = j + q * M0;
            a[q] = blk[((((x >> p.l_cw) << p.l_rk) + (ky & (rk - 1))) << p.l_cw) + (x & cwm)];  // element (ky, x) of [x / CW][ky % RK][x % CW]
            if (addback) c[q] = cr[x];
        }
    }
    if (addback) {  // + wx[x] (alpha_x What0[ky] + gamma_x What1[ky]): the plane, subtracted in the spectral domain
#pragma unroll
        for (int q = 0; q < R; ++q) {
            a[q].re = fma(c[q].re, h0.re, fma(c[q].im, h1.re, a[q].re));
            a[q].im = fma(c[q].re, h0.im, fma(c[q].im, h1.im, a[q].im));
        }
    }
    dft_r<T, R>(a);
    n_chain<T, R>(a, w0);
    CT* s = lds + t * g.str + n_pad(j, g.inv_pdq);
    const int st = g.step[0];
#pragma unroll
    for (int k = 0; k < R; ++k) s[k * st] = a[k];
}

template <typename T, int MODE, bool ISO, int CAP>
__global__ void __launch_bounds__(fastn_max_threads<T>(), (sizeof(T) == 4 ? 4 : 3)) fastn_rows_kernel(FastN P) {
    static_assert(!ISO || MODE == 1 || MODE == 2, "radial sums exist for power and cross spectra");
    typedef C2<T> CT;
    constexpr bool TWO = MODE >= 2;
    NGeoRef g = *P.g;
    const FastM& p = P.f;
    XRFT_DYN_SMEM(smem_raw);
    CT* lds = reinterpret_cast<CT*>(smem_raw);
    CT* twl = lds + g.g * g.str;
    const int tid = threadIdx.x, nthr = g.thr, NX = p.nx, STR = g.str, RPU = P.rpu;
    const int upr = p.nrow_pad / RPU, slab = blockIdx.x / upr, unit = blockIdx.x - slab * upr, ky0 = unit * RPU, nyh = p.ny >> 1;
    constexpr int NTW = 2;  // (the middle passes' twiddles: loaded now, parked in registers, written to LDS behind the first pass -- no load waits for another)
    CT twr[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) { const int e = tid + i * nthr; twr[i] = e < g.twn ? reinterpret_cast<const CT*>(P.twm)[e] : mk<T>((T)0, (T)0); }
    // first pass from registers: item (sequence t, butterfly j) loads x = j + q M0, q < r[0], of its row.  Item order (row pair, j, row in the pair):
    // the RK rows that share the lines of W2 sit in adjacent lanes, so a wave consumes whole lines
    {
        const int nit = g.m[0] << g.lg;
        for (int w = tid; w < nit && !(P.dbg & 4); w += nthr) {
#define NF_(RR) n_first_rows<T, RR>(P, lds, w, ky0, slab, TWO)
            XRFT_N_SWITCH(g.r[0], NF_)
#undef NF_
        }
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) { const int e = tid + i * nthr; if (e < g.twn) twl[e] = twr[i]; }
    for (int e = tid + NTW * nthr; e < g.twn; e += nthr) twl[e] = reinterpret_cast<const CT*>(P.twm)[e];
    if (!(P.dbg & 1)) n_fft_tail<T, CAP>(lds, g, tid, nthr, twl); else __syncthreads();
    const int sx = p.shift_x, sy = p.shift_y;
    const T sc = (T)p.scale;
    const float ipn = g.inv_pnq;
    if (ISO) {
        // A RADIAL bin map (verified on the host, fastm_build_tfirst): the bins of a row are contiguous ranges of |kx| on either side of kx = 0; task =
        // (bin, row, side), 2 RPU adjacent lanes share a bin and their float64 sums meet in lane order by shuffles; only the bins the unit's rows reach
        constexpr int HW = MODE == 2 ? 2 : 1;
        const unsigned bw = p.twin[unit];
        const int blo = (int)(bw & 0xffffu), bhi = (int)(bw >> 16);
        const int H = NX / 2, HM = (NX - 1) / 2;  // |kx| = 0 .. H; kx = nx - |kx| exists for |kx| = 1 .. HM
        const int TPB = 2 * RPU;                  // (a power of two <= 64 dividing the thread count: host)
        const int sub = tid & (TPB - 1), r = sub >> 1, side = sub & 1, kyr = ky0 + r;
        const int ltpb = ilog2c(TPB);
        const bool rlive = kyr <= nyh, twin = kyr != 0 && 2 * kyr != p.ny;
        double* __restrict__ part = p.iso_part + ((size_t)slab * upr + unit) * p.nbins * HW;
        for (int b0 = blo; b0 < bhi; b0 += nthr >> ltpb) {
            const int bn = b0 + (tid >> ltpb);
            double rr = 0.0, ri = 0.0;
            if (rlive && bn < bhi) {
                const unsigned short* __restrict__ fr = p.tfirst + (size_t)kyr * (p.nbins + 1) + bn;
                const int s = fr[0], e = fr[1];  // the bin holds |kx| = s .. e - 1 of this row
                auto take = [&](int kx) {
                    const int ps_ = n_pad(kx, ipn);
                    const CT va = lds[r * STR + ps_];
                    if (MODE == 1) rr += (double)((va.re * va.re + va.im * va.im) * sc);
                    else { const CT v = cscale(cmulc(va, lds[(RPU + r) * STR + ps_]), sc); rr += (double)v.re; ri += (double)v.im; }
                };
                if (side == 0) { for (int m = s; m < min(e, H + 1); ++m) take(m); }
                else { for (int m = max(s, 1); m < min(e, HM + 1); ++m) take(NX - m); }
                if (twin) { rr *= 2.0; ri = 0.0; }  // + the twin row (-ky): V + conj V
            }
            for (int m = 1; m < TPB; m <<= 1) {  // (row, side) in a fixed tree order
                rr += __shfl_down(rr, m, TPB);
                if (MODE == 2) ri += __shfl_down(ri, m, TPB);
            }
            if (sub == 0 && bn < bhi) {
                part[bn * HW] = rr;
                if (MODE == 2) part[2 * bn + 1] = ri;
            }
        }
    }
    if (p.out == nullptr || (P.dbg & 2)) return;
    typedef typename std::conditional<MODE == 0 || MODE == 2, CT, T>::type OutT;
    constexpr int VW = 16 / (int)sizeof(OutT);
    if (p.half || !P.vec_ok) {
        // one sample per lane and store (rows of nx/2 + 1 samples -- real_dim --, or a row length the 16-byte stores do not divide): whole lines per wave still
        const int W = p.half ? NX / 2 + 1 : NX;
        const float inv_w = 1.0f / (float)W;
        OutT* __restrict__ oh = reinterpret_cast<OutT*>(p.out) + (size_t)slab * p.ny * W;
        const int tot = RPU * 2 * W;
        for (int e = tid; e < tot; e += nthr) {
            const int rr = fdiv(e, inv_w), oc = e - rr * W, r = rr >> 1, mir = rr & 1, ky = ky0 + r;
            if (ky > nyh || (mir && (ky == 0 || 2 * ky == p.ny))) continue;
            int fx = oc;
            if (!p.half) { fx = oc - sx; if (fx < 0) fx += NX; }  // unshifted frequency of output column oc
            const int fy = mir ? p.ny - ky : ky, kx = mir ? (fx == 0 ? 0 : NX - fx) : fx;
            int orow = fy + sy; if (orow >= p.ny) orow -= p.ny;
            const T f2 = (p.half && p.realdim2 && fx != 0 && 2 * fx != NX) ? (T)2 : (T)1;
            const int ps_ = n_pad(kx, ipn);
            CT va = lds[r * STR + ps_];
            OutT* dst = oh + (size_t)orow * W + oc;
            if (MODE == 1) {
                *reinterpret_cast<T*>(dst) = (va.re * va.re + va.im * va.im) * (sc * f2);
            } else {
                if (TWO) va = cmulc(va, lds[(RPU + r) * STR + ps_]);
                va = cscale(va, sc * f2);
                if (mir) va = cconj(va);
                if (p.ph_on) va = cmul(va, cmul(reinterpret_cast<const CT*>(p.ph_y)[fy], reinterpret_cast<const CT*>(p.ph_x)[fx]));
                if (MODE == 3) *reinterpret_cast<T*>(dst) = (T)atan2((double)va.im, (double)va.re);
                else *reinterpret_cast<CT*>(dst) = va;
            }
        }
        return;
    }
    // whole rows, 16 bytes per lane and store: VW samples
    const int CPR = NX / VW, tot = RPU * 2 * CPR;
    const float inv_cpr = 1.0f / (float)CPR;
    OutT* __restrict__ outs = reinterpret_cast<OutT*>(p.out) + (size_t)slab * p.ny * NX;
    for (int e = tid; e < tot; e += nthr) {
        const int rr = fdiv(e, inv_cpr), chunk = e - rr * CPR, r = rr >> 1, mir = rr & 1;
        const int ky = ky0 + r;
        if (ky > nyh || (mir && (ky == 0 || 2 * ky == p.ny))) continue;
        const CT* rowA = lds + r * STR;
        const CT* rowB = lds + (RPU + r) * STR;  // (TWO)
        const int fy = mir ? p.ny - ky : ky;
        int orow = fy + sy; if (orow >= p.ny) orow -= p.ny;
        const int c = chunk * VW;
        alignas(16) OutT o[VW];
        CT py = mk<T>((T)1, (T)0);
        if (MODE != 1 && p.ph_on) py = reinterpret_cast<const CT*>(p.ph_y)[fy];
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            int fx = c + i - sx; if (fx < 0) fx += NX;      // unshifted frequency of output column c + i
            const int kx = mir ? (fx == 0 ? 0 : NX - fx) : fx;    // F(-ky, fx) = conj F(ky, -fx)
            const int ps_ = n_pad(kx, ipn);
            CT va = rowA[ps_];
            if (MODE == 1) {
                reinterpret_cast<T*>(o)[i] = (va.re * va.re + va.im * va.im) * sc;
            } else {
                if (TWO) va = cmulc(va, rowB[ps_]);  // F0 conj(F1)
                va = cscale(va, sc);
                if (mir) va = cconj(va);
                if (p.ph_on) va = cmul(va, cmul(py, reinterpret_cast<const CT*>(p.ph_x)[fx]));
                if (MODE == 3) reinterpret_cast<T*>(o)[i] = (T)atan2((double)va.im, (double)va.re);
                else reinterpret_cast<CT*>(o)[i] = va;
            }
        }
        mr_store16_nt<T>(outs + (size_t)orow * NX + c, o);
    }
}

// ------------------------------------------------------------------------------------------------
// Two ADJACENT transform axes with the independent elements INNERMOST -- dim = ["y", "x"] of a (y, x, time) array, xrfthip_desc.inner:
// [batch][ny][nx][inner] -- as the same two passes (round 5; before: a detrend pass and two one-axis plans, 32 bytes per sample through memory):
//   pass 1  fastn_cols_kernel on the [ny][nx inner] view: every (x, e) is a column, two adjacent ones (e, e + 1) a packed sequence; the window along x rides on a
//           table expanded to the view's columns; the per-column sums are those of the (x, e) columns
//   [fit]   fastn_fit_inner_kernel: one plane per (slab, e) from its nx column sums
//   pass 2  fastn_irows_kernel: a workgroup owns GE consecutive e of ONE row ky of the intermediate -- GE complex sequences of nx points, x strided by `inner` --
//           adds the plane back, transforms along x and writes (ky, kx, e) and its Hermitian twin (-ky, -kx, e) as runs of GE elements where they lie.
// 16 algorithmic-plus-intermediate bytes per sample instead of 32; no transposed copy.      (xrft/xrft.py:395-409: any axes where they lie)
// ------------------------------------------------------------------------------------------------
struct FastNI {
    const void* w2;      // [slab][ky / RK][c / CW][ky % RK][CW] complex T, c = x inner + e (pitch columns per row)
    const void* corr;    // [slab][nx inner] complex T: wx (subtracted line - plane) as (offset at ibar, slope)
    const void* what0;   // FFT_y(wy)[ky], FFT_y(wy (i - ibar))[ky]
    const void* what1;
    const void* tw_x;    // W_nx^k
    const void* twm;     // staged twiddles of the middle passes
    NGeoPtr g;           // the nx-point transform, g.g = GE sequences per workgroup
    const void* ph_y;    // complex mode: combined phase factors per unshifted frequency
    const void* ph_x;
    void* out;           // [slab][ny][nx][inner]: T (power) or complex T
    int ph_on, vec, dbg; // vec: the result is written in 16-byte pieces; dbg: ablations for measurements (1 no LDS passes, 2 no stores, 4 no loads)
    int ny, nx, inner, nrow_pad, pitch, l_cw, l_rk, detrend, shift_y, shift_x, neb, nunits;
    // `inner` = the independent elements e of a row; sample (x, e) is column x sx + e se of the view -- the elements INNERMOST ([ny][nx][inner]: sx = inner, se = 1) or
    // BETWEEN the two axes ([ny][mid][nx], dim = ["time", "lon"] of a (time, lat, lon) array: sx = 1, se = nx; `midlay`: the lanes then run along x, not along e)
    int sx, se, midlay;
    // real_dim along the second axis (XRFTHIP_HALF_X, xrft.py:400-404): rows of nx/2 + 1 samples, unshifted along x; realdim2: 0 < kx < nx/2 counts twice (xrft.py:673-682)
    int half, realdim2;
    int half_y;          // real_dim along the FIRST axis (XRFTHIP_HALF_Y): rows ky = 0 .. ny/2 only, no twin rows; realdim2 then doubles 0 < ky < ny/2
    // MODE 2, the cross spectrum of two fields (xrft.cross_spectrum, xrft.py:825): the workgroup's sequences are GE/2 of field 0 followed by the SAME GE/2 elements of field 1
    // (its intermediate and plane corrections: w2b, corrb); the result is F0 conj(F1), the twin row its conjugate
    const void* w2b;
    const void* corrb;
    double scale;
};

template <typename T, int R>
__device__ __forceinline__ void n_first_irows(const FastNI& p, NGeoRef g, C2<T>* lds, int w, int ky, int slab, int e0) {
    typedef C2<T> CT;
    const int M0 = g.m[0], rk = 1 << p.l_rk, cwm = (1 << p.l_cw) - 1;
    int ge, j;
    if (p.midlay) { ge = w / M0; j = w - ge * M0; }  // (lanes along x: the samples of a sequence are contiguous)
    else { ge = w & (g.g - 1); j = w >> g.lg; }
    const int gh = p.w2b ? (g.g >> 1) : g.g;   // sequences per field
    const bool fb = ge >= gh;                  // (cross spectrum: the second half of the workgroup's sequences is field 1)
    const int e = min(e0 + (fb ? ge - gh : ge), p.inner - 1);  // (a ragged last block re-reads the last element; never stored)
    const CT w0 = reinterpret_cast<const CT*>(p.tw_x)[j];
    const CT* __restrict__ blk = reinterpret_cast<const CT*>(fb ? p.w2b : p.w2) + ((size_t)slab * p.nrow_pad + (size_t)((ky >> p.l_rk) << p.l_rk)) * p.pitch;
    const CT* __restrict__ cr = reinterpret_cast<const CT*>(fb ? p.corrb : p.corr) + (size_t)slab * p.nx * p.inner;
    const bool addback = p.detrend != 0;
    CT a[R], c[R];
    CT h0 = mk<T>((T)0, (T)0), h1 = h0;
    if (addback) { h0 = reinterpret_cast<const CT*>(p.what0)[ky]; h1 = reinterpret_cast<const CT*>(p.what1)[ky]; }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int col = (j + q * M0) * p.sx + e * p.se;
        if (p.dbg & 4) { a[q] = mk<T>((T)col, (T)ky); c[q] = a[q]; continue; }
        a[q] = blk[((((col >> p.l_cw) << p.l_rk) + (ky & (rk - 1))) << p.l_cw) + (col & cwm)];
        c[q] = addback ? cr[col] : mk<T>((T)0, (T)0);
    }
    if (addback) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            a[q].re = fma(c[q].re, h0.re, fma(c[q].im, h1.re, a[q].re));
            a[q].im = fma(c[q].re, h0.im, fma(c[q].im, h1.im, a[q].im));
        }
    }
    dft_r<T, R>(a);
    n_chain<T, R>(a, w0);
    CT* s = lds + ge * g.str + n_pad(j, g.inv_pdq);
    const int st = g.step[0];
#pragma unroll
    for (int k = 0; k < R; ++k) s[k * st] = a[k];
}

// MODE 0: complex spectrum, 1: power spectrum, 2: cross spectrum of two fields (complex)
template <typename T, int MODE, int CAP>
__global__ void __launch_bounds__(fastn_max_threads<T>(), (sizeof(T) == 4 ? 4 : 3)) fastn_irows_kernel(FastNI p) {
    typedef C2<T> CT;
    NGeoRef g = *p.g;
    XRFT_DYN_SMEM(smem_raw);
    CT* lds = reinterpret_cast<CT*>(smem_raw);
    CT* twl = lds + g.g * g.str;
    const int tid = threadIdx.x, nthr = g.thr, GE = g.g, NX = p.nx, nyh = p.ny >> 1;
    // unit = (slab, ky, block of GE elements e); every XCD gets a contiguous range of units: the workgroups that fill the 128-byte lines of a result row share an L2
    const int per = (p.nunits + 7) >> 3, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int unit = xcd * per + jb;
    if (jb >= per || unit >= p.nunits) return;
    constexpr int LX = MODE == 2 ? 1 : 0;  // a cross spectrum's workgroup holds GE >> 1 elements of both fields
    const int GEO = GE >> LX, lgo = g.lg - LX;  // elements (and their log2) the workgroup writes
    const int ups = (nyh + 1) * p.neb, slab = unit / ups, rem = unit - slab * ups, ky = rem / p.neb, e0 = (rem - ky * p.neb) * GEO;
    for (int e = tid; e < g.twn; e += nthr) twl[e] = reinterpret_cast<const CT*>(p.twm)[e];
    {
        const int nit = g.m[0] << g.lg;
        for (int w = tid; w < nit; w += nthr) {
#define NF_(RR) n_first_irows<T, RR>(p, g, lds, w, ky, slab, e0)
            XRFT_N_SWITCH(g.r[0], NF_)
#undef NF_
        }
    }
    if (!(p.dbg & 1)) n_fft_tail<T, CAP>(lds, g, tid, nthr, twl); else __syncthreads();
    const int sx = p.shift_x, sy = p.shift_y;
    const T sc = (T)p.scale;
    // WO: samples of a result row -- every kx, or kx = 0 .. nx/2 as it lies (real_dim; the twin row (-ky) then takes its samples from kx = nx - fx of the same sequences)
    const int WO = p.half ? NX / 2 + 1 : NX;
    const float ipn = g.inv_pnq, inv_nx = 1.0f / (float)WO;
    if ((p.dbg & 2) && lds[tid].re != (T)1.2345) return;
    const bool interior_y = ky != 0 && 2 * ky != p.ny, twin = interior_y && !p.half_y;
    const int tot = (WO << lgo) * (twin ? 2 : 1);
    const int NYO = p.half_y ? nyh + 1 : p.ny;  // rows of a slab of the result
    const bool dbl_y = p.half_y && p.realdim2 && interior_y;  // (the kept half of the real FIRST axis counts twice)
    const int fbo = GEO * g.str;  // (MODE 2: field 1's sequence of an element, behind field 0's)
    typedef typename std::conditional<MODE != 1, CT, T>::type OutT;
    OutT* __restrict__ outs = reinterpret_cast<OutT*>(p.out) + (size_t)slab * NYO * WO * p.inner;
    if (p.vec) {  // 16-byte pieces of the result: VW consecutive elements e per thread (the host checked inner % VW == 0 and GE % VW == 0)
        constexpr int VW = 16 / (int)sizeof(OutT), LV = VW == 4 ? 2 : VW == 2 ? 1 : 0;
        const int lgq = lgo - LV, totv = tot >> LV;
        for (int idx = tid; idx < totv; idx += nthr) {
            const int ge = (idx & ((1 << lgq) - 1)) << LV, rest = idx >> lgq, mir = fdiv(rest, inv_nx), oc = rest - mir * WO, e = e0 + ge;
            if (e >= p.inner) continue;
            int fx = oc - sx; if (fx < 0) fx += NX;
            const int kx = mir ? (fx == 0 ? 0 : NX - fx) : fx;
            const int fy = mir ? p.ny - ky : ky;
            int orow = fy + sy; if (orow >= p.ny) orow -= p.ny;
            const CT* src = lds + ge * g.str + n_pad(kx, ipn);
            OutT o[VW];
            const T scv = (dbl_y || (!p.half_y && p.realdim2 && fx != 0 && 2 * fx != NX)) ? sc + sc : sc;
            CT ph = mk<T>((T)1, (T)0);
            if (MODE != 1 && p.ph_on) ph = cmul(reinterpret_cast<const CT*>(p.ph_y)[fy], reinterpret_cast<const CT*>(p.ph_x)[fx]);
#pragma unroll
            for (int i = 0; i < VW; ++i) {
                CT v = src[i * g.str];
                if (MODE == 2) v = cmulc(v, src[i * g.str + fbo]);  // F0 conj(F1)
                if (MODE == 1) {
                    *reinterpret_cast<T*>(&o[i]) = (v.re * v.re + v.im * v.im) * scv;
                } else {
                    v = cscale(v, MODE == 2 ? scv : sc);  // (a cross spectrum's kept half of a real axis counts twice, too: xrft.py:673-682)
                    if (mir) v = cconj(v);
                    if (p.ph_on) v = cmul(v, ph);
                    *reinterpret_cast<CT*>(&o[i]) = v;
                }
            }
            mr_store16_nt<T, true>(outs + ((size_t)orow * WO + oc) * p.inner + e, o);  // (a plain store: the pieces of a 128-byte line come from several workgroups and meet in the L2; non-temporal pieces go to memory one by one, 652 us against 244)
        }
        return;
    }
    const size_t osx = p.midlay ? 1 : (size_t)p.inner, ose = p.midlay ? (size_t)WO : 1;  // the result's strides along kx and along e
    for (int idx = tid; idx < tot; idx += nthr) {
        int ge, mir, oc;
        if (p.midlay) { const int rest = fdiv(idx, inv_nx); oc = idx - rest * WO; mir = rest >> lgo; ge = rest & (GEO - 1); }  // (lanes along kx: contiguous stores)
        else { ge = idx & (GEO - 1); const int rest = idx >> lgo; mir = fdiv(rest, inv_nx); oc = rest - mir * WO; }
        const int e = e0 + ge;
        if (e >= p.inner) continue;
        int fx = oc - sx; if (fx < 0) fx += NX;                 // unshifted frequency of output column oc
        const int kx = mir ? (fx == 0 ? 0 : NX - fx) : fx;      // F(-ky, fx) = conj F(ky, -fx)
        const int fy = mir ? p.ny - ky : ky;
        int orow = fy + sy; if (orow >= p.ny) orow -= p.ny;
        CT v = lds[ge * g.str + n_pad(kx, ipn)];
        if (MODE == 2) v = cmulc(v, lds[ge * g.str + fbo + n_pad(kx, ipn)]);  // F0 conj(F1)
        OutT* dst = outs + (size_t)orow * WO * p.inner + (size_t)oc * osx + (size_t)e * ose;
        if (MODE == 1) {
            *reinterpret_cast<T*>(dst) = (v.re * v.re + v.im * v.im) * ((dbl_y || (!p.half_y && p.realdim2 && fx != 0 && 2 * fx != NX)) ? sc + sc : sc);
        } else {
            v = cscale(v, (MODE == 2 && (dbl_y || (!p.half_y && p.realdim2 && fx != 0 && 2 * fx != NX))) ? sc + sc : sc);
            if (mir) v = cconj(v);
            if (p.ph_on) v = cmul(v, cmul(reinterpret_cast<const CT*>(p.ph_y)[fy], reinterpret_cast<const CT*>(p.ph_x)[fx]));
            *reinterpret_cast<CT*>(dst) = v;
        }
    }
}

// one plane per (slab, e) from the sums of its nx columns (fastm_fit_kernel with the columns of an element sx apart, the elements se): one 256-thread block each
template <typename T>
__global__ void __launch_bounds__(256) fastn_fit_inner_kernel(const double* colfit, const T* win_x_exp, C2<T>* corr, int nx, int inner, int ny, int detrend, int sx, int se) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);
    const int slab = blockIdx.x / inner, e = blockIdx.x - slab * inner, tid = threadIdx.x;
    const double* cf4 = colfit + (size_t)slab * nx * inner * 4;
    const double xbar = 0.5 * (nx - 1), sxx = (double)nx * ((double)nx * nx - 1.0) / 12.0;
    const double inv_n = 1.0 / ny, inv_sii = 12.0 / ((double)ny * ((double)ny * ny - 1.0));
    double s[3] = {0.0, 0.0, 0.0};
    for (int x = tid; x < nx; x += 256) {
        const size_t c = (size_t)x * sx + (size_t)e * se;
        const double m = cf4[4 * c] * inv_n, sl = cf4[4 * c + 1] * inv_sii;
        s[0] += m;
        s[1] += ((double)x - xbar) * m;
        s[2] += sl;
    }
    block_sum<3>(s, red);
    __syncthreads();
    if (tid == 0) { red[0] = s[0]; red[1] = s[1]; red[2] = s[2]; }
    __syncthreads();
    const double a = red[0] / nx;
    const double b = (detrend == 2 && nx > 1) ? red[1] / sxx : 0.0;
    const double cc = detrend == 2 ? red[2] / nx : 0.0;
    C2<T>* out = corr + (size_t)slab * nx * inner;
    for (int x = tid; x < nx; x += 256) {
        const size_t c = (size_t)x * sx + (size_t)e * se;
        const double wx = (double)win_x_exp[c];
        out[c] = mk<T>((T)(wx * (cf4[4 * c + 2] - a - b * ((double)x - xbar))), (T)(wx * (cf4[4 * c + 3] - cc)));
    }
}

}  // namespace xrft

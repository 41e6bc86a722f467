// fastg.h -- ONE pass over a small real slab of ANY smooth shape, either precision: the whole 2-D transform of a slab inside the LDS of one
// workgroup, the lengths as DATA (run-time radices), not as template arguments (xrft.power_spectrum / fft over the last two axes of
// (nt, ny, nx) arrays: reference xrft/xrft.py:307-476, 685-750, detrend.py:100-113 -- the reference's documented workload is thousands of
// 50 x 50 boxes).
//
// Where fasts.h (float32, 64 | 128 | 256 points per axis) keeps a slab in registers with compile-time radices, this kernel keeps the slab's
// half spectrum -- ny rows of nx/2 + 1 complex values -- in LDS and runs the radix passes of tile_fft.h over it with the radices of a
// parameter block: any ny, nx = 2^a 3^b 5^c whose half spectrum fits the LDS (19 500 complex64 / 9 700 complex128 values:
// 50 x 50, 96 x 96, 100 x 100, 120 x 240, 150 x 150, 180 x 90; in float64 also 64 ... 128 points per axis).  Before it these shapes took the
// generic tile kernels in two passes through memory with a moments pass in front (50-115 GFFT/s in float32, 40-60 in float64).
//
//   load      the slab, rows packed in pairs of samples z[i][m] = x[i][2m] + i x[i][2m+1], into LDS (coalesced); the exact plane of
//             detrend='linear' from three float64 sums over the workgroup (fixed order), subtracted, the window multiplied, in a second sweep
//   x         the radix passes of length n = nx/2 over the ny rows (tile_fft.h run_pass: decimation in frequency, digit-reversed result),
//             then the unpack of the packed rows in place: X[k] = E + W_nx^k O, X[n - k] = conj(E - W_nx^k O) at the positions of Z[k], Z[n - k];
//             X[n] in column n
//   y         the radix passes of length ny over the n + 1 columns (lanes along the columns: contiguous)
//   out       every output sample in output order (coalesced stores): its source (ky, kx) or the Hermitian twin (-ky, -kx), looked up through
//             the digit-reversal tables of the two axes; |F|^2 scale, or F scale x the true-phase factors (conjugated for the twin)
//
// The same kernel, by parameter: an ODD nx (rows as complex sequences, the whole spectrum in the tile); real_dim (the half spectrum as it lies);
// the radial sums of isotropic_power_spectrum / isotropic_cross_spectrum (per-bin lists of LDS positions, any bin map, no atomics); MODE 2, the cross
// spectrum of two fields (two tiles); `one_d`, 1-D transforms along x on groups of rows (no y passes, a mean / line per row).
// fastgy_kernel (below): ONE transform axis that is NOT the contiguous one on any smooth length -- real columns packed in pairs, complex columns, two
// real fields; Bluestein inside the tile for a prime factor that has no butterfly.
#pragma once
#include "tile_fft.h"
#include "fastr.h"  // fastr_store4 / fastr_store8

namespace xrft {

constexpr int kFastGMaxPasses = 8;
constexpr int kFastGWaves = 16;  // at most: 1024 threads in float32, 512 in float64 (its radix-16 butterflies want more than 128 registers)
template <typename T> constexpr int fastg_max_threads() { return sizeof(T) == 4 ? 1024 : 512; }

struct FastG {
    const void* in;     // [slabs][ny][nx] real T
    const void* in_b;   // cross spectrum: the second field (same layout); the result is F(in) conj F(in_b) (xrft.py:825)
    void* out;          // [slabs][ny][nx] real T (power) or complex T
    long long nslabs;
    int ny, nx, n, rs;  // n = nx / 2 (packed rows) or nx (an odd nx: the rows as complex sequences, imaginary parts zero); rs = LDS row stride in complex elements (>= n + 1 | nx)
    int packed;         // nx even: rows packed in pairs of samples, the half spectrum in the tile; else the whole spectrum
    int one_d;          // a 1-D transform along x of `nrows` rows, ny of them per workgroup: no y passes (nry = 0, rev_y the identity), a mean / line per ROW
    int lpr;            // ... lanes that share a row in the per-row sums (a power of two <= 64)
    int nred;           // doubles of the sums' scratch: 3 per wave (plane), 2 per row (one_d)
    long long nrows;
    // complex input (xrft.fft of complex data, and every inverse transform: xrft.ifft, xrft.py:479-646): [slabs][ny][nx] complex T, the whole spectrum in the tile
    // (packed = 0), no detrend.  inv: conj(FFT(conj(z))); ishy / ishx: sample (i, m) of the tile is source (i + ishy, m + ishx) mod (ny, nx) -- the ifftshift of an
    // fftshifted spectrum; ph_in: the phase tables multiply the INPUT at its source position (the lag's phase, xrft.py:574-576)
    int cin, inv, ishy, ishx, ph_in;
    // irfftn (XRFTHIP_C2R_X, xrft.ifft with real_dim): the input is the HALF spectrum [slabs][ny][nx/2 + 1] complex T, the output REAL [slabs][ny][nx] T (MODE 1).
    // conj(X) phase is loaded into the packed geometry, the y passes run FIRST (on the nx/2 + 1 columns), every row is re-packed in place --
    // W[k] = T[k] + conj T[n-k] - i (T[k] - conj T[n-k]) W_nx^k = conj Z[k], Z the spectrum of z[m] = x[2m] + i x[2m+1] -- the x passes, and
    // x[2m] = Re R[m], x[2m+1] = -Im R[m] on the way out
    int c2r;
    int nrx, nry;
    int rx[kFastGMaxPasses], ry[kFastGMaxPasses];
    const void* tw_x;   // W_n^k,  k < n   (complex T)
    const void* tw_y;   // W_ny^k, k < ny
    const void* tw_r;   // W_nx^k, k <= n
    const unsigned* rev_x;  // position of frequency k after the passes of length n
    const unsigned* rev_y;  // ... of length ny
    const void* win_y;  // T, or null (then win_x is null, too)
    const void* win_x;
    const void* ph_y;   // complex mode: true-phase factors by unshifted frequency index (complex T)
    const void* ph_x;
    int ph_on;
    int detrend;        // 0 none, 1 constant, 2 linear (plane)
    int shift_y, shift_x;  // 0 or n/2 (the fftshift offsets, xrft.py:446-447)
    int half;              // real_dim: only kx = 0..nx/2 is stored, rows of nx/2 + 1 samples, unshifted (xrft.py:400-404)
    int realdim2;          // ... and 0 < kx < nx/2 counts twice (xrft.py:673-682)
    double scale;
    // radial sums (xrft.isotropic_power_spectrum, xrft.py:895-906): per bin the LDS positions of its samples (any bin map; a sample of the
    // right half plane is its Hermitian twin's position), [nbins + 1] starts into the list; iso[slab][nbins].  out may then be null.
    const unsigned short* iso_pos;
    const unsigned* iso_start;
    double* iso;
    int nbins;
};

// one radix pass over the COLUMNS of the tile: sequences of length len, element stride rs, ncols of them; lanes run along the columns
// (tws: the stride of W^(j k) in `tw` when that is not len / L -- the passes along q of the prime-factor form run on blocks of a length-n tile with a table of W_q alone)
template <typename T, int R>
__device__ __forceinline__ void fastg_pass_cols(C2<T>* tile, int ncols, int len, int rs, int L, int tid, int nthreads, const C2<T>* __restrict__ tw, int tws = 0) {
    const int m = L / R, per = len / R, nb = ncols * per, twstep = tws ? tws : len / L;
    const float inv_c = 1.0f / (float)ncols, inv_m = 1.0f / (float)m;
    for (int w = tid; w < nb; w += nthreads) {
        const int gg = fdiv(w, inv_c), c = w - gg * ncols;
        const int blk = fdiv(gg, inv_m), j = gg - blk * m;
        C2<T>* s = tile + c + (blk * L + j) * rs;
        C2<T> a[R];
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = s[q * m * rs];
        dft_r<T, R>(a);
        if (m > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) a[k] = cmul(a[k], tw[j * k * twstep]);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) s[k * m * rs] = a[k];
    }
}

// exact inverse of fastg_pass_cols up to the factor R (the Bluestein convolution of fastgy_kernel): undo the twiddles with their conjugates, then the
// unnormalised inverse butterfly (as run_pass_inv of tile_fft.h)
template <typename T, int R>
__device__ __forceinline__ void fastg_pass_cols_inv(C2<T>* tile, int ncols, int len, int rs, int L, int tid, int nthreads, const C2<T>* __restrict__ tw) {
    const int m = L / R, per = len / R, nb = ncols * per, twstep = len / L;
    const float inv_c = 1.0f / (float)ncols, inv_m = 1.0f / (float)m;
    for (int w = tid; w < nb; w += nthreads) {
        const int gg = fdiv(w, inv_c), c = w - gg * ncols;
        const int blk = fdiv(gg, inv_m), j = gg - blk * m;
        C2<T>* s = tile + c + (blk * L + j) * rs;
        C2<T> a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = cconj(s[k * m * rs]);
        if (m > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) a[k] = cmul(a[k], tw[j * k * twstep]);  // conj(a conj(w)) = conj(a) w
        }
        dft_r<T, R>(a);
#pragma unroll
        for (int q = 0; q < R; ++q) s[q * m * rs] = cconj(a[q]);
    }
}
// The middle of Rader's convolution in ONE trip through the LDS: the LAST forward pass, the product by the transformed kernel and the FIRST inverse pass all work on
// the same blocks of R consecutive rows, without twiddles (two trips and two barriers less than pass, product, pass); and the two frequency-0 exchanges of the algorithm: row 0 holds the sum S of the samples with n2 != 0,
// block zoff the sample z with n2 = 0: X[.][0] = z + S goes to zoff, S B[0] + z (z reaches every other output) into the convolution.
template <typename T, int R>
__device__ __forceinline__ void fastg_pass_cols_inv_first(C2<T>* tile, int ncols, int len, int rs, int tid, int nthreads, const C2<T>* __restrict__ bh, int zoff) {
    const int per = len / R, nb = ncols * per;
    const float inv_c = 1.0f / (float)ncols;
    for (int w = tid; w < nb; w += nthreads) {
        const int blk = fdiv(w, inv_c), c = w - blk * ncols;
        C2<T>* s = tile + c + (blk * R) * rs;
        C2<T> a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = s[k * rs];
        dft_r<T, R>(a);  // (the LAST forward pass works on the same blocks of R rows, without twiddles: it runs here, in registers)
        if (blk == 0) {
            const C2<T> S = a[0], z = tile[zoff + c];
            tile[zoff + c] = mk<T>(z.re + S.re, z.im + S.im);
            a[0] = cmul(S, bh[0]);
            a[0] = mk<T>(a[0].re + z.re, a[0].im + z.im);
#pragma unroll
            for (int k = 1; k < R; ++k) a[k] = cmul(a[k], bh[k]);
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) a[k] = cmul(a[k], bh[blk * R + k]);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = cconj(a[k]);
        dft_r<T, R>(a);
#pragma unroll
        for (int q = 0; q < R; ++q) s[q * rs] = cconj(a[q]);
    }
}
template <typename T, bool X17 = false>
__device__ __forceinline__ void fastg_cols_pass_inv_first(C2<T>* tile, int ncols, int len, int rs, int R, int tid, int nthr, const C2<T>* bh, int zoff) {
    if (X17 && R == 17) { fastg_pass_cols_inv_first<T, X17 ? 17 : 2>(tile, ncols, len, rs, tid, nthr, bh, zoff); return; }
    switch (R) {
        case 2: fastg_pass_cols_inv_first<T, 2>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 3: fastg_pass_cols_inv_first<T, 3>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 4: fastg_pass_cols_inv_first<T, 4>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 5: fastg_pass_cols_inv_first<T, 5>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 6: fastg_pass_cols_inv_first<T, 6>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 7: fastg_pass_cols_inv_first<T, 7>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 8: fastg_pass_cols_inv_first<T, 8>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 9: fastg_pass_cols_inv_first<T, 9>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 10: fastg_pass_cols_inv_first<T, 10>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 11: fastg_pass_cols_inv_first<T, 11>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 12: fastg_pass_cols_inv_first<T, 12>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 13: fastg_pass_cols_inv_first<T, 13>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 14: fastg_pass_cols_inv_first<T, 14>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        case 15: fastg_pass_cols_inv_first<T, 15>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
        default: fastg_pass_cols_inv_first<T, 16>(tile, ncols, len, rs, tid, nthr, bh, zoff); break;
    }
}

template <typename T, bool X17 = false>  // (X17: the 17-point butterfly, too -- the Rader forms only: it would cost every other kernel registers)
__device__ __forceinline__ void fastg_cols_pass_inv(C2<T>* tile, int ncols, int len, int rs, int R, int L, int tid, int nthr, const C2<T>* tw) {
    if (X17 && R == 17) { fastg_pass_cols_inv<T, X17 ? 17 : 2>(tile, ncols, len, rs, L, tid, nthr, tw); return; }
    switch (R) {
        case 2: fastg_pass_cols_inv<T, 2>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 3: fastg_pass_cols_inv<T, 3>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 4: fastg_pass_cols_inv<T, 4>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 5: fastg_pass_cols_inv<T, 5>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 6: fastg_pass_cols_inv<T, 6>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 7: fastg_pass_cols_inv<T, 7>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 8: fastg_pass_cols_inv<T, 8>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 9: fastg_pass_cols_inv<T, 9>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 10: fastg_pass_cols_inv<T, 10>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 11: fastg_pass_cols_inv<T, 11>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 12: fastg_pass_cols_inv<T, 12>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 13: fastg_pass_cols_inv<T, 13>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 14: fastg_pass_cols_inv<T, 14>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        case 15: fastg_pass_cols_inv<T, 15>(tile, ncols, len, rs, L, tid, nthr, tw); break;
        default: fastg_pass_cols_inv<T, 16>(tile, ncols, len, rs, L, tid, nthr, tw); break;
    }
}

template <typename T>
__device__ __forceinline__ void fastg_rows_pass(C2<T>* tile, const TileGeom& g, int R, int L, int tid, int nthr, const C2<T>* tw) {
    switch (R) {
        case 2: run_pass<T, 2>(tile, g, L, tid, nthr, tw); break;
        case 3: run_pass<T, 3>(tile, g, L, tid, nthr, tw); break;
        case 4: run_pass<T, 4>(tile, g, L, tid, nthr, tw); break;
        case 5: run_pass<T, 5>(tile, g, L, tid, nthr, tw); break;
        case 6: run_pass<T, 6>(tile, g, L, tid, nthr, tw); break;
        case 7: run_pass<T, 7>(tile, g, L, tid, nthr, tw); break;
        case 8: run_pass<T, 8>(tile, g, L, tid, nthr, tw); break;
        case 9: run_pass<T, 9>(tile, g, L, tid, nthr, tw); break;
        case 10: run_pass<T, 10>(tile, g, L, tid, nthr, tw); break;
        case 11: run_pass<T, 11>(tile, g, L, tid, nthr, tw); break;
        case 12: run_pass<T, 12>(tile, g, L, tid, nthr, tw); break;
        case 13: run_pass<T, 13>(tile, g, L, tid, nthr, tw); break;
        case 14: run_pass<T, 14>(tile, g, L, tid, nthr, tw); break;
        case 15: run_pass<T, 15>(tile, g, L, tid, nthr, tw); break;
        default: run_pass<T, 16>(tile, g, L, tid, nthr, tw); break;
    }
}
template <typename T, bool X17 = false>
__device__ __forceinline__ void fastg_cols_pass(C2<T>* tile, int ncols, int len, int rs, int R, int L, int tid, int nthr, const C2<T>* tw, int tws = 0) {
    if (X17 && R == 17) { fastg_pass_cols<T, X17 ? 17 : 2>(tile, ncols, len, rs, L, tid, nthr, tw, tws); return; }
    switch (R) {
        case 2: fastg_pass_cols<T, 2>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 3: fastg_pass_cols<T, 3>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 4: fastg_pass_cols<T, 4>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 5: fastg_pass_cols<T, 5>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 6: fastg_pass_cols<T, 6>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 7: fastg_pass_cols<T, 7>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 8: fastg_pass_cols<T, 8>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 9: fastg_pass_cols<T, 9>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 10: fastg_pass_cols<T, 10>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 11: fastg_pass_cols<T, 11>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 12: fastg_pass_cols<T, 12>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 13: fastg_pass_cols<T, 13>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 14: fastg_pass_cols<T, 14>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        case 15: fastg_pass_cols<T, 15>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
        default: fastg_pass_cols<T, 16>(tile, ncols, len, rs, L, tid, nthr, tw, tws); break;
    }
}

// MODE 1: power spectrum (real T out), 0: complex spectrum, 2: cross spectrum of two fields (two tiles in LDS, complex out)
// CIN: the complex-input forms (cin, inverse, c2r) -- kernels of their own: their branches cost the real-input forms registers
template <typename T, int MODE, bool CIN>
__global__ void __launch_bounds__(fastg_max_threads<T>(), (sizeof(T) == 4 ? 4 : 3)) fastg_kernel(FastG p) {  // (float64: three waves per SIMD = 168 registers)
    typedef C2<T> CT;
    XRFT_DYN_SMEM(smem_raw);
    CT* tile0 = reinterpret_cast<CT*>(smem_raw);
    constexpr int NF = MODE == 2 ? 2 : 1;  // fields = tiles
    const int tid = threadIdx.x, nthr = blockDim.x, ny = p.ny, nx = p.nx, n = p.n, rs = p.rs;
    const bool packed = p.packed != 0;
    const int ncol = packed ? n + 1 : nx;  // columns of the tile after the x transforms
    // behind the tile: the tables of the plan, staged once per workgroup (the passes' twiddles and the digit-reversal look-ups of the unpack and
    // of the output loop sit on every inner loop's critical path: from global memory each was an L2 round trip), and the plane's wave sums
    unsigned char* tb = smem_raw + (((size_t)NF * ny * rs * sizeof(CT) + 15) & ~(size_t)15);
    CT* twx = reinterpret_cast<CT*>(tb); tb += (size_t)n * sizeof(CT);
    CT* twy = reinterpret_cast<CT*>(tb); tb += (size_t)ny * sizeof(CT);
    CT* twr = reinterpret_cast<CT*>(tb); tb += (size_t)(n + 1) * sizeof(CT);
    double* red = reinterpret_cast<double*>(tb); tb += (size_t)p.nred * sizeof(double);  // [waves][3]; one_d: [row][2]
    T* wys = reinterpret_cast<T*>(tb); tb += (size_t)ny * sizeof(T);  // the windows (with a window; else unused)
    T* wxs = reinterpret_cast<T*>(tb); tb += (size_t)nx * sizeof(T);
    unsigned short* revx = reinterpret_cast<unsigned short*>(tb); tb += (((size_t)n * 2 + 3) & ~(size_t)3);
    unsigned short* revy = reinterpret_cast<unsigned short*>(tb);
    if (p.win_y) {
        for (int k = tid; k < ny; k += nthr) wys[k] = reinterpret_cast<const T*>(p.win_y)[k];
        for (int k = tid; k < nx; k += nthr) wxs[k] = reinterpret_cast<const T*>(p.win_x)[k];
    }
    for (int k = tid; k < n; k += nthr) { twx[k] = reinterpret_cast<const CT*>(p.tw_x)[k]; revx[k] = (unsigned short)p.rev_x[k]; }
    for (int k = tid; k < ny; k += nthr) { twy[k] = reinterpret_cast<const CT*>(p.tw_y)[k]; revy[k] = (unsigned short)p.rev_y[k]; }
    if (packed) for (int k = tid; k <= n; k += nthr) twr[k] = reinterpret_cast<const CT*>(p.tw_r)[k];
    const float inv_n = 1.0f / (float)n, inv_nx = 1.0f / (float)nx;
    TileGeom g{};
    g.n = n; g.T = ny; g.seq_stride = rs; g.pad_shift = 30;
    for (long long slab = blockIdx.x; slab < p.nslabs; slab += gridDim.x) {
      // one_d: the "slab" is a group of ny ROWS of a 1-D transform along x (no y passes; the last group may be short)
      const int nyv = p.one_d ? (int)(p.nrows - slab * ny < (long long)ny ? p.nrows - slab * ny : (long long)ny) : ny;
      const int npk = nyv * n;  // packed samples
      __syncthreads();         // (the previous slab's output loop is done with the tile; the tables are in place)
#pragma unroll 1
      for (int f = 0; f < NF; ++f) {  // (a cross spectrum: field 0 into tile 0, field 1 into tile 1, the same code)
        CT* tile = tile0 + f * (ny * rs);
        const CT* __restrict__ src = (CIN && p.c2r) ? reinterpret_cast<const CT*>(p.in) + (size_t)slab * ny * (n + 1)  // (half spectra)
                                   : (CIN && p.cin) ? reinterpret_cast<const CT*>(p.in) + (size_t)slab * ny * nx  // (complex samples)
                                           : reinterpret_cast<const CT*>(reinterpret_cast<const T*>(f ? p.in_b : p.in) + (size_t)slab * ny * nx);
        // ---- load; the plane's sums on the way (float64 per thread, then the threads in a fixed order)
        double s0 = 0.0, si = 0.0, sj = 0.0;
        const bool plane = p.detrend && !p.one_d;  // (the slab's plane; a 1-D transform fits a line per row below)
        const double ibar = 0.5 * (ny - 1), jbar = 0.5 * (nx - 1);
        if ((CIN && p.c2r)) {  // (nyv (n + 1) complex samples of the half spectrum, conjugated: the inverse transform is conj(FFT(conj .)))
            const CT* __restrict__ wyc = reinterpret_cast<const CT*>(p.ph_y);
            const CT* __restrict__ wxc = reinterpret_cast<const CT*>(p.ph_x);
            const int hw = n + 1, toth = nyv * hw;
            const float inv_hw = 1.0f / (float)hw;
            for (int e = tid; e < toth; e += nthr) {
                const int i = fdiv(e, inv_hw), k = e - i * hw;
                int is = i + p.ishy; if (is >= ny) is -= ny;
                CT z = src[(size_t)is * hw + k];
                if ((CIN && p.ph_in)) z = cmul(z, p.one_d ? wxc[k] : cmul(wyc[is], wxc[k]));
                tile[i * rs + k] = mk<T>(z.re, -z.im);
            }
            if (!p.one_d) {  // the y passes first: every row must be the half spectrum of a real sequence before its c2r step
                __syncthreads();
                int L = ny;
                for (int ps = 0; ps < p.nry; ++ps) {
                    fastg_cols_pass<T>(tile, ncol, ny, rs, p.ry[ps], L, tid, nthr, twy);
                    L /= p.ry[ps];
                    __syncthreads();
                }
            } else {
                __syncthreads();
            }
            // re-pack the rows in place: pairs (k, n - k), k <= n / 2
            const int hp = n / 2 + 1, nb = nyv * hp;
            const float inv_hp = 1.0f / (float)hp;
            for (int w = tid; w < nb; w += nthr) {
                const int i = fdiv(w, inv_hp), k = w - i * hp, km = n - k;
                CT* row = tile + i * rs;
                CT tk = row[k], tm = row[km];
                if (k == 0) { tk.im = (T)0; tm.im = (T)0; }  // (numpy's irfft takes the real parts of the zero-frequency and Nyquist samples of a spectrum that is not a real sequence's)
                // W[k] = tk + conj tm - i (tk - conj tm) w_k,  W[n-k] = tm + conj tk - i (tm - conj tk) w_(n-k)
                const CT d1 = mk<T>(tk.re - tm.re, tk.im + tm.im), d2 = mk<T>(tm.re - tk.re, tm.im + tk.im);
                const CT p1 = cmul(d1, twr[k]), p2 = cmul(d2, twr[km]);
                row[k] = mk<T>(tk.re + tm.re + p1.im, tk.im - tm.im - p1.re);  // (-i (a + i b) = b - i a)
                if (km != k && km != n) row[km] = mk<T>(tm.re + tk.re + p2.im, tm.im - tk.im - p2.re);
            }
        } else if (packed) {
            for (int e = tid; e < npk; e += nthr) {
                const int i = fdiv(e, inv_n), m = e - i * n;
                const CT z = src[e];
                tile[i * rs + m] = z;
                if (plane) {
                    const double u = (double)z.re + (double)z.im;
                    s0 += u;
                    si = fma((double)i - ibar, u, si);
                    sj += ((double)(2 * m) - jbar) * u + (double)z.im;
                }
            }
        } else if ((CIN && p.cin)) {  // (npk = ny nx complex samples)
            const CT* __restrict__ wyc = reinterpret_cast<const CT*>(p.ph_y);
            const CT* __restrict__ wxc = reinterpret_cast<const CT*>(p.ph_x);
            for (int e = tid; e < npk; e += nthr) {
                const int i = fdiv(e, inv_n), m = e - i * n;
                int is = i + p.ishy; if (is >= ny) is -= ny;
                int ms = m + p.ishx; if (ms >= nx) ms -= nx;
                CT z = src[(size_t)is * nx + ms];
                if ((CIN && p.ph_in)) z = cmul(z, p.one_d ? wxc[ms] : cmul(wyc[is], wxc[ms]));
                if ((CIN && p.inv)) z.im = -z.im;
                if (p.win_y) { const T w = (p.one_d ? (T)1 : wys[i]) * wxs[m]; z = mk<T>(z.re * w, z.im * w); }
                tile[i * rs + m] = z;
            }
        } else {  // (npk = ny nx real samples)
            const T* __restrict__ srcr = reinterpret_cast<const T*>(src);
            for (int e = tid; e < npk; e += nthr) {
                const int i = fdiv(e, inv_n), m = e - i * n;
                const T v = srcr[e];
                tile[i * rs + m] = mk<T>(v, (T)0);
                if (plane) {
                    s0 += (double)v;
                    si = fma((double)i - ibar, (double)v, si);
                    sj = fma((double)m - jbar, (double)v, sj);
                }
            }
        }
        if ((CIN && p.cin) || (CIN && p.c2r)) {
            // (complex input: the window rode on the load, there is no detrend)
        } else if (p.one_d && (p.detrend || p.win_y)) {
            // per-row mean / least-squares line (scipy.signal.detrend along x, xrft/detrend.py:54-71): lpr lanes share a row (a power of two <= 64, so a
            // row's lanes sit in one wave), the lanes' float64 sums meet in a fixed shuffle tree; the rows in rounds
            __syncthreads();
            const int lpr = p.lpr, lane = tid & (lpr - 1);
            if (p.detrend) {
                const double sjj = (double)nx * ((double)nx * (double)nx - 1.0) / 12.0;
                for (int r0 = 0; r0 < nyv; r0 += nthr / lpr) {  // (every thread takes every round: the shuffles are wave-wide)
                    const int row = r0 + tid / lpr;
                    double a0 = 0.0, a1 = 0.0;
                    for (int m = lane; m < n && row < nyv; m += lpr) {
                        const CT z = tile[row * rs + m];
                        if (packed) {
                            a0 += (double)z.re + (double)z.im;
                            a1 += ((double)(2 * m) - jbar) * ((double)z.re + (double)z.im) + (double)z.im;
                        } else {
                            a0 += (double)z.re;
                            a1 = fma((double)m - jbar, (double)z.re, a1);
                        }
                    }
                    for (int mm = 1; mm < lpr; mm <<= 1) { a0 += __shfl_xor(a0, mm); a1 += __shfl_xor(a1, mm); }
                    if (lane == 0 && row < nyv) { red[2 * row] = a0 / (double)nx; red[2 * row + 1] = (p.detrend == 2 && nx > 1) ? a1 / sjj : 0.0; }
                }
                __syncthreads();
            }
            const T* wx = wxs;
            for (int e = tid; e < npk; e += nthr) {
                const int i = fdiv(e, inv_n), m = e - i * n;
                CT z = tile[i * rs + m];
                const double c0 = p.detrend ? red[2 * i] : 0.0, c2 = p.detrend ? red[2 * i + 1] : 0.0;
                if (packed) {
                    if (p.detrend) {
                        const double l = c0 + c2 * ((double)(2 * m) - jbar);
                        z = mk<T>((T)((double)z.re - l), (T)((double)z.im - (l + c2)));
                    }
                    if (p.win_y) z = mk<T>(z.re * wx[2 * m], z.im * wx[2 * m + 1]);
                } else {
                    if (p.detrend) z.re = (T)((double)z.re - (c0 + c2 * ((double)m - jbar)));
                    if (p.win_y) z.re *= wx[m];
                }
                tile[i * rs + m] = z;
            }
        } else if (p.detrend || p.win_y) {
            double c0 = 0.0, c1 = 0.0, c2 = 0.0;
            if (p.detrend) {  // wave shuffles, then the waves' sums in wave order
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { s0 += __shfl_xor(s0, m); si += __shfl_xor(si, m); sj += __shfl_xor(sj, m); }
                if ((tid & 63) == 0) { red[(tid >> 6) * 3] = s0; red[(tid >> 6) * 3 + 1] = si; red[(tid >> 6) * 3 + 2] = sj; }
                __syncthreads();
                double t0 = 0.0, t1 = 0.0, t2 = 0.0;
                for (int w = 0; w < (nthr >> 6); ++w) { t0 += red[3 * w]; t1 += red[3 * w + 1]; t2 += red[3 * w + 2]; }
                const double npts = (double)ny * (double)nx;
                c0 = t0 / npts;
                if (p.detrend == 2) {
                    if (ny > 1) c1 = t1 * 12.0 / (npts * ((double)ny * ny - 1.0));
                    if (nx > 1) c2 = t2 * 12.0 / (npts * ((double)nx * nx - 1.0));
                }
            } else {
                __syncthreads();
            }
            const T* wy = p.win_y ? wys : nullptr;
            const T* wx = wxs;
            for (int e = tid; e < npk; e += nthr) {  // (each thread revisits the elements it loaded)
                const int i = fdiv(e, inv_n), m = e - i * n;
                CT z = tile[i * rs + m];
                if (packed) {
                    if (p.detrend) {
                        const double l = c0 + c1 * ((double)i - ibar) + c2 * ((double)(2 * m) - jbar);
                        z = mk<T>((T)((double)z.re - l), (T)((double)z.im - (l + c2)));
                    }
                    if (wy) { const T w = wy[i]; z = mk<T>(z.re * (w * wx[2 * m]), z.im * (w * wx[2 * m + 1])); }
                } else {
                    if (p.detrend) z.re = (T)((double)z.re - (c0 + c1 * ((double)i - ibar) + c2 * ((double)m - jbar)));
                    if (wy) z.re *= wy[i] * wx[m];
                }
                tile[i * rs + m] = z;
            }
        }
        __syncthreads();
        // ---- x: the passes of length n over the rows
        {
            int L = n;
            for (int ps = 0; ps < p.nrx; ++ps) {
                fastg_rows_pass<T>(tile, g, p.rx[ps], L, tid, nthr, twx);
                L /= p.rx[ps];
                __syncthreads();
            }
        }
        // ---- unpack the packed rows in place: pairs (k, n - k), k <= n / 2; X[n] goes to column n
        if (packed && !(CIN && p.c2r)) {
            const int hp = n / 2 + 1, nb = nyv * hp;
            const float inv_hp = 1.0f / (float)hp;
            for (int w = tid; w < nb; w += nthr) {
                const int i = fdiv(w, inv_hp), k = w - i * hp;
                CT* row = tile + i * rs;
                const int pk = (int)revx[k], pm = (int)revx[k == 0 ? 0 : n - k];
                const CT zk = row[pk], zm = row[pm];
                const CT E = mk<T>((T)0.5 * (zk.re + zm.re), (T)0.5 * (zk.im - zm.im));   // (Zk + conj Zm) / 2
                const CT O = mk<T>((T)0.5 * (zk.im + zm.im), (T)0.5 * (zm.re - zk.re));   // -i (Zk - conj Zm) / 2
                const CT t = cmul(twr[k], O);
                if (k == 0) {
                    row[pk] = mk<T>(E.re + t.re, (T)0);    // X[0]  = Re Z0 + Im Z0
                    row[n] = mk<T>(E.re - t.re, (T)0);     // X[n]  = Re Z0 - Im Z0
                } else {
                    row[pk] = mk<T>(E.re + t.re, E.im + t.im);        // X[k]
                    if (pm != pk) row[pm] = mk<T>(E.re - t.re, -(E.im - t.im));  // X[n - k] = conj(E - t)
                }
            }
        }
        __syncthreads();
        // ---- y: the passes of length ny over the n + 1 columns
        if (!(CIN && p.c2r)) {
            int L = ny;
            for (int ps = 0; ps < p.nry; ++ps) {
                fastg_cols_pass<T>(tile, ncol, ny, rs, p.ry[ps], L, tid, nthr, twy);
                L /= p.ry[ps];
                __syncthreads();
            }
        }
      }
      {
        CT* tile = tile0;
        const CT* tileb = tile0 + ny * rs;  // (MODE 2)
        (void)tileb;
        // ---- out, in output order: (orow, ocol) <- F[ky][kx], or conj F[-ky][-kx] for kx > n (a real field's spectrum is Hermitian)
        const int tot = nyv * nx;
        const size_t obase = (size_t)slab * ((size_t)ny * (p.half ? nx / 2 + 1 : nx));
        const T sc = (T)p.scale;
        if (MODE != 0 && p.iso != nullptr) {
            // a bin per wave: lane l adds the samples l, l + 64, ... of the bin's list in float64, the lanes meet in a fixed shuffle tree -- no
            // atomics, the same bits every time; a nan / inf stays in its bin.  A cross spectrum's sums are complex; bit 15 of a position: the sample
            // is the Hermitian twin of the stored one (the conjugate)
            const int lane = tid & 63, nw = nthr >> 6;
            for (int b = tid >> 6; b < p.nbins; b += nw) {
                const unsigned q0 = p.iso_start[b], q1 = p.iso_start[b + 1];
                double acc = 0.0, aci = 0.0;
                for (unsigned q = q0 + (unsigned)lane; q < q1; q += 64u) {
                    const unsigned pq = p.iso_pos[q];
                    if (MODE == 2) {
                        const CT v = cmulc(tile[pq & 0x7fffu], tileb[pq & 0x7fffu]);
                        acc += (double)(v.re * sc);
                        aci += (double)(((pq & 0x8000u) ? -v.im : v.im) * sc);
                    } else {
                        const CT v = tile[pq];
                        acc += (double)((v.re * v.re + v.im * v.im) * sc);
                    }
                }
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { acc += __shfl_xor(acc, m); if (MODE == 2) aci += __shfl_xor(aci, m); }
                if (lane == 0) {
                    if (MODE == 2) { p.iso[((size_t)slab * p.nbins + b) * 2] = acc; p.iso[((size_t)slab * p.nbins + b) * 2 + 1] = aci; }
                    else p.iso[(size_t)slab * p.nbins + b] = acc;
                }
            }
            if (p.out == nullptr) continue;
        }
        if (MODE == 1 && (CIN && p.c2r)) {  // real samples in output order: x[2m] = Re R[m], x[2m+1] = -Im R[m], R[m] at the digit-reversed position of m in its row
#pragma unroll 2
            for (int e = tid; e < tot; e += nthr) {
                const int orow = fdiv(e, inv_nx), ocol = e - orow * nx;
                int i = orow - p.shift_y; if (i < 0) i += ny;
                int j = ocol - p.shift_x; if (j < 0) j += nx;
                const CT r = tile[(int)revy[i] * rs + (int)revx[j >> 1]];
                reinterpret_cast<T*>(p.out)[obase + e] = ((j & 1) ? -r.im : r.re) * sc;
            }
            continue;
        }
        if (p.half) {  // rows of n + 1 samples, kx = 0 .. n as they lie in the tile: no twin, no shift
            const int W = nx / 2 + 1, toth = nyv * W;
            const float inv_w = 1.0f / (float)W;
            for (int e = tid; e < toth; e += nthr) {
                const int ky = fdiv(e, inv_w), kx = e - ky * W;
                const int ps_ = (int)revy[ky] * rs + ((packed && kx == n) ? n : (int)revx[kx]);
                CT v = tile[ps_];
                if (MODE == 2) v = cmulc(v, tileb[ps_]);  // F0 conj(F1)
                if (MODE == 1) {
                    T pw = (v.re * v.re + v.im * v.im) * sc;
                    if (p.realdim2 && kx != 0 && 2 * kx != nx) pw *= (T)2;
                    reinterpret_cast<T*>(p.out)[obase + e] = pw;
                } else {
                    CT o = mk<T>(v.re * sc, v.im * sc);
                    if (MODE == 2 && p.realdim2 && kx != 0 && 2 * kx != nx) o = mk<T>(o.re * (T)2, o.im * (T)2);
                    if (p.ph_on) o = cmul(o, p.one_d ? reinterpret_cast<const CT*>(p.ph_x)[kx] : cmul(reinterpret_cast<const CT*>(p.ph_y)[ky], reinterpret_cast<const CT*>(p.ph_x)[kx]));
                    reinterpret_cast<CT*>(p.out)[obase + e] = o;
                }
            }
            continue;
        }
        for (int e = tid; e < tot; e += nthr) {
            const int orow = fdiv(e, inv_nx), ocol = e - orow * nx;
            int ky = orow - p.shift_y; if (ky < 0) ky += ny;
            int kx = ocol - p.shift_x; if (kx < 0) kx += nx;
            const bool mir = packed && kx > n;
            const int sy = (mir && !p.one_d) ? (ky == 0 ? 0 : ny - ky) : ky, sx = mir ? nx - kx : kx;  // (one_d: a row's twin is in the row)
            const int ps_ = (int)revy[sy] * rs + ((packed && sx == n) ? n : (int)revx[sx]);
            CT v = tile[ps_];
            if (MODE == 2) v = cmulc(v, tileb[ps_]);  // F0 conj(F1); its Hermitian twin is the conjugate, like a spectrum's
            if (MODE == 1) {
                const T pw = (v.re * v.re + v.im * v.im) * sc;
                reinterpret_cast<T*>(p.out)[obase + e] = pw;
            } else {
                CT o = mk<T>(v.re * sc, ((mir != ((CIN && p.inv) != 0)) ? -v.im : v.im) * sc);  // (the twin's conjugate; an inverse transform's conj out)
                if (p.ph_on) o = cmul(o, p.one_d ? reinterpret_cast<const CT*>(p.ph_x)[kx] : cmul(reinterpret_cast<const CT*>(p.ph_y)[ky], reinterpret_cast<const CT*>(p.ph_x)[kx]));
                reinterpret_cast<CT*>(p.out)[obase + e] = o;
            }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// ONE transform axis that is NOT the contiguous one -- xrft.fft / power_spectrum along "time" of a (time, y, x) array, the reference's most
// common call (xrft.py:395-409) -- on ANY smooth length, lengths as data: XRFTHIP_AXIS_Y, [batch][ny][nx] real T, y transformed where it lies.
// (The lengths of fastm.h's table have fastm_yonly_kernel; every other one took the generic column tiles at 0.3-0.5 TB/s.)
//
// A workgroup owns C = 2 G adjacent real columns of one batch element: columns 2g, 2g + 1 are the real and imaginary part of sequence g, the tile is
// [ny][G] complex with the lanes along g (every load, LDS access and store of a wave is contiguous).  Per-column mean / least-squares line
// (xrft/detrend.py:54-71) from float64 sums -- row groups of threads, partial sums in LDS, added in group order -- subtracted and the window multiplied
// in place; the radix passes along y with the radices of the parameter block; the two columns' spectra A[k] = (Z[k] + conj Z[-k]) / 2,
// B[k] = (Z[k] - conj Z[-k]) / 2i are split on the way out, every output row whole: |F|^2 scale or F scale x the true-phase factor, fftshift as a
// rotation of the rows.
struct FastGY {
    const void* in;    // [batch][ny][nx] real T
    void* out;         // [batch][ny][nx] real T (power) or complex T
    long long nunits;  // batch x column blocks
    int ny, nx, nblk;  // nblk = ceil(nx / (2 G))
    int G, lg;         // complex sequences per workgroup (a power of two), its log2
    int nry, ry[kFastGMaxPasses];
    const void* tw_y;  // W_ny^k (complex T)
    const unsigned* rev_y;
    const void* win_y; // T, or null
    const void* ph_y;  // complex mode: combined phase factors by unshifted frequency (complex T)
    int ph_on, detrend, shift_y;
    double scale;
    // a length with a prime factor that has no butterfly (365 = 5 x 73 daily samples of a year, 730, 1460): Bluestein inside the tile -- the passes run
    // on blue_m = 2^a 3^b 5^c >= 2 ny - 1 rows (ry, tw_y belong to blue_m; rev_y is unused: the result comes out in natural order):
    //   x[i] conj(c[i]) zero-padded -> forward passes -> * blue_b -> inverse passes -> * conj(c[k]),  c[k] = exp(i pi k^2 / ny)
    int blue_m;
    const void* blue_c;  // c[k], k < ny (complex T)
    const void* blue_b;  // FFT_m(chirp kernel) / m at the row the forward passes leave each frequency
    int tw_lds;          // the twiddles of the passes are staged in LDS (always, unless a Bluestein tile leaves no room)
    int cin;             // the input is COMPLEX T (the later stages of N-D transforms, xrft.fft of complex data): one sequence per column, G columns per workgroup, no split
    const void* in_b;    // two real fields (cross spectrum / cross phase along the axis, xrft.py:753-874): column c of `in` and of `in_b` are the real and the
    int two, angle;      // imaginary part of sequence c -- G columns per workgroup; the result is F(in) conj F(in_b) (MODE 0), or its phase as real T (angle)
    // inverse transforms (xrft.ifft along the axis, xrft.py:479-646; complex input): conj(FFT(conj(z))); row i of the tile is source row i + ishift_in (mod ny:
    // the ifftshift of an fftshifted spectrum), ph_in: ph_y multiplies the INPUT at its source position (the lag's phase, xrft.py:574-576)
    int inv, ishift_in, ph_in;
    // ny = q p, p ONE prime 17 ... 127 with a smooth p - 1, q smooth and prime to p (365 = 5 x 73 days, 1460 = 20 x 73 six-hourly samples, 366 = 6 x 61):
    // the prime-factor form -- input row i = n1 p + n2 q (mod ny), frequency k = (k mod q, k mod p): a q x p transform with no twiddles between the two
    // dimensions -- with RADER's algorithm along p: the tile is [p][q][G], block j < p - 1 holds n2 = g^-j (g a generator of the units mod p), block
    // p - 1 holds n2 = 0;  ry/tw_y (W_ny) run along q inside every block (the tail passes of a length-ny transform), then along j over the first p - 1
    // blocks: rp forward passes (tw_p = W_(p-1)) -> * rad_b, the sum of the p - 1 samples and the n2 = 0 sample exchanged at frequency 0 -> the inverse
    // passes: X[k1][g^k] at block k, X[k1][0] at block p - 1.  perm_in[i] = row of input i; rev_y[k] = row of frequency k.  ~2.4 transforms of the
    // length where the chirp convolution takes two of 2.1 x the length: (1460, 128, 256) float64 16 -> 59 GFFT/s.
    int half, realdim2;  // real_dim along the axis (xrft.py:400-404, 673-682): only k = 0 .. ny/2 is stored (ny/2 + 1 rows, unshifted); 0 < k < ny/2 counts twice
    int rad_p, rad_q, nrp, rp[kFastGMaxPasses];
    const void* tw_p;            // W_(p-1)^k (complex T)
    const void* rad_b;           // FFT_(p-1)(W_p^(g^m)) / (p - 1) at the row the forward passes leave each frequency (complex T)
    const unsigned* perm_in;
};

// MODE 1: power spectrum (real T out), 0: complex spectrum; FORM 1: the Bluestein form (its inverse passes cost the plain form 25 registers: a kernel of its own),
// FORM 2: the prime-factor form with Rader's algorithm along the prime; FORM 3: the same along the CONTIGUOUS axis -- [rows][ny samples], a sequence = two rows (1-D spectra of
// (station, time) series on 365 / 730 / 1460 samples): the lanes of the load and of the output loop run along the samples, the tile's rows are one sequence wider (odd stride)
template <typename T, int MODE, int FORM>
__global__ void __launch_bounds__(256, (sizeof(T) == 4 ? 3 : 2)) fastgy_kernel(FastGY p) {  // (two waves per SIMD at least: the float64 Rader forms wanted 256 + registers -- one wave, 76 -> 46 GFFT/s)
    constexpr bool BLUE = FORM == 1, RADER = FORM >= 2, ROWS = FORM == 3;
    typedef C2<T> CT;
    XRFT_DYN_SMEM(smem_raw);
    CT* tile = reinterpret_cast<CT*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x, ny = p.ny, nx = p.nx, G = p.G, lg = p.lg;
    int C = 2 * G;  // columns of a unit
    const int GS = ROWS ? G + 1 : G;        // sequences of a tile row incl. padding (ROWS: an odd stride -- consecutive lanes hold consecutive samples of ONE sequence)
    const int nrow = BLUE ? p.blue_m : ny;  // rows of the tile = length of the passes
    unsigned char* tb = smem_raw + (((size_t)nrow * GS * sizeof(CT) + 15) & ~(size_t)15);
    double* part = reinterpret_cast<double*>(tb); tb += (size_t)nthr * 4 * sizeof(double);  // [row group][g][4]
    double* coef = reinterpret_cast<double*>(tb); tb += (size_t)G * 4 * sizeof(double);     // [g][mean re, slope re, mean im, slope im]
    const bool twin_lds = !BLUE || p.tw_lds;  // (the plain form always)
    CT* twl = reinterpret_cast<CT*>(tb); tb += twin_lds ? (size_t)(RADER ? p.rad_q : nrow) * sizeof(CT) : 0;  // (RADER: W_q alone, W_ny^(k p))
    T* wys = reinterpret_cast<T*>(tb); tb += (size_t)ny * sizeof(T);
    unsigned short* revy = reinterpret_cast<unsigned short*>(tb); tb += (size_t)ny * 2;
    tb += (size_t)(-(reinterpret_cast<intptr_t>(tb) - reinterpret_cast<intptr_t>(smem_raw))) & 15;  // (what follows holds complex values: 16-byte aligned)
    unsigned short* pin = reinterpret_cast<unsigned short*>(tb); tb += RADER ? (((size_t)ny * 2 + 15) & ~(size_t)15) : 0;
    CT* twp = reinterpret_cast<CT*>(tb);  // (RADER: W_(p-1))
    if (RADER) {
        for (int k = tid; k < ny; k += nthr) pin[k] = (unsigned short)p.perm_in[k];
        for (int k = tid; k < p.rad_p - 1; k += nthr) twp[k] = reinterpret_cast<const CT*>(p.tw_p)[k];
    }
    const CT* twy = twl;
    if (BLUE && !p.tw_lds) twy = reinterpret_cast<const CT*>(p.tw_y);
    if (RADER) { for (int k = tid; k < p.rad_q; k += nthr) twl[k] = reinterpret_cast<const CT*>(p.tw_y)[k * p.rad_p]; }
    else if (twin_lds) for (int k = tid; k < nrow; k += nthr) twl[k] = reinterpret_cast<const CT*>(p.tw_y)[k];
    for (int k = tid; k < ny; k += nthr) {
        if (!BLUE) revy[k] = (unsigned short)p.rev_y[k];
        if (p.win_y) wys[k] = reinterpret_cast<const T*>(p.win_y)[k];
    }
    const int RQ = nthr >> lg;
    // (lane along the sequences, row group; ROWS: lanes along the samples of a sequence -- they are contiguous in memory)
    const int g = ROWS ? tid / RQ : tid & (G - 1), rq = ROWS ? tid - g * RQ : tid >> lg;
    const bool one_col = p.cin || p.two;                            // a sequence is ONE column (complex input, or a column of each of two fields)
    const int cpg = one_col ? 1 : 2, lc = one_col ? lg : lg + 1;    // columns per sequence; log2 of the columns of a unit
    C = G * cpg;
    const T sc = (T)p.scale;
    const double ibar = 0.5 * (ny - 1);
    for (long long unit = blockIdx.x; unit < p.nunits; unit += gridDim.x) {
        const long long b = unit / p.nblk;
        const int c0 = (int)(unit - b * p.nblk) * C;
        const bool live = c0 + cpg * g < nx;  // (real input: nx is even, a pair of columns is whole or absent)
        const bool live1 = !ROWS || one_col || c0 + cpg * g + 1 < nx;  // (ROWS: an odd number of rows leaves the last sequence one row)
        // ROWS: [nx rows][ny samples], the transform axis is the contiguous one; "column" c of the unit is row c0 + c
        const T* __restrict__ src = reinterpret_cast<const T*>(p.in) + (ROWS ? (size_t)(c0 + cpg * g) * ny : (size_t)b * ny * nx + c0 + cpg * g) * (p.cin ? 2 : 1);
        const size_t rowstep = ROWS ? (size_t)(p.cin ? 2 : 1) : (size_t)nx * (p.cin ? 2 : 1);  // (in T elements)
        __syncthreads();  // (the previous unit's output loop is done with the tile; the tables are in place)
        // ---- load (rows rq, rq + RQ, ... of sequence g); without a detrend the window rides along
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = rq; i < ny; i += RQ) {
            CT z = mk<T>((T)0, (T)0);
            int is = i + p.ishift_in; if (is >= ny) is -= ny;  // (source row)
            if (live) {
                if (p.two) z = mk<T>(src[(size_t)is * rowstep], (reinterpret_cast<const T*>(p.in_b) + (ROWS ? (size_t)(c0 + g) * ny : (size_t)b * ny * nx + c0 + g))[(size_t)is * rowstep]);
                else if (ROWS && !p.cin) z = mk<T>(src[is], live1 ? src[(size_t)ny + is] : (T)0);  // (the two rows of the sequence)
                else z = *reinterpret_cast<const CT*>(src + (size_t)is * rowstep);
            }
            if (p.ph_in) z = cmul(z, reinterpret_cast<const CT*>(p.ph_y)[is]);
            if (p.inv) z.im = -z.im;
            if (p.detrend) {
                const double ri = (double)i - ibar;
                s[0] += (double)z.re; s[2] += (double)z.im;
                s[1] = fma(ri, (double)z.re, s[1]); s[3] = fma(ri, (double)z.im, s[3]);
            } else if (p.win_y) {
                const T w = reinterpret_cast<const T*>(p.win_y)[i];
                z = mk<T>(z.re * w, z.im * w);
            }
            tile[(RADER ? (int)pin[i] : i) * GS + g] = z;
        }
        if (p.detrend) {
#pragma unroll
            for (int c = 0; c < 4; ++c) part[(rq * G + g) * 4 + c] = s[c];
            __syncthreads();
            if (tid < G) {  // the row groups' sums in group order; mean and slope of the two columns of sequence tid
                double a[4] = {0.0, 0.0, 0.0, 0.0};
                for (int r = 0; r < RQ; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] += part[(r * G + tid) * 4 + c];
                const double sii = (double)ny * ((double)ny * (double)ny - 1.0) / 12.0;
                const bool lin = p.detrend == 2 && ny > 1;
                coef[tid * 4 + 0] = a[0] / (double)ny; coef[tid * 4 + 1] = lin ? a[1] / sii : 0.0;
                coef[tid * 4 + 2] = a[2] / (double)ny; coef[tid * 4 + 3] = lin ? a[3] / sii : 0.0;
            }
            __syncthreads();
            const double m0 = coef[g * 4], b0 = coef[g * 4 + 1], m1 = coef[g * 4 + 2], b1 = coef[g * 4 + 3];
            for (int i = rq; i < ny; i += RQ) {  // (each thread revisits the elements it loaded)
                const int slot = (RADER ? (int)pin[i] : i) * GS + g;
                CT z = tile[slot];
                const double ri = (double)i - ibar;
                z = mk<T>((T)((double)z.re - fma(b0, ri, m0)), (T)((double)z.im - fma(b1, ri, m1)));
                if (p.win_y) { const T w = wys[i]; z = mk<T>(z.re * w, z.im * w); }
                tile[slot] = z;
            }
        }
        if (BLUE) {  // x conj(c), zero padding up to blue_m rows
            const CT* __restrict__ ch = reinterpret_cast<const CT*>(p.blue_c);
            for (int i = rq; i < nrow; i += RQ) tile[i * G + g] = i < ny ? cmulc(tile[i * G + g], ch[i]) : mk<T>((T)0, (T)0);
        }
        __syncthreads();
        // ---- the passes of length ny (blue_m) over the G sequences (lanes along the sequences); RADER: along q inside every block of q rows
        {
            int L = RADER ? p.rad_q : nrow;
            for (int ps = 0; ps < p.nry; ++ps) {
                fastg_cols_pass<T>(tile, G, nrow, GS, p.ry[ps], L, tid, nthr, twy, RADER ? p.rad_q / L : 0);
                L /= p.ry[ps];
                __syncthreads();
            }
        }
        if (BLUE) {  // circular convolution with the chirp: * B, the inverse passes in reverse order, * conj(c): Z[k] at row k
            const CT* __restrict__ bh = reinterpret_cast<const CT*>(p.blue_b);
            const CT* __restrict__ ch = reinterpret_cast<const CT*>(p.blue_c);
            for (int i = rq; i < nrow; i += RQ) tile[i * G + g] = cmul(tile[i * G + g], bh[i]);
            __syncthreads();
            int Li = 1;
            for (int ip = p.nry - 1; ip >= 0; --ip) {
                Li *= p.ry[ip];
                fastg_cols_pass_inv<T>(tile, G, nrow, G, p.ry[ip], Li, tid, nthr, twy);
                __syncthreads();
            }
            for (int i = rq; i < ny; i += RQ) tile[i * G + g] = cmulc(tile[i * G + g], ch[i]);
            __syncthreads();
        }
        if (RADER) {  // along the prime: a cyclic convolution of the p - 1 blocks with n2 != 0 (q G sequences side by side, element stride q G)
            const int P1 = p.rad_p - 1, qg = p.rad_q * GS;
            const CT* __restrict__ bh = reinterpret_cast<const CT*>(p.rad_b);
            int L = P1;
            for (int ps = 0; ps + 1 < p.nrp; ++ps) {
                fastg_cols_pass<T, sizeof(T) == 4>(tile, qg, P1, qg, p.rp[ps], L, tid, nthr, twp);
                L /= p.rp[ps];
                __syncthreads();
            }
            // the last forward pass, * the transformed kernel, the frequency-0 exchanges and the first inverse pass in one (frequency 0 of the convolution = the sum of the samples with
            // n2 != 0: X[.][0] = x0 + sum, and x0 joins every other frequency)
            fastg_cols_pass_inv_first<T, sizeof(T) == 4>(tile, qg, P1, qg, p.rp[p.nrp - 1], tid, nthr, bh, P1 * qg);
            __syncthreads();
            int Li = p.rp[p.nrp - 1];
            for (int ip = p.nrp - 2; ip >= 0; --ip) {
                Li *= p.rp[ip];
                fastg_cols_pass_inv<T, sizeof(T) == 4>(tile, qg, P1, qg, p.rp[ip], Li, tid, nthr, twp);
                __syncthreads();
            }
        }
        // ---- out: row orow of the C columns = frequency k of the two spectra packed in every sequence
        const int nyo = p.half ? (ny >> 1) + 1 : ny;  // rows of the result
        const int tot = nyo << lc;
        const float inv_ny = 1.0f / (float)nyo;
        for (int e = tid; e < tot; e += nthr) {
            int orow, c;
            if (ROWS) { c = fdiv(e, inv_ny); orow = e - c * nyo; }  // (lanes along the frequencies of one row: contiguous stores)
            else { orow = e >> lc; c = e & (C - 1); }
            const int col = c0 + c;
            if (col >= nx) continue;
            int k = orow - p.shift_y; if (k < 0) k += ny;
            const int km = k == 0 ? 0 : ny - k;
            CT v;
            if (p.cin) {
                v = tile[(BLUE ? k : (int)revy[k]) * GS + c];
                if (p.inv) v.im = -v.im;
            } else if (p.two) {  // A = (Zk + conj Zm) / 2, B = (Zk - conj Zm) / 2i: A conj(B)
                const CT zk = tile[(BLUE ? k : (int)revy[k]) * GS + c], zm = tile[(BLUE ? km : (int)revy[km]) * GS + c];
                v = cmulc(mk<T>((T)0.5 * (zk.re + zm.re), (T)0.5 * (zk.im - zm.im)), mk<T>((T)0.5 * (zk.im + zm.im), (T)0.5 * (zm.re - zk.re)));
            } else {
                const CT zk = tile[(BLUE ? k : (int)revy[k]) * GS + (c >> 1)], zm = tile[(BLUE ? km : (int)revy[km]) * GS + (c >> 1)];
                v = (c & 1) ? mk<T>((T)0.5 * (zk.im + zm.im), (T)0.5 * (zm.re - zk.re))   // (Zk - conj Zm) / 2i
                            : mk<T>((T)0.5 * (zk.re + zm.re), (T)0.5 * (zk.im - zm.im));  // (Zk + conj Zm) / 2
            }
            const size_t o = ROWS ? (size_t)col * nyo + orow : ((size_t)b * nyo + orow) * nx + col;
            const T sck = (p.realdim2 && k != 0 && 2 * k != ny) ? sc + sc : sc;
            if (MODE == 1) {
                reinterpret_cast<T*>(p.out)[o] = (v.re * v.re + v.im * v.im) * sck;
            } else {
                CT w = mk<T>(v.re * sck, v.im * sck);
                if (p.ph_on) w = cmul(w, reinterpret_cast<const CT*>(p.ph_y)[k]);
                if (p.angle) reinterpret_cast<T*>(p.out)[o] = (T)atan2((double)w.im, (double)w.re);  // cross phase (xrft.py:838-874)
                else reinterpret_cast<CT*>(p.out)[o] = w;
            }
        }
    }
}

}  // namespace xrft

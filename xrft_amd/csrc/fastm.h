// fastm.h -- transforms whose length is a product of three small radices, held in LDS: the float64 / float32 kernels beside the
// register-resident float32 power-of-two ones of fasty.h.
//
// (1) The two-pass "y first" pipeline (fasty.h) for real slabs whose two lengths are in the table -- the regular lat/lon grids
//     180 ... 1440 in both precisions, 256 / 512 / 1024 in float64.  BASELINE.json configs[4] is power_spectrum of
//     (64, 1440, 720) float64 slabs with a linear detrend and a Hann window.
//       pass 1  fastm_cols_kernel   FFT along y of the real columns (window fused), half spectra ky = 0..ny/2, exact column sums
//       [fit]   fastm_fit_kernel    plane from the per-column sums (xrft/detrend.py:100-113)
//       pass 2  fastm_rows_kernel   trend added back in the spectral domain, FFT along x of rows ky = 0..ny/2, result rows ky AND -ky;
//                                   power / complex / cross spectrum / cross phase, full or half (real_dim) rows, radial sums
//     (xrft.power_spectrum / fft / cross_spectrum / cross_phase / isotropic_*, reference xrft/xrft.py:307-476, 685-1187.)
// (2) ONE transform axis in ONE pass: fastm_yonly_kernel (an axis that is not the contiguous one: XRFTHIP_AXIS_Y) and
//     fastm_xonly_kernel (short contiguous rows, packed in pairs), real or complex input, one or two fields, with the
//     reference's per-sequence detrend done inside the workgroup.
//
// Same plan as fasty.h -- the last pass owns whole result rows, fftshift is a rotation, the Hermitian mirror a reversed read
// of a row that is in LDS, detrending costs no pass over the data -- but the transforms are not register-resident: the first
// pass runs on operands loaded straight from global memory (every load of a thread in flight at once), then two more in-place
// decimation-in-frequency passes through LDS with compile-time radices R0 x R1 x R2 (one butterfly per thread and pass: no
// loops, no index tables), the last of which leaves the spectrum in natural order, and the result streams out of LDS.
// The generic tile kernel (tile_fft.h) does the same with run-time geometry; at (1440, 720) float64 it ran 2.5x above the
// memory floor of both passes with its waves parked half of the time (profiles/r02_pmc_generic_c5_summary.txt).
//
// Intermediate W2 (complex T): [slab][ky / RK][x / CW][ky % RK][CW], CW = 2 G columns of one pass-1 workgroup, RK rows per
// 128-byte line; pass 1 writes whole lines, pass 2 reads RPU (a multiple of RK) consecutive ky = one contiguous block.
#pragma once
#include "aux_kernels.h"
#include "fasty.h"  // ilog2c
#include "tile_fft.h"

namespace xrft {

template <int N> struct MRad { static constexpr int R0 = 0, R1 = 0, R2 = 0; };
#define XRFT_MRAD(NN, A, B, C) \
    template <> struct MRad<NN> { static constexpr int R0 = A, R1 = B, R2 = C; static_assert(A * B * C == NN, "radices"); }
XRFT_MRAD(180, 5, 6, 6);
XRFT_MRAD(240, 5, 6, 8);
XRFT_MRAD(360, 6, 6, 10);
XRFT_MRAD(480, 6, 8, 10);
XRFT_MRAD(720, 8, 9, 10);
XRFT_MRAD(960, 8, 10, 12);
XRFT_MRAD(1440, 10, 12, 12);
XRFT_MRAD(900, 9, 10, 10);    // (round 3: lengths the generic tile kernels ran 3-5x slower -- 0.1-degree grids 3600 x 1800, and 1500 / 2000 / 3000)
XRFT_MRAD(1500, 10, 10, 15);
XRFT_MRAD(1800, 10, 12, 15);
XRFT_MRAD(2000, 10, 10, 20);
XRFT_MRAD(3000, 10, 15, 20);
XRFT_MRAD(3600, 15, 15, 16);
XRFT_MRAD(320, 5, 8, 8);      // (Gaussian grids N80 ... N640: 320 x 160, 640 x 320, 1280 x 640, 2560 x 1280; 1/3, 1/6, 1/8, 1/12-degree grids: 1080 x 540, 2160 x 1080, 2880 x 1440, 4320 x 2160)
XRFT_MRAD(540, 6, 9, 10);
XRFT_MRAD(640, 8, 8, 10);
XRFT_MRAD(1080, 9, 10, 12);
XRFT_MRAD(1280, 8, 10, 16);
XRFT_MRAD(2160, 12, 12, 15);
XRFT_MRAD(2560, 10, 16, 16);
XRFT_MRAD(2880, 12, 15, 16);
XRFT_MRAD(4320, 15, 16, 18);
XRFT_MRAD(192, 4, 6, 8);      // (3 x 2^k: the Gaussian grids T63 ... T511 and their halves -- 192 x 96, 384 x 192, 768 x 384, 1536 x 768; TL799: 1600 x 800; HD / 4K frames: 1920 x 1080, 3840 x 2160)
XRFT_MRAD(384, 6, 8, 8);
XRFT_MRAD(768, 8, 8, 12);
XRFT_MRAD(1536, 8, 12, 16);
XRFT_MRAD(1600, 10, 10, 16);
XRFT_MRAD(1920, 10, 12, 16);
XRFT_MRAD(2400, 10, 12, 20);
XRFT_MRAD(3072, 12, 16, 16);
XRFT_MRAD(3840, 15, 16, 16);
XRFT_MRAD(256, 4, 8, 8);   // (powers of two: float64 only -- float32 has the register-resident kernels of fasty.h)
XRFT_MRAD(512, 8, 8, 8);
XRFT_MRAD(1024, 8, 8, 16);
XRFT_MRAD(2048, 8, 16, 16);  // (one transform axis only)
XRFT_MRAD(4096, 16, 16, 16); // (one transform axis only)
// lengths of time-like axes, one transform axis only (fastm_yonly_kernel)
XRFT_MRAD(100, 4, 5, 5);
XRFT_MRAD(128, 4, 4, 8);
XRFT_MRAD(200, 5, 5, 8);
XRFT_MRAD(400, 5, 8, 10);
XRFT_MRAD(500, 5, 10, 10);
XRFT_MRAD(600, 6, 10, 10);
XRFT_MRAD(800, 8, 10, 10);
XRFT_MRAD(1000, 10, 10, 10);
XRFT_MRAD(1200, 10, 10, 12);
#undef XRFT_MRAD
// the lengths the host dispatches on: X(N) for every entry
#define XRFT_M_LATLON(X) X(180) X(192) X(240) X(320) X(360) X(384) X(480) X(500) X(540) X(640) X(720) X(768) X(900) X(960) X(1000) X(1080) X(1200) X(1280) X(1440) X(1500) X(1800) X(1920) X(2000) X(2160)  /* both axes of a slab: the lat/lon and Gaussian-grid lengths + 500, 1000, 1200, 1500, 2000 */
#define XRFT_M_F32ONLY(X) X(1536) X(1600) X(2400) X(2560) X(2880) X(3072) X(3840)  /* float32 only: a pair of complex128 sequences of this length does not fit the LDS beside a second workgroup */
#define XRFT_M_F32_1AX(X) X(3000) X(3600) X(4320)  /* float32, ONE transform axis only: as both axes of a slab these lengths left the table in round 5 -- the lengths-as-data pipeline (fastn.h) runs them within 10 % (profiles/r05_fastn_vs_table.txt: 150 / 162, 162 / 176, 137 / 131 GFFT/s) */
#define XRFT_M_POW2(X) X(256) X(512) X(1024)
#define XRFT_M_WIDE32(X) X(1800) X(2000) X(2160)  /* float32: pass 1 also exists with four sequences per workgroup (832 threads at most) */
#define XRFT_M_YONLY(X) X(100) X(128) X(200) X(400) X(600) X(800)

constexpr int mr_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// LDS geometry of one N-point sequence.  Positions during the passes: pd(i) = i + i / PDQ (the last pass reads runs of R2
// with a lane stride of R2: padded to an odd stride), natural-order result: pn(k) = k + k / PNQ (the last pass writes with a
// lane stride of R0).  STR = 4 (mod 8) elements: the two rows / two sequences that eight lanes touch land on disjoint banks.
// (GOV > 0 overrides the number of sequences per workgroup: the y-only kernel at 4096 / 2048 points needs two -- four columns --
// whatever the LDS they take)
template <typename T, int N, int GOV = 0> struct MGeom {
    typedef MRad<N> R;
    static_assert(R::R0 > 0, "length not in the table");
    static constexpr int R0 = R::R0, R1 = R::R1, R2 = R::R2;
    static constexpr int M0 = N / R0;                  // = R1 R2
    static constexpr int B0 = N / R0, B1 = N / R1, B2 = N / R2;  // butterflies per sequence and pass
    static constexpr int BMAX = mr_max3(B0, B1, B2);
    static constexpr int PDQ = (R2 % 2 == 0) ? R2 : (1 << 30);
    static constexpr int PNQ = (R0 % 2 == 0) ? R0 : (1 << 30);
    static constexpr int PQ = PDQ < PNQ ? PDQ : PNQ;
    static constexpr int STR = ((N + N / PQ + 3) / 8) * 8 + 4;  // the smallest s >= N + N / PQ with s = 4 (mod 8)
    static constexpr size_t CS = 2 * sizeof(T);
    // sequences per workgroup: as many (a power of two) as keep three workgroups on a CU
#ifndef XRFT_M_LDSCAP
#define XRFT_M_LDSCAP (N >= 1800 ? 78 * 1024 : 52 * 1024)  /* three workgroups per CU; two for the long sequences (a pair of 2000-point complex128 sequences is 70 KB) */
#endif
    // (at most 4: 8 columns per workgroup divide every length of the table; and at most XRFT_M_MAXTHR threads: one butterfly per thread and pass.
    // 640 until the lengths 1800 ... 2160 joined: four float32 sequences of them fit the LDS, and 16-byte row segments -- two sequences --
    // load at half the rate of 32-byte ones: (64, 2000, 2000) float32 pass 1 12.5 us per slab for 6 us of bytes)
#ifndef XRFT_M_MAXTHR
#define XRFT_M_MAXTHR 640
#endif
    static constexpr int G0 = ((size_t)4 * STR * CS <= XRFT_M_LDSCAP && 4 * BMAX <= XRFT_M_MAXTHR) ? 4 : (size_t)2 * STR * CS <= XRFT_M_LDSCAP ? 2 : 1;
    static constexpr int G = GOV > 0 ? GOV : G0;
    static constexpr int THR = ((G * BMAX + 63) / 64) * 64;
    static constexpr size_t LDS_ROWS = ((size_t)G * STR + M0) * CS;                                    // sequences + pass-1 twiddles
    static constexpr size_t LDS = LDS_ROWS + (size_t)(THR / 64) * G * 4 * sizeof(double);              // + pass 1's partial column sums
    static constexpr int WGS = (int)((160 * 1024) / LDS) < 1 ? 1 : (int)((160 * 1024) / LDS);
    static constexpr int WPS = (WGS * (THR / 64) + 3) / 4 > 5 ? 5 : (WGS * (THR / 64) + 3) / 4;  // waves per SIMD the launch bounds ask for (>= 102 VGPRs)
    // pass 2, one field: sequences (= rows) per workgroup, its threads, LDS and launch bound.  Half of pass 1's (but whole lines of
    // W2: >= 2 rows): its reads and writes are contiguous whatever the count, and six small workgroups per CU interleave their
    // load / transform / store phases better than three (C5 row pass 4.22 -> 3.74 us per slab)
#ifndef XRFT_M_ROWS_SHIFT
#define XRFT_M_ROWS_SHIFT 1
#endif
    static constexpr int GR1 = (G >> XRFT_M_ROWS_SHIFT) < 2 ? (G < 2 ? G : 2) : (G >> XRFT_M_ROWS_SHIFT);
    template <int GG> struct Rows {
        static constexpr int THR = ((GG * BMAX + 63) / 64) * 64;
        static constexpr size_t LDS = ((size_t)GG * STR + M0) * CS;
        static constexpr int WGS = (int)((160 * 1024) / LDS) < 1 ? 1 : ((int)((160 * 1024) / LDS) > 8 ? 8 : (int)((160 * 1024) / LDS));
        static constexpr int WPS = (WGS * (THR / 64) + 3) / 4 > 5 ? 5 : (WGS * (THR / 64) + 3) / 4;
    };
    __device__ static __forceinline__ int pd(int i) { return i + i / PDQ; }
    __device__ static __forceinline__ int pn(int k) { return k + k / PNQ; }
};

struct FastM {
    const void* in;      // [slab][ny][nx] real T
    const void* in_b;    // one-axis cross spectra / phases: the second field (same layout)
    void* w2;            // intermediate (see above)
    const void* w2b;     // cross spectra: field 1's intermediate (pass 2 only)
    void* out;           // [slab][ny][nx]: T (power, phase) or complex T
    const void* tw_x;    // W_nx^k, k < nx  (complex T)
    const void* tw_y;
    const void* win_y;   // T, never null
    const void* win_x;
    double* colfit;      // [slab][nx][4]: sum d, sum (i - ibar) d, and the line pass 1 subtracted (0, 0 here)
    const void* corr;    // [slab][nx] complex T: wx[x] * (subtracted line - plane fit) as (offset at ibar, slope)
    const void* corr_b;  // ... of field 1
    const void* ph_y;    // complex modes: combined phase factors per unshifted frequency (complex T), never null
    const void* ph_x;
    const void* what0;   // FFT_y(wy)[ky], ky < nrow_pad (complex T)
    const void* what1;   // FFT_y(wy (i - ibar))[ky]
    const int* binmap;   // radial sums fused into pass 2 (ISO): bin of (ky, kx), unshifted indices, [ny][nx]; < 0 = none
    double* iso_part;    // [slab][row workgroup][nbins (x2 complex)]: per-workgroup sums, reduced in order by iso_reduce_kernel
    const unsigned short* tfirst;  // radial bin map: [ky <= ny/2][nbins + 1], the smallest |kx| <= nx/2 of a row whose bin is >= b (null: any map, the atomic tables)
    const unsigned* twin;          // ... [unit of rows] first bin | (last bin + 1) << 16 its rows reach
    int nbins, iso_ncopy;
    int cin;             // one-axis kernels: the input is COMPLEX (one sequence per column / row, no packing): the later stages of N-D transforms
    int angle;           // one-axis two-field kernels: store the cross PHASE (float) instead of the cross spectrum (xrft.py:838-874)
    int half;            // real_dim: only kx = 0..nx/2 is stored, rows of nx/2 + 1 samples, unshifted along x (xrft.py:400-404)
    int realdim2;        // ... and 0 < kx < nx/2 counts twice (xrft.py:673-682)
    int ph_on;
    int c2r;             // x-only kernel: irfft along the contiguous axis (XRFTHIP_C2R_X) -- rows of n/2 + 1 complex values in, n real samples out, TWO rows per transform (C = A + i B)
    int inv, ishift_in, ph_in;  // x-only kernel, complex input: an inverse transform (xrft.ifft along the contiguous axis, xrft.py:479-646) = conj(FFT(conj z)); source sample
                                // (x + ishift_in) mod n feeds position x (the ifftshift of an fftshifted spectrum); ph_x multiplies the INPUT at its source position (the lag's phase, :574-576)
    int ny, nx, nrow_pad;
    int l_cw, l_rk;      // log2 of CW and RK
    int detrend, nslab, nunits;
    int shift_y, shift_x;
    double scale;
};

// profiling builds (scripts/build_ablate_m.sh, -DXRFT_MDBG=bits; 0 in the product): 1 = the intermediate is written with plain
// stores, 2 = the result too, 4 = no transforms (passes skipped), 8 = no stores, 16 = no loads of the sequences
#ifndef XRFT_MDBG
#define XRFT_MDBG 0
#endif

// 16-byte store that bypasses the caches' retention (the next reader is another kernel, a whole group of slabs later)
template <typename T, bool PLAIN = false> __device__ __forceinline__ void mr_store16_nt(void* dst, const void* src16) {
#ifdef XRFT_EMULATE
    memcpy(dst, src16, 16);
#else
    if (PLAIN) { *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src16); return; }
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(*reinterpret_cast<const v4f*>(src16), reinterpret_cast<v4f*>(dst));
#endif
}

// one complex value (16 bytes float64, 8 bytes float32), non-temporal
template <typename T, bool PLAIN = false> __device__ __forceinline__ void mr_store_ct_nt(void* dst, const C2<T>& v) {
    if (sizeof(T) == 8) { mr_store16_nt<T, PLAIN>(dst, &v); return; }
#ifdef XRFT_EMULATE
    memcpy(dst, &v, sizeof(v));
#else
    if (PLAIN) { *reinterpret_cast<C2<T>*>(dst) = v; return; }
    typedef float v2f __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(*reinterpret_cast<const v2f*>(&v), reinterpret_cast<v2f*>(dst));
#endif
}

// First pass of sequence t, butterfly j, on operands the thread already holds (a[q] = x[j + q M0], straight from global
// memory: the sequences are never staged): y_k[j] W_N^(j k) goes to k M0 + j.  w0 = W_N^j (loaded by the caller long before).
template <typename T, int N>
__device__ __forceinline__ void mr_pass0(C2<T>* a, C2<T>* s, int j, C2<T> w0) {
    typedef MGeom<T, N> M;
    dft_r<T, M::R0>(a);
    C2<T> w = w0;
#pragma unroll
    for (int k = 1; k < M::R0; ++k) {
        a[k] = cmul(a[k], w);
        if (k + 1 < M::R0) w = cmul(w, w0);
    }
#pragma unroll
    for (int k = 0; k < M::R0; ++k) s[M::pd(j + k * M::M0)] = a[k];
}

// The other two passes over the G sequences of a workgroup (sequence t at lds + t STR, positions pd(i); the result is left in
// natural order at pn(k)).  tw1[j R1 + k] = W_(N/R0)^(j k).  Starts and ends with a barrier.
template <typename T, int N, int G, int THR>
__device__ __forceinline__ void mr_fft_tail(C2<T>* lds, int tid, const C2<T>* tw1) {
    typedef MGeom<T, N> M;
    constexpr int R0 = M::R0, R1 = M::R1, R2 = M::R2, M0 = M::M0, STR = M::STR;
    __syncthreads();
    if (tid < G * M::B1) {  // pass 1: blocks of M0 = R1 R2, butterflies over stride R2
        const int t = tid / M::B1, gg = tid % M::B1, blk = gg / R2, j = gg % R2, base = blk * M0 + j;
        C2<T>* s = lds + t * STR;
        C2<T> a[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) a[q] = s[M::pd(base + q * R2)];
        dft_r<T, R1>(a);
#pragma unroll
        for (int k = 1; k < R1; ++k) a[k] = cmul(a[k], tw1[j * R1 + k]);
#pragma unroll
        for (int k = 0; k < R1; ++k) s[M::pd(base + k * R2)] = a[k];
    }
    __syncthreads();
    {   // pass 2: runs of R2; frequency k0 + R0 (k1 + R1 k2) of run k0 R1 + k1 goes to its natural slot (after everyone has read)
        const bool on = tid < G * M::B2;
        const int t = tid / M::B2, blk = tid % M::B2, k0 = blk / R1, k1 = blk % R1;
        C2<T>* s = lds + (on ? t : 0) * STR;
        C2<T> a[R2];
        if (on) {
#pragma unroll
            for (int q = 0; q < R2; ++q) a[q] = s[M::pd(blk * R2 + q)];
        }
        __syncthreads();
        if (on) {
            dft_r<T, R2>(a);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) s[M::pn(k0 + R0 * (k1 + R1 * k2))] = a[k2];
        }
    }
    __syncthreads();
}

template <typename T, int N> __device__ __forceinline__ void mr_fill_tw1(C2<T>* tw1, const C2<T>* __restrict__ tw, int tid, int nthreads) {
    typedef MGeom<T, N> M;
    for (int e = tid; e < M::M0; e += nthreads) {
        const int j = e / M::R1, k = e % M::R1;
        tw1[e] = tw[M::R0 * j * k];  // W_(N/R0)^(j k) = W_N^(R0 j k), j k < M0
    }
}

// ------------------------------------------------------------------------------------------------
// pass 1: a workgroup owns CW = 2 G adjacent real columns of one slab (lane order (row, g), g fastest: the lanes of a row read
// 16 G contiguous bytes); columns 2g, 2g+1 are the real and imaginary part of sequence g.     window: xrft.py:96-103, 430-433
// Detrending (xrft/detrend.py:100-113): the plane needs sums over the whole slab, which exist only after this pass, so the
// pass transforms the raw windowed data, produces the exact per-column sums (sum d, sum (i - ibar) d) on the side -- float64
// per thread, wave shuffles, one small LDS table summed after the transforms -- and pass 2 subtracts the plane in the spectral
// domain: wx[x] (alpha_x What0[ky] + gamma_x What1[ky]).  In float64 nothing cancels visibly (trend / signal of 10^4 costs 13 of
// 53 bits; the tests hold 1e-10).
// ------------------------------------------------------------------------------------------------
// GOV = 4 (float32, 1800 / 2000 / 2160 rows: XRFT_M_WIDE32): four sequences = 8 real columns = 32-byte row segments where the default
// geometry (at most 640 threads) takes two -- 16-byte segments load at half the rate: (64, 2000, 2000) float32 12.5 us per slab for 6 us of
// bytes.  Taken when the row length divides into 8-column blocks; the intermediate's layout follows (FastM::l_cw, l_rk).
template <typename T, int NY, bool DET, int GOV = 0>
__global__ void __launch_bounds__((MGeom<T, NY, GOV>::THR), (MGeom<T, NY, GOV>::WPS)) fastm_cols_kernel(FastM p) {
    typedef MGeom<T, NY, GOV> M;
    typedef C2<T> CT;
    constexpr int G = M::G, THR = M::THR, STR = M::STR, CW = 2 * G;
    static_assert(THR >= G * M::B0 && THR % G == 0, "one first-pass butterfly per thread");
    XRFT_DYN_SMEM(smem_raw);
    CT* lds = reinterpret_cast<CT*>(smem_raw);
    CT* tw1 = lds + G * STR;
    double* part = reinterpret_cast<double*>(tw1 + M::M0);  // [wave][g][4]
    const int tid = threadIdx.x, g = tid % G, r0 = tid / G;
    // unit = (slab, column block).  Workgroups b, b + 8, ... run on one XCD: every XCD gets a contiguous range of units, so
    // that the workgroups that share the 128-byte lines of the input rows share an L2 (fasty.h; 45.9 vs 27.8 us measured there)
    const int per = (p.nunits + 7) >> 3, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int unit = xcd * per + jb;
    if (jb >= per || unit >= p.nunits) return;
    const int nxb = p.nx / CW, slab = unit / nxb, xb = unit % nxb;
    mr_fill_tw1<T, NY>(tw1, reinterpret_cast<const CT*>(p.tw_y), tid, THR);
    // first pass from registers: thread (g, j = r0) loads rows j + q M0, q < R0, of its sequence (lanes (j, g), g fastest: the
    // lanes of a row read 16 G contiguous bytes) -- all of them in flight at once, none staged in LDS
    constexpr int R0 = M::R0, M0 = M::M0;
    const bool on = r0 < M::B0;
    const int j = on ? r0 : 0;
    const CT w0 = reinterpret_cast<const CT*>(p.tw_y)[j];
    const char* __restrict__ src = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.in) + (size_t)slab * NY * p.nx + (size_t)xb * CW);
    const unsigned rowb = (unsigned)p.nx * (unsigned)sizeof(T), off0 = (unsigned)j * rowb + (unsigned)g * (unsigned)sizeof(CT), rstep = (unsigned)M0 * rowb;
    const T* __restrict__ wy = reinterpret_cast<const T*>(p.win_y);
    const CT wx = *reinterpret_cast<const CT*>(reinterpret_cast<const T*>(p.win_x) + xb * CW + 2 * g);
    CT a[R0];
    T wyv[R0];
    // float32: what is subtracted HERE, in y-space, only has to take the bulk of the trend out so that nothing cancels in float32 --
    // a line per column estimated from the medians of the KREF adjacent rows around ny/4 and around 3 ny/4 (away from the edges,
    // robust to a spike in one row; every thread of the sequence loads them itself: the same addresses in all lanes), rounded to
    // a power-of-two grid on which T + S i is exact (fasty.h); pass 2 corrects whatever was subtracted.  float64 subtracts nothing.
    constexpr bool PRE = DET && sizeof(T) == 4;
    constexpr int KREF = 3, ITOP = NY / 4, IBOT = 3 * NY / 4;
    CT rt[KREF], rb[KREF];
    if (PRE) {
#pragma unroll
        for (int k = 0; k < KREF; ++k) {
            rt[k] = *reinterpret_cast<const CT*>(src + ((unsigned)g * (unsigned)sizeof(CT) + rowb * (unsigned)(ITOP - 1 + k)));
            rb[k] = *reinterpret_cast<const CT*>(src + ((unsigned)g * (unsigned)sizeof(CT) + rowb * (unsigned)(IBOT - 1 + k)));
        }
    }
#pragma unroll
    for (int q = 0; q < R0; ++q) {
        a[q] = mk<T>((T)0, (T)0); wyv[q] = (T)0;
        if (on) {
            if (!(XRFT_MDBG & 16)) a[q] = *reinterpret_cast<const CT*>(src + (off0 + rstep * (unsigned)q));
            wyv[q] = wy[j + q * M0];
        }
    }
    constexpr double IBAR = 0.5 * (NY - 1);
    float Tl[2] = {0.f, 0.f}, Sl[2] = {0.f, 0.f};
    if (PRE) {
        auto med3 = [](float x, float y, float z) { return fmaxf(fminf(x, y), fminf(fmaxf(x, y), z)); };
        const float mt[2] = {med3((float)rt[0].re, (float)rt[1].re, (float)rt[2].re), med3((float)rt[0].im, (float)rt[1].im, (float)rt[2].im)};
        const float mb[2] = {med3((float)rb[0].re, (float)rb[1].re, (float)rb[2].re), med3((float)rb[0].im, (float)rb[1].im, (float)rb[2].im)};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float top = mt[c], bot = mb[c];  // the column near i = ITOP and i = IBOT
            const float Se = p.detrend == 2 ? (bot - top) * (1.0f / (IBOT - ITOP)) : 0.f;
            const float Te = p.detrend == 2 ? top - Se * (float)ITOP : 0.5f * (top + bot);
            const float mag = fabsf(Te) + fabsf(Se) * (float)NY;
            const float C = __uint_as_float((__float_as_uint(mag) & 0x7f800000u) + (3u << 23)) * 1.5f;  // rounds to 2^(e-20), 2^e <= mag
            Tl[c] = (Te + C) - C; Sl[c] = (Se + C) - C;
        }
        if (r0 == 0) {  // what is subtracted, as (offset at ibar, slope)
            double* cfp = p.colfit + ((size_t)slab * p.nx + xb * CW + 2 * g) * 4;
#pragma unroll
            for (int c = 0; c < 2; ++c) { cfp[4 * c + 2] = (double)Tl[c] + (double)Sl[c] * IBAR; cfp[4 * c + 3] = (double)Sl[c]; }
        }
    }
    if (DET) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            const double ri = (double)(j + q * M0) - IBAR;
            s[0] += (double)a[q].re; s[1] += (double)a[q].im;
            s[2] = fma(ri, (double)a[q].re, s[2]); s[3] = fma(ri, (double)a[q].im, s[3]);
        }
#pragma unroll
        for (int m = G; m < 64; m <<= 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += __shfl_xor(s[c], m);
        if ((tid & 63) < G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) part[((tid >> 6) * G + g) * 4 + c] = s[c];
        }
    }
#pragma unroll
    for (int q = 0; q < R0; ++q) {
        if (PRE) {
            const float fi = (float)(j + q * M0);
            a[q] = mk<T>((T)((float)a[q].re - fmaf(Sl[0], fi, Tl[0])), (T)((float)a[q].im - fmaf(Sl[1], fi, Tl[1])));
        }
        a[q] = mk<T>(a[q].re * (wyv[q] * wx.re), a[q].im * (wyv[q] * wx.im));
    }
    if (XRFT_MDBG & 4) {
        if (on) for (int q = 0; q < R0; ++q) lds[g * STR + M::pn(j + q * M0)] = a[q];
        __syncthreads();
    } else {
        if (on) mr_pass0<T, NY>(a, lds + g * STR, j, w0);
        mr_fft_tail<T, NY, G, THR>(lds, tid, tw1);
    }
    if (DET && tid < 4 * G) {  // (sum d, sum (i - ibar) d, 0, 0) per column
        const int c = tid / G, gg = tid % G;  // c: 0, 1 = sum d of columns 2gg, 2gg+1; 2, 3 = the first moments
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < THR / 64; ++w) acc += part[(w * G + gg) * 4 + c];
        double* cfp = p.colfit + ((size_t)slab * p.nx + xb * CW + 2 * gg + (c & 1)) * 4;
        cfp[c >> 1] = acc;
        if (!PRE) cfp[2 + (c >> 1)] = 0.0;
    }
    // split the packed spectra: Ra[k] = (Z[k] + conj Z[N-k]) / 2, Rb[k] = (Z[k] - conj Z[N-k]) / (2i) = the spectra of columns 2g and
    // 2g+1.  One 16-byte value per lane, lanes (ky, column): CW consecutive lanes write the CW columns of a row, RK rows
    // complete a 128-byte line -- every store instruction writes whole lines.  (Two stores per lane -- Ra then Rb, each instruction
    // half of every 32-byte sector -- ran the stores at 2.8 TB/s: 2.99 us per slab alone, profiles/r02_experiments.txt.)
    const int rk = 1 << p.l_rk;
    char* __restrict__ w2s = reinterpret_cast<char*>(reinterpret_cast<CT*>(p.w2) + (size_t)slab * p.nrow_pad * p.nx);
    constexpr int NST = (CW * (NY / 2 + 1) + THR - 1) / THR;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int l = tid + i * THR, col = l % CW, k = l / CW;
        if (k <= NY / 2) {
            const CT* z = lds + (col >> 1) * STR;
            const CT zk = z[M::pn(k)], zc = cconj(z[M::pn(k == 0 ? 0 : NY - k)]);
            const CT o = (col & 1) ? cscale(mul_mi(zk - zc), (T)0.5) : cscale(zk + zc, (T)0.5);
            const unsigned off = ((((unsigned)(k >> p.l_rk) * (unsigned)nxb + (unsigned)xb) << p.l_rk) + (unsigned)(k & (rk - 1))) * (unsigned)CW + (unsigned)col;
            if (!(XRFT_MDBG & 8) || o.re == (T)1.2345) mr_store_ct_nt<T, (XRFT_MDBG & 1) != 0>(w2s + (size_t)off * sizeof(CT), o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One transform axis that is NOT the contiguous one (XRFTHIP_AXIS_Y: xrft.fft / power_spectrum along "time" of a
// (..., time, ...) array, in place in memory order -- the reference's most common call): pass 1 alone IS the transform.  A
// workgroup owns CW adjacent columns of one slab [ny][nx] exactly as fastm_cols_kernel does; the whole column is inside the
// workgroup, so the per-column detrend of the reference (scipy.signal.detrend along the axis, xrft/detrend.py:54-71) is exact
// and local: column sums -> wave shuffles -> LDS -> every thread subtracts its columns' mean / least-squares line in float64
// from the samples it holds, then windows them.  The spectrum of every column leaves as rows ky and -ky (conjugate) of the
// caller's [ny][nx] result, rotated by the fftshift: |F|^2 scale (MODE 1) or F scale phase[ky] (MODE 0; the phase table carries
// the true-phase factor and the (-1)^k of an ifftshifted input).  xrft.py:425-447, 462-469, 740-748.
// ------------------------------------------------------------------------------------------------
// (MODE 0 complex, 1 power, 2 two fields: cross spectrum, or its phase with p.angle; the detrend is a run-time switch: fewer
// instantiations -- 20 lengths x 2 precisions x 3 modes)
// (a workgroup needs at least two sequences = four real columns here: 4096 points, and 2048 in float64, get them by override)
template <typename T, int NY> struct MYGeom { typedef MGeom<T, NY, (MGeom<T, NY>::G0 < 2 ? 2 : 0)> type; };

template <typename T, int NY, int MODE>
__global__ void __launch_bounds__((MYGeom<T, NY>::type::THR), (MYGeom<T, NY>::type::WPS)) fastm_yonly_kernel(FastM p) {
    typedef typename MYGeom<T, NY>::type M;
    typedef C2<T> CT;
    constexpr bool TWO = MODE >= 2;
    const bool DET = p.detrend != 0;  // cross spectrum / cross phase: column c of field 0 and of field 1 are the two halves of sequence c
    constexpr int G = M::G, THR = M::THR, STR = M::STR, R0 = M::R0, M0 = M::M0;
    const bool CIN = !TWO && p.cin != 0;          // complex input: column c IS sequence c
    const int CW = (TWO || CIN) ? G : 2 * G;     // columns per workgroup
    XRFT_DYN_SMEM(smem_raw);
    CT* lds = reinterpret_cast<CT*>(smem_raw);
    CT* tw1 = lds + G * STR;
    double* part = reinterpret_cast<double*>(tw1 + M::M0);  // [wave][g][4]
    const int tid = threadIdx.x, g = tid % G, r0 = tid / G;
    const int per = (p.nunits + 7) >> 3, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int unit = xcd * per + jb;
    if (jb >= per || unit >= p.nunits) return;
    const int nxb = CIN ? (p.nx + CW - 1) / CW : p.nx / CW, slab = unit / nxb, xb = unit % nxb;  // (complex columns: the last block of a slab may be short)
    mr_fill_tw1<T, NY>(tw1, reinterpret_cast<const CT*>(p.tw_y), tid, THR);
    const bool colin = !CIN || xb * CW + g < p.nx;
    const bool on = r0 < M::B0;
    const int j = on ? r0 : 0;
    const CT w0 = reinterpret_cast<const CT*>(p.tw_y)[j];
    const size_t esz = CIN ? sizeof(CT) : sizeof(T);  // bytes per input element
    const char* __restrict__ src = reinterpret_cast<const char*>(p.in) + ((size_t)slab * NY * p.nx + (size_t)xb * CW) * esz;
    const char* __restrict__ srcb = reinterpret_cast<const char*>(TWO ? p.in_b : p.in) + ((size_t)slab * NY * p.nx + (size_t)xb * CW) * esz;
    // (64-bit offsets: nx is everything behind the axis -- a (time, y, x) cube is ONE slab of ny * nx elements, far beyond 4 GB)
    const size_t rowb = (size_t)p.nx * esz, off0 = (size_t)j * rowb + (size_t)g * (TWO ? sizeof(T) : sizeof(CT)), rstep = (size_t)M0 * rowb;
    const T* __restrict__ wy = reinterpret_cast<const T*>(p.win_y);
    CT a[R0];
    T wyv[R0];
#pragma unroll
    for (int q = 0; q < R0; ++q) {
        a[q] = mk<T>((T)0, (T)0); wyv[q] = (T)0;
        if (on && colin) {
            if (TWO) a[q] = mk<T>(*reinterpret_cast<const T*>(src + (off0 + rstep * (size_t)q)), *reinterpret_cast<const T*>(srcb + (off0 + rstep * (size_t)q)));
            else if (CIN && (p.inv | p.ishift_in | p.ph_in)) {  // an inverse transform along the axis (xrft.ifft, xrft.py:479-646): the fftshifted input rotated, the lag's phase on the input, conj in
                int rs = j + q * M0 + p.ishift_in; if (rs >= NY) rs -= NY;
                CT z = *reinterpret_cast<const CT*>(src + ((size_t)rs * rowb + (size_t)g * sizeof(CT)));
                if (p.ph_in) z = cmul(z, reinterpret_cast<const CT*>(p.ph_y)[rs]);
                if (p.inv) z.im = -z.im;
                a[q] = z;
            } else a[q] = *reinterpret_cast<const CT*>(src + (off0 + rstep * (size_t)q));
            wyv[q] = wy[j + q * M0];
        }
    }
    constexpr double IBAR = 0.5 * (NY - 1);
    if (DET) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            const double ri = (double)(j + q * M0) - IBAR;
            s[0] += (double)a[q].re; s[1] += (double)a[q].im;
            s[2] = fma(ri, (double)a[q].re, s[2]); s[3] = fma(ri, (double)a[q].im, s[3]);
        }
#pragma unroll
        for (int m = G; m < 64; m <<= 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += __shfl_xor(s[c], m);
        if ((tid & 63) < G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) part[((tid >> 6) * G + g) * 4 + c] = s[c];
        }
        __syncthreads();
        double tot[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int w = 0; w < THR / 64; ++w)
#pragma unroll
            for (int c = 0; c < 4; ++c) tot[c] += part[(w * G + g) * 4 + c];
        // mean, and the slope of the least-squares line through (i - ibar): sum (i - ibar)^2 = n (n^2 - 1) / 12
        constexpr double INV_N = 1.0 / NY, INV_SII = 12.0 / ((double)NY * ((double)NY * NY - 1.0));
        const double m0 = tot[0] * INV_N, m1 = tot[1] * INV_N;
        const double sl0 = p.detrend == 2 ? tot[2] * INV_SII : 0.0, sl1 = p.detrend == 2 ? tot[3] * INV_SII : 0.0;
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            const double ri = (double)(j + q * M0) - IBAR;
            a[q] = mk<T>((T)((double)a[q].re - fma(sl0, ri, m0)), (T)((double)a[q].im - fma(sl1, ri, m1)));
        }
    }
#pragma unroll
    for (int q = 0; q < R0; ++q) a[q] = cscale(a[q], wyv[q]);
    if (on) mr_pass0<T, NY>(a, lds + g * STR, j, w0);
    mr_fft_tail<T, NY, G, THR>(lds, tid, tw1);
    // split the packed spectra (fastm_cols_kernel) and store rows ky and -ky of the result; lanes (ky, column), column fastest
    // (p.half -- real_dim along this axis, xrft.py:400-404: only k = 0 .. NY/2 is stored, NY/2 + 1 rows per slab, unshifted; p.realdim2: 0 < k < NY/2 counts twice, xrft.py:673-682)
    const int orows = p.half ? NY / 2 + 1 : NY;
    char* __restrict__ outs = reinterpret_cast<char*>(p.out) + ((size_t)slab * orows * p.nx + (size_t)xb * CW) * ((MODE == 1 || (MODE == 2 && p.angle)) ? sizeof(T) : sizeof(CT));
    const T sc = (T)p.scale;
    if (CIN) {  // every frequency of every column, no mirror
        for (int l = tid; l < CW * NY; l += THR) {
            const int col = l % CW, k = l / CW;
            if (xb * CW + col >= p.nx) continue;
            CT o = lds[col * STR + M::pn(k)];
            if (p.inv) o.im = -o.im;
            int rd = k + p.shift_y; if (rd >= NY) rd -= NY;
            if (MODE == 1) reinterpret_cast<T*>(outs)[(size_t)rd * p.nx + col] = (o.re * o.re + o.im * o.im) * sc;
            else {
                o = cscale(o, sc);
                if (p.ph_on) o = cmul(o, reinterpret_cast<const CT*>(p.ph_y)[k]);
                reinterpret_cast<CT*>(outs)[(size_t)rd * p.nx + col] = o;
            }
        }
        return;
    }
    constexpr int NST = (2 * G * (NY / 2 + 1) + THR - 1) / THR;
    const int cwsh = CW == G ? ilog2c(G) : ilog2c(2 * G);  // (CW is a power of two)
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int l = tid + i * THR, col = l & (CW - 1), k = l >> cwsh;
        if (k <= NY / 2) {
            const CT* z = lds + (TWO ? col : (col >> 1)) * STR;
            const CT zk = z[M::pn(k)], zc = cconj(z[M::pn(k == 0 ? 0 : NY - k)]);
            CT o;
            if (TWO) o = cmulc(cscale(zk + zc, (T)0.5), cscale(mul_mi(zk - zc), (T)0.5));  // F0 conj(F1) of this column (xrft.py:825)
            else o = (col & 1) ? cscale(mul_mi(zk - zc), (T)0.5) : cscale(zk + zc, (T)0.5);
            const int km = k == 0 ? 0 : NY - k;
            int rd = k + p.shift_y; if (rd >= NY) rd -= NY;
            int rm = km + p.shift_y; if (rm >= NY) rm -= NY;
            const bool interior = k != 0 && 2 * k != NY, mirror = interior && !p.half;
            const T sck = (p.realdim2 && interior) ? sc + sc : sc;
            if (MODE == 1) {
                const T v = (o.re * o.re + o.im * o.im) * sck;
                reinterpret_cast<T*>(outs)[(size_t)rd * p.nx + col] = v;
                if (mirror) reinterpret_cast<T*>(outs)[(size_t)rm * p.nx + col] = v;
            } else {
                o = cscale(o, sck);
                CT om = cconj(o);
                if (p.ph_on) { o = cmul(o, reinterpret_cast<const CT*>(p.ph_y)[k]); om = cmul(om, reinterpret_cast<const CT*>(p.ph_y)[km]); }
                if (MODE == 2 && p.angle) {  // cross phase (xrft.py:838-874)
                    reinterpret_cast<T*>(outs)[(size_t)rd * p.nx + col] = (T)atan2((double)o.im, (double)o.re);
                    if (mirror) reinterpret_cast<T*>(outs)[(size_t)rm * p.nx + col] = (T)atan2((double)om.im, (double)om.re);
                } else {
                    reinterpret_cast<CT*>(outs)[(size_t)rd * p.nx + col] = o;
                    if (mirror) reinterpret_cast<CT*>(outs)[(size_t)rm * p.nx + col] = om;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One transform axis that IS the contiguous one, short rows (ndim = 1: xrft.fft / power_spectrum along the last axis of
// (..., n) arrays, n in the table): the same transform with ROWS packed in pairs -- rows 2g, 2g+1 of the workgroup's 2 G rows are
// the real and imaginary part of sequence g.  Per-row detrend (mean / least-squares line along the row, scipy.signal.detrend,
// xrft/detrend.py:54-71) from the sums of the samples the threads hold, as in fastm_yonly_kernel; window; three passes; each
// row's spectrum (all n frequencies, or n/2 + 1 with real_dim) leaves rotated by the fftshift, contiguous along k.
// ------------------------------------------------------------------------------------------------
template <typename T, int N, int MODE>
__global__ void __launch_bounds__((MGeom<T, N>::THR), (MGeom<T, N>::WPS)) fastm_xonly_kernel(FastM p) {
    typedef MGeom<T, N> M;
    typedef C2<T> CT;
    constexpr bool TWO = MODE >= 2;
    const bool DET = p.detrend != 0;  // cross spectrum / cross phase: row r of field 0 and of field 1 are the two halves of a sequence
    constexpr int G = M::G, THR = M::THR, STR = M::STR, R0 = M::R0, M0 = M::M0;
    const bool CIN = !TWO && p.cin != 0;           // complex input: row r IS sequence r
    // irfft (xrft.ifft with real_dim, xrft.py:612-621): the Hermitian extensions A, B of TWO stored half rows travel as one sequence C = A + i B; its
    // inverse transform z = conj(FFT(conj C)) holds row a in its real part and row b in its imaginary part
    const bool C2R = CIN && p.c2r != 0;
    const int RPW = (TWO || (CIN && !C2R)) ? G : 2 * G;     // rows per workgroup
    XRFT_DYN_SMEM(smem_raw);
    CT* lds = reinterpret_cast<CT*>(smem_raw);
    CT* tw1 = lds + G * STR;
    double* part = reinterpret_cast<double*>(tw1 + M::M0);  // [wave][g][4]
    const int tid = threadIdx.x, g = tid % G, r0 = tid / G;
    const long long row0 = (long long)blockIdx.x * RPW, nrows = p.nslab;  // (nslab: rows in all)
    mr_fill_tw1<T, N>(tw1, reinterpret_cast<const CT*>(p.tw_x), tid, THR);
    const bool on = r0 < M::B0;
    const int j = on ? r0 : 0;
    const CT w0 = reinterpret_cast<const CT*>(p.tw_x)[j];
    const long long ra = (TWO || (CIN && !C2R)) ? row0 + g : row0 + 2 * g, rb = (TWO || (CIN && !C2R)) ? ra : ra + 1;
    const bool ha = on && ra < nrows, hb = on && rb < nrows;
    const CT* __restrict__ sc_in = reinterpret_cast<const CT*>(p.in) + (size_t)(ha ? ra : 0) * N;  // (complex input)
    const T* __restrict__ sa = reinterpret_cast<const T*>(p.in) + (size_t)(ha ? ra : 0) * N;
    const T* __restrict__ sb = reinterpret_cast<const T*>(TWO ? p.in_b : p.in) + (size_t)(hb ? rb : 0) * N;
    const T* __restrict__ wx = reinterpret_cast<const T*>(p.win_x);
    CT a[R0];
    T wv[R0];
#pragma unroll
    for (int q = 0; q < R0; ++q) {
        const int x = j + q * M0;
        if (C2R) {
            constexpr int HW = N / 2 + 1;
            const int xs = 2 * x <= N ? x : N - x;  // the stored sample; beyond n/2 its conjugate
            CT A = ha ? (reinterpret_cast<const CT*>(p.in) + (size_t)ra * HW)[xs] : mk<T>((T)0, (T)0);
            CT B = hb ? (reinterpret_cast<const CT*>(p.in) + (size_t)rb * HW)[xs] : mk<T>((T)0, (T)0);
            if (p.ph_in) { const CT f = reinterpret_cast<const CT*>(p.ph_x)[xs]; A = cmul(A, f); B = cmul(B, f); }
            if (x == 0 || 2 * x == N) { A.im = (T)0; B.im = (T)0; }  // (numpy's irfft takes the real parts of the zero-frequency and Nyquist samples)
            if (2 * x > N) { A.im = -A.im; B.im = -B.im; }
            a[q] = mk<T>(A.re - B.im, -(A.im + B.re));  // conj(A + i B)
        } else if (CIN) {
            int xs = x + p.ishift_in; if (xs >= N) xs -= N;
            CT z = ha ? sc_in[xs] : mk<T>((T)0, (T)0);
            if (p.ph_in) z = cmul(z, reinterpret_cast<const CT*>(p.ph_x)[xs]);
            if (p.inv) z.im = -z.im;
            a[q] = z;
        } else a[q] = mk<T>(ha ? sa[x] : (T)0, hb ? sb[x] : (T)0);
        wv[q] = wx[x];
    }
    constexpr double XBAR = 0.5 * (N - 1);
    if (DET) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        if (on) {
#pragma unroll
            for (int q = 0; q < R0; ++q) {
                const double ri = (double)(j + q * M0) - XBAR;
                s[0] += (double)a[q].re; s[1] += (double)a[q].im;
                s[2] = fma(ri, (double)a[q].re, s[2]); s[3] = fma(ri, (double)a[q].im, s[3]);
            }
        }
#pragma unroll
        for (int m = G; m < 64; m <<= 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += __shfl_xor(s[c], m);
        if ((tid & 63) < G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) part[((tid >> 6) * G + g) * 4 + c] = s[c];
        }
        __syncthreads();
        double tot[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int w = 0; w < THR / 64; ++w)
#pragma unroll
            for (int c = 0; c < 4; ++c) tot[c] += part[(w * G + g) * 4 + c];
        constexpr double INV_N = 1.0 / N, INV_SII = 12.0 / ((double)N * ((double)N * N - 1.0));
        const double m0 = tot[0] * INV_N, m1 = tot[1] * INV_N;
        const double sl0 = p.detrend == 2 ? tot[2] * INV_SII : 0.0, sl1 = p.detrend == 2 ? tot[3] * INV_SII : 0.0;
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            const double ri = (double)(j + q * M0) - XBAR;
            a[q] = mk<T>((T)((double)a[q].re - fma(sl0, ri, m0)), (T)((double)a[q].im - fma(sl1, ri, m1)));
        }
    }
#pragma unroll
    for (int q = 0; q < R0; ++q) a[q] = cscale(a[q], wv[q]);
    if (on) mr_pass0<T, N>(a, lds + g * STR, j, w0);
    mr_fft_tail<T, N, G, THR>(lds, tid, tw1);
    // split and store: lanes run along k of one row
    const bool real_out = MODE == 1 || (MODE == 2 && p.angle) || C2R;
    const int W = p.half ? N / 2 + 1 : N;
    const T sc = (T)p.scale;
    for (int e = tid; e < RPW * W; e += THR) {
        const int t = e / W, k = e - t * W;
        const long long row = row0 + t;
        if (row >= nrows) break;  // (t grows with e)
        if (C2R) {  // sample k of row t: Re z (the even row of the pair) or Im z = -Im FFT(conj C)
            const CT zr = lds[(t >> 1) * STR + M::pn(k)];
            int ocr = k + p.shift_x; if (ocr >= N) ocr -= N;
            reinterpret_cast<T*>(p.out)[(size_t)row * N + ocr] = ((t & 1) ? -zr.im : zr.re) * sc;
            continue;
        }
        const CT* z = lds + ((TWO || CIN) ? t : (t >> 1)) * STR;
        const CT zk = z[M::pn(k)], zc = cconj(z[M::pn(k == 0 ? 0 : N - k)]);
        CT o;
        if (CIN) { o = zk; if (p.inv) o.im = -o.im; }
        else if (TWO) o = cmulc(cscale(zk + zc, (T)0.5), cscale(mul_mi(zk - zc), (T)0.5));  // F0 conj(F1) of this row (xrft.py:825)
        else o = (t & 1) ? cscale(mul_mi(zk - zc), (T)0.5) : cscale(zk + zc, (T)0.5);
        int oc = k + p.shift_x; if (oc >= N) oc -= N;  // (half output: shift_x = 0)
        const T f = (p.realdim2 && k != 0 && 2 * k != N) ? sc * (T)2 : sc;
        char* dst = reinterpret_cast<char*>(p.out) + ((size_t)row * W + oc) * (real_out ? sizeof(T) : sizeof(CT));
        if (MODE == 1) {
            *reinterpret_cast<T*>(dst) = (o.re * o.re + o.im * o.im) * f;
        } else {
            o = cscale(o, f);
            if (p.ph_on) o = cmul(o, reinterpret_cast<const CT*>(p.ph_x)[k]);
            if (MODE == 2 && p.angle) *reinterpret_cast<T*>(dst) = (T)atan2((double)o.im, (double)o.re);
            else *reinterpret_cast<CT*>(dst) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: a workgroup owns RPU consecutive rows ky0.. of W2 (one contiguous block), adds the plane back, transforms along x and
// writes every row twice: as output row ky (rotated by the fftshift) and, reversed, as row -ky (Hermitian mirror of the
// spectrum of a real field).  MODE = xrfthip_out_mode: 1 power, 0 complex (fft), 2 cross / 3 cross phase (the first G/2
// sequences are rows of field 0, the others the same rows of field 1: F0 conj(F1) is formed on the way out).
//   xrft.py:446-447 (fftshift), :462-469 (true phase), :740-748 / :825-833 (|F|^2, F0 conj F1 and the scalings, in `scale`)
// ------------------------------------------------------------------------------------------------
//   xrft.py:895-906 (ISO: radial sums, here bit-reproducible: per-bin exponent bound by atomicMax, int64 fixed-point adds; aux_kernels.h)
// (two fields, or radial sums -- whose tables and partial-sum rows are per workgroup --: pass 1's count)
// (round 3: the kernels with the radial sums fused take the small one-field workgroup too -- a radial map's sums are gathered without tables,
// nothing is amortised over the rows of a workgroup any more)
template <typename T, int NX, int MODE, bool ISO> struct MRowsG { static constexpr int G = MODE >= 2 ? MGeom<T, NX>::G : MGeom<T, NX>::GR1; };

template <typename T, int NX, int MODE, bool ISO = false>
__global__ void __launch_bounds__((MGeom<T, NX>::template Rows<MRowsG<T, NX, MODE, ISO>::G>::THR), (MGeom<T, NX>::template Rows<MRowsG<T, NX, MODE, ISO>::G>::WPS)) fastm_rows_kernel(FastM p) {
    static_assert(!ISO || MODE == 1 || MODE == 2, "radial sums exist for power and cross spectra");
    typedef MGeom<T, NX> M;
    typedef C2<T> CT;
    constexpr bool TWO = MODE >= 2;
    constexpr int G = MRowsG<T, NX, MODE, ISO>::G, THR = M::template Rows<G>::THR, STR = M::STR, RPU = TWO ? G / 2 : G;
    static_assert(!TWO || G >= 2, "two fields need two sequences");
    XRFT_DYN_SMEM(smem_raw);
    CT* lds = reinterpret_cast<CT*>(smem_raw);
    CT* tw1 = lds + G * STR;
    const int tid = threadIdx.x;
    const int upr = p.nrow_pad / RPU, slab = blockIdx.x / upr, unit = blockIdx.x % upr, ky0 = unit * RPU, nyh = p.ny >> 1;
    mr_fill_tw1<T, NX>(tw1, reinterpret_cast<const CT*>(p.tw_x), tid, THR);
    // first pass from registers: thread (sequence t, j) loads x = j + q M0, q < R0, of its row.  Lane order (row pair, j, row in
    // the pair): the RK rows that share the lines of W2 sit in adjacent lanes, so a wave consumes whole lines.
    constexpr int R0 = M::R0, M0 = M::M0;
    const int rk = 1 << p.l_rk, cwm = (1 << p.l_cw) - 1;
    const int pairi = (tid >> p.l_rk) / M::B0, j = (tid >> p.l_rk) % M::B0, t = (pairi << p.l_rk) + (tid & (rk - 1));
    const bool on = t < G;
    const int f = TWO && t >= RPU ? 1 : 0, row = t - f * RPU;  // (two fields: sequences RPU.. are the same rows of field 1)
    const int ky = ky0 + row, kyc = min(ky, nyh);
    const CT w0 = reinterpret_cast<const CT*>(p.tw_x)[j];
    const CT* __restrict__ blk = reinterpret_cast<const CT*>(f ? p.w2b : p.w2) + ((size_t)slab * p.nrow_pad + (size_t)((ky >> p.l_rk) << p.l_rk)) * NX;
    const CT* __restrict__ cr = reinterpret_cast<const CT*>(f ? p.corr_b : p.corr) + (size_t)slab * NX;
    const bool addback = p.detrend != 0;
    const bool live = on && ky <= nyh;  // (padding rows of the last unit were never written by pass 1: they stay zero)
    CT a[R0], c[R0];
    CT h0 = mk<T>((T)0, (T)0), h1 = h0;
    if (addback && live) { h0 = reinterpret_cast<const CT*>(p.what0)[kyc]; h1 = reinterpret_cast<const CT*>(p.what1)[kyc]; }
#pragma unroll
    for (int q = 0; q < R0; ++q) {
        a[q] = mk<T>((T)0, (T)0); c[q] = a[q];
        if (live) {
            const int x = j + q * M0;
            // element (ky, x) of the block [x / CW][ky % RK][x % CW]
            if (!(XRFT_MDBG & 16)) a[q] = blk[((((x >> p.l_cw) << p.l_rk) + (ky & (rk - 1))) << p.l_cw) + (x & cwm)];
            if (addback) c[q] = cr[x];
        }
    }
    if (addback) {  // + wx[x] (alpha_x What0[ky] + gamma_x What1[ky]): the plane, subtracted in the spectral domain
#pragma unroll
        for (int q = 0; q < R0; ++q) {
            a[q].re = fma(c[q].re, h0.re, fma(c[q].im, h1.re, a[q].re));
            a[q].im = fma(c[q].re, h0.im, fma(c[q].im, h1.im, a[q].im));
        }
    }
    if (XRFT_MDBG & 4) {
        if (on) for (int q = 0; q < R0; ++q) lds[t * STR + M::pn(j + q * M0)] = a[q];
        __syncthreads();
    } else {
        if (on) mr_pass0<T, NX>(a, lds + t * STR, j, w0);
        mr_fft_tail<T, NX, G, THR>(lds, tid, tw1);
    }
    const int sx = p.shift_x, sy = p.shift_y;
    const T sc = (T)p.scale;
    if (ISO) {
        // Radial sums of this workgroup's samples -- rows ky0.. and their Hermitian mirrors -- straight from the spectra in LDS: two
        // sweeps (largest exponent per bin, then integer fixed-point adds: exact, so the order in which lanes arrive does not
        // matter), the sums go to this workgroup's row of the partial table.  The tables sit behind the twiddle table.
        constexpr int HW = MODE == 2 ? 2 : 1, NIT = (RPU * NX + THR - 1) / THR;
        if (p.tfirst != nullptr) {
            // A RADIAL bin map (verified on the host, fastm_build_tfirst: along a row the bin depends on |kx| only and never decreases
            // with it, every sample is binned, the Hermitian twin of a sample shares its bin) needs no atomics and no tables: the
            // bins of a row are contiguous ranges of kx on either side of kx = 0.  The samples of a bin are added in float64 in a fixed
            // order (each range in ascending |kx|, the ranges by a shuffle tree): bit-reproducible, inf / nan propagate as in any sum.
            // Only the bins the unit's rows reach are visited (p.twin), written and reduced.
            const unsigned bw = p.twin[unit];
            const int blo = (int)(bw & 0xffffu), bhi = (int)(bw >> 16);
            constexpr int H = NX / 2, HM = (NX - 1) / 2;  // |kx| = 0 .. H; kx = nx - |kx| exists for |kx| = 1 .. HM
            // task = (bin, row, side of kx = 0): 2 RPU adjacent lanes share a bin and their sums meet in lane order by shuffles (a thread per
            // bin left most of the workgroup idle behind chains of LDS round trips: the row pass with the radial sums ran 25 % behind the one
            // that stores the spectrum)
            constexpr int TPB = 2 * RPU;
            static_assert((TPB & (TPB - 1)) == 0 && TPB <= 64 && THR % TPB == 0, "tasks per bin");
            const int sub = tid % TPB, r = sub >> 1, side = sub & 1, kyr = ky0 + r;
            const bool rlive = kyr <= nyh, twin = kyr != 0 && 2 * kyr != p.ny;
            double* __restrict__ part = p.iso_part + ((size_t)slab * upr + unit) * p.nbins * HW;
            for (int b0 = blo; b0 < bhi; b0 += THR / TPB) {
                const int bn = b0 + tid / TPB;
                double rr = 0.0, ri = 0.0;
                if (rlive && bn < bhi) {
                    const unsigned short* __restrict__ fr = p.tfirst + (size_t)kyr * (p.nbins + 1) + bn;
                    const int s = fr[0], e = fr[1];  // the bin holds |kx| = s .. e - 1 of this row
                    auto take = [&](int kx) {
                        const CT va = lds[r * STR + M::pn(kx)];
                        if (MODE == 1) rr += (double)((va.re * va.re + va.im * va.im) * sc);
                        else { const CT v = cscale(cmulc(va, lds[(RPU + r) * STR + M::pn(kx)]), sc); rr += (double)v.re; ri += (double)v.im; }
                    };
                    if (side == 0) { for (int m = s; m < min(e, H + 1); ++m) take(m); }
                    else { for (int m = max(s, 1); m < min(e, HM + 1); ++m) take(NX - m); }
                    if (twin) { rr *= 2.0; ri = 0.0; }  // + the twin row (-ky): V + conj V
                }
#pragma unroll
                for (int m = 1; m < TPB; m <<= 1) {  // (row, side) in a fixed tree order
                    rr += __shfl_down(rr, m, TPB);
                    if (MODE == 2) ri += __shfl_down(ri, m, TPB);
                }
                if (sub == 0 && bn < bhi) {
                    part[bn * HW] = rr;
                    if (MODE == 2) part[2 * bn + 1] = ri;
                }
            }
        } else {
        const int nc = p.iso_ncopy, nbn = p.nbins;  // copies of the tables (lane l uses copy l % nc: neighbouring samples share bins)
        unsigned long long* acc_all = reinterpret_cast<unsigned long long*>(tw1 + M::M0);  // [nc][nbins][HW]
        unsigned* bmax_all = reinterpret_cast<unsigned*>(acc_all + (size_t)nc * nbn * HW);   // [nc][nbins]
        unsigned long long* acc = acc_all + (size_t)(tid & (nc - 1)) * nbn * HW;
        unsigned* bmax = bmax_all + (size_t)(tid & (nc - 1)) * nbn;
        for (int i = tid; i < nc * nbn * HW; i += THR) acc_all[i] = 0ull;
        for (int i = tid; i < nc * nbn; i += THR) bmax_all[i] = 0u;
        // a thread's samples: (row r, kx) and its mirror (-ky, -kx); their bins first, all loads in flight together
        int bd[NIT], bm[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * THR, r = e / NX, kx = e % NX, ky = ky0 + r;
            bd[it] = -1; bm[it] = -1;
            if (e < RPU * NX && ky <= nyh) {
                bd[it] = p.binmap[(size_t)ky * NX + kx];
                if (ky != 0 && 2 * ky != p.ny) bm[it] = p.binmap[(size_t)(p.ny - ky) * NX + (kx == 0 ? 0 : NX - kx)];
            }
        }
        __syncthreads();
        for (int sweep = 0; sweep < 2; ++sweep) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (bd[it] < 0 && bm[it] < 0) continue;
                const int e = tid + it * THR, r = e / NX, kx = e % NX, ky = ky0 + r;
                CT va = lds[r * STR + M::pn(kx)];
                double dr, di = 0.0, mr = 0.0, mi = 0.0;  // direct and mirrored value
                if (MODE == 1) {
                    dr = (double)((va.re * va.re + va.im * va.im) * sc);
                    mr = dr;
                } else {
                    va = cscale(cmulc(va, lds[(RPU + r) * STR + M::pn(kx)]), sc);
                    CT vm = cconj(va);
                    if (p.ph_on) {
                        va = cmul(va, cmul(reinterpret_cast<const CT*>(p.ph_y)[ky], reinterpret_cast<const CT*>(p.ph_x)[kx]));
                        vm = cmul(vm, cmul(reinterpret_cast<const CT*>(p.ph_y)[ky == 0 ? 0 : p.ny - ky], reinterpret_cast<const CT*>(p.ph_x)[kx == 0 ? 0 : NX - kx]));
                    }
                    dr = (double)va.re; di = (double)va.im; mr = (double)vm.re; mi = (double)vm.im;
                }
                const bool same = MODE == 1 && bd[it] == bm[it];  // (a power spectrum's two samples are equal: one add of twice the value)
                if (sweep == 0) {
                    if (bd[it] >= 0) atomicMax(&bmax[bd[it]], (unsigned)((unsigned long long)__double_as_longlong(fabs(dr) + fabs(di)) >> 32));
                    if (bm[it] >= 0 && !same) atomicMax(&bmax[bm[it]], (unsigned)((unsigned long long)__double_as_longlong(fabs(mr) + fabs(mi)) >> 32));
                } else {
                    if (bd[it] >= 0) {
                        const int eb = (int)(bmax_all[bd[it]] >> 20);
                        atomicAdd(&acc[HW * bd[it]], (unsigned long long)(iso_fixed(dr, eb) * (same ? 2 : 1)));
                        if (MODE == 2) atomicAdd(&acc[2 * bd[it] + 1], (unsigned long long)iso_fixed(di, eb));
                    }
                    if (bm[it] >= 0 && !same) {
                        const int eb = (int)(bmax_all[bm[it]] >> 20);
                        atomicAdd(&acc[HW * bm[it]], (unsigned long long)iso_fixed(mr, eb));
                        if (MODE == 2) atomicAdd(&acc[2 * bm[it] + 1], (unsigned long long)iso_fixed(mi, eb));
                    }
                }
            }
            __syncthreads();
            if (sweep == 0) {  // one bound per bin: the maximum over the copies, kept in copy 0
                for (int i = tid; i < nbn; i += THR) {
                    unsigned m = bmax_all[i];
                    for (int k = 1; k < nc; ++k) m = max(m, bmax_all[(size_t)k * nbn + i]);
                    bmax_all[i] = m;
                }
                __syncthreads();
            }
        }
        double* __restrict__ part = p.iso_part + ((size_t)slab * upr + unit) * nbn * HW;
        for (int i = tid; i < nbn * HW; i += THR) {
            long long sum = 0;
            for (int k = 0; k < nc; ++k) sum += (long long)acc_all[(size_t)k * nbn * HW + i];
            const unsigned bm = bmax_all[i / HW];
            double v = ldexp((double)sum, (int)(bm >> 20) - 1023 - kIsoFR);
            // a bin with an inf / nan member is +inf (a power spectrum whose only offenders are +inf) or nan, as the IEEE sum is
            if ((bm >> 20) == 0x7ffu) v = __longlong_as_double((MODE == 1 && bm == 0x7ff00000u) ? 0x7ff0000000000000ll : 0x7ff8000000000000ll);
            part[i] = v;
        }
        }
    }
    if (p.out == nullptr) return;
    // ---- the result leaves as whole rows, 16 bytes per lane and store: VW samples
    typedef typename std::conditional<MODE == 0 || MODE == 2, CT, T>::type OutT;
    constexpr int VW = 16 / (int)sizeof(OutT), CPR = NX / VW;
    static_assert(NX % VW == 0, "row length");
    if (p.half) {  // rows of nx/2 + 1 samples (an odd length: one sample per lane and store, still whole lines per wave); F(-ky, kx) = conj F(ky, -kx)
        constexpr int W = NX / 2 + 1;
        OutT* __restrict__ oh = reinterpret_cast<OutT*>(p.out) + (size_t)slab * p.ny * W;
        for (int e = tid; e < RPU * 2 * W; e += THR) {
            const int fx = e % W, rr = e / W, r = rr >> 1, mir = rr & 1, ky = ky0 + r;
            if (ky > nyh || (mir && (ky == 0 || 2 * ky == p.ny))) continue;
            const int fy = mir ? p.ny - ky : ky, kx = mir ? (fx == 0 ? 0 : NX - fx) : fx;
            int orow = fy + sy; if (orow >= p.ny) orow -= p.ny;
            const T f2 = (p.realdim2 && fx != 0 && 2 * fx != NX) ? (T)2 : (T)1;
            CT va = lds[r * STR + M::pn(kx)];
            OutT* dst = oh + (size_t)orow * W + fx;
            if (MODE == 1) {
                *reinterpret_cast<T*>(dst) = (va.re * va.re + va.im * va.im) * (sc * f2);
            } else {
                if (TWO) va = cmulc(va, lds[(RPU + r) * STR + M::pn(kx)]);
                va = cscale(va, sc * f2);
                if (mir) va = cconj(va);
                if (p.ph_on) va = cmul(va, cmul(reinterpret_cast<const CT*>(p.ph_y)[fy], reinterpret_cast<const CT*>(p.ph_x)[fx]));
                if (MODE == 3) *reinterpret_cast<T*>(dst) = (T)atan2((double)va.im, (double)va.re);
                else *reinterpret_cast<CT*>(dst) = va;
            }
        }
        return;
    }
    OutT* __restrict__ outs = reinterpret_cast<OutT*>(p.out) + (size_t)slab * p.ny * NX;
    for (int e = tid; e < RPU * 2 * CPR; e += THR) {
        const int chunk = e % CPR, rr = e / CPR, r = rr >> 1, mir = rr & 1;
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
            int kx = mir ? (fx == 0 ? 0 : NX - fx) : fx;    // F(-ky, fx) = conj F(ky, -fx)
            CT va = rowA[M::pn(kx)];
            if (MODE == 1) {
                reinterpret_cast<T*>(o)[i] = (va.re * va.re + va.im * va.im) * sc;
            } else {
                if (TWO) va = cmulc(va, rowB[M::pn(kx)]);  // F0 conj(F1)
                va = cscale(va, sc);
                if (mir) va = cconj(va);
                if (p.ph_on) va = cmul(va, cmul(py, reinterpret_cast<const CT*>(p.ph_x)[fx]));
                if (MODE == 3) reinterpret_cast<T*>(o)[i] = (T)atan2((double)va.im, (double)va.re);
                else reinterpret_cast<CT*>(o)[i] = va;
            }
        }
        if (!(XRFT_MDBG & 8) || reinterpret_cast<T*>(o)[0] == (T)1.2345) mr_store16_nt<T, (XRFT_MDBG & 2) != 0>(outs + (size_t)orow * NX + c, o);
    }
}

// ------------------------------------------------------------------------------------------------
// plane fit from the per-column sums (fasty_fit_kernel with the correction in the pipeline's own precision): one 256-thread
// block per slab, float64, fixed summation order.  corr[x] = wx[x] * (line pass 1 subtracted - plane) as (offset at ibar, slope).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) fastm_fit_kernel(const double* colfit, const T* win_x, C2<T>* corr, int nx, int ny, int detrend) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);
    const int slab = blockIdx.x, tid = threadIdx.x;
    const double* cf4 = colfit + (size_t)slab * nx * 4;
    const double xbar = 0.5 * (nx - 1), sxx = (double)nx * ((double)nx * nx - 1.0) / 12.0;
    const double inv_n = 1.0 / ny, inv_sii = 12.0 / ((double)ny * ((double)ny * ny - 1.0));
    double s[3] = {0.0, 0.0, 0.0};
    for (int x = tid; x < nx; x += 256) {
        const double m = cf4[4 * x] * inv_n, sl = cf4[4 * x + 1] * inv_sii;
        s[0] += m;
        s[1] += ((double)x - xbar) * m;
        s[2] += sl;
    }
    block_sum<3>(s, red);
    __syncthreads();
    if (tid == 0) { red[0] = s[0]; red[1] = s[1]; red[2] = s[2]; }
    __syncthreads();
    const double a = red[0] / nx;
    const double b = detrend == 2 ? red[1] / sxx : 0.0;
    const double c = detrend == 2 ? red[2] / nx : 0.0;
    C2<T>* out = corr + (size_t)slab * nx;
    for (int x = tid; x < nx; x += 256) {
        const double wx = (double)win_x[x];
        out[x] = mk<T>((T)(wx * (cf4[4 * x + 2] - a - b * ((double)x - xbar))), (T)(wx * (cf4[4 * x + 3] - c)));
    }
}

}  // namespace xrft

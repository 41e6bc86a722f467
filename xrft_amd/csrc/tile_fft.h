// tile_fft.h -- the generic batched tile FFT kernel (any length with prime factors <= 128).
//
// One workgroup owns a tile of T sequences x n complex points in LDS and runs an in-place decimation-in-
// frequency mixed-radix FFT on it: early passes touch lane-contiguous LDS addresses, the result is left in
// digit-reversed order and the store loop gathers it through a host-built table, so no ping-pong buffer is
// needed and twice as many sequences fit per tile (better HBM coalescing for strided "column" passes).
// Everything the reference does around numpy.fft is folded into the first pass' loads (detrend, window,
// flip, ifftshift -- xrft/xrft.py:425-442) and the last pass' stores (fftshift, true-phase factor, prod(dx),
// |F|^2 / F conj(G), real-dim doubling, window/density scaling, Hermitian mirror --
// xrft/xrft.py:446-472, 740-748, 825-833, 993-1004), so intermediates never exist as arrays.
//
// The shapes of BASELINE.json have their own kernels (fasty.h: float32 powers of two; fastm.h: float64 lat/lon lengths); this
// kernel takes every other shape and is the parity reference for the specialised ones.  Radial bin sums (isotropic spectra,
// xrft.py:895-906) are a separate, bit-reproducible pass over the stored spectrum (aux_kernels.h, radial_binsum_det_kernel).
#pragma once
#include "gpu_rt.h"

namespace xrft {

// ------------------------------------------------------------------------------------------------
// complex arithmetic
// ------------------------------------------------------------------------------------------------
template <typename T>
struct alignas(2 * sizeof(T)) C2 {
    T re, im;
};
template <typename T> __device__ __forceinline__ C2<T> mk(T a, T b) { C2<T> r; r.re = a; r.im = b; return r; }
template <typename T> __device__ __forceinline__ C2<T> operator+(C2<T> a, C2<T> b) { return mk<T>(a.re + b.re, a.im + b.im); }
template <typename T> __device__ __forceinline__ C2<T> operator-(C2<T> a, C2<T> b) { return mk<T>(a.re - b.re, a.im - b.im); }
template <typename T> __device__ __forceinline__ C2<T> cmul(C2<T> a, C2<T> b) { return mk<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
template <typename T> __device__ __forceinline__ C2<T> cmulc(C2<T> a, C2<T> b) { return mk<T>(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im); }  // a * conj(b)
template <typename T> __device__ __forceinline__ C2<T> cscale(C2<T> a, T s) { return mk<T>(a.re * s, a.im * s); }
template <typename T> __device__ __forceinline__ C2<T> cconj(C2<T> a) { return mk<T>(a.re, -a.im); }
template <typename T> __device__ __forceinline__ C2<T> mul_mi(C2<T> a) { return mk<T>(a.im, -a.re); }  // a * (-i)
template <typename T> __device__ __forceinline__ C2<T> mul_pi(C2<T> a) { return mk<T>(-a.im, a.re); }  // a * (+i)

// ------------------------------------------------------------------------------------------------
// in-register DFTs (forward, e^{-2 pi i qk/R}), natural order in and out
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void dft2(C2<T>& a0, C2<T>& a1) { C2<T> t = a0 - a1; a0 = a0 + a1; a1 = t; }

template <typename T> __device__ __forceinline__ void dft3(C2<T>* a) {
    const T s = (T)0.86602540378443864676;
    C2<T> t = a[1] + a[2];
    C2<T> m = mk<T>(a[0].re - (T)0.5 * t.re, a[0].im - (T)0.5 * t.im);
    C2<T> d = cscale(a[1] - a[2], s);
    a[0] = a[0] + t;
    a[1] = m + mul_mi(d);
    a[2] = m + mul_pi(d);
}

template <typename T> __device__ __forceinline__ void dft4(C2<T>* a) {
    C2<T> s02 = a[0] + a[2], d02 = a[0] - a[2], s13 = a[1] + a[3], d13 = a[1] - a[3];
    a[0] = s02 + s13;
    a[2] = s02 - s13;
    a[1] = d02 + mul_mi(d13);
    a[3] = d02 + mul_pi(d13);
}

template <typename T> __device__ __forceinline__ void dft5(C2<T>* a) {
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;
    const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;
    C2<T> t1 = a[1] + a[4], t2 = a[2] + a[3], t3 = a[1] - a[4], t4 = a[2] - a[3];
    C2<T> m1 = mk<T>(a[0].re + c1 * t1.re + c2 * t2.re, a[0].im + c1 * t1.im + c2 * t2.im);
    C2<T> m2 = mk<T>(a[0].re + c2 * t1.re + c1 * t2.re, a[0].im + c2 * t1.im + c1 * t2.im);
    C2<T> n1 = mk<T>(s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im);
    C2<T> n2 = mk<T>(s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im);
    a[0] = a[0] + t1 + t2;
    a[1] = m1 + mul_mi(n1);
    a[4] = m1 + mul_pi(n1);
    a[2] = m2 + mul_mi(n2);
    a[3] = m2 + mul_pi(n2);
}

template <typename T> __device__ __forceinline__ void dft8(C2<T>* a) {
    const T h = (T)0.70710678118654752440;
    C2<T> u[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { u[k] = a[k] + a[k + 4]; v[k] = a[k] - a[k + 4]; }
    // v[k] *= W8^k
    v[1] = mk<T>(h * (v[1].re + v[1].im), h * (v[1].im - v[1].re));
    v[2] = mul_mi(v[2]);
    v[3] = mk<T>(h * (v[3].im - v[3].re), -h * (v[3].re + v[3].im));
    dft4(u);
    dft4(v);
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[2 * k] = u[k]; a[2 * k + 1] = v[k]; }
}

// ------------------------------------------------------------------------------------------------
// radix-16 butterfly, natural order in and out (4 x 4 with constant twiddles)
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void dft16(C2<T>* a) {
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173, h = (T)0.70710678118654752440;
    C2<T> t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // stage 1: DFT4 over elements q, q+4, q+8, q+12  -> t[q][m]
        C2<T> b[4] = {a[q], a[q + 4], a[q + 8], a[q + 12]};
        dft4(b);
#pragma unroll
        for (int m = 0; m < 4; ++m) t[q][m] = b[m];
    }
    // twiddle t[q][m] *= W16^(q m)
    t[1][1] = mk<T>(c1 * t[1][1].re + s1 * t[1][1].im, c1 * t[1][1].im - s1 * t[1][1].re);  // W16^1
    t[1][2] = mk<T>(h * (t[1][2].re + t[1][2].im), h * (t[1][2].im - t[1][2].re));          // W16^2
    t[1][3] = mk<T>(s1 * t[1][3].re + c1 * t[1][3].im, s1 * t[1][3].im - c1 * t[1][3].re);  // W16^3
    t[2][1] = mk<T>(h * (t[2][1].re + t[2][1].im), h * (t[2][1].im - t[2][1].re));          // W16^2
    t[2][2] = mul_mi(t[2][2]);                                                              // W16^4 = -i
    t[2][3] = mk<T>(h * (t[2][3].im - t[2][3].re), -h * (t[2][3].re + t[2][3].im));         // W16^6
    t[3][1] = mk<T>(s1 * t[3][1].re + c1 * t[3][1].im, s1 * t[3][1].im - c1 * t[3][1].re);  // W16^3
    t[3][2] = mk<T>(h * (t[3][2].im - t[3][2].re), -h * (t[3][2].re + t[3][2].im));         // W16^6
    t[3][3] = mk<T>(-c1 * t[3][3].re - s1 * t[3][3].im, s1 * t[3][3].re - c1 * t[3][3].im); // W16^9 = -W16^1
#pragma unroll
    for (int m = 0; m < 4; ++m) {  // stage 2: DFT4 over q -> X[m + 4 p]
        C2<T> b[4] = {t[0][m], t[1][m], t[2][m], t[3][m]};
        dft4(b);
#pragma unroll
        for (int p = 0; p < 4; ++p) a[m + 4 * p] = b[p];
    }
}

template <typename T, int R> __device__ __forceinline__ void dft_r(C2<T>* a);

// Composite radices with coprime factors (6, 10, 12, 15, 20) by the prime-factor (Good-Thomas) mapping: no twiddles at all,
//   n = (B n1 + A n2) mod AB,   k = k1 mod A, k = k2 mod B (CRT);  every index is a compile-time constant.
constexpr int xrft_modinv(int a, int m) { for (int x = 1; x < m; ++x) if ((a * x) % m == 1) return x; return 0; }
template <typename T, int A, int B> __device__ __forceinline__ void dft_pfa(C2<T>* a) {
    constexpr int N = A * B, EB = B * xrft_modinv(B % A, A), EA = A * xrft_modinv(A % B, B);  // k = (k1 EB + k2 EA) mod N
    C2<T> y[N];
#pragma unroll
    for (int n2 = 0; n2 < B; ++n2) {
        C2<T> t[A];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) t[n1] = a[(B * n1 + A * n2) % N];
        dft_r<T, A>(t);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) y[k1 * B + n2] = t[k1];
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
        C2<T> t[B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) t[n2] = y[k1 * B + n2];
        dft_r<T, B>(t);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) a[(k1 * EB + k2 * EA) % N] = t[k2];
    }
}

// radix 9 = 3 x 3 (Cooley-Tukey, constant twiddles W9^1, W9^2, W9^4)
template <typename T> __device__ __forceinline__ void dft9(C2<T>* a) {
    const C2<T> w1 = mk<T>((T)0.76604444311897803520, (T)-0.64278760968653932632);
    const C2<T> w2 = mk<T>((T)0.17364817766693034885, (T)-0.98480775301220805937);
    const C2<T> w4 = mk<T>((T)-0.93969262078590838405, (T)-0.34202014332566873304);
    C2<T> y[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {  // n = 3 n1 + n2
        C2<T> t[3] = {a[n2], a[3 + n2], a[6 + n2]};
        dft3(t);
        y[0][n2] = t[0]; y[1][n2] = t[1]; y[2][n2] = t[2];
    }
    y[1][1] = cmul(y[1][1], w1); y[1][2] = cmul(y[1][2], w2);
    y[2][1] = cmul(y[2][1], w2); y[2][2] = cmul(y[2][2], w4);
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {  // k = k1 + 3 k2
        C2<T> t[3] = {y[k1][0], y[k1][1], y[k1][2]};
        dft3(t);
        a[k1] = t[0]; a[k1 + 3] = t[1]; a[k1 + 6] = t[2];
    }
}

// odd primes 7, 11, 13 (numpy's pocketfft has hard-coded passes for 7 and 11): with s_k = a[k] + a[P-k], d_k = a[k] - a[P-k],
//   X[m], X[P-m] = (a0 + sum_k cos(2 pi m k / P) s_k) -/+ i (sum_k sin(2 pi m k / P) d_k)  --  H^2 real x complex products each, H = (P - 1) / 2.
// Every table index is a compile-time constant after unrolling.
template <typename T, int P> struct PrimeTab;
template <typename T> struct PrimeTab<T, 7> {
    static __device__ __forceinline__ T c(int k) { const T t[3] = {(T)0.623489801858733530525, (T)-0.2225209339563144042889, (T)-0.9009688679024191262361}; return t[k - 1]; }
    static __device__ __forceinline__ T s(int k) { const T t[3] = {(T)0.7818314824680298087084, (T)0.9749279121818236070181, (T)0.4338837391175581204758}; return t[k - 1]; }
};
template <typename T> struct PrimeTab<T, 11> {
    static __device__ __forceinline__ T c(int k) { const T t[5] = {(T)0.8412535328311811688618, (T)0.4154150130018864255293, (T)-0.1423148382732851404438, (T)-0.6548607339452850640569, (T)-0.9594929736144973898904}; return t[k - 1]; }
    static __device__ __forceinline__ T s(int k) { const T t[5] = {(T)0.5406408174555975821076, (T)0.9096319953545183714117, (T)0.9898214418809327323761, (T)0.755749574354258283774, (T)0.2817325568414296977114}; return t[k - 1]; }
};
template <typename T> struct PrimeTab<T, 13> {
    static __device__ __forceinline__ T c(int k) { const T t[6] = {(T)0.8854560256532098959004, (T)0.5680647467311558025118, (T)0.1205366802553230533491, (T)-0.3546048870425356259696, (T)-0.7485107481711010986346, (T)-0.970941817426052027157}; return t[k - 1]; }
    static __device__ __forceinline__ T s(int k) { const T t[6] = {(T)0.464723172043768545656, (T)0.8229838658936563945796, (T)0.9927088740980539928008, (T)0.9350162426854148234398, (T)0.6631226582407952023768, (T)0.2393156642875577671488}; return t[k - 1]; }
};
template <typename T> struct PrimeTab<T, 17> {  // (round 5: 102 = 6 x 17 -- Rader's convolution for the 103 of the ERA5 grid's 721 = 7 x 103 latitudes; the Rader forms only)
    static __device__ __forceinline__ T c(int k) { const T t[8] = {(T)0.9324722294043558045731, (T)0.7390089172206591159245, (T)0.4457383557765382673965, (T)0.09226835946330199523965, (T)-0.2736629900720828635391, (T)-0.6026346363792563891786, (T)-0.8502171357296141521341, (T)-0.9829730996839017782819}; return t[k - 1]; }
    static __device__ __forceinline__ T s(int k) { const T t[8] = {(T)0.3612416661871529487447, (T)0.6736956436465572117127, (T)0.895163291355062322067, (T)0.9957341762950345218712, (T)0.9618256431728190704088, (T)0.7980172272802395033328, (T)0.5264321628773558002446, (T)0.1837495178165703315744}; return t[k - 1]; }
};
template <typename T, int P> __device__ __forceinline__ void dft_prime(C2<T>* a) {
    constexpr int H = (P - 1) / 2;
    C2<T> sm[H], df[H];
#pragma unroll
    for (int k = 1; k <= H; ++k) { sm[k - 1] = a[k] + a[P - k]; df[k - 1] = a[k] - a[P - k]; }
    const C2<T> a0 = a[0];
    C2<T> tot = a0;
#pragma unroll
    for (int k = 0; k < H; ++k) tot = tot + sm[k];
    a[0] = tot;
#pragma unroll
    for (int m = 1; m <= H; ++m) {
        C2<T> A = a0, B = mk<T>((T)0, (T)0);
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            const int mk_ = (m * k) % P, idx = mk_ <= H ? mk_ : P - mk_;   // cos is even, sin odd around P / 2
            const T cc = PrimeTab<T, P>::c(idx), ss = mk_ <= H ? PrimeTab<T, P>::s(idx) : -PrimeTab<T, P>::s(idx);
            A = mk<T>(A.re + cc * sm[k - 1].re, A.im + cc * sm[k - 1].im);
            B = mk<T>(B.re + ss * df[k - 1].re, B.im + ss * df[k - 1].im);
        }
        a[m] = A + mul_mi(B);       // A - i B
        a[P - m] = A + mul_pi(B);   // A + i B
    }
}

template <typename T, int R> __device__ __forceinline__ void dft_r(C2<T>* a) {
    if (R == 1) return;
    if (R == 2) dft2(a[0], a[1]);
    else if (R == 3) dft3(a);
    else if (R == 4) dft4(a);
    else if (R == 5) dft5(a);
    else if (R == 6) dft_pfa<T, 2, 3>(a);
    else if (R == 7) dft_prime<T, 7>(a);
    else if (R == 8) dft8(a);
    else if (R == 9) dft9(a);
    else if (R == 10) dft_pfa<T, 2, 5>(a);
    else if (R == 11) dft_prime<T, 11>(a);
    else if (R == 12) dft_pfa<T, 4, 3>(a);
    else if (R == 13) dft_prime<T, 13>(a);
    else if (R == 14) dft_pfa<T, 2, 7>(a);
    else if (R == 15) dft_pfa<T, 3, 5>(a);
    else if (R == 16) dft16(a);
    else if (R == 17) dft_prime<T, 17>(a);
    else if (R == 18) dft_pfa<T, 2, 9>(a);
    else if (R == 20) dft_pfa<T, 4, 5>(a);
}

// ------------------------------------------------------------------------------------------------
// parameter blocks (plain data, passed by value)
// ------------------------------------------------------------------------------------------------
#define XRFT_MAX_PASSES 24

struct TileGeom {
    int n;      // complex FFT length held in LDS per sequence
    int n_out;  // output points per sequence (n, n/…: r2c n+1, or truncated nxh)
    int nr;     // radix passes
    int radix[XRFT_MAX_PASSES];
    int T;           // sequences per tile
    int seq_stride;  // LDS elements between sequences
    int pad_shift;   // phys(pos) = pos + (pos >> pad_shift)
    int r2c;         // real input packed z[m] = y[2m] + i y[2m+1]; n = n_real/2, n_out = n+1
    int tile_axis;   // 0: a tile is T consecutive o (sequence contiguous: "row" pass); 1: T consecutive q ("column" pass)
    int in_fast;     // lanes iterate fastest over: 0 the point index, 1 the tile axis (pick the contiguous one)
    int out_fast;
    long long n_outer, inner, n_tiles, tiles_per_outer;
    // raw complex addressing (elements): o*so + q*sq + p*sp
    long long in_so, in_sq, in_sp, out_so, out_sq, out_sp;
    const void* in;
    void* out;
    const void* tw;         // W_n^k, k < n
    int tw_lds;             // 1: the kernel copies the table into LDS behind the tile (7 twiddle loads per radix-8 butterfly
                            // from global memory made the passes load-issue bound)
    const unsigned* rev;    // rev[k] = LDS position of frequency k after the DIF passes
    int rev_lds;            // 1: staged in LDS behind the twiddles (the store loop's lookups are on its critical path)
    const void* tw_r2c;     // W_{2n}^k, k <= n   (r2c unpack)
    // Bluestein: blue_n > 0 is the logical sequence length; n (a power of two >= 2 blue_n - 1) is what the LDS passes run.
    //   x[p] conj(c[p]) zero-padded -> forward passes -> * blue_b -> inverse passes -> * conj(c[k]),  c[k] = exp(i pi k^2 / blue_n)
    int rowc_off;           // > 0: byte offset of T per-row constant records in LDS; the first pass over contiguous rows of
                            // real input then takes the lean loader (see tile_fft_kernel)
    // tiled intermediate between a row pass and a column pass of Tc = til columns per tile: W[slab][kx / Tc][i][kx % Tc].
    // The column pass then reads each of its tiles as ONE contiguous block (reading Tc-column strips out of row-major rows
    // fetched every 128-byte line twice or more), the row pass writes Tc-element chunks.
    int out_tiled, in_tiled;      // Tc (a power of two) or 0
    long long til_stride;         // ny * Tc: elements between column tiles
    long long til_slab;           // (padded width / Tc) * ny * Tc: elements per slab
    int til_ny;
    int lean_final;         // 1: plain column pass writing |F|^2 or F: the lean epilogue applies (host-checked, see tile_fft_kernel)
                            // 2: last pass of a four-step transform along x (kx = q + q_mul' ... see there)
    int lean_col;           // column tiles of a four-step first pass: bit 0 = lean loader (real 1-D input), bit 1 = lean store
    int dbg;                // ablation switches for profiling (XRFTHIP_DBG): 1 skip the passes, 2 skip the store, 4 skip the load
    int blue_n;
    const void* blue_c;
    const void* blue_b;     // FFT_n(chirp) / n at the LDS position the forward passes leave each frequency
    const void* tw_big;     // four-step: W_bigN^k table, k < bigN
    long long tw_bigN, tw_qdiv, tw_qmod;  // factor = W_bigN^{((q / qdiv) % qmod) * k}
};

struct Prologue {  // first pass: xrft.py:425-442
    int in_complex;
    int detrend;
    long long rows;            // rows per slab seen by this pass (ny for 2-D, 1 for 1-D): b = o / rows, i = o % rows
    long long j_mul_p, j_mul_q;  // logical x index j = p*j_mul_p + q*j_mul_q
    int ny, nx;
    int flip_y, ishift_y, flip_x, ishift_x;
    long long slab_stride, row_stride;  // source strides in elements
    const void* in;
    const void* win_y;
    const void* win_x;
    const double* coef;  // [slab][6]: c0.re c0.im c1.re c1.im c2.re c2.im ; trend = c0 + c1*i + c2*j (source indices)
    // inverse transforms (xrft.ifft): complex input only
    const void* ph_y;    // complex tables multiplying the input, indexed by source position (nullable)
    const void* ph_x;
    int conj_in;         // conjugate after the phase multiply (ifft = conj(FFT(conj z)) / N)
    int herm_nxh;        // > 0: the source row holds only kx = 0..herm_nxh-1; kx >= herm_nxh is conj(src[-ky][nx-kx]) (irfftn)
    int p_is_row;        // XRFTHIP_AXIS_Y: the pass transforms y of [slab][ny][nx] in place: point p is the row i, q the column j,
                         // and the trend coefficients are per column: coef[(slab * nx + j) * 6]
};

struct Epilogue {  // last pass: xrft.py:446-472, 740-748, 825-833, 993-1004
    int mode;      // xrfthip_out_mode
    int p_axis;    // 0: this pass transforms x (1-D): kx = k_m, ky = 0;  1: it transforms y: ky = k_m, kx = q
    long long odiv;  // slab b = o / odiv, r = o % odiv
    long long r_mul, q_mul, p_mul;  // k_m = r*r_mul + q*q_mul + p*p_mul  (four-step final passes interleave k1 + n1*k2)
    int ny, nx, nx_out;
    int mirror;    // real input, full output: also store the Hermitian mirror (ny-ky, nx-kx)
    int shift_y, shift_x;
    int realdim_x2;
    int conj_out, real_out;  // inverse transforms: conjugate the result / store only its real part
    long long slab_stride, row_stride;  // output strides in elements
    double scale;
    void* out;  // may be null (iso only)
    const void* ph_y;
    const void* ph_x;
    const void* other;  // CROSS: raw F0, complex [slab][ny][nxh], unshifted
    long long other_slab_stride, other_row_stride;
};

__device__ __forceinline__ int phys(int pos, int sh) { return pos + (pos >> sh); }

// x / d for 0 <= x < 2^22 with inv = 1.0f / d: (x + 0.5) / d is at least 0.5 / d away from an integer and the float
// evaluation is off by at most (x + 0.5) / d * 2^-23, so the truncation is exact.  Every per-element index of a tile
// (< 20 K elements) goes through this instead of a ~30-instruction integer division by a runtime divisor.
__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

__device__ __forceinline__ int map_src(int m, int n, int flip, int ishift) {
    int t = m;
    if (ishift) { t += n / 2; if (t >= n) t -= n; }  // ifftshift(y)[m] = y[(m + n//2) % n]
    if (flip) t = n - 1 - t;
    return t;
}
__device__ __forceinline__ int shift_dst(int k, int n, int shift) {
    if (!shift) return k;
    int t = k + n / 2;  // fftshift(f)[(k + n//2) % n] = f[k]
    return t >= n ? t - n : t;
}

// ------------------------------------------------------------------------------------------------
// first-pass element fetch: source value -> detrend -> window      (one real or complex sample)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ C2<T> fetch_src(const Prologue& pr, long long b, int i, int j) {
    bool herm = false;
    if (pr.herm_nxh > 0 && j >= pr.herm_nxh) {  // irfftn: F[ky, kx] = conj(F[-ky, nx - kx]) for the half that is not stored
        j = pr.nx - j;
        i = i == 0 ? 0 : pr.ny - i;
        herm = true;
    }
    const int si = map_src(i, pr.ny, pr.flip_y, pr.ishift_y);
    const int sj = map_src(j, pr.herm_nxh > 0 ? pr.herm_nxh : pr.nx, pr.flip_x, pr.ishift_x);
    const long long off = b * pr.slab_stride + (long long)si * pr.row_stride + sj;
    C2<T> v;
    if (pr.in_complex) v = reinterpret_cast<const C2<T>*>(pr.in)[off];
    else v = mk<T>(reinterpret_cast<const T*>(pr.in)[off], (T)0);
    if (pr.detrend) {
        const double* c = pr.coef + (pr.p_is_row ? b * pr.nx + sj : b) * 6;
        v.re = (T)((double)v.re - (c[0] + c[2] * si + c[4] * sj));
        if (pr.in_complex) v.im = (T)((double)v.im - (c[1] + c[3] * si + c[5] * sj));
    }
    T w = (T)1;
    if (pr.win_y) w = reinterpret_cast<const T*>(pr.win_y)[si];
    if (pr.win_x) w *= reinterpret_cast<const T*>(pr.win_x)[sj];
    if (pr.win_y || pr.win_x) v = cscale(v, w);
    if (pr.ph_y) v = cmul(v, reinterpret_cast<const C2<T>*>(pr.ph_y)[si]);
    if (pr.ph_x) v = cmul(v, reinterpret_cast<const C2<T>*>(pr.ph_x)[sj]);
    if (herm != (pr.conj_in != 0)) v.im = -v.im;
    return v;
}

// ------------------------------------------------------------------------------------------------
// last-pass store of one frequency sample V = F(b, ky, kx) (unshifted indices)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void emit(const Epilogue& ep, long long b, int ky, int kx, C2<T> V, bool conj_it) {
    // V is F (COMPLEX), or F0 conj(F1) (CROSS), or (|F|^2, 0) (POWER), before phase and scale
    if (conj_it) V.im = -V.im;
    if (ep.mode != 1) {
        if (ep.ph_y) V = cmul(V, reinterpret_cast<const C2<T>*>(ep.ph_y)[ky]);
        if (ep.ph_x) V = cmul(V, reinterpret_cast<const C2<T>*>(ep.ph_x)[kx]);
    }
    T s = (T)ep.scale;
    if (ep.realdim_x2 && !(kx == 0 || ((ep.nx & 1) == 0 && kx == ep.nx / 2))) s *= (T)2;
    V = cscale(V, s);
    if (ep.conj_out) V.im = -V.im;
    if (ep.out) {
        const long long off = b * ep.slab_stride + (long long)shift_dst(ky, ep.ny, ep.shift_y) * ep.row_stride + shift_dst(kx, ep.nx, ep.shift_x);
        if (ep.mode == 3) reinterpret_cast<T*>(ep.out)[off] = (T)atan2((double)V.im, (double)V.re);  // np.angle
        else if (ep.mode == 1 || ep.real_out) reinterpret_cast<T*>(ep.out)[off] = V.re;
        else reinterpret_cast<C2<T>*>(ep.out)[off] = V;
    }
}

// (b0, r0) = (o0 / odiv, o0 % odiv) of the tile's first sequence, dt = sequence offset inside the tile: the only division
// per element is a 32-bit one (64-bit integer division costs >100 instructions on the GPU)
template <typename T>
__device__ __forceinline__ void epi_store(const Epilogue& ep, long long b0, int r0, int dt, int q, int p, C2<T> F) {
    unsigned rr = (unsigned)(r0 + dt), db = 0;
    const unsigned od = (unsigned)ep.odiv;
    if (od == 1) { db = rr; rr = 0; } else { while (rr >= od) { rr -= od; ++db; } }  // r0 < od, dt < T: a step or two
    const long long b = b0 + db;
    const int r = (int)rr;
    const int km = (int)(r * (int)ep.r_mul + q * (int)ep.q_mul + p * (int)ep.p_mul);
    int ky, kx;
    if (ep.p_axis) { ky = km; kx = (int)q; } else { ky = 0; kx = km; }
    if (kx >= ep.nx_out && !ep.mirror) return;  // full-complex path of a real input with HALF_X
    C2<T> V;
    if (ep.mode == 0) V = F;
    else if (ep.mode == 1) V = mk<T>(F.re * F.re + F.im * F.im, (T)0);
    else {  // CROSS (2) and PHASE (3)
        const C2<T> G = reinterpret_cast<const C2<T>*>(ep.other)[b * ep.other_slab_stride + (long long)ky * ep.other_row_stride + kx];
        V = cmulc(G, F);  // F0 * conj(F1): `other` holds field 0, this pass transforms field 1
    }
    emit<T>(ep, b, ky, kx, V, false);
    if (ep.mirror && kx > 0 && kx < ep.nx - kx) {
        const int my = ky == 0 ? 0 : ep.ny - ky;
        emit<T>(ep, b, my, ep.nx - kx, V, true);
    }
}

// ------------------------------------------------------------------------------------------------
// radix passes
// ------------------------------------------------------------------------------------------------
template <typename T, int R, typename TWP>
__device__ __forceinline__ void run_pass(C2<T>* tile, const TileGeom& g, int L, int tid, int nthreads, TWP tw) {
    const int m = L / R;
    const int per_seq = g.n / R;
    const int nb = g.T * per_seq;
    const int twstep = g.n / L;
    const float inv_ps = 1.0f / (float)per_seq, inv_m = 1.0f / (float)m;
    for (int w = tid; w < nb; w += nthreads) {
        const int t = fdiv(w, inv_ps);
        const int gg = w - t * per_seq;
        const int blk = fdiv(gg, inv_m);
        const int j = gg - blk * m;
        C2<T>* s = tile + t * g.seq_stride;
        const int base = blk * L + j;
        C2<T> a[R];
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = s[phys(base + q * m, g.pad_shift)];
        dft_r<T, R>(a);
        if (m > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) a[k] = cmul(a[k], tw[j * k * twstep]);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) s[phys(base + k * m, g.pad_shift)] = a[k];
    }
}

// exact inverse of run_pass up to the factor R (used by the Bluestein convolution, whose transform length is 2^a 3^b 5^c): undo the twiddles with their conjugates, then the unnormalised inverse butterfly
template <typename T, int R, typename TWP>
__device__ __forceinline__ void run_pass_inv(C2<T>* tile, const TileGeom& g, int L, int tid, int nthreads, TWP tw) {
    const int m = L / R;
    const int per_seq = g.n / R;
    const int nb = g.T * per_seq;
    const int twstep = g.n / L;
    const float inv_ps = 1.0f / (float)per_seq, inv_m = 1.0f / (float)m;
    for (int w = tid; w < nb; w += nthreads) {
        const int t = fdiv(w, inv_ps);
        const int gg = w - t * per_seq;
        const int blk = fdiv(gg, inv_m);
        const int j = gg - blk * m;
        C2<T>* s = tile + t * g.seq_stride;
        const int base = blk * L + j;
        C2<T> a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = cconj(s[phys(base + k * m, g.pad_shift)]);
        if (m > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) a[k] = cmul(a[k], tw[j * k * twstep]);  // conj(a conj(w)) = conj(a) w
        }
        dft_r<T, R>(a);
#pragma unroll
        for (int q = 0; q < R; ++q) s[phys(base + q * m, g.pad_shift)] = cconj(a[q]);
    }
}

// any radix up to XRFTHIP_MAX_RADIX (O(R^2) butterfly; operands live in scratch) -- odd lengths only
template <typename T>
__device__ void run_pass_generic(C2<T>* tile, const TileGeom& g, int R, int L, int tid, int nthreads) {
    const int m = L / R;
    const int per_seq = g.n / R;
    const int nb = g.T * per_seq;
    const int twstep = g.n / L;
    const int rstep = g.n / R;
    const C2<T>* __restrict__ tw = reinterpret_cast<const C2<T>*>(g.tw);
    const float inv_ps = 1.0f / (float)per_seq, inv_m = 1.0f / (float)m;
    C2<T> a[128];
    for (int w = tid; w < nb; w += nthreads) {
        const int t = fdiv(w, inv_ps);
        const int gg = w - t * per_seq;
        const int blk = fdiv(gg, inv_m);
        const int j = gg - blk * m;
        C2<T>* s = tile + t * g.seq_stride;
        const int base = blk * L + j;
        for (int q = 0; q < R; ++q) a[q] = s[phys(base + q * m, g.pad_shift)];
        for (int k = 0; k < R; ++k) {
            C2<T> acc = a[0];
            int idx = 0;
            for (int q = 1; q < R; ++q) {
                idx += k; if (idx >= R) idx -= R;
                acc = acc + cmul(a[q], tw[(long long)idx * rstep]);
            }
            if (m > 1 && k > 0) acc = cmul(acc, tw[(long long)j * k * twstep]);
            s[phys(base + k * m, g.pad_shift)] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// MAXT: the largest block the instantiation is launched with -- 512 leaves the register allocator 256 VGPRs per lane
// (the float64 radix-16 butterfly alone holds 64), 1024 caps it at 128
// PATH: 0 = every loader and epilogue (the general instantiation).  1..4 = instantiations that contain only the lean code
// of one pass kind, chosen by the host when all of its preconditions hold at plan time -- smaller code, no spills, deeper
// load batches:  1 row first pass (lean row loader; generic store)   2 column last pass (tiled loader, lean epilogue)
//                3 four-step first pass (lean column loader + store)  4 four-step last pass along x (lean epilogue)
template <typename T, bool FIRST, bool FINAL, bool GENERIC, int MAXT, int PATH>
__global__ void __launch_bounds__(MAXT) tile_fft_kernel(TileGeom g, Prologue pr, Epilogue ep) {
    constexpr bool ALL = PATH == 0;
    XRFT_DYN_SMEM(smem_raw);
    C2<T>* tile = reinterpret_cast<C2<T>*>(smem_raw);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const C2<T>* __restrict__ gin = reinterpret_cast<const C2<T>*>(g.in);
    C2<T>* __restrict__ gout = reinterpret_cast<C2<T>*>(g.out);
    const C2<T>* __restrict__ twg = reinterpret_cast<const C2<T>*>(g.tw);
    C2<T>* twl = nullptr;
    unsigned* revl = nullptr;
    if (g.tw_lds || g.rev_lds) {
        size_t off = (size_t)g.T * g.seq_stride * sizeof(C2<T>);
        off = (off + 15) & ~(size_t)15;
        off = (off + 15) & ~(size_t)15;
        if (g.tw_lds) {
            twl = reinterpret_cast<C2<T>*>(smem_raw + off);
            for (int i = tid; i < g.n; i += nthreads) twl[i] = twg[i];
            off += (size_t)g.n * sizeof(C2<T>);
            off = (off + 15) & ~(size_t)15;
        }
        if (g.rev_lds) {
            revl = reinterpret_cast<unsigned*>(smem_raw + off);
            for (int i = tid; i < g.n; i += nthreads) revl[i] = g.rev[i];
        }
    }

    for (long long tile_id = blockIdx.x; tile_id < g.n_tiles; tile_id += gridDim.x) {
        long long o0, q0;
        int tv;
        if (g.tile_axis == 0) {
            o0 = tile_id * g.T; q0 = 0;
            long long rem = g.n_outer - o0; tv = rem < g.T ? (int)rem : g.T;
        } else {
            o0 = tile_id / g.tiles_per_outer;
            q0 = (tile_id - o0 * g.tiles_per_outer) * g.T;
            long long rem = g.inner - q0; tv = rem < g.T ? (int)rem : g.T;
        }
        // per-tile 64-bit divisions (once); everything per element is 32-bit
        long long pb0 = 0, eb0 = 0;
        int pi0 = 0, er0 = 0;
        if (FIRST) { pb0 = o0 / pr.rows; pi0 = (int)(o0 - pb0 * pr.rows); }
        if (FINAL) { eb0 = o0 / ep.odiv; er0 = (int)(o0 - eb0 * ep.odiv); }
        // ------------------------------------------------------------------ load
        const int total_in = g.T * g.n;
        const float inv_n = 1.0f / (float)g.n, inv_T = 1.0f / (float)g.T;
        bool loaded = false;
        // everything that is constant along a row of a row tile -- slab, source row, trend at j = 0, trend slope, y window,
        // base of the row in the (tiled) intermediate -- is computed once per tile by T lanes and read back from LDS
        struct RowC { long long base, obase; double t0, t1, wy, pad_; };
        RowC* rc = reinterpret_cast<RowC*>(smem_raw + (g.rowc_off > 0 ? g.rowc_off : 0));
        if ((ALL || PATH == 1) && FIRST && g.rowc_off > 0) {
            for (int rt = tid; rt < g.T; rt += nthreads) {
                unsigned ii = (unsigned)(pi0 + rt), db = 0;
                const unsigned rws = (unsigned)pr.rows;
                if (rws == 1) { db = ii; ii = 0; } else { while (ii >= rws) { ii -= rws; ++db; } }
                const long long b = pb0 + db;
                const int si = map_src((int)ii, pr.ny, pr.flip_y, pr.ishift_y);
                RowC r;
                r.base = b * pr.slab_stride + (long long)si * pr.row_stride;
                r.obase = g.out_tiled ? b * g.til_slab + (long long)ii * g.out_tiled : 0;
                r.t0 = 0.0; r.t1 = 0.0; r.pad_ = 0.0;
                if (pr.detrend && rt < tv) { const double* c = pr.coef + b * 6; r.t0 = c[0] + c[2] * si; r.t1 = c[4]; }
                r.wy = pr.win_y ? (double)reinterpret_cast<const T*>(pr.win_y)[si] : 1.0;
                rc[rt] = r;
            }
            __syncthreads();
        }
        if ((PATH == 1) || (ALL && FIRST && g.rowc_off > 0 && pr.ph_y == nullptr && pr.ph_x == nullptr && !(g.dbg & 4))) {
            // lean loader for rows of real samples (xrft.py:425-442 without flips / input phases): per sample one
            // address, the trend FMA in float64, the x window
            const T* __restrict__ src = reinterpret_cast<const T*>(pr.in);
            const T* __restrict__ wx = reinterpret_cast<const T*>(pr.win_x);
            const C2<T>* __restrict__ chirp = reinterpret_cast<const C2<T>*>(g.blue_c);
            const int nx = pr.nx, hx = pr.ishift_x ? nx / 2 : 0, nlog = g.blue_n ? g.blue_n : g.n;
            const bool det = pr.detrend != 0, r2c = g.r2c != 0;
            constexpr int UR = ALL ? 4 : (sizeof(T) == 4 ? 8 : 4);
            for (int e0 = tid; e0 < total_in; e0 += UR * nthreads) {
                T x0[UR], x1[UR], w0[UR], w1[UR];
                int dst[UR], sj0[UR], sj1[UR], tt[UR], pp[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const int e = e0 + u * nthreads;
                    dst[u] = -1; tt[u] = -1; pp[u] = 0; sj0[u] = sj1[u] = 0;
                    x0[u] = x1[u] = (T)0; w0[u] = w1[u] = (T)1;
                    if (e < total_in) {
                        const int t = fdiv(e, inv_n), p = e - t * g.n;
                        dst[u] = t * g.seq_stride + phys(p, g.pad_shift);
                        if (t < tv && p < nlog) {
                            tt[u] = t; pp[u] = p;
                            int a = (r2c ? 2 * p : p) + hx;
                            if (a >= nx) a -= nx;
                            int b1 = a + 1;
                            if (b1 >= nx) b1 -= nx;
                            sj0[u] = a; sj1[u] = b1;
                            const long long base = rc[t].base;
                            x0[u] = src[base + a];
                            if (r2c) x1[u] = src[base + b1];
                            if (wx) { w0[u] = wx[a]; if (r2c) w1[u] = wx[b1]; }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    if (dst[u] < 0) continue;
                    C2<T> v = mk<T>((T)0, (T)0);
                    if (tt[u] >= 0) {
                        const RowC r = rc[tt[u]];
                        T a0 = x0[u], a1 = x1[u];
                        if (det) {
                            a0 = (T)((double)a0 - (r.t0 + r.t1 * (double)sj0[u]));
                            if (r2c) a1 = (T)((double)a1 - (r.t0 + r.t1 * (double)sj1[u]));
                        }
                        const T wy = (T)r.wy;
                        v = mk<T>(a0 * (w0[u] * wy), r2c ? a1 * (w1[u] * wy) : (T)0);
                        if (g.blue_n) v = cmulc(v, chirp[pp[u]]);
                    }
                    tile[dst[u]] = v;
                }
            }
            loaded = true;
        }
        // U independent elements per thread and trip: all their global loads are in flight before the first LDS store
        constexpr int U = 4;
        if ((PATH == 3) || (ALL && FIRST && (g.lean_col & 1) && pr.ph_x == nullptr && !(g.dbg & 4))) {
            // four-step first pass over a real 1-D series: the tile is T consecutive columns q of the row viewed as
            // [n1][n2]; a lane keeps its column, per sample: j = p n2 + q, one load, trend, window
            const int tsh = 31 - __builtin_clz((unsigned)g.T);
            const int t = tid & (g.T - 1);
            const long long q = q0 + t;
            const bool ok = t < tv;
            const T* __restrict__ src = reinterpret_cast<const T*>(pr.in) + o0 * pr.slab_stride;
            const T* __restrict__ wx = reinterpret_cast<const T*>(pr.win_x);
            const int nx = pr.nx, hx = pr.ishift_x ? nx / 2 : 0, nlog = g.blue_n ? g.blue_n : g.n;
            double t0 = 0.0, t1 = 0.0;
            if (pr.detrend) { const double* c = pr.coef + o0 * 6; t0 = c[0]; t1 = c[4]; }
            const int pstep = nthreads >> tsh, jq = (int)(q * pr.j_mul_q), jp = (int)pr.j_mul_p;
            constexpr int UC = 8;
            for (int p0 = tid >> tsh; p0 < g.n; p0 += UC * pstep) {
                T xv[UC], wv[UC];
                int sjv[UC];
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const int p = p0 + u * pstep;
                    xv[u] = (T)0; wv[u] = (T)1; sjv[u] = 0;
                    if (p < nlog && ok) {
                        int a = p * jp + jq + hx;
                        if (a >= nx) a -= nx;
                        sjv[u] = a;
                        xv[u] = src[a];
                        if (wx) wv[u] = wx[a];
                    }
                }
#pragma unroll
                for (int u = 0; u < UC; ++u) {
                    const int p = p0 + u * pstep;
                    if (p >= g.n) continue;
                    C2<T> v = mk<T>((T)0, (T)0);
                    if (p < nlog && ok) {
                        T a0 = xv[u];
                        if (pr.detrend) a0 = (T)((double)a0 - (t0 + t1 * (double)sjv[u]));
                        v = mk<T>(a0 * wv[u], (T)0);
                        if (g.blue_n) v = cmulc(v, reinterpret_cast<const C2<T>*>(g.blue_c)[p]);
                    }
                    tile[t * g.seq_stride + phys(p, g.pad_shift)] = v;
                }
            }
            loaded = true;
        }
        if ((PATH == 2) || (ALL && !FIRST && g.in_tiled && !(g.dbg & 4))) {  // one contiguous block: element e of the tile is element e of the block
            const int tsh = 31 - __builtin_clz((unsigned)g.T);
            const C2<T>* __restrict__ blk = gin + o0 * g.til_slab + (q0 >> tsh) * g.til_stride;
            const int nlog = g.blue_n ? g.blue_n : g.n;
            const int tot = nlog * g.T;
            constexpr int UT = 8;
            for (int e0 = tid; e0 < tot; e0 += UT * nthreads) {
                C2<T> vv[UT];
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    const int e = e0 + u * nthreads;
                    vv[u] = mk<T>((T)0, (T)0);
                    if (e < tot && (e & (g.T - 1)) < tv) vv[u] = blk[e];
                }
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    const int e = e0 + u * nthreads;
                    if (e < tot) {
                        const int p = e >> tsh, t = e & (g.T - 1);
                        C2<T> v = vv[u];
                        if (g.blue_n) v = cmulc(v, reinterpret_cast<const C2<T>*>(g.blue_c)[p]);
                        tile[t * g.seq_stride + phys(p, g.pad_shift)] = v;
                    }
                }
            }
            if (g.blue_n) {  // zero padding of the Bluestein transform
                for (int e = tot + tid; e < total_in; e += nthreads) {
                    const int p = e >> tsh, t = e & (g.T - 1);
                    tile[t * g.seq_stride + phys(p, g.pad_shift)] = mk<T>((T)0, (T)0);
                }
            }
            loaded = true;
        }
        if constexpr (ALL || PATH == 4)
        for (int e0 = tid; e0 < ((loaded || (g.dbg & 4)) ? 0 : total_in); e0 += U * nthreads) {
            C2<T> vv[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * nthreads;
                vv[u] = mk<T>((T)0, (T)0);
                dst[u] = -1;
                if (e < total_in) {
                    int t, p;
                    if (g.in_fast == 0) { t = fdiv(e, inv_n); p = e - t * g.n; } else { p = fdiv(e, inv_T); t = e - p * g.T; }
                    dst[u] = t * g.seq_stride + phys(p, g.pad_shift);
                    if (t < tv && (g.blue_n == 0 || p < g.blue_n)) {
                        const long long o = g.tile_axis == 0 ? o0 + t : o0;
                        const long long q = g.tile_axis == 0 ? 0 : q0 + t;
                        C2<T> v;
                        if (FIRST) {
                            unsigned ii = (unsigned)(pi0 + (g.tile_axis == 0 ? t : 0)), db = 0;
                            const unsigned rws = (unsigned)pr.rows;
                            if (rws == 1) { db = ii; ii = 0; } else { while (ii >= rws) { ii -= rws; ++db; } }  // t < T: a step or two
                            const long long b = pb0 + db;
                            const int i = (int)ii;
                            if (g.r2c) {
                                const C2<T> x0 = fetch_src<T>(pr, b, i, 2 * p);
                                const C2<T> x1 = fetch_src<T>(pr, b, i, 2 * p + 1);
                                v = mk<T>(x0.re, x1.re);
                            } else {
                                v = pr.p_is_row ? fetch_src<T>(pr, b, p, (int)q) : fetch_src<T>(pr, b, i, (int)(p * pr.j_mul_p + q * pr.j_mul_q));
                            }
                        } else {
                            v = gin[o * g.in_so + q * g.in_sq + (long long)p * g.in_sp];
                        }
                        if (g.blue_n) v = cmulc(v, reinterpret_cast<const C2<T>*>(g.blue_c)[p]);
                        vv[u] = v;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) tile[dst[u]] = vv[u];
        }
        __syncthreads();
        // ------------------------------------------------------------------ in-place DIF passes
        int L = g.n;
        for (int ip = 0; ip < ((g.dbg & 1) ? 0 : g.nr); ++ip) {
            const int R = g.radix[ip];
            if (twl) {
                const C2<T>* tw = twl;
                switch (R) {
                    case 2: run_pass<T, 2>(tile, g, L, tid, nthreads, tw); break;
                    case 3: run_pass<T, 3>(tile, g, L, tid, nthreads, tw); break;
                    case 4: run_pass<T, 4>(tile, g, L, tid, nthreads, tw); break;
                    case 5: run_pass<T, 5>(tile, g, L, tid, nthreads, tw); break;
                    case 6: run_pass<T, 6>(tile, g, L, tid, nthreads, tw); break;
                    case 8: run_pass<T, 8>(tile, g, L, tid, nthreads, tw); break;
                    case 9: run_pass<T, 9>(tile, g, L, tid, nthreads, tw); break;
                    case 10: run_pass<T, 10>(tile, g, L, tid, nthreads, tw); break;
                    case 12: run_pass<T, 12>(tile, g, L, tid, nthreads, tw); break;
                    case 15: run_pass<T, 15>(tile, g, L, tid, nthreads, tw); break;
                    case 16: run_pass<T, 16>(tile, g, L, tid, nthreads, tw); break;
                    default:
                        if (GENERIC) run_pass_generic<T>(tile, g, R, L, tid, nthreads);
                        break;
                }
            } else {
                switch (R) {
                    case 2: run_pass<T, 2>(tile, g, L, tid, nthreads, twg); break;
                    case 3: run_pass<T, 3>(tile, g, L, tid, nthreads, twg); break;
                    case 4: run_pass<T, 4>(tile, g, L, tid, nthreads, twg); break;
                    case 5: run_pass<T, 5>(tile, g, L, tid, nthreads, twg); break;
                    case 6: run_pass<T, 6>(tile, g, L, tid, nthreads, twg); break;
                    case 8: run_pass<T, 8>(tile, g, L, tid, nthreads, twg); break;
                    case 9: run_pass<T, 9>(tile, g, L, tid, nthreads, twg); break;
                    case 10: run_pass<T, 10>(tile, g, L, tid, nthreads, twg); break;
                    case 12: run_pass<T, 12>(tile, g, L, tid, nthreads, twg); break;
                    case 15: run_pass<T, 15>(tile, g, L, tid, nthreads, twg); break;
                    case 16: run_pass<T, 16>(tile, g, L, tid, nthreads, twg); break;
                    default:
                        if (GENERIC) run_pass_generic<T>(tile, g, R, L, tid, nthreads);
                        break;
                }
            }
            L /= R;
            __syncthreads();
        }
        if (g.blue_n) {  // circular convolution with the chirp: * B, inverse passes in reverse order, * conj(c)
            const C2<T>* __restrict__ bh = reinterpret_cast<const C2<T>*>(g.blue_b);
            const C2<T>* __restrict__ ch = reinterpret_cast<const C2<T>*>(g.blue_c);
            for (int e = tid; e < total_in; e += nthreads) {
                const int t = fdiv(e, inv_n), p = e - t * g.n;
                C2<T>* x = tile + t * g.seq_stride + phys(p, g.pad_shift);
                *x = cmul(*x, bh[p]);
            }
            __syncthreads();
            int Li = 1;
            for (int ip = g.nr - 1; ip >= 0; --ip) {
                const int R = g.radix[ip];
                Li *= R;
                if (twl) {
                    const C2<T>* tw = twl;
                    switch (R) {
                        case 2: run_pass_inv<T, 2>(tile, g, Li, tid, nthreads, tw); break;
                        case 3: run_pass_inv<T, 3>(tile, g, Li, tid, nthreads, tw); break;
                        case 4: run_pass_inv<T, 4>(tile, g, Li, tid, nthreads, tw); break;
                        case 5: run_pass_inv<T, 5>(tile, g, Li, tid, nthreads, tw); break;
                        case 6: run_pass_inv<T, 6>(tile, g, Li, tid, nthreads, tw); break;
                        case 8: run_pass_inv<T, 8>(tile, g, Li, tid, nthreads, tw); break;
                        case 9: run_pass_inv<T, 9>(tile, g, Li, tid, nthreads, tw); break;
                        case 10: run_pass_inv<T, 10>(tile, g, Li, tid, nthreads, tw); break;
                        case 12: run_pass_inv<T, 12>(tile, g, Li, tid, nthreads, tw); break;
                        case 15: run_pass_inv<T, 15>(tile, g, Li, tid, nthreads, tw); break;
                        default: run_pass_inv<T, 16>(tile, g, Li, tid, nthreads, tw); break;
                    }
                } else {
                    switch (R) {
                        case 2: run_pass_inv<T, 2>(tile, g, Li, tid, nthreads, twg); break;
                        case 3: run_pass_inv<T, 3>(tile, g, Li, tid, nthreads, twg); break;
                        case 4: run_pass_inv<T, 4>(tile, g, Li, tid, nthreads, twg); break;
                        case 5: run_pass_inv<T, 5>(tile, g, Li, tid, nthreads, twg); break;
                        case 6: run_pass_inv<T, 6>(tile, g, Li, tid, nthreads, twg); break;
                        case 8: run_pass_inv<T, 8>(tile, g, Li, tid, nthreads, twg); break;
                        case 9: run_pass_inv<T, 9>(tile, g, Li, tid, nthreads, twg); break;
                        case 10: run_pass_inv<T, 10>(tile, g, Li, tid, nthreads, twg); break;
                        case 12: run_pass_inv<T, 12>(tile, g, Li, tid, nthreads, twg); break;
                        case 15: run_pass_inv<T, 15>(tile, g, Li, tid, nthreads, twg); break;
                        default: run_pass_inv<T, 16>(tile, g, Li, tid, nthreads, twg); break;
                    }
                }
                __syncthreads();
            }
            const int tot_b = g.T * g.blue_n;
            const float inv_bn = 1.0f / (float)g.blue_n;
            for (int e = tid; e < tot_b; e += nthreads) {
                const int t = fdiv(e, inv_bn), k = e - t * g.blue_n;
                C2<T>* x = tile + t * g.seq_stride + phys(k, g.pad_shift);
                *x = cmulc(*x, ch[k]);
            }
            __syncthreads();
        }
        // ------------------------------------------------------------------ store
        const int nl = g.blue_n ? g.blue_n : g.n;  // logical transform length
        bool stored = false;
        if ((PATH == 3) || (ALL && !FINAL && (g.lean_col & 2) && !(g.dbg & 2))) {
            // four-step first pass, store: W2[o][k n2 + q] = F[k] W_N^(q k); the lane's column q is fixed
            const int tsh = 31 - __builtin_clz((unsigned)g.T);
            const int t = tid & (g.T - 1);
            const long long q = q0 + t;
            const bool ok = t < tv;
            const unsigned a = ((unsigned)q / (unsigned)g.tw_qdiv) % (unsigned)g.tw_qmod;
            const C2<T>* __restrict__ big = reinterpret_cast<const C2<T>*>(g.tw_big);
            C2<T>* __restrict__ dstc = gout + o0 * g.out_so + q * g.out_sq;
            const C2<T>* s = tile + t * g.seq_stride;
            const int kstep = nthreads >> tsh;
            constexpr int VS = ALL ? 4 : 8;  // (16 spills in the 128-VGPR float32 instantiation)
            for (int k0 = tid >> tsh; k0 < nl; k0 += VS * kstep) {
                C2<T> FF[VS], WW[VS];
#pragma unroll
                for (int u = 0; u < VS; ++u) {
                    const int k = k0 + u * kstep;
                    FF[u] = mk<T>((T)0, (T)0); WW[u] = mk<T>((T)1, (T)0);
                    if (k < nl && ok) {
                        const int pk = g.blue_n ? k : (revl ? (int)revl[k] : (int)g.rev[k]);
                        FF[u] = s[phys(pk, g.pad_shift)];
                        WW[u] = big[(long long)a * k];
                    }
                }
#pragma unroll
                for (int u = 0; u < VS; ++u) {
                    const int k = k0 + u * kstep;
                    if (k < nl && ok) dstc[(long long)k * g.out_sp] = cmul(FF[u], WW[u]);
                }
            }
            stored = true;
        }
        if ((PATH == 4) || (ALL && FINAL && g.lean_final == 2 && ep.out != nullptr && !(g.dbg & 2))) {
            // last pass of a four-step transform along x (1-D): kx = q + p_mul k with the lane's q fixed; no mirror (the
            // four-step path transforms real input as complex), row = slab
            const int tsh = 31 - __builtin_clz((unsigned)g.T);
            const int t = tid & (g.T - 1);
            const int kq = (int)(q0 + t) * (int)ep.q_mul, pm = (int)ep.p_mul;
            const bool ok = t < tv;
            const bool cplx = ep.mode == 0;
            const C2<T>* __restrict__ phx = reinterpret_cast<const C2<T>*>(ep.ph_x);
            const long long row_off = eb0 * ep.slab_stride;
            const C2<T>* s = tile + t * g.seq_stride;
            const int kstep = nthreads >> tsh;
            constexpr int VL = ALL ? 4 : 8;
            for (int k0 = tid >> tsh; k0 < nl; k0 += VL * kstep) {
                C2<T> FF[VL], PH[VL];
#pragma unroll
                for (int u = 0; u < VL; ++u) {
                    const int k = k0 + u * kstep;
                    FF[u] = mk<T>((T)0, (T)0); PH[u] = mk<T>((T)1, (T)0);
                    if (k < nl && ok) {
                        const int pk = g.blue_n ? k : (revl ? (int)revl[k] : (int)g.rev[k]);
                        FF[u] = s[phys(pk, g.pad_shift)];
                        if (cplx && phx) PH[u] = phx[kq + pm * k];
                    }
                }
#pragma unroll
                for (int u = 0; u < VL; ++u) {
                    const int k = k0 + u * kstep;
                    if (k >= nl || !ok) continue;
                    const int kx = kq + pm * k;
                    if (kx >= ep.nx_out) continue;
                    T sc = (T)ep.scale;
                    if (ep.realdim_x2 && !(kx == 0 || ((ep.nx & 1) == 0 && kx == ep.nx / 2))) sc *= (T)2;
                    const long long off = row_off + shift_dst(kx, ep.nx, ep.shift_x);
                    if (cplx) reinterpret_cast<C2<T>*>(ep.out)[off] = cscale(cmul(FF[u], PH[u]), sc);
                    else reinterpret_cast<T*>(ep.out)[off] = (FF[u].re * FF[u].re + FF[u].im * FF[u].im) * sc;
                }
            }
            stored = true;
        }
        if ((PATH == 2) || (ALL && FINAL && g.lean_final == 1 && ep.out != nullptr && !(g.dbg & 2))) {
            // lean epilogue of a plain column pass (xrft.py:446-472, 740-748): T is a power of two dividing the block size,
            // so a lane keeps its column for the whole tile -- column index, shifted destination column, mirror column,
            // x phase factors and the scale are per-lane constants; per sample: one LDS read, the y factors, two stores.
            const int tsh = 31 - __builtin_clz((unsigned)g.T);
            const int t = tid & (g.T - 1), kx = (int)q0 + t;
            const bool col_ok = t < tv && (kx < ep.nx_out || ep.mirror);
            const bool has_m = ep.mirror && kx > 0 && kx < ep.nx - kx;
            const bool direct_ok = kx < ep.nx_out;
            const int mkx = ep.nx - kx;
            const long long dcol = shift_dst(kx < ep.nx ? kx : 0, ep.nx, ep.shift_x), mcol = has_m ? shift_dst(mkx, ep.nx, ep.shift_x) : 0;
            T sc = (T)ep.scale;
            if (ep.realdim_x2 && !(kx == 0 || ((ep.nx & 1) == 0 && kx == ep.nx / 2))) sc *= (T)2;
            const bool cplx = ep.mode == 0;
            const C2<T>* __restrict__ phy = reinterpret_cast<const C2<T>*>(ep.ph_y);
            const C2<T>* __restrict__ phx = reinterpret_cast<const C2<T>*>(ep.ph_x);
            C2<T> px = mk<T>(sc, (T)0), pxm = mk<T>(sc, (T)0);  // scale folded into the x factor
            if (cplx && phx && col_ok) {
                px = cscale(phx[kx < ep.nx ? kx : 0], sc);
                if (has_m) pxm = cscale(phx[mkx], sc);
            }
            const long long slab_off = eb0 * ep.slab_stride;
            const C2<T>* s = tile + t * g.seq_stride;
            const int kstep = nthreads >> tsh;
            constexpr int VL = ALL ? 4 : 8;
            for (int k0 = tid >> tsh; k0 < nl; k0 += VL * kstep) {
                C2<T> FF[VL];
#pragma unroll
                for (int u = 0; u < VL; ++u) {
                    const int k = k0 + u * kstep;
                    FF[u] = mk<T>((T)0, (T)0);
                    if (k < nl && col_ok) {
                        const int pk = g.blue_n ? k : (revl ? (int)revl[k] : (int)g.rev[k]);
                        FF[u] = s[phys(pk, g.pad_shift)];
                    }
                }
#pragma unroll
                for (int u = 0; u < VL; ++u) {
                    const int k = k0 + u * kstep;
                    if (k >= nl || !col_ok) continue;
                    const C2<T> F = FF[u];
                    const long long drow = slab_off + (long long)shift_dst(k, ep.ny, ep.shift_y) * ep.row_stride;
                    const int mk_ = k == 0 ? 0 : ep.ny - k;
                    const long long mrow = slab_off + (long long)shift_dst(mk_, ep.ny, ep.shift_y) * ep.row_stride;
                    if (cplx) {
                        C2<T> vd = cmul(F, px), vm = cmul(cconj(F), pxm);
                        if (phy) { vd = cmul(vd, phy[k]); if (has_m) vm = cmul(vm, phy[mk_]); }
                        if (direct_ok) reinterpret_cast<C2<T>*>(ep.out)[drow + dcol] = vd;
                        if (has_m) reinterpret_cast<C2<T>*>(ep.out)[mrow + mcol] = vm;
                    } else {
                        const T v = (F.re * F.re + F.im * F.im) * sc;
                        if (direct_ok) reinterpret_cast<T*>(ep.out)[drow + dcol] = v;
                        if (has_m) reinterpret_cast<T*>(ep.out)[mrow + mcol] = v;
                    }
                }
            }
            stored = true;
        }
        const int total_out = g.T * g.n_out;
        const float inv_no = 1.0f / (float)g.n_out;
        // V independent results per thread and trip: table lookups and LDS reads of all of them overlap
        constexpr int V = 4;
        if constexpr (ALL || PATH == 1)
        for (int e0 = tid; e0 < ((stored || (g.dbg & 2)) ? 0 : total_out); e0 += V * nthreads) {
            C2<T> FF[V];
            int tt[V], kk[V];
#pragma unroll
            for (int u = 0; u < V; ++u) {
                const int e = e0 + u * nthreads;
                tt[u] = -1; kk[u] = 0;
                FF[u] = mk<T>((T)0, (T)0);
                if (e < total_out) {
                    int t, k;
                    if (g.out_fast == 0) { t = fdiv(e, inv_no); k = e - t * g.n_out; } else { k = fdiv(e, inv_T); t = e - k * g.T; }
                    if (t < tv) {
                        tt[u] = t; kk[u] = k;
                        const C2<T>* s = tile + t * g.seq_stride;
                        if (g.r2c) {
                            const int ka = k == nl ? 0 : k;
                            const int kb = k == 0 ? 0 : nl - k;
                            int pa, pb;
                            if (g.blue_n) { pa = ka; pb = kb; }
                            else if (revl) { pa = (int)revl[ka]; pb = (int)revl[kb]; }
                            else { pa = (int)g.rev[ka]; pb = (int)g.rev[kb]; }
                            const C2<T> zk = s[phys(pa, g.pad_shift)];
                            const C2<T> zc = cconj(s[phys(pb, g.pad_shift)]);
                            const C2<T> E = cscale(zk + zc, (T)0.5);
                            const C2<T> O = cscale(mul_mi(zk - zc), (T)0.5);
                            FF[u] = E + cmul(reinterpret_cast<const C2<T>*>(g.tw_r2c)[k], O);
                        } else {
                            const int pk = g.blue_n ? k : (revl ? (int)revl[k] : (int)g.rev[k]);
                            FF[u] = s[phys(pk, g.pad_shift)];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < V; ++u) {
                if (tt[u] < 0) continue;
                const int t = tt[u], k = kk[u];
                C2<T> F = FF[u];
                const long long o = g.tile_axis == 0 ? o0 + t : o0;
                const long long q = g.tile_axis == 0 ? 0 : q0 + t;
                if (FINAL) {
                    epi_store<T>(ep, eb0, er0, g.tile_axis == 0 ? t : 0, (int)q, k, F);
                } else {
                    if (g.tw_big) {
                        const unsigned a = ((unsigned)q / (unsigned)g.tw_qdiv) % (unsigned)g.tw_qmod;
                        F = cmul(F, reinterpret_cast<const C2<T>*>(g.tw_big)[(long long)a * k]);  // a*k < bigN by construction
                    }
                    if (g.out_tiled) {
                        const int tsh = 31 - __builtin_clz((unsigned)g.out_tiled);
                        gout[rc[t].obase + (long long)(k >> tsh) * g.til_stride + (k & (g.out_tiled - 1))] = F;
                    } else {
                        gout[o * g.out_so + q * g.out_sq + (long long)k * g.out_sp] = F;
                    }
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace xrft

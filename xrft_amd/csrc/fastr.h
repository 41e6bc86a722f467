// fastr.h -- ONE pass over a long real float32 row: the whole 65536-point transform of a row on one CU (BASELINE.json configs[1]:
// xrft.dft along x of (1024, 65536) float32; reference xrft/xrft.py:237-250 -> fft :307-476, numpy.fft.fft at :439-447).
//
// Why: a 65536-sample float32 row is 256 KB = the packed complex sequence z[n] = x[2n] + i x[2n+1] of M = 32768 points = 32 complex
// values (64 VGPRs) in each of the 1024 threads of one workgroup: it FITS the register file of a CU (512 KB).  The four-step form of
// fasty.h moves 28 bytes per sample through memory (4 read + 8 written + 8 read + 8 written); this kernel moves the 12 algorithmic
// ones: the row is read once (coalesced 8-byte loads, 512 contiguous bytes per wave), transformed in registers, and the full
// complex64 spectrum (or |X|^2) leaves once, 512 contiguous, aligned bytes per wave-instruction.
//
//   M = 32 x 32 x 32, decimation in frequency, n = n1 + 32 n2 + 1024 n3, k = k1 + 32 k2 + 1024 k3 (all digits < 32):
//     stage 1   thread p = n1 + 32 n2 holds n3 = 0..31:   DFT32 over n3 -> k1,   x W_M^(p k1)
//     exchange  (n1 + 32 n2 | k1) -> (k1 + 32 n1 | n2)                                        [threads | registers]
//     stage 2   DFT32 over n2 -> k2,   x W_1024^(n1 k2)
//     exchange  (k1 + 32 n1 | k2) -> (k1 + 32 k2 | n1)
//     stage 3   DFT32 over n1 -> k3:   thread p = k1 + 32 k2 holds Z[p + 1024 k3], k3 = 0..31 -- natural order, lanes along k
//     split     X[k] = A - i W_N^k B,  X[k + M] = A + i W_N^k B,  A = (Z[k] + conj Z[M-k]) / 2,  B = (Z[k] - conj Z[M-k]) / 2:
//               Z[M-k] is register 31 - k3 of thread 1024 - p, fetched through the LDS; both results are stored by thread p
//               at p + 1024 k3 (+ M), so every store is lane-contiguous and 512-byte aligned, fftshift or not.
//   Each exchange moves the 256 KB of the row through the 160-KB LDS in two halves of 16 registers (136 KB with the padding that makes
//   every 8-byte access conflict-free: strides = 2 dwords mod 32 for the 16-lane groups of ds_write_b64, mod 64 for the 32-lane groups
//   of ds_read_b64).  Whole waves write, whole waves read (a half is the registers [16h, 16h + 16) of the READER, i.e. the threads
//   [512h, 512h + 512) of the writers).
//   Per-row mean / least-squares line (scipy.signal.detrend along the row, xrft/detrend.py:54-71) from float64 sums over the registers,
//   wave shuffles and one LDS table added in wave order (bit-reproducible); window multiply on the loaded samples; true-phase table,
//   scale, |X|^2, real_dim half output on the way out.
#pragma once
#include "fastp2.h"

namespace xrft {

struct FastR {
    const float* in;   // [rows][N] float32
    void* out;         // [rows][N] complex64 | float32 (half: [rows][N/2 + 1])
    const cf* tw_m;    // W_M^p,    p < 1024   (M = N / 2)
    const cf* tw_s;    // W_1024^n, n < 32
    const cf* tw_n;    // W_N^p,    p < 1024
    const float* win;  // N samples (null: none)
    const cf* ph;      // N factors by unshifted frequency index (true phase, times (-1)^k for an ifftshifted input); read when ph_on
    long long nrows;
    int detrend;       // 0 none, 1 constant, 2 linear
    int ph_on;
    int half;          // real_dim: k = 0..N/2 only, rows of N/2 + 1 samples, unshifted
    int realdim2;      // ... and 0 < k < N/2 counts twice (xrft.py:673-682)
    int shift;         // fftshift of the output (xrft.py:446-447)
    float scale;       // complex: multiplies X; power: multiplies |X|^2
    int inv;           // fastc_kernel (complex rows): the inverse transform -- conjugate in, conjugate out (xrft.ifft, xrft.py:586-621)
    int ishift;        // ... its input is rotated by n/2 on load (an fftshifted spectrum: xrft.py:612-617)
    int ph_in;         // ... `ph` multiplies the INPUT samples (by source index) instead of the output (the true-phase factor of xrft.ifft, xrft.py:596-606)
    int stagger;       // fastr_kernel: start delay of workgroup class c = (block / 8) % classes, c x (stagger & 0xff) x 3.4 us; classes = stagger >> 8 (0: none)
};

constexpr int kFastRThreads = 1024;
constexpr int kFastRLdsElems = 17440;                                // complex64 elements: 31 * 545 + 31 * 17 + 16 rounded up
constexpr size_t kFastRLds = (size_t)kFastRLdsElems * 8 + 16 * 2 * 8;  // + the detrend sums of 16 waves

// forward DFT of 32 points in registers, natural order in and out: two DFT16 over the even / odd samples and one radix-2 level
__device__ __forceinline__ void dft32f(cf* a) {
    cf e[16], o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { e[k] = a[2 * k]; o[k] = a[2 * k + 1]; }
    dft16<float>(e);
    dft16<float>(o);
    // cos / sin of 2 pi k / 32, k = 0..15
    constexpr float C[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                             0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f, -0.19509032201612826785f,
                             -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                             -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float S[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                             0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.0f, 0.98078528040323044913f,
                             0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                             0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        cf t;
        if (k == 0) t = o[0];
        else if (k == 8) t = mul_mi(o[8]);
        else t = mk<float>(C[k] * o[k].re + S[k] * o[k].im, C[k] * o[k].im - S[k] * o[k].re);  // o[k] W32^k
        a[k] = e[k] + t;
        a[k + 16] = e[k] - t;
    }
}

// a[k] *= w1^k, k = 1..31: powers w^(8 a + b) = w^(8 a) w^b from w1..w7, w8, w16, w24 (at most 7 roundings deep)
__device__ __forceinline__ void twiddle32f(cf* a, cf w1) {
    cf w[8];
    w[1] = w1; w[2] = cmul(w1, w1); w[3] = cmul(w[2], w1); w[4] = cmul(w[2], w[2]);
    w[5] = cmul(w[4], w1); w[6] = cmul(w[4], w[2]); w[7] = cmul(w[4], w[3]);
    const cf w8 = cmul(w[4], w[4]), w16 = cmul(w8, w8), w24 = cmul(w16, w8);
#pragma unroll
    for (int b = 1; b < 8; ++b) a[b] = cmul(a[b], w[b]);
    a[8] = cmul(a[8], w8); a[16] = cmul(a[16], w16); a[24] = cmul(a[24], w24);
#pragma unroll
    for (int b = 1; b < 8; ++b) {
        a[8 + b] = cmul(a[8 + b], cmul(w8, w[b]));
        a[16 + b] = cmul(a[16 + b], cmul(w16, w[b]));
        a[24 + b] = cmul(a[24 + b], cmul(w24, w[b]));
    }
}

__device__ __forceinline__ void fastr_store8(cf* dst, cf v) {
#ifdef XRFT_EMULATE
    *dst = v;
#else
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f t = {v.re, v.im};
    __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(dst));
#endif
}
__device__ __forceinline__ void fastr_store4(float* dst, float v) {
#ifdef XRFT_EMULATE
    *dst = v;
#else
    __builtin_nontemporal_store(v, dst);
#endif
}

// nothing may be scheduled across this point: bounds how many loads the compiler keeps in flight beside the row's 64 registers
__device__ __forceinline__ void fastr_sched_fence() {
#ifndef XRFT_EMULATE
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// The split and the stores of one half (see the kernel): own registers 31 - 16 h - q, q < 16, against the partner values in slots q, in batches
// of BQ (the partner values and, with PH, the true-phase factors of a batch live beside the row's 64 registers: 8 per batch spilled 20).
template <int MODE, bool HALF, bool PH, int BQ>
__device__ __forceinline__ void fastr_emit(const FastR& p, const cf* a, const cf* __restrict__ pl, cf wn, int h, int tid, size_t orow, int pos0, int pos1, float sc) {
    constexpr int N = 65536, M = N / 2, T = kFastRThreads;
    // W_64^k = cos - i sin of 2 pi k / 64, k < 32 (compile-time indices after unrolling)
    constexpr float C64[32] = {
        1.0000000000f, 0.9951847267f, 0.9807852804f, 0.9569403357f, 0.9238795325f, 0.8819212643f, 0.8314696123f, 0.7730104534f, 0.7071067812f,
        0.6343932842f, 0.5555702330f, 0.4713967368f, 0.3826834324f, 0.2902846773f, 0.1950903220f, 0.0980171403f, 0.0000000000f, -0.0980171403f,
        -0.1950903220f, -0.2902846773f, -0.3826834324f, -0.4713967368f, -0.5555702330f, -0.6343932842f, -0.7071067812f, -0.7730104534f,
        -0.8314696123f, -0.8819212643f, -0.9238795325f, -0.9569403357f, -0.9807852804f, -0.9951847267f};
    constexpr float S64[32] = {
        0.0000000000f, -0.0980171403f, -0.1950903220f, -0.2902846773f, -0.3826834324f, -0.4713967368f, -0.5555702330f, -0.6343932842f,
        -0.7071067812f, -0.7730104534f, -0.8314696123f, -0.8819212643f, -0.9238795325f, -0.9569403357f, -0.9807852804f, -0.9951847267f,
        -1.0000000000f, -0.9951847267f, -0.9807852804f, -0.9569403357f, -0.9238795325f, -0.8819212643f, -0.8314696123f, -0.7730104534f,
        -0.7071067812f, -0.6343932842f, -0.5555702330f, -0.4713967368f, -0.3826834324f, -0.2902846773f, -0.1950903220f, -0.0980171403f};
#pragma unroll
    for (int g = 0; g < 16 / BQ; ++g) {
        cf zm[BQ], f0[BQ], f1[BQ];
#pragma unroll
        for (int q = 0; q < BQ; ++q) zm[q] = pl[(BQ * g + q) * T];
        if (PH) {
#pragma unroll
            for (int q = 0; q < BQ; ++q) {
                const int k = tid + T * (31 - 16 * h - BQ * g - q);
                f0[q] = p.ph[k];
                if (!HALF) f1[q] = p.ph[k + M];
            }
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int k3 = 31 - 16 * h - BQ * g - q;      // own register paired with the partner's register 16 h + BQ g + q
            const cf z = a[k3];
            const cf A = mk<float>(z.re + zm[q].re, z.im - zm[q].im), B = mk<float>(z.re - zm[q].re, z.im + zm[q].im);  // 2A, 2B
            const cf w = k3 == 0 ? wn : cmul(wn, mk<float>(C64[k3], S64[k3]));  // W_N^k = W_N^tid W_64^k3
            const cf wb = cmul(w, B);
            const cf t = mk<float>(-wb.im, wb.re);  // i W B
            cf x0 = mk<float>(A.re - t.re, A.im - t.im), x1 = mk<float>(A.re + t.re, A.im + t.im);  // 2 X[k], 2 X[k + M]
            if (MODE == 1) {
                float* o = reinterpret_cast<float*>(p.out) + orow;
                // (real_dim: 0 < k < M counts twice; k = 0 and the Nyquist sample X[M] = x1 of k = 0 once: xrft.py:673-682)
                const float f = (HALF && p.realdim2 && (k3 != 0 || tid != 0)) ? 2.0f * sc : sc;
                fastr_store4(o + pos0 + T * k3, (x0.re * x0.re + x0.im * x0.im) * f);
                if (!HALF) fastr_store4(o + pos1 + T * k3, (x1.re * x1.re + x1.im * x1.im) * sc);
                else if (k3 == 0) { if (tid == 0) fastr_store4(o + M, (x1.re * x1.re + x1.im * x1.im) * sc); }
            } else {
                x0 = cscale(x0, sc); x1 = cscale(x1, sc);
                cf* o = reinterpret_cast<cf*>(p.out) + orow;
                if (PH) { x0 = cmul(x0, f0[q]); if (!HALF) x1 = cmul(x1, f1[q]); }
                fastr_store8(o + pos0 + T * k3, x0);
                if (!HALF) fastr_store8(o + pos1 + T * k3, x1);
                else if (k3 == 0) { if (tid == 0) fastr_store8(o + M, PH ? cmul(x1, p.ph[M]) : x1); }
            }
        }
        fastr_sched_fence();
    }
}

// MODE 0: complex spectrum (xrft.fft / dft), 1: power spectrum; HALF: real_dim output, k = 0..N/2
// start delay of workgroup class c = (block / 8) % classes: c x (stagger & 0xff) sleeps of 127 x 64 cycles (~3.4 us each); classes = stagger >> 8
__device__ __forceinline__ void fastr_stagger(int stagger) {
#ifndef XRFT_EMULATE
    if ((stagger >> 8) > 1) {
        const int cls = (int)((blockIdx.x >> 3) % (unsigned)(stagger >> 8)), n = cls * (stagger & 0xff);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
}

template <int MODE, bool HALF>
__global__ void __launch_bounds__(kFastRThreads) fastr_kernel(FastR p) {
    constexpr int N = 65536, M = N / 2, T = kFastRThreads;
    constexpr int A1 = 545, B1 = 17;  // exchange 1: element (k1, n1, n2') at k1 A1 + n1 B1 + n2' (8-byte elements)
    XRFT_DYN_SMEM(smem_raw);
    cf* L = reinterpret_cast<cf*>(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw + (size_t)kFastRLdsElems * 8);  // [16 waves][2]
#ifndef XRFT_EMULATE
    // One workgroup owns a CU and walks its rows load -> transform -> store with nothing overlapped, and all CUs start together: the chip
    // alternates between memory phases (every CU loading or storing) and a phase in which every CU computes and the memory idles.  Classes of
    // workgroups that start a fraction of a row period apart keep the memory busy while the others transform.
    fastr_stagger(p.stagger);
#endif
    for (long long row = blockIdx.x; row < p.nrows; row += gridDim.x) {
        // (everything derived from the thread index is re-derived per row from an opaque copy: hoisted out of the loop, the 62 twiddle
        // powers and the store offsets were 165 spilled registers)
        int tid = threadIdx.x;
        XRFT_OPAQUE(tid);
        const int lo = tid & 31, hi = tid >> 5;
        const cf* __restrict__ src = reinterpret_cast<const cf*>(p.in + (size_t)row * N) + tid;
        cf a[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) a[j] = src[j * T];  // z[n], n = tid + 1024 j: samples 2n, 2n + 1
        if (p.detrend) {
            // sums over the row of x and of (i - ibar) x in float64.  With u_j = x[2 n_j] + x[2 n_j + 1], n_j = tid + 1024 j and
            // c_j = 2 n_j - ibar = c_0 + 2048 j:   sum (i - ibar) x = c_0 sum u_j + 2048 sum j u_j + sum x[2 n_j + 1]  -- no per-sample
            // index is kept.  Wave shuffles, then the 16 wave sums in wave order: bit-reproducible.
            constexpr double IBAR = 0.5 * (N - 1);
            const double c0 = (double)(2 * tid) - IBAR;
            double U = 0.0, V = 0.0, I = 0.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const double u = (double)a[j].re + (double)a[j].im;
                U += u;
                V = fma((double)j, u, V);
                I += (double)a[j].im;
            }
            double s0 = U, s1 = fma(c0, U, fma(2048.0, V, I));
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); }
            if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s0; red[(tid >> 6) * 2 + 1] = s1; }
            __syncthreads();
            double t0 = 0.0, t1 = 0.0;  // (`red` is next written a row later, behind the barriers of the exchanges)
#pragma unroll
            for (int w = 0; w < T / 64; ++w) {
                t0 += red[2 * w]; t1 += red[2 * w + 1];
                if ((w & 3) == 3) fastr_sched_fence();  // (four waves' sums at a time: all 32 values at once are 64 registers beside the row's 64)
            }
            constexpr double INV_N = 1.0 / N, INV_SII = 12.0 / ((double)N * ((double)N * N - 1.0));
            const double slope = p.detrend == 2 ? t1 * INV_SII : 0.0;
            const double l0 = fma(slope, c0, t0 * INV_N), dl = 2048.0 * slope;  // the line at sample 2 n_j: l0 + dl j
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                XRFT_OPAQUE(a[j].re); XRFT_OPAQUE(a[j].im);  // (else the 64 float64 conversions of the first loop are kept, spilled, for this one)
                const double lj = fma(dl, (double)j, l0);
                a[j] = mk<float>((float)((double)a[j].re - lj), (float)((double)a[j].im - (lj + slope)));
            }
        }
        if (p.win) {  // two batches of 16 window pairs beside the 64 registers of the row
            const cf* __restrict__ wsrc = reinterpret_cast<const cf*>(p.win) + tid;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                cf w[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = wsrc[(16 * g + j) * T];
#pragma unroll
                for (int j = 0; j < 16; ++j) a[16 * g + j] = mk<float>(a[16 * g + j].re * w[j].re, a[16 * g + j].im * w[j].im);
                fastr_sched_fence();
            }
        }
        // ---- stage 1: over n3 -> k1
        dft32f(a);
        twiddle32f(a, p.tw_m[tid]);
        // ---- exchange 1: writer (n1 = lo, n2 = hi) registers k1  ->  reader (k1 = lo, n1 = hi) registers n2
        {
            cf b[32];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __syncthreads();
                if ((hi >> 4) == h) {
                    cf* dst = L + lo * B1 + (hi & 15);
#pragma unroll
                    for (int k = 0; k < 32; ++k) dst[k * A1] = a[k];
                }
                __syncthreads();
                const cf* s = L + lo * A1 + hi * B1;
#pragma unroll
                for (int q = 0; q < 16; ++q) b[16 * h + q] = s[q];
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k] = b[k];
        }
        // ---- stage 2: over n2 -> k2
        dft32f(a);
        twiddle32f(a, p.tw_s[hi]);
        // ---- exchange 2: writer (k1 = lo, n1 = hi) registers k2  ->  reader (k1 = lo, k2 = hi) registers n1
        {
            cf b[32];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __syncthreads();
                if ((hi >> 4) == h) {
                    cf* dst = L + lo * B1 + (hi & 15);
#pragma unroll
                    for (int k = 0; k < 32; ++k) dst[k * 32 * B1] = a[k];
                }
                __syncthreads();
                const cf* s = L + (hi * 32 + lo) * B1;
#pragma unroll
                for (int q = 0; q < 16; ++q) b[16 * h + q] = s[q];
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k] = b[k];
        }
        // ---- stage 3: over n1 -> k3.  a[k3] = Z[tid + 1024 k3]
        dft32f(a);
        // ---- split + store.  Partner Z[M - k]: register 31 - k3 of thread 1024 - tid (thread 0: its own register (32 - k3) mod 32, which
        // travels in an extra slot).  Half h carries registers [16 h, 16 h + 16) in slots 0..15 and pairs them with the readers' own
        // registers 31 - 16 h .. 16 - 16 h.
        constexpr int W = HALF ? M + 1 : N;
        const size_t orow = (size_t)row * W;
        const int pos0 = tid + ((p.shift && !HALF) ? M : 0), pos1 = tid + (p.shift ? 0 : M);
        const float sc = MODE == 1 ? 0.25f * p.scale : 0.5f * p.scale;
        const cf wn = p.tw_n[tid];
        const cf* __restrict__ pl = L + ((T - tid) & (T - 1)) + (tid == 0 ? T : 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) L[q * T + tid] = a[16 * h + q];
            if (tid == 0) L[16 * T] = a[(16 * h + 16) & 31];
            __syncthreads();
            if (MODE == 0 && p.ph_on) fastr_emit<MODE, HALF, true, 2>(p, a, pl, wn, h, tid, orow, pos0, pos1, sc);
            else fastr_emit<MODE, HALF, false, 4>(p, a, pl, wn, h, tid, orow, pos0, pos1, sc);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// The same for rows of N = 8192, 16384, 32768 samples: M = N / 2 = 32 R2 R3 packed points on T = R2 R3 = 128, 256, 512 threads, 32 per thread
// (R2, R3 = 16, 8 | 16, 16 | 32, 16).  The whole row fits the LDS, so every exchange is ONE trip (all write, barrier, all read), and 4 / 2 / 1
// workgroups share a CU (35 / 70 / 135 KB of LDS): below 32768 samples a CU's rows overlap their load, transform and store phases.
//   stage 1   thread p = a + R3 b holds n = p + T j, j < 32:      DFT32 over j -> k1,  x W_M^(p k1)
//   exchange  (a + R3 b | k1) -> (a + R3 c | s, b),  k1 = c + R2 s, s < 32 / R2           element (k1, p) at k1 (T + R3 mod 32) + p
//   stage 2   DFT_R2 over b -> k2,  x W_T^(a k2)                                           [u = s R2 + k2]
//   exchange  (a + R3 c | u) -> (d + R3 c | t, a),  u = d + R3 t, t < 32 / R3             element (u, a + R3 c) at u (T + 1) + a + R3 c
//   stage 3   DFT_R3 over a -> k3:  Z[k], k = k1 + 32 (k2 + R2 k3)
//   split     Z in natural order through the LDS (k + k / 32); thread v takes k = v + T m, m < 32, and its partner M - k:
//             X[k] = A - i W_N^k B,  X[k + M] = A + i W_N^k B,  W_N^k = W_N^v W_64^m  (N / T = 64 for every size) -- lane-contiguous stores.
// Every access is an 8-byte one with a lane stride that is conflict-free (1 element; T + R3 or T + 1 elements between the classes that share a
// 32-lane group), except the natural-order writes of the last trip (2-way).
// ------------------------------------------------------------------------------------------------------------------------------------------
template <int R2, int R3> struct R2Geom {
    static_assert((R2 == 32 && R3 == 16) || (R2 == 16 && R3 == 16) || (R2 == 16 && R3 == 8) || (R2 == 8 && R3 == 8), "32768, 16384, 8192 or 4096 samples");
    static constexpr int T = R2 * R3, M = 32 * T, N = 2 * M;
    static constexpr int K2 = 32 / R2, K3 = 32 / R3;
    static constexpr int S1 = T + (R3 % 32), S2 = T + 1;
    static constexpr size_t LDS_MAIN = (size_t)(32 * S1 > M + M / 32 ? 32 * S1 : M + M / 32) * 8;
    static constexpr int NW = T / 64;  // (>= 1: a 4096-sample row is one wave)
    static constexpr size_t LDS = LDS_MAIN + (size_t)NW * 2 * 8;
    static constexpr int WPS = 4;  // (113-123 registers, none spilled: four waves per SIMD)
};

template <int R2, int R3, int MODE, bool HALF>
__global__ void __launch_bounds__((R2Geom<R2, R3>::T), (R2Geom<R2, R3>::WPS)) fastr2_kernel(FastR p) {
    typedef R2Geom<R2, R3> G;
    constexpr int T = G::T, M = G::M, N = G::N, K2 = G::K2, K3 = G::K3, S1 = G::S1, S2 = G::S2;
    constexpr float C64[32] = {
        1.0000000000f, 0.9951847267f, 0.9807852804f, 0.9569403357f, 0.9238795325f, 0.8819212643f, 0.8314696123f, 0.7730104534f, 0.7071067812f,
        0.6343932842f, 0.5555702330f, 0.4713967368f, 0.3826834324f, 0.2902846773f, 0.1950903220f, 0.0980171403f, 0.0000000000f, -0.0980171403f,
        -0.1950903220f, -0.2902846773f, -0.3826834324f, -0.4713967368f, -0.5555702330f, -0.6343932842f, -0.7071067812f, -0.7730104534f,
        -0.8314696123f, -0.8819212643f, -0.9238795325f, -0.9569403357f, -0.9807852804f, -0.9951847267f};
    constexpr float S64[32] = {
        0.0000000000f, -0.0980171403f, -0.1950903220f, -0.2902846773f, -0.3826834324f, -0.4713967368f, -0.5555702330f, -0.6343932842f,
        -0.7071067812f, -0.7730104534f, -0.8314696123f, -0.8819212643f, -0.9238795325f, -0.9569403357f, -0.9807852804f, -0.9951847267f,
        -1.0000000000f, -0.9951847267f, -0.9807852804f, -0.9569403357f, -0.9238795325f, -0.8819212643f, -0.8314696123f, -0.7730104534f,
        -0.7071067812f, -0.6343932842f, -0.5555702330f, -0.4713967368f, -0.3826834324f, -0.2902846773f, -0.1950903220f, -0.0980171403f};
    XRFT_DYN_SMEM(smem_raw);
    cf* L = reinterpret_cast<cf*>(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw + G::LDS_MAIN);  // [waves][2]
    fastr_stagger(p.stagger);  // (a resident set walking the rows: XRFTHIP_FASTR_STAGGER / XRFTHIP_FASTR_GRID)
    for (long long row = blockIdx.x; row < p.nrows; row += gridDim.x) {
        int tid = threadIdx.x;
        XRFT_OPAQUE(tid);
        const cf* __restrict__ src = reinterpret_cast<const cf*>(p.in + (size_t)row * N) + tid;
        cf a[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) a[j] = src[j * T];  // z[n], n = tid + T j: samples 2n, 2n + 1
        if (p.detrend) {  // (as in fastr_kernel: c_j = 2 n_j - ibar = c_0 + 2 T j)
            constexpr double IBAR = 0.5 * (N - 1);
            const double c0 = (double)(2 * tid) - IBAR;
            double U = 0.0, V = 0.0, I = 0.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const double u = (double)a[j].re + (double)a[j].im;
                U += u;
                V = fma((double)j, u, V);
                I += (double)a[j].im;
            }
            double s0 = U, s1 = fma(c0, U, fma((double)(2 * T), V, I));
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); }
            if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s0; red[(tid >> 6) * 2 + 1] = s1; }
            __syncthreads();
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int w = 0; w < G::NW; ++w) { t0 += red[2 * w]; t1 += red[2 * w + 1]; }
            constexpr double INV_N = 1.0 / N, INV_SII = 12.0 / ((double)N * ((double)N * N - 1.0));
            const double slope = p.detrend == 2 ? t1 * INV_SII : 0.0;
            const double l0 = fma(slope, c0, t0 * INV_N), dl = (double)(2 * T) * slope;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                XRFT_OPAQUE(a[j].re); XRFT_OPAQUE(a[j].im);
                const double lj = fma(dl, (double)j, l0);
                a[j] = mk<float>((float)((double)a[j].re - lj), (float)((double)a[j].im - (lj + slope)));
            }
        }
        if (p.win) {
            const cf* __restrict__ wsrc = reinterpret_cast<const cf*>(p.win) + tid;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                cf w[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = wsrc[(16 * g + j) * T];
#pragma unroll
                for (int j = 0; j < 16; ++j) a[16 * g + j] = mk<float>(a[16 * g + j].re * w[j].re, a[16 * g + j].im * w[j].im);
                fastr_sched_fence();
            }
        }
        // ---- stage 1
        dft32f(a);
        twiddle32f(a, p.tw_m[tid]);
        // ---- exchange 1
        const int aa = tid % R3, cc = tid / R3;  // before: (a, b); after: (a, c)
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) L[k1 * S1 + tid] = a[k1];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < K2; ++s)
#pragma unroll
            for (int b = 0; b < R2; ++b) a[s * R2 + b] = L[(cc + R2 * s) * S1 + aa + R3 * b];
        // ---- stage 2: DFT_R2 over b, x W_T^(a k2)
        {
            const cf wt = p.tw_s[aa];  // W_T^a
            if (R2 == 32) { dft32f(a); twiddle32f(a, wt); }
            else if (R2 == 16) {
#pragma unroll
                for (int s = 0; s < K2; ++s) { dft16<float>(a + 16 * s); twiddle16<float>(a + 16 * s, wt); }
            } else {
                const cf w2 = cmul(wt, wt), w3 = cmul(w2, wt), w4 = cmul(w2, w2), w5 = cmul(w4, wt), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
#pragma unroll
                for (int s = 0; s < K2; ++s) {
                    cf* g = a + 8 * s;
                    dft8<float>(g);
                    g[1] = cmul(g[1], wt); g[2] = cmul(g[2], w2); g[3] = cmul(g[3], w3); g[4] = cmul(g[4], w4);
                    g[5] = cmul(g[5], w5); g[6] = cmul(g[6], w6); g[7] = cmul(g[7], w7);
                }
            }
        }
        // ---- exchange 2
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 32; ++u) L[u * S2 + tid] = a[u];
        __syncthreads();
        const int dd = tid % R3;  // after: (d, c), the same c
#pragma unroll
        for (int t = 0; t < K3; ++t)
#pragma unroll
            for (int e = 0; e < R3; ++e) a[t * R3 + e] = L[(dd + R3 * t) * S2 + e + R3 * cc];
        // ---- stage 3: DFT_R3 over a -> k3
#pragma unroll
        for (int t = 0; t < K3; ++t) dft_r<float, R3>(a + R3 * t);
        // ---- Z in natural order: register (t, k3) is k = k1 + 32 (k2 + R2 k3), u = d + R3 t = s R2 + k2, k1 = c + R2 s
        __syncthreads();
#pragma unroll
        for (int t = 0; t < K3; ++t) {
            const int u = dd + R3 * t, s2 = u / R2, k2 = u % R2, k1 = cc + R2 * s2;
#pragma unroll
            for (int k3 = 0; k3 < R3; ++k3) {
                const int k = k1 + 32 * (k2 + R2 * k3);
                L[k + (k >> 5)] = a[t * R3 + k3];
            }
        }
        __syncthreads();
        // ---- split + store: thread v takes k = v + T m
        constexpr int W = HALF ? M + 1 : N;
        const size_t orow = (size_t)row * W;
        const int pos0 = tid + ((p.shift && !HALF) ? M : 0), pos1 = tid + (p.shift ? 0 : M);
        const float sc = MODE == 1 ? 0.25f * p.scale : 0.5f * p.scale;
        const cf wn = p.tw_n[tid];
        const int km0 = (M - tid) & (M - 1);  // partner of m = 0; of m: km0 - T m (mod M)
#pragma unroll
        for (int g = 0; g < 8; ++g) {  // batches of 4
            cf z[4], zm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = 4 * g + q, k = tid + T * m, km = (km0 - T * m) & (M - 1);
                z[q] = L[k + (k >> 5)];
                zm[q] = L[km + (km >> 5)];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = 4 * g + q;
                const cf A = mk<float>(z[q].re + zm[q].re, z[q].im - zm[q].im), B = mk<float>(z[q].re - zm[q].re, z[q].im + zm[q].im);  // 2A, 2B
                const cf w = m == 0 ? wn : cmul(wn, mk<float>(C64[m], S64[m]));  // W_N^k = W_N^tid W_64^m
                const cf wb = cmul(w, B);
                const cf t = mk<float>(-wb.im, wb.re);  // i W B
                cf x0 = mk<float>(A.re - t.re, A.im - t.im), x1 = mk<float>(A.re + t.re, A.im + t.im);  // 2 X[k], 2 X[k + M]
                if (MODE == 1) {
                    float* o = reinterpret_cast<float*>(p.out) + orow;
                    const float f = (HALF && p.realdim2 && (m != 0 || tid != 0)) ? 2.0f * sc : sc;
                    fastr_store4(o + pos0 + T * m, (x0.re * x0.re + x0.im * x0.im) * f);
                    if (!HALF) fastr_store4(o + pos1 + T * m, (x1.re * x1.re + x1.im * x1.im) * sc);
                    else if (m == 0) { if (tid == 0) fastr_store4(o + M, (x1.re * x1.re + x1.im * x1.im) * sc); }
                } else {
                    x0 = cscale(x0, sc); x1 = cscale(x1, sc);
                    cf* o = reinterpret_cast<cf*>(p.out) + orow;
                    if (p.ph_on) {
                        const int k = tid + T * m;
                        x0 = cmul(x0, p.ph[k]);
                        if (!HALF) x1 = cmul(x1, p.ph[k + M]);
                    }
                    fastr_store8(o + pos0 + T * m, x0);
                    if (!HALF) fastr_store8(o + pos1 + T * m, x1);
                    else if (m == 0) { if (tid == 0) fastr_store8(o + M, p.ph_on ? cmul(x1, p.ph[M]) : x1); }
                }
            }
            fastr_sched_fence();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// COMPLEX rows of M = 2048 | 4096 | 8192 | 16384 points in ONE pass (xrft.ifft / xrft.fft of complex data along the contiguous axis, xrft.py:586-621,
// :439-447): the transform core of fastr2_kernel on the row itself -- no packing in front, no split behind.  T = M / 32 threads hold the row
// (32 complex values each), three stages through two LDS exchanges, Z in natural order through the LDS, 8-byte lane-contiguous stores.
//   load    z[n] at n = tid + T j; an fftshifted input is the register rotation j -> j + 16 (n + M/2 = tid + T (j + 16)); the true-phase
//           factor of an inverse transform multiplies the SOURCE sample (ph_in); the inverse conjugates on the way in and on the way out
//   store   Z[k] * scale (times the output phase table, unshifted k) at (k + shift) mod M: k = v + T m -> the rotation m -> m + 16
// 16 bytes per point through memory (8 read + 8 written); the LDS-resident fastm_xonly_kernel it replaces on these lengths ran them at
// 2.4 TB/s (ifft (16384, 4096): 149 GFFT/s, profiles/r05_inverse.txt).   MODE 0: complex result, 1: |Z|^2 (power spectrum of complex rows)
// ------------------------------------------------------------------------------------------------------------------------------------------
template <int R2, int R3, int MODE>
__global__ void __launch_bounds__((R2Geom<R2, R3>::T), (R2Geom<R2, R3>::WPS)) fastc_kernel(FastR p) {
    typedef R2Geom<R2, R3> G;
    constexpr int T = G::T, M = G::M, K2 = G::K2, K3 = G::K3, S1 = G::S1, S2 = G::S2;
    XRFT_DYN_SMEM(smem_raw);
    cf* L = reinterpret_cast<cf*>(smem_raw);
    fastr_stagger(p.stagger);  // (a resident set walking the rows: XRFTHIP_FASTR_STAGGER / XRFTHIP_FASTR_GRID)
    for (long long row = blockIdx.x; row < p.nrows; row += gridDim.x) {
        int tid = threadIdx.x;
        XRFT_OPAQUE(tid);
        const cf* __restrict__ src = reinterpret_cast<const cf*>(p.in) + (size_t)row * M + tid;
        const int rot = p.ishift ? 16 : 0;
        cf a[32];
        if (p.ishift) {
#pragma unroll
            for (int j = 0; j < 32; ++j) a[j] = src[((j + 16) & 31) * T];
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) a[j] = src[j * T];
        }
        if (p.ph_in) {  // two batches of 16 factors beside the 64 registers of the row
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                cf f[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) f[j] = p.ph[tid + T * ((16 * g + j + rot) & 31)];
#pragma unroll
                for (int j = 0; j < 16; ++j) a[16 * g + j] = cmul(a[16 * g + j], f[j]);
                fastr_sched_fence();
            }
        }
        if (p.win) {  // a real window over the M samples
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float w[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = p.win[tid + T * (16 * g + j)];  // (indexed by the transform's sample index, as fastm_xonly_kernel)
#pragma unroll
                for (int j = 0; j < 16; ++j) a[16 * g + j] = cscale(a[16 * g + j], w[j]);
                fastr_sched_fence();
            }
        }
        if (p.inv) {
#pragma unroll
            for (int j = 0; j < 32; ++j) a[j].im = -a[j].im;
        }
        // ---- stage 1
        dft32f(a);
        twiddle32f(a, p.tw_m[tid]);
        // ---- exchange 1
        const int aa = tid % R3, cc = tid / R3;
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) L[k1 * S1 + tid] = a[k1];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < K2; ++s)
#pragma unroll
            for (int b = 0; b < R2; ++b) a[s * R2 + b] = L[(cc + R2 * s) * S1 + aa + R3 * b];
        // ---- stage 2: DFT_R2 over b, x W_T^(a k2)
        {
            const cf wt = p.tw_s[aa];
            if (R2 == 32) { dft32f(a); twiddle32f(a, wt); }
            else if (R2 == 16) {
#pragma unroll
                for (int s = 0; s < K2; ++s) { dft16<float>(a + 16 * s); twiddle16<float>(a + 16 * s, wt); }
            } else {
                const cf w2 = cmul(wt, wt), w3 = cmul(w2, wt), w4 = cmul(w2, w2), w5 = cmul(w4, wt), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
#pragma unroll
                for (int s = 0; s < K2; ++s) {
                    cf* g = a + 8 * s;
                    dft8<float>(g);
                    g[1] = cmul(g[1], wt); g[2] = cmul(g[2], w2); g[3] = cmul(g[3], w3); g[4] = cmul(g[4], w4);
                    g[5] = cmul(g[5], w5); g[6] = cmul(g[6], w6); g[7] = cmul(g[7], w7);
                }
            }
        }
        // ---- exchange 2
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 32; ++u) L[u * S2 + tid] = a[u];
        __syncthreads();
        const int dd = tid % R3;
#pragma unroll
        for (int t = 0; t < K3; ++t)
#pragma unroll
            for (int e = 0; e < R3; ++e) a[t * R3 + e] = L[(dd + R3 * t) * S2 + e + R3 * cc];
        // ---- stage 3
#pragma unroll
        for (int t = 0; t < K3; ++t) dft_r<float, R3>(a + R3 * t);
        // ---- Z in natural order: register (t, k3) is k = k1 + 32 (k2 + R2 k3), u = d + R3 t = s R2 + k2, k1 = c + R2 s
        __syncthreads();
#pragma unroll
        for (int t = 0; t < K3; ++t) {
            const int u = dd + R3 * t, s2 = u / R2, k2 = u % R2, k1 = cc + R2 * s2;
#pragma unroll
            for (int k3 = 0; k3 < R3; ++k3) {
                const int k = k1 + 32 * (k2 + R2 * k3);
                L[k + (k >> 5)] = a[t * R3 + k3];
            }
        }
        __syncthreads();
        // ---- store: thread v takes k = v + T m at (k + shift) mod M = v + T ((m + 16) mod 32)
        const size_t orow = (size_t)row * M;
        const int orot = p.shift ? 16 : 0;
        const float sc = p.scale;
#pragma unroll
        for (int g = 0; g < 8; ++g) {  // batches of 4
            cf z[4], f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = tid + T * (4 * g + q);
                z[q] = L[k + (k >> 5)];
                if (MODE == 0 && p.ph_on) f[q] = p.ph[k];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = 4 * g + q, pos = tid + T * ((m + orot) & 31);
                if (MODE == 1) {
                    fastr_store4(reinterpret_cast<float*>(p.out) + orow + pos, (z[q].re * z[q].re + z[q].im * z[q].im) * sc);
                } else {
                    cf o = cscale(z[q], sc);
                    if (p.inv) o.im = -o.im;
                    if (p.ph_on) o = cmul(o, f[q]);
                    fastr_store8(reinterpret_cast<cf*>(p.out) + orow + pos, o);
                }
            }
            fastr_sched_fence();
        }
    }
}

}  // namespace xrft
